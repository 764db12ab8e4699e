"""Stand-alone attention fwd/bwd at the training step's shape (for rocprofv3 --pmc and timing)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

B, L, H = 8, 1536, 12
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
dO = torch.randn(B * L, H * 64, device="cuda").to(torch.bfloat16)
for _ in range(2):
    o, lse = hip.attention_fwd(qkv, B, L, H)
    d = hip.attention_bwd(qkv, o, dO, lse, B, L, H)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
for _ in range(reps):
    o, lse = hip.attention_fwd(qkv, B, L, H)
e[1].record()
for _ in range(reps):
    d = hip.attention_bwd(qkv, o, dO, lse, B, L, H)
e[2].record()
torch.cuda.synchronize()
f = 4.0 * B * H * L * L * 64
print(f"fwd {e[0].elapsed_time(e[1]) / reps * 1e3:.1f} us ({f / (e[0].elapsed_time(e[1]) / reps) / 1e9:.0f} TF/s)  "
      f"bwd(delta+dq+dkv) {e[1].elapsed_time(e[2]) / reps * 1e3:.1f} us ({3.5 * f / (e[1].elapsed_time(e[2]) / reps) / 1e9:.0f} TF/s)")
