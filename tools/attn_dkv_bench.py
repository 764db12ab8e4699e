"""dK/dV kernel timing in isolation via rocprof-free event timing of attention_bwd minus... (ablation helper): prints bwd total."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

B, L, H = 8, 1536, 12
qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
dO = torch.randn(B * L, H * 64, device="cuda").to(torch.bfloat16)
o, lse = hip.attention_fwd(qkv, B, L, H)
for _ in range(3):
    hip.attention_bwd(qkv, o, dO, lse, B, L, H)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    hip.attention_bwd(qkv, o, dO, lse, B, L, H)
e1.record()
torch.cuda.synchronize()
print(f"bwd (dq + dkv) {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
