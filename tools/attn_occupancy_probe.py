"""Do co-resident workgroups of the attention kernels overlap?  Same per-workgroup work (L = 1536), grids of ~1, 2, 3, 4
workgroups per CU: perfect overlap keeps the time flat, none makes it proportional.  (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

L, H = 1536, 1
for BH in (21, 42, 64, 85, 96):
    B = BH
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
    dO = torch.randn(B * L, H * 64, device="cuda").to(torch.bfloat16)
    o, lse = hip.attention_fwd(qkv, B, L, H)
    hip.attention_bwd(qkv, o, dO, lse, B, L, H)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(20):
        hip.attention_fwd(qkv, B, L, H)
    ev[1].record()
    for _ in range(20):
        hip.attention_bwd(qkv, o, dO, lse, B, L, H)
    ev[2].record()
    torch.cuda.synchronize()
    print(f"workgroups {12 * BH:5d} ({12 * BH / 256:.2f} per CU): fwd {ev[0].elapsed_time(ev[1]) / 20 * 1e3:7.1f} us   bwd (dq+dkv) {ev[1].elapsed_time(ev[2]) / 20 * 1e3:7.1f} us", flush=True)
