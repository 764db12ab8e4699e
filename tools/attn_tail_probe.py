"""How much do the partly filled last rounds of the attention kernels cost?  Same per-workgroup work (L = 1536, 12 key / query blocks per head),
head count varied so that the number of workgroups is a whole or a fractional number of resident rounds (forward 1024, dQ 768, dK/dV 512 slots).
Prints time per launch and per workgroup-item.  python tools/attn_tail_probe.py (GPU)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

L, H = 1536, 4
for B in (16, 21, 22, 24, 26, 28, 32):        # x 4 heads x 12 blocks = 768, 1008, 1056, 1152 (the step), 1248, 1344, 1536 workgroups
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
    dO = torch.randn(B * L, H * 64, device="cuda").to(torch.bfloat16)
    for _ in range(2):
        o, lse = hip.attention_fwd(qkv, B, L, H)
        d = hip.attention_bwd(qkv, o, dO, lse, B, L, H)
    torch.cuda.synchronize()
    reps = 10
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(reps):
        o, lse = hip.attention_fwd(qkv, B, L, H)
    e[1].record()
    for _ in range(reps):
        d = hip.attention_bwd(qkv, o, dO, lse, B, L, H)
    e[2].record()
    torch.cuda.synchronize()
    n = B * H * 12
    tf, tb = e[0].elapsed_time(e[1]) / reps * 1e3, e[1].elapsed_time(e[2]) / reps * 1e3
    print(f"{n:5d} workgroups (fwd {n / 1024:.2f} / dQ {n / 768:.2f} / dKdV {n / 512:.2f} rounds): fwd {tf:6.1f} us = {tf / n * 1e3:5.1f} ns/item   "
          f"bwd {tb:6.1f} us = {tb / n * 1e3:5.1f} ns/item")
