"""Attention kernels on random vs constant vs zero operands (same instruction stream, different switching activity): how much of the
kernel time is the chip's power / clock management rather than its instruction schedule.  usage: python tools/attn_data_effect.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

B, L, H = 8, 1536, 12
reps = 20


def run(name, qkv, dO):
    for _ in range(3):
        o, lse = hip.attention_fwd(qkv, B, L, H)
        hip.attention_bwd(qkv, o, dO, lse, B, L, H)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(reps):
        o, lse = hip.attention_fwd(qkv, B, L, H)
    e[1].record()
    for _ in range(reps):
        hip.attention_bwd(qkv, o, dO, lse, B, L, H)
    e[2].record()
    torch.cuda.synchronize()
    print(f"{name:28s} fwd {e[0].elapsed_time(e[1]) / reps * 1e3:6.1f} us   bwd {e[1].elapsed_time(e[2]) / reps * 1e3:6.1f} us", flush=True)


g = torch.Generator(device="cuda").manual_seed(0)
rq = torch.randn(B * L, 3 * H * 64, device="cuda", generator=g).to(torch.bfloat16)
rd = torch.randn(B * L, H * 64, device="cuda", generator=g).to(torch.bfloat16)
for rnd in range(2):
    run("random normal", rq, rd)
    run("zeros", torch.zeros_like(rq), torch.zeros_like(rd))
    run("constant 0.5", torch.full_like(rq, 0.5), torch.full_like(rd, 0.5))
    run("random, small (x 0.01)", rq * 0.01, rd * 0.01)
