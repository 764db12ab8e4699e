"""Interleaved A/B in ONE process of the two attention backward forms at the training step's shapes (random data):
split = attn_bwd_dq + attn_bwd_dkv (7 products), fused = attn_delta + attn_bwd_fused (5 products, ordered dQ hand-off).
Usage: python tools/attn_bwd_ab.py [rounds] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
H = 12
for (B, L, q_begin) in [(8, 1536, 0), (8, 1536, 1024), (8, 1536, 512), (1, 1536, 0), (8, 2048, 0), (2, 5120, 0)]:
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
    dO = torch.randn(B * (L - q_begin), H * 64, device="cuda").to(torch.bfloat16)
    o, lse = hip.attention_fwd(qkv, B, L, H, q_begin=q_begin)
    out = {}
    for fused in (False, True):
        for _ in range(2):
            out[fused] = hip.attention_bwd(qkv, o, dO, lse, B, L, H, q_begin=q_begin, fused=fused)
    st = hip.attention_bwd_fused_status(hip.attention_bwd.last_ws)
    diff = (out[True].float() - out[False].float()).abs().max().item() / out[False].float().abs().max().item()
    times = {False: [], True: []}
    for r in range(rounds):
        for fused in (False, True):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                hip.attention_bwd(qkv, o, dO, lse, B, L, H, dqkv=out[fused], q_begin=q_begin, fused=fused)
            e1.record()
            torch.cuda.synchronize()
            times[fused].append(e0.elapsed_time(e1) / reps * 1e3)
    f5 = 10.0 * B * H * (L - q_begin) * L * 64
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    print(f"B={B} L={L} q_begin={q_begin}: split {med[False]:.1f} us (min {min(times[False]):.1f})  fused {med[True]:.1f} us (min {min(times[True]):.1f}) "
          f"= {f5 / med[True] / 1e6:.0f} TF/s of 5 products; ratio {med[True] / med[False]:.3f}; max diff {diff:.2e}; status {st}", flush=True)
