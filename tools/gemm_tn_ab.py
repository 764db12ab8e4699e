"""Grouped weight-gradient GEMM (gemm_tn192*), the step's 4-block launch: 16 problems = 768 tiles of 192x192 = 3 rounds of the 256 CUs,
every problem on its own operands (1.2 GB per launch, as in the step: nothing stays in L2 / MALL between launches).
Interleaved A/B in one process: tile 7 = round-1 kernel (fragments read behind the barrier), tile 2 = register-pipelined (round 4).
usage: python tools/gemm_tn_ab.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
M, D = 12288, 768
wg = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]
flops = 0.0
probs = []
for blk in range(4):
    for P, Q in wg:
        probs.append(dict(A=torch.randn(M, P, device="cuda").to(torch.bfloat16), B=torch.randn(M, Q, device="cuda").to(torch.bfloat16),
                          out=torch.empty(P, Q, device="cuda")))
        flops += 2.0 * M * P * Q


def run(tile, n):
    for pr in probs:
        pr["tile"] = tile
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        hip.gemm_tn_grouped(probs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for t in (7, 2):
    run(t, 2)
ref = [pr["out"].clone() for pr in probs]       # tile 2 ran last
run(7, 1)
same = all(torch.equal(r, pr["out"]) for r, pr in zip(ref, probs))
print(f"pipelined == burst kernel bit for bit: {same}")
for rnd in range(4):
    a, b = run(7, reps), run(2, reps)
    print(f"round {rnd}: burst {a:7.1f} us = {flops / a / 1e6:6.0f} TF/s | pipelined {b:7.1f} us = {flops / b / 1e6:6.0f} TF/s | ratio {b / a:.3f}", flush=True)
# one block's group (192 tiles) and the 3-block group for reference
for nb in (1, 2, 3):
    sub = probs[: 4 * nb]
    f = sum(2.0 * M * p["A"].shape[1] * p["B"].shape[1] for p in sub)
    for tile in (7, 2):
        for pr in sub:
            pr["tile"] = tile
        hip.gemm_tn_grouped(sub)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hip.gemm_tn_grouped(sub)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"{nb} block(s), tile {tile}: {us:7.1f} us = {f / us / 1e6:6.0f} TF/s")
