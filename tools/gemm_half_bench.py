"""192x192 one-workgroup-per-CU NT GEMM (variant 2) against the 192x96 two-per-CU instantiation (variants 5/6/7 =
no / dispatch-order / interleaved stagger) on the training step's shapes.  (GPU box)
usage: python tools/gemm_half_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402
from tools.gemm_bench import bench_nt  # noqa: E402

if __name__ == "__main__":
    M, D = 12288, 768
    names = {hip.EPI_BF16: "bf16", hip.EPI_BF16_GELU: "gelu", hip.EPI_F32: "f32res", hip.EPI_BF16_DGELU: "dgelu"}
    cases = [("qkv fwd", M, 3 * D, D, hip.EPI_BF16), ("proj fwd", M, D, D, hip.EPI_F32), ("fc1 fwd", M, 4 * D, D, hip.EPI_BF16_GELU),
             ("fc2 fwd", M, D, 4 * D, hip.EPI_F32), ("fc2 dgrad", M, 4 * D, D, hip.EPI_BF16_DGELU), ("fc1 dgrad", M, D, 4 * D, hip.EPI_BF16),
             ("proj dgrad", M, D, D, hip.EPI_BF16), ("qkv dgrad", M, D, 3 * D, hip.EPI_BF16)]
    variants = (2, 5)
    tot = {v: 0.0 for v in variants}
    for name, m, n, k, epi in cases:
        row = f"{name:11s} N={n:5d} K={k:5d} {names[epi]:7s}"
        for rnd in range(2):
            for v in variants:
                us = bench_nt(m, n, k, epi, v, reps=30)
                if rnd == 1:
                    tot[v] += us
                    row += f" | v{v}: {us:6.1f} us {2.0 * m * n * k / us / 1e6:6.1f} TF"
        print(row, flush=True)
    print("sum per block: " + ", ".join(f"v{v} {tot[v]:.0f} us" for v in variants))
    hip.GEMM_TILE = 0
