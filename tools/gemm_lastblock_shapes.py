"""Tile generations on the compact last-block shapes (M = 4096 / 8192 kept rows, width 768).  (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402
from tools.gemm_bench import bench_nt  # noqa: E402

D = 768
for M in (4096, 8192):
    for name, n, k, epi in (("proj", D, D, hip.EPI_F32), ("fc1", 4 * D, D, hip.EPI_BF16_GELU), ("fc2", D, 4 * D, hip.EPI_F32),
                            ("fc2 dgrad", 4 * D, D, hip.EPI_BF16_DGELU), ("fc1 dgrad", D, 4 * D, hip.EPI_BF16), ("proj dgrad", D, D, hip.EPI_BF16)):
        row = f"M={M:6d} {name:10s} N={n:5d} K={k:5d}"
        for v in (1, 2):
            bench_nt(M, n, k, epi, v, reps=10)
            us = bench_nt(M, n, k, epi, v, reps=40)
            row += f" | v{v}: {us:6.1f} us {2.0 * M * n * k / us / 1e6:6.1f} TF"
        t192 = -(-M // 192) * -(-n // 192)
        t128 = -(-M // 128) * -(-n // 128)
        row += f" | tiles192 {t192} tiles128 {t128}"
        print(row, flush=True)
hip.GEMM_TILE = 0
