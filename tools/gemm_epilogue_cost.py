"""Prologue + epilogue cost of the NT192 GEMM: time vs K at fixed M, N (the K -> 0 intercept is what one tile pays
outside its main loop; the slope is the main loop's time per 64-wide K-tile).  (GPU box)
usage: python tools/gemm_epilogue_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402
from tools.gemm_bench import bench_nt  # noqa: E402

if __name__ == "__main__":
    M = 12288
    names = {hip.EPI_BF16: "bf16", hip.EPI_BF16_GELU: "gelu", hip.EPI_F32: "f32res", hip.EPI_BF16_DGELU: "dgelu"}
    for N in (768, 2304, 3072):
        for epi in (hip.EPI_BF16, hip.EPI_F32, hip.EPI_BF16_GELU, hip.EPI_BF16_DGELU):
            row = f"N={N:5d} {names[epi]:7s} rounds={N // 768}:"
            ts = {}
            for K in (64, 192, 384, 768, 1536):
                bench_nt(M, N, K, epi, 2, reps=5)
                ts[K] = bench_nt(M, N, K, epi, 2, reps=50)
                row += f"  K={K}: {ts[K]:6.1f}"
            slope = (ts[1536] - ts[768]) / 12.0
            row += f"  | us/K-tile/round {slope / (N // 768):5.2f}, intercept/round {(ts[768] - 12 * slope) / (N // 768):5.2f} us"
            print(row, flush=True)
