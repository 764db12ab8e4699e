"""How much of the NT192 GEMM's distance from the MFMA peak is operand-data dependent (clock/power management)
rather than kernel structure?  Same launches, operands = zeros / constant / random.  (GPU box)
usage: python tools/gemm_data_effect.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402


def run(M, N, K, fill, reps=200):
    if fill == "zeros":
        A = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
        B = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
    elif fill == "ones":
        A = torch.ones(M, K, device="cuda", dtype=torch.bfloat16)
        B = torch.ones(N, K, device="cuda", dtype=torch.bfloat16)
    else:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    hip.GEMM_TILE = 2
    for _ in range(5):
        hip.gemm_nt(A, B, hip.EPI_BF16, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.gemm_nt(A, B, hip.EPI_BF16, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    for M, N, K in [(12288, 768, 3072), (12288, 2304, 768), (12288, 768, 12288)]:
        for rnd in range(2):
            for fill in ("zeros", "ones", "randn"):
                us = run(M, N, K, fill)
                if rnd:
                    print(f"M={M} N={N} K={K} {fill:6s}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
