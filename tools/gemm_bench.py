"""Micro-benchmark of the GEMM tile generations on the training step's shapes (GPU box).
usage: python tools/gemm_bench.py  -> prints us and TFLOP/s per shape and variant (interleaved rounds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402


def bench_nt(M, N, K, epi, variant, reps=20):
    hip.GEMM_TILE = variant
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    kw = {}
    if epi == hip.EPI_F32:
        kw = dict(bias=torch.randn(N, device="cuda"), residual=torch.randn(M, N, device="cuda"))
        out = torch.empty(M, N, device="cuda")
    elif epi == hip.EPI_BF16_GELU:
        kw = dict(bias=torch.randn(N, device="cuda"), out2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    elif epi == hip.EPI_BF16_DGELU:
        kw = dict(aux=torch.randn(M, N, device="cuda").to(torch.bfloat16))
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        hip.gemm_nt(A, B, epi, out=out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.gemm_nt(A, B, epi, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bench_tn(M, shapes, variant, reps=10):
    hip.GEMM_TILE = variant
    probs = []
    for P, Q in shapes:
        probs.append(dict(A=torch.randn(M, P, device="cuda").to(torch.bfloat16), B=torch.randn(M, Q, device="cuda").to(torch.bfloat16),
                          out=torch.empty(P, Q, device="cuda")))
    for _ in range(2):
        hip.gemm_tn_grouped(probs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.gemm_tn_grouped(probs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    M, D = 12288, 768
    names = {hip.EPI_BF16: "bf16", hip.EPI_BF16_GELU: "gelu", hip.EPI_F32: "f32res", hip.EPI_BF16_DGELU: "dgelu"}
    cases = [("qkv fwd", M, 3 * D, D, hip.EPI_BF16), ("proj fwd", M, D, D, hip.EPI_F32), ("fc1 fwd", M, 4 * D, D, hip.EPI_BF16_GELU),
             ("fc2 fwd", M, D, 4 * D, hip.EPI_F32), ("fc2 dgrad", M, 4 * D, D, hip.EPI_BF16_DGELU), ("fc1 dgrad", M, D, 4 * D, hip.EPI_BF16),
             ("proj dgrad", M, D, D, hip.EPI_BF16), ("qkv dgrad", M, D, 3 * D, hip.EPI_BF16)]
    tot = {1: 0.0, 2: 0.0}
    for name, m, n, k, epi in cases:
        row = f"{name:11s} M={m} N={n:5d} K={k:5d} {names[epi]:7s}"
        for rnd in range(2):
            for v in (1, 2):
                us = bench_nt(m, n, k, epi, v)
                if rnd == 1:
                    tot[v] += us
                    row += f" | v{v}: {us:7.1f} us {2.0 * m * n * k / us / 1e6:7.1f} TF/s"
        print(row, flush=True)
    print(f"sum per block: v1 {tot[1]:.0f} us, v2 {tot[2]:.0f} us")
    wg = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]
    f = sum(2.0 * M * p * q for p, q in wg)
    for v in (1, 2):
        us = bench_tn(M, wg, v)
        print(f"wgrad group (4 problems) v{v}: {us:7.1f} us {f / us / 1e6:7.1f} TF/s", flush=True)
    # 4 blocks' weight gradients in one launch (768 tiles of 192x192 = 3 full rounds)
    wg4 = wg * 2
    us = bench_tn(M, wg4, 2)
    print(f"wgrad group (8 problems = 2 blocks) v2: {us:7.1f} us {2 * f / us / 1e6:7.1f} TF/s")
    hip.GEMM_TILE = 0
