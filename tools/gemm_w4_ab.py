"""Four-wave prototype of the 192x192x64 NT GEMM (vtGemmNT.tile = 20: one wave per SIMD, 96 x 96 outputs per wave, accumulators pinned in AGPRs) against
the eight-wave default (tile = 2): bit-equality of the plain bf16 epilogue, then interleaved timings; the K = 768 -> 3072 difference at one shape gives each
kernel's cost per K-tile with the (prototype) epilogue taken out."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

torch.manual_seed(2)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


bad = 0
for M, N, K in ((192, 192, 64), (384, 1536, 768), (3840, 3072, 768), (12288, 768, 3072), (12288, 2304, 768)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    for b in (None, bias):
        kw = dict(bias=b) if b is not None else {}
        old = hip.gemm_nt(A, B, hip.EPI_BF16, tile=2, **kw).clone()
        new = hip.gemm_nt(A, B, hip.EPI_BF16, tile=20, **kw)
        torch.cuda.synchronize()
        ok = torch.equal(old, new)
        bad += 0 if ok else 1
        print(f"M{M} N{N} K{K} bias={b is not None}: {'equal' if ok else 'DIFFERENT max|diff| %g' % float((old.float() - new.float()).abs().max())}", flush=True)
print("bit-equality:", "all equal" if bad == 0 else f"{bad} cases differ")
M = 12288
res = {}
for name, N, K in (("qkv fwd", 2304, 768), ("fc1 plain", 3072, 768), ("proj", 768, 768), ("qkv dgrad", 768, 2304), ("fc2 fwd / fc1 dgrad", 768, 3072), ("N3072 K3072 (slope)", 3072, 3072)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    rows = []
    for rnd in range(3):
        rows.append((timeit(lambda: hip.gemm_nt(A, B, hip.EPI_BF16, out=out, tile=2)), timeit(lambda: hip.gemm_nt(A, B, hip.EPI_BF16, out=out, tile=20))))
    o, n = min(r[0] for r in rows), min(r[1] for r in rows)
    f = 2.0 * M * N * K
    res[name] = (o, n)
    print(f"{name:22s} M{M} N{N} K{K}: eight waves {o:6.1f} us ({f / o / 1e6:5.0f} TF/s)   four waves {n:6.1f} us ({f / n / 1e6:5.0f} TF/s)   rounds " + " ".join(f"{a:.1f}/{b:.1f}" for a, b in rows), flush=True)
o1, n1 = res["fc1 plain"]
o4, n4 = res["N3072 K3072 (slope)"]
print(f"per K-tile and round of 256 tiles (N = 3072: 4 rounds, 36 more K-tiles each): eight waves {(o4 - o1) / 4 / 36:.3f} us, four waves {(n4 - n1) / 4 / 36:.3f} us")
sys.exit(1 if bad else 0)
