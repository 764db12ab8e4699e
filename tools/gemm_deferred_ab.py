"""The K = 768 NT GEMM with the epilogue inside the next tile's K loop (vtGemmNT.tile = 19, gemm_nt192d_kernel) against the kernel that
runs it behind each tile (tile = 2): bit-equality of every output (and of u / g / column sums) on whole-round, uneven and one-tile grids,
then interleaved timings at the step's shapes on random data, with torch.matmul (hipBLASLt) beside the plain epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

torch.manual_seed(1)
K = 768


def run(epi, tile, A, B, bias=None, aux=None, cs=None):
    kw = {}
    if bias is not None:
        kw["bias"] = bias
    if aux is not None:
        kw["aux"] = aux
    if cs is not None:
        kw["colsum_partial"] = cs
    return hip.gemm_nt(A, B, epi, tile=tile, **kw)


def check():
    bad = 0
    for M, N in ((12288, 2304), (12288, 3072), (3840, 3072), (1536, 768), (192, 192), (384, 1536)):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        u = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        for epi, name in ((hip.EPI_BF16, "bf16"), (hip.EPI_BF16_GELU, "gelu"), (hip.EPI_BF16_DGELU, "dgelu")):
            for with_bias in ((False, True) if epi != hip.EPI_BF16_DGELU else (False,)):
                b = bias if with_bias else None
                aux = u if epi == hip.EPI_BF16_DGELU else None
                cs_old = torch.full(((M + 191) // 192, N), float("nan"), device="cuda") if epi == hip.EPI_BF16_DGELU else None
                cs_new = cs_old.clone() if cs_old is not None else None
                cs_one = cs_old.clone() if cs_old is not None else None
                old = run(epi, 2, A, B, b, aux, None)
                old = tuple(t.clone() for t in old) if isinstance(old, tuple) else (old.clone(),)
                new = run(epi, 19, A, B, b, aux, cs_new)
                new = tuple(t.clone() for t in new) if isinstance(new, tuple) else (new.clone(),)
                torch.cuda.synchronize()
                ok = all(torch.equal(x, y) for x, y in zip(old, new))
                msg = ""
                if cs_new is not None:
                    want = torch.stack([new[0][t * 192:(t + 1) * 192].float().sum(0) for t in range((M + 191) // 192)])
                    err = float((cs_new - want).abs().max() / (want.abs().max() + 1e-9))
                    one = run(epi, 2, A, B, b, aux, cs_one)           # the kernel it would replace: same outputs, its own order of the column sums
                    torch.cuda.synchronize()
                    e2 = float((cs_one - cs_new).abs().max() / (want.abs().max() + 1e-9))
                    ok = ok and err < 1e-5 and torch.equal(one, new[0]) and e2 < 1e-5
                    msg = f" colsum rel err {err:.1e} (vs the other kernel's sums {e2:.1e})"
                if not ok:
                    bad += 1
                    d = [float((x.float() - y.float()).abs().max()) for x, y in zip(old, new)]
                    msg += f" MISMATCH max|diff| {d}"
                print(f"M{M} N{N} {name:5s} bias={with_bias}: {'equal' if ok else 'DIFFERENT'}{msg}", flush=True)
    return bad


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


def bench():
    M = 12288
    cases = (("qkv fwd (bf16)", 2304, hip.EPI_BF16), ("fc1 fwd plain (bf16)", 3072, hip.EPI_BF16), ("fc1 fwd (GELU)", 3072, hip.EPI_BF16_GELU),
             ("fc2 dgrad (gelu')", 3072, hip.EPI_BF16_DGELU))
    if os.environ.get("VT_BENCH_CASES"):
        cases = tuple(c for i, c in enumerate(cases) if str(i) in os.environ["VT_BENCH_CASES"])
    for name, N, epi in cases:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda") if epi == hip.EPI_BF16_GELU else None
        u = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi == hip.EPI_BF16_DGELU else None
        cs = torch.empty(M // 192, N, device="cuda") if epi == hip.EPI_BF16_DGELU else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == hip.EPI_BF16_GELU else None
        ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        kw = dict(out=out)
        if out2 is not None:
            kw["out2"] = out2
        if bias is not None:
            kw["bias"] = bias
        if u is not None:
            kw["aux"] = u
            kw["colsum_partial"] = cs
        rows = []
        for rnd in range(3):
            t_old = timeit(lambda: hip.gemm_nt(A, B, epi, tile=2, **kw))
            t_new = timeit(lambda: hip.gemm_nt(A, B, epi, tile=19, **kw))
            t_lib = timeit(lambda: torch.matmul(A, B.t(), out=ref)) if epi == hip.EPI_BF16 else float("nan")
            rows.append((t_old, t_new, t_lib))
        f = 2.0 * M * N * K
        o, n, l = (min(r[i] for r in rows) for i in range(3))
        print(f"{name:22s} M{M} N{N} K{K}: behind each tile {o:6.1f} us ({f / o / 1e6:5.0f} TF/s)   inside the next K loop {n:6.1f} us ({f / n / 1e6:5.0f} TF/s)"
              f"   hipBLASLt {l:6.1f} us   rounds " + " ".join(f"{a:.1f}/{b:.1f}" for a, b, _ in rows), flush=True)


if __name__ == "__main__":
    bad = 0 if os.environ.get("VT_BENCH_ONLY") else check()
    print("bit-equality:", "all equal" if bad == 0 else f"{bad} cases differ")
    bench()
    sys.exit(1 if bad else 0)
