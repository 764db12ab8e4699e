"""Known-good reference for the NT GEMM shapes of one transformer block: torch.matmul (hipBLASLt / rocBLAS behind it) vs this build's
gemm_nt192 at the same shapes and data, bf16 in / bf16 out, interleaved in one process (the reference is a yardstick, not a
dependency: the product does not link it)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

M, D = 12288, 768
shapes = [("qkv fwd", M, 3 * D, D), ("proj fwd/dgrad", M, D, D), ("fc1 fwd (plain)", M, 4 * D, D), ("fc2 fwd / fc1 dgrad", M, D, 4 * D),
          ("qkv dgrad", M, D, 3 * D), ("fc2 dgrad (plain)", M, 4 * D, D)]


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


# two passes over the list: the first launches of a process run several per cent slow for every kernel, the library's included (clocks, caches), so
# the second pass is the one to read; `rows` = this build's kernel forced to the row-major tile list (vtGemmNT.tile = 19), the order of rounds 1-4
for pass_, (name, m, n, k) in [(p, sh) for p in (1, 2) for sh in shapes]:
    A = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    B = (torch.randn(n, k, device="cuda") * 0.03).to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ref = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    res = []
    for rnd in range(3):
        t_ours = timeit(lambda: hip.gemm_nt(A, B, hip.EPI_BF16, out=out))
        t_rows = timeit(lambda: hip.gemm_nt(A, B, hip.EPI_BF16, out=out, tile=19))
        t_lib = timeit(lambda: torch.matmul(A, B.t(), out=ref))
        res.append((t_ours, t_lib, t_rows))
    to, tl, tr = min(r[0] for r in res), min(r[1] for r in res), min(r[2] for r in res)
    f = 2.0 * m * n * k
    err = float((out.float() - ref.float()).abs().max())
    print(f"pass {pass_} {name:22s} M{m} N{n} K{k}: rows {tr:6.1f}  ours {to:6.1f} us ({f / to / 1e6:6.0f} TF/s)   hipBLASLt {tl:6.1f} us ({f / tl / 1e6:6.0f} TF/s)   ratio {to / tl:.2f}   max|diff| {err:.3f}")
