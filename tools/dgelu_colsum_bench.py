"""fc2-dgrad GEMM (x gelu') + fc1 bias gradient: separate column-sum pass vs the sums taken in the epilogue.  (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

M, N, K = 12288, 3072, 768
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
u = torch.randn(M, N, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
part = torch.empty(M // 192, N, device="cuda")
bias = torch.empty(N, device="cuda")


def separate():
    hip.gemm_nt(A, B, hip.EPI_BF16_DGELU, aux=u, out=out)
    return hip.colsum(out)


def fused():
    hip.gemm_nt(A, B, hip.EPI_BF16_DGELU, aux=u, out=out, colsum_partial=part)
    hip.check(hip.lib().vt_sum_slabs(hip.ptr(part), M // 192, N, N, hip.ptr(bias), hip.stream()), "vt_sum_slabs")
    return bias


for rnd in range(3):
    for name, fn in (("separate", separate), ("fused", fused)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:9s}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us", flush=True)
a, b = separate().clone(), fused().clone()
print("max abs diff of the bias gradient:", float((a - b).abs().max()), "of", float(a.abs().max()))
