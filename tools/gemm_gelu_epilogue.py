"""The two GELU epilogues of the 192x192 NT kernel at the step's shapes against the plain bf16 epilogue on the same flops:
fc1 forward (u and gelu(u) out) and fc2 dgrad (dy W2 * gelu'(u), + column sums), interleaved in one process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

M, D = 12288, 768
A = torch.randn(M, D, device="cuda").to(torch.bfloat16)
B = (torch.randn(4 * D, D, device="cuda") * 0.03).to(torch.bfloat16)
bias = torch.randn(4 * D, device="cuda") * 0.1
u = (torch.randn(M, 4 * D, device="cuda")).to(torch.bfloat16)
out = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
out2 = torch.empty_like(out)
cs = torch.empty((M + 191) // 192, 4 * D, device="cuda")
cases = {"plain bf16": dict(epi=hip.EPI_BF16, bias=bias, out=out),
         "fc1 GELU": dict(epi=hip.EPI_BF16_GELU, bias=bias, out=out, out2=out2),
         "fc2-dgrad DGELU": dict(epi=hip.EPI_BF16_DGELU, aux=u, out=out, colsum_partial=cs),
         # timing ablations (wrong results): where does the epilogue's time go?
         "plain, no stores": dict(epi=hip.EPI_BF16, bias=bias, out=out, tile=17),
         "plain, stores stay in cache": dict(epi=hip.EPI_BF16, bias=bias, out=out, tile=18),
         "GELU, no stores / no gelu": dict(epi=hip.EPI_BF16_GELU, bias=bias, out=out, out2=out2, tile=17),
         "GELU, stores stay in cache": dict(epi=hip.EPI_BF16_GELU, bias=bias, out=out, out2=out2, tile=18),
         "plain, one tile per WG": dict(epi=hip.EPI_BF16, bias=bias, out=out, tile=6)}
res = {}
for _ in range(5):
    for name, kw in cases.items():
        for _w in range(2):
            hip.gemm_nt(A, B, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _r in range(20):
            hip.gemm_nt(A, B, **kw)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
for name, v in res.items():
    med = sorted(v)[len(v) // 2]
    print(f"{name:28s} {med:6.1f} us  ({2.0 * M * 4 * D * D / med / 1e6:.0f} TF/s)", flush=True)
