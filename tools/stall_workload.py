"""Workload for tools/pmc_stalls.sh: the long-K and short-K NT GEMM, the grouped weight-gradient GEMM, attention fwd/bwd."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

M, D = 12288, 768
for (m, n, k) in [(M, D, 4 * D), (M, 3 * D, D)]:
    A = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    B = (torch.randn(n, k, device="cuda") * 0.03).to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        hip.gemm_nt(A, B, hip.EPI_BF16, out=out)
probs = [dict(A=torch.randn(M, p, device="cuda").to(torch.bfloat16), B=torch.randn(M, q, device="cuda").to(torch.bfloat16),
              out=torch.empty(p, q, device="cuda")) for p, q in [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)] * 4]
for _ in range(2):
    hip.gemm_tn_grouped(probs)
B_, L, H = 8, 1536, 12
qkv = torch.randn(B_ * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
dO = torch.randn(B_ * L, H * 64, device="cuda").to(torch.bfloat16)
for _ in range(3):
    o, lse = hip.attention_fwd(qkv, B_, L, H)
    d = hip.attention_bwd(qkv, o, dO, lse, B_, L, H)
torch.cuda.synchronize()
