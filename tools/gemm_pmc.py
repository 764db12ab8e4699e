"""Launch the dominant kernel (gemm_nt192_kernel<VT_EPI_BF16>) at the training step's four shapes, a few times each,
for `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (see tools/pmc_traffic.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

M, D = 12288, 768
for (m, n, k) in [(M, 3 * D, D), (M, D, 4 * D), (M, D, D), (M, D, 3 * D)]:
    A = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    B = (torch.randn(n, k, device="cuda") * 0.03).to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(4):
        hip.gemm_nt(A, B, hip.EPI_BF16, out=out)
    torch.cuda.synchronize()
