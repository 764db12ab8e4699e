"""Ablation of the 192x192 NT kernel (timing only, results are wrong by construction): full kernel vs
'no LDS-DMA after the prologue' (compute + LDS reads + barriers only) vs 'no MFMA / LDS reads' (staging only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

M, D = 12288, 768
for (m, n, k) in [(M, D, 4 * D), (M, 3 * D, D), (M, D, D)]:
    A = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    B = (torch.randn(n, k, device="cuda") * 0.03).to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rnd in range(2):
        for name, v in (("full", 2), ("no_dma", 3), ("no_mfma", 4)):
            hip.GEMM_TILE = v
            for _ in range(3):
                hip.gemm_nt(A, B, hip.EPI_BF16, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip.gemm_nt(A, B, hip.EPI_BF16, out=out)
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M={m} N={n} K={k}: " + "  ".join(f"{kk} {vv:.1f} us" for kk, vv in res.items()), flush=True)
hip.GEMM_TILE = 0
