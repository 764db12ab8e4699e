"""NT GEMMs at one and two clips per GPU (M = 1536 / 3072): the 128x128 kernel behind its 2-deep ring (tile 1) vs the 4-deep ring
(tile 16) vs the automatic choice, interleaved in one process on random data."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

D = 768
for M in (1536, 3072):
    for name, N, K, epi in (("qkv fwd", 3 * D, D, hip.EPI_BF16), ("proj", D, D, hip.EPI_F32), ("fc1 fwd", 4 * D, D, hip.EPI_BF16_GELU), ("fc2 fwd", D, 4 * D, hip.EPI_F32),
                            ("fc1 dgrad", D, 4 * D, hip.EPI_BF16), ("qkv dgrad", D, 3 * D, hip.EPI_BF16)):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == hip.EPI_F32 else torch.bfloat16)
        out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == hip.EPI_BF16_GELU else None
        res = {}
        for _ in range(3):
            for tile in (1, 16, 0):
                for _w in range(3):
                    hip.gemm_nt(A, B, epi, out=out, out2=out2, tile=tile)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _r in range(20):
                    hip.gemm_nt(A, B, epi, out=out, out2=out2, tile=tile)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(tile, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        med = {k: sorted(v)[1] for k, v in res.items()}
        print(f"M={M} {name:9s} N={N} K={K}: 2-deep {med[1]:6.1f} us   4-deep {med[16]:6.1f} us   auto {med[0]:6.1f} us   ({2.0 * M * N * K / med[16] / 1e6:.0f} TF/s)", flush=True)
