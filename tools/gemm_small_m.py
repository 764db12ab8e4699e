"""NT GEMMs at one and two clips per GPU (M = 1536 / 3072) on the 128x128 kernel: 2-deep ring (tile 1) vs 4-deep ring (tile 16), K unsplit,
then K split over 2..8 workgroups per tile (vtGemmNT.splitk) and the automatic rule -- interleaved in one process on random data."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

D = 768
CASES = [("2-deep", dict(tile=1, splitk=1)), ("4-deep", dict(tile=16, splitk=1))] + [(f"split{s}", dict(tile=0, splitk=s)) for s in (2, 3, 4, 6, 7, 8)] + [("auto", dict(tile=0, splitk=None))]
for M in (1536, 3072):
    for name, N, K, epi in (("qkv fwd", 3 * D, D, hip.EPI_BF16), ("proj", D, D, hip.EPI_F32), ("fc1 fwd", 4 * D, D, hip.EPI_BF16_GELU), ("fc2 fwd", D, 4 * D, hip.EPI_F32),
                            ("fc1 dgrad", D, 4 * D, hip.EPI_BF16), ("qkv dgrad", D, 3 * D, hip.EPI_BF16)):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == hip.EPI_F32 else torch.bfloat16)
        out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == hip.EPI_BF16_GELU else None
        res = {}
        for _ in range(3):
            for label, kw in CASES:
                if kw.get("splitk", 0) > K // 64 or kw.get("splitk", 0) * ((M + 127) // 128) * ((N + 127) // 128) > 512:
                    continue        # the workspace holds 512 partial tiles
                for _w in range(3):
                    hip.gemm_nt(A, B, epi, out=out, out2=out2, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _r in range(20):
                    hip.gemm_nt(A, B, epi, out=out, out2=out2, **kw)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(label, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        med = {k: sorted(v)[1] for k, v in res.items()}
        print(f"M={M} {name:9s} N={N} K={K}: " + "  ".join(f"{k} {v:5.1f}" for k, v in med.items()) + f"  us   (auto {2.0 * M * N * K / med['auto'] / 1e6:.0f} TF/s)", flush=True)
