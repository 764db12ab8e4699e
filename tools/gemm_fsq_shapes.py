"""128x128 vs 192x192 tiles on the GEMM shapes of autoencoder_large (width 1024, GEGLU 2752) at 4 and 8 clips.  (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402
from tools.gemm_bench import bench_nt  # noqa: E402

if __name__ == "__main__":
    for M in (8192, 16384):
        for N, K, epi, what in [(4096, 1024, hip.EPI_BF16, "to_qkv fwd"), (1024, 1024, hip.EPI_F32, "out_proj fwd"), (5504, 1024, hip.EPI_BF16, "fc1 fwd"),
                                (1024, 2752 + 64, hip.EPI_F32, "fc2 fwd"), (2752 + 64, 1024, hip.EPI_BF16, "fc2 dgrad"), (1024, 5504, hip.EPI_BF16, "fc1 dgrad"),
                                (1024, 1024, hip.EPI_BF16, "out_proj dgrad"), (1024, 4096, hip.EPI_F32, "to_qkv dgrad")]:
            row = f"M={M:6d} {what:14s} N={N:5d} K={K:5d}"
            for v in (1, 2):
                bench_nt(M, N, K, epi, v, reps=5)
                t = bench_nt(M, N, K, epi, v, reps=40)
                row += f"  v{v} {t:7.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF/s"
            t192 = ((M + 191) // 192) * ((N + 191) // 192)
            t128 = ((M + 127) // 128) * ((N + 127) // 128)
            row += f"  | rounds 192: {(t192 + 255) // 256}  128: {(t128 + 511) // 512}"
            print(row, flush=True)
