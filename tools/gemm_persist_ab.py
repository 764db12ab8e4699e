"""Persistent 192x192 NT kernel (workgroup b walks tiles b, b + 256, ...; the next tile's first K-tile is in flight during the
epilogue) vs the same kernel launched one tile per workgroup (vt_set_gemm_variant 6), on the training step's shapes.  (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402
from tools.gemm_bench import bench_nt  # noqa: E402

if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    names = {hip.EPI_BF16: "bf16", hip.EPI_BF16_GELU: "gelu", hip.EPI_F32: "f32res", hip.EPI_BF16_DGELU: "dgelu"}
    shapes = [(2304, 768, hip.EPI_BF16, "qkv fwd"), (768, 768, hip.EPI_F32, "proj fwd"), (3072, 768, hip.EPI_BF16_GELU, "fc1 fwd"),
              (768, 3072, hip.EPI_F32, "fc2 fwd"), (3072, 768, hip.EPI_BF16_DGELU, "fc2 dgrad"), (768, 3072, hip.EPI_BF16, "fc1 dgrad"),
              (768, 768, hip.EPI_BF16, "proj dgrad"), (768, 2304, hip.EPI_F32, "qkv dgrad")]
    tot = {2: 0.0, 6: 0.0}
    for N, K, epi, what in shapes:
        row = f"{what:10s} N={N:5d} K={K:5d} {names[epi]:7s}"
        for rep in range(2):
            for v in (6, 2):
                bench_nt(M, N, K, epi, v, reps=5)
                t = bench_nt(M, N, K, epi, v, reps=60)
                row += f"  {'one-tile' if v == 6 else 'persist'} {t:6.1f}"
                tot[v] += t / 2
        print(row, flush=True)
    print(f"sum per block: one tile per workgroup {tot[6]:.1f} us, persistent {tot[2]:.1f} us")
