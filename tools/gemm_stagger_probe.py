"""Timing experiment: persistent 192x192 NT kernel with every other CU of an XCD starting 1..8 us late (vtGemmNT.tile 8..15), so that one half's
HBM-bound epilogues fall into the other half's main loops.  python tools/gemm_stagger_probe.py (GPU)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402
from tools.gemm_bench import bench_nt  # noqa: E402

M = 12288
names = {hip.EPI_BF16: "bf16", hip.EPI_BF16_GELU: "gelu", hip.EPI_F32: "f32res", hip.EPI_BF16_DGELU: "dgelu"}
for N, K, epi, what in [(2304, 768, hip.EPI_BF16, "qkv fwd"), (3072, 768, hip.EPI_BF16_GELU, "fc1 fwd"), (3072, 768, hip.EPI_BF16_DGELU, "fc2 dgrad"),
                        (768, 3072, hip.EPI_F32, "fc2 fwd"), (768, 768, hip.EPI_BF16, "proj dgrad")]:
    row = f"{what:10s} N={N:5d} K={K:5d} {names[epi]:7s}"
    for v in (2, 9, 11, 13, 15, 2):
        bench_nt(M, N, K, epi, v, reps=5)
        t = bench_nt(M, N, K, epi, v, reps=60)
        row += f"  {'plain' if v == 2 else f'+{v - 7}us'} {t:6.1f}"
    print(row, flush=True)
