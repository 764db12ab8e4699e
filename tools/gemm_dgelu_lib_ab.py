"""fc2 input gradient x gelu'(u) (192x192 NT kernel, DGELU epilogue) on the library named by VT_HIP_LIB (or the default build): time of the launch at
the step's shape beside the plain-epilogue launch of the same shape.  Run once per library, interleaved by the calling script.  (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402
from tools.gemm_bench import bench_nt  # noqa: E402

if __name__ == "__main__":
    M, N, K = 12288, 3072, 768
    for epi in (hip.EPI_BF16, hip.EPI_BF16_DGELU):       # warm the chip on both
        bench_nt(M, N, K, epi, 0, reps=40)
    best = {}
    for rep in range(3):
        for name, epi in (("plain", hip.EPI_BF16), ("dgelu", hip.EPI_BF16_DGELU)):
            best[name] = min(best.get(name, 1e9), bench_nt(M, N, K, epi, 0, reps=60))
    print(f"{os.path.basename(os.environ.get('VT_HIP_LIB', 'libvt_hip.so')):34s} plain {best['plain']:6.1f} us   gelu' {best['dgelu']:6.1f} us   epilogue cost {best['dgelu'] - best['plain']:5.1f} us", flush=True)
