"""Codebook search for rocprofv3 passes and timing.
  python tools/vq_pmc.py        : training shape N = K = 8192, d = 24, the three index modes
  python tools/vq_pmc.py sq     : the 'sq' bottleneck's search, N = 8192 tokens against the frozen K = 196 560 x 24 codebook (cosine argmax),
                                  prints us per launch, TFLOP/s of exact fp32 (2 N K d) and the algorithmic bytes"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

sq = len(sys.argv) > 1 and sys.argv[1] == "sq"
N, K, d = (8192, 196560, 24) if sq else (8192, 8192, 24)
z = torch.randn(N, 64, device="cuda")
W = (torch.rand(K, d, device="cuda") - 0.5)
modes = (1,) if sq else (0, 1, 2)
for mode in modes:
    for _ in range(5):
        hip.vq_forward(z, W, mode, inv_tau=1 / 0.03 if not sq else 1.0, seed=7, ldp=64)
torch.cuda.synchronize()
if sq:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        hip.vq_forward(z, W, 1, inv_tau=1.0, seed=7, ldp=64)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    alg = N * d * 4 + K * d * 4 + N * (d * 4 * 2 + 8)
    print(f"sq search N={N} K={K} d={d}: {us:.1f} us per forward (all its kernels), {2.0 * N * K * d / us / 1e6:.1f} TFLOP/s exact fp32 (peak 157), "
          f"algorithmic bytes {alg / 1e6:.1f} MB = {alg / us / 1e3:.1f} GB/s; the N x K score matrix ({N * K * 4 / 1e9:.1f} GB in the reference) is never written")
