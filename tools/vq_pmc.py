"""Codebook search at the training shape (N = K = 8192, d = 24) in the three index modes, for rocprofv3 passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

N, K, d = 8192, 8192, 24
z = torch.randn(N, 64, device="cuda")
W = (torch.rand(K, d, device="cuda") - 0.5)
for mode in (0, 1, 2):
    for _ in range(5):
        hip.vq_forward(z, W, mode, inv_tau=1 / 0.03, seed=7, ldp=64)
torch.cuda.synchronize()
