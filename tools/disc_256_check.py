"""Discriminator of cfgs/larp_tokenizer_large.yaml at 16x256x256 (L = 4097, head_dim 32): runs, finite, linear backward.  (GPU box)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd as vt  # noqa: E402

m = vt.TransformerDiscriminator(384, 12, 8, 256, 4, 8, 3, frame_num=16).cuda()
x = torch.rand(2, 3, 16, 256, 256, device="cuda", requires_grad=True)
y = m(x)
y.sum().backward()
torch.cuda.synchronize()
assert y.shape == (2, 1) and torch.isfinite(y).all() and torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
g1 = x.grad.clone()
x.grad = None
(2.0 * m(x)).sum().backward()
torch.cuda.synchronize()
rel = float((x.grad - 2 * g1).norm() / (2 * g1).norm())
t0 = time.perf_counter()
for _ in range(5):
    x.grad = None
    m(x).sum().backward()
torch.cuda.synchronize()
print(f"L = {m.video_token_num + 1}, logits {y.flatten().tolist()}, backward linearity rel err {rel:.2e}, fwd+bwd {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms for 2 clips")
