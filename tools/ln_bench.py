"""LayerNorm forward / backward at the step's shape (12288 x 768), HBM bytes and achieved TB/s.  python tools/ln_bench.py (GPU)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

M, D = 12288, 768
x = torch.randn(M, D, device="cuda")
g, b = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda")
dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
dres = torch.randn(M, D, device="cuda")
y, mean, rstd = hip.layernorm_fwd(x, g, b, 1e-5)
dx = torch.empty_like(x)
dxb = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    hip.layernorm_fwd(x, g, b, 1e-5, y=y)
    hip.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dx=dx, dxb=dxb)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 50
e[0].record()
for _ in range(n):
    hip.layernorm_fwd(x, g, b, 1e-5, y=y)
e[1].record()
for _ in range(n):
    hip.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dx=dx, dxb=dxb)
e[2].record()
torch.cuda.synchronize()
tf, tb = e[0].elapsed_time(e[1]) / n * 1e3, e[1].elapsed_time(e[2]) / n * 1e3
bf, bb = M * D * 6, M * D * (2 + 4 + 4 + 4 + 2)
print(f"ln_fwd {tf:.1f} us = {bf / tf / 1e6:.2f} TB/s ({bf / 1e6:.1f} MB)   ln_bwd (+reduce) {tb:.1f} us = {bb / tb / 1e6:.2f} TB/s ({bb / 1e6:.1f} MB)")
