"""Can memory-bound kernels (LayerNorm backward) run UNDER the grouped weight-gradient GEMM (MFMA-bound, one 144 KiB
workgroup per CU) when they sit on different HIP streams?  And what does the attention backward do next to it?  (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

M, D = 12288, 768
probs = [dict(A=torch.randn(M, p, device="cuda").to(torch.bfloat16), B=torch.randn(M, q, device="cuda").to(torch.bfloat16),
              out=torch.empty(p, q, device="cuda")) for p, q in [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)] * 4]
x = torch.randn(M, D, device="cuda")
dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
g = torch.ones(D, device="cuda")
_, mean, rstd = hip.layernorm_fwd(x, g, torch.zeros(D, device="cuda"), 1e-5)
dX = torch.randn(M, D, device="cuda")
dxb = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
B_, L, H = 8, 1536, 12
qkv = torch.randn(B_ * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
dO = torch.randn(B_ * L, H * 64, device="cuda").to(torch.bfloat16)
o, lse = hip.attention_fwd(qkv, B_, L, H)


def tn():
    hip.gemm_tn_grouped(probs)


def ln(n=16):
    for _ in range(n):
        hip.layernorm_bwd(dy, x, g, mean, rstd, dres=dX, dx=dX, dxb=dxb)


def attn(n=2):
    for _ in range(n):
        hip.attention_bwd(qkv, o, dO, lse, B_, L, H)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


s2 = torch.cuda.Stream()


def par(other):
    def f():
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            tn()
        other()
        main.wait_stream(s2)
    return f


for name, other in (("16 x LayerNorm backward", ln), ("2 x attention backward", attn)):
    a, b = timed(tn), timed(other)
    c = timed(par(other))
    print(f"wgrad group {a:7.1f} us | {name} {b:7.1f} us | serial {a + b:7.1f} us | two streams {c:7.1f} us  (hidden: {a + b - c:6.1f} us)", flush=True)
