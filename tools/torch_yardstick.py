"""Yardstick: the same training step written with STOCK PyTorch-ROCm ops, on the same GPU.

What the reference would run on this box: its tokenizer is timm `Block`s (nn.LayerNorm, nn.Linear, F.scaled_dot_product_attention,
nn.GELU), a Conv3d patch embed and a VQ bottleneck made of torch ops, under torch.autocast(bf16) (models/larp_tokenizer.py:400-496,
models/transformer.py:52-70, models/bottleneck.py:262-324, trainers/larp_tokenizer_trainer.py:244).  The reference itself cannot be imported
(timm and half a dozen other packages are absent), so this file is a self-contained nn.Module of the same architecture, sizes and
dtype policy -- NOT the checker (nothing under oracle/ is imported) and not a product path: a measuring stick for bench.py's number.
Library kernels do all the work: hipBLASLt / rocBLAS GEMMs, the SDPA backend PyTorch picks on gfx950, native LayerNorm / GELU.

    python3 tools/torch_yardstick.py [--batch 8] [--steps 10] [--ops] [--compile]

--ops adds per-op timings at the step's shapes (SDPA forward / backward per backend, LayerNorm forward / backward) next to this
library's kernels.  One JSON line per measurement on stdout."""
import argparse
import json
import math
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

D, H, DEPTH, NQ, KCODES, DBN = 768, 12, 12, 1024, 8192, 24
T, S, PT, P = 16, 128, 2, 16
NV = (T // PT) * (S // P) ** 2


class Block(nn.Module):
    """pre-LN attention + MLP block, timm defaults as the reference constructs it (qkv without bias, no LayerScale, no dropout)"""

    def __init__(self):
        super().__init__()
        self.norm1 = nn.LayerNorm(D)            # timm Block: nn.LayerNorm default eps 1e-5
        self.qkv = nn.Linear(D, 3 * D, bias=False)
        self.proj = nn.Linear(D, D)
        self.norm2 = nn.LayerNorm(D)
        self.fc1 = nn.Linear(D, 4 * D)
        self.fc2 = nn.Linear(4 * D, D)

    def forward(self, x):
        B, L, _ = x.shape
        qkv = self.qkv(self.norm1(x)).reshape(B, L, 3, H, D // H).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        x = x + self.proj(a.transpose(1, 2).reshape(B, L, D))
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class Stack(nn.Module):
    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList(Block() for _ in range(DEPTH))

    def forward(self, context, query):
        h = torch.cat([context, query], 1)
        for b in self.blocks:
            h = b(h)
        return h[:, -query.shape[1]:]


class Tokenizer(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = nn.Conv3d(3, D, (PT, P, P), (PT, P, P))
        self.register_buffer("pe_x", torch.randn(1, NV, D) * 0.02)
        self.register_buffer("pe_z", torch.randn(1, NQ, D) * 0.02)
        self.q_emb = nn.Parameter(torch.randn(NQ, D) * 0.02)
        self.patch_q = nn.Parameter(torch.randn(1, NV, D) * 0.02)
        self.encoder, self.decoder = Stack(), Stack()
        self.in_linear, self.out_linear = nn.Linear(D, DBN), nn.Linear(DBN, D)
        self.codebook = nn.Embedding(KCODES, DBN)
        self.final_norm = nn.LayerNorm(D, eps=1e-6)
        self.final = nn.Linear(D, PT * P * P * 3)

    def quantize(self, z, stochastic):
        with torch.autocast("cuda", enabled=False):
            z = F.normalize(z.float(), dim=-1)
            e = F.normalize(self.codebook.weight, dim=-1)
            flat = z.reshape(-1, DBN)
            cos = flat @ e.t()
            probs = F.softmax(cos / 0.03, dim=-1)
            idx = torch.multinomial(probs, 1).squeeze(-1) if stochastic else probs.argmax(-1)
            q = e[idx].view_as(z)
            loss_q = 0.25 * F.mse_loss(q.detach(), z) + F.mse_loss(q, z.detach())
            return z + (q - z).detach(), loss_q

    def forward(self, x, stochastic=True):
        B = x.shape[0]
        tok = self.embed(x).flatten(2).transpose(1, 2) + self.pe_x
        enc = self.encoder(tok, self.q_emb.unsqueeze(0).repeat(B, 1, 1))
        zq, loss_q = self.quantize(self.in_linear(enc), stochastic)
        z = self.out_linear(zq) + self.pe_z
        dec = self.decoder(z, self.patch_q.expand(B, -1, -1))
        y = self.final(self.final_norm(dec))
        t, h = T // PT, S // P
        y = y.reshape(B, t, h, h, PT, P, P, 3).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, 3, T, S, S)
        return y.contiguous(), loss_q


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def ev_time(fn, n=20, warmup=5):
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3     # us


def step_yardstick(a):
    torch.manual_seed(0)
    m = Tokenizer().cuda()
    nn.init.xavier_uniform_(m.final.weight)
    x = torch.rand(a.batch, 3, T, S, S, device="cuda")
    run = torch.compile(m) if a.compile else m

    def step():
        for p in m.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y, lq = run(x, not a.deterministic)
        loss = (y.float() - x).abs().mean() + 0.1 * lq
        loss.backward()

    dt = timed(step, a.steps, a.warmup)
    print(json.dumps({"what": "stock PyTorch-ROCm step (nn.Linear / LayerNorm / SDPA / GELU under autocast bf16), forward + backward, no optimizer",
                      "torch": torch.__version__, "compiled": bool(a.compile), "clips_per_gpu": a.batch, "ms_per_step": round(dt * 1e3, 3),
                      "clips_per_s": round(a.batch / dt, 2), "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)


def ops_yardstick(a):
    from torch.nn.attention import SDPBackend, sdpa_kernel
    B, L, hd = a.batch, NV + NQ, D // H
    q, k, v = (torch.randn(B, H, L, hd, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    go = torch.randn(B, H, L, hd, device="cuda", dtype=torch.bfloat16)
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
        try:
            with sdpa_kernel(be):
                f = ev_time(lambda: F.scaled_dot_product_attention(q, k, v))
                o = F.scaled_dot_product_attention(q, k, v)
                bw = ev_time(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True))
            print(json.dumps({"op": f"torch SDPA [{name}] B={B} H={H} L={L} hd={hd} bf16", "fwd_us": round(f, 1), "bwd_us": round(bw, 1)}), flush=True)
        except Exception as e:      # a backend this build does not have for gfx950
            print(json.dumps({"op": f"torch SDPA [{name}]", "error": str(e).splitlines()[0][:160]}), flush=True)
    # this library's kernels on the same operands (packed qkv layout [B, L, 3, H, hd])
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import video_tokenizer_amd as vt
    from video_tokenizer_amd import hip
    qkv = torch.randn(B * L, 3 * H * hd, device="cuda", dtype=torch.bfloat16)
    out, lse = hip.attention_fwd(qkv, B, L, H, hd)
    do, dqkv = torch.randn_like(out), torch.empty_like(qkv)
    f = ev_time(lambda: hip.attention_fwd(qkv, B, L, H, hd, o=out))
    bw = ev_time(lambda: hip.attention_bwd(qkv, out, do, lse, B, L, H, hd, dqkv=dqkv))
    print(json.dumps({"op": "this library: vt_attention_fwd_rows / vt_attention_bwd_rows (delta + dQ + dK/dV), same shape", "fwd_us": round(f, 1), "bwd_us": round(bw, 1)}), flush=True)
    # LayerNorm at the step's shape
    M = B * L
    x = torch.randn(M, D, device="cuda", requires_grad=True)
    w, b = torch.ones(D, device="cuda", requires_grad=True), torch.zeros(D, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        f = ev_time(lambda: F.layer_norm(x, (D,), w, b))
        y = F.layer_norm(x, (D,), w, b)
    gy = torch.randn_like(y)
    bw = ev_time(lambda: torch.autograd.grad(y, (x, w, b), gy, retain_graph=True))
    print(json.dumps({"op": f"torch layer_norm under autocast, {M} x {D} fp32 rows", "fwd_us": round(f, 1), "bwd_us": round(bw, 1), "out_dtype": str(y.dtype),
                      "this_library_in_step_us": "ln_fwd 10.5, ln_bwd 26.1 (profiles/r04_kernel_stats.csv; the backward also adds into the residual gradient and writes its bf16 copy)"}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--deterministic", action="store_true", help="argmax instead of multinomial sampling in the quantizer")
    ap.add_argument("--compile", action="store_true", help="torch.compile the module (the reference's optional `compile` flag)")
    ap.add_argument("--ops", action="store_true")
    a = ap.parse_args()
    if a.ops:
        ops_yardstick(a)
    else:
        step_yardstick(a)
