"""Where an OUTPUT tile of the 192x192 NT kernel spends its cycles (diagnostic build with s_memtime stamps, tools/gemm_stamps.sh).
Segments per output tile and wave: 0 main loop | 1 barrier + next tile's K-tile 0 issued | 2 accumulators -> LDS image | 3 barrier |
4 read-back + global stores | 5 barrier, K-tiles 1-2 issued, wait for K-tile 0, barrier | 6 fragments of K-tile 0.  Mean over the
waves of all 256 workgroups; the launch's measured time next to it."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

M, D = 12288, 768
fn = hip.lib().vt_gemm_nt_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
names = ("main loop", "barrier+next K0", "acc -> LDS", "barrier", "read-back+stores", "barrier+K1,K2+wait+barrier", "fragments K0")
shapes = (("qkv forward   N=2304 K= 768 bf16", 3 * D, D, hip.EPI_BF16), ("fc1 forward   N=3072 K= 768 GELU", 4 * D, D, hip.EPI_BF16_GELU),
          ("proj dgrad    N= 768 K= 768 bf16", D, D, hip.EPI_BF16), ("qkv dgrad     N= 768 K=2304 bf16", D, 3 * D, hip.EPI_BF16),
          ("fc1 dgrad     N= 768 K=3072 bf16", D, 4 * D, hip.EPI_BF16))
for label, N, K, epi in shapes:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.1
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    kw = dict(epi=epi, bias=bias, out=out)
    if epi == hip.EPI_BF16_GELU:
        kw["out2"] = torch.empty_like(out)
    for _ in range(30):
        hip.gemm_nt(A, B, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.gemm_nt(A, B, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    buf = (ctypes.c_uint64 * (256 * 8 * 8))()
    assert fn(buf) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).reshape(256, 8, 8)
    tiles = t[..., 7].clamp(min=1)
    per = t[..., :7] / tiles[..., None]
    tot = per.sum(-1).mean().item()
    print(f"{label}: {us:.1f} us per launch (stamped build), {tiles.mean().item():.2f} output tiles per workgroup, {K // 64} K-tiles; cycles per output tile {tot:.0f}")
    for i, n in enumerate(names):
        v = per[..., i].mean().item()
        print(f"     {n:30s} {v:8.0f}  {100 * v / tot:5.1f} %   (min wave {per[..., i].min().item():.0f}, max {per[..., i].max().item():.0f})")
    print(f"     main loop per K-tile: {per[..., 0].mean().item() / (K // 64):.0f} cycles;  outside the main loop: {tot - per[..., 0].mean().item():.0f} cycles per output tile", flush=True)
