"""Where a tile iteration of the attention kernels spends its cycles (diagnostic build with s_memtime stamps, tools/attn_stamps.sh).
Segments per iteration and wave: tile body (staging issue + fragment reads + MFMA + softmax) | s_waitcnt vmcnt(0) on the NEXT tile's
LDS-DMA | workgroup barrier.  Prints, per kernel, the mean cycles per iteration of each segment over all waves, and by dispatch
round (workgroups that start on an empty chip vs the thin last round)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

B, L, H = 8, 1536, 12
qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
dO = torch.randn(B * L, H * 64, device="cuda").to(torch.bfloat16)
for _ in range(20):     # clocks settle
    o, lse = hip.attention_fwd(qkv, B, L, H)
    d = hip.attention_bwd(qkv, o, dO, lse, B, L, H)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * (3 * 2048 * 4 * 6))()
fn = hip.lib().vt_attention_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
t = torch.tensor(list(buf), dtype=torch.float64).reshape(3, 2048, 4, 6)
nwg = B * H * ((L + 127) // 128)
for k, name in enumerate(("fwd", "dq", "dkv")):
    x = t[k, :nwg]                                   # [wg, wave, 4]
    it = x[..., 3].clamp(min=1)
    per = x[..., :3] / it[..., None]
    tot = per.sum(-1)
    print(f"{name}: {nwg} workgroups, iterations/wave {it.mean().item():.1f}; cycles per iteration: body {per[..., 0].mean():.0f}  dma-wait {per[..., 1].mean():.0f}  "
          f"barrier {per[..., 2].mean():.0f}  total {tot.mean():.0f}   (shares {100 * per[..., 0].mean() / tot.mean():.0f} / {100 * per[..., 1].mean() / tot.mean():.0f} / {100 * per[..., 2].mean() / tot.mean():.0f} %)")
    for lo, hi, label in ((0, 1024, "blockIdx < 1024"), (1024, nwg, "blockIdx >= 1024 (thin round)")):
        if hi > lo:
            y = per[lo:hi]
            print(f"     {label:30s} body {y[..., 0].mean():.0f}  dma-wait {y[..., 1].mean():.0f}  barrier {y[..., 2].mean():.0f}")
    w = per.mean(0)                                  # per wave index
    print("     by wave: " + "  ".join(f"w{i}: {w[i, 0]:.0f}/{w[i, 1]:.0f}/{w[i, 2]:.0f}" for i in range(4)))
    # timeline (s_memrealtime, 100 MHz): when do workgroups start / end relative to the first start, and the shader clock they ran at
    t0 = x[..., 4].min()
    start, end = (x[:, 0, 4] - t0) / 100.0, (x[:, 0, 5] - t0) / 100.0          # us, wave 0 of each workgroup
    cyc = x[:, 0, :3].sum(-1)
    clk = cyc / ((end - start) * 1e3)                                          # cycles per ns = GHz
    order = torch.argsort(start)
    q = lambda v, f: v.kthvalue(max(1, int(f * v.numel())))[0].item()
    print(f"     loop start (us after the first): 50% {q(start, .5):.1f}  89% {q(start, .89):.1f}  95% {q(start, .95):.1f}  last {start.max().item():.1f};  "
          f"loop end: first {end.min().item():.1f}  50% {q(end, .5):.1f}  last {end.max().item():.1f};  loop duration: 10% {q(end - start, .1):.1f}  50% {q(end - start, .5):.1f}  90% {q(end - start, .9):.1f} us;  "
          f"shader clock in the loop: 10% {q(clk, .1):.2f}  50% {q(clk, .5):.2f}  90% {q(clk, .9):.2f} GHz")
    late = start > 0.5 * end.max()
    print(f"     workgroups that start in the second half of the kernel: {int(late.sum())} (loop duration {((end - start)[late]).mean().item() if late.any() else 0:.1f} us vs {((end - start)[~late]).mean().item():.1f} us for the others)")
