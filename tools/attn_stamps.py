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
buf = (ctypes.c_uint64 * (3 * 2048 * 4 * 4))()
fn = hip.lib().vt_attention_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
t = torch.tensor(list(buf), dtype=torch.float64).reshape(3, 2048, 4, 4)
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
