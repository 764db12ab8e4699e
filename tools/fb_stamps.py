"""Where an iteration of attn_bwd_fused_kernel spends its cycles (diagnostic build with s_memtime stamps, tools/fb_stamps.sh).
Segments: 0 phase A compute | 1 announce + drain | 2 barrier 1 | 3 phase B prologue (drain, fetch) | 4 dQ MFMAs + exchange |
5 barrier 2 | 6 finalize | 7 loop bookkeeping / item prologue.  Prints the share of each, for waves 0-3 and 4-7 separately."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

names = ["A compute", "announce+drain", "barrier1", "B prologue", "dQ mfma+xchg", "barrier2", "finalize", "bookkeeping"]
for (B, L, H) in [(8, 1536, 12), (1, 1536, 12)]:
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
    dO = torch.randn(B * L, H * 64, device="cuda").to(torch.bfloat16)
    o, lse = hip.attention_fwd(qkv, B, L, H)
    for _ in range(3):
        hip.attention_bwd(qkv, o, dO, lse, B, L, H, fused=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_uint64 * (2048 * 8))()
    fn = hip.lib().vt_attention_bwd_fused_stamps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf) == 0
    t = torch.tensor(list(buf), dtype=torch.float64).reshape(256, 8, 8)
    print(f"B={B} L={L}: cycles per workgroup (median over workgroups of the per-wave sums) {t.sum(-1).median().item():.0f}")
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        x = t[:, sl].mean(dim=(0, 1))
        tot = x.sum().item()
        print(f"  {grp}: " + "  ".join(f"{n} {100 * v / tot:.1f}%" for n, v in zip(names, x.tolist())))
