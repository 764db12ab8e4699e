"""Attention forward / backward at 1, 2, 4, 8 clips per GPU (B x 12 heads, L = 1536).  Round 3 used it to A/B a four-deep K/V ring for
launches of at most two workgroups per CU against the two-deep ring (VT_ATTN_RING, since removed): bit-identical and no faster
(one clip: forward 25.3 vs 25.4 us, backward 65.4 vs 61.8 us)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402

H, L = 12, 1536
for B in (1, 2, 4, 8):
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
    dO = torch.randn(B * L, H * 64, device="cuda").to(torch.bfloat16)
    o, lse = hip.attention_fwd(qkv, B, L, H)
    dqkv = hip.attention_bwd(qkv, o, dO, lse, B, L, H)
    res = {}
    for _ in range(5):
        for name, fn in (("fwd", lambda: hip.attention_fwd(qkv, B, L, H, o=o)), ("bwd", lambda: hip.attention_bwd(qkv, o, dO, lse, B, L, H, dqkv=dqkv))):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"B={B}: fwd {sorted(res['fwd'])[2]:6.1f} us   bwd (dq + dkv + delta) {sorted(res['bwd'])[2]:6.1f} us", flush=True)
