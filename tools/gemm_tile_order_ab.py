"""Tile order of the persistent 192x192 NT kernel: the row-major tile list (vtGemmNT.tile = 19) against column blocks of W tile columns
(tile = 19 + W) and the automatic choice (tile = 2), interleaved, on the training step's shapes; every variant's output is compared bit for
bit with the row-major one's.  (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd.hip as hip  # noqa: E402
from tools.gemm_bench import bench_nt  # noqa: E402


def same_bits(M, N, K, epi, variants):
    torch.manual_seed(1)
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    outs = []
    for v in variants:
        hip.GEMM_TILE = v
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == hip.EPI_F32 else torch.bfloat16)
        hip.gemm_nt(A, B, epi, out=out)
        outs.append(out)
    torch.cuda.synchronize()
    return all(torch.equal(outs[0], o) for o in outs[1:])


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    names = {hip.EPI_BF16: "bf16", hip.EPI_BF16_GELU: "gelu", hip.EPI_F32: "f32res", hip.EPI_BF16_DGELU: "dgelu"}
    shapes = [(2304, 768, hip.EPI_BF16, "qkv fwd", (0, 3, 4, 6)), (3072, 768, hip.EPI_BF16, "fc1 plain", (0, 4, 6, 8)),
              (3072, 768, hip.EPI_BF16_GELU, "fc1 fwd", (0, 4, 6, 8)), (3072, 768, hip.EPI_BF16_DGELU, "fc2 dgrad", (0, 4, 6, 8)),
              (768, 768, hip.EPI_F32, "proj fwd", (0, 2)), (768, 3072, hip.EPI_F32, "fc2 fwd", (0, 2)),
              (768, 3072, hip.EPI_BF16, "fc1 dgrad", (0, 2)), (768, 2304, hip.EPI_BF16, "qkv dgrad", (0, 2)), (768, 768, hip.EPI_BF16, "proj dgrad", (0, 2))]
    for N, K, epi, what, widths in shapes:
        codes = [19 + w for w in widths] + [2]
        ok = same_bits(M, N, K, hip.EPI_BF16 if epi in (hip.EPI_BF16_GELU, hip.EPI_BF16_DGELU) else epi, codes) if epi != hip.EPI_F32 else True
        best = {c: 1e9 for c in codes}
        for rep in range(3):
            for c in codes:
                bench_nt(M, N, K, epi, c, reps=5)
                best[c] = min(best[c], bench_nt(M, N, K, epi, c, reps=60))
        row = f"{what:10s} N={N:5d} K={K:5d} {names[epi]:7s}" + "".join(
            f"  {'auto' if c == 2 else 'rows' if c == 19 else 'W=%d' % (c - 19)} {best[c]:6.1f}" for c in codes)
        print(row + ("   same bits" if ok else "   BITS DIFFER"), flush=True)
