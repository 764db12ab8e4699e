"""The 'sq' bottleneck on the GPU: `VectorQuantizer` of models/model_new/quantizer/fsq.py:144-230 behind
LARPTokenizer(bottleneck_type='sq') (models/larp_tokenizer.py:225-229, 423-428) -- cosine nearest neighbour over the frozen
196 560 x 24 codebook on the exact-fp32 MFMA search kernel.  Indices bit-exact against the fixed-order C oracle and against
the reference class's own outputs (tests/golden/sq_*.npz); full-size properties at N = 8192."""
import os

import numpy as np
import pytest
import torch

from oracle import inputs as gen
from oracle import larp_oracle as O
from oracle import vq_c

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("case", [((2, 128), 4096, 601), ((1, 256), 196560, 602)])
def test_sq_module_matches_reference_fixture_and_c_oracle(case):
    import video_tokenizer_amd as vt
    from tests.golden.make_golden import sq_codebook
    (b, n), K, seed = case
    f = np.load(os.path.join(G, f"sq_N{b * n}_K{K}.npz"))
    W = sq_codebook(K, seed)
    q = vt.sq.VectorQuantizer(n_embed=K, embed_dim=24, l2_norm=True, beta=0.25, input_format="blc", predefined_codebook=W).cuda()
    z = torch.from_numpy(gen.normal((b, n, 24), seed + 1)).cuda().requires_grad_(True)
    g = torch.from_numpy(gen.normal((b, n, 24), seed + 2)).cuda()
    o = q(z)
    (o["output"] * g).sum().add(0.7 * o["loss_codebook"]).backward()
    torch.cuda.synchronize()
    idx = o["indices"].reshape(-1).cpu().numpy()
    c = vq_c.vq_forward(gen.normal((b * n, 24), seed + 1), W, "D", temperature=1.0)
    assert np.array_equal(idx, c["idx"])                                             # bit-exact vs the fixed-order oracle
    mism = np.nonzero(idx != f["idx"].astype(np.int64))[0]
    assert all(f["margin"][i] < 1e-6 for i in mism), (mism, f["margin"][mism])        # and vs the reference, up to fp32 near-ties
    ok = idx == f["idx"].astype(np.int64)
    np.testing.assert_allclose(o["output"].detach().cpu().numpy().reshape(-1, 24)[ok], f["output"].reshape(-1, 24)[ok], rtol=0, atol=2e-7)
    np.testing.assert_allclose(o["loss_codebook"].item(), f["loss_codebook"], rtol=1e-5)
    np.testing.assert_allclose(z.grad.cpu().numpy().reshape(-1, 24)[ok], f["dz"].reshape(-1, 24)[ok], rtol=1e-4, atol=1e-6)
    assert q.embedding.weight.grad is None                                             # frozen codebook
    rows = q.get_codebook_entry(o["indices"])
    assert torch.equal(rows, o["output"].detach() - 0 * rows) or rel(rows, o["output"].detach()) < 1e-6


def test_sq_full_size_properties():
    """N = 8192 tokens (8 clips x 1024 latents) against all 196 560 codewords: 77 GFLOP of exact fp32, no N x K matrix.
    Properties that hold at any size: the chosen code is the cosine argmax (dense fp32 check on the GPU, chunked), outputs
    are unit codebook rows, the loss equals (beta + 1) x mean squared distance summed over d, repeated calls agree."""
    import video_tokenizer_amd as vt
    W = torch.from_numpy(vt.sq.leech_minimal_vectors()).cuda()
    q = vt.sq.VectorQuantizer(n_embed=196560, embed_dim=24, l2_norm=True, beta=0.25, input_format="blc", predefined_codebook=W.cpu().numpy()).cuda()
    z = torch.from_numpy(gen.normal((8, 1024, 24), 611)).cuda()
    a, b = q(z), q(z)
    torch.cuda.synchronize()
    assert torch.equal(a["indices"], b["indices"]) and torch.equal(a["output"], b["output"])
    idx = a["indices"].reshape(-1)
    zn = torch.nn.functional.normalize(z.reshape(-1, 24), dim=-1)
    assert int(idx.min()) >= 0 and int(idx.max()) < 196560 and len(torch.unique(idx)) > 4000
    chosen = (zn * W[idx]).sum(-1)
    for i in range(0, 8192, 1024):
        best = (zn[i:i + 1024] @ W.t()).max(dim=-1).values
        assert float((best - chosen[i:i + 1024]).max()) < 2e-6
    out = a["output"].reshape(-1, 24)
    assert rel(out, W[idx]) < 1e-6
    msd = ((W[idx] - zn) ** 2).sum(-1).mean()
    np.testing.assert_allclose(a["loss_codebook"].item(), 1.25 * msd.item(), rtol=1e-5)


def test_sq_tokenizer_forward_backward_matches_oracle():
    """end to end: LARPTokenizer(bottleneck_type='sq') vs the oracle's restatement of the same branch, with the full Leech codebook"""
    import video_tokenizer_amd as vt
    cfg = O.make_cfg("tiny", bottleneck_type="sq")
    spec = vt.config.model_spec(cfg)
    spec["args"]["bottleneck_type"] = "sq"
    model = vt.make(spec)
    sd = O.init_state_dict(cfg, seed=17, query_std=1.0)
    sd["bottleneck.embedding.weight"] = model.bottleneck.embedding.weight.detach().clone()
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    x = torch.from_numpy(gen.video_clips(2, cfg["frame_num"], cfg["input_size"], 19))
    w = torch.from_numpy(gen.normal(tuple(x.shape), 20))
    out = model(x.cuda())
    assert set(out.keys()) == {"pred_frames", "encoded", "loss_codebook"}          # larp_tokenizer.py:423-428, 489-496
    ((out["pred_frames"] * w.cuda()).sum() + 0.7 * out["loss_codebook"]).backward()
    torch.cuda.synchronize()
    # oracle on the same weights, following the GPU's discrete indices so every float comparison is like-for-like
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("_pe") and k not in ("decoder_patch_query_embed", "bottleneck.embedding.weight"))
         for k, v in sd.items()}
    idx_gpu = model.last_indices.cpu()
    free = O.tokenizer_forward(sd, cfg, x, emu=True)
    agree = (free["_indices"].reshape(-1) == idx_gpu.reshape(-1)).float().mean().item()
    assert agree >= 0.97, agree                                                     # bf16 GEMMs upstream: only near-ties may flip
    assert len(torch.unique(idx_gpu)) >= 0.25 * idx_gpu.numel()
    ref = O.tokenizer_forward(p, cfg, x, emu=True, force_idx=idx_gpu)
    ((ref["pred_frames"] * w).sum() + 0.7 * ref["loss_codebook"]).backward()
    assert rel(out["pred_frames"].cpu(), ref["pred_frames"].detach()) < 2e-2
    assert rel(out["encoded"].cpu(), ref["encoded"].detach()) < 2e-2
    np.testing.assert_allclose(out["loss_codebook"].item(), ref["loss_codebook"].item(), rtol=2e-2)
    bad = [(n, rel(q.grad.cpu(), p[n].grad)) for n, q in model.named_parameters() if q.requires_grad and rel(q.grad.cpu(), p[n].grad) > 6e-2]
    assert not bad, bad
    assert model.bottleneck.embedding.weight.grad is None


def test_bench_sq_leg_runs():
    """the `bench.py --sq` leg: the step with the bottleneck the shipped yaml carries (cfgs/larp_tokenizer.yaml:73), here at the tiny geometry"""
    import bench
    import video_tokenizer_amd as vt
    cfg = vt.config.geometry("A")
    spec = vt.config.model_spec(cfg, stochastic=True)
    spec["args"]["encoder_depth"] = spec["args"]["decoder_depth"] = 2
    x = torch.from_numpy(vt.config.synthetic_clips(2, cfg["frame_num"], cfg["input_size"], 5)).cuda()
    r = bench.sq_step(vt, cfg, spec, x, steps=2, warmup=1)
    assert r["bottleneck_type"] == "sq" and r["clips_per_s"] > 0 and np.isfinite(r["loss"]) and r["distinct_codes_in_batch"] >= 1
