"""Finite scalar quantizer (csrc/vt_fsq.hip) against oracle/fsq_oracle.c and the reference
class's own outputs (tests/golden/fsq_*.npz).  Indices and codes are integer/lattice work: bit-exact.  GPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import fsq_c
from oracle import inputs as gen
from tests.golden.make_golden import fsq_cases
from tests.test_oracle_golden import _near_rounding_boundary

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def vt():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import video_tokenizer_amd as v
    v.hip.lib()
    return v


@pytest.mark.parametrize("case", fsq_cases())
def test_fsq_forward_backward_bit_exact_vs_oracle_and_reference_vectors(vt, case):
    levels, N, seed = case
    z = gen.normal((N, len(levels)), seed, std=1.5)
    up = gen.normal((N, len(levels)), seed + 1000)
    zc = torch.from_numpy(z).cuda()
    codes, idx = vt.hip.fsq_forward(zc, levels)
    dz = vt.hip.fsq_backward(zc, torch.from_numpy(up).cuda(), levels)
    torch.cuda.synchronize()
    o_codes, o_idx, o_bounded = fsq_c.forward(z, levels)
    # device double-precision tanh vs libm's: both round to the correctly rounded fp32 value => identical lattices;
    # should the two ever disagree by an ulp it can only show on an element sitting on a rounding boundary
    tie = _near_rounding_boundary(o_bounded).any(axis=1)
    assert np.array_equal(idx.cpu().numpy()[~tie], o_idx[~tie])
    assert np.array_equal(codes.cpu().numpy()[~tie], o_codes[~tie])
    assert (idx.cpu().numpy() != o_idx).sum() <= tie.sum()
    np.testing.assert_allclose(dz.cpu().numpy(), fsq_c.backward(z, up, levels), rtol=1e-6, atol=1e-6)
    # the reference class's own outputs on the same bytes
    f = np.load(os.path.join(G, "fsq_" + "x".join(str(v) for v in levels) + ".npz"))
    tie_ref = _near_rounding_boundary(f["bounded"]).any(axis=1)
    assert np.array_equal(idx.cpu().numpy()[~tie_ref], f["indices"][~tie_ref])
    assert np.array_equal(codes.cpu().numpy()[~tie_ref], f["codes"][~tie_ref])
    np.testing.assert_allclose(dz.cpu().numpy(), f["dz"], rtol=2e-6, atol=1e-6)
    back = vt.hip.fsq_indices_to_codes(torch.from_numpy(f["indices"]).cuda(), levels)
    assert np.array_equal(back.cpu().numpy(), f["codes_from_indices"])


def test_fsq_full_size_properties(vt):
    """config-sized batch (B=8 x 1024 latent tokens, levels of autoencoder_large): size-independent properties --
    codes lie on the lattice k / half_width, indices are in range, indices -> codes -> quantize is the identity,
    and re-quantizing atanh-free lattice points is idempotent in index space"""
    levels = [8, 8, 8, 5, 5, 5]
    N = 8 * 1024
    z = torch.from_numpy(gen.normal((N, 6), 77, std=2.0)).cuda()
    codes, idx = vt.hip.fsq_forward(z, levels)
    hw = torch.tensor([v // 2 for v in levels], device="cuda", dtype=torch.float32)
    lvl = codes * hw + hw
    assert torch.equal(lvl, lvl.round()) and int(lvl.min()) >= 0
    assert bool((lvl <= torch.tensor(levels, device="cuda") - 1).all())
    assert int(idx.min()) >= 0 and int(idx.max()) < 64000
    assert torch.equal(vt.hip.fsq_indices_to_codes(idx, levels), codes)
    # every level of every channel is reachable with std-2 inputs, and the checksum of indices is reproducible
    for c, l in enumerate(levels):
        assert lvl[:, c].unique().numel() == l
    o_codes, o_idx, _ = fsq_c.forward(z.cpu().numpy(), levels)
    assert int(idx.sum()) == int(o_idx.astype(np.int64).sum())


def test_fsq_bf16_path_matches_fp32_on_the_same_values(vt):
    """the reference up-casts a bf16 latent to fp32, quantizes, and casts the codes back (fsq.py:122-129)"""
    levels = [8, 8, 8, 8, 5, 5, 5, 5]
    zb = torch.from_numpy(gen.normal((4096, 8), 78, std=1.5)).cuda().to(torch.bfloat16)
    codes_b, idx_b = vt.hip.fsq_forward(zb, levels)
    codes_f, idx_f = vt.hip.fsq_forward(zb.float(), levels)
    assert codes_b.dtype == torch.bfloat16 and torch.equal(idx_b, idx_f)
    assert torch.equal(codes_b, codes_f.to(torch.bfloat16))
    g = torch.from_numpy(gen.normal((4096, 8), 79)).cuda().to(torch.bfloat16)
    dz_b = vt.hip.fsq_backward(zb, g, levels)
    dz_f = vt.hip.fsq_backward(zb.float(), g.float(), levels)
    assert torch.equal(dz_b, dz_f.to(torch.bfloat16))


def test_fsq_module_surface_and_autograd(vt):
    """FSQ(levels) used the way model_new/autoencoder.py does: forward -> (codes, {'indices'}), straight-through
    gradient into the encoder output, indices_to_codes for decode_indices; leading dims are free"""
    q = vt.FSQ(levels=[8, 8, 8, 5, 5, 5]).cuda()
    assert q.codebook_size == 64000 and q.codebook_dim == 6 and len(list(q.parameters())) == 0 and len(q.state_dict()) == 0
    z = torch.from_numpy(gen.normal((2, 1024, 6), 80, std=1.5)).cuda().requires_grad_(True)
    codes, info = q(z)
    assert codes.shape == z.shape and info["indices"].shape == (2, 1024) and info["indices"].dtype == torch.int32
    up = torch.from_numpy(gen.normal((2, 1024, 6), 81)).cuda()
    (codes * up).sum().backward()
    ref = fsq_c.backward(z.detach().cpu().numpy().reshape(-1, 6), up.cpu().numpy().reshape(-1, 6), [8, 8, 8, 5, 5, 5])
    np.testing.assert_allclose(z.grad.cpu().numpy().reshape(-1, 6), ref, rtol=1e-6, atol=1e-6)
    assert torch.equal(q.indices_to_codes(info["indices"]), codes.detach())
    assert torch.equal(q.codes_to_indices(codes.detach()), info["indices"])
    lv = q.indices_to_level_indices(info["indices"])
    assert lv.shape == (2, 1024, 6) and int(lv.max()) == 7
    assert torch.equal(q.quantize(z.detach()), codes.detach())
    # bf16 latent under autocast-style use
    cb, ib = q(z.detach().to(torch.bfloat16))
    assert cb.dtype == torch.bfloat16 and ib["indices"].shape == (2, 1024)


def test_fsq_bad_arguments_are_refused(vt):
    z = torch.zeros(16, 3, device="cuda")
    with pytest.raises(vt.hip.HipError):
        vt.hip.fsq_forward(z, [8, 1, 5])                     # a level below 2
    with pytest.raises(vt.hip.HipError):
        vt.hip.fsq_forward(torch.zeros(16, 9, device="cuda"), [8] * 9)   # 8^9 > 2^24: fp32 index sum would be inexact
    with pytest.raises(vt.hip.HipError):
        vt.hip.fsq_forward(torch.zeros(16, 17, device="cuda"), [2] * 17)  # d > 16
    with pytest.raises(TypeError):
        vt.hip.fsq_forward(z.half(), [8, 8, 5])
    with pytest.raises(vt.hip.HipError):
        vt.hip.fsq_forward(torch.zeros(16, 3), [8, 8, 5])    # CPU tensor: no CPU path
