"""Op-level parity of every HIP kernel behind include/vt_hip.h against the CPU oracle / plain
fp32 math on the same seeded inputs.  All calls go through the C ABI (ctypes).  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import inputs as gen
from oracle import larp_oracle as O
from oracle import vq_c
from tests.golden.make_golden import vq_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import video_tokenizer_amd.hip as h
    h.lib()
    return h


def bf(a):
    return torch.from_numpy(a).to(torch.bfloat16)


def _rand(shape, seed, std=1.0):
    return gen.normal(shape, seed, std)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------------------------------- GEMM
@pytest.fixture(params=[1, 2, 5, 6], ids=["tile128", "tile192", "tile192x96", "tile192_one_tile_per_wg"], autouse=False)
def gemm_variant(request, hip):
    """run a GEMM test once per tile generation (hip.GEMM_TILE -> vtGemmNT.tile), then restore auto dispatch"""
    hip.GEMM_TILE = request.param
    yield request.param
    hip.GEMM_TILE = 0


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (256, 24, 768), (130, 1000, 192)])
def test_gemm_nt_bf16_and_bias(hip, M, N, K, gemm_variant):
    A, B = bf(_rand((M, K), 1)), bf(_rand((N, K), 2))
    bias = torch.from_numpy(_rand((N,), 3))
    ref = A.float() @ B.float().t() + bias
    out = hip.gemm_nt(A.cuda(), B.cuda(), hip.EPI_BF16, bias=bias.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.to(torch.bfloat16).float().numpy(), rtol=2e-2, atol=2e-2)


def test_gemm_nt_exact_integers_asymmetric(hip, gemm_variant):
    """A = I (padded), asymmetric integer B: catches a transposed or permuted C write exactly."""
    M, N, K = 128, 256, 128
    A = torch.zeros(M, K)
    A[torch.arange(M), torch.arange(M) % K] = 1.0
    B = (torch.arange(N).reshape(N, 1) * 3 + torch.arange(K).reshape(1, K) % 7).float() % 64
    ref = A @ B.t()
    out = hip.gemm_nt(A.to(torch.bfloat16).cuda(), B.to(torch.bfloat16).cuda(), hip.EPI_F32)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)


def test_gemm_nt_gelu_and_dgelu(hip, gemm_variant):
    M, N, K = 200, 320, 128
    A, B = bf(_rand((M, K), 4, 0.5)), bf(_rand((N, K), 5, 0.5))
    bias = torch.from_numpy(_rand((N,), 6, 0.1))
    pre = (A.float() @ B.float().t() + bias).to(torch.bfloat16)
    u, g = hip.gemm_nt(A.cuda(), B.cuda(), hip.EPI_BF16_GELU, bias=bias.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(u.float().cpu().numpy(), pre.float().numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(g.float().cpu().numpy(), O.gelu_erf(u.float().cpu()).to(torch.bfloat16).float().numpy(), rtol=2e-2, atol=1e-2)
    # dgelu: out = (A @ B^T) * gelu'(aux)
    aux = bf(_rand((M, N), 7))
    xr = aux.float().requires_grad_(True)
    O.gelu_erf(xr).sum().backward()
    ref = (A.float() @ B.float().t()) * xr.grad
    out = hip.gemm_nt(A.cuda(), B.cuda(), hip.EPI_BF16_DGELU, aux=aux.cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=2e-2, atol=3e-2)


def test_gemm_nt_f32_residual_rowmod_rowmap(hip, gemm_variant):
    Bt, n, seq, off, N, K = 3, 40, 100, 60, 256, 128
    M = Bt * n
    A, B = bf(_rand((M, K), 8)), bf(_rand((N, K), 9))
    bias = torch.from_numpy(_rand((N,), 10))
    res = torch.from_numpy(_rand((Bt * seq, N), 11))
    pe = torch.from_numpy(_rand((n, N), 12))
    out = torch.full((Bt * seq, N), 7.0)
    out_gpu, out2 = out.cuda(), torch.zeros(Bt * seq, N, dtype=torch.bfloat16).cuda()
    hip.gemm_nt(A.cuda(), B.cuda(), hip.EPI_F32, bias=bias.cuda(), out=out_gpu, out2=out2, residual=res.cuda(),
                rowmod=pe.cuda(), rowmod_period=n, omap=hip.RowMap(n, seq, off), round_bf16=True)
    torch.cuda.synchronize()
    core = (A.float() @ B.float().t() + bias).to(torch.bfloat16).float()
    ref = out.clone()
    for b in range(Bt):
        rows = slice(b * seq + off, b * seq + off + n)
        ref[rows] = core[b * n:(b + 1) * n] + res[rows] + pe
    got = out_gpu.cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-2, atol=3e-2)
    assert torch.equal(got[:off], out[:off])  # untouched rows stay untouched
    np.testing.assert_allclose(out2.float().cpu().numpy()[off:off + n], ref.to(torch.bfloat16).float().numpy()[off:off + n], rtol=1e-2, atol=3e-2)


@pytest.mark.parametrize("M", [1, 7, 16, 33, 64])
@pytest.mark.parametrize("N,K", [(16, 64), (100, 384), (768, 2048), (2304, 768), (8192, 1280)])
def test_gemm_nt_skinny_matches_tile_kernel_and_torch(hip, M, N, K):
    """M <= 64 dispatches to the weight-streaming kernel (tile 7): every epilogue against the 128-row tile kernel and fp32 torch"""
    A, B = bf(_rand((M, K), 81 + M, 0.5)).cuda(), bf(_rand((N, K), 82 + N, 0.5)).cuda()
    ref = A.float() @ B.float().t()
    Np = (N + 3) // 4 * 4
    bias = torch.from_numpy(_rand((N,), 83, 0.2)).cuda()
    res = torch.from_numpy(_rand((M, Np), 84)).cuda()
    # integers: exact, any summation order
    Ai = torch.from_numpy((np.arange(M * K).reshape(M, K) % 5 - 2).astype(np.float32)).to(torch.bfloat16).cuda()
    Bi = torch.from_numpy((np.arange(N * K).reshape(N, K) % 7 - 3).astype(np.float32)).to(torch.bfloat16).cuda()
    assert torch.equal(hip.gemm_nt(Ai, Bi, hip.EPI_F32, out=torch.empty(M, Np, device="cuda"), tile=7)[:, :N], Ai.float() @ Bi.float().t())
    for tile in (0, 7):     # 0 = automatic: must pick the same kernel
        o_bf = hip.gemm_nt(A, B, hip.EPI_BF16, bias=bias, out=torch.empty(M, Np, device="cuda", dtype=torch.bfloat16), tile=tile)[:, :N]
        o_f = hip.gemm_nt(A, B, hip.EPI_F32, out=torch.full((M, Np), 3.0, device="cuda"), residual=res, round_bf16=True, tile=tile)
        assert rel(o_bf, (ref + bias).to(torch.bfloat16)) < 4e-3
        assert rel(o_f[:, :N], ref.to(torch.bfloat16).float() + res[:, :N]) < 2e-3
        assert torch.equal(o_f[:, N:], torch.full((M, Np - N), 3.0, device="cuda"))          # pad columns untouched
    t_bf = hip.gemm_nt(A, B, hip.EPI_BF16, bias=bias, out=torch.empty(M, Np, device="cuda", dtype=torch.bfloat16), tile=1)[:, :N]
    assert rel(o_bf, t_bf) < 3e-3
    if N % 4 == 0:
        u, g = hip.gemm_nt(A, B, hip.EPI_BF16_GELU, bias=bias, tile=7)
        u1, g1 = hip.gemm_nt(A, B, hip.EPI_BF16_GELU, bias=bias, tile=1)
        assert rel(u, u1) < 3e-3 and rel(g, g1) < 4e-3


def test_gemm_tn_grouped(hip, gemm_variant):
    M = 256
    dY1, X1 = bf(_rand((M, 192), 13)), bf(_rand((M, 136), 14))
    dY2, X2 = bf(_rand((M, 64), 15)), bf(_rand((M, 320), 16))
    out1 = torch.zeros(192, 136).cuda()
    out2 = torch.full((40, 24), -5.0).cuda()
    perm = torch.randperm(24, generator=torch.Generator().manual_seed(0)).to(torch.int32)
    hip.gemm_tn_grouped([dict(A=dY1.cuda(), B=X1.cuda(), out=out1),
                         dict(A=dY2.cuda(), B=X2.cuda(), out=out2, p_lim=24, q_lim=24, row_perm=perm.cuda())])
    torch.cuda.synchronize()
    ref1 = dY1.float().t() @ X1.float()
    np.testing.assert_allclose(out1.cpu().numpy(), ref1.numpy(), rtol=1e-3, atol=1e-2)
    ref2 = (dY2.float().t() @ X2.float())[:24, :24]
    got2 = out2.cpu()
    np.testing.assert_allclose(got2[perm.long()].numpy()[:, :24], ref2.numpy(), rtol=1e-3, atol=1e-2)
    assert torch.all(got2[24:] == -5.0)


@pytest.mark.parametrize("M,N,K", [(384, 384, 512), (200, 400, 320)])
def test_gemm_nt_exact_integers_multi_ktile(hip, gemm_variant, M, N, K):
    """small-integer operands: every product and partial sum is exact in fp32, so the result must be bit-exact
    whatever the tile size, pipeline depth or LDS ring position (catches stale-buffer races)."""
    g = torch.Generator().manual_seed(3)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    for _ in range(3):
        out = hip.gemm_nt(A.to(torch.bfloat16).cuda(), B.to(torch.bfloat16).cuda(), hip.EPI_F32)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), A @ B.t())


@pytest.mark.parametrize("M,P,Q", [(512, 384, 192), (320, 200, 392)])
def test_gemm_tn_exact_integers_multi_ktile(hip, gemm_variant, M, P, Q):
    g = torch.Generator().manual_seed(4)
    dY = torch.randint(-3, 4, (M, P), generator=g).float()
    X = torch.randint(-3, 4, (M, Q), generator=g).float()
    for _ in range(3):
        out = torch.zeros(P, Q).cuda()
        hip.gemm_tn_grouped([dict(A=dY.to(torch.bfloat16).cuda(), B=X.to(torch.bfloat16).cuda(), out=out)])
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), dY.t() @ X)


@pytest.mark.parametrize("M,P,Q", [(64, 192, 192), (128, 384, 200), (192, 200, 392), (256, 192, 576), (448, 768, 392), (1024, 392, 768)])
def test_gemm_tn_register_pipeline_is_bit_identical_to_the_burst_kernel(hip, M, P, Q):
    """Round 4: gemm_tn192p_kernel (fragments of K-tile t+1 read into a second register set while tile t is multiplied, LDS-DMA three
    tiles ahead) against the round-1 kernel it replaces (vtGemmTN.tile = 7): same MFMA order per accumulator, so the same bits, for
    1 ... 16 K-tiles (ring wrap-around, both register sets as the last one), ragged P / Q, limits and a row permutation; and exact on
    small integers (a stale LDS slot or a fragment read before its DMA landed would show)."""
    g = torch.Generator().manual_seed(M + P)
    dY = (torch.randn(M, P, generator=g)).to(torch.bfloat16).cuda()
    X = (torch.randn(M, Q, generator=g)).to(torch.bfloat16).cuda()
    perm = torch.randperm(P - 8, generator=g).to(torch.int32).cuda()
    outs = {}
    for tile in (7, 2):
        o1 = torch.full((P, Q), -3.0).cuda()
        o2 = torch.full((P, Q), -3.0).cuda()
        hip.gemm_tn_grouped([dict(A=dY, B=X, out=o1, tile=tile), dict(A=dY, B=X, out=o2, p_lim=P - 8, q_lim=Q - 8, row_perm=perm, tile=tile)])
        torch.cuda.synchronize()
        outs[tile] = (o1, o2)
    assert torch.equal(outs[7][0], outs[2][0]) and torch.equal(outs[7][1], outs[2][1])
    ref = dY.float().t() @ X.float()
    assert (outs[2][0] - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert torch.all(outs[2][1][P - 8:] == -3.0) and torch.all(outs[2][1][:, Q - 8:] == -3.0)
    di = torch.randint(-3, 4, (M, P), generator=g).float()
    xi = torch.randint(-3, 4, (M, Q), generator=g).float()
    for _ in range(2):
        o = torch.zeros(P, Q).cuda()
        hip.gemm_tn_grouped([dict(A=di.to(torch.bfloat16).cuda(), B=xi.to(torch.bfloat16).cuda(), out=o, tile=2)])
        torch.cuda.synchronize()
        assert torch.equal(o.cpu(), di.t() @ xi)


def test_gemm_tn_exact_integers(hip, gemm_variant):
    M, P, Q = 128, 128, 128
    dY = ((torch.arange(M).reshape(M, 1) * 5 + torch.arange(P).reshape(1, P) * 3) % 9).float()
    X = ((torch.arange(M).reshape(M, 1) + torch.arange(Q).reshape(1, Q) * 7) % 5).float()
    out = torch.zeros(P, Q).cuda()
    hip.gemm_tn_grouped([dict(A=dY.to(torch.bfloat16).cuda(), B=X.to(torch.bfloat16).cuda(), out=out)])
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), dY.t() @ X)


# ------------------------------------------------------------------------------------- row kernels
@pytest.mark.parametrize("dim", [768, 384, 128, 1024])   # 384 = the discriminator's width (8-B-per-lane variant)
def test_layernorm_fwd_bwd(hip, dim):
    rows = 200
    x = torch.from_numpy(_rand((rows, dim), 20)) * 2 + 0.3
    g = torch.from_numpy(gen.uniform((dim,), 21, 0.5, 1.5))
    b = torch.from_numpy(_rand((dim,), 22, 0.1))
    y, mean, rstd = hip.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), 1e-5)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (dim,), gr, br, 1e-5)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.detach().to(torch.bfloat16).float().numpy(), rtol=1e-2, atol=1e-2)
    dy = bf(_rand((rows, dim), 23))
    dres = torch.from_numpy(_rand((rows, dim), 24))
    ref.backward(dy.float())
    dx, dxb, dg, db, ds = hip.layernorm_bwd(dy.cuda(), x.cuda(), g.cuda(), mean, rstd, dres=dres.cuda())
    torch.cuda.synchronize()
    want = xr.grad + dres
    np.testing.assert_allclose(dx.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dxb.float().cpu().numpy(), want.to(torch.bfloat16).float().numpy(), rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(ds.cpu().numpy(), want.sum(0).numpy(), rtol=1e-4, atol=1e-3)


def test_layernorm_rowmap(hip):
    Bt, seq, off, n, dim = 2, 50, 30, 20, 768
    x = torch.from_numpy(_rand((Bt * seq, dim), 25))
    g, b = torch.ones(dim), torch.zeros(dim)
    y, mean, rstd = hip.layernorm_fwd(x.cuda(), g.cuda(), b.cuda(), 1e-6, rows=Bt * n, xmap=hip.RowMap(n, seq, off))
    torch.cuda.synchronize()
    sel = torch.cat([x[bb * seq + off: bb * seq + off + n] for bb in range(Bt)])
    np.testing.assert_allclose(y.float().cpu().numpy(), F.layer_norm(sel, (dim,), eps=1e-6).to(torch.bfloat16).float().numpy(), rtol=1e-2, atol=1e-2)


def test_colsum_batchsum_cast_assemble_pack(hip):
    rows, width = 1000, 3072
    s = bf(_rand((rows, width), 30))
    np.testing.assert_allclose(hip.colsum(s.cuda()).cpu().numpy(), s.float().sum(0).numpy(), rtol=1e-4, atol=1e-2)
    f = torch.from_numpy(_rand((6 * 50, 768), 31))
    m = hip.RowMap(20, 50, 30)
    sel = torch.stack([f[bb * 50 + 30: bb * 50 + 50] for bb in range(6)])
    np.testing.assert_allclose(hip.colsum(f.cuda(), rows=120, rmap=m).cpu().numpy(), sel.reshape(-1, 768).sum(0).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(hip.batch_sum(f.cuda(), 6, 20, rmap=m).cpu().numpy(), sel.sum(0).numpy(), rtol=1e-5, atol=1e-4)
    c = hip.cast_rows(f.cuda(), rows=120, rmap=m)
    assert torch.equal(c.cpu(), sel.reshape(-1, 768).to(torch.bfloat16))
    dst = torch.zeros(6 * 50, 768).cuda()
    src = torch.from_numpy(_rand((6 * 20, 768), 32))
    tab = torch.from_numpy(_rand((20, 768), 33))
    vec = torch.from_numpy(_rand((768,), 34))
    hip.assemble_rows(dst, 50, 30, 6, 20, src=src.cuda(), table=tab.cuda(), vec=vec.cuda())
    torch.cuda.synchronize()
    want = torch.zeros(6, 50, 768)
    want[:, 30:] = src.reshape(6, 20, 768) + tab + vec
    np.testing.assert_allclose(dst.cpu().numpy(), want.reshape(-1, 768).numpy(), rtol=1e-6, atol=1e-6)
    w = torch.from_numpy(_rand((100, 72), 35))
    perm = torch.randperm(100, generator=torch.Generator().manual_seed(1)).to(torch.int32)
    wb, wt = hip.pack_weight(w.cuda(), row_perm=perm.cuda(), n_pad=128, k_pad=128)
    torch.cuda.synchronize()
    wp = w[perm.long()].to(torch.bfloat16)
    assert torch.equal(wb.cpu()[:100, :72], wp) and torch.all(wb.cpu()[100:] == 0) and torch.all(wb.cpu()[:, 72:] == 0)
    assert torch.equal(wt.cpu()[:72, :100], wp.t())


@pytest.mark.parametrize("pt,p,T,S", [(2, 16, 4, 32), (4, 8, 4, 32), (2, 16, 16, 128)])
def test_patchify_unpatchify(hip, pt, p, T, S):
    B = 2
    x = torch.from_numpy(gen.video_clips(B, T, S, 40))
    rows = hip.patchify(x.cuda(), pt, p)
    torch.cuda.synchronize()
    ref = O.patchify(x, pt, p).reshape(-1, 3 * pt * p * p)
    assert torch.equal(rows.cpu(), ref.to(torch.bfloat16))
    # unpatchify takes (c,dt,dy,dx)-ordered rows; the reference head emits (dt,dy,dx,c): permute columns
    kp = 3 * pt * p * p
    y = torch.from_numpy(_rand((B * rows.shape[0] // B, kp), 41))
    perm = torch.arange(kp).reshape(pt, p, p, 3).permute(3, 0, 1, 2).reshape(-1)  # packed col j <- reference col perm[j]
    want = O.unpatchify(y.reshape(B, -1, kp), pt, p, S // p)
    got = hip.unpatchify(y[:, perm].contiguous().cuda(), B, 3, T, S, pt, p)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("M,N,K", [(1536, 768, 3072), (1536, 2304, 768), (3072, 768, 768), (1536, 768, 256), (200, 392, 448), (128, 128, 64)])
def test_gemm_nt_deep_ring_is_bit_identical_to_the_two_stage_kernel(hip, M, N, K):
    """vtGemmNT.tile = 16: the 128x128 kernel behind a 4-deep LDS ring with counted waits (picked automatically when the launch has at
    most one workgroup per CU: one or two clips per GPU).  Same MFMA order as the 2-deep ring (tile 1) => every epilogue's output must be
    bit-identical, for K from 1 to 48 K-tiles, ragged M / N, and the automatic choice must be one of the two."""
    A = bf(_rand((M, K), 700 + K))
    B = bf(_rand((N, K), 701 + N))
    bias = torch.from_numpy(_rand((N,), 702)).cuda()
    res = torch.from_numpy(_rand((M, N), 703)).cuda()
    u = bf(_rand((M, N), 704)).cuda()
    a, b = A.cuda(), B.cuda()
    for kw in (dict(epi=hip.EPI_BF16, bias=bias), dict(epi=hip.EPI_F32, bias=bias, residual=res, round_bf16=True), dict(epi=hip.EPI_BF16_GELU, bias=bias),
               dict(epi=hip.EPI_BF16_DGELU, aux=u)):
        two = hip.gemm_nt(a, b, tile=1, splitk=1, **kw)
        deep = hip.gemm_nt(a, b, tile=16, splitk=1, **kw)
        auto = hip.gemm_nt(a, b, tile=0, splitk=1, **kw)
        torch.cuda.synchronize()
        for x, y, z in zip(*(t if isinstance(t, tuple) else (t,) for t in (two, deep, auto))):
            assert torch.equal(x, y), kw["epi"]
            if N >= 192 and M >= 192 and kw["epi"] != hip.EPI_BF16_DGELU:
                assert torch.equal(x, z), kw["epi"]       # 192-tile kernels accumulate K in the same order too
    ref = (A.float() @ B.float().t() + bias.cpu()).to(torch.bfloat16)
    got = hip.gemm_nt(a, b, tile=16, epi=hip.EPI_BF16, bias=bias).cpu()
    assert (got.float() - ref.float()).abs().max() <= 2e-2 * max(1.0, ref.float().abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,split", [(1536, 768, 3072, 0), (1536, 768, 3072, 7), (1536, 768, 2304, 4), (1536, 768, 768, 3), (3072, 768, 3072, 0),
                                          (200, 136, 448, 3), (128, 128, 128, 2), (130, 260, 1024, 8), (1536, 768, 3072, 2)])
def test_gemm_nt_split_k_is_deterministic_and_close_to_the_unsplit_kernel(hip, M, N, K, split):
    """vtGemmNT.splitk_ws: launches of the 128x128 kernel that leave most CUs idle (one / two clips per GPU: N = 768, 72 or 144 tiles) give each
    tile to several workgroups over contiguous K shares; the last to arrive adds the fp32 partial sums in split order and runs the
    epilogue.  Checked: every epilogue, forced splits 2..8 and the automatic rule (split 0), uneven K shares (7 K-tiles over 3), ragged
    M / N; two runs bit-identical (no atomics in the sum); against the unsplit kernel only the fp32 summation order differs; the arrival
    counters are back to zero; the automatic rule does split the one-clip shapes and leaves the small ragged ones alone."""
    A = bf(_rand((M, K), 800 + K))
    B = bf(_rand((N, K), 801 + N))
    bias = torch.from_numpy(_rand((N,), 802)).cuda()
    res = torch.from_numpy(_rand((M, N), 803)).cuda()
    u = bf(_rand((M, N), 804)).cuda()
    a, b = A.cuda(), B.cuda()
    ws = hip.splitk_workspace(a.device)
    for kw in (dict(epi=hip.EPI_BF16, bias=bias), dict(epi=hip.EPI_F32, bias=bias, residual=res), dict(epi=hip.EPI_BF16_GELU, bias=bias),
               dict(epi=hip.EPI_BF16_DGELU, aux=u)):
        plain = hip.gemm_nt(a, b, splitk=1, **kw)
        one = hip.gemm_nt(a, b, splitk=split or None, **kw)
        one = tuple(t.clone() for t in one) if isinstance(one, tuple) else one.clone()
        two = hip.gemm_nt(a, b, splitk=split or None, **kw)
        torch.cuda.synchronize()
        assert int(ws[:4096].view(torch.int32).abs().sum()) == 0
        for x, y, z in zip(*(t if isinstance(t, tuple) else (t,) for t in (plain, one, two))):
            assert torch.equal(y, z), kw["epi"]
            scale = max(1.0, x.float().abs().max().item())
            tol = 1e-5 if x.dtype == torch.float32 else 2.0 ** -7          # fp32 order / one bf16 rounding step
            assert (x.float() - y.float()).abs().max().item() <= tol * scale, (kw["epi"], (x.float() - y.float()).abs().max().item())
            if kw["epi"] == hip.EPI_F32 and M >= 1536:
                assert not torch.equal(x, y), "the split did not happen (same bits as the unsplit kernel)"
    ref = (A.float() @ B.float().t() + bias.cpu()).to(torch.bfloat16)
    got = hip.gemm_nt(a, b, epi=hip.EPI_BF16, bias=bias, splitk=split or None).cpu()
    assert (got.float() - ref.float()).abs().max() <= 2e-2 * max(1.0, ref.float().abs().max().item())


@pytest.mark.gpu
def test_gemm_nt_split_k_hand_off_under_uneven_concurrent_load(hip):
    """The split-K hand-off (write-through partial tiles, an arrival counter, the last arriver reads every partial back past its L2) is the one
    cross-workgroup exchange inside the tokenizer's backward at one or two clips per GPU -- and under data parallelism it runs NEXT TO the
    side stream's weight-gradient launches and the collective, not alone on the chip (the round-4 record has one unexplained gradient
    mismatch in exactly that setting).  The CDNA4 guide's rule for such exchanges: test them under UNEVEN load, checking every word.  Here
    60 launches of the one-clip shapes run while a second stream keeps part of the chip busy with copies and a grouped weight-gradient GEMM
    of changing size; every output must equal the first quiet run bit for bit and the counters must be back at zero."""
    torch.manual_seed(0)
    side = torch.cuda.Stream()
    big_src = torch.randn(64 << 20, device="cuda")
    big_dst = torch.empty_like(big_src)
    dy = bf(_rand((1536, 768), 71)).cuda()
    xx = bf(_rand((1536, 3072), 72)).cuda()
    wg = torch.empty(768, 3072, device="cuda")
    ws = hip.splitk_workspace(torch.device("cuda", 0))
    shapes = [(128, 768, 3072, hip.EPI_BF16), (1536, 768, 3072, hip.EPI_F32), (1536, 768, 2304, hip.EPI_BF16), (3072, 768, 3072, hip.EPI_BF16)]
    for M, N, K, epi in shapes:
        a, b = bf(_rand((M, K), 900 + M)).cuda(), bf(_rand((N, K), 901 + K)).cuda()
        quiet = hip.gemm_nt(a, b, epi=epi, splitk=None).clone()
        torch.cuda.synchronize()
        outs = [torch.empty_like(quiet) for _ in range(60)]
        go = torch.cuda.Event()
        go.record()
        side.wait_event(go)
        with torch.cuda.stream(side):
            for i in range(12):                      # uneven: bursts of HBM traffic, then a matrix-bound launch, then nothing
                big_dst[: (8 << 20) * (1 + i % 4)].copy_(big_src[: (8 << 20) * (1 + i % 4)])
                if i % 3 == 0:
                    hip.gemm_tn_grouped([dict(A=dy, B=xx, out=wg, p_lim=768, q_lim=3072)])
        for o in outs:
            hip.gemm_nt(a, b, epi=epi, splitk=None, out=o)
        torch.cuda.synchronize()
        diff = [i for i, o in enumerate(outs) if not torch.equal(o, quiet)]
        assert not diff, (M, N, K, epi, diff[:8], float((outs[diff[0]].float() - quiet.float()).abs().max()))
        assert int(ws[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.gpu
def test_gemm_nt192_gelu_table_equals_the_arithmetic_on_every_bf16_magnitude(hip):
    """The 192x192 kernel's fc1 epilogue looks gelu(u) up in an LDS table of the bf16 patterns with |u| in [2^-16, 16) and falls back to
    the arithmetic for a 4-column group that holds anything else; the 128x128 kernel keeps the arithmetic.  B = identity rows, so u is
    exactly A: every bf16 exponent from 2^-40 to 2^20 in both signs, zeros, and the table's edge patterns land in the epilogue --
    gelu(u) of the two kernels must be bit-identical (and u itself too)."""
    import numpy as np
    M, K = 384, 192
    rng = np.random.default_rng(5)
    exps = rng.integers(-40, 21, size=(M, K))
    vals = np.ldexp(1.0 + rng.integers(0, 128, size=(M, K)) / 128.0, exps) * rng.choice([-1.0, 1.0], size=(M, K))
    vals[::7, ::5] = 0.0
    vals[1::7, ::5] = -0.0
    vals[0, :8] = [2.0 ** -16, -2.0 ** -16, 2.0 ** -17 * 1.9921875, 15.9375, -15.9375, 16.0, -16.0, 2.0 ** -16 * 1.0078125]   # the table's edges
    vals[2:6] = rng.normal(size=(4, K))          # rows that stay inside the table
    a = torch.from_numpy(vals.astype(np.float32)).to(torch.bfloat16).cuda()
    b = torch.eye(K, dtype=torch.bfloat16).cuda()
    u128, g128 = hip.gemm_nt(a, b, epi=hip.EPI_BF16_GELU, tile=1, splitk=1)
    u192, g192 = hip.gemm_nt(a, b, epi=hip.EPI_BF16_GELU, tile=2)
    torch.cuda.synchronize()
    assert torch.equal(u192, a) and torch.equal(u128, a)
    assert torch.equal(g128.view(torch.int16), g192.view(torch.int16))
    ref = torch.nn.functional.gelu(a.float()).to(torch.bfloat16)
    # against torch: one bf16 rounding step, plus the arithmetic's own absolute error (Abramowitz-Stegun 7.1.26: 1.5e-7 on erf, times |u|)
    assert ((g192.float() - ref.float()).abs() <= 2.0 ** -7 * ref.float().abs() + 4e-7 * a.float().abs().clamp(min=1.0)).all()


# -------------------------------------------------------------------------------------- attention
def _attn_ref(qkv, B, L, H, hd=64):
    q, k, v = qkv.reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    att = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    return (att @ v).transpose(1, 2).reshape(B * L, H * hd)


@pytest.mark.gpu
@pytest.mark.parametrize("L,H,hd,Bbig", [(1536, 12, 64, 8), (256, 12, 32, 24), (320, 3, 64, 64)])
def test_attention_of_a_clip_does_not_depend_on_its_batch(hip, L, H, hd, Bbig):
    """Sequences are independent through the three attention kernels and nothing in them depends on the launch size: outputs, log-sum-exps
    and gradients of a sequence must be bit-identical whether it is attended alone (144 workgroups at the step's shape) or inside a large
    batch -- the property the engine's batch-independent forward rests on, and the check that held a four-deep K/V ring for small launches
    to its claim (tools/attn_small_batch.py: bit-identical, and no faster -- 25.3 vs 25.4 us forward, 65.4 vs 61.8 us backward at one
    clip: a lone wave per SIMD is bound by its own S -> softmax -> P.V sequence, not by the loads; the ring was not kept)."""
    qkv = bf(_rand((Bbig * L, 3 * H * hd), 900 + L)).cuda()
    dO = bf(_rand((Bbig * L, H * hd), 901 + L)).cuda()
    o_big, lse_big = hip.attention_fwd(qkv, Bbig, L, H, hd=hd)
    d_big = hip.attention_bwd(qkv, o_big, dO, lse_big, Bbig, L, H, hd=hd)
    for b in (0, Bbig - 1):
        rows = slice(b * L, (b + 1) * L)
        q1 = qkv[rows].contiguous()
        o1, lse1 = hip.attention_fwd(q1, 1, L, H, hd=hd)
        d1 = hip.attention_bwd(q1, o1, dO[rows].contiguous(), lse1, 1, L, H, hd=hd)
        torch.cuda.synchronize()
        assert torch.equal(o1, o_big[rows]), b
        assert torch.equal(lse1.reshape(-1), lse_big.reshape(Bbig, -1)[b]), b
        assert torch.equal(d1, d_big[rows]), b                     # dQ | dK | dV: the backward kernels' rings are chosen the same way


# head_dim 32 + odd L: the GAN discriminator's attention (loss.py:119-204: 12 heads of 32, cls token => L = 1025)
@pytest.mark.parametrize("B,L,H,hd", [(1, 64, 1, 64), (2, 192, 3, 64), (1, 100, 2, 64), (1, 333, 1, 64),
                                      (1, 64, 1, 32), (2, 192, 3, 32), (1, 129, 2, 32), (1, 333, 4, 32)])
def test_attention_fwd_bwd(hip, B, L, H, hd):
    qkv = bf(_rand((B * L, 3 * H * hd), 50 + L))
    dO = bf(_rand((B * L, H * hd), 51 + L))
    x = qkv.float().requires_grad_(True)
    ref = _attn_ref(x, B, L, H, hd)
    ref.backward(dO.float())
    o, lse2 = hip.attention_fwd(qkv.cuda(), B, L, H, hd)
    torch.cuda.synchronize()
    np.testing.assert_allclose(o.float().cpu().numpy(), ref.detach().numpy(), rtol=2e-2, atol=2e-2)
    q, k, _ = qkv.float().reshape(B, L, 3, H, hd).permute(2, 0, 3, 1, 4)
    lse_ref = torch.logsumexp((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1) / np.log(2.0)
    np.testing.assert_allclose(lse2.cpu().numpy(), lse_ref.numpy(), rtol=1e-3, atol=1e-2)
    dqkv = hip.attention_bwd(qkv.cuda(), o, dO.cuda(), lse2, B, L, H, hd)
    torch.cuda.synchronize()
    g = x.grad
    err = (dqkv.float().cpu() - g).abs().max() / g.abs().max()
    assert err < 3e-2, err


@pytest.mark.parametrize("B,L,H,hd,q_begin", [(2, 192, 2, 64, 64), (1, 320, 3, 64, 128), (2, 200, 2, 32, 64), (1, 1536, 1, 64, 1024)])
def test_attention_kept_query_suffix(hip, B, L, H, hd, q_begin):
    """vt_attention_*_rows: queries q_begin..L-1 only (last block of a stack), compact o / dO, full-length keys and dqkv"""
    qkv = bf(_rand((B * L, 3 * H * hd), 150 + L))
    Lq = L - q_begin
    dOc = bf(_rand((B * Lq, H * hd), 151 + L))
    x = qkv.float().requires_grad_(True)
    full = _attn_ref(x, B, L, H, hd).reshape(B, L, H * hd)[:, q_begin:].reshape(B * Lq, H * hd)
    full.backward(dOc.float())
    o, lse2 = hip.attention_fwd(qkv.cuda(), B, L, H, hd, q_begin=q_begin)
    torch.cuda.synchronize()
    assert o.shape == (B * Lq, H * hd)
    np.testing.assert_allclose(o.float().cpu().numpy(), full.detach().numpy(), rtol=2e-2, atol=2e-2)
    # same kernels as the full call: the kept rows are bit-equal to it
    o_all, lse_all = hip.attention_fwd(qkv.cuda(), B, L, H, hd)
    assert torch.equal(o, o_all.reshape(B, L, H * hd)[:, q_begin:].reshape(B * Lq, H * hd))
    assert torch.equal(lse2[:, :, q_begin:], lse_all[:, :, q_begin:])
    dqkv = hip.attention_bwd(qkv.cuda(), o, dOc.cuda(), lse2, B, L, H, hd, q_begin=q_begin)
    torch.cuda.synchronize()
    g = x.grad
    assert (dqkv.float().cpu() - g).abs().max() / g.abs().max() < 3e-2
    dq_rows = dqkv.reshape(B, L, 3, H * hd)[:, :q_begin, 0]
    assert torch.all(dq_rows == 0)                                   # queries before q_begin: exactly zero gradient


def _attn_grad_ref_cuda(qkv, dO, B, L, H, q_begin):
    """fp32 torch math on the device (the checker, not the product): gradient of softmax(q k^T / 8) v wrt the packed projection"""
    x = qkv.float().cuda().requires_grad_(True)
    Lq = L - q_begin
    out = _attn_ref(x, B, L, H, 64).reshape(B, L, H * 64)[:, q_begin:].reshape(B * Lq, H * 64)
    out.backward(dO.float().cuda())
    return x.grad


@pytest.mark.parametrize("hd", [64, 32])
def test_attention_integer_identity(hip, hd):
    """One-hot V columns + peaked scores: checks the transposed-read PV product element by element."""
    B, L, H = 1, 64, 1
    q = torch.zeros(L, hd)
    k = torch.zeros(L, hd)
    v = torch.zeros(L, hd)
    for i in range(L):
        q[i, i % hd] = 16.0
        k[i, i % hd] = 16.0           # score(i,i) = 256/sqrt(hd) >> others (0); hd = 32: rows i, i+32 tie and average
        v[i, (i * 7 + 3) % hd] = float(i % 13 + 1)
    qkv = torch.stack([q, k, v], dim=1).reshape(L, 3 * hd).to(torch.bfloat16)
    o, _ = hip.attention_fwd(qkv.cuda(), B, L, H, hd)
    torch.cuda.synchronize()
    ref = _attn_ref(qkv.float(), B, L, H, hd)
    np.testing.assert_allclose(o.float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)


def test_attention_refuses_a_misaligned_output(hip):
    """The attention kernels read and (since the widened stores of round 5) write 16 bytes per lane: an output that is only
    8-byte aligned must come back as a Python error from the argument check, not reach a kernel."""
    B, L, H = 1, 128, 2
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
    backing = torch.empty(B * L * H * 64 + 8, device="cuda", dtype=torch.bfloat16)
    o_bad = backing[4:4 + B * L * H * 64].view(B * L, H * 64)            # 8 bytes past a 16-byte boundary
    assert o_bad.data_ptr() % 16 == 8
    with pytest.raises(hip.HipError):
        hip.attention_fwd(qkv, B, L, H, 64, o=o_bad)
    o, lse = hip.attention_fwd(qkv, B, L, H, 64)
    dq_bad = torch.empty(qkv.numel() + 8, device="cuda", dtype=torch.bfloat16)[4:4 + qkv.numel()].view_as(qkv)
    with pytest.raises(hip.HipError):
        hip.attention_bwd(qkv, o, torch.randn_like(o), lse, B, L, H, 64, dqkv=dq_bad)
    torch.cuda.synchronize()


# --------------------------------------------------------------------------------------------- VQ
@pytest.mark.parametrize("case", vq_cases())
@pytest.mark.parametrize("mode", ["L", "D"])
def test_vq_forward_bit_exact_vs_c_oracle(hip, case, mode, golden_dir):
    (b, n), K, d, seed = case
    W = gen.kaiming_uniform_codebook(K, d, seed)
    z = gen.normal((b * n, d), seed + 1000)
    inv_tau = float(np.float32(1.0 / 0.03))
    ref = vq_c.vq_forward(z, W, mode)
    o = hip.vq_forward(torch.from_numpy(z).cuda(), torch.from_numpy(W).cuda(), 0 if mode == "L" else 1, inv_tau=inv_tau, ldp=64)
    torch.cuda.synchronize()
    assert np.array_equal(o["zn"].cpu().numpy(), ref["z"])            # normalisation bit-exact
    assert np.array_equal(o["E"].cpu().numpy(), ref["emb"])
    assert np.array_equal(o["idx"].cpu().numpy(), ref["idx"])          # indices bit-exact
    assert np.array_equal(o["rz"].cpu().numpy(), ref["regularized_z"])
    np.testing.assert_allclose(o["losses"].cpu().numpy()[:3], [ref["loss_q"], ref["loss_commit"], ref["loss_codebook"]], rtol=1e-5)
    assert torch.equal(o["rz_pad"].cpu()[:, :d], torch.from_numpy(ref["regularized_z"]).to(torch.bfloat16))
    assert torch.all(o["rz_pad"].cpu()[:, d:] == 0)
    # and against the reference's own indices (golden fixture)
    f = np.load(f"{golden_dir}/vq_N{b * n}_K{K}_d{d}_{mode}.npz")
    assert np.array_equal(o["idx"].cpu().numpy(), f["idx"].reshape(-1).astype(np.int64))


@pytest.mark.parametrize("case", vq_cases())    # all three fixtures: d = 24 twice and the d = 16 one (configs C / D / E)
def test_vq_backward_matches_reference_grads(hip, case, golden_dir):
    (b, n), K, d, seed = case
    f = np.load(f"{golden_dir}/vq_N{b * n}_K{K}_d{d}_L.npz")
    W = gen.kaiming_uniform_codebook(K, d, seed)
    z = gen.normal((b * n, d), seed + 1000)
    g = gen.normal((b * n, d), seed + 2000)
    o = hip.vq_forward(torch.from_numpy(z).cuda(), torch.from_numpy(W).cuda(), 0)
    gscal = torch.tensor([0.7, 0.0, 0.0]).cuda()
    dz, _, dW = hip.vq_backward(torch.from_numpy(g).cuda(), gscal, o)
    torch.cuda.synchronize()
    np.testing.assert_allclose(dz.cpu().numpy(), f["dz"].reshape(-1, d), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(dW.cpu().numpy(), f["dE"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("N,K,d,collapse", [(700, 512, 24, False), (2048, 1024, 16, False), (1500, 256, 24, True)])
def test_vq_codebook_gradient_bit_exact_vs_c_oracle(hip, N, K, d, collapse):
    """dW = OneHot^T (Q - Z) on the exact-fp32 MFMA: equal, bit for bit, to the C restatement of its summation order --
    also when most tokens sit on a handful of codes (the regime the kernel was rewritten for)."""
    W = gen.kaiming_uniform_codebook(K, d, 301)
    z = gen.normal((N, d), 302)
    if collapse:
        z[: N - 100] = W[7] + 0.05 * z[: N - 100]            # most tokens next to code 7
    o = hip.vq_forward(torch.from_numpy(z).cuda(), torch.from_numpy(W).cuda(), 0)
    gs = np.array([0.7, 0.0, 0.3], dtype=np.float32)
    dz, _, dW = hip.vq_backward(None, torch.from_numpy(gs).cuda(), o)
    torch.cuda.synchronize()
    s_b = np.float32(np.float32(gs[0] * np.float32(1.0) + gs[2]) * np.float32(2.0)) / np.float32(np.float32(N) * np.float32(d))
    ref = vq_c.codebook_grad(o["zn"].cpu().numpy(), o["E"].cpu().numpy(), o["wnorm"].cpu().numpy(), o["idx"].cpu().numpy(), s_b)
    if collapse:
        assert np.bincount(o["idx"].cpu().numpy(), minlength=K).max() > N // 2
    assert np.array_equal(dW.cpu().numpy(), ref)


def test_vq_stochastic_mode_matches_softmax_distribution(hip):
    """mode 2 has no bit-exact oracle (torch.multinomial CPU != GPU): chi-square of the sampled index
    frequencies of ONE token replicated N times against softmax(cos/tau) from the oracle."""
    K, d, N = 256, 24, 65536
    W = gen.kaiming_uniform_codebook(K, d, 77)
    z1 = gen.normal((1, d), 78)
    z = np.repeat(z1, N, axis=0)
    o = hip.vq_forward(torch.from_numpy(z).cuda(), torch.from_numpy(W).cuda(), 2, inv_tau=1.0 / 0.3, seed=1234)
    torch.cuda.synchronize()
    zn = F.normalize(torch.from_numpy(z1), dim=-1)
    en = F.normalize(torch.from_numpy(W), dim=-1)
    probs = torch.softmax((zn @ en.t()) / 0.3, dim=-1).double().numpy().reshape(-1)
    cnt = np.bincount(o["idx"].cpu().numpy(), minlength=K).astype(np.float64)
    keep = probs * N >= 5
    chi2 = (((cnt - probs * N) ** 2) / (probs * N))[keep].sum()
    dof = keep.sum() - 1
    assert chi2 < dof + 6 * np.sqrt(2 * dof), (chi2, dof)
    # a different seed gives a different draw
    o2 = hip.vq_forward(torch.from_numpy(z).cuda(), torch.from_numpy(W).cuda(), 2, inv_tau=1.0 / 0.3, seed=99)
    assert (o2["idx"] != o["idx"]).float().mean() > 0.5
    # consecutive per-forward counters must give unrelated noise: with the counter XORed into the token index, call s + 1
    # re-used call s's noise vectors permuted over tokens (token t of call s == token t ^ s ^ (s + 1) of call s + 1)
    o3 = hip.vq_forward(torch.from_numpy(z).cuda(), torch.from_numpy(W).cuda(), 2, inv_tau=1.0 / 0.3, seed=1235)
    perm = torch.arange(N, device="cuda") ^ (1234 ^ 1235)
    assert (o3["idx"][perm] != o["idx"]).float().mean() > 0.5


@pytest.mark.parametrize("M,N,K,variant", [(200, 320, 128, 2), (700, 768, 192, 2), (400, 96, 64, 5)])
def test_gemm_nt_dgelu_fused_column_sums(hip, M, N, K, variant):
    """vtGemmNT.colsum_partial: per-192-row column sums of the rounded output, out of the epilogue (fc1 bias gradient)"""
    hip.GEMM_TILE = variant
    try:
        A, B = bf(_rand((M, K), 91, 0.5)), bf(_rand((N, K), 92, 0.5))
        u = bf(_rand((M, N), 93))
        tiles = (M + 191) // 192
        part = torch.full((tiles, N), float("nan"), device="cuda")
        out = hip.gemm_nt(A.cuda(), B.cuda(), hip.EPI_BF16_DGELU, aux=u.cuda(), colsum_partial=part)
        plain = hip.gemm_nt(A.cuda(), B.cuda(), hip.EPI_BF16_DGELU, aux=u.cuda())
        torch.cuda.synchronize()
        assert torch.equal(out, plain)                                   # the output does not depend on the option
        want = torch.stack([out[t * 192:(t + 1) * 192].float().sum(0) for t in range(tiles)])
        np.testing.assert_allclose(part.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(part.sum(0).cpu().numpy(), hip.colsum(out).cpu().numpy(), rtol=1e-5, atol=1e-3)
    finally:
        hip.GEMM_TILE = 0


@pytest.mark.parametrize("ragged", [False, True])
def test_pack_weights_grouped_equals_single_packs(hip, ragged):
    """vt_pack_weights_grouped (all bf16 operand copies of the model in a few launches) == a cast / transpose in torch.  Extents that
    are all multiples of 4 take the 64x64 kernel (16 B in, 8 B out per thread); one ragged job in a group sends that group to the
    32x32 element-wise kernel"""
    import ctypes
    shapes = [(768, 768), (2304, 768), (24, 768), (768, 24), (100, 36)] * 8     # 40 jobs: two grouped launches
    if ragged:
        shapes[3], shapes[37] = (33, 50), (7, 129)
    ws = [torch.from_numpy(_rand(sh, 400 + i)).cuda() for i, sh in enumerate(shapes)]
    perm = torch.randperm(100).to(torch.int32).cuda()
    jobs = (hip.PackJob * len(shapes))()
    outs = []
    for i, (w, (N, K)) in enumerate(zip(ws, shapes)):
        wb = torch.zeros(N, K + 8, device="cuda", dtype=torch.bfloat16)
        wt = torch.zeros(K, N + 8, device="cuda", dtype=torch.bfloat16) if i % 3 else None
        rp = perm if (N == 100) else None
        j = jobs[i]
        j.w, j.N, j.K, j.row_perm = w.data_ptr(), N, K, (rp.data_ptr() if rp is not None else None)
        j.wb, j.ldd, j.wt, j.lddT = wb.data_ptr(), K + 8, (wt.data_ptr() if wt is not None else None), N + 8
        outs.append((wb, wt, rp))
    hip.check(hip.lib().vt_pack_weights_grouped(jobs, len(shapes), hip.stream()), "vt_pack_weights_grouped")
    torch.cuda.synchronize()
    for w, (N, K), (wb, wt, rp) in zip(ws, shapes, outs):
        src = w[rp.long()] if rp is not None else w
        assert torch.equal(wb[:, :K], src.to(torch.bfloat16)) and torch.all(wb[:, K:] == 0)
        if wt is not None:
            assert torch.equal(wt[:, :N], src.t().to(torch.bfloat16)) and torch.all(wt[:, N:] == 0)


@pytest.mark.parametrize("K", [64, 128, 192, 768])
@pytest.mark.parametrize("epi", ["bf16", "gelu", "f32res", "dgelu"])
def test_gemm_nt_persistent_walk_equals_one_tile_per_workgroup(hip, K, epi):
    """the persistent 192x192 kernel (a workgroup walks 2-3 tiles, the next tile's first K-tile is DMA-ed during the epilogue) must
    give bit-identical results to the same kernel launched one tile per workgroup, for 1, 2, 3 and 12 K-tiles, ragged M and N
    (edge tiles first / last in a workgroup's walk), every epilogue; and agree with integer-exact fp32 math"""
    M, N = 192 * 31 + 77, 192 * 17 + 52                      # 32 x 18 = 576 tiles on 256 workgroups
    g = torch.Generator().manual_seed(K)
    A = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16).cuda()
    B = torch.randint(-3, 4, (N, K), generator=g).to(torch.bfloat16).cuda()
    res = torch.randint(-8, 9, (M, N), generator=g).float().cuda()
    aux = (torch.randint(-6, 7, (M, N), generator=g).float() / 4).to(torch.bfloat16).cuda()
    kw = {"bf16": dict(epi=hip.EPI_BF16), "gelu": dict(epi=hip.EPI_BF16_GELU), "f32res": dict(epi=hip.EPI_F32, residual=res),
          "dgelu": dict(epi=hip.EPI_BF16_DGELU, aux=aux)}[epi]
    outs = {}
    try:
        for v in (2, 6):
            hip.GEMM_TILE = v
            o = hip.gemm_nt(A, B, **kw)
            outs[v] = [t.clone() for t in (o if isinstance(o, tuple) else (o,))]
    finally:
        hip.GEMM_TILE = 0
    for a, b in zip(outs[2], outs[6]):
        assert torch.equal(a, b)
    exact = A.float() @ B.float().t()                           # small integers: exact in fp32, and in bf16 up to |x| <= 256
    if epi == "bf16":
        assert torch.equal(outs[2][0].float(), exact.to(torch.bfloat16).float())
    if epi == "f32res":
        assert torch.equal(outs[2][0], exact + res)


@pytest.mark.parametrize("M,N", [(192 * 31 + 77, 192 * 17 + 52), (192 * 9 + 1, 192 * 6), (192 * 3, 192 * 12 + 4), (192 * 64, 192 * 7)])
@pytest.mark.parametrize("epi", ["bf16", "dgelu"])
def test_gemm_nt_tile_order_does_not_change_the_result(hip, M, N, epi):
    """the 192x192 kernel walks wide outputs in column blocks of W tile columns (automatic: 6 or 8) instead of as a row-major list: every
    forced order (vtGemmNT.tile 19 + W, W = 0 .. 12: blocks that divide the tile columns, that leave a narrower last block, that are wider
    than the matrix), persistent and one tile per workgroup, must visit every tile exactly once -- outputs (and the fused column sums)
    bit-identical to the row-major walk and equal to exact integer math"""
    K = 128
    g = torch.Generator().manual_seed(M + N)
    A = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16).cuda()
    B = torch.randint(-3, 4, (N, K), generator=g).to(torch.bfloat16).cuda()
    aux = (torch.randint(-6, 7, (M, N), generator=g).float() / 4).to(torch.bfloat16).cuda()
    kw = dict(epi=hip.EPI_BF16) if epi == "bf16" else dict(epi=hip.EPI_BF16_DGELU, aux=aux)
    outs = {}
    try:
        for v in [19 + w for w in range(13)] + [2, 6]:
            hip.GEMM_TILE = v
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)     # a tile nobody visits stays NaN
            part = torch.full(((M + 191) // 192, N), float("nan"), device="cuda") if epi == "dgelu" else None
            o = hip.gemm_nt(A, B, out=out, colsum_partial=part, **kw)
            outs[v] = [o.clone()] + ([part] if part is not None else [])
    finally:
        hip.GEMM_TILE = 0
    for v, o in outs.items():
        for a, b in zip(outs[19], o):
            assert torch.equal(a, b), v
    if epi == "bf16":
        assert torch.equal(outs[19][0].float(), (A.float() @ B.float().t()).to(torch.bfloat16).float())
