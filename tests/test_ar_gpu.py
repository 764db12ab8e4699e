"""LARP_AR, the consumer of the tokenizer's `bottleneck_rep`, on the GPU: the HIP kernels it adds (RMSNorm, SwiGLU,
causal attention, KV-cache decode attention) against plain torch fp32 math, and the module against the CPU restatement
(oracle/ar_oracle.py, itself pinned to the reference's LARP_AR by tests/test_ar_oracle_cpu.py) and the reference's own
greedy generations (tests/golden/ar_*.npz)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ar_oracle as A
from oracle import inputs as gen
from tests.golden.make_golden import ar_cases, ar_inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rb(t):
    return t.to(torch.bfloat16).float()


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("dim", [384, 768, 1024, 1280, 1536, 2560])
@pytest.mark.parametrize("rows", [1, 67, 2048])
def test_rmsnorm_forward_backward(dim, rows):
    from video_tokenizer_amd import hip
    x = torch.from_numpy(gen.normal((rows, dim), 11 + dim)).cuda() * 1.7
    w = 1.0 + torch.from_numpy(gen.normal((dim,), 12, 0.2)).cuda()
    dy = rb(torch.from_numpy(gen.normal((rows, dim), 13)).cuda())
    dres = torch.from_numpy(gen.normal((rows, dim), 14)).cuda()
    y, rstd = hip.rmsnorm_fwd(x, w, 1e-5)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = xr * torch.rsqrt((xr * xr).mean(-1, keepdim=True) + 1e-5) * wr
    assert torch.equal(y.float(), rb(ref.detach())) or (y.float() - rb(ref.detach())).abs().max() <= 2 ** -7 * ref.detach().abs().max()   # 1 bf16 ulp
    assert rel(y, ref.detach()) < 4e-3
    assert rel(rstd, torch.rsqrt((x * x).mean(-1) + 1e-5)) < 1e-6
    ref.backward(dy)
    dx, dxb, dw = hip.rmsnorm_bwd(dy.to(torch.bfloat16), x, w, rstd, dres=dres, want_bf16=True)
    assert rel(dx, xr.grad + dres) < 1e-5
    assert rel(dxb, xr.grad + dres) < 4e-3
    assert rel(dw, wr.grad) < 1e-5
    dx2, none, dw2 = hip.rmsnorm_bwd(dy.to(torch.bfloat16), x, w, rstd)
    assert none is None and rel(dx2, xr.grad) < 1e-5 and torch.equal(dw2, dw)


def test_rmsnorm_refuses_other_widths_and_cpu_tensors():
    from video_tokenizer_amd import hip
    with pytest.raises(hip.HipError, match="unsupported"):
        hip.rmsnorm_fwd(torch.zeros(4, 512, device="cuda"), torch.ones(512, device="cuda"), 1e-5)
    with pytest.raises(hip.HipError, match="GPU tensors only"):
        hip.rmsnorm_fwd(torch.zeros(4, 384), torch.ones(384), 1e-5)


@pytest.mark.parametrize("M,I", [(1, 1024), (130, 2048), (2048, 2816), (512, 6912)])
def test_swiglu_forward_backward(M, I):
    from video_tokenizer_amd import hip
    h = torch.from_numpy(gen.normal((M, 2 * I), 21, 2.0)).cuda().to(torch.bfloat16)
    da = torch.from_numpy(gen.normal((M, I), 22)).cuda().to(torch.bfloat16)
    a = hip.swiglu_fwd(h)
    x, g = h[:, :I].float().requires_grad_(True), h[:, I:].float().requires_grad_(True)
    ref = rb(F.silu(g)) * x                                     # autocast: silu -> bf16, product -> bf16
    assert rel(a, ref.detach()) < 3e-3
    assert (a.float() - rb(ref.detach())).abs().max() <= 2 ** -7 * ref.detach().abs().max()
    (F.silu(g) * x).backward(da.float())
    dh = hip.swiglu_bwd(da, h)
    assert rel(dh[:, :I], x.grad) < 6e-3 and rel(dh[:, I:], g.grad) < 6e-3


def _sdpa_ref(qkv, B, L, H):
    q, k, v = (t.reshape(B, L, H, 64).transpose(1, 2).float() for t in qkv.split(H * 64, dim=-1))
    s = (q @ k.transpose(-2, -1)) / 8.0
    s = s.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool, device=qkv.device)), float("-inf"))
    return torch.softmax(s, dim=-1) @ v, s


@pytest.mark.parametrize("B,L,H", [(2, 64, 6), (1, 1, 6), (3, 100, 2), (2, 129, 3), (1, 1024, 12), (2, 63, 6), (1, 1500, 2)])
def test_causal_attention_forward_backward_vs_torch(B, L, H):
    from video_tokenizer_amd import hip
    D = H * 64
    qkv = torch.from_numpy(gen.normal((B * L, 3 * D), 31 + L, 1.3)).cuda().to(torch.bfloat16)
    dO = torch.from_numpy(gen.normal((B * L, D), 32 + L)).cuda().to(torch.bfloat16)
    o, lse2 = hip.attention_causal_fwd(qkv, B, L, H)
    qr = qkv.float().requires_grad_(True)
    ref, s = _sdpa_ref(qr, B, L, H)
    ref2 = ref.transpose(1, 2).reshape(B * L, D)
    assert rel(o, ref2.detach()) < 5e-3, rel(o, ref2.detach())
    assert rel(lse2, torch.logsumexp(s.detach(), dim=-1) * math.log2(math.e)) < 1e-3
    ref2.backward(dO.float())
    dqkv = hip.attention_causal_bwd(qkv, o, dO, lse2, B, L, H)
    for i, nm in enumerate("qkv"):
        mine, want = dqkv[:, i * D:(i + 1) * D], qr.grad[:, i * D:(i + 1) * D]
        if L == 1 and nm in "qk":       # one visible key: the softmax is constant, dq = dk = 0 (up to the bf16 rounding of o in delta)
            assert float(mine.float().abs().max()) < 0.05
            continue
        r = rel(mine, want)
        assert r < 1.2e-2, (nm, r)
    assert torch.isfinite(dqkv.float()).all()


def test_causal_attention_ignores_later_keys():
    """position t must not depend on keys after t (the property the KV cache relies on).  Not bit-for-bit: the kernel's lazy
    softmax rescaling is decided per wave (32 queries), so a later query of the same wave can move the reference point of an
    earlier one -- the same value, rounded differently."""
    from video_tokenizer_amd import hip
    B, L, H = 2, 200, 4
    qkv = torch.from_numpy(gen.normal((B * L, 3 * H * 64), 41)).cuda().to(torch.bfloat16)
    o1, _ = hip.attention_causal_fwd(qkv, B, L, H)
    q2 = qkv.clone().reshape(B, L, -1)
    q2[:, 150:] = torch.from_numpy(gen.normal((B, 50, 3 * H * 64), 42, 3.0)).cuda().to(torch.bfloat16)
    o2, _ = hip.attention_causal_fwd(q2.reshape(B * L, -1), B, L, H)
    assert torch.equal(o1.reshape(B, L, -1)[:, :128], o2.reshape(B, L, -1)[:, :128])        # other workgroups: untouched
    assert rel(o1.reshape(B, L, -1)[:, 128:150], o2.reshape(B, L, -1)[:, 128:150]) < 3e-3


@pytest.mark.parametrize("n_keys", [1, 2, 63, 64, 65, 1000, 1032])
def test_decode_attention_vs_torch(n_keys):
    from video_tokenizer_amd import hip
    B, Bmax, H, Lmax = 3, 4, 6, 1032
    q = torch.from_numpy(gen.normal((B, H, 64), 51, 1.5)).cuda().to(torch.bfloat16)
    kc = torch.from_numpy(gen.normal((Bmax, H, Lmax, 64), 52)).cuda().to(torch.bfloat16)
    vc = torch.from_numpy(gen.normal((Bmax, H, Lmax, 64), 53)).cuda().to(torch.bfloat16)
    o = hip.decode_attention(q, kc, vc, n_keys)
    s = torch.einsum("bhd,bhkd->bhk", q.float(), kc[:B, :, :n_keys].float()) / 8.0
    ref = torch.einsum("bhk,bhkd->bhd", torch.softmax(s, -1), vc[:B, :, :n_keys].float())
    assert rel(o, ref) < 4e-3, rel(o, ref)
    with pytest.raises(hip.HipError):
        hip.decode_attention(q, kc, vc, Lmax + 1)


@pytest.mark.parametrize("pos", [0, 1, 7, 8, 63, 500, 1031])
def test_decode_attention_step_updates_cache_and_matches_torch(pos):
    """the graph-capturable decode step: position read from device memory, k / v of the new token stored into the caches"""
    from video_tokenizer_amd import hip
    B, Bmax, H, Lmax = 3, 4, 6, 1032
    D = H * 64
    qkv = torch.from_numpy(gen.normal((B, 3 * D), 61, 1.2)).cuda().to(torch.bfloat16)
    kc = torch.from_numpy(gen.normal((Bmax, H, Lmax, 64), 62)).cuda().to(torch.bfloat16)
    vc = torch.from_numpy(gen.normal((Bmax, H, Lmax, 64), 63)).cuda().to(torch.bfloat16)
    k0, v0 = kc.clone(), vc.clone()
    o = hip.decode_attention_step(qkv, kc, vc, torch.tensor([pos], device="cuda", dtype=torch.int32))
    q, k, v = (t.reshape(B, H, 64) for t in qkv.split(D, dim=-1))
    k0[:B, :, pos], v0[:B, :, pos] = k, v
    assert torch.equal(kc, k0) and torch.equal(vc, v0)                          # exactly one row per (b, h) changed, to the new k / v
    s = torch.einsum("bhd,bhkd->bhk", q.float(), k0[:B, :, :pos + 1].float()) / 8.0
    ref = torch.einsum("bhk,bhkd->bhd", torch.softmax(s, -1), v0[:B, :, :pos + 1].float()).reshape(B, D)
    assert rel(o, ref) < 4e-3, rel(o, ref)
    o2 = hip.decode_attention(q.contiguous(), kc, vc, pos + 1)                  # same kernel body, position by value, keys from the cache
    assert torch.equal(o2.reshape(B, D), o)


@pytest.mark.parametrize("M,K,N", [(1, 384, 1152), (16, 768, 2304), (33, 1024, 4096), (64, 2560, 512)])
def test_decode_norm_linear_is_bit_identical_to_the_separate_kernels(M, K, N):
    """RMSNorm folded into the weight-streaming GEMM (and SwiGLU into its epilogue): same bits as rmsnorm -> skinny GEMM -> swiglu"""
    from video_tokenizer_amd import hip
    x = torch.from_numpy(gen.normal((M, K), 91, 1.5)).cuda()
    nw = 1.0 + torch.from_numpy(gen.normal((K,), 92, 0.2)).cuda()
    W = torch.from_numpy(gen.normal((N, K), 93, 0.05)).cuda().to(torch.bfloat16)
    y, _ = hip.rmsnorm_fwd(x, nw, 1e-5)
    ref_bf = hip.gemm_nt(y, W, hip.EPI_BF16, tile=7)
    assert torch.equal(hip.decode_norm_linear(x, nw, 1e-5, W, mode=0), ref_bf)
    assert torch.equal(hip.decode_norm_linear(x, nw, 1e-5, W, mode=2), hip.gemm_nt(y, W, hip.EPI_F32, round_bf16=True, tile=7))
    I = N // 2                                                   # rows 0..I-1 play w3, I..2I-1 play w1
    Wi = torch.cat([W[:I].reshape(I // 8, 8, K), W[I:].reshape(I // 8, 8, K)], dim=1).reshape(N, K).contiguous()
    assert torch.equal(hip.decode_norm_linear(x, nw, 1e-5, Wi, mode=1), hip.swiglu_fwd(ref_bf))
    assert torch.equal(hip.decode_norm_linear(y, None, 0.0, Wi, mode=1), hip.swiglu_fwd(ref_bf))        # pre-normalised bf16 operand
    assert torch.equal(hip.decode_norm_linear(y, None, 0.0, W, mode=0), ref_bf)
    ref = (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5) * nw).to(torch.bfloat16).float() @ W.float().t()
    assert rel(ref_bf, ref) < 4e-3


# ------------------------------------------------------------------------------------------------ the module
def build(name, **over):
    import video_tokenizer_amd as vt
    kw, B, seed = ar_cases()[name]
    cfg = A.make_cfg(**kw)
    sd = A.init_state_dict(cfg, seed)
    args = dict(dim=cfg["dim"], n_layer=cfg["n_layer"], n_head=cfg["n_head"], n_kv_head=cfg["n_kv_head"], vocab_size=cfg["vocab_size"], max_seq_len=cfg["max_seq_len"],
                num_classes=cfg["num_classes"], cls_token_num=cfg["cls_token_num"], frame_prediction=cfg["frame_prediction"], use_fixed_pe=cfg["use_fixed_pe"],
                token_dropout_p=0.0, resid_dropout_p=0.0, ffn_dropout_p=0.0, class_dropout_prob=0.1)
    args.update(over)
    m = vt.LARP_AR(vt.larp_ar.ModelArgs(**args))
    m.load_state_dict(sd, strict=True)
    return m.cuda(), cfg, sd, ar_inputs(cfg, B, seed), B


@pytest.mark.parametrize("name", list(ar_cases()))
def test_module_matches_oracle_forward_loss_gradients(name):
    m, cfg, sd, (tok, cond), B = build(name)
    g = np.load(os.path.join(G, f"ar_{name}.npz"))
    assert sorted(m.state_dict().keys()) == g["sd_keys"].tolist()
    m.train()
    if not cfg["frame_prediction"]:
        m.cls_embedding.dropout_prob = 0.0
    logits, loss = m(tok[:, :-1].cuda(), cond.cuda(), targets=tok.cuda())
    loss.backward()
    torch.cuda.synchronize()
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    lo, ls = A.forward(p, cfg, tok[:, :-1], cond, targets=tok, emu=True)
    ls.backward()
    assert logits.shape == lo.shape
    assert rel(logits, lo.detach()) < 1e-2, rel(logits, lo.detach())
    assert abs(loss.item() - ls.item()) < 2e-3
    assert abs(loss.item() - float(g["loss"])) < 2e-2                   # and the reference's own fp32 loss
    worst = {}
    for k, v in m.named_parameters():
        if p[k].grad is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            continue
        worst[k] = rel(v.grad, p[k].grad)
    bad = {k: r for k, r in worst.items() if r > 3e-2}
    assert not bad, bad
    _, lv = m(tok[:, :-1].cuda(), cond.cuda(), targets=tok.cuda(), valid=torch.tensor([1.0] + [0.0] * (B - 1), device="cuda"))
    assert abs(lv.item() - float(g["loss_valid"])) < 3e-2
    m.eval()
    with torch.no_grad():
        le, none = m(tok[:, :-1].cuda(), cond.cuda())
    assert none is None and list(le.shape) == g["logits_eval_shape"].tolist()


@pytest.mark.parametrize("name", list(ar_cases()))
def test_greedy_generation_through_kv_cache_matches_reference(name):
    """tokens are index work: equal to the reference's generation except where the oracle's top-2 gap is a near tie at bf16"""
    m, cfg, sd, (tok, cond), B = build(name)
    g = np.load(os.path.join(G, f"ar_{name}.npz"))
    m.eval()
    from video_tokenizer_amd.larp_ar import generate
    for scale in (1.0,) if cfg["frame_prediction"] else (1.0, 3.0):
        n_new = cfg["max_seq_len"]
        with m.sampling():
            seq = generate(m, cond.cuda(), n_new, cfg_scale=scale, temperature=1.0, top_k=0, top_p=1.0, sample_logits=False)
        m.reset_caches()
        torch.cuda.synchronize()
        assert seq.shape == (B, n_new) and seq.dtype == torch.int32
        got = seq.cpu().numpy()
        n_chk = 24
        want, margin = A.generate_greedy(sd, cfg, cond, n_chk, cfg_scale=scale, emu=True, return_margins=True)
        for b in range(B):
            diff = np.nonzero(got[b, :n_chk] != want[b].numpy())[0]
            assert diff.size == 0 or margin[b, diff[0]] < 2e-3, (name, scale, b, diff[:4], float(margin[b, diff[0]]))
        if f"greedy_cfg{scale:g}" in g:           # none for grouped-query attention: the reference's own KV cache raises there; this build's works
            agree = (got == g[f"greedy_cfg{scale:g}"]).mean()
            print(name, "cfg", scale, "agreement with the reference's fp32 generation:", agree)


def test_graph_replayed_generation_equals_eager_loop():
    """the hipGraph-captured decode step replays the same kernels on the same buffers: greedy tokens are identical to the eager loop's"""
    from video_tokenizer_amd.larp_ar import generate
    m, cfg, sd, (tok, cond), B = build("class_S2")
    m.eval()
    outs = []
    for use_graph in (False, True):
        with m.sampling():
            outs.append(generate(m, cond.cuda(), cfg["max_seq_len"], cfg_scale=3.0, cfg_interval=20, use_graph=use_graph, temperature=1.0, top_k=0, top_p=1.0,
                                 sample_logits=False))
        m.reset_caches()
    assert torch.equal(outs[0], outs[1])
    for level in ("0", "norm"):                                                 # and to the other fusion levels of the decode kernels (same bits by construction)
        os.environ["VT_AR_FUSED_DECODE"] = level
        try:
            with m.sampling():
                other = generate(m, cond.cuda(), cfg["max_seq_len"], cfg_scale=3.0, cfg_interval=20, use_graph=False, temperature=1.0, top_k=0, top_p=1.0,
                                 sample_logits=False)
            m.reset_caches()
        finally:
            del os.environ["VT_AR_FUSED_DECODE"]
        assert torch.equal(outs[0], other), level
    torch.manual_seed(3)
    with m.sampling():                                                          # sampled path (top-k / top-p / multinomial inside the graph): valid tokens
        s = generate(m, cond.cuda(), cfg["max_seq_len"], cfg_scale=1.0, use_graph=True, temperature=0.8, top_k=40, top_p=0.9, sample_logits=True)
    m.reset_caches()
    assert int(s.min()) >= 0 and int(s.max()) < cfg["vocab_size"]


def test_cached_decode_equals_full_recomputation():
    """logits of the newest position from prefill + one-token steps == the training-branch forward over the same prefix"""
    m, cfg, sd, (tok, cond), B = build("class_S2")
    m.eval()
    dev = "cuda"
    n = 40
    with torch.no_grad():
        full, _ = m(tok[:, :n].to(dev), cond.to(dev))                      # [B, 1 + n, V]
        m.setup_caches(B, cfg["max_seq_len"] + 1)
        with m.sampling():
            outs = [m(None, cond.to(dev), torch.arange(0, 1, device=dev))[0][:, -1]]
            for t in range(n):
                outs.append(m(tok[:, t:t + 1].to(dev), None, torch.tensor([t + 1], device=dev, dtype=torch.int))[0][:, -1])
        m.reset_caches()
    cached = torch.stack(outs, 1)
    assert rel(cached, full) < 1e-2, rel(cached, full)


def test_frame_prediction_prefill_equals_full_forward():
    m, cfg, sd, (tok, cond), B = build("frame_S2")
    m.eval()
    T = cfg["cls_token_num"]
    with torch.no_grad():
        m.setup_caches(B, cfg["max_seq_len"] + T)
        with m.sampling():
            pre, _ = m(None, cond.cuda(), torch.arange(0, T, device="cuda"))
        m.reset_caches()
        with m.sampling():
            full, _ = m(tok[:, :8].cuda(), cond.cuda(), torch.arange(0, T + 8, device="cuda"))
    assert rel(pre, full[:, :T]) < 2e-3


def test_sampling_api_seeded_and_in_range():
    m, cfg, sd, (tok, cond), B = build("class_S2")
    m.eval()
    torch.manual_seed(5)
    a = m.sample(cond.cuda(), cfg_scale=2.0, temperature=0.9, top_k=50, top_p=0.95)
    m.reset_caches()
    torch.manual_seed(5)
    b = m.sample(cond.cuda(), cfg_scale=2.0, temperature=0.9, top_k=50, top_p=0.95)
    m.reset_caches()
    assert a.shape == (B, cfg["max_seq_len"]) and torch.equal(a, b)
    assert int(a.min()) >= 0 and int(a.max()) < cfg["vocab_size"]
    assert not m.is_sampling and m.layers[0].attention.kv_cache is None


def test_bf16_cast_model_like_sample_py():
    """sample.py:105,263 casts the AR model with .to(dtype) before sampling and scoring"""
    m, cfg, sd, (tok, cond), B = build("class_S2")
    with torch.no_grad():
        ref, _ = m.eval()(tok[:, :-1].cuda(), cond.cuda())
        mb = m.to(torch.bfloat16).eval()
        out, loss = mb(tok[:, :-1].cuda(), cond.cuda(), targets=None)
        seq = mb.sample(cond.cuda(), cfg_scale=1.0, top_k=10)
    assert rel(out[:, :], ref) < 3e-2 and seq.shape == (B, cfg["max_seq_len"])


def test_registry_names_checkpoint_and_training_step():
    import video_tokenizer_amd as vt
    assert set(vt.larp_ar.larp_ar_models) == {"llama-abs-" + s for s in ("S", "B", "L", "LP", "XL", "XXL", "XXXL")}
    spec = {"name": "llama-abs-S", "args": dict(vocab_size=1024, max_seq_len=128, num_classes=101, token_dropout_p=0.1, resid_dropout_p=0.1, ffn_dropout_p=0.1,
                                                 drop_path_rate=0.1)}
    m = vt.registry.make(spec).cuda()
    assert m.config.dim == 384 and m.config.n_layer == 12 and len(m.layers) == 12
    assert float(m.output.weight.abs().max()) == 0.0                      # larp_ar.py:286: zero-initialised head
    spec["sd"] = {k: v.cpu() for k, v in m.state_dict().items()}
    m2 = vt.LARP_AR.from_checkpoint({"model": spec}).cuda()
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    torch.nn.init.normal_(m2.output.weight, std=0.02)
    opt = torch.optim.AdamW(m2.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
    B = 4
    tok = torch.from_numpy((gen.hash_u64(B * 128, 71) % np.uint64(1024)).astype(np.int64)).reshape(B, 128).cuda()
    lab = torch.from_numpy((gen.hash_u64(B, 72) % np.uint64(101)).astype(np.int64)).cuda()
    m2.train()
    torch.manual_seed(0)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        _, loss = m2(tok[:, :-1], lab, targets=tok)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.3, losses      # memorises the batch


def test_tokenizer_indices_feed_the_prior():
    """the §8f rank-4 hand-off: LARPTokenizer.encode(...)['bottleneck_rep'] -> LARP_AR(idx[:, :-1], label, targets=idx)"""
    import video_tokenizer_amd as vt
    from oracle import larp_oracle as O
    from tests.test_model_gpu import build as build_tok
    cfg = O.make_cfg("tiny")
    tokz, _ = build_tok(cfg)
    x = torch.from_numpy(gen.video_clips(4, cfg["frame_num"], cfg["input_size"], 81)).cuda()
    tokz.eval()
    with torch.no_grad():
        idx = tokz.encode(x)["bottleneck_rep"]
    assert idx.dim() == 2
    idx = idx.long()
    n = idx.shape[1]
    ar = vt.registry.make({"name": "llama-abs-S", "args": dict(vocab_size=cfg["codebook_size"], max_seq_len=n, num_classes=3)}).cuda()
    torch.nn.init.normal_(ar.output.weight, std=0.02)
    logits, loss = ar(idx[:, :-1], torch.tensor([0, 1, 2, 1], device="cuda"), targets=idx)
    loss.backward()
    assert logits.shape == (4, n, cfg["codebook_size"]) and torch.isfinite(loss)
    assert abs(loss.item() - math.log(cfg["codebook_size"])) < 0.5
    # and back (sample.py:168-190): sampled token ids -> decode_from_bottleneck -> video
    ar.eval()
    ids = ar.sample(torch.tensor([0, 2], device="cuda"), cfg_scale=1.5, temperature=1.0, top_k=100, top_p=0.95)
    ar.reset_caches()
    assert ids.shape == (2, n) and int(ids.max()) < cfg["codebook_size"]
    video = tokz.decode_from_bottleneck(ids.long())
    assert video.shape == (2, 3, cfg["frame_num"], cfg["input_size"], cfg["input_size"]) and bool(torch.isfinite(video).all())


def test_bench_ar_leg_runs():
    """the `bench.py --ar SIZE` leg (training step + KV-cache generation of the prior) on a short sequence"""
    import bench
    import video_tokenizer_amd as vt
    r = bench.ar_prior_step(vt, "S", steps=2, warmup=1, batch=2, seq=128, gen_batch=2, vocab=512)
    assert r["model"] == "llama-abs-S" and r["train"]["tokens_per_s"] > 0 and np.isfinite(r["train"]["loss"])
    assert r["generate_cfg1"]["new_tokens"] == 128 and r["generate_cfg2"]["tokens_per_s"] > 0
