"""CPU restatement of the reference's TiTok-style FSQ autoencoder (`autoencoder_*`, models/model_new/).

TEST INFRASTRUCTURE ONLY (same rule as larp_oracle.py: only tests/, smoke() and bench.py's cpu_baseline leg import it).

Restated from (paths relative to /root/reference):
  * models/model_new/base/rope.py:18-24 apply_rotary_emb, :27-46 get_1d_rotary_pos_embed, :49-84 get_grid,
    :87-105 interleave_freqs, :108-121 get_freqs          -- PINNED: tests/golden/titok_rope.npz holds outputs of that file
    (pure torch/einops, loaded by file path in the build container, tests/golden/make_golden.py)
  * models/model_new/base/transformer.py:11-17 GEGLU, :20-29 ffd, :32-63 Attn, :66-91 ResidualAttentionBlock
    -- PARITY UNPINNED: the module imports flash_attn (third-party, absent, requirements.txt pins no version) at its top, so
    it cannot be loaded; restated from the source text.  flash_attn_func(q, k, v) = softmax(q k^T / sqrt(hd)) v per head
    on [B, L, H, hd] operands, no mask, no dropout (its published contract).
  * models/model_new/base/blocks.py:18-82 Encoder, :85-149 Decoder; models/model_new/autoencoder.py:589-669
    `autoencoder_large`; base/utils.py:6-41 get_model_dims, :44-51 init_weights      -- composition, unpinned beyond its parts
  * models/model_new/quantizer/fsq.py -- PINNED separately (oracle/fsq_oracle.c, tests/golden/fsq_*.npz); restated here
    in torch so gradients flow (straight-through round).

All arithmetic torch CPU fp32; `emu=True` rounds to bf16 where autocast(bf16) materialises bf16 tensors (Linear / conv
outputs, LayerNorm_hd output cast `.to(q)`, rotary output `.type_as(x)`, flash-attn output, sigmoid, gelu and the
elementwise products), which are also the rounding points of the HIP path.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .larp_oracle import _rb, linear, patchify


# ------------------------------------------------------------------------------------------ rope.py
def rotary_angles_1d(dim, pos, theta=10000.0):
    """rope.py:27-46: angle[p, i] = pos[p] * (pi / 2) * theta ** linspace(0, 1, dim // 2)[i], float64"""
    assert dim % 2 == 0
    freqs = theta ** torch.linspace(math.log(1.0, theta), math.log(theta, theta), dim // 2, dtype=torch.float64)
    freqs = freqs * math.pi / 2.0
    return freqs * pos.unsqueeze(-1)


def rope_grid(in_grid, in_tokens):
    """rope.py:49-84: the first in_tokens rows carry (i, i, i); the grid rows carry (t, h, w) + in_tokens"""
    frames, height, width = in_grid
    seq_len = math.prod(in_grid) + in_tokens
    ids = torch.zeros(seq_len, 3, dtype=torch.int64)
    ids[:in_tokens] = torch.arange(in_tokens, dtype=torch.int64).unsqueeze(-1)
    t, h, w = torch.meshgrid(torch.arange(frames), torch.arange(height), torch.arange(width), indexing="ij")
    ids[in_tokens:, 0], ids[in_tokens:, 1], ids[in_tokens:, 2] = t.flatten(), h.flatten(), w.flatten()
    ids[in_tokens:] += in_tokens
    return ids


def interleave(parts):
    """rope.py:87-105: round-robin over the axes (largest first) while the shortest still has entries, then the
    leftovers of the longer ones: T H W T H W ... T T"""
    parts = sorted(parts, key=lambda a: a.shape[-1], reverse=True)
    total = sum(a.shape[-1] for a in parts)
    out = torch.zeros(*parts[0].shape[:-1], total, dtype=parts[0].dtype)
    offset = last = 0
    parts = list(parts)
    for _ in range(len(parts)):
        idx = torch.arange(parts[-1].shape[-1] - offset)
        for i, f in enumerate(parts):
            out[..., idx * len(parts) + i + last] = f[..., idx + offset]
        offset += idx.shape[0]
        last += idx.shape[0] * len(parts)
        parts.pop(-1)
    return out


def rope_angles(in_tokens, in_grid, head_dim=64, theta=10000.0):
    """rope.py:108-121 get_freqs, as angles (freqs_cis = exp(i * angle)): float64 [L, head_dim // 2]"""
    axes = head_dim / 3
    axes = [int(axes - (axes % 2))] * 3
    axes[0] += head_dim - sum(axes)
    grid = rope_grid(in_grid, in_tokens)
    return interleave([rotary_angles_1d(axes[i], grid[:, i], theta) for i in range(3)])


def apply_rotary(x, angles):
    """rope.py:18-24: x [B, L, H, hd] (pairs (2j, 2j+1) are complex numbers) times exp(i angle[L, hd/2]); computed in
    float64 like the reference's complex128 product, returned in fp32"""
    xr = x.double().reshape(*x.shape[:-1], -1, 2)
    c, s = torch.cos(angles).unsqueeze(-2), torch.sin(angles).unsqueeze(-2)   # [L, 1, hd/2]
    re = xr[..., 0] * c - xr[..., 1] * s
    im = xr[..., 0] * s + xr[..., 1] * c
    return torch.stack([re, im], dim=-1).flatten(-2).float()


# ------------------------------------------------------------------------------------------ transformer.py
def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def ffd_inner(dim, mult=4, mult_of=32):
    """transformer.py:20-22"""
    inner = int(mult * (2 / 3) * dim)
    return mult_of * ((inner + mult_of - 1) // mult_of)


def attn(x, p, pre, heads, angles, emu=False):
    """transformer.py:45-63"""
    b, n, d = x.shape
    hd = d // heads
    q, k, v, gate = linear(x, p[pre + "to_qkv.weight"], None, emu).chunk(4, dim=-1)
    q, k, v = (t.reshape(b, n, heads, hd) for t in (q, k, v))
    q = _rb(F.layer_norm(q, (hd,), p[pre + "q_norm.weight"], p[pre + "q_norm.bias"], 1e-5), emu)
    k = _rb(F.layer_norm(k, (hd,), p[pre + "k_norm.weight"], p[pre + "k_norm.bias"], 1e-5), emu)
    q, k = _rb(apply_rotary(q, angles), emu), _rb(apply_rotary(k, angles), emu)
    att = torch.softmax(torch.einsum("blhd,bmhd->bhlm", q, k) * (hd ** -0.5), dim=-1)
    o = _rb(torch.einsum("bhlm,bmhd->blhd", att, v), emu).reshape(b, n, d)
    o = _rb(o * _rb(torch.sigmoid(gate), emu), emu)
    return linear(o, p[pre + "out_proj.weight"], None, emu)


def ffd(x, p, pre, emu=False):
    """transformer.py:20-29: LayerNorm -> Linear(no bias) -> GEGLU -> Linear(no bias)"""
    d = x.shape[-1]
    y = _rb(F.layer_norm(x, (d,), p[pre + "0.weight"], p[pre + "0.bias"], 1e-5), emu)
    h = linear(y, p[pre + "1.weight"], None, emu)
    a, gate = h.chunk(2, dim=-1)
    a = _rb(_rb(gelu_erf(gate), emu) * a, emu)
    return linear(a, p[pre + "3.weight"], None, emu)


def residual_attention_block(x, p, pre, num_layer, heads, angles, emu=False):
    """transformer.py:82-91, incl. the 1/sqrt(i+1) rescale of the whole stream after every layer"""
    for i in range(num_layer):
        x = x + attn(x, p, f"{pre}attn_layer.{i}.", heads, angles, emu)
        x = x + ffd(x, p, f"{pre}ffd_layer.{i}.", emu)
        x = x * (1 / math.sqrt(i + 1))
    return x


# ------------------------------------------------------------------------------------------ fsq.py (torch, differentiable)
def fsq(z, levels):
    lv = torch.tensor(levels, dtype=torch.int32)
    half_l = (lv - 1) * (1 + 1e-3) / 2
    offset = torch.where(lv % 2 == 0, 0.5, 0.0)
    shift = (offset / half_l).atanh()
    bounded = (z.float() + shift).tanh() * half_l - offset
    q = bounded + (bounded.round() - bounded).detach()
    half_w = lv // 2
    codes = q / half_w
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1])), dim=0).to(torch.int32)
    idx = ((codes * half_w + half_w) * basis).sum(dim=-1).to(torch.int32)
    return codes, idx, bounded


# ------------------------------------------------------------------------------------------ blocks.py / autoencoder.py
MODEL_DIMS = {"tiny": (256, 4, 4), "small": (512, 8, 8), "base": (768, 12, 12), "large": (1024, 24, 16)}   # width, layers, heads


def autoencoder_forward(p, cfg, video, emu=False, force_codes=None):
    """autoencoder.py:650-669 with Encoder/Decoder of blocks.py.  cfg: width, layers, heads, patch (pt, p, p), grid (T', H', W'),
    tokens, levels.  Returns pred_frames, z (encoder output), codes, indices, bounded."""
    b = video.shape[0]
    width, heads, layers = cfg["width"], cfg["heads"], cfg["layers"]
    pt, ps = cfg["patch"][0], cfg["patch"][1]
    n_lat, n_grid = cfg["tokens"], math.prod(cfg["grid"])
    ang = rope_angles(n_lat, cfg["grid"], width // heads)
    # Encoder (blocks.py:57-82)
    w = p["encoder.proj_in.weight"]
    tok = linear(patchify(video, pt, ps), w.reshape(width, -1), p["encoder.proj_in.bias"], emu)
    x = torch.cat([p["encoder.mask_token"].expand(b, n_lat, width), tok], dim=1)
    x = residual_attention_block(x, p, "encoder.model_layers.", layers, heads, ang, emu)
    z = linear(x[:, :n_lat], p["encoder.proj_out.weight"], p["encoder.proj_out.bias"], emu)
    codes, idx, bounded = fsq(z, cfg["levels"])
    if force_codes is not None:                      # follow the device's codes through the decoder (near-tie flips)
        codes = codes + (force_codes - codes).detach()
    # Decoder (blocks.py:125-149)
    y = linear(codes, p["decoder.proj_in.weight"], p["decoder.proj_in.bias"], emu)
    y = torch.cat([y, p["decoder.mask_token"].expand(b, n_grid, width)], dim=1)
    y = residual_attention_block(y, p, "decoder.model_layers.", layers, heads, ang, emu)
    y = y[:, n_lat:]
    # ConvTranspose3d(kernel = stride = patch): out patch (c, dt, dy, dx) = y . W[width, c*pt*p*p] + bias[c]
    wt = p["decoder.proj_out.weight"]                # [width, 3, pt, p, p]
    bias = p["decoder.proj_out.bias"].repeat_interleave(pt * ps * ps)
    rows = linear(y, wt.reshape(width, -1).t(), bias, emu)                   # [B, n_grid, 3*pt*p*p]
    t_, h_, w_ = cfg["grid"]
    pred = rows.reshape(b, t_, h_, w_, 3, pt, ps, ps).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(b, 3, t_ * pt, h_ * ps, w_ * ps)
    return {"pred_frames": pred, "z": z, "codes": codes, "indices": idx, "bounded": bounded}


# ------------------------------------------------------------------------------------------ first-token family (autoencoder.py:672-913)
def rope_angles_unify(cond_tokens, in_tokens, grid, head_dim=64, theta=10000.0):
    """Angles for Decoder_unify's sequence [cond latents | latents | grid].  get_freqs_multi (rope.py:124-146) builds pair i =
    get_grid(grid_i, tokens_i) + max(pair i-1); the reference calls it with the hard-coded pairs [[256, [1,16,16]], [1024,
    [4,16,16]]] (blocks.py:724) = 2560 rows for a 2304 / 2048 / 1792-row sequence and raises in apply_rotary_emb, so NO reference
    output exists for this decoder (parity unpinned AND deviating).  Restated fix, identical to the product's
    (video-tokenizer_amd/titok.py::rope_positions_unify): pair 0's latent rows only (its grid rows have no token in the decoder),
    then pair 1 with the model's own in_tokens."""
    axes = head_dim / 3
    axes = [int(axes - (axes % 2))] * 3
    axes[0] += head_dim - sum(axes)
    g0 = rope_grid([1, grid[1], grid[2]], cond_tokens)
    g1 = rope_grid(list(grid), in_tokens) + g0.max()
    pos = torch.cat([g0[:cond_tokens], g1], dim=0)
    return interleave([rotary_angles_1d(axes[i], pos[:, i], theta) for i in range(3)])


def encoder_forward(p, pre, video, width, heads, layers, patch, grid, n_lat, emu=False):
    """blocks.py:57-82 (mask token of any of the three shapes: (1,1,1), (1,1,width), (1,n,width))"""
    b = video.shape[0]
    ang = rope_angles(n_lat, grid, width // heads)
    w = p[pre + "proj_in.weight"]
    tok = linear(patchify(video, patch[0], patch[1]), w.reshape(width, -1), p[pre + "proj_in.bias"], emu)
    x = torch.cat([p[pre + "mask_token"].expand(b, n_lat, width), tok], dim=1)
    x = residual_attention_block(x, p, pre + "model_layers.", layers, heads, ang, emu)
    return linear(x[:, :n_lat], p[pre + "proj_out.weight"], p[pre + "proj_out.bias"], emu)


def conv_transpose_patch(y, wt, bias3, patch, grid, emu=False):
    """ConvTranspose3d(kernel = stride = patch): out patch (c, dt, dy, dx) = y . W[width, c*pt*p*p] + bias[c]"""
    b, width = y.shape[0], wt.shape[0]
    pt, ps = patch[0], patch[1]
    bias = bias3.repeat_interleave(pt * ps * ps)
    rows = linear(y, wt.reshape(width, -1).t(), bias, emu)
    t_, h_, w_ = grid
    return rows.reshape(b, t_, h_, w_, 3, pt, ps, ps).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(b, 3, t_ * pt, h_ * ps, w_ * ps)


def first_token_forward(p, cfg, video, emu=False, force_codes=None, force_first=None):
    """AutoEncoder_first_token.forward (autoencoder.py:717-752): video encoder + first-frame encoder (patch (1, p, p)) -> ONE shared
    FSQ -> Decoder_unify(codes, first-frame codes) (blocks.py:758-787, with the rotary fix of rope_angles_unify).
    cfg: enc = (width, layers, heads), dec = (width, layers, heads), patch, grid, tokens, cond_tokens, levels."""
    b = video.shape[0]
    (ew, el, eh), (dw, dl, dh) = cfg["enc"], cfg["dec"]
    patch, grid, n_lat, n_cond = cfg["patch"], cfg["grid"], cfg["tokens"], cfg["cond_tokens"]
    z = encoder_forward(p, "encoder.", video, ew, eh, el, patch, grid, n_lat, emu)
    z1 = encoder_forward(p, "encoder1.", video[:, :, 0:1], ew, eh, el, [1, patch[1], patch[2]], [1, grid[1], grid[2]], n_cond, emu)
    codes, idx, _ = fsq(z, cfg["levels"])
    codes1, idx1, _ = fsq(z1, cfg["levels"])
    if force_codes is not None:
        codes = codes + (force_codes - codes).detach()
    if force_first is not None:
        codes1 = codes1 + (force_first - codes1).detach()
    n_grid = math.prod(grid)
    c = linear(codes1, p["decoder.proj_cond.weight"], p["decoder.proj_cond.bias"], emu)
    y = linear(codes, p["decoder.proj_in.weight"], p["decoder.proj_in.bias"], emu)
    y = torch.cat([c, y, p["decoder.mask_token"].expand(b, n_grid, dw)], dim=1)
    ang = rope_angles_unify(n_cond, n_lat, grid, dw // dh)
    y = residual_attention_block(y, p, "decoder.model_layers.", dl, dh, ang, emu)
    pred = conv_transpose_patch(y[:, n_cond + n_lat:], p["decoder.proj_out.weight"], p["decoder.proj_out.bias"], patch, grid, emu)
    return {"pred_frames": pred, "z": z, "z1": z1, "codes": codes, "codes1": codes1, "indices": idx, "indices1": idx1}


THIN_DIMS = {"tiny_thin": (512, 2, 8, 2.0), "small_thin": (768, 5, 12, 2.0), "base_thin": (1024, 7, 16, 2.0)}   # utils.py:7-20: width, layers, heads, mlp_ratio


def make_first_token_cfg(enc="tiny", dec="tiny", frames=8, side=32, patch=(4, 8, 8), tokens=32, cond_tokens=16, levels=(8, 8, 8, 5, 5, 5)):
    def dims(size):
        if size.endswith("_thin"):
            return THIN_DIMS[size]
        w, l, h = MODEL_DIMS[size]
        return (w, l, h, 4.0)
    e, d = dims(enc), dims(dec)
    grid = [frames // patch[0], side // patch[1], side // patch[2]]
    return dict(enc=e[:3], dec=d[:3], enc_mlp=e[3], dec_mlp=d[3], patch=list(patch), grid=grid, tokens=tokens, cond_tokens=cond_tokens,
                levels=list(levels), frames=frames, side=side)


def init_first_token_state_dict(cfg, seed=888):
    """deterministic weights in the reference's key layout for AutoEncoder_first_token (encoder.*, encoder1.*, decoder.* incl.
    decoder.proj_cond.*); distributions as init_state_dict below"""
    from . import inputs as gen
    s = [seed]

    def nxt():
        s[0] += 1
        return s[0]

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    d = len(cfg["levels"])
    pt, ps = cfg["patch"][0], cfg["patch"][1]
    sd = {}

    def layers(side, width, n, mlp):
        inner = ffd_inner(width, mlp)
        sd[f"{side}.mask_token"] = T(gen.normal((1, 1, 1), nxt(), width ** -0.5))
        for i in range(n):
            a = f"{side}.model_layers.attn_layer.{i}."
            sd[a + "to_qkv.weight"] = T(gen.normal((4 * width, width), nxt(), 0.02))
            for nm in ("q_norm", "k_norm"):
                sd[a + nm + ".weight"] = T(1.0 + gen.normal((64,), nxt(), 0.05))
                sd[a + nm + ".bias"] = T(gen.normal((64,), nxt(), 0.05))
            sd[a + "out_proj.weight"] = T(gen.normal((width, width), nxt(), 0.02))
            f = f"{side}.model_layers.ffd_layer.{i}."
            sd[f + "0.weight"] = T(1.0 + gen.normal((width,), nxt(), 0.05))
            sd[f + "0.bias"] = T(gen.normal((width,), nxt(), 0.05))
            sd[f + "1.weight"] = T(gen.normal((2 * inner, width), nxt(), 0.02))
            sd[f + "3.weight"] = T(gen.normal((width, inner), nxt(), 0.02))
    (ew, el, eh), (dw, dl, dh) = cfg["enc"], cfg["dec"]
    for side, pk in (("encoder", (pt, ps, ps)), ("encoder1", (1, ps, ps))):
        layers(side, ew, el, cfg["enc_mlp"])
        sd[f"{side}.proj_in.weight"] = T(gen.xavier_uniform((ew, 3) + pk, nxt()))
        sd[f"{side}.proj_in.bias"] = T(gen.uniform((ew,), nxt(), -0.02, 0.02))
        sd[f"{side}.proj_out.weight"] = T(gen.normal((d, ew), nxt(), 0.05))
        sd[f"{side}.proj_out.bias"] = T(gen.uniform((d,), nxt(), -0.02, 0.02))
    layers("decoder", dw, dl, cfg["dec_mlp"])
    for nm in ("proj_in", "proj_cond"):
        sd[f"decoder.{nm}.weight"] = T(gen.normal((dw, d), nxt(), 0.05))
        sd[f"decoder.{nm}.bias"] = T(gen.uniform((dw,), nxt(), -0.02, 0.02))
    sd["decoder.proj_out.weight"] = T(gen.xavier_uniform((dw, 3, pt, ps, ps), nxt()))
    sd["decoder.proj_out.bias"] = T(gen.uniform((3,), nxt(), -0.02, 0.02))
    return sd


def make_cfg(size="tiny", frames=8, side=32, patch=(4, 8, 8), tokens=32, levels=(8, 8, 8, 5, 5, 5)):
    width, layers, heads = MODEL_DIMS[size]
    grid = [frames // patch[0], side // patch[1], side // patch[2]]
    return dict(size=size, width=width, layers=layers, heads=heads, patch=list(patch), grid=grid, tokens=tokens, levels=list(levels),
                frames=frames, side=side)


def init_state_dict(cfg, seed=777):
    """Build-owned deterministic weights in the reference's key layout.  Distributions follow init_weights (utils.py:44-51:
    trunc-normal(0.02) Linear weights, zero biases, unit LayerNorm, xavier convs) except that biases / LayerNorm affine
    parameters are perturbed so that their gradients and code paths are exercised; mask_token ~ width^-0.5 * N(0, 1)."""
    from . import inputs as gen
    width, layers = cfg["width"], cfg["layers"]
    pt, ps = cfg["patch"][0], cfg["patch"][1]
    inner = ffd_inner(width)
    d = len(cfg["levels"])
    s = [seed]

    def nxt():
        s[0] += 1
        return s[0]

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    sd = {}
    for side in ("encoder", "decoder"):
        sd[f"{side}.mask_token"] = T(gen.normal((1, 1, 1), nxt(), width ** -0.5))
        for i in range(layers):
            a = f"{side}.model_layers.attn_layer.{i}."
            sd[a + "to_qkv.weight"] = T(gen.normal((4 * width, width), nxt(), 0.02))
            for n in ("q_norm", "k_norm"):
                sd[a + n + ".weight"] = T(1.0 + gen.normal((64,), nxt(), 0.05))
                sd[a + n + ".bias"] = T(gen.normal((64,), nxt(), 0.05))
            sd[a + "out_proj.weight"] = T(gen.normal((width, width), nxt(), 0.02))
            f = f"{side}.model_layers.ffd_layer.{i}."
            sd[f + "0.weight"] = T(1.0 + gen.normal((width,), nxt(), 0.05))
            sd[f + "0.bias"] = T(gen.normal((width,), nxt(), 0.05))
            sd[f + "1.weight"] = T(gen.normal((2 * inner, width), nxt(), 0.02))
            sd[f + "3.weight"] = T(gen.normal((width, inner), nxt(), 0.02))
    sd["encoder.proj_in.weight"] = T(gen.xavier_uniform((width, 3, pt, ps, ps), nxt()))
    sd["encoder.proj_in.bias"] = T(gen.uniform((width,), nxt(), -0.02, 0.02))
    sd["encoder.proj_out.weight"] = T(gen.normal((d, width), nxt(), 0.05))
    sd["encoder.proj_out.bias"] = T(gen.uniform((d,), nxt(), -0.02, 0.02))
    sd["decoder.proj_in.weight"] = T(gen.normal((width, d), nxt(), 0.05))
    sd["decoder.proj_in.bias"] = T(gen.uniform((width,), nxt(), -0.02, 0.02))
    sd["decoder.proj_out.weight"] = T(gen.xavier_uniform((width, 3, pt, ps, ps), nxt()))
    sd["decoder.proj_out.bias"] = T(gen.uniform((3,), nxt(), -0.02, 0.02))
    return sd
