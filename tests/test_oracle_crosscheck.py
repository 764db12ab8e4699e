"""Independent cross-checks of the oracle's PARITY-UNPINNED pieces.

`timm.models.vision_transformer.Block` (models/transformer.py:52-59) and `flash_attn_func`
(models/model_new/base/transformer.py:6,56) are third-party packages that are absent here, so no reference output can
pin `oracle.larp_oracle.block/attention`, `discriminator_forward` or `oracle.titok_oracle.attn`.  What IS importable
is torch's own, separately written implementation of the same published recipes:
  * `torch.nn.TransformerEncoderLayer(norm_first=True, activation='gelu')` == pre-LN MHSA + pre-LN MLP(GELU-erf),
    LayerNorm eps 1e-5, softmax(q k^T / sqrt(hd)) v -- timm's Block at the reference's arguments
    (qkv_bias=False -> zero in_proj bias; no LayerScale / DropPath / dropout);
  * `torch.nn.functional.scaled_dot_product_attention` == timm Attention's fused kernel call and flash_attn_func's
    contract (no mask, no dropout, scale hd^-0.5).
Forward values, input gradients and every weight gradient must agree to 1e-5 in fp32.  CPU only.
"""
import math

import torch
import torch.nn.functional as F

from oracle import inputs as gen
from oracle import larp_oracle as O
from oracle import titok_oracle as TO


def _t(shape, seed, std=1.0):
    return torch.from_numpy(gen.normal(shape, seed, std))


def _block_params(d, seed):
    s = [seed]

    def nxt():
        s[0] += 1
        return s[0]
    p = {"norm1.weight": 1.0 + _t((d,), nxt(), 0.1), "norm1.bias": _t((d,), nxt(), 0.1),
         "attn.qkv.weight": _t((3 * d, d), nxt(), d ** -0.5), "attn.proj.weight": _t((d, d), nxt(), d ** -0.5),
         "attn.proj.bias": _t((d,), nxt(), 0.1),
         "norm2.weight": 1.0 + _t((d,), nxt(), 0.1), "norm2.bias": _t((d,), nxt(), 0.1),
         "mlp.fc1.weight": _t((4 * d, d), nxt(), d ** -0.5), "mlp.fc1.bias": _t((4 * d,), nxt(), 0.1),
         "mlp.fc2.weight": _t((d, 4 * d), nxt(), (4 * d) ** -0.5), "mlp.fc2.bias": _t((d,), nxt(), 0.1)}
    return p


def _torch_layer_from(p, d, heads):
    layer = torch.nn.TransformerEncoderLayer(d, heads, 4 * d, dropout=0.0, activation="gelu", layer_norm_eps=1e-5,
                                             batch_first=True, norm_first=True)
    with torch.no_grad():
        layer.self_attn.in_proj_weight.copy_(p["attn.qkv.weight"])      # rows [q | k | v] x heads x hd: timm's reshape(B,N,3,H,hd)
        layer.self_attn.in_proj_bias.zero_()                            # qkv_bias=False
        layer.self_attn.out_proj.weight.copy_(p["attn.proj.weight"])
        layer.self_attn.out_proj.bias.copy_(p["attn.proj.bias"])
        layer.norm1.weight.copy_(p["norm1.weight"]); layer.norm1.bias.copy_(p["norm1.bias"])
        layer.norm2.weight.copy_(p["norm2.weight"]); layer.norm2.bias.copy_(p["norm2.bias"])
        layer.linear1.weight.copy_(p["mlp.fc1.weight"]); layer.linear1.bias.copy_(p["mlp.fc1.bias"])
        layer.linear2.weight.copy_(p["mlp.fc2.weight"]); layer.linear2.bias.copy_(p["mlp.fc2.bias"])
    return layer.train()   # train mode: the python path (eval mode may take the fused "fast path")


def _close(a, b, tol=1e-5):
    err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    assert err < tol, err


def test_block_equals_torch_transformer_encoder_layer():
    for (d, heads, b, n, seed) in [(64, 4, 2, 19, 100), (96, 12, 1, 33, 200), (768, 12, 1, 8, 300)]:
        p = {k: v.clone().requires_grad_(True) for k, v in _block_params(d, seed).items()}
        x = _t((b, n, d), seed + 50).requires_grad_(True)
        w = _t((b, n, d), seed + 51)
        y = O.block(x, {"blk." + k: v for k, v in p.items()}, "blk.", heads, emu=False)
        (y * w).sum().backward()

        layer = _torch_layer_from(p, d, heads)
        x2 = x.detach().clone().requires_grad_(True)
        y2 = layer(x2)
        (y2 * w).sum().backward()
        _close(y.detach(), y2.detach())
        _close(x.grad, x2.grad)
        pairs = {"attn.qkv.weight": layer.self_attn.in_proj_weight, "attn.proj.weight": layer.self_attn.out_proj.weight,
                 "attn.proj.bias": layer.self_attn.out_proj.bias, "norm1.weight": layer.norm1.weight, "norm1.bias": layer.norm1.bias,
                 "norm2.weight": layer.norm2.weight, "norm2.bias": layer.norm2.bias, "mlp.fc1.weight": layer.linear1.weight,
                 "mlp.fc1.bias": layer.linear1.bias, "mlp.fc2.weight": layer.linear2.weight, "mlp.fc2.bias": layer.linear2.bias}
        for k, q in pairs.items():
            _close(p[k].grad, q.grad)
        assert float(layer.self_attn.in_proj_bias.grad.abs().max()) >= 0.0  # exists; the oracle has no such parameter


def test_attention_equals_sdpa_and_multihead_attention():
    d, heads, b, n = 96, 12, 2, 21       # head_dim 8; the discriminator's head_dim 32 and the tokenizer's 64 below
    for (d, heads) in [(96, 12), (384, 12), (768, 12)]:
        qkv_w, proj_w, proj_b = _t((3 * d, d), 1, d ** -0.5), _t((d, d), 2, d ** -0.5), _t((d,), 3, 0.1)
        x = _t((b, n, d), 4)
        y = O.attention(x, qkv_w, proj_w, proj_b, heads, emu=False)
        hd = d // heads
        qkv = F.linear(x, qkv_w).reshape(b, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], dropout_p=0.0, is_causal=False)   # default scale hd^-0.5
        y2 = F.linear(o.transpose(1, 2).reshape(b, n, d), proj_w, proj_b)
        _close(y, y2)
        mha = torch.nn.MultiheadAttention(d, heads, dropout=0.0, bias=True, batch_first=True)
        with torch.no_grad():
            mha.in_proj_weight.copy_(qkv_w); mha.in_proj_bias.zero_()
            mha.out_proj.weight.copy_(proj_w); mha.out_proj.bias.copy_(proj_b)
        y3, _ = mha(x, x, x, need_weights=False)
        _close(y, y3.detach())


def test_encoder_parallel_and_discriminator_compose_blocks_like_torch():
    """models/transformer.py:62-70 (cat -> blocks -> last len(query) rows) and models/loss.py:188-201 (cls row of a
    fused stack -> LayerNorm(1e-6) -> Linear) built from torch's encoder layers with the same weights."""
    d, heads, depth, b = 64, 4, 3, 2
    sd, layers = {}, []
    for i in range(depth):
        p = _block_params(d, 1000 + 37 * i)
        layers.append(_torch_layer_from(p, d, heads))
        sd.update({f"enc.blocks.{i}.{k}": v for k, v in p.items()})
    ctx, qry = _t((b, 7, d), 5), _t((b, 5, d), 6)
    y = O.encoder_parallel(ctx, qry, sd, "enc.", depth, heads, emu=False)
    h = torch.cat([ctx, qry], dim=1)
    with torch.no_grad():
        for layer in layers:
            h = layer(h)
    _close(y, h[:, -5:])

    # discriminator: hidden 96, 12 heads (head_dim 8), 2 layers, 4x16x16 clips, pt 2, p 8
    hid, nh, nl = 96, 12, 2
    dsd = O.init_discriminator_state_dict(hid, nh, nl, 16, 4, 2, 8, seed=99)
    x = torch.from_numpy(gen.video_clips(2, 4, 16, 77))
    logit = O.discriminator_forward(dsd, {"n_heads": nh, "n_layers": nl}, x, emu=False)
    conv = torch.nn.Conv3d(3, hid, kernel_size=(2, 8, 8), stride=(2, 8, 8))
    with torch.no_grad():
        conv.weight.copy_(dsd["x_embedder.proj.weight"]); conv.bias.copy_(dsd["x_embedder.proj.bias"])
        tok = conv(x).flatten(2).transpose(1, 2) + dsd["encoder_pos_embed"]
        h = torch.cat([dsd["cls_token"].expand(2, -1, -1), tok], dim=1)
        for i in range(nl):
            p = {k[len(f"transformer_encoder.blocks.{i}."):]: v for k, v in dsd.items() if k.startswith(f"transformer_encoder.blocks.{i}.")}
            h = _torch_layer_from(p, hid, nh)(h)
        z = F.layer_norm(h[:, 0], (hid,), dsd["norm_final.weight"], dsd["norm_final.bias"], 1e-6)
        ref = F.linear(z, dsd["fc.weight"], dsd["fc.bias"])
    _close(logit, ref, 2e-5)


def test_gated_attention_layer_core_equals_sdpa():
    """models/model_new/base/transformer.py:45-63: between the per-head LayerNorm + rotary embedding and the sigmoid gate the
    layer calls flash_attn_func(q, k, v) on [B, L, H, hd]; the oracle's einsum softmax must equal torch's SDPA on the same
    rotated operands, and the whole layer must equal a composition of torch primitives written here independently."""
    width, heads, b = 256, 4, 2
    cfg = TO.make_cfg("tiny", frames=8, side=16, patch=(4, 8, 8), tokens=6)
    sd = TO.init_state_dict(cfg, seed=31)
    n = cfg["tokens"] + math.prod(cfg["grid"])
    ang = TO.rope_angles(cfg["tokens"], cfg["grid"], width // heads)
    x = _t((b, n, width), 9)
    pre = "encoder.model_layers.attn_layer.0."
    y = TO.attn(x, sd, pre, heads, ang, emu=False)

    hd = width // heads
    q, k, v, gate = F.linear(x, sd[pre + "to_qkv.weight"]).chunk(4, dim=-1)
    q, k, v = (t.reshape(b, n, heads, hd) for t in (q, k, v))
    q = F.layer_norm(q, (hd,), sd[pre + "q_norm.weight"], sd[pre + "q_norm.bias"])
    k = F.layer_norm(k, (hd,), sd[pre + "k_norm.weight"], sd[pre + "k_norm.bias"])
    cis = torch.polar(torch.ones_like(ang), ang)                                  # rope.py:18-24 as complex numbers

    def rot(t):
        tc = torch.view_as_complex(t.double().reshape(*t.shape[:-1], -1, 2))
        return torch.view_as_real(tc * cis.unsqueeze(-2)).flatten(-2).float()
    o = F.scaled_dot_product_attention(rot(q).transpose(1, 2), rot(k).transpose(1, 2), v.transpose(1, 2))
    o = o.transpose(1, 2).reshape(b, n, width) * torch.sigmoid(gate)
    _close(y, F.linear(o, sd[pre + "out_proj.weight"]))
