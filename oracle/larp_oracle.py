"""CPU restatement of the LARP tokenizer encode -> quantize -> decode step.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product path (video-tokenizer_amd/) never
does and fails loudly when its HIP library is missing.

Every function cites the reference lines it restates (paths relative to
/root/reference).  Pinning status:
  * sincos tables, PatchEmbed3D, Bottleneck + SimpleVectorQuantizer (modes L and D):
    pinned against outputs of the reference's own modules, generated in the build
    container by tests/golden/make_golden.py and committed under tests/golden/.
  * the attention+MLP block is `timm.models.vision_transformer.Block`
    (models/transformer.py:3,52-59), a third-party dependency that is absent from
    /root/reference and not installed here (requirements.txt:4 pins no version).  It is
    restated from timm's published Block/Attention/Mlp recipe; no reference test or
    fixture covers it  =>  **parity unpinned** for `block()`; the only structural pin is
    the checkpoint key layout.
  * the full `forward()` composition follows models/larp_tokenizer.py line by line but
    the file itself cannot be imported (needs vjepa2, easydict, flash_attn, timm)  =>
    composition pinned only through its pinned parts.

All arithmetic is torch CPU fp32 (the reference's CPU path).  `emulate_bf16=True`
rounds activations/weights to bf16 at the points where the HIP path does (the points
torch autocast(bf16) would, SURVEY §7 "bf16 autocast numerics"), so GPU results can be
compared at a tight tolerance; gradients flow through the same casts.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# sin-cos position tables  (models/embed.py:269-331)
# --------------------------------------------------------------------------------------


def sincos_1d(embed_dim, pos, scale_factor=10000):
    """embed.py:312-331: [sin(pos*w) | cos(pos*w)], w_i = 1/scale^(i/(D/2)), float64."""
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / scale_factor ** omega
    pos = np.asarray(pos).reshape(-1)
    out = np.einsum("m,d->md", pos, omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim, grid_size):
    """embed.py:283-309: meshgrid(w, h) "w goes first" => first half of the channels
    encodes the column index, second half the row index."""
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape(2, 1, grid_size, grid_size)
    emb_a = sincos_1d(embed_dim // 2, grid[0])
    emb_b = sincos_1d(embed_dim // 2, grid[1])
    return np.concatenate([emb_a, emb_b], axis=1)


def sincos_3d(embed_dim, grid_size, frame_num):
    """embed.py:269-277: 2-D table + (added, not concatenated) 1-D temporal table."""
    emb_2d = sincos_2d(embed_dim, grid_size).reshape(1, grid_size, grid_size, embed_dim)
    emb_1d = sincos_1d(embed_dim, np.arange(frame_num, dtype=np.float32)).reshape(frame_num, 1, 1, embed_dim)
    return (emb_2d + emb_1d).reshape(-1, embed_dim)


# --------------------------------------------------------------------------------------
# casts
# --------------------------------------------------------------------------------------


def _rb(t, on):
    """round-trip through bf16 when emulating the mixed-precision path."""
    return t.to(torch.bfloat16).to(torch.float32) if on else t


def linear(x, w, b=None, emu=False, round_out=True):
    """nn.Linear under autocast(bf16): bf16 inputs, fp32 accumulate, bf16 output."""
    y = F.linear(_rb(x, emu), _rb(w, emu))
    if b is not None:
        y = y + b
    return _rb(y, emu and round_out)


# --------------------------------------------------------------------------------------
# PatchEmbed3D  (models/embed.py:37-116)
# --------------------------------------------------------------------------------------


def patchify(x, pt, p):
    """Gather non-overlapping (pt,p,p) patches in (c,dt,dy,dx) order; token order
    (t,h,w) t-major == Conv3d(kernel=stride=patch) followed by flatten(2).transpose(1,2)
    (embed.py:82,110-112)."""
    b, c, t, h, w = x.shape
    x = x.reshape(b, c, t // pt, pt, h // p, p, w // p, p)
    x = x.permute(0, 2, 4, 6, 1, 3, 5, 7)  # b, T, H, W, c, dt, dy, dx
    return x.reshape(b, (t // pt) * (h // p) * (w // p), c * pt * p * p)


def patch_embed3d(x, weight, bias, emu=False):
    """embed.py:85-116 with norm=Identity, flatten=True. weight (D,C,pt,p,p)."""
    d, c, pt, p, _ = weight.shape
    return linear(patchify(x, pt, p), weight.reshape(d, -1), bias, emu)


# --------------------------------------------------------------------------------------
# timm Block  (third-party; constructed at models/transformer.py:52-59)   PARITY UNPINNED
# --------------------------------------------------------------------------------------


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def attention(x, qkv_w, proj_w, proj_b, num_heads, emu=False):
    """timm Attention: qkv=Linear(D,3D,bias=False) reshaped (B,N,3,H,hd).permute(2,0,3,1,4);
    softmax(q k^T * hd^-0.5) v; proj=Linear(D,D).  No mask, no dropout (p=0)."""
    b, n, d = x.shape
    hd = d // num_heads
    qkv = linear(x, qkv_w, None, emu).reshape(b, n, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.softmax((q @ k.transpose(-2, -1)) * (hd ** -0.5), dim=-1)
    o = _rb(att @ v, emu).transpose(1, 2).reshape(b, n, d)
    # HIP path: projection epilogue adds bias and residual in fp32 (no bf16 round of the branch)
    return linear(o, proj_w, proj_b, emu, round_out=False)


def block(x, p, prefix, num_heads, emu=False):
    """timm Block(dim, heads, mlp_ratio=4, qkv_bias=False): pre-LN, LayerNorm eps 1e-5,
    x = x + attn(norm1(x)); x = x + fc2(gelu(fc1(norm2(x)))).  LayerScale/DropPath are
    Identity at the reference's arguments.  Residual stream stays fp32."""
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], 1e-5)
    x = x + attention(h, p[prefix + "attn.qkv.weight"], p[prefix + "attn.proj.weight"],
                      p[prefix + "attn.proj.bias"], num_heads, emu)
    h = F.layer_norm(x, (d,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], 1e-5)
    u = linear(h, p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"], emu)
    g = _rb(gelu_erf(u), emu)
    x = x + linear(g, p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"], emu, round_out=False)
    return x


def encoder_parallel(context, query, p, prefix, depth, num_heads, emu=False):
    """models/transformer.py:62-70: cat([context, query]) -> blocks -> last len(query) rows."""
    nq = query.shape[1]
    h = torch.cat([context, query], dim=1)
    for i in range(depth):
        h = block(h, p, f"{prefix}blocks.{i}.", num_heads, emu)
    return h[:, -nq:, :]


# --------------------------------------------------------------------------------------
# SimpleVectorQuantizer  (models/bottleneck.py:203-344)
# --------------------------------------------------------------------------------------


def entropy_loss(affinity, temperature=0.01):
    """bottleneck.py:12-33 (loss_type 'softmax')"""
    flat = affinity.reshape(-1, affinity.shape[-1]) / temperature
    probs = F.softmax(flat, dim=-1)
    log_probs = F.log_softmax(flat + 1e-5, dim=-1)
    avg_probs = probs.mean(dim=0)
    avg_entropy = -torch.sum(avg_probs * torch.log(avg_probs + 1e-5))
    sample_entropy = -torch.mean(torch.sum(probs * log_probs, dim=-1))
    return sample_entropy - avg_entropy, sample_entropy, avg_entropy


def vq_forward(z, emb_weight, mode="L", l2_normalized=True, beta=0.25, codebook_w=1.0,
               temperature=0.03, generator=None, force_idx=None, entropy_w=0.0, entropy_temperature=0.01):
    """bottleneck.py:262-324 in fp32.  mode 'L': stochastic=False argmin of the 3-term
    distance (:282-290); 'D': eval-deterministic argmax(softmax(cos/tau)) (:275-278);
    'S': multinomial sample (:280).  Returns the same dict keys."""
    z = z.float()
    if l2_normalized:
        z = F.normalize(z, p=2, dim=-1)
        emb = F.normalize(emb_weight, p=2, dim=-1)
    else:
        emb = emb_weight
    zf = z.reshape(-1, z.shape[-1])
    if mode in ("D", "S"):
        cos_sim = torch.einsum("bd,nd->bn", zf, emb)
        probs = F.softmax(cos_sim * (1.0 / temperature), dim=-1)
        if mode == "D":
            idx = torch.argmax(probs, dim=-1)
        else:
            idx = torch.multinomial(probs, 1, generator=generator).squeeze(-1)
    else:
        dist = (torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1)
                - 2 * torch.einsum("bd,dn->bn", zf, emb.t()))
        idx = torch.argmin(dist, dim=1)
    if force_idx is not None:  # test hook: follow a given discrete path (e.g. the GPU's indices)
        idx = force_idx.reshape(-1).to(torch.int64)
    quantized = F.embedding(idx, emb).view(z.shape)
    loss_commit = ((quantized.detach() - z) ** 2).mean()
    loss_codebook = ((quantized - z.detach()) ** 2).mean()
    zero = torch.tensor(0.0)
    le = se = ae = zero
    if entropy_w > 0:                                       # :298-303 (only defined on the stochastic=False branch, which computes `d`)
        assert mode == "L"
        le, se, ae = entropy_loss(-dist, entropy_temperature)
    loss = beta * loss_commit + codebook_w * loss_codebook + entropy_w * le
    quantized_st = z + (quantized - z).detach()
    return {
        "unregularized_z": z, "emb": emb, "regularized_z": quantized_st,
        "bottleneck_rep": idx.reshape(z.shape[0], z.shape[1]),
        "loss_q": loss, "loss_commit": loss_commit, "loss_codebook": loss_codebook,
        "loss_entropy": le, "per_sample_entropy": se, "codebook_entropy": ae,
    }


def bottleneck_forward(x, p, prefix, mode="L", emu=False, **vq_kw):
    """bottleneck.py:170-188: norm stats, in_linear (+ the LayerNorm of norm = 'ln_d' / 'ln_nd' or the batch norm of 'bn_bn' / 'bn_b',
    :146-159: fp32, autocast off; which one is read off the state dict), regulariser, out_linear."""
    n_first = torch.norm(x[:, 0, :], dim=-1).mean()
    n_last = torch.norm(x[:, -1, :], dim=-1).mean()
    z = linear(x, p[prefix + "in_linear.weight"], p[prefix + "in_linear.bias"], emu)
    if prefix + "norm_layer.running_mean" in p:
        # norm = 'bn_bn' / 'bn_b' (:115-119, 149-156): SyncBatchNorm in training mode = batch statistics (biased variance, eps 1e-5) per
        # latent channel over (batch, tokens), or per (token, channel) over the batch; which of the two is read off the weight's length
        w, b_ = p[prefix + "norm_layer.weight"], p[prefix + "norm_layer.bias"]
        zf = z.float()
        if w.numel() == zf.shape[-1]:
            z = F.batch_norm(zf.transpose(1, 2), None, None, w, b_, True, 0.1, 1e-5).transpose(1, 2)
        else:
            z = F.batch_norm(zf.reshape(zf.shape[0], -1), None, None, w, b_, True, 0.1, 1e-5).reshape(zf.shape)
    elif prefix + "norm_layer.weight" in p:
        w = p[prefix + "norm_layer.weight"]
        z = F.layer_norm(z.float(), tuple(w.shape), w, p[prefix + "norm_layer.bias"], 1e-5)
    reg = vq_forward(z, p[prefix + "regularizer.embedding.weight"], mode, **vq_kw)
    x_hat = linear(reg["regularized_z"], p[prefix + "out_linear.weight"], p[prefix + "out_linear.bias"], emu)
    rep = reg.pop("bottleneck_rep")
    return {"output": x_hat, "bottleneck_rep": rep, "projected_z": z,
            "input_norm_first": n_first, "input_norm_last": n_last, **reg}


def sq_forward(z, emb_weight, beta=0.25, force_idx=None, chunk=2048):
    """`VectorQuantizer.forward` of models/model_new/quantizer/fsq.py:170-207 with l2_norm=True, input_format='blc' -- the
    'sq' bottleneck of LARPTokenizer (larp_tokenizer.py:225-229, 423-428): z / |z|, argmin(-z E^T) over the unit-normalised
    codebook, z_q = normalize(E[idx]), loss = beta * mean_n sum_d (sg(z_q) - z)^2 + mean_n sum_d (z_q - sg(z))^2, output
    z + sg(z_q - z).  PINNED by tests/golden/sq_*.npz (outputs of that class, loaded by file path).  The N x K product is taken
    in row chunks (K = 196 560) -- same arithmetic per row."""
    z = F.normalize(z.float(), dim=-1)
    zf = z.reshape(-1, z.shape[-1])
    emb = F.normalize(emb_weight, dim=-1)
    if force_idx is None:
        with torch.no_grad():
            idx = torch.cat([torch.argmin(-zf[i:i + chunk] @ emb.t(), dim=1) for i in range(0, zf.shape[0], chunk)])
    else:
        idx = force_idx.reshape(-1).to(torch.int64)
    zq = F.normalize(F.embedding(idx, emb_weight), dim=-1).view(z.shape)
    loss = beta * torch.mean(((zq.detach() - z) ** 2).sum(dim=-1)) + torch.mean(((zq - z.detach()) ** 2).sum(dim=-1))
    return {"output": z + (zq - z).detach(), "loss_codebook": loss, "indices": idx.reshape(z.shape[:-1]), "unregularized_z": z}


def vq_decode(indices, emb_weight, l2_normalized=True):
    """bottleneck.py:327-344 get_codebook_entry."""
    zq = F.embedding(indices.reshape(-1), emb_weight)
    if l2_normalized:
        zq = F.normalize(zq, p=2, dim=-1)
    return zq.reshape(*indices.shape, emb_weight.shape[1])


# --------------------------------------------------------------------------------------
# output head + unpatchify  (models/larp_tokenizer.py:31-41, 441-454)
# --------------------------------------------------------------------------------------


def unpatchify(x, pt, p, token_h, c=3):
    """larp_tokenizer.py:441-454: (b, n, pt*p*p*c) with channel LAST inside the patch
    -> 'b t h w pt p1 p2 c -> b c (t pt) (h p1) (w p2)'."""
    b = x.shape[0]
    h = w = token_h
    t = x.shape[1] // (h * w)
    x = x.reshape(b, t, h, w, pt, p, p, c).permute(0, 7, 1, 4, 2, 5, 3, 6)
    return x.reshape(b, c, t * pt, h * p, w * p)


def output_layer(x, p, emu=False):
    """larp_tokenizer.py:37-41: LayerNorm(eps 1e-6) -> Linear.  HIP path keeps the
    head output in fp32 (pixels), so no output rounding."""
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), p["final_layer.norm_final.weight"], p["final_layer.norm_final.bias"], 1e-6)
    return linear(h, p["final_layer.linear.weight"], p["final_layer.linear.bias"], emu, round_out=False)


# --------------------------------------------------------------------------------------
# LARPTokenizer.forward  (models/larp_tokenizer.py:400-428, 456-469, 489-496)
# --------------------------------------------------------------------------------------


def tokenizer_forward(p, cfg, x, mode="L", emu=False, **vq_kw):
    """`p`: state dict in the reference layout (SURVEY §5 checkpoint row); `cfg`: dict with
    encoder_depth, decoder_depth, encoder_num_heads, decoder_num_heads, temporal_patch_size,
    patch_size, token_h (and bottleneck_type == 'vq')."""
    b = x.shape[0]
    wpe = p["x_embedder.proj.weight"]
    if wpe.dim() == 4:                                                  # VideoPatchEmbed (temporal_patch_size 1, embed.py:16-34): Conv2d per frame, tokens (t, h, w)
        wpe = wpe.unsqueeze(2)
    tok = patch_embed3d(x, wpe, p["x_embedder.proj.bias"], emu)
    if "encoder_h_embed" in p:                                          # learned factorised PE (:121-127)
        pe = (p["encoder_h_embed"] + p["encode_w_embed"] + p["encoder_t_embed"]).reshape(1, -1, tok.shape[-1])
    else:
        pe = p["encoder_patch_pe"]
    if "encoder_patch_token_type_embed" in p:
        pe = pe + p["encoder_patch_token_type_embed"]                   # :131-134
    q_emb = p["encoder_latent_query_embed"].unsqueeze(0)
    if "encoder_latent_query_token_type_embed" in p:
        q_emb = q_emb + p["encoder_latent_query_token_type_embed"]      # :147-150
    q_emb = q_emb.repeat(b, 1, 1)                                       # :410
    if _mrope(p):                                                       # train_type 'mrope' (:401-405): no additive PE, Encoder111 = gated RoPE layers
        z = _mrope_stack(torch.cat([q_emb, tok], dim=1), p, "encoder111.", cfg, q_emb.shape[1], emu)[:, : q_emb.shape[1]]
    else:
        tok = tok + pe[:, : tok.shape[1]]                               # :407
        z = encoder_parallel(tok, q_emb, p, "encoder.", cfg["encoder_depth"], cfg["encoder_num_heads"], emu)
    if cfg.get("bottleneck_type", "vq") == "sq":                        # :423-428
        zp = linear(z, p["sq_in_linear.weight"], p["sq_in_linear.bias"], emu)
        sq = sq_forward(zp, p["bottleneck.embedding.weight"], force_idx=vq_kw.get("force_idx"))
        encoded = linear(sq["output"], p["sq_out_linear.weight"], p["sq_out_linear.bias"], emu)
        out = {"encoded": encoded, "loss_codebook": sq["loss_codebook"], "_indices": sq["indices"], "_projected_z": zp}
    elif cfg.get("bottleneck_type", "vq") == "fsq":                     # :412-418: LayerNorm -> Linear(768, 6) -> FSQ([8,8,8,5,5,5]) -> Linear(6, 768)
        from .titok_oracle import fsq as _fsq                            # models/model_new/quantizer/fsq.py:54-131, pinned by tests/golden/fsq_*.npz
        zn = F.layer_norm(z, (z.shape[-1],), p["fsq_norm.weight"], p["fsq_norm.bias"], 1e-5)
        zp = linear(zn, p["fsq_in_linear.weight"], p["fsq_in_linear.bias"], emu)
        codes, idx, bounded = _fsq(zp, (8, 8, 8, 5, 5, 5))
        if vq_kw.get("force_codes") is not None:                         # follow the device's codes downstream (rounding-boundary flips)
            codes = codes + (vq_kw["force_codes"] - codes).detach()
        encoded = linear(codes, p["fsq_out_linear.weight"], p["fsq_out_linear.bias"], emu)
        out = {"encoded": encoded, "_codes": codes, "_indices": idx, "_bounded": bounded, "_projected_z": zp}   # the reference returns {'encoded'} only
    else:
        bo = bottleneck_forward(z, p, "bottleneck.", mode, emu, **vq_kw)     # :420
        encoded = bo.pop("output")
        out = {"encoded": encoded, **bo}
    return {"pred_frames": tokenizer_decode(p, cfg, encoded, emu), **out}


def _mrope(p):
    return "encoder111.model_layers.attn_layer.0.to_qkv.weight" in p


def _mrope_stack(h, p, pre, cfg, n_lat, emu):
    """Encoder111 / Decoder111 (models/model_new/base/blocks.py:1110-1178): the gated RoPE layer stack of titok_oracle over [latents ; grid
    tokens]; depth and heads are read off the state dict, the RoPE grid off the tokenizer's patching"""
    from . import titok_oracle as TO
    depth = sum(1 for k in p if k.startswith(pre + "model_layers.attn_layer.") and k.endswith(".to_qkv.weight"))
    heads = h.shape[-1] // 64
    grid = [cfg["frame_num"] // cfg["temporal_patch_size"], cfg["token_h"], cfg["token_h"]]
    angles = TO.rope_angles(n_lat, grid, head_dim=64)
    return TO.residual_attention_block(h, p, pre + "model_layers.", depth, heads, angles, emu)


def tokenizer_decode(p, cfg, encoded, emu=False):
    """LARPTokenizer.decode (/root/reference/models/larp_tokenizer.py:456-469): latents (b, Nq, D) -> video, on its own -- the
    reference's decode() is an ordinary differentiable method (decoder-only fine-tuning on cached latents)"""
    b = encoded.shape[0]
    lpe = p["decoder_latent_pe"]
    if "decoder_latent_token_type_embed" in p:
        lpe = lpe + p["decoder_latent_token_type_embed"]                # :160-163
    zz = encoded + lpe                                                  # :463-464
    if "decoder_h_embed" in p:                                          # :167-171
        dq = (p["decoder_h_embed"] + p["decoder_w_embed"] + p["decoder_t_embed"]).reshape(1, -1, encoded.shape[-1])
    else:
        dq = p["decoder_patch_query_embed"]
    if "decoder_patch_query_token_type_embed" in p:
        dq = dq + p["decoder_patch_query_token_type_embed"]             # :178
    dq = dq.expand(b, -1, -1)
    if _mrope(p):                                                       # :459-461: no latent PE, Decoder111
        y = _mrope_stack(torch.cat([encoded, dq], dim=1), p, "decoder111.", cfg, encoded.shape[1], emu)[:, encoded.shape[1]:]
    else:
        y = encoder_parallel(zz, dq, p, "decoder.", cfg["decoder_depth"], cfg["decoder_num_heads"], emu)
    y = output_layer(y, p, emu)                                         # :467
    return unpatchify(y, cfg["temporal_patch_size"], cfg["patch_size"], cfg["token_h"]).contiguous()


def init_state_dict(cfg, seed=1234, zero_head=False, query_std=0.02):
    """Build-owned deterministic weights in the reference state-dict layout (shapes per
    larp_tokenizer.py:110,128,141,157,173,177,209-210,217,237 + timm Block names).
    Distributions follow initialize_weights (:249-328) but values come from
    oracle/inputs.py, not torch RNG; the head is xavier unless zero_head (the
    reference zero-inits it, :327-328, which makes every gradient but the head's zero).
    `query_std`: std of encoder_latent_query_embed.  The reference's 0.02 (:281) leaves the latent queries of a freshly
    initialised model nearly identical, so every token quantises to the same 1-6 codes; parity tests pass 1.0 so that the
    end-to-end comparison exercises many codebook rows (a trained model's queries are spread too)."""
    from . import inputs as gen
    D = cfg["hidden"]
    pt, ps = cfg["temporal_patch_size"], cfg["patch_size"]
    th = cfg["input_size"] // ps
    tt = cfg["frame_num"] // pt
    nv = th * th * tt
    nq = cfg["bottleneck_token_num"]
    d, K = cfg["bottleneck_dim"], cfg["codebook_size"]
    sd = {}
    s = [seed]

    def nxt():
        s[0] += 1
        return s[0]

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    sd["x_embedder.proj.weight"] = T(gen.xavier_uniform((D, 3, pt, ps, ps), nxt()))
    sd["x_embedder.proj.bias"] = T(gen.uniform((D,), nxt(), -0.02, 0.02))
    sd["encoder_patch_pe"] = T(sincos_3d(D, th, tt)).float().reshape(1, nv, D)
    sd["encoder_latent_query_embed"] = T(gen.normal((nq, D), nxt(), query_std))
    sd["decoder_latent_pe"] = T(sincos_1d(D, np.arange(nq), cfg.get("latent_pe_scale_factor", 10000))).float().reshape(1, nq, D)
    sd["decoder_patch_query_embed"] = T(sincos_3d(D, th, tt)).float().reshape(1, nv, D)
    sd["decoder_patch_query_token_type_embed"] = T(gen.normal((1, 1, D), nxt(), 0.02))
    for side, depth in (("encoder", cfg["encoder_depth"]), ("decoder", cfg["decoder_depth"])):
        for i in range(depth):
            pre = f"{side}.blocks.{i}."
            sd[pre + "norm1.weight"] = T(gen.uniform((D,), nxt(), 0.9, 1.1))
            sd[pre + "norm1.bias"] = T(gen.uniform((D,), nxt(), -0.05, 0.05))
            sd[pre + "attn.qkv.weight"] = T(gen.xavier_uniform((3 * D, D), nxt()))
            sd[pre + "attn.proj.weight"] = T(gen.xavier_uniform((D, D), nxt()))
            sd[pre + "attn.proj.bias"] = T(gen.uniform((D,), nxt(), -0.02, 0.02))
            sd[pre + "norm2.weight"] = T(gen.uniform((D,), nxt(), 0.9, 1.1))
            sd[pre + "norm2.bias"] = T(gen.uniform((D,), nxt(), -0.05, 0.05))
            sd[pre + "mlp.fc1.weight"] = T(gen.xavier_uniform((4 * D, D), nxt()))
            sd[pre + "mlp.fc1.bias"] = T(gen.uniform((4 * D,), nxt(), -0.02, 0.02))
            sd[pre + "mlp.fc2.weight"] = T(gen.xavier_uniform((D, 4 * D), nxt()))
            sd[pre + "mlp.fc2.bias"] = T(gen.uniform((D,), nxt(), -0.02, 0.02))
    if cfg.get("bottleneck_type", "vq") == "sq":   # larp_tokenizer.py:225-229; the codebook is INPUT DATA supplied by the caller (sd["bottleneck.embedding.weight"])
        sd["sq_in_linear.weight"] = T(gen.xavier_uniform((24, D), nxt()))
        sd["sq_in_linear.bias"] = T(gen.uniform((24,), nxt(), -0.02, 0.02))
        sd["sq_out_linear.weight"] = T(gen.xavier_uniform((D, 24), nxt()))
        sd["sq_out_linear.bias"] = T(gen.uniform((D,), nxt(), -0.02, 0.02))
    elif cfg.get("bottleneck_type", "vq") == "fsq":   # larp_tokenizer.py:219-228 (then initialize_weights xavier-inits every Linear, :318-321)
        sd["fsq_in_linear.weight"] = T(gen.xavier_uniform((6, D), nxt()))
        sd["fsq_in_linear.bias"] = T(gen.uniform((6,), nxt(), -0.02, 0.02))
        sd["fsq_out_linear.weight"] = T(gen.xavier_uniform((D, 6), nxt()))
        sd["fsq_out_linear.bias"] = T(gen.uniform((D,), nxt(), -0.02, 0.02))
        sd["fsq_norm.weight"] = T(gen.uniform((D,), nxt(), 0.9, 1.1))
        sd["fsq_norm.bias"] = T(gen.uniform((D,), nxt(), -0.05, 0.05))
    else:
        sd["bottleneck.in_linear.weight"] = T(gen.xavier_uniform((d, D), nxt()))
        sd["bottleneck.in_linear.bias"] = T(gen.uniform((d,), nxt(), -0.02, 0.02))
        sd["bottleneck.out_linear.weight"] = T(gen.xavier_uniform((D, d), nxt()))
        sd["bottleneck.out_linear.bias"] = T(gen.uniform((D,), nxt(), -0.02, 0.02))
        sd["bottleneck.regularizer.embedding.weight"] = T(gen.kaiming_uniform_codebook(K, d, nxt()))
    sd["final_layer.norm_final.weight"] = T(gen.uniform((D,), nxt(), 0.9, 1.1))
    sd["final_layer.norm_final.bias"] = T(gen.uniform((D,), nxt(), -0.05, 0.05))
    kp = pt * ps * ps * 3
    if zero_head:
        sd["final_layer.linear.weight"] = torch.zeros(kp, D)
        sd["final_layer.linear.bias"] = torch.zeros(kp)
    else:
        sd["final_layer.linear.weight"] = T(gen.xavier_uniform((kp, D), nxt()))
        sd["final_layer.linear.bias"] = T(gen.uniform((kp,), nxt(), -0.02, 0.02))
    return sd


# --------------------------------------------------------------------------------------
# TransformerDiscriminator  (models/loss.py:119-204)   PARITY UNPINNED: loss.py imports `lpips`
# at module top and is not importable here; the pieces it is made of are pinned separately
# (PatchEmbed3D + sincos tables by golden vectors, timm Block by the restatement above).
# --------------------------------------------------------------------------------------


def discriminator_forward(p, cfg, x, emu=False):
    """loss.py:188-201: x_embedder(x) + pos_embed -> cat(cls, .) -> transformer_encoder_fused (nn.Sequential of
    timm Blocks, transformer.py:8-31) -> cls row -> LayerNorm(1e-6) -> Linear(D, 1).  `p` uses the module's
    state-dict keys; cfg: dict(n_heads, n_layers)."""
    b = x.shape[0]
    w = p["x_embedder.proj.weight"]
    w = w if w.dim() == 5 else w.unsqueeze(2)        # temporal_patch_size == 1: VideoPatchEmbed's Conv2d per frame (embed.py:16-34)
    tok = patch_embed3d(x, w, p["x_embedder.proj.bias"], emu) + p["encoder_pos_embed"]
    h = torch.cat([p["cls_token"].expand(b, -1, -1), tok], dim=1)
    for i in range(cfg["n_layers"]):
        h = block(h, p, f"transformer_encoder.blocks.{i}.", cfg["n_heads"], emu)
    d = h.shape[-1]
    z = F.layer_norm(h[:, 0], (d,), p["norm_final.weight"], p["norm_final.bias"], 1e-6)
    return linear(z, p["fc.weight"], p["fc.bias"], emu)


def init_discriminator_state_dict(hidden, n_heads, n_layers, input_size, frame_num, pt, ps, seed=4321):
    """Deterministic weights in TransformerDiscriminator's state-dict layout (loss.py:131-186 for shapes and
    distributions; values from oracle/inputs.py)."""
    from . import inputs as gen
    D = hidden
    th, tt = input_size // ps, frame_num // pt
    s = [seed]

    def nxt():
        s[0] += 1
        return s[0]

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    sd = {"x_embedder.proj.weight": T(gen.xavier_uniform((D, 3, pt, ps, ps) if pt > 1 else (D, 3, ps, ps), nxt())),
          "x_embedder.proj.bias": T(gen.uniform((D,), nxt(), -0.02, 0.02)),
          "cls_token": T(gen.uniform((1, 1, D), nxt(), -0.1, 0.1)),
          "encoder_pos_embed": T(sincos_3d(D, th, tt)).float().reshape(1, th * th * tt, D)}
    for i in range(n_layers):
        pre = f"transformer_encoder.blocks.{i}."
        sd[pre + "norm1.weight"] = T(gen.uniform((D,), nxt(), 0.9, 1.1))
        sd[pre + "norm1.bias"] = T(gen.uniform((D,), nxt(), -0.05, 0.05))
        sd[pre + "attn.qkv.weight"] = T(gen.xavier_uniform((3 * D, D), nxt()))
        sd[pre + "attn.proj.weight"] = T(gen.xavier_uniform((D, D), nxt()))
        sd[pre + "attn.proj.bias"] = T(gen.uniform((D,), nxt(), -0.02, 0.02))
        sd[pre + "norm2.weight"] = T(gen.uniform((D,), nxt(), 0.9, 1.1))
        sd[pre + "norm2.bias"] = T(gen.uniform((D,), nxt(), -0.05, 0.05))
        sd[pre + "mlp.fc1.weight"] = T(gen.xavier_uniform((4 * D, D), nxt()))
        sd[pre + "mlp.fc1.bias"] = T(gen.uniform((4 * D,), nxt(), -0.02, 0.02))
        sd[pre + "mlp.fc2.weight"] = T(gen.xavier_uniform((D, 4 * D), nxt()))
        sd[pre + "mlp.fc2.bias"] = T(gen.uniform((D,), nxt(), -0.02, 0.02))
    sd["norm_final.weight"] = T(gen.uniform((D,), nxt(), 0.9, 1.1))
    sd["norm_final.bias"] = T(gen.uniform((D,), nxt(), -0.05, 0.05))
    sd["fc.weight"] = T(gen.xavier_uniform((1, D), nxt()))
    sd["fc.bias"] = T(gen.uniform((1,), nxt(), -0.02, 0.02))
    return sd


# geometry points of SURVEY §8(d)
CONFIGS = {
    # name: (frame_num, input_size, pt, p, enc_depth, dec_depth, Nq, d)
    "A": dict(frame_num=2, input_size=64, temporal_patch_size=2, patch_size=16, encoder_depth=12, decoder_depth=12, bottleneck_token_num=1024, bottleneck_dim=24),
    "B": dict(frame_num=16, input_size=128, temporal_patch_size=2, patch_size=16, encoder_depth=12, decoder_depth=12, bottleneck_token_num=1024, bottleneck_dim=24),
    "Bp": dict(frame_num=16, input_size=128, temporal_patch_size=4, patch_size=8, encoder_depth=12, decoder_depth=12, bottleneck_token_num=1024, bottleneck_dim=24),
    "C": dict(frame_num=16, input_size=128, temporal_patch_size=4, patch_size=8, encoder_depth=6, decoder_depth=6, bottleneck_token_num=512, bottleneck_dim=16),
    "D": dict(frame_num=16, input_size=128, temporal_patch_size=4, patch_size=8, encoder_depth=6, decoder_depth=6, bottleneck_token_num=1024, bottleneck_dim=16),
    "E": dict(frame_num=16, input_size=256, temporal_patch_size=4, patch_size=8, encoder_depth=6, decoder_depth=6, bottleneck_token_num=1024, bottleneck_dim=16),
    # small case for parity tests the CPU oracle finishes in seconds
    "tiny": dict(frame_num=4, input_size=32, temporal_patch_size=2, patch_size=16, encoder_depth=2, decoder_depth=2, bottleneck_token_num=56, bottleneck_dim=24, codebook_size=512),
}


def make_cfg(name, **over):
    c = dict(hidden=768, encoder_num_heads=12, decoder_num_heads=12, codebook_size=8192,
             latent_pe_scale_factor=10000)
    c.update(CONFIGS[name])
    c.update(over)
    c["token_h"] = c["input_size"] // c["patch_size"]
    return c
