"""The reference's TiTok-style FSQ autoencoders (`autoencoder_convpatchify`, `autoencoder_convpatchify_greatfsq`,
`autoencoder_large`: models/model_new/autoencoder.py:8-87, 89-170, 589-669; the mask-token variants `autoencoder_mask3`,
`autoencoder_convpatchify_mask2[_greatfsq]`: :173-416; and `autoencoder_first_token_f256t512 / t768 / t1024a` with
`Decoder_unify`: :672-913, base/blocks.py:690-787 -- what the f256t* yamls name) on the MI355X kernels.  SURVEY §8f rank 3.
Not built: `autoencoder_convpatchify_simplytransformer` (a different layer type, base/simpletransformer.py).

Same module tree and state-dict keys as the reference (`encoder.proj_in`, `encoder.mask_token`,
`encoder.model_layers.attn_layer.{i}.{to_qkv,q_norm,k_norm,out_proj}`, `encoder.model_layers.ffd_layer.{i}.{0,1,3}`,
`encoder.proj_out`, `decoder.*`; `quantize` has no entries), same constructor keywords (the reference ignores every
size keyword and hard-codes 16x128x128 clips, (4,8,8) patches, 1024 latent tokens: so do these classes, the private
`_geometry=` override exists for small parity tests), same `encode / decode / decode_indices / forward -> {'pred_frames'}`.

One transformer layer (models/model_new/base/transformer.py:45-91) = 4 MFMA GEMMs + the LARP path's attention and
LayerNorm kernels + the single-pass glue of csrc/vt_gated.hip, composed in `GatedLayer` (forward and hand-written
backward; torch only owns tensors).  Mixed precision = autocast(bf16) as the reference trains: fp32 residual stream,
bf16 GEMM operands with fp32 accumulation, fp32 LayerNorm statistics.  GPU tensors only; B * L must be a multiple of 64.
"""
import math
import os

import numpy as np
import torch
from torch import nn

from . import hip
from .fsq import FSQ
from .functional import Linear as LinearFn, PatchEmbed as PatchEmbedFn, _pad64
from .registry import register

import itertools
_PACK_UID = itertools.count(1)   # identities of the modules that own packed weight copies (see ResidualAttentionBlock._identity)


def get_model_dims(model_size="tiny", head_dim=64, mlp_ratio=4.0):
    """models/model_new/base/utils.py:6-41"""
    if model_size.endswith("_thin"):
        model_size = model_size[:-5]
        layers = {"tiny": 2, "small": 5, "base": 7, "large": 8}[model_size]
        heads = {"tiny": 8, "small": 12, "base": 16, "large": 32}[model_size]
        mlp_ratio = mlp_ratio / 2
    else:
        layers = {"tiny": 4, "small": 8, "base": 12, "large": 24}[model_size]
        heads = {"tiny": 4, "small": 8, "base": 12, "large": 16}[model_size]
    return int(head_dim * heads), layers, heads, mlp_ratio


def ffd_inner_dim(dim, mult=4, mult_of=32):
    """models/model_new/base/transformer.py:20-22"""
    inner = int(mult * (2 / 3) * dim)
    return mult_of * ((inner + mult_of - 1) // mult_of)


def rope_positions(in_tokens, in_grid):
    """get_grid (models/model_new/base/rope.py:49-84): float64 [in_tokens + prod(grid), 3]; latent row i sits at (i, i, i), grid token
    (t, h, w) at (t, h, w) + in_tokens"""
    f, h, w = in_grid
    pos = np.zeros((in_tokens + f * h * w, 3), dtype=np.float64)
    pos[:in_tokens] = np.arange(in_tokens, dtype=np.float64)[:, None]
    tt, hh, ww = np.meshgrid(np.arange(f), np.arange(h), np.arange(w), indexing="ij")
    pos[in_tokens:] = np.stack([tt.ravel(), hh.ravel(), ww.ravel()], axis=1) + in_tokens
    return pos


def rope_tables_from_positions(pos, head_dim=64, theta=10000.0):
    """fp32 (cos, sin) tables [L, head_dim/2] of the 3-axis rotary embedding at positions pos [L, 3] (rope.py:27-46, 87-121): axis
    dims [24, 20, 20] for head_dim 64; per-axis angle = pos * (pi/2) * theta**linspace(0, 1, n) in float64; the three axes are
    interleaved T H W T H W ... with the temporal leftovers last."""
    per = head_dim / 3
    dims = [int(per - (per % 2))] * 3
    dims[0] += head_dim - sum(dims)
    axes = []
    for a, dim in enumerate(dims):
        fr = torch.linspace(math.log(1.0, theta), math.log(theta, theta), dim // 2, dtype=torch.float64).numpy()
        axes.append(pos[:, a:a + 1] * ((theta ** fr) * math.pi / 2.0)[None, :])
    order = sorted(range(3), key=lambda a: -axes[a].shape[1])            # stable: largest first
    n_short = min(a.shape[1] for a in axes)
    cols = [axes[a][:, j] for j in range(n_short) for a in order]
    longer = [a for a in order if axes[a].shape[1] > n_short]
    assert len(longer) <= 1, "interleave restated for at most one longer axis (true for every head_dim the reference uses)"
    for a in longer:
        cols += [axes[a][:, j] for j in range(n_short, axes[a].shape[1])]
    ang = np.stack(cols, axis=1)
    return torch.from_numpy(np.cos(ang).astype(np.float32)), torch.from_numpy(np.sin(ang).astype(np.float32))


def rope_tables(in_tokens, in_grid, head_dim=64, theta=10000.0):
    """get_freqs (rope.py:108-121) as fp32 (cos, sin) tables [L, head_dim/2], L = in_tokens + prod(grid)"""
    return rope_tables_from_positions(rope_positions(in_tokens, in_grid), head_dim, theta)


def rope_positions_unify(cond_tokens, in_tokens, grid):
    """Positions of Decoder_unify's sequence [cond latents | latents | grid tokens].

    The reference builds its table with get_freqs_multi([[256, [1,16,16]], [1024, [4,16,16]]]) (blocks.py:724-736), i.e. for
    the rows [256 first-frame latents | 256 first-frame grid tokens | 1024 latents | 1024 grid tokens] = 2560 rows, hard-coded,
    while the sequence it is applied to has cond + in_tokens + grid = 2304 / 2048 / 1792 rows: apply_rotary_emb (rope.py:18-24)
    cannot broadcast the two and raises -- as shipped none of the autoencoder_first_token_* models can run a forward pass.
    This build keeps get_freqs_multi's construction (pair i is offset by the maximum coordinate of pair i-1, rope.py:134-136) and
    takes the rows of the tokens that are actually present: the latents of pair 0 (the first-frame GRID rows have no token in the
    decoder), then pair 1 = [in_tokens latents | grid] with the model's own in_tokens.

    PARITY UNPINNED: no reference output exists for these models (their forward raises as shipped), so this repair is this build's
    own reading of the intent; a checkpoint trained with a differently repaired reference would need its own positions --
    `positions=` of Decoder_unify accepts an override ([rows, 3] integer coordinates) for that case.
    """
    p0 = rope_positions(cond_tokens, [1, grid[1], grid[2]])
    p1 = rope_positions(in_tokens, grid) + p0.max()
    return np.concatenate([p0[:cond_tokens], p1], axis=0)


def pack_layer_weights(w_qkv, w_out, w_fc1, w_fc2):
    """bf16 [N, K] copies and [K, N] transposes of a layer's four matrices; fc2's contraction dim padded to a multiple of 64"""
    with torch.no_grad():
        return (*hip.pack_weight(w_qkv), *hip.pack_weight(w_out), *hip.pack_weight(w_fc1),
                *hip.pack_weight(w_fc2, k_pad=_pad64(w_fc2.shape[1])))


class GatedLayer(torch.autograd.Function):
    """x -> (x + Attn(x) ; + ffd(.)) * scale for one layer of ResidualAttentionBlock (transformer.py:45-63, 20-29, 82-91)."""

    @staticmethod
    def forward(ctx, x, cos, sin, n_head, scale, packs, w_qkv, q_w, q_b, k_w, k_b, w_out, ln_w, ln_b, w_fc1, w_fc2):
        hip.require_gpu(x, cos, sin, w_qkv, w_out, w_fc1, w_fc2)
        B, L, D = x.shape
        M = B * L
        if M % 64 or D != 64 * n_head:
            raise hip.HipError(f"GatedLayer: B * L = {M} must be a multiple of 64 and width {D} = 64 * heads")
        inner = w_fc2.shape[1]
        ipad = _pad64(inner)
        x2 = x.contiguous().reshape(M, D).float()
        xb = hip.cast_rows(x2)
        if packs is None:                                            # bf16 operand copies (and their transposes for the dgrads)
            packs = pack_layer_weights(w_qkv, w_out, w_fc1, w_fc2)
        wqkv_b, wqkv_t, wout_b, wout_t, wfc1_b, wfc1_t, wfc2_b, wfc2_t = packs
        qkvg = hip.gemm_nt(xb, wqkv_b, hip.EPI_BF16)
        qkv = hip.qknorm_rope_fwd(qkvg, L, n_head, q_w, q_b, k_w, k_b, 1e-5, cos, sin)
        o, lse = hip.attention_fwd(qkv, B, L, n_head, 64)
        og = hip.sigmoid_gate_fwd(o, qkvg)
        x1 = hip.gemm_nt(og, wout_b, hip.EPI_F32, residual=x2, round_bf16=True)
        y, mean, rstd = hip.layernorm_fwd(x1, ln_w, ln_b, 1e-5)
        h = hip.gemm_nt(y, wfc1_b, hip.EPI_BF16)
        a = hip.geglu_fwd(h, lda=ipad)
        out = hip.gemm_nt(a, wfc2_b, hip.EPI_F32, residual=x1, round_bf16=True)
        if scale != 1.0:
            out.mul_(scale)
        ctx.save_for_backward(xb, qkvg, qkv, o, lse, og, x1, y, mean, rstd, h, a, cos, sin, q_w, k_w, ln_w, wqkv_t, wout_t, wfc1_t, wfc2_t)
        ctx.geom = (B, L, D, n_head, scale, inner)
        return out.reshape(B, L, D)

    @staticmethod
    def backward(ctx, dout):
        xb, qkvg, qkv, o, lse, og, x1, y, mean, rstd, h, a, cos, sin, q_w, k_w, ln_w, wqkv_t, wout_t, wfc1_t, wfc2_t = ctx.saved_tensors
        B, L, D, n_head, scale, inner = ctx.geom
        M = B * L
        dev = dout.device
        d2 = dout.contiguous().reshape(M, D).float()
        if scale != 1.0:
            d2 = d2 * scale
        # ffd backward
        gb = hip.cast_rows(d2)
        da = hip.gemm_nt(gb, wfc2_t, hip.EPI_BF16)                                   # [M, ipad]
        dw_fc2 = torch.empty(D, a.shape[1], device=dev)
        dh = hip.geglu_bwd(da, h)
        dy = hip.gemm_nt(dh, wfc1_t, hip.EPI_BF16)
        dw_fc1 = torch.empty(h.shape[1], D, device=dev)
        dx1, dx1b, d_ln_w, d_ln_b, _ = hip.layernorm_bwd(dy, x1, ln_w, mean, rstd, dres=d2, want_dxsum=False)
        # attention backward
        dog = hip.gemm_nt(dx1b, wout_t, hip.EPI_BF16)
        dw_out = torch.empty(D, D, device=dev)
        dqkvg = torch.empty_like(qkvg)
        d_o = hip.sigmoid_gate_bwd(dog, o, qkvg, dqkvg)
        dqkv = hip.attention_bwd(qkv, o, d_o, lse, B, L, n_head, 64)
        dq_w, dq_b, dk_w, dk_b = hip.qknorm_rope_bwd(qkvg, dqkv, L, n_head, q_w, k_w, 1e-5, cos, sin, dqkvg)
        dx = hip.gemm_nt(dqkvg, wqkv_t, hip.EPI_F32, residual=dx1, round_bf16=True)
        dw_qkv = torch.empty(4 * D, D, device=dev)
        hip.gemm_tn_grouped([dict(A=gb, B=a, out=dw_fc2), dict(A=dh, B=y, out=dw_fc1), dict(A=dx1b, B=og, out=dw_out),
                             dict(A=dqkvg, B=xb, out=dw_qkv)])
        return (dx.reshape(B, L, D), None, None, None, None, None, dw_qkv, dq_w, dq_b, dk_w, dk_b, dw_out, d_ln_w, d_ln_b, dw_fc1,
                dw_fc2[:, :inner].contiguous())


class ConvTransposePatch(torch.autograd.Function):
    """nn.ConvTranspose3d(width, C, kernel = stride = (pt, p, p)) on a token grid (blocks.py:116-147): one GEMM into patch
    rows (c, dt, dy, dx) + the LARP path's unpatchify scatter; bf16-rounded output like the conv under autocast."""

    @staticmethod
    def forward(ctx, tok, weight, bias, geom):
        hip.require_gpu(tok, weight, bias)
        B, C, T, S, pt, p = geom
        width = weight.shape[0]
        M = tok.shape[0] * tok.shape[1]
        if M % 64:
            raise hip.HipError("ConvTransposePatch: B * tokens must be a multiple of 64")
        xb = hip.cast_rows(tok.contiguous().reshape(M, width).float())
        w2 = weight.reshape(width, -1)                                         # [width, Kp]
        wb, wt = hip.pack_weight(w2.t().contiguous())                          # B operand [Kp, width]; wt = [width, Kp]
        rows = hip.gemm_nt(xb, wb, hip.EPI_F32, bias=bias.repeat_interleave(pt * p * p).contiguous(), round_bf16=True)
        ctx.save_for_backward(xb, wt)
        ctx.geom = geom
        return hip.unpatchify(rows, B, C, T, S, pt, p)

    @staticmethod
    def backward(ctx, dvideo):
        xb, wt = ctx.saved_tensors
        B, C, T, S, pt, p = ctx.geom
        dev = dvideo.device
        drows = hip.patchify(dvideo.contiguous().float(), pt, p)              # bf16 [M, Kp], (c, dt, dy, dx) inside a patch
        M, Kp = drows.shape
        width = xb.shape[1]
        dtok = hip.gemm_nt(drows, wt, hip.EPI_F32, round_bf16=True)            # [M, width]
        dw = torch.empty(width, Kp, device=dev)
        hip.gemm_tn_grouped([dict(A=xb, B=drows, out=dw)])
        db = hip.colsum(drows, rows=M).reshape(C, pt * p * p).sum(dim=1)
        return dtok.reshape(B, M // B, width), dw.reshape(width, C, pt, p, p), db, None


class GEGLU(nn.Module):
    """transformer.py:11-17; parameterless -- it keeps the reference's Sequential indices (ffd_layer.{i}.{0,1,3}); the
    arithmetic runs fused inside GatedLayer (vt_geglu_fwd / vt_geglu_bwd)"""

    def forward(self, x):
        raise NotImplementedError("GEGLU runs inside ResidualAttentionBlock's fused layer")


def ffd(dim, mult=4, mult_of=32):
    inner = ffd_inner_dim(dim, mult, mult_of)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner * 2, bias=False), GEGLU(), nn.Linear(inner, dim, bias=False))


class Attn(nn.Module):
    """transformer.py:32-43 (parameters; the forward is GatedLayer)"""

    def __init__(self, dim, heads):
        super().__init__()
        self.dim, self.heads, self.head_dim = dim, heads, dim // heads
        assert self.head_dim == 64, "the kernels are built for head_dim 64 (every reference model size, utils.py:6)"
        self.to_qkv = nn.Linear(dim, dim * 4, bias=False)
        self.q_norm = nn.LayerNorm(self.head_dim)
        self.k_norm = nn.LayerNorm(self.head_dim)
        self.out_proj = nn.Linear(dim, dim, bias=False)


class ResidualAttentionBlock(nn.Module):
    """transformer.py:66-91"""

    def __init__(self, embed_dim=512, heads=8, mlp_ratio=4, num_layer=2):
        super().__init__()
        self.num_layer, self.heads = num_layer, heads
        self.attn_layer = nn.Sequential(*[Attn(embed_dim, heads) for _ in range(num_layer)])
        self.ffd_layer = nn.Sequential(*[ffd(embed_dim, mlp_ratio) for _ in range(num_layer)])
        self._pack_cache = {}
        self._vt_epoch = 0

    def _identity(self):
        """(unique id of THIS module object, invalidation epoch): part of every pack key.  Parameter addresses and `_version`
        counters alone do not identify a weight: the caching allocator hands a second model of the same geometry the addresses of
        a deleted first one, and freshly initialised parameters all carry the same version.  The id is re-drawn after
        copy.deepcopy (the copy's `__dict__` names the original as owner)."""
        d = self.__dict__
        if d.get("_vt_uid_owner") != id(self):
            d["_vt_uid"], d["_vt_uid_owner"] = next(_PACK_UID), id(self)
        return (d["_vt_uid"], self._vt_epoch)

    def invalidate_packs(self):
        """call after writing weights through `.data` / under no_grad in a way that does not bump `_version` (`p.data.copy_`)"""
        self._vt_epoch += 1
        self._pack_cache.clear()

    def _packs(self, i):
        """the layer's bf16 operand copies, re-made only when a weight changed (optimizer step, load_state_dict, .to())"""
        at, ff = self.attn_layer[i], self.ffd_layer[i]
        ws = (at.to_qkv.weight, at.out_proj.weight, ff[1].weight, ff[3].weight)
        key = (self._identity(),) + tuple((w.data_ptr(), w._version) for w in ws)
        hit = self._pack_cache.get(i)
        if hit is None or hit[0] != key:
            hit = (key, pack_layer_weights(*ws))
            self._pack_cache[i] = hit
        return hit[1]

    def forward(self, x, freqs):
        """the whole stack as one engine call per direction (functional.GatedStack -> vt_gated_stack_forward / _backward).
        VT_GATED_PYTHON=1 selects the round-1 composition of one autograd Function per layer instead (A/B and tests)."""
        cos, sin = freqs
        if os.environ.get("VT_GATED_PYTHON") == "1":
            for i in range(self.num_layer):
                at, ff = self.attn_layer[i], self.ffd_layer[i]
                x = GatedLayer.apply(x, cos, sin, self.heads, 1.0 / math.sqrt(i + 1), self._packs(i), at.to_qkv.weight, at.q_norm.weight, at.q_norm.bias,
                                     at.k_norm.weight, at.k_norm.bias, at.out_proj.weight, ff[0].weight, ff[0].bias, ff[1].weight, ff[3].weight)
            return x
        from .functional import GatedStack
        params, key = [], [self._identity()]
        for i in range(self.num_layer):
            at, ff = self.attn_layer[i], self.ffd_layer[i]
            params += [at.to_qkv.weight, at.q_norm.weight, at.q_norm.bias, at.k_norm.weight, at.k_norm.bias, at.out_proj.weight, ff[0].weight, ff[0].bias,
                       ff[1].weight, ff[3].weight]
            key += [(w.data_ptr(), w._version) for w in (at.to_qkv.weight, at.out_proj.weight, ff[1].weight, ff[3].weight)]
        return GatedStack.apply(x, cos, sin, self.heads, tuple(key), *params)


def init_weights(module):
    """models/model_new/base/utils.py:44-51 (ConvTranspose3d is not an nn.Conv3d subclass: it keeps torch's default init)"""
    if isinstance(module, nn.Linear):
        nn.init.trunc_normal_(module.weight.data, mean=0.0, std=0.02)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)
    elif isinstance(module, nn.LayerNorm):
        nn.init.constant_(module.bias, 0)
        nn.init.constant_(module.weight, 1.0)
    elif isinstance(module, (nn.Conv3d, nn.Conv2d)):
        nn.init.xavier_uniform_(module.weight)
        nn.init.zeros_(module.bias)


def _mask_shape(kind, n, width):
    """the reference's block variants differ only in the shape of the learned mask token that is expanded to [B, n, width]:
    Encoder/Decoder (1, 1, 1) (blocks.py:47,103), Encoder4/Decoder4 (1, 1, width) ('mask3', :456,512), Encoder1/Decoder1 (1, n, width)
    ('mask2', :324,380)"""
    return {"scalar": (1, 1, 1), "vector": (1, 1, width), "full": (1, n, width)}[kind]


class _RopeMixin:
    def _freqs(self, device):
        if self._freqs_dev is None or self._freqs_dev[0].device != device:
            self._freqs_dev = (self.freqs[0].to(device), self.freqs[1].to(device))
        return self._freqs_dev


class Encoder(nn.Module, _RopeMixin):
    """blocks.py:18-82: Conv3d patchify, `out_tokens` scalar mask tokens in FRONT of the patch tokens, layers, first
    out_tokens rows -> Linear(width, token_size)"""

    def __init__(self, model_size="tiny", patch_size=(4, 8, 8), in_channels=3, out_channels=5, in_grid=(16, 128, 128), out_tokens=2048, mask="scalar"):
        super().__init__()
        self.patch_size, self.token_size, self.in_channels, self.out_tokens = tuple(patch_size), out_channels, in_channels, out_tokens
        self.grid = [x // y for x, y in zip(in_grid, patch_size)]
        self.width, self.num_layers, self.heads, mlp_ratio = get_model_dims(model_size)
        assert patch_size[1] == patch_size[2] and in_grid[1] == in_grid[2], "square frames and patches (the reference's only geometry)"
        self.proj_in = nn.Conv3d(in_channels, self.width, kernel_size=self.patch_size, stride=self.patch_size, bias=True)
        self.mask_token = nn.Parameter(self.width ** -0.5 * torch.randn(*_mask_shape(mask, out_tokens, self.width)))
        self.freqs = rope_tables(out_tokens, self.grid, head_dim=self.width // self.heads)
        self._freqs_dev = None
        self.model_layers = ResidualAttentionBlock(self.width, self.heads, mlp_ratio, self.num_layers)
        self.proj_out = nn.Linear(self.width, self.token_size, bias=True)
        self.apply(init_weights)

    def forward(self, x):
        B = x.shape[0]
        tok = PatchEmbedFn.apply(x, self.proj_in.weight, self.proj_in.bias, None)
        h = torch.cat([self.mask_token.expand(B, self.out_tokens, self.width), tok], dim=1)
        h = self.model_layers(h, freqs=self._freqs(x.device))
        return LinearFn.apply(h[:, :self.out_tokens], self.proj_out.weight, self.proj_out.bias)


class Encoder111(nn.Module, _RopeMixin):
    """blocks.py:1110-1144, the encoder of LARPTokenizer(train_type='mrope'): [latent queries ; patch tokens] through the gated RoPE layer
    stack, first `out_tokens` rows back.  No projections of its own: the tokenizer's patch embed and bottleneck stay."""

    def __init__(self, model_size="small", patch_size=(4, 8, 8), in_channels=3, out_channels=5, in_grid=(16, 128, 128), out_tokens=1024):
        super().__init__()
        self.patch_size, self.token_size, self.in_channels, self.out_tokens = tuple(patch_size), out_channels, in_channels, out_tokens
        self.grid = [x // y for x, y in zip(in_grid, patch_size)]
        self.width, self.num_layers, self.heads, mlp_ratio = get_model_dims(model_size)
        self.freqs = rope_tables(out_tokens, self.grid, head_dim=self.width // self.heads)
        self._freqs_dev = None
        self.model_layers = ResidualAttentionBlock(self.width, self.heads, mlp_ratio, self.num_layers)
        self.apply(init_weights)

    def forward(self, x, query):
        h = self.model_layers(torch.cat([query.float(), x.float()], dim=1), freqs=self._freqs(x.device))
        return h[:, :self.out_tokens]


class Decoder111(nn.Module, _RopeMixin):
    """blocks.py:1147-1178: [latents ; patch queries] through the same kind of stack, the rows behind the `in_tokens` latents back"""

    def __init__(self, model_size="small", patch_size=(4, 8, 8), in_tokens=1024, out_grid=(16, 128, 128)):
        super().__init__()
        self.patch_size, self.in_tokens = tuple(patch_size), in_tokens
        self.grid = [x // y for x, y in zip(out_grid, patch_size)]
        self.grid_size = math.prod(self.grid)
        self.width, self.num_layers, self.heads, mlp_ratio = get_model_dims(model_size)
        self.freqs = rope_tables(in_tokens, self.grid, head_dim=self.width // self.heads)
        self._freqs_dev = None
        self.model_layers = ResidualAttentionBlock(self.width, self.heads, mlp_ratio, self.num_layers)
        self.apply(init_weights)

    def forward(self, x, masktoken):
        h = self.model_layers(torch.cat([x.float(), masktoken.float()], dim=1), freqs=self._freqs(x.device))
        return h[:, self.in_tokens:]


class Decoder(nn.Module, _RopeMixin):
    """blocks.py:85-149: Linear(token_size, width), grid_size scalar mask tokens BEHIND the latents, layers, last grid_size
    rows -> ConvTranspose3d unpatchify"""

    def __init__(self, model_size="tiny", patch_size=(4, 8, 8), in_channels=5, out_channels=3, in_tokens=2048, out_grid=(32, 256, 256), mask="scalar"):
        super().__init__()
        self.patch_size, self.token_size, self.in_channels, self.in_tokens = tuple(patch_size), in_channels, out_channels, in_tokens
        self.out_grid = tuple(out_grid)
        self.grid = [x // y for x, y in zip(out_grid, patch_size)]
        self.grid_size = math.prod(self.grid)
        self.width, self.num_layers, self.heads, mlp_ratio = get_model_dims(model_size)
        assert patch_size[1] == patch_size[2] and out_grid[1] == out_grid[2], "square frames and patches (the reference's only geometry)"
        self.proj_in = nn.Linear(self.token_size, self.width, bias=True)
        self.mask_token = nn.Parameter(self.width ** -0.5 * torch.randn(*_mask_shape(mask, self.grid_size, self.width)))
        self.freqs = rope_tables(in_tokens, self.grid, head_dim=self.width // self.heads)
        self._freqs_dev = None
        self.model_layers = ResidualAttentionBlock(self.width, self.heads, mlp_ratio, self.num_layers)
        self.proj_out = nn.ConvTranspose3d(self.width, out_channels, kernel_size=self.patch_size, stride=self.patch_size, bias=True)
        self.apply(init_weights)

    def forward(self, x):
        B = x.shape[0]
        h = LinearFn.apply(x, self.proj_in.weight, self.proj_in.bias)
        h = torch.cat([h, self.mask_token.expand(B, self.grid_size, self.width)], dim=1)
        h = self.model_layers(h, freqs=self._freqs(x.device))
        geom = (B, self.in_channels, self.out_grid[0], self.out_grid[1], self.patch_size[0], self.patch_size[1])
        return ConvTransposePatch.apply(h[:, self.in_tokens:], self.proj_out.weight, self.proj_out.bias, geom)


class DecoderUnify(nn.Module, _RopeMixin):
    """`Decoder_unify` (blocks.py:690-787): sequence [proj_cond(first-frame latents) | proj_in(latents) | grid_size mask tokens], layers,
    last grid_size rows -> ConvTranspose3d.  Two shipped quirks are NOT reproduced (see rope_positions_unify and DESIGN.md): the
    hard-coded 2560-row rotary table that makes the reference's forward raise, and the debug print in forward (:776)."""

    def __init__(self, model_size="tiny", patch_size=(4, 8, 8), in_channels=5, out_channels=3, in_tokens=1024, cond_tokens=256, out_grid=(16, 128, 128),
                 positions=None):
        super().__init__()
        self.patch_size, self.token_size, self.in_channels = tuple(patch_size), in_channels, out_channels
        self.in_tokens, self.cond_tokens, self.out_grid = in_tokens, cond_tokens, tuple(out_grid)
        self.grid = [x // y for x, y in zip(out_grid, patch_size)]
        self.grid_size = math.prod(self.grid)
        self.width, self.num_layers, self.heads, mlp_ratio = get_model_dims(model_size)
        assert patch_size[1] == patch_size[2] and out_grid[1] == out_grid[2], "square frames and patches (the reference's only geometry)"
        self.proj_in = nn.Linear(self.token_size, self.width, bias=True)
        if self.cond_tokens > 0:
            self.proj_cond = nn.Linear(self.token_size, self.width, bias=True)
        self.mask_token = nn.Parameter(self.width ** -0.5 * torch.randn(1, 1, 1))
        self.freqs = rope_tables_from_positions(rope_positions_unify(cond_tokens, in_tokens, self.grid) if positions is None else positions,
                                                head_dim=self.width // self.heads)   # `positions`: see rope_positions_unify (parity unpinned)
        self._freqs_dev = None
        self.model_layers = ResidualAttentionBlock(self.width, self.heads, mlp_ratio, self.num_layers)
        self.proj_out = nn.ConvTranspose3d(self.width, out_channels, kernel_size=self.patch_size, stride=self.patch_size, bias=True)
        self.apply(init_weights)

    def forward(self, x, cond=None):
        if self.cond_tokens > 0 and cond is None:
            raise NotImplementedError("Decoder_unify without the first-frame tokens: the rotary table is built for [cond | latents | grid]")
        B = x.shape[0]
        toks = []
        if self.cond_tokens > 0:
            toks.append(LinearFn.apply(cond, self.proj_cond.weight, self.proj_cond.bias))
        toks.append(LinearFn.apply(x, self.proj_in.weight, self.proj_in.bias))
        toks.append(self.mask_token.expand(B, self.grid_size, self.width))
        h = self.model_layers(torch.cat(toks, dim=1), freqs=self._freqs(x.device))
        prefix = self.in_tokens + self.cond_tokens
        geom = (B, self.in_channels, self.out_grid[0], self.out_grid[1], self.patch_size[0], self.patch_size[1])
        return ConvTransposePatch.apply(h[:, prefix:], self.proj_out.weight, self.proj_out.bias, geom)


class _AutoEncoderFirstToken(nn.Module):
    """`AutoEncoder_first_token` (autoencoder.py:672-913; what cfgs/larp_tokenizerf256t512.yaml / t768 / t1024 name): a video encoder,
    a first-frame encoder (patch (1, 8, 8), 256 tokens), ONE shared FSQ, Decoder_unify conditioned on the first-frame codes.
    The reference ignores every constructor keyword; `_geometry` = dict(in_grid, patch_size, tokens, cond_tokens) is for small tests."""
    ENC_SIZE, DEC_SIZE, TOKENS, LEVELS = "base", "base", 512, [8, 8, 8, 5, 5, 5]
    output_format = "bcthw"

    def __init__(self, bottleneck=None, prior_model=None, _geometry=None, **kwargs):
        super().__init__()
        g = dict(in_grid=[16, 128, 128], patch_size=[4, 8, 8], tokens=self.TOKENS, cond_tokens=256)
        g.update(_geometry or {})
        token_size = len(self.LEVELS)
        grid, ps = g["in_grid"], g["patch_size"]
        self.encoder = Encoder(model_size=self.ENC_SIZE, patch_size=ps, in_channels=3, out_channels=token_size, in_grid=grid, out_tokens=g["tokens"])
        self.encoder1 = Encoder(model_size=self.ENC_SIZE, patch_size=[1, ps[1], ps[2]], in_channels=3, out_channels=token_size,
                                in_grid=[1, grid[1], grid[2]], out_tokens=g["cond_tokens"])
        self.quantize = FSQ(levels=self.LEVELS)
        self.decoder = DecoderUnify(model_size=self.DEC_SIZE, patch_size=ps, in_channels=token_size, out_channels=3, in_tokens=g["tokens"],
                                    cond_tokens=g["cond_tokens"], out_grid=grid)
        self.prior_model = None

    def encode(self, data, **kwargs):
        x_q, _ = self.quantize(self.encoder(data))
        first_q, _ = self.quantize(self.encoder1(data[:, :, 0:1]))
        return x_q, first_q

    def decode(self, x, first_token):
        return self.decoder(x, first_token)

    def decode_indices(self, indices, first_indices):
        return self.decoder(self.quantize.indices_to_codes(indices), self.quantize.indices_to_codes(first_indices))

    def forward(self, x):
        x_q, first_q = self.encode(x)
        return {"pred_frames": self.decode(x_q, first_q)}


@register("autoencoder_first_token_f256t512")
class AutoEncoderFirstTokenT512(_AutoEncoderFirstToken):
    ENC_SIZE, DEC_SIZE, TOKENS = "base", "base", 512


@register("autoencoder_first_token_f256t768")
class AutoEncoderFirstTokenT768(_AutoEncoderFirstToken):
    ENC_SIZE, DEC_SIZE, TOKENS = "base", "base", 768


@register("autoencoder_first_token_f256t1024a")
class AutoEncoderFirstTokenT1024(_AutoEncoderFirstToken):
    ENC_SIZE, DEC_SIZE, TOKENS = "small_thin", "small", 1024


# cfgs/larp_tokenizerf256t1024.yaml:37 names `autoencoder_first_token_f256t1024`, which the reference never registers (only
# `...t1024a`, autoencoder.py:672): models.make raises KeyError there.  Registered here as an alias so the shipped yaml resolves.
register("autoencoder_first_token_f256t1024")(AutoEncoderFirstTokenT1024)


class _AutoEncoderBase(nn.Module):
    """autoencoder.py:9-87 / 90-170 / 590-669: every size keyword of the reference constructor is accepted and ignored (the
    reference hard-codes the geometry); `_geometry` = dict(in_grid, patch_size, tokens) overrides it for small tests."""
    MODEL_SIZE, LEVELS, MASK = "small", [8, 8, 8, 5, 5, 5], "scalar"
    output_format = "bcthw"

    def __init__(self, bottleneck=None, prior_model=None, _geometry=None, **kwargs):
        super().__init__()
        g = dict(in_grid=[16, 128, 128], patch_size=[4, 8, 8], tokens=1024)
        g.update(_geometry or {})
        token_size = len(self.LEVELS)
        self.encoder = Encoder(model_size=self.MODEL_SIZE, patch_size=g["patch_size"], in_channels=3, out_channels=token_size,
                               in_grid=g["in_grid"], out_tokens=g["tokens"], mask=self.MASK)
        self.quantize = FSQ(levels=self.LEVELS)
        self.decoder = Decoder(model_size=self.MODEL_SIZE, patch_size=g["patch_size"], in_channels=token_size, out_channels=3,
                               in_tokens=g["tokens"], out_grid=g["in_grid"], mask=self.MASK)
        self.prior_model = None

    def encode(self, data, **kwargs):
        return self.quantize(self.encoder(data))

    def decode(self, x):
        return self.decoder(x)

    def decode_indices(self, indices):
        return self.decoder(self.quantize.indices_to_codes(indices))

    def forward(self, x):
        x_q, _ = self.encode(x)
        return {"pred_frames": self.decode(x_q)}


@register("autoencoder_convpatchify")
class AutoEncoderConvPatchify(_AutoEncoderBase):
    MODEL_SIZE, LEVELS = "small", [8, 8, 8, 5, 5, 5]


@register("autoencoder_convpatchify_greatfsq")
class AutoEncoderConvPatchifyGreatFSQ(_AutoEncoderBase):
    MODEL_SIZE, LEVELS = "base", [8, 8, 8, 8, 5, 5, 5, 5]


@register("autoencoder_large")
class AutoEncoderLarge(_AutoEncoderBase):
    MODEL_SIZE, LEVELS = "large", [8, 8, 8, 5, 5, 5]


@register("autoencoder_convpatchify_mask2")
class AutoEncoderMask2(_AutoEncoderBase):
    """autoencoder.py:256-335: Encoder1 / Decoder1 = one learned mask token per position"""
    MODEL_SIZE, LEVELS, MASK = "base", [8, 8, 8, 5, 5, 5], "full"


@register("autoencoder_convpatchify_mask2_greatfsq")
class AutoEncoderMask2GreatFSQ(_AutoEncoderBase):
    """autoencoder.py:337-416"""
    MODEL_SIZE, LEVELS, MASK = "base", [8, 8, 8, 8, 5, 5, 5, 5], "full"


@register("autoencoder_mask3")
class AutoEncoderMask3(_AutoEncoderBase):
    """autoencoder.py:173-254: Encoder4 / Decoder4 = one learned width-vector shared by all positions"""
    MODEL_SIZE, LEVELS, MASK = "base", [8, 8, 8, 5, 5, 5], "vector"
