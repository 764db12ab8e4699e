"""`LARP_AR`: the autoregressive prior that consumes the tokenizer's `bottleneck_rep`.

Mirrors /root/reference/models/larp_ar.py:33-471 (Llama-style blocks with ABSOLUTE position embeddings: RMSNorm, fused wqkv,
causal attention, SwiGLU FeedForward, class or frame-prediction conditioning, KV cache) and /root/reference/ar/generate.py:12-174
(top-k / top-p sampling, classifier-free guidance, prefill + one-token decode loop).  Same module tree, state-dict keys and
registry names (`llama-abs-S` ... `llama-abs-XXXL`), same `forward(idx, cond_idx, input_pos, targets, mask, valid) -> (logits, loss)`,
`setup_caches / reset_caches / sampling() / sample() / from_checkpoint`.

Every matrix product, RMSNorm, SwiGLU and attention is a libvt_hip call (vt_gemm_nt / vt_gemm_tn_grouped, vt_rmsnorm_*, vt_swiglu_*,
vt_attention_causal_* for training / prefill, vt_decode_attention against the KV cache); torch owns tensors, autograd and the O(B x L x D)
glue the reference also leaves to elementwise ops: embedding gathers, the absolute-PE add, dropout, residual adds, cross-entropy and the
sampling arithmetic over [B, vocab] logits.  Mixed precision = autocast(bf16) (trainers/larp_ar_trainer.py runs under it): fp32
parameters and residual stream, bf16 GEMM operands with fp32 accumulation, Linear outputs rounded to bf16.  GPU tensors only.
Grouped-query attention (n_kv_head < n_head; no llama-abs size uses it) runs on the multi-head kernels through a row-expanded wqkv
(Attention.__init__).  Not built: `emb_masks` in generate (frame-prediction masking of the prefix).
"""
import os
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from .embed import get_1d_sincos_pos_embed_from_grid
from .functional import Linear as LinearFn
from .pretrained import LocalPretrainedMixin
from .registry import models as _registry


def find_multiple(n, k):
    return n if n % k == 0 else n + k - (n % k)


@dataclass
class ModelArgs:
    dim: int = 4096
    n_layer: int = 32
    n_head: int = 32
    n_kv_head: Optional[int] = None
    multiple_of: int = 256
    ffn_dim_multiplier: Optional[float] = None
    rope_base: float = 10000
    norm_eps: float = 1e-5
    initializer_range: float = 0.02
    token_dropout_p: float = 0.1
    attn_dropout_p: float = 0.0
    resid_dropout_p: float = 0.1
    ffn_dropout_p: float = 0.1
    drop_path_rate: float = 0.0
    num_classes: int = 101
    class_dropout_prob: float = 0.1
    model_type: str = "class_cond"
    vocab_size: int = 8192
    cls_token_num: int = 1
    max_batch_size: int = 32
    max_seq_len: int = 1024
    use_fixed_pe: bool = False
    frame_prediction: bool = False


# ------------------------------------------------------------------------------------------------ autograd functions over the kernels
def _pack(mod, *names):
    """bf16 [N, K] and [K, N] operand copies of one weight, or of several concatenated along N; re-made when a weight changes"""
    ws = [getattr(mod, n).weight for n in names]
    key = tuple((w.data_ptr(), w._version) for w in ws)
    cache = mod.__dict__.setdefault("_vt_pack", {})
    hit = cache.get(names)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            w = torch.cat([x.detach() for x in ws], dim=0) if len(ws) > 1 else ws[0].detach()
            hit = (key, hip.pack_weight(w.float().contiguous()))
        cache[names] = hit
    return hit[1]


def _wqkv(at):
    """the [3 dim, dim] weight the attention kernels consume: the stored one, or its grouped-query row expansion (differentiable)"""
    return at.wqkv.weight if at._gqa_rows is None else at.wqkv.weight.index_select(0, at._gqa_rows)


def _pack_qkv(at):
    """bf16 operand copies of _wqkv(at)"""
    if at._gqa_rows is None:
        return _pack(at, "wqkv")
    w = at.wqkv.weight
    key = (w.data_ptr(), w._version)
    hit = at.__dict__.get("_vt_pack_gqa")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            hit = (key, hip.pack_weight(w.detach().index_select(0, at._gqa_rows).float().contiguous()))
        at.__dict__["_vt_pack_gqa"] = hit
    return hit[1]


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


def _rows64(*ts):
    """the weight-gradient GEMMs contract over rows in steps of 64: zero-pad ragged row counts (never at the training shapes)"""
    M = ts[0].shape[0]
    if M % 64 == 0:
        return ts
    out = []
    for t in ts:
        z = torch.zeros((M + 63) // 64 * 64, t.shape[1], device=t.device, dtype=t.dtype)
        z[:M].copy_(t)
        out.append(z)
    return out


class _AttnBranch(torch.autograd.Function):
    """wo(causal_attention(wqkv(rmsnorm(x)))) of TransformerBlock (larp_ar.py:153-213), training / full-sequence path"""

    @staticmethod
    def forward(ctx, x, norm_w, wqkv, wo, n_head, eps, pk_qkv, pk_o):
        hip.require_gpu(x, norm_w, wqkv, wo)
        B, L, D = x.shape
        x2 = x.contiguous().reshape(B * L, D).float()
        norm_w = _f32(norm_w)
        y, rstd = hip.rmsnorm_fwd(x2, norm_w, eps)
        qkv = hip.gemm_nt(y, pk_qkv[0], hip.EPI_BF16)
        o, lse = hip.attention_causal_fwd(qkv, B, L, n_head)
        out = hip.gemm_nt(o, pk_o[0], hip.EPI_F32, round_bf16=True)
        ctx.save_for_backward(x2, norm_w, rstd, y, qkv, o, lse, pk_qkv[1], pk_o[1])
        ctx.geom = (B, L, D, n_head)
        return out.reshape(B, L, D)

    @staticmethod
    def backward(ctx, dout):
        x2, norm_w, rstd, y, qkv, o, lse, wqkv_t, wo_t = ctx.saved_tensors
        B, L, D, H = ctx.geom
        M = B * L
        dev = dout.device
        gb = hip.cast_rows(dout.contiguous().reshape(M, D).float())
        d_o = hip.gemm_nt(gb, wo_t, hip.EPI_BF16)
        dqkv = hip.attention_causal_bwd(qkv, o, d_o, lse, B, L, H)
        dy = hip.gemm_nt(dqkv, wqkv_t, hip.EPI_BF16)
        dwo, dwqkv = torch.empty(D, D, device=dev), torch.empty(3 * D, D, device=dev)
        gb_, o_, dqkv_, y_ = _rows64(gb, o, dqkv, y)
        hip.gemm_tn_grouped([dict(A=gb_, B=o_, out=dwo), dict(A=dqkv_, B=y_, out=dwqkv)])
        dx, _, dnw = hip.rmsnorm_bwd(dy, x2, norm_w, rstd)
        return dx.reshape(B, L, D), dnw, dwqkv, dwo, None, None, None, None


class _FfnBranch(torch.autograd.Function):
    """w2(silu(w1 y) * w3 y), y = rmsnorm(h) (larp_ar.py:122-136, 211); w3 and w1 run as ONE GEMM on the concatenated weight"""

    @staticmethod
    def forward(ctx, x, norm_w, w1, w3, w2, eps, pk_31, pk_2):
        hip.require_gpu(x, norm_w, w1, w3, w2)
        B, L, D = x.shape
        x2 = x.contiguous().reshape(B * L, D).float()
        norm_w = _f32(norm_w)
        y, rstd = hip.rmsnorm_fwd(x2, norm_w, eps)
        h = hip.gemm_nt(y, pk_31[0], hip.EPI_BF16)             # [M, 2I] = [w3 y | w1 y]
        a = hip.swiglu_fwd(h)
        out = hip.gemm_nt(a, pk_2[0], hip.EPI_F32, round_bf16=True)
        ctx.save_for_backward(x2, norm_w, rstd, y, h, a, pk_31[1], pk_2[1])
        ctx.geom = (B, L, D, w1.shape[0])
        return out.reshape(B, L, D)

    @staticmethod
    def backward(ctx, dout):
        x2, norm_w, rstd, y, h, a, w31_t, w2_t = ctx.saved_tensors
        B, L, D, I = ctx.geom
        M = B * L
        dev = dout.device
        gb = hip.cast_rows(dout.contiguous().reshape(M, D).float())
        da = hip.gemm_nt(gb, w2_t, hip.EPI_BF16)
        dh = hip.swiglu_bwd(da, h)
        dy = hip.gemm_nt(dh, w31_t, hip.EPI_BF16)
        dw2, dw31 = torch.empty(D, I, device=dev), torch.empty(2 * I, D, device=dev)
        gb_, a_, dh_, y_ = _rows64(gb, a, dh, y)
        hip.gemm_tn_grouped([dict(A=gb_, B=a_, out=dw2), dict(A=dh_, B=y_, out=dw31)])
        dx, _, dnw = hip.rmsnorm_bwd(dy, x2, norm_w, rstd)
        return dx.reshape(B, L, D), dnw, dw31[I:].contiguous(), dw31[:I].contiguous(), dw2, None, None, None


class _RMSNormFn(torch.autograd.Function):
    """final RMSNorm in front of the output head (larp_ar.py:405); returns the bf16-rounded values in an fp32 tensor"""

    @staticmethod
    def forward(ctx, x, w, eps):
        hip.require_gpu(x, w)
        shp = x.shape
        x2 = x.contiguous().reshape(-1, shp[-1]).float()
        w = _f32(w)
        y, rstd = hip.rmsnorm_fwd(x2, w, eps)
        ctx.save_for_backward(x2, w, rstd)
        return y.float().reshape(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, w, rstd = ctx.saved_tensors
        dx, _, dw = hip.rmsnorm_bwd(hip.cast_rows(dy.contiguous().reshape(x2.shape).float()), x2, w, rstd)
        return dx.reshape(dy.shape), dw, None


# ------------------------------------------------------------------------------------------------ modules (the reference's tree)
class RMSNorm(nn.Module):
    """models/norm.py:6-17"""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return _RMSNormFn.apply(x, self.weight, self.eps)


class LabelEmbedder(nn.Module):
    """models/embed.py:229-259 (class embedding with label dropout for classifier-free guidance)"""

    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes + (dropout_prob > 0), hidden_size)
        self.num_classes, self.dropout_prob = num_classes, dropout_prob

    def forward(self, labels, train, force_drop_ids=None):
        if (train and self.dropout_prob > 0) or force_drop_ids is not None:
            drop = torch.rand(labels.shape[0], device=labels.device) < self.dropout_prob if force_drop_ids is None else force_drop_ids == 1
            labels = torch.where(drop, self.num_classes, labels)
        labels = torch.where(labels < 0, self.num_classes, labels)
        return self.embedding_table(labels)


class FeedForward(nn.Module):
    def __init__(self, config):
        super().__init__()
        hidden = int(2 * (4 * config.dim) / 3)
        if config.ffn_dim_multiplier is not None:
            hidden = int(config.ffn_dim_multiplier * hidden)
        hidden = find_multiple(hidden, config.multiple_of)
        self.w1 = nn.Linear(config.dim, hidden, bias=False)
        self.w3 = nn.Linear(config.dim, hidden, bias=False)
        self.w2 = nn.Linear(hidden, config.dim, bias=False)
        self.ffn_dropout = nn.Dropout(config.ffn_dropout_p)


class KVCache(nn.Module):
    """larp_ar.py:138-151, bf16"""

    def __init__(self, max_batch_size, max_seq_length, n_head, head_dim, dtype=torch.bfloat16):
        super().__init__()
        shape = (max_batch_size, n_head, max_seq_length, head_dim)
        self.register_buffer("k_cache", torch.zeros(shape, dtype=torch.bfloat16), persistent=False)
        self.register_buffer("v_cache", torch.zeros(shape, dtype=torch.bfloat16), persistent=False)

    def update(self, input_pos, k_val, v_val):
        assert input_pos.shape[0] == k_val.shape[2], f"{input_pos.shape[0]} != {k_val.shape[2]}"
        self.k_cache[: k_val.shape[0], :, input_pos] = k_val.to(torch.bfloat16)
        self.v_cache[: v_val.shape[0], :, input_pos] = v_val.to(torch.bfloat16)
        return self.k_cache, self.v_cache


class Attention(nn.Module):
    def __init__(self, config):
        super().__init__()
        assert config.dim % config.n_head == 0
        self.dim, self.n_head = config.dim, config.n_head
        self.head_dim = config.dim // config.n_head
        self.n_kv_head = config.n_kv_head if config.n_kv_head is not None else config.n_head
        if self.head_dim != 64:
            raise NotImplementedError("LARP_AR on this build: head_dim 64 (true for every llama-abs size, larp_ar.py:449-468)")
        if config.attn_dropout_p > 0:
            raise NotImplementedError("attn_dropout_p > 0 is not built (the shipped configs use 0.0)")
        assert self.n_head % self.n_kv_head == 0
        # larp_ar.py:171-175: rows = q (n_head heads) | k (n_kv_head heads) | v (n_kv_head heads)
        self.wqkv = nn.Linear(config.dim, (self.n_head + 2 * self.n_kv_head) * self.head_dim, bias=False)
        # Grouped-query attention (n_kv_head < n_head; no shipped size uses it): `keys.repeat_interleave(n_head // n_kv_head)` (:202-203) is the
        # same function as multi-head attention with every K / V head's weight rows repeated for its group, so the kernels run on that
        # expanded [3 dim, dim] weight (a row gather of the stored one: its gradient sums each group back, which is the GQA gradient).
        # It costs the repeated K / V projections and a KV cache of n_head heads; the stored parameter keeps the reference's shape.
        if self.n_kv_head != self.n_head:
            g, hd = self.n_head // self.n_kv_head, self.head_dim
            kv = (torch.arange(self.n_head) // g).repeat_interleave(hd) * hd + torch.arange(hd).repeat(self.n_head)
            rows = torch.cat([torch.arange(self.dim), self.dim + kv, self.dim + self.n_kv_head * hd + kv])
            self.register_buffer("_gqa_rows", rows, persistent=False)
        else:
            self._gqa_rows = None
        self.wo = nn.Linear(config.dim, config.dim, bias=False)
        self.kv_cache = None
        self.resid_dropout = nn.Dropout(config.resid_dropout_p)


class TransformerBlock(nn.Module):
    def __init__(self, config, drop_path):
        super().__init__()
        self.attention = Attention(config)
        self.feed_forward = FeedForward(config)
        self.attention_norm = RMSNorm(config.dim, eps=config.norm_eps)
        self.ffn_norm = RMSNorm(config.dim, eps=config.norm_eps)
        self.drop_path_prob = float(drop_path)

    def _drop_path(self, x):
        if self.drop_path_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_path_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask.div_(keep)

    def forward(self, x, input_pos=None, cached=False):
        at, ff = self.attention, self.feed_forward
        if cached:                       # inference with a KV cache: prefill (several tokens from position 0) or one-token decode
            return self._forward_cached(x, input_pos)
        a = _AttnBranch.apply(x, self.attention_norm.weight, _wqkv(at), at.wo.weight, at.n_head, self.attention_norm.eps,
                              _pack_qkv(at), _pack(at, "wo"))
        h = x + self._drop_path(at.resid_dropout(a))
        f = _FfnBranch.apply(h, self.ffn_norm.weight, ff.w1.weight, ff.w3.weight, ff.w2.weight, self.ffn_norm.eps, _pack(ff, "w3", "w1"), _pack(ff, "w2"))
        return h + self._drop_path(ff.ffn_dropout(f))

    @torch.no_grad()
    def _forward_cached(self, x, input_pos):
        at, ff = self.attention, self.feed_forward
        B, T, D = x.shape
        H = at.n_head
        M = B * T
        x2 = x.contiguous().reshape(M, D).float()
        fuse = _decode_fusion() if M <= 64 else "0"         # decode-sized row counts: SwiGLU (and optionally RMSNorm) inside the weight-streaming GEMM
        if fuse == "norm":
            qkv = hip.decode_norm_linear(x2, _f32(self.attention_norm.weight), self.attention_norm.eps, _pack_qkv(at)[0])
        else:
            y, _ = hip.rmsnorm_fwd(x2, _f32(self.attention_norm.weight), self.attention_norm.eps)
            qkv = hip.gemm_nt(y, _pack_qkv(at)[0], hip.EPI_BF16)                          # [B * T, 3D] = q | k | v
        if T == 1:      # one new token: cache update + attention in one launch, position read on the device (graph-capturable)
            pos = input_pos if input_pos.dtype == torch.int32 else input_pos.to(torch.int32)
            o = hip.decode_attention_step(qkv, at.kv_cache.k_cache, at.kv_cache.v_cache, pos[-1:])
        else:
            if int(input_pos[0]) != 0 or T != int(input_pos[-1]) + 1:
                raise NotImplementedError("KV-cache prefill is built for a prefix that starts at position 0")
            k, v = (t.reshape(B, T, H, 64).transpose(1, 2) for t in qkv.split(D, dim=-1)[1:])  # [B, H, T, 64]
            at.kv_cache.update(input_pos, k, v)
            o, _ = hip.attention_causal_fwd(qkv, B, T, H)
        h = hip.gemm_nt(o, _pack(at, "wo")[0], hip.EPI_F32, round_bf16=True, residual=x2)
        if fuse == "norm":
            g = hip.decode_norm_linear(h, _f32(self.ffn_norm.weight), self.ffn_norm.eps, _pack_swiglu(ff), mode=1)
        elif fuse == "swiglu":
            y2, _ = hip.rmsnorm_fwd(h, _f32(self.ffn_norm.weight), self.ffn_norm.eps)
            g = hip.decode_norm_linear(y2, None, 0.0, _pack_swiglu(ff), mode=1)
        else:
            y2, _ = hip.rmsnorm_fwd(h, _f32(self.ffn_norm.weight), self.ffn_norm.eps)
            g = hip.swiglu_fwd(hip.gemm_nt(y2, _pack(ff, "w3", "w1")[0], hip.EPI_BF16))
        out = hip.gemm_nt(g, _pack(ff, "w2")[0], hip.EPI_F32, round_bf16=True, residual=h)
        return out.reshape(B, T, D)


def _decode_fusion():
    """VT_AR_FUSED_DECODE: 'swiglu' (default) = SwiGLU in the epilogue of the w3|w1 GEMM; 'norm' = also RMSNorm in the operand load of
    the wqkv / w3|w1 / head GEMMs (fewer graph nodes, but every workgroup re-reads the fp32 rows: measured slower); '0' = neither.
    All three give the same bits."""
    return os.environ.get("VT_AR_FUSED_DECODE", "swiglu")


def _pack_swiglu(ff):
    """bf16 [2I, D]: w3 and w1 interleaved in slabs of 8 + 8 rows, the operand layout of vt_decode_norm_linear(mode 1)"""
    key = tuple((w.data_ptr(), w._version) for w in (ff.w3.weight, ff.w1.weight))
    hit = ff.__dict__.get("_vt_pack_swiglu")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            I, K = ff.w1.weight.shape
            w = torch.cat([ff.w3.weight.detach().reshape(I // 8, 8, K), ff.w1.weight.detach().reshape(I // 8, 8, K)], dim=1).reshape(2 * I, K)
            hit = (key, w.to(torch.bfloat16).contiguous())
        ff.__dict__["_vt_pack_swiglu"] = hit
    return hit[1]


class LARP_AR(nn.Module, LocalPretrainedMixin):   # from_pretrained / save_pretrained: local directory, PyTorchModelHubMixin's layout (larp_ar.py:233)
    def __init__(self, config: ModelArgs):
        super().__init__()
        self.config = config
        self.vocab_size, self.n_layer = config.vocab_size, config.n_layer
        self.max_seq_length, self.num_classes = config.max_seq_len, config.num_classes
        self.model_type, self.cls_token_num = config.model_type, config.cls_token_num
        self.is_sampling = False
        self.frame_prediction = config.frame_prediction
        if self.frame_prediction:
            self.cls_embedding = None
        elif self.model_type == "class_cond":
            self.cls_embedding = LabelEmbedder(config.num_classes, config.dim, config.class_dropout_prob)
        else:
            raise Exception("please check model type")
        self.tok_embeddings = nn.Embedding(config.vocab_size + (1 if self.frame_prediction else 0), config.dim)
        self.tok_dropout = nn.Dropout(config.token_dropout_p)
        dpr = [x.item() for x in torch.linspace(0, config.drop_path_rate, config.n_layer)]
        self.layers = nn.ModuleList([TransformerBlock(config, dpr[i]) for i in range(config.n_layer)])
        self.norm = RMSNorm(config.dim, eps=config.norm_eps)
        self.output = nn.Linear(config.dim, config.vocab_size, bias=False)
        n_pe = config.max_seq_len + config.cls_token_num - 1
        if config.use_fixed_pe:
            pe = get_1d_sincos_pos_embed_from_grid(embed_dim=config.dim, pos=np.arange(n_pe))
            self.register_buffer("abs_pe", torch.from_numpy(pe).float().reshape(1, n_pe, config.dim))
        else:
            self.abs_pe = nn.Parameter(torch.randn(1, n_pe, config.dim) * 0.02)
        self.causal_mask = None
        self.initialize_weights()

    def initialize_weights(self):
        std = self.config.initializer_range
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(mean=0.0, std=std)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.Embedding):
                m.weight.data.normal_(mean=0.0, std=std)
        nn.init.constant_(self.output.weight, 0)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @contextmanager
    def sampling(self):
        self.is_sampling = True
        try:
            yield
        finally:
            self.is_sampling = False

    def setup_caches(self, max_batch_size, max_seq_length, dtype=None):
        assert max_seq_length == self.max_seq_length + self.cls_token_num, f"{max_seq_length} != {self.max_seq_length} + {self.cls_token_num=}"
        max_seq_length = find_multiple(max_seq_length, 8)
        dev = self.device
        for b in self.layers:
            b.attention.kv_cache = KVCache(max_batch_size, max_seq_length, self.config.n_head, 64).to(dev)
        self.causal_mask = torch.tril(torch.ones(max_seq_length, max_seq_length, dtype=torch.bool, device=dev)).unsqueeze(0).repeat(max_batch_size, 1, 1)

    def reset_caches(self):
        for b in self.layers:
            b.attention.kv_cache = None

    def forward(self, idx, cond_idx, input_pos=None, targets=None, mask=None, valid=None):
        """larp_ar.py:346-409.  Cached inference (`input_pos` given): positions must lie in [0, max_seq_len of setup_caches).  A HOST
        `input_pos` is checked here and raises like the reference's index_put would; a DEVICE `input_pos` (the generation loop keeps
        it on the GPU so that a decode step is graph-capturable) cannot be checked without a synchronisation: the decode kernel then
        CLAMPS an out-of-range position to the last cache row instead of faulting -- generate() never produces one."""
        if input_pos is not None and torch.is_tensor(input_pos) and not input_pos.is_cuda and input_pos.numel():
            lmax = self.layers[0].attention.kv_cache.k_cache.shape[2] if getattr(self.layers[0].attention, "kv_cache", None) is not None else None
            lo, hi = int(input_pos.min()), int(input_pos.max())
            if lo < 0 or (lmax is not None and hi >= lmax):
                raise IndexError(f"LARP_AR.forward: input_pos in [{lo}, {hi}] is outside the KV cache (0 .. {lmax - 1 if lmax else '?'})")
            input_pos = input_pos.to(self.tok_embeddings.weight.device)
        if mask is not None:
            raise NotImplementedError("an explicit attention mask is not built: training is causal, cached inference uses positions")
        cached = False
        if idx is not None and cond_idx is not None:          # training or naive inference
            if self.frame_prediction:
                assert cond_idx.ndim == 2
                cond = self.tok_embeddings(cond_idx)
                assert cond.shape[1] == self.cls_token_num
            else:
                cond = self.cls_embedding(cond_idx, train=self.training).unsqueeze(1)[:, : self.cls_token_num]
            h = self.tok_dropout(torch.cat((cond, self.tok_embeddings(idx)), dim=1))
        else:
            if cond_idx is not None:                          # prefill in inference
                if self.frame_prediction:
                    tok = self.tok_embeddings(cond_idx)
                    assert tok.shape[1] == self.cls_token_num
                else:
                    tok = self.cls_embedding(cond_idx, train=self.training).unsqueeze(1)[:, : self.cls_token_num]
            else:                                             # decode_n_tokens (KV cache)
                tok = self.tok_embeddings(idx)
            cached = True
            h = self.tok_dropout(tok)
        if not h.is_cuda:
            raise hip.HipError("LARP_AR: tensors are on the CPU; this build runs on MI355X only (no CPU fallback)")
        h = h + (self.abs_pe[:, input_pos] if self.is_sampling else self.abs_pe[:, : h.shape[1]])
        h = h.float()
        for layer in self.layers:
            h = layer(h, input_pos, cached)
        if cached:
            V = self.output.weight.shape[0]
            h2 = h.reshape(-1, h.shape[-1])
            if h2.shape[0] <= 64 and V % 16 == 0 and _decode_fusion() == "norm":
                logits = hip.decode_norm_linear(h2, _f32(self.norm.weight), self.norm.eps, _pack(self, "output")[0], mode=2)
            else:
                y, _ = hip.rmsnorm_fwd(h2, _f32(self.norm.weight), self.norm.eps)
                logits = hip.gemm_nt(y, _pack(self, "output")[0], hip.EPI_F32, round_bf16=True, out=torch.empty(y.shape[0], (V + 3) // 4 * 4, device=y.device))
            logits = logits[:, :V].reshape(h.shape[0], h.shape[1], V)
        else:
            logits = LinearFn.apply(self.norm(h), self.output.weight, None)
        if self.training or (self.frame_prediction and not self.is_sampling):
            logits = logits[:, self.cls_token_num - 1:].contiguous()
        loss = None
        if valid is not None:
            loss_all = F.cross_entropy(logits.view(-1, logits.size(-1)), targets.view(-1), reduction="none")
            valid_all = valid[:, None].repeat(1, targets.shape[1]).view(-1)
            loss = (loss_all * valid_all).sum() / max(valid_all.sum(), 1)
        elif targets is not None:
            loss = F.cross_entropy(logits.view(-1, logits.size(-1)), targets.view(-1))
        return logits, loss

    @torch.inference_mode()
    def sample(self, c, cfg_scale=2.0, cfg_interval=-1, temperature=1.0, top_k=0, top_p=1.0, seq_length=None):
        seq_length = self.max_seq_length if seq_length is None else seq_length
        with self.sampling():
            return generate(self, c, seq_length, cfg_scale=cfg_scale, cfg_interval=cfg_interval, temperature=temperature, top_k=top_k, top_p=top_p,
                            sample_logits=True)

    @classmethod
    def from_checkpoint(cls, ckpt, load_state_dict=True):
        """larp_ar.py:431-442; files are read with weights_only=True"""
        from . import registry
        if isinstance(ckpt, str):
            assert os.path.exists(ckpt), f"checkpoint {ckpt} does not exist"
            ckpt = torch.load(ckpt, map_location="cpu", weights_only=True)
        else:
            assert isinstance(ckpt, dict), "checkpoint must be a dict or a path to a checkpoint"
        return registry.make(ckpt["model"], load_sd=load_state_dict)


# ------------------------------------------------------------------------------------------------ ar/generate.py
def top_k_top_p_filtering(logits, top_k=0, top_p=1.0, filter_value=-float("Inf"), min_tokens_to_keep=1):
    """ar/generate.py:13-52"""
    if top_k > 0:
        top_k = min(max(top_k, min_tokens_to_keep), logits.size(-1))
        logits = logits.masked_fill(logits < torch.topk(logits, top_k)[0][..., -1, None], filter_value)
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        remove = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1) > top_p
        if min_tokens_to_keep > 1:
            remove[..., :min_tokens_to_keep] = 0
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        logits = logits.masked_fill(remove.scatter(1, sorted_indices, remove), filter_value)
    return logits


def sample(logits, temperature=1.0, top_k=0, top_p=1.0, sample_logits=True):
    """ar/generate.py:55-67"""
    logits = logits[:, -1, :] / max(temperature, 1e-5)
    if top_k > 0 or top_p < 1.0:
        logits = top_k_top_p_filtering(logits, top_k=top_k, top_p=top_p)
    probs = F.softmax(logits.float(), dim=-1)
    idx = torch.multinomial(probs, num_samples=1) if sample_logits else torch.topk(probs, k=1, dim=-1)[1]
    return idx, probs


def _guided(logits, cfg_scale, use_cfg=True):
    if cfg_scale > 1.0:
        cond, uncond = torch.split(logits, len(logits) // 2, dim=0)
        return uncond + (cond - uncond) * cfg_scale if use_cfg else cond
    return logits


GRAPH_WARM = 2          # decode iterations run eagerly before the step is captured
GRAPH_MIN_STEPS = 8     # shorter generations are not worth a capture


@torch.no_grad()
def generate(model, cond, max_new_tokens, emb_masks=None, cfg_scale=1.0, cfg_interval=-1, use_graph=None, **sampling_kwargs):
    """ar/generate.py:126-174: prefill the conditioning token(s), then decode max_new_tokens - 1 tokens one at a time through the KV cache.

    A decode iteration (embedding, n_layer x [RMSNorm, wqkv, cache update + attention, wo, RMSNorm, w3|w1, SwiGLU, w2], head, guidance,
    sampling, position += 1, token -> seq) has no host-side dependency -- the position lives on the device (vt_decode_attention_step) --
    so after GRAPH_WARM eager iterations it is captured ONCE as a hipGraph (torch.cuda.CUDAGraph) and replayed: the loop is launch-bound
    (~100-400 small kernels per token), the replay costs one enqueue.  use_graph=False (or VT_AR_GRAPH=0) keeps the eager loop."""
    if emb_masks is not None:
        raise NotImplementedError("emb_masks (masked frame-prediction prefixes) are not built")
    if model.frame_prediction:
        assert cfg_scale == 1.0, "frame prediction requires cfg_scale=1.0 (no classifier-free guidance)"
        cond_combined, T = cond, cond.shape[1]
    elif model.model_type == "class_cond":
        cond_combined = torch.cat([cond, torch.ones_like(cond) * model.num_classes]) if cfg_scale > 1.0 else cond
        T = 1
    else:
        raise Exception("please check model type")
    T_new = T + max_new_tokens
    B = cond.shape[0]
    dev = cond.device
    model.setup_caches(max_batch_size=B * 2 if cfg_scale > 1.0 else B, max_seq_length=T_new)
    seq = torch.empty((B, T_new), dtype=torch.int, device=dev)
    input_pos = torch.arange(0, T, device=dev)
    logits, _ = model(None, cond_combined, input_pos)
    cur = sample(_guided(logits, cfg_scale), **sampling_kwargs)[0].view(-1, 1).clone()
    seq[:, T:T + 1] = cur
    input_pos = torch.tensor([T], device=dev, dtype=torch.int)

    def one_step(use_cfg):          # everything on the device; `cur`, `input_pos`, `seq` are updated in place
        x = torch.cat([cur, cur]) if cfg_scale > 1.0 else cur
        lg, _ = model(x, cond_idx=None, input_pos=input_pos)
        nxt, _ = sample(_guided(lg, cfg_scale, use_cfg), **sampling_kwargs)
        input_pos.add_(1)
        seq.index_copy_(1, input_pos.long(), nxt.to(torch.int32))     # iteration i fills column T + 1 + i = the incremented position
        cur.copy_(nxt.view(-1, 1))

    n_rest = max_new_tokens - 1
    if use_graph is None:
        use_graph = os.environ.get("VT_AR_GRAPH", "1") != "0"
    if use_graph and os.environ.get("VT_AR_GRAPH") != "1":
        # a replayed decode step reads device scalars written by earlier nodes of the same replay (position, sampled token): exactly the
        # pattern that went stale under the runtime's graph packet capture.  Unless the switch is known to be off, stay eager
        # (VT_AR_GRAPH=1 insists).
        import video_tokenizer_amd as _pkg
        use_graph = _pkg.graph_replay_safe()
    use_graph = use_graph and n_rest >= GRAPH_MIN_STEPS
    graphs = {}
    for i in range(n_rest):
        use_cfg = not (cfg_interval > -1 and i > cfg_interval)
        if not use_graph or i < GRAPH_WARM:
            one_step(use_cfg)
            continue
        g = graphs.get(use_cfg)
        if g is None:
            g = graphs[use_cfg] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                one_step(use_cfg)
        g.replay()
    out = seq[:, T:].clone()
    del graphs
    return out


def _make(n_layer, n_head, dim):
    def f(**kwargs):
        return LARP_AR(ModelArgs(n_layer=n_layer, n_head=n_head, dim=dim, **kwargs))
    return f


larp_ar_models = {"llama-abs-S": _make(12, 6, 384), "llama-abs-B": _make(12, 12, 768), "llama-abs-L": _make(24, 16, 1024),
                  "llama-abs-LP": _make(30, 20, 1280), "llama-abs-XL": _make(36, 20, 1280), "llama-abs-XXL": _make(48, 24, 1536),
                  "llama-abs-XXXL": _make(48, 40, 2560)}
_registry.update(larp_ar_models)
