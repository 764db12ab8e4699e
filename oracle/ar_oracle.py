"""CPU restatement of LARP_AR, the autoregressive prior over the tokenizer's `bottleneck_rep`.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product path (video-tokenizer_amd/larp_ar.py) never does and fails loudly without its HIP library.

Restates /root/reference/models/larp_ar.py (RMSNorm models/norm.py:6-17, LabelEmbedder models/embed.py:229-259) and the
generation loop of /root/reference/ar/generate.py:126-174 as plain functions over a state dict with the reference's keys.
Pinning: forward logits, loss, parameter gradients and greedy generation (KV cache in the reference; full recomputation
here, which is the same function of the prefix) are checked against outputs of the reference's own `LARP_AR` run on the
CPU in the build container (tests/golden/make_golden.py::make_ar -> tests/golden/ar_*.npz).

Arithmetic is torch CPU fp32.  `emu=True` rounds to bf16 where autocast(bf16) does (trainers/larp_ar_trainer.py runs
the model under it): Linear inputs, weights and outputs, the attention output, silu(w1 x) and its product with w3 x;
RMSNorm, the residual stream, softmax statistics and the loss stay fp32.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import inputs as gen
from .larp_oracle import _rb, linear, sincos_1d


def find_multiple(n, k):
    return n if n % k == 0 else n + k - (n % k)


def make_cfg(dim=384, n_layer=2, n_head=6, vocab_size=512, max_seq_len=64, num_classes=10, cls_token_num=1, norm_eps=1e-5,
             multiple_of=256, frame_prediction=False, use_fixed_pe=False, class_dropout_prob=0.1, n_kv_head=None):
    hidden = find_multiple(int(2 * (4 * dim) / 3), multiple_of)                       # larp_ar.py:125-131
    return dict(dim=dim, n_layer=n_layer, n_head=n_head, vocab_size=vocab_size, max_seq_len=max_seq_len, num_classes=num_classes,
                cls_token_num=cls_token_num, norm_eps=norm_eps, hidden=hidden, frame_prediction=frame_prediction, use_fixed_pe=use_fixed_pe,
                class_dropout_prob=class_dropout_prob, multiple_of=multiple_of, n_kv_head=n_head if n_kv_head is None else n_kv_head)


def init_state_dict(cfg, seed=77, head_std=0.02):
    """N(0, 0.02) Linear / Embedding weights (larp_ar.py:281-296) from the counter-based generator; the output head, which
    the reference zero-initialises, gets N(0, head_std) so that logits are not all equal."""
    D, I, V = cfg["dim"], cfg["hidden"], cfg["vocab_size"]
    p, s = {}, [seed * 1000]

    def nrm(shape, std=0.02):
        s[0] += 1
        return torch.from_numpy(gen.normal(shape, s[0], std))

    if not cfg["frame_prediction"]:
        p["cls_embedding.embedding_table.weight"] = nrm((cfg["num_classes"] + (cfg["class_dropout_prob"] > 0), D))
    p["tok_embeddings.weight"] = nrm((V + (1 if cfg["frame_prediction"] else 0), D))
    for i in range(cfg["n_layer"]):
        pre = f"layers.{i}."
        p[pre + "attention.wqkv.weight"] = nrm(((cfg["n_head"] + 2 * cfg.get("n_kv_head", cfg["n_head"])) * (D // cfg["n_head"]), D))   # larp_ar.py:171-175
        p[pre + "attention.wo.weight"] = nrm((D, D))
        p[pre + "feed_forward.w1.weight"] = nrm((I, D))
        p[pre + "feed_forward.w3.weight"] = nrm((I, D))
        p[pre + "feed_forward.w2.weight"] = nrm((D, I))
        p[pre + "attention_norm.weight"] = 1.0 + nrm((D,), 0.1)
        p[pre + "ffn_norm.weight"] = 1.0 + nrm((D,), 0.1)
    p["norm.weight"] = 1.0 + nrm((D,), 0.1)
    p["output.weight"] = nrm((V, D), head_std)
    n_pe = cfg["max_seq_len"] + cfg["cls_token_num"] - 1
    if cfg["use_fixed_pe"]:
        p["abs_pe"] = torch.from_numpy(sincos_1d(D, np.arange(n_pe))).float().reshape(1, n_pe, D)
    else:
        p["abs_pe"] = nrm((1, n_pe, D))
    return p


def rmsnorm(x, w, eps):
    """models/norm.py:12-17 (computed in fp32 with autocast disabled)"""
    return x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps) * w


def attention(x, wqkv, wo, n_head, emu=False, n_kv_head=None):
    """larp_ar.py:182-213 without a cache: causal softmax(q k^T / sqrt(hd)) v; with n_kv_head < n_head every K / V head serves
    n_head // n_kv_head consecutive query heads (`repeat_interleave`, :202-203)"""
    b, n, d = x.shape
    hd = d // n_head
    n_kv = n_head if n_kv_head is None else n_kv_head
    q, k, v = linear(x, wqkv, None, emu).split([d, n_kv * hd, n_kv * hd], dim=-1)
    q = q.reshape(b, n, n_head, hd).transpose(1, 2)
    k, v = (t.reshape(b, n, n_kv, hd).transpose(1, 2).repeat_interleave(n_head // n_kv, dim=1) for t in (k, v))
    s = (q @ k.transpose(-2, -1)) / math.sqrt(hd)
    s = s.masked_fill(~torch.tril(torch.ones(n, n, dtype=torch.bool)), float("-inf"))
    o = _rb(torch.softmax(s, dim=-1) @ v, emu).transpose(1, 2).reshape(b, n, d)
    return linear(o, wo, None, emu)


def feed_forward(x, w1, w3, w2, emu=False):
    """larp_ar.py:135-136: w2(silu(w1 x) * w3 x)"""
    g = _rb(F.silu(linear(x, w1, None, emu)), emu)
    return linear(_rb(g * linear(x, w3, None, emu), emu), w2, None, emu)


def block(x, p, pre, cfg, emu=False):
    """larp_ar.py:216-229 (drop_path / dropouts are identity: evaluation, or rates 0)"""
    h = x + attention(rmsnorm(x, p[pre + "attention_norm.weight"], cfg["norm_eps"]), p[pre + "attention.wqkv.weight"], p[pre + "attention.wo.weight"],
                      cfg["n_head"], emu, cfg.get("n_kv_head"))
    return h + feed_forward(rmsnorm(h, p[pre + "ffn_norm.weight"], cfg["norm_eps"]), p[pre + "feed_forward.w1.weight"], p[pre + "feed_forward.w3.weight"],
                            p[pre + "feed_forward.w2.weight"], emu)


def cond_embed(p, cfg, cond_idx):
    """larp_ar.py:355-361 / embed.py:251-259 in evaluation (no label dropout); negative labels -> the unconditional row"""
    if cfg["frame_prediction"]:
        return p["tok_embeddings.weight"][cond_idx]
    lab = torch.where(cond_idx < 0, torch.full_like(cond_idx, cfg["num_classes"]), cond_idx)
    return p["cls_embedding.embedding_table.weight"][lab].unsqueeze(1)[:, : cfg["cls_token_num"]]


def trunk(p, cfg, h, emu=False):
    for i in range(cfg["n_layer"]):
        h = block(h, p, f"layers.{i}.", cfg, emu)
    return linear(rmsnorm(h, p["norm.weight"], cfg["norm_eps"]), p["output.weight"], None, emu)


def forward(p, cfg, idx, cond_idx, targets=None, valid=None, training=True, emu=False):
    """larp_ar.py:346-409, the `idx is not None and cond_idx is not None` branch; returns (logits, loss).
    `training` only selects the logits slice of line 397 (dropout rates are taken as 0)."""
    h = torch.cat((cond_embed(p, cfg, cond_idx), p["tok_embeddings.weight"][idx]), dim=1)
    h = h + p["abs_pe"][:, : h.shape[1]]
    logits = trunk(p, cfg, h, emu)
    if training or cfg["frame_prediction"]:
        logits = logits[:, cfg["cls_token_num"] - 1:].contiguous()
    loss = None
    if valid is not None:
        la = F.cross_entropy(logits.reshape(-1, logits.size(-1)), targets.reshape(-1), reduction="none")
        va = valid[:, None].repeat(1, targets.shape[1]).reshape(-1)
        loss = (la * va).sum() / max(va.sum(), 1)
    elif targets is not None:
        loss = F.cross_entropy(logits.reshape(-1, logits.size(-1)), targets.reshape(-1))
    return logits, loss


@torch.no_grad()
def generate_greedy(p, cfg, cond, max_new_tokens, cfg_scale=1.0, emu=False, return_margins=False):
    """ar/generate.py:126-174 with sample_logits=False (argmax), temperature 1, no top-k/top-p.  The reference walks a KV
    cache; attention is causal, so recomputing the whole prefix each step gives the same logits for the newest position.
    Returns tokens [B, max_new_tokens] (and the top-1/top-2 probability gaps, for near-tie screening)."""
    if cfg["frame_prediction"]:
        assert cfg_scale == 1.0
        cc = cond
    else:
        cc = torch.cat([cond, torch.ones_like(cond) * cfg["num_classes"]]) if cfg_scale > 1.0 else cond
    h0 = cond_embed(p, cfg, cc)
    toks, margins = [], []
    for step in range(max_new_tokens):
        h = h0 if not toks else torch.cat((h0, p["tok_embeddings.weight"][torch.cat([torch.stack(toks, 1)] * (2 if cfg_scale > 1.0 else 1))]), dim=1)
        h = h + p["abs_pe"][:, : h.shape[1]]
        logits = trunk(p, cfg, h, emu)[:, -1]
        if cfg_scale > 1.0:
            c, u = torch.split(logits, len(logits) // 2, dim=0)
            logits = u + (c - u) * cfg_scale
        probs = F.softmax(logits.float(), dim=-1)
        top = torch.topk(probs, k=2, dim=-1)
        toks.append(top[1][:, 0])
        margins.append(top[0][:, 0] - top[0][:, 1])
    out = torch.stack(toks, 1)
    return (out, torch.stack(margins, 1)) if return_margins else out


def top_k_top_p_filtering(logits, top_k=0, top_p=1.0, filter_value=-float("Inf"), min_tokens_to_keep=1):
    """ar/generate.py:13-52, restated with explicit loops over rows (small inputs only)"""
    logits = logits.clone()
    for r in range(logits.shape[0]):
        row = logits[r]
        if top_k > 0:
            k = min(max(top_k, min_tokens_to_keep), row.numel())
            kth = torch.sort(row, descending=True)[0][k - 1]
            row[row < kth] = filter_value
        if top_p < 1.0:
            sv, si = torch.sort(row, descending=True)
            cum = torch.cumsum(F.softmax(sv, dim=-1), dim=-1)
            drop = [bool(c > top_p) for c in cum]
            if min_tokens_to_keep > 1:
                for j in range(min_tokens_to_keep):
                    drop[j] = False
            drop = [False] + drop[:-1]
            for j, d_ in enumerate(drop):
                if d_:
                    row[si[j]] = filter_value
    return logits
