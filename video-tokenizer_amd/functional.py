"""Autograd functions over the C-ABI ops: the standalone forwards of the reference's sub-modules.

The tokenizer step itself runs in the fused C++ engine (engine.py).  The reference resolves its
sub-modules by registry name as well (`transformer_encoder_parallel`, `transformer_encoder_fused`,
`bottleneck`, `vq`) and calls them on their own -- and the GAN branch of the step is a stack
of the same timm Blocks (models/loss.py:119-204).  These functions give those call sites the same
kernels: every matrix product, LayerNorm, attention and codebook search below is a libvt_hip call; torch
only owns the tensors, the autograd graph and a few O(B x D) glue ops.

Mixed-precision contract = the engine's: fp32 residual stream and LayerNorm statistics,
bf16 MFMA operands with fp32 accumulation, Linear outputs rounded to bf16 where autocast would.
No CPU path: CPU tensors are refused by hip.ptr().
"""
import ctypes

import torch

from . import hip


def _pad64(m):
    return (m + 63) // 64 * 64


def _zeros(rows, cols, dev, dtype=torch.bfloat16):
    return torch.zeros(rows, cols, device=dev, dtype=dtype)


PARAMS_PER_BLOCK = 11
BLOCK_PARAM_NAMES = ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.proj.weight", "attn.proj.bias",
                     "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")


def block_params(blocks):
    """flat parameter list of an nn.ModuleList / nn.Sequential of transformer.Block in BLOCK_PARAM_NAMES order"""
    out = []
    for b in blocks:
        out += [b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.proj.weight, b.attn.proj.bias,
                b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight, b.mlp.fc1.bias, b.mlp.fc2.weight, b.mlp.fc2.bias]
    return out


_STACKS = {}       # (B, L, D, H, depth) -> (handle, workspace bytes)
_WS_POOL = {}      # same key -> zero-initialised workspaces whose backward is done (their padded rows are still zero)


def _stack(key):
    h = _STACKS.get(key)
    if h is None:
        cfg = hip.StackConfig(*key)
        out = ctypes.c_void_p()
        hip.check(hip.lib().vt_stack_create(ctypes.byref(cfg), ctypes.byref(out)), "vt_stack_create")
        h = (out, int(hip.lib().vt_stack_workspace_bytes(out)))
        _STACKS[key] = h
    return h


def _take_ws(key, nbytes, dev):
    pool = _WS_POOL.setdefault((key, str(dev)), [])
    if pool:
        return pool.pop()
    return torch.zeros(nbytes, dtype=torch.uint8, device=dev)   # = vt_stack_init_workspace


def _block_array(tensors, depth):
    arr = (hip.BlockTensors * depth)()
    for i in range(depth):
        for j, name in enumerate(hip.BLOCK_FIELDS):
            setattr(arr[i], name, tensors[i * PARAMS_PER_BLOCK + j].data_ptr())
    return arr


class BlockStack(torch.autograd.Function):
    """depth x timm Block(dim, heads, mlp_ratio=4, qkv_bias=False) on x fp32 [B, L, D]; head_dim 64 or 32, any L.
    One vt_stack_forward / vt_stack_backward call each: the C++ engine enqueues the tokenizer's own block sequence
    (8 launches per block forward, 13 backward + one grouped weight-gradient launch per 4 blocks; reference: timm Block
    as built at models/transformer.py:18-25, 52-59).  Frozen parameters (requires_grad False everywhere: the
    generator-side pass through the discriminator) skip the weight-gradient GEMMs."""

    @staticmethod
    def forward(ctx, x, n_head, *params):
        hip.require_gpu(x, *params)
        assert x.dim() == 3 and x.dtype == torch.float32
        B, L, D = x.shape
        depth = len(params) // PARAMS_PER_BLOCK
        assert depth * PARAMS_PER_BLOCK == len(params) and D % n_head == 0
        key = (B, L, D, n_head, depth)
        handle, nbytes = _stack(key)
        params = tuple(p_.detach().float().contiguous() for p_ in params)
        ws = _take_ws(key, nbytes, x.device)
        xin = x.contiguous()
        out = torch.empty_like(xin)
        hip.check(hip.lib().vt_stack_forward(handle, _block_array(params, depth), hip.ptr(xin), hip.ptr(ws), hip.ptr(out), hip.stream()), "vt_stack_forward")
        if any(ctx.needs_input_grad):     # a backward may follow: the activations stay in ws until then
            ctx.key, ctx.ws, ctx.params, ctx.done = key, ws, params, False
        else:                             # inference: the workspace is free again behind this forward (stream order)
            _WS_POOL[(key, str(x.device))].append(ws)
        return out

    @staticmethod
    def backward(ctx, dy):
        if ctx.done:
            raise RuntimeError("BlockStack: second backward through the same forward (its workspace was recycled)")
        B, L, D, H, depth = ctx.key
        handle, _ = _stack(ctx.key)
        need = ctx.needs_input_grad[2:]
        grads = [torch.empty_like(p_) for p_ in ctx.params]     # LayerNorm / bias gradients are always produced
        dx = torch.empty(B, L, D, device=dy.device, dtype=torch.float32)
        dyc = dy.contiguous().float()
        hip.check(hip.lib().vt_stack_backward(handle, _block_array(ctx.params, depth), hip.ptr(dyc), hip.ptr(ctx.ws), _block_array(grads, depth),
                                              hip.ptr(dx), int(any(need)), hip.stream()), "vt_stack_backward")
        ctx.done = True
        _WS_POOL[(ctx.key, str(dy.device))].append(ctx.ws)
        ctx.ws = None
        return (dx, None, *[g if n else None for g, n in zip(grads, need)])


def block_stack(x, blocks, n_head):
    return BlockStack.apply(x, n_head, *block_params(blocks))


# ------------------------------------------------------------------------------------------------------------------------
# stack of the TiTok-style gated layers (models/model_new/base/transformer.py:66-91) as one engine call per direction
# ------------------------------------------------------------------------------------------------------------------------
PARAMS_PER_GATED_LAYER = 10   # hip.GATED_FIELDS order: to_qkv, q_norm w/b, k_norm w/b, out_proj, ffd LayerNorm w/b, ffd.1, ffd.3
_GSTACKS = {}


def _gstack(key):
    h = _GSTACKS.get(key)
    if h is None:
        cfg = hip.GatedStackConfig(*key)
        out = ctypes.c_void_p()
        hip.check(hip.lib().vt_gated_stack_create(ctypes.byref(cfg), ctypes.byref(out)), "vt_gated_stack_create")
        h = (out, int(hip.lib().vt_gated_stack_workspace_bytes(out)))
        _GSTACKS[key] = h
    return h


def _gated_array(tensors, depth):
    arr = (hip.GatedLayerTensors * depth)()
    for i in range(depth):
        for j, name in enumerate(hip.GATED_FIELDS):
            setattr(arr[i], name, tensors[i * PARAMS_PER_GATED_LAYER + j].data_ptr())
    return arr


class GatedStack(torch.autograd.Function):
    """depth x {x += Attn(x); x += ffd(x); x /= sqrt(i+1)} on x fp32 [B, L, D] (B * L % 64 == 0, D = 64 * heads): ONE
    vt_gated_stack_forward / _backward call each (11 launches per layer forward, 17 backward, issued by the C++ engine).  The
    workspace carries the saved activations and the bf16 operand copies of the weights; `pack_key` identifies the weight
    versions a workspace was packed for."""

    @staticmethod
    def forward(ctx, x, cos, sin, n_head, pack_key, *params):
        hip.require_gpu(x, cos, sin, *params)
        assert x.dim() == 3
        B, L, D = x.shape
        depth = len(params) // PARAMS_PER_GATED_LAYER
        assert depth * PARAMS_PER_GATED_LAYER == len(params)
        inner = params[9].shape[1]
        key = (B, L, D, n_head, depth, inner)
        handle, nbytes = _gstack(key)
        params = tuple(p_.detach().float().contiguous() for p_ in params)
        ws = _take_ws(("gated",) + key, nbytes, x.device)
        repack = getattr(ws, "_vt_pack_key", None) != pack_key
        ws._vt_pack_key = pack_key
        xin = x.contiguous().float()
        out = torch.empty_like(xin)
        hip.check(hip.lib().vt_gated_stack_forward(handle, _gated_array(params, depth), hip.ptr(cos), hip.ptr(sin), hip.ptr(xin), hip.ptr(ws), hip.ptr(out),
                                                   int(repack), hip.stream()), "vt_gated_stack_forward")
        if any(ctx.needs_input_grad):
            ctx.key, ctx.ws, ctx.params, ctx.tabs, ctx.done = key, ws, params, (cos, sin), False
        else:
            _WS_POOL[(("gated",) + key, str(x.device))].append(ws)
        return out

    @staticmethod
    def backward(ctx, dy):
        if ctx.done:
            raise RuntimeError("GatedStack: second backward through the same forward (its workspace was recycled)")
        B, L, D, H, depth, inner = ctx.key
        handle, _ = _gstack(ctx.key)
        grads = [torch.empty_like(p_) for p_ in ctx.params]
        dx = torch.empty(B, L, D, device=dy.device, dtype=torch.float32)
        dyc = dy.contiguous().float()
        cos, sin = ctx.tabs
        hip.check(hip.lib().vt_gated_stack_backward(handle, _gated_array(ctx.params, depth), hip.ptr(cos), hip.ptr(sin), hip.ptr(dyc), hip.ptr(ctx.ws),
                                                    _gated_array(grads, depth), hip.ptr(dx), hip.stream()), "vt_gated_stack_backward")
        ctx.done = True
        _WS_POOL[(("gated",) + ctx.key, str(dy.device))].append(ctx.ws)
        ctx.ws = None
        need = ctx.needs_input_grad[5:]
        return (dx, None, None, None, None, *[g if n else None for g, n in zip(grads, need)])


class PatchEmbed(torch.autograd.Function):
    """PatchEmbed3D.forward (models/embed.py:85-116): Conv3d(kernel = stride = (pt,p,p)) as patch gather + one GEMM,
    bf16-rounded like the conv under autocast, then (optionally) + pos_embed [N, D] in fp32, fused in the epilogue."""

    @staticmethod
    def forward(ctx, video, weight, bias, pos_embed):
        hip.require_gpu(video, weight, bias, pos_embed)
        D, C, pt, p, _ = weight.shape
        B = video.shape[0]
        video = video.contiguous().float()
        patches = hip.patchify(video, pt, p)                       # [B*N, Kp] bf16, (c,dt,dy,dx) inside a patch
        M, Kp = patches.shape
        wb, wt = hip.pack_weight(weight)
        kw = dict(rowmod=pos_embed.reshape(-1, D).contiguous(), rowmod_period=M // B) if pos_embed is not None else {}
        tok = hip.gemm_nt(patches, wb, hip.EPI_F32, bias=bias, round_bf16=True, **kw)
        ctx.save_for_backward(patches, wt)
        ctx.geom = (B, C, video.shape[2], video.shape[3], pt, p, D)
        return tok.reshape(B, M // B, D)

    @staticmethod
    def backward(ctx, dtok):
        patches, wt = ctx.saved_tensors
        B, C, T, S, pt, p, D = ctx.geom
        M, Kp = patches.shape
        Mp = _pad64(M)
        dev = dtok.device
        dT = _zeros(Mp, D, dev)
        hip.cast_rows(dtok.contiguous().reshape(M, D), dst=dT)
        dvideo = dw = db = None
        if ctx.needs_input_grad[0]:
            rows = hip.gemm_nt(dT[:M], wt, hip.EPI_F32)            # [M, Kp] fp32 in patch order
            dvideo = hip.unpatchify(rows, B, C, T, S, pt, p)
        if ctx.needs_input_grad[1]:
            pp = patches
            if Mp != M:
                pp = _zeros(Mp, Kp, dev)
                pp[:M].copy_(patches)
            dw = torch.empty(D, Kp, device=dev)
            hip.gemm_tn_grouped([dict(A=dT, B=pp, out=dw)])
            dw = dw.reshape(D, C, pt, p, p)
        if ctx.needs_input_grad[2]:
            db = hip.colsum(dT, rows=M)
        return dvideo, dw, db, None


def _pad_cols(t2d, k_to):
    """bf16 copy of a 2-D tensor with the column count zero-padded to k_to (GEMM contraction dims are multiples of 64)"""
    r, k = t2d.shape
    out = _zeros(r, k_to, t2d.device)
    out[:, :k].copy_(t2d)
    return out


class Linear(torch.autograd.Function):
    """nn.Linear under autocast(bf16) for the small projections outside the block stacks (bottleneck in/out_linear,
    models/bottleneck.py:140-164): bf16 operands, fp32 accumulate, output rounded to bf16 (returned in an fp32 tensor)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        hip.require_gpu(x, weight, bias)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).float()
        M, K = x2.shape
        N = weight.shape[0]
        Kp_, Np_, Mp = _pad64(K), _pad64(N), _pad64(M)
        xb = _zeros(Mp, Kp_, x.device)
        xb[:M, :K].copy_(x2)
        wb = _pad_cols(weight.detach(), Kp_)                        # [N, Kp]
        y = hip.gemm_nt(xb[:M], wb, hip.EPI_F32, bias=bias, round_bf16=True, out=torch.empty(M, (N + 3) // 4 * 4, device=x.device))
        ctx.save_for_backward(xb, weight)
        ctx.geom = (shp, M, K, N, Kp_, Np_, Mp)
        return y[:, :N].reshape(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xb, weight = ctx.saved_tensors
        shp, M, K, N, Kp_, Np_, Mp = ctx.geom
        dev = dy.device
        dyb = _zeros(Mp, Np_, dev)                                  # grad of a bf16 output is bf16 under autocast
        dyb[:M, :N].copy_(dy.reshape(M, N))
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = _pad_cols(weight.detach().t().contiguous(), Np_)   # [K, Np]: B operand of dX = dY . W
            dx = hip.gemm_nt(dyb[:M], wt, hip.EPI_F32, out=torch.empty(M, (K + 3) // 4 * 4, device=dev))[:, :K].reshape(shp)
        if ctx.needs_input_grad[1]:
            full = torch.empty(Np_, Kp_, device=dev)
            hip.gemm_tn_grouped([dict(A=dyb, B=xb, out=full)])
            dw = full[:N, :K].contiguous()
        if ctx.needs_input_grad[2]:
            db = hip.colsum(dyb, rows=M)[:N].contiguous()
        return dx, dw, db


class VectorQuantize(torch.autograd.Function):
    """SimpleVectorQuantizer.forward (models/bottleneck.py:262-324).  Returns (regularized_z, indices, loss_q, loss_commit,
    loss_codebook, unregularized_z, emb); gradients flow through regularized_z (straight-through) and the three losses,
    exactly what autograd derives from :292-307; unregularized_z / emb are returned detached."""

    @staticmethod
    def forward(ctx, z, codebook, mode, l2_normalized, inv_tau, beta, codebook_w, seed):
        hip.require_gpu(z, codebook)
        shp = z.shape
        z2 = z.reshape(-1, shp[-1]).float().contiguous()
        o = hip.vq_forward(z2, codebook, mode, l2_normalized=l2_normalized, inv_tau=inv_tau, beta=beta, codebook_w=codebook_w, seed=seed)
        ctx.saved, ctx.cfg, ctx.shp = o, (beta, codebook_w, l2_normalized), shp
        L = o["losses"]
        idx, zn, emb = o["idx"].clone(), o["zn"].reshape(shp).clone(), o["E"].clone()
        ctx.mark_non_differentiable(idx, zn, emb)
        return o["rz"].reshape(shp), idx, L[0].clone(), L[1].clone(), L[2].clone(), zn, emb

    @staticmethod
    def backward(ctx, g_rz, _gi, g_q, g_c, g_cb, _gz, _ge):
        beta, cw, l2n = ctx.cfg
        o = ctx.saved
        dev = o["zn"].device
        gs = torch.stack([t if t is not None else torch.zeros((), device=dev) for t in (g_q, g_c, g_cb)]).float().contiguous()
        g2 = g_rz.reshape(o["zn"].shape).float().contiguous() if g_rz is not None else None
        dz, _, dW = hip.vq_backward(g2, gs, o, beta=beta, codebook_w=cw, l2_normalized=l2n, need_dW=ctx.needs_input_grad[1])
        return dz.reshape(ctx.shp), dW, None, None, None, None, None, None


def codebook_entries(indices, codebook, l2_normalized):
    """get_codebook_entry (bottleneck.py:327-344): rows of the (normalised) codebook; fp32 [*indices.shape, d]"""
    hip.require_gpu(indices, codebook)
    K, d = codebook.shape
    idx = indices.reshape(-1).to(torch.int64).contiguous()
    dev = codebook.device
    E, wn = torch.empty(K, d, device=dev), torch.empty(K, device=dev)
    ws = hip._ws(hip.lib().vt_vq_workspace_bytes(max(idx.numel(), 1), K, d), dev)
    hip.check(hip.lib().vt_vq_prep_codebook(hip.ptr(codebook), K, d, int(l2_normalized), hip.ptr(E), hip.ptr(wn), hip.ptr(ws), hip.stream()), "vt_vq_prep_codebook")
    out = torch.empty(idx.numel(), d, device=dev)
    hip.check(hip.lib().vt_vq_gather(hip.ptr(E), hip.ptr(idx), idx.numel(), K, d, hip.ptr(out), None, 0, hip.stream()), "vt_vq_gather")
    return out.reshape(*indices.shape, d)


class FiniteScalarQuantize(torch.autograd.Function):
    """FSQ.forward (models/model_new/quantizer/fsq.py:119-131): one launch per direction, straight-through gradient."""

    @staticmethod
    def forward(ctx, z, levels):
        zc = z.contiguous()
        codes, idx = hip.fsq_forward(zc, levels)
        ctx.save_for_backward(zc)
        ctx.levels = levels
        ctx.mark_non_differentiable(idx)
        return codes, idx

    @staticmethod
    def backward(ctx, dcodes, _didx):
        (z,) = ctx.saved_tensors
        return hip.fsq_backward(z, dcodes.contiguous().to(z.dtype), ctx.levels), None


class LayerNormRows(torch.autograd.Function):
    """nn.LayerNorm over the last dim on the LayerNorm kernels (fp32 in, statistics in fp32, bf16-rounded output returned in an
    fp32 tensor -- what the Linear that follows reads under autocast).  Widths of vt_layernorm_fwd (128 ... 1024)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        hip.require_gpu(x, weight, bias)
        shp = x.shape
        x2 = x.contiguous().reshape(-1, shp[-1]).float()
        y, mean, rstd = hip.layernorm_fwd(x2, weight, bias, eps)
        ctx.save_for_backward(x2, weight, mean, rstd)
        return y.float().reshape(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, mean, rstd = ctx.saved_tensors
        dyb = hip.cast_rows(dy.contiguous().reshape(x2.shape).float())
        dx, _, dg, db, _ = hip.layernorm_bwd(dyb, x2, weight, mean, rstd, want_dxsum=False)
        return dx.reshape(dy.shape), dg, db, None


class Unpatchify(torch.autograd.Function):
    """rows [B * N, C * pt * p * p] in (c, dt, dy, dx) column order -> video [B, C, T, S, S] (the scatter of larp_tokenizer.py:441-454
    once the head's rows are permuted to that order); backward = the patch gather."""

    @staticmethod
    def forward(ctx, rows, geom):
        B, C, T, S, pt, p = geom
        ctx.geom = geom
        return hip.unpatchify(rows.contiguous().float(), B, C, T, S, pt, p)

    @staticmethod
    def backward(ctx, dvideo):
        B, C, T, S, pt, p = ctx.geom
        return hip.patchify(dvideo.contiguous().float(), pt, p).float(), None
