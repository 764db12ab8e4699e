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


class BlockStack(torch.autograd.Function):
    """depth x timm Block(dim, heads, mlp_ratio=4, qkv_bias=False) on x fp32 [B, L, D]; head_dim 64 or 32, any L.
    Forward = 8 launches per block, backward = 13 + one grouped weight-gradient launch, the same sequence as
    vt_engine.hip::block_forward/block_backward (reference: timm Block as built at models/transformer.py:18-25, 52-59)."""

    @staticmethod
    def forward(ctx, x, n_head, *params):
        hip.require_gpu(x, *params)
        assert x.dim() == 3 and x.dtype == torch.float32
        B, L, D = x.shape
        depth = len(params) // PARAMS_PER_BLOCK
        assert depth * PARAMS_PER_BLOCK == len(params) and D % n_head == 0
        hd = D // n_head
        M, Mp, dev = B * L, _pad64(B * L), x.device
        cur = x.contiguous().reshape(M, D)
        saved = []
        for i in range(depth):
            g1, b1, wqkv, wproj, bproj, g2, b2, wfc1, bfc1, wfc2, bfc2 = params[i * PARAMS_PER_BLOCK:(i + 1) * PARAMS_PER_BLOCK]
            qkv_b, qkv_t = hip.pack_weight(wqkv)
            proj_b, proj_t = hip.pack_weight(wproj)
            fc1_b, fc1_t = hip.pack_weight(wfc1)
            fc2_b, fc2_t = hip.pack_weight(wfc2)
            # rows >= M of every buffer a weight-gradient GEMM contracts over stay zero
            h1 = _zeros(Mp, D, dev)
            _, mean1, rstd1 = hip.layernorm_fwd(cur, g1, b1, 1e-5, y=h1)
            qkv = hip.gemm_nt(h1[:M], qkv_b, hip.EPI_BF16)
            o = _zeros(Mp, D, dev)
            _, lse = hip.attention_fwd(qkv, B, L, n_head, hd, o=o)
            x_mid = hip.gemm_nt(o[:M], proj_b, hip.EPI_F32, bias=bproj, residual=cur)
            h2 = _zeros(Mp, D, dev)
            _, mean2, rstd2 = hip.layernorm_fwd(x_mid, g2, b2, 1e-5, y=h2)
            u = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
            g = _zeros(Mp, 4 * D, dev)
            hip.gemm_nt(h2[:M], fc1_b, hip.EPI_BF16_GELU, bias=bfc1, out=u, out2=g[:M])
            x_out = hip.gemm_nt(g[:M], fc2_b, hip.EPI_F32, bias=bfc2, residual=x_mid)
            saved.append(dict(x_in=cur, h1=h1, mean1=mean1, rstd1=rstd1, qkv=qkv, lse=lse, o=o, x_mid=x_mid, h2=h2, mean2=mean2,
                              rstd2=rstd2, u=u, g=g, qkv_t=qkv_t, proj_t=proj_t, fc1_t=fc1_t, fc2_t=fc2_t))
            cur = x_out
        ctx.saved, ctx.geom, ctx.params = saved, (B, L, D, n_head, hd), params
        return cur.reshape(B, L, D)

    @staticmethod
    def backward(ctx, dy):
        B, L, D, H, hd = ctx.geom
        M, Mp, dev = B * L, _pad64(B * L), dy.device
        depth = len(ctx.saved)
        need = ctx.needs_input_grad[2:]
        grads = [None] * len(ctx.params)
        dX = dy.contiguous().reshape(M, D).clone()
        dXa = _zeros(Mp, D, dev)
        hip.cast_rows(dX, dst=dXa)
        if need[(depth - 1) * PARAMS_PER_BLOCK + 10]:
            grads[(depth - 1) * PARAMS_PER_BLOCK + 10] = hip.colsum(dX)
        for i in range(depth - 1, -1, -1):
            s = ctx.saved[i]
            g1, _, wqkv, wproj, _, g2, _, wfc1, _, wfc2, _ = ctx.params[i * PARAMS_PER_BLOCK:(i + 1) * PARAMS_PER_BLOCK]
            k = i * PARAMS_PER_BLOCK
            du = _zeros(Mp, 4 * D, dev)
            hip.gemm_nt(dXa[:M], s["fc2_t"], hip.EPI_BF16_DGELU, aux=s["u"], out=du[:M])
            if need[k + 8]:
                grads[k + 8] = hip.colsum(du, rows=M)
            dh = hip.gemm_nt(du[:M], s["fc1_t"], hip.EPI_BF16)
            dXm = _zeros(Mp, D, dev)
            _, _, dg2, db2, dpb = hip.layernorm_bwd(dh, s["x_mid"], g2, s["mean2"], s["rstd2"], dres=dX, dx=dX, dxb=dXm)
            grads[k + 5], grads[k + 6], grads[k + 4] = dg2, db2, dpb
            dob = hip.gemm_nt(dXm[:M], s["proj_t"], hip.EPI_BF16)
            dqkv = _zeros(Mp, 3 * D, dev)
            hip.attention_bwd(s["qkv"], s["o"], dob, s["lse"], B, L, H, hd, dqkv=dqkv)
            dh = hip.gemm_nt(dqkv[:M], s["qkv_t"], hip.EPI_BF16)
            wg = []
            for idx, A, Bm, w in ((k + 9, dXa, s["g"], wfc2), (k + 7, du, s["h2"], wfc1), (k + 3, dXm, s["o"], wproj), (k + 2, dqkv, s["h1"], wqkv)):
                if need[idx]:
                    grads[idx] = torch.empty_like(w, dtype=torch.float32)
                    wg.append(dict(A=A, B=Bm, out=grads[idx]))
            if wg:
                hip.gemm_tn_grouped(wg)
            dXa = _zeros(Mp, D, dev)
            _, _, dg1, db1, dxs = hip.layernorm_bwd(dh, s["x_in"], g1, s["mean1"], s["rstd1"], dres=dX, dx=dX, dxb=dXa)
            grads[k + 0], grads[k + 1] = dg1, db1
            if i > 0:
                grads[k - PARAMS_PER_BLOCK + 10] = dxs  # fc2 bias of the block below = column sum of dL/dx_in
        grads = [g if n else None for g, n in zip(grads, need)]
        return (dX.reshape(B, L, D), None, *grads)


def block_stack(x, blocks, n_head):
    return BlockStack.apply(x, n_head, *block_params(blocks))


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
        dz, _, dW = hip.vq_backward(g2, gs, o, beta=beta, codebook_w=cw, l2_normalized=l2n)
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
