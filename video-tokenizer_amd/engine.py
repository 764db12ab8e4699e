"""Host side of the fused tokenizer engine: binds a LARPTokenizer's parameters to the C ABI
(vt_tokenizer_* in include/vt_hip.h) and exposes forward/backward as ONE torch.autograd.Function.

PyTorch is plumbing here: it owns the device memory (parameters, one flat gradient buffer, the
engine workspace, the outputs) and the stream; every FLOP of the step runs in libvt_hip.so.
Gradients are written by the kernels straight into slices of one flat fp32 buffer laid out in the
order backward produces them, so a data-parallel reducer can all-reduce finished slices while later
stages are still running (parallel.py).
"""
import ctypes
import os

import torch

from . import hip


def _flat_order(model):
    """(name, tensor, stage) for every trainable parameter in the order the backward stages write
    their gradients.  Stage numbering = vt_tokenizer_backward: 0 head, 1..depth_dec decoder blocks
    (last first), then bottleneck, encoder blocks (last first), patch-embed."""
    de, dd = model.encoder.depth, model.decoder.depth
    out = []

    def blk(side, i, stage, with_fc2_b_of=None):
        b = getattr(model, side).blocks[i]
        pre = f"{side}.blocks.{i}."
        for n, p in (("norm1.weight", b.norm1.weight), ("norm1.bias", b.norm1.bias), ("attn.qkv.weight", b.attn.qkv.weight),
                     ("attn.proj.weight", b.attn.proj.weight), ("attn.proj.bias", b.attn.proj.bias), ("norm2.weight", b.norm2.weight),
                     ("norm2.bias", b.norm2.bias), ("mlp.fc1.weight", b.mlp.fc1.weight), ("mlp.fc1.bias", b.mlp.fc1.bias),
                     ("mlp.fc2.weight", b.mlp.fc2.weight)):
            out.append((pre + n, p, stage))

    def fc2b(side, i, stage):
        out.append((f"{side}.blocks.{i}.mlp.fc2.bias", getattr(model, side).blocks[i].mlp.fc2.bias, stage))

    fl = model.final_layer
    out += [("final_layer.linear.weight", fl.linear.weight, 0), ("final_layer.linear.bias", fl.linear.bias, 0),
            ("final_layer.norm_final.weight", fl.norm_final.weight, 0), ("final_layer.norm_final.bias", fl.norm_final.bias, 0)]
    fc2b("decoder", dd - 1, 0)
    for st in range(1, dd + 1):
        i = dd - st
        blk("decoder", i, st)
        if i > 0:
            fc2b("decoder", i - 1, st)
    st = dd + 1
    if model.use_decoder_patch_query_token_type_embed:
        out.append(("decoder_patch_query_token_type_embed", model.decoder_patch_query_token_type_embed, st))
    named = dict(model.named_parameters())
    bn = model._bt_names   # 'vq': bottleneck.{in,out}_linear.* + bottleneck.regularizer.embedding.weight; 'sq': sq_{in,out}_linear.* + bottleneck.embedding.weight
    out += [(bn[k], named[bn[k]], st) for k in ("out_b", "out_w", "codebook", "in_b", "in_w")]
    fc2b("encoder", de - 1, st)
    for k in range(1, de + 1):
        i = de - k
        blk("encoder", i, st + k)
        if i > 0:
            fc2b("encoder", i - 1, st + k)
    last = st + de + 1
    out += [("encoder_latent_query_embed", model.encoder_latent_query_embed, last),
            ("x_embedder.proj.bias", model.x_embedder.proj.bias, last), ("x_embedder.proj.weight", model.x_embedder.proj.weight, last)]
    return out


class _Tensors:
    """ctypes vtTokenizerTensors over a name->tensor mapping (parameters or gradient views)."""

    def __init__(self, model, get):
        de, dd = model.encoder.depth, model.decoder.depth
        self.enc = (hip.BlockTensors * de)()
        self.dec = (hip.BlockTensors * dd)()
        names = {"norm1_w": "norm1.weight", "norm1_b": "norm1.bias", "qkv_w": "attn.qkv.weight", "proj_w": "attn.proj.weight",
                 "proj_b": "attn.proj.bias", "norm2_w": "norm2.weight", "norm2_b": "norm2.bias", "fc1_w": "mlp.fc1.weight",
                 "fc1_b": "mlp.fc1.bias", "fc2_w": "mlp.fc2.weight", "fc2_b": "mlp.fc2.bias"}
        for side, arr, depth in (("encoder", self.enc, de), ("decoder", self.dec, dd)):
            for i in range(depth):
                for f, n in names.items():
                    setattr(arr[i], f, get(f"{side}.blocks.{i}.{n}"))
        t = hip.TokenizerTensors()
        t.pe_w, t.pe_b = get("x_embedder.proj.weight"), get("x_embedder.proj.bias")
        t.enc_patch_pe = get("encoder_patch_pe")
        t.enc_query = get("encoder_latent_query_embed")
        t.dec_latent_pe = get("decoder_latent_pe")
        t.dec_patch_query = get("decoder_patch_query_embed")
        t.dec_token_type = get("decoder_patch_query_token_type_embed")
        bn = model._bt_names
        t.in_w, t.in_b = get(bn["in_w"]), get(bn["in_b"])
        t.out_w, t.out_b = get(bn["out_w"]), get(bn["out_b"])
        t.codebook = get(bn["codebook"])
        t.head_norm_w, t.head_norm_b = get("final_layer.norm_final.weight"), get("final_layer.norm_final.bias")
        t.head_w, t.head_b = get("final_layer.linear.weight"), get("final_layer.linear.bias")
        t.enc_blocks = ctypes.cast(self.enc, ctypes.POINTER(hip.BlockTensors))
        t.dec_blocks = ctypes.cast(self.dec, ctypes.POINTER(hip.BlockTensors))
        self.struct = t


class _State:
    """One engine handle + workspace for a fixed (batch, frames, size, vq mode) geometry."""

    def __init__(self, engine, key, cfg, device):
        self.key = key
        self.cfg = cfg
        h = ctypes.c_void_p()
        hip.check(hip.lib().vt_tokenizer_create(ctypes.byref(cfg), ctypes.byref(h)), "vt_tokenizer_create")
        self.handle = h
        nbytes = hip.lib().vt_tokenizer_workspace_bytes(h)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        hip.check(hip.lib().vt_tokenizer_init_workspace(h, hip.ptr(self.ws), hip.stream()), "vt_tokenizer_init_workspace")
        self.nstages = hip.lib().vt_tokenizer_num_backward_stages(h)
        self.packed_version = None
        self.fwd_id = 0
        self.graphed = False      # this geometry's step is captured in a GraphedStep: device-side seed counter, weights re-packed inside the graph

    def __del__(self):
        try:
            if self.handle:
                hip.lib().vt_tokenizer_destroy(self.handle)
        except Exception:
            pass


class TokenizerEngine:
    def __init__(self, model):
        self.model = model
        self.states = {}
        self.flat_grad = None
        self.order = None
        self.segments = None  # stage -> (lo, hi) element range of the flat gradient buffer
        self.reducer = None   # set by parallel.DataParallelTokenizer
        self.seed_counter = 0
        self.split_k = None   # None = the library's default (on unless VT_GEMM_SPLITK=0); set_split_k() overrides it for every geometry
        self.wgrad_tail = 0   # set_wgrad_tail(): flush the encoder's first blocks' weight gradients block by block (data-parallel runs)
        self.wgrad_batch = 0    # set_wgrad_batch(): blocks per grouped weight-gradient launch (0 = the library's 4)
        self.wgrad_stream = None   # set_wgrad_stream(): torch.cuda.Stream the deferred weight-gradient launches run on (data-parallel runs)
        self.data_parallel = False  # set_data_parallel(): a collective shares the chip with the backward (one tile per workgroup for its multi-round GEMMs)
        self.graph_mode = False   # GraphedStep: weights are re-packed inside the captured step, the VQ seed counter lives on the device

    def __deepcopy__(self, memo):
        """copy.deepcopy(model) (the reference trainer builds its EMA model that way, base_trainer.py:396-405) must not
        copy native handles or workspaces: the copy gets a fresh, empty engine bound to the copied module."""
        new = TokenizerEngine(memo.get(id(self.model), self.model))
        memo[id(self)] = new
        return new

    def __getstate__(self):  # torch.save(model) / pickling: drop native state
        return {"model": self.model}

    def __setstate__(self, st):
        self.__init__(st["model"])

    # ------------------------------------------------------------------ parameters / gradients
    def _named(self):
        m = self.model
        d = dict(m.named_parameters())
        d.update(dict(m.named_buffers()))
        return d

    def param_struct(self):
        named = self._named()

        def get(name):
            t = named.get(name)
            if t is None:
                return None
            if not t.is_cuda:
                raise hip.HipError(f"{name} is on {t.device}: the tokenizer hot path runs on the GPU only (no CPU fallback)")
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise hip.HipError(f"{name}: fp32 contiguous master weights required (got {t.dtype})")
            return t.data_ptr()
        return _Tensors(self.model, get)

    def ensure_flat_grad(self, device):
        if self.flat_grad is not None and self.flat_grad.device == device:
            return
        self.order = _flat_order(self.model)
        total, segs, views, off = 0, {}, {}, 0
        for name, p, st in self.order:
            n = p.numel()
            lo, hi = segs.get(st, (off, off))
            segs[st] = (lo, off + n)
            off += n
        total = off
        self.flat_grad = torch.zeros(total + ((-total) % 4), dtype=torch.float32, device=device)  # padded like the flat parameter buffer
        off, offsets = 0, {}
        for name, p, st in self.order:
            views[name] = self.flat_grad[off:off + p.numel()].view(p.shape)
            offsets[name] = off
            off += p.numel()
        self.grad_views = views
        self.grad_offsets = offsets
        self.segments = segs
        self.grad_struct = _Tensors(self.model, lambda n: views[n].data_ptr() if n in views else None)

    def params_version(self):
        return (sum(p._version for p in self.model.parameters()) + sum(b._version for b in self.model.buffers()),
                getattr(self, "param_epoch", 0), getattr(self, "flat_param", None) is not None)

    # ------------------------------------------------------------------ states
    def state_for(self, B, T, S, device):
        m = self.model
        mode, l2n, inv_tau, beta, cw, frozen = m._vq_engine_cfg()
        key = (B, T, S, mode, inv_tau, str(device))
        st = self.states.get(key)
        if st is None:
            c = hip.TokenizerConfig()
            c.B, c.C, c.T, c.S, c.pt, c.p = B, m.in_channels, T, S, m.temporal_patch_size, m.patch_size
            c.D, c.H, c.depth_enc, c.depth_dec = m.encoder_hidden_size, m.encoder_num_heads, m.encoder.depth, m.decoder.depth
            c.Nq, c.d, c.K = m.bottleneck_token_num, m.bottleneck_dim, m.codebook_size
            c.vq_mode, c.l2_normalized = mode, int(l2n)
            c.inv_tau, c.beta, c.codebook_w = inv_tau, beta, cw
            c.freeze_codebook = int(frozen)
            st = _State(self, key, c, device)
            if self.split_k is not None:
                hip.check(hip.lib().vt_tokenizer_set_split_k(st.handle, int(self.split_k)), "vt_tokenizer_set_split_k")
            if self.wgrad_tail:
                hip.check(hip.lib().vt_tokenizer_set_wgrad_tail(st.handle, int(self.wgrad_tail)), "vt_tokenizer_set_wgrad_tail")
            if self.wgrad_stream is not None:
                hip.check(hip.lib().vt_tokenizer_set_wgrad_stream(st.handle, self.wgrad_stream.cuda_stream), "vt_tokenizer_set_wgrad_stream")
            if self.wgrad_batch:
                hip.check(hip.lib().vt_tokenizer_set_wgrad_batch(st.handle, self.wgrad_batch), "vt_tokenizer_set_wgrad_batch")
            if self.data_parallel:
                hip.check(hip.lib().vt_tokenizer_set_data_parallel(st.handle, 1), "vt_tokenizer_set_data_parallel")
            self.states[key] = st
        return st

    def set_data_parallel(self, on):
        """A collective's workgroups will hold CUs during the backward (vt_tokenizer_set_data_parallel): its GEMMs with more tiles than
        the chip has CUs go out one tile per workgroup.  Bit-identical gradients."""
        self.data_parallel = bool(on)
        for st in self.states.values():
            hip.check(hip.lib().vt_tokenizer_set_data_parallel(st.handle, int(self.data_parallel)), "vt_tokenizer_set_data_parallel")

    def set_wgrad_batch(self, n):
        """Blocks per grouped weight-gradient launch, 1..4 (vt_tokenizer_set_wgrad_batch; the library's default is 4).  Bit-identical gradients."""
        self.wgrad_batch = int(n)
        for st in self.states.values():
            hip.check(hip.lib().vt_tokenizer_set_wgrad_batch(st.handle, self.wgrad_batch), "vt_tokenizer_set_wgrad_batch")

    def set_wgrad_stream(self, stream):
        """Data-parallel runs (vt_tokenizer_set_wgrad_stream): the deferred weight-gradient launches and the partial-sum reductions of the same
        blocks run on `stream` (a torch.cuda.Stream, or None for the single-stream schedule) next to the backward's critical path; the engine
        orders both streams with events and joins them at the last stage of backward.  Bit-identical gradients.  Whoever consumes a finished
        gradient slice before the end of backward (parallel.GradReducer) has to wait for this stream as well."""
        if any(st.graphed for st in self.states.values()) and stream is not self.wgrad_stream:
            raise RuntimeError("set_wgrad_stream: a geometry of this engine is captured in a hipGraph (GraphedStep) with the schedule it had at "
                               "capture time; close() the GraphedStep before changing streams")
        self.wgrad_stream = stream
        for st in self.states.values():
            hip.check(hip.lib().vt_tokenizer_set_wgrad_stream(st.handle, stream.cuda_stream if stream is not None else None), "vt_tokenizer_set_wgrad_stream")

    def set_wgrad_tail(self, n):
        """Data-parallel runs (vt_tokenizer_set_wgrad_tail): the encoder's blocks below n flush their weight gradients block by block, so
        the gradient slice that becomes final at the very end of backward -- whose all-reduce nothing can hide -- is one block's instead of
        four blocks'.  Same kernels on the same operands: bit-identical gradients; ~0.17 ms more compute per step at config B."""
        self.wgrad_tail = max(0, int(n))
        for st in self.states.values():
            hip.check(hip.lib().vt_tokenizer_set_wgrad_tail(st.handle, self.wgrad_tail), "vt_tokenizer_set_wgrad_tail")

    def set_split_k(self, on):
        """Split K in the backward input-gradient GEMMs at one / two clips per GPU (vt_tokenizer_set_split_k): +8 % there, at the price of
        gradients that agree with a larger batch's only to the bf16 noise level.  Forward results never depend on it."""
        self.split_k = bool(on)
        for st in self.states.values():
            hip.check(hip.lib().vt_tokenizer_set_split_k(st.handle, int(self.split_k)), "vt_tokenizer_set_split_k")

    def ensure_packed(self, st, pstruct):
        v = self.params_version()
        if st.packed_version != v or st.graphed:      # under capture the pack is part of the graph: weights change every replay
            hip.check(hip.lib().vt_tokenizer_pack(st.handle, ctypes.byref(pstruct.struct), hip.ptr(st.ws), hip.stream()), "vt_tokenizer_pack")
            st.packed_version = v

    def next_seed(self, st=None):
        base = int(torch.initial_seed()) & 0xFFFFFFFF
        # Only the geometry a GraphedStep captured reads its per-call word from the device counter (vt_vq_forward_ctr); an eager
        # forward at ANY other geometry while the graph is alive (validation at another batch size) keeps drawing from the host
        # counter -- engine-wide, this flag gave every such call seed base << 32, i.e. identical noise (advisor finding, round 3).
        if st is not None and st.graphed:
            return base << 32
        self.seed_counter += 1
        return (base << 32) | (self.seed_counter & 0xFFFFFFFF)


def _outputs(m, B, device):
    Nq, D, d, K = m.bottleneck_token_num, m.decoder_hidden_size, m.bottleneck_dim, m.codebook_size
    f = dict(device=device, dtype=torch.float32)
    return {
        "encoded": torch.empty(B, Nq, D, **f), "indices": torch.empty(B, Nq, device=device, dtype=torch.int64),
        "projected_z": torch.empty(B, Nq, d, **f), "unregularized_z": torch.empty(B, Nq, d, **f),
        "regularized_z": torch.empty(B, Nq, d, **f), "emb": torch.empty(K, d, **f),
        "losses": torch.empty(4, **f), "input_norms": torch.empty(2, **f),
    }


def _out_struct(o, pred=None):
    s = hip.TokenizerOutputs()
    s.pred_frames = pred.data_ptr() if pred is not None else None
    for k in ("encoded", "indices", "projected_z", "unregularized_z", "regularized_z", "emb", "losses", "input_norms"):
        setattr(s, k, o[k].data_ptr())
    return s


def _check_video(m, x):
    if not x.is_cuda:
        raise hip.HipError("LARPTokenizer: input is on the CPU; the hot path runs on MI355X only (no CPU fallback)")
    assert x.dim() == 5 and x.shape[1] == m.in_channels, "data: video in shape (b, c, t, h, w)"
    m.x_embedder.check_input(x)
    return x.contiguous().float()


def run_encode(engine, x):
    m = engine.model
    x = _check_video(m, x)
    B, _, T, S, _ = x.shape
    st = engine.state_for(B, T, S, x.device)
    ps = engine.param_struct()
    engine.ensure_packed(st, ps)
    o = _outputs(m, B, x.device)
    os_ = _out_struct(o)
    hip.check(hip.lib().vt_tokenizer_encode(st.handle, ctypes.byref(ps.struct), hip.ptr(x), hip.ptr(st.ws), ctypes.byref(os_),
                                            engine.next_seed(st), hip.stream()), "vt_tokenizer_encode")
    st.fwd_id += 1
    return st, ps, o


def run_decode(engine, st, ps, encoded, B, T, S):
    m = engine.model
    pred = torch.empty(B, m.out_channels, T, S, S, device=encoded.device, dtype=torch.float32)
    hip.check(hip.lib().vt_tokenizer_decode(st.handle, ctypes.byref(ps.struct), hip.ptr(encoded), hip.ptr(st.ws), hip.ptr(pred), hip.stream()),
              "vt_tokenizer_decode")
    return pred


class TokenizerFunction(torch.autograd.Function):
    """forward(video) -> (pred_frames, losses[4], + non-differentiable VQ outputs); backward fills
    the flat gradient buffer through vt_tokenizer_backward and hands views of it to autograd."""

    @staticmethod
    def forward(ctx, engine, x, *params):
        m = engine.model
        st, ps, o = run_encode(engine, x)
        B, _, T, S, _ = x.shape
        pred = run_decode(engine, st, ps, o["encoded"], B, T, S)
        ctx.engine, ctx.st, ctx.fwd_id = engine, st, st.fwd_id
        ctx.n_params = len(params)
        ctx.x_shape = x.shape
        nd = (o["encoded"], o["indices"], o["projected_z"], o["unregularized_z"], o["regularized_z"], o["emb"], o["input_norms"])
        ctx.mark_non_differentiable(*nd)
        return (pred, o["losses"]) + nd

    @staticmethod
    def backward(ctx, d_pred, d_losses, *unused):
        engine, st = ctx.engine, ctx.st
        if st.fwd_id != ctx.fwd_id:
            raise hip.HipError("LARPTokenizer.backward: the engine workspace was overwritten by a later forward with the same "
                               "geometry; run backward before the next forward")
        dev = st.ws.device
        engine.ensure_flat_grad(dev)
        if d_pred is None:
            d_pred = torch.zeros(ctx.x_shape, device=dev, dtype=torch.float32)
        d_pred = d_pred.contiguous().float()
        gscal = d_losses.contiguous().float() if d_losses is not None else None
        ps = engine.param_struct()
        lib = hip.lib()
        red = engine.reducer
        named = dict(engine.model.named_parameters())
        # Gradient accumulation: a parameter whose .grad already IS a view of the flat buffer (no zero_grad(set_to_none=True)
        # since the last backward) would lose its accumulated value when the kernels overwrite the slot.  Save those slots,
        # let the kernels write, then hand autograd the NEW gradient as a tensor of its own and put the old value back --
        # AccumulateGrad adds in place (the slot ends as old + new), and torch.autograd.grad() gets a real tensor without
        # .grad being touched.  Only aliased slots are copied; the common path (grads set to None each step) copies nothing.
        lo_ptr, hi_ptr = engine.flat_grad.data_ptr(), engine.flat_grad.data_ptr() + engine.flat_grad.numel() * 4
        aliased = [n for n in engine.apply_names if named[n].grad is not None and lo_ptr <= named[n].grad.data_ptr() < hi_ptr]
        saved = {n: engine.grad_views[n].clone() for n in aliased}
        final = ctypes.c_int32(0)
        done = 0
        # without a reducer the whole backward is one enqueue; with one, the engine runs until the next group of weight gradients is
        # flushed (a slice of the flat buffer becomes final), the slice goes to the reducer, and so on: ~8 calls at 12 + 12 blocks
        if red is None:
            hip.check(lib.vt_tokenizer_backward(st.handle, ctypes.byref(ps.struct), hip.ptr(d_pred), hip.ptr(gscal), hip.ptr(st.ws),
                                                ctypes.byref(engine.grad_struct.struct), 0, st.nstages, ctypes.byref(final), hip.stream()),
                      "vt_tokenizer_backward")
        else:
            if getattr(red, "early_release", False):
                red.total = max(hi for _, hi in engine.segments.values())    # the reducer lets the last slices go as they are reported
            nxt, stage = ctypes.c_int32(0), 0
            while stage < st.nstages:
                hip.check(lib.vt_tokenizer_backward_until_flush(st.handle, ctypes.byref(ps.struct), hip.ptr(d_pred), hip.ptr(gscal), hip.ptr(st.ws),
                                                                ctypes.byref(engine.grad_struct.struct), stage, ctypes.byref(nxt), ctypes.byref(final),
                                                                hip.stream()), "vt_tokenizer_backward_until_flush")
                stage = nxt.value
                while done < final.value:  # stages whose gradients are final (deferred weight gradients flushed)
                    red.segment_ready(engine.flat_grad, *engine.segments[done])
                    done += 1
        if red is not None:
            red.finish()
        grads = []
        # autograd wants gradients in the order the parameters were passed to apply().  NOTE for callers: except in the aliased
        # case these are views of ONE reused buffer -- the next backward of this model overwrites them.
        for name in engine.apply_names:
            if not named[name].requires_grad:
                grads.append(None)
            elif name in saved:
                fresh = engine.grad_views[name].clone()
                engine.grad_views[name].copy_(saved[name])
                grads.append(fresh)
            else:
                # a FRESH view object (use_count 1) so AccumulateGrad adopts it instead of cloning 694 MB per step
                grads.append(engine.grad_views[name].view(named[name].shape))
        return (None, None) + tuple(grads)


def apply(engine, x):
    order = _flat_order(engine.model)
    engine.apply_names = [n for n, _, _ in order]
    return TokenizerFunction.apply(engine, x, *[p for _, p, _ in order])


class GraphedStep:
    """The whole forward + loss + backward of a fixed geometry captured ONCE as a hipGraph (torch.cuda.CUDAGraph) and replayed.

    Why: at the reference's own recipe -- global batch 8 on 8 GPUs = ONE clip per GPU (scripts/train_larp_tokenizer_reproduce.sh:8,
    trainers/base_trainer.py:316) -- the host spends ~5 ms per step enqueueing ~900 launches (measured 5.2 ms next to 7.75 ms of GPU
    time): a replay costs the host 0.3-0.4 ms and frees it for the loader and the optimizer, the GPU time per step stays
    what it was.  What makes the step capturable: the engine never allocates or synchronises; the weight
    re-pack runs inside the graph; the stochastic quantizer's per-call seed word is a DEVICE counter incremented in the graph
    (vt_vq_forward_ctr), giving the same noise sequence as eager calls; gradients land in the engine's flat buffer, whose views
    the parameters' .grad keep pointing at.  The optimizer step stays outside (FusedAdam.step() is 4-5 launches).

        graphed = GraphedStep(model, x_example, loss_fn)          # loss_fn(out_dict, x) -> scalar tensor
        for x in loader:
            loss, out = graphed(x)                                # replay; `loss` and the kept outputs are static buffers, valid until the next replay
            opt.step()                                            # do NOT set grads to None in between (zero_grad(set_to_none=False) or nothing)

    Under parallel.DataParallelTokenizer (round 5) the capture includes the data-parallel schedule: the stage-by-stage backward, the
    reducer's event / wait pairs, its RCCL all-reduces on the communication stream (torch's NCCL process group captures its collectives
    like any other launch) and, when it is on, the side stream of the weight-gradient launches -- every forked stream joins the capturing
    stream before the backward returns (GradReducer.finish, join_wgrad_stream), which is what stream capture asks for.  Every rank must
    construct its GraphedStep at the same point and replay in lock-step, like any other collective call.  Pass the wrapper or its .module.

    self_check (default on) runs loss_fn for two eager steps and four replays on two clips and requires bit equality, so loss_fn must be a pure
    function of (out, x) and the module's parameters: a loss with state of its own that advances per call (the discriminator's spectral-norm power
    iteration, a global_step counter, an adaptive-weight EMA) either fails the check spuriously or has its state advanced six times before
    training starts -- snapshot / restore that state around the constructor or pass self_check=False and verify the first replays yourself.
    The check leaves every .grad pointing at the flat buffer's views holding its LAST replay's gradients; the first real replay overwrites them."""

    def __init__(self, model, x, loss_fn, warmup=2, outputs=("bottleneck_rep", "loss_q", "loss_commit", "loss_codebook"), self_check=True):
        if getattr(model, "_engine", None) is None and hasattr(model, "module"):
            model = model.module                    # a DataParallelTokenizer: its forward is the module's, the reducer hangs on the engine
        eng = model._engine
        if eng is None:
            raise NotImplementedError("GraphedStep needs the fused engine (this model runs on the composed path)")
        if eng.reducer is not None and (getattr(eng.reducer, "check_late_writers", False) or getattr(eng.reducer, "record_events", False)):
            raise RuntimeError("GraphedStep: GradReducer.check_late_writers (a device-wide synchronisation at every reported slice) and "
                               "record_events (timing events) are inspection aids of the eager step and cannot be captured; switch them off first")
        self.model, self.engine, self.loss_fn = model, eng, loss_fn
        from .optim import flatten_parameters
        flatten_parameters(model)       # parameter ADDRESSES are baked into the graph: move them into the flat buffer FusedAdam uses now, not later
        self.x = x.detach().clone().contiguous().float()
        B, _, T, S, _ = self.x.shape
        self.state = eng.state_for(B, T, S, self.x.device)
        self.ctr = torch.full((1,), eng.seed_counter & 0x7FFFFFFF, dtype=torch.int32, device=self.x.device)
        hip.check(hip.lib().vt_tokenizer_set_seed_counter(self.state.handle, hip.ptr(self.ctr)), "vt_tokenizer_set_seed_counter")
        eng.graph_mode = True
        self.state.graphed = True
        # What the caller gets back.  (1) The LOSS VALUE is recomputed by ordinary launches after the replay (same ops, same bits as
        # eager; ~6 tiny kernels).  Measured on this ROCm build (round 3; the probe that remains is tools/graph_scalar_probe.py): inside a replayed graph a torch
        # elementwise kernel that reads a 0-dim tensor written by an earlier node of the same replay can see the value a previous replay
        # left at that address -- `loss = a + 0.1 * b` came out as 1.0 + 0.1 * b, 1.0 being the backward seed that had reused `a`'s block
        # -- depending on which single-workgroup kernels ran in between (a `torch.equal` outside was enough).  Every kernel of this
        # library reads its operands with vector loads and is unaffected: gradients and weights stayed bit-equal to eager in all nine
        # patterns tried.  Round 4 found the trigger and the switch (tools/graph_stale_scalar_check.sh, profiles/r04_graph_stale_scalar.log):
        # a loss whose SCALAR factors change from step to step (an adaptive weight, a product of two loss terms) back-propagates the
        # PREVIOUS replay's factors -- the engine's d_losses, which torch kernels of the same replay produce, arrives stale -- under the
        # runtime's graph packet capture (pre-recorded kernel packets, the default of this ROCm build) and is correct, twice out of
        # twice, with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0.  It is not a cache effect on this side: a chip-wide L2 write-back + L1 / L2 /
        # scalar-cache invalidation kernel between the nodes changed nothing.  A torch-only graph of the same shape does not show it
        # (tools/graph_scalar_probe.py).  Consequences: (a) video_tokenizer_amd sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 when it is
        # imported before the HIP runtime has read its flags (an existing value is respected); (b) the constructor CHECKS the caller's
        # own loss_fn on two different clips in alternation -- loss and every gradient of each replay against an eager step, bit for
        # bit -- and raises on a difference, so a process where the switch came too late fails loudly instead of training on stale
        # weights.  (2) Small outputs are copied, inside the graph, into buffers from the ordinary pool; large ones are the graph's
        # static tensors.
        self._keep = tuple(outputs)
        self._out = None
        self._graph_out = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # warm-up on a side stream: kernel attributes, workspaces, autograd buffers
            for _ in range(max(1, warmup)):
                self._step()
                eng.seed_counter += 1
        torch.cuda.current_stream().wait_stream(side)
        for p in model.parameters():
            p.grad = None                           # capture with empty .grad: AccumulateGrad adopts the flat-buffer views, no add kernels
        self.stream = side
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._graph_loss, self._graph_out = self._step()
        if self_check:
            self._self_check()

    def _self_check(self):
        """Replays against eager steps, bit for bit, on TWO clips in alternation (the example clip and a scrambled copy), each from its
        own quantizer seed: a scalar that a replay reads as the previous replay left it -- the failure round 4 pinned on the runtime's
        graph packet capture -- shows as soon as consecutive replays differ in their loss terms.  A single-workgroup kernel reads the
        graph-resident loss between replays (the pattern that exposed the stale read in round 3).  Raises RuntimeError on a difference."""
        eng, model = self.engine, self.model
        k0 = int(self.ctr.item())
        x_a = self.x.detach().clone()
        x_b = (1.0 - x_a).flip(-1).contiguous()                    # same range, different content: every loss term changes
        bad = []
        with torch.cuda.stream(self.stream):
            want = []
            for j, xin in enumerate((x_a, x_b)):
                self.x.copy_(xin)
                self.ctr.fill_(k0 + 16 * j)
                for p in model.parameters():
                    p.grad = None
                loss, _ = self._step()
                want.append((loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
            for rep, j in enumerate((0, 1, 0, 1)):
                self.x.copy_((x_a, x_b)[j])
                self.ctr.fill_(k0 + 16 * j)
                self.graph.replay()
                got_loss = self._graph_loss.detach().clone()
                torch.equal(self._graph_loss, got_loss)        # a kernel (not .item()) reads the graph-resident scalar
                if not torch.equal(got_loss, want[j][0]):
                    bad.append(f"replay {rep}: in-graph loss {got_loss.item()!r} != eager {want[j][0].item()!r}")
                for n, g in want[j][1].items():
                    if not torch.equal(eng.grad_views[n].view(g.shape), g):
                        bad.append(f"replay {rep}: gradient of {n} differs from the eager step")
                        break
            self.x.copy_(x_a)
            self.ctr.fill_(k0)
        torch.cuda.current_stream().wait_stream(self.stream)
        for name, p in model.named_parameters():
            if p.requires_grad and name in eng.grad_views:
                p.grad = eng.grad_views[name].view(p.shape)
        if bad:
            self.close()
            raise RuntimeError("GraphedStep self-check failed -- the captured step does not reproduce the eager step:\n  " + "\n  ".join(bad[:6]) +
                               "\nOn this ROCm build replays read stale scalars under the runtime's graph packet capture: set "
                               "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment BEFORE the process touches the GPU (importing video_tokenizer_amd "
                               "first does it), or train with the eager step (DESIGN 6b).  (A loss_fn with state of its own that advances per call fails "
                               "this check by construction: see the class docstring, self_check=False.)")

    def _step(self):
        self.ctr.add_(1)
        out = self.model(self.x)
        loss = self.loss_fn(out, self.x)
        loss.backward()
        if self._out is None:
            self._out = {k: torch.empty_like(out[k]) for k in self._keep if k in out}
        with torch.no_grad():
            for k, buf in self._out.items():
                buf.copy_(out[k])
        return loss, out

    def set_seed_counter(self, k):
        """the next replay draws the quantizer noise of eager call number k + 1"""
        self.ctr.fill_(int(k) & 0x7FFFFFFF)
        self.engine.seed_counter = int(k)

    def __call__(self, x):
        # the replay runs on a stream of its own (not the legacy null stream), ordered against the caller's stream by events on both sides
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.x.copy_(x, non_blocking=True)
            self.graph.replay()
            with torch.no_grad():
                loss = self.loss_fn(self._graph_out, self.x)        # the reported value: ordinary launches, see __init__
        cur.wait_stream(self.stream)
        out = dict(self._graph_out)
        out.update(self._out)
        eng = self.engine
        eng.seed_counter += 1
        for name, p in self.model.named_parameters():          # a caller that dropped .grad gets the (re-written) views back
            if p.grad is None and p.requires_grad and name in eng.grad_views:
                p.grad = eng.grad_views[name].view(p.shape)
        return loss, out

    def close(self):
        hip.check(hip.lib().vt_tokenizer_set_seed_counter(self.state.handle, None), "vt_tokenizer_set_seed_counter")
        self.engine.graph_mode = False
        self.state.graphed = False
