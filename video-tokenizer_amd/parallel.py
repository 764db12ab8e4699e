"""Data-parallel training of the tokenizer: one process per GPU, gradient averaging over RCCL.

The reference wraps the model in torch DDP (trainers/base_trainer.py:388: NCCL, 25 MB buckets,
find_unused_parameters=True).  Here the engine writes all gradients into ONE flat fp32 buffer in the
order backward produces them, and reports each finished contiguous slice (head, decoder blocks,
bottleneck, encoder blocks, patch-embed).  `GradReducer` coalesces finished slices into buckets and
all-reduces each bucket on a dedicated HIP stream behind an event recorded on the compute stream, so
the collective of bucket k overlaps the backward kernels of later stages.  MI355X: the 8 GPUs of a
node are fully connected by xGMI (7 links x ~153 GB/s per GPU); buckets are sized large (64 MB) so
every RCCL call is bandwidth- not latency-bound, and there are only ~11 of them per step at 694 MB.
Semantics preserved from DDP: gradient MEAN over ranks; parameters broadcast from rank 0 at wrap time.
The same code runs on CPU tensors with the gloo backend (tests), synchronously.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


class _HipStreams:
    """The reducer's stream plumbing on the device: events, waits and the collective itself.  Kept apart from the bucket logic so
    that tests/test_engine_ledger_cpu.py can run the SAME `GradReducer` against a recording stand-in and check, without a GPU, that
    every writer of a slice happens-before the collective that reads it."""

    def __init__(self, device, process_group, use_avg, world):
        self.device, self.pg, self.use_avg, self.world = device, process_group, use_avg, world

    def current(self):
        return torch.cuda.current_stream(self.device)

    def new_stream(self):
        return torch.cuda.Stream(device=self.device)

    def record(self, stream, timing=False):
        ev = torch.cuda.Event(enable_timing=timing)
        ev.record(stream)
        return ev

    def wait(self, stream, event):
        stream.wait_event(event)

    def wait_stream(self, stream, other):
        stream.wait_stream(other)

    def all_reduce(self, stream, view):
        with torch.cuda.stream(stream):
            if self.use_avg:
                dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
                view.mul_(1.0 / self.world)


class GradReducer:
    def __init__(self, process_group=None, bucket_bytes=64 << 20, early_release=True):
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.early_release = early_release   # once less than a bucket is left to come, finished slices go out as they are reported
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.pending = None  # (flat, lo, hi)
        self.total = None    # elements that will be reported in one backward (set by the engine); None: only full buckets go out early
        self.comm_stream = None
        self.streams = None  # _HipStreams, made at the first device-side launch (tests substitute a recording one)
        self.launched = []   # (lo, hi) ranges reduced in this backward (for tests/inspection)
        # RCCL averages inside the collective (ncclAvg): saves a read-modify-write pass over the 694 MB of gradients that
        # would compete with the backward kernels for HBM; gloo (CPU tests) has no AVG, so sum then scale there
        self.use_avg = dist.get_backend(process_group) == "nccl"
        # inspection (tests / tools): with record_events set, every bucket's collective is bracketed by timing events on the
        # comm stream and finish() records one on the compute stream behind the last backward kernel -> `events`
        # a second stream that also writes gradients (engine.set_wgrad_stream): a slice is ready when BOTH streams have passed this point
        self.extra_stream = None
        self.record_events = False
        self.events = []      # [(start, stop)] per launched bucket of the last backward
        self.compute_done = None
        self._alive = []      # the ordering events of this backward: kept until the next backward starts (never destroyed while a wait is queued)
        # debugging aid (tests/test_parallel_gpu.py; the advisor's round-4 suggestion): with check_late_writers set, every reported slice is
        # snapshotted behind a device-wide synchronisation BEFORE its collective is enqueued; check_snapshots() after the backward names
        # every slice that a kernel still wrote after the engine had reported it final (world size 1: the collective changes no value)
        self.check_late_writers = False
        self._snapshots = []

    def _launch(self, flat, lo, hi):
        if hi <= lo:
            return
        view = flat[lo:hi]
        if self.streams is None and flat.is_cuda:
            self.streams = _HipStreams(flat.device, self.pg, self.use_avg, self.world)
        st = self.streams
        if st is not None:
            if self.comm_stream is None:
                self.comm_stream = st.new_stream()
            ev = st.record(st.current())               # slice fully written by the kernels enqueued on the compute stream so far
            st.wait(self.comm_stream, ev)
            self._alive.append(ev)
            if self.extra_stream is not None:
                ev2 = st.record(self.extra_stream)     # ... and by the weight-gradient launches enqueued on the side stream so far
                st.wait(self.comm_stream, ev2)
                self._alive.append(ev2)
            e0 = st.record(self.comm_stream, timing=True) if self.record_events else None
            st.all_reduce(self.comm_stream, view)
            if self.record_events:
                self.events.append((e0, st.record(self.comm_stream, timing=True)))
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
            view.mul_(1.0 / self.world)
        self.launched.append((lo, hi))

    def segment_ready(self, flat, lo, hi):
        """The slice flat[lo:hi] is final (kernels enqueued).  Slices arrive in increasing, contiguous order."""
        if self.pending is None:
            self.launched = []
            self.events = []
            self._alive = []
            self._snapshots = []
            self.pending = (flat, lo, hi)
        else:
            f, plo, phi = self.pending
            assert f is flat and phi == lo, "gradient slices must be reported contiguously in order"
            self.pending = (flat, plo, hi)
        if self.check_late_writers and flat.is_cuda:
            torch.cuda.synchronize(flat.device)
            self._snapshots.append((lo, hi, flat[lo:hi].clone()))
        f, plo, phi = self.pending
        # a full bucket goes out; so does anything finished once less than a bucket is left to come (`total` elements in all): the
        # last slices are the ones no later backward kernel can hide, so they do not wait for each other
        if phi - plo >= self.bucket_elems or (self.total is not None and self.total - phi < self.bucket_elems):
            self._launch(f, plo, phi)
            self.pending = (flat, phi, phi)

    def check_snapshots(self, flat):
        """[(lo, hi, differing elements)] for every reported slice whose content changed after it was reported (see check_late_writers)"""
        torch.cuda.synchronize(flat.device)
        return [(lo, hi, int((flat[lo:hi] != snap).sum())) for lo, hi, snap in self._snapshots if not torch.equal(flat[lo:hi], snap)]

    def finish(self):
        """Flush the tail bucket and order the compute stream after every collective."""
        if self.pending is not None:
            f, plo, phi = self.pending
            self._launch(f, plo, phi)
            self.pending = None
            if self.streams is not None and self.comm_stream is not None:
                st = self.streams
                if self.record_events:
                    self.compute_done = st.record(st.current(), timing=True)   # behind the last backward kernel
                st.wait_stream(st.current(), self.comm_stream)


class DataParallelTokenizer(nn.Module):
    """DDP-like wrapper: `.module`, forward passthrough, parameter broadcast at construction."""

    def __init__(self, module, process_group=None, bucket_bytes=64 << 20):
        super().__init__()
        self.module = module
        self.process_group = process_group
        if getattr(module, "_engine", None) is None:
            why = getattr(module, "_composed_why", None)
            raise NotImplementedError("this model runs on the composed path (no fused engine" + (f": {why}" if why else "") + "): its gradients are "
                                      "ordinary .grad tensors, wrap it in torch.nn.parallel.DistributedDataParallel")
        if dist.get_world_size(process_group) > 1:
            # rank 0's weights to everyone (DDP's constructor does the same): the trainable parameters sit in ONE flat buffer
            # (optim.flatten_parameters, idempotent), so they travel as one collective instead of 277; frozen parameters and the three
            # position-embedding buffers follow one by one
            from .optim import flatten_parameters
            with torch.no_grad():
                flat = flatten_parameters(module)
                dist.broadcast(flat, src=0, group=process_group)
                inside = {p.data_ptr() for p in module.parameters()
                          if flat.data_ptr() <= p.data_ptr() < flat.data_ptr() + flat.numel() * flat.element_size()}
                for t in list(module.parameters()) + list(module.buffers()):
                    if t.data_ptr() not in inside:
                        dist.broadcast(t, src=0, group=process_group)
        module._engine.reducer = GradReducer(process_group, bucket_bytes)
        # the slice that becomes final with the LAST weight-gradient launch is reduced with no backward left to hide it: flush the encoder's
        # first three blocks one group at a time (3-2 | 1 | 0) so that slice is one block's 28 MB instead of four blocks' 113 MB
        # (vt_tokenizer_set_wgrad_tail; VT_WGRAD_TAIL=0 keeps the single-GPU schedule)
        module._engine.set_wgrad_tail(int(os.environ.get("VT_WGRAD_TAIL", "3")))
        # a collective's workgroups will hold CUs during the backward: its multi-round GEMMs go out one tile per workgroup (engine.set_data_parallel)
        module._engine.set_data_parallel(os.environ.get("VT_DP_ONE_TILE", "1") != "0")
        if os.environ.get("VT_WGRAD_BATCH"):   # blocks per grouped weight-gradient launch (1..4, default 4): smaller groups report finished
            module._engine.set_wgrad_batch(int(os.environ["VT_WGRAD_BATCH"]))   # slices sooner; measured neutral on compute with the second stream
        # the deferred weight-gradient launches on a stream of their own (vt_tokenizer_set_wgrad_stream; VT_WGRAD_STREAM=0 = single stream).
        # While a collective's workgroups hold CUs, every exact-fit GEMM launch of the backward runs an extra, nearly empty round
        # (tools/cu_thief_probe.py: + 33 % on the step while something is resident); independent weight-gradient work fills part of those
        # rounds.  With the stand-in resident for 2.7 / 5.4 / 12.6 ms of the backward (what an all-reduce of the step's 694 MB keeps
        # resident at 8 / 4 / 2 GPUs) the step takes 24.8 / 25.1 / 26.0 ms against 24.9 / 25.4 / 26.5 ms on one stream
        # (profiles/r04_cu_thief_probe.log); it costs 0.1 ms when nothing is resident, which a data-parallel run never sees.
        if os.environ.get("VT_WGRAD_STREAM", "1") != "0" and next(module.parameters()).is_cuda:
            side = torch.cuda.Stream(device=next(module.parameters()).device)
            module._engine.set_wgrad_stream(side)
            module._engine.reducer.extra_stream = side

    def forward(self, *a, **k):
        return self.module(*a, **k)
