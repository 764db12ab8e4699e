"""Stream-order check of the whole training step WITHOUT a GPU (round-4 verdict item 1, advisor finding 1).

The product's host code -- vt_engine.hip's launch sequence, engine.py's stage-by-stage backward, parallel.GradReducer's bucket
logic and event protocol -- runs unchanged against tests/c/engine_ledger.hip, which records every launch / copy / event record /
stream wait with its stream and the byte ranges it touches.  tests/ledger.py then looks for conflicting accesses on different
streams that no event chain orders.  Covered: the single-stream schedule, the data-parallel schedule (weight gradients on a
second stream, block-by-block tail, the reducer's early release of the last slices at every bucket size down to "each slice the
moment it is reported"), two consecutive steps (the next forward rewrites what the side stream still reads), and the detector
itself (a reducer that forgets the side stream, a slice reported one stage early: both must be caught).
"""
import ctypes
import os
import shutil

import pytest
import torch

from oracle import larp_oracle as O
from tests import ledger as LG

pytestmark = pytest.mark.skipif(shutil.which(LG.HIPCC) is None and not os.path.exists(LG.HIPCC), reason="needs hipcc to link the recording engine")


class FakeStream:
    def __init__(self, handle):
        self.cuda_stream = handle


class LedgerStreams:
    """parallel._HipStreams with every call turned into a ledger entry"""

    def __init__(self, lg, forget_side=False):
        self.lg, self.forget_side = lg, forget_side
        self.main = FakeStream(LG.MAIN)

    def current(self):
        return self.main

    def new_stream(self):
        return FakeStream(LG.COMM)

    def record(self, stream, timing=False):
        if self.forget_side and stream.cuda_stream == LG.SIDE:
            return None
        return self.lg.record(stream.cuda_stream)

    def wait(self, stream, event):
        if event is not None:
            self.lg.wait(stream.cuda_stream, event)

    def wait_stream(self, stream, other):
        self.lg.wait(stream.cuda_stream, self.lg.record(other.cuda_stream))

    def all_reduce(self, stream, view):
        lo = view.data_ptr()
        self.lg.access("all_reduce", stream.cuda_stream, lo, lo + view.numel() * 4, False)
        self.lg.access("all_reduce", stream.cuda_stream, lo, lo + view.numel() * 4, True)


@pytest.fixture(scope="module")
def lg():
    return LG.Ledger()


@pytest.fixture()
def host_on_ledger(lg, monkeypatch):
    """engine.py / hip.py bound to the recording library, CPU tensors standing in for device memory"""
    import video_tokenizer_amd as vt
    from video_tokenizer_amd import engine as E
    from video_tokenizer_amd import hip
    L = lg.lib
    for name, (res, args) in hip.ENGINE_SIGNATURES.items():
        if name.startswith("vt_tokenizer_"):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
    L.vt_last_error.restype, L.vt_last_error.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t]
    monkeypatch.setattr(hip, "_lib", L)
    monkeypatch.setattr(hip, "ptr", lambda t: None if t is None else ctypes.c_void_p(t.data_ptr()))
    monkeypatch.setattr(hip, "stream", lambda: ctypes.c_void_p(LG.MAIN))
    monkeypatch.setattr(E, "_check_video", lambda m, x: x.contiguous().float())

    def param_struct(self):
        named = self._named()
        return E._Tensors(self.model, lambda n: named[n].data_ptr() if named.get(n) is not None else None)
    monkeypatch.setattr(E.TokenizerEngine, "param_struct", param_struct)
    return vt, E


def make_model(vt, name, **over):
    from video_tokenizer_amd.config import model_spec
    cfg = O.make_cfg(name, **over)
    return vt.make(model_spec(cfg, False)), cfg


def run_steps(lg, E, model, cfg, B, steps, reducer=None, side=False, tail=0, total=True, report_early=False):
    eng = model._engine
    eng.reducer = reducer
    if reducer is not None:
        eng.set_wgrad_tail(tail)
        eng.set_data_parallel(True)
        if side:
            eng.set_wgrad_stream(FakeStream(LG.SIDE))
            reducer.extra_stream = FakeStream(LG.SIDE)
    lg.reset()
    x = torch.zeros(B, 3, cfg["frame_num"], cfg["input_size"], cfg["input_size"])
    if report_early:          # a deliberately wrong engine: every stage reported final one stage too early
        real = E.hip.lib().vt_tokenizer_backward_until_flush

        def early(*a):
            rc = real(*a)
            a[8]._obj.value = min(a[8]._obj.value + 1, eng.states[next(iter(eng.states))].nstages)
            return rc
        E.hip.lib().vt_tokenizer_backward_until_flush = early
    try:
        for _ in range(steps):
            for p in model.parameters():
                p.grad = None
            out = E.apply(eng, x)
            (out[0].sum() + out[1][0]).backward()
            # the optimizer reads every gradient and rewrites every parameter on the compute stream
            fg = eng.flat_grad
            lg.access("optimizer_reads_grads", LG.MAIN, fg.data_ptr(), fg.data_ptr() + fg.numel() * 4, False)
    finally:
        if report_early:
            E.hip.lib().vt_tokenizer_backward_until_flush = real
    return lg.ops()


def fake_reducer(lg, bucket_bytes, total=True, forget_side=False):
    from video_tokenizer_amd.parallel import GradReducer
    r = GradReducer.__new__(GradReducer)
    r.pg, r.world, r.bucket_elems = None, 1, max(1, bucket_bytes // 4)
    r.pending, r.total, r.comm_stream, r.launched, r.use_avg = None, None, None, [], True
    r.extra_stream, r.record_events, r.events, r.compute_done, r._alive = None, False, [], None, []
    r.streams = LedgerStreams(lg, forget_side)
    r.early_release, r.check_late_writers, r._snapshots = total, False, []
    return r


GEOMS = {
    # 2 + 2 blocks, L = 64: the geometry of test_data_parallel_wrapper_single_rank_rccl (the test that failed once in round 4)
    "tiny": ("tiny", {}, 2),
    # both stacks' last blocks on the kept rows only (compact buffers that do not rotate with the gradient sets)
    "lastskip": ("tiny", {"frame_num": 8, "input_size": 64, "bottleneck_token_num": 128}, 2),
    # 12 + 12 blocks: 4-block weight-gradient groups, 9 gradient sets in rotation, sets re-used within one backward
    "deep": ("tiny", {"encoder_depth": 12, "decoder_depth": 12}, 1),
    # the headline geometry (config B, 8 clips): the addresses the real step uses, workspace reserved not touched
    "B": ("B", {}, 8),
}


@pytest.mark.parametrize("geom", list(GEOMS))
def test_single_stream_schedule_has_no_cross_stream_access(lg, host_on_ledger, geom):
    vt, E = host_on_ledger
    name, over, B = GEOMS[geom]
    model, cfg = make_model(vt, name, **over)
    ops = run_steps(lg, E, model, cfg, B, steps=2)
    assert {o.stream for o in ops} == {LG.MAIN}, {hex(o.stream) for o in ops}
    bad, n = LG.races(ops)
    assert not bad, bad


def _dp_cases():
    out = []
    for geom in GEOMS:
        for bucket in (64 << 20, 8 << 20, 1024):
            for side, tail in ((True, 3), (False, 3), (True, 0), (False, 0)):
                # the headline geometry builds a 173 M-parameter model per case: the shipped schedule and the plain one, largest and smallest bucket
                if geom == "B" and not ((side, tail) in ((True, 3), (False, 0)) and bucket != 8 << 20):
                    continue
                out.append(pytest.param(geom, bucket, side, tail, id=f"{geom}-bucket{bucket}-side{int(side)}-tail{tail}"))
    return out


@pytest.mark.parametrize("geom,bucket,side,tail", _dp_cases())
def test_data_parallel_schedule_orders_every_conflicting_access(lg, host_on_ledger, geom, bucket, side, tail):
    """Two steps under the reducer.  bucket 1024 = every slice goes to the collective the moment the engine reports it: the strongest
    form of the early release that round 4 took out after one unexplained mismatch."""
    vt, E = host_on_ledger
    name, over, B = GEOMS[geom]
    model, cfg = make_model(vt, name, **over)
    red = fake_reducer(lg, bucket)
    ops = run_steps(lg, E, model, cfg, B, steps=2, reducer=red, side=side, tail=tail)
    streams = {o.stream for o in ops}
    assert LG.COMM in streams and (LG.SIDE in streams) == side
    total = sum(p.numel() for p in model.parameters())
    assert red.launched[0][0] == 0 and red.launched[-1][1] == total and all(a[1] == b[0] for a, b in zip(red.launched, red.launched[1:]))
    bad, n = LG.races(ops)
    assert not bad, (n, bad)


def test_detector_catches_a_reducer_that_ignores_the_side_stream(lg, host_on_ledger):
    vt, E = host_on_ledger
    model, cfg = make_model(vt, "tiny")
    red = fake_reducer(lg, 1024, forget_side=True)
    ops = run_steps(lg, E, model, cfg, 2, steps=1, reducer=red, side=True, tail=3)
    bad, n = LG.races(ops)
    assert n > 0 and any("gemm_tn_grouped@side" in b and "all_reduce@comm" in b for b in bad), bad


def test_detector_catches_a_slice_reported_before_its_last_writer(lg, host_on_ledger):
    vt, E = host_on_ledger
    model, cfg = make_model(vt, "tiny", encoder_depth=12, decoder_depth=12)
    red = fake_reducer(lg, 1024)
    ops = run_steps(lg, E, model, cfg, 1, steps=1, reducer=red, side=True, tail=3, report_early=True)
    bad, n = LG.races(ops)
    assert n > 0 and any("all_reduce@comm" in b for b in bad), bad
