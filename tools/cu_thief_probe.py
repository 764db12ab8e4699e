"""What the training step loses while a collective's kernel holds CUs -- measured on ONE GPU with a stand-in.

RCCL's all-reduce runs as persistent workgroups (one per channel) next to the backward of the step (parallel.GradReducer, side
stream).  Such a workgroup cannot share a CU with a 192x192 GEMM workgroup (LDS and registers are full), so while it runs the
GEMMs of the step see 256 - n CUs.  tools/probes/cu_thief.hip holds n CUs the same way (n workgroups x 256 threads x 96 KiB LDS,
spinning on the real-time counter).  This script runs bench.py's step next to it

  * for the WHOLE step (the upper bound), n = 4 ... 64, single-stream schedule;
  * and in BURSTS: a thief of `burst_ms` started when the first gradient slice of every backward is final -- where the first bucket's
    collective would start -- under the single-GPU schedule, under DataParallelTokenizer's (block-by-block tail, multi-round GEMMs one
    tile per workgroup, weight gradients on their own stream: engine.set_wgrad_tail / set_wgrad_stream) and under that without the stream.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/cu_thief.hip -o tools/probes/_bin/libcu_thief.so   # in the dev container
    python3 tools/cu_thief_probe.py [clips] [burst_ms,burst_ms,...]"""
import ctypes
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import video_tokenizer_amd as vt  # noqa: E402
from video_tokenizer_amd.config import geometry, model_spec  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
BURST = 3.5
thief = ctypes.CDLL(os.path.join(R, "tools", "probes", "_bin", "libcu_thief.so"))
thief.thief_launch.restype = ctypes.c_int
thief.thief_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]

c = geometry("B")
torch.manual_seed(1234)
model = vt.make(model_spec(c, stochastic=True))
with torch.no_grad():
    torch.nn.init.xavier_uniform_(model.final_layer.linear.weight)
model = model.cuda().train()
eng = model._engine
x = torch.from_numpy(vt.config.synthetic_clips(B, c["frame_num"], c["input_size"], 100)).cuda()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
thief_stream = torch.cuda.Stream()
wg_stream = torch.cuda.Stream(priority=int(os.environ.get("VT_THIEF_WG_PRIORITY", "0")))


class ThiefAtFirstBucket:
    """Stands where parallel.GradReducer stands (engine.reducer): the backward then runs stage by stage, and when the FIRST gradient
    slice is reported final -- the moment the first bucket's collective would be launched -- the thief starts, behind the kernels
    enqueued so far on both streams."""

    def __init__(self):
        self.cus, self.fired = 0, False

    def segment_ready(self, flat, lo, hi):
        if self.cus and not self.fired:
            thief_stream.wait_stream(torch.cuda.current_stream())
            if eng.wgrad_stream is not None:
                thief_stream.wait_stream(eng.wgrad_stream)
            assert thief.thief_launch(self.cus, BURST, sink.data_ptr(), thief_stream.cuda_stream) == 0
            self.fired = True

    def finish(self):
        self.fired = False


at_bucket = ThiefAtFirstBucket()


def step(burst_cus=0):
    out = model(x)
    loss = (out["pred_frames"] - x).abs().mean() + 0.1 * out["loss_q"]
    for p in model.parameters():
        p.grad = None
    at_bucket.cus = burst_cus
    loss.backward()


def timed(hold_cus=0, burst_cus=0, steps=10):
    for _ in range(3):
        step(burst_cus)
    torch.cuda.synchronize()
    if hold_cus:
        # the thief outlives the timed steps (bounded: 0.6 s) and is released by its own clock; start it first so it is resident
        assert thief.thief_launch(hold_cus, 600.0, sink.data_ptr(), thief_stream.cuda_stream) == 0
        time.sleep(0.02)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step(burst_cus)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / steps
    torch.cuda.synchronize()          # the thief's time runs out
    assert ms * steps < 550.0, "the timed steps outlived the thief: lower `steps`"
    return ms


def schedule(kind):
    """0: the single-GPU schedule; 1: DataParallelTokenizer's default (block-by-block tail, the backward's multi-round GEMMs one tile per
    workgroup); 2: 1 + the weight gradients on their own stream (VT_WGRAD_STREAM=1)"""
    eng.set_wgrad_stream(wg_stream if kind >= 2 else None)
    eng.set_wgrad_tail(3 if kind else 0)
    eng.set_wgrad_batch({3: 2, 4: 1}.get(kind, 4))


NAMES = ("single-GPU schedule", "data-parallel default (tail 3-2|1|0, one tile per workgroup)", "data-parallel + weight gradients on their own stream",
         "... + groups of 2 blocks", "... + groups of 1 block")
KINDS = tuple(int(v) for v in os.environ.get("VT_THIEF_KINDS", "0,1,2").split(","))
eng.reducer = at_bucket            # every backward of this script runs stage by stage, as under DataParallelTokenizer
schedule(0)
base = timed()
print(f"{B} clips per GPU, forward + backward, ms per step (10 steps, events).  single-GPU schedule, no thief: {base:.2f}", flush=True)
for n in (4, 16, 64):
    ms = timed(hold_cus=n)
    print(f"  {n:3d} CUs held for the whole step: {ms:.2f} ms (+{100 * (ms / base - 1):.1f} %; CUs lost {100 * n / 256:.1f} %)", flush=True)
bursts = [float(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [3.5]
print("bursts: a thief of 16 CUs for b ms, started when the first gradient slice of every backward is final (where the first bucket's collective starts)", flush=True)
for rnd in range(2):
    for kind in KINDS:
        schedule(kind)
        free = timed()
        line = f"  {NAMES[kind]}: no thief {free:.2f}"
        for b in bursts:
            BURST = b
            ms = timed(burst_cus=16)
            line += f" | {b} ms: {ms:.2f}"
        hold = timed(hold_cus=16)
        line += f" | whole step: {hold:.2f}"
        print(line, flush=True)
schedule(0)
