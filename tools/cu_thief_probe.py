"""What the training step loses while a collective's kernel holds CUs -- measured on ONE GPU with a stand-in.

RCCL's all-reduce runs as persistent workgroups (one per channel) next to the backward of the step (parallel.GradReducer, side
stream).  Such a workgroup cannot share a CU with a 192x192 GEMM workgroup (LDS and registers are full), so while it runs the
GEMMs of the step see 256 - n CUs.  tools/probes/cu_thief.hip holds n CUs the same way (n workgroups x 256 threads x 96 KiB LDS,
spinning on the real-time counter); this script runs bench.py's step with the thief on a side stream for the WHOLE step (the upper
bound: an all-reduce covers only part of the backward) and for n = 0, 4, 8, 16, 32, 64 prints ms per step.

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/cu_thief.hip -o tools/probes/_bin/libcu_thief.so   # in the dev container
    python3 tools/cu_thief_probe.py [clips]"""
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
thief = ctypes.CDLL(os.path.join(R, "tools", "probes", "_bin", "libcu_thief.so"))
thief.thief_launch.restype = ctypes.c_int
thief.thief_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]

c = geometry("B")
torch.manual_seed(1234)
model = vt.make(model_spec(c, stochastic=True))
with torch.no_grad():
    torch.nn.init.xavier_uniform_(model.final_layer.linear.weight)
model = model.cuda().train()
x = torch.from_numpy(vt.config.synthetic_clips(B, c["frame_num"], c["input_size"], 100)).cuda()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()


def step():
    out = model(x)
    loss = (out["pred_frames"] - x).abs().mean() + 0.1 * out["loss_q"]
    for p in model.parameters():
        p.grad = None
    loss.backward()


def timed(n_cus, steps=10):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    if n_cus:
        # the thief outlives the timed steps (bounded: 0.6 s) and is released by its own clock; start it first so it is resident
        rc = thief.thief_launch(n_cus, 600.0, sink.data_ptr(), side.cuda_stream)
        assert rc == 0, rc
        time.sleep(0.02)
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / steps
    torch.cuda.synchronize()          # the thief's 0.6 s run out
    assert ms * steps < 550.0, "the timed steps outlived the thief: lower `steps`"
    return ms


base = timed(0)
print(f"{B} clips per GPU, forward + backward, ms per step (10 steps, events): no thief {base:.2f}", flush=True)
for n in (4, 8, 16, 32, 64):
    ms = timed(n)
    again = timed(0)
    print(f"  {n:3d} CUs held for the whole step: {ms:.2f} ms (+{100 * (ms / base - 1):.1f} %; CUs lost {100 * n / 256:.1f} %)   no thief again: {again:.2f}", flush=True)
