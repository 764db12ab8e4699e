"""N training steps (config B, forward + backward) with n CUs held by tools/probes/cu_thief.hip for their whole duration: the workload
tools/cu_thief_stats.sh traces with rocprofv3 --kernel-trace --stats to see WHICH kernels pay for the missing CUs.
    python3 tools/cu_thief_steps.py <n_cus> [steps]"""
import ctypes
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import video_tokenizer_amd as vt  # noqa: E402
from video_tokenizer_amd.config import geometry, model_spec  # noqa: E402

n_cus = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
thief = ctypes.CDLL(os.path.join(R, "tools", "probes", "_bin", "libcu_thief.so"))
thief.thief_launch.restype = ctypes.c_int
thief.thief_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
c = geometry("B")
torch.manual_seed(1234)
model = vt.make(model_spec(c, stochastic=True))
with torch.no_grad():
    torch.nn.init.xavier_uniform_(model.final_layer.linear.weight)
model = model.cuda().train()
if os.environ.get("VT_THIEF_DP") in ("1", "2"):   # the schedule parallel.DataParallelTokenizer switches on: block-by-block tail + one tile per workgroup for the
    model._engine.set_wgrad_tail(3)               # backward's multi-round GEMMs (1), and the weight gradients on their own stream as well (2)
    if os.environ["VT_THIEF_DP"] == "2":
        wg_stream = torch.cuda.Stream()
        model._engine.set_wgrad_stream(wg_stream)
x = torch.from_numpy(vt.config.synthetic_clips(8, c["frame_num"], c["input_size"], 100)).cuda()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()


def step():
    out = model(x)
    loss = (out["pred_frames"] - x).abs().mean() + 0.1 * out["loss_q"]
    for p in model.parameters():
        p.grad = None
    loss.backward()


torch.cuda.synchronize()
if n_cus:
    assert thief.thief_launch(n_cus, 900.0, sink.data_ptr(), side.cuda_stream) == 0
    time.sleep(0.02)
t0 = time.perf_counter()
for _ in range(steps):
    step()
ev = torch.cuda.Event()
ev.record()
ev.synchronize()
print(f"{n_cus} CUs held: {1e3 * (time.perf_counter() - t0) / steps:.2f} ms per step (including the first, cold ones)", flush=True)
torch.cuda.synchronize()
