"""Experiment: one training step of 8 clips on the whole chip vs two concurrent steps of 4 clips each, every one on its own
stream restricted to half of the CUs (hipExtStreamCreateWithCUMask).  The step is close to the SUM of its matrix-bound and its
memory-bound kernel time; two de-phased half-chip pipelines let one half's HBM-bound kernels run under the other
half's MFMA-bound ones.  Prints clips/s of both arrangements (same total work).
usage: python tools/cu_split_probe.py [mask_kind: xcd|half|none] [steps]"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import video_tokenizer_amd as vt  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "xcd"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
hiprt = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hiprt.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def build(seed):
    c, spec = bench.yaml_model_args("B")
    torch.manual_seed(seed)
    m = vt.make(spec)
    with torch.no_grad():
        torch.nn.init.xavier_uniform_(m.final_layer.linear.weight)
    return c, m.cuda().train()


def make_step(model, x):
    def step():
        out = model(x)
        loss = (out["pred_frames"] - x).abs().mean() + 0.1 * out["loss_q"]
        for p in model.parameters():
            p.grad = None
        loss.backward()
    return step


c, full = build(1)
x8 = torch.from_numpy(vt.config.synthetic_clips(8, c["frame_num"], c["input_size"], 100)).cuda()
s_full = make_step(full, x8)
for _ in range(3):
    s_full()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    s_full()
torch.cuda.synchronize()
t_full = (time.perf_counter() - t0) / steps
print(f"whole chip, 8 clips/step: {t_full * 1e3:.2f} ms/step = {8 / t_full:.1f} clips/s", flush=True)

if kind == "xcd":
    sets = [{i for i in range(256) if i % 8 < 4}, {i for i in range(256) if i % 8 >= 4}]
elif kind == "half":
    sets = [set(range(128)), set(range(128, 256))]
elif kind == "even":
    sets = [{i for i in range(256) if i % 2 == 0}, {i for i in range(256) if i % 2 == 1}]
else:
    sets = [set(range(256)), set(range(256))]
streams = [masked_stream(s) for s in sets]
halves = []
for k in range(2):
    _, m = build(2 + k)
    x4 = x8[4 * k:4 * k + 4].contiguous()
    halves.append((m, make_step(m, x4)))


def both():
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            halves[k][1]()


for k in range(2):          # warm each replica alone (workspace allocation on its stream)
    with torch.cuda.stream(streams[k]):
        for _ in range(2):
            halves[k][1]()
torch.cuda.synchronize()
# one half alone on its half chip
t0 = time.perf_counter()
with torch.cuda.stream(streams[0]):
    for _ in range(steps):
        halves[0][1]()
torch.cuda.synchronize()
t_half = (time.perf_counter() - t0) / steps
print(f"half chip alone ({kind}), 4 clips/step: {t_half * 1e3:.2f} ms/step = {4 / t_half:.1f} clips/s", flush=True)
for offset in (False, True):
    if offset:      # de-phase: stream 1 starts half a step later
        with torch.cuda.stream(streams[0]):
            halves[0][1]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if offset:
        with torch.cuda.stream(streams[0]):
            x8.mul_(1.0)        # placeholder op; the real offset comes from enqueue order below
    for _ in range(steps):
        both()
    torch.cuda.synchronize()
    t_two = (time.perf_counter() - t0) / steps
    print(f"two half-chip pipelines ({kind}{', offset' if offset else ''}), 2 x 4 clips/step: {t_two * 1e3:.2f} ms/step = {8 / t_two:.1f} clips/s "
          f"({t_full / t_two:.3f}x the whole-chip step)", flush=True)
