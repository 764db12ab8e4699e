"""Step time of the FSQ autoencoders at the reference's geometry: 16x128x128 clips, 1024 + 1024 tokens.
python tools/titok_bench.py [name] [clips]      (GPU)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import tests.conftest  # noqa: F401,E402  (package alias)
import video_tokenizer_amd as vt  # noqa: E402
from oracle import inputs as gen  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "autoencoder_large"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m = vt.make({"name": name, "args": {"bottleneck": None, "prior_model": None}}).cuda()
video = torch.from_numpy(gen.video_clips(B, 16, 128, 7)).cuda()
n_par = sum(p.numel() for p in m.parameters())
W, layers = m.encoder.width, m.encoder.num_layers
inner = vt.titok.ffd_inner_dim(W)
L = 2048
flops_layer = 2 * L * W * (4 * W + W + 3 * inner) + 4 * L * L * W           # per clip, forward
flops = 3 * 2 * layers * flops_layer


def step():
    for p in m.parameters():
        p.grad = None
    out = m(video)["pred_frames"]
    (out - video).abs().mean().backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{name}: {n_par / 1e6:.0f} M parameters, {B} clips/step, {dt * 1e3:.1f} ms/step, {B / dt:.1f} clips/s, "
      f"{B * flops / dt / 1e12:.0f} TFLOP/s (transformer layers only, fwd+bwd = 3 x fwd)")
