"""GAN branch timing at the shipped size (cfgs/larp_tokenizer.yaml:113-136): generator-side pass (D frozen, fwd + bwd to the
reconstruction) and discriminator update (real + fake fwd, bwd to the parameters), 8 clips of 16x128x128.  (GPU box)
usage: python tools/disc_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_tokenizer_amd as vt  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, host


if __name__ == "__main__":
    B = 8
    lm = vt.make({"name": "lpips_disc_loss", "args": dict(
        disc_type="transformer", disc_start=0, disc_self_start=-1, pixelloss_weight=1.0, perceptual_weight=0.0, pixel_loss="l1",
        lecam_weight=0.001, disc_loss="ns_smooth", disc_weight=0.3, r1_gp_weight=0.0, d_update_freq=5, spectral_norm=False,
        disc_tran_hidden_size=384, disc_tran_n_heads=12, disc_tran_n_layers=8, disc_tran_temporal_patch_size=4, disc_tran_patch_size=8,
        input_spatial_size=128, frame_num=16)}).cuda()
    real = torch.rand(B, 3, 16, 128, 128, device="cuda")
    fake = torch.rand(B, 3, 16, 128, 128, device="cuda", requires_grad=True)

    def g_step():
        lm.trainable_requires_grad_(False)
        loss, _, _ = lm(real, fake, 10, for_discriminator=False)
        loss.backward()

    def d_step():
        lm.trainable_requires_grad_(True)
        loss, _, _ = lm(real, fake.detach(), 10, for_discriminator=True)
        loss.backward()

    flops_fwd = B * 8 * (24 * 1025 * 384 ** 2 + 4 * 1025 ** 2 * 384) + B * 2 * 1024 * 768 * 384
    for name, fn, mult in (("generator-side (fwd + input-grad bwd)", g_step, 1 + 2 * 2 / 3.0), ("discriminator update (2 fwd + full bwd)", d_step, 2 + 2 * 2)):
        gpu, host = timed(fn)
        print(f"{name}: {gpu:.2f} ms GPU ({host:.2f} ms host enqueue)  ~{flops_fwd * mult / gpu / 1e9:.0f} TFLOP/s", flush=True)
