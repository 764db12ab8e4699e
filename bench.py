"""Headline benchmark: clips/sec of the LARP tokenizer training step (fwd + bwd, + gradient
all-reduce when N > 1) on synthetic 16x128x128 clips -- BASELINE.json config[1]:
cfgs/larp_tokenizer.yaml base geometry (pt2 p16, 12+12 blocks, 1024 latent tokens, d=24, K=8192),
bs=8 per GPU, bf16 MFMA with fp32 accumulate, `LARPTokenizer(bottleneck_type='vq')`.

  python bench.py [--gpus N --steps K --warmup W]          (N > 1 without a launcher: spawns its own N rank processes)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = model(data) -> L1 reconstruction loss + 0.1*loss_q (the model-side part of
trainers/larp_tokenizer_trainer.py:_iter_step; LPIPS/GAN are out of scope) -> backward through the HIP
engine (-> bucketed RCCL gradient all-reduce); `--optimizer` additionally runs torch.optim.Adam in the step.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# hipGraph replays of this package's steps need the runtime's graph packet capture off; the runtime reads the flag when the process
# first touches HIP, so it is set before torch is imported -- the package would set it too, but only at its own import, which comes after
# torch.cuda.set_device here (it then warns and generation keeps its eager loop)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md (spec, no sparsity)


def yaml_model_args(cfg_name):
    """Model args of the benchmark workload = cfgs/larp_tokenizer.yaml surface with the --opts the SURVEY
    prescribes (model.name larp_tokenizer, bottleneck_type vq, input_size 128)."""
    from video_tokenizer_amd.config import geometry, model_spec
    c = geometry(cfg_name)
    return c, model_spec(c, stochastic=True)  # yaml default: stochastic sampling, tau 0.03


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N rank processes ourselves
    (the reference does the same with mp.spawn, train.py:162-169), as a `torch.distributed.run` CHILD process -- this
    parent has not touched the GPU and never execs.  Rank 0's JSON line passes through on stdout; the exit code is the
    child's (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver (RCCL needs it)
    return subprocess.call(cmd, env=env)


def flops_per_clip(c):
    """fwd+bwd algorithmic FLOPs per clip: F_blk = 24 L D^2 + 4 L^2 D; bwd = 2x fwd."""
    D = 768
    nv = (c["frame_num"] // c["temporal_patch_size"]) * (c["input_size"] // c["patch_size"]) ** 2
    L = nv + c["bottleneck_token_num"]
    kp = 3 * c["temporal_patch_size"] * c["patch_size"] ** 2
    blk = 24 * L * D * D + 4 * L * L * D
    nq = c["bottleneck_token_num"]
    fwd = (c["encoder_depth"] + c["decoder_depth"]) * blk + 2 * (2 * nv * kp * D) + 2 * nq * c["codebook_size"] * c["bottleneck_dim"] \
        + 2 * (2 * nq * D * c["bottleneck_dim"])
    attn = (c["encoder_depth"] + c["decoder_depth"]) * 4 * L * L * D
    return 3 * fwd, 3 * attn


def time_dominant_kernel(B, c, reps=50):
    """Roofline of the dominant kernel = the bf16 NT GEMM (gemm_nt_kernel<VT_EPI_BF16>: qkv forward and
    all three input-gradient GEMMs of a block).  Its launches of one training step are replayed through the
    C ABI with the same shapes and timed with HIP events on the stream they run on; achieved = sum of
    algorithmic FLOPs / sum of durations."""
    import video_tokenizer_amd.hip as hip
    D = 768
    nv = (c["frame_num"] // c["temporal_patch_size"]) * (c["input_size"] // c["patch_size"]) ** 2
    M = B * (nv + c["bottleneck_token_num"])
    nblk = c["encoder_depth"] + c["decoder_depth"]
    shapes = [(M, 3 * D, D, nblk), (M, D, 4 * D, nblk), (M, D, D, nblk), (M, D, 3 * D, nblk)]  # qkv fwd, fc1 dgrad, proj dgrad, qkv dgrad
    tot_f, tot_t, per = 0.0, 0.0, []
    ops = []
    for (m, n, k, mult) in shapes:
        A = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        Bm = (torch.randn(n, k, device="cuda") * 0.03).to(torch.bfloat16)
        ops.append((A, Bm, torch.empty(m, n, device="cuda", dtype=torch.bfloat16)))
    # The GPU has been idle while the CPU baseline ran: its first ~10 ms of matrix work run 5-10 % slow (clocks, caches), which the
    # training step never sees.  ~25 ms of the same launches go first, untimed; then each shape is timed over `reps` launches.
    for _ in range(15):
        for A, Bm, out in ops:
            for _ in range(10):
                hip.gemm_nt(A, Bm, hip.EPI_BF16, out=out)
    for (m, n, k, mult), (A, Bm, out) in zip(shapes, ops):
        for _ in range(5):
            hip.gemm_nt(A, Bm, hip.EPI_BF16, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hip.gemm_nt(A, Bm, hip.EPI_BF16, out=out)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        f = 2.0 * m * n * k
        per.append({"M": m, "N": n, "K": k, "us": round(t * 1e6, 1), "TFLOPs": round(f / t / 1e12, 1)})
        tot_f += mult * f
        tot_t += mult * t
    return tot_f / tot_t / 1e12, per, tot_t


def time_attention_kernels(B, c, reps=20):
    """The north star's own figure: the fused attention kernels (forward, dQ, dK/dV: vt_attention_fwd / vt_attention_bwd) replayed at the
    step's shape through the C ABI, timed with HIP events on their stream.  Returns (us forward, us backward, algorithmic TFLOP/s over
    forward + backward) per layer: algorithmic FLOPs = 4 L^2 D per clip forward, twice that backward (12 L^2 D per layer)."""
    import video_tokenizer_amd.hip as hip
    D, H = 768, 12
    nv = (c["frame_num"] // c["temporal_patch_size"]) * (c["input_size"] // c["patch_size"]) ** 2
    L = nv + c["bottleneck_token_num"]
    qkv = torch.randn(B * L, 3 * D, device="cuda").to(torch.bfloat16)
    dO = torch.randn(B * L, D, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        o, lse = hip.attention_fwd(qkv, B, L, H)
        hip.attention_bwd(qkv, o, dO, lse, B, L, H)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(reps):
        o, lse = hip.attention_fwd(qkv, B, L, H)
    e[1].record()
    for _ in range(reps):
        hip.attention_bwd(qkv, o, dO, lse, B, L, H)
    e[2].record()
    torch.cuda.synchronize()
    tf, tb = e[0].elapsed_time(e[1]) / reps * 1e-3, e[1].elapsed_time(e[2]) / reps * 1e-3
    return tf * 1e6, tb * 1e6, 12.0 * L * L * D * B / (tf + tb) / 1e12


def time_vq_forward(B, c, reps=20):
    """The codebook search the north star names (`vt_vq_forward`: normalise, nearest code over the whole codebook on the exact-fp32 MFMA, gather,
    straight-through output, loss partials -- the N x K score matrix is never materialised) replayed at the step's shape through the C ABI and
    timed with HIP events on its stream, in the training default (mode 2: sampling) and the deterministic mode 0.  algorithmic flops = 2 N K d;
    algorithmic HBM bytes = z in + codebook once + indices + q out = 4 N d + 4 K d + 8 N + 4 N d."""
    import video_tokenizer_amd.hip as hip
    N, K, d = B * c["bottleneck_token_num"], c["codebook_size"], c["bottleneck_dim"]
    z = torch.randn(N, 64, device="cuda")
    cb = torch.randn(K, d, device="cuda")
    out = {"N": N, "K": K, "d": d, "algorithmic_bytes": 4 * N * d + 4 * K * d + 8 * N + 4 * N * d, "algorithmic_flops": 2 * N * K * d,
           "peak_fp32_matrix_TFLOPs": 157.3, "note": "whole forward quantizer call (prep + search + finalize kernels); compute-bound by construction: "
           "the GB/s figure is algorithmic bytes / time as the north star asks, not a bandwidth the kernel is limited by"}
    for mode, name in ((2, "sample"), (0, "argmin")):
        for _ in range(3):
            hip.vq_forward(z, cb, mode, inv_tau=1.0 / 0.03, seed=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            hip.vq_forward(z, cb, mode, inv_tau=1.0 / 0.03, seed=2 + i)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        out[name] = {"us": round(t * 1e6, 1), "TFLOPs_fp32": round(out["algorithmic_flops"] / t / 1e12, 1), "GBps_algorithmic": round(out["algorithmic_bytes"] / t / 1e9, 1),
                     "frac_of_fp32_matrix_peak": round(out["algorithmic_flops"] / t / 1e12 / 157.3, 3)}
    return out


MEASURED_MFMA_CEILING_TFLOPS = 1742.0    # bare 192x192 MFMA pattern with its LDS fragment reads and a barrier per K-tile, random data, all 256 CUs
MEASURED_MFMA_CEILING_SOURCE = "profiles/r04_mfma_probe_larger_tile.log"


def committed_profile(kind):
    """Figures that cannot be collected inside this process (rocprofv3 runs around it), read from the newest record under profiles/:
    kind 'mfma_busy' -> whole-step MFMA-pipe busy fraction (tools/pmc_step.sh); kind 'in_step_us' -> the dominant kernel's average
    duration inside the traced step (rocprofv3 --kernel-trace --stats of this script)."""
    import csv
    import glob
    for r in (5, 4, 3, 2, 1):
        if kind == "mfma_busy":
            # the newest record of a round sorts last (…_step.json < …_step_final_build.json < …_step_second_session.json)
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r0{r}_pmc_mfma_busy_step*.json")), reverse=True):
                with open(path) as f:
                    d = json.load(f)
                for k in ("whole_step_mfma_pipe_busy", "whole_step_mfma_busy_frac", "whole_step"):
                    if isinstance(d.get(k), (int, float)):
                        return {"value": round(float(d[k]), 4), "source": os.path.relpath(path, ROOT)}
        else:
            # a trace of the step alone (bench.py --no-roofline: no replays of the dominant kernel among its calls) first, then the round's traces of the default command, newest first
            names = [f"r0{r}_kernel_stats_step_only.csv", f"r0{r}_kernel_stats_second_session_final.csv", f"r0{r}_kernel_stats_final_build.csv", f"r0{r}_kernel_stats.csv"]
            for path in [q for q in (os.path.join(ROOT, "profiles", n) for n in names) if os.path.exists(q)]:
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if "gemm_nt192_kernel<0, 4>" in row["Name"]:
                            return {"value": round(float(row["AverageNs"]) / 1e3, 1), "calls": int(row["Calls"]), "source": os.path.relpath(path, ROOT)}
    return None


def torch_yardstick(batch):
    """The same step written with stock PyTorch-ROCm ops (nn.Linear / LayerNorm / SDPA / GELU under autocast bf16) on an MI355X of this
    pool: tools/torch_yardstick.py, measured by tools/torch_yardstick.sh next to this script and committed under profiles/.  Context
    for `value`, not a baseline in BASELINE.md's sense (the reference publishes no throughput): `vs_baseline` stays null."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_torch_yardstick.jsonl")), reverse=True):
        out = {}
        with open(path) as f:
            for line in f:
                try:
                    d = json.loads(line)
                except ValueError:
                    continue
                if d.get("clips_per_gpu") == batch and "clips_per_s" in d:
                    out.setdefault("torch_compile_clips_per_s" if d.get("compiled") else "eager_clips_per_s", d["clips_per_s"])
        if out:
            out["source"] = os.path.relpath(path, ROOT)
            return out
    return None


def _traffic_path():
    return next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_traffic_gemm_nt192.json") for r in (5, 4, 3, 2, 1)) if os.path.exists(q)), None)


def measured_traffic():
    """HBM-side bytes per launch of the dominant kernel from PMC counters (cannot be collected inside this process):
    the rocprofv3 passes of tools/pmc_traffic.sh, committed under profiles/ (`traffic_source` names the file)."""
    path = _traffic_path()
    if path is None:
        return None
    with open(path) as f:
        return int(json.load(f)["traffic_bytes_per_launch_mean"])


def traffic_source():
    path = _traffic_path()
    return None if path is None else os.path.relpath(path, ROOT) + " (committed rocprofv3 --pmc record, not collected in this run)"


def fsq_autoencoder_step(vt, name, steps, warmup, clips=4):
    """Secondary figure: forward + backward (L1 reconstruction loss) of an FSQ autoencoder of
    models/model_new/autoencoder.py at its hard-coded geometry (16x128x128 clips, 1024 + 1024 tokens).  No optimizer step, so the
    bf16 operand copies of the weights are re-used between steps; `tflops` counts the layers' matrix products only."""
    m = vt.make({"name": name, "args": {"bottleneck": None, "prior_model": None}}).cuda()
    video = torch.from_numpy(vt.config.synthetic_clips(clips, 16, 128, 7)).cuda()
    # matrix-product flops of every layer stack of the model (encoder, first-frame encoder, decoder): per layer and row
    # 2 * W * (4W + W + 3 * inner) for the four projections + 4 * L * W for the two attention products; fwd + bwd = 3x
    flops = 0
    for st in (getattr(m, "encoder", None), getattr(m, "encoder1", None), getattr(m, "decoder", None)):
        if st is None:
            continue
        W, layers = st.width, st.num_layers
        L = st.freqs[0].shape[0]
        inner = st.model_layers.ffd_layer[0][3].weight.shape[1]
        flops += 3 * layers * (2 * L * W * (4 * W + W + 3 * inner) + 4 * L * L * W)

    def step():
        for q in m.parameters():
            q.grad = None
        (m(video)["pred_frames"] - video).abs().mean().backward()

    for _ in range(max(warmup, 2)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"model": name, "parameters_M": round(sum(q.numel() for q in m.parameters()) / 1e6, 1), "clips_per_step": clips,
            "ms_per_step": round(dt * 1e3, 3), "clips_per_s": round(clips / dt, 2), "tflops": round(clips * flops / dt / 1e12, 1)}


def ar_prior_step(vt, size, steps, warmup, batch=8, seq=1024, gen_batch=16, vocab=8192, mode="both"):
    """Secondary figure: LARP_AR `llama-abs-<size>` over the tokenizer's indices (vocab 8192, 1024 tokens per clip:
    cfgs/larp_ar.yaml geometry) -- training step (forward, cross-entropy, backward; dropout at the reference's 0.1) and class-conditional
    generation through the KV cache (ar/generate.py: prefill + one token per position, with and without classifier-free guidance)."""
    m = vt.registry.make({"name": f"llama-abs-{size}", "args": dict(vocab_size=vocab, max_seq_len=seq, num_classes=101)}).cuda()
    torch.nn.init.normal_(m.output.weight, std=0.02)
    g = torch.Generator(device="cuda").manual_seed(5)
    tok = torch.randint(0, vocab, (batch, seq), device="cuda", generator=g)
    lab = torch.randint(0, 101, (batch,), device="cuda", generator=g)
    res = {"model": f"llama-abs-{size}", "parameters_M": round(sum(p.numel() for p in m.parameters()) / 1e6, 1)}
    if mode in ("both", "train"):
        m.train()

        def step():
            for p in m.parameters():
                p.grad = None
            loss = m(tok[:, :-1], lab, targets=tok)[1]
            loss.backward()
            return loss

        for _ in range(max(warmup, 2)):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        c = m.config
        hidden = m.layers[0].feed_forward.w1.weight.shape[0]
        per_tok = c.n_layer * (2 * c.dim * (4 * c.dim + 3 * hidden) + 2 * seq * c.dim) + 2 * c.dim * c.vocab_size      # forward flops per token, causal attention
        res["train"] = {"batch": batch, "seq": seq, "ms_per_step": round(dt * 1e3, 2), "tokens_per_s": round(batch * seq / dt),
                        "model_tflops": round(3 * per_tok * batch * seq / dt / 1e12, 1), "loss": round(loss.item(), 4)}
    if mode in ("both", "gen"):
        m.eval()
        cond = torch.randint(0, 101, (gen_batch,), device="cuda", generator=g)
        for scale in (1.0, 2.0):
            for timed in (False, True):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = m.sample(cond, cfg_scale=scale, temperature=1.0, top_k=0, top_p=1.0)
                m.reset_caches()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            res[f"generate_cfg{scale:g}"] = {"sequences": gen_batch, "new_tokens": int(out.shape[1]), "s": round(dt, 3), "tokens_per_s": round(gen_batch * out.shape[1] / dt),
                                             "ms_per_position": round(dt / out.shape[1] * 1e3, 3)}
    return res


def sq_step(vt, c, spec, x, steps, warmup):
    """Secondary figure: the same step with the bottleneck the shipped cfgs/larp_tokenizer.yaml:73 names, bottleneck_type 'sq' -- cosine search over
    the frozen 196 560 x 24 codebook (models/larp_tokenizer.py:225-229, 423-428); loss = L1 + 0.1 * loss_codebook."""
    import copy
    sp = copy.deepcopy(spec)
    sp["args"]["bottleneck_type"] = "sq"
    m = vt.make(sp)
    with torch.no_grad():
        torch.nn.init.xavier_uniform_(m.final_layer.linear.weight)
    m = m.to(x.device).train()

    def step():
        out = m(x)
        loss = (out["pred_frames"] - x).abs().mean() + 0.1 * out["loss_codebook"]
        for p in m.parameters():
            p.grad = None
        loss.backward()
        return loss

    for _ in range(max(warmup, 2)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"bottleneck_type": "sq", "codebook": "196560 x 24 (frozen, generated Leech shell)", "ms_per_step": round(dt * 1e3, 3), "clips_per_s": round(x.shape[0] / dt, 2),
            "loss": round(loss.item(), 5), "distinct_codes_in_batch": int(torch.unique(m.last_indices).numel())}


def gan_step(vt, model, x, steps, warmup):
    """Secondary figure: the trainer's step with its GAN branch (larp_tokenizer_trainer.py:263-345) --
    tokenizer forward, discriminator update on the detached reconstruction every d_update_freq-th step, generator loss
    (L1 + 0.3 * ns_g_loss through the frozen discriminator + 0.1 * loss_q), backward through discriminator and tokenizer.
    LPIPS is off (its VGG weights are not available offline), no optimizer steps: comparable to the headline step."""
    lm = vt.make({"name": "lpips_disc_loss", "args": dict(
        disc_type="transformer", disc_start=0, disc_self_start=-1, pixelloss_weight=1.0, perceptual_weight=0.0, pixel_loss="l1",
        lecam_weight=0.001, disc_loss="ns_smooth", disc_weight=0.3, r1_gp_weight=0.0, d_update_freq=5, spectral_norm=False,
        disc_tran_hidden_size=384, disc_tran_n_heads=12, disc_tran_n_layers=8, disc_tran_temporal_patch_size=4, disc_tran_patch_size=8,
        input_spatial_size=x.shape[-1], frame_num=x.shape[2])}).to(x.device)
    it = [0]

    def step():
        out = model(x)
        pred = out["pred_frames"]
        if it[0] % lm.d_update_freq == 0:
            lm.trainable_requires_grad_(True)
            d_loss, _, _ = lm(x, pred.detach(), global_step=it[0], for_discriminator=True)
            for q in lm.discriminator.parameters():
                q.grad = None
            d_loss.backward()
        lm.trainable_requires_grad_(False)
        loss, _, _ = lm(x, pred, global_step=it[0], for_discriminator=False)
        loss = loss + 0.1 * out["loss_q"]
        for q in model.parameters():
            q.grad = None
        loss.backward()
        it[0] += 1

    n = max(steps, 10) // 5 * 5          # whole d_update_freq periods
    for _ in range(max(warmup, 5)):
        step()
    torch.cuda.synchronize()
    it[0] = 0
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"ms_per_step": round(dt * 1e3, 3), "clips_per_s": round(x.shape[0] / dt, 2), "steps": n,
            "what": "tokenizer fwd+bwd + generator-side pass through the discriminator every step + discriminator update every 5th step; LPIPS off"}


def cpu_baseline(c, sd_seed=7):
    """The oracle (CPU restatement, fp32, reference semantics) timed on this host's cores: ONE clip of the
    same workload, forward + backward, stochastic=False index path (multinomial is not the cost)."""
    from oracle import inputs as gen   # the checker's own generator: same bytes as config.synthetic_clips
    from oracle import larp_oracle as O
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = min(ncpu, 16)  # the GPU box grants a 16-core share per GPU; more threads only oversubscribe it
    torch.set_num_threads(ncpu)
    sd = O.init_state_dict(c, seed=sd_seed)
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("_pe") and k != "decoder_patch_query_embed") for k, v in sd.items()}
    x = torch.from_numpy(gen.video_clips(1, c["frame_num"], c["input_size"], 3))
    times = []
    for i in range(4):      # one warm-up (thread pool, allocator, page faults), then the median of three
        for q in p.values():
            q.grad = None
        t0 = time.time()
        out = O.tokenizer_forward(p, c, x, "L")
        ((out["pred_frames"] - x).abs().mean() + 0.1 * out["loss_q"]).backward()
        if i > 0:
            times.append(time.time() - t0)
    dt = sorted(times)[1]
    return {"value": round(1.0 / dt, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 clip of the same workload, fwd+bwd, fp32 torch-CPU oracle/larp_oracle.py; median of 3 runs after one warm-up "
                      f"({', '.join(f'{t:.1f}' for t in times)} s); index mode L (stochastic: false, argmin of distances) -- the GPU leg samples (mode S), "
                      f"whose multinomial is not the cost on either side"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fsq-ae", default=None, metavar="NAME",
                    help="also time fwd+bwd of one of the FSQ autoencoders (autoencoder_large, autoencoder_convpatchify, "
                         "autoencoder_convpatchify_greatfsq; cfgs/larp_tokenizer_large.yaml:37) at the reference geometry, 4 clips: "
                         "extra key 'fsq_autoencoder_step'; the headline metric is unchanged")
    ap.add_argument("--sq", action="store_true",
                    help="also time the step with bottleneck_type 'sq' (the value the shipped cfgs/larp_tokenizer.yaml:73 carries): cosine search over the frozen "
                         "196 560 x 24 codebook; extra key 'sq_step'; the headline metric (bottleneck_type vq, BASELINE.json) is unchanged")
    ap.add_argument("--ar", default=None, metavar="SIZE",
                    help="also time the AR prior llama-abs-SIZE (S, B, L, LP, XL, XXL, XXXL; models/larp_ar.py:449-468) over 8192-code indices, "
                         "1024 tokens per clip: training step and KV-cache generation; extra key 'ar_prior'; the headline metric is unchanged")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="B")
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--optimizer", choices=["none", "fused", "torch"], default="none",
                    help="also step Adam(lr 1e-4, betas (0.5,0.9)) inside the timed step: 'fused' = vt_adam_step over flat buffers")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole forward + loss + backward as ONE captured hipGraph (engine.GraphedStep): the small-batch regime "
                         "(the reference recipe is 1 clip per GPU) is host-bound when enqueued launch by launch")
    ap.add_argument("--gan", action="store_true",
                    help="also time the step with the GAN branch of the trainer (discriminator of cfgs/larp_tokenizer.yaml:113-136, LPIPS off): "
                         "reported under the extra key 'gan_step'; the headline metric is unchanged")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal of the N > 1 code path on one GPU: process group, DataParallelTokenizer, barriers and the MAX all-reduce at world size 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launcher / rendezvous rehearsal only: every rank joins the process group (gloo when there is no GPU), "
                         "barrier + MAX all-reduce, rank 0 prints one JSON line; no model is built")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:     # bare `python bench.py --gpus N`: be our own launcher
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    if a.rehearse_launch:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        gpu = torch.cuda.device_count() > local
        dist.init_process_group("nccl" if gpu else "gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank)], dtype=torch.float64, device=torch.device("cuda", local) if gpu else "cpu")
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"rehearsal": True, "n_gpus": world, "backend": dist.get_backend(), "max_rank": int(t.item())}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or a.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL over xGMI

    import video_tokenizer_amd as vt
    from video_tokenizer_amd.parallel import DataParallelTokenizer

    c, spec = yaml_model_args(a.config)
    torch.manual_seed(1234 + rank)
    model = vt.make(spec)
    with torch.no_grad():  # the reference zero-inits the head (larp_tokenizer.py:327-328) => all-zero output and dead gradients
        torch.nn.init.xavier_uniform_(model.final_layer.linear.weight)
    model = model.to(dev).train()
    net = DataParallelTokenizer(model) if multi else model
    if a.optimizer == "torch":
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.9))
    elif a.optimizer == "fused":
        from video_tokenizer_amd.optim import FusedAdam
        opt = FusedAdam(model, lr=1e-4, betas=(0.5, 0.9))
    else:
        opt = None

    B = a.batch
    x = torch.from_numpy(vt.config.synthetic_clips(B, c["frame_num"], c["input_size"], 100 + rank)).to(dev)

    def step():
        out = net(x)
        loss = (out["pred_frames"] - x).abs().mean() + 0.1 * out["loss_q"]
        if opt is not None:
            opt.zero_grad(set_to_none=True)
        else:
            for p in model.parameters():
                p.grad = None
        loss.backward()
        if opt is not None:
            opt.step()
        return loss

    if a.graph:
        # the whole forward + loss + backward as ONE hipGraph replay (engine.GraphedStep); the optimizer step stays outside
        # (under the data-parallel wrapper the capture holds the reducer's collectives and stream forks as well; every rank captures here)
        from video_tokenizer_amd.engine import GraphedStep
        graphed = GraphedStep(net, x, lambda out, xin: (out["pred_frames"] - xin).abs().mean() + 0.1 * out["loss_q"])

        def step():  # noqa: F811
            loss, _ = graphed(x)
            if opt is not None:
                opt.step()
            return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(loss).item(), "loss is not finite"
    # host-side cost of ENQUEUEING one step, measured outside the timed region on an empty queue (a sync before each probe
    # step): inside the timed loop the host runs ahead until the HIP queue is full and then measures back-pressure instead
    host = []
    for _ in range(3):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        step()
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()

    if rank == 0:
        clips_s = world * B * a.steps / dt
        f_clip, f_attn = flops_per_clip(c)
        res = {
            "metric": "clips/sec (16x128x128) tokenizer fwd+bwd", "value": round(clips_s, 3), "unit": "clips/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"cfgs/larp_tokenizer.yaml base geometry as LARPTokenizer(bottleneck_type=vq): config {a.config}, "
                                   f"{c['frame_num']}x{c['input_size']}x{c['input_size']} clips, {B} clips/GPU, "
                                   f"{c['encoder_depth']}+{c['decoder_depth']} blocks, Nq={c['bottleneck_token_num']}, d={c['bottleneck_dim']}, K={c['codebook_size']}, "
                                   f"stochastic VQ (tau 0.03), loss = L1 + 0.1*loss_q" + (f", Adam ({a.optimizer})" if opt else ""),
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "host_enqueue_ms_per_step": round(sorted(host)[len(host) // 2] * 1e3, 3),
            "model_tflops_per_gpu": round(clips_s / world * f_clip / 1e12, 1),
            "attention_gemm_tflops_per_gpu": round(clips_s / world * f_attn / 1e12, 1),
        }
        # the secondary legs must never cost the headline line: a failure there is reported in place of the object
        if a.gan and world == 1:
            try:
                res["gan_step"] = gan_step(vt, model, x, a.steps, a.warmup)
            except Exception as e:  # noqa: BLE001
                res["gan_step"] = {"error": repr(e)}
        if a.fsq_ae and world == 1:
            try:
                res["fsq_autoencoder_step"] = fsq_autoencoder_step(vt, a.fsq_ae, a.steps, a.warmup)
            except Exception as e:  # noqa: BLE001
                res["fsq_autoencoder_step"] = {"error": repr(e)}
        if a.sq and world == 1:
            try:
                res["sq_step"] = sq_step(vt, c, spec, x, a.steps, a.warmup)
            except Exception as e:  # noqa: BLE001
                res["sq_step"] = {"error": repr(e)}
        if a.ar and world == 1:
            try:
                res["ar_prior"] = ar_prior_step(vt, a.ar, a.steps, a.warmup)
            except Exception as e:  # noqa: BLE001
                res["ar_prior"] = {"error": repr(e)}
        if not a.no_roofline:
            try:
                ach, per, _ = time_dominant_kernel(B, c)
                res["roofline"] = {"bound": "mfma", "kernel": "gemm_nt192_kernel<VT_EPI_BF16> (qkv fwd + fc1/proj/qkv dgrad)", "achieved": round(ach, 1),
                                   "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": measured_traffic(),
                                   "traffic_source": traffic_source(),
                                   # against what a bare MFMA loop of this tile pattern sustains on this chip under its own power management
                                   "frac_of_measured_mfma_ceiling": {"value": round(ach / MEASURED_MFMA_CEILING_TFLOPS, 4), "ceiling_TFLOPs": MEASURED_MFMA_CEILING_TFLOPS,
                                                                     "source": MEASURED_MFMA_CEILING_SOURCE},
                                   "per_shape": per, "whole_step_frac": round(clips_s / world * f_clip / 1e12 / PEAK_BF16_TFLOPS, 4)}
                # the north star's own figures next to it: the attention kernels against the same peak (live), the dominant kernel's
                # average INSIDE a traced step and the whole-step MFMA-pipe busy fraction (both from the committed rocprofv3 records)
                us_f, us_b, attn_tf = time_attention_kernels(B, c)
                res["roofline"]["attention_frac"] = round(attn_tf / PEAK_BF16_TFLOPS, 4)
                res["roofline"]["attention"] = {"achieved": round(attn_tf, 1), "unit": "TFLOP/s", "us_forward": round(us_f, 1), "us_backward": round(us_b, 1),
                                                "flops": "12 L^2 D per clip and layer (forward 4, backward 8; the two recompute kernels execute 7 products)"}
                res["roofline"]["vq_search"] = time_vq_forward(B, c)
                res["roofline"]["in_step_avg_us"] = committed_profile("in_step_us")
                res["roofline"]["mfma_busy_step"] = committed_profile("mfma_busy")
            except Exception as e:  # noqa: BLE001
                res["roofline"] = {"bound": "mfma", "achieved": None, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None, "error": repr(e)}
        if a.config == "B" and not a.gan and a.optimizer == "none":
            res["stock_pytorch_same_gpu"] = torch_yardstick(B)
        if not a.no_cpu_baseline and world == 1:  # the CPU leg runs at N = 1 only (other ranks would sit in teardown meanwhile)
            try:
                res["cpu_baseline"] = cpu_baseline(c)
            except Exception as e:  # noqa: BLE001
                res["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": None, "kind": "port", "sample": None, "error": repr(e)}
        print(json.dumps(res), flush=True)
    if multi:
        dist.barrier()  # rank 0 may still be replaying the dominant kernel for `roofline`: nobody tears the group down early
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
