"""Data-parallel path on CPU: world_size 2, gloo.  The reducer is the same code that runs over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import video_tokenizer_amd as vt
        from oracle import larp_oracle as O
        from tests.test_model_gpu import spec_from_cfg
        from video_tokenizer_amd.engine import _flat_order
        from video_tokenizer_amd.parallel import DataParallelTokenizer, GradReducer
        torch.manual_seed(100 + rank)  # different init per rank: the wrapper must broadcast rank 0's weights
        m = vt.make(spec_from_cfg(O.make_cfg("tiny"), stochastic=True))
        dp = DataParallelTokenizer(m, bucket_bytes=4 << 20)
        # every element of every parameter and buffer (the trainable ones travel as ONE flat-buffer broadcast since round 4)
        chk = torch.cat([p.detach().double().reshape(-1)[:64] for p in list(m.parameters()) + list(m.buffers())] +
                        [torch.stack([p.detach().double().sum() for p in list(m.parameters()) + list(m.buffers())])])
        assert m._engine.flat_param is not None and all(p.data_ptr() >= m._engine.flat_param.data_ptr() for p in m.parameters() if p.requires_grad)
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        # feed the reducer the stage slices exactly as the engine reports them
        eng = m._engine
        eng.ensure_flat_grad(torch.device("cpu"))
        nst = max(eng.segments) + 1
        assert sorted(eng.segments) == list(range(nst))
        total = sum(p.numel() for p in m.parameters())
        assert eng.segments[0][0] == 0 and eng.segments[nst - 1][1] == total <= eng.flat_grad.numel() < total + 4
        assert all(eng.segments[s][1] == eng.segments[s + 1][0] for s in range(nst - 1))
        g = torch.Generator().manual_seed(5 + rank)
        local = torch.randn(eng.flat_grad.numel(), generator=g)
        eng.flat_grad.copy_(local)
        red = eng.reducer
        assert isinstance(red, GradReducer)
        for s in range(nst):
            red.segment_ready(eng.flat_grad, *eng.segments[s])
        red.finish()
        others = [torch.randn(eng.flat_grad.numel(), generator=torch.Generator().manual_seed(5 + r)) for r in range(world)]
        mean = sum(others) / world
        ok_mean = torch.allclose(eng.flat_grad[:total], mean[:total], atol=1e-6)
        covered = red.launched[0][0] == 0 and red.launched[-1][1] == total and \
            all(a[1] == b[0] for a, b in zip(red.launched, red.launched[1:]))
        n_full = len(red.launched)
        # the early release the engine switches on (red.total = elements to come): once less than a bucket is left, every reported slice goes
        # out at once -- more collectives than with full buckets only, the same tiling of the buffer, the same mean.  (A bucket of 12 M
        # elements, so that slices have to wait for each other at all: the tiny model's stages are larger than the 1 M-element bucket above.)
        counts = []
        for tot in (None, total):
            eng.flat_grad.copy_(local)
            red.bucket_elems, red.total = 12_000_000, tot
            for s in range(nst):
                red.segment_ready(eng.flat_grad, *eng.segments[s])
            red.finish()
            ok_mean = ok_mean and torch.allclose(eng.flat_grad[:total], mean[:total], atol=1e-6)
            covered = covered and red.launched[0][0] == 0 and red.launched[-1][1] == total and all(a[1] == b[0] for a, b in zip(red.launched, red.launched[1:]))
            counts.append(len(red.launched))
        covered = covered and counts[1] > counts[0]
        q.put((rank, same, ok_mean, covered, n_full, len(_flat_order(m))))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, ok_mean, covered, nb, nparams in res:
        assert same, "parameters were not broadcast from rank 0"
        assert ok_mean, "gradient mean over ranks is wrong"
        assert covered, "buckets do not tile the flat gradient buffer"
        assert nb >= 2 and nparams == 57


def test_bench_spawns_its_own_ranks_and_reaches_the_process_group():
    """`python bench.py --gpus 2` with no launcher environment (how the driver may call it): bench.py starts two rank
    processes itself (torch.distributed.run child, 127.0.0.1 rendezvous) before touching any GPU; with --rehearse-launch each
    rank joins the group (gloo here: no GPU), passes a barrier and a MAX all-reduce, and rank 0 prints ONE JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-launch"], capture_output=True, text=True,
                         timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r == {"rehearsal": True, "n_gpus": 2, "backend": "gloo", "max_rank": 1}
    # a failing rank must surface as a non-zero exit code of the parent
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-launch", "--config", "nope", "--bogus-flag"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert bad.returncode != 0
