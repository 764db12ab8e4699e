"""Data-parallel path with TWO ranks on one GPU (gloo moves the CUDA gradients; RCCL refuses two ranks on one device): the wrapper's
broadcast, the engine's stage-by-stage backward under the reducer, the data-parallel launch schedule (block-by-block tail, weight gradients
on their own stream) and the bucketed mean, end to end on real gradients.  GPU only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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
        torch.cuda.set_device(0)
        import video_tokenizer_amd as vt
        from oracle import inputs as gen
        from oracle import larp_oracle as O
        from tests.test_model_gpu import spec_from_cfg
        from video_tokenizer_amd.parallel import DataParallelTokenizer
        cfg = O.make_cfg("tiny", encoder_depth=6, decoder_depth=5)
        torch.manual_seed(100 + rank)                      # different init per rank: the wrapper must broadcast rank 0's weights
        m = vt.make(spec_from_cfg(cfg, stochastic=False)).cuda().train()
        with torch.no_grad():
            torch.nn.init.xavier_uniform_(m.final_layer.linear.weight)
            m.encoder_latent_query_embed.normal_(0.0, 1.0)  # spread queries: many different codes
        dp = DataParallelTokenizer(m, bucket_bytes=1 << 20)
        eng = m._engine
        assert eng.wgrad_stream is not None and eng.wgrad_tail == 3 and eng.reducer is not None
        xs = [torch.from_numpy(gen.video_clips(2, cfg["frame_num"], cfg["input_size"], 500 + r)).cuda() for r in range(world)]
        w = torch.from_numpy(gen.normal(tuple(xs[0].shape), 77)).cuda()

        def grads(x, net):
            for p in m.parameters():
                p.grad = None
            out = net(x)
            ((out["pred_frames"] * w).sum() + 0.7 * out["loss_q"]).backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

        got = None
        for _ in range(2):                                  # second pass: everything warm, side stream and buckets in steady state
            got = grads(xs[rank], dp)
        nb = len(eng.reducer.launched)
        # what it must equal: the mean of both ranks' local gradients, each computed here on one stream without a reducer
        red, side = eng.reducer, eng.wgrad_stream
        eng.reducer = None
        eng.set_wgrad_stream(None)
        eng.set_wgrad_tail(0)
        local = [grads(x, m) for x in xs]
        want = {k: (local[0][k] + local[1][k]) * 0.5 for k in local[0]}
        bad = [k for k in want if not torch.equal(got[k], want[k])]
        worst = max((float((got[k] - want[k]).abs().max() / (want[k].abs().max() + 1e-30)) for k in want), default=0.0)
        differ = sum(1 for k in want if not torch.equal(local[0][k], local[1][k]))
        eng.reducer = red
        q.put((rank, len(want), bad[:5], worst, nb, differ))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reduce_to_the_mean_of_their_gradients():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n, bad, worst, nb, differ in res:
        assert n > 60 and differ > 60, (n, differ)           # the two ranks really had different gradients
        assert nb >= 3, nb                                   # several buckets
        assert not bad, (rank, bad, worst)                   # (a + b) * 0.5 is exact: bit-equal to the locally computed mean
