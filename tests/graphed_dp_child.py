"""Child process of tests/test_model_gpu.py::test_graphed_step_under_the_data_parallel_wrapper_single_rank_rccl.

engine.GraphedStep around a model wrapped in parallel.DataParallelTokenizer (RCCL, world size 1: the only size a 1-GPU box allows):
the capture then holds the stage-by-stage backward, the reducer's event / wait pairs, its all-reduces on the communication stream
and -- mode "side" -- the weight-gradient launches on their own stream.  Replays on new clips must equal EAGER WRAPPED steps bit for
bit (loss, sampled token ids, every gradient), with an optimizer step between replays.  A process of its own: a capture that fails
leaves streams in capture mode, which must not reach the other tests of the suite.

    python tests/graphed_dp_child.py side|noside       -> one JSON line, exit code 0 on success
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(mode):
    if mode == "noside":
        os.environ["VT_WGRAD_STREAM"] = "0"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    import video_tokenizer_amd as vt   # noqa: F401  (sets the graph switch before torch touches the GPU)
    import torch
    import torch.distributed as dist
    from oracle import inputs as gen
    from oracle import larp_oracle as O
    from video_tokenizer_amd.config import model_spec
    from video_tokenizer_amd.engine import GraphedStep
    from video_tokenizer_amd.optim import FusedAdam
    from video_tokenizer_amd.parallel import DataParallelTokenizer

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg = O.make_cfg("tiny", frame_num=8, input_size=64, bottleneck_token_num=128)
    xs = [torch.from_numpy(gen.video_clips(2, cfg["frame_num"], cfg["input_size"], 500 + i)).cuda() for i in range(4)]

    def loss_fn(out, x):
        rec = (out["pred_frames"] - x).abs().mean()
        return rec + (rec.detach() * 0.5 + 0.1) * out["loss_q"] + out["loss_commit"] * out["loss_codebook"]

    def fresh():
        model = vt.make(model_spec(cfg, True))
        model.load_state_dict(O.init_state_dict(cfg, seed=7, query_std=1.0), strict=True)
        model = model.cuda().train()
        model.bottleneck.regularizer.set_stochastic_temperature(1.0)
        dp = DataParallelTokenizer(model, bucket_bytes=1 << 20)          # several collectives even on the tiny model
        return model, dp, FusedAdam(model, lr=1e-3, betas=(0.5, 0.9))

    torch.manual_seed(1234)
    model, dp, opt = fresh()
    eng = model._engine
    assert eng.reducer is not None and (eng.wgrad_stream is not None) == (mode == "side")
    eng.seed_counter = 100
    eager = []
    for i in range(3):
        opt.zero_grad(set_to_none=True)
        out = dp(xs[i])
        loss = loss_fn(out, xs[i])
        loss.backward()
        eager.append((loss.detach().clone(), out["bottleneck_rep"].clone(), {n: p.grad.clone() for n, p in model.named_parameters()}))
        opt.step()
    torch.cuda.synchronize()
    n_coll = len(eng.reducer.launched)
    assert n_coll >= 3, n_coll

    torch.manual_seed(1234)
    model2, dp2, opt2 = fresh()
    graphed = GraphedStep(dp2, xs[3], loss_fn)             # self_check on: two eager + four replayed wrapped steps, bit for bit
    assert model2._engine.reducer is not None and len(model2._engine.reducer.launched) == n_coll
    graphed.set_seed_counter(100)
    bad = []
    for i in range(3):
        loss, out = graphed(xs[i])
        torch.cuda.synchronize()
        if not torch.equal(loss, eager[i][0]):
            bad.append(("loss", i, float(loss), float(eager[i][0])))
        if not torch.equal(out["bottleneck_rep"], eager[i][1]):
            bad.append(("bottleneck_rep", i))
        for n, p in model2.named_parameters():
            if p.grad is None or not torch.equal(p.grad, eager[i][2][n]):
                bad.append((n, i))
        opt2.step()
    torch.cuda.synchronize()
    for (n, a), (_, b) in zip(model.named_parameters(), model2.named_parameters()):
        if not torch.equal(a, b):
            bad.append(("weight", n))
    graphed.close()
    dist.destroy_process_group()
    print(json.dumps({"mode": mode, "collectives_per_step": n_coll, "differences": bad[:8], "ok": not bad}), flush=True)
    return 0 if not bad else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "side"))
