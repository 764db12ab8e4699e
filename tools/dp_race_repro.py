"""Repro harness for an intermittent mismatch between the plain backward and the FIRST backward under DataParallelTokenizer (world 1, RCCL):
N times {fresh tiny model, plain step, wrap, first wrapped step, compare every gradient}, per setting.  VT_REPRO_BUCKET=<bytes>: the
reducer's bucket size (1024 = every stage's slice is reduced the moment it is final)."""
import gc
import os
import sys

import torch
import torch.distributed as dist

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from oracle import inputs as gen  # noqa: E402  (tools-only harness next to the tests: the checker's generator gives the tests' clips)
from oracle import larp_oracle as O  # noqa: E402
from tests.test_model_gpu import build  # noqa: E402
from video_tokenizer_amd.parallel import DataParallelTokenizer  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = O.make_cfg("tiny")
x = torch.from_numpy(gen.video_clips(2, cfg["frame_num"], cfg["input_size"], 31)).cuda()
w = torch.from_numpy(gen.normal(tuple(x.shape), 32)).cuda()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run(model, net):
    for p in model.parameters():
        p.grad = None
    out = net(x)
    ((out["pred_frames"] * w).sum() + 0.7 * out["loss_q"]).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in model.named_parameters()}


for label, env in (("default (stream, tail, early flush)", {}), ("VT_WGRAD_STREAM=0", {"VT_WGRAD_STREAM": "0"}), ("VT_WGRAD_TAIL=0", {"VT_WGRAD_TAIL": "0"}),
                   ("default again", {})):
    for k in ("VT_WGRAD_STREAM", "VT_WGRAD_TAIL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    fails = []
    for i in range(N):
        model, _ = build(cfg, seed=7 + i)
        plain = run(model, model)
        dp = DataParallelTokenizer(model, bucket_bytes=int(os.environ.get('VT_REPRO_BUCKET', 8 << 20)))
        got = run(model, dp)
        bad = {n: float((plain[n] - got[n]).abs().max()) for n in plain if not torch.equal(plain[n], got[n])}
        if bad:
            fails.append((i, len(bad), list(bad.items())[:4]))
        del model, dp, plain, got
        gc.collect()
    print(f"{label:40s}: {len(fails)} of {N} first wrapped steps differed", fails[:3], flush=True)
dist.destroy_process_group()
