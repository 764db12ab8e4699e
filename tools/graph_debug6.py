# RECORD of a round-3 investigation, as run against the FIRST GraphedStep (which returned the graph-resident loss tensor): see engine.GraphedStep.__init__
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import inputs as gen
from oracle import larp_oracle as O
from tests.test_model_gpu import build
from video_tokenizer_amd.engine import GraphedStep
from video_tokenizer_amd.optim import FusedAdam
cfg = O.make_cfg("tiny", frame_num=8, input_size=64, bottleneck_token_num=128)
xs = [torch.from_numpy(gen.video_clips(2, cfg["frame_num"], cfg["input_size"], 300 + i)).cuda() for i in range(4)]
dbg = {k: torch.zeros((), device="cuda") for k in ("a", "b", "xsum", "psum")}
def loss_fn(out, x):
    a = (out["pred_frames"] - x).abs().mean()
    b = out["loss_q"]
    with torch.no_grad():
        dbg["a"].copy_(a); dbg["b"].copy_(b); dbg["xsum"].copy_(x.sum()); dbg["psum"].copy_(out["pred_frames"].sum())
    return a + 0.1 * b
def fresh():
    model, _ = build(cfg, stochastic=True)
    model.train()
    model.bottleneck.regularizer.set_stochastic_temperature(1.0)
    return model, FusedAdam(model, lr=1e-3, betas=(0.5, 0.9))
torch.manual_seed(1234)
model, opt = fresh()
model._engine.seed_counter = 100
E = []
for i in range(3):
    opt.zero_grad(set_to_none=True)
    out = model(xs[i]); loss = loss_fn(out, xs[i]); loss.backward()
    E.append((loss.item(), {k: v.item() for k, v in dbg.items()}))
    opt.step()
torch.cuda.synchronize()
torch.manual_seed(1234)
model2, opt2 = fresh()
graphed = GraphedStep(model2, xs[3], loss_fn)
graphed.set_seed_counter(100)
for i in range(3):
    l2, o2 = graphed(xs[i])
    torch.cuda.synchronize()
    d = {k: v.item() for k, v in dbg.items()}       # .item() reads: harmless
    lv = l2.item()
    torch.equal(l2, l2.clone())                      # a KERNEL reads the loss buffer: the trigger
    print(f"step {i}: graph loss {lv:.6f} parts {d}\n        eager loss {E[i][0]:.6f} parts {E[i][1]}")
    opt2.step()
