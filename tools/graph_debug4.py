# RECORD of a round-3 investigation, as run against the FIRST GraphedStep (which returned the graph-resident loss tensor): see engine.GraphedStep.__init__
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    import subprocess
    for v in ["nothing", "sync_only", "equal_live", "equal_detached", "equal_unrelated", "eq_kept_detached", "item_live", "item_unrelated", "equal_live_nograd"]:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        print(v, (r.stdout.strip().splitlines() or ["?" + r.stderr[-200:]])[-1], flush=True)
    sys.exit(0)
V = sys.argv[1]
import torch
from oracle import inputs as gen
from oracle import larp_oracle as O
from tests.test_model_gpu import build
from video_tokenizer_amd.engine import GraphedStep
from video_tokenizer_amd.optim import FusedAdam
cfg = O.make_cfg("tiny", frame_num=8, input_size=64, bottleneck_token_num=128)
xs = [torch.from_numpy(gen.video_clips(2, cfg["frame_num"], cfg["input_size"], 300 + i)).cuda() for i in range(4)]
def loss_fn(out, x):
    return (out["pred_frames"] - x).abs().mean() + 0.1 * out["loss_q"]
def fresh():
    model, _ = build(cfg, stochastic=True)
    model.train()
    model.bottleneck.regularizer.set_stochastic_temperature(1.0)
    return model, FusedAdam(model, lr=1e-3, betas=(0.5, 0.9))
torch.manual_seed(1234)
model, opt = fresh()
model._engine.seed_counter = 100
X = []
for i in range(3):
    opt.zero_grad(set_to_none=True)
    out = model(xs[i]); loss = loss_fn(out, xs[i]); loss.backward()
    X.append(loss.detach().clone())
    opt.step()
torch.cuda.synchronize()
torch.manual_seed(1234)
model2, opt2 = fresh()
graphed = GraphedStep(model2, xs[3], loss_fn)
graphed.set_seed_counter(100)
keep = []
for i in range(3):
    l2, o2 = graphed(xs[i])
    if V != "nothing":
        torch.cuda.synchronize()
    if V == "equal_live":
        torch.equal(l2, X[i])
    elif V == "equal_live_nograd":
        with torch.no_grad():
            torch.equal(l2, X[i])
    elif V == "equal_detached":
        torch.equal(l2.detach(), X[i])
    elif V == "equal_unrelated":
        torch.equal(X[0], X[0])
    elif V == "eq_kept_detached":
        keep.append(l2.detach() == X[i])
    elif V == "item_live":
        l2.item()
    elif V == "item_unrelated":
        X[0].item()
    opt2.step()
torch.cuda.synchronize()
wd = max((a - b).abs().max().item() for a, b in zip(model.parameters(), model2.parameters()))
print(f"final weights max diff vs eager {wd:.3e}; last loss {l2.item():.6f} vs {X[2].item():.6f}")
