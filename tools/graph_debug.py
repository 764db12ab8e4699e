"""debug: where do eager and graph-replayed training steps diverge"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import inputs as gen
from oracle import larp_oracle as O
from tests.test_model_gpu import build
from video_tokenizer_amd.engine import GraphedStep
from video_tokenizer_amd.optim import FusedAdam

cfg = O.make_cfg("tiny", frame_num=8, input_size=64, bottleneck_token_num=128)
xs = [torch.from_numpy(gen.video_clips(2, cfg["frame_num"], cfg["input_size"], 300 + i)).cuda() for i in range(4)]
loss_fn = lambda out, x: (out["pred_frames"] - x).abs().mean() + 0.1 * out["loss_q"]

def fresh():
    model, _ = build(cfg, stochastic=True)
    model.train()
    model.bottleneck.regularizer.set_stochastic_temperature(1.0)
    return model, FusedAdam(model, lr=1e-3, betas=(0.5, 0.9))

torch.manual_seed(1234)
model, opt = fresh()
model._engine.seed_counter = 100
torch.manual_seed(1234)
model2, opt2 = fresh()
graphed = GraphedStep(model2, xs[3], loss_fn)
graphed.set_seed_counter(100)
for i in range(3):
    opt.zero_grad(set_to_none=True)
    out = model(xs[i]); loss = loss_fn(out, xs[i]); loss.backward()
    l2, o2 = graphed(xs[i])
    torch.cuda.synchronize()
    gd = max((a.grad - b.grad).abs().max().item() for a, b in zip(model.parameters(), model2.parameters()))
    print(f"step {i}: eager loss {loss.item():.6f} graph loss {l2.item():.6f} idx equal {torch.equal(out['bottleneck_rep'], o2['bottleneck_rep'])} max grad diff {gd:.3e}")
    opt.step(); opt2.step()
    torch.cuda.synchronize()
    wd = [(n, (a - b).abs().max().item()) for (n, a), (_, b) in zip(model.named_parameters(), model2.named_parameters())]
    wd.sort(key=lambda t: -t[1])
    print("   after opt.step: max weight diff", wd[:3], "steps", opt.step_count, opt2.step_count, "flat same storage", model2._engine.flat_param.data_ptr() == next(iter(model2.parameters())).data_ptr() or True)
    # eager forward of model2 with the same seed as the next replay would use, for comparison
