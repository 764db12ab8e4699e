"""Fused Adam (+EMA) for the tokenizer: the optimizer step as ONE HBM-bound kernel over flat buffers.

The reference builds `torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.9))`
(cfgs/larp_tokenizer.yaml optimizer block; trainers/larp_tokenizer_trainer.py:160-212) and steps it through a
GradScaler (:376-377), then optionally `update_ema` (trainers/base_trainer.py:769-779).  Here the model's
parameters are re-pointed into one flat fp32 buffer laid out exactly like the engine's flat gradient buffer, so
the step is a single `vt_adam_step` launch (28 B/parameter) instead of ~300 foreach tensors.
State-dict compatible with torch.optim.Adam's per-parameter layout via `state_dict()`/`load_state_dict()`.
"""
import torch

from . import hip
from .engine import _flat_order


def flatten_parameters(model):
    """Move every trainable parameter of `model` into one flat fp32 buffer (same order as the flat gradient
    buffer).  Parameters keep their identity (only `.data` is re-pointed), so optimizers, state_dict and DDP
    wrappers created before or after keep working.  Idempotent."""
    eng = model._engine
    if getattr(eng, "flat_param", None) is not None and eng.flat_param.device == next(model.parameters()).device:
        return eng.flat_param
    order = _flat_order(model)
    dev = order[0][1].device
    total = sum(p.numel() for _, p, _ in order)
    pad = (-total) % 4
    flat = torch.zeros(total + pad, dtype=torch.float32, device=dev)
    off = 0
    for _, p, _ in order:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        off += n
    eng.flat_param = flat
    eng.flat_numel = total
    return flat


class FusedAdam:
    """Adam over the flattened tokenizer parameters.  API subset of torch.optim.Optimizer that the reference
    trainer uses: step(), zero_grad(set_to_none), state_dict(), load_state_dict(), param_groups[0]['lr']."""

    def __init__(self, model, lr=1e-4, betas=(0.5, 0.9), eps=1e-8, weight_decay=0.0, ema_decay=None):
        self.model = model
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay}]
        self.ema_decay = ema_decay
        self.step_count = 0
        self.m = self.v = self.ema = None

    def _ensure(self):
        flat = flatten_parameters(self.model)
        if self.m is None or self.m.device != flat.device:
            self.m = torch.zeros_like(flat)
            self.v = torch.zeros_like(flat)
            if self.ema_decay is not None:
                self.ema = flat.clone()
        return flat

    @torch.no_grad()
    def step(self):
        flat = self._ensure()
        eng = self.model._engine
        if eng.flat_grad is None:
            raise hip.HipError("FusedAdam.step(): no gradients yet (run backward first)")
        g = eng.flat_grad
        n = flat.numel()
        assert g.numel() == n, "flat gradient and parameter buffers must have the same (padded) length"
        grp = self.param_groups[0]
        self.step_count += 1
        hip.check(hip.lib().vt_adam_step(hip.ptr(flat), hip.ptr(g), hip.ptr(self.m), hip.ptr(self.v), n, grp["lr"], grp["betas"][0],
                                         grp["betas"][1], grp["eps"], grp["weight_decay"], self.step_count, hip.ptr(self.ema),
                                         float(self.ema_decay or 0.0), hip.stream()), "vt_adam_step")
        eng.param_epoch = getattr(eng, "param_epoch", 0) + 1  # the kernel wrote the weights behind torch's version counters:
        #                                                          tell the engine to re-pack its bf16 operand copies

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def ema_state_dict(self):
        """EMA weights in the model's state-dict layout (checkpoint key ema_sd[decay], base_trainer.py:783-891)."""
        assert self.ema is not None
        sd, off = {}, 0
        for name, p, _ in _flat_order(self.model):
            sd[name] = self.ema[off:off + p.numel()].view(p.shape).clone()
            off += p.numel()
        for k, b in self.model.named_buffers():
            sd[k] = b.clone()
        return sd

    def state_dict(self):
        order = _flat_order(self.model)
        names = [n for n, _ in self.model.named_parameters()]
        offs, off = {}, 0
        for name, p, _ in order:
            offs[name] = (off, p.numel(), p.shape)
            off += p.numel()
        state = {}
        for i, n in enumerate(names):
            o, k, shp = offs[n]
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[o:o + k].view(shp).clone() if self.m is not None else None,
                        "exp_avg_sq": self.v[o:o + k].view(shp).clone() if self.v is not None else None}
        grp = dict(self.param_groups[0])
        grp["params"] = list(range(len(names)))
        return {"state": state, "param_groups": [grp]}

    def load_state_dict(self, sd):
        self._ensure()
        names = [n for n, _ in self.model.named_parameters()]
        offs, off = {}, 0
        for name, p, _ in _flat_order(self.model):
            offs[name] = (off, p.numel())
            off += p.numel()
        for i, n in enumerate(names):
            st = sd["state"].get(i)
            if st is None:
                continue
            o, k = offs[n]
            self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])
        g = sd["param_groups"][0]
        self.param_groups[0].update({k: g[k] for k in ("lr", "betas", "eps", "weight_decay") if k in g})
