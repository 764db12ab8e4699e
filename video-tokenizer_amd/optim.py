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


def _composed_reason(model):
    """which constructor option sent the model to the composed path (LARPTokenizer records it), for error messages"""
    why = getattr(model, "_composed_why", None)
    return f" because of {why}" if why else ""


def flatten_parameters(model):
    """Move every trainable parameter of `model` into one flat fp32 buffer (same order as the flat gradient
    buffer).  Parameters keep their identity (only `.data` is re-pointed), so optimizers, state_dict and DDP
    wrappers created before or after keep working.  Idempotent."""
    eng = model._engine
    if eng is None:
        raise NotImplementedError("this model runs on the composed path (no fused engine, hence no flat parameter / gradient buffers)"
                                  + _composed_reason(model) + ": use torch.optim.Adam / torch DDP")
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
        if getattr(model, "_engine", None) is None:
            raise NotImplementedError("FusedAdam works on the fused engine's flat buffers; this model runs on the composed path"
                                      + _composed_reason(model) + ": use torch.optim.Adam")
        self.model = model
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay}]
        self.ema_decay = ema_decay
        self.step_count = 0          # the largest per-parameter step (torch.optim.Adam keeps one `step` per parameter)
        self.steps = {}              # name -> steps taken by that parameter: one frozen at first and unfrozen later starts at 1
        self.m = self.v = self.ema = None

    def _ensure(self):
        flat = flatten_parameters(self.model)
        if self.m is None or self.m.device != flat.device:
            self.m = torch.zeros_like(flat)
            self.v = torch.zeros_like(flat)
            if self.ema_decay is not None:
                self.ema = flat.clone()
        return flat

    def _active_runs(self, eng):
        """Contiguous [lo, hi) element ranges of the flat buffers that torch.optim.Adam would update: parameters with
        requires_grad and a gradient.  It skips the others entirely (no moment decay, no weight decay, no step) -- e.g. after
        decoder_requires_grad_(False) / others_requires_grad_(False).  Also makes sure the flat gradient slot holds what
        p.grad holds: autograd normally adopts the engine's view of the slot; if it made its own tensor, copy it back."""
        runs, off = [], 0
        for name, p, _ in _flat_order(self.model):
            n = p.numel()
            if p.requires_grad and p.grad is not None:
                view = eng.grad_views[name]
                if p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad.reshape(view.shape))
                k = self.steps.get(name, 0) + 1        # bias correction uses THIS parameter's step count, as torch.optim.Adam does
                self.steps[name] = k
                if runs and runs[-1][1] == off and runs[-1][2] == k:
                    runs[-1][1] = off + n
                else:
                    runs.append([off, off + n, k])
            off += n
        return runs, off

    @torch.no_grad()
    def step(self):
        flat = self._ensure()
        eng = self.model._engine
        if eng.flat_grad is None:
            raise hip.HipError("FusedAdam.step(): no gradients yet (run backward first)")
        g = eng.flat_grad
        assert g.numel() == flat.numel(), "flat gradient and parameter buffers must have the same (padded) length"
        grp = self.param_groups[0]
        runs, total = self._active_runs(eng)
        self.step_count = max([self.step_count] + [k for _, _, k in runs])
        if runs and runs[-1][1] == total:
            runs[-1][1] = flat.numel()          # the zero padding behind the last parameter rides along (n % 4 == 0)
        lr, (b1, b2), eps, wd = grp["lr"], grp["betas"], grp["eps"], grp["weight_decay"]
        covered = 0
        for lo, hi, k in runs:
            # every parameter of this model has a multiple of 4 elements, so runs start and end on 16-byte boundaries; a
            # model that breaks this gets the unaligned edge elements from the same formula in torch ops on the device
            a, b = (lo + 3) // 4 * 4, hi // 4 * 4
            for (x, y) in ((lo, min(a, hi)), (max(b, a), hi)):
                if y > x:
                    self._adam_slice_torch(flat, g, x, y, lr, b1, b2, eps, wd, k)
            if b > a:
                ema = self.ema[a:b] if self.ema is not None else None
                hip.check(hip.lib().vt_adam_step(hip.ptr(flat[a:b]), hip.ptr(g[a:b]), hip.ptr(self.m[a:b]), hip.ptr(self.v[a:b]), b - a, lr, b1, b2,
                                                 eps, wd, k, hip.ptr(ema), float(self.ema_decay or 0.0), hip.stream()), "vt_adam_step")
            covered += hi - lo
        if self.ema is not None and covered < flat.numel():
            # update_ema (base_trainer.py:769-779) runs over EVERY parameter, frozen ones included: their EMA keeps relaxing
            # towards the (unchanged) weight.  Frozen slices: ema = d * ema + (1 - d) * p, outside the fused launch.
            d, prev = float(self.ema_decay), 0
            for lo, hi, _ in runs + [[flat.numel(), flat.numel(), 0]]:
                if lo > prev:
                    self.ema[prev:lo].mul_(d).add_(flat[prev:lo], alpha=1.0 - d)
                prev = hi
        eng.param_epoch = getattr(eng, "param_epoch", 0) + 1  # the kernel wrote the weights behind torch's version counters:
        #                                                          tell the engine to re-pack its bf16 operand copies

    def _adam_slice_torch(self, flat, g, x, y, lr, b1, b2, eps, wd, k):
        gg = g[x:y] + wd * flat[x:y]
        self.m[x:y].mul_(b1).add_(gg, alpha=1.0 - b1)
        self.v[x:y].mul_(b2).addcmul_(gg, gg, value=1.0 - b2)
        bc1, bc2 = 1.0 - b1 ** k, 1.0 - b2 ** k
        flat[x:y].sub_((lr / bc1) * self.m[x:y] / (self.v[x:y].sqrt() / bc2 ** 0.5 + eps))
        if self.ema is not None:
            d = float(self.ema_decay)
            self.ema[x:y].mul_(d).add_(flat[x:y], alpha=1.0 - d)

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
            state[i] = {"step": torch.tensor(float(self.steps.get(n, 0))), "exp_avg": self.m[o:o + k].view(shp).clone() if self.m is not None else None,
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
            self.steps[n] = int(st["step"])
            self.step_count = max(self.step_count, self.steps[n])
        g = sd["param_groups"][0]
        self.param_groups[0].update({k: g[k] for k in ("lr", "betas", "eps", "weight_decay") if k in g})
