"""`from_pretrained` / `save_pretrained` for a LOCAL directory, in the file layout `huggingface_hub.PyTorchModelHubMixin` writes.

The reference's `LARPTokenizer` and `LARP_AR` inherit `PyTorchModelHubMixin` (/root/reference/models/larp_tokenizer.py:45,
models/larp_ar.py:233) and its callers load released weights with `cls.from_pretrained(<hub id or directory>)`
(eval/eval_larp_tokenizer.py:40, sample.py:409,415, trainers/larp_ar_trainer.py:51).  The mixin's directory layout is

    <dir>/config.json          the constructor's keyword arguments as JSON; an argument NAMED `config` that is a dataclass (LARP_AR's
                               `config: ModelArgs`) is the file itself: its fields sit at the top level, next to any other keyword
    <dir>/model.safetensors    the state dict

This module reads and writes exactly that, and ONLY on the local file system: there is no network on the machines this build runs on,
so a string that is not an existing directory raises instead of being resolved as a hub id (download the snapshot elsewhere and pass
its path).  safetensors files hold plain tensors: nothing is executed from them.
"""
import dataclasses
import functools
import inspect
import json
import os

import torch

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "model.safetensors"


def _jsonable(v):
    if dataclasses.is_dataclass(v) and not isinstance(v, type):
        return {k: _jsonable(x) for k, x in dataclasses.asdict(v).items()}
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (str, int, float, bool)) or v is None:
        return v
    if isinstance(v, torch.Tensor) and v.numel() == 1:
        return v.item()
    raise TypeError(f"constructor argument of type {type(v).__name__} cannot be written to {CONFIG_NAME}")


class LocalPretrainedMixin:
    """Adds `save_pretrained(dir)` and `from_pretrained(dir, **overrides)`; the constructor's arguments are recorded when it runs."""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        init = cls.__dict__.get("__init__")
        if init is None or getattr(init, "_records_hub_config", False):
            return
        sig = inspect.signature(init)

        @functools.wraps(init)
        def recording_init(self, *args, **kwargs):
            if not hasattr(self, "_hub_init_kwargs"):      # the outermost constructor call wins (subclasses calling super().__init__)
                bound = sig.bind(self, *args, **kwargs)
                rec = {}
                for name, val in list(bound.arguments.items())[1:]:
                    kind = sig.parameters[name].kind
                    if kind is inspect.Parameter.VAR_KEYWORD:
                        rec.update(val)
                    elif kind is not inspect.Parameter.VAR_POSITIONAL:
                        rec[name] = val
                object.__setattr__(self, "_hub_init_kwargs", rec)
            init(self, *args, **kwargs)

        recording_init._records_hub_config = True
        cls.__init__ = recording_init

    def save_pretrained(self, save_directory, config=None):
        """Write `config.json` + `model.safetensors` into `save_directory` (created if missing); returns the directory."""
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = config if config is not None else getattr(self, "_hub_init_kwargs", None)
        if cfg is None:
            raise RuntimeError("save_pretrained: the constructor arguments were not recorded; pass config=")
        cfg = dict(cfg)
        if dataclasses.is_dataclass(cfg.get("config")) and not isinstance(cfg.get("config"), type):
            top = _jsonable(cfg.pop("config"))          # the hub mixin's encoding of a dataclass `config` argument
            top.update(_jsonable(cfg))
            cfg = top
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            json.dump(_jsonable(cfg), f, indent=2, sort_keys=True)
        # safetensors refuses aliased storage: every entry becomes its own contiguous CPU tensor
        sd = {k: v.detach().to("cpu").contiguous().clone() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(save_directory, WEIGHTS_NAME), metadata={"format": "pt"})
        return save_directory

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *, strict=True, map_location="cpu", **model_kwargs):
        """Build the model from `<dir>/config.json` (keyword overrides in `model_kwargs`) and load `<dir>/model.safetensors` with
        `load_state_dict(strict=strict)`.  Only an existing local directory is accepted (no hub access on this platform)."""
        from safetensors.torch import load_file
        path = os.fspath(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"{cls.__name__}.from_pretrained('{path}'): not a local directory.  This build never contacts the Hugging Face hub; "
                f"download the snapshot ({CONFIG_NAME} + {WEIGHTS_NAME}) on a connected machine and pass its directory.")
        cfg_path, w_path = os.path.join(path, CONFIG_NAME), os.path.join(path, WEIGHTS_NAME)
        if not os.path.isfile(w_path):
            raise FileNotFoundError(f"{cls.__name__}.from_pretrained: {w_path} is missing")
        kwargs = {}
        if os.path.isfile(cfg_path):
            with open(cfg_path) as f:
                kwargs = json.load(f)
        kwargs.update(model_kwargs)
        # a dataclass-typed `config` parameter is rebuilt from the top-level keys (PyTorchModelHubMixin's encoding); any other
        # dataclass-typed parameter from the dict stored under its name
        sig = inspect.signature(cls.__init__)
        for name, prm in sig.parameters.items():
            ann = prm.annotation
            if not (isinstance(ann, type) and dataclasses.is_dataclass(ann)):
                continue
            known = {f.name for f in dataclasses.fields(ann)}
            if isinstance(kwargs.get(name), dict):
                kwargs[name] = ann(**{k: v for k, v in kwargs[name].items() if k in known})
            elif name == "config" and name not in kwargs:
                kwargs[name] = ann(**{k: kwargs.pop(k) for k in list(kwargs) if k in known})
        accepts_extra = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values())
        if not accepts_extra:
            kwargs = {k: v for k, v in kwargs.items() if k in sig.parameters}     # hub bookkeeping keys some snapshots carry
        model = cls(**kwargs)
        sd = load_file(w_path, device=str(map_location))
        model.load_state_dict(sd, strict=strict)
        model.eval()        # as PyTorchModelHubMixin._from_pretrained leaves it
        return model
