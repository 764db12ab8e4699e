"""Config surface of the reference's launcher, without the launcher.

Mirrors /root/reference/train.py:55-138: a yaml file whose string values of the form `$name$` are
replaced by the command-line argument `name` (:66-75), then `--opts key.path value ...` overrides,
each value converted to the type of the value it replaces (bool from 'true'/'false', lists/tuples
from 'a_b_c', :105-124).  Returns plain nested dicts (attribute access via `AttrDict`), so
`make(cfg['model'])` builds the model exactly as trainers/base_trainer.py:353-393 does.
"""
import ast
import copy

import yaml


class AttrDict(dict):
    __setattr__ = dict.__setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    @staticmethod
    def wrap(x):
        if isinstance(x, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in x.items()})
        if isinstance(x, list):
            return [AttrDict.wrap(v) for v in x]
        return x


def _substitute(d, args):
    for k, v in d.items():
        if isinstance(v, dict):
            _substitute(v, args)
        elif isinstance(v, str) and v.startswith("$") and v.endswith("$"):
            name = v.replace("$", "")
            d[k] = args[name] if isinstance(args, dict) else getattr(args, name)


def _convert(typ, x):
    if typ is bool and isinstance(x, str):
        if x.lower() == "true":
            return True
        if x.lower() == "false":
            return False
        raise ValueError(f"Cannot convert {x} to bool")
    if typ in (list, tuple) and isinstance(x, str):
        return [ast.literal_eval(p) for p in x.split("_")]
    if typ is type(None):
        return x
    return typ(x)


def apply_opts(cfg, opts):
    assert len(opts) % 2 == 0, "--opts takes key value pairs"
    cfg = copy.deepcopy(cfg)
    for key, v in zip(opts[::2], opts[1::2]):
        keys = key.split(".")
        node = cfg
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = _convert(type(node[keys[-1]]), v)  # KeyError if the path does not exist, like the reference
    return cfg


def load_cfg(path_or_text, args=None, opts=()):
    if "\n" in path_or_text:
        cfg = yaml.safe_load(path_or_text)
    else:
        with open(path_or_text) as f:
            cfg = yaml.safe_load(f)
    _substitute(cfg, args or {})
    return AttrDict.wrap(apply_opts(cfg, list(opts)))
