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


# --------------------------------------------------------------------------------------------------------------------
# Geometry points of the benchmark / parity configs (BASELINE.json `configs`), the registry spec that the
# cfgs/larp_tokenizer.yaml surface produces with `--opts model.name larp_tokenizer model.args.bottleneck_type vq`, and the
# synthetic clip generator (the distribution of the reference's built-in fake dataset).  Product-side: bench.py and the
# tests build their models and inputs from here.
# --------------------------------------------------------------------------------------------------------------------
GEOMETRIES = {
    # BASELINE configs[0]: yaml base geometry on 2x64x64 clips (plumbing case)
    "A": dict(frame_num=2, input_size=64, temporal_patch_size=2, patch_size=16, encoder_depth=12, decoder_depth=12, bottleneck_token_num=1024, bottleneck_dim=24),
    # configs[1], the headline: yaml base geometry on 16x128x128 clips; Bp = the upstream LARP patching (pt 4, p 8)
    "B": dict(frame_num=16, input_size=128, temporal_patch_size=2, patch_size=16, encoder_depth=12, decoder_depth=12, bottleneck_token_num=1024, bottleneck_dim=24),
    "Bp": dict(frame_num=16, input_size=128, temporal_patch_size=4, patch_size=8, encoder_depth=12, decoder_depth=12, bottleneck_token_num=1024, bottleneck_dim=24),
    # configs[2], [3]: f256t512 / f256t1024 yaml values; configs[4]: large yaml at 16x256x256
    "C": dict(frame_num=16, input_size=128, temporal_patch_size=4, patch_size=8, encoder_depth=6, decoder_depth=6, bottleneck_token_num=512, bottleneck_dim=16),
    "D": dict(frame_num=16, input_size=128, temporal_patch_size=4, patch_size=8, encoder_depth=6, decoder_depth=6, bottleneck_token_num=1024, bottleneck_dim=16),
    "E": dict(frame_num=16, input_size=256, temporal_patch_size=4, patch_size=8, encoder_depth=6, decoder_depth=6, bottleneck_token_num=1024, bottleneck_dim=16),
    # small case for parity tests the CPU oracle finishes in seconds
    "tiny": dict(frame_num=4, input_size=32, temporal_patch_size=2, patch_size=16, encoder_depth=2, decoder_depth=2, bottleneck_token_num=56, bottleneck_dim=24, codebook_size=512),
}


def geometry(name, **over):
    c = dict(hidden=768, encoder_num_heads=12, decoder_num_heads=12, codebook_size=8192, latent_pe_scale_factor=10000)
    c.update(GEOMETRIES[name])
    c.update(over)
    c["token_h"] = c["input_size"] // c["patch_size"]
    return c


def model_spec(cfg, stochastic=False):
    """registry spec {'name', 'args'} of LARPTokenizer(bottleneck_type='vq') at a geometry (cfgs/larp_tokenizer.yaml:37-78 keys)"""
    return {"name": "larp_tokenizer", "args": {
        "bottleneck": {"name": "bottleneck", "args": {"bottleneck_dim": cfg["bottleneck_dim"], "norm": "none", "regularizer": {
            "name": "vq", "args": {"codebook_size": cfg["codebook_size"], "commitment_loss_weight": 0.25, "codebook_loss_weight": 1.0,
                                   "entropy_loss_weight": 0.0, "entropy_loss_temperature": 0.01, "l2_normalized": True,
                                   "stochastic": stochastic, "stochastic_temperature": 0.03}}}},
        "prior_model": {"name": "none"}, "bottleneck_token_num": cfg["bottleneck_token_num"], "input_size": cfg["input_size"],
        "frame_num": cfg["frame_num"], "temporal_patch_size": cfg["temporal_patch_size"], "patch_size": cfg["patch_size"],
        "decoder_temporal_patch_size": cfg["temporal_patch_size"], "decoder_patch_size": cfg["patch_size"], "in_channels": 3,
        "bottleneck_type": "vq", "transformer_name": "transformer_encoder_parallel", "encoder_name": "none", "decoder_name": "none",
        "encoder_hidden_size": 768, "decoder_hidden_size": 768, "encoder_num_heads": 12, "decoder_num_heads": 12,
        "encoder_depth": cfg["encoder_depth"], "decoder_depth": cfg["decoder_depth"],
        "use_decoder_patch_query_token_type_embed": True, "use_pe": "yes"}}


def synthetic_clips(batch, frames, size, seed):
    """(B, 3, T, S, S) float32 in [0, 1]: uniform uint8 frames / 255 -- the distribution of the reference's fake dataset
    (datasets/video_dataset.py:315-316 randint(0, 256) of (T, H, W, 3) uint8; :344 permute(-1, 0, 1, 2).float() / 255).
    Counter-based (splitmix64 of seed and flat index), so any process regenerates the same bytes."""
    import numpy as np
    m64 = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mix(x):
        with np.errstate(over="ignore"):
            x = (x + np.uint64(0x9E3779B97F4A7C15)) & m64
            z = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & m64
            z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & m64
            return z ^ (z >> np.uint64(31))
    n = batch * frames * size * size * 3
    s = mix(np.array([seed], dtype=np.uint64))[0]
    u8 = (mix(np.arange(n, dtype=np.uint64) ^ s) >> np.uint64(56)).astype(np.uint8).reshape(batch, frames, size, size, 3)
    return np.ascontiguousarray(u8.transpose(0, 4, 1, 2, 3)).astype(np.float32) / np.float32(255.0)
