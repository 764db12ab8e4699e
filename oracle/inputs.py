"""Counter-based deterministic input generator (TEST INFRASTRUCTURE).

Every parity test, golden-fixture script and the bench's cpu_baseline leg draws its
inputs from here, so the CPU oracle and the HIP path see identical bytes on any
machine and only *outputs* have to be committed under tests/golden/.

The generator is a pure function of (seed, flat index): splitmix64 finaliser.
Nothing here comes from the reference; it replaces `torch.randint`-style synthetic
data of the reference's fake dataset (datasets/video_dataset.py:315-316,344) with a
reproducible equivalent of the same distribution.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def hash_u64(n: int, seed: int, offset: int = 0) -> np.ndarray:
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = _splitmix64(np.array([seed], dtype=np.uint64))[0]
        return _splitmix64(idx ^ s)


def uniform(shape, seed: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """float32 uniform in [lo, hi): 24 random mantissa bits."""
    n = int(np.prod(shape))
    u = (hash_u64(n, seed) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(shape, seed: int, std: float = 1.0) -> np.ndarray:
    """float32 N(0, std^2) by Box-Muller in float64 on two hashed uniforms."""
    n = int(np.prod(shape))
    h1 = hash_u64(n, seed)
    h2 = hash_u64(n, seed ^ 0x5DEECE66D)
    u1 = ((h1 >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / (1 << 53))  # (0,1]
    u2 = (h2 >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (std * z).astype(np.float32).reshape(shape)


def video_clips(batch: int, frames: int, size: int, seed: int) -> np.ndarray:
    """(B,3,T,S,S) float32 in [0,1]: uniform uint8 frames /255, the distribution of the
    reference's fake dataset (video_dataset.py:315-316 randint(0,256) (T,H,W,3) uint8;
    :344 permute(-1,0,1,2).float()/255)."""
    n = batch * frames * size * size * 3
    u8 = (hash_u64(n, seed) >> np.uint64(56)).astype(np.uint8).reshape(batch, frames, size, size, 3)
    return np.ascontiguousarray(u8.transpose(0, 4, 1, 2, 3)).astype(np.float32) / np.float32(255.0)


def xavier_uniform(shape, seed: int) -> np.ndarray:
    """xavier-uniform bound for a (out, in...) weight viewed 2-D, like
    larp_tokenizer.py:251-256,322-323 (values are hash-generated, not torch RNG)."""
    fan_out = shape[0]
    fan_in = int(np.prod(shape[1:]))
    a = float(np.sqrt(6.0 / (fan_in + fan_out)))
    return uniform(shape, seed, -a, a)


def kaiming_uniform_codebook(k: int, d: int, seed: int) -> np.ndarray:
    """nn.init.kaiming_uniform_ default (a=0 => gain sqrt(2), bound sqrt(6/fan_in)),
    bottleneck.py:237-238."""
    a = float(np.sqrt(6.0 / d))
    return uniform((k, d), seed, -a, a)
