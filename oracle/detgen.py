"""Deterministic, RNG-free tensor generators (test infrastructure).

Values come from a 32-bit integer hash of (element index, tag), so the reference side
(tests/golden/make_goldens.py) and the build side (tests) regenerate bit-identical
inputs and weights without shipping either and without depending on any RNG version.
"""
import zlib

import numpy as np
import torch


def _hash_u32(n, seed):
    """Integer avalanche hash of arange(n) xor seed -> uint32 array (exact integer math)."""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(seed & 0xFFFFFFFF) * np.uint64(40503)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x45D9F3B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x45D9F3B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def tag_seed(tag):
    return zlib.crc32(tag.encode()) & 0xFFFFFFFF


def uniform(shape, tag, lo=0.0, hi=1.0, dtype=torch.float32):
    """U[lo, hi) tensor of `shape`, a pure function of (shape, tag)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = _hash_u32(n, tag_seed(tag)).astype(np.float64) / 4294967296.0
    v = lo + (hi - lo) * u
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def bernoulli(shape, tag, p):
    n = int(np.prod(shape))
    u = _hash_u32(n, tag_seed(tag)).astype(np.float64) / 4294967296.0
    return torch.from_numpy((u < p).reshape(shape))


def fill_state_dict(sd, prefix="w"):
    """Overwrite every floating tensor of a state_dict in place with formula values.

    conv / linear weights: U(-b, b) with the xavier bound b = sqrt(6/(fan_in+fan_out));
    conv biases U(-0.05, 0.05); BN gamma U(0.8, 1.2), beta U(-0.1, 0.1);
    running_mean -> 0, running_var -> 1 (module defaults); integer buffers untouched.
    Keys of the (unused, 123.6 M parameter) VGG classifier are skipped.
    """
    for k, t in sd.items():
        if not torch.is_floating_point(t) or ".classifier." in k:
            continue
        tag = prefix + ":" + k
        if k.endswith("running_mean"):
            t.zero_()
        elif k.endswith("running_var"):
            t.fill_(1.0)
        elif t.dim() >= 2:
            rf = int(np.prod(t.shape[2:])) if t.dim() > 2 else 1
            bound = float(np.sqrt(6.0 / ((t.shape[0] + t.shape[1]) * rf)))
            t.copy_(uniform(tuple(t.shape), tag, -bound, bound))
        elif k.endswith("weight"):  # BN gamma (1-D weight)
            t.copy_(uniform(tuple(t.shape), tag, 0.8, 1.2))
        else:  # biases: BN beta or conv bias
            is_bn = (k[: -len("bias")] + "running_mean") in sd
            b = 0.1 if is_bn else 0.05
            t.copy_(uniform(tuple(t.shape), tag, -b, b))
    return sd


def image_batch(b, h, w, tag="img"):
    """U(0,1) image then the reference's default normalisation (x-0.5)/0.5 (train.py:131-132)."""
    return (uniform((b, 3, h, w), tag) - 0.5) / 0.5


def sparse_depth(b, h, w, tag="gt", density=0.05, lo=1.0, hi=80.0):
    """KITTI-like sparse GT: U(lo,hi) where a Bernoulli(density) mask is 1, else 0."""
    d = uniform((b, h, w), tag + ":v", lo, hi)
    m = bernoulli((b, h, w), tag + ":m", density)
    return d * m.to(d.dtype)


def summarize(t, stride=97):
    """Sparse samples + fp64 checksums of a tensor (for config-shape goldens)."""
    a = t.detach().to(torch.float64).reshape(-1).numpy()
    return {
        "shape": np.array(t.shape, dtype=np.int64),
        "samples": a[::stride].copy(),
        "sum": np.float64(a.sum()),
        "abssum": np.float64(np.abs(a).sum()),
        "max": np.float64(a.max()),
        "min": np.float64(a.min()),
    }
