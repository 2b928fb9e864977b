"""TEST INFRASTRUCTURE ONLY — deterministic, framework-independent tensor generator.

Golden fixtures store only *outputs*; weights and inputs are rebuilt on any box from a
(name, shape) recipe so a 25 MB state dict never has to be committed.  The stream is
numpy's PCG64 seeded by crc32(name) ^ base_seed, i.e. it does not depend on torch's RNG.

Key names and shapes follow the reference ``state_dict`` (reference
``src/models.py:193-219``, ``src/modules.py:65-78``, ``src/losses.py:31,68``).
"""
import zlib

import numpy as np


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))


def tensor_for(name, shape, seed=0):
    """Value recipe by key suffix.  float64 ndarray (callers cast)."""
    g = _rng(name, seed)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return g.normal(0.0, 0.2, shape)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape)
    is_bn = len(shape) == 1 and (name.endswith("conv_block.1." + leaf) or name.endswith("skip_connection.1." + leaf)
                                 or name.endswith("pool.1." + leaf) or name.endswith("linear.1." + leaf))
    if is_bn and leaf == "weight":
        return g.uniform(0.5, 1.5, shape)
    if is_bn and leaf == "bias":
        return g.uniform(-0.3, 0.3, shape)
    if leaf == "bias":
        return g.uniform(-0.1, 0.1, shape)
    # conv / linear weights: uniform(-b, b), b = sqrt(3 / fan_in)  (unit-variance preserving)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    b = np.sqrt(3.0 / max(fan_in, 1))
    return g.uniform(-b, b, shape)


def fill_state_dict(shapes, seed=0):
    """shapes: ordered {key: shape}.  Returns {key: float64/int64 ndarray}."""
    return {k: tensor_for(k, s, seed) for k, s in shapes.items()}


def spectrograms(batch, n_mels, frames, seed=0):
    """Synthetic L2-normalised-dB-like mel batch [B, M, T] (values ~[-0.2, 0])."""
    g = _rng("input.spectrograms", seed)
    return (g.normal(0.0, 1.0, (batch, n_mels, frames)) * 0.11 - 0.10)


def speakers(batch, n_classes, seed=0):
    g = _rng("input.speakers", seed)
    return g.integers(0, n_classes, (batch,), dtype=np.int64)
