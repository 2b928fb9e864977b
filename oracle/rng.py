"""TEST INFRASTRUCTURE ONLY — numpy restatement of the counter-based dropout mask.

The reference draws dropout masks from torch's ``bernoulli_`` stream
(``src/modules.py:133``, ``src/models.py:470-472``); that stream cannot be matched by any
other implementation, so parity with dropout > 0 is checked against THIS restatement of
the HIP kernels' counter-based generator (``titanet_amd/csrc/tn_common.h: tn_drop8``)
while the statistical contract (keep rate 1-p, survivors scaled by 1/(1-p)) is what is
checked against the reference semantics.

Element index space: activations are stored rows x channels ("NTC": row = b*T + t), the
element index is e = row*C + c.  Each group of 8 consecutive elements shares one mixing round
x = f(e>>3 + key); element pair j of the group takes h = x*C_j ^ (x*C_j >> 16): low 16 bits -> even
element, high 16 bits -> odd element; keep iff bits >= round(p * 65536).
"""
import numpy as np

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)
_GOLD = np.uint32(0x9E3779B9)


def mix32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= _M1
        x ^= x >> np.uint32(15)
        x *= _M2
        x ^= x >> np.uint32(16)
    return x


def layer_key(seed, layer):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo = seed & 0xFFFFFFFF
    hi = seed >> 32
    inner = int(mix32(np.uint32((hi + int(layer) * 0x9E3779B9 + 1) & 0xFFFFFFFF)))
    return np.uint32(int(mix32(np.uint32(lo ^ inner))))


def threshold(p):
    return int(round(float(p) * 65536.0))


_DROP_C = np.array([0x846CA68B, 0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D], dtype=np.uint32)


def keep_mask_rows(seed, layer, rows, channels, p):
    """bool [rows, channels] keep mask in the kernels' row-major (NTC) element order
    (restates tn_common.h: tn_drop_shared / tn_drop_final)."""
    n = rows * channels
    e = np.arange(n, dtype=np.uint64)
    idx8 = (e >> np.uint64(3)).astype(np.uint32)
    with np.errstate(over="ignore"):
        x = idx8 + layer_key(seed, layer)
        x ^= x >> np.uint32(16)
        x *= _M1
        x ^= x >> np.uint32(15)
        c = _DROP_C[((e >> np.uint64(1)) & np.uint64(3)).astype(np.int64)]
        h = x * c
        h ^= h >> np.uint32(16)
    bits = np.where((e & np.uint64(1)) == 0, h & np.uint32(0xFFFF), h >> np.uint32(16))
    return (bits >= np.uint32(threshold(p))).reshape(rows, channels)


def keep_mask_bct(seed, layer, batch, channels, frames, p):
    """bool [B, C, T] keep mask (reference layout) for layer id `layer`."""
    m = keep_mask_rows(seed, layer, batch * frames, channels, p)
    return m.reshape(batch, frames, channels).transpose(0, 2, 1)
