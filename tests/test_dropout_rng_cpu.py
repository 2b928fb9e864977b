"""Statistical contract of the counter-based dropout generator (titanet_amd/csrc/tn_common.h: tn_drop8, restated bit for
bit by oracle/rng.py; the GPU tests check kernel masks == this restatement): keep rate, and INDEPENDENCE of the keep bits
inside a group of 8 elements (they share one mixing round), between neighbouring groups, between layers and between
consecutive steps — the reference draws i.i.d. Bernoulli masks (src/modules.py:133, src/models.py:470-472)."""
import numpy as np
import pytest

from oracle import rng

ROWS, C = 4096, 256


def _corr(a, b):
    a = a.astype(np.float64) - a.mean()
    b = b.astype(np.float64) - b.mean()
    return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_keep_rate_and_independence_inside_a_group_of_eight(p):
    m = rng.keep_mask_rows(seed=0x1234ABCD5678, layer=7, rows=ROWS, channels=C, p=p)
    n = m.size
    assert abs(m.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n)
    g = m.reshape(-1, 8)                                     # the 8 elements that share a mixing round
    sigma = 1 / np.sqrt(g.shape[0])
    worst = max(abs(_corr(g[:, i], g[:, j])) for i in range(8) for j in range(i + 1, 8))
    assert worst < 4.5 * sigma, (worst, sigma)               # 28 pairs: 4.5 sigma ~ 2e-4 false-alarm rate
    # per-position keep rate (no position of the group is biased)
    for i in range(8):
        assert abs(g[:, i].mean() - (1 - p)) < 4.5 * np.sqrt(p * (1 - p) / g.shape[0]), i
    # neighbouring groups (consecutive counters), same position
    for i in (0, 3, 7):
        assert abs(_corr(g[:-1, i], g[1:, i])) < 4.5 * sigma
    # along time for one channel (row stride = C / 8 counters) and along channels for one row
    assert abs(_corr(m[:-1, 5], m[1:, 5])) < 4.5 / np.sqrt(ROWS)
    assert abs(_corr(m[:, :-8].ravel(), m[:, 8:].ravel())) < 4.5 / np.sqrt(ROWS * (C - 8))


def test_independence_across_layers_and_steps():
    p, base = 0.1, 0xC0FFEE
    a = rng.keep_mask_rows(base, 3, ROWS, C, p).ravel()
    sigma = 1 / np.sqrt(a.size)
    for layer in (4, 5, 67):
        assert abs(_corr(a, rng.keep_mask_rows(base, layer, ROWS, C, p).ravel())) < 4.5 * sigma, layer
    # consecutive steps of the eager path: seed = base + step * 0x9E3779B97F4A7C15 (titanet_amd/models.py)
    for step in (1, 2, 3):
        s = (base + step * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        assert abs(_corr(a, rng.keep_mask_rows(s, 3, ROWS, C, p).ravel())) < 4.5 * sigma, step
    # graph replay: the per-step word is ADDED to the layer key (tn_act_key): key + mix32(step * golden + ...)
    k0 = int(rng.layer_key(base, 3))
    for step in (1, 2):
        word = int(rng.mix32(np.uint32((step * 0x9E3779B9 + 0x85EBCA6B) & 0xFFFFFFFF)))
        e = np.arange(ROWS * C, dtype=np.uint64)

        def mask_for(key):
            idx8 = (e >> np.uint64(3)).astype(np.uint32)
            with np.errstate(over="ignore"):
                x = idx8 + np.uint32(key & 0xFFFFFFFF)
                x ^= x >> np.uint32(16)
                x *= np.uint32(0x7FEB352D)
                x ^= x >> np.uint32(15)
                h = x * rng._DROP_C[((e >> np.uint64(1)) & np.uint64(3)).astype(np.int64)]
                h ^= h >> np.uint32(16)
            bits = np.where((e & np.uint64(1)) == 0, h & np.uint32(0xFFFF), h >> np.uint32(16))
            return bits >= np.uint32(rng.threshold(p))
        assert abs(_corr(mask_for(k0), mask_for(k0 + word))) < 4.5 * sigma, step
