"""EER parity: the batched verification harness on the HIP path gives the same EER / minDCF as the CPU oracle
embeddings pushed through the same (reference-pinned) metric code."""
import itertools

import numpy as np
import pytest
import torch

from oracle import titanet_oracle as O
from tests.golden.cases import CASES
from tests.test_forward_gpu import build
from tests.util import case_state_dict, oracle_cfg
from titanet_amd import metrics

pytestmark = pytest.mark.gpu


def test_verification_eer_parity():
    case = CASES["tiny_k3"]
    m = build(case, None).eval()
    g = torch.Generator().manual_seed(3)
    n_spk, per_spk = 4, 3
    specs, spk = [], []
    for s in range(n_spk):
        base = torch.randn(case["cfg"]["n_mels"], 1, generator=g) * 0.1
        for u in range(per_spk):
            T = int(torch.randint(20, 60, (1,), generator=g))
            specs.append(base + torch.randn(case["cfg"]["n_mels"], T, generator=g) * 0.05 - 0.1)
            spk.append(s)
    got, scores, labels = metrics.verification_test(m, specs, spk)
    got5, scores5, _ = metrics.verification_test(m, specs, spk, batch_size=5)        # three padded batches instead of one
    assert np.abs(scores5 - scores).max() < 2e-5 and abs(got5["test/eer"] - got["test/eer"]) < 1e-6
    assert len(scores) == (n_spk * per_spk) ** 2                      # all ordered pairs incl. self pairs
    sd = case_state_dict(case, None, torch.float64)
    embs = []
    with torch.no_grad():
        for s in specs:
            embs.append(O.titanet_forward(sd, s.double().unsqueeze(0), oracle_cfg(case), training=False).normalized)
    e = torch.cat(embs).numpy()
    sim = e @ e.T
    pairs = list(itertools.product(range(len(specs)), repeat=2))
    o_scores = np.array([sim[i, j] for i, j in pairs])
    assert np.abs(scores - o_scores).max() < 1e-4
    want = metrics.get_test_metrics(o_scores, labels, prefix="test")
    assert abs(got["test/eer"] - want["test/eer"]) < 1e-6
    assert abs(got["test/mindcf"] - want["test/mindcf"]) < 1e-6


def test_verification_with_the_mean_pool_decoder_and_ragged_utterances():
    """Decoder(simple_pool=True) has no padding-mask forward: verification_test embeds one utterance per forward then (as the
    reference does) instead of raising on utterances of different lengths."""
    case = CASES["tiny_simple_pool"]
    m = build(case, None).eval()
    g = torch.Generator().manual_seed(5)
    specs = [torch.randn(case["cfg"]["n_mels"], int(torch.randint(20, 50, (1,), generator=g)), generator=g) * 0.05 - 0.1 for _ in range(6)]
    spk = [0, 0, 1, 1, 2, 2]
    got, scores, labels = metrics.verification_test(m, specs, spk, batch_size=4)
    assert len(scores) == 36 and np.isfinite(scores).all()
    with torch.no_grad():
        alone = torch.cat([m(s.unsqueeze(0).cuda()) for s in specs]).cpu().numpy()
    alone /= np.linalg.norm(alone, axis=1, keepdims=True)
    assert np.abs(scores.reshape(6, 6) - alone @ alone.T).max() < 1e-5


def test_verification_keeps_the_plan_cache_small():
    """utterances sorted by length, frames padded to multiples of 128: a sweep over many lengths creates a handful of plans"""
    case = CASES["tiny_k3"]
    m = build(case, None).eval()
    g = torch.Generator().manual_seed(6)
    specs = [torch.randn(case["cfg"]["n_mels"], 20 + 7 * i, generator=g) * 0.05 - 0.1 for i in range(24)]      # 20 .. 181 frames
    metrics.verification_test(m, specs, list(range(24)), batch_size=8)
    assert len(m._plans) <= 3, len(m._plans)
