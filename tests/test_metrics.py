"""EER / minDCF: our numpy restatement against golden values produced by the reference's own
utils.compute_eer / compute_mindcf (tests/golden/make_golden.py::metrics_golden)."""
import numpy as np

from tests.util import load_golden
from titanet_amd import metrics


def test_eer_mindcf_match_reference_goldens():
    g = load_golden("metrics")
    rng = np.random.default_rng(0)
    labels = rng.integers(0, 2, 1000)
    scores = labels * 0.3 + rng.normal(0, 0.3, 1000)
    assert abs(metrics.compute_eer(scores, labels) - float(g["eer"])) < 1e-9        # 0.2850971922
    assert abs(metrics.compute_mindcf(scores, labels, 0.01, 1, 1) - float(g["mindcf"])) < 1e-12   # 0.9663840205
    assert abs(metrics.compute_eer(g["scores2"], g["labels2"]) - float(g["eer2"])) < 1e-9
    assert abs(metrics.compute_mindcf(g["scores2"], g["labels2"], 0.01, 1, 1) - float(g["mindcf2"])) < 1e-12
    m = metrics.get_test_metrics(scores, labels, prefix="test")
    assert set(m) == {"test/eer", "test/mindcf"}


def test_eer_edge_cases():
    assert metrics.compute_eer([0.9, 0.8, 0.2, 0.1], [1, 1, 0, 0]) == 0.0          # perfectly separable
    assert abs(metrics.compute_eer([0.1, 0.2, 0.8, 0.9], [1, 1, 0, 0]) - 1.0) < 1e-12   # perfectly wrong
