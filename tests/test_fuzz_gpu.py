"""A short run of the differential fuzz (tools/fuzz_paths.py): the specialised kernels against the generic templates
(TN_GENERIC=1) on random model sizes, shapes, heads, dropout rates, padding masks, gradient groups and fp8 plans.  The long runs
(hundreds of cases per seed) found the two bugs pinned in tests/test_v2_shapes_gpu.py."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_configurations_fast_vs_generic():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_paths", os.path.join(root, "tools", "fuzz_paths.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    failed = fz.main(n=20, seed=51)
    assert not failed, failed
