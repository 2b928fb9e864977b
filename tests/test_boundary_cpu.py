"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, and its parameter layout IS the reference state_dict (names, shapes, order)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import titanet_oracle as O
from tests.util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "titanet_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tn_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from titanet_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(_lib.EXPORTS)
    assert b"gfx950" in lib.tn_version()


@pytest.mark.parametrize("loss", [None, "ce", "arc"])
def test_layout_is_reference_state_dict(loss):
    from titanet_amd import LOSSES, TitaNet
    lf = None if loss is None else (LOSSES[loss](192, 251) if loss == "ce" else LOSSES[loss](192, 251, scale=30, margin=0.2))
    m = TitaNet.get_titanet(n_mega_blocks=2, model_size="s", loss_function=lf)
    sd = m.state_dict()
    want = O.state_dict_shapes(O.OracleConfig.titanet("s", n_mega_blocks=2), loss=("ce" if loss == "ce" else ("m" if loss else None)),
                               n_classes=251)
    assert list(sd.keys()) == list(want.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(want[k]), k
        assert v.dtype == (torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    # every tensor is a view into the flat buffers
    base = m.flat_parameters()
    for n, p in m.named_parameters():
        assert p.untyped_storage().data_ptr() == base.untyped_storage().data_ptr(), n


@pytest.mark.parametrize("name,kw,loss", [
    ("s2_none", dict(n_mega_blocks=2, model_size="s"), None),
    ("s2_ce251", dict(n_mega_blocks=2, model_size="s"), "ce"),
    ("s2_arc251", dict(n_mega_blocks=2, model_size="s"), "arc"),
    ("m1_none", dict(n_mega_blocks=1, model_size="m"), None),
    ("l1_none", dict(n_mega_blocks=1, model_size="l"), None),
    ("s1_simple_pool", dict(n_mega_blocks=1, model_size="s", simple_pool=True), None)])
def test_state_dict_keys_match_the_reference_listing(name, kw, loss):
    """key ORDER, shapes and dtypes against tests/golden/state_dict_keys.json, listed from the reference's own
    ``TitaNet.get_titanet(...).state_dict()`` by tests/golden/make_state_dict_keys.py (a checkpoint written by the reference
    loads key for key; torch.save keeps the order)"""
    import json
    import os
    from titanet_amd import LOSSES, TitaNet
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")))[name]
    lf = None if loss is None else (LOSSES["ce"](192, 251) if loss == "ce" else LOSSES["arc"](192, 251, scale=30, margin=0.2))
    sd = TitaNet.get_titanet(loss_function=lf, **kw).state_dict()
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    assert got == want


def test_simple_pool_layout_is_reference_state_dict():
    """Decoder(simple_pool=True), reference src/models.py:497-502: keys decoder.pool.2.{weight,bias}, no pool BatchNorm"""
    from titanet_amd import TitaNet
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s", simple_pool=True)
    sd = m.state_dict()
    want = O.state_dict_shapes(O.OracleConfig.titanet("s", n_mega_blocks=1, simple_pool=True))
    assert list(sd.keys()) == list(want.keys())
    assert tuple(sd["decoder.pool.2.weight"].shape) == (3072, 1536) and "decoder.pool.1.weight" not in sd


def test_param_counts_match_reference_known_answers():
    from titanet_amd import LOSSES, TitaNet
    sizing = load_golden("sizing")
    for size, n in (("s", 17), ("m", 10), ("l", 5)):
        m = TitaNet.get_titanet(n_mega_blocks=n, model_size=size)
        assert int(m.get_n_params()) == int(sizing[f"params.{size}{n}"])
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s", loss_function=LOSSES["ce"](192, 251))
    assert int(m.get_n_params()) == 1776379            # titanet.ipynb:961 ("1.78M")
    assert abs(m.get_n_params(div=1e6) - 1.78) < 0.01


def test_find_n_mega_blocks_known_answers():
    from titanet_amd import TitaNet
    # titanet.ipynb:743,765,787
    assert TitaNet.find_n_mega_blocks(192, 80, "s") == 18
    assert TitaNet.find_n_mega_blocks(192, 80, "m") == 10
    assert TitaNet.find_n_mega_blocks(192, 80, "l") == 5


def test_state_dict_roundtrip_and_default_init():
    from titanet_amd import LOSSES, TitaNet
    torch.manual_seed(0)
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s", loss_function=LOSSES["ce"](192, 10))
    sd = m.state_dict()
    assert float(sd["encoder.prolog.conv_block.1.weight"].min()) == 1.0
    assert float(sd["encoder.prolog.conv_block.1.running_var"].min()) == 1.0
    w = sd["encoder.mega_blocks.0.sub_blocks.0.conv_block.0.conv.1.weight"]
    assert abs(float(w.abs().max()) - 1 / 16) < 0.01          # kaiming_uniform(a=sqrt(5)): bound = 1/sqrt(fan_in)
    m2 = TitaNet.get_titanet(n_mega_blocks=1, model_size="s", loss_function=LOSSES["ce"](192, 10))
    m2.load_state_dict(sd)
    assert torch.equal(m2.flat_parameters(), m.flat_parameters())


def test_forward_without_gpu_fails_loudly():
    from titanet_amd import TitaNet
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s").eval()
    with pytest.raises(RuntimeError, match="no CPU execution path"):
        m(torch.zeros(2, 80, 50))


def test_reference_assertions():
    from titanet_amd import LOSSES, TitaNet
    with pytest.raises(AssertionError, match="Unsupported model size"):
        TitaNet.get_titanet(n_mega_blocks=1, model_size="x")
    with pytest.raises(AssertionError, match="Unsupported loss function"):
        TitaNet.get_titanet(n_mega_blocks=1, loss_function=torch.nn.CrossEntropyLoss())
    with pytest.raises(AssertionError, match="Margin out of bounds"):
        LOSSES["arc"](192, 10, margin=1.5)
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s")
    with pytest.raises(AssertionError, match="Loss function should not be None in training mode"):
        m(torch.zeros(2, 80, 50), speakers=torch.zeros(2, dtype=torch.int64))


def test_hot_path_kernels_do_not_spill():
    """tools/resource_usage.py --check: compiles the HIP sources with -Rpass-analysis=kernel-resource-usage (cross-compiles
    without a GPU) and fails on spilled VGPRs in the specialised / pipelined kernels beyond the documented exceptions"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "resource_usage.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "pgemm_nt_kernel" in r.stdout and "dgrad_dw_v6_kernel" in r.stdout


def test_no_reads_of_in_flight_inline_asm_load_registers():
    """tools/check_asm_hazards.py: in the generated gfx950 assembly no instruction reads a register that an inline-asm
    `global_load_dword` is still filling before the inline-asm `s_waitcnt vmcnt` that retires it (hipcc schedules copies
    there when it can: it believes the value exists once the asm statement has run)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_asm_hazards.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "tn_bwd.hip: 0 reads" in r.stdout


def test_store_data_hazard_checker_flags_the_pattern_it_was_written_for():
    """tools/check_asm_hazards.py check_store_data on listings of the two forms met in round 5 (se_combine_fwd_v3_kernel, DESIGN.md
    6): a 16-byte buffer store with an SGPR scalar offset followed by a VALU write of its first data register is flagged; the
    same store with scalar offset 0 (where hipcc inserts the hazard slots itself) and a write of an unrelated register are not."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(root, "tools", "check_asm_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = """_Z4kernv:
	v_cvt_pk_bf16_f32 v47, v62, v63
	buffer_store_dwordx4 v[44:47], v8, s[48:51], s36 offen
	v_or_b32_e32 v44, 0xc0, v154
	s_endpgm
"""
    good = """_Z4kernv:
	buffer_store_dwordx4 v[44:47], v60, s[48:51], 0 offen
	s_nop 1
	v_or_b32_e32 v44, 0xc0, v154
	buffer_store_dwordx4 v[20:23], v8, s[48:51], s36 offen
	v_or_b32_e32 v24, 0xc0, v154
	v_cmp_gt_u32_e64 s[2:3], s74, v20
	s_endpgm
"""
    f = mod.check_store_data(bad)
    assert len(f) == 1 and f[0][3] == [44], f
    assert mod.check_store_data(good) == []
