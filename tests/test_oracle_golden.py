"""Pin the CPU oracle (oracle/titanet_oracle.py) against golden vectors produced by the real
reference (tests/golden/make_golden.py).  CPU-only; runs in the build container and on the GPU box."""
import numpy as np
import pytest
import torch

from oracle import titanet_oracle as O
from tests.golden.cases import CASES
from tests.util import LOSS_KW, case_inputs, case_state_dict, load_golden, oracle_cfg, rel_err

FAST = ["tiny_k3", "tiny_k7", "tiny_k11_short", "mid_k3", "tiny_simple_pool"]


@pytest.mark.parametrize("name", FAST + ["s17_b8"])
def test_eval_embeddings(name):
    case, g = CASES[name], load_golden(name)
    for dtype, tag, tol in ((torch.float64, "f64", 1e-10), (torch.float32, "f32", 2e-5)):
        sd = case_state_dict(case, None, dtype)
        x, _ = case_inputs(case, dtype)
        with torch.no_grad():
            out = O.titanet_forward(sd, x, oracle_cfg(case), training=False, keep_inter=(tag == "f64"))
        assert rel_err(out.normalized.numpy(), g[f"eval.{tag}.embeddings"]) < tol
        if tag == "f64" and case.get("inter"):
            for k, v in g.items():
                if not k.startswith("eval.f64.inter."):
                    continue
                key = k[len("eval.f64.inter."):]
                key = key.replace(".excitation.gate", ".excitation.gate")
                if key.endswith(".excitation.gate"):
                    got = out.inter[key][:, :]
                    assert rel_err(got.numpy(), v.reshape(got.shape)) < 1e-6, key
                elif key in out.inter:
                    assert rel_err(out.inter[key].numpy(), v) < 1e-6, key


@pytest.mark.parametrize("name", FAST + ["s17_b8"])
def test_train_forward_backward(name):
    case, g = CASES[name], load_golden(name)
    for loss in case["losses"]:
        sd = case_state_dict(case, loss, torch.float64)
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running_" not in k:
                v.requires_grad_(True)
        x, y = case_inputs(case)
        x.requires_grad_(True)
        kw = LOSS_KW[loss]
        out = O.titanet_forward(sd, x, oracle_cfg(case), training=True, speakers=y,
                                loss="ce" if loss == "ce" else "margin", loss_kwargs=kw)
        p = f"train.{loss}"
        assert abs(out.loss.item() - float(g[p + ".loss"])) < 1e-9 * max(1.0, abs(float(g[p + ".loss"])))
        assert rel_err(out.normalized.detach().numpy(), g[p + ".embeddings"]) < 1e-9
        assert np.array_equal(out.preds.numpy(), g[p + ".preds"])
        assert rel_err(out.logits.detach().numpy().clip(-1e30, 1e30), np.clip(g[p + ".logits"], -1, 1) if loss != "ce" else g[p + ".logits"]) < 1e-9
        if loss != "ce":
            assert rel_err(out.new_fc_weight.numpy(), g[p + ".fc_weight_after"]) < 1e-12
        out.loss.backward()
        n_checked = 0
        gscale = max([float(np.linalg.norm(v)) for k, v in g.items() if k.startswith(p + ".grad.")] + [1.0])
        for k, v in g.items():
            if not k.startswith(p + ".grad."):
                continue
            key = k[len(p + ".grad."):]
            got = x.grad if key == "input" else sd[key].grad
            assert got is not None, key
            # goldens are stored as float32
            # (a conv bias feeding a train-mode BN has an exactly-zero gradient: absolute floor)
            err = float(np.linalg.norm(got.numpy() - v))
            assert err < 2e-6 * float(np.linalg.norm(v)) + 1e-12 * gscale, (key, err)
            n_checked += 1
        assert n_checked > 0
        for k, v in g.items():
            if k.startswith(p + ".buffer."):
                key = k[len(p + ".buffer."):]
                assert rel_err(out.new_buffers[key].numpy(), v) < 1e-10, key


def test_state_dict_layout_matches_reference_listing():
    # 68 keys for N=1 + CE head; params 1.78 M (titanet.ipynb:961)
    cfg = O.OracleConfig.titanet("s", n_mega_blocks=1)
    shapes = O.state_dict_shapes(cfg, "ce", 251)
    assert len(shapes) == 68
    sizing = load_golden("sizing")
    n = sum(int(np.prod(s)) for k, s in shapes.items() if "running_" not in k and "num_batches" not in k)
    assert n == int(sizing["params.s1.ce251"]) == 1776379
    for size, nb in (("s", 17), ("s", 18), ("m", 10), ("l", 5)):
        shapes = O.state_dict_shapes(O.OracleConfig.titanet(size, n_mega_blocks=nb))
        n = sum(int(np.prod(s)) for k, s in shapes.items() if "running_" not in k and "num_batches" not in k)
        assert n == int(sizing[f"params.{size}{nb}"])


@pytest.mark.parametrize("name", ["arc", "cos"])
def test_margin_head_at_the_cos_clamp_vs_reference_golden(name):
    """a batch whose cosines reach exactly +-1 before the clamp (reference src/losses.py:94-98): loss,
    predictions, normalised inputs, in-place weight normalisation and both gradients against the real reference (f64)."""
    from tests.golden.cases import head_clamp_inputs
    g = load_golden("head_clamp")
    x, w, y = head_clamp_inputs()
    xo = torch.from_numpy(x).double().requires_grad_(True)
    wo = torch.from_numpy(w).double().requires_grad_(True)
    kw = O.margin_kwargs(name, scale=30 if name == "arc" else 64, margin=0.2)
    norm, preds, loss, cos, w_after = O.angular_margin_loss(xo, torch.from_numpy(y), {"loss_function.fc.weight": wo}, **kw)
    loss.backward()
    assert np.abs(g[name + ".raw_cos"]).max() >= 1.0 - 1e-12            # the fixture really sits on the clamp
    assert abs(loss.item() - float(g[name + ".loss"])) < 1e-10 * max(1.0, abs(float(g[name + ".loss"])))
    assert np.array_equal(preds.numpy(), g[name + ".preds"])
    assert rel_err(norm.detach().numpy(), g[name + ".normalized"]) < 1e-12
    assert rel_err(w_after.numpy(), g[name + ".weight_after"]) < 1e-12
    # the reference takes arccos of EVERY clamped cosine (src/losses.py:100): at +-1 its backward is 0 * inf = NaN for the
    # whole input row and the touched weight rows — the restatement reproduces exactly that pattern, finite entries to 1e-9
    for got, want in ((xo.grad.numpy(), g[name + ".grad.inputs"]), (wo.grad.numpy(), g[name + ".grad.weight"])):
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.isnan(want).any() and np.isfinite(want).any()
        ok = np.isfinite(want)
        assert rel_err(got[ok], want[ok]) < 1e-9


@pytest.mark.parametrize("name", ["tiny_k3", "mid_k3"])
def test_eager_module_graph_matches_reference_goldens(name):
    """oracle/eager_modules.py (what bench.py's cpu_baseline times: the reference's nn.Module graph rebuilt from its
    structure) loads the reference's state_dict keys and reproduces the reference's outputs."""
    from oracle.eager_modules import EagerTitaNet
    case, g = CASES[name], load_golden(name)
    c = case["cfg"]
    for loss in ("ce", "arc"):
        m = EagerTitaNet(n_mels=c["n_mels"], n_mega_blocks=c["n_mega_blocks"], hidden=c["hidden"], enc_out=c["enc_out"], emb=c["emb"],
                         kernel=c["kernel"], attn_hidden=c["attn_hidden"], dropout=0.0, loss=loss, n_classes=case["n_classes"]).double()
        sd = case_state_dict(case, loss, torch.float64)
        assert list(m.state_dict().keys()) == list(sd.keys())
        m.load_state_dict(sd)
        x, y = case_inputs(case, torch.float64)
        m.eval()
        with torch.no_grad():
            assert rel_err(m(x).numpy(), g["eval.f64.embeddings"]) < 1e-10
        m.train()
        emb, preds, lv = m(x, y)
        assert abs(lv.item() - float(g[f"train.{loss}.loss"])) < 1e-9 * max(1.0, abs(float(g[f"train.{loss}.loss"])))
        assert rel_err(emb.detach().numpy(), g[f"train.{loss}.embeddings"]) < 1e-9
        assert np.array_equal(preds.numpy(), g[f"train.{loss}.preds"])


def test_padding_mask_definition_matches_unpadded_utterances():
    """The padding-mask extension of the oracle (lengths=...) is DEFINED by: a zero-padded batch gives every utterance the
    result it has on its own.  Eval mode: embeddings of the padded batch == embeddings of each utterance run un-padded
    (through the reference-pinned, unmasked oracle).  Train mode: lengths == T reproduces the unmasked oracle exactly, and
    the masked statistics of a ragged batch equal the statistics of the concatenated valid frames."""
    case = CASES["tiny_k3"]
    cfg = oracle_cfg(case)
    sd = case_state_dict(case, "ce", torch.float64)
    x, y = case_inputs(case, torch.float64)
    B, T = x.shape[0], x.shape[2]
    lengths = torch.tensor([T, 9, 23, 30])[:B]
    xp = x.clone()
    for b in range(B):
        xp[b, :, lengths[b]:] = 0.0
    with torch.no_grad():
        padded = O.titanet_forward(sd, xp, cfg, training=False, lengths=lengths).normalized
        for b in range(B):
            alone = O.titanet_forward(sd, xp[b:b + 1, :, :lengths[b]], cfg, training=False).normalized
            assert rel_err(padded[b:b + 1].numpy(), alone.numpy()) < 1e-12, b
        full = O.titanet_forward(sd, x, cfg, training=True, speakers=y, loss="ce", lengths=torch.full((B,), T))
        ref = O.titanet_forward(sd, x, cfg, training=True, speakers=y, loss="ce")
        assert abs(full.loss.item() - ref.loss.item()) < 1e-12
        assert rel_err(full.normalized.numpy(), ref.normalized.numpy()) < 1e-12
        # masked train-mode statistics: garbage in the padded frames must not change anything
        xg = xp.clone()
        for b in range(B):
            xg[b, :, lengths[b]:] = 7.0
        a = O.titanet_forward(sd, xp, cfg, training=True, speakers=y, loss="ce", lengths=lengths)
        g = O.titanet_forward(sd, xg, cfg, training=True, speakers=y, loss="ce", lengths=lengths)
        assert abs(a.loss.item() - g.loss.item()) < 1e-12
        k = "encoder.prolog.conv_block.1.running_mean"
        assert rel_err(a.new_buffers[k].numpy(), g.new_buffers[k].numpy()) < 1e-12
