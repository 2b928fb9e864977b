"""GPU parity of the HIP backward path (loss.backward() through the C ABI) against float64
gradients of the real reference (golden fixtures) and of the CPU oracle.

Gradient tolerance: SURVEY.md §0.4 — train-mode BatchNorm over small batches makes parameter
gradients ill-conditioned (the reference's own fp32-vs-fp64 gradients differ by 1.2e-2 at B=8), so
the fp32 HIP path is held to 3e-2 relative per tensor at B<=8 against the float64 reference and to
a much tighter whole-gradient cosine."""
import numpy as np
import pytest
import torch

from oracle import titanet_oracle as O
from tests.golden.cases import CASES
from tests.test_forward_gpu import build
from tests.util import LOSS_KW, case_inputs, case_state_dict, load_golden, mask_fn_for, oracle_cfg, rel_err

pytestmark = pytest.mark.gpu


def _check_grads(model, g, prefix, tol, floor_rel=2e-4, x_grad=None):
    grads = {k: v.grad for k, v in model.named_parameters()}
    gscale = max(float(np.linalg.norm(v)) for k, v in g.items() if k.startswith(prefix + ".grad."))
    n = 0
    worst = (0.0, None)
    for k, want in g.items():
        if not k.startswith(prefix + ".grad."):
            continue
        key = k[len(prefix + ".grad."):]
        got = x_grad if key == "input" else grads[key]
        if got is None:
            assert key == "input", key
            continue
        got = got.detach().cpu().numpy()
        err = float(np.linalg.norm(got - want))
        ok = err <= tol * float(np.linalg.norm(want)) + floor_rel * tol * gscale
        rel = err / max(float(np.linalg.norm(want)), 1e-30)
        if rel > worst[0] and float(np.linalg.norm(want)) > 1e-3 * gscale:
            worst = (rel, key)
        assert ok, (key, err, float(np.linalg.norm(want)), gscale)
        n += 1
    assert n > 0
    return worst


@pytest.mark.parametrize("name", ["tiny_k3", "tiny_k7", "tiny_k11_short", "mid_k3", "tiny_simple_pool", "s17_b8"])
def test_backward_fp32_vs_reference_golden(name):
    case, g = CASES[name], load_golden(name)
    for loss in case["losses"]:
        m = build(case, loss).train()
        x, y = case_inputs(case, torch.float32)
        xin = x.cuda().requires_grad_(case.get("grads") == "all")
        emb, preds, lv = m(xin, speakers=y.cuda())
        lv.backward()
        torch.cuda.synchronize()
        worst = _check_grads(m, g, f"train.{loss}", tol=3e-2, x_grad=xin.grad)
        print(name, loss, "worst significant grad rel err", worst)
        # whole-gradient direction vs the float64 reference
        keys = [k[len(f"train.{loss}.grad."):] for k in g if k.startswith(f"train.{loss}.grad.") and not k.endswith(".input")]
        named = dict(m.named_parameters())
        a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in keys])
        b = np.concatenate([g[f"train.{loss}.grad.{k}"].ravel() for k in keys])
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        assert cos > 0.9995, cos


@pytest.mark.parametrize("name,p", [("tiny_k3", 0.25), ("mid_k3", 0.1), ("tiny_simple_pool", 0.2)])
def test_backward_with_dropout_vs_oracle(name, p):
    case = CASES[name]
    m = build(case, "ce", dropout=p).train()
    m._seed_base, m._step = 424242, 0
    x, y = case_inputs(case, torch.float32)
    emb, preds, lv = m(x.cuda(), speakers=y.cuda())
    lv.backward()
    sd = case_state_dict(case, "ce", torch.float64)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
    xo, yo = case_inputs(case, torch.float64)
    out = O.titanet_forward(sd, xo, oracle_cfg(case, dropout=p), training=True, speakers=yo, loss="ce",
                            mask_fn=mask_fn_for(424242, p))
    out.loss.backward()
    assert abs(lv.item() - out.loss.item()) < 1e-3 * max(1.0, abs(out.loss.item()))
    named = dict(m.named_parameters())
    a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in named])
    b = np.concatenate([sd[k].grad.numpy().ravel() for k in named])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    assert cos > 0.9995, cos
    assert rel_err(a, b) < 3e-2, rel_err(a, b)


@pytest.mark.parametrize("name", ["tiny_k3", "mid_k3", "tiny_simple_pool", "s17_b8"])
def test_backward_bf16_direction(name):
    """bf16 is the throughput mode: gradients must point the same way as the float64 reference."""
    case, g = CASES[name], load_golden(name)
    m = build(case, "ce", precision="bf16").train()
    x, y = case_inputs(case, torch.float32)
    emb, preds, lv = m(x.cuda(), speakers=y.cuda())
    lv.backward()
    assert abs(lv.item() - float(g["train.ce.loss"])) < 0.1 * max(1.0, abs(float(g["train.ce.loss"])))
    keys = [k[len("train.ce.grad."):] for k in g if k.startswith("train.ce.grad.") and not k.endswith(".input")]
    named = dict(m.named_parameters())
    a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in keys])
    b = np.concatenate([g[f"train.ce.grad.{k}"].ravel() for k in keys])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    # B=8 train-mode BatchNorm (decoder BN over 8 samples) is ill-conditioned: bf16 noise is amplified
    assert cos > (0.7 if case["batch"] <= 8 and case["cfg"]["n_mega_blocks"] > 4 else 0.97), cos


def test_backward_bf16_vs_oracle_larger_batch():
    """bf16 gradients at a batch where BatchNorm is well conditioned: S-width, 3 mega blocks, B=48."""
    case = dict(cfg=dict(n_mels=80, n_mega_blocks=3, hidden=256, enc_out=512, emb=64, kernel=3, attn_hidden=64),
                batch=48, frames=120, n_classes=40, seed=7)
    m = build(case, "ce", precision="bf16").train()
    x, y = case_inputs(case, torch.float32)
    emb, preds, lv = m(x.cuda(), speakers=y.cuda())
    lv.backward()
    sd = case_state_dict(case, "ce", torch.float64)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
    xo, yo = case_inputs(case, torch.float64)
    out = O.titanet_forward(sd, xo, oracle_cfg(case), training=True, speakers=yo, loss="ce")
    out.loss.backward()
    assert abs(lv.item() - out.loss.item()) < 0.05 * max(1.0, abs(out.loss.item()))
    named = dict(m.named_parameters())
    a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in named])
    b = np.concatenate([sd[k].grad.numpy().ravel() for k in named])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    print("bf16 B=48 whole-gradient cosine", cos, "rel", rel_err(a, b))
    assert cos > 0.98, cos
    # fp32 path on the same case: tight
    m32 = build(case, "ce", precision="fp32").train()
    m32(x.cuda(), speakers=y.cuda())[2].backward()
    named = dict(m32.named_parameters())
    a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in named])
    assert rel_err(a, b) < 2e-3, rel_err(a, b)


def test_grad_accumulation_and_zero_grad_semantics():
    case = CASES["tiny_k3"]
    m = build(case, "ce").train()
    x, y = case_inputs(case, torch.float32)
    xc, yc = x.cuda(), y.cuda()
    m(xc, speakers=yc)[2].backward()
    g1 = m.flat_gradients().clone()
    p = next(m.parameters())
    ga = p.grad.clone()
    m(xc, speakers=yc)[2].backward()        # second backward accumulates (torch semantics)
    assert rel_err(p.grad.cpu().numpy(), 2 * ga.cpu().numpy()) < 1e-5   # (atomics reorder fp32 sums: not bitwise)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.zero_grad()                          # set_to_none=True
    assert p.grad is None
    m(xc, speakers=yc)[2].backward()
    assert rel_err(p.grad.cpu().numpy(), ga.cpu().numpy()) < 1e-5
    before = m.flat_parameters().clone()
    opt.step()
    assert not torch.equal(before, m.flat_parameters())


def test_chart_dependencies_batch_independence():
    """reference src/utils.py:451-468: in eval mode the gradient of one sample's embedding w.r.t. the
    other samples' inputs is exactly zero."""
    case = CASES["tiny_k3"]
    m = build(case, None).eval()
    x, _ = case_inputs(case, torch.float32)
    xin = x.cuda().requires_grad_(True)
    out = m(xin)
    idx = 2
    out[idx].sum().backward()
    gi = xin.grad
    assert gi is not None
    for b in range(x.shape[0]):
        if b == idx:
            assert float(gi[b].abs().sum()) > 0
        else:
            assert float(gi[b].abs().sum()) == 0.0
    # and the value matches the oracle's autograd
    sd = case_state_dict(case, None, torch.float64)
    xo = x.double().requires_grad_(True)
    o = O.titanet_forward(sd, xo, oracle_cfg(case), training=False)
    o.normalized[idx].sum().backward()
    assert rel_err(gi.cpu().numpy(), xo.grad.numpy()) < 2e-3
