"""TN_PREC_FP8 (BASELINE.json configs[4]: "TitaNet-L, fp8 MFMA pointwise-conv path"): the forward pointwise GEMMs of the
mega-block sub-blocks on v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales) (OCP e4m3 operands, per-output-channel weight scales, f32
accumulation), everything else the bf16 plan.  Stated tolerance of the mode, against the float64 oracle of the reference
path: embeddings within 0.12 relative, loss within 5 %, whole-gradient cosine > 0.9 at the TitaNet-L width (e4m3 carries 3
mantissa bits: 6 % per operand element, averaged down by the K = 1024 contraction); and the SHARP statement: against the
oracle that rounds the same operands to e4m3 (and stores bf16 where the plan does) the forward agrees at the bf16 noise
floor — the kernels add nothing beyond the declared formats."""
import numpy as np
import pytest
import torch

from oracle import titanet_oracle as O
from tests.test_forward_gpu import build
from tests.util import case_inputs, case_state_dict, oracle_cfg, rel_err

pytestmark = pytest.mark.gpu


def _case(hidden, kernel, blocks=1, batch=24, frames=120):
    return dict(cfg=dict(n_mels=80, n_mega_blocks=blocks, hidden=hidden, enc_out=1536, emb=192, kernel=kernel, attn_hidden=128),
                batch=batch, frames=frames, n_classes=30, seed=31)


@pytest.mark.parametrize("hidden,kernel", [(1024, 11), (512, 7), (256, 3)])
def test_fp8_train_step_vs_oracles(hidden, kernel):
    case = _case(hidden, kernel, blocks=2 if hidden < 1024 else 1)
    m = build(case, "ce", precision="fp8").train()
    x, y = case_inputs(case, torch.float32)
    emb, preds, lv = m(x.cuda(), speakers=y.cuda())
    lv.backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    res = {}
    for tag in ("plain", "emu"):
        sd = case_state_dict(case, "ce", torch.float64 if tag == "plain" else torch.float32)
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running_" not in k:
                v.requires_grad_(True)
        xo, yo = case_inputs(case, torch.float64 if tag == "plain" else torch.float32)
        cfg = oracle_cfg(case)
        if tag == "emu":
            cfg.store_round, cfg.pw_operand_round = O.bf16_store, O.fp8_operands
        out = O.titanet_forward(sd, xo, cfg, training=True, speakers=yo, loss="ce")
        out.loss.backward()
        a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in named])
        b = np.concatenate([sd[k].grad.numpy().ravel() for k in named])
        res[tag] = (rel_err(emb.detach().cpu().numpy(), out.normalized.detach().numpy()), abs(float(lv) - float(out.loss)) / abs(float(out.loss)),
                    float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))))
        print(f"H={hidden} fp8 vs {tag} oracle: emb {res[tag][0]:.3e} loss rel {res[tag][1]:.3e} grad cos {res[tag][2]:.4f}")
    assert res["plain"][0] < 0.12 and res["plain"][1] < 0.05 and res["plain"][2] > 0.9
    assert res["emu"][0] < 0.08 and res["emu"][1] < 0.02 and res["emu"][2] > 0.95


def test_fp8_eval_and_training_run():
    case = _case(1024, 11, blocks=2, batch=16, frames=120)
    from titanet_amd.trainer import Trainer
    m = build(case, "ce", precision="fp8").train()
    tr = Trainer(m, lr=1e-3)
    x, y = case_inputs(case, torch.float32)
    losses = [float(tr.step(x.cuda(), y.cuda())[2]) for _ in range(30)]
    assert losses[-1] < 0.5 * losses[0], losses[::5]            # overfits a fixed batch through the fp8 forward
    m.eval()
    with torch.no_grad():
        e8 = m(x.cuda()).cpu()
    mb = build(case, "ce", precision="bf16").eval()
    mb.load_state_dict(m.state_dict())
    with torch.no_grad():
        eb = mb(x.cuda()).cpu()
    d = float((e8 - eb).norm() / eb.norm())
    print("eval embeddings fp8 vs bf16 plan:", d)
    assert d < 0.1


@pytest.mark.parametrize("hidden,kernel,masked", [(1024, 11, False), (512, 7, False), (1024, 11, True)])
def test_fp8_data_gradient_vs_bf16_data_gradient(hidden, kernel, masked, monkeypatch):
    """Round 4: under the fp8 plan the sub-block data gradients dS * W run on the f8f6f4 MFMA too (e4m3 dS rows with one
    power-of-two scale per row as the MFMA's block scale, e4m3 W^T rows with per-input-channel scales; tn_pgemm.h F8 + rowexp).
    Same weights, batch and dropout stream with precision="fp8_fwd" (TN_PREC_FP8_FWD: bf16 backward of the same plan): every large gradient tensor
    within 8e-2, whole-gradient cosine > 0.998 — and not identical, i.e. the fp8 kernels did run."""
    case = _case(hidden, kernel, blocks=2, batch=16, frames=128)
    x, y = case_inputs(case, torch.float32)
    lengths = None
    if masked:
        g = torch.Generator().manual_seed(5)
        lengths = torch.randint(20, 129, (16,), generator=g)
        lengths[3] = 128
    grads = {}
    monkeypatch.delenv("TN_FP8_BWD", raising=False)
    for tag, prec in (("fp8", "fp8"), ("bf16", "fp8_fwd")):      # fp8_fwd == TN_PREC_FP8_FWD: the same plan, backward in bf16
        m = build(case, "ce", precision=prec).train()
        m._seed_base, m._step = 11, 0
        emb, preds, lv = m(x.cuda(), speakers=y.cuda(), lengths=lengths)
        lv.backward()
        torch.cuda.synchronize()
        grads[tag] = {k: p.grad.detach().float().cpu().numpy().copy() for k, p in m.named_parameters()}
        assert all(np.isfinite(v).all() for v in grads[tag].values())
    a = np.concatenate([grads["fp8"][k].ravel() for k in grads["bf16"]])
    b = np.concatenate([grads["bf16"][k].ravel() for k in grads["bf16"]])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    per = {k: rel_err(grads["fp8"][k], v) for k, v in grads["bf16"].items() if v.size >= 16384}
    worst = sorted(per.items(), key=lambda kv: -kv[1])[:3]
    print(f"H={hidden} masked={masked}: fp8 vs bf16 data gradient: cosine {cos:.5f}, worst large tensors {worst}")
    assert cos > 0.998 and worst[0][1] < 8e-2
    assert not np.array_equal(a, b)


@pytest.mark.parametrize("hidden,kernel,masked", [(1024, 11, False), (512, 7, False), (512, 7, True)])
def test_fp8_weight_gradient_vs_bf16_weight_gradient(hidden, kernel, masked, monkeypatch):
    """Round 5: the sub-block pointwise WEIGHT gradients of an fp8 plan run on the f8f6f4 MFMA too (tn_pgemm.h:
    pgemm_tn_f8_batched_kernel): e4m3 dS scaled per column by the previous backward's column maxima (delayed scaling; the
    scale is the MFMA's own block-scale operand), the kept e4m3 depthwise outputs, byte-transposing LDS reads.  A plan's FIRST
    backward has no maxima yet and runs the bf16 contraction, so the same step is run twice on one plan (same weights, batch and
    dropout stream) and the SECOND gradient is compared with the same plan under TN_FP8_WGRAD=0 (bf16 weight gradients, read
    at plan creation): whole-gradient cosine > 0.998, every large tensor within 8e-2 — and the pointwise weight gradients are
    not identical, i.e. the fp8 contraction did run."""
    case = _case(hidden, kernel, blocks=2, batch=16, frames=128)
    x, y = case_inputs(case, torch.float32)
    lengths = None
    if masked:
        g = torch.Generator().manual_seed(5)
        lengths = torch.randint(20, 129, (16,), generator=g)
        lengths[3] = 128
    grads = {}
    for tag, env in (("fp8", None), ("bf16", "0")):
        if env is None:
            monkeypatch.delenv("TN_FP8_WGRAD", raising=False)
        else:
            monkeypatch.setenv("TN_FP8_WGRAD", env)
        m = build(case, "ce", precision="fp8").train()
        for rep in range(2):
            m.zero_grad(set_to_none=False)
            m._seed_base, m._step = 11, 0
            emb, preds, lv = m(x.cuda(), speakers=y.cuda(), lengths=lengths)
            lv.backward()
        torch.cuda.synchronize()
        grads[tag] = {k: p.grad.detach().float().cpu().numpy().copy() for k, p in m.named_parameters()}
        assert all(np.isfinite(v).all() for v in grads[tag].values())
    monkeypatch.delenv("TN_FP8_WGRAD", raising=False)
    a = np.concatenate([grads["fp8"][k].ravel() for k in grads["bf16"]])
    b = np.concatenate([grads["bf16"][k].ravel() for k in grads["bf16"]])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    per = {k: rel_err(grads["fp8"][k], v) for k, v in grads["bf16"].items() if v.size >= 16384}
    worst = sorted(per.items(), key=lambda kv: -kv[1])[:3]
    pw = [k for k in grads["bf16"] if k.endswith("conv_block.0.conv.1.weight")]
    print(f"H={hidden} masked={masked}: fp8 vs bf16 weight gradient: cosine {cos:.5f}, worst large tensors {worst}; "
          f"pointwise weights {[(k.split('mega_blocks.')[1][:14], round(per[k], 4)) for k in pw]}")
    assert cos > 0.998 and worst[0][1] < 8e-2
    assert any(not np.array_equal(grads["fp8"][k], grads["bf16"][k]) for k in pw)
