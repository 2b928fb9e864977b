"""Gradient parity at TRAINED weights (the per-tensor bf16 tolerances at a random initialisation are 0.2-0.35
because ~0.5 % of the ReLU / dropout-survivor decisions of a randomly initialised network flip under bf16 rounding; a
mis-scaled slab could hide in that).  Here a 2-block S-width model is first trained for 250 fused-Adam steps in fp32 on a
small deterministic task (separable class means, so the activations settle away from the decision boundaries), then the
bf16 / fp8 plans' gradients AT THOSE WEIGHTS are compared tensor by tensor with the float64 oracle (same dropout masks)."""
import numpy as np
import pytest
import torch

from oracle import titanet_oracle as O
from tests.test_forward_gpu import build
from tests.util import mask_fn_for, oracle_cfg, rel_err

pytestmark = pytest.mark.gpu

P, SEED, NCLS = 0.1, 777, 16


def _task(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(NCLS, 80, 1, generator=torch.Generator().manual_seed(5)) * 0.08          # the classes: fixed spectral shapes
    y = torch.randint(0, NCLS, (B,), generator=g)
    x = means[y] + torch.randn(B, 80, T, generator=g) * 0.05 - 0.10
    return x, y


@pytest.mark.parametrize("hidden,kernel,precision", [(256, 3, "bf16"), (512, 7, "bf16"), (512, 7, "fp8"), (1024, 11, "bf16"), (1024, 11, "fp8")])
def test_gradients_at_trained_weights_vs_float64_oracle(hidden, kernel, precision):
    from titanet_amd.trainer import Trainer
    case = dict(cfg=dict(n_mels=80, n_mega_blocks=2, hidden=hidden, enc_out=1536, emb=192, kernel=kernel, attn_hidden=128),
                batch=64, frames=120, n_classes=NCLS, seed=31)
    m32 = build(case, "ce", precision="fp32", dropout=P).train()
    m32._seed_base, m32._step = 20240917, 0           # the dropout stream of the training run: not whatever torch.initial_seed()
                                                      # happens to be after the tests that ran before this one
    tr = Trainer(m32, lr=1e-3)
    first = last = None
    for step in range(250):
        x, y = _task(64, 120, 1000 + step % 8)
        lv = tr.step(x.cuda(), y.cuda())[2]
        if step == 0:
            first = float(lv)
    last = float(lv)
    assert last < 0.5 * first, (first, last)                       # it did train
    sd_trained = {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
    del tr, m32
    torch.cuda.empty_cache()
    # ---- the plan under test at the trained weights
    x, y = _task(64, 120, 4242)
    m = build(case, "ce", precision=precision, dropout=P).train()
    m.load_state_dict(sd_trained)
    # (fp8: the pointwise WEIGHT gradients run on the f8f6f4 MFMA with column scales taken from the plan's previous backward —
    #  a plan's first backward runs the bf16 contraction and records the maxima — so the step is run twice and the second
    #  gradient, the one every training step after the first sees, is the one compared)
    for rep in range(2 if precision == "fp8" else 1):
        m.zero_grad(set_to_none=False)
        m._seed_base, m._step = SEED, 0
        emb, _, lv = m(x.cuda(), speakers=y.cuda())
        lv.backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    got = {k: named[k].grad.detach().cpu().numpy() for k in named}
    # ---- float64 oracle at the same weights, same masks
    sd = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in sd_trained.items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
    out = O.titanet_forward(sd, x.double(), oracle_cfg(case, dropout=P), training=True, speakers=y, loss="ce", mask_fn=mask_fn_for(SEED, P))
    out.loss.backward()
    want = {k: sd[k].grad.numpy() for k in got}
    big = [k for k in got if got[k].size >= 4096 and float(np.abs(want[k]).max()) > 1e-9]
    per = {k: rel_err(got[k], want[k]) for k in big}
    a = np.concatenate([got[k].ravel() for k in got]); b = np.concatenate([want[k].ravel() for k in got])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    worst = sorted(per.items(), key=lambda kv: -kv[1])[:4]
    e_emb = rel_err(emb.detach().cpu().numpy(), out.normalized.detach().numpy())
    print(f"H={hidden} {precision}: loss {float(lv):.4f} / {float(out.loss):.4f} (trained from {first:.3f} to {last:.3f}), emb {e_emb:.2e}, "
          f"gradient cosine {cos:.5f}, worst large tensors {[(k, round(v, 4)) for k, v in worst]}")
    lim = 5e-2 if precision == "bf16" else 1.5e-1                  # fp8: e4m3 forward operands (3 mantissa bits)
    assert cos > (0.999 if precision == "bf16" else 0.99), cos
    for k, v in per.items():
        # (4096-element tensors — the SE weights — sum fewer terms: their bf16 noise averages out less)
        assert v < (lim if got[k].size >= 16384 else 1.6 * lim), (k, v)


def test_full_depth_s17_bf16_vs_fp32_plan_at_trained_weights():
    """The benched configuration's depth (TitaNet-S, 17 mega blocks, train mode, dropout 0.1, bf16) at TRAINED weights
    (VERDICT r3 item 2): after 300 fused-Adam steps in fp32 on the separable task, the bf16 plan against the fp32 plan (the
    1e-3 parity path) of the same weights and the same dropout stream — the output of EVERY mega block down to the 17th, the
    embeddings, the whole gradient, one pointwise weight gradient per block, and the large gradient tensors of the last block
    one by one (a mis-scaled weight-gradient slab in block 17 is a relative error of O(1) there).

    What "close" can mean at this depth is MEASURED in the same test, not assumed: the fp32 plan itself, with nothing changed but
    its weight matrices rounded to bf16 once, moves by 0.002 (block 1) .. 0.036 (block 17) in the block outputs, by 0.41 (block
    1) .. 0.04 (block 17) in the per-block weight gradients and to a whole-gradient cosine of 0.95 (tools/depth_probe.py: the
    input rounded once instead: 0.022 / 0.30 / 0.97; run-to-run noise of the fp32 plan 0 / 0.008 / 0.99998) — a train-mode
    network this deep, trained to a loss of 3e-4, amplifies ANY bf16-sized perturbation that much.  The bf16 plan rounds at
    ~100 storage points and lands at 1.3 - 2.4x that single-rounding sensitivity; the test bounds it by 3x block by block (the
    fp32 training run that produces the weights is not bit-reproducible: atomics), so a kernel error of the size of a few extra
    bf16 roundings per block would already trip it."""
    from titanet_amd.trainer import Trainer
    NB = 17
    case = dict(cfg=dict(n_mels=80, n_mega_blocks=NB, hidden=256, enc_out=1536, emb=192, kernel=3, attn_hidden=128),
                batch=64, frames=120, n_classes=NCLS, seed=33)
    m32 = build(case, "ce", precision="fp32", dropout=P).train()
    m32._seed_base, m32._step = 20240918, 0
    tr = Trainer(m32, lr=1e-3)
    first = None
    for step in range(300):
        x, y = _task(64, 120, 2000 + step % 8)
        lv = tr.step(x.cuda(), y.cuda())[2]
        if step == 0:
            first = float(lv)
    last = float(lv)
    assert last < 0.5 * first, (first, last)
    sd_trained = {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
    del tr, m32
    torch.cuda.empty_cache()
    B, T = 256, 300                      # the benchmarked batch shape (BASELINE configs[1])
    x, y = _task(B, T, 5151)
    wkey = lambda i: f"encoder.mega_blocks.{i}.sub_blocks.2.conv_block.0.conv.1.weight"

    def run(prec, sd):
        m = build(dict(case, batch=B, frames=T), "ce", precision=prec, dropout=P).train()
        m.load_state_dict(sd)
        m._seed_base, m._step = SEED, 0
        emb, _, lv = m(x.cuda(), speakers=y.cuda())
        blocks = [m.debug_fetch(f"block_out:{i}", (B, 256, T)).cpu() for i in range(NB)]
        lv.backward()
        torch.cuda.synchronize()
        out = (blocks, emb.detach().cpu().numpy(), float(lv), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()})
        del m
        torch.cuda.empty_cache()
        return out

    def dist(a, b):
        errs = [float((p - q).norm() / q.norm()) for p, q in zip(a[0], b[0])]
        ga = np.concatenate([a[3][k].ravel() for k in b[3]]); gb = np.concatenate([b[3][k].ravel() for k in b[3]])
        cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
        return errs, rel_err(a[1], b[1]), cos, [rel_err(a[3][wkey(i)], b[3][wkey(i)]) for i in range(NB)]

    ref = run("fp32", sd_trained)
    low = run("bf16", sd_trained)
    sd_rounded = {k: (v.to(torch.bfloat16).float() if (v.dtype == torch.float32 and v.dim() >= 2) else v) for k, v in sd_trained.items()}
    sens = run("fp32", sd_rounded)                # the parity path's own sensitivity to ONE bf16 rounding of its weight matrices
    errs, e_emb, cos, wg = dist(low, ref)
    s_errs, s_emb, s_cos, s_wg = dist(sens, ref)
    last_blk = {k: rel_err(low[3][k], ref[3][k]) for k in ref[3] if f"mega_blocks.{NB - 1}." in k and ref[3][k].size >= 16384}
    print(f"S/17 trained ({first:.3f} -> {last:.4f}): loss {low[2]:.4f} / {ref[2]:.4f}\n  block outputs bf16-vs-fp32", [f"{e:.4f}" for e in errs],
          "\n  block outputs fp32(weights rounded once)-vs-fp32", [f"{e:.4f}" for e in s_errs],
          f"\n  emb {e_emb:.2e} ({s_emb:.2e}), gradient cosine {cos:.5f} ({s_cos:.5f})\n  pointwise weight gradient per block", [f"{e:.3f}" for e in wg],
          "\n  ... of the rounded-weights fp32 plan", [f"{e:.3f}" for e in s_wg],
          "\n  last block tensors", {k.split(f"mega_blocks.{NB - 1}.")[1]: round(v, 4) for k, v in last_blk.items()})
    try:      # the yardstick numbers, kept with the round's profiles (gpurun_out/ travels back from the GPU box)
        import os
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "full_depth_drift.txt"), "w") as fh:
            fh.write(f"TitaNet-S/17, batch {B} x {T} frames, train mode, dropout {P}, weights after 300 fp32 Adam steps (loss {first:.3f} -> {last:.4f})\n")
            fh.write(f"loss bf16 {low[2]:.5f} fp32 {ref[2]:.5f}\n")
            fh.write("block | output bf16-vs-fp32 | output fp32(weights rounded once)-vs-fp32 | pointwise dW bf16-vs-fp32 | ... rounded-weights fp32\n")
            for i in range(NB):
                fh.write(f"{i:5d} | {errs[i]:.4f} | {s_errs[i]:.4f} | {wg[i]:.3f} | {s_wg[i]:.3f}\n")
            fh.write(f"embeddings {e_emb:.3e} (yardstick {s_emb:.3e}); whole-gradient cosine {cos:.5f} (yardstick {s_cos:.5f})\n")
    except OSError:
        pass
    assert abs(low[2] - ref[2]) < 2e-2 * max(1.0, abs(ref[2]))
    for i in range(NB):
        assert errs[i] < 3.0 * s_errs[i] + 5e-3, (i, errs[i], s_errs[i])           # every block output down to the 17th
        assert wg[i] < 2.5 * s_wg[i] + 3e-2, (i, wg[i], s_wg[i])                   # the fused-tail flow's tensors, block by block
    assert max(errs) < 0.12 and e_emb < 5e-2 and e_emb < 3.0 * s_emb + 2e-3, (max(errs), e_emb, s_emb)
    assert 1.0 - cos < 3.0 * (1.0 - s_cos), (cos, s_cos)
    assert len(last_blk) >= 4
    for k, v in last_blk.items():
        assert v < 0.25, (k, v)                   # (measured 0.04 - 0.12, the trained state varies run to run; a mis-scaled slab: >= 0.5)


def test_deep_m10_fp8_plan_vs_fp32_plan_at_trained_weights():
    """the fp8 data gradient was only checked on 2-block nets.  TitaNet-M at its full depth (10 mega blocks, hidden
    512, 7 taps: 30 pointwise layers whose forward GEMM and data gradient run on the f8f6f4 MFMA, 10 skip connections likewise),
    train mode, dropout 0.1, weights after 250 fp32 Adam steps: the fp8 plan and the bf16 plan against the fp32 plan (the parity
    path) on the same batch and dropout stream.  The bf16 plan's distance is the yardstick (what reduced-precision storage costs
    at this depth).  Measured (round 5): e4m3 forward operands cost 2.7 % of the first block's output and the error grows like
    the bf16 plan's, x 1.25 per block, to 17 % at block 10 — 5 - 6 x the bf16 plan's distance at every depth — with a
    whole-gradient cosine of 0.969 (bf16: 0.995).  Bounds: 8 x the bf16 distance block by block, cosine above 0.93; that the plan
    still TRAINS like the bf16 plan at this depth is tests/test_train_compare_gpu.py's statement."""
    from titanet_amd.trainer import Trainer
    NB = 10
    case = dict(cfg=dict(n_mels=80, n_mega_blocks=NB, hidden=512, enc_out=1536, emb=192, kernel=7, attn_hidden=128),
                batch=64, frames=120, n_classes=NCLS, seed=35)
    m32 = build(case, "ce", precision="fp32", dropout=P).train()
    m32._seed_base, m32._step = 20240919, 0
    tr = Trainer(m32, lr=1e-3)
    first = None
    for step in range(250):
        x, y = _task(64, 120, 3000 + step % 8)
        lv = tr.step(x.cuda(), y.cuda())[2]
        if step == 0:
            first = float(lv)
    last = float(lv)
    assert last < 0.5 * first, (first, last)
    sd_trained = {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
    del tr, m32
    torch.cuda.empty_cache()
    B, T = 64, 256
    x, y = _task(B, T, 6161)

    def run(prec):
        m = build(dict(case, batch=B, frames=T), "ce", precision=prec, dropout=P).train()
        m.load_state_dict(sd_trained)
        m._seed_base, m._step = SEED, 0
        emb, _, lv = m(x.cuda(), speakers=y.cuda())
        blocks = [m.debug_fetch(f"block_out:{i}", (B, 512, T)).cpu() for i in range(NB)]
        lv.backward()
        torch.cuda.synchronize()
        out = (blocks, emb.detach().cpu().numpy(), float(lv), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()})
        del m
        torch.cuda.empty_cache()
        return out

    def dist(a, b):
        errs = [float((p - q).norm() / q.norm()) for p, q in zip(a[0], b[0])]
        ga = np.concatenate([a[3][k].ravel() for k in b[3]]); gb = np.concatenate([b[3][k].ravel() for k in b[3]])
        cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
        return errs, rel_err(a[1], b[1]), cos

    ref, low, f8 = run("fp32"), run("bf16"), run("fp8")
    b_errs, b_emb, b_cos = dist(low, ref)
    f_errs, f_emb, f_cos = dist(f8, ref)
    print(f"M/10 trained ({first:.3f} -> {last:.4f}): loss fp32 {ref[2]:.4f} bf16 {low[2]:.4f} fp8 {f8[2]:.4f}\n  block outputs bf16-vs-fp32", [f"{e:.4f}" for e in b_errs],
          "\n  block outputs fp8-vs-fp32 ", [f"{e:.4f}" for e in f_errs],
          f"\n  embeddings bf16 {b_emb:.2e} fp8 {f_emb:.2e}; whole-gradient cosine bf16 {b_cos:.5f} fp8 {f_cos:.5f}")
    assert all(np.isfinite(v).all() for v in f8[3].values())
    assert abs(f8[2] - ref[2]) < 5e-2 * max(1.0, abs(ref[2]))
    for i in range(NB):
        assert f_errs[i] < 8.0 * b_errs[i] + 3e-2, (i, f_errs[i], b_errs[i])
    assert f_emb < 8.0 * b_emb + 2e-2, (f_emb, b_emb)
    assert b_cos > 0.98 and f_cos > 0.93, (f_cos, b_cos)
