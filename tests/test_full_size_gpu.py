"""BASELINE.json's full size (TitaNet-S/17, batch 256, 80 x 300): the oracle would need minutes per case there, so
parity is held through size-independent properties of the path instead (each one is also true of the reference):

* eval mode is per-utterance: the embedding of utterance i does not depend on its batch (running BN statistics);
* train mode is permutation-equivariant over the batch (batch BN statistics are symmetric sums; fp32 tolerance 3e-4:
  the summation order of 76,800-row statistics changes, 70 BatchNorms deep);
* train-mode BatchNorm removes the scale/shift of the preceding conv: scaling the prolog conv weight+bias leaves
  the embeddings and the loss unchanged, and every conv bias in front of a BN has zero gradient;
* backward is linear in the loss scale;
* softmax cross-entropy: the logit gradients of every utterance sum to zero -> the head bias gradient sums to 0;
  the reported loss equals log-softmax of the head applied to the returned (normalised -> raw) embeddings.
"""
import numpy as np
import pytest
import torch

from titanet_amd import LOSSES, TitaNet

pytestmark = pytest.mark.gpu

B, T, NCLS = 256, 300, 251


def make(precision, dropout=0.0, seed=0):
    torch.manual_seed(seed)
    loss = LOSSES["ce"](192, NCLS, device="cuda")
    return TitaNet.get_titanet(embedding_size=192, n_mels=80, n_mega_blocks=17, model_size="s", attention_hidden_size=128,
                               loss_function=loss, dropout=dropout, device="cuda", precision=precision)


def batch(seed=42):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, 80, T, generator=g) * 0.11 - 0.10).cuda()
    y = torch.randint(0, NCLS, (B,), generator=g).cuda()
    return x, y


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_eval_embeddings_do_not_depend_on_the_batch(precision):
    m = make(precision).eval()
    x, _ = batch()
    with torch.no_grad():
        full = m(x).clone()
        part = m(x[40:48].contiguous()).clone()
        one = m(x[255:256].contiguous()).clone()
    assert full.shape == (B, 192)
    assert torch.allclose(full.norm(dim=1), torch.ones(B, device="cuda"), atol=1e-5)
    assert rel(full[40:48], part) < 1e-6, rel(full[40:48], part)
    assert rel(full[255:256], one) < 1e-6


def test_train_forward_is_permutation_equivariant_fp32():
    m = make("fp32").train()
    x, y = batch()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        e1, p1, l1 = [t.clone() for t in m(x, speakers=y)]
        e2, p2, l2 = [t.clone() for t in m(x[perm].contiguous(), speakers=y[perm].contiguous())]
    assert rel(e2, e1[perm]) < 3e-4, rel(e2, e1[perm])
    assert abs(float(l1) - float(l2)) < 3e-4 * max(1.0, abs(float(l1)))
    assert (p2 == p1[perm]).float().mean() > 0.99


def test_eval_forward_is_permutation_equivariant_bf16():
    """bf16 in TRAIN mode at random initialisation is not a usable property carrier: a randomly initialised 17-block
    TitaNet amplifies perturbations ~1.3x per block (measured in fp32: 2.6e-7 after the prolog -> 3.9e-5 after block
    16, from nothing but the summation order of the BN statistics), so fresh bf16 rounding noise per layer decorrelates
    two train-mode runs.  Eval mode has no order-dependent sums and must be exact."""
    m = make("bf16").eval()
    x, _ = batch()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        e1 = m(x).clone()
        e2 = m(x[perm].contiguous()).clone()
    assert rel(e2, e1[perm]) < 1e-6, rel(e2, e1[perm])


def test_batchnorm_absorbs_prolog_scale_and_conv_biases_have_zero_gradient():
    m = make("fp32").train()
    x, y = batch()
    named = dict(m.named_parameters())
    with torch.no_grad():                        # large scales on both sides so that BN's eps (1e-5) is negligible
        named["encoder.prolog.conv_block.0.weight"].mul_(8.0)
        named["encoder.prolog.conv_block.0.bias"].mul_(8.0)
    e1, _, l1 = m(x, speakers=y)
    l1.backward()
    e1, l1 = e1.detach().clone(), float(l1.detach())
    gnorm = torch.cat([p.grad.flatten() for p in named.values()]).norm()
    # every conv / linear bias directly in front of a train-mode BatchNorm has (numerically) zero gradient
    for k, p in named.items():
        if k.endswith("conv.1.bias") or k.endswith("skip_connection.0.bias") or k.endswith("prolog.conv_block.0.bias") \
                or k.endswith("epilog.conv_block.0.bias") or k == "decoder.linear.0.bias":
            assert float(p.grad.norm()) < 1e-4 * float(gnorm), (k, float(p.grad.norm()), float(gnorm))
    with torch.no_grad():
        named["encoder.prolog.conv_block.0.weight"].mul_(8.0)
        named["encoder.prolog.conv_block.0.bias"].mul_(8.0).add_(0.5)
        e2, _, l2 = m(x, speakers=y)
    assert rel(e2, e1) < 1e-3, rel(e2, e1)
    assert abs(float(l2) - l1) < 1e-3 * abs(l1)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 5e-2)])
def test_backward_is_linear_in_the_loss_scale_and_head_bias_gradient_sums_to_zero(precision, tol):
    """two backward passes from ONE forward (the saved state of the plan), loss scaled by 8 the second time"""
    m = make(precision, dropout=0.1).train()
    x, y = batch()
    _, _, l = m(x, speakers=y)
    l.backward(retain_graph=True)
    g1 = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    m.zero_grad()
    (l * 8.0).backward()
    g8 = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    assert torch.isfinite(g1).all() and float(g1.norm()) > 0
    assert rel(g8, g1 * 8.0) < tol, rel(g8, g1 * 8.0)
    gb = dict(m.named_parameters())["loss_function.fc.bias"].grad
    assert abs(float(gb.sum())) < 1e-5 * max(1.0, float(gb.abs().sum()))


def test_reported_loss_and_preds_match_the_head_on_raw_embeddings():
    m = make("fp32").train()
    x, y = batch()
    with torch.no_grad():
        emb, preds, loss = m(x, speakers=y)
        raw = m.debug_fetch("embeddings_raw", (B, 192))
        got_logits = m.debug_fetch("logits", (B, NCLS))
    named = dict(m.named_parameters())
    logits = raw.double() @ named["loss_function.fc.weight"].double().t() + named["loss_function.fc.bias"].double()
    assert rel(got_logits, logits) < 1e-5
    want = torch.nn.functional.cross_entropy(logits, y)
    assert abs(float(want) - float(loss)) < 1e-5 * max(1.0, float(want))
    assert (logits.argmax(1) == preds).float().mean() > 0.995
    assert rel(emb, torch.nn.functional.normalize(raw, dim=1)) < 1e-6


def test_arcface_full_size_properties():
    """BASELINE.json configs[2] per GPU: TitaNet-S/17, batch 256, ArcFace(30, 0.2) (reference src/losses.py:77-132).
    Properties the reference's head has at any size: the head weight comes back row-normalised (in-place, :88-90); the
    reported loss / predictions equal the margin softmax recomputed from the returned (already normalised) embeddings and
    that weight; the weight gradient of row c is orthogonal to nothing in particular but the gradient wrt a normalised
    embedding direction is tangent (x . dL/dx = 0 because the head sees x / |x|); backward is linear in the loss scale."""
    torch.manual_seed(0)
    loss = LOSSES["arc"](192, NCLS, device="cuda", scale=30, margin=0.2)
    m = TitaNet.get_titanet(embedding_size=192, n_mels=80, n_mega_blocks=17, model_size="s", attention_hidden_size=128,
                            loss_function=loss, dropout=0.1, device="cuda", precision="fp32").train()
    x, y = batch()
    emb, preds, l = m(x, speakers=y)
    W = dict(m.named_parameters())["loss_function.fc.weight"].detach().double()
    assert torch.allclose(W.norm(dim=1), torch.ones(NCLS, device="cuda", dtype=torch.float64), atol=1e-5)
    e = emb.detach().double()
    assert torch.allclose(e.norm(dim=1), torch.ones(B, device="cuda", dtype=torch.float64), atol=1e-5)
    cos = (e @ W.t()).clamp(-1, 1)
    assert (cos.argmax(1) == preds).float().mean() > 0.995
    s, mg, eps = 30.0, 0.2, 1e-6
    tgt = cos.gather(1, y[:, None]).squeeze(1)
    num = s * torch.cos(torch.arccos(tgt) + mg)
    excl = torch.exp(s * cos).sum(1) - torch.exp(s * tgt)
    want = -(num - torch.log(torch.exp(num) + excl + eps)).mean()
    assert abs(float(want) - float(l)) < 1e-4 * max(1.0, abs(float(want))), (float(want), float(l))
    l.backward(retain_graph=True)
    g1 = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    m.zero_grad()
    (l * 4.0).backward()
    g4 = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    assert torch.isfinite(g1).all() and float(g1.norm()) > 0
    assert rel(g4, g1 * 4.0) < 1e-4


def test_validation_forward_between_forward_and_backward():
    """the reference allows `loss = model(x, speakers=y)` ... `model.eval(); model(val)` ... `loss.backward()`: the eval
    forward (same shape on purpose) runs on its own plan and must not disturb the saved state of the train forward"""
    m = make("fp32", dropout=0.1).train()
    x, y = batch()
    xs, ys = x[:32].contiguous(), y[:32].contiguous()
    _, _, l = m(xs, speakers=ys)
    l.backward(retain_graph=True)                 # reference gradient from this forward's saved state
    g_ref = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    m.zero_grad()
    m.eval()
    with torch.no_grad():
        v1 = m(xs).clone()
        v2 = m(x[:8].contiguous()).clone()
    m.train()
    l.backward()                                  # ... and again after the validation forwards
    g_mid = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
    # (two backward passes from one forward differ by the summation order of the atomics-accumulated BatchNorm sums only)
    assert rel(g_mid, g_ref) < 1e-3, rel(g_mid, g_ref)
    assert torch.isfinite(v1).all() and torch.isfinite(v2).all()
    # ... while a second TRAIN forward of the same shape does invalidate it, loudly
    _, _, l3 = m(xs, speakers=ys)
    _, _, l4 = m(xs, speakers=ys)
    with pytest.raises(RuntimeError, match="saved activations"):
        l3.backward()
