"""Variable-length batches with a padding mask (BASELINE.json configs[3]) on the HIP path:
``model(spectrograms, speakers, lengths=...)`` -> ``tn_forward_masked``.  The extension is defined by the oracle
(``O.titanet_forward(..., lengths=)``, itself checked against un-padded runs of the reference-pinned restatement):
  * all lengths == T  ==  the unmasked path;
  * eval: a padded batch gives every utterance the embedding it has on its own (un-padded, batch of 1) to 1e-5 in fp32;
  * whatever the padded frames of the input hold is ignored;
  * train: loss, embeddings, BatchNorm running statistics and every gradient against the float64 oracle, fp32 tolerance;
    bf16 and the TitaNet-M width (the configs[3] model: generic kernel templates) at the bf16 tolerance.
"""
import numpy as np
import pytest
import torch

from oracle import titanet_oracle as O
from tests.golden.cases import CASES
from tests.test_forward_gpu import build
from tests.util import case_inputs, case_state_dict, oracle_cfg, rel_err

pytestmark = pytest.mark.gpu


def _pad(x, lengths, fill=0.0):
    x = x.clone()
    for b, n in enumerate(lengths.tolist()):
        x[b, :, n:] = fill
    return x


def test_full_lengths_equal_the_unmasked_path():
    case = CASES["tiny_k3"]
    x, y = case_inputs(case, torch.float32)
    B, T = x.shape[0], x.shape[2]
    res = []
    for lengths in (None, torch.full((B,), T)):
        m = build(case, "ce").train()
        emb, _, lv = m(x.cuda(), speakers=y.cuda(), lengths=lengths)
        lv.backward()
        torch.cuda.synchronize()
        res.append((emb.detach().cpu(), float(lv), m.flat_gradients().clone().cpu(), m.state_dict()["encoder.epilog.conv_block.1.running_var"].cpu()))
    assert abs(res[0][1] - res[1][1]) < 1e-6
    assert rel_err(res[1][0].numpy(), res[0][0].numpy()) < 1e-6
    assert rel_err(res[1][2].numpy(), res[0][2].numpy()) < 1e-5
    assert rel_err(res[1][3].numpy(), res[0][3].numpy()) < 1e-6


@pytest.mark.parametrize("name", ["tiny_k3", "tiny_k7"])
def test_eval_padded_batch_equals_each_utterance_alone(name):
    case = CASES[name]
    m = build(case, None).eval()
    x, _ = case_inputs(case, torch.float32)
    B, T = x.shape[0], x.shape[2]
    lengths = torch.tensor([T, 9, 23, 30][:B]).clamp(max=T)
    with torch.no_grad():
        padded = m(_pad(x, lengths, fill=5.0).cuda(), lengths=lengths).cpu()       # garbage in the padding must not matter
        for b in range(B):
            alone = m(x[b:b + 1, :, :lengths[b]].contiguous().cuda()).cpu()
            assert rel_err(padded[b:b + 1].numpy(), alone.numpy()) < 1e-5, (b, rel_err(padded[b:b + 1].numpy(), alone.numpy()))
        sd = case_state_dict(case, None, torch.float64)
        want = O.titanet_forward(sd, _pad(x, lengths).double(), oracle_cfg(case), training=False, lengths=lengths).normalized
    assert rel_err(padded.numpy(), want.numpy()) < 5e-5


def _train_case(case, loss, precision, lengths, p=0.0):
    m = build(case, loss, precision=precision, dropout=p).train()
    x, y = case_inputs(case, torch.float32)
    xp = _pad(x, lengths)
    emb, preds, lv = m(xp.cuda(), speakers=y.cuda(), lengths=lengths)
    lv.backward()
    torch.cuda.synchronize()
    sd = case_state_dict(case, loss, torch.float64)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
    kw = dict(loss="ce") if loss == "ce" else dict(loss="margin", loss_kwargs=O.margin_kwargs("arc", scale=30, margin=0.2))
    out = O.titanet_forward(sd, xp.double(), oracle_cfg(case), training=True, speakers=y, lengths=lengths, **kw)
    out.loss.backward()
    named = dict(m.named_parameters())
    per = {k: rel_err(named[k].grad.detach().cpu().numpy(), sd[k].grad.numpy()) for k in named
           if float(sd[k].grad.abs().max()) > 1e-9}
    a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in named])
    b = np.concatenate([sd[k].grad.numpy().ravel() for k in named])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    bufs = {k: rel_err(m.state_dict()[k].cpu().numpy(), v.numpy()) for k, v in out.new_buffers.items() if "num_batches" not in k}
    return rel_err(emb.detach().cpu().numpy(), out.normalized.detach().numpy()), abs(float(lv) - float(out.loss)), per, cos, bufs


@pytest.mark.parametrize("name,loss", [("tiny_k3", "ce"), ("tiny_k3", "arc"), ("tiny_k7", "ce")])
def test_train_step_with_padding_mask_fp32_vs_oracle(name, loss):
    case = CASES[name]
    B, T = case["batch"], case["frames"]
    lengths = torch.tensor([T, 9, 23, 30][:B]).clamp(max=T)
    e_emb, d_loss, per, cos, bufs = _train_case(case, loss, "fp32", lengths)
    worst = max(per.items(), key=lambda kv: kv[1])
    print(f"{name}/{loss}: emb {e_emb:.2e} dloss {d_loss:.2e} grad cos {cos:.6f} worst tensor {worst} worst buffer {max(bufs.values()):.2e}")
    assert e_emb < 1e-3 and d_loss < 1e-3
    assert cos > 0.9999 and worst[1] < 3e-2, worst
    assert max(bufs.values()) < 1e-4


@pytest.mark.parametrize("hidden,kernel,precision", [(256, 3, "bf16"), (512, 7, "bf16"), (512, 7, "fp32")])
def test_masked_train_step_s_and_m_width(hidden, kernel, precision):
    """S width in bf16 (a plan that would otherwise run the specialised kernels) and TitaNet-M width (configs[3])"""
    case = dict(cfg=dict(n_mels=80, n_mega_blocks=2, hidden=hidden, enc_out=1536, emb=192, kernel=kernel, attn_hidden=128),
                batch=6, frames=120, n_classes=20, seed=9)
    lengths = torch.tensor([120, 64, 101, 33, 120, 77])
    e_emb, d_loss, per, cos, bufs = _train_case(case, "ce", precision, lengths)
    print(f"H={hidden} K={kernel} {precision}: emb {e_emb:.2e} dloss {d_loss:.2e} grad cos {cos:.5f}")
    if precision == "fp32":
        assert e_emb < 1e-3 and d_loss < 1e-3 and cos > 0.9995
    else:
        assert e_emb < 6e-2 and d_loss < 5e-2 and cos > 0.93


def test_bad_lengths_are_rejected():
    case = CASES["tiny_k3"]
    m = build(case, None).eval()
    x, _ = case_inputs(case, torch.float32)
    with pytest.raises(ValueError):
        m(x.cuda(), lengths=torch.tensor([0, 5, 5, 5]))
    with pytest.raises(ValueError):
        m(x.cuda(), lengths=torch.tensor([5, 5, 5, x.shape[2] + 1]))


def test_more_than_512_utterances_per_batch():
    """The valid-frame counts travel to the device as kernel arguments in chunks of 512 (lens_write_kernel): a padded batch of 600
    utterances gives utterances on both sides of the chunk boundary the embedding they have on their own."""
    from titanet_amd import TitaNet
    torch.manual_seed(3)
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s", device="cuda", precision="bf16").eval()
    B, T = 600, 96
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(B, 80, T, generator=g) * 0.11 - 0.1)
    lengths = torch.randint(20, T + 1, (B,), generator=g)
    lengths[0] = T
    for b in range(B):
        x[b, :, lengths[b]:] = 0
    x = x.cuda()
    with torch.no_grad():
        full = m(x, lengths=lengths).clone()
        for b in (3, 511, 512, 513, 599):
            alone = m(x[b:b + 1, :, :int(lengths[b])].contiguous())
            assert rel_err(full[b:b + 1].cpu().numpy(), alone.cpu().numpy()) < 2e-2, (b, int(lengths[b]))
