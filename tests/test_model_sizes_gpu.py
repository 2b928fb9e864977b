"""TitaNet-M / -L widths and kernels (hidden 512 / 1024, K = 7 / 11: BASELINE configs[3], [4]) on the generic
kernel path, forward + backward against the CPU oracle, plus the S width in fp32 and bf16 at a batch that
exercises multi-tile rows."""
import numpy as np
import pytest
import torch

from oracle import titanet_oracle as O
from tests.test_forward_gpu import build
from tests.util import case_inputs, case_state_dict, oracle_cfg, rel_err

pytestmark = pytest.mark.gpu


def _case(size, n_blocks, batch, frames, seed):
    h, k = {"s": (256, 3), "m": (512, 7), "l": (1024, 11)}[size]
    return dict(cfg=dict(n_mels=80, n_mega_blocks=n_blocks, hidden=h, enc_out=1536, emb=192, kernel=k, attn_hidden=128),
                batch=batch, frames=frames, n_classes=30, seed=seed)


@pytest.mark.parametrize("size,precision", [("m", "fp32"), ("l", "fp32"), ("s", "fp32"), ("m", "bf16"), ("l", "bf16")])
def test_model_family_forward_backward(size, precision):
    case = _case(size, 2 if size != "l" else 1, 6, 77, {"s": 11, "m": 12, "l": 13}[size])
    m = build(case, "ce", precision=precision).train()
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
    named = dict(m.named_parameters())
    a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in named])
    b = np.concatenate([sd[k].grad.numpy().ravel() for k in named])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    e_emb = rel_err(emb.detach().cpu().numpy(), out.normalized.detach().numpy())
    print(size, precision, "emb rel", e_emb, "loss", lv.item(), out.loss.item(), "grad cos", cos, "grad rel", rel_err(a, b))
    if precision == "fp32":
        assert e_emb < 1e-3 and abs(lv.item() - out.loss.item()) < 1e-3 * max(1.0, abs(out.loss.item()))
        assert cos > 0.9995 and rel_err(a, b) < 3e-2
    else:
        assert e_emb < 8e-2 and abs(lv.item() - out.loss.item()) < 0.1 * max(1.0, abs(out.loss.item()))
        assert cos > 0.9
        # every pointwise weight gradient on its own (bf16 wide models: 256 x 256 slab units of the batched launch — a
        # misplaced slab would hide inside the whole-gradient cosine)
        for k in named:
            if k.endswith("conv.1.weight") or k.endswith("skip_connection.0.weight") or k == "encoder.epilog.conv_block.0.weight":
                e = rel_err(named[k].grad.detach().cpu().numpy(), sd[k].grad.numpy())
                assert e < 0.35, (k, e)


def test_eval_forward_sizes():
    for size in ("m", "l"):
        case = _case(size, 1, 3, 130, 21)
        m = build(case, None).eval()
        x, _ = case_inputs(case, torch.float32)
        sd = case_state_dict(case, None, torch.float64)
        with torch.no_grad():
            e = m(x.cuda())
            o = O.titanet_forward(sd, x.double(), oracle_cfg(case), training=False)
        assert rel_err(e.cpu().numpy(), o.normalized.numpy()) < 5e-5, size
