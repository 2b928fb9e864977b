"""Shared helpers for the parity tests (TEST INFRASTRUCTURE)."""
import os

import numpy as np
import torch

from oracle import detgen, rng
from oracle import titanet_oracle as O
from tests.golden.cases import CASES

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def oracle_cfg(case, dropout=0.0):
    c = case["cfg"]
    return O.OracleConfig(n_mels=c["n_mels"], n_mega_blocks=c["n_mega_blocks"], hidden=c["hidden"],
                          enc_out=c["enc_out"], emb=c["emb"], kernel=c["kernel"], attn_hidden=c["attn_hidden"],
                          dropout=dropout, simple_pool=bool(case.get("simple_pool", False)))


def case_state_dict(case, loss=None, dtype=torch.float64):
    cfg = oracle_cfg(case)
    shapes = O.state_dict_shapes(cfg, loss=("ce" if loss == "ce" else ("margin" if loss else None)),
                                 n_classes=case["n_classes"])
    vals = detgen.fill_state_dict(shapes, seed=case["seed"])
    sd = {}
    for k, v in vals.items():
        t = torch.from_numpy(np.asarray(v))
        sd[k] = t if t.dtype == torch.int64 else t.float().to(dtype)  # canonical weights are float32 values
    return sd


def case_inputs(case, dtype=torch.float64):
    c = case["cfg"]
    x = torch.from_numpy(detgen.spectrograms(case["batch"], c["n_mels"], case["frames"], seed=case["seed"])).float().to(dtype)
    y = torch.from_numpy(detgen.speakers(case["batch"], case["n_classes"], seed=case["seed"]))
    return x, y


LOSS_KW = {
    "ce": None,
    "arc": O.margin_kwargs("arc", scale=30, margin=0.2),
    "cos": O.margin_kwargs("cos", scale=64, margin=0.2),
    "sphere": O.margin_kwargs("sphere", margin=4),
}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def mask_fn_for(seed, p):
    def fn(layer_id, shape):
        B, C, T = shape
        return torch.from_numpy(rng.keep_mask_bct(seed, layer_id, B, C, T, p))
    return fn
