"""Differential fuzz: the specialised bf16 kernels (default) vs the generic templates (TN_GENERIC=1) on random shapes,
dropout rates, heads and modes.  Both paths share the same counter-based dropout masks, so embeddings / loss / the whole
gradient must agree to bf16 noise.  Usage: python tools/fuzz_paths.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from titanet_amd import LOSSES, TitaNet


def run(mask, cfg):
    if mask is None:
        os.environ.pop("TN_GENERIC", None)
    else:
        os.environ["TN_GENERIC"] = "1"
    torch.manual_seed(cfg["wseed"])
    if cfg["head"] == "ce":
        lf = LOSSES["ce"](192, cfg["ncls"], device="cuda")
    else:
        lf = LOSSES["arc"](192, cfg["ncls"], device="cuda", scale=30, margin=0.2)
    m = TitaNet.get_titanet(n_mega_blocks=cfg["blocks"], model_size=cfg.get("size", "s"), loss_function=lf, dropout=cfg["p"], device="cuda",
                            precision=cfg.get("prec", "bf16"), simple_pool=cfg["simple"])
    if cfg.get("groups", 1) > 1:
        m.grad_groups = cfg["groups"]          # backward finalises the gradient in 1 + groups buckets (the data-parallel layout)
    m._seed_base, m._step = 777, 0
    g = torch.Generator().manual_seed(cfg["xseed"])
    x = (torch.randn(cfg["B"], 80, cfg["T"], generator=g) * 0.11 - 0.1).cuda()
    y = torch.randint(0, cfg["ncls"], (cfg["B"],), generator=g).cuda()
    lengths = None
    if cfg.get("masked"):
        lengths = torch.randint(1, cfg["T"] + 1, (cfg["B"],), generator=g)
        lengths[int(torch.randint(0, cfg["B"], (1,), generator=g))] = cfg["T"]
    if cfg["train"]:
        m.train()
        # (fp8 plans: the pointwise weight gradients run on the f8f6f4 MFMA from the plan's SECOND backward on — delayed column
        #  scales, include/titanet_amd.h TN_PREC_FP8 — so the same step is run twice and the second gradient is the one compared)
        for rep in range(2 if cfg.get("prec") == "fp8" else 1):
            m.zero_grad(set_to_none=False)
            m._seed_base, m._step = 777, 0
            emb, preds, loss = m(x, speakers=y, lengths=lengths)
            loss.backward()
        grad = torch.cat([p.grad.flatten() for p in m.parameters()]).float().cpu()
        return emb.detach().float().cpu(), float(loss), grad
    m.eval()
    with torch.no_grad():
        emb = m(x, lengths=lengths)
    return emb.float().cpu(), 0.0, None


def main(n=None, seed=None):
    n = n if n is not None else (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
    rng = np.random.default_rng(seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    worst, failed = 0.0, []
    for i in range(n):
        cfg = dict(B=int(rng.integers(6, 40)), T=int(rng.choice([33, 64, 65, 100, 151, 201, 300, 301, 417, 520, 700])),
                   p=float(rng.choice([0.0, 0.1, 0.3])), head=str(rng.choice(["ce", "arc"])), blocks=int(rng.integers(1, 4)),
                   ncls=int(rng.integers(5, 60)), simple=bool(rng.random() < 0.2), train=bool(rng.random() < 0.8),
                   wseed=int(rng.integers(1 << 30)), xseed=int(rng.integers(1 << 30)),
                   size=str(rng.choice(["s", "s", "m", "l"])), masked=bool(rng.random() < 0.6))
        cfg["groups"] = int(rng.choice([1, 1, 2, 3]))
        cfg["prec"] = "fp8" if (cfg["size"] != "s" and rng.random() < 0.3) else "bf16"
        if cfg["size"] != "s":
            cfg["blocks"] = min(cfg["blocks"], 2)
        if cfg["masked"]:
            cfg["simple"] = False              # (the mean-pool decoder has no masked variant)
        e0, l0, g0 = run("0", cfg)
        e1, l1, g1 = run(None, cfg)
        er = float((e1 - e0).norm() / e0.norm())
        msg = f"{i:3d} {cfg}  emb rel {er:.2e}"
        f8 = cfg.get("prec") == "fp8"          # (TN_GENERIC plans compute in bf16: an fp8 case compares e4m3 forward GEMMs with bf16 ones)
        ok = er < (1e-1 if f8 else 5e-2)
        if cfg["train"]:
            cos = float((g0 @ g1) / (g0.norm() * g1.norm()))
            msg += f"  dloss {abs(l1 - l0):.2e}  grad cos {cos:.4f}"
            ok = ok and abs(l1 - l0) < 5e-2 * max(1.0, abs(l0)) and cos > ((0.95 if f8 else 0.97) if cfg["B"] * cfg["T"] >= 1500 else 0.9)
        worst = max(worst, er)
        print(("ok  " if ok else "FAIL") + msg, flush=True)
        if not ok:
            failed.append(msg)
    print("worst emb rel", worst)
    return failed


if __name__ == "__main__":
    main()
