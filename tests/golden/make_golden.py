#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REAL reference (Wadaboa/titanet).

Run in the build container only (it imports /root/reference/src, which never travels):

    python tests/golden/make_golden.py

Weights and inputs are NOT stored: they are rebuilt anywhere from ``oracle/detgen.py``
(name+shape -> values).  Only outputs of the reference are stored, as small ``.npz`` files
next to this script.  The fixtures pin ``oracle/titanet_oracle.py`` (CPU tests) and the HIP
path (GPU tests).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")

import torch  # noqa: E402

import losses as ref_losses  # noqa: E402  (reference)
import models as ref_models  # noqa: E402  (reference)

from oracle import detgen  # noqa: E402
from tests.golden.cases import CASES, head_clamp_inputs  # noqa: E402


def build_reference(case, loss_name, dtype):
    cfg = case["cfg"]
    loss_fn = None
    if loss_name == "ce":
        loss_fn = ref_losses.CELoss(cfg["emb"], case["n_classes"])
    elif loss_name == "arc":
        loss_fn = ref_losses.ArcFaceLoss(cfg["emb"], case["n_classes"], scale=30, margin=0.2)  # parameters.yml:42-44
    elif loss_name == "cos":
        loss_fn = ref_losses.CosFaceLoss(cfg["emb"], case["n_classes"], scale=64, margin=0.2)
    elif loss_name == "sphere":
        loss_fn = ref_losses.SphereFaceLoss(cfg["emb"], case["n_classes"], margin=4)  # parameters.yml:38-39
    model = ref_models.TitaNet(
        n_mels=cfg["n_mels"], n_mega_blocks=cfg["n_mega_blocks"], n_sub_blocks=3,
        encoder_hidden_size=cfg["hidden"], encoder_output_size=cfg["enc_out"], embedding_size=cfg["emb"],
        mega_block_kernel_size=cfg["kernel"], attention_hidden_size=cfg["attn_hidden"],
        simple_pool=bool(case.get("simple_pool", False)), loss_function=loss_fn, dropout=0.0)
    sd = model.state_dict()
    vals = detgen.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed=case["seed"])
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})
    return model.to(dtype)


def hooks_for(model, store):
    hs = []

    def add(mod, name):
        hs.append(mod.register_forward_hook(lambda m, i, o, n=name: store.__setitem__(n, o.detach().double().numpy())))

    add(model.encoder.prolog, "encoder.prolog.out")
    for i, mb in enumerate(model.encoder.mega_blocks):
        add(mb, f"encoder.mega_blocks.{i}.out")
        add(mb.sub_blocks[3].excitation, f"encoder.mega_blocks.{i}.sub_blocks.3.excitation.gate")
    add(model.encoder.epilog, "encoder.epilog.out")
    add(model.decoder.pool[0], "decoder.pool.0.out")
    return hs


def run_case(name, case):
    out = {}
    cfg = case["cfg"]
    # canonical inputs and weights are float32 values (what the GPU path receives), widened for f64 runs
    x64 = torch.from_numpy(detgen.spectrograms(case["batch"], cfg["n_mels"], case["frames"], seed=case["seed"])).float().double()
    y = torch.from_numpy(detgen.speakers(case["batch"], case["n_classes"], seed=case["seed"]))

    # ---- eval forward, float64 and float32 (inference mode returns normalised embeddings)
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        model = build_reference(case, None, dtype).eval()
        inter = {}
        hs = hooks_for(model, inter) if (dtype == torch.float64 and case.get("inter")) else []
        with torch.no_grad():
            emb = model(x64.to(dtype))
        for h in hs:
            h.remove()
        out[f"eval.{tag}.embeddings"] = emb.double().numpy()
        for k, v in inter.items():
            out[f"eval.f64.inter.{k}"] = v.astype(np.float32)

    # ---- train-mode forward+backward (dropout 0) per loss
    for loss_name in case["losses"]:
        model = build_reference(case, loss_name, torch.float64).train()
        logits_store = {}
        h = model.loss_function.fc.register_forward_hook(
            lambda m, i, o: logits_store.__setitem__("logits", o.detach().numpy()))
        xin = x64.clone().requires_grad_(True)
        emb, preds, loss = model(xin, speakers=y)
        loss.backward()
        h.remove()
        p = f"train.{loss_name}"
        out[p + ".embeddings"] = emb.detach().numpy()
        out[p + ".preds"] = preds.numpy()
        out[p + ".loss"] = np.asarray(loss.item())
        out[p + ".logits"] = logits_store["logits"]
        sd = model.state_dict()
        if loss_name != "ce":
            out[p + ".fc_weight_after"] = sd["loss_function.fc.weight"].numpy()
        if case.get("grads") == "all":
            for k, v in model.named_parameters():
                out[p + ".grad." + k] = v.grad.numpy().astype(np.float32)
            out[p + ".grad.input"] = xin.grad.numpy().astype(np.float32)
        else:
            for k, v in model.named_parameters():
                if k in case.get("grads", ()):
                    out[p + ".grad." + k] = v.grad.numpy().astype(np.float32)
        # BN buffers after one step (a7)
        for k in case.get("buffers", ()):
            out[p + ".buffer." + k] = sd[k].numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def metrics_golden():
    """EER / minDCF known answers from the reference's own utils (src/utils.py:294-367),
    imported with its presentation-only dependencies stubbed."""
    import types
    for mod in ("wandb", "umap", "IPython", "IPython.display", "torchaudio", "librosa", "soundfile",
                "librosa.display", "matplotlib", "matplotlib.pyplot", "seaborn"):
        if mod not in sys.modules:
            try:
                __import__(mod)
            except Exception:
                sys.modules[mod] = types.ModuleType(mod)
    try:
        import utils as ref_utils
    except Exception as e:  # pragma: no cover
        print("metrics golden skipped:", e)
        return
    rng = np.random.default_rng(0)
    labels = rng.integers(0, 2, 1000)
    scores = labels * 0.3 + rng.normal(0, 0.3, 1000)
    eer = ref_utils.compute_eer(scores, labels)
    mindcf = ref_utils.compute_mindcf(scores, labels, 0.01, 1, 1)
    # second, degenerate-ish case: small, with ties
    labels2 = np.array([1, 0, 1, 1, 0, 0, 1, 0, 1, 0])
    scores2 = np.array([0.9, 0.8, 0.8, 0.4, 0.35, 0.3, 0.3, 0.2, 0.6, 0.6])
    eer2 = ref_utils.compute_eer(scores2, labels2)
    mindcf2 = ref_utils.compute_mindcf(scores2, labels2, 0.01, 1, 1)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), eer=eer, mindcf=np.asarray(mindcf, dtype=np.float64),
                        eer2=eer2, mindcf2=np.asarray(mindcf2, dtype=np.float64),
                        labels2=labels2, scores2=scores2)
    print("metrics:", eer, mindcf, eer2, mindcf2)


def head_clamp_golden():
    x, w, y = head_clamp_inputs()
    out = {}
    for name, ctor in (("arc", lambda: ref_losses.ArcFaceLoss(16, 9, scale=30, margin=0.2)),
                       ("cos", lambda: ref_losses.CosFaceLoss(16, 9, scale=64, margin=0.2))):
        head = ctor().double()
        head.fc.weight.data = torch.from_numpy(w).double()
        xin = torch.from_numpy(x).double().requires_grad_(True)
        cos_store = {}
        h = head.fc.register_forward_hook(lambda m, i, o: cos_store.__setitem__("cos", o.detach().numpy()))
        norm, preds, loss = head(xin, torch.from_numpy(y))
        loss.backward()
        h.remove()
        out[name + ".raw_cos"] = cos_store["cos"]           # BEFORE the clamp: rows 0 / 1 reach +-1 (up to rounding)
        out[name + ".normalized"] = norm.detach().numpy()
        out[name + ".preds"] = preds.numpy()
        out[name + ".loss"] = np.asarray(loss.item())
        out[name + ".grad.inputs"] = xin.grad.numpy()
        out[name + ".grad.weight"] = head.fc.weight.grad.numpy()
        out[name + ".weight_after"] = head.fc.weight.data.numpy()
    np.savez_compressed(os.path.join(HERE, "head_clamp.npz"), **out)
    print("head_clamp: raw cos extremes", out["arc.raw_cos"].max(), out["arc.raw_cos"].min(), "loss", out["arc.loss"])


def sizing_golden():
    """Known answers for model sizing (titanet.ipynb:743,765,787,961)."""
    out = {}
    for size, n in (("s", 17), ("s", 18), ("m", 10), ("l", 5)):
        m = ref_models.TitaNet.get_titanet(n_mega_blocks=n, model_size=size)
        out[f"params.{size}{n}"] = np.asarray(int(m.get_n_params()))
    m = ref_models.TitaNet.get_titanet(n_mega_blocks=1, model_size="s", loss_function=ref_losses.CELoss(192, 251))
    out["params.s1.ce251"] = np.asarray(int(m.get_n_params()))
    np.savez_compressed(os.path.join(HERE, "sizing.npz"), **out)
    print({k: int(v) for k, v in out.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, case in CASES.items():
        if only and name not in only:
            continue
        run_case(name, case)
    if not only or "metrics" in only:
        metrics_golden()
    if not only or "sizing" in only:
        sizing_golden()
    if not only or "head_clamp" in only:
        head_clamp_golden()
