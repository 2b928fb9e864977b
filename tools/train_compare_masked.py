"""Training on a learnable synthetic task with VARIABLE-LENGTH batches (zero-padded + lengths): the fast masked kernels and the
generic masked templates (TN_GENERIC=1) must drive the loss down the same way.  Class-specific spectral templates + noise,
utterance lengths U(T/4, T); the template only lives in the valid frames, the padding holds a constant that must be ignored.
    python tools/train_compare_masked.py [size] [blocks] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer

NCLS, B, T = 24, 48, 260


def data(step, templates):
    g = torch.Generator().manual_seed(1000 + step)
    y = torch.randint(0, NCLS, (B,), generator=g)
    ln = torch.randint(T // 4, T + 1, (B,), generator=g)
    ln[int(torch.randint(0, B, (1,), generator=g))] = T
    x = templates[y] * 0.05 + torch.randn(B, 80, T, generator=g) * 0.11 - 0.10
    for b in range(B):
        x[b, :, ln[b]:] = 3.0                       # whatever the padding holds must not matter
    return x.cuda(), y.cuda(), ln


def run(size, blocks, generic, steps):
    if generic:
        os.environ["TN_GENERIC"] = "1"
    else:
        os.environ.pop("TN_GENERIC", None)
    torch.manual_seed(0)
    m = TitaNet.get_titanet(n_mega_blocks=blocks, model_size=size, loss_function=LOSSES["ce"](192, NCLS, device="cuda"), dropout=0.1,
                            device="cuda", precision="bf16").train()
    os.environ.pop("TN_GENERIC", None)
    tr = Trainer(m, lr=1e-3)
    templates = torch.randn(NCLS, 80, 1, generator=torch.Generator().manual_seed(5)).expand(NCLS, 80, T).contiguous()
    hist, accs = [], []
    for s in range(steps):
        x, y, ln = data(s, templates)
        emb, preds, loss = tr.step(x, y, lengths=ln)
        hist.append(float(loss)); accs.append(float((preds == y).float().mean()))
    return hist, accs, bool(torch.isfinite(m.flat_parameters()).all())


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "s"
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    res = {name: run(size, blocks, name == "generic", steps) for name in ("generic", "fast")}
    for k in range(0, steps, max(1, steps // 8)):
        print(f"step {k:4d}  " + "  ".join(f"{n}: loss {res[n][0][k]:.3f} acc {res[n][1][k]:.2f}" for n in res))
    tail = max(5, steps // 10)
    for n in res:
        print(f"{n:8s} last-{tail} mean loss {sum(res[n][0][-tail:]) / tail:.4f}  acc {sum(res[n][1][-tail:]) / tail:.3f}  finite {res[n][2]}")


if __name__ == "__main__":
    main()
