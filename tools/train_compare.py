"""fp32 vs bf16 training on a learnable synthetic task (class-specific spectral templates + noise): both precisions
must drive the loss down at the same rate, although single bf16 forwards of a randomly initialised 17-block net are
noise-dominated."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer

NCLS, B, T = 32, 64, 201


def data(step, templates):
    g = torch.Generator().manual_seed(1000 + step)
    y = torch.randint(0, NCLS, (B,), generator=g)
    x = templates[y] * 0.05 + torch.randn(B, 80, T, generator=g) * 0.11 - 0.10
    return x.cuda(), y.cuda()


def run(prec, head, steps=150):
    torch.manual_seed(0)
    if head == "ce":
        loss = LOSSES["ce"](192, NCLS, device="cuda")
    else:
        loss = LOSSES["arc"](192, NCLS, device="cuda", scale=30, margin=0.2)
    m = TitaNet.get_titanet(n_mega_blocks=17, model_size="s", loss_function=loss, dropout=0.1, device="cuda", precision=prec).train()
    tr = Trainer(m, lr=1e-3)
    templates = torch.randn(NCLS, 80, 1, generator=torch.Generator().manual_seed(5)).expand(NCLS, 80, T).contiguous()
    hist, accs = [], []
    for s in range(steps):
        x, y = data(s, templates)
        _, preds, l = tr.step(x, y)
        hist.append(float(l)); accs.append(float((preds == y).float().mean()))
    k = 10
    print(f"{prec} {head}: loss first{k} {sum(hist[:k])/k:.3f} last{k} {sum(hist[-k:])/k:.3f}  acc last{k} {sum(accs[-k:])/k:.3f}", flush=True)


if __name__ == "__main__":
    for head in ("ce", "arc"):
        for prec in ("fp32", "bf16"):
            run(prec, head)
