import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from titanet_amd import LOSSES, TitaNet
import ast
cfg = ast.literal_eval(sys.argv[1]) if len(sys.argv) > 1 else {'B': 25, 'T': 151, 'p': 0.0, 'head': 'ce', 'blocks': 1, 'ncls': 45, 'simple': False, 'train': True, 'wseed': 329160110, 'xseed': 976519419, 'size': 's', 'masked': True}
def run(generic):
    if generic: os.environ["TN_GENERIC"] = "1"
    else: os.environ.pop("TN_GENERIC", None)
    torch.manual_seed(cfg["wseed"])
    lf = LOSSES["ce"](192, cfg["ncls"], device="cuda") if cfg["head"] == "ce" else LOSSES["arc"](192, cfg["ncls"], device="cuda", scale=30, margin=0.2)
    m = TitaNet.get_titanet(n_mega_blocks=cfg["blocks"], model_size=cfg["size"], loss_function=lf, dropout=cfg["p"], device="cuda", precision="bf16", simple_pool=cfg["simple"])
    m._seed_base, m._step = 777, 0
    g = torch.Generator().manual_seed(cfg["xseed"])
    x = (torch.randn(cfg["B"], 80, cfg["T"], generator=g) * 0.11 - 0.1).cuda()
    y = torch.randint(0, cfg["ncls"], (cfg["B"],), generator=g).cuda()
    lengths = None
    if cfg["masked"]:
        lengths = torch.randint(1, cfg["T"] + 1, (cfg["B"],), generator=g)
        lengths[int(torch.randint(0, cfg["B"], (1,), generator=g))] = cfg["T"]
    m.train()
    emb, preds, loss = m(x, speakers=y, lengths=lengths)
    loss.backward(); torch.cuda.synchronize()
    return {k: p.grad.detach().float().cpu().numpy().copy() for k, p in m.named_parameters()}, lengths
g0, ln = run(True); g1, _ = run(False)
print("lengths", None if ln is None else ln.tolist())
rows = []
for k in g0:
    a, b = g1[k].ravel(), g0[k].ravel()
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    cos = float(a @ b / (na * nb + 1e-30)); rel = float(np.linalg.norm(a - b) / (nb + 1e-30))
    rows.append((rel, cos, k, nb))
for rel, cos, k, nb in sorted(rows, reverse=True)[:14]:
    print(f"{k:70s} rel {rel:.3e} cos {cos:.4f} |g| {nb:.3e}")
a = np.concatenate([g1[k].ravel() for k in g0]); b = np.concatenate([g0[k].ravel() for k in g0])
print("total cos", float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))))
