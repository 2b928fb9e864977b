"""S/17 at trained weights: bf16 plan vs fp32 plan, and the fp32 plan's own sensitivity to ONE bf16 rounding of its input /
of its weights (how much of the bf16 plan's gradient distance is the network amplifying a perturbation, not the kernels).
    python tools/depth_probe.py [steps]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_forward_gpu import build
from tests.test_trained_parity_gpu import _task, P, SEED, NCLS
from tests.util import rel_err
from titanet_amd.trainer import Trainer

NB = 17
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
case = dict(cfg=dict(n_mels=80, n_mega_blocks=NB, hidden=256, enc_out=1536, emb=192, kernel=3, attn_hidden=128),
            batch=64, frames=120, n_classes=NCLS, seed=33)
m32 = build(case, "ce", precision="fp32", dropout=P).train()
m32._seed_base, m32._step = 20240918, 0
tr = Trainer(m32, lr=1e-3)
for step in range(steps):
    x, y = _task(64, 120, 2000 + step % 8)
    lv = tr.step(x.cuda(), y.cuda())[2]
    if step == 0:
        first = float(lv)
print("trained", first, "->", float(lv))
sd = {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
del tr, m32
B, T = 128, 200
x, y = _task(B, T, 5151)

def run(prec, xin, sdict, drop=P):
    m = build(dict(case, batch=B, frames=T), "ce", precision=prec, dropout=drop).train()
    m.load_state_dict(sdict)
    m._seed_base, m._step = SEED, 0
    emb, _, lv = m(xin.cuda(), speakers=y.cuda())
    blocks = [m.debug_fetch(f"block_out:{i}", (B, 256, T)).cpu() for i in range(NB)]
    lv.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
    return blocks, emb.detach().cpu().numpy(), float(lv), g

def report(tag, a, b):
    errs = [float((p - q).norm() / q.norm()) for p, q in zip(a[0], b[0])]
    ga = np.concatenate([a[3][k].ravel() for k in b[3]]); gb = np.concatenate([b[3][k].ravel() for k in b[3]])
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    w = [rel_err(a[3][f"encoder.mega_blocks.{i}.sub_blocks.2.conv_block.0.conv.1.weight"], b[3][f"encoder.mega_blocks.{i}.sub_blocks.2.conv_block.0.conv.1.weight"]) for i in range(NB)]
    print(f"[{tag}] loss {a[2]:.4f}/{b[2]:.4f} block errs", [f"{e:.4f}" for e in errs], f"emb {rel_err(a[1], b[1]):.2e} grad cos {cos:.5f} wgrad per block", [f"{e:.3f}" for e in w])

for drop in (P, 0.0):
    ref = run("fp32", x, sd, drop)
    report(f"bf16 vs fp32, dropout {drop}", run("bf16", x, sd, drop), ref)
    xb = x.to(torch.bfloat16).float()
    report(f"fp32 with the INPUT rounded to bf16 vs fp32, dropout {drop}", run("fp32", xb, sd, drop), ref)
    sdb = {k: (v.to(torch.bfloat16).float() if (v.dtype == torch.float32 and v.dim() >= 2) else v) for k, v in sd.items()}
    report(f"fp32 with the WEIGHT MATRICES rounded to bf16 vs fp32, dropout {drop}", run("fp32", x, sdb, drop), ref)
    report(f"fp32 twice (run-to-run), dropout {drop}", run("fp32", x, sd, drop), ref)
