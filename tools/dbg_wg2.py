import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if len(sys.argv) > 1:
    from tests.test_forward_gpu import build
    from tests.util import case_inputs
    case = dict(cfg=dict(n_mels=80, n_mega_blocks=3, hidden=256, enc_out=512, emb=64, kernel=3, attn_hidden=64),
                batch=48, frames=120, n_classes=40, seed=7)
    m = build(case, "ce", precision="bf16").train()
    x, y = case_inputs(case, torch.float32)
    m(x.cuda(), speakers=y.cuda())[2].backward()
    np.savez(sys.argv[1], **{k: v.grad.detach().cpu().numpy() for k, v in m.named_parameters()})
else:
    for v in ("3", "7"):
        subprocess.run([sys.executable, __file__, f"/tmp/g{v}.npz"], env=dict(os.environ, TN_V2=v), check=True)
    a, b = np.load("/tmp/g3.npz"), np.load("/tmp/g7.npz")
    for k in a.files:
        e = np.linalg.norm(a[k] - b[k]) / max(np.linalg.norm(a[k]), 1e-20)
        if e > 0.3:
            print(f"{k:70s} rel diff {e:.3e}  |a|={np.linalg.norm(a[k]):.3e} |b|={np.linalg.norm(b[k]):.3e}")
