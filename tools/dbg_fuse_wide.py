import os, sys, subprocess, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from titanet_amd import LOSSES, TitaNet
    torch.manual_seed(0)
    size = sys.argv[2]
    m = TitaNet.get_titanet(n_mega_blocks=2, model_size=size, loss_function=LOSSES["ce"](192, 20, device="cuda"), dropout=0.1, device="cuda", precision="bf16").train()
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(12, 80, 150, generator=g) * 0.11 - 0.1).cuda(); y = torch.randint(0, 20, (12,), generator=g).cuda()
    m._seed_base, m._step = 5, 0
    emb, _, lv = m(x, speakers=y); lv.backward(); torch.cuda.synchronize()
    np.savez(sys.argv[1], **{k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()})
    sys.exit(0)
for size in ("m", "l"):
    subprocess.run([sys.executable, __file__, "/tmp/a.npz", size], check=True)
    subprocess.run([sys.executable, __file__, "/tmp/b.npz", size], check=True, env=dict(os.environ, TN_DBG_NO_FUSE_WIDE="1"))
    a, b = np.load("/tmp/a.npz"), np.load("/tmp/b.npz")
    for k in a.files:
        e = float(np.linalg.norm(a[k] - b[k]) / (np.linalg.norm(b[k]) + 1e-30))
        if e > 2e-2: print(size, k, a[k].shape, round(e, 4))
    print(size, "done")
