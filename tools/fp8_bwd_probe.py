"""The fp8 data gradient of the fp8 plan against its bf16 backward (TN_FP8_BWD=0), TitaNet-L/5 and -M/10: runs of the same
forward (same weights, batch, dropout stream) with (i) the bf16 backward, (ii) dS rounded through e4m3 with one power-of-two
scale per row but multiplied in bf16 (TN_FP8_BWD_EMU=1: the experiment that preceded the kernels), (iii) the built path — e4m3
dS rows x e4m3 W^T rows on the f8f6f4 MFMA (default), (iv) bf16 again = the run-to-run noise floor (atomics); compared tensor
by tensor.     python tools/fp8_bwd_probe.py"""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    import torch
    from titanet_amd import LOSSES, TitaNet
    size, nb = sys.argv[2], int(sys.argv[3])
    torch.manual_seed(0)
    m = TitaNet.get_titanet(n_mega_blocks=nb, model_size=size, loss_function=LOSSES["ce"](192, 251, device="cuda"), dropout=0.1, device="cuda", precision="fp8").train()
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(128, 80, 300, generator=g) * 0.11 - 0.1).cuda(); y = torch.randint(0, 251, (128,), generator=g).cuda()
    m._seed_base, m._step = 5, 0
    emb, _, lv = m(x, speakers=y); lv.backward(); torch.cuda.synchronize()
    np.savez(sys.argv[1], **{k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()})
    sys.exit(0)
for size, nb in (("l", 5), ("m", 10)):
    runs = {}
    for tag, env in (("bf16", {"TN_FP8_BWD": "0"}), ("e4m3_emulated", {"TN_FP8_BWD_EMU": "1"}), ("fp8_built", {}), ("bf16_again", {"TN_FP8_BWD": "0"})):
        subprocess.run([sys.executable, __file__, f"/tmp/fp8p_{tag}.npz", size, str(nb)], check=True, env=dict(os.environ, **env))
        runs[tag] = dict(np.load(f"/tmp/fp8p_{tag}.npz"))
    def dist(a, b):
        ga = np.concatenate([a[k].ravel() for k in b]); gb = np.concatenate([b[k].ravel() for k in b])
        cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
        per = {k: float(np.linalg.norm(a[k] - b[k]) / (np.linalg.norm(b[k]) + 1e-30)) for k in b if b[k].size >= 16384}
        return cos, per
    for tag in ("e4m3_emulated", "fp8_built", "bf16_again"):
        cos, per = dist(runs[tag], runs["bf16"])
        blk = [max(v for k, v in per.items() if f"mega_blocks.{i}." in k) for i in range(nb)]
        print(f"TitaNet-{size.upper()}/{nb} fp8 plan, backward {tag} vs bf16: whole-gradient cosine {cos:.5f}; worst large tensor per mega block (first .. last)",
              [round(v, 3) for v in blk], "prolog", round(per.get("encoder.prolog.conv_block.0.weight", 0.0), 3))
