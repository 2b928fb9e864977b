"""debug: fused SE + combine vs the two-kernel form, train mode, intermediates (GPU box)"""
import os, sys
import torch
sys.path.insert(0, ".")

def run(fused, nb, B, T, masked, pd=0.1):
    from titanet_amd import LOSSES, TitaNet
    os.environ["TN_SE_FUSED"] = "1" if fused else "0"
    torch.manual_seed(5)
    m = TitaNet.get_titanet(n_mega_blocks=nb, model_size="s", loss_function=LOSSES["ce"](192, 40, device="cuda"), dropout=pd,
                            device="cuda", precision="bf16")
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(B, 80, T, generator=g) * 0.11 - 0.1).cuda()
    y = torch.randint(0, 40, (B,), generator=g).cuda()
    lengths = None
    if masked:
        lengths = torch.randint(1, T + 1, (B,), generator=g); lengths[0] = T
    m.train(); m._seed_base, m._step = 777, 0
    emb, _, loss = m(x, speakers=y, lengths=lengths)
    out = {}
    for i in range(nb):
        out[f"gate{i}"] = m.debug_fetch(f"se_gate:{i}", (B, 256)).cpu()
        out[f"blk{i}"] = m.debug_fetch(f"block_out:{i}", (B, 256, T)).cpu()
    out["emb"] = emb.detach().float().cpu()
    return out, lengths

for nb, B, T, masked, pd in [(1, 64, 320, False, 0.1)]:
    print("dropout", pd)
    a, L = run(False, nb, B, T, masked, pd)
    a2, _ = run(False, nb, B, T, masked, pd)
    b, _ = run(True, nb, B, T, masked, pd)
    for k in a:
        d = (a[k] - b[k]).abs(); d0 = (a[k] - a2[k]).abs()
        print(nb, B, T, masked, k, "max diff fused", float(d.max()), "rerun", float(d0.max()), "nonzero", int((d > 0).sum()), "of", d.numel())
        if k.startswith("blk") and float(d.max()) > 0:
            idx = (d > 0).nonzero()
            print("  first diffs (b, c, t):", idx[:12].tolist())
            bs = sorted(set(idx[:, 0].tolist())); ts = sorted(set(idx[:, 2].tolist())); cs = sorted(set(idx[:, 1].tolist()))
            print("  utterances", bs[:20], "frames", ts[:40], "channels", cs[:40], len(cs))
            if L is not None: print("  lengths", L[:20].tolist())
            A, Bv = a[k][d > 0], b[k][d > 0]
            print("  unfused==0:", int((A == 0).sum()), " fused==0:", int((Bv == 0).sum()), " both nonzero:", int(((A != 0) & (Bv != 0)).sum()))
            print("  per element-in-vector c%8:", [int((idx[:, 1] % 8 == i).sum()) for i in range(8)])
            print("  per vector lane c//8:", [int((idx[:, 1] // 8 == i).sum()) for i in range(32)])
            print("  per row phase t%16:", [int((idx[:, 2] % 16 == i).sum()) for i in range(16)])
            big = (d > 0.1)
            bi = big.nonzero()
            vec = (bi[:, 0] * 100000 + bi[:, 2]) * 32 + bi[:, 1] // 8
            uniq, cnt = torch.unique(vec, return_counts=True)
            print("  large diffs:", int(big.sum()), "in", len(uniq), "vectors; elements per affected vector histogram:", torch.bincount(cnt, minlength=9).tolist())
            for j in range(0, min(len(bi), 400), 20):
                bb, cc, tt = bi[j].tolist()
                c8 = cc // 8 * 8
                print("   ", (bb, cc, tt), "unfused", [round(float(v), 3) for v in a[k][bb, c8:c8 + 8, tt]], "fused", [round(float(v), 3) for v in b[k][bb, c8:c8 + 8, tt]])
            nzv = ((a[k] != 0) | (b[k] != 0))
            print("  fraction of nonzero outputs overall:", float(nzv.float().mean()))
            # for affected vectors: is the fused vector equal to the unfused vector of another row (stale row)?
            us = (idx[:, 2] // 16).tolist()
            print("  diffs per 16-row step u:", [us.count(u) for u in range(20)])
