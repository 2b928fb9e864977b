import os, sys, torch, numpy as np
sys.path.insert(0, ".")
from titanet_amd import LOSSES, TitaNet
torch.manual_seed(0)
def run(B, T, nb, mode):
    g = torch.Generator().manual_seed(1)
    lengths = torch.randint(T // 10, T, (B,), generator=g); lengths[0] = T
    if mode == "full": lengths[:] = T
    x = torch.randn(B, 80, T, generator=g) * 0.11 - 0.1
    for b in range(B): x[b, :, lengths[b]:] = 0
    if mode == "generic": os.environ["TN_GENERIC"] = "1"
    m = TitaNet.get_titanet(n_mega_blocks=nb, model_size="m", loss_function=LOSSES["ce"](192, 251, device="cuda"), dropout=0.1, device="cuda", precision="bf16").train()
    y = torch.randint(0, 251, (B,), generator=g).cuda()
    emb, preds, lv = m(x.cuda(), speakers=y, lengths=None if mode == "nomask" else lengths)
    os.environ.pop("TN_GENERIC", None)
    lv.backward(); torch.cuda.synchronize()
    bad = [k for k, p in m.named_parameters() if not torch.isfinite(p.grad).all()]
    print(B, T, nb, mode, "loss", float(lv.detach()), "non-finite grads:", len(bad), bad[:6], bad[-3:], flush=True)
    for k in bad[:3] + bad[-2:]:
        gk = dict(m.named_parameters())[k].grad
        print("   ", k, tuple(gk.shape), "non-finite elems", int((~torch.isfinite(gk)).sum()), "of", gk.numel())
for (B, T, nb, mode) in ((32, 1969, 2, "nomask"), (64, 1000, 2, "nomask"), (32, 1969, 2, "fast"), (32, 1969, 10, "fast"), (32, 1969, 10, "nomask"), (256, 300, 10, "nomask")):
    run(B, T, nb, mode)
