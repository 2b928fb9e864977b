"""Eval forward (the verification path: padded batches of 64 + lengths) masked vs unmasked, per model size."""
import sys, time, torch
sys.path.insert(0, ".")
from titanet_amd import TitaNet
def ms(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for size, nb in (("s", 17), ("m", 10), ("l", 5)):
    for (B, T) in ((64, 1000), (256, 300)):
        m = TitaNet.get_titanet(n_mega_blocks=nb, model_size=size, device="cuda", precision="bf16").eval()
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(B, 80, T, generator=g) * 0.11 - 0.1).cuda()
        ln = torch.randint(T // 5, T, (B,), generator=g); ln[0] = T
        with torch.no_grad():
            a = ms(lambda: m(x)); b = ms(lambda: m(x, lengths=ln))
        print(f"{size}/{nb} eval B={B} T={T}: unmasked {a:.2f} ms, masked {b:.2f} ms ({int(ln.sum())} of {B*T} frames valid)", flush=True)
        del m

# ---- train step (fwd + bwd + Adam), masked vs unmasked
from titanet_amd import LOSSES
from titanet_amd.trainer import Trainer
for size, nb in (("s", 17), ("m", 10)):
    B, T = 256, 300
    m = TitaNet.get_titanet(n_mega_blocks=nb, model_size=size, loss_function=LOSSES["ce"](192, 251, device="cuda"), dropout=0.1, device="cuda", precision="bf16").train()
    tr = Trainer(m)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(B, 80, T, generator=g) * 0.11 - 0.1).cuda()
    y = torch.randint(0, 251, (B,), generator=g).cuda()
    ln = torch.randint(T // 5, T, (B,), generator=g); ln[0] = T
    a = ms(lambda: tr.step(x, y)); b = ms(lambda: tr.step(x, y, lengths=ln))
    print(f"{size}/{nb} train B={B} T={T}: unmasked {a:.2f} ms, masked {b:.2f} ms ({int(ln.sum())} of {B*T} frames valid), params finite {bool(torch.isfinite(m.flat_parameters()).all())}", flush=True)
    del m, tr
