"""Eager (4 native calls, ~470 launches) vs one-hipGraph-per-step training step."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer


def run(B, T, prec, use_graph, steps=50):
    loss = LOSSES["ce"](192, 251, device="cuda")
    m = TitaNet.get_titanet(n_mega_blocks=17, model_size="s", loss_function=loss, dropout=0.1, device="cuda", precision=prec).train()
    tr = Trainer(m, use_graph=use_graph)
    x = torch.randn(B, 80, T, device="cuda") * 0.11 - 0.1
    y = torch.randint(0, 251, (B,), device="cuda")
    for _ in range(5):
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tr.step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"B={B} T={T} {prec} graph={use_graph}: {dt*1e3:.3f} ms/step  {B/dt:.0f} utt/s  loss {float(out[2]):.3f}", flush=True)


if __name__ == "__main__":
    for B, T in ((8, 300), (32, 300), (256, 300)):
        for g in (False, True):
            run(B, T, "bf16", g, steps=50 if B < 256 else 20)
