"""Ad-hoc timing of forward (and backward when available) on the GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd import LOSSES, TitaNet

def run(prec, B=256, T=300, steps=10, train=True, bwd=False):
    loss = LOSSES["ce"](192, 251, device="cuda")
    m = TitaNet.get_titanet(n_mega_blocks=17, model_size="s", loss_function=loss, dropout=0.1, device="cuda", precision=prec)
    m.train(train)
    x = torch.randn(B, 80, T, device="cuda") * 0.11 - 0.1
    y = torch.randint(0, 251, (B,), device="cuda")
    def step():
        if bwd:
            e, p, l = m(x, speakers=y); l.backward()
        else:
            with torch.no_grad():
                m(x, speakers=y)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"prec={prec} B={B} T={T} train={train} bwd={bwd}: {dt*1e3:.2f} ms/step  {B/dt:.0f} utt/s", flush=True)

if __name__ == "__main__":
    bwd = "--bwd" in sys.argv
    for prec in ("bf16", "fp32"):
        run(prec, bwd=bwd)
