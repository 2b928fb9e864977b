"""Step time of the train step (Trainer.step) and eval forward for the model sizes / heads BASELINE.json's configs name."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def run(size, nblocks, prec, head, B=256, T=300, steps=10):
    if head == "ce":
        loss = LOSSES["ce"](192, 251, device="cuda")
    else:
        loss = LOSSES["arc"](192, 251, device="cuda", scale=30, margin=0.2)
    m = TitaNet.get_titanet(n_mega_blocks=nblocks, model_size=size, loss_function=loss, dropout=0.1, device="cuda", precision=prec).train()
    tr = Trainer(m)
    x = torch.randn(B, 80, T, device="cuda") * 0.11 - 0.1
    y = torch.randint(0, 251, (B,), device="cuda")
    dt = timed(lambda: tr.step(x, y), steps)
    m.eval()
    with torch.no_grad():
        de = timed(lambda: m(x), steps)
    print(f"{size}/{nblocks} {prec} {head} B={B} T={T}: train {dt*1e3:.2f} ms/step {B/dt:.0f} utt/s | eval fwd {de*1e3:.2f} ms {B/de:.0f} utt/s", flush=True)
    del m, tr
    torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1:            # e.g. `python tools/size_bench.py m l`: only the named sizes, bf16
        for sz in sys.argv[1:]:           # "l" -> bf16, "l:fp8" -> the fp8 pointwise plan
            name, _, prec = sz.partition(":")
            run(name, {"s": 17, "m": 10, "l": 5}[name], prec or "bf16", "ce")
        sys.exit(0)
    run("s", 17, "bf16", "ce")
    run("s", 17, "bf16", "arc")
    run("s", 17, "fp32", "ce")
    run("m", 10, "bf16", "ce")
    run("m", 10, "fp32", "ce", steps=5)
    run("l", 5, "bf16", "ce")
    run("l", 5, "fp32", "ce", steps=5)
    run("s", 17, "bf16", "ce", B=8, steps=30)
    run("s", 17, "bf16", "ce", B=1024, steps=5)
