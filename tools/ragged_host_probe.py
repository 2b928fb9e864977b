import sys, time, torch, random
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer
dev = torch.device("cuda", 0)
mode = sys.argv[1]
if mode in ("s_first", "s_first_keep"):
    loss = LOSSES["ce"](192, 251, device=dev)
    model = TitaNet.get_titanet(embedding_size=192, n_mels=80, n_mega_blocks=17, model_size="s", attention_hidden_size=128,
                                loss_function=loss, dropout=0.1, device=dev, precision="bf16").train()
    tr = Trainer(model, lr=1e-3, n_buckets=2)
    x = (torch.randn(256, 80, 300) * 0.11 - 0.10).to(dev); y = torch.randint(0, 251, (256,)).to(dev)
    for _ in range(10): tr.step(x, y)
    torch.cuda.synchronize()
    if mode == "s_first":
        del tr, model
        torch.cuda.empty_cache()
import bench as B
# replicate ragged_m with host timestamps
orig = B._timed_steps
def timed(fn, warm, steps):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(mode, "step ms", round(dt * 1e3, 2), "host ms per call", [round(t * 1e3, 2) for t in ts])
    return dt
B._timed_steps = timed
print(B.other_configs(dev, ["m10_ragged_mel_specaug_masked"])["m10_ragged_mel_specaug_masked"].get("ms_per_step"))
