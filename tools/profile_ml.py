"""rocprofv3 target: a few train steps of TitaNet-M/10 or -L/5 (bf16, B=256, 80x300) — kernel-trace stats of the generic path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer
size, nb = sys.argv[1], int(sys.argv[2])
loss = LOSSES["ce"](192, 251, device="cuda")
m = TitaNet.get_titanet(n_mega_blocks=nb, model_size=size, loss_function=loss, dropout=0.1, device="cuda", precision=(sys.argv[3] if len(sys.argv) > 3 else "bf16")).train()
tr = Trainer(m)
x = torch.randn(256, 80, 300, device="cuda") * 0.11 - 0.1
y = torch.randint(0, 251, (256,), device="cuda")
if os.environ.get('POISON'):
    x[:] = float('nan')
for _ in range(6):
    out = tr.step(x, y)
torch.cuda.synchronize()
print('loss', float(out[2]), 'params finite', bool(torch.isfinite(m.flat_parameters()).all()))
