"""Run a few train steps of one model size (for rocprofv3 --kernel-trace): python tools/profile_size.py m 10 bf16"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer
size, nb, prec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
loss = LOSSES["ce"](192, 251, device="cuda")
m = TitaNet.get_titanet(n_mega_blocks=nb, model_size=size, loss_function=loss, dropout=0.1, device="cuda", precision=prec).train()
tr = Trainer(m)
x = torch.randn(B, 80, 300, device="cuda") * 0.11 - 0.1
y = torch.randint(0, 251, (B,), device="cuda")
for _ in range(6):
    tr.step(x, y)
torch.cuda.synchronize()
