"""Small driver for rocprofv3: a few full fwd+bwd steps of TitaNet-S/17, B=256, T=300."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd import LOSSES, TitaNet

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
loss = LOSSES["ce"](192, 251, device="cuda")
m = TitaNet.get_titanet(n_mega_blocks=17, model_size="s", loss_function=loss, dropout=0.1, device="cuda", precision=prec).train()
x = torch.randn(256, 80, 300, device="cuda") * 0.11 - 0.1
y = torch.randint(0, 251, (256,), device="cuda")
for _ in range(steps):
    e, p, l = m(x, speakers=y)
    l.backward()
torch.cuda.synchronize()
print("loss", l.item())
