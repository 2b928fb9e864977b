"""rocprofv3 target: what do utterance boundaries cost the specialised kernels?  Same 76800 rows as 256 x 300 and as 2 x 38400."""
import sys, torch
sys.path.insert(0, ".")
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer
B, T = int(sys.argv[1]), int(sys.argv[2])
m = TitaNet.get_titanet(n_mega_blocks=17, model_size="s", loss_function=LOSSES["ce"](192, 251, device="cuda"), dropout=0.1, device="cuda", precision="bf16").train()
tr = Trainer(m)
x = torch.randn(B, 80, T, device="cuda") * 0.11 - 0.1
y = torch.randint(0, 251, (B,), device="cuda")
for _ in range(6):
    tr.step(x, y)
torch.cuda.synchronize()
