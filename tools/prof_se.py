"""rocprofv3 target: where does se_squeeze_fc_kernel's time go?  eval forward of TitaNet-M at T = 300 and T = 600 (same batch)."""
import sys, torch
sys.path.insert(0, ".")
from titanet_amd import TitaNet
T = int(sys.argv[1]); mode = sys.argv[2]
m = TitaNet.get_titanet(n_mega_blocks=10, model_size="m", device="cuda", precision="bf16")
m = m.eval() if mode == "eval" else m.train()
x = torch.randn(256, 80, T, device="cuda") * 0.11 - 0.1
with torch.no_grad():
    for _ in range(4):
        if mode == "eval": m(x)
        else: m(x, speakers=None) if False else None
torch.cuda.synchronize()
