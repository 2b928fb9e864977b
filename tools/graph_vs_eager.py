"""Step time at the bench shape: eager launches vs one hipGraph per step (tuning tool)."""
import time, torch, sys
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
for use_graph in (False, True):
    torch.manual_seed(0)
    loss = LOSSES["ce"](192, 251, device=dev)
    model = TitaNet.get_titanet(embedding_size=192, n_mels=80, n_mega_blocks=17, model_size="s", attention_hidden_size=128,
                                loss_function=loss, dropout=0.1, device=dev, precision="bf16").train()
    tr = Trainer(model, lr=1e-3, use_graph=use_graph)
    x = (torch.randn(B, 80, 300) * 0.11 - 0.1).to(dev); y = torch.randint(0, 251, (B,)).to(dev)
    for _ in range(6): tr.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tr.step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"batch {B} graph={use_graph}: {dt * 1e3:.3f} ms/step")
    del tr, model
