import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd import LOSSES, TitaNet

def run(fused, B, T, spw=None, exact=0):
    os.environ["TN_ASP_FUSED"] = "1" if fused else "0"
    os.environ["TN_ASP_EXACT"] = str(exact)
    if spw: os.environ["TN_ASP_SPW"] = str(spw)
    else: os.environ.pop("TN_ASP_SPW", None)
    torch.manual_seed(0)
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s", loss_function=LOSSES["ce"](192, 251, device="cuda"), dropout=0.1,
                            device="cuda", precision="bf16")
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(B, 80, T, generator=g) * 0.11 - 0.10).cuda()
    m.eval()
    with torch.no_grad():
        return m(x).float().cpu()

for B, T, spw, ex in [(256, 300, 1, 0), (256, 300, 2, 0), (256, 300, 3, 0), (256, 300, 6, 0), (256, 300, 3, 1), (256, 64, 3, 0), (256, 150, 3, 0), (256, 257, 3, 0), (256, 320, 3, 0), (64, 300, 3, 0), (64, 300, 6, 0), (256, 200, 6, 0)]:
    e1 = run(True, B, T, spw, ex)
    e0 = run(False, B, T)
    bad = (~torch.isfinite(e1)).any(dim=1).nonzero().flatten().tolist()
    ok = torch.isfinite(e1).all(dim=1)
    rel = float((e1[ok] - e0[ok]).norm() / e0[ok].norm())
    print(B, T, "spw", spw, "exact", ex, "rel(finite rows)", rel, "bad rows", bad[:12], len(bad), flush=True)
