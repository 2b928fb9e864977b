import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_full_size_gpu import make, batch, rel, B
x, y = batch()
perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()
res = {}
for prec in ("fp32", "bf16"):
    m = make(prec).train()
    with torch.no_grad():
        e1 = m(x, speakers=y)[0].clone()
        e1b = m(x, speakers=y)[0].clone()
        e2 = m(x[perm].contiguous(), speakers=y[perm].contiguous())[0].clone()
    res[prec] = (e1, e2)
    print(prec, "rerun", rel(e1b, e1), "perm", rel(e2, e1[perm]))
    for name in ("prolog_out", "block_out:0", "block_out:4", "block_out:16"):
        C = 256
        with torch.no_grad():
            m(x, speakers=y); a = m.debug_fetch(name, (B, C, 300)).clone()
            m(x[perm].contiguous(), speakers=y[perm].contiguous()); b = m.debug_fetch(name, (B, C, 300)).clone()
        print("   ", name, rel(b, a[perm]))
print("bf16 vs fp32", rel(res["bf16"][0], res["fp32"][0]), "perm", rel(res["bf16"][1], res["fp32"][1]))
