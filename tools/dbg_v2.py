import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden.cases import CASES
from tests.test_forward_gpu import build
from tests.util import case_inputs, load_golden, rel_err
case = CASES["s17_b8"]
for mode in (sys.argv[1:] or ["train", "eval"]):
    m = build(case, "ce", precision="bf16")
    m.train(mode == "train")
    x, y = case_inputs(case, torch.float32)
    with torch.no_grad():
        out = m(x.cuda(), speakers=y.cuda()) if mode == "train" else m(x.cuda())
    torch.cuda.synchronize()
    print(mode, "ok", flush=True)
