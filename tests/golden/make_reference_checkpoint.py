#!/usr/bin/env python3
"""A checkpoint WRITTEN BY THE REFERENCE's objects in the reference's layout (src/learn.py:188-195: ``{"epoch", "model",
"optimizer", "lr_scheduler"}`` through torch.save), plus the reference's parameters after ONE MORE Adam step on the case's
deterministic batch — the resume test's known answer.  Build container only (imports /root/reference/src):

    python tests/golden/make_reference_checkpoint.py  ->  ref_checkpoint_tiny.pth, ref_checkpoint_next.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
import torch  # noqa: E402

from tests.golden.cases import CASES  # noqa: E402
from tests.golden.make_golden import build_reference  # noqa: E402
from tests.util import case_inputs  # noqa: E402

case = CASES["tiny_k3"]
model = build_reference(case, "ce", torch.float32).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0)       # reference src/train.py:130-135
x, y = case_inputs(case, torch.float32)
for _ in range(2):                                                           # reference src/learn.py:95-117
    _, _, loss = model(x, speakers=y)
    opt.zero_grad()
    loss.backward()
    opt.step()
torch.save({"epoch": 2, "model": model.state_dict(), "optimizer": opt.state_dict(), "lr_scheduler": dict()},
           os.path.join(HERE, "ref_checkpoint_tiny.pth"))
_, _, loss = model(x, speakers=y)
opt.zero_grad()
loss.backward()
opt.step()
np.savez_compressed(os.path.join(HERE, "ref_checkpoint_next.npz"), loss=np.float64(loss.item()),
                    **{k: v.detach().numpy() for k, v in model.state_dict().items()})
print("loss at the resumed step", float(loss))
