#!/usr/bin/env python3
"""state_dict key order / shapes / dtypes of the REAL reference (Wadaboa/titanet), written to state_dict_keys.json.
Run in the build container only (imports /root/reference/src):  python tests/golden/make_state_dict_keys.py"""
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
import losses as ref_losses  # noqa: E402  (reference)
import models as ref_models  # noqa: E402  (reference)

out = {}
for name, kw, loss in (
        ("s2_none", dict(n_mega_blocks=2, model_size="s"), None),
        ("s2_ce251", dict(n_mega_blocks=2, model_size="s"), ref_losses.CELoss(192, 251)),
        ("s2_arc251", dict(n_mega_blocks=2, model_size="s"), ref_losses.ArcFaceLoss(192, 251, scale=30, margin=0.2)),
        ("m1_none", dict(n_mega_blocks=1, model_size="m"), None),
        ("l1_none", dict(n_mega_blocks=1, model_size="l"), None),
        ("s1_simple_pool", dict(n_mega_blocks=1, model_size="s", simple_pool=True), None)):
    m = ref_models.TitaNet.get_titanet(loss_function=loss, **kw)
    out[name] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "state_dict_keys.json"), "w") as fh:
    json.dump(out, fh, indent=0)
print({k: len(v) for k, v in out.items()})
