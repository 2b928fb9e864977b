"""fp32 vs bf16 (and fp8 vs bf16 at L width) training on the synthetic speaker task of tests/train_task.py: loss / accuracy
plateau and held-out verification EER per precision.  The `-m gpu` form with assertions is tests/test_train_compare_gpu.py;
this script sweeps the task difficulty (`sig`) and prints the table.

    python tools/train_compare.py [sig ...]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.train_task import SpeakerTask, train_and_verify  # noqa: E402

if __name__ == "__main__":
    sigs = [float(a) for a in sys.argv[1:]] or [0.02]
    for sig in sigs:
        task = SpeakerTask(sig=sig)
        for kw in (dict(precision="fp32"), dict(precision="bf16"), dict(precision="bf16", head="arc"), dict(precision="fp32", head="arc")):
            r = train_and_verify(task, **kw)
            r.update({"sig": sig, "head": kw.get("head", "ce")})
            print(json.dumps(r), flush=True)
        for prec in ("bf16", "fp8"):
            r = train_and_verify(task, precision=prec, size="l", n_blocks=2)
            r.update({"sig": sig, "model": "L/2"})
            print(json.dumps(r), flush=True)
