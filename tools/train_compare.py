"""fp32 vs bf16 (and fp8 vs bf16 at L width) training on the synthetic speaker task of tests/train_task.py: loss / accuracy
plateau and held-out verification EER per precision, with a second data stream per precision as the yardstick (how far two
runs of the SAME precision end up from each other).  The `-m gpu` form with assertions is tests/test_train_compare_gpu.py.

    python tools/train_compare.py [--sig S] [--steps N] [--train K] [--held H] [--what s17,l2]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.train_task import SpeakerTask, train_and_verify  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sig", type=float, nargs="+", default=[0.02])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--train", type=int, default=32)
    ap.add_argument("--held", type=int, default=12)
    ap.add_argument("--what", default="s17,l2")
    ap.add_argument("--heads", default="ce")
    args = ap.parse_args()
    for sig in args.sig:
        task = SpeakerTask(n_train=args.train, n_heldout=args.held, sig=sig)
        if "s17" in args.what:
            for head in args.heads.split(","):
                for prec in ("fp32", "bf16"):
                    for stream in (0, 1):
                        r = train_and_verify(task, precision=prec, head=head, steps=args.steps, stream=stream)
                        r.update({"sig": sig, "head": head, "model": "S/17", "steps": args.steps, "train": args.train, "held": args.held})
                        print(json.dumps(r), flush=True)
        if "l2" in args.what:
            for prec in ("bf16", "fp8"):
                for stream in (0, 1):
                    r = train_and_verify(task, precision=prec, size="l", n_blocks=2, steps=args.steps, stream=stream)
                    r.update({"sig": sig, "model": "L/2", "steps": args.steps, "train": args.train, "held": args.held})
                    print(json.dumps(r), flush=True)
