"""Does the benchmarked precision TRAIN to the same quality as the parity path?  (`metric` says "EER
parity"; the reference's verification protocol is src/learn.py:409-459, its step protocol src/learn.py:88-135.)

TitaNet-S at full depth (17 mega blocks, dropout 0.1 — the benchmarked model) is trained for 300 fused-Adam steps on the
synthetic speaker task of tests/train_task.py (fresh noise every step, confusable speakers: the loss settles at a
noise-limited plateau) once in fp32 — the path that is bit-close to the reference (tests/test_forward_gpu.py,
test_backward_gpu.py) — and once in bf16, from the same initial weights on the same data stream.  Asserted: the mean loss of
the last 20 steps within 10 % relative, the training accuracy within 2 points, and the verification EER of HELD-OUT
speakers (utterances of 150-300 frames, embedded through metrics.verification_test, all ordered pairs) within 1 point
absolute.  The same for the fp8 plan against the bf16 plan at TitaNet-L width (2 mega blocks).
"""
import json
import os

import pytest

from tests.train_task import SpeakerTask, train_and_verify

pytestmark = pytest.mark.gpu

PROFILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "train_compare.jsonl")


def _log(rows):
    try:
        os.makedirs(os.path.dirname(PROFILE), exist_ok=True)
        with open(PROFILE, "a") as fh:
            for r in rows:
                fh.write(json.dumps(r) + "\n")
    except OSError:
        pass


def _compare(a, b, what):
    print(what)
    for r in (a, b):
        print("   ", json.dumps(r))
    _log([dict(a, what=what), dict(b, what=what)])
    assert a["params_finite"] and b["params_finite"]
    # both must have learnt the task (far below the untrained loss, far above chance) for the comparison to mean anything
    for r in (a, b):
        assert r["loss_last"] < 0.6 * r["loss_first"], r
        assert r["eer"] < 0.35, r
    assert abs(a["loss_last"] - b["loss_last"]) <= 0.10 * max(a["loss_last"], b["loss_last"]), (a["loss_last"], b["loss_last"])
    assert abs(a["acc_last"] - b["acc_last"]) <= 0.02, (a["acc_last"], b["acc_last"])
    assert abs(a["eer"] - b["eer"]) <= 0.01, (a["eer"], b["eer"])


@pytest.mark.parametrize("head", ["ce", "arc"])
def test_bf16_trains_like_fp32_at_full_depth(head):
    task = SpeakerTask()
    a = train_and_verify(task, "fp32", size="s", n_blocks=17, head=head, steps=300)
    b = train_and_verify(task, "bf16", size="s", n_blocks=17, head=head, steps=300)
    _compare(a, b, f"TitaNet-S/17 {head}: fp32 vs bf16, 300 steps")


def test_fp8_trains_like_bf16_at_l_width():
    task = SpeakerTask()
    a = train_and_verify(task, "bf16", size="l", n_blocks=2, steps=300)
    b = train_and_verify(task, "fp8", size="l", n_blocks=2, steps=300)
    _compare(a, b, "TitaNet-L/2 ce: bf16 vs fp8, 300 steps")
