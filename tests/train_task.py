"""A learnable synthetic speaker task for the "trains the same" tests (TEST INFRASTRUCTURE).

Speakers are points z_s of a low-dimensional latent space; an utterance of speaker s is the spectral template
``U z_s`` (U: n_mels x latent, fixed), slowly modulated over time, buried in the dB-mel-like noise of the benchmark inputs
(``randn * 0.11 - 0.10``).  Fresh noise every step (nothing to memorise), close speakers stay confusable: the loss and the
accuracy settle at a noise-limited plateau that is a property of the task, not of one trajectory — which is what makes
"fp32 and bf16 reach the same plateau" a testable statement.  Held-out speakers (new z) give a verification EER through
``titanet_amd.metrics.verification_test`` — the reference's test protocol (reference src/learn.py:409-459).
"""
import math

import torch

from titanet_amd import LOSSES, TitaNet, metrics
from titanet_amd.trainer import Trainer

N_MELS, LATENT = 80, 6


class SpeakerTask:
    def __init__(self, n_train=32, n_heldout=12, sig=0.02, seed=5):
        g = torch.Generator().manual_seed(seed)
        self.U = torch.randn(N_MELS, LATENT, generator=g) / math.sqrt(LATENT)
        self.z_train = torch.randn(n_train, LATENT, generator=g)
        self.z_held = torch.randn(n_heldout, LATENT, generator=g)
        self.sig = sig

    def utterances(self, z, T, g, device="cpu"):
        """z: [B, LATENT] -> [B, N_MELS, T] float32 (on `device`, drawn from generator `g` of that device)"""
        B = z.shape[0]
        tmpl = (z @ self.U.t()).unsqueeze(2).to(device)                        # [B, 80, 1]
        t = torch.arange(T, dtype=torch.float32, device=device).view(1, 1, T)
        phase = torch.rand(B, 1, 1, generator=g, device=device) * 2 * math.pi
        mod = 1.0 + 0.5 * torch.sin(2 * math.pi * t / 50.0 + phase)           # slow amplitude modulation, random phase
        return tmpl * mod * self.sig + torch.randn(B, N_MELS, T, generator=g, device=device) * 0.11 - 0.10

    def batch(self, step, B, T, stream=0, device="cuda"):
        """the training batch of `step` on data stream `stream`: drawn ON the device (a host-side randn of 1 M values per step
        would be most of a 1200-step run's wall time); labels from a host generator"""
        gh = torch.Generator().manual_seed(1000 + step + 1000003 * stream)
        y = torch.randint(0, self.z_train.shape[0], (B,), generator=gh)
        g = torch.Generator(device=device).manual_seed(1000 + step + 1000003 * stream)
        return self.utterances(self.z_train[y], T, g, device), y

    def heldout(self, per_speaker=6, seed=77):
        """variable-length utterances of speakers the training never saw (list of [80, T_i]), speaker ids"""
        g = torch.Generator().manual_seed(seed)
        specs, spk = [], []
        for s in range(self.z_held.shape[0]):
            for _ in range(per_speaker):
                T = int(torch.randint(150, 301, (1,), generator=g))
                specs.append(self.utterances(self.z_held[s:s + 1], T, g)[0])
                spk.append(s)
        return specs, spk


def train_and_verify(task, precision, size="s", n_blocks=17, head="ce", steps=300, B=64, T=201, dropout=0.1, lr=1e-3, tail=20, stream=0):
    """``steps`` fused-Adam steps of the reference's step protocol (reference src/learn.py:88-135) on ``task``; returns the
    mean loss / training accuracy of the last ``tail`` steps and the verification metrics of the held-out speakers."""
    torch.manual_seed(0)
    n_cls = task.z_train.shape[0]
    if head == "ce":
        loss = LOSSES["ce"](192, n_cls, device="cuda")
    else:
        loss = LOSSES["arc"](192, n_cls, device="cuda", scale=30, margin=0.2)
    m = TitaNet.get_titanet(n_mega_blocks=n_blocks, model_size=size, loss_function=loss, dropout=dropout, device="cuda",
                            precision=precision).train()
    tr = Trainer(m, lr=lr)
    hist, accs = [], []
    for s in range(steps):
        x, y = task.batch(s, B, T, stream)
        y = y.cuda()
        _, preds, l = tr.step(x, y)
        hist.append(l)
        accs.append((preds == y).float().mean())
    hist = [float(v) for v in hist]
    accs = [float(v) for v in accs]
    specs, spk = task.heldout()
    ver, _, _ = metrics.verification_test(m, specs, spk, batch_size=36)
    finite = bool(torch.isfinite(m.flat_parameters()).all())
    return {"precision": precision, "stream": stream, "loss_first": sum(hist[:tail]) / tail, "loss_last": sum(hist[-tail:]) / tail,
            "acc_last": sum(accs[-tail:]) / tail, "eer": ver["test/eer"], "mindcf": ver["test/mindcf"], "params_finite": finite}
