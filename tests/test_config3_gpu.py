"""BASELINE.json configs[3] at its stated shape: TitaNet-M width, ragged waveforms of U(2, 20) s at 16 kHz (frames up to
2001) -> on-GPU mel + SpecAugment (time stretch, frequency + time masks) -> zero-padded batch + lengths -> masked TRAIN
step (reference path: src/transforms.py:158-203 -> src/datasets.py:48-73 -> src/models.py:318-339 -> src/learn.py:95-117).

  * the front end of every utterance of the ragged batch against the mel oracle with that utterance's stretch rate and masks;
  * the masked train step of a 2-block M-width model fed with the GPU spectrograms against the float64 oracle (loss,
    embeddings, whole gradient);
  * TitaNet-M/10 (the configs[3] model), B = 32: eval embeddings of the padded batch == each utterance embedded alone
    (its own length, batch of one), and one bf16 train step is finite and changes with the lengths.
"""
import random

import numpy as np
import pytest
import torch

from oracle import mel_oracle as MO
from oracle import titanet_oracle as O
from tests.test_forward_gpu import build
from tests.util import case_state_dict, oracle_cfg, rel_err

pytestmark = pytest.mark.gpu
SR, HOP = 16000, 160


def ragged_waveforms(B, seed, lo=2.0, hi=20.0):
    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    nsamp = [int(rnd.uniform(lo, hi) * SR) for _ in range(B)]
    nsamp[0], nsamp[1] = int(hi * SR), int(lo * SR)                 # both ends of the range are present: T = 2001 and 201
    wav = torch.zeros(B, max(nsamp))
    for b, n in enumerate(nsamp):
        wav[b, :n] = torch.randn(n, generator=g) * 0.05
    return wav, nsamp, rnd


def front_end(wav, nsamp, rnd, stretch=True, into=None, align_frames=None):
    from titanet_amd.transforms import MelSpectrogram
    B = wav.shape[0]
    mel = MelSpectrogram(SR, n_fft=512, win_length=400, hop_length=HOP, n_mels=80)
    rates = [rnd.uniform(0.95, 1.05) if stretch else 1.0 for _ in range(B)]
    frames = [mel.n_frames(n, r) for n, r in zip(nsamp, rates)]
    T = max(frames)
    fms, tms = [], []
    fm = torch.zeros(B, 80, dtype=torch.bool)
    tm = torch.zeros(B, T, dtype=torch.bool)
    for b in range(B):
        f0 = rnd.randrange(0, 60); f1 = f0 + rnd.randrange(1, 28)            # <= 0.35 * 80 bins (parameters.yml:103-107)
        t0 = rnd.randrange(0, frames[b] - 4); t1 = min(frames[b], t0 + rnd.randrange(1, max(2, int(0.15 * frames[b]))))
        fm[b, f0:f1] = True; tm[b, t0:t1] = True
        fms.append((f0, f1)); tms.append((t0, t1))
    x = mel.batch(wav.cuda(), lengths=nsamp, rates=rates, freq_masks=fm, time_masks=tm, into=into, align_frames=align_frames)
    return x, frames, rates, fms, tms


def test_ragged_front_end_matches_the_mel_oracle():
    wav, nsamp, rnd = ragged_waveforms(8, seed=1)
    x, frames, rates, fms, tms = front_end(wav, nsamp, rnd)
    assert x.shape == (8, 80, max(frames)) and max(frames) >= 1900 and min(frames) <= 215
    xs = x.cpu().numpy()
    for b in (0, 1, 5):
        want = MO.mel_spectrogram(wav[b, :nsamp[b]].numpy().astype(np.float64), rate=rates[b], freq_masks=[fms[b]], time_masks=[tms[b]])
        assert want.shape == (80, frames[b])
        assert rel_err(xs[b, :, :frames[b]], want) < 2e-3, (b, rel_err(xs[b, :, :frames[b]], want))
        assert (xs[b, :, frames[b]:] == 0).all()                    # collate_fn layout: zeros beyond the utterance


def test_masked_train_step_from_gpu_spectrograms_vs_oracle():
    """M width (hidden 512, 7 taps), 2 mega blocks, 6 ragged utterances of 2 .. 20 s, fp32 plan vs the float64 oracle"""
    case = dict(cfg=dict(n_mels=80, n_mega_blocks=2, hidden=512, enc_out=1536, emb=192, kernel=7, attn_hidden=128),
                batch=6, frames=0, n_classes=20, seed=13)
    wav, nsamp, rnd = ragged_waveforms(6, seed=2)
    x, frames, *_ = front_end(wav, nsamp, rnd)
    lengths = torch.tensor(frames)
    y = torch.tensor([3, 1, 4, 1, 5, 9])
    m = build(case, "ce").train()
    emb, preds, lv = m(x, speakers=y.cuda(), lengths=lengths)
    lv.backward()
    torch.cuda.synchronize()
    sd = case_state_dict(case, "ce", torch.float64)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
    out = O.titanet_forward(sd, x.cpu().double(), oracle_cfg(case), training=True, speakers=y, lengths=lengths, loss="ce")
    out.loss.backward()
    named = dict(m.named_parameters())
    a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in named])
    b = np.concatenate([sd[k].grad.numpy().ravel() for k in named])
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    e_emb = rel_err(emb.detach().cpu().numpy(), out.normalized.detach().numpy())
    print(f"T up to {max(frames)}: emb {e_emb:.2e}, loss {float(lv):.5f} vs {float(out.loss):.5f}, gradient cosine {cos:.6f}")
    assert e_emb < 1e-3 and abs(float(lv) - float(out.loss)) < 1e-3 and cos > 0.9995


def test_m10_padded_batch_equals_each_utterance_alone_and_trains():
    from titanet_amd import LOSSES, TitaNet
    from titanet_amd.trainer import Trainer
    B = 32
    wav, nsamp, rnd = ragged_waveforms(B, seed=3)
    x, frames, *_ = front_end(wav, nsamp, rnd)
    assert max(frames) >= 1900
    lengths = torch.tensor(frames)
    torch.manual_seed(0)
    m = TitaNet.get_titanet(n_mega_blocks=10, model_size="m", loss_function=LOSSES["ce"](192, 251, device="cuda"), dropout=0.1,
                            device="cuda", precision="bf16")
    m.eval()
    with torch.no_grad():
        padded = m(x, lengths=lengths).clone()
        for b in (0, 1, 7, 19, 31):
            alone = m(x[b:b + 1, :, :frames[b]].contiguous())
            assert rel_err(padded[b:b + 1].cpu().numpy(), alone.cpu().numpy()) < 2e-2, (b, frames[b])
    m.train()
    tr = Trainer(m)
    y = torch.randint(0, 251, (B,), generator=torch.Generator().manual_seed(1)).cuda()
    l1 = float(tr.step(x, y, lengths=lengths)[2])
    l2 = float(tr.step(x, y, lengths=lengths)[2])
    assert np.isfinite(l1) and np.isfinite(l2) and torch.isfinite(m.flat_parameters()).all()
    assert torch.isfinite(m.flat_gradients()).all() and float(m.flat_gradients().norm()) > 0


def test_front_end_fused_into_the_prolog_operand():
    """configs[3] "mel + SpecAugment fused into the prolog kernel": MelSpectrogram.batch(..., into=model) writes the prolog
    conv's packed bf16 operand itself (tn_mel_forward_batch_packed -> tn_plan_prolog_input -> tn_forward_prepacked): same
    embeddings / loss / gradient as the float32 spectrogram tensor fed through model(x, lengths=...)."""
    from titanet_amd import LOSSES, TitaNet
    B = 12
    torch.manual_seed(0)
    m = TitaNet.get_titanet(n_mega_blocks=2, model_size="m", loss_function=LOSSES["ce"](192, 40, device="cuda"), dropout=0.1,
                            device="cuda", precision="bf16").train()
    y = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(1)).cuda()
    res = []
    for fused in (False, True):
        wav, nsamp, rnd = ragged_waveforms(B, seed=4, lo=2.0, hi=8.0)
        # (same frame axis on both paths: the dropout masks are a hash of the row index b * T + t)
        x, frames, *_ = front_end(wav, nsamp, rnd, into=m if fused else None, align_frames=1)
        m._seed_base, m._step = 11, 0
        m.zero_grad()
        if fused:
            assert type(x).__name__ == "PackedSpectrograms" and x.shape == (B, 80, max(frames))
            emb, _, lv = m(x, speakers=y)                       # the lengths ride in the handle
        else:
            emb, _, lv = m(x, speakers=y, lengths=torch.tensor(frames))
        lv.backward()
        torch.cuda.synchronize()
        res.append((emb.detach().cpu().numpy(), float(lv), m.flat_gradients().clone().cpu().numpy()))
    e = rel_err(res[1][0], res[0][0])
    cos = float(res[1][2] @ res[0][2] / (np.linalg.norm(res[1][2]) * np.linalg.norm(res[0][2])))
    print(f"fused vs tensor path: emb {e:.2e}, loss {res[1][1]:.5f} / {res[0][1]:.5f}, gradient cosine {cos:.6f}")
    # identical bf16 operand bits; what differs is the summation order of the atomics-accumulated statistics
    assert e < 2e-2 and abs(res[1][1] - res[0][1]) < 2e-2 and cos > 0.995
    # by default the packed path pads the frame axis to a multiple of 256 (every utterance starts on a row-tile boundary):
    # the padding is invisible behind the lengths mask (eval: no dropout hash of the row index)
    m.eval()
    with torch.no_grad():
        wav, nsamp, rnd = ragged_waveforms(B, seed=4, lo=2.0, hi=8.0)
        xa, frames, *_ = front_end(wav, nsamp, rnd, into=m)
        assert xa.shape == (B, 80, -(-max(frames) // 256) * 256) and xa.shape[2] > max(frames)
        ea = m(xa).cpu().numpy()
        wav, nsamp, rnd = ragged_waveforms(B, seed=4, lo=2.0, hi=8.0)
        xb, *_ = front_end(wav, nsamp, rnd, into=m, align_frames=1)
        eb = m(xb).cpu().numpy()
    assert rel_err(ea, eb) < 2e-3, rel_err(ea, eb)
    # eval, equal lengths, the f32 twin of the packed operand
    from titanet_amd.transforms import MelSpectrogram
    mel = MelSpectrogram(SR, n_fft=512, win_length=400, hop_length=HOP, n_mels=80)
    w = torch.randn(4, 32000, generator=torch.Generator().manual_seed(2)) * 0.05
    with torch.no_grad():
        a = m(mel.batch(w.cuda())).cpu().numpy()
        b = m(mel.batch(w.cuda(), into=m)).cpu().numpy()
    assert rel_err(b, a) < 1e-6, rel_err(b, a)


def test_prefetching_a_packed_batch_invalidates_a_pending_backward():
    """MelSpectrogram.batch(into=model) overwrites the plan's prolog operand, which backward re-reads for the prolog weight
    gradient: a batch prefetched between forward and backward must make that backward raise, not use the wrong input."""
    from titanet_amd import LOSSES, TitaNet
    from titanet_amd.transforms import MelSpectrogram
    B = 4
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s", loss_function=LOSSES["ce"](192, 10, device="cuda"), dropout=0.0,
                            device="cuda", precision="bf16").train()
    mel = MelSpectrogram(SR, n_fft=512, win_length=400, hop_length=HOP, n_mels=80)
    w = (torch.randn(B, 32000, generator=torch.Generator().manual_seed(2)) * 0.05).cuda()
    y = torch.arange(B).cuda()
    _, _, lv = m(mel.batch(w, into=m), speakers=y)
    nxt = mel.batch(w * 0.5, into=m)                      # same shape, same mode: the same plan
    with pytest.raises(RuntimeError, match="saved activations"):
        lv.backward()
    _, _, lv2 = m(nxt, speakers=y)                        # the prefetched batch itself is consumed normally
    lv2.backward()
    assert torch.isfinite(m.flat_gradients()).all()
    # a batch of one utterance in training mode: the intended error, not an AttributeError on the packed handle
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        m(mel.batch(w[:1], into=m), speakers=y[:1])
