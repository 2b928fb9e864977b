"""GPU parity of the mel front end (tn_mel_forward) against the CPU oracle restatement
(oracle/mel_oracle.py; STFT pinned to torch.stft, torchaudio-defined pieces unpinned — see its header)."""
import numpy as np
import pytest
import torch

from oracle import mel_oracle as MO
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_samples", [16000, 24000, 48000, 3 * 1600 + 77, 700])
def test_mel_matches_oracle(n_samples):
    from titanet_amd.transforms import MelSpectrogram
    rng = np.random.default_rng(n_samples)
    B = 3
    waves = rng.normal(0, 0.05, (B, n_samples)).astype(np.float32)
    mel = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, specaugment_probability=0.0)
    out = mel.batch(torch.from_numpy(waves)).cpu().numpy()
    assert out.shape == (B, 80, 1 + n_samples // 160)
    for b in range(B):
        want = MO.mel_spectrogram(waves[b].astype(np.float64))
        assert rel_err(out[b], want) < 1e-3, rel_err(out[b], want)
        assert np.abs(out[b] - want).max() < 2e-3


def test_mel_masks_and_example_protocol():
    """the reference's per-example transform protocol incl. the whole SpecAugment branch (time stretch, 2 + 2 masks)"""
    import random
    from titanet_amd.transforms import MelSpectrogram
    rng = np.random.default_rng(5)
    wave = torch.from_numpy(rng.normal(0, 0.05, (1, 32000)).astype(np.float32))
    mel = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, specaugment_probability=1.0,
                         specaugment_freq_mask_num=2, specaugment_time_mask_num=2)
    random.seed(0); torch.manual_seed(0)
    ex = mel({"waveform": wave, "sample_rate": 16000, "speaker": 3})
    spec = ex["spectrogram"].cpu().numpy()
    # replay the same draws on the host and compare against the oracle with that rate and those masks
    random.seed(0); torch.manual_seed(0)
    random.random()
    rate = random.uniform(0.95, 1.05)
    T = mel.n_frames(32000, rate)
    assert spec.shape == (1, 80, T) and ex["speaker"] == 3 and T != 201
    fms = [MelSpectrogram._mask_bounds(80, 0.35 * 80) for _ in range(2)]
    tms = [MelSpectrogram._mask_bounds(T, 0.15 * T) for _ in range(2)]
    want = MO.mel_spectrogram(wave[0].numpy().astype(np.float64), rate=rate, freq_masks=fms, time_masks=tms)
    assert want.shape == (80, T)
    assert rel_err(spec[0], want) < 1e-3, rel_err(spec[0], want)
    for f in fms:
        assert (spec[0][f[0]:f[1]] == 0).all()
    for t in tms:
        assert (spec[0][:, t[0]:t[1]] == 0).all()


@pytest.mark.parametrize("rate", [0.95, 1.0, 1.05, 1.3])
def test_time_stretch_matches_vocoder_magnitudes(rate):
    from titanet_amd.transforms import MelSpectrogram
    rng = np.random.default_rng(11)
    wave = rng.normal(0, 0.05, (2, 20000)).astype(np.float32)
    mel = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, specaugment_probability=0.0)
    out = mel.batch(torch.from_numpy(wave), rates=[rate, rate]).cpu().numpy()
    for b in range(2):
        want = MO.mel_spectrogram(wave[b].astype(np.float64), rate=rate)
        assert out[b].shape == want.shape
        assert rel_err(out[b], want) < 1e-3, rel_err(out[b], want)


def test_ragged_batch_equals_each_utterance_alone():
    """variable-length batch (BASELINE configs[3]): zero-padded waveforms + lengths -> the collate_fn layout, every
    utterance with its own frame count and its own reflect padding, zeros beyond its end"""
    from titanet_amd.transforms import MelSpectrogram
    rng = np.random.default_rng(3)
    lens = [32000, 16000 + 77, 48000, 700]
    A = max(lens)
    waves = np.zeros((4, A), dtype=np.float32)
    for b, n in enumerate(lens):
        waves[b, :n] = rng.normal(0, 0.05, n)
    mel = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, specaugment_probability=0.0)
    out = mel.batch(torch.from_numpy(waves), lengths=lens).cpu().numpy()
    assert out.shape == (4, 80, 1 + A // 160)
    for b, n in enumerate(lens):
        t = 1 + n // 160
        want = MO.mel_spectrogram(waves[b, :n].astype(np.float64))
        assert rel_err(out[b, :, :t], want) < 1e-3
        assert (out[b, :, t:] == 0).all()
    # and the network consumes it with the matching frame lengths (padding mask)
    from titanet_amd import TitaNet
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s", device="cuda").eval()
    frames = torch.tensor([1 + n // 160 for n in lens])
    with torch.no_grad():
        e = m(torch.from_numpy(out).cuda(), lengths=frames)
        alone = m(torch.from_numpy(out[1:2, :, :frames[1]]).contiguous().cuda())
    assert float((e[1:2] - alone).abs().max()) < 1e-5


def test_mel_feeds_the_network():
    """waveform -> mel (GPU) -> TitaNet (GPU) without leaving the device (BASELINE config 4 plumbing)."""
    from titanet_amd import TitaNet
    from titanet_amd.transforms import MelSpectrogram
    mel = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, specaugment_probability=0.0)
    x = mel.batch(torch.randn(4, 24000) * 0.05)
    m = TitaNet.get_titanet(n_mega_blocks=1, model_size="s", device="cuda").eval()
    with torch.no_grad():
        e = m(x)
    assert e.shape == (4, 192) and torch.isfinite(e).all()
    assert torch.allclose(e.norm(dim=1), torch.ones(4, device=e.device), atol=1e-4)


def test_wave_per_frame_kernel_equals_the_generic_kernel(monkeypatch):
    """n_fft = 512 runs mel512_batch_kernel (one wave per frame, radix-8 passes); TN_MEL_GENERIC=1 at tn_mel_create keeps the
    generic radix-2 workgroup-per-frame kernel, which every other n_fft uses: same spectrograms on a ragged, time-stretched,
    masked batch (float32 rounding of two FFT factorizations apart), and the generic kernel still meets the oracle."""
    from titanet_amd.transforms import MelSpectrogram
    rng = np.random.default_rng(5)
    lens = [40000, 16000 + 77, 52000, 700, 31999]
    rates = [0.95, 1.0, 1.05, 1.0, 1.3]
    A = max(lens)
    waves = np.zeros((len(lens), A), dtype=np.float32)
    for b, n in enumerate(lens):
        waves[b, :n] = rng.normal(0, 0.05, n)
    fast = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, specaugment_probability=0.0)
    fast._mel()
    monkeypatch.setenv("TN_MEL_GENERIC", "1")
    slow = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, specaugment_probability=0.0)
    slow._mel()
    monkeypatch.delenv("TN_MEL_GENERIC")
    T = max(fast.n_frames(n, r) for n, r in zip(lens, rates))
    fm = torch.zeros(len(lens), 80, dtype=torch.bool); fm[1, 10:31] = True; fm[4, 70:] = True
    tm = torch.zeros(len(lens), T, dtype=torch.bool); tm[0, 5:40] = True; tm[2, 100:101] = True
    w = torch.from_numpy(waves)
    a = fast.batch(w, lengths=lens, rates=rates, freq_masks=fm, time_masks=tm).cpu().numpy()
    b = slow.batch(w, lengths=lens, rates=rates, freq_masks=fm, time_masks=tm).cpu().numpy()
    assert a.shape == b.shape == (len(lens), 80, T)
    assert np.abs(a - b).max() < 2e-4, np.abs(a - b).max()
    assert ((a == 0) == (b == 0)).all()
    # equal lengths, interval masks (tn_mel_forward)
    mk = torch.tensor([[3, 9, 10, 30]] * 2, dtype=torch.int32)
    a2 = fast.batch(w[:2, :32000], masks=mk).cpu().numpy()
    b2 = slow.batch(w[:2, :32000], masks=mk).cpu().numpy()
    assert np.abs(a2 - b2).max() < 2e-4 and ((a2 == 0) == (b2 == 0)).all()
    assert (a2[:, 3:9] == 0).all() and (a2[:, :, 10:30] == 0).all()
    want = MO.mel_spectrogram(waves[0, :lens[0]].astype(np.float64), rate=rates[0])
    got = slow.batch(w[:1, :lens[0]], rates=[rates[0]]).cpu().numpy()[0]
    assert rel_err(got, want) < 1e-3
