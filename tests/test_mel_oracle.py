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
