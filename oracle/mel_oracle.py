"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference mel front end.

Reference: ``MelSpectrogram.__call__`` (/root/reference/src/transforms.py:158-203) with the
constructor of :118-156 and the parameters ``get_transforms`` passes (:57-74; parameters.yml:79-85):
``Spectrogram(n_fft=512, win_length=400, hop_length=160, power=None)`` -> ``abs().pow(2)`` ->
``MelScale(n_mels=80, sample_rate=16000, n_stft=257)`` -> ``AmplitudeToDB()`` ->
``F.normalize(dim=1)`` -> (SpecAugment) one frequency mask and one time mask.

The arithmetic lives in **torchaudio 0.13.0** (init/requirements.txt:4), which is NOT vendored in
/root/reference and NOT installed in this image: **parity unpinned** for the torchaudio-defined pieces
(mel filter bank, AmplitudeToDB constants, mask_along_axis).  They are restated here from torchaudio's
published definitions (``torchaudio.functional.melscale_fbanks`` HTK scale / norm=None,
``amplitude_to_DB`` with multiplier 10, amin 1e-10, ref 1.0, top_db None; ``mask_along_axis``).  The
STFT itself IS pinned: ``tests/test_mel_oracle.py`` checks it against ``torch.stft`` with the exact
argument set torchaudio's Spectrogram uses (hann periodic window zero-padded to n_fft, center=True,
reflect padding, onesided).  The torchaudio-defined pieces are additionally held to hand-computed known
answers of their published formulas (tests/test_mel_oracle.py): HTK mel points, triangle values and their
partition of unity, the dB constants, the mask index arithmetic, the phase vocoder's time grid.

The SpecAugment time stretch (torchaudio.transforms.TimeStretch -> functional.phase_vocoder,
transforms.py:168-175) is followed by ``.abs().pow(2)`` in the reference (:177), so only the vocoder's
MAGNITUDE path reaches the output: mag[j] = alpha_j |S[idx_j + 1]| + (1 - alpha_j) |S[idx_j]| on the time
grid j * rate (the accumulated phase is discarded by the abs) — ``time_stretch_power`` below.
"""
import numpy as np


def hann_window_periodic(win_length):
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def padded_window(n_fft, win_length):
    """torch.stft centres a shorter window inside n_fft (left pad (n_fft - win_length) // 2)."""
    w = np.zeros(n_fft, dtype=np.float64)
    left = (n_fft - win_length) // 2
    w[left:left + win_length] = hann_window_periodic(win_length)
    return w


def n_frames(n_samples, hop_length):
    return 1 + n_samples // hop_length


def stft_power(wave, n_fft=512, win_length=400, hop_length=160):
    """|STFT|^2, [n_fft//2+1, frames]; center=True with reflect padding (torch.stft semantics)."""
    wave = np.asarray(wave, dtype=np.float64)
    pad = n_fft // 2
    x = np.pad(wave, (pad, pad), mode="reflect")
    T = n_frames(len(wave), hop_length)
    w = padded_window(n_fft, win_length)
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(T)[:, None]
    frames = x[idx] * w[None, :]
    spec = np.fft.rfft(frames, axis=1)            # [T, n_fft//2+1]
    return (spec.real ** 2 + spec.imag ** 2).T


def stft_magnitude(wave, n_fft=512, win_length=400, hop_length=160):
    return np.sqrt(stft_power(wave, n_fft, win_length, hop_length))


def time_stretch_power(mag, rate):
    """|phase_vocoder(S, rate)|^2 for S with magnitudes ``mag`` [n_freq, T] (torchaudio.functional.phase_vocoder):
    time_steps = arange(0, T, rate); idx = floor(time_steps); alpha = time_steps % 1; S is zero-padded by two frames;
    mag_out = alpha * |S[idx + 1]| + (1 - alpha) * |S[idx]|.  rate == 1 returns mag^2 unchanged (torchaudio short-cuts)."""
    if rate == 1.0:
        return mag ** 2
    T = mag.shape[1]
    steps = np.arange(0, T, rate, dtype=np.float64)
    idx = np.floor(steps).astype(np.int64)
    alpha = steps - idx
    padded = np.concatenate([mag, np.zeros((mag.shape[0], 2))], axis=1)
    out = alpha[None, :] * padded[:, idx + 1] + (1.0 - alpha[None, :]) * padded[:, idx]
    return out ** 2


def melscale_fbanks(n_freqs=257, f_min=0.0, f_max=8000.0, n_mels=80, sample_rate=16000):
    """HTK mel triangles, norm=None (torchaudio.functional.melscale_fbanks definition): [n_freqs, n_mels]."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def mel_spectrogram(wave, sample_rate=16000, n_fft=512, win_length=400, hop_length=160, n_mels=80,
                    freq_mask=None, time_mask=None, rate=None, freq_masks=(), time_masks=()):
    """[n_mels, frames] float64: power mel spectrogram in dB, L2-normalised over the mel axis per frame,
    then optional masks given as (start, end) index pairs (reference transforms.py:177-201).
    rate: SpecAugment time-stretch rate (frames become ceil(T / rate)); freq_masks / time_masks: any number of
    (start, end) intervals applied in sequence (specaugment_*_mask_num > 1, transforms.py:189-201)."""
    if rate is None or rate == 1.0:
        power = stft_power(wave, n_fft, win_length, hop_length)
    else:
        power = time_stretch_power(stft_magnitude(wave, n_fft, win_length, hop_length), rate)
    fb = melscale_fbanks(n_fft // 2 + 1, 0.0, sample_rate / 2.0, n_mels, sample_rate)
    mel = fb.T @ power                                            # [n_mels, T]
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))                  # AmplitudeToDB(power, ref=1, top_db=None)
    norm = np.sqrt((db ** 2).sum(axis=0, keepdims=True))
    out = db / np.maximum(norm, 1e-12)                            # F.normalize(dim=1) on [1, M, T]
    if freq_mask is not None:
        out[freq_mask[0]:freq_mask[1], :] = 0.0
    if time_mask is not None:
        out[:, time_mask[0]:time_mask[1]] = 0.0
    for f in freq_masks:
        out[f[0]:f[1], :] = 0.0
    for t in time_masks:
        out[:, t[0]:t[1]] = 0.0
    return out


def mask_along_axis_bounds(size, mask_param, u_value, u_start):
    """torchaudio.functional.mask_along_axis index arithmetic for given uniforms u in [0,1):
    value = u_value*mask_param; min_value = u_start*(size - value); [long(min_value), long(min_value)+long(value))."""
    value = u_value * mask_param
    min_value = u_start * (size - value)
    start = int(min_value)
    return start, start + int(value)
