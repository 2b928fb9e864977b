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
    import random
    from titanet_amd.transforms import MelSpectrogram
    rng = np.random.default_rng(5)
    wave = torch.from_numpy(rng.normal(0, 0.05, (1, 32000)).astype(np.float32))
    mel = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, specaugment_probability=1.0)
    random.seed(0); torch.manual_seed(0)
    ex = mel({"waveform": wave, "sample_rate": 16000, "speaker": 3})
    spec = ex["spectrogram"].cpu().numpy()
    assert spec.shape == (1, 80, 201) and ex["speaker"] == 3
    # replay the same draws on the host and compare against the oracle with those masks
    random.seed(0); torch.manual_seed(0)
    random.random(); random.uniform(0.95, 1.05)
    f = MelSpectrogram._mask_bounds(80, 0.35 * 80)
    t = MelSpectrogram._mask_bounds(201, 0.15 * 201)
    want = MO.mel_spectrogram(wave[0].numpy().astype(np.float64), freq_mask=f, time_mask=t)
    assert rel_err(spec[0], want) < 1e-3
    assert (spec[0][f[0]:f[1]] == 0).all() and (spec[0][:, t[0]:t[1]] == 0).all()


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
