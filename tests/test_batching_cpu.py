"""Batch assembly (reference src/datasets.py:48-73) and RandomChunk (reference src/transforms.py:206-233) semantics.
The reference's datasets module does not import here (SyntaxError in src/datasets.py:36-44), so the
checks are against the documented contract on hand-made cases."""
import random

import torch

from titanet_amd.datasets import collate_fn
from titanet_amd.transforms import RandomChunk, Resample


def test_collate_pads_right_with_zeros_and_keeps_dtypes():
    g = torch.Generator().manual_seed(0)
    batch = [{"spectrogram": torch.randn(1, 80, t, generator=g, dtype=torch.float64), "speaker_id": s}
             for t, s in ((151, 3), (301, 7), (201, 3))]
    spec, lengths, speakers = collate_fn(batch)
    assert spec.shape == (3, 80, 301) and spec.dtype == torch.float32
    assert lengths.dtype == torch.int64 and lengths.tolist() == [151, 301, 201]
    assert speakers.dtype == torch.int64 and speakers.tolist() == [3, 7, 3]
    for i, e in enumerate(batch):
        t = lengths[i]
        assert torch.equal(spec[i, :, :t], e["spectrogram"][0].float())
        assert torch.count_nonzero(spec[i, :, t:]) == 0


def test_random_chunk_draws_and_bounds():
    sr = 16000
    ex = {"waveform": torch.arange(5 * sr, dtype=torch.float32).reshape(1, -1), "sample_rate": sr, "speaker_id": 1}
    random.seed(123)
    out = RandomChunk(3, [1.5, 2, 3])(ex)
    random.seed(123)
    length = random.choice([1.5, 2, 3])
    start = random.randint(0, 5 * sr - int(length * sr))
    assert out["waveform"].shape == (1, int(length * sr))
    assert out["waveform"][0, 0].item() == float(start)          # contiguous slice starting at `start`
    assert ex["waveform"].shape == (1, 5 * sr)                    # input not modified
    short = {"waveform": torch.zeros(1, 2 * sr), "sample_rate": sr}
    assert RandomChunk(3, [1.5, 2, 3])(short)["waveform"].shape == (1, 2 * sr)   # <= max_length: untouched
    # chunk lengths map to the frame counts the model sees: 1 + A // 160
    assert [1 + int(l * sr) // 160 for l in (1.5, 2, 3)] == [151, 201, 301]


def test_resample_identity_only():
    ex = {"waveform": torch.zeros(1, 10), "sample_rate": 16000}
    assert Resample(16000)(ex)["waveform"].shape == (1, 10)
    try:
        Resample(8000)(ex)
    except NotImplementedError:
        return
    raise AssertionError("expected NotImplementedError")
