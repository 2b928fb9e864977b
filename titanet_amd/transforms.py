"""Mel front end with the reference's transform surface, computed on the GPU.

Mirrors ``transforms.MelSpectrogram`` (reference src/transforms.py:111-203): same constructor arguments,
same ``__call__(example)`` contract (``example["waveform"]`` ``[1, A]`` -> ``new_example["spectrogram"]``
``[1, n_mels, 1 + A // hop]``), same Python-``random`` / ``torch.rand`` draws for the SpecAugment
decisions and the mask bounds of ``torchaudio.functional.mask_along_axis``.  The arithmetic is the HIP
kernel behind ``tn_mel_forward`` (include/titanet_amd.h).  Deviation: the phase-vocoder ``TimeStretch`` of
the SpecAugment branch (src/transforms.py:168-175) is not implemented — the stretch rate is drawn (to keep
the random stream aligned) and ignored.  ``Resample`` (src/transforms.py:320-341) is the identity at the
target rate and refuses anything else.
"""
import ctypes as C
import random

import torch

from . import _lib
from ._lib import check


def copy_example(example):
    """reference src/transforms.py:12-22"""
    return {k: (torch.clone(v) if isinstance(v, torch.Tensor) else v) for k, v in example.items()}


class MelSpectrogram:
    def __init__(self, sample_rate, n_fft=400, win_length=None, hop_length=None, n_mels=128,
                 specaugment_min_speed=0.95, specaugment_max_speed=1.05, specaugment_freq_mask_ratio=0.35,
                 specaugment_freq_mask_num=1, specaugment_time_mask_ratio=0.15, specaugment_time_mask_num=1,
                 specaugment_probability=1.0, device="cuda"):
        self.sample_rate, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
        self.win_length = win_length if win_length is not None else n_fft            # torchaudio defaults
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        self.specaugment_min_speed = specaugment_min_speed
        self.specaugment_max_speed = specaugment_max_speed
        self.specaugment_freq_mask_ratio = specaugment_freq_mask_ratio
        self.specaugment_freq_mask_num = specaugment_freq_mask_num
        self.specaugment_time_mask_ratio = specaugment_time_mask_ratio
        self.specaugment_time_mask_num = specaugment_time_mask_num
        self.specaugment_probability = specaugment_probability
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._handle = None

    def _mel(self):
        if self._handle is None:
            h = C.c_void_p()
            check(self._lib.tn_mel_create(self.sample_rate, self.n_fft, self.win_length, self.hop_length, self.n_mels,
                                          C.byref(h)), "tn_mel_create")
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                self._lib.tn_mel_destroy(self._handle)
        except Exception:
            pass

    @staticmethod
    def _mask_bounds(size, mask_param):
        """index arithmetic of torchaudio.functional.mask_along_axis (one torch.rand per quantity)"""
        value = torch.rand(1) * mask_param
        min_value = torch.rand(1) * (size - value)
        start = int(min_value.long())
        return start, start + int(value.long())

    def batch(self, waveforms, masks=None):
        """[B, A] float waveforms (equal length) -> [B, n_mels, 1 + A // hop] on the GPU.
        masks: optional int32 [B, 4] = (f_start, f_end, t_start, t_end)."""
        if waveforms.dim() != 2:
            raise ValueError("expected waveforms of shape [B, A]")
        if not torch.cuda.is_available():
            raise RuntimeError("titanet_amd.transforms.MelSpectrogram needs a ROCm device; there is no CPU execution path")
        w = waveforms.to(device=self.device, dtype=torch.float32).contiguous()
        B, A = w.shape
        T = 1 + A // self.hop_length
        out = torch.empty(B, self.n_mels, T, dtype=torch.float32, device=self.device)
        m = None
        if masks is not None:
            m = masks.to(device=self.device, dtype=torch.int32).contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        vp = C.c_void_p
        check(self._lib.tn_mel_forward(self._mel(), vp(w.data_ptr()), B, A, vp(m.data_ptr() if m is not None else 0),
                                       vp(out.data_ptr()), vp(stream)), "tn_mel_forward")
        return out

    def __call__(self, example):
        assert isinstance(example, dict) and "waveform" in example, "Wrong input structure"
        new_example = copy_example(example)
        wave = new_example["waveform"]
        apply_specaugment = random.random() < self.specaugment_probability
        masks = None
        if apply_specaugment:
            random.uniform(self.specaugment_min_speed, self.specaugment_max_speed)   # time-stretch rate: drawn, not applied
            T = 1 + wave.shape[-1] // self.hop_length
            f0 = f1 = t0 = t1 = 0
            # the reference applies `num` masks in sequence; with the default num = 1 that is one interval per axis
            for _ in range(self.specaugment_freq_mask_num):
                f0, f1 = self._mask_bounds(self.n_mels, self.specaugment_freq_mask_ratio * self.n_mels)
            for _ in range(self.specaugment_time_mask_num):
                t0, t1 = self._mask_bounds(T, self.specaugment_time_mask_ratio * T)
            if self.specaugment_freq_mask_num > 1 or self.specaugment_time_mask_num > 1:
                raise NotImplementedError("more than one mask per axis is not supported by tn_mel_forward yet")
            masks = torch.tensor([[f0, f1, t0, t1]], dtype=torch.int32)
        new_example["spectrogram"] = self.batch(wave.reshape(1, -1), masks)
        return new_example


class Resample:
    """reference src/transforms.py:320-341: identity at the target rate (LibriSpeech is 16 kHz)."""

    def __init__(self, sample_rate):
        self.sample_rate = sample_rate

    def __call__(self, example):
        assert isinstance(example, dict) and "waveform" in example and "sample_rate" in example, "Wrong input structure"
        if example["sample_rate"] != self.sample_rate:
            raise NotImplementedError("resampling is outside the MI355X hot-path scope (SURVEY.md §8a14)")
        return copy_example(example)


class RandomChunk:
    """reference src/transforms.py:206-233: if the waveform is longer than ``max_length`` seconds, keep a random
    chunk of one of ``lengths`` seconds (same ``random.choice`` / ``random.randint`` draws, in the same order)."""

    def __init__(self, max_length, lengths):
        self.max_length = max_length
        self.lengths = lengths

    def __call__(self, example):
        assert isinstance(example, dict) and "waveform" in example and "sample_rate" in example, "Wrong input structure"
        new_example = copy_example(example)
        num_samples = new_example["waveform"].size(-1)
        if num_samples / new_example["sample_rate"] > self.max_length:
            length = random.choice(self.lengths)
            samples = int(length * new_example["sample_rate"])
            start = random.randint(0, num_samples - samples)
            new_example["waveform"] = new_example["waveform"][:, start:start + samples]
        return new_example


def get_transforms(enabled, rir_corpora_path=None, max_length=3, chunk_lengths=(1.5, 2, 3), min_speed=0.95, max_speed=1.05,
                   sample_rate=16000, n_fft=512, win_length=25, hop_length=10, n_mels=80, freq_mask_ratio=0.35,
                   freq_mask_num=1, time_mask_ratio=0.15, time_mask_num=1, probability=1.0, device="cuda", training=True):
    """reference src/transforms.py:24-74: Resample -> [RandomChunk] -> MelSpectrogram (window / hop given in ms).
    "reverb" (sox + RIR corpora, src/transforms.py:236-317) is outside the hot-path scope and is refused."""
    enabled = list(enabled or [])
    if "reverb" in enabled and training:
        raise NotImplementedError("reverb augmentation is outside the MI355X hot-path scope (SURVEY.md §8f)")
    transformations = [Resample(sample_rate)]
    if "chunk" in enabled:
        transformations.append(RandomChunk(max_length, list(chunk_lengths)))
    transformations.append(MelSpectrogram(
        sample_rate, n_fft=n_fft, win_length=int(win_length / 1000 * sample_rate), hop_length=int(hop_length / 1000 * sample_rate),
        n_mels=n_mels, specaugment_min_speed=min_speed, specaugment_max_speed=max_speed,
        specaugment_freq_mask_ratio=freq_mask_ratio, specaugment_freq_mask_num=freq_mask_num,
        specaugment_time_mask_ratio=time_mask_ratio, specaugment_time_mask_num=time_mask_num,
        specaugment_probability=(probability if "specaugment" in enabled and training else 0.0), device=device))
    return transformations
