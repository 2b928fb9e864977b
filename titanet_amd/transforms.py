"""Mel front end with the reference's transform surface, computed on the GPU.

Mirrors ``transforms.MelSpectrogram`` (reference src/transforms.py:111-203): same constructor arguments,
same ``__call__(example)`` contract (``example["waveform"]`` ``[1, A]`` -> ``new_example["spectrogram"]``
``[1, n_mels, 1 + A // hop]``), same Python-``random`` / ``torch.rand`` draws for the SpecAugment
decisions and the mask bounds of ``torchaudio.functional.mask_along_axis`` (any number of masks per axis).
The arithmetic is the HIP kernel behind ``tn_mel_forward_batch`` (include/titanet_amd.h), including the
SpecAugment time stretch: the reference squares the magnitude of the phase vocoder's output right away
(src/transforms.py:173-177), so the vocoder's magnitude interpolation on the stretched time grid is what
reaches the spectrogram, and that is what the kernel computes.  ``Resample`` (src/transforms.py:320-341) is
the identity at the target rate and refuses anything else.
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
        self._ring, self._ring_i, self._marks = [None] * 32, 0, None

    def _upload(self, parts):
        """Small per-batch host arrays (lengths, rates, masks) -> device tensors through ONE asynchronous copy from a pinned
        staging slot (a ring of 32, each guarded by a host-visible completion word).  ``tensor.to(device)`` from pageable memory blocks the calling
        thread until the stream has drained — one sleep / wake-up round trip per array and step, which on a busy host cost more
        than the whole network step (configs[3] leg: 26 -> 30-70 ms).  ``parts``: (source tensor, dtype, shape) — the source
        is converted and zero-padded to ``shape`` INSIDE the pinned slot (no temporaries: host allocations were the next stall)."""
        import math
        offs, total = [], 0
        for _, dt, shape in parts:
            offs.append(total)
            total += (math.prod(shape) * torch.empty(0, dtype=dt).element_size() + 15) // 16 * 16
        k = self._ring_i % len(self._ring)
        self._ring_i += 1
        slot = self._ring[k]
        if self._marks is None:
            from ._lib import HostMarks
            self._marks = HostMarks(len(self._ring))
        if slot is None or slot[0].numel() < total:
            cap = max(4096, 1 << (total - 1).bit_length())
            self._marks.wait(k)
            slot = self._ring[k] = (torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=self.device))
        else:
            # the copy that last used this slot (32 batches ago) has run: a word of pinned memory the stream stored to behind it
            # (tn_mark_host).  With 32 slots the wait never happens in practice (the HIP runtime's own back-pressure stops the
            # host a few steps ahead of the GPU), and it sleeps instead of spinning: see _lib.HostMarks
            self._marks.wait(k)
        host, dev = slot
        out = []
        for (src, dt, shape), o in zip(parts, offs):
            n = math.prod(shape) * torch.empty(0, dtype=dt).element_size()
            # numpy, not torch, for the host-side copies: a torch CPU op over more than 32 k elements (the time masks) wakes
            # the intra-op thread pool — one thread per VISIBLE core (256 on the GPU boxes), all spinning for a few ms after the
            # op — which burns through the container's CPU quota (16 cores) and gets the whole process frozen for the rest of the
            # 100 ms accounting period, mid launch sequence (round 4: 60-85 ms GPU-idle gaps on the configs[3] leg)
            view = host[o:o + n].view(dt).view(shape).numpy()
            src_np = src.detach().cpu().numpy() if isinstance(src, torch.Tensor) else src
            if tuple(src_np.shape) == tuple(shape):
                view[...] = src_np
            else:                                      # time masks narrower than the padded batch: zero beyond them
                view[...] = 0
                view[:, :src_np.shape[1]] = src_np
            out.append(dev[o:o + n].view(dt).view(shape))
        dev[:total].copy_(host[:total], non_blocking=True)
        self._marks.mark(k, torch.cuda.current_stream(self.device).cuda_stream)
        return out

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

    def batch(self, waveforms, masks=None, lengths=None, rates=None, freq_masks=None, time_masks=None, into=None, align_frames=None):
        """[B, A] float waveforms -> [B, n_mels, frames] on the GPU (the collate_fn layout, zero beyond an utterance's end).

        masks: optional int32 [B, 4] = (f_start, f_end, t_start, t_end), one interval per axis (``tn_mel_forward``).
        lengths: int64 [B] valid samples of each zero-padded waveform (ragged batch); rates: float64 [B] SpecAugment
        time-stretch rates; freq_masks / time_masks: bool [B, n_mels] / [B, frames] unions of mask intervals.
        into: a bf16 / fp8 ``TitaNet`` on the same device — the spectrogram is then written STRAIGHT into that model's prolog
        operand (bf16 rows x n_mels; no float32 tensor, no packing pass) and a ``PackedSpectrograms`` handle is returned;
        ``model(handle, speakers)`` runs the network on it, ragged batches with their padding mask (BASELINE configs[3]).
        align_frames: the frame axis is padded to a multiple of this (default: 256 for a ragged batch written ``into`` a model,
        1 otherwise).  The kernels of a variable-length batch skip 256-row tiles without valid frames; with every utterance
        starting on a tile boundary only the tile at its END is partly padding (row b * T is otherwise somewhere inside a tile
        shared with utterance b - 1), and batches whose longest utterance falls in the same 256-frame bucket share one plan.
        The result does not depend on the padding (lengths mask)."""
        if waveforms.dim() != 2:
            raise ValueError("expected waveforms of shape [B, A]")
        if not torch.cuda.is_available():
            raise RuntimeError("titanet_amd.transforms.MelSpectrogram needs a ROCm device; there is no CPU execution path")
        w = waveforms.to(device=self.device, dtype=torch.float32).contiguous()
        B, A = w.shape
        stream = torch.cuda.current_stream(self.device).cuda_stream
        vp = C.c_void_p
        if lengths is None and rates is None and freq_masks is None and time_masks is None and into is None:
            T = 1 + A // self.hop_length
            out = torch.empty(B, self.n_mels, T, dtype=torch.float32, device=self.device)
            m = masks.to(device=self.device, dtype=torch.int32).contiguous() if masks is not None else None
            check(self._lib.tn_mel_forward(self._mel(), vp(w.data_ptr()), B, A, vp(m.data_ptr() if m is not None else 0),
                                           vp(out.data_ptr()), vp(stream)), "tn_mel_forward")
            return out
        assert masks is None, "interval masks go with the equal-length entry point; pass freq_masks / time_masks here"
        ln = torch.as_tensor(lengths if lengths is not None else [A] * B, dtype=torch.int64)
        rt = torch.as_tensor(rates if rates is not None else [1.0] * B, dtype=torch.float64)
        # the C entry point trusts these: a length beyond the row or a mask of another batch size would read out of bounds
        if ln.numel() != B or int(ln.min()) < 1 or int(ln.max()) > A:
            raise ValueError(f"lengths must be {B} values in [1, {A}]")
        if rt.numel() != B or not bool((rt > 0).all()):
            raise ValueError(f"rates must be {B} positive values")
        if freq_masks is not None and tuple(freq_masks.shape) != (B, self.n_mels):
            raise ValueError(f"freq_masks must have shape ({B}, {self.n_mels}), got {tuple(freq_masks.shape)}")
        if time_masks is not None and (time_masks.dim() != 2 or time_masks.shape[0] != B):
            raise ValueError(f"time_masks must have shape ({B}, frames), got {tuple(time_masks.shape)}")
        frames = [self.n_frames(int(n), float(r)) for n, r in zip(ln.tolist(), rt.tolist())]
        T = max(frames)
        if time_masks is not None:
            T = max(T, time_masks.shape[1])
        if align_frames is None:
            align_frames = 256 if (into is not None and lengths is not None) else 1
        if align_frames > 1:
            T = -(-T // align_frames) * align_frames      # (_upload zero-pads a narrower time mask inside its pinned slot)
        out = None if into is not None else torch.empty(B, self.n_mels, T, dtype=torch.float32, device=self.device)
        parts = [(ln, torch.int64, (B,)), (rt, torch.float64, (B,))]
        if freq_masks is not None:
            parts.append((freq_masks.detach(), torch.uint8, (B, self.n_mels)))
        if time_masks is not None:
            parts.append((time_masks.detach(), torch.uint8, (B, T)))
        up = self._upload(parts)
        ln_d, rt_d = up[0], up[1]
        fm = up[2] if freq_masks is not None else None
        tm = up[-1] if time_masks is not None else None
        self.last_frames = frames
        if into is not None:
            from .models import PackedSpectrograms
            plan = into._get_plan(B, T)
            dst = self._lib.tn_plan_prolog_input(plan.handle)
            if not dst:
                raise RuntimeError("MelSpectrogram.batch(into=model) needs a bf16 / fp8 model (fp32 plans read the float32 tensor)")
            check(self._lib.tn_mel_forward_batch_packed(self._mel(), vp(w.data_ptr()), B, A, vp(ln_d.data_ptr()), vp(rt_d.data_ptr()),
                                                        vp(fm.data_ptr() if fm is not None else 0), vp(tm.data_ptr() if tm is not None else 0),
                                                        T, vp(dst), vp(0), vp(stream)), "tn_mel_forward_batch_packed")
            # the write replaced the plan's prolog operand, which a pending backward of an earlier forward on this plan would
            # re-read for the prolog weight gradient: a new generation makes that backward raise instead of using this batch
            plan.generation += 1
            return PackedSpectrograms(plan, plan.generation, B, self.n_mels, T, torch.tensor(frames, dtype=torch.int64),
                                      into.flat_parameters().device)
        check(self._lib.tn_mel_forward_batch(self._mel(), vp(w.data_ptr()), B, A, vp(ln_d.data_ptr()), vp(rt_d.data_ptr()),
                                             vp(fm.data_ptr() if fm is not None else 0), vp(tm.data_ptr() if tm is not None else 0),
                                             T, vp(out.data_ptr()), vp(stream)), "tn_mel_forward_batch")
        return out

    def n_frames(self, n_samples, rate=1.0):
        """frames of an utterance: 1 + n // hop (center=True), then ceil(. / rate) under a time stretch"""
        import math
        t = 1 + n_samples // self.hop_length
        return t if rate == 1.0 else int(math.ceil(t / rate))

    def __call__(self, example):
        assert isinstance(example, dict) and "waveform" in example, "Wrong input structure"
        new_example = copy_example(example)
        wave = new_example["waveform"]
        apply_specaugment = random.random() < self.specaugment_probability
        if not apply_specaugment:
            new_example["spectrogram"] = self.batch(wave.reshape(1, -1))
            return new_example
        # reference src/transforms.py:168-201: stretch rate, then `num` frequency masks, then `num` time masks, each
        # mask_along_axis call drawing its own two uniforms
        rate = random.uniform(self.specaugment_min_speed, self.specaugment_max_speed)
        T = self.n_frames(wave.shape[-1], rate)
        fm = torch.zeros(1, self.n_mels, dtype=torch.bool)
        tm = torch.zeros(1, T, dtype=torch.bool)
        for _ in range(self.specaugment_freq_mask_num):
            f0, f1 = self._mask_bounds(self.n_mels, self.specaugment_freq_mask_ratio * self.n_mels)
            fm[0, f0:f1] = True
        for _ in range(self.specaugment_time_mask_num):
            t0, t1 = self._mask_bounds(T, self.specaugment_time_mask_ratio * T)
            tm[0, t0:t1] = True
        new_example["spectrogram"] = self.batch(wave.reshape(1, -1), rates=[rate], freq_masks=fm, time_masks=tm)
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
