"""Timing of the mel front end on the GPU (tn_mel_forward / tn_mel_forward_batch): 256 utterances of 3 s, 16 kHz."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from titanet_amd.transforms import MelSpectrogram

mel = MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=160, n_mels=80, specaugment_probability=0.0)
w = (torch.randn(256, 48000) * 0.05).cuda()
rates = [1.0 + 0.0002 * i for i in range(256)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


a = timed(lambda: mel.batch(w))
b = timed(lambda: mel.batch(w, lengths=[48000] * 256))
c = timed(lambda: mel.batch(w, rates=rates))
print(f"mel 256 x 3 s: one frame per workgroup {a:.3f} ms | batched kernel (16 frames / workgroup, T-contiguous stores) {b:.3f} ms | "
      f"with time stretch {c:.3f} ms  ({256 / b * 1e3:.0f} utt/s)")
