"""Host-side timeline of the configs[3] leg: where the wall-clock of a step goes (front end call, forward+backward enqueue, GPU)."""
import random, sys, time, torch
sys.path.insert(0, ".")
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer
from titanet_amd.transforms import MelSpectrogram
dev = torch.device("cuda", 0)
B, sr, hop = 32, 16000, 160
rnd = random.Random(3); g = torch.Generator().manual_seed(3)
nsamp = [int(rnd.uniform(2.0, 20.0) * sr) for _ in range(B)]
wav = torch.zeros(B, max(nsamp))
for b, n in enumerate(nsamp): wav[b, :n] = torch.randn(n, generator=g) * 0.05
wav = wav.to(dev)
mel = MelSpectrogram(sr, n_fft=512, win_length=400, hop_length=hop, n_mels=80, device=dev)
rates = [rnd.uniform(0.95, 1.05) for _ in range(B)]
frames = [mel.n_frames(n, r) for n, r in zip(nsamp, rates)]
T = max(frames)
fm = torch.zeros(B, 80, dtype=torch.bool); tm = torch.zeros(B, T, dtype=torch.bool)
for b in range(B):
    f0 = rnd.randrange(0, 60); fm[b, f0:f0 + rnd.randrange(1, 28)] = True
    t0 = rnd.randrange(0, max(1, frames[b] - 10)); tm[b, t0:t0 + rnd.randrange(1, max(2, int(0.15 * frames[b])))] = True
m = TitaNet.get_titanet(n_mega_blocks=10, model_size="m", loss_function=LOSSES["ce"](192, 251, device=dev), dropout=0.1, device=dev, precision="bf16").train()
tr = Trainer(m)
y = torch.randint(0, 251, (B,), generator=g).to(dev)
ln = torch.tensor(frames, dtype=torch.int64)
def sync(): torch.cuda.synchronize()
for it in range(6):
    sync(); t0 = time.perf_counter()
    x = mel.batch(wav, lengths=nsamp, rates=rates, freq_masks=fm, time_masks=tm, into=m)
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    tr.step(x, y, lengths=ln)
    t3 = time.perf_counter(); sync(); t4 = time.perf_counter()
    print(f"iter {it}: mel.batch call {1e3*(t1-t0):.2f} ms (+{1e3*(t2-t1):.2f} to drain), tr.step call {1e3*(t3-t2):.2f} ms (+{1e3*(t4-t3):.2f} to drain), total {1e3*(t4-t0):.2f}")
# unsynchronised steps
sync(); t0 = time.perf_counter()
for it in range(5):
    x = mel.batch(wav, lengths=nsamp, rates=rates, freq_masks=fm, time_masks=tm, into=m)
    tr.step(x, y, lengths=ln)
sync(); print("5 steps back to back:", 1e3 * (time.perf_counter() - t0) / 5, "ms/step")


xf = mel.batch(wav, lengths=nsamp, rates=rates, freq_masks=fm, time_masks=tm)
def series(name, fn, n=12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    sync(); ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    sync()
    print(name, " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.1f}" for i in range(n)))
series("masked   ", lambda: tr.step(xf, y, lengths=ln))
series("unmasked ", lambda: tr.step(xf, y))
series("full-len ", lambda: tr.step(xf, y, lengths=torch.full((B,), T, dtype=torch.int64)))
series("masked   ", lambda: tr.step(xf, y, lengths=ln))
series("mel+masked", lambda: tr.step(mel.batch(wav, lengths=nsamp, rates=rates, freq_masks=fm, time_masks=tm, into=m), y, lengths=ln))

sync()
for it in range(14):
    t0 = time.perf_counter()
    x = mel.batch(wav, lengths=nsamp, rates=rates, freq_masks=fm, time_masks=tm, into=m)
    t1 = time.perf_counter()
    tr.step(x, y, lengths=ln)
    t2 = time.perf_counter()
    print(f"host it {it}: mel.batch {1e3*(t1-t0):.2f} ms, tr.step {1e3*(t2-t1):.2f} ms")
sync()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for it in range(10):
    x = mel.batch(wav, lengths=nsamp, rates=rates, freq_masks=fm, time_masks=tm, into=m)
    tr.step(x, y, lengths=ln)
sync(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
