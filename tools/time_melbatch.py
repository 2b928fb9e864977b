"""Where does MelSpectrogram.batch(into=model) stall on the host?  Per-statement wall clock over many calls."""
import random, sys, time, torch
sys.path.insert(0, ".")
from titanet_amd import LOSSES, TitaNet
from titanet_amd.trainer import Trainer
from titanet_amd.transforms import MelSpectrogram
import titanet_amd.transforms as TR
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
m = TitaNet.get_titanet(n_mega_blocks=10, model_size="m", loss_function=LOSSES["ce"](192, 251, device=dev), dropout=0.1, device=dev, precision="bf16").train()
tr = Trainer(m)
y = torch.randint(0, 251, (B,), generator=g).to(dev)
ln = torch.tensor(frames, dtype=torch.int64)
import cProfile, pstats, io
for rep in range(3):
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    for it in range(20):
        x = mel.batch(wav, lengths=nsamp, rates=rates, freq_masks=fm, time_masks=tm, into=m)
        tr.step(x, y, lengths=ln)
    pr.disable()
    torch.cuda.synchronize()
    print("rep", rep, "ms/step", 1e3 * (time.perf_counter() - t0) / 20)
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(7); print(st.getvalue()[-1500:])
