"""Pure-write / pure-read / copy bandwidth with torch kernels (cold: rotating over 64 x 39.3 MB tensors)."""
import torch, time
n = 256 * 300 * 256
bufs = [torch.empty(n, dtype=torch.bfloat16, device="cuda") for _ in range(64)]
src = [torch.randn(n, device="cuda").to(torch.bfloat16) for _ in range(64)]
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
t = timeit(lambda: [b.zero_() for b in bufs]); print(f"zero_  : {64 * n * 2 / t / 1e12:.2f} TB/s written")
t = timeit(lambda: [b.fill_(1.5) for b in bufs]); print(f"fill_  : {64 * n * 2 / t / 1e12:.2f} TB/s written")
t = timeit(lambda: [b.copy_(s) for b, s in zip(bufs, src)]); print(f"copy_  : {2 * 64 * n * 2 / t / 1e12:.2f} TB/s (r+w)")
t = timeit(lambda: [s.sum() for s in src]); print(f"sum    : {64 * n * 2 / t / 1e12:.2f} TB/s read")
big = torch.empty(64 * n, dtype=torch.bfloat16, device="cuda")
t = timeit(lambda: big.zero_()); print(f"zero_ one 2.5 GB tensor: {64 * n * 2 / t / 1e12:.2f} TB/s written")
t = timeit(lambda: hipmemset(big)) if False else 0
