#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): utterances/sec of TitaNet-S fwd+bwd on 80-mel x 300-frame
synthetic batches on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch per GPU: forward + backward of
TitaNet-S (17 mega blocks, parameters.yml:54) with the CE-251 head, bf16 compute, batch 256 per GPU
(BASELINE.json configs[1]) + gradient all-reduce (N > 1, overlapped with backward) + fused Adam.  Inputs are resident
in HBM before the timed region.  Prints ONE JSON line (rank 0).

The `roofline` object:
  * achieved / frac        SURVEY.md 8(d) algorithmic bytes of the step (70.66 MB per utterance, fwd+bwd, bf16) / step time,
                           against the 8.0 TB/s HBM3E spec — the number the 70 % target is quoted on;
  * traffic, traffic_ratio HBM bytes the step really moves (rocprofv3 PMC passes of this same command, committed under
                           profiles/, file named in traffic_source) and their ratio to the algorithmic bytes;
  * stream_ceiling         what a plain 2-reads-1-write streaming pass over tensors of the step's size reaches on THIS box
                           in THIS run (cold = rotating over 2.4 GB like the step's 4.7 GB workspace, hot = Infinity-Cache
                           resident): the practical ceiling every kernel of the path is measured against;
  * dominant_kernel        the class with the most time per step (HIP events on the launch stream, live): its own traffic
                           model and the bytes the 8(d) model attributes to it.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
ALG_BYTES_PER_UTT_BF16 = 70.66e6   # 7,065,600 elements x 5 passes x 2 B (S/17, T=300, fwd+bwd)

PROF_CLASSES = {1: "fwd_subblock_gemm", 2: "bwd_pointwise_wgrad", 3: "bwd_skip_dgrad", 4: "bwd_subblock_dgrad_depthwise"}
# kernel behind each class on the headline shape (for the PMC traffic lookup)
PROF_KERNELS = {1: ("sub_fwd_v5_kernel<3, true, 7",), 2: ("pgemm_tn_batched_kernel", "wgrad_batched_v2_kernel<3, false>"),
                3: ("dgrad_v2_kernel<64>",), 4: ("dgrad_dw_v6_kernel<7",)}       # (name prefixes: every instance of the class's main flag set)


def _pmc_file():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    return files[-1] if files else None


def _kernel_digest():
    """Digest of the kernel sources the library is built from (titanet_amd/csrc/build.py): a PMC summary only describes
    the kernels it was taken with."""
    try:
        from titanet_amd.csrc.build import _digest
        return _digest()[:16]
    except Exception:
        return None


def _pmc_bytes(d):
    # FETCH_SIZE is doubled: on gfx950 it reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM
    # section); units are KiB
    return (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0


def pmc_traffic(cls):
    """(HBM bytes per launch of the class's kernel, HBM bytes per step of the whole path, source file) from the committed
    rocprofv3 PMC passes (profiles/*pmc_traffic.json: tools/pmc_summary.py over separate `--pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` runs of this same command) — NOT measured in this run (PMC needs rocprofv3 around the process)."""
    path = _pmc_file()
    if not path:
        return None, None, None
    try:
        data = json.load(open(path))
        meta = data.get("_meta", {})
        # stale counters are worse than none: the file names the kernel-source digest it was taken at (tools/pmc_summary.py)
        if meta.get("kernel_digest") != _kernel_digest():
            return None, None, os.path.relpath(path, ROOT) + " (STALE: taken at kernel digest %s, this build is %s)" % (
                meta.get("kernel_digest"), _kernel_digest())
        steps = float(meta.get("steps", 0))
        ks = [v for n, v in data.items() if n != "_meta" and n.startswith(PROF_KERNELS[cls])]      # (a tuple of prefixes)
        per_launch = int(sum(_pmc_bytes(v) * v.get("launches", 0) for v in ks) / max(sum(v.get("launches", 0) for v in ks), 1)) if ks else None
        per_step = None
        if steps > 0:
            per_step = int(sum(_pmc_bytes(v) * v.get("launches", 0) for n, v in data.items() if n != "_meta") / steps)
        return per_launch, per_step, os.path.relpath(path, ROOT)
    except Exception:
        return None, None, None


def device_note(dev):
    """Which GPU rank 0 ran on, for whoever samples `rocm-smi` beside the run: HIP index 0 is the FIRST VISIBLE device
    (HIP_/ROCR_VISIBLE_DEVICES), matched to an SMI card by its PCI bus id, not by number; and the GPU is busy for seconds
    only — the timed region is steps x ms_per_step, the other legs a few hundred steps, the CPU baseline leg tens of seconds
    of host-only work — so a 5-second sampler mostly sees it idle."""
    pr = torch.cuda.get_device_properties(dev)
    return {"hip_index": dev.index, "name": pr.name, "pci_bus_id": getattr(pr, "pci_bus_id", None),
            "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES"),
            "gpu_busy_note": "GPU legs total a few seconds of the run; match the SMI card by pci_bus_id"}


def kernel_own_bytes(cls, rows, hidden, esz):
    """HBM bytes one launch of the kernel class moves by its own design (kept depthwise outputs and kept
    per-layer gradients included)."""
    t = rows * hidden * esz
    return {
        1: 3 * t,                          # read input rows once, write raw output once + the kept depthwise output (for wgrad)
        2: 2 * t + hidden * hidden * 4,    # ONE unit of the class since round 4: the stored BatchNorm-backward'd dS + the kept
                                           # depthwise output (pgemm_tn_batched_kernel: 67 such units per step); the units left
                                           # in wgrad_batched_v2_kernel (block 0's skip conv, 6 epilog slabs: 3t each, 12
                                           # pooling units: 1.5t) are added by the caller (class_units_per_step)
        3: 3 * t,                          # read dYbn, Y; write dD
        4: (4 * t * 32 // 30) + t,         # fused data gradient + depthwise backward: read dYbn, Y, previous raw output; write
                                           # dYbn(prev); 32-row tiles yield 30 rows (the overlap is re-read); + the stored
                                           # BatchNorm-backward'd dS (every launch since round 4: the weight gradient's operand)
    }[cls]


def kernel_attributed_bytes(cls, rows, hidden, esz):
    """Bytes the SURVEY.md 8(d) model (5 passes per inter-kernel tensor: forward write + read, backward read of the saved
    tensor, gradient write + read) attributes to one launch of the class.  The model fuses the weight gradients into the
    data-gradient pass and never materialises the depthwise output or its gradient, so it attributes NOTHING to a separate
    weight-gradient launch and only one tensor to the depthwise backward."""
    t = rows * hidden * esz
    return {1: 2 * t, 2: 0, 3: 2 * t, 4: 3 * t}[cls]


def stream_ceiling(dev, rows=256 * 300, hidden=256, sets=10, reps=40):
    """2 reads + 1 write over rows x 256 bf16 tensors (39.3 MB each at the bench shape) with a plain element-wise kernel:
    cold = rotating over `sets` disjoint buffer triples (no Infinity-Cache reuse, like the step's workspace), hot = one
    triple.  GB/s of 3 x tensor bytes / launch time, HIP events on the current stream."""
    n = rows * hidden
    bufs = [[torch.empty(n, dtype=torch.bfloat16, device=dev).normal_() for _ in range(3)] for _ in range(sets)]
    out = {}
    for tag, nset in (("cold", sets), ("hot", 1)):
        for i in range(nset + 2):
            a, b, c = bufs[i % nset]
            torch.add(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            a, b, c = bufs[i % nset]
            torch.add(a, b, out=c)
        e1.record()
        torch.cuda.synchronize()
        out[tag] = 3 * n * 2 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del bufs
    torch.cuda.empty_cache()
    return out


def effective_cores():
    """Host cores this process may actually use: the visible count capped by the container's CPU quota (cgroup v2 cpu.max /
    v1 cfs_quota).  The GPU boxes show 256 cores under a 16-core quota: thread pools sized by os.cpu_count() burn the quota in
    a few ms of spinning and the whole process is frozen for the rest of the 100 ms accounting period."""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(budget_s=24.0):
    """The reference's CPU path timed on this box's host cores: the reference's module graph
    (nn.Conv1d / nn.BatchNorm1d / ... leaf modules, F.pad per conv, the margin loss's per-row loop) rebuilt in
    oracle/eager_modules.py (the reference's files do not travel to the GPU box; pinned to its golden vectors by
    tests/test_oracle_golden.py), float32, TitaNet-S/17, 80 x 300 synthetic batches, dropout 0.1.  Legs: the reference's own
    batch 8 (parameters.yml:13) — eval forward, train fwd+bwd with CE and with ArcFace(30, 0.2) — at the reference's
    default 2 threads (src/train.py:21-22, parameters.yml:74) and at the best thread count for that batch; batch 256
    (the GPU workload; 64 on hosts with < 32 cores) train fwd+bwd with CE on all cores, one iteration.  Bounded: each batch-8
    leg stops after ~budget_s / 8 seconds."""
    from oracle.eager_modules import EagerTitaNet
    ncpu = effective_cores()
    g = torch.Generator().manual_seed(42)

    def leg(batch, mode, loss, threads, max_s, max_it, warm=True):
        torch.set_num_threads(threads)
        m = EagerTitaNet(n_mega_blocks=17, dropout=0.1, loss=loss, n_classes=251)
        x = torch.randn(batch, 80, 300, generator=g) * 0.11 - 0.10
        y = torch.randint(0, 251, (batch,), generator=g)

        def step():
            if mode == "eval":
                with torch.no_grad():
                    m(x)
            else:
                m.zero_grad(set_to_none=True)
                m(x, y)[2].backward()
        m.eval() if mode == "eval" else m.train()
        if warm:
            step()
        n, t0 = 0, time.perf_counter()
        while n < max_it:
            step()
            n += 1
            if time.perf_counter() - t0 > max_s:
                break
        dt = (time.perf_counter() - t0) / n
        return {"batch": batch, "mode": mode, "loss": loss, "threads": threads, "iterations": n, "utt_per_s": round(batch / dt, 2)}

    many = min(ncpu, 8)      # batch 8: beyond 8 intra-op threads the per-layer ops lose to synchronisation (measured r1: 32 thr slower)
    per = budget_s / 8.0
    legs = [leg(8, "eval", None, 2, per, 10), leg(8, "train", "ce", 2, per, 6), leg(8, "train", "arc", 2, per, 6),
            leg(8, "eval", None, many, per, 20), leg(8, "train", "ce", many, per, 12), leg(8, "train", "arc", many, per, 12),
            # the GPU workload's own batch on all cores: ONE un-warmed iteration (tens of seconds of CPU work); 64 utterances
            # instead of 256 on small hosts so that the default bench run stays within minutes
            leg(256 if ncpu >= 32 else 64, "train", "ce", min(ncpu, 64), 0.0, 1, warm=False)]
    best = max((lg for lg in legs if lg["mode"] == "train" and lg["loss"] == "ce"), key=lambda lg: lg["utt_per_s"])
    torch.set_num_threads(min(ncpu, 8))
    return {"value": best["utt_per_s"], "unit": "utterances/s", "cores": best["threads"], "kind": "port", "host_cores": os.cpu_count(),
            "usable_cores": ncpu,
            "sample": f"best train fwd+bwd (CE) leg: batch {best['batch']}, {best['iterations']} iterations, {best['threads']} threads; "
                      "eager nn.Module graph of the reference (oracle/eager_modules.py), fp32, TitaNet-S/17, 80x300",
            "legs": legs}


# algorithmic FLOPs (fwd + bwd) and bf16 bytes per utterance at T = 300
FLOPS_PER_UTT = {"s": 9.62e9, "m": 21.4e9, "l": 42.0e9}
ELEMS_PER_FRAME = {"m": (1 + 10 * 5) * 512 + 1536, "l": (1 + 5 * 5) * 1024 + 1536}     # inter-kernel tensor elements per frame
MFMA_PEAK_BF16 = 2.5e15        # dense (MI355X_MICROARCH.md)


def _timed_steps(fn, warm, steps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def other_configs(dev, only=None):
    """BASELINE.json configs[2..4] as bounded legs OUTSIDE the headline's timed region (rank 0, one GPU): ms per step,
    utterances/s, the governing roofline of SURVEY.md 8(d) and the MFMA utilisation FLOPs x utt/s / 2.5 PFLOP/s (bf16 dense)."""
    import random
    from titanet_amd import LOSSES, TitaNet
    from titanet_amd.trainer import Trainer
    from titanet_amd.transforms import MelSpectrogram
    out = {}
    n_classes = 251

    def leg(name, fn):
        if only and name not in only:
            return
        try:
            out[name] = fn()
        except Exception as e:      # a leg must never take the headline line down with it
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()

    def fixed(size, nb, prec, head, B=256, T=300, steps=8):
        loss = LOSSES["ce"](192, n_classes, device=dev) if head == "ce" else LOSSES["arc"](192, n_classes, device=dev, scale=30, margin=0.2)
        m = TitaNet.get_titanet(n_mega_blocks=nb, model_size=size, loss_function=loss, dropout=0.1, device=dev, precision=prec).train()
        tr = Trainer(m)
        g = torch.Generator().manual_seed(7)
        x = (torch.randn(B, 80, T, generator=g) * 0.11 - 0.10).to(dev)
        y = torch.randint(0, n_classes, (B,), generator=g).to(dev)
        dt = _timed_steps(lambda: tr.step(x, y), 3, steps)
        ups = B / dt
        # (checked AFTER the timed steps: parameters gone non-finite make the matrix pipe run ~8 % faster — constant operands,
        #  less power, higher clocks — and would flatter the number)
        r = {"workload": f"TitaNet-{size.upper()}/{nb} fwd+bwd+Adam, {head} head, batch {B}, 80x{T}, {prec}", "ms_per_step": round(dt * 1e3, 3),
             "utt_per_s": round(ups, 1), "mfma_util": round(FLOPS_PER_UTT[size] * ups / MFMA_PEAK_BF16, 4),
             "params_finite": bool(torch.isfinite(m.flat_parameters()).all())}
        if size == "l" and prec == "fp8":
            # BASELINE.md 3 / SURVEY.md 8(d): with e4m3 pointwise operands (5 PFLOP/s dense) L is HBM-bound again: 84.48 MB per
            # utterance at 8 TB/s = 94.7 k utterances/s
            nbytes = ELEMS_PER_FRAME["l"] * 300 * 10.0
            r.update({"bound": "hbm", "frac": round(nbytes * ups / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_MB_per_utt": round(nbytes / 1e6, 2),
                      "mfma_util_fp8_peak": round(FLOPS_PER_UTT[size] * ups / (2 * MFMA_PEAK_BF16), 4),
                      "note": "8(d): HBM-bound under fp8 (95 k utt/s roof); forward AND backward pointwise GEMMs on the f8f6f4 MFMA where the fp8 plan provides them, the rest in bf16"})
        elif size == "l":
            r.update({"bound": "mfma", "frac": r["mfma_util"], "peak_TFLOPs": MFMA_PEAK_BF16 / 1e12,
                      "note": "8(d): L is MFMA-bound in bf16"})
        else:
            nbytes = {"s": ALG_BYTES_PER_UTT_BF16, "m": ELEMS_PER_FRAME["m"] * 300 * 10.0}[size]
            r.update({"bound": "hbm", "frac": round(nbytes * ups / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_MB_per_utt": round(nbytes / 1e6, 2)})
        return r

    def ragged_m():
        # configs[3]: TitaNet-M, waveforms of U(2, 20) s at 16 kHz -> on-GPU mel + SpecAugment (time stretch, 1 freq + 1 time
        # mask, parameters.yml:79-107) -> zero-padded batch + lengths -> masked train step
        B, sr, hop = 32, 16000, 160
        rnd = random.Random(3)
        g = torch.Generator().manual_seed(3)
        nsamp = [int(rnd.uniform(2.0, 20.0) * sr) for _ in range(B)]
        A = max(nsamp)
        wav = torch.zeros(B, A)
        for b, n in enumerate(nsamp):
            wav[b, :n] = torch.randn(n, generator=g) * 0.05
        wav = wav.to(dev)
        mel = MelSpectrogram(sr, n_fft=512, win_length=400, hop_length=hop, n_mels=80, device=dev)
        rates = [rnd.uniform(0.95, 1.05) for _ in range(B)]
        frames = [mel.n_frames(n, r) for n, r in zip(nsamp, rates)]
        T = max(frames)
        fm = torch.zeros(B, 80, dtype=torch.bool)
        tm = torch.zeros(B, T, dtype=torch.bool)
        for b in range(B):
            f0 = rnd.randrange(0, 80 - 20); fm[b, f0:f0 + rnd.randrange(1, 28)] = True
            t0 = rnd.randrange(0, max(1, frames[b] - 10)); tm[b, t0:t0 + rnd.randrange(1, max(2, int(0.15 * frames[b])))] = True
        loss = LOSSES["ce"](192, n_classes, device=dev)
        m = TitaNet.get_titanet(n_mega_blocks=10, model_size="m", loss_function=loss, dropout=0.1, device=dev, precision="bf16").train()
        tr = Trainer(m)
        y = torch.randint(0, n_classes, (B,), generator=g).to(dev)
        ln = torch.tensor(frames, dtype=torch.int64)

        shape = {}

        def step():
            # the front end writes the prolog conv's packed bf16 operand itself (no float32 [B, 80, T] tensor, no pack pass); it
            # pads the frame axis to a multiple of 256 (every utterance starts on a row-tile boundary, MelSpectrogram.batch)
            x = mel.batch(wav, lengths=nsamp, rates=rates, freq_masks=fm, time_masks=tm, into=m)
            shape["T"] = x.shape[2]
            tr.step(x, y, lengths=ln)
        dt = _timed_steps(step, 6, 8)          # (6 warm-up steps: the front end allocates its 4 pinned staging slots on first use)
        valid = sum(frames)
        nbytes = ELEMS_PER_FRAME["m"] * valid * 10.0
        flops = FLOPS_PER_UTT["m"] * valid / 300.0
        T = shape["T"]
        return {"workload": f"TitaNet-M/10, {B} waveforms of U(2,20) s -> GPU mel + SpecAugment -> padded [B,80,{T}] + lengths -> masked fwd+bwd+Adam, bf16",
                "ms_per_step": round(dt * 1e3, 3), "utt_per_s": round(B / dt, 1), "audio_s_per_s": round(sum(nsamp) / sr / dt, 1),
                "valid_frames": valid, "padded_frames": B * T, "bound": "hbm", "frac": round(nbytes / dt / 1e9 / HBM_PEAK_GBS, 4),
                "mfma_util": round(flops / dt / MFMA_PEAK_BF16, 4), "includes": "mel front end + SpecAugment inside the step",
                "params_finite": bool(torch.isfinite(m.flat_parameters()).all())}

    def masked_s():
        # the headline model on a zero-padded batch with lengths (the collate_fn layout, lengths U(60, 300) frames): the padding
        # mask rides through the specialised kernels (tile masks, zero-stored padding rows, corrected statistics)
        B, T = 256, 300
        loss = LOSSES["ce"](192, n_classes, device=dev)
        m = TitaNet.get_titanet(n_mega_blocks=17, model_size="s", loss_function=loss, dropout=0.1, device=dev, precision="bf16").train()
        tr = Trainer(m)
        g = torch.Generator().manual_seed(7)
        x = (torch.randn(B, 80, T, generator=g) * 0.11 - 0.10).to(dev)
        y = torch.randint(0, n_classes, (B,), generator=g).to(dev)
        ln = torch.randint(T // 5, T + 1, (B,), generator=g)
        ln[0] = T
        dt = _timed_steps(lambda: tr.step(x, y, lengths=ln), 3, 8)
        valid = int(ln.sum())
        return {"workload": f"TitaNet-S/17 fwd+bwd+Adam, ce head, batch {B}, zero-padded 80x{T} + lengths (padding mask), bf16",
                "ms_per_step": round(dt * 1e3, 3), "utt_per_s": round(B / dt, 1), "valid_frames": valid, "padded_frames": B * T,
                "bound": "hbm", "frac": round(ALG_BYTES_PER_UTT_BF16 * (B / dt) / 1e9 / HBM_PEAK_GBS, 4),
                "params_finite": bool(torch.isfinite(m.flat_parameters()).all())}

    leg("s17_arcface_b256", lambda: fixed("s", 17, "bf16", "arc"))
    leg("s17_padded_masked_b256", masked_s)
    leg("m10_ragged_mel_specaug_masked", ragged_m)
    leg("m10_b256", lambda: fixed("m", 10, "bf16", "ce", steps=6))
    leg("l5_fp8_b256", lambda: fixed("l", 5, "fp8", "ce", steps=5))
    leg("l5_bf16_b256", lambda: fixed("l", 5, "bf16", "ce", steps=5))
    return out


def _spawn_entry(local_rank, argv, world, port):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    sys.argv = argv
    main()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--loss", default="ce", choices=["ce", "arc"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the bounded legs of BASELINE configs[2..4]")
    ap.add_argument("--only-config", action="append", default=None, help="run only this other_configs leg (repeatable), e.g. m10_ragged_mel_specaug_masked")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the streaming-ceiling microbenchmark (PMC passes: keeps foreign kernels out of the counters)")
    ap.add_argument("--prof-class", type=int, default=0, help="kernel class timed with HIP events in the timed region (0 = the dominant one)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm; gloo for single-GPU smoke tests)")
    ap.add_argument("--same-device", action="store_true", help="debug: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--grad-groups", type=int, default=2, help="gradient buckets of mega blocks for the overlapped all-reduce (N > 1)")
    ap.add_argument("--median-steps", type=int, default=100, help="extra steps timed one by one with HIP events after the timed region (median step time; 0 = skip)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU execution path for the product)")
    torch.set_num_threads(min(effective_cores(), 8))       # host-side torch ops: never a pool of one thread per VISIBLE core
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched as a plain `python bench.py --gpus N`: start the N ranks here (one process per GPU, RCCL)
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not args.same_device:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} device(s) visible — refusing to report a {args.gpus}-GPU line")
        import socket
        import torch.multiprocessing as mp
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        mp.spawn(_spawn_entry, args=(sys.argv, args.gpus, port), nprocs=args.gpus, join=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    if args.same_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} device(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
        probe = torch.ones(1, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(probe)                    # a real collective before anything is timed
        rccl_ranks = int(round(float(probe.item())))
        assert rccl_ranks == dist.get_world_size() == world, (rccl_ranks, dist.get_world_size(), world)

    from titanet_amd import LOSSES, TitaNet
    from titanet_amd.trainer import Trainer

    torch.manual_seed(42)
    n_classes, T = 251, 300
    if args.loss == "ce":
        loss = LOSSES["ce"](192, n_classes, device=dev)
    else:
        loss = LOSSES["arc"](192, n_classes, device=dev, scale=30, margin=0.2)      # parameters.yml:42-44
    model = TitaNet.get_titanet(embedding_size=192, n_mels=80, n_mega_blocks=17, model_size="s", attention_hidden_size=128,
                                loss_function=loss, dropout=0.1, device=dev, precision=args.precision).train()
    trainer = Trainer(model, lr=1e-3, n_buckets=args.grad_groups)
    g = torch.Generator(device="cpu").manual_seed(42 + rank)       # per-rank shard of the synthetic global batch
    x = (torch.randn(args.batch, 80, T, generator=g) * 0.11 - 0.10).to(dev)
    y = torch.randint(0, n_classes, (args.batch,), generator=g).to(dev)

    lib = model._lib
    for _ in range(max(args.warmup, 1)):
        trainer.step(x, y)
    torch.cuda.synchronize()

    # pick the dominant kernel class with one profiled step each (outside the timed region)
    plan = model._active_plan
    cls_ms = {}
    for cls in PROF_CLASSES:
        lib.tn_profile_begin(plan.handle, cls)
        trainer.step(x, y)
        ms, cnt = C.c_double(), C.c_int64()
        lib.tn_profile_read(plan.handle, C.byref(ms), C.byref(cnt))
        cls_ms[cls] = (ms.value, cnt.value)
    dom = args.prof_class if args.prof_class in cls_ms else max(cls_ms, key=lambda k: cls_ms[k][0])
    # two event records per launch cost stream time: a class with many launches per step is SAMPLED (every n-th launch, the
    # counter running on across steps so that every position of the class in the step is visited)
    per_step = cls_ms[dom][1]
    stride = 1 if per_step <= 8 else 7
    lib.tn_profile_sample(plan.handle, stride)
    lib.tn_profile_begin(plan.handle, dom)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    trainer.reducer.measure = world > 1           # (two event records per step on the rank's streams: nothing at N = 1)
    trainer.host_enqueue_s, trainer.host_enqueue_steps = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        emb, preds, lv = trainer.step(x, y)
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0            # this rank's own clock (spread across ranks reported below)
    host_enqueue_ms = trainer.host_enqueue_s / max(trainer.host_enqueue_steps, 1) * 1e3
    exposed = sorted(trainer.reducer.exposed_ms()) if world > 1 else []
    trainer.reducer.measure = False
    barrier()
    dt = time.perf_counter() - t0
    ms, cnt = C.c_double(), C.c_int64()
    lib.tn_profile_read(plan.handle, C.byref(ms), C.byref(cnt))
    lib.tn_profile_begin(plan.handle, 0)
    lib.tn_profile_sample(plan.handle, 1)
    rank_ms = [dt_rank / args.steps * 1e3]
    if world > 1:
        cdev = dev if args.backend == "nccl" else "cpu"
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        mine = torch.tensor([dt_rank / args.steps * 1e3], device=cdev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_ms = [float(t.item()) for t in allr]
    loss_value = float(lv.item())
    params_finite = bool(torch.isfinite(model.flat_parameters()).all())
    # SURVEY.md 8(d) protocol beside the driver's flags: >= 100 further steps, each bracketed by its own HIP events on the
    # launch stream -> median / spread of the step time (one GPU; the headline `value` stays the wall-clock mean of --steps)
    step_events = None
    if world == 1 and args.median_steps > 0:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.median_steps)]
        for e0, e1 in evs:
            e0.record()
            trainer.step(x, y)
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        step_events = {"steps": len(ts), "warmup_before": args.warmup + 4 + args.steps, "median_ms": round(ts[len(ts) // 2], 3),
                       "p10_ms": round(ts[len(ts) // 10], 3), "p90_ms": round(ts[(len(ts) * 9) // 10], 3), "mean_ms": round(sum(ts) / len(ts), 3)}

    if rank == 0:
        esz = 2 if args.precision == "bf16" else 4
        rows = args.batch * T
        headline = args.precision == "bf16" and args.batch == 256
        per_step = cnt.value / max(args.steps, 1)
        # the v2 weight-gradient kernel covers all 17 x (3 sub-blocks + skip) pointwise layers plus the 6 256-channel slabs
        # of the epilog conv in ONE launch (each unit reads dZ, Y and the kept depthwise output once: 3t); with gradient
        # groups (N > 1) the same units are spread over 1 + groups launches
        # ... and the two attentive-pooling weight gradients as 12 units of 1.5t (one 128-wide operand)
        # class 2 = both weight-gradient kernels (72 % of its time is pgemm_tn_batched_kernel): per step 67 stored-operand units of
        # 2t (51 sub-block layers + the skip convs of blocks > 0) + block 0's skip conv and the 6 epilog slabs at 3t (dZ, Y and the
        # operand: BatchNorm backward on load) + 12 pooling units at 1.5t, spread over the class's launches of the step
        units = 1.0
        own = int(kernel_own_bytes(dom, rows, 256, esz) * units)
        if dom == 2:
            t_ = rows * 256 * esz
            own = int((67 * 2 * t_ + 7 * 3 * t_ + 12 * 1.5 * t_ + 86 * 256 * 256 * 4) / max(per_step, 1))
        attributed = int(kernel_attributed_bytes(dom, rows, 256, esz) * units)
        avg_s = (ms.value / 1e3) / max(cnt.value, 1)
        value = args.batch * world * args.steps / dt
        step_s = dt / args.steps
        step_alg = ALG_BYTES_PER_UTT_BF16 * (esz / 2) * args.batch          # per GPU
        achieved = step_alg / step_s / 1e9
        k_traffic, s_traffic, src = pmc_traffic(dom) if headline else (None, None, None)
        ceil = {"cold": None, "hot": None} if args.no_ceiling else stream_ceiling(dev)
        rnd = lambda v, n=1: None if v is None else round(v, n)
        out = {
            "metric": "utterances/sec TitaNet-S fwd+bwd (80-mel x 300f)",
            "value": round(value, 1),
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": f"TitaNet-S/17 fwd+bwd+Adam, {args.loss.upper()}-251 head, batch {args.batch}/GPU, 80x300 (BASELINE configs[1])",
                       "global_batch": args.batch * world, "frames": T, "parallelism": f"dp{world}", "dropout": 0.1,
                       "loss": loss_value, "params_finite": params_finite, "rccl_ranks": rccl_ranks, "backend": args.backend if world > 1 else None,
                       "grad_groups": args.grad_groups if world > 1 else 1,
                       "grad_groups_note": ("N > 1 runs the weight-gradient launch per gradient bucket (overlapped all-reduce): the same rank "
                                            "program costs ~0.1 ms/step more than the single-bucket N = 1 program (DESIGN.md 5)") if world > 1 else None,
                       "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
                       # rank 0's host time to ENQUEUE a step (asynchronous launches; must stay below ms_per_step or the GPU starves:
                       # N ranks share the host's cores) and, at N > 1, the part of the gradient all-reduce that backward did not
                       # hide: end of the last bucket's collective - end of backward, HIP events on the rank's two streams
                       "host_enqueue_ms_per_step": round(host_enqueue_ms, 3),
                       "exposed_allreduce_ms": ({"median": round(exposed[len(exposed) // 2], 3), "max": round(exposed[-1], 3)} if exposed else None),
                       "device": device_note(dev)},
            "roofline": {
                "bound": "hbm", "scope": "whole step: SURVEY.md 8(d) algorithmic bytes / step time (per GPU)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_step": int(step_alg),
                "mfma_util": round(FLOPS_PER_UTT["s"] * (args.batch / step_s) / MFMA_PEAK_BF16, 4),
                "traffic": s_traffic, "traffic_source": src,
                "traffic_ratio": round(s_traffic / step_alg, 3) if s_traffic else None,
                "stream_ceiling": {"cold_GBps": rnd(ceil["cold"]), "hot_GBps": rnd(ceil["hot"]),
                                   "what": "2 reads + 1 write over 39.3 MB bf16 tensors, plain element-wise kernel, this run"},
                "frac_of_cold_stream_ceiling": rnd(achieved / ceil["cold"], 4) if ceil["cold"] else None,
                "moved_GBps": round(s_traffic / step_s / 1e9, 1) if s_traffic else None,
                "dominant_kernel": {
                    "class": PROF_CLASSES[dom], "name": " + ".join(PROF_KERNELS[dom]), "avg_launch_us": round(avg_s * 1e6, 2), "launches": cnt.value,
                    "sampled_every": stride,
                    "own_bytes_per_launch": own, "achieved_own_GBps": round(own / avg_s / 1e9, 1),
                    "attributed_8d_bytes_per_launch": attributed, "frac_attributed": round(attributed / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                    "traffic_per_launch": k_traffic},
                "class_ms_per_step": {PROF_CLASSES[k]: round(v[0], 3) for k, v in cls_ms.items()},
                "step_time_events": step_events},
        }
        del trainer, model
        torch.cuda.empty_cache()
        if world == 1 and not args.no_other_configs:
            out["other_configs"] = other_configs(dev, args.only_config)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
