"""Data parallelism on the real model step (BASELINE.json configs[2] partitions the batch over ranks):
two ranks sharing cuda:0 (the GPU box has one device; gloo moves the CUDA buffers) run Trainer.step on DIFFERENT shards of
one global batch with the overlapped bucketed all-reduce, and
  (i)  the replicas stay bit-identical after 3 steps,
  (ii) the all-reduced gradient equals the sum of the two ranks' local gradients (Adam then applies 1 / world),
  (iii) gradient grouping (the per-bucket weight-gradient launches behind the overlap) does not change the gradient.
"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _model(loss="arc", groups=1, seed=11):
    from titanet_amd import LOSSES, TitaNet
    torch.manual_seed(seed)
    kw = {"scale": 30, "margin": 0.2} if loss == "arc" else {}
    lf = LOSSES[loss](192, 32, device="cuda", **kw)
    m = TitaNet.get_titanet(n_mega_blocks=5, model_size="s", loss_function=lf, dropout=0.1, device="cuda", precision="bf16").train()
    m.grad_groups = groups
    return m


def _shard(rank, per=24, T=151):
    g = torch.Generator().manual_seed(42)              # one global batch, contiguous shards
    x = torch.randn(2 * per, 80, T, generator=g) * 0.11 - 0.10
    y = torch.randint(0, 32, (2 * per,), generator=g)
    return x[rank * per:(rank + 1) * per].cuda(), y[rank * per:(rank + 1) * per].cuda()


def _worker(rank, world, init_file, q):
    torch.cuda.set_device(0)
    # file rendezvous: no TCP-store port to lose between "pick a free port" and "bind it"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        from titanet_amd.trainer import Trainer
        tr = Trainer(_model(seed=11 + rank), lr=1e-3, n_buckets=3)          # different inits: rank 0's weights must win
        assert tr.model.grad_groups == 3
        x, y = _shard(rank)
        # (ii) one forward/backward: local gradient, then the overlapped all-reduce
        tr.forward_backward(x, y)
        plan = tr._last_plan
        assert len(plan.buckets) == 4 and plan.buckets[0][1] == tr.model.flat_parameters().numel()
        assert sorted(b for b, _ in plan.buckets)[0] == 0 and sum(e - b for b, e in plan.buckets) == plan.buckets[0][1]
        torch.cuda.synchronize()
        local = tr.model.flat_gradients().clone()
        tr.reducer.all_reduce_overlapped_(tr.model.flat_gradients(), plan, tr.model._lib)
        torch.cuda.synchronize()
        red = tr.model.flat_gradients().clone()
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        torch.cuda.synchronize()
        err = float((red - (gathered[0] + gathered[1])).abs().max() / red.abs().max())
        # (i) three full steps, then compare replicas bit for bit
        losses = []
        for _ in range(3):
            losses.append(float(tr.step(x, y)[2]))
        torch.cuda.synchronize()
        flat = tr.model.flat_parameters().clone()
        others = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(others, flat)
        torch.cuda.synchronize()
        same = bool(torch.equal(others[0], others[1]))
        q.put((rank, err, same, losses, bool(torch.isfinite(flat).all())))
        dist.barrier()                  # nobody tears its connections down while the peer is still in a collective
    finally:
        dist.destroy_process_group()


def test_two_rank_model_step_same_device(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, str(tmp_path / "rendezvous"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, same, losses, finite in res:
        print(f"rank {rank}: |allreduce - sum of locals| / max = {err:.2e}, replicas identical = {same}, losses {losses}")
        assert err < 1e-6, err          # float32 sums of two terms: exact up to the collective's own rounding
        assert same and finite
    assert res[0][3] != res[1][3]       # the ranks really saw different shards


def test_gradient_grouping_does_not_change_the_gradient():
    """grad_groups = 1 (one deferred weight-gradient launch at the end) vs 4 (per-bucket launches + events) on the SAME
    forward: backward is repeatable from one forward and linear in it, so the two gradients may differ only by the
    summation order of the split-K slabs and of the atomics-accumulated BatchNorm sums.  (Two separate FORWARD runs would
    not be comparable this tightly: bf16 storage makes a randomly initialised train-mode network chaotic)"""
    import ctypes as C
    from titanet_amd._lib import check
    x, y = _shard(0, per=32, T=300)
    m = _model("ce", groups=4, seed=5)          # the workspace is sized for the larger slab region
    m._seed_base, m._step = 99, 0
    m._native_forward(x, y)
    plan = m._active_plan
    stream = torch.cuda.current_stream().cuda_stream
    gs = {}
    for groups in (4, 1, 4):
        torch.cuda.synchronize()
        check(m._lib.tn_plan_set_grad_groups(plan.handle, groups), "tn_plan_set_grad_groups")
        m._read_buckets(plan)
        assert len(plan.buckets) == (1 if groups == 1 else 5)
        check(m._lib.tn_backward(plan.handle, C.c_float(1.0), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_void_p(stream)), "tn_backward")
        torch.cuda.synchronize()
        gs.setdefault(groups, []).append(m.flat_gradients().clone())
    ref = gs[1][0]
    d_group = float((gs[4][0] - ref).norm() / ref.norm())
    d_repeat = float((gs[4][0] - gs[4][1]).norm() / ref.norm())
    print(f"grouped vs ungrouped gradient rel diff {d_group:.2e}; same grouping repeated {d_repeat:.2e}")
    # run-to-run noise of the bf16 backward itself (atomics-ordered BatchNorm sums flip a few bf16 roundings of the stored
    # gradients): grouping must not add to it
    assert d_group < 3 * d_repeat + 1e-3 and d_group < 3e-2, (d_group, d_repeat)
    assert torch.isfinite(ref).all() and float(ref.norm()) > 0


# ---------------------------------------------------------------------------------------------------------------------
# parameters.yml-driven data-parallel training (titanet_amd.train under torchrun; reference src/train.py:11-183)
# ---------------------------------------------------------------------------------------------------------------------
def _train_worker(rank, world, init_file, ckpt, q):
    import yaml
    from tests.test_train_gpu import PARAMS
    from titanet_amd import train
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        cfg = yaml.safe_load(yaml.safe_dump(PARAMS))
        cfg["training"]["batch_size"] = 24              # GLOBAL batch: 12 per rank
        cfg["training"]["loss"] = "arc"                 # BASELINE configs[2]: ArcFace(30, 0.2) from the yml's loss section
        params = train.Struct(**cfg)
        model, trainer, hist = train.run(params, steps=4, n_classes=16, precision="bf16", log_every=2, rank=rank, world=world,
                                         device=torch.device("cuda", 0), grad_groups=2)
        torch.cuda.synchronize()
        flat = model.flat_parameters().clone()
        others = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(others, flat)
        torch.cuda.synchronize()
        if rank == 0:
            train.save_checkpoint(model, trainer, 4, ckpt)
        q.put((rank, bool(torch.equal(others[0], others[1])), [h[1] for h in hist], len(trainer._last_plan.buckets)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_yaml_driven_data_parallel_training(tmp_path):
    """two ranks (one device, gloo) run titanet_amd.train.run from the parameters.yml schema on their shards of the global
    batch: replicas identical after 4 steps, different shards (different losses), overlapped buckets in use, rank-0
    checkpoint loadable with weights_only=True and resumable."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ck = str(tmp_path / "dp.pth")
    procs = [ctx.Process(target=_train_worker, args=(r, 2, str(tmp_path / "rdv"), ck, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "replicas diverged"
    assert res[0][2] != res[1][2], "both ranks saw the same shard"
    assert res[0][3] == 3, res[0][3]            # tail bucket + 2 groups of mega blocks
    d = torch.load(ck, weights_only=True)
    assert d["epoch"] == 4 and d["dropout_stream"]["step"] == 4


def test_bench_spawn_path_two_ranks_same_device():
    """`python bench.py --gpus 2` as the driver's 8-GPU box will first run it (bench.py's own mp.spawn, port pick,
    init_process_group, the collective probe, the per-rank spread gather), here with both ranks on cuda:0 over gloo: ONE JSON
    line with n_gpus 2, rccl_ranks 2, the rank spread and the whole-job value."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--steps", "2",
                          "--warmup", "1", "--batch", "64", "--no-cpu-baseline", "--no-other-configs", "--no-ceiling"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and cfg["rccl_ranks"] == 2 and cfg["global_batch"] == 128 and cfg["parallelism"] == "dp2"
    assert cfg["grad_groups"] == 2 and cfg["grad_groups_note"] and cfg["params_finite"]
    assert 0 < cfg["rank_ms_per_step"]["min"] <= cfg["rank_ms_per_step"]["max"]
    # the two multi-rank health numbers: host enqueue time per step, and what of the all-reduce backward did not hide
    assert cfg["host_enqueue_ms_per_step"] > 0
    assert cfg["exposed_allreduce_ms"] is not None and 0 <= cfg["exposed_allreduce_ms"]["median"] <= cfg["exposed_allreduce_ms"]["max"]
    assert abs(d["value"] - 128 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3 and d["scaling"] == "weak"
