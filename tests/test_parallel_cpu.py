"""Data-parallel layer on CPU: world_size-2 gloo processes all-reduce a flat gradient buffer in buckets
and end up with identical (mean) gradients; bucket ranges tile the buffer."""
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from titanet_amd.trainer import FlatAllReducer, bucket_ranges


def test_bucket_ranges_tile_the_buffer():
    for n in (1, 1000, 1024, 6_200_059, 24_690_368):
        for nb in (1, 3, 4, 8):
            r = bucket_ranges(n, nb)
            assert r[0][0] == 0 and r[-1][1] == n and len(r) <= nb
            for (a, b), (c, d) in zip(r, r[1:]):
                assert b == c and a < b
            assert all(lo % 1024 == 0 for lo, _ in r)


def _worker(rank, world, init_file, n, q):
    # file rendezvous: no TCP-store port to lose to another process between "pick a free port" and "bind it"
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(42 + rank)          # per-rank shard seed, as bench.py
    grad = torch.randn(n, generator=g)
    red = FlatAllReducer(n_buckets=4)
    red.all_reduce_(grad)
    grad /= world
    q.put((rank, grad[:5].clone(), float(grad.sum())))
    dist.barrier()                                        # nobody tears its connections down while the peer still reduces
    dist.destroy_process_group()


def _run_world2(n, world, tmp_path, attempt):
    init_file = tmp_path / f"rendezvous_{attempt}"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, str(init_file), n, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return res


def test_flat_allreduce_world2_gloo(tmp_path):
    n, world = 300_001, 2
    res = None
    for attempt in range(3):                              # the loopback rendezvous of two fresh processes is retried, the maths is not
        try:
            res = _run_world2(n, world, tmp_path, attempt)
            break
        except Exception as e:                            # noqa: BLE001 - queue timeout / connection reset during set-up
            last = e
    assert res is not None, last
    want = sum(torch.randn(n, generator=torch.Generator().manual_seed(42 + r)) for r in range(world)) / world
    for rank, head, total in res:
        assert torch.allclose(head, want[:5], atol=1e-6)
        assert abs(total - float(want.sum())) < 1e-2
