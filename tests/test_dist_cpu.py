"""N>1 path on CPU: gloo, world_size 2 (the GPU run uses the same code over RCCL)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from diffute_amd import dist as D
    dist = D.init_from_env("gloo")
    lo, hi = D.shard_range(10, rank, world)
    D.barrier_sync(dist)
    elapsed = 0.5 if rank == 0 else 2.0                 # the slow rank defines the job time
    t, thr = D.whole_job_throughput(dist, elapsed, hi - lo)
    q.put((rank, lo, hi, t, thr))
    D.barrier_sync(dist)
    dist.destroy_process_group()


def test_weak_scaling_aggregation_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert [(r[1], r[2]) for r in res] == [(0, 5), (5, 10)]          # disjoint, covering shards
    for r in res:
        assert abs(r[3] - 2.0) < 1e-9 and abs(r[4] - 10 / 2.0) < 1e-9     # max time over ranks, total units / that time


def test_shard_range_properties():
    from diffute_amd.dist import shard_range
    for n in (0, 1, 7, 64):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


# ---------------------------------------------------------------------------------------------- training exchange (D1-D3)
def test_bucket_planning():
    from diffute_amd.dist import clip_ranges, merge_ranges, plan_buckets
    assert merge_ranges([(40, 48), (0, 16), (16, 24), (100, 120)]) == [(0, 24), (40, 48), (100, 120)]
    assert merge_ranges([(0, 16), (20, 32)], gap=8) == [(0, 32)]
    assert clip_ranges([(0, 24), (40, 48)], 8, 44) == [(8, 24), (40, 44)]
    # two buckets in completion order (the second one first in memory), a derived (non-trainable) hole at [64, 128)
    params = [(0, 64), (128, 192), (192, 200), (512, 600)]
    plan = plan_buckets(params, [[(256, 1024)], [(0, 256)]], itemsize=4, gap=0)
    assert plan == [[(128, 150)], [(0, 16), (32, 50)]]
    assert plan_buckets(params, [[(0, 1024)]], itemsize=4, gap=64) == [[(0, 50), (128, 150)]]


def _train_sync_worker(rank, world, port, q, mode="rs_ag"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from diffute_amd import dist as D
    dist = D.init_from_env("gloo")
    # a fake gradient arena: trainable ranges get rank-dependent values, the holes a sentinel that must survive
    n = 4096
    flat = torch.full((n,), -7.0)
    params = [(0, 1000 * 4), (1500 * 4, 3000 * 4), (3501 * 4, 4096 * 4)]
    for b, e in params:
        flat[b // 4:e // 4] = torch.arange(b // 4, e // 4, dtype=torch.float32) * (rank + 1)
    buckets = [[(2048 * 4, 4096 * 4)], [(0, 2048 * 4)]]                 # completion order: back half first
    plan = D.plan_buckets(params, buckets, gap=0)
    order = []
    D.reduce_buckets(flat, plan, dist, wait_bucket=order.append, average_by=world, mode=mode)
    w = [torch.ones(3) * (rank + 5)]
    D.broadcast_parameters(w, dist, src=0)
    mean_loss = D.gather_scalar(float(rank + 1), dist, world)
    q.put((rank, flat.numpy().copy(), order, w[0].numpy().copy(), mean_loss))     # numpy: no fd passing races with process exit
    D.barrier_sync(dist)
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("mode", ["rs_ag", "all_reduce"])
def test_bucketed_gradient_exchange_gloo_world2(mode):
    """D1/D2/D3 with gloo, world_size 2, for both exchange schedules (in-place reduce-scatter + all-gather per arena slice -
    slices here have odd lengths, so the < world leftover path runs too - and one all-reduce per slice): bucket by bucket in
    completion order, only the trainable ranges are touched, the result is the mean over ranks and identical on every rank;
    parameters broadcast from rank 0; scalar loss mean."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_train_sync_worker, args=(r, world, port, q, mode)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    idx = torch.arange(4096, dtype=torch.float32)
    want = torch.full((4096,), -7.0)
    for b, e in [(0, 1000), (1500, 3000), (3501, 4096)]:
        want[b:e] = idx[b:e] * (1 + 2) / 2.0                             # mean of rank-scaled values
    for rank, flat, order, w, mean_loss in res:
        assert torch.equal(torch.from_numpy(flat), want), f"rank {rank}: reduced arena differs"
        assert order == [0, 1]
        assert torch.equal(torch.from_numpy(w), torch.ones(3) * 5)
        assert abs(mean_loss - 1.5) < 1e-6



def _accum_worker(rank, world, port, q, n_acc, mode):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from diffute_amd import dist as D
    dist = D.init_from_env("gloo")
    calls = {"n": 0}

    class Counting:                                        # the process group seen through a counter of its collectives
        def __getattr__(self, name):
            f = getattr(dist, name)
            if name in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor"):
                def g(*a, **k):
                    calls["n"] += 1
                    return f(*a, **k)
                return g
            return f
    cd = Counting()
    n = 1024
    params = [(0, 700 * 4), (800 * 4, 1024 * 4)]
    buckets = [[(512 * 4, 1024 * 4)], [(0, 512 * 4)]]
    plan = D.plan_buckets(params, buckets, gap=0)
    acc = D.GradientAccumulator(accumulate_steps=n_acc)
    flat = torch.empty(n)
    micro = []
    exchanges = 0
    for k in range(2 * n_acc):                              # two windows: the state must reset at the boundary
        g = (torch.arange(n, dtype=torch.float32) * 0.25 + 3 * k + 100 * rank) / world      # what a backward writes (1 / world folded in)
        micro.append(g.clone())
        flat.copy_(g)                                       # the backward WRITES the arena
        if not acc.boundary():
            acc.stash(flat)
            continue
        D.reduce_buckets(flat, plan, cd, wait_bucket=lambda i: acc.pre_add(flat, plan[i]), mode=mode)
        acc.exchanged()
        exchanges += 1
        q.put((rank, k, flat.numpy().copy(), calls["n"]))
    # a no_sync() block: skip_ctx > 0 makes every backward non-boundary whatever n says
    acc2 = D.GradientAccumulator(1)
    acc2.skip_ctx += 1
    assert not acc2.boundary()
    acc2.skip_ctx -= 1
    assert acc2.boundary()
    D.barrier_sync(dist)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rs_ag", "all_reduce"])
def test_gradient_accumulation_times_sync_gloo_world2(mode):
    """`accelerator.accumulate(unet)` x DDP (train_diffute_v1.py:873,926): with accumulate_steps = 3 only every third backward exchanges, and what it
    exchanges is the window's accumulated gradient - the result equals the sum over ranks of the sum over the window's micro-steps (1 / world is folded
    into each backward), is identical on both ranks, the holes of the arena are untouched, the second window starts from zero, and the number of
    collectives per window is that of ONE exchange."""
    world, port, n_acc = 2, _free_port(), 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_accum_worker, args=(r, world, port, q, n_acc, mode)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(2 * world)]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    idx = torch.arange(1024, dtype=torch.float32)
    by = {}
    for rank, k, flat, ncalls in res:
        by.setdefault(k, {})[rank] = (torch.from_numpy(flat), ncalls)
    assert sorted(by) == [n_acc - 1, 2 * n_acc - 1], "exactly the boundary micro-steps exchanged"
    per_window = None
    for w, k in enumerate(sorted(by)):
        ks = range(w * n_acc, (w + 1) * n_acc)
        want_live = sum((idx * 0.25 + 3 * kk + 100 * r) / world for kk in ks for r in range(world))
        (f0, c0), (f1, c1) = by[k][0], by[k][1]
        assert torch.equal(f0[:700], f1[:700]) and torch.equal(f0[800:], f1[800:]), "ranks hold different exchanged gradients"
        for f in (f0, f1):
            assert torch.allclose(f[:700], want_live[:700], rtol=1e-6, atol=1e-4) and torch.allclose(f[800:], want_live[800:], rtol=1e-6, atol=1e-4)
        for r, f in ((0, f0), (1, f1)):                     # the hole [700, 800) is not a parameter: it keeps what the LAST backward of that rank wrote
            assert torch.equal(f[700:800], ((idx * 0.25 + 3 * k + 100 * r) / world)[700:800])
        assert c0 == c1
        per_window = c0 if per_window is None else per_window
        assert c0 == per_window * (w + 1), "one exchange worth of collectives per window"
