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
