"""Multi-GPU plumbing of the inference path: images are independent, so ranks share nothing on the data path
("replicas + batch split").  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only
to line ranks up and to agree on the slowest rank's time."""
import os

import torch


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) slice of n_items for this rank (first n%world ranks get one extra)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def init_from_env(backend):
    """Process group from torchrun's env (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); None when single-rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world, **kw)
    return dist


def barrier_sync(dist, device=None):
    if dist is not None:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def whole_job_throughput(dist, elapsed_s, units_this_rank, device=None):
    """(max elapsed over ranks, total units over ranks / that time): the contract of bench.py's `value`."""
    if dist is None:
        return elapsed_s, units_this_rank / elapsed_s
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    u = torch.tensor([float(units_this_rank)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item()) / float(t.item())
