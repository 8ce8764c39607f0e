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


# ---------------------------------------------------------------------------------------------------------------------
# Training: gradient exchange over the packed fp32 gradient arena.
#
# The backward (dmx_unet_train_backward) writes every parameter gradient into one flat fp32 arena and completes it in
# 11 buckets (conv_out, up_blocks 3..0, mid, down_blocks 3..0, conv_in + time embedding), recording an event per bucket.
# Each bucket is reduced on a side stream as soon as its event fires, so the exchange runs under the rest of the
# backward; the ranges are contiguous slices of the arena (no flattening copies).  RCCL picks ring / direct algorithms
# per message size over the xGMI links; the ranges skip the arena regions that hold derived (non-trainable) data.

def merge_ranges(ranges, gap=0):
    """Sorted union of [begin, end) ranges; ranges closer than `gap` are fused."""
    out = []
    for b, e in sorted(ranges):
        if out and b <= out[-1][1] + gap:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([b, e])
    return [(b, e) for b, e in out]


def clip_ranges(ranges, lo, hi):
    """Parts of the ranges that lie inside [lo, hi)."""
    out = []
    for b, e in ranges:
        b2, e2 = max(b, lo), min(e, hi)
        if b2 < e2:
            out.append((b2, e2))
    return out


def plan_buckets(param_ranges, bucket_ranges, itemsize=4, gap=4096):
    """param_ranges: byte ranges of the trainable gradients; bucket_ranges: list (in completion order) of lists of byte
    ranges.  Returns, per bucket, the element ranges to all-reduce (merged, clipped, multiples of itemsize)."""
    merged = merge_ranges(param_ranges, gap)
    plan = []
    for spans in bucket_ranges:
        rs = []
        for lo, hi in spans:
            rs += clip_ranges(merged, lo, hi)
        plan.append([(b // itemsize, (e + itemsize - 1) // itemsize) for b, e in rs])
    return plan


def reduce_buckets(flat, plan, dist, group=None, wait_bucket=None, average_by=None, mode="rs_ag"):
    """Sum the planned ranges of the 1-D tensor `flat` over the ranks, bucket by bucket in completion order;
    wait_bucket(i) is called before bucket i is touched (GPU: make the side stream wait for the bucket's event).

    mode "rs_ag" (default): each contiguous range is exchanged as an in-place REDUCE-SCATTER (rank r ends
    up owning the summed r-th 1/world slice) followed by an in-place ALL-GATHER of the slices - on a fully connected xGMI
    node both halves are direct exchanges with the 7 peers (every link carries 1/world of the range each way) instead of
    whatever ring RCCL's all-reduce heuristic would pick for that message size; the 1/world averaging touches only the
    owned slice.  The < world leftover elements of a range go through one tiny all-reduce.
    mode "all_reduce": one all-reduce per range (round 1 behaviour; kept for A/B runs)."""
    world = dist.get_world_size(group) if hasattr(dist, "get_world_size") else 1
    rank = dist.get_rank(group) if hasattr(dist, "get_rank") else 0
    for i, ranges in enumerate(plan):
        if wait_bucket is not None:
            wait_bucket(i)
        for b, e in ranges:
            v = flat[b:e]
            n = e - b
            main = (n // world) * world if mode == "rs_ag" else 0      # (world 1 too: the in-place aliasing of the two collectives is then exercised on a 1-rank RCCL group)
            if main:
                c = main // world
                body = v[:main]
                mine = body[rank * c:(rank + 1) * c]
                dist.reduce_scatter_tensor(mine, body, group=group)        # in place: output is the rank's own slice of the input
                if average_by:
                    mine.div_(average_by)
                dist.all_gather_into_tensor(body, mine, group=group)       # in place
            if main < n:
                tail = v[main:]
                dist.all_reduce(tail, group=group)
                if average_by:
                    tail.div_(average_by)


class GradientAccumulator:
    """Bookkeeping of `accelerator.accumulate(unet)` (train_diffute_v1.py:873,926) for the in-backward exchange: on the non-boundary
    micro-steps of a gradient-accumulation window the reference (DDP under `no_sync`) skips the all-reduce and keeps the local gradient; only
    the boundary micro-step exchanges - the ACCUMULATED gradient.  Here the backward WRITES the packed gradient arena, so the local sum of the
    skipped micro-steps lives in a second arena (`acc`); at the boundary it is added to the fresh gradient bucket by bucket ON THE EXCHANGE
    STREAM, right behind that bucket's completion event and in front of its collective: one exchange per optimizer step (not one per
    micro-step), still overlapped with the backward.  Pure torch: shared by UNet2DConditionModel._train_backward and the gloo tests."""

    def __init__(self, accumulate_steps=1):
        self.n = max(1, int(accumulate_steps))
        self.micro = 0          # backward passes since the last exchange
        self.skip_ctx = 0       # depth of no_sync() contexts
        self.acc = None
        self.acc_n = 0          # micro-step gradients summed in `acc`

    def boundary(self):
        """does the backward that is about to run exchange?"""
        return self.skip_ctx == 0 and (self.micro + 1) % self.n == 0

    def stash(self, flat):
        """a non-boundary backward has written `flat`: keep its gradient"""
        if self.acc is None or self.acc.shape != flat.shape or self.acc.device != flat.device:
            self.acc = torch.empty_like(flat)
        if self.acc_n == 0:
            self.acc.copy_(flat)
        else:
            self.acc.add_(flat)
        self.acc_n += 1
        self.micro += 1

    def pre_add(self, flat, ranges):
        """boundary step, bucket complete: flat[ranges] += what the skipped micro-steps left (call on the exchange stream)"""
        if self.acc_n:
            for b, e in ranges:
                flat[b:e].add_(self.acc[b:e])

    def exchanged(self):
        self.acc_n = 0
        self.micro = 0


def broadcast_parameters(params, dist, src=0, group=None, module=None):
    """D3 (accelerator.prepare, train_diffute_v1.py:780): every rank starts from rank `src`'s parameters.

    The broadcast writes the Parameter itself under no_grad (an in-place write that bumps `p._version`), not `p.data`:
    writes through `.data` are invisible to the packed-arena change detection of the HIP models (their version counter is
    detached), so a broadcast after the first forward / after FusedAdamW(unet) would leave the arena and the fp32 masters
    on the old weights.  Pass `module=` (or call `module.mark_parameters_changed()` yourself) to force the re-pack whatever
    the backend does to the version counter."""
    params = list(params)
    with torch.no_grad():
        for p in params:
            dist.broadcast(p, src=src, group=group)
    if module is not None and hasattr(module, "mark_parameters_changed"):
        module.mark_parameters_changed()


def gather_scalar(value, dist, world, device=None):
    """D2 (accelerator.gather(loss.repeat(bs)).mean(), train_diffute_v1.py:921): mean of a per-rank scalar."""
    t = torch.tensor([float(value)], dtype=torch.float32, device=device if device is not None else torch.device("cpu"))
    dist.all_reduce(t)
    return float(t.item()) / world
