"""Data-parallel exchange steps of the MoCo iteration (one process per GPU, RCCL over xGMI via
torch.distributed backend "nccl"; the same code runs on gloo for the CPU logic tests).

Reference (tools/train_video_contrast_dis.py):
  :183-187 _global_gather   all_gather of k (b,128) -> all_k, rank-major         -> gather_keys
  :189-231 _shuffle_bn      all_gather of the raw key clips inside the node (b*G clips land on
                            EVERY GPU, 617 MB/GPU at config 3), 2 id broadcasts, keep b rows
                                                                                 -> shuffle_exchange
  DDP                       bucketed all-reduce (mean) of gradients              -> allreduce_mean_

MI355X-native choices: xGMI is point-to-point, so instead of materialising the whole node batch
on every GPU the ShuffleBN step moves each clip exactly once (row all-to-all: 1/G of the bytes);
the permutation comes from a seed shared by construction, so the two id broadcasts disappear; the
gradient all-reduce is ONE collective over the flat gradient arena (or a few large buckets
overlapped with backward) instead of per-tensor buckets.

Row packing/unpacking is delegated to a `gather(src, idx)` callable: the trainer passes the HIP
row-gather kernel; the gloo tests pass plain indexing.  There is deliberately no default.
"""
import torch
import torch.distributed as dist


class DistCtx(object):
    """rank / world / process group.  `host_staged` routes device tensors through pinned host memory around
    each collective: that is how the N>1 code path is rehearsed with the gloo backend when the ranks share
    one GPU (tests/test_gpu_dist.py); production (backend "nccl" = RCCL) leaves it False."""

    def __init__(self, rank=0, world=1, group=None, host_staged=False, force_active=False):
        self.rank, self.world, self.group, self.host_staged = rank, world, group, bool(host_staged)
        # force_active: run the multi-GPU control flow (exchange, gathers, bucketed all-reduce, segment graphs) even at
        # world size 1 -- how the RCCL branches are exercised on a one-GPU box (tests/test_gpu_dist.py)
        self.force_active = bool(force_active)

    @property
    def active(self):
        return self.world > 1 or self.force_active


def shared_permutation(n, seed, step):
    """Same permutation on every rank without communication (replaces the reference's randperm on
    each rank + broadcast from rank 0, :207-211).  Host generator => no device sync."""
    g = torch.Generator()
    g.manual_seed((int(seed) * 1000003 + int(step)) % (2 ** 63 - 1))
    return torch.randperm(n, generator=g)


def exchange_plan(shuffle_ids, b, rank, world):
    """Index bookkeeping for the row all-to-all.  shuffle_ids: host int64 (b*world,).
    Rank r must end up with this_x[j] = node_x[shuffle_ids[r*b + j]].
    Returns (send_idx, send_counts, recv_counts, place_idx):
      send_idx   local row indices in send order (grouped by destination rank)
      place_idx  this_x = received_rows[place_idx]"""
    ids = shuffle_ids.view(world, b)                      # ids[d] = rows rank d wants (global indices)
    owner = ids // b
    send_idx, send_counts = [], []
    for d in range(world):
        mine = ids[d][owner[d] == rank] - rank * b        # in the order d wants them
        send_idx.append(mine)
        send_counts.append(int(mine.numel()))
    want_owner = owner[rank]
    recv_counts = [int((want_owner == s).sum()) for s in range(world)]
    # received buffer is grouped by source rank s (ascending), each group in my wanted order
    order = torch.argsort(want_owner, stable=True)        # positions j sorted by source
    place_idx = torch.empty(b, dtype=torch.long)
    place_idx[order] = torch.arange(b)
    return torch.cat(send_idx), send_counts, recv_counts, place_idx


def shuffle_exchange(x, shuffle_ids, ctx, gather):
    """ShuffleBN input exchange: returns this_x with this_x[j] = node_x[shuffle_ids[rank*b+j]]."""
    b = x.shape[0]
    if not ctx.active:
        return gather(x, shuffle_ids.to(x.device))
    send_idx, send_counts, recv_counts, place_idx = exchange_plan(shuffle_ids, b, ctx.rank, ctx.world)
    send = gather(x, send_idx.to(x.device)) if send_idx.numel() else x[:0]
    recv = torch.empty((b,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if ctx.host_staged:
        recv_h = torch.empty(recv.shape, dtype=x.dtype)
        dist.all_to_all_single(recv_h, send.contiguous().cpu(), output_split_sizes=recv_counts,
                               input_split_sizes=send_counts, group=ctx.group)
        recv.copy_(recv_h)
    else:
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts,
                               group=ctx.group)
    return gather(recv, place_idx.to(x.device))


def gather_keys(k, ctx):
    """all_k = cat over ranks of k (rank-major order, :183-187)."""
    if not ctx.active:
        return k
    out = torch.empty((ctx.world * k.shape[0],) + tuple(k.shape[1:]), dtype=k.dtype, device=k.device)
    if ctx.host_staged:
        parts = [torch.empty(k.shape, dtype=k.dtype) for _ in range(ctx.world)]
        dist.all_gather(parts, k.contiguous().cpu(), group=ctx.group)
        out.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(out, k.contiguous(), group=ctx.group)
    return out


def unshuffle_keys(all_k, shuffle_ids, b, ctx, gather):
    """k[j] = all_k[reverse_ids[rank*b + j]]  (:225-229; single node, so node_k == all_k)."""
    reverse_ids = torch.argsort(shuffle_ids)
    mine = reverse_ids[ctx.rank * b:(ctx.rank + 1) * b]
    return gather(all_k, mine.to(all_k.device))


def allreduce_sum_(flat, ctx):
    """Gradient all-reduce over the flat arena (SUM; the 1/world of DDP's mean is folded into the loss
    gradient scale by the trainer, so no extra pass over the buffer)."""
    if ctx.active:
        if ctx.host_staged:
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=ctx.group)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
    return flat


class BucketReducer(object):
    """Gradient all-reduce in arena buckets, overlapped with the backward pass (what DDP does with its 25 MB buckets,
    tools/train_video_contrast_dis.py:143,419 -- here a few LARGE buckets: xGMI rings are per-link bound, so fewer,
    larger collectives win).  `launch(lo, hi)` starts the SUM all-reduce of flat[lo:hi] without blocking the host; the
    collective runs on the process group's own stream behind everything already queued on the current stream, so the
    backward kernels of the next stage overlap it.  `wait()` makes the current stream wait for all of them."""

    def __init__(self, flat, ctx):
        self.flat, self.ctx, self.pending = flat, ctx, []

    def launch(self, lo, hi):
        if not self.ctx.active or hi <= lo:
            return
        view = self.flat[lo:hi]
        if self.ctx.host_staged:                       # gloo rehearsal on one GPU: synchronous, through the host
            h = view.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.ctx.group)
            view.copy_(h)
        else:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.ctx.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()                                   # stream-side wait: the host does not block
        self.pending = []


def plan_buckets(first_closure, offsets, sizes, total, target_elems):
    """Partition the flat gradient arena [0, total) into contiguous buckets of about `target_elems` elements and give
    each the index of the backward closure after which it is complete.  first_closure[i] = smallest tape index that
    writes parameter i's gradient (None: never written).  Returns [(closure index, lo, hi)] sorted by closure index
    DESCENDING -- the order in which a reverse sweep of the tape completes them."""
    n = len(offsets)
    bounds, acc = [total], 0
    for i in range(n - 1, -1, -1):                     # walk the arena from its end (the head: finished first)
        acc += (offsets[i + 1] if i + 1 < n else total) - offsets[i]
        if acc >= target_elems and i > 0:
            bounds.append(offsets[i])
            acc = 0
    bounds.append(0)
    bounds = sorted(set(bounds))
    out = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        idx = [first_closure[i] for i in range(n) if lo <= offsets[i] < hi and first_closure[i] is not None]
        out.append((min(idx) if idx else 0, lo, hi))
    out.sort(key=lambda b: -b[0])
    return out


class ExchangePlans(object):
    """ShuffleBN index bookkeeping for step t prepared while step t-1 runs: the permutation comes from the shared seed
    (no broadcast), the send / place / un-shuffle indices are built on the host ahead of time and uploaded through
    pinned memory with non-blocking copies, so nothing but the collectives sits between the trainer's graph segments."""

    def __init__(self, b, ctx, seed, device):
        self.b, self.ctx, self.seed, self.device = b, ctx, seed, device
        self.cache = {}

    def _build(self, step, shuffle_ids=None):
        b, ctx = self.b, self.ctx
        ids = shared_permutation(b * ctx.world, self.seed, step) if shuffle_ids is None else shuffle_ids
        send_idx, send_counts, recv_counts, place_idx = exchange_plan(ids, b, ctx.rank, ctx.world)
        rev = torch.argsort(ids)[ctx.rank * b:(ctx.rank + 1) * b]
        on_gpu = torch.device(self.device).type == 'cuda'
        up = lambda t: (t.pin_memory() if (on_gpu and t.numel()) else t).to(self.device, non_blocking=True)
        return dict(ids=ids, send_idx=up(send_idx), place_idx=up(place_idx), unshuffle_idx=up(rev),
                    send_counts=send_counts, recv_counts=recv_counts)

    def get(self, step, shuffle_ids=None):
        if shuffle_ids is not None:
            return self._build(step, shuffle_ids)
        p = self.cache.pop(step, None)
        return p if p is not None else self._build(step)

    def prefetch(self, step):
        if step not in self.cache:
            self.cache[step] = self._build(step)


def shuffle_exchange_planned(x, plan, ctx, gather):
    """shuffle_exchange with the indices already on the device (ExchangePlans)."""
    b = x.shape[0]
    send = gather(x, plan['send_idx']) if plan['send_idx'].numel() else x[:0]
    recv = torch.empty((b,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if ctx.host_staged:
        recv_h = torch.empty(recv.shape, dtype=x.dtype)
        dist.all_to_all_single(recv_h, send.contiguous().cpu(), output_split_sizes=plan['recv_counts'],
                               input_split_sizes=plan['send_counts'], group=ctx.group)
        recv.copy_(recv_h)
    else:
        dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=plan['recv_counts'],
                               input_split_sizes=plan['send_counts'], group=ctx.group)
    return gather(recv, plan['place_idx'])


def broadcast_(t, ctx, src=0):
    """Initial queue / parameter sync (:233-241)."""
    if ctx.active:
        if ctx.host_staged:
            h = t.cpu()
            dist.broadcast(h, src, group=ctx.group)
            t.copy_(h)
        else:
            dist.broadcast(t, src, group=ctx.group)
    return t
