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

    def __init__(self, rank=0, world=1, group=None, host_staged=False):
        self.rank, self.world, self.group, self.host_staged = rank, world, group, bool(host_staged)

    @property
    def active(self):
        return self.world > 1


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
