"""Multi-GPU sharding of the Sph2Pob path: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Box pairs are independent, so the path shards with NO data-path collective: every rank evaluates a contiguous
slice of the pairs (aligned) or of the anchor axis (pairwise; the <=64 GT rows are replicated).  The only exchange
is the optional assembly of the per-shard IoU vectors with ONE `all_gather_into_tensor`, enqueued on the compute
stream right behind the kernel (RCCL is stream-ordered: no host synchronisation between kernel and collective).
The reference never shards this path (its collectives are DDP's).

`op` is the per-shard operator (default: the HIP kernels).  Tests inject a CPU operator to exercise the sharding
and assembly logic under gloo without a GPU.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank):
    """Contiguous slice [lo, hi) of n items owned by `rank`; the first n % world_size ranks get one extra item."""
    base, rem = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _world(group):
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def gather_shards(shard, counts=None, group=None, out=None):
    """Assemble per-rank 1-D (or row-sharded N-D) results on every rank, in rank order.
    Equal shard sizes take the single-collective path `all_gather_into_tensor`; ragged shards are padded to the
    largest shard and trimmed.  `counts` (list of per-rank sizes) avoids the size exchange."""
    world, rank = _world(group)
    if world == 1:
        return shard
    if counts is None:
        mine = torch.tensor([shard.size(0)], dtype=torch.int64, device=shard.device)
        allc = torch.empty(world, dtype=torch.int64, device=shard.device)
        dist.all_gather_into_tensor(allc, mine, group=group)
        counts = [int(c) for c in allc.tolist()]
    assert counts[rank] == shard.size(0), 'counts[rank] must match the local shard'
    tail = tuple(shard.shape[1:])
    if len(set(counts)) == 1:
        if out is None:
            out = torch.empty((sum(counts),) + tail, dtype=shard.dtype, device=shard.device)
        dist.all_gather_into_tensor(out, shard.contiguous(), group=group)
        return out
    mx = max(counts)
    padded = shard.new_zeros((mx,) + tail)
    padded[:shard.size(0)] = shard
    buf = torch.empty((world * mx,) + tail, dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * mx:r * mx + counts[r]] for r in range(world)], dim=0)


def _default_op(name):
    from . import iou as _iou
    return getattr(_iou, name)


def sharded_aligned_iou(bboxes1, bboxes2, op='sph2pob_standard_iou', gather=True, counts=None, group=None, **kw):
    """Aligned IoU of THIS RANK's shard of pairs; with gather=True every rank gets the whole IoU vector
    (rank order = pair order when shards are the contiguous slices of `shard_bounds`)."""
    fn = _default_op(op) if isinstance(op, str) else op
    local = fn(bboxes1, bboxes2, is_aligned=True, **kw)
    return gather_shards(local, counts, group) if gather else local


def sharded_pairwise_iou(gt_bboxes, anchors_shard, op='sph2pob_standard_iou', gather=False, counts=None, group=None,
                         **kw):
    """(k, n_r) overlaps of the replicated GT rows against THIS RANK's slice of the anchor axis (the MaxIoUAssigner
    call pattern, mmdet/core/bbox/assigners/max_iou_assigner.py:113).  Per-anchor max/argmax is local; gather=True
    assembles the full (k, n) matrix (column order = rank order)."""
    fn = _default_op(op) if isinstance(op, str) else op
    local = fn(gt_bboxes, anchors_shard, is_aligned=False, **kw)
    if not gather:
        return local
    return gather_shards(local.t().contiguous(), counts, group).t()
