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


# ---- sharded MaxIoUAssigner ("per-anchor max local; per-GT max = all_reduce(MAX) of k values + index resolution") ----
def pack_assign_keys(values, indices):
    """Per-GT (max IoU, global column) as int64 keys whose SIGNED order is (value ascending, then index DESCENDING), so that
    `all_reduce(MAX)` yields the maximum value and, among equal values, the smallest global column — `torch.max(dim=1)` of the
    unsharded matrix.  The exchange format of `sph2pob_iou_assign_reduce_f32` / `_finalize_f32` (include/sph2pob_hip.h):
    high word = the float's bits mapped to an order-preserving int32, low word = (0x7fffffff - index) << 1."""
    bits = values.contiguous().view(torch.int32).to(torch.int64)
    hi = torch.where(bits >= 0, bits, ~(bits & 0x7fffffff))
    return (hi << 32) | ((0x7fffffff - indices.to(torch.int64)) << 1)


def unpack_assign_keys(keys):
    """-> (values fp32, global column indices int64)."""
    hi = keys >> 32                                                     # arithmetic shift: the signed high word
    bits = torch.where(hi >= 0, hi, (~hi) | 0x80000000)                 # back to the float's bit pattern (as 0 .. 2^32 - 1)
    values = torch.where(bits >= 0x80000000, bits - (1 << 32), bits).to(torch.int32).view(torch.float32)
    return values, 0x7fffffff - ((keys & 0xffffffff) >> 1)


class _HipAssignOp:
    """The per-shard operator on the MI355X: the two halves of the fused assigner through the C ABI."""

    def __init__(self, variant='standard', rbb_edge='arc'):
        self.variant, self.rbb_edge = variant, rbb_edge

    def reduce(self, gt, shard, col_offset, ignore_mask=None):
        from . import _lib, _torch_glue as G
        lib = _lib.lib()
        G.require_hip(gt, shard)
        gt, shard = G.as_f32_nograd(gt), G.as_f32_nograd(shard)
        k, n, dim = gt.size(0), shard.size(0), gt.size(1)
        dev = shard.device
        ws, state = G.assign_workspace(dev, lib.sph2pob_iou_assign_workspace_bytes(k, n), lib.sph2pob_iou_assign_state_bytes(k, n))
        keys = torch.empty((k,), dtype=torch.int64, device=dev)
        ign = None if ignore_mask is None else ignore_mask.to(device=dev, dtype=torch.uint8).contiguous()
        try:
            G.call('sph2pob_iou_assign_reduce_f32', dev, G.ptr(gt), k, G.ptr(shard), n, dim, G.VARIANTS[self.variant],
                   G.EDGES[self.rbb_edge], G.ptr(ign), int(col_offset), None, G.ptr(keys), G.ptr(ws), G.ptr(state), G.raw_stream_of(dev))
        except Exception:
            G.drop_assign_workspace(dev)
            raise
        return keys, (gt, shard, int(col_offset), ws)

    def finalize(self, ctx, keys, gt_labels, pos_iou_thr, neg_lo, neg_hi, min_pos_iou, match_low_quality, gt_max_assign_all):
        from . import _torch_glue as G
        gt, shard, col_offset, ws = ctx
        k, n, dim = gt.size(0), shard.size(0), gt.size(1)
        dev = shard.device
        max_ov = torch.empty((n,), dtype=torch.float32, device=dev)
        gt_inds = torch.empty((n,), dtype=torch.int64, device=dev)
        labels = gl = None
        if gt_labels is not None:
            gl = gt_labels.to(device=dev, dtype=torch.int64).contiguous()
            labels = torch.empty((n,), dtype=torch.int64, device=dev)
        G.call('sph2pob_iou_assign_finalize_f32', dev, G.ptr(gt), k, G.ptr(shard), n, dim, G.VARIANTS[self.variant],
               G.EDGES[self.rbb_edge], col_offset, G.ptr(keys), pos_iou_thr, neg_lo, neg_hi, min_pos_iou, int(bool(match_low_quality)),
               int(bool(gt_max_assign_all)), G.ptr(gl), G.ptr(max_ov), None, None, None, G.ptr(gt_inds), G.ptr(labels), G.ptr(ws),
               G.raw_stream_of(dev))
        return gt_inds, max_ov, labels


def sharded_assign(gt_bboxes, anchors_shard, col_offset, gt_labels=None, pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0,
                   gt_max_assign_all=True, match_low_quality=True, ignore_mask=None, op='standard', group=None):
    """`MaxIoUAssigner.assign` (mmdet/core/bbox/assigners/max_iou_assigner.py:113, :135-220) for THIS RANK's slice of the
    anchor axis, equal to the slice [col_offset, col_offset + n_r) of the single-process result bit for bit.

    The GT rows are replicated; `col_offset` is the global index of the shard's first anchor (`shard_bounds`).  Per-anchor
    max / argmax over the GTs is local.  The per-GT max over ALL anchors — the low-quality step needs it
    (`overlaps[i, :] == gt_max_overlaps[i]`, :200-207) — is ONE `all_reduce(MAX)` of k packed int64 keys (512 bytes for 64
    GTs); the (k, n) matrix is neither gathered nor materialised.  `ignore_mask`: this shard's slice of the columns the
    reference sets to -1 (:115-126).  Returns (assigned_gt_inds, max_overlaps, assigned_labels | None) of the shard;
    `gt_argmax_overlaps` are global column indices inside the keys (`unpack_assign_keys`).

    `op`: 'standard' | 'efficient' (the HIP kernels), or an object with the same `reduce` / `finalize` pair (the gloo
    tests inject a CPU operator: the product's own is HIP-only by design)."""
    if isinstance(neg_iou_thr, (tuple, list)):
        neg_lo, neg_hi = float(neg_iou_thr[0]), float(neg_iou_thr[1])
    else:
        neg_lo, neg_hi = 0.0, float(neg_iou_thr)
    fn = _HipAssignOp(op) if isinstance(op, str) else op
    keys, ctx = fn.reduce(gt_bboxes, anchors_shard, col_offset, ignore_mask)
    world, _rank = _world(group)
    if world > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MAX, group=group)   # stream-ordered on RCCL: no host synchronisation
    return fn.finalize(ctx, keys, gt_labels, pos_iou_thr, neg_lo, neg_hi, min_pos_iou, match_low_quality, gt_max_assign_all)
