"""CPU, world_size 2, gloo: the N > 1 path (shard bounds, equal and ragged assembly, anchor-axis sharding).
The per-shard operator is injected (the CPU oracle) — the product's own operator is HIP-only by design."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_op(variant):
    from oracle import oracle as O

    def op(b1, b2, is_aligned=False, **kw):
        f = O.iou_aligned if is_aligned else O.iou_pairwise
        return torch.from_numpy(f(b1.numpy(), b2.numpy(), variant=variant))
    return op


def _worker(rank, world, port, n, ragged, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from sph_retina_amd import parallel as P
        b1 = torch.from_numpy(O.generate_boxes(n, 0))
        b2 = torch.from_numpy(O.generate_boxes(n, 1))
        if ragged:
            lo, hi = P.shard_bounds(n, world, rank)
        else:
            per = n // world
            lo, hi = rank * per, (rank + 1) * per
        full = P.sharded_aligned_iou(b1[lo:hi], b2[lo:hi], op=_oracle_op('efficient'))
        gt = b1[:5]
        pw_local = P.sharded_pairwise_iou(gt, b2[lo:hi], op=_oracle_op('efficient'))
        pw_full = P.sharded_pairwise_iou(gt, b2[lo:hi], op=_oracle_op('efficient'), gather=True)
        q.put((rank, lo, hi, full.numpy(), pw_local.numpy(), pw_full.numpy()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('n,ragged', [(64, False), (77, True)])
def test_two_rank_sharded_iou_matches_single_process(n, ragged):
    from oracle import oracle as O
    O.build()
    world, port = 2, 29500 + (os.getpid() % 2000) + (1 if ragged else 0)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b1, b2 = O.generate_boxes(n, 0), O.generate_boxes(n, 1)
    used = n if ragged else (n // world) * world
    want = O.iou_aligned(b1[:used], b2[:used], variant='efficient')
    want_pw = O.iou_pairwise(b1[:5], b2[:used], variant='efficient')
    covered = 0
    for rank, lo, hi, full, pw_local, pw_full in sorted(results):
        np.testing.assert_array_equal(full, want)          # every rank holds the whole vector, pair order kept
        np.testing.assert_array_equal(pw_local, want_pw[:, lo:hi])
        np.testing.assert_array_equal(pw_full, want_pw)
        covered += hi - lo
    assert covered == used


def test_shard_bounds_partition():
    from sph_retina_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 1_000_000, 8_000_001):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


# ---- sharded assigner: local column maxima, ONE all_reduce(MAX) of k packed keys, local finalize ----
class _OracleAssignOp:
    """CPU stand-in for the HIP reduce / finalize pair: the shard's (k, n_r) overlaps from the C oracle and the numpy
    restatement of assign_wrt_overlaps' steps with the GLOBAL per-GT maxima (test infrastructure)."""

    def __init__(self, ov_fn):
        self.ov_fn = ov_fn

    def reduce(self, gt, shard, col_offset, ignore_mask=None):
        from sph_retina_amd import parallel as P
        ov = torch.from_numpy(self.ov_fn(gt.numpy(), shard.numpy()))
        if ignore_mask is not None:
            ov[:, ignore_mask] = -1
        mx, am = ov.max(dim=1)
        return P.pack_assign_keys(mx, am + col_offset), (ov, col_offset)

    def finalize(self, ctx, keys, gt_labels, pos, neg_lo, neg_hi, min_pos, mlq, assign_all):
        from sph_retina_amd import parallel as P
        ov, off = ctx
        gmax, garg = P.unpack_assign_keys(keys)
        mo, amo = ov.max(dim=0)
        gi = torch.full((ov.size(1),), -1, dtype=torch.int64)
        gi[(mo >= neg_lo) & (mo < neg_hi)] = 0
        gi[mo >= pos] = amo[mo >= pos] + 1
        if mlq:
            for i in range(ov.size(0)):
                if gmax[i] >= min_pos:
                    if assign_all:
                        gi[ov[i] == gmax[i]] = i + 1
                    elif off <= int(garg[i]) < off + ov.size(1):
                        gi[int(garg[i]) - off] = i + 1
        lab = None
        if gt_labels is not None:
            lab = torch.full_like(gi, -1)
            lab[gi > 0] = gt_labels[gi[gi > 0] - 1]
        return gi, mo, lab


def _assign_scene():
    """12 GT x 301 anchors: GT 0's best anchor lives on rank 1 only, GT 1's best value occurs on BOTH ranks (an exact copy of
    the anchor: the smaller global column must win the argmax and both must be assigned), GT 2 overlaps nothing."""
    from oracle import oracle as O
    gt = O.generate_boxes(12, 5, alpha=(10, 60), beta=(10, 60), phi=(30, 150))
    rng = np.random.default_rng(9)
    anchors = gt[rng.integers(0, 12, 301)] + rng.standard_normal((301, 4)).astype(np.float32) * 6
    anchors[:, 0] %= 360
    anchors[:, 1] = anchors[:, 1].clip(20, 179)
    anchors[:, 2:] = anchors[:, 2:].clip(2, 100)
    anchors[:150][np.abs(anchors[:150, 0] - gt[0, 0]) < 40] += np.float32(120)   # nothing near GT 0 on rank 0
    anchors[:, 0] %= 360
    anchors[250] = gt[0] + np.float32(0.5)
    anchors[40] = gt[1] + np.float32(0.25)
    anchors[200] = anchors[40]
    gt[2] = np.array([10.0, 1.0, 1.0, 1.0], np.float32)
    return gt, anchors, rng.integers(0, 37, 12)


CFGS_ASSIGN = (dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0),
               dict(pos_iou_thr=0.6, neg_iou_thr=(0.1, 0.5), min_pos_iou=0.2, gt_max_assign_all=False),
               dict(pos_iou_thr=0.5, neg_iou_thr=0.5, match_low_quality=False))


def _assign_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from sph_retina_amd import parallel as P
        gt, anchors, labels = _assign_scene()
        lo, hi = P.shard_bounds(len(anchors), world, rank)
        op = _OracleAssignOp(lambda a, b: O.iou_pairwise(a, b, variant='standard'))
        ign = torch.zeros(len(anchors), dtype=torch.bool)
        ign[::17] = True
        out = []
        for kw in CFGS_ASSIGN:
            for mask in (None, ign[lo:hi]):
                gi, mo, lab = P.sharded_assign(torch.from_numpy(gt), torch.from_numpy(anchors[lo:hi]), lo, torch.from_numpy(labels),
                                               ignore_mask=mask, op=op, **kw)
                out.append((gi.numpy(), mo.numpy(), lab.numpy()))
        q.put((rank, lo, hi, out))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_sharded_assign_equals_single_process():
    from oracle import oracle as O
    O.build()
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_assign_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gt, anchors, labels = _assign_scene()
    ov = O.iou_pairwise(gt, anchors, variant='standard')
    # the scene holds what it claims
    assert ov[0].argmax() == 250 and ov[0, :150].max() < ov[0, 250]
    assert ov[1, 40] == ov[1, 200] == ov[1].max() and ov[2].max() == 0
    ign = np.zeros(len(anchors), bool)
    ign[::17] = True
    idx = 0
    for kw in CFGS_ASSIGN:
        for masked in (False, True):
            o = ov.copy()
            if masked:
                o[:, ign] = -1
            gi, mo, _a, _g, _ga, lab = O.assign_wrt_overlaps(o, labels, **kw)
            got_gi = np.concatenate([r[3][idx][0] for r in results])
            got_mo = np.concatenate([r[3][idx][1] for r in results])
            got_lab = np.concatenate([r[3][idx][2] for r in results])
            np.testing.assert_array_equal(got_gi, gi)
            np.testing.assert_array_equal(got_mo, mo)
            np.testing.assert_array_equal(got_lab, lab)
            idx += 1
    assert results[0][2] == results[1][1]   # contiguous shards


def test_assign_keys_order_and_round_trip():
    from sph_retina_amd import parallel as P
    v = torch.tensor([0.0, 0.5, 1.0, -1.0, 0.25, 0.25, 1e-30, 0.99999994])
    i = torch.tensor([3, 7, 0, 5, 9, 2, 2147483000, 1])
    k = P.pack_assign_keys(v, i)
    vv, ii = P.unpack_assign_keys(k)
    assert torch.equal(vv, v) and torch.equal(ii, i)
    order = torch.argsort(k)
    assert v[order].tolist() == sorted(v.tolist())
    assert i[order][3:5].tolist() == [9, 2]            # equal values: the larger key is the SMALLER index
    assert int(torch.maximum(k[4], k[5])) == int(k[5])
