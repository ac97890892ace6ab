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
