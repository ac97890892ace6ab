"""GPU: fused MaxIoUAssigner epilogue (sph2pob_assign_f32) vs the numpy restatement of mmdet's assign_wrt_overlaps,
mmdet's own known-answer cases, and the config-4 call pattern end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def A():
    import sph_retina_amd.bbox.assigners as assigners
    assert torch.cuda.is_available()
    return assigners


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def check(A, oracle, ov, labels=None, **kw):
    res, ex = A.assign_wrt_overlaps(cu(ov), None if labels is None else cu(labels), return_extras=True, **kw)
    gi, mo, amo, gm, gam, lab = oracle.assign_wrt_overlaps(ov, labels, **kw)
    np.testing.assert_array_equal(res.gt_inds.cpu().numpy(), gi)
    np.testing.assert_array_equal(res.max_overlaps.cpu().numpy(), mo)
    np.testing.assert_array_equal(ex['argmax_overlaps'].cpu().numpy(), amo)
    np.testing.assert_array_equal(ex['gt_max_overlaps'].cpu().numpy(), gm)
    np.testing.assert_array_equal(ex['gt_argmax_overlaps'].cpu().numpy(), gam)
    if labels is not None:
        np.testing.assert_array_equal(res.labels.cpu().numpy(), lab)
    return res


@pytest.mark.parametrize('k,n', [(1, 1), (3, 70), (64, 1000), (17, 4099), (200, 333), (32, 257), (33, 65), (65, 31)])
def test_random_matrices_bit_exact(A, oracle, k, n):
    rng = np.random.default_rng(k * 1000 + n)
    ov = rng.random((k, n)).astype(np.float32)
    ov[ov < 0.7] = 0.0                                   # sparse like real anchor/GT overlaps, many exact ties at 0
    ov[:, rng.integers(0, n, max(1, n // 10))] = -1.0     # ignored columns (max_iou_assigner.py:126)
    if n > 5:
        ov[0, 3] = ov[0, 5] = ov[0].max()                 # ties on a row maximum -> gt_max_assign_all hits both
    labels = rng.integers(0, 37, k)
    for kw in (dict(pos_iou_thr=0.8, neg_iou_thr=0.75, min_pos_iou=0.0),
               dict(pos_iou_thr=0.9, neg_iou_thr=(0.1, 0.8), min_pos_iou=0.75, gt_max_assign_all=False),
               dict(pos_iou_thr=0.8, neg_iou_thr=0.8, match_low_quality=False)):
        check(A, oracle, ov, labels, **kw)
        check(A, oracle, ov, None, **kw)


def test_mmdet_known_answers(A):
    """tests/test_utils/test_assigner.py:19-40 of the vendored mmdet: planar IoUs of its 4 boxes vs 2 GTs."""
    ov = np.array([[0.81, 0.0476, 0.0, 0.0], [0.0, 0.6328, 0.0, 0.0]], np.float32)  # IoU(gt, boxes), same ordering
    res = A.assign_wrt_overlaps(cu(ov), cu(np.array([2, 3])), pos_iou_thr=0.5, neg_iou_thr=0.5)
    assert res.gt_inds.tolist() == [1, 2, 0, 0]        # expected_gt_inds at test_assigner.py:38
    assert res.labels.tolist() == [2, 3, -1, -1]
    empty = A.assign_wrt_overlaps(torch.zeros((0, 4), device='cuda'), None, pos_iou_thr=0.5, neg_iou_thr=0.5)
    assert empty.gt_inds.tolist() == [0, 0, 0, 0] and empty.num_gts == 0
    none = A.assign_wrt_overlaps(torch.zeros((2, 0), device='cuda'), cu(np.array([1, 2])), pos_iou_thr=0.5, neg_iou_thr=0.5)
    assert none.gt_inds.numel() == 0 and none.labels.numel() == 0


@pytest.mark.parametrize('hw,count', [((512, 1024), 98208), ((1024, 2048), 392832)])
def test_config4_call_pattern_end_to_end(A, oracle, hw, count):
    """RetinaNet assigner of the reference config (pos 0.5 / neg 0.4 / min_pos 0): 64 GT x 98 208 anchors (the
    reference's default 512 x 1024 ERP: the "~100k" of BASELINE configs[3]) and x 392 832 (the literal 1024 x 2048 grid)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from bench_configs import retina_anchors
    anchors = retina_anchors(*hw)
    assert anchors.size(0) == count
    g = torch.Generator().manual_seed(0)
    u = torch.rand((64, 4), generator=g)
    gt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1).cuda()
    labels = torch.randint(0, 37, (64,), generator=g).cuda()
    assigner = A.SphMaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1)
    res = assigner.assign(anchors, gt, gt_labels=labels)
    assert res.gt_inds.shape == (anchors.size(0),) and res.num_gts == 64
    ov = assigner.iou_calculator(gt, anchors).cpu().numpy()
    gi, mo, *_rest, lab = oracle.assign_wrt_overlaps(ov, labels.cpu().numpy(), pos_iou_thr=0.5, neg_iou_thr=0.4)
    np.testing.assert_array_equal(res.gt_inds.cpu().numpy(), gi)
    np.testing.assert_array_equal(res.max_overlaps.cpu().numpy(), mo)
    np.testing.assert_array_equal(res.labels.cpu().numpy(), lab)
    assert (gi > 0).sum() >= 64  # every GT got at least its best anchor (low-quality matching)
    # and the overlaps themselves against the CPU oracle on a column sample
    cols = np.arange(0, anchors.size(0), 97)
    want = oracle.iou_pairwise(gt.cpu().numpy(), anchors.cpu().numpy()[cols], variant='standard')
    assert np.abs(ov[:, cols] - want).mean() < 1e-6


# ---- the fused route: no (k, n) matrix (sph2pob_iou_assign_f32), SURVEY §8f-1 ----
def _matrix_route(A, S, gt, boxes, labels, variant, ignore=None, **kw):
    """The two-step route the fused one must equal bit for bit: pairwise kernel -> (ignored columns = -1) -> sph2pob_assign_f32."""
    fn = S.sph2pob_standard_iou if variant == 'standard' else S.sph2pob_efficient_iou
    ov = fn(gt, boxes)
    if ignore is not None:
        ov[:, ignore] = -1
    res, ex = A.assign_wrt_overlaps(ov, labels, return_extras=True, **kw)
    return ov, res, ex


def _same(res, ex, res2, ex2, ov=None, ov2=None):
    assert torch.equal(res.gt_inds, res2.gt_inds)
    assert torch.equal(res.max_overlaps.view(torch.int32), res2.max_overlaps.view(torch.int32))
    if res.labels is not None or res2.labels is not None:
        assert torch.equal(res.labels, res2.labels)
    for key in ('argmax_overlaps', 'gt_argmax_overlaps'):
        assert torch.equal(ex[key], ex2[key]), key
    assert torch.equal(ex['gt_max_overlaps'].view(torch.int32), ex2['gt_max_overlaps'].view(torch.int32))
    if ov is not None:
        assert torch.equal(ov.view(torch.int32), ov2.view(torch.int32))


CFGS = (dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0),
        dict(pos_iou_thr=0.6, neg_iou_thr=(0.1, 0.5), min_pos_iou=0.3, gt_max_assign_all=False),
        dict(pos_iou_thr=0.5, neg_iou_thr=0.5, match_low_quality=False),
        dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.2))


def _scene(k, n, dim, seed):
    """GT + boxes with everything the epilogue has to get right: boxes scattered around the GTs, exact copies of boxes
    (column ties on a row maximum, far apart and inside one tile), a duplicated GT (row ties on a column maximum: first row
    wins, the later GT wins the low-quality step), a GT that overlaps nothing, far boxes that are culled."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand((k, 5), generator=g)
    gt = torch.stack([u[:, 0] * 360, 25 + u[:, 1] * 130, 5 + u[:, 2] * 60, 5 + u[:, 3] * 60, -60 + 120 * u[:, 4]], 1)[:, :dim]
    near = gt[torch.randint(0, k, (n // 2,), generator=g)] + torch.randn((n // 2, dim), generator=g) * 5
    v = torch.rand((n - n // 2, 5), generator=g)
    far = torch.stack([v[:, 0] * 360, v[:, 1] * 180, 1 + v[:, 2] * 80, 1 + v[:, 3] * 80, -90 + 180 * v[:, 4]], 1)[:, :dim]
    b = torch.cat([near, far])[torch.randperm(n, generator=g)]
    b[:, 0] %= 360
    b[:, 1] = b[:, 1].clamp(0.5, 179.5)
    b[:, 2:4] = b[:, 2:4].clamp(1, 170)
    if n > 600:
        b[5] = b[n - 3]            # identical boxes far apart (different tiles)
        b[300] = b[301]            # ... and next to each other
    if k > 3:
        gt[2] = gt[0]              # duplicated GT
        gt[3] = torch.tensor([181.0, 1.0, 1.0, 1.0, 0.0][:dim])   # a tiny GT at the pole
        b[:, 1] = b[:, 1].clamp(60, 179.5)                      # ... that no box reaches: >= 59 deg away, and the
        b[:, 2:4] = b[:, 2:4].clamp(1, 75)                      # boxes' circumscribed circles have radii <= 53 deg
    return gt.cuda(), b.cuda(), torch.randint(0, 37, (k,), generator=g).cuda()


@pytest.mark.parametrize('k,n,dim', [(1, 1, 4), (3, 70, 4), (64, 1000, 4), (17, 4099, 5), (200, 333, 4), (33, 65, 5),
                                     (65, 2000, 4), (9, 256, 4), (8, 513, 4)])
@pytest.mark.parametrize('variant', ['standard', 'efficient'])
def test_fused_equals_matrix_route_bit_for_bit(A, k, n, dim, variant):
    import sph_retina_amd as S
    gt, boxes, labels = _scene(k, n, dim, 1000 * k + n)
    ign = (torch.rand(n, generator=torch.Generator().manual_seed(n)) < 0.1).cuda()
    for ignore in (None, ign, torch.ones_like(ign)):
        for kw in CFGS:
            ov, res, ex = _matrix_route(A, S, gt, boxes, labels, variant, ignore, **kw)
            res2, ex2 = A.fused_assign(gt, boxes, labels, variant, ignore_mask=ignore, return_extras=True, **kw)
            _same(res, ex, res2, ex2)
            res3, ov3, ex3 = A.fused_assign(gt, boxes, labels, variant, ignore_mask=ignore, return_overlaps=True,
                                            return_extras=True, **kw)
            _same(res, ex, res3, ex3, ov, ov3)
    if k > 3 and n > 600:   # the scene holds what it claims
        ov, res, ex = _matrix_route(A, S, gt, boxes, labels, variant, None, **CFGS[0])
        assert float(ov[3].max()) == 0.0 and int((res.gt_inds == 4).sum()) > n // 2   # zero-overlap GT takes every free box
        assert torch.equal(ov[0], ov[2])


@pytest.mark.parametrize('which', ['box', 'gt', 'both'])
def test_fused_equals_matrix_route_with_nan_boxes(A, which):
    """A NaN coordinate (a diverged regression head) makes that row / column of the overlaps NaN in the reference
    (torch.max treats NaN as the maximum: max_iou_assigner.py:171-175 then assigns through it); the fused reductions carry
    the NaN the same way the matrix route does — bit for bit, the NaN's payload included."""
    import sph_retina_amd as S
    gt, boxes, labels = _scene(12, 1500, 4, 77)
    if which in ('box', 'both'):
        boxes[40, 2] = float('nan')
        boxes[900, 0] = float('nan')
    if which in ('gt', 'both'):
        gt[5, 1] = float('nan')
    for kw in CFGS:
        ov, res, ex = _matrix_route(A, S, gt, boxes, labels, 'standard', None, **kw)
        res2, ex2 = A.fused_assign(gt, boxes, labels, 'standard', return_extras=True, **kw)
        _same(res, ex, res2, ex2)
    assert bool(torch.isnan(ov).any())


def test_matrix_route_with_nan_follows_torch_max(A):
    """The (k, n) epilogue on a matrix holding NaN columns, rows and single entries against torch.max on the same device
    tensor (what the reference computes, max_iou_assigner.py:171-175)."""
    g = torch.Generator().manual_seed(5)
    ov = torch.rand((70, 3000), generator=g)
    ov[:, 17] = float('nan')
    ov[2, 40] = float('nan')
    ov[45, 40] = float('nan')
    ov[33, :] = float('nan')
    ov = ov.cuda()
    res, ex = A.assign_wrt_overlaps(ov, torch.arange(70).cuda(), pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0, return_extras=True)
    mo, amo = ov.max(dim=0)
    gmo, gamo = ov.max(dim=1)
    assert torch.equal(torch.isnan(res.max_overlaps), torch.isnan(mo))
    assert torch.equal(torch.nan_to_num(res.max_overlaps, nan=-5.0), torch.nan_to_num(mo, nan=-5.0))
    assert torch.equal(ex['argmax_overlaps'], amo)
    assert torch.equal(torch.isnan(ex['gt_max_overlaps']), torch.isnan(gmo)) and torch.equal(ex['gt_argmax_overlaps'], gamo)


def test_fused_assign_equals_reference_fixture(A):
    """The reference's real MaxIoUAssigner.assign on spherical boxes (tests/golden/assign.npz, part b): same assignment
    except where an anchor's IoU sits within fp32 noise of a threshold or of a row maximum tie."""
    from conftest import load_golden
    g = load_golden('assign')
    gt, anchors, labels = cu(g['b_gt']), cu(g['b_anchors']), cu(g['b_labels'])
    from test_assign_golden import CFGS as REF_CFGS
    for ci, cfg in enumerate(REF_CFGS):
        res = A.fused_assign(gt, anchors, labels, 'standard', **cfg)
        want = g[f'b_c{ci}_plain_gt_inds'].astype(np.int64)
        got = res.gt_inds.cpu().numpy()
        mo = g['b_plain_max_overlaps']
        lo, hi = (cfg['neg_iou_thr'] if isinstance(cfg['neg_iou_thr'], tuple) else (0.0, cfg['neg_iou_thr']))
        edge = np.zeros_like(mo, dtype=bool)
        for thr in (lo, hi, cfg['pos_iou_thr']):
            edge |= np.abs(mo - thr) < 2e-4
        bad = (got != want) & ~edge
        assert bad.sum() == 0, (ci, np.nonzero(bad)[0][:10], got[bad][:10], want[bad][:10])
        assert np.abs(res.max_overlaps.cpu().numpy() - mo).max() < 2e-3
        np.testing.assert_array_equal(res.labels.cpu().numpy()[~edge], g[f'b_c{ci}_plain_labels'][~edge])


@pytest.mark.parametrize('hw,count', [((512, 1024), 98208), ((1024, 2048), 392832)])
def test_config4_fused_equals_matrix_route(A, oracle, hw, count):
    import sys, os
    import sph_retina_amd as S
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from bench_configs import retina_anchors
    anchors = retina_anchors(*hw)
    g = torch.Generator().manual_seed(0)
    u = torch.rand((64, 4), generator=g)
    gt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1).cuda()
    labels = torch.randint(0, 37, (64,), generator=g).cuda()
    kw = dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0)
    for variant in ('standard', 'efficient'):
        ov, res, ex = _matrix_route(A, S, gt, anchors, labels, variant, None, **kw)
        res2, ex2 = A.fused_assign(gt, anchors, labels, variant, return_extras=True, **kw)
        _same(res, ex, res2, ex2)
        res3, ov3, ex3 = A.fused_assign(gt, anchors, labels, variant, return_overlaps=True, return_extras=True, **kw)
        _same(res, ex, res3, ex3, ov, ov3)
    # the registry-built assigner takes the fused route by itself and the matrix route on request
    fused = A.SphMaxIoUAssigner(ignore_iof_thr=-1, **kw).assign(anchors, gt, gt_labels=labels)
    plain = A.SphMaxIoUAssigner(ignore_iof_thr=-1, fused=False, **kw).assign(anchors, gt, gt_labels=labels)
    assert torch.equal(fused.gt_inds, plain.gt_inds) and torch.equal(fused.labels, plain.labels)
    assert torch.equal(fused.max_overlaps, plain.max_overlaps)
    gi, mo, *_r, lab = oracle.assign_wrt_overlaps(ov.cpu().numpy(), labels.cpu().numpy(), **kw)   # ov: efficient, last loop
    res_e = A.fused_assign(gt, anchors, labels, 'efficient', **kw)
    np.testing.assert_array_equal(res_e.gt_inds.cpu().numpy(), gi)


def test_fused_two_shards_with_key_exchange_equal_one_device(A):
    """shard the box axis, reduce per shard, MAX the k keys (the all-reduce), finalize per shard — on one GPU."""
    import ctypes
    from sph_retina_amd import _lib, _torch_glue as G
    lib = _lib.lib()
    gt, boxes, labels = _scene(40, 3000, 4, 77)
    boxes[2999] = boxes[10]   # a cross-shard tie on a row maximum
    kw = dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0)
    whole, exw = A.fused_assign(gt, boxes, labels, 'standard', return_extras=True, **kw)
    st = G.raw_stream_of(gt.device)
    cuts = [(0, 1100), (1100, 3000)]
    keys, wss, shards = [], [], []
    for lo, hi in cuts:
        sh = boxes[lo:hi].contiguous()
        n = hi - lo
        ws = torch.empty((lib.sph2pob_iou_assign_workspace_bytes(40, n) // 8,), dtype=torch.int64, device='cuda')
        state = torch.zeros((lib.sph2pob_iou_assign_state_bytes(40, n) // 8,), dtype=torch.int64, device='cuda')
        key = torch.empty((40,), dtype=torch.int64, device='cuda')
        rc = lib.sph2pob_iou_assign_reduce_f32(G.ptr(gt), 40, G.ptr(sh), n, 4, 0, 0, None, lo, None, G.ptr(key), G.ptr(ws), G.ptr(state), st)
        assert int(state.abs().sum()) == 0      # left clean
        assert rc == 0
        keys.append(key); wss.append(ws); shards.append(sh)
    allk = torch.maximum(keys[0], keys[1])    # all_reduce(MAX) on int64
    gi, mo = [], []
    for (lo, hi), sh, ws in zip(cuts, shards, wss):
        n = hi - lo
        a = torch.empty((n,), dtype=torch.int64, device='cuda')
        m = torch.empty((n,), dtype=torch.float32, device='cuda')
        gm = torch.empty((40,), dtype=torch.float32, device='cuda')
        gam = torch.empty((40,), dtype=torch.int64, device='cuda')
        rc = lib.sph2pob_iou_assign_finalize_f32(G.ptr(gt), 40, G.ptr(sh), n, 4, 0, 0, lo, G.ptr(allk), 0.5, 0.0, 0.4, 0.0, 1, 1,
                                                 None, G.ptr(m), None, G.ptr(gm), G.ptr(gam), G.ptr(a), None, G.ptr(ws), st)
        assert rc == 0
        gi.append(a); mo.append(m)
        assert torch.equal(gm, exw['gt_max_overlaps']) and torch.equal(gam, exw['gt_argmax_overlaps'])
    assert torch.equal(torch.cat(gi), whole.gt_inds) and torch.equal(torch.cat(mo), whole.max_overlaps)


def test_sharded_assign_hip_operator_and_key_format(A):
    """parallel.sharded_assign with the HIP operator (one rank: no collective) equals the fused call, and the keys the reduce
    half hands to the all-reduce are exactly parallel.pack_assign_keys(gt_max, gt_argmax) — the format the gloo tests exercise."""
    from sph_retina_amd import parallel as P
    gt, boxes, labels = _scene(40, 3000, 5, 78)
    kw = dict(pos_iou_thr=0.5, neg_iou_thr=(0.05, 0.4), min_pos_iou=0.1)
    whole, ex = A.fused_assign(gt, boxes, labels, 'efficient', return_extras=True, **kw)
    gi, mo, lab = P.sharded_assign(gt, boxes, 0, labels, op='efficient', **kw)
    assert torch.equal(gi, whole.gt_inds) and torch.equal(mo, whole.max_overlaps) and torch.equal(lab, whole.labels)
    op = P._HipAssignOp('efficient')
    keys, _ctx = op.reduce(gt, boxes[1000:].contiguous(), 1000)
    ov = A.fused_assign(gt, boxes[1000:].contiguous(), None, 'efficient', return_overlaps=True, **kw)[1]
    mx, am = ov.max(dim=1)
    assert torch.equal(keys, P.pack_assign_keys(mx, am + 1000))
    v, i = P.unpack_assign_keys(keys)
    assert torch.equal(v, mx) and torch.equal(i, am + 1000)
