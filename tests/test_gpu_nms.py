"""GPU: SphNMS (suppression bit-matrix + single-wave sweep) vs the reference keep lists (fixtures) and the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def N():
    import sph_retina_amd.bbox.nms as nms
    assert torch.cuda.is_available()
    return nms


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_reference_scenario_and_random_scenes(N):
    g = load_golden('nms')
    nms = N.SphNMS(iou_calculator='sph2pob_efficient')
    dets, keep = nms(cu(g['boxes']), cu(g['scores']), cu(g['idxs']), dict(type='nms', iou_threshold=0.5))
    assert keep.tolist() == [0, 5, 7, 3, 8, 9]  # reference tests/test_nms.py scenario, SURVEY App. C.4
    np.testing.assert_allclose(dets.cpu().numpy(), g['dets'], atol=1e-6)
    assert keep.dtype == torch.int64 and dets.shape == (6, 5)
    dets, keep = nms(cu(g['rboxes']), cu(g['rscores']), cu(g['ridxs']), dict(type='nms', iou_threshold=0.5, max_num=100))
    assert keep.tolist() == g['rkeep'].tolist()
    np.testing.assert_allclose(dets.cpu().numpy(), g['rdets'], atol=1e-6)
    dets, keep = nms(cu(g['r5boxes']), cu(g['r5scores']), cu(g['r5idxs']), dict(type='nms', iou_threshold=0.4))
    assert keep.tolist() == g['r5keep'].tolist()
    assert dets.shape[1] == 6


def test_config4_size_5000_boxes_37_classes_vs_oracle(N, oracle):
    """BASELINE config 4 NMS half: K = 5 x nms_pre(1000) boxes, 37 classes, thr 0.5, max_per_img 100."""
    rng = np.random.default_rng(4)
    k = 5000
    centres = oracle.generate_boxes(300, 8, alpha=(5, 60), beta=(5, 60))
    boxes = centres[rng.integers(0, 300, k)] + rng.standard_normal((k, 4)).astype(np.float32) * 2.0
    boxes[:, 0] %= 360
    boxes[:, 1] = boxes[:, 1].clip(1, 179)
    boxes[:, 2:] = boxes[:, 2:].clip(2, 120)
    scores = rng.random(k).astype(np.float32)
    idxs = rng.integers(0, 37, k)
    dets, keep = N.SphNMS()(cu(boxes), cu(scores), cu(idxs), dict(type='nms', iou_threshold=0.5, max_num=100))
    odets, okeep = oracle.batched_nms(boxes, scores, idxs, 0.5, max_num=100)
    assert keep.tolist() == okeep.tolist()
    np.testing.assert_allclose(dets.cpu().numpy(), odets, atol=1e-6)
    # full keep set (no max_num): allow the rare threshold flip (an IoU within fp32 noise of 0.5)
    _, keep_all = N.SphNMS()(cu(boxes), cu(scores), cu(idxs), dict(type='nms', iou_threshold=0.5))
    _, okeep_all = oracle.batched_nms(boxes, scores, idxs, 0.5)
    a, b = set(keep_all.tolist()), set(okeep_all.tolist())
    assert len(a ^ b) <= 2, (len(a), len(b), len(a ^ b))
    s = scores[keep_all.cpu().numpy()]
    assert (np.diff(s) <= 0).all()  # descending score order


def test_single_class_op_and_edge_cases(N, oracle):
    rng = np.random.default_rng(1)
    b = oracle.generate_boxes(700, 3, alpha=(10, 50), beta=(10, 50))
    b[:, :2] = np.array([100, 90], np.float32) + rng.standard_normal((700, 2)).astype(np.float32) * 15
    s = rng.random(700).astype(np.float32)
    keep = N.sph_nms_op(cu(b), cu(s), 0.3)
    okeep = oracle.nms_op(b, s, 0.3)
    assert len(set(keep.tolist()) ^ set(okeep.tolist())) <= 1
    # the reference hands the IoU FUNCTION to sph_nms_op (sph_nms.py:19, :62): accepted, mapped by name
    import sph_retina_amd as S
    assert torch.equal(N.sph_nms_op(cu(b), cu(s), 0.3, S.sph2pob_efficient_iou), keep)
    assert torch.equal(N.sph_nms_op(cu(b), cu(s), 0.3, 'sph2pob_efficient_iou'), keep)
    with pytest.raises(TypeError):
        N.sph_nms_op(cu(b), cu(s), 0.3, len)
    with pytest.raises(TypeError):
        N.sph_nms_op(cu(b), cu(s), 0.3, 42)
    # idempotence: NMS of the survivors keeps all of them
    kb, ks = cu(b)[keep], cu(s)[keep]
    again = N.sph_nms_op(kb, ks, 0.3)
    assert again.numel() == keep.numel()
    # empty, single box, all identical boxes, RBFoV
    dets, keep = N.SphNMS()(cu(b[:0]), cu(s[:0]), cu(np.zeros(0, np.int64)), dict(iou_threshold=0.5))
    assert dets.shape == (0, 5) and keep.numel() == 0
    dets, keep = N.SphNMS()(cu(b[:1]), cu(s[:1]), cu(np.zeros(1, np.int64)), dict(iou_threshold=0.5))
    assert keep.tolist() == [0]
    same = np.repeat(b[:1], 130, axis=0)
    dets, keep = N.SphNMS()(cu(same), cu(s[:130]), cu(np.zeros(130, np.int64)), dict(iou_threshold=0.5))
    assert keep.tolist() == [int(np.argmax(s[:130]))]
    with pytest.raises(ValueError):
        N.SphNMS()(cu(b), cu(s), cu(np.zeros(700, np.int64)), None)
    with pytest.raises(TypeError):
        N.SphNMS('planar_iou')


def test_multiclass_nms_wrapper(N, oracle):
    rng = np.random.default_rng(2)
    n, c = 300, 4
    boxes = oracle.generate_boxes(n, 5, alpha=(10, 50), beta=(10, 50))
    boxes[:, :2] = np.array([200, 80], np.float32) + rng.standard_normal((n, 2)).astype(np.float32) * 20
    ms = rng.random((n, c + 1)).astype(np.float32)
    dets, labels, inds = N.multiclass_nms(cu(boxes), cu(ms), 0.3, dict(type='nms', iou_threshold=0.5), max_num=50,
                                          return_inds=True, nms_op=N.SphNMS(), box_version=4)
    assert dets.shape[1] == 5 and dets.shape[0] <= 50 and labels.shape[0] == dets.shape[0]
    flat_scores = ms[:, :-1].reshape(-1)
    np.testing.assert_allclose(dets[:, 4].cpu().numpy(), flat_scores[inds.cpu().numpy()], atol=0)
    assert (np.diff(dets[:, 4].cpu().numpy()) <= 0).all() and float(dets[:, 4].min()) > 0.3


def test_many_candidates_beyond_one_matrix(N, oracle):
    """multiclass_nms hands every (box, class) candidate above score_thr to the NMS (sphdet/bbox/nms/utils.py:6-15):
    far more than 32 k rows in total, a few thousand per class — the segment-relative suppression matrix handles it."""
    rng = np.random.default_rng(11)
    k, ncls = 60000, 37
    centres = oracle.generate_boxes(400, 3, alpha=(5, 40), beta=(5, 40))
    b = centres[rng.integers(0, 400, k)] + rng.normal(0, 1.5, (k, 4)).astype(np.float32)
    b[:, 0] %= 360
    b[:, 1] = np.clip(b[:, 1], 1, 179)
    b[:, 2:] = np.clip(b[:, 2:], 2, 120)
    scores = rng.random(k).astype(np.float32)
    idxs = rng.integers(0, ncls, k)
    dets, keep = N.SphNMS()(cu(b), cu(scores), cu(idxs), dict(type='nms', iou_threshold=0.5, max_num=300))
    keep = keep.cpu().numpy()
    assert len(keep) == 300 and len(set(keep.tolist())) == 300
    # check three classes completely against the restatement
    for c in (0, 17, 36):
        m = np.nonzero(idxs == c)[0]
        ref = set(m[oracle.nms_op(b[m], scores[m], 0.5, variant='efficient')].tolist())
        d_all, keep_all = N.SphNMS()(cu(b[m]), cu(scores[m]), cu(np.zeros(len(m), np.int64)), dict(iou_threshold=0.5))
        got = set(m[keep_all.cpu().numpy()].tolist())
        assert len(got ^ ref) <= 2, (c, len(got), len(ref))          # a borderline IoU may flip one decision
    # and the joint call equals the per-class calls
    full_d, full_k = N.SphNMS()(cu(b), cu(scores), cu(idxs), dict(type='nms', iou_threshold=0.5))
    per = []
    for c in range(ncls):
        m = np.nonzero(idxs == c)[0]
        _, kc = N.SphNMS()(cu(b[m]), cu(scores[m]), cu(np.zeros(len(m), np.int64)), dict(iou_threshold=0.5))
        per.append(m[kc.cpu().numpy()])
    assert set(np.concatenate(per).tolist()) == set(full_k.cpu().numpy().tolist())


def test_planar_nms_is_naive_iou_nms_class_agnostic_by_default(N, oracle):
    rng = np.random.default_rng(5)
    centres = oracle.generate_boxes(30, 2, alpha=(8, 50), beta=(8, 50))
    k = 500
    b = centres[rng.integers(0, 30, k)] + rng.normal(0, 2.0, (k, 4)).astype(np.float32)
    b[:, 0] %= 360
    b[:, 1] = np.clip(b[:, 1], 1, 179)
    b[:, 2:] = np.clip(b[:, 2:], 2, 120)
    scores, idxs = rng.random(k).astype(np.float32), rng.integers(0, 5, k)
    cfg = dict(type='nms', iou_threshold=0.5)
    d_ag, k_ag = N.PlanarNMS()(cu(b), cu(scores), cu(idxs), cfg)                       # across classes
    rd, rk = oracle.batched_nms(b, scores, np.zeros(k, np.int64), 0.5, variant='naive')
    assert np.array_equal(k_ag.cpu().numpy(), rk)
    np.testing.assert_allclose(d_ag.cpu().numpy(), rd, atol=0)
    d_pc, k_pc = N.PlanarNMS()(cu(b), cu(scores), cu(idxs), dict(cfg, class_agnostic=False))
    rd, rk = oracle.batched_nms(b, scores, idxs, 0.5, variant='naive')
    assert np.array_equal(k_pc.cpu().numpy(), rk) and len(rk) >= len(k_ag)
    assert N.PlanarNMS('sph2tan').box_formator == 'sph2tan'      # served since round 3 (tests/test_unbiased_naive.py)
    with pytest.raises(AssertionError):
        N.PlanarNMS('sph2kent')


def test_long_single_class_segment_through_the_pipelined_sweep(N, oracle):
    """one class of 9 000 boxes = 141 blocks of 64 rows: the sweep's OR stage then has more (word, 16-row) tasks than
    worker threads for the early blocks, and every block's removed word arrives half through the LDS bit-vector (blocks up
    to b - 2) and half through wave 0's register (block b - 1); keep list against the CPU restatement, and idempotence"""
    rng = np.random.default_rng(5)
    k = 9000
    centres = oracle.generate_boxes(500, 9, alpha=(5, 50), beta=(5, 50))
    b = centres[rng.integers(0, 500, k)] + rng.normal(0, 2.0, (k, 4)).astype(np.float32)
    b[:, 0] %= 360
    b[:, 1] = np.clip(b[:, 1], 1, 179)
    b[:, 2:] = np.clip(b[:, 2:], 2, 120)
    s = rng.random(k).astype(np.float32)
    keep = N.sph_nms_op(cu(b), cu(s), 0.5)
    ref = oracle.nms_op(b, s, 0.5, variant='efficient')
    got, want = set(keep.tolist()), set(ref.tolist())
    assert len(got ^ want) <= 4, (len(got), len(want), len(got ^ want))      # a borderline IoU may flip a decision (and what it suppressed)
    assert torch.equal(s_sorted_desc(cu(s)[keep]), cu(s)[keep])
    again = N.sph_nms_op(cu(b)[keep], cu(s)[keep], 0.5)
    assert again.numel() == keep.numel()


def s_sorted_desc(t):
    return torch.sort(t, descending=True, stable=True)[0]


# ---- round 3: the host-free route (sph2pob_batched_nms_f32) and the chunked sweep of over-long classes ----
def _general_route(N, boxes, scores, idxs, cfg, variant='efficient'):
    """sph_batched_nms with the host-free route switched off: torch sorts + the same two NMS kernels (round 2's path)."""
    from sph_retina_amd import _lib
    lib = _lib.lib()
    real = lib.sph2pob_batched_nms_max_boxes
    try:
        lib.sph2pob_batched_nms_max_boxes = lambda: 0
        return N.sph_batched_nms(boxes, scores, idxs, cfg, variant)
    finally:
        lib.sph2pob_batched_nms_max_boxes = real


@pytest.mark.parametrize('k,dim,ncls', [(1, 4, 1), (63, 4, 3), (2048, 4, 37), (2049, 5, 5), (5000, 4, 37), (8192, 4, 2), (16384, 4, 80)])
def test_host_free_route_equals_general_route(N, oracle, k, dim, ncls):
    rng = np.random.default_rng(k + dim)
    centres = oracle.generate_boxes(max(k // 16, 1), 8, box='bfov' if dim == 4 else 'rbfov', alpha=(5, 60), beta=(5, 60))
    b = centres[rng.integers(0, len(centres), k)] + rng.standard_normal((k, dim)).astype(np.float32) * 2.0
    b[:, 0] %= 360
    b[:, 1] = b[:, 1].clip(1, 179)
    b[:, 2:4] = b[:, 2:4].clip(2, 120)
    scores = rng.random(k).astype(np.float32)
    scores[rng.integers(0, k, k // 8)] = np.float32(0.5)        # many exact score ties: broken by the original index
    idxs = rng.integers(0, ncls, k)
    tb, ts, ti = cu(b), cu(scores), cu(idxs)
    for cfg in (dict(type='nms', iou_threshold=0.5, max_num=100), dict(type='nms', iou_threshold=0.3)):
        dets, keep = N.sph_batched_nms(tb, ts, ti, cfg, 'efficient')
        gdets, gkeep = _general_route(N, tb, ts, ti, cfg)
        assert torch.equal(keep, gkeep) and torch.equal(dets, gdets)
        assert keep.dtype == torch.int64 and dets.shape == (keep.numel(), dim + 1)
    if k == 5000:   # and against the CPU restatement of the reference's loop
        odets, okeep = oracle.batched_nms(b, scores, idxs, 0.5, max_num=100)
        dets, keep = N.sph_batched_nms(tb, ts, ti, dict(iou_threshold=0.5, max_num=100), 'efficient')
        sk = scores[keep.cpu().numpy()]
        assert (np.diff(sk) <= 0).all() and len(set(keep.tolist()) ^ set(okeep.tolist())) <= 2


def test_host_free_route_leaves_inputs_alone_and_handles_odd_class_ids(N):
    g = torch.Generator().manual_seed(3)
    k = 900
    b = torch.stack([torch.rand(k, generator=g) * 40 + 100, torch.rand(k, generator=g) * 30 + 70,
                     torch.rand(k, generator=g) * 30 + 5, torch.rand(k, generator=g) * 30 + 5], 1).cuda()
    s = torch.rand(k, generator=g).cuda()
    cfg = dict(iou_threshold=0.5)
    small = torch.randint(0, 5, (k,), generator=g).cuda()
    b0, s0 = b.clone(), s.clone()
    d1, k1 = N.sph_batched_nms(b, s, small, cfg)
    assert torch.equal(b, b0) and torch.equal(s, s0)
    # class ids beyond the composite key's 18 bits, and negative ones: the call silently takes the general route — same result
    for ids in (small * 1_000_003, small - 3, small.to(torch.int32)):
        d2, k2 = N.sph_batched_nms(b, s, ids, cfg)
        assert torch.equal(k1, k2) and torch.equal(d1, d2)
    # scores as torch's device sort (cub radix keys) orders them: +NaN first, -NaN last, -0 == +0 (ties keep the index order)
    odd = s.clone()
    odd[5], odd[77] = float('nan'), float('nan')
    odd[77] = torch.copysign(odd[77], torch.tensor(-1.0, device='cuda'))
    odd[[10, 12, 300]] = 0.0
    odd[[11, 301]] = -0.0
    d3, k3 = N.sph_batched_nms(b, odd, small, cfg)
    g3, gk3 = _general_route(N, b, odd, small, cfg)
    assert torch.equal(k3, gk3) and torch.equal(torch.nan_to_num(d3, nan=-7.0), torch.nan_to_num(g3, nan=-7.0))
    assert k3[0].item() == 5 and torch.isnan(d3[0, -1]) and k3[-1].item() == 77


def test_one_class_of_40000_boxes_is_swept_in_chunks(N, oracle):
    """A class beyond the sweep kernel's 32 704 boxes (round 2 raised; the reference has no limit, sph_nms.py:62-74):
    chunks of 16 384, each first tested against every box kept so far.  Keep list against the CPU restatement of the loop."""
    rng = np.random.default_rng(40)
    k = 40000
    centres = oracle.generate_boxes(4000, 9, alpha=(3, 25), beta=(3, 25))
    b = centres[rng.integers(0, 4000, k)] + rng.normal(0, 1.0, (k, 4)).astype(np.float32)
    b[:, 0] %= 360
    b[:, 1] = np.clip(b[:, 1], 1, 179)
    b[:, 2:] = np.clip(b[:, 2:], 1, 120)
    s = rng.random(k).astype(np.float32)
    keep = N.sph_nms_op(cu(b), cu(s), 0.5)
    ref = oracle.nms_op(b, s, 0.5, variant='efficient', nthreads=16)
    got, want = set(keep.tolist()), set(ref.tolist())
    assert len(want) > 3000 and len(got ^ want) <= 8, (len(got), len(want), len(got ^ want))
    sk = s[keep.cpu().numpy()]
    assert (np.diff(sk) <= 0).all()
    # the same class inside a multi-class call: the long class in chunks, the others through the kernels
    idxs = np.zeros(k + 600, np.int64)
    idxs[k:] = rng.integers(1, 4, 600)
    b2 = np.concatenate([b, b[:600] + np.float32(0.3)])
    s2 = np.concatenate([s, rng.random(600).astype(np.float32)])
    dets, keep2 = N.SphNMS()(cu(b2), cu(s2), cu(idxs), dict(iou_threshold=0.5))
    k2 = keep2.cpu().numpy()
    assert set(k2[k2 < k].tolist()) == got
    rest = np.nonzero(idxs > 0)[0]
    _, kr = N.SphNMS()(cu(b2[rest]), cu(s2[rest]), cu(idxs[rest]), dict(iou_threshold=0.5))
    assert set(rest[kr.cpu().numpy()].tolist()) == set(k2[k2 >= k].tolist())
    assert (np.diff(s2[k2]) <= 0).all()


def _nan_scene():
    rng = np.random.default_rng(12)
    k = 240
    b = np.stack([100 + rng.random(k) * 40, 70 + rng.random(k) * 30, 5 + rng.random(k) * 30, 5 + rng.random(k) * 30], 1).astype(np.float32)
    s = rng.random(k).astype(np.float32)
    idxs = rng.integers(0, 3, k)
    top0 = np.flatnonzero(idxs == 0)[np.argmax(s[idxs == 0])]        # the best box of class 0 is NaN: it is kept and, its IoU
    b[top0, 2] = np.nan                                              # with everything being NaN, nothing else of the class is
    mid1 = np.flatnonzero(idxs == 1)[np.argsort(-s[idxs == 1])[7]]   # a NaN box further down class 1: removed by the first kept box
    b[mid1, 0] = np.nan
    return b, s, idxs, top0, mid1


def _loop_with(iou_fn, boxes, scores, idxs, thr):
    """The reference's loops (sph_nms.py:39-52 per class, :62-74 greedy) on the IoUs `iou_fn` gives: whether a NaN box yields a
    NaN IoU is the IoU operator's business (mmcv's planar kernel is absent: unpinned); what NMS does with one is the
    loop's — `iou <= thr` keeps."""
    keep_all = []
    for c in torch.unique(idxs).tolist():
        ids = torch.nonzero(idxs == c).flatten()
        order = ids[torch.argsort(scores[ids], descending=True, stable=True)]
        while order.numel() > 0:
            keep_all.append(int(order[0]))
            if order.numel() == 1:
                break
            iou = iou_fn(boxes[order[:1]], boxes[order[1:]]).reshape(-1)
            order = order[1:][iou <= thr]
    keep = torch.tensor(sorted(keep_all), device=scores.device)
    return keep[torch.argsort(scores[keep], descending=True, stable=True)].tolist()


def test_nan_boxes_suppress_like_the_reference_loop(N):
    """sph_nms.py:69-73 keeps `iou <= thr`: a NaN IoU fails the test, so a NaN box suppresses (when kept) or is suppressed
    (otherwise).  Fused route, torch-sorted route and the single-class operator against the loop run on this package's IoUs."""
    import sph_retina_amd as S
    b, s, idxs, top0, mid1 = _nan_scene()
    tb, ts, ti = cu(b), cu(s), cu(idxs)
    want = _loop_with(S.sph2pob_efficient_iou, tb, ts, ti, 0.5)
    assert top0 in want and mid1 not in want and int((idxs[want] == 0).sum()) == 1
    dets, keep = N.sph_batched_nms(tb, ts, ti, dict(iou_threshold=0.5), 'efficient')
    gdets, gkeep = _general_route(N, tb, ts, ti, dict(iou_threshold=0.5))
    assert keep.tolist() == want and gkeep.tolist() == want
    one = torch.from_numpy(idxs == 1)
    w1 = _loop_with(S.sph2pob_efficient_iou, tb[one], ts[one], torch.zeros(int(one.sum()), dtype=torch.long, device=tb.device), 0.5)
    assert N.sph_nms_op(tb[one], ts[one], 0.5).tolist() == w1
