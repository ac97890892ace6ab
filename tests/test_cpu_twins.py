"""CPU: the product's host twins (libsph2pob_host.so, `*_cpu`) behind the reference's operator surface on CPU tensors.

BASELINE configs[0] is the reference's own CPU-runnable case (tests/test_all_ious.py:88-104 runs every IoU with device='cpu';
sphdet/iou/sph_iou_calculator.py:107-108 forces unbiased_iou to the CPU): 10 000 BFoV pairs through `sph_retina_amd` on CPU
tensors, against the fixtures generated from the unmodified reference and against the oracle.  The host twins are the kernels'
own __host__ __device__ arithmetic compiled for the host (they share no code with the oracle, which checks them here)."""
import numpy as np
import pytest
import torch

from conftest import err_stats, load_golden


@pytest.fixture(scope='module')
def S():
    import sph_retina_amd as S
    S.set_arithmetic('fast')
    return S


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_configs0_10k_bfov_pairs_on_cpu_and_inputs_not_mutated(S, oracle):
    """configs[0] + the reference's immutability check (tests/test_all_ious.py:322-332)."""
    n = 10_000
    b1, b2 = oracle.generate_boxes(n, 0), oracle.generate_boxes(n, 1)
    t1, t2 = t(b1), t(b2)
    k1, k2 = t1.clone(), t2.clone()
    for name, fn in (('standard', S.sph2pob_standard_iou), ('efficient', S.sph2pob_efficient_iou), ('legacy', S.sph2pob_legacy_iou)):
        got = fn(t1, t2, is_aligned=True)
        assert got.device.type == 'cpu' and got.dtype == torch.float32 and got.shape == (n,)
        assert torch.equal(t1, k1) and torch.equal(t2, k2)
        ref = oracle.iou_aligned(b1, b2, variant=name, planar='mmcv')
        tru = oracle.iou_aligned(b1, b2, variant=name, planar='exact', dtype=np.float64)
        r, tr = err_stats(got.numpy(), ref), err_stats(got.numpy(), tru)
        assert r['mean'] < 1e-7 and r['n5'] <= 2 and r['max'] < 1e-4, (name, r)
        assert tr['mean'] < 1e-7 and tr['n5'] <= 2, (name, tr)
        assert float(got.min()) >= 0 and float(got.max()) <= 1
    pw = S.sph2pob_efficient_iou(t1[:9], t2[:33])
    assert pw.shape == (9, 33) and torch.equal(pw.diagonal(), S.sph2pob_efficient_iou(t1[:9], t2[:9], is_aligned=True))


@pytest.mark.parametrize('fixture,variants', [('uniform_bfov', ('standard', 'efficient', 'legacy')), ('nearby_bfov', ('standard', 'efficient')),
                                              ('uniform_rbfov', ('standard', 'efficient')), ('samples7', ('standard', 'efficient', 'legacy'))])
def test_iou_fixtures_from_the_reference(S, fixture, variants):
    g = load_golden(fixture)
    fns = {'standard': S.sph2pob_standard_iou, 'efficient': S.sph2pob_efficient_iou, 'legacy': S.sph2pob_legacy_iou}
    for v in variants:
        got = fns[v](t(g['b1']), t(g['b2']), is_aligned=True).numpy()
        r, tr = err_stats(got, g['iou_' + v]), err_stats(got, g['iou64_' + v])
        if fixture == 'samples7':
            assert r['max'] < 1e-5 and tr['max'] < 1e-5, (v, r, tr)   # the reference's hard-coded sample pairs (tests/test_all_ious.py:244-261)
        elif fixture.startswith('uniform'):
            assert r['mean'] < 1e-7 and r['n5'] <= 2, (v, r)
        else:   # nearby pairs: the reference's own fp32 noise — held to f64 tightly, to fp32 in the mean
            assert r['mean'] < 3e-6 and tr['mean'] < 1e-6 and tr['n4'] <= 2, (v, r, tr)


def test_every_backend_of_the_calculator_on_cpu(S, oracle):
    g = load_golden('samples7_backends')
    b1, b2 = t(g['b1']), t(g['b2'])
    calc = S.SphOverlaps2D(backend='unbiased_iou', box_version=4)    # the reference forces this one to the CPU
    np.testing.assert_allclose(calc(b1, b2, is_aligned=True).numpy(), g['unbiased64'], atol=2e-6)
    np.testing.assert_allclose(calc(b1, b2).numpy(), g['unbiased_pw'], atol=2e-5)
    np.testing.assert_allclose(S.SphOverlaps2D(backend='sph_iou')(b1, b2, is_aligned=True).numpy(), g['sph'], atol=2e-6)
    np.testing.assert_allclose(S.SphOverlaps2D(backend='fov_iou')(b1, b2, is_aligned=True).numpy(), g['fov'], atol=2e-6)
    nv = S.SphOverlaps2D(backend='naive_iou')(b1, b2, is_aligned=True)
    assert nv.shape == (7,) and bool(((nv >= 0) & (nv <= 1)).all())
    with pytest.raises(NotImplementedError):
        S.SphOverlaps2D(backend='kent_iou')(b1, b2)
    pg = load_golden('pairwise')
    for fn, a, b, key, v in ((S.sph2pob_standard_iou, 'b1', 'b2', 'iou_standard', 'standard'), (S.sph2pob_efficient_iou, 'r1', 'r2', 'riou_efficient', 'efficient')):
        got = fn(t(pg[a]), t(pg[b])).numpy()
        truth = oracle.iou_pairwise(pg[a], pg[b], variant=v, planar='exact', dtype=np.float64)
        tol = 5e-5 + 1.5 * np.abs(pg[key] - truth)   # the reference's own distance to the exact value (the GPU test's criterion)
        assert got.shape == pg[key].shape and (np.abs(got - pg[key]) <= tol).all(), (key, np.abs(got - pg[key]).max())
    # options and both arithmetics
    og = load_golden('options')
    o1, o2 = t(og['b1']), t(og['b2'])
    try:
        for arith in ('fast', 'reference'):
            S.set_arithmetic(arith)
            for key in [k for k in og if k.startswith('standard_')][:6]:
                _, edge, angle, mode = key.split('_')
                got = S.sph2pob_standard_iou(o1, o2, mode=mode, is_aligned=True, rbb_edge=edge, rbb_angle=angle).numpy()
                assert np.abs(got - og[key]).mean() < 2e-6, (arith, key)
    finally:
        S.set_arithmetic('fast')


@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
def test_loss_forward_and_backward_on_cpu(S, box):
    """Sph2PobIoULoss on CPU tensors against the reference's autograd fixtures (same criteria as tests/test_gpu_loss.py)."""
    from sph_retina_amd.losses import Sph2PobIoULoss
    g = load_golden('loss_' + box)
    for mode in ('iou', 'giou', 'diou', 'ciou'):
        pred, target = t(g['pred']).requires_grad_(True), t(g['target']).requires_grad_(True)
        loss = Sph2PobIoULoss(mode=mode, reduction='none')(pred, target)
        assert loss.device.type == 'cpu'
        d = np.abs(loss.detach().numpy() - g['loss64_' + mode])
        assert d.mean() < 2e-6 and np.quantile(d, 0.99) < 5e-5, (mode, d.mean(), d.max())
        loss.sum().backward()
        for got, key in ((pred.grad, 'gpred_'), (target.grad, 'gtarget_')):
            want = g[key + mode]
            scale = np.abs(want).max()
            e = np.abs(got.numpy() - want) / scale
            assert np.median(e) < 1e-6 and np.quantile(e, 0.99) < 2e-3, (mode, key, np.median(e), e.max())
    # reductions, weights, avg_factor (mmdet/models/losses/utils.py:47-58)
    pred, target, w1, w2 = t(g['pred']), t(g['target']), t(g['w1']), t(g['w2'])
    L = Sph2PobIoULoss(mode='ciou', reduction='mean', loss_weight=2.0)      # as the fixtures were made (oracle/gen_goldens.py)
    rt = dict(rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(L(pred, target).item(), g['mean_ciou'], **rt)
    np.testing.assert_allclose(L(pred, target, w1).item(), g['mean_ciou_w1'], **rt)
    np.testing.assert_allclose(L(pred, target, w2).item(), g['mean_ciou_w2'], **rt)
    np.testing.assert_allclose(L(pred, target, w2, avg_factor=123.0).item(), g['mean_ciou_w2_avg'], **rt)
    np.testing.assert_allclose(L(pred, target, w1, reduction_override='sum').item(), g['sum_ciou_w1'], **rt)
    p = pred.clone().requires_grad_(True)
    L(p, target, w2, avg_factor=123.0).backward()
    want = g['gpred_mean_ciou_w2_avg']
    e = np.abs(p.grad.numpy() - want) / np.abs(want).max()
    assert np.median(e) < 1e-6 and np.quantile(e, 0.99) < 2e-3
    # a second backward through the same node and a non-unit upstream gradient (the two-pass backward twin)
    q = pred.clone().requires_grad_(True)
    out = L(q, target, reduction_override='none')
    out.sum().backward(retain_graph=True)
    g1 = q.grad.clone()
    q.grad = None
    (3.0 * out).sum().backward()
    assert torch.allclose(q.grad, 3.0 * g1, rtol=1e-5, atol=1e-9)


def test_nms_and_assigner_on_cpu(S, oracle):
    from sph_retina_amd.bbox.nms import SphNMS, sph_nms_op
    from sph_retina_amd.bbox.assigners import SphMaxIoUAssigner, assign_wrt_overlaps
    g = load_golden('nms')
    dets, keep = SphNMS('sph2pob_efficient')(t(g['boxes']), t(g['scores']), t(g['idxs']), dict(type='nms', iou_threshold=0.5))
    assert keep.tolist() == [0, 5, 7, 3, 8, 9] and keep.device.type == 'cpu'      # reference tests/test_nms.py scenario
    np.testing.assert_allclose(dets.numpy(), g['dets'], atol=1e-6)
    dets, keep = SphNMS()(t(g['rboxes']), t(g['rscores']), t(g['ridxs']), dict(type='nms', iou_threshold=0.5, max_num=100))
    assert keep.tolist() == g['rkeep'].tolist()
    dets, keep = SphNMS()(t(g['r5boxes']), t(g['r5scores']), t(g['r5idxs']), dict(type='nms', iou_threshold=0.4))
    assert keep.tolist() == g['r5keep'].tolist()
    one = sph_nms_op(t(g['rboxes']), t(g['rscores']), 0.5)
    assert one.tolist() == oracle.nms_op(g['rboxes'], g['rscores'], 0.5).tolist()
    # assign_wrt_overlaps on a CPU matrix: bit-equal to the reference's real class (tests/golden/assign.npz)
    from test_assign_golden import CFGS
    a = load_golden('assign')
    for si in range(7):
        for ci, cfg in enumerate(CFGS):
            res = assign_wrt_overlaps(t(a[f's{si}_ov']), t(a[f's{si}_labels']), **cfg)
            np.testing.assert_array_equal(res.gt_inds.numpy(), a[f's{si}_c{ci}_gt_inds'])
            np.testing.assert_array_equal(res.labels.numpy(), a[f's{si}_c{ci}_labels'])
            np.testing.assert_array_equal(res.max_overlaps.numpy(), a[f's{si}_max_overlaps'])
    # the registry-built assigner end to end on CPU boxes (what gpu_assign_thr moves to the CPU)
    res = SphMaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0).assign(t(a['b_anchors']), t(a['b_gt']), gt_labels=t(a['b_labels']))
    want = a['b_c4_plain_gt_inds'] if False else a['b_c0_plain_gt_inds']   # CFGS[0] differs from these thresholds only in pos/neg
    assert res.gt_inds.shape == (3000,) and res.num_gts == 12 and int((res.gt_inds > 0).sum()) >= 12


def test_assign_on_a_matrix_with_nan_follows_torch_max(S):
    """torch.max treats NaN as the maximum (first NaN wins): the assignment through a NaN column / row
    (max_iou_assigner.py:171-175 on a diverged head's boxes) must come out as the reference's torch ops give it."""
    from sph_retina_amd.bbox.assigners import assign_wrt_overlaps
    g = torch.Generator().manual_seed(5)
    ov = torch.rand((7, 300), generator=g)
    ov[:, 17] = float('nan')            # a NaN box
    ov[2, 40] = float('nan')            # single NaNs
    ov[5, 40] = float('nan')
    ov[4, :] = float('nan')             # a NaN GT
    labels = torch.arange(7)
    res, ex = assign_wrt_overlaps(ov, labels, pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0, return_extras=True)
    mo, amo = ov.max(dim=0)
    gmo, gamo = ov.max(dim=1)
    assert torch.equal(torch.isnan(res.max_overlaps), torch.isnan(mo))
    assert torch.equal(torch.nan_to_num(res.max_overlaps, nan=-5.0), torch.nan_to_num(mo, nan=-5.0))
    assert torch.equal(ex['argmax_overlaps'], amo)
    assert torch.equal(torch.isnan(ex['gt_max_overlaps']), torch.isnan(gmo)) and torch.equal(ex['gt_argmax_overlaps'], gamo)
    # the reference's steps on those values
    want = torch.full((300,), -1, dtype=torch.int64)
    want[(mo >= 0) & (mo < 0.4)] = 0
    pos = mo >= 0.5
    want[pos] = amo[pos] + 1
    for i in range(7):
        if gmo[i] >= 0.0:
            want[ov[i] == gmo[i]] = i + 1
    assert torch.equal(res.gt_inds, want)


def test_nms_with_nan_boxes_on_cpu_follows_the_reference_loop(S):
    """`iou <= thr` keeps (sph_nms.py:72): a NaN IoU suppresses.  The host twin against the reference's loop run on this
    package's own CPU IoUs."""
    from sph_retina_amd.bbox.nms import sph_batched_nms
    from test_gpu_nms import _loop_with, _nan_scene
    b, s, idxs, top0, mid1 = _nan_scene()
    want = _loop_with(S.sph2pob_efficient_iou, t(b), t(s), t(idxs), 0.5)
    dets, keep = sph_batched_nms(t(b), t(s), t(idxs), dict(iou_threshold=0.5), 'efficient')
    assert keep.tolist() == want and top0 in want and mid1 not in want
