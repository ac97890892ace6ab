"""CPU: pin the oracle (oracle/sph2pob_oracle.c) against fixtures produced by the unmodified reference Python.

Tolerances: the reference's fp32 transform computes centre distance / edge angles as acos(clamp(dot)) which
amplifies 1-ulp libm differences (torch/Sleef vs glibc) at small arcs; SURVEY App. C.2 measured the reference's
own fp32-vs-fp64 spread at >1e-5 on ~0.5 % of detector-like nearby pairs.  So: exact agreement is asserted on
the mean (the reference's own criterion, tests/test_sph_iou_loss.py:34: mean |d| < 1e-6) and on the bulk
(>= 99 % of pairs within 1e-5), plus tight bounds on well-conditioned sets (uniform / samples / edge cases).
"""
import numpy as np
import pytest

from conftest import err_stats, load_golden

VARIANTS3 = ['standard', 'efficient', 'legacy']


def test_samples7_known_answers(oracle):
    g = load_golden('samples7')
    # SURVEY App. C.3 / reference tests/test_all_ious.py:244-261
    expect = {'standard': [0.233804, 0.334935, 0.617312, 0.135759, 0.283415, 0.203682, 0.554219],
              'efficient': [0.233804, 0.334934, 0.617307, 0.135759, 0.283414, 0.203681, 0.554219],
              'legacy': [0.232635, 0.333749, 0.617413, 0.138463, 0.286415, 0.203624, 0.554315]}
    for v in VARIANTS3:
        np.testing.assert_allclose(g['iou_' + v], expect[v], atol=2e-6)
        for planar in ('diff', 'mmcv', 'exact'):
            o = oracle.iou_aligned(g['b1'], g['b2'], variant=v, planar=planar)
            np.testing.assert_allclose(o, g['iou_' + v], atol=3e-6, err_msg=f'{v}/{planar}')


@pytest.mark.parametrize('v', VARIANTS3)
def test_edge_cases(oracle, v):
    g = load_golden('edge_cases')
    for mode in ('iou', 'iof'):
        ref = g[f'{mode}_{v}']
        o = oracle.iou_aligned(g['b1'], g['b2'], variant=v, mode=mode, planar='diff')
        ok = np.isfinite(ref)
        # rows 7/8 (pole, identical after jitter) are ill-conditioned by construction: loose bound there
        np.testing.assert_allclose(o[ok], ref[ok], atol=2e-3, err_msg=f'{v}/{mode}')
        well = np.array([1, 2, 3, 4, 5, 6, 9, 10])
        np.testing.assert_allclose(o[well], ref[well], atol=2e-5, err_msg=f'{v}/{mode}')


@pytest.mark.parametrize('name,variants', [('uniform_bfov', VARIANTS3), ('nearby_bfov', VARIANTS3),
                                           ('int_bfov', VARIANTS3), ('uniform_rbfov', VARIANTS3[:2]),
                                           ('nearby_rbfov', VARIANTS3[:2])])
def test_random_sets_vs_reference_fp32(oracle, name, variants):
    g = load_golden(name)
    for v in variants:
        ref = g['iou_' + v]
        o = oracle.iou_aligned(g['b1'], g['b2'], variant=v, planar='diff')
        ok = np.isfinite(ref) & np.isfinite(o)
        assert ok.mean() > 0.995, (name, v)
        s = err_stats(o[ok], ref[ok])
        # int_bfov = integer-degree boxes a few degrees apart: small arcs, the reference's noisiest regime
        assert s['mean'] < (3e-6 if name == 'int_bfov' else 1e-6), (name, v, s)
        assert s['n5'] <= 0.025 * s['n'], (name, v, s)
        if name.startswith('uniform'):
            assert s['max'] < 1e-4, (name, v, s)


@pytest.mark.parametrize('name,variants', [('uniform_bfov', VARIANTS3), ('nearby_rbfov', VARIANTS3[:2])])
def test_fp64_instantiation_matches_reference_fp64(oracle, name, variants):
    g = load_golden(name)
    for v in variants:
        ref = g['iou64_' + v]
        o = oracle.iou_aligned(g['b1'], g['b2'], variant=v, planar='diff', dtype=np.float64)
        ok = np.isfinite(ref) & np.isfinite(o)
        s = err_stats(o[ok], ref[ok])
        assert s['max'] < 1e-9, (name, v, s)


@pytest.mark.parametrize('name,variants', [('uniform_bfov', VARIANTS3), ('nearby_bfov', VARIANTS3),
                                           ('nearby_rbfov', VARIANTS3[:2])])
def test_transform_stage(oracle, name, variants):
    """Planar boxes out of the transforms (before jitter) against the reference's own transform outputs."""
    g = load_golden(name)
    for v in variants:
        o1, o2 = oracle.transform(g['b1'], g['b2'], variant=v)
        for o, key in ((o1, 'planar1_'), (o2, 'planar2_')):
            ref = g[key + v]
            ok = np.isfinite(ref).all(1) & np.isfinite(o).all(1)
            d = np.abs(o[ok] - ref[ok])
            assert np.median(d) < 1e-6, (name, v, key)
            assert (d.max(1) > 1e-4).mean() < 0.01, (name, v, key, d.max())


def test_options_matrix(oracle):
    g = load_golden('options')
    for key, ref in g.items():
        if key in ('b1', 'b2', 'r1', 'r2'):
            continue
        box, v, edge, ang, mode = key.split('_')
        b1, b2 = (g['b1'], g['b2']) if box == 'bfov' else (g['r1'], g['r2'])
        o = oracle.iou_aligned(b1, b2, variant=v, mode=mode, edge=edge, angle=ang, planar='diff')
        s = err_stats(o, ref)
        assert s['mean'] < 2e-6 and s['n4'] <= 3, (key, s)


def test_pairwise_rows_are_first_argument(oracle):
    g = load_golden('pairwise')
    for v in VARIANTS3:
        o = oracle.iou_pairwise(g['b1'], g['b2'], variant=v, planar='diff')
        assert o.shape == (7, 11)
        np.testing.assert_allclose(o, g['iou_' + v], atol=5e-5)
    for v in VARIANTS3[:2]:
        o = oracle.iou_pairwise(g['r1'], g['r2'], variant=v, planar='diff')
        np.testing.assert_allclose(o, g['riou_' + v], atol=5e-5)


def test_planar_diff_restatement(oracle):
    g = load_golden('planar')
    o = oracle.planar_iou(g['p1'], g['p2'], planar='diff')
    s = err_stats(o, g['iou'])
    assert s['mean'] < 1e-7 and s['n5'] <= 3, s
    o64 = oracle.planar_iou(g['p1'], g['p2'], planar='diff', dtype=np.float64)
    s = err_stats(o64, g['iou64'])
    assert s['n5'] == 0, s


def test_planar_mmcv_restatement_agrees_with_vendored_diff(oracle):
    """Reference's own pin of the un-vendored mmcv kernel: mean |box_iou_rotated - diff_iou_rotated_2d| < 1e-6
    (tests/test_sph_iou_loss.py:34).  Also the tolerance-free exact clip as referee."""
    g = load_golden('planar')
    m = oracle.planar_iou(g['p1'], g['p2'], planar='mmcv')
    e = oracle.planar_iou(g['p1'], g['p2'], planar='exact', dtype=np.float64)
    assert np.abs(m - g['iou']).mean() < 1e-6
    assert np.abs(m - e).mean() < 1e-6
    assert np.abs(g['iou'] - e).mean() < 1e-6


@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
def test_loss_values(oracle, box):
    g = load_golden('loss_' + box)
    for mode in ('iou', 'giou', 'diou', 'ciou'):
        o = oracle.loss_elements(g['pred'], g['target'], mode=mode)
        s = err_stats(o, g['loss_' + mode])
        assert s['mean'] < 2e-6 and s['n4'] <= 4, (mode, s)
        o64 = oracle.loss_elements(g['pred'], g['target'], mode=mode, dtype=np.float64)
        s = err_stats(o64, g['loss64_' + mode])
        assert s['max'] < 1e-8, (mode, s)
    kw = dict(mode='ciou', loss_weight=2.0)
    np.testing.assert_allclose(oracle.sph2pob_iou_loss(g['pred'], g['target'], **kw), g['mean_ciou'], rtol=2e-5)
    np.testing.assert_allclose(oracle.sph2pob_iou_loss(g['pred'], g['target'], g['w1'], **kw), g['mean_ciou_w1'],
                               rtol=2e-5)
    np.testing.assert_allclose(oracle.sph2pob_iou_loss(g['pred'], g['target'], g['w2'], **kw), g['mean_ciou_w2'],
                               rtol=2e-5)
    np.testing.assert_allclose(oracle.sph2pob_iou_loss(g['pred'], g['target'], g['w2'], avg_factor=123.0, **kw),
                               g['mean_ciou_w2_avg'], rtol=2e-5)
    np.testing.assert_allclose(oracle.sph2pob_iou_loss(g['pred'], g['target'], g['w1'], reduction='sum', **kw),
                               g['sum_ciou_w1'], rtol=2e-5)


@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
def test_loss_grad_fd_matches_reference_autograd(oracle, box):
    """fp64 finite differences of the oracle vs the reference's fp32 autograd (bulk agreement)."""
    g = load_golden('loss_' + box)
    sl = slice(0, 120)
    for mode in ('iou', 'ciou'):
        gp, gt = oracle.loss_grad_fd(g['pred'][sl], g['target'][sl], mode=mode)
        for fd, ref in ((gp, g['gpred_' + mode][sl]), (gt, g['gtarget_' + mode][sl])):
            d = np.abs(fd - ref)
            scale = np.abs(ref).max()
            assert np.median(d) < 2e-4 * scale, (box, mode, np.median(d), scale)
            assert (d > 0.05 * scale).mean() < 0.03, (box, mode)


def test_nms_keep_lists(oracle):
    g = load_golden('nms')
    dets, keep = oracle.batched_nms(g['boxes'], g['scores'], g['idxs'], 0.5)
    assert keep.tolist() == [0, 5, 7, 3, 8, 9]  # SURVEY App. C.4 / reference tests/test_nms.py scenario
    assert keep.tolist() == g['keep'].tolist()
    np.testing.assert_allclose(dets, g['dets'], atol=1e-6)
    dets, keep = oracle.batched_nms(g['rboxes'], g['rscores'], g['ridxs'], 0.5, max_num=100)
    assert keep.tolist() == g['rkeep'].tolist()
    dets, keep = oracle.batched_nms(g['r5boxes'], g['r5scores'], g['r5idxs'], 0.4)
    assert keep.tolist() == g['r5keep'].tolist()
    np.testing.assert_allclose(dets, g['r5dets'], atol=1e-6)


def test_cheap_backends_sph_iou_fov_iou(oracle):
    """Sph-IoU / FoV-IoU closed forms (sphdet/iou/approximate_ious.py:3-54 behind sph_iou_api.py:128-175)."""
    g = load_golden('approx')
    for v in ('sph_iou', 'fov_iou'):
        np.testing.assert_allclose(oracle.iou_aligned(g['b1'], g['b2'], variant=v), g[v], atol=3e-7)
        np.testing.assert_allclose(oracle.iou_aligned(g['b1'], g['b2'], variant=v, dtype=np.float64), g[v + '64'], atol=1e-12)
        np.testing.assert_allclose(oracle.iou_pairwise(g['pa'], g['pb'], variant=v), g[v + '_pw'], atol=3e-7)
