"""CPU: pin the numpy restatement of MaxIoUAssigner (oracle.assign_wrt_overlaps / oracle.assign) against fixtures produced by
the reference's REAL class (tests/golden/assign.npz from oracle/gen_goldens.py `assign`:
mmdet/core/bbox/assigners/max_iou_assigner.py:67-220 loaded unmodified).  The GPU kernels are then held to this
restatement bit for bit (tests/test_gpu_assigner.py)."""
import numpy as np
import pytest

from conftest import load_golden

# the configurations of oracle/gen_goldens.py:ASSIGN_CFGS, in order
CFGS = (dict(pos_iou_thr=0.8, neg_iou_thr=0.75, min_pos_iou=0.0),
        dict(pos_iou_thr=0.9, neg_iou_thr=(0.1, 0.8), min_pos_iou=0.75, gt_max_assign_all=False),
        dict(pos_iou_thr=0.8, neg_iou_thr=0.8, match_low_quality=False),
        dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.3),
        dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0, gt_max_assign_all=False))


def test_fixture_covers_what_it_claims():
    g = load_golden('assign')
    assert int(g['n_cfg']) == len(CFGS)
    ov = g['s2_ov']
    assert (ov == -1).all(0).any() and (ov[0] == ov[0].max()).sum() >= 2          # ignored columns, a row-maximum tie
    assert (ov[1][ov[1] >= 0] == 0).all()                                         # a GT overlapping nothing
    col = ov.shape[1] // 2
    assert ov[0, col] == ov[2, col] == ov[:, col].max()                           # a column-maximum tie
    assert (g['s6_ov'] == -1).all()                                               # everything ignored
    # the reference really does hand a zero-overlap GT every anchor it does not overlap (min_pos_iou = 0, assign-all)
    assert (g['s2_c0_gt_inds'] == 2).sum() > 500


@pytest.mark.parametrize('si', range(7))
def test_assign_wrt_overlaps_equals_reference(oracle, si):
    g = load_golden('assign')
    ov, labels = g[f's{si}_ov'], g[f's{si}_labels']
    for ci, cfg in enumerate(CFGS):
        gi, mo, _amo, _gm, _gam, lab = oracle.assign_wrt_overlaps(ov, labels, **cfg)
        np.testing.assert_array_equal(gi, g[f's{si}_c{ci}_gt_inds'], err_msg=f'cfg {ci}')
        np.testing.assert_array_equal(lab, g[f's{si}_c{ci}_labels'], err_msg=f'cfg {ci}')
        np.testing.assert_array_equal(mo, g[f's{si}_max_overlaps'])
        assert oracle.assign_wrt_overlaps(ov, None, **cfg)[5] is None


@pytest.mark.parametrize('tag,kw', [('plain', {}), ('ignc', dict(ignore_iof_thr=0.5)),
                                    ('ignb', dict(ignore_iof_thr=0.5, ignore_wrt_candidates=False))])
def test_assign_on_spherical_boxes_equals_reference(oracle, tag, kw):
    """`assign` with the reference's overlaps: the C oracle's IoUs are within fp32 noise of the reference's, which can flip
    an anchor sitting on a threshold, so the reference's own matrix is used where the matrix is not what is tested; the
    ignore step runs on the oracle's IoF (its decisions are far from the 0.5 threshold for all but a few anchors)."""
    g = load_golden('assign')
    gt, anchors, ignore, labels = g['b_gt'], g['b_anchors'], g['b_ignore'], g['b_labels']
    ref_ov = g['b_overlaps']
    mine = oracle.iou_pairwise(gt, anchors, variant='standard', planar='diff')
    assert np.abs(mine - ref_ov).max() < 2e-3 and np.abs(mine - ref_ov).mean() < 1e-6

    def iou_fn(a, b, mode):
        if mode == 'iou':
            return ref_ov
        return oracle.iou_pairwise(a, b, variant='standard', mode='iof', planar='diff')
    for ci, cfg in enumerate(CFGS):
        ov, (gi, mo, *_r, lab) = oracle.assign(anchors, gt, iou_fn, gt_bboxes_ignore=ignore if kw else None, gt_labels=labels,
                                              **cfg, **kw)
        want = g[f'b_c{ci}_{tag}_gt_inds']
        bad = gi != want
        # anchors whose ignore decision sits on the threshold in the oracle's rounding may differ; nothing else may
        assert bad.sum() <= (3 if kw else 0), (ci, tag, int(bad.sum()))
        np.testing.assert_array_equal(lab[~bad], g[f'b_c{ci}_{tag}_labels'][~bad])
        if ci == 0:
            np.testing.assert_array_equal(mo[~bad], g[f'b_{tag}_max_overlaps'][~bad])
    if kw:
        assert (g[f'b_c0_{tag}_gt_inds'] == -1).sum() > 20   # the ignore path really ignored something
