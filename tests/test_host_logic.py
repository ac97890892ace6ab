"""CPU: host-side logic of the boundary that needs no GPU — registries / config strings of the reference, option
assertions, empty-input conventions, sharding arithmetic, the torch restatements of the two jitter helpers."""
import numpy as np
import pytest
import torch

import sph_retina_amd as S
from sph_retina_amd import registry as R


def test_registry_builds_reference_config_strings():
    """configs/retinanet/sph_retinanet_r50_fpn_120e_indoor360.py:37-40 and ..._obb_ciou_loss.py:5-13."""
    if R.IOU_CALCULATORS_IS_MMDET:
        pytest.skip('mmdet present: its own registries are used')
    calc = R.build_iou_calculator(dict(type='SphOverlaps2D', backend='sph2pob_standard_iou', box_version=4))
    assert isinstance(calc, S.SphOverlaps2D) and calc.backend == 'sph2pob_standard_iou' and calc.box_version == 4
    loss = R.build_loss(dict(type='Sph2PobIoULoss', mode='ciou', loss_weight=1.0))
    assert isinstance(loss, S.Sph2PobIoULoss) and loss.mode == 'ciou' and loss.reduction == 'mean' and loss.eps == 1e-6
    assigner = R.build_assigner(dict(type='SphMaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                                     ignore_iof_thr=-1))
    assert isinstance(assigner.iou_calculator, S.SphOverlaps2D)
    for name in ('SphOverlaps2D',):
        assert name in R.IOU_CALCULATORS
    for name in ('Sph2PobIoULoss', 'SphIoULoss'):
        assert name in R.LOSSES
    with pytest.raises(KeyError):
        R.build_loss(dict(type='NoSuchLoss'))


def test_option_assertions_match_reference():
    a, b = torch.rand(3, 4), torch.rand(3, 4)
    for kw in (dict(mode='giou'), dict(calculator='fast'), dict(rbb_edge='secant'), dict(rbb_angle='pole')):
        with pytest.raises(AssertionError):   # sph_iou_api.py:49-51
            S.sph2pob_standard_iou(a, b, **kw)
    with pytest.raises(AssertionError):       # sph_iou_calculator.py:75-76
        S.sph_overlaps(a, b, backend='xinyuan')
    with pytest.raises(AssertionError):
        S.SphOverlaps2D(backend='sph2pob_standard_iou')(torch.rand(3, 7), b)
    with pytest.raises(NotImplementedError):  # Kent backends are outside the hot path
        S.sph_overlaps(a, b, backend='kent_iou')
    with pytest.raises(AssertionError):
        S.Sph2PobIoULoss(mode='linear')       # sph2pob_iou_loss.py:19
    with pytest.raises(TypeError):
        S.SphNMS('planar')                    # the reference's `raise NotImplemented(...)` is a TypeError too


def test_empty_conventions_without_gpu():
    e4 = torch.rand(0, 4)
    assert S.sph2pob_efficient_iou(e4, torch.rand(5, 4)).shape == (0, 5)
    assert S.sph2pob_efficient_iou(torch.rand(5, 4), e4).shape == (5, 0)
    assert S.sph2pob_legacy_iou(e4, e4, is_aligned=True).shape == (0, 1)      # sph_iou_api.py:56-57
    assert S.SphOverlaps2D('sph2pob_standard_iou')(torch.rand(0, 5), torch.rand(3, 5)).shape == (0, 3)
    from sph_retina_amd.bbox.assigners import assign_wrt_overlaps
    r = assign_wrt_overlaps(torch.zeros(0, 6), gt_labels=torch.zeros(0, dtype=torch.long))
    assert r.gt_inds.tolist() == [0] * 6 and r.labels.tolist() == [-1] * 6 and r.num_gts == 0
    r = assign_wrt_overlaps(torch.zeros(3, 0))
    assert r.gt_inds.numel() == 0 and r.labels is None


def test_jitter_helpers_match_oracle():
    """jiter_spherical_bboxes / jiter_rotated_bboxes (sph_iou_api.py:222-260) as importable in-place helpers."""
    from sph_retina_amd.iou import jiter_rotated_bboxes, jiter_spherical_bboxes
    from oracle import oracle as O
    b1 = torch.from_numpy(O.generate_boxes(500, 1, box='rbfov'))
    b2 = b1.clone()
    b2[::3] += 0.5
    b2[1::3, 2] += 3.0
    b2[:, 1:4].clamp_(0.0, 200.0)
    j1, j2 = jiter_spherical_bboxes(b1.clone(), b2.clone())
    # oracle: transform(jitter=True) applies the same spherical jitter before the transform; compare through IoU inputs
    same = ((b1 - b2).abs() < 1.2345678e-4).any(1)
    assert torch.equal(j1[~same][:, 0], b1[~same][:, 0].clamp(2.4691356e-4, 360 - 1.2345678e-4))
    assert same.any() and torch.allclose(j1[same][:, 4], b1[same][:, 4] - 2.4691356e-4, atol=1e-6)  # gamma of box 1: shifted, never clamped
    assert float(j2[:, 1:4].max()) <= 180 - 2 * 1.2345678e-4 + 1e-4 and float(j1[:, 1:4].max()) <= 180
    p1 = torch.tensor([[0.0, 0.0, 0.5, 0.25, 0.10], [0.0, 0.0, 1e-5, 0.3, 3.0]])
    p2 = torch.tensor([[0.2, 0.0, 0.5, 0.40, 0.1005], [0.3, 0.0, 0.2, 0.3, -3.0]])
    q1, q2 = jiter_rotated_bboxes(p1.clone(), p2.clone())
    e, ea = 1.2345678e-4, 1.2345678e-3
    # row 0: w equal -> similar -> += (e,e,2e,2e,e) / (2e,2e,e,e,5e); then |a1-a2| < ea -> += ea / 2ea
    np.testing.assert_allclose(q1[0].numpy(), [e, e, 0.5 + 2 * e, 0.25 + 2 * e, 0.10 + e + ea], atol=1e-7)
    np.testing.assert_allclose(q2[0].numpy(), [0.2 + 2 * e, 2 * e, 0.5 + e, 0.40 + e, 0.1005 + 5 * e + 2 * ea], atol=1e-7)
    # row 1: h equal -> similar; tiny w clamped from below
    assert abs(float(q1[1, 2]) - max(1e-5 + 2 * e, 2 * ea / 10)) < 1e-7


def test_arithmetic_switch_and_flags():
    from sph_retina_amd import _torch_glue as G
    assert S.get_arithmetic() in ('fast', 'reference')
    S.set_arithmetic('reference')
    assert G.VARIANTS['standard'] == 0x100 and G.VARIANTS['legacy'] == 0x102
    S.set_arithmetic('fast')
    assert G.VARIANTS['efficient'] == 1
    with pytest.raises(AssertionError):
        S.set_arithmetic('sloppy')
