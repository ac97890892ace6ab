"""GPU: the reference head's hot path end to end (tools/demo_hot_path.py) — assigner -> targets -> decode -> loss ->
backward -> NMS compose on device tensors the way sph_retina_head.py strings them together."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('backend,nms', [('sph2pob_standard_iou', 'sph2pob_efficient'), ('unbiased_iou', 'unbiased_iou')])
def test_hot_path_end_to_end(oracle, backend, nms):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import demo_hot_path
    info, d = demo_hot_path.run(num_gt=16, backend=backend, nms_calculator=nms)
    assert info['anchors'] == 98208 and info['num_pos'] > 0
    assert np.isfinite(info['loss']) and info['loss'] > 0
    # gradients reach exactly the positive anchors' deltas (weight 0 elsewhere)
    grad_rows = d['deltas'].grad.abs().sum(1) > 0
    assert torch.equal(grad_rows & ~d['pos'], torch.zeros_like(grad_rows)) and int(grad_rows.sum()) >= 0.9 * info['num_pos']
    # assigner result against the restatement on the same overlaps
    import sph_retina_amd as S
    ov = S.SphOverlaps2D(backend=backend, box_version=4)(d['gt'], d['anchors'])
    gi = oracle.assign_wrt_overlaps(ov.cpu().numpy(), None, pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0)[0]
    assert np.array_equal(d['assign'].gt_inds.cpu().numpy(), gi)
    # encode(anchor, target) decodes back to the target
    coder = S.DeltaXYWHSphBBoxCoder()
    back = coder.decode(d['anchors'][d['pos']], d['enc'])
    assert torch.allclose(back, d['bbox_targets'][d['pos']], atol=2e-3)
    # detections: <= 100, sorted by score, valid boxes
    dets = d['dets']
    assert dets.shape[1] == 5 and dets.shape[0] <= 100 and bool((dets[:-1, 4] >= dets[1:, 4]).all())
    assert bool((dets[:, 2:4] > 0).all())
