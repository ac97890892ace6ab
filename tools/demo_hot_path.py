"""GPU box: the reference head's hot path end to end on synthetic data, the way sphdet/models/heads/sph_retina_head.py
strings it together (no network — the regression deltas are a leaf tensor):

  training   anchors x GT -> SphMaxIoUAssigner(SphOverlaps2D)            (get_targets, _get_targets_single)
             decode(anchors, deltas) -> Sph2PobIoULoss(ciou), weights 0 on negatives, avg_factor = #pos  (loss_single :246-264)
             backward to the deltas
  inference  decode -> multiclass_nms(SphNMS)                              (_get_bboxes_single / test_cfg)

Prints one JSON line with the stage times; `run()` is also used by tests/test_gpu_pipeline.py.
"""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sph_retina_amd as S  # noqa: E402
from sph_retina_amd.bbox.nms import multiclass_nms  # noqa: E402


def retina_anchors(h=512, w=1024, device='cuda'):
    """5-level, 9-anchor RetinaNet grid (configs/_base_/models/sph_retinanet_r50_fpn.py:28-35) in spherical degrees
    (sphdet/bbox/box_formator.py:85-92): 98 208 anchors for the reference's default 512 x 1024 ERP."""
    out = []
    for stride in (8, 16, 32, 64, 128):
        fh, fw = math.ceil(h / stride), math.ceil(w / stride)
        ys, xs = torch.meshgrid(torch.arange(fh, dtype=torch.float32), torch.arange(fw, dtype=torch.float32), indexing='ij')
        cx, cy = (xs.reshape(-1) + 0.5) * stride, (ys.reshape(-1) + 0.5) * stride
        for scale in (2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)):
            for ratio in (0.5, 1.0, 2.0):
                bw = 4 * stride * scale / math.sqrt(ratio)
                bh = 4 * stride * scale * math.sqrt(ratio)
                out.append(torch.stack([cx / w * 360, cy / h * 180, torch.full_like(cx, bw / w * 360),
                                        torch.full_like(cx, bh / h * 180)], 1))
    a = torch.cat(out).to(device)
    a[:, 2:] = a[:, 2:].clamp(max=179.0)
    return a.contiguous()


def run(num_gt=64, num_classes=37, backend='sph2pob_standard_iou', nms_calculator='sph2pob_efficient', seed=0, reps=1):
    dev = 'cuda'
    g = torch.Generator().manual_seed(seed)
    anchors = retina_anchors()
    u = torch.rand((num_gt, 4), generator=g)
    gt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1).to(dev)
    gt_labels = torch.randint(0, num_classes, (num_gt,), generator=g).to(dev)
    assigner = S.SphMaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1,
                                   iou_calculator=dict(type='SphOverlaps2D', backend=backend, box_version=4))
    coder = S.DeltaXYWHSphBBoxCoder(target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.))
    loss_bbox = S.Sph2PobIoULoss(mode='ciou', loss_weight=1.0)
    deltas = (torch.randn((anchors.size(0), 4), generator=g) * 0.05).to(dev).requires_grad_(True)
    cls_scores = torch.rand((anchors.size(0), num_classes + 1), generator=g).to(dev) ** 8   # sparse high scores
    t = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = fn()
        torch.cuda.synchronize()
        t[name + '_ms'] = (time.perf_counter() - t0) / reps * 1e3
        return r

    assign = timed('assign', lambda: assigner.assign(anchors, gt, gt_labels=gt_labels))
    pos = assign.gt_inds > 0
    num_pos = int(pos.sum())
    # regression targets / weights as _get_targets_single builds them with reg_decoded_bbox=True
    bbox_targets = torch.zeros_like(anchors)
    bbox_weights = torch.zeros_like(anchors)
    bbox_targets[pos] = gt[assign.gt_inds[pos] - 1]
    bbox_weights[pos] = 1.0

    def train_step():
        deltas.grad = None
        pred = coder.decode(anchors, deltas)
        loss = loss_bbox(pred, bbox_targets, bbox_weights, avg_factor=max(num_pos, 1))
        loss.backward()
        return loss.detach(), pred.detach()
    loss, pred = timed('decode_loss_backward', train_step)
    enc = timed('encode', lambda: coder.encode(anchors[pos], bbox_targets[pos]))

    def infer():
        boxes = coder.decode(anchors, deltas.detach())
        nms_pre = 1000                                           # per-level nms_pre, here one top-k over all levels
        top = cls_scores[:, :-1].max(1).values.topk(5 * nms_pre).indices
        return multiclass_nms(boxes[top], cls_scores[top], 0.05, dict(type='nms', iou_threshold=0.5), max_num=100,
                              nms_op=S.SphNMS(nms_calculator), box_version=4)
    dets, labels = timed('decode_topk_nms', infer)
    out = {'anchors': anchors.size(0), 'num_gt': num_gt, 'num_pos': num_pos, 'loss': float(loss),
           'grad_nonzero_rows': int((deltas.grad.abs().sum(1) > 0).sum()), 'dets': tuple(dets.shape),
           'backend': backend, 'nms': nms_calculator, **t}
    return out, dict(assign=assign, pos=pos, pred=pred, deltas=deltas, enc=enc, dets=dets, labels=labels, anchors=anchors,
                     gt=gt, bbox_targets=bbox_targets)


if __name__ == '__main__':
    run()                                   # warm-up (lazy initialisation, workspace allocation)
    for backend, nms in (('sph2pob_standard_iou', 'sph2pob_efficient'), ('unbiased_iou', 'unbiased_iou'),
                         ('naive_iou', 'naive_iou')):
        print(json.dumps(run(backend=backend, nms_calculator=nms, reps=5)[0]), flush=True)
