"""`multiclass_nms(..., nms_op=, box_version=)` — the reference's variant of mmdet's helper
(sphdet/bbox/nms/utils.py:6-94) whose `nms_op` is a SphNMS instance and whose boxes have `box_version` columns."""
import torch

from .sph_nms import SphNMS


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None, return_inds=False,
                   nms_op=None, box_version=4):
    """-> (dets (k, box_version + 1), labels (k,)[, inds (k,)]); the last score column (background) is ignored."""
    nms_op = nms_op if nms_op is not None else SphNMS()
    n, num_classes = multi_scores.size(0), multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > box_version:
        bboxes = multi_bboxes.view(n, -1, box_version)
    else:
        bboxes = multi_bboxes[:, None].expand(n, num_classes, box_version)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device).view(1, -1).expand_as(scores)
    bboxes, scores, labels = bboxes.reshape(-1, box_version), scores.reshape(-1), labels.reshape(-1)
    valid = scores > score_thr
    if score_factors is not None:
        scores = scores * score_factors.view(-1, 1).expand(n, num_classes).reshape(-1)
    inds = valid.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if bboxes.numel() == 0:
        dets = torch.cat([bboxes, scores[:, None]], -1)
        return (dets, labels, inds) if return_inds else (dets, labels)
    dets, keep = nms_op(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return (dets, labels[keep], inds[keep]) if return_inds else (dets, labels[keep])
