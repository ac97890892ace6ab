"""Container-only: emit tests/golden/*.npz from the UNMODIFIED reference Python (see oracle/ref_loader.py).

Run:  python -B oracle/gen_goldens.py        (needs /root/reference; never runs on the GPU box)

Every fixture is DATA: seeded inputs + the reference's outputs (fp32, and the same code under
torch.set_default_dtype(float64) as accuracy truth).  The planar stage behind ``box_iou_rotated`` is the
reference's vendored ``sphdet/iou/diff_iou_rotated.py`` (mmcv is not installable offline) — recorded in every
file as ``planar='diff'``.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ref_loader import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
R = load_reference()
IOU = {'standard': R.api.sph2pob_standard_iou, 'efficient': R.api.sph2pob_efficient_iou,
       'legacy': R.api.sph2pob_legacy_iou}
TRANS = {'standard': R.std.sph2pob_standard, 'efficient': R.eff.sph2pob_efficient, 'legacy': R.leg.sph2pob_legacy}


def f64(fn, *tensors, **kw):
    torch.set_default_dtype(torch.float64)
    try:
        return fn(*[t.double() for t in tensors], **kw)
    finally:
        torch.set_default_dtype(torch.float32)


def gen(n, box='bfov', near=False, alpha=(1, 100), beta=(1, 100), gamma=(-90, 90)):
    g = R.gen.generate_boxes(n, (0, 360), (0, 180), alpha, beta, gamma, dtype='float', box=box)
    if near:
        sig = torch.tensor([8., 8., 6., 6., 10.])[:g.shape[1]]
        p = g + torch.randn_like(g) * sig
        p[:, 0] = p[:, 0] % 360
        p[:, 1] = p[:, 1].clamp(0.5, 179.5)
        p[:, 2:4] = p[:, 2:4].clamp(1, 170)
        if g.shape[1] == 5:
            p[:, 4] = p[:, 4].clamp(-89, 89)
    else:
        p = R.gen.generate_boxes(n, (0, 360), (0, 180), alpha, beta, gamma, dtype='float', box=box)
    return g, p


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, {k: tuple(v.shape) for k, v in out.items()})


def iou_set(name, b1, b2, variants, aligned=True):
    arrs = dict(b1=b1, b2=b2)
    for v in variants:
        arrs['iou_' + v] = IOU[v](b1, b2, is_aligned=aligned)
        arrs['iou64_' + v] = f64(IOU[v], b1, b2, is_aligned=aligned)
        t1, t2 = TRANS[v](b1.clone(), b2.clone(), rbb_angle_version='rad')
        arrs['planar1_' + v], arrs['planar2_' + v] = t1, t2
    save(name, **arrs)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(20231024)

    # 1. the reference's own hard-coded sample pairs: tests/test_all_ious.py:244-261 (after geo2sph)
    b1 = torch.tensor([[40, 50, 35, 55], [30, 60, 60, 60], [50, -78, 25, 46], [30, 75, 30, 60], [40, 70, 25, 30],
                       [30, 75, 30, 30], [30, 60, 60, 60]]).float()
    b2 = torch.tensor([[35, 20, 37, 50], [55, 40, 60, 60], [30, -75, 26, 45], [60, 40, 60, 60], [60, 85, 30, 30],
                       [60, 55, 40, 50], [60, 60, 60, 60]]).float()
    iou_set('samples7', R.box_formator.geo2sph(b1), R.box_formator.geo2sph(b2), ['standard', 'efficient', 'legacy'])

    # 2. edge cases incl. identical / seam / pole / antipodal / contained / clamped alpha
    e1 = torch.tensor([[100, 80, 40, 30], [1, 90, 30, 30], [10, 2, 30, 30], [10, 60, 50, 50], [100, 90, 60, 60],
                       [100, 90, 60, 20], [100, 90, 200, 20], [0, 0, 10, 10], [359.9999, 179.9999, 10, 10],
                       [180, 90, 1, 1], [180, 90, 179, 179], [20, 40, 30, 30]]).float()
    e2 = torch.tensor([[100, 80, 40, 30], [359, 90, 30, 30], [200, 2, 30, 30], [190, 120, 50, 50],
                       [100, 90, 20, 20], [100, 90, 20, 60], [120, 90, 200, 20], [0, 0, 10, 10],
                       [0.0001, 179.9999, 10, 10], [180.5, 90.2, 1, 1], [181, 91, 179, 179],
                       [20.00005, 40, 30, 30]]).float()
    arrs = dict(b1=e1, b2=e2)
    for v in IOU:
        arrs['iou_' + v] = IOU[v](e1, e2, is_aligned=True)
        arrs['iof_' + v] = IOU[v](e1, e2, mode='iof', is_aligned=True)
    save('edge_cases', **arrs)

    # 3. seeded random sets, BFoV: uniform (benchmark distribution), detector-like nearby, integer degrees
    g, p = gen(2000)
    iou_set('uniform_bfov', g, p, ['standard', 'efficient', 'legacy'])
    g, p = gen(2000, near=True)
    iou_set('nearby_bfov', g, p, ['standard', 'efficient', 'legacy'])
    gi = R.gen.generate_boxes(1500, (0, 360), (1, 180), (1, 100), (1, 100), dtype='int')
    pi = (gi + torch.randint(-6, 7, gi.shape).float())
    pi[:, 0] = pi[:, 0] % 360
    pi[:, 1] = pi[:, 1].clamp(1, 179)
    pi[:, 2:] = pi[:, 2:].clamp(1, 170)
    iou_set('int_bfov', gi, pi, ['standard', 'efficient', 'legacy'])

    # 4. RBFoV (5-DoF)
    g, p = gen(2000, box='rbfov')
    iou_set('uniform_rbfov', g, p, ['standard', 'efficient'])
    g, p = gen(2000, box='rbfov', near=True)
    iou_set('nearby_rbfov', g, p, ['standard', 'efficient'])

    # 5. options: rbb_edge x rbb_angle x mode on nearby pairs
    g, p = gen(300, near=True)
    g5, p5 = gen(300, box='rbfov', near=True)
    arrs = dict(b1=g, b2=p, r1=g5, r2=p5)
    for v in ('standard', 'efficient'):
        for edge in ('arc', 'chord', 'tangent'):
            for ang in ('equator', 'project'):
                for mode in ('iou', 'iof'):
                    key = f'{v}_{edge}_{ang}_{mode}'
                    arrs['bfov_' + key] = IOU[v](g, p, mode=mode, is_aligned=True, rbb_edge=edge, rbb_angle=ang)
                    arrs['rbfov_' + key] = IOU[v](g5, p5, mode=mode, is_aligned=True, rbb_edge=edge, rbb_angle=ang)
    for edge in ('arc', 'chord', 'tangent'):
        for mode in ('iou', 'iof'):
            arrs[f'bfov_legacy_{edge}_equator_{mode}'] = IOU['legacy'](g, p, mode=mode, is_aligned=True, rbb_edge=edge)
    save('options', **arrs)

    # 6. pairwise (rows = first argument), incl. the MaxIoUAssigner call pattern iou(gt, anchors)
    a, _ = gen(7)
    b, _ = gen(11, near=False)
    b[:7] = a + torch.randn(7, 4) * 3
    b[:, 1] = b[:, 1].clamp(1, 179)
    b[:, 2:] = b[:, 2:].clamp(1, 170)
    arrs = dict(b1=a, b2=b)
    for v in IOU:
        arrs['iou_' + v] = IOU[v](a, b, is_aligned=False)
    a5, _ = gen(5, box='rbfov')
    b5 = a5[[0, 1, 2, 3, 4, 0, 2]] + torch.randn(7, 5) * 4
    b5[:, 1] = b5[:, 1].clamp(1, 179)
    b5[:, 2:4] = b5[:, 2:4].clamp(1, 170)
    arrs.update(r1=a5, r2=b5)
    for v in ('standard', 'efficient'):
        arrs['riou_' + v] = IOU[v](a5, b5, is_aligned=False)
    save('pairwise', **arrs)

    # 7. planar rotated boxes -> vendored diff_iou_rotated_2d (sphdet/iou/diff_iou_rotated.py:325-343)
    n = 3000
    c = torch.randn(n, 2) * 0.3
    p1 = torch.cat([c, torch.rand(n, 2) * 1.5 + 0.02, (torch.rand(n, 1) - 0.5) * 6.2], 1)
    p2 = torch.cat([c + torch.randn(n, 2) * 0.3, torch.rand(n, 2) * 1.5 + 0.02, (torch.rand(n, 1) - 0.5) * 6.2], 1)
    save('planar', p1=p1, p2=p2, iou=R.diff.diff_iou_rotated_2d(p1[None], p2[None])[0],
         iou64=f64(R.diff.diff_iou_rotated_2d, p1[None], p2[None])[0])

    # 8. losses: values + grads for iou/giou/diou/ciou; BFoV & RBFoV; weights / avg_factor / reductions
    for box, dim in (('bfov', 4), ('rbfov', 5)):
        tgt, pred = gen(400, box=box, near=True, alpha=(5, 90), beta=(5, 90), gamma=(-60, 60))
        arrs = dict(pred=pred, target=tgt)
        for mode in ('iou', 'giou', 'diou', 'ciou'):
            L = R.iou_loss.Sph2PobIoULoss(mode=mode, reduction='none')
            pr = pred.clone().requires_grad_(True)
            tg = tgt.clone().requires_grad_(True)
            el = L(pr, tg)
            el.sum().backward()
            arrs[f'loss_{mode}'] = el
            arrs[f'gpred_{mode}'] = pr.grad
            arrs[f'gtarget_{mode}'] = tg.grad
            el64 = f64(lambda a, b: L(a, b), pred, tgt)
            arrs[f'loss64_{mode}'] = el64
        w1 = torch.rand(400)
        w2 = torch.rand(400, dim)
        arrs.update(w1=w1, w2=w2)
        Lm = R.iou_loss.Sph2PobIoULoss(mode='ciou', reduction='mean', loss_weight=2.0)
        arrs['mean_ciou'] = Lm(pred, tgt)
        arrs['mean_ciou_w1'] = Lm(pred, tgt, w1)
        arrs['mean_ciou_w2'] = Lm(pred, tgt, w2)
        arrs['mean_ciou_w2_avg'] = Lm(pred, tgt, w2, avg_factor=123.0)
        arrs['sum_ciou_w1'] = Lm(pred, tgt, w1, reduction_override='sum')
        pr = pred.clone().requires_grad_(True)
        Lm(pr, tgt, w2, avg_factor=123.0).backward()
        arrs['gpred_mean_ciou_w2_avg'] = pr.grad
        save('loss_' + box, **arrs)

    # 9. NMS: the reference's 10-box scenario (tests/test_nms.py:6-27) + a seeded random scene
    boxes = torch.tensor([[20, 40, 30, 30], [20, 40, 30, 30], [22, 38, 32, 28], [60, 60, 10, 10], [60, 60, 10, 10],
                          [60, 60, 10, 10], [60, 60, 10, 10], [30, 10, 10, 10], [30, 45, 45, 45],
                          [80, 20, 66, 66]]).float()
    scores = torch.tensor([0.9, 0.8, 0.7, 0.6, 0.5, 0.85, 0.75, 0.65, 0.4, 0.3])
    idxs = torch.tensor([1, 1, 1, 1, 1, 2, 2, 2, 3, 3])
    nms = R.nms.SphNMS('sph2pob_efficient')
    dets, keep = nms(boxes, scores, idxs, dict(type='nms', iou_threshold=0.5))
    arrs = dict(boxes=boxes, scores=scores, idxs=idxs, dets=dets, keep=keep)
    centers, _ = gen(40, alpha=(5, 60), beta=(5, 60))
    rb = centers[torch.randint(0, 40, (400,))] + torch.randn(400, 4) * 2.5
    rb[:, 0] = rb[:, 0] % 360
    rb[:, 1] = rb[:, 1].clamp(1, 179)
    rb[:, 2:] = rb[:, 2:].clamp(2, 120)
    rs = torch.rand(400)
    ri = torch.randint(0, 5, (400,))
    dets, keep = nms(rb, rs, ri, dict(type='nms', iou_threshold=0.5, max_num=100))
    arrs.update(rboxes=rb, rscores=rs, ridxs=ri, rdets=dets, rkeep=keep)
    c5, _ = gen(30, box='rbfov', alpha=(5, 60), beta=(5, 60), gamma=(-60, 60))
    rb5 = c5[torch.randint(0, 30, (300,))] + torch.randn(300, 5) * 2.5
    rb5[:, 0] = rb5[:, 0] % 360
    rb5[:, 1] = rb5[:, 1].clamp(1, 179)
    rb5[:, 2:4] = rb5[:, 2:4].clamp(2, 120)
    rs5 = torch.rand(300)
    ri5 = torch.randint(0, 4, (300,))
    dets, keep = nms(rb5, rs5, ri5, dict(type='nms', iou_threshold=0.4))
    arrs.update(r5boxes=rb5, r5scores=rs5, r5idxs=ri5, r5dets=dets, r5keep=keep)
    save('nms', **arrs)


def approx_only():
    """Sph-IoU / FoV-IoU fixtures (sphdet/iou/sph_iou_api.py:128-175): own seed, does not touch the other files."""
    torch.manual_seed(20231025)
    g, p = gen(1500)
    gn, pn = gen(1500, near=True)
    b1, b2 = torch.cat([g, gn]), torch.cat([p, pn])
    a, _ = gen(6)
    b, _ = gen(9)
    b[:6] = a + torch.randn(6, 4) * 3
    b[:, 1] = b[:, 1].clamp(1, 179)
    b[:, 2:] = b[:, 2:].clamp(1, 170)
    save('approx', b1=b1, b2=b2, sph_iou=R.api.sph_iou(b1, b2, is_aligned=True), fov_iou=R.api.fov_iou(b1, b2, is_aligned=True),
         sph_iou64=f64(R.api.sph_iou, b1, b2, is_aligned=True), fov_iou64=f64(R.api.fov_iou, b1, b2, is_aligned=True),
         pa=a, pb=b, sph_iou_pw=R.api.sph_iou(a, b), fov_iou_pw=R.api.fov_iou(a, b))


def coder_only():
    """Box-coder fixtures (sphdet/bbox/coder/delta_xywh_sph_bbox_coder.py, delta_xywha_rsph_bbox_coder.py): encode,
    decode (single and multi-class, both clamp modes) and the autograd gradient of decode w.r.t. the deltas."""
    torch.manual_seed(20231026)
    arrs = {}
    for dim, mod, cls in ((4, R.coder4, 'DeltaXYWHSphBBoxCoder'), (5, R.coder5, 'DeltaXYWHASphBBoxCoder')):
        n = 600
        anchors, gts = gen(n, box='bfov' if dim == 4 else 'rbfov', near=True)
        anchors[:5, 2] = 0.0                                    # width clip at eps
        means = (0.01, -0.02, 0.03, 0.0, 0.05)[:dim]
        stds = (0.1, 0.1, 0.2, 0.2, 0.1)[:dim]
        coder = getattr(mod, cls)(target_means=means, target_stds=stds)
        enc = coder.encode(anchors, gts)
        deltas = torch.randn(n, dim) * torch.tensor((1.0, 1.0, 1.5, 1.5, 3.0)[:dim])
        deltas[:40] *= 4                                         # drive the ratio clip and the border clamp
        deltas[40:60, 2:4] = 25.0
        dreq = deltas.clone().requires_grad_(True)
        dec = coder.decode(anchors, dreq)
        gout = torch.randn(n, dim)
        (dec * gout).sum().backward()
        ctr = getattr(mod, cls)(target_means=means, target_stds=stds, add_ctr_clamp=True, ctr_clamp=6, clip_border=False)
        dreq2 = deltas.clone().requires_grad_(True)
        dec_ctr = ctr.decode(anchors, dreq2, wh_ratio_clip=0.05)
        (dec_ctr * gout).sum().backward()
        nc = 3
        dm = torch.randn(50, nc * dim)
        dec_mc = mod.delta2bbox(anchors[:50], dm, means, stds)
        plain = mod.delta2bbox(anchors, deltas)                  # default means / stds
        rt = mod.delta2bbox(anchors[5:], mod.bbox2delta(anchors[5:], gts[5:]))  # round trip
        k = f'd{dim}_'
        arrs.update({k + 'anchors': anchors, k + 'gts': gts, k + 'means': torch.tensor(means), k + 'stds': torch.tensor(stds),
                     k + 'enc': enc, k + 'deltas': deltas, k + 'dec': dec.detach(), k + 'gout': gout,
                     k + 'gdeltas': dreq.grad, k + 'dec_ctr': dec_ctr.detach(), k + 'gdeltas_ctr': dreq2.grad,
                     k + 'deltas_mc': dm, k + 'dec_mc': dec_mc, k + 'dec_plain': plain, k + 'roundtrip': rt})
    save('coder', **arrs)


def unbiased_only():
    """Unbiased-IoU fixtures (sphdet/iou/sph_iou_api.py:103-126 over unbiased_iou_bfov.py / unbiased_iou_rbfov.py): the
    reference on float32 tensors (its own mixed float32/float64 numpy arithmetic) and on float64 tensors (truth)."""
    torch.manual_seed(20231027)
    arrs = {}
    for box in ('bfov', 'rbfov'):
        d = 4 if box == 'bfov' else 5
        g, p = gen(1500, box=box)
        gn, pn = gen(2500, box=box, near=True)
        # structured cases: identical, contained, shared edge planes (integer degrees), poles, seam, wide boxes
        e1 = torch.tensor([[40., 60., 30., 20., 10.], [40., 60., 30., 20., 10.], [10., 90., 20., 20., 0.],
                           [10., 90., 20., 20., 0.], [359., 90., 30., 30., 5.], [180., 2., 40., 40., -30.],
                           [90., 178., 50., 30., 45.], [120., 70., 170., 160., 0.], [200., 45., 1., 1., 0.],
                           [200., 45., 1., 1., 0.], [30., 60., 60., 60., 0.], [300., 120., 90., 10., 80.]])[:, :d]
        e2 = torch.tensor([[40., 60., 30., 20., 10.], [40., 60., 10., 8., 10.], [30., 90., 20., 20., 0.],
                           [10., 110., 20., 20., 0.], [1., 92., 30., 30., -5.], [0., 3., 40., 40., 60.],
                           [270., 177., 50., 30., -45.], [130., 80., 165., 150., 20.], [200.4, 45.3, 1., 1., 0.],
                           [200., 45., 1.5, 0.5, 30.], [30., 60., 20., 90., 0.], [300., 120., 10., 90., -10.]])[:, :d]
        ig = torch.floor(g[:400])
        ip = torch.floor(ig + torch.randint(-6, 7, ig.shape).float())
        ip[:, 0] = ip[:, 0] % 360
        ip[:, 1] = ip[:, 1].clamp(1, 179)
        ip[:, 2:4] = ip[:, 2:4].clamp(1, 170)
        if d == 5:
            ip[:, 4] = ip[:, 4].clamp(-89, 89)
        ig[:, 1] = ig[:, 1].clamp(1, 179)
        ig[:, 2:4] = ig[:, 2:4].clamp(1, 170)
        b1, b2 = torch.cat([g, gn, e1, ig]), torch.cat([p, pn, e2, ip])
        pa, pb = gn[:7].clone(), torch.cat([pn[:5], gn[2:6] + 1.5])
        arrs.update({box + '_b1': b1, box + '_b2': b2,
                     box + '_iou32': R.api.unbiased_iou(b1, b2, is_aligned=True),
                     box + '_iou64': f64(R.api.unbiased_iou, b1, b2, is_aligned=True),
                     box + '_pa': pa, box + '_pb': pb, box + '_pw32': R.api.unbiased_iou(pa, pb),
                     box + '_pw64': f64(R.api.unbiased_iou, pa, pb)})
    save('unbiased', **arrs)


def samples_backends_only():
    """The reference's hard-coded sample pairs (tests/test_all_ious.py:244-261, geo2sph applied) through the other
    backends its own test prints (:271-279): unbiased, Sph-IoU, FoV-IoU (naive needs mmcv.ops.bbox_overlaps: absent)."""
    b1 = torch.tensor([[40, 50, 35, 55], [30, 60, 60, 60], [50, -78, 25, 46], [30, 75, 30, 60], [40, 70, 25, 30],
                       [30, 75, 30, 30], [30, 60, 60, 60]]).float()
    b2 = torch.tensor([[35, 20, 37, 50], [55, 40, 60, 60], [30, -75, 26, 45], [60, 40, 60, 60], [60, 85, 30, 30],
                       [60, 55, 40, 50], [60, 60, 60, 60]]).float()
    b1, b2 = R.box_formator.geo2sph(b1), R.box_formator.geo2sph(b2)
    save('samples7_backends', b1=b1, b2=b2, unbiased=R.api.unbiased_iou(b1, b2, is_aligned=True),
         unbiased64=f64(R.api.unbiased_iou, b1, b2, is_aligned=True), sph=R.api.sph_iou(b1, b2, is_aligned=True),
         fov=R.api.fov_iou(b1, b2, is_aligned=True), unbiased_pw=R.api.unbiased_iou(b1, b2))


def l1_only():
    """Sph2PobL1Loss fixtures (sphdet/losses/sph2pob_l1_loss.py:9-88 on the vendored mmdet L1Loss): the object is built
    without its constructor (which stops in pdb.set_trace(), :24) and its real, decorator-wrapped forward is run."""
    import torch.nn as nn
    torch.manual_seed(20231028)
    cls = R.l1_loss.Sph2PobL1Loss

    def make(encode, swap, modifier, reduction='mean', loss_weight=1.0):
        o = cls.__new__(cls)
        nn.Module.__init__(o)
        o.reduction, o.loss_weight, o.encode, o.swap, o.angle_modifier = reduction, loss_weight, encode, swap, modifier
        return o
    arrs = {}
    for box in ('bfov', 'rbfov'):
        t, p = gen(700, box=box, near=True)
        w = torch.rand(700, t.shape[1])
        w[::5] = 0
        gout = torch.randn(700, 5)
        arrs.update({box + '_pred': p, box + '_target': t, box + '_weight': w, box + '_gout': gout})
        for name, cfg in (('enc', (True, False, 'original')), ('swapmod', (True, True, 'modulus')), ('raw', (False, False, 'original'))):
            o = make(*cfg)
            pr, tr = p.clone().requires_grad_(True), t.clone().requires_grad_(True)
            el = o(pr, tr, reduction_override='none')
            (el * gout).sum().backward()
            k = f'{box}_{name}_'
            arrs.update({k + 'elements': el.detach(), k + 'gpred': pr.grad, k + 'gtarget': tr.grad,
                         k + 'elements64': f64(lambda a, b: o(a, b, reduction_override='none'), p, t),
                         k + 'mean': o(p, t), k + 'mean_w_avg': make(*cfg, loss_weight=2.0)(p, t, w, avg_factor=97.0),
                         k + 'sum_w': o(p, t, w, reduction_override='sum')})
    save('l1', **arrs)


ASSIGN_CFGS = (dict(pos_iou_thr=0.8, neg_iou_thr=0.75, min_pos_iou=0.0),
               dict(pos_iou_thr=0.9, neg_iou_thr=(0.1, 0.8), min_pos_iou=0.75, gt_max_assign_all=False),
               dict(pos_iou_thr=0.8, neg_iou_thr=0.8, match_low_quality=False),
               dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.3),
               dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0, gt_max_assign_all=False))


def assign_only():
    """MaxIoUAssigner fixtures from the reference's real class (mmdet/core/bbox/assigners/max_iou_assigner.py:67-220):
    (a) `assign_wrt_overlaps` on synthetic (k, n) matrices — exact zeros, ties on row and column maxima, ignored (-1)
    columns, an all-zero GT row, an all-ignored matrix — under five threshold / flag configurations (float and tuple
    `neg_iou_thr`, `gt_max_assign_all` and `match_low_quality` both ways); (b) `assign` on spherical boxes with the
    reference's own `sph2pob_standard_iou` as the calculator, with and without `gt_bboxes_ignore` (both
    `ignore_wrt_candidates` settings)."""
    torch.manual_seed(20231101)
    M = R.assigner.MaxIoUAssigner
    arrs = {'n_cfg': len(ASSIGN_CFGS)}
    shapes = [(1, 1), (3, 70), (64, 1000), (17, 4099), (33, 65), (65, 31), (5, 300)]
    arrs['shapes'] = np.asarray(shapes)
    for si, (k, n) in enumerate(shapes):
        ov = torch.rand(k, n)
        ov[ov < 0.7] = 0.0
        ov[:, torch.randint(0, n, (max(1, n // 10),))] = -1.0
        if n > 5:
            ov[0, 3] = ov[0, 5] = ov[0].max()                     # tie on a row maximum
        if k > 2:
            ov[1, :] = torch.where(ov[1] < 0, ov[1], torch.zeros(()))   # a GT that overlaps nothing (all 0 / ignored)
            ov[2, n // 2] = ov[0, n // 2] = 0.95                  # tie on a column maximum: first row wins
        if si == 6:
            ov[:] = -1.0                                          # every column ignored
        labels = torch.randint(0, 37, (k,))
        arrs[f's{si}_ov'], arrs[f's{si}_labels'] = ov, labels
        for ci, cfg in enumerate(ASSIGN_CFGS):
            a = M(iou_calculator=None, **cfg)
            r = a.assign_wrt_overlaps(ov.clone(), labels)
            r0 = a.assign_wrt_overlaps(ov.clone(), None)              # without labels: same indices, labels None
            assert r0.labels is None and torch.equal(r0.gt_inds, r.gt_inds) and torch.equal(r0.max_overlaps, r.max_overlaps)
            arrs[f's{si}_c{ci}_gt_inds'] = r.gt_inds.to(torch.int16)   # (stored narrow: the files stay small)
            arrs[f's{si}_c{ci}_labels'] = r.labels.to(torch.int8)
            if ci == 0:
                arrs[f's{si}_max_overlaps'] = r.max_overlaps           # the same for every configuration
            else:
                assert torch.equal(arrs[f's{si}_max_overlaps'], r.max_overlaps)
    # (b) spherical boxes: anchors scattered around the GTs + far ones
    k, n = 12, 3000
    gt = R.gen.generate_boxes(k, (0, 360), (20, 160), (5, 90), (5, 90), (-90, 90), dtype='float', box='bfov')
    near = gt[torch.randint(0, k, (n // 2,))] + torch.randn(n // 2, 4) * torch.tensor([6., 6., 8., 8.])
    near[:, 0] %= 360
    near[:, 1] = near[:, 1].clamp(1, 179)
    near[:, 2:] = near[:, 2:].clamp(1, 170)
    far = R.gen.generate_boxes(n - n // 2, (0, 360), (0, 180), (1, 100), (1, 100), (-90, 90), dtype='float', box='bfov')
    anchors = torch.cat([near, far])[torch.randperm(n)]
    anchors[7] = gt[3]                                            # an exact copy of a GT
    ignore = R.gen.generate_boxes(4, (0, 360), (30, 150), (30, 90), (30, 90), (-90, 90), dtype='float', box='bfov')
    labels = torch.randint(0, 37, (k,))
    calc = lambda a, b, mode='iou': R.api.sph2pob_standard_iou(a, b, mode=mode)   # noqa: E731
    arrs.update(b_gt=gt, b_anchors=anchors, b_ignore=ignore, b_labels=labels, b_overlaps=calc(gt, anchors))
    for ci, cfg in enumerate(ASSIGN_CFGS):
        for tag, kw, ign in (('plain', {}, None), ('ignc', dict(ignore_iof_thr=0.5), ignore),
                             ('ignb', dict(ignore_iof_thr=0.5, ignore_wrt_candidates=False), ignore)):
            r = M(iou_calculator=calc, **cfg, **kw).assign(anchors, gt, gt_bboxes_ignore=ign, gt_labels=labels)
            arrs[f'b_c{ci}_{tag}_gt_inds'], arrs[f'b_c{ci}_{tag}_labels'] = r.gt_inds.to(torch.int16), r.labels.to(torch.int8)
            if ci == 0:
                arrs[f'b_{tag}_max_overlaps'] = r.max_overlaps
    save('assign', **arrs)


def naive_only():
    """Naive IoU, RBFoV branch (sph_iou_api.py:179-197: Sph2PlanarBoxTransform + box_iou_rotated) with both box formators
    ('sph2pix', 'sph2tan': box_formator.py:76-106, :161-183), aligned.  `box_iou_rotated` is the reference's vendored planar IoU
    (mmcv absent); the BFoV branch needs mmcv.ops.bbox_overlaps and stays unpinned."""
    torch.manual_seed(20231103)
    g, p = gen(1500, box='rbfov', near=True, alpha=(2, 80), beta=(2, 80))
    arrs = dict(b1=g, b2=p)
    for f in ('sph2pix', 'sph2tan'):
        arrs['iou_' + f] = R.api.naive_iou(g, p, is_aligned=True, box_formator=f)
        arrs['iou64_' + f] = f64(R.api.naive_iou, g, p, is_aligned=True, box_formator=f)
    save('naive', **arrs)


def transform_bwd_only():
    """Gradients of the transforms that have no closed-form backward in the kernels — sph2pob_legacy and
    rbb_angle='project' of sph2pob_standard / sph2pob_efficient — from the reference's own torch autograd
    (sphdet/iou/sph2pob_legacy.py:8-31, sph2pob_standard.py:8-80, sph2pob_efficient.py:9-73), float32 and float64."""
    torch.manual_seed(20261004)
    arrs = {}
    cases = [('legacy', R.leg.sph2pob_legacy, 'bfov', {}), ('standard_project', R.std.sph2pob_standard, 'bfov', dict(rbb_angle='project')),
             ('standard_project', R.std.sph2pob_standard, 'rbfov', dict(rbb_angle='project')),
             ('efficient_project', R.eff.sph2pob_efficient, 'bfov', dict(rbb_angle='project')),
             ('efficient_project', R.eff.sph2pob_efficient, 'rbfov', dict(rbb_angle='project'))]
    for name, fn, box, kw in cases:
        for near in (False, True):
            g, p = gen(300, box=box, near=near)
            go1, go2 = torch.randn(300, 5), torch.randn(300, 5)
            for edge in ('arc', 'chord'):
                key = f'{name}_{box}_{"near" if near else "uni"}_{edge}_'

                def run(gg, pp, o1, o2):
                    a, b = gg.clone().requires_grad_(True), pp.clone().requires_grad_(True)
                    q1, q2 = fn(a.clone(), b.clone(), rbb_angle_version='rad', rbb_edge=edge, **kw)   # the reference mutates its inputs (legacy :254): clones, as sph2pob_transform.py:26-27 passes
                    ((q1 * o1).sum() + (q2 * o2).sum()).backward()
                    return q1.detach(), q2.detach(), a.grad, b.grad
                q1, q2, ga, gb = run(g, p, go1, go2)
                torch.set_default_dtype(torch.float64)
                try:
                    _, _, ga64, gb64 = run(g.double(), p.double(), go1.double(), go2.double())
                finally:
                    torch.set_default_dtype(torch.float32)
                arrs.update({key + 'b1': g, key + 'b2': p, key + 'go1': go1, key + 'go2': go2, key + 'p1': q1, key + 'p2': q2,
                             key + 'g1': ga, key + 'g2': gb, key + 'g1_64': ga64, key + 'g2_64': gb64})
    # the jittered form the Sph2PobTransfrom('sph2pob_legacy') decorator runs (sph2pob_transform.py:25-30): clone,
    # jiter_spherical_bboxes, sph2pob_legacy(..., 'rad'), jiter_rotated_bboxes — near-identical pairs included so that the
    # jitters' branches and clamps take part
    for near in (False, True):
        g, p = gen(300, box='bfov', near=near)
        if near:
            p[:60] = g[:60] + torch.randn(60, 4) * 1e-4
        go1, go2 = torch.randn(300, 5), torch.randn(300, 5)
        key = f'legacyjit_bfov_{"near" if near else "uni"}_arc_'

        def runj(gg, pp, o1, o2):
            a, b = gg.clone().requires_grad_(True), pp.clone().requires_grad_(True)
            x, y = R.api.jiter_spherical_bboxes(a.clone(), b.clone())
            q1, q2 = R.leg.sph2pob_legacy(x, y, rbb_angle_version='rad')
            q1, q2 = R.api.jiter_rotated_bboxes(q1, q2)
            ((q1 * o1).sum() + (q2 * o2).sum()).backward()
            return q1.detach(), q2.detach(), a.grad, b.grad
        q1, q2, ga, gb = runj(g, p, go1, go2)
        torch.set_default_dtype(torch.float64)
        try:
            _, _, ga64, gb64 = runj(g.double(), p.double(), go1.double(), go2.double())
        finally:
            torch.set_default_dtype(torch.float32)
        arrs.update({key + 'b1': g, key + 'b2': p, key + 'go1': go1, key + 'go2': go2, key + 'p1': q1, key + 'p2': q2,
                     key + 'g1': ga, key + 'g2': gb, key + 'g1_64': ga64, key + 'g2_64': gb64})
    save('transform_bwd', **arrs)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'transform_bwd':
        transform_bwd_only()
    elif len(sys.argv) > 1 and sys.argv[1] == 'samples_backends':
        samples_backends_only()
    elif len(sys.argv) > 1 and sys.argv[1] == 'l1':
        l1_only()
    elif len(sys.argv) > 1 and sys.argv[1] == 'unbiased':
        unbiased_only()
    elif len(sys.argv) > 1 and sys.argv[1] == 'approx':
        approx_only()
    elif len(sys.argv) > 1 and sys.argv[1] == 'coder':
        coder_only()
    elif len(sys.argv) > 1 and sys.argv[1] == 'assign':
        assign_only()
    elif len(sys.argv) > 1 and sys.argv[1] == 'naive':
        naive_only()
    else:
        main()
        approx_only()
        coder_only()
        unbiased_only()
        l1_only()
        samples_backends_only()
        transform_bwd_only()
        assign_only()
        naive_only()
