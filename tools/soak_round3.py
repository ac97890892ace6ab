#!/usr/bin/env python3
"""GPU box: differential soak of round 3's fused routes against the routes they replace, on random shapes.

  * fused assigner (sph2pob_iou_assign_f32, no k x n matrix) vs pairwise kernel + matrix epilogue: AssignResult and extras bit for bit,
    random k in [1, 300], n in [1, 60 000] (weighted towards tile / chunk boundaries), both variants, both box kinds, the four
    threshold configurations of the tests, with / without an ignore mask;
  * host-free batched NMS (sph2pob_batched_nms_f32) vs the torch-sorted route: keep lists and dets equal, random K in [1, 16 384],
    1 ... 200 classes, score ties, both box kinds.
usage: python tools/soak_round3.py [seconds=240]"""
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import sph_retina_amd as S                                               # noqa: E402
import sph_retina_amd.bbox.assigners as A                                # noqa: E402
import sph_retina_amd.bbox.nms as N                                      # noqa: E402
from test_gpu_assigner import CFGS, _matrix_route, _same, _scene         # noqa: E402
from test_gpu_nms import _general_route                                  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(20260)
t_end = time.time() + budget
n_assign = n_nms = 0
bad = []
while time.time() < t_end:
    # ---- assigner ----
    k = int(rng.choice([1, 2, 3, 7, 8, 9, 21, 22, 23, 63, 64, 65, 128, 200, 300, int(rng.integers(1, 301))]))
    base = int(rng.choice([1, 64, 255, 256, 257, 511, 512, 513, 1024, 4096, 8192, 32768, int(rng.integers(1, 60001))]))
    n = max(1, base + int(rng.integers(-2, 3)))
    dim = int(rng.choice([4, 5]))
    variant = str(rng.choice(['standard', 'efficient']))
    gt, boxes, labels = _scene(k, n, dim, int(rng.integers(0, 2 ** 31)))
    ignore = None
    if rng.random() < 0.4:
        ignore = (torch.rand(n, generator=torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))) < rng.random() * 0.5).cuda()
    kw = CFGS[int(rng.integers(0, len(CFGS)))]
    try:
        ov, res, ex = _matrix_route(A, S, gt, boxes, labels, variant, ignore, **kw)
        res2, ex2 = A.fused_assign(gt, boxes, labels, variant, ignore_mask=ignore, return_extras=True, **kw)
        _same(res, ex, res2, ex2)
    except AssertionError as e:
        bad.append(('assign', k, n, dim, variant, ignore is not None, kw, str(e)[:200]))
        print('MISMATCH', bad[-1], flush=True)
    n_assign += 1
    # ---- NMS ----
    kk = int(rng.choice([1, 2, 63, 64, 65, 2047, 2048, 2049, 4096, 4097, 6144, 8192, 8193, 12288, 16384, int(rng.integers(1, 16385))]))
    dim = int(rng.choice([4, 5]))
    ncls = int(rng.choice([1, 2, 37, 80, 200]))
    centres = torch.rand((max(kk // 16, 1), 5)) * torch.tensor([360., 150., 55., 55., 120.]) + torch.tensor([0., 15., 5., 5., -60.])
    b = centres[torch.randint(0, centres.size(0), (kk,))] + torch.randn((kk, 5)) * 2.0
    b[:, 0] %= 360
    b[:, 1] = b[:, 1].clamp(1, 179)
    b[:, 2:4] = b[:, 2:4].clamp(2, 120)
    b = b[:, :dim].contiguous().cuda()
    s = torch.rand(kk)
    if rng.random() < 0.5:
        s[torch.randint(0, kk, (max(kk // 8, 1),))] = 0.5
    s = s.cuda()
    idxs = torch.randint(0, ncls, (kk,)).cuda()
    cfg = dict(iou_threshold=float(rng.choice([0.3, 0.5, 0.7])))
    if rng.random() < 0.5:
        cfg['max_num'] = int(rng.choice([1, 100, 1000]))
    try:
        dets, keep = N.sph_batched_nms(b, s, idxs, cfg, 'efficient')
        gdets, gkeep = _general_route(N, b, s, idxs, cfg)
        assert torch.equal(keep, gkeep) and torch.equal(dets, gdets)
    except AssertionError as e:
        bad.append(('nms', kk, dim, ncls, cfg, str(e)[:200]))
        print('MISMATCH', bad[-1], flush=True)
    n_nms += 1
    if (n_assign % 50) == 0:
        print(f'{n_assign} assignments, {n_nms} NMS calls, {len(bad)} mismatches', flush=True)
print(f'done: {n_assign} assignments, {n_nms} NMS calls, {len(bad)} mismatches')
sys.exit(1 if bad else 0)
