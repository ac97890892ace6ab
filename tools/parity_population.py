#!/usr/bin/env python3
"""GPU box: whole-population parity of the default arithmetic at BASELINE's full sizes — all 8 M pairs of configs[4] and all
25 M pairs of the 64 x 392 832 assigner matrix — against the oracle's two instantiations, with every pair beyond 1e-4 listed
(index, boxes, the four IoUs, the planar angle difference the rotated jitter decides on).  The fixed-count bounds of
tests/test_gpu_iou_parity.py come from this output (profiles/r05*_parity_population.log)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def stats(got, want):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    return dict(max=float(d.max()), mean=float(d.mean()), n5=int((d > 1e-5).sum()), n4=int((d > 1e-4).sum()))


def explain(O, S, g, p, idx_label):
    """One outlier pair: the reference-order kernel, both oracles, and what the rotated jitter sees."""
    g1, p1 = g[None].astype(np.float32), p[None].astype(np.float32)
    t1, t2 = torch.from_numpy(g1).cuda(), torch.from_numpy(p1).cuda()
    S.set_arithmetic('fast')
    fast = float(S.sph2pob_standard_iou(t1, t2, is_aligned=True)[0])
    S.set_arithmetic('reference')
    refk = float(S.sph2pob_standard_iou(t1, t2, is_aligned=True)[0])
    S.set_arithmetic('fast')
    r32 = float(O.iou_aligned(g1, p1, variant='standard', planar='mmcv')[0])
    f64 = float(O.iou_aligned(g1, p1, variant='standard', planar='exact', dtype=np.float64)[0])
    a32 = O.transform(g1, p1, variant='standard', jitter=False)
    a64 = O.transform(g1, p1, variant='standard', jitter=False, dtype=np.float64)
    da32, da64 = float(a32[0][0, 4] - a32[1][0, 4]), float(a64[0][0, 4] - a64[1][0, 4])
    arc = float(a64[1][0, 0] - a64[0][0, 0])
    print(f'   {idx_label}: g={g.tolist()} p={p.tolist()}\n      hip fast {fast:.7f}  hip reference-order {refk:.7f}  ref32 {r32:.7f}  f64 {f64:.7f}'
          f'\n      planar angle difference a_g - a_p: fp32 {da32:+.7e}  f64 {da64:+.7e}  (rotated jitter threshold 1.2345678e-3; |.| within 2e-6 of it = a flip)'
          f'  centre distance {arc:.5f} rad', flush=True)


def main():
    from oracle import oracle as O
    O.build()
    import sph_retina_amd as S
    from bench_configs import retina_anchors
    S.set_arithmetic('fast')
    n = 8_000_000
    g = torch.Generator(device='cpu').manual_seed(4)
    u = torch.rand((2, n, 4), generator=g)
    mk = lambda v: torch.stack([v[:, 0] * 360, v[:, 1] * 180, v[:, 2] * 99 + 1, v[:, 3] * 99 + 1], 1)  # noqa: E731
    h1, h2 = mk(u[0]).numpy(), mk(u[1]).numpy()
    iou = S.sph2pob_standard_iou(torch.from_numpy(h1).cuda(), torch.from_numpy(h2).cuda(), is_aligned=True).cpu().numpy()
    ref = O.iou_aligned(h1, h2, variant='standard', planar='mmcv', nthreads=64)
    tru = O.iou_aligned(h1, h2, variant='standard', planar='exact', dtype=np.float64, nthreads=64)
    print('8M uniform BFoV standard: vs ref32', stats(iou, ref), 'vs f64', stats(iou, tru), 'ref32 vs f64', stats(ref, tru), flush=True)
    for i in np.nonzero((np.abs(iou - ref) > 1e-4) | (np.abs(iou - tru) > 1e-4))[0]:
        explain(O, S, h1[i], h2[i], f'pair {i}')
    anchors = retina_anchors(1024, 2048)
    gg = torch.Generator().manual_seed(0)
    uu = torch.rand((64, 4), generator=gg)
    gt = torch.stack([uu[:, 0] * 360, 20 + uu[:, 1] * 140, 5 + uu[:, 2] * 85, 5 + uu[:, 3] * 85], 1)
    for variant in ('standard', 'efficient'):
        fn = S.sph2pob_standard_iou if variant == 'standard' else S.sph2pob_efficient_iou
        ov = fn(gt.cuda(), anchors).cpu().numpy()
        a = anchors.cpu().numpy()
        ref = O.iou_pairwise(gt.numpy(), a, variant=variant, planar='mmcv', nthreads=64)
        tru = O.iou_pairwise(gt.numpy(), a, variant=variant, planar='exact', dtype=np.float64, nthreads=64)
        print(f'64 x 392832 anchors {variant}: vs ref32', stats(ov, ref), 'vs f64', stats(ov, tru), 'ref32 vs f64', stats(ref, tru),
              'zero agreement', int(((ov == 0) != (ref == 0)).sum()), flush=True)
        if variant == 'standard':
            for r, c in zip(*np.nonzero((np.abs(ov - ref) > 1e-4) | (np.abs(ov - tru) > 1e-4))):
                explain(O, S, gt.numpy()[r], a[c], f'GT {r} x anchor {c}')


if __name__ == '__main__':
    main()
