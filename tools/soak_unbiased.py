"""GPU box: Unbiased-IoU kernel against the C restatement (same precision model) on many million random pairs."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from sph_retina_amd.iou import unbiased_iou
from oracle import oracle as O
millions = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1_000_000
tot = {4: [0, 0, 0.0], 5: [0, 0, 0.0]}
for it in range(millions):
    dim = 4 if it % 2 == 0 else 5
    box = 'bfov' if dim == 4 else 'rbfov'
    a = O.generate_boxes(n, 1000 + it, box=box)
    if it % 4 < 2:
        b = O.generate_boxes(n, 5000 + it, box=box)
    else:
        b = a + np.random.default_rng(it).normal(0, 1, a.shape).astype(np.float32) * np.array([8, 8, 6, 6, 10], np.float32)[:dim]
        b[:, 0] %= 360; b[:, 1] = b[:, 1].clip(0.5, 179.5); b[:, 2:4] = b[:, 2:4].clip(1, 170)
    k = unbiased_iou(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), is_aligned=True).cpu().numpy()
    o = O.unbiased_iou(a, b, prec='kernel')
    d = np.abs(k - o)
    tot[dim][0] += int((d > 1e-6).sum()); tot[dim][1] += int((d > 1e-3).sum()); tot[dim][2] = max(tot[dim][2], float(d.max()))
    assert np.isfinite(k).all() and (k >= 0).all() and (k <= 1).all()
    if it % 5 == 4:
        print('iteration', it + 1, tot, flush=True)
print('pairs with |kernel - restatement| > 1e-6 / > 1e-3 / max, per box type over %d M pairs each:' % (millions // 2), tot)
