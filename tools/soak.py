"""GPU box: large differential soak — closed-form ('fast', 'robust') against reference-order arithmetic on many million
random pairs generated on the device; every pair on which two modes differ by more than `thr` is re-evaluated with the
f64 oracle on the CPU and attributed.  usage: python tools/soak.py [millions=100] [thr=3e-3]"""
import os, sys, json
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import sph_retina_amd as S
from oracle import oracle as O

millions = int(sys.argv[1]) if len(sys.argv) > 1 else 100
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-3
n = 1_000_000
g = torch.Generator(device='cuda').manual_seed(2024)
fn = {'standard': S.sph2pob_standard_iou, 'efficient': S.sph2pob_efficient_iou}
kept = []
counts = {}
for it in range(millions):
    dim = 4 if it % 2 == 0 else 5
    u = torch.rand((n, 5), generator=g, device='cuda')
    a = torch.stack([u[:, 0] * 360, u[:, 1] * 180, 1 + u[:, 2] * 99, 1 + u[:, 3] * 99, -90 + u[:, 4] * 180], 1)[:, :dim].contiguous()
    kind = it % 4
    if kind < 2:      # independent second draw (benchmark distribution)
        v = torch.rand((n, 5), generator=g, device='cuda')
        b = torch.stack([v[:, 0] * 360, v[:, 1] * 180, 1 + v[:, 2] * 99, 1 + v[:, 3] * 99, -90 + v[:, 4] * 180], 1)[:, :dim].contiguous()
    else:             # detector-like nearby pairs
        sig = torch.tensor([8., 8., 6., 6., 10.], device='cuda')[:dim]
        b = a + torch.randn((n, dim), generator=g, device='cuda') * sig
        b[:, 0] %= 360
        b[:, 1].clamp_(0.5, 179.5)
        b[:, 2:4].clamp_(1, 170)
    for vname, f in fn.items():
        res = {}
        for mode in ('fast', 'robust', 'reference'):
            S.set_arithmetic(mode)
            res[mode] = f(a, b, is_aligned=True)
        S.set_arithmetic('fast')
        bad = ((res['fast'] - res['reference']).abs() > thr) | ((res['robust'] - res['reference']).abs() > thr) | \
              ~torch.isfinite(res['fast']) | ~torch.isfinite(res['robust'])
        idx = bad.nonzero().view(-1)
        key = (dim, 'uniform' if kind < 2 else 'nearby', vname)
        counts[key] = counts.get(key, 0) + int(idx.numel())
        if idx.numel():
            for i in idx[:50].tolist():
                kept.append((dim, kind < 2, vname, a[i].cpu().numpy(), b[i].cpu().numpy(), float(res['fast'][i]), float(res['robust'][i]), float(res['reference'][i])))
    if it % 10 == 9:
        print('iteration', it + 1, 'of', millions, 'flagged so far', sum(counts.values()), flush=True)
print('pairs per (dim, dist, variant):', millions // 4, 'M; flagged (|mode - reference-order| >', thr, '):', {str(k): v for k, v in counts.items()})
summary = {'fast_worse': 0, 'robust_worse': 0, 'reference_worse': 0}
for dim, uni, vname, x, y, ff, rr, ref in kept:
    tru = float(O.iou_aligned(x[None], y[None], variant=vname, planar='exact', dtype=np.float64)[0])
    ef, er, eo = abs(ff - tru), abs(rr - tru), abs(ref - tru)
    worst = max((ef, 'fast_worse'), (er, 'robust_worse'), (eo, 'reference_worse'))[1]
    summary[worst] += 1
    if max(ef, er) > thr:
        print('dim%d %s %s fast %.6f (err %.1e) robust %.6f (err %.1e) reforder %.6f (err %.1e) truth %.6f' % (dim, 'uniform' if uni else 'nearby', vname, ff, ef, rr, er, ref, eo, tru), x, y)
print('which mode is farthest from the f64 truth on the flagged pairs:', summary)
