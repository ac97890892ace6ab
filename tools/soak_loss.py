"""GPU box: differential soak of the fused loss kernels — fast vs reference-order arithmetic on many million nearby
pairs (values), finite gradients; outliers attributed with the f64 oracle.  usage: python tools/soak_loss.py [millions=40]"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import sph_retina_amd as S
from sph_retina_amd.losses import Sph2PobIoULoss
from oracle import oracle as O

millions = int(sys.argv[1]) if len(sys.argv) > 1 else 40
thr = 3e-3
n = 1_000_000
g = torch.Generator(device='cuda').manual_seed(7)
modes = ('iou', 'giou', 'diou', 'ciou')
flag = {m: 0 for m in modes}
nonfinite = 0
kept = []
for it in range(millions):
    dim = 4 if it % 2 == 0 else 5
    mode = modes[(it // 2) % 4]
    u = torch.rand((n, 5), generator=g, device='cuda')
    t = torch.stack([u[:, 0] * 360, u[:, 1] * 180, 1 + u[:, 2] * 99, 1 + u[:, 3] * 99, -90 + u[:, 4] * 180], 1)[:, :dim].contiguous()
    sig = torch.tensor([8., 8., 6., 6., 10.], device='cuda')[:dim] * (0.05 if it % 8 >= 4 else 1.0)   # also near convergence
    p = t + torch.randn((n, dim), generator=g, device='cuda') * sig
    p[:, 0] %= 360
    p[:, 1].clamp_(0.5, 179.5)
    p[:, 2:4].clamp_(1, 170)
    out = {}
    for arith in ('fast', 'reference'):
        S.set_arithmetic(arith)
        pr = p.clone().requires_grad_(True)
        el = Sph2PobIoULoss(mode=mode, reduction='none')(pr, t)
        el.sum().backward()
        nonfinite += int((~torch.isfinite(el)).sum()) + int((~torch.isfinite(pr.grad)).sum())
        out[arith] = el.detach()
    S.set_arithmetic('fast')
    idx = ((out['fast'] - out['reference']).abs() > thr).nonzero().view(-1)
    flag[mode] += int(idx.numel())
    for i in idx[:30].tolist():
        kept.append((mode, p[i].cpu().numpy(), t[i].cpu().numpy(), float(out['fast'][i]), float(out['reference'][i])))
    if it % 10 == 9:
        print('iteration', it + 1, 'flagged', flag, 'non-finite', nonfinite, flush=True)
print('flagged per loss mode (|fast - reference-order| > %g):' % thr, flag, '| non-finite losses / gradients:', nonfinite)
worse = {'fast': 0, 'reference': 0}
for mode, x, y, ff, rr in kept:
    tru = float(O.loss_elements(x[None], y[None], mode=mode, dtype=np.float64)[0])
    ef, er = abs(ff - tru), abs(rr - tru)
    worse['fast' if ef > er else 'reference'] += 1
    if ef > thr:
        print('%s fast %.6f (err %.1e) reforder %.6f (err %.1e) truth %.6f' % (mode, ff, ef, rr, er, tru), x, y)
print('farther from the f64 oracle on the flagged pairs:', worse)
