"""GPU box: print the worst pairs (fast arithmetic vs f64 oracle) of one adversarial set of tools/stress_compare.py."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
os.environ.setdefault('SPH2POB_STRESS_N', '2000000')
import sph_retina_amd as S
from oracle import oracle as O
import stress_compare as SC
dim, want, variant = int(sys.argv[1]), sys.argv[2], sys.argv[3]
fn = {'standard': S.sph2pob_standard_iou, 'efficient': S.sph2pob_efficient_iou}[variant]
for name, b1, b2 in SC.sets(dim):
    if name != want:
        continue
    t1, t2 = torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda()
    S.set_arithmetic('fast'); fast = fn(t1, t2, is_aligned=True).cpu().numpy()
    S.set_arithmetic('reference'); ref = fn(t1, t2, is_aligned=True).cpu().numpy()
    S.set_arithmetic('fast')
    tru = O.iou_aligned(b1, b2, variant=variant, planar='exact', dtype=np.float64, nthreads=64)
    r32 = O.iou_aligned(b1, b2, variant=variant, planar='mmcv', nthreads=64)
    d = np.abs(fast - tru)
    np.set_printoptions(precision=6, suppress=True, linewidth=200)
    for i in np.argsort(-d)[:6]:
        print('err %.3e fast %.6f reforder %.6f oracle32 %.6f truth %.6f' % (d[i], fast[i], ref[i], r32[i], tru[i]), b1[i], b2[i])
