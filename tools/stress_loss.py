"""GPU box: adversarial check of the loss kernels — values vs the f64 oracle in both arithmetic modes, gradients finite."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))
import sph_retina_amd as S  # noqa: E402
from sph_retina_amd.losses import Sph2PobIoULoss  # noqa: E402
from oracle import oracle as O  # noqa: E402
os.environ.setdefault('SPH2POB_STRESS_N', '50000')
import stress_compare as SC  # noqa: E402


def run():
    """Returns the number of (set, mode, arithmetic) combinations with a non-finite loss or gradient."""
    bad = 0
    for dim in (4, 5):
        for name, b1, b2 in SC.sets(dim):
            for mode in ('iou', 'giou', 'diou', 'ciou'):
                tru = O.loss_elements(b1, b2, mode=mode, dtype=np.float64, nthreads=32)
                row = f'dim{dim} {name:15s} {mode:5s}'
                for arith in ('fast', 'reference'):
                    S.set_arithmetic(arith)
                    p = torch.from_numpy(b1).cuda().requires_grad_(True)
                    t = torch.from_numpy(b2).cuda().requires_grad_(True)
                    el = Sph2PobIoULoss(mode=mode, reduction='none')(p, t)
                    el.sum().backward()
                    v = el.detach().cpu().numpy()
                    ok = np.isfinite(tru)
                    d = np.abs(v - tru)[ok]
                    fin = bool(torch.isfinite(p.grad).all() and torch.isfinite(t.grad).all() and np.isfinite(v).all())
                    gmax = float(max(p.grad.abs().max(), t.grad.abs().max()))
                    row += f' | {arith}: mean {d.mean():.1e} p99.9 {np.quantile(d, 0.999):.1e} max {d.max():.1e} gradmax {gmax:.1e}{"" if fin else " NONFINITE!"}'
                    bad += 0 if fin else 1
                S.set_arithmetic('fast')
                print(row, flush=True)
    return bad


if __name__ == '__main__':
    sys.exit(1 if run() else 0)
