"""GPU box: adversarial differential test — fast core vs reference-order kernels vs the f64 oracle on distributions that
exercise the rare branches (poles, seam, tiny / huge boxes, |gamma| near 180 / 360, identical and near-identical pairs,
integer degrees).  Prints per-set statistics; exits non-zero on a gross discrepancy."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tests'))
import sph_retina_amd as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(12345)
n = int(os.environ.get('SPH2POB_STRESS_N', 200_000))


def clampb(b):
    b[:, 0] %= 360
    b[:, 1] = b[:, 1].clip(0, 180)
    b[:, 2:4] = b[:, 2:4].clip(0.01, 179.9)
    return b.astype(np.float32)


def sets(dim):
    base = O.generate_boxes(n, 1, box='rbfov' if dim == 5 else 'bfov', gamma=(-180, 180))
    d = lambda s: rng.standard_normal(base.shape).astype(np.float32) * s  # noqa: E731
    yield 'poles', clampb(np.concatenate([base[:, :1], rng.choice([0.0, 0.5, 179.5, 180.0, 3.0], n)[:, None], base[:, 2:]], 1)), \
        clampb(np.concatenate([base[:, :1] + 90, rng.choice([0.0, 1.0, 179.0, 180.0, 2.0], n)[:, None], base[:, 2:]], 1).astype(np.float32))
    s1 = base.copy(); s1[:, 0] = rng.choice([0.0, 0.001, 359.999, 1.0, 359.0], n)
    s2 = s1 + d(2.0); s2[:, 0] = rng.choice([359.9995, 0.0, 0.5, 358.0, 2.0], n)
    yield 'seam', clampb(s1), clampb(s2)
    t1 = base.copy(); t1[:, 2:4] = rng.choice([0.01, 0.05, 0.3, 1.0], (n, 2))
    yield 'tiny', clampb(t1), clampb(t1 + d(0.2))
    h1 = base.copy(); h1[:, 2:4] = rng.choice([120.0, 150.0, 179.0, 179.9], (n, 2))
    yield 'huge', clampb(h1), clampb(h1 + d(20.0))
    yield 'identical', clampb(base.copy()), clampb(base.copy())
    yield 'near-identical', clampb(base.copy()), clampb(base + d(0.01))
    i1 = np.round(base); i2 = i1 + rng.integers(-3, 4, base.shape)
    yield 'integer', clampb(i1), clampb(i2)
    if dim == 5:
        g1 = base.copy(); g1[:, 4] = rng.choice([-360.0, -270.0, -180.0, -179.9, 179.9, 180.0, 270.0, 360.0, 90.0, -90.0], n)
        g2 = g1 + d(5.0); g2[:, 4] = rng.choice([-360.0, -181.0, 179.0, 181.0, 359.0, 0.0], n)
        yield 'gamma-extreme', clampb(g1), clampb(g2)


def run():
  bad = 0
  rows = []
  for dim in (4, 5):
      for name, b1, b2 in sets(dim):
          t1, t2 = torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda()
          for v, fn in (('standard', S.sph2pob_standard_iou), ('efficient', S.sph2pob_efficient_iou)):
              S.set_arithmetic('fast')
              fast = fn(t1, t2, is_aligned=True).cpu().numpy()
              S.set_arithmetic('reference')
              ref = fn(t1, t2, is_aligned=True).cpu().numpy()
              S.set_arithmetic('fast')
              tru = O.iou_aligned(b1, b2, variant=v, planar='exact', dtype=np.float64, nthreads=32)
              ok = np.isfinite(tru)
              df, dr, dd = np.abs(fast - tru)[ok], np.abs(ref - tru)[ok], np.abs(fast - ref)[ok]
              flag = ''
              if not (np.isfinite(fast).all() and fast.min() >= 0 and fast.max() <= 1):
                  flag = ' RANGE!'
                  bad += 1
              if np.quantile(df, 0.999) > 10 * max(np.quantile(dr, 0.999), 1e-4):
                  flag += ' WORSE-THAN-REFERENCE-ORDER!'
                  bad += 1
              rows.append((dim, name, v, df, dr, dd))
              print(f'dim{dim} {name:15s} {v:9s} fast-truth mean {df.mean():.1e} p99.9 {np.quantile(df, 0.999):.1e} max {df.max():.1e} | '
                    f'reforder-truth mean {dr.mean():.1e} p99.9 {np.quantile(dr, 0.999):.1e} max {dr.max():.1e} | fast-reforder max {dd.max():.1e}{flag}',
                    flush=True)
  return bad, rows


if __name__ == '__main__':
    bad, _ = run()
    sys.exit(1 if bad else 0)
