"""GPU box: print |dIoU| statistics of the HIP kernels against the CPU oracle (f32 reference arithmetic and
f64 exact truth) for every variant / box type / distribution.  Test infrastructure (uses oracle/)."""
import json
import sys
import os

import numpy as np
import torch  # noqa: F401  (initialises the HIP runtime before the product is imported)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from oracle import oracle as O  # noqa: E402
import sph_retina_amd as S  # noqa: E402
from conftest import err_stats  # noqa: E402
from test_gpu_iou_parity import hip_iou, nearby  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rows = []
for box in ('bfov', 'rbfov'):
    for dist in ('uniform', 'nearby'):
        b1 = O.generate_boxes(n, 0, box=box)
        b2 = O.generate_boxes(n, 1, box=box) if dist == 'uniform' else nearby(b1, 7)
        for v in ('standard', 'efficient', 'legacy'):
            if v == 'legacy' and box == 'rbfov':
                continue
            got = hip_iou(S, v, b1, b2)
            ref = O.iou_aligned(b1, b2, variant=v, planar='mmcv', nthreads=64)
            tru = O.iou_aligned(b1, b2, variant=v, planar='exact', dtype=np.float64, nthreads=64)
            ok = np.isfinite(ref) & np.isfinite(tru)
            r = dict(box=box, dist=dist, variant=v, n=int(ok.sum()),
                     hip_vs_ref32=err_stats(got[ok], ref[ok]), hip_vs_truth=err_stats(got[ok], tru[ok]),
                     ref32_vs_truth=err_stats(ref[ok], tru[ok]))
            rows.append(r)
            print(json.dumps(r), flush=True)
