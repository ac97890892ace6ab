#!/usr/bin/env python3
"""GPU box: per-wave timeline of the dominant kernel from a DIAGNOSTIC build (-DSPH_STAMPS, build/ab/lib_stamps.so):
s_memrealtime (100 MHz) stamps at wave start (0), first slice culled = its data arrived (1), loop done (2), workgroup
barrier passed (3), last in-loop finishing pass done (4), wave done (5).  Prints, relative to the first wave's start,
percentiles of each stamp over the waves of ONE launch in the middle of a back-to-back stream, and the launch period.
usage: python tools/stamp_timeline.py [pairs] [ENV=val ...]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for kv in sys.argv[2:]:
    k, v = kv.split('=')
    os.environ[k] = v
import torch  # noqa: E402
from bench import make_boxes  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, 'build', 'ab', 'lib_stamps.so'))
fn = lib.sph2pob_iou_aligned_f32
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
lib.sph2pob_debug_set_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda', 0)
b1, b2 = make_boxes(pairs, 0, dev), make_boxes(pairs, 1, dev)
out = torch.empty(pairs, device=dev)
nw = 7 * 256 * 4
stamps = torch.zeros((nw, 8), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def launch():
    assert fn(b1.data_ptr(), b2.data_ptr(), out.data_ptr(), pairs, 4, 0, 0, 0, 0, st) == 0


for _ in range(3000):
    launch()
torch.cuda.synchronize()
assert lib.sph2pob_debug_set_stamps(stamps.data_ptr()) == 0
rows = []
for rep in range(5):
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64)
    live = s[:, 0] > 0
    s = s[live] * 10.0   # ns
    t0 = s[:, 0].min()
    rows.append(s)
    rel = lambda c: s[:, c][s[:, c] > 0] - t0  # noqa: E731
    pct = lambda a: ' '.join('%7.2f' % (np.percentile(a, q) / 1e3) for q in (0, 10, 50, 90, 100))  # noqa: E731
    print(f'rep {rep}: waves {int(live.sum())}  (us from the first wave start; percentiles 0 10 50 90 100)')
    for c, name in ((0, 'wave start'), (1, 'first slice culled'), (4, 'in-loop pass done'), (2, 'loop done'), (3, 'barrier passed'), (5, 'wave done')):
        a = rel(c)
        print(f'   {name:20s} n={a.size:5d}  {pct(a)}')
    d = s[:, 5] - s[:, 0]
    print(f'   wave lifetime        n={d.size:5d}  {pct(d)}    kernel span {(s[:, 5].max() - t0) / 1e3:.2f} us')
