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
nw = max(7 * 256 * 4, (pairs + 127) // 128 + 64)
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
    for c, name in ((0, 'wave start'), (1, 'first slice culled'),) + (() if os.environ.get('SPH2POB_CHUNK_SLICES') else ((4, 'in-loop pass done'),)) + ( (2, 'loop done'), (3, 'barrier passed'), (5, 'wave done')):
        a = rel(c)
        if a.size == 0:
            continue
        print(f'   {name:20s} n={a.size:5d}  {pct(a)}')
    d = s[:, 5] - s[:, 0]
    print(f'   wave lifetime        n={d.size:5d}  {pct(d)}    kernel span {(s[:, 5].max() - t0) / 1e3:.2f} us')

# ---- where the tail comes from: the last repetition's launch, by CU, by SIMD, by role of the wave ----
s = rows[-1]
raw = stamps.cpu().numpy()[stamps.cpu().numpy()[:, 0] > 0]
hw = raw[:, 6] & 0xffffffff
xcc = (raw[:, 6] >> 32) & 15
cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)
simd = (hw >> 4) & 3
t0 = s[:, 0].min()
done = (s[:, 5] - t0) / 1e3
loop = (s[:, 2] - t0) / 1e3
barrier = (s[:, 3] - t0) / 1e3
first = (s[:, 1] - t0) / 1e3
nwaves = s.shape[0]
wave_in_wg = np.arange(nwaves) % 4
left = raw[:, 7]
cus = np.unique(cu)
print(f'distinct CUs {cus.size}; waves per CU min / max {min((cu == c).sum() for c in cus)} / {max((cu == c).sum() for c in cus)}')
cu_done = np.array([done[cu == c].max() for c in cus])
cu_loop = np.array([loop[cu == c].max() for c in cus])
print('last wave done per CU (us): percentiles 0 10 50 90 100 ' + ' '.join('%.2f' % np.percentile(cu_done, q) for q in (0, 10, 50, 90, 100)))
print('last loop done per CU (us):                             ' + ' '.join('%.2f' % np.percentile(cu_loop, q) for q in (0, 10, 50, 90, 100)))
key = cu * 4 + simd
ks = np.unique(key)
simd_done = np.array([done[key == k].max() for k in ks])
simd_n = np.array([(key == k).sum() for k in ks])
print(f'SIMDs {ks.size}; waves per SIMD min / max {simd_n.min()} / {simd_n.max()}; last wave done per SIMD: ' + ' '.join('%.2f' % np.percentile(simd_done, q) for q in (0, 10, 50, 90, 100)))
for w in range(4):
    m = wave_in_wg == w
    print(f'wave {w} of its workgroup: loop done median {np.median(loop[m]):.2f}  barrier {np.median(barrier[m]):.2f}  done median {np.median(done[m]):.2f}  90th {np.percentile(done[m], 90):.2f}  max {done[m].max():.2f}  leftover mean {left[m].mean():.1f}')
nsl = (pairs + 63) // 64
three = np.arange(nwaves) < (nsl - 2 * nwaves) if nsl > 2 * nwaves else np.zeros(nwaves, bool)
for name, m in (('waves with the extra slice', three), ('waves without', ~three)):
    if m.any():
        print(f'{name}: n {m.sum()}  first data {np.median(first[m]):.2f}  loop done median {np.median(loop[m]):.2f} 90th {np.percentile(loop[m], 90):.2f}  done median {np.median(done[m]):.2f} 90th {np.percentile(done[m], 90):.2f} max {done[m].max():.2f}')
n4 = nwaves // 4 * 4
wg_barrier = barrier[:n4].reshape(-1, 4).max(1)
wg_done = done[:n4].reshape(-1, 4).max(1)
print('workgroup barrier release (us): ' + ' '.join('%.2f' % np.percentile(wg_barrier, q) for q in (0, 10, 50, 90, 100)) + '   workgroup done: ' + ' '.join('%.2f' % np.percentile(wg_done, q) for q in (0, 10, 50, 90, 100)))
print('pass length by leftover count: ' + '  '.join('%d-%d: %.2f' % (a, b, np.median((done - barrier)[(left >= a) & (left < b)])) for a, b in ((1, 40), (40, 50), (50, 60), (60, 65), (65, 129)) if ((left >= a) & (left < b)).any()))
print('tail pass length (done - barrier) of the waves that ran one: ' + ' '.join('%.2f' % np.percentile((done - barrier)[(done - barrier) > 0.2], q) for q in (0, 10, 50, 90, 100)))
late = np.argsort(done)[-12:]
for i in late:
    print(f'  late wave {i}: wg {i // 4} wave {i % 4} cu {cu[i]} simd {simd[i]} first {first[i]:.2f} loop {loop[i]:.2f} barrier {barrier[i]:.2f} done {done[i]:.2f} leftover {left[i]}')

if os.environ.get('SPH2POB_CHUNK_SLICES'):
    life_ticks = raw[:, 5] - raw[:, 0]   # 100 MHz ticks
    clk = raw[:, 4] / np.maximum(life_ticks, 1) * 0.1   # GHz
    print('in-kernel shader clock (GHz) per wave, delta s_memtime / delta s_memrealtime: ' + ' '.join('%.3f' % np.percentile(clk, q) for q in (0, 10, 50, 90, 100)))
