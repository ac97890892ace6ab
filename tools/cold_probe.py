#!/usr/bin/env python3
"""GPU box: why the cold-HBM rotation of bench.py reads 11 us on most boxes and ~40 us on some.  Times the rotation over
10 buffer sets (360 MB) laid out three ways — 30 separate allocations (what bench.py does), views into ONE 360 MB
allocation, and the same views after a long run — and prints where the buffers sit (address mod 2 MiB)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_boxes
from sph_retina_amd import _lib
lib = _lib.lib()
dev = torch.device('cuda', 0)
n = 1_000_000
st = torch.cuda.current_stream().cuda_stream


def run(sets, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    k = len(sets)
    for r in range(reps):
        a, b, o = sets[r % k]
        lib.sph2pob_iou_aligned_f32(a.data_ptr(), b.data_ptr(), o.data_ptr(), n, 4, 0, 0, 0, 0, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


sep = [(make_boxes(n, 100 + 2 * k, dev), make_boxes(n, 101 + 2 * k, dev), torch.empty(n, device=dev)) for k in range(10)]
big = torch.empty(10 * 9 * n, device=dev)
one = []
for k in range(10):
    o = k * 9 * n
    a, b, c = big[o:o + 4 * n].view(n, 4), big[o + 4 * n:o + 8 * n].view(n, 4), big[o + 8 * n:o + 9 * n]
    a.copy_(sep[k][0]); b.copy_(sep[k][1])
    one.append((a, b, c))
print('separate allocations: address mod 2 MiB (KiB):', sorted({t.data_ptr() % (2 << 20) >> 10 for s in sep for t in s}))
print('one allocation: base mod 2 MiB (KiB):', big.data_ptr() % (2 << 20) >> 10)
run(sep[:1], 3000)
print(f'same buffers           {run(sep[:1], 2000):7.2f} us')
for name, sets in (('30 allocations', sep), ('one allocation', one), ('30 allocations', sep), ('one allocation', one)):
    run(sets, 500)
    print(f'rotation, {name:15s} ' + ' '.join(f'{run(sets, 2000):7.2f}' for _ in range(3)) + ' us')
run(sep, 30000)
print('after 30 000 more launches of the rotation:')
for name, sets in (('30 allocations', sep), ('one allocation', one)):
    print(f'rotation, {name:15s} ' + ' '.join(f'{run(sets, 2000):7.2f}' for _ in range(3)) + ' us')
print('memory:', torch.cuda.mem_get_info(dev))
