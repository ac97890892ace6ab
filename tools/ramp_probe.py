#!/usr/bin/env python3
"""How the headline kernel's time per launch depends on how many launches follow a device synchronisation.

The bench contract brackets K steps by a barrier + synchronize; the driver uses K = 20, the builder's default is 5 000, and
the two figures differ (8.1 vs 7.2 us per step).  This probe separates the per-launch time from what a burst pays once:

    python tools/ramp_probe.py run                # event-timed bursts of 20 ... 5000 launches, least-squares fixed + slope
    rocprofv3 --kernel-trace -d DIR -- python3 tools/ramp_probe.py bursts    # 12 bursts of 200 launches after a sync
    python tools/ramp_probe.py read DIR           # duration and gap by position inside the burst
"""
import csv
import ctypes
import glob
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def setup(n=1_000_000):
    import torch
    from sph_retina_amd import _lib, _torch_glue as G
    import bench
    dev = torch.device('cuda', 0)
    b1, b2 = bench.make_boxes(n, 0, dev), bench.make_boxes(n, 1, dev)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    lib = _lib.lib()
    stream = torch.cuda.current_stream(dev)
    G.set_arithmetic('fast')
    sp = ctypes.c_void_p(stream.cuda_stream)
    p1, p2, po, cn = G.ptr(b1), G.ptr(b2), G.ptr(out), ctypes.c_int64(n)

    def launch():
        lib.sph2pob_iou_aligned_f32(p1, p2, po, cn, 4, 0, 0, 0, 0, sp)
    return torch, dev, stream, launch, (b1, b2, out)


def run():
    torch, dev, stream, launch, keep = setup()
    for _ in range(3000):
        launch()
    torch.cuda.synchronize(dev)
    rows = []
    for reps in (20, 50, 100, 200, 500, 1000, 2000, 5000):
        ev_t, wall_t = [], []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(reps):
                launch()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            wall_t.append((time.perf_counter() - t0) * 1e6)
            ev_t.append(e0.elapsed_time(e1) * 1e3)
        ev, wall = statistics.median(ev_t), statistics.median(wall_t)
        rows.append((reps, ev, wall))
        print(f'burst of {reps:5d}: events {ev:9.1f} us = {ev / reps:6.3f} per launch   wall {wall:9.1f} us = {wall / reps:6.3f} per launch', flush=True)
    # host-only launch rate: how fast the launching thread can submit
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(2000):
        launch()
    sub = (time.perf_counter() - t0) * 1e6 / 2000
    torch.cuda.synchronize(dev)
    print(f'host submission: {sub:.3f} us per launch (2000 launches, before the sync)')
    for a, b in ((0, 3), (3, 7)):
        (r0, e0, w0), (r1, e1, w1) = rows[a], rows[b]
        k = (e1 - e0) / (r1 - r0)
        kw = (w1 - w0) / (r1 - r0)
        print(f'bursts {r0} -> {r1}: events slope {k:.3f} us per launch, fixed {e0 - k * r0:.1f} us; wall slope {kw:.3f}, fixed {w0 - kw * r0:.1f} us')


def bracket():
    """the contract's bracket at the driver's K = 20: synchronize, K launches, synchronize — where its fixed cost sits"""
    torch, dev, stream, launch, keep = setup()
    for _ in range(3000):
        launch()
    torch.cuda.synchronize(dev)
    for K in (20, 100):
        sub, tot = [], []
        for _ in range(101):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(K):
                launch()
            t1 = time.perf_counter()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            sub.append((t1 - t0) * 1e6)
            tot.append((t2 - t0) * 1e6)
        print(f'ROC_ACTIVE_WAIT_TIMEOUT={os.environ.get("ROC_ACTIVE_WAIT_TIMEOUT")}: K={K}: submission {statistics.median(sub):7.1f} us, '
              f'bracket {statistics.median(tot):7.1f} us = {statistics.median(tot) / K:6.3f} per step (min {min(tot) / K:6.3f})', flush=True)


def bursts():
    torch, dev, stream, launch, keep = setup()
    for _ in range(3000):
        launch()
    torch.cuda.synchronize(dev)
    for _ in range(12):
        torch.cuda.synchronize(dev)
        for _ in range(200):
            launch()
        torch.cuda.synchronize(dev)


def read(path):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True))[0]
    ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(path))
                if 'iou_aligned' in r['Kernel_Name'])
    groups, cur = [], [ks[0]]
    for a, b in zip(ks, ks[1:]):
        if b[0] - a[1] > 30_000:
            groups.append(cur)
            cur = []
        cur.append(b)
    groups.append(cur)
    groups = [g for g in groups if len(g) == 200]
    print(f'{len(groups)} bursts of 200')
    for lo, hi in ((0, 1), (1, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 200)):
        dur = [g[p][1] - g[p][0] for g in groups for p in range(lo, hi)]
        gap = [g[p][0] - g[p - 1][1] for g in groups for p in range(max(lo, 1), hi)]
        print(f'position {lo:3d}..{hi - 1:3d}: duration median {statistics.median(dur) / 1e3:6.2f} us   gap to previous '
              f'{(statistics.median(gap) / 1e3 if gap else float("nan")):6.2f} us')
    span = [(g[19][1] - g[0][0]) / 20 for g in groups]
    print(f'first 20 of a burst: {statistics.median(span) / 1e3:.3f} us per launch;  last 100: '
          f'{statistics.median([(g[199][1] - g[100][0]) / 100 for g in groups]) / 1e3:.3f}')


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'run'
    {'run': run, 'bursts': bursts, 'bracket': bracket}.get(mode, lambda: read(sys.argv[2]))()
