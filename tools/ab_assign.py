#!/usr/bin/env python3
"""GPU box: A/B timing of the assigner epilogue (`sph2pob_assign_f32`: three kernels over the (k, n) overlaps) of several
builds of libsph2pob_hip.so on ONE box in ONE process, on the overlaps of configs[3] (64 GT x the ERP anchor grid).

    python tools/ab_assign.py [--rounds 5] [--launches 300] label=path/to/lib.so ...

Arms are interleaved in rounds; prints the median per call (HIP events) per arm and anchor grid, and whether every
arm's outputs equal the first arm's.
"""
import argparse
import ctypes
import os
import shutil
import statistics
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--launches', type=int, default=300)
    ap.add_argument('arms', nargs='+')
    args = ap.parse_args()

    import torch
    from sph_retina_amd import _lib, _torch_glue as G
    import sph_retina_amd as S
    from tools.bench_configs import retina_anchors
    tmp = tempfile.mkdtemp(prefix='aba_')
    arms = []
    for i, spec in enumerate(args.arms):
        label, path = spec.split('=', 1)
        copy = os.path.join(tmp, f'arm{i}.so')
        shutil.copy(path, copy)
        h = ctypes.CDLL(copy)
        h.sph2pob_assign_f32.argtypes = _lib.SIGNATURES['sph2pob_assign_f32']
        h.sph2pob_assign_f32.restype = ctypes.c_int
        h.sph2pob_assign_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64]
        h.sph2pob_assign_workspace_bytes.restype = ctypes.c_int64
        arms.append((label, h))
    st = torch.cuda.current_stream().cuda_stream
    for grid in ((512, 1024), (1024, 2048)):
        anchors = retina_anchors(*grid)
        g = torch.Generator().manual_seed(0)
        u = torch.rand((64, 4), generator=g)
        gt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1).cuda()
        ov = S.SphOverlaps2D(backend='sph2pob_standard_iou', box_version=4)(gt, anchors).contiguous()
        k, n = ov.shape
        labels = torch.arange(k, device='cuda') % 37
        outs = {}

        def make(h):
            mo, amo = torch.empty(n, device='cuda'), torch.empty(n, dtype=torch.int64, device='cuda')
            gm, gam = torch.empty(k, device='cuda'), torch.empty(k, dtype=torch.int64, device='cuda')
            gi, lab = torch.empty(n, dtype=torch.int64, device='cuda'), torch.empty(n, dtype=torch.int64, device='cuda')
            ws = torch.empty(h.sph2pob_assign_workspace_bytes(k, n) // 8, dtype=torch.int64, device='cuda')

            def call():
                rc = h.sph2pob_assign_f32(G.ptr(ov), k, n, 0.5, 0.0, 0.4, 0.0, 1, 1, G.ptr(labels), G.ptr(mo), G.ptr(amo), G.ptr(gm),
                                          G.ptr(gam), G.ptr(gi), G.ptr(lab), G.ptr(ws), st)
                if rc != 0:
                    raise SystemExit(f'sph2pob_assign_f32 returned {rc}')
            return call, (mo, amo, gm, gam, gi, lab)

        calls = []
        for label, h in arms:
            call, res = make(h)
            call()
            torch.cuda.synchronize()
            outs[label] = [t.clone() for t in res]
            same = all(torch.equal(a, b) for a, b in zip(outs[label], outs[arms[0][0]]))
            print(f'{k} x {n} {label:10s} positives {int((outs[label][4] > 0).sum())} equal to first arm: {same}')
            calls.append((label, call))
        for _ in range(1000):
            calls[0][1]()
        times = {label: [] for label, _ in calls}
        for _ in range(args.rounds):
            for label, call in calls:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.launches):
                    call()
                b.record()
                torch.cuda.synchronize()
                times[label].append(a.elapsed_time(b) * 1e3 / args.launches)
        for label, t in times.items():
            print(f'{k} x {n} {label:10s} median {statistics.median(t):8.3f} us  min {min(t):8.3f}  all ' + ' '.join(f'{v:.2f}' for v in t))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
