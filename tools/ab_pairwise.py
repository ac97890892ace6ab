#!/usr/bin/env python3
"""GPU box: A/B timing of the pairwise (assigner-pattern) IoU launcher of several builds of libsph2pob_hip.so on ONE box in
ONE process: 64 GT x the RetinaNet ERP anchor grids of configs[3] (98 208 and 392 832 anchors), arms interleaved.

    python tools/ab_pairwise.py [--rounds 5] [--launches 300] label=path/to/lib.so[:ENV=val] ...
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
    from tools.bench_configs import retina_anchors
    tmp = tempfile.mkdtemp(prefix='abp_')
    arms = []
    for i, spec in enumerate(args.arms):
        label, rest = spec.split('=', 1)
        path, _, envs = rest.partition(':')
        for kv in filter(None, envs.split(',')):
            k, v = kv.split('=')
            os.environ[k] = v
        copy = os.path.join(tmp, f'arm{i}.so')
        shutil.copy(path, copy)
        h = ctypes.CDLL(copy)   # load-time knobs are read from the environment here
        for kv in filter(None, envs.split(',')):
            os.environ.pop(kv.split('=')[0], None)
        h.sph2pob_iou_pairwise_f32.argtypes = _lib.SIGNATURES['sph2pob_iou_pairwise_f32']
        h.sph2pob_iou_pairwise_f32.restype = ctypes.c_int
        arms.append((label, h))
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    u = torch.rand((64, 4), generator=g)
    gt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1).cuda().contiguous()
    for grid in ((512, 1024), (1024, 2048)):
        anchors = retina_anchors(*grid).contiguous()
        m, n = gt.size(0), anchors.size(0)
        outs = [torch.empty((m, n), device='cuda') for _ in arms]

        def call(h, out):
            rc = h.sph2pob_iou_pairwise_f32(G.ptr(gt), m, G.ptr(anchors), n, G.ptr(out), 4, G.VARIANTS['standard'], 0, 0, 0, st)
            if rc != 0:
                raise SystemExit(f'launcher returned {rc}')
        for (label, h), out in zip(arms, outs):
            call(h, out)
            torch.cuda.synchronize()
            print(f'{m} x {n} {label:12s} checksum {float(out.double().sum()):.6f} equal to first arm: {torch.equal(out, outs[0])}')
        for _ in range(1000):
            call(arms[0][1], outs[0])
        times = {label: [] for label, _ in arms}
        for _ in range(args.rounds):
            for (label, h), out in zip(arms, outs):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.launches):
                    call(h, out)
                b.record()
                torch.cuda.synchronize()
                times[label].append(a.elapsed_time(b) * 1e3 / args.launches)
        for label, t in times.items():
            print(f'{m} x {n} {label:12s} median {statistics.median(t):8.3f} us  min {min(t):8.3f}  all ' + ' '.join(f'{v:.2f}' for v in t))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
