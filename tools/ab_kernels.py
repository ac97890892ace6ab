#!/usr/bin/env python3
"""GPU box: A/B timing of the aligned IoU launcher of several builds of libsph2pob_hip.so on ONE box in ONE process
(boxes differ by a few percent, and so do the first thousand launches of a process: arms are interleaved in rounds).

    python tools/ab_kernels.py [--pairs N[,N..]] [--dim 4|5] [--variant standard] [--rounds 5] [--launches 1000] \
        label=path/to/lib.so[:ENV=val[,ENV=val]] ...

Each arm is its own copy of the shared library (its load-time knobs such as SPH2POB_WGS_PER_CU are read from the
environment when it is loaded).  Prints the median / min per-launch time per arm (HIP events around `launches`
back-to-back launches) and the output difference of every arm against the first one.
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
    ap.add_argument('--pairs', default='1000000')
    ap.add_argument('--dim', type=int, default=4)
    ap.add_argument('--variant', default='standard')
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--launches', type=int, default=1000)
    ap.add_argument('--settle', type=int, default=3000)
    ap.add_argument('--reference-order', action='store_true', help="the reference's fp32 operation order (SPH2POB_FLAG_REFERENCE_ORDER)")
    ap.add_argument('--nearby', type=float, default=0.0, help='sigma (deg) of box2 = box1 + noise; 0 = independent uniform boxes')
    ap.add_argument('arms', nargs='+')
    args = ap.parse_args()

    import torch
    from bench import make_boxes
    dev = torch.device('cuda', 0)
    tmp = tempfile.mkdtemp(prefix='ab_')
    arms = []
    for k, spec in enumerate(args.arms):
        label, rest = spec.split('=', 1)
        path, _, envs = rest.partition(':')
        saved = {}
        for kv in filter(None, envs.split(',')):
            key, val = kv.split('=')
            saved[key] = os.environ.get(key)
            os.environ[key] = val
        copy = os.path.join(tmp, f'arm{k}.so')
        shutil.copy(os.path.join(ROOT, path) if not os.path.isabs(path) else path, copy)
        lib = ctypes.CDLL(copy)
        for key, old in saved.items():
            if old is None:
                del os.environ[key]
            else:
                os.environ[key] = old
        fn = lib.sph2pob_iou_aligned_f32
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
        fn.restype = ctypes.c_int
        arms.append((label, fn))
    variant = {'standard': 0, 'efficient': 1, 'legacy': 2}[args.variant] | (0x100 if args.reference_order else 0)
    stream = torch.cuda.current_stream(dev)
    sp = ctypes.c_void_p(stream.cuda_stream)
    for n in [int(x) for x in args.pairs.split(',')]:
        if args.dim == 4:
            b1, b2 = make_boxes(n, 0, dev), make_boxes(n, 1, dev)
        else:
            g = torch.Generator(device='cpu'); g.manual_seed(0)
            u = torch.rand((2, n, 5), generator=g)
            mk = lambda v: torch.stack([v[:, 0] * 360, v[:, 1] * 180, v[:, 2] * 99 + 1, v[:, 3] * 99 + 1, v[:, 4] * 180 - 90], 1)
            b1, b2 = mk(u[0]).to(dev), mk(u[1]).to(dev)
        if args.nearby > 0:
            g = torch.Generator(device='cpu'); g.manual_seed(5)
            b2 = b1 + (torch.randn(b1.shape, generator=g) * args.nearby).to(dev)
            b2[:, 0] %= 360
            b2[:, 1:4] = b2[:, 1:4].clamp(1, 179)
            b2 = b2.contiguous()
        outs = [torch.empty(n, dtype=torch.float32, device=dev) for _ in arms]

        def launch(k):
            rc = arms[k][1](b1.data_ptr(), b2.data_ptr(), outs[k].data_ptr(), n, args.dim, variant, 0, 0, 0, sp)
            assert rc == 0, (arms[k][0], rc)

        for k in range(len(arms)):
            launch(k)
        torch.cuda.synchronize()
        base = outs[0]
        for k, (label, _) in enumerate(arms):
            d = (outs[k] - base).abs()
            print(f'pairs {n} {label}: checksum {float(outs[k].double().sum()):.6f} zeros {float((outs[k] == 0).float().mean()):.4f} '
                  f'vs {arms[0][0]}: max {float(d.max()):.2e} ndiff {int((d > 0).sum())} n>1e-6 {int((d > 1e-6).sum())} '
                  f'nan {int(torch.isnan(outs[k]).sum())}')
        for i in range(args.settle):
            launch(i % len(arms))
        torch.cuda.synchronize()
        times = [[] for _ in arms]
        for r in range(args.rounds):
            for k in range(len(arms)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(args.launches):
                    launch(k)
                e1.record(stream)
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / args.launches * 1e3)
        for k, (label, _) in enumerate(arms):
            t = times[k]
            print(f'pairs {n} dim {args.dim} {args.variant} {label:24s} median {statistics.median(t):8.3f} us  min {min(t):8.3f}  '
                  f'all {" ".join("%.2f" % x for x in t)}', flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
