#!/usr/bin/env python3
"""GPU box: A/B timing of the loss launchers of several builds of libsph2pob_hip.so on ONE box in ONE process.

    python tools/ab_loss.py [--pairs N] [--rounds 5] [--launches 300] label=path/to/lib.so ...

Per arm (interleaved in rounds; median per step, HIP events): configs[2]'s step through the C ABI with an upstream
gradient of 1 (`unit`), with an upstream gradient of 0.5 (`scaled`: the 40 MB scale pass runs), with an (n, 5) weight
whose rows are zero for half of the waves (`weighted`), and the plain two-pass sum of 1 M floats (`sum`).  Every arm's
loss value and gradient checksum are printed against the first arm's.
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
    ap.add_argument('--pairs', type=int, default=1_000_000)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--launches', type=int, default=300)
    ap.add_argument('arms', nargs='+')
    args = ap.parse_args()

    import torch
    from sph_retina_amd import _lib, _torch_glue as G
    from tools.bench_configs import boxes
    n = args.pairs
    tgt = boxes(n, 0, 5, alpha=(5, 90), gamma=(-60, 60))
    g = torch.Generator().manual_seed(1)
    pred = tgt + (torch.randn((n, 5), generator=g) * torch.tensor([8., 8., 6., 6., 10.])).cuda()
    pred[:, 0] %= 360
    pred[:, 1].clamp_(1, 179)
    pred[:, 2:4].clamp_(1, 170)
    pred = pred.contiguous()
    weight = torch.ones((n, 5), device='cuda')
    weight[(torch.arange(n, device='cuda') // 64) % 2 == 1] = 0.0   # every other wave holds only negatives
    x = torch.rand(n, device='cuda')
    tmp = tempfile.mkdtemp(prefix='abl_')
    arms = []
    for k, spec in enumerate(args.arms):
        label, path = spec.split('=', 1)
        copy = os.path.join(tmp, f'arm{k}.so')
        shutil.copy(path, copy)
        h = ctypes.CDLL(copy)
        for name in ('sph2pob_loss_fwd_grad_f32', 'sph2pob_loss_grad_scale_f32', 'sph2pob_sum_f32'):
            getattr(h, name).argtypes = _lib.SIGNATURES[name]
            getattr(h, name).restype = ctypes.c_int
        h.sph2pob_loss_sum_workspace_floats.argtypes = [ctypes.c_int64]
        h.sph2pob_loss_sum_workspace_floats.restype = ctypes.c_int64
        h.sph2pob_sum_workspace_floats.restype = ctypes.c_int
        arms.append((label, h))

    st = torch.cuda.current_stream().cuda_stream
    out = torch.empty((), device='cuda')
    stash, gp = torch.empty_like(pred), torch.empty_like(pred)
    one, half = torch.ones((), device='cuda'), torch.full((), 0.5, device='cuda')
    ws2 = torch.empty(arms[0][1].sph2pob_loss_sum_workspace_floats(n), device='cuda')
    ws = torch.empty(arms[0][1].sph2pob_sum_workspace_floats(), device='cuda')

    def check(rc):
        if rc != 0:
            raise SystemExit(f'launcher returned {rc}')

    def steps(h):
        def unit():
            check(h.sph2pob_loss_fwd_grad_f32(G.ptr(pred), G.ptr(tgt), None, 0, 1.0 / n, None, G.ptr(out), G.ptr(ws2), G.ptr(stash),
                                              None, n, 5, 3, 1e-6, st))
            check(h.sph2pob_loss_grad_scale_f32(G.ptr(stash), G.ptr(one), 0, G.ptr(stash), n, 5, st))

        def scaled():
            check(h.sph2pob_loss_fwd_grad_f32(G.ptr(pred), G.ptr(tgt), None, 0, 1.0 / n, None, G.ptr(out), G.ptr(ws2), G.ptr(stash),
                                              None, n, 5, 3, 1e-6, st))
            check(h.sph2pob_loss_grad_scale_f32(G.ptr(stash), G.ptr(half), 0, G.ptr(gp), n, 5, st))

        def weighted():
            check(h.sph2pob_loss_fwd_grad_f32(G.ptr(pred), G.ptr(tgt), G.ptr(weight), 5, 1.0 / n, None, G.ptr(out), G.ptr(ws2),
                                              G.ptr(stash), None, n, 5, 3, 1e-6, st))

        def plain_sum():
            check(h.sph2pob_sum_f32(G.ptr(x), n, 1.0, G.ptr(out), G.ptr(ws), st))
        return {'unit': unit, 'scaled': scaled, 'weighted': weighted, 'sum': plain_sum}

    def timeit(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.launches):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / args.launches

    fns = [(label, steps(h)) for label, h in arms]
    ref = {}
    for label, f in fns:   # values first: every arm against the first
        for name, fn in f.items():
            fn()
            torch.cuda.synchronize()
            val = (float(out), float(gp.double().abs().sum()) if name == 'scaled' else float(stash.double().abs().sum()) if name != 'sum' else 0.0)
            ref.setdefault(name, val)
            print(f'{label:10s} {name:9s} value {val[0]:.9g} |grad| {val[1]:.9g}  same as first arm: {val == ref[name]}')
    for _ in range(2000):
        fns[0][1]['unit']()
    times = {(label, name): [] for label, f in fns for name in f}
    for _ in range(args.rounds):
        for label, f in fns:
            for name, fn in f.items():
                times[(label, name)].append(timeit(fn))
    for (label, name), t in times.items():
        print(f'{label:10s} {name:9s} median {statistics.median(t):8.3f} us  min {min(t):8.3f}  all ' + ' '.join(f'{v:.2f}' for v in t))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
