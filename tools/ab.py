#!/usr/bin/env python3
"""A/B timing of several builds / load-time settings of libsph2pob_hip.so on ONE box in ONE process.

Boxes differ by a few percent and so do the first thousand launches of a process, so arms are interleaved in rounds and
the median over rounds is what counts.  Two steps, because hipcc time on the GPU box is GPU budget:

  build (here, no GPU):   python tools/ab.py build base fence=-DSPH_ASSIGN_FENCE,-DFOO=1 ...
                          -> build/ab/<label>.so   (label alone = the shipped flags; the tree's .so files travel with gpurun)
  run   (GPU box):        python tools/ab.py run --workload assign --tag r05_assign base fence other=base:SPH2POB_PW_ROWS=12
                          an arm is <label>[=<built label>][:ENV=val[,ENV=val]]  (load-time knobs are read when the copy is loaded)
                          -> one line per arm and size on stdout and, with --tag, in gpurun_out/<tag>.log (copy to profiles/)

Workloads (all through the C ABI, buffers allocated once, HIP events around `--launches` back-to-back calls):
  aligned   sph2pob_iou_aligned_f32, --pairs N[,N..] --dim 4|5 --variant standard|efficient|legacy [--nearby SIGMA] [--reference-order]
  pairwise  sph2pob_iou_pairwise_f32 on configs[3] (64 GT x the ERP anchor grids)
  assign    pairwise + sph2pob_assign_f32 (the matrix route) on the same
  fused     sph2pob_iou_assign_f32 (no matrix) on the same
  loss      sph2pob_loss_fwd_grad_f32 + final sum + grad_scale, 1 M nearby RBFoV pairs, CIoU (configs[2])
  nms       sph2pob_nms_segmented_f32 on 5 000 sorted boxes x 37 classes and on one class of 5 000
  bnms      sph2pob_batched_nms_f32 (unsorted input, no host work) on the same two scenes
Every arm's outputs are compared with the first arm's (bit equality is reported, not assumed).
"""
import argparse
import ctypes
import os
import shutil
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB_DIR = os.path.join(ROOT, 'build', 'ab')


def build(specs):
    from sph_retina_amd import _lib
    os.makedirs(AB_DIR, exist_ok=True)
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    procs = []
    for spec in specs:
        label, _, flags = spec.partition('=')
        out = os.path.join(AB_DIR, label + '.so')
        cmd = [hipcc] + _lib.HIPCC_FLAGS + [f for f in flags.split(',') if f] + ['-o', out] + \
            [os.path.join(_lib.CSRC, s) for s in _lib.SOURCES]
        procs.append((label, out, subprocess.Popen(cmd, cwd=_lib.CSRC)))
        if len(procs) % 4 == 0:   # four compilers at a time (8 cores, ~1 GB each)
            for _l, _o, p in procs[-4:]:
                p.wait()
    for label, out, p in procs:
        assert p.wait() == 0, label
        print('built', os.path.relpath(out, ROOT))


def load_arms(specs, tmp):
    from sph_retina_amd import _lib
    arms = []
    for k, spec in enumerate(specs):
        head, _, envs = spec.partition(':')
        label, _, built = head.partition('=')
        path = os.path.join(AB_DIR, (built or label) + '.so')
        if not os.path.exists(path) and (built or label) == 'shipped':
            path = _lib.LIB_PATH
        saved = {}
        for kv in filter(None, envs.split(',')):
            key, val = kv.split('=')
            saved[key] = os.environ.get(key)
            os.environ[key] = val
        copy = os.path.join(tmp, f'arm{k}.so')
        shutil.copy(path, copy)
        lib = ctypes.CDLL(copy)
        for key, old in saved.items():
            if old is None:
                del os.environ[key]
            else:
                os.environ[key] = old
        for name, argtypes in _lib.SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.argtypes = argtypes
                fn.restype = _lib._RESTYPES.get(name, ctypes.c_int)
        arms.append((label, lib))
    return arms


def _gt64(torch):
    g = torch.Generator().manual_seed(0)
    u = torch.rand((64, 4), generator=g)
    return torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1).cuda()


def workloads(args, torch, G):
    """Yields (title, make(lib) -> (launch, outputs))."""
    st = G.raw_stream_of(torch.device('cuda', 0))
    if args.workload == 'aligned':
        from bench import make_boxes
        variant = {'standard': 0, 'efficient': 1, 'legacy': 2, 'sph_iou': 3, 'fov_iou': 4, 'unbiased': 5, 'naive': 6}[args.variant] | (0x100 if args.reference_order else 0)
        for n in [int(x) for x in args.pairs.split(',')]:
            if args.dim == 4:
                b1, b2 = make_boxes(n, 0, 'cuda'), make_boxes(n, 1, 'cuda')
            else:
                g = torch.Generator().manual_seed(0)
                u = torch.rand((2, n, 5), generator=g)
                mk = lambda v: torch.stack([v[:, 0] * 360, v[:, 1] * 180, v[:, 2] * 99 + 1, v[:, 3] * 99 + 1, v[:, 4] * 180 - 90], 1)  # noqa: E731
                b1, b2 = mk(u[0]).cuda(), mk(u[1]).cuda()
            if args.nearby > 0:
                g = torch.Generator().manual_seed(5)
                b2 = b1 + (torch.randn(b1.shape, generator=g) * args.nearby).cuda()
                b2[:, 0] %= 360
                b2[:, 1:4] = b2[:, 1:4].clamp(1, 179)
                b2 = b2.contiguous()

            def make(lib, b1=b1, b2=b2, n=n):
                out = torch.empty(n, device='cuda')
                return (lambda: lib.sph2pob_iou_aligned_f32(G.ptr(b1), G.ptr(b2), G.ptr(out), n, args.dim, variant, 0, 0, 0, st)), [out]
            yield f'aligned {args.variant} dim {args.dim} pairs {n}', make
    elif args.workload in ('pairwise', 'assign', 'fused'):
        from tools.bench_configs import retina_anchors
        for grid in ((512, 1024), (1024, 2048)):
            anchors, gt = retina_anchors(*grid), _gt64(torch)
            k, n = 64, anchors.size(0)
            labels = (torch.arange(k) % 37).cuda()

            def make(lib, anchors=anchors, gt=gt, k=k, n=n, labels=labels):
                mo, gi, lab = torch.empty(n, device='cuda'), torch.empty(n, dtype=torch.int64, device='cuda'), torch.empty(n, dtype=torch.int64, device='cuda')
                amo, gm, gam = torch.empty(n, dtype=torch.int64, device='cuda'), torch.empty(k, device='cuda'), torch.empty(k, dtype=torch.int64, device='cuda')
                if args.workload == 'fused':
                    ws = torch.empty(lib.sph2pob_iou_assign_workspace_bytes(k, n) // 8, dtype=torch.int64, device='cuda')
                    state = torch.zeros(lib.sph2pob_iou_assign_state_bytes(k, n) // 8, dtype=torch.int64, device='cuda')

                    def launch():
                        return lib.sph2pob_iou_assign_f32(G.ptr(gt), k, G.ptr(anchors), n, 4, 0, 0, None, None, 0.5, 0.0, 0.4, 0.0, 1, 1,
                                                          G.ptr(labels), G.ptr(mo), G.ptr(amo), G.ptr(gm), G.ptr(gam), G.ptr(gi), G.ptr(lab),
                                                          G.ptr(ws), G.ptr(state), st)
                    return launch, [mo, amo, gm, gam, gi, lab]
                ov = torch.empty((k, n), device='cuda')
                ws = torch.empty(lib.sph2pob_assign_workspace_bytes(k, n) // 8, dtype=torch.int64, device='cuda')

                def launch():
                    rc = lib.sph2pob_iou_pairwise_f32(G.ptr(gt), k, G.ptr(anchors), n, G.ptr(ov), 4, 0, 0, 0, 0, st)
                    if args.workload == 'assign':
                        rc |= lib.sph2pob_assign_f32(G.ptr(ov), k, n, 0.5, 0.0, 0.4, 0.0, 1, 1, G.ptr(labels), G.ptr(mo), G.ptr(amo), G.ptr(gm),
                                                     G.ptr(gam), G.ptr(gi), G.ptr(lab), G.ptr(ws), st)
                    return rc
                return launch, ([ov] if args.workload == 'pairwise' else [ov, mo, amo, gm, gam, gi, lab])
            yield f'{args.workload} 64 x {n}', make
    elif args.workload == 'loss':
      for n in ([int(v) for v in args.pairs.split(',')] if args.pairs != '1000000' else [1_000_000]):
          g = torch.Generator().manual_seed(2)
          u = torch.rand((n, 5), generator=g)
          tgt = torch.stack([u[:, 0] * 360, u[:, 1] * 180, u[:, 2] * 99 + 1, u[:, 3] * 99 + 1, u[:, 4] * 180 - 90], 1)
          pred = tgt + torch.randn((n, 5), generator=g) * torch.tensor([8., 8., 6., 6., 10.])
          pred[:, 0] %= 360
          pred[:, 1] = pred[:, 1].clamp(0.5, 179.5)
          pred[:, 2:4] = pred[:, 2:4].clamp(1, 170)
          pred[:, 4] = pred[:, 4].clamp(-89, 89)
          pred, tgt = pred.cuda().contiguous(), tgt.cuda().contiguous()
          for mode_name, mode in (('ciou', 3), ('iou', 0)):
              def make(lib, mode=mode):
                  gp, out = torch.empty((n, 5), device='cuda'), torch.empty(1, device='cuda')
                  ws = torch.empty(lib.sph2pob_loss_sum_workspace_floats(n) + 1024, device='cuda')
                  one = torch.ones(1, device='cuda')

                  def launch():
                      rc = lib.sph2pob_loss_fwd_grad_f32(G.ptr(pred), G.ptr(tgt), None, 0, 1.0 / n, None, G.ptr(out), G.ptr(ws), G.ptr(gp), None, n, 5,
                                                         mode, 1e-6, st)
                      return rc | lib.sph2pob_loss_grad_scale_f32(G.ptr(gp), G.ptr(one), 0, G.ptr(gp), n, 5, st)
                  return launch, [out, gp]
              yield f'loss {mode_name} fwd+grad {n} RBFoV', make
    elif args.workload in ('nms', 'bnms'):
        import numpy as np
        from tools.bench_configs import boxes
        k = 5000
        rng = np.random.default_rng(4)
        centres = boxes(300, 8, alpha=(5, 60)).cpu().numpy()
        b = centres[rng.integers(0, 300, k)] + rng.standard_normal((k, 4)).astype(np.float32) * 2.0
        b[:, 0] %= 360
        b[:, 1] = b[:, 1].clip(1, 179)
        b[:, 2:] = b[:, 2:].clip(2, 120)
        scores = torch.rand(k, generator=torch.Generator().manual_seed(1))
        for title, cls in (('5000 x 37 classes', torch.randint(0, 37, (k,), generator=torch.Generator().manual_seed(2))),
                           ('5000 x 1 class', torch.zeros(k, dtype=torch.int64))):
            order = torch.argsort(cls.double() * 2 - scores.double(), stable=True)
            bs, cs = torch.from_numpy(b)[order].cuda().contiguous(), cls[order].cuda().contiguous()
            seg = int(torch.bincount(cls).max())

            if args.workload == 'bnms':
                ub, us, uc = torch.from_numpy(b).cuda().contiguous(), scores.cuda(), (cls.cuda() if title.endswith('classes') else None)
                mx = 100 if uc is not None else k

                def make(lib, ub=ub, us=us, uc=uc, mx=mx):
                    ws = torch.empty(lib.sph2pob_batched_nms_workspace_bytes(k, 4), dtype=torch.uint8, device='cuda')
                    ko, do, stt = torch.zeros(mx, dtype=torch.int64, device='cuda'), torch.zeros((mx, 5), device='cuda'), torch.zeros(1, dtype=torch.int32, device='cuda')
                    return (lambda: lib.sph2pob_batched_nms_f32(G.ptr(ub), G.ptr(us), G.ptr(uc), k, 4, 1, 0.5, mx, G.ptr(ws), G.ptr(ko), G.ptr(do),
                                                                G.ptr(stt), st)), [ko, do, stt]
                yield f'bnms {title}', make
                continue

            def make(lib, bs=bs, cs=cs, seg=seg):
                keep = torch.empty(k, dtype=torch.uint8, device='cuda')
                ws = torch.empty(lib.sph2pob_nms_segmented_workspace_bytes(k, seg) // 8 + 8, dtype=torch.int64, device='cuda')
                return (lambda: lib.sph2pob_nms_segmented_f32(G.ptr(bs), G.ptr(cs), k, 4, 1, 0.5, seg, G.ptr(ws), G.ptr(keep), st)), [keep]
            yield f'nms {title}', make
    else:
        raise SystemExit('unknown workload ' + args.workload)


def run(args):
    import torch
    from sph_retina_amd import _torch_glue as G
    tmp = tempfile.mkdtemp(prefix='ab_')
    arms = load_arms(args.arms, tmp)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)
    for title, make in workloads(args, torch, G):
        made = [make(lib) for _label, lib in arms]
        for (label, _), (launch, _o) in zip(arms, made):
            rc = launch()
            assert rc == 0, (label, rc)
        torch.cuda.synchronize()
        for (label, _), (_l, outs) in zip(arms[1:], made[1:]):
            same = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(outs, made[0][1]))
            say(f'{title}: {label} vs {arms[0][0]}: outputs {"bit-equal" if same else "DIFFER"}')
        for i in range(args.settle):
            made[i % len(arms)][0]()
        torch.cuda.synchronize()
        times = [[] for _ in arms]
        for _r in range(args.rounds):
            for k, (launch, _o) in enumerate(made):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.launches):
                    launch()
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / args.launches * 1e3)
        for (label, _), t in zip(arms, times):
            say(f'{title}: {label:24s} median {statistics.median(t):8.3f} us  min {min(t):8.3f}  all {" ".join("%.2f" % x for x in t)}')
    shutil.rmtree(tmp, ignore_errors=True)
    if args.tag:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', args.tag + '.log'), 'w') as f:
            f.write('# python tools/ab.py ' + ' '.join(sys.argv[1:]) + '\n' + '\n'.join(lines) + '\n')


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest='cmd', required=True)
    b = sub.add_parser('build')
    b.add_argument('specs', nargs='+')
    r = sub.add_parser('run')
    r.add_argument('--workload', default='aligned')
    r.add_argument('--pairs', default='1000000')
    r.add_argument('--dim', type=int, default=4)
    r.add_argument('--variant', default='standard')
    r.add_argument('--nearby', type=float, default=0.0)
    r.add_argument('--reference-order', action='store_true')
    r.add_argument('--rounds', type=int, default=5)
    r.add_argument('--launches', type=int, default=300)
    r.add_argument('--settle', type=int, default=2000)
    r.add_argument('--tag', default='')
    r.add_argument('arms', nargs='+')
    args = ap.parse_args()
    if args.cmd == 'build':
        build(args.specs)
    else:
        run(args)


if __name__ == '__main__':
    main()
