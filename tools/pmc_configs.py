#!/usr/bin/env python3
"""The non-headline kernels of configs[2] / configs[3], launched through the C ABI on fixed inputs — the program that
tools/profile_configs_pmc.sh puts directly behind `rocprofv3 ... --` (no wrapper hop: the profiler's library has initialised
the GPU before this process starts).  Every kernel runs `--launches` times per size after `--settle` untimed calls.

  loss      sph2pob_loss_fwd_grad_f32 (+ final sum + grad scale)   1 M nearby RBFoV pairs, CIoU             configs[2]
  pairwise  sph2pob_iou_pairwise_f32                               64 GT x 98 208 / 392 832 anchors        configs[3]
  assign    sph2pob_assign_f32 on that matrix (the matrix route)
  fused     sph2pob_iou_assign_f32 (no matrix)
  nms       sph2pob_batched_nms_f32                                5 000 boxes x 37 classes, and x 1 class
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--launches', type=int, default=20)
    ap.add_argument('--settle', type=int, default=30)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    import torch
    from sph_retina_amd import _lib, _torch_glue as G
    from tools import ab
    lib = _lib.lib()
    for name, fn in _lib.SIGNATURES.items():
        getattr(lib, name)
    for wl in [w for w in ('loss', 'pairwise', 'assign', 'fused', 'bnms') if not a.only or w in a.only.split(',')]:
        ns = argparse.Namespace(workload=wl, pairs='1000000', dim=4, variant='standard', nearby=0.0, reference_order=False)
        for title, make in ab.workloads(ns, torch, G):
            if wl == 'loss' and 'ciou' not in title:
                continue
            launch, _outs = make(lib)
            for _ in range(a.settle):
                launch()
            torch.cuda.synchronize()
            for _ in range(a.launches):
                launch()
            torch.cuda.synchronize()
            print('ran', title, flush=True)


if __name__ == '__main__':
    main()
