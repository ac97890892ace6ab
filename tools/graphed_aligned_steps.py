#!/usr/bin/env python3
"""GPU box: K stream-ordered launches of the aligned IoU kernel issued from the host one by one against the same K
launches captured once into a hipGraph (K kernel nodes in a chain) and replayed: what the command processor's
per-launch handling costs a launch-bound step.  us per launch, median of 5 rounds."""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from bench import make_boxes
    from sph_retina_amd import _lib, _torch_glue as G
    lib = _lib.lib()
    dev = torch.device('cuda', 0)
    for n in (250_000, 1_000_000, 2_000_000):
        b1, b2 = make_boxes(n, 0, dev), make_boxes(n, 1, dev)
        out = torch.empty(n, device=dev)
        K = 500

        def launch(stream):
            rc = lib.sph2pob_iou_aligned_f32(G.ptr(b1), G.ptr(b2), G.ptr(out), n, 4, G.VARIANTS['standard'], 0, 0, 0, stream)
            assert rc == 0, rc
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3000):
            launch(st)
        torch.cuda.synchronize()
        ref = out.clone()
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch(side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph, stream=side):
            for _ in range(K):
                launch(side.cuda_stream)
        eager, graphed = [], []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(K):
                launch(st)
            b.record()
            torch.cuda.synchronize()
            eager.append(a.elapsed_time(b) * 1e3 / K)
            a.record()
            graph.replay()
            b.record()
            torch.cuda.synchronize()
            graphed.append(a.elapsed_time(b) * 1e3 / K)
        print(f'pairs {n:8d}  host-issued {statistics.median(eager):7.3f} us / launch   graph of {K} nodes {statistics.median(graphed):7.3f} us / launch'
              f'   equal outputs {torch.equal(out, ref)}')


if __name__ == '__main__':
    main()
