#!/usr/bin/env python3
"""bench.py — Sph2Pob spherical-IoU throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic boxes resident in HBM:
aligned `sph2pob_standard_iou` over 1,000,000 BFoV pairs per GPU (BASELINE.json configs[1]); for N > 1 every
rank owns its own 1 M-pair shard (N = 8: the 8 M pairs of configs[4] sharded 8x).  Box pairs are independent, so the
path shards with NO data-path collective: results stay on the rank that owns the shard (their consumers — assigner,
loss, NMS — are sharded the same way).  `--gather` additionally assembles the per-shard IoU vectors on every rank with
one RCCL all_gather_into_tensor per step, double-buffered so that the collective of step i (RCCL's stream) overlaps
the kernel of step i+1 (4 MB per rank per step: wire + launch time of the collective exceeds the 10 us kernel, so that
variant is communication-bound by construction).
`value` = pairs processed by all ranks / max-over-ranks wall time.  Weak scaling (per-GPU work fixed).
Timing: W warm-up steps (topped up to 3 000 untimed steps — the clocks only settle after a few thousand back-to-back
launches; the count is reported as config.untimed_steps_before_timing), barrier + synchronize, exactly K timed steps,
barrier + synchronize, max over ranks.

Extra objects on the JSON line:
  roofline      the dominant kernel (iou_aligned) against the HBM roofline: algorithmic bytes = 36 B/pair
                (2 x 16 B boxes in + 4 B IoU out) / average launch duration measured here with
                HIP events on the launch stream; `traffic` = PMC-measured HBM bytes per launch when
                profiles/ holds a summary for this round (collected in separate rocprofv3 --pmc passes), else null.
  cpu_baseline  the CPU oracle (C restatement of the reference path, "port") timed on this host's cores on a
                bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAIRS_PER_GPU = 1_000_000
BYTES_PER_PAIR = 36          # aligned BFoV: 2 * 16 B read + 4 B written
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VARIANT = 'standard'


CLOCK_SETTLE_STEPS = 3000


def rank_env():
    return int(os.environ.get('RANK', '0'))


def make_boxes(n, seed, device):
    """tests/utils/generate_data.py:31-42 (dtype='float') with the harness ranges of tests/test_all_ious.py:141-147."""
    import torch
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    u = torch.rand((n, 4), generator=g)
    boxes = torch.stack([u[:, 0] * 360, u[:, 1] * 180, u[:, 2] * 99 + 1, u[:, 3] * 99 + 1], dim=1)
    return boxes.to(device)


def cpu_baseline(n_sample=1_000_000):
    """Oracle timed on the host: the checker used as a reported baseline, never as the thing shipped."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    b1 = O.generate_boxes(n_sample, 0)
    b2 = O.generate_boxes(n_sample, 1)
    cores = min(O.max_threads(), len(os.sched_getaffinity(0)))
    O.iou_aligned(b1[:20000], b2[:20000], variant=VARIANT, planar='mmcv', nthreads=cores)  # spin up the pool
    reps, t_total = 0, 0.0
    while t_total < 10.0 and reps < 400:   # ~10 s of CPU work, bounded
        t0 = time.perf_counter()
        O.iou_aligned(b1, b2, variant=VARIANT, planar='mmcv', nthreads=cores)
        t_total += time.perf_counter() - t0
        reps += 1
    t1 = time.perf_counter()
    O.iou_aligned(b1[:200000], b2[:200000], variant=VARIANT, planar='mmcv', nthreads=1)
    single = 200000 / (time.perf_counter() - t1)
    return {'value': reps * n_sample / t_total, 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'{reps} x {n_sample} uniform BFoV pairs, sph2pob_{VARIANT}_iou, C oracle (OpenMP)',
            'single_core_pairs_per_s': single}


def pmc_traffic():
    """HBM bytes per launch from the committed PMC summary of this round (profiles/pmc_summary.json)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
    try:
        with open(path) as f:
            return json.load(f).get('iou_aligned', {}).get('hbm_bytes_per_launch')
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5000)
    ap.add_argument('--warmup', type=int, default=500)
    ap.add_argument('--pairs', type=int, default=PAIRS_PER_GPU, help='pairs per GPU per step')
    ap.add_argument('--variant', default=VARIANT, choices=['standard', 'efficient', 'legacy'])
    ap.add_argument('--arithmetic', default='fast', choices=['fast', 'robust', 'reference'],
                    help="'fast' = default closed-form core; 'reference' = the reference's fp32 operation order")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--gather', action='store_true',
                    help='N > 1: also assemble the per-shard IoU vectors on every rank with one RCCL all-gather per step '
                         '(pipelined one step deep); default: shards stay on their ranks, no data-path collective')
    ap.add_argument('--force-dist', action='store_true', help='initialise the process group even with one rank (tests)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sph_retina_amd import _lib, _torch_glue as G
    if rank_env() == 0:
        _lib.build()   # no-op when sph_retina_amd/lib/libsph2pob_hip.so is up to date
        if not args.no_cpu_baseline:
            # the CPU-baseline leg's checker is built here, BEFORE this process initialises the GPU: on this pool a
            # process must not fork + exec (make / gcc) once it has touched the device
            from oracle import oracle as _O
            _O.build()
    else:
        for _ in range(600):   # other ranks wait for rank 0's build instead of racing it
            if not _lib._stale():
                break
            time.sleep(0.5)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    gather = use_dist and args.gather

    n = args.pairs
    b1 = make_boxes(n, 2 * rank, dev)        # rank r owns its own contiguous shard, generated per rank
    b2 = make_boxes(n, 2 * rank + 1, dev)
    # this rank's IoU vector and the assembled vector, double-buffered so that the RCCL all-gather of step i (on the
    # process group's own stream) overlaps the kernel of step i+1 (pre-allocated)
    shards = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2 if gather else 1)]
    gathered = [torch.empty(world * n, dtype=torch.float32, device=dev) for _ in range(2)] if gather else None
    shard = shards[0]
    pending = [None, None]
    lib = _lib.lib()
    stream = torch.cuda.current_stream(dev)
    G.set_arithmetic(args.arithmetic)
    variant_c = G.VARIANTS[args.variant]

    def step(i):
        k = i & 1 if gather else 0
        if gather and pending[k] is not None:
            pending[k].wait()          # stream-ordered: the kernel below waits for the collective that read shards[k]
        rc = lib.sph2pob_iou_aligned_f32(G.ptr(b1), G.ptr(b2), G.ptr(shards[k]), ctypes.c_int64(n), 4, variant_c, 0, 0, 0,
                                         ctypes.c_void_p(stream.cuda_stream))
        if rc:
            _lib.check(rc, 'sph2pob_iou_aligned_f32')
        if gather:
            pending[k] = dist.all_gather_into_tensor(gathered[k], shards[k], async_op=True)

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def barrier():
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # W warm-up steps as asked, topped up to CLOCK_SETTLE_STEPS untimed steps: the GPU clocks only settle after a few
    # thousand back-to-back launches (20 launches in a row run at 10.9 us each, 5 000 at 9.3 us), whatever W is
    for i in range(max(args.warmup, CLOCK_SETTLE_STEPS)):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant-kernel duration: HIP events on the launch stream around K back-to-back launches (no collective)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k_launch = max(args.steps, 1000)   # long enough for a stable average whatever K is
    torch.cuda.synchronize(dev)
    ev0.record(stream)
    for _ in range(k_launch):
        lib.sph2pob_iou_aligned_f32(G.ptr(b1), G.ptr(b2), G.ptr(shard), ctypes.c_int64(n), 4, variant_c, 0, 0, 0,
                                    ctypes.c_void_p(stream.cuda_stream))
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    kernel_ms = ev0.elapsed_time(ev1) / k_launch
    checksum = float(shard.double().sum().item())

    if rank == 0:
        achieved = BYTES_PER_PAIR * n / (kernel_ms * 1e-3) / 1e9
        out = {
            'metric': 'box-pairs/sec, Sph2Pob spherical IoU (aligned BFoV, fp32), 1M pairs per MI355X',
            'value': world * n * args.steps / elapsed,
            'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,   # BASELINE.json:published is empty (README T_cuda has no stated hardware)
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': f'{n:,} uniform random BFoV pairs per GPU, sph2pob_{args.variant}_iou aligned '
                                   f'(BASELINE configs[1]{"; x%d shards" % world if world > 1 else ""}{" + RCCL all-gather of the shards, pipelined one step deep" if gather else ""})',
                       'pairs_per_gpu': n, 'variant': args.variant, 'arithmetic': args.arithmetic, 'parallelism': f'shard{world}', 'gather': bool(gather),
                       'untimed_steps_before_timing': max(args.warmup, CLOCK_SETTLE_STEPS)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': pmc_traffic(),
                         'kernel': 'iou_aligned_compact_kernel' if args.arithmetic == 'fast' and args.variant != 'legacy'
                         else 'iou_aligned_kernel', 'kernel_ms': kernel_ms,
                         'algorithmic_bytes_per_launch': BYTES_PER_PAIR * n},
            'readme_t_cuda_ratio': (n / (kernel_ms * 1e-3)) / (1e6 / 0.0096),
            'checksum': checksum,
        }
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
