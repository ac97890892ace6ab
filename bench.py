#!/usr/bin/env python3
"""bench.py — Sph2Pob spherical-IoU throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 starts its own ranks: when WORLD_SIZE is not in the environment the parent — before
it imports torch or touches a GPU — starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ... bench.py <same arguments>` as a child process and relays rank 0's JSON line and the exit code;
launched by torch.distributed.run directly (WORLD_SIZE set) it is one of the ranks.

One "step" = one pass of the hot path over one batch of synthetic boxes resident in HBM: aligned
`sph2pob_standard_iou` (closed-form arithmetic, the default) over the batch.
  N = 1   1,000,000 BFoV pairs (BASELINE.json configs[1]).
  N > 1   the same batch on EVERY GPU (1,000,000 pairs per rank, generated on the rank), no data-path collective: box pairs are
          independent and the consumers (assigner, loss, NMS) shard the same way, so computing needs no exchange
          ("scaling": "weak").  The north star's exchange — configs[4]: 8,000,000 pairs split N ways and ONE RCCL
          `all_gather_into_tensor` of the per-shard IoU vectors per step, double-buffered so that the collective of step i
          overlaps the kernel of step i+1 — is measured in the same job and reported beside it (`north_star_configs4`: with
          the gather, without it, the gather alone, and the speed-ups against the whole batch on one GPU): the 7 us kernel is
          shorter than any collective of 32 MB, so that figure is the collective's, not the kernel's.
          `--scaling strong [--no-gather]` makes configs[4] the timed step instead.
  `--pairs P` changes the pairs per GPU; `--total-pairs T` (strong) the batch.
`value` = pairs processed by all ranks / max-over-ranks wall time of the K timed steps.
Timing: W warm-up steps as asked, topped up to 3 000 untimed steps (the clocks only settle after a few thousand
back-to-back launches; reported as config.untimed_steps_before_timing, and the figure measured with exactly W warm-up
steps is reported next to it as `unsettled`), barrier + synchronize, exactly K timed steps, barrier + synchronize, max
over ranks.

Extra objects on the JSON line:
  roofline      the dominant kernel (iou_aligned_chunk_kernel) against the HBM roofline: algorithmic bytes = 36 B/pair
                (2 x 16 B boxes in + 4 B IoU out) / average launch duration measured here with HIP events
                on the launch stream, median of 5 runs of K/5 back-to-back launches (`rocprof_traced_kernel_ms`: the committed rocprofv3 kernel-trace average of the same
                command, which is 1-1.5 us higher: the tracer brackets every dispatch).  `traffic` (PMC-measured HBM bytes
                per launch) and `valu_active_frac` come from
                this round's committed rocprofv3 summary (separate --pmc passes) and are only attached when this run's
                configuration matches the one profiled; `cold` repeats the measurement rotating through 10 distinct
                input / output sets (360 MB > the 256 MiB Infinity Cache: every launch streams from HBM), `at_8m` is one
                8 M-pair launch (288 MB).
  two_streams   N = 1: the same steps issued alternately on two HIP streams (independent launches: the ramp-up of one
                overlaps the tail of the other); a side figure, never `value`.
                `frac` divides by the event-timed launch (`kernel_ms`); `frac_step` by the driver-visible `ms_per_step`;
                `frac_rocprof` by the committed rocprofv3 average (`rocprof_traced_kernel_ms`) — all three on the line so
                that no reader has to recompute them.
  parity        N = 1: the second half of BASELINE.json's metric ("...; max |dIoU|"), computed OUTSIDE the timed region on
                the very boxes of this run (what the reference's own harness prints, tests/test_all_ious.py:61-78): max /
                mean / p99.9 / count > 1e-5 / count > 1e-4 of |IoU_hip - IoU_ref| against ref32 (the C oracle in the
                reference's fp32 operation order with mmcv's planar algorithm) and against f64 (the same algorithm in
                double with an exact clip), for the benched arithmetic AND for the reference-order mode (with its time per
                launch, so that the trade is on one line), next to ref32's own distance to f64 (the reference's noise
                floor).  The oracle is the checker here, never the thing timed or shipped.
  cpu_baseline  the CPU oracle (C restatement of the reference path, "port") timed on this host's cores on a bounded
                sample of the same workload, next to the reference's own Python timings.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAIRS_ONE_GPU = 1_000_000    # configs[1]
PAIRS_SHARDED = 8_000_000    # configs[4]
BYTES_PER_PAIR = 36          # aligned BFoV: 2 * 16 B read + 4 B written
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VARIANT = 'standard'
CLOCK_SETTLE_STEPS = 3000
COLD_SETS = 10               # x 36 MB per 1 M pairs = 360 MB > 256 MiB Infinity Cache


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5000)
    ap.add_argument('--warmup', type=int, default=500)
    ap.add_argument('--scaling', default=None, choices=['strong', 'weak'],
                    help='N > 1: weak = --pairs per GPU, no collective (default); strong = --total-pairs split N ways with the '
                         'all-gather in the step (the north-star configuration, otherwise measured beside the default)')
    ap.add_argument('--total-pairs', type=int, default=None, help='strong scaling: pairs per step over all ranks '
                    f'(default {PAIRS_ONE_GPU:,} for one GPU, {PAIRS_SHARDED:,} for several)')
    ap.add_argument('--pairs', type=int, default=None, help='pairs per GPU per step (weak scaling; one GPU: the batch)')
    ap.add_argument('--variant', default=VARIANT, choices=['standard', 'efficient', 'legacy'])
    ap.add_argument('--arithmetic', default='fast', choices=['fast', 'robust', 'reference'],
                    help="'fast' = default closed-form core ('robust' is an alias of it since round 2); 'reference' = the "
                         "reference's fp32 operation order")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the max |dIoU| block (oracle on the host, outside the timed region)')
    ap.add_argument('--gather', dest='gather', action='store_true', default=None,
                    help='N > 1: one RCCL all-gather of the per-shard IoU vectors per step, pipelined one step deep '
                         '(default with --scaling strong)')
    ap.add_argument('--no-gather', dest='gather', action='store_false', help='N > 1: time the sharded step only')
    ap.add_argument('--no-extras', action='store_true', help='skip the cold-HBM / 8 M-pair / unsettled side measurements')
    ap.add_argument('--force-dist', action='store_true', help='initialise the process group even with one rank (tests)')
    ap.add_argument('--north-star', action='store_true', help='with --force-dist: run the configs[4] exchange phase with one rank too (tests)')
    ap.add_argument('--dry-run', action='store_true',
                    help='CPU rehearsal of the launch / shard / gather logic on gloo with a stand-in operator (no kernel, '
                         'no roofline): what the CPU test-suite drives')
    return ap.parse_args(argv)


def shard_plan(args, world):
    """(pairs per rank as a list, total, label).  Default: the same batch on every GPU (configs[1] per rank: 'weak'); with
    --scaling strong (or --total-pairs) one batch split N ways."""
    from sph_retina_amd.parallel import shard_bounds
    scaling = args.scaling or ('strong' if args.total_pairs else 'weak')
    if scaling == 'weak':
        per = args.pairs or PAIRS_ONE_GPU
        return [per] * world, per * world, 'weak'
    total = args.total_pairs or (args.pairs if world == 1 and args.pairs else (PAIRS_ONE_GPU if world == 1 else PAIRS_SHARDED))
    counts = [hi - lo for lo, hi in (shard_bounds(total, world, r) for r in range(world))]
    return counts, total, 'strong'


def north_star_exchange(kernel, make, world, rank, dev, dist, steps, warmup, barrier_fn, dry):
    """configs[4] as the north star writes it, measured beside the default: 8 M pairs split over the ranks, ONE
    `all_gather_into_tensor` of the per-shard IoU vectors per step, double-buffered so that the collective of step i overlaps
    the kernel of step i + 1 — then the same without the gather, and the gather alone.  Every rank runs this."""
    import torch
    from sph_retina_amd.parallel import shard_bounds
    total = PAIRS_SHARDED
    lo, hi = shard_bounds(total, world, rank)
    n = hi - lo
    if any((b - a) != n for a, b in (shard_bounds(total, world, r) for r in range(world))):
        return None   # the collective needs equal shards
    b1, b2 = make(n, 300 + 2 * rank, dev), make(n, 301 + 2 * rank, dev)
    shards = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty(world * n, dtype=torch.float32, device=dev) for _ in range(2)]
    pending = [None, None]

    def step(i, with_gather):
        k = i & 1
        if with_gather and pending[k] is not None:
            pending[k].wait()
        kernel(b1, b2, shards[k], n)
        if with_gather:
            pending[k] = dist.all_gather_into_tensor(gathered[k], shards[k], async_op=True)

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def timed(fn):
        drain()
        barrier_fn()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        drain()
        barrier_fn()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for i in range(warmup):
        step(i, True)
    el_g = timed(lambda i: step(i, True))
    el_n = timed(lambda i: step(i, False))
    el_o = timed(lambda i: dist.all_gather_into_tensor(gathered[i & 1], shards[i & 1]))
    ok = bool(torch.equal(gathered[0][lo:hi], shards[0]))
    return {'config': f'BASELINE configs[4]: {total:,} pairs split over {world} GPUs, one RCCL all-gather of the shards per step '
                      '(pipelined one step deep)', 'total_pairs': total, 'pairs_per_gpu': n, 'steps': steps,
            'with_gather': {'ms_per_step': el_g / steps * 1e3, 'value': total * steps / el_g},
            'no_gather': {'ms_per_step': el_n / steps * 1e3, 'value': total * steps / el_n},
            'gather_only': {'ms_per_step': el_o / steps * 1e3, 'bytes_received_per_rank': (world - 1) * n * 4},
            'own_shard_intact_in_gathered': ok}


def self_launch(args, argv):
    """Parent of an N > 1 run: start the ranks as a child process tree and relay rank 0's line.  Nothing here touches a
    GPU (a GPU-initialised process must not fork + exec on this pool), and the native libraries are built first so the
    ranks do not race on them."""
    from sph_retina_amd import _lib
    if not args.dry_run:
        _lib.build()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
    for ln in proc.stdout.splitlines():
        if not ln.startswith('{'):
            print(ln, file=sys.stderr)
    if proc.returncode == 0 and len(lines) != 1:
        print(f'bench.py: expected one JSON line from rank 0, got {len(lines)}', file=sys.stderr)
        return 1
    for ln in lines:
        print(ln)
    return proc.returncode


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
            'single_core_pairs_per_s': single,
            # the reference's own Python on the same workload cannot run on the GPU box (it never travels); its timings
            # from the build container are quoted beside the port so that "the reference's CPU path" is in one place
            'reference_python': {'value': 1e6 / (0.184 + 2.24), 'unit': 'pairs/s', 'cores': 8,
                                 'sample': '1 x 1,000,000 uniform BFoV pairs: sph2pob_standard 0.184 s + vendored planar '
                                           'rotated IoU 2.24 s, torch CPU, 8 threads',
                                 'source': 'BASELINE.md §3 (measured in the build container, not on this host)'}}


def err_stats(got, want):
    import numpy as np
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    d = d[np.isfinite(d)]
    return {'max': float(d.max()), 'mean': float(d.mean()), 'p99_9': float(np.quantile(d, 0.999)),
            'n_gt_1e-5': int((d > 1e-5).sum()), 'n_gt_1e-4': int((d > 1e-4).sum())}


def parity_block(h1, h2, results, variant):
    """max |dIoU| and friends of every (name -> IoU vector computed on the GPU from h1, h2) in `results` against the
    oracle's two instantiations.  Runs on the host after the timed region."""
    import numpy as np
    from oracle import oracle as O
    cores = min(O.max_threads(), len(os.sched_getaffinity(0)))
    ref32 = O.iou_aligned(h1, h2, variant=variant, planar='mmcv', nthreads=cores)
    f64 = O.iou_aligned(h1, h2, variant=variant, planar='exact', dtype=np.float64, nthreads=cores)
    ref32d = O.iou_aligned(h1, h2, variant=variant, planar='diff', nthreads=cores)
    out = {'pairs': int(len(h1)), 'tolerance_north_star': 1e-5,
           'ref32': 'C oracle, reference fp32 operation order, mmcv box_iou_rotated planar algorithm (restated: mmcv is absent)',
           'ref32_diff': "the same with the reference's vendored planar IoU (sphdet/iou/diff_iou_rotated.py: pinned by fixtures)",
           'f64': 'same algorithm in double, exact clip',
           'ref32_vs_f64': err_stats(ref32, f64), 'ref32_diff_vs_f64': err_stats(ref32d, f64)}
    for name, (iou, extra) in results.items():
        out[name] = dict(extra, vs_ref32=err_stats(iou, ref32), vs_ref32_diff=err_stats(iou, ref32d), vs_f64=err_stats(iou, f64))
    return out


def pmc_summary(pairs, variant, arithmetic, kernel):
    """This round's committed PMC summary (profiles/pmc_summary.json), ONLY when it was collected for this very
    configuration: a figure measured for another batch size / variant / build is not this run's measurement."""
    path = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
    try:
        with open(path) as f:
            d = json.load(f).get('iou_aligned', {})
    except (OSError, ValueError):
        return {}
    cfg = d.get('config', {})
    if (cfg.get('pairs'), cfg.get('variant'), cfg.get('arithmetic')) != (pairs, variant, arithmetic) or \
            kernel not in d.get('kernel', ''):
        return {}
    return d


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args, argv))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    import torch
    import torch.distributed as dist
    from sph_retina_amd import _lib, _torch_glue as G
    dry = args.dry_run
    # host-side tensor work here is only the synthetic boxes' RNG: torch sizes its CPU pool by the cores it SEES (128 on a
    # GPU box) while the box grants a share of them; an oversubscribed pool runs into the cgroup's CPU quota and the
    # launching thread is stalled with it (seen as one ~60 ms hole in a later measurement)
    torch.set_num_threads(min(torch.get_num_threads(), 8))
    if not dry:
        if rank == 0:
            _lib.build()   # no-op when sph_retina_amd/lib/libsph2pob_hip.so is up to date
            if not (args.no_cpu_baseline and args.no_parity) and world == 1:
                # the CPU-baseline leg's checker is built here, BEFORE this process initialises the GPU: on this pool a
                # process must not fork + exec (make / gcc) once it has touched the device
                from oracle import oracle as _O
                _O.build()
        else:
            for _ in range(600):   # other ranks wait for rank 0's build instead of racing it
                if not _lib._stale():
                    break
                time.sleep(0.5)

    use_dist = world > 1 or args.force_dist
    if dry:
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if dry:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
    counts, total, scaling = shard_plan(args, world)
    n = counts[rank]
    equal = len(set(counts)) == 1
    gather = (world > 1 and scaling == 'strong') if args.gather is None else (args.gather and use_dist)
    if gather and not equal:
        raise SystemExit('--gather needs equal shards (total pairs divisible by the number of GPUs)')

    b1 = make_boxes(n, 2 * rank, dev)        # rank r owns its own contiguous shard, generated per rank
    b2 = make_boxes(n, 2 * rank + 1, dev)
    # this rank's IoU vector and the assembled vector, double-buffered so that the all-gather of step i (on the
    # process group's own stream) overlaps the kernel of step i+1 (pre-allocated)
    shards = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty(world * n, dtype=torch.float32, device=dev) for _ in range(2)] if gather else None
    pending = [None, None]
    if dry:
        def kernel(x1, x2, out, m, stream=None):   # stand-in operator: the rehearsal is about launch / shard / gather
            torch.sub(x1[:, 0], x2[:, 0], out=out)
    else:
        lib = _lib.lib()
        stream = torch.cuda.current_stream(dev)
        G.set_arithmetic(args.arithmetic)
        variant_c = G.VARIANTS[args.variant]

        def kernel(x1, x2, out, m, stream=stream):
            rc = lib.sph2pob_iou_aligned_f32(G.ptr(x1), G.ptr(x2), G.ptr(out), ctypes.c_int64(m), 4, variant_c, 0, 0, 0,
                                             ctypes.c_void_p(stream.cuda_stream))
            if rc:
                _lib.check(rc, 'sph2pob_iou_aligned_f32')

    if dry:
        def launch_step(k):
            kernel(b1, b2, shards[k], n)
    else:
        # the step's two argument lists, marshalled once (the launch itself is the step; ~2 us of ctypes conversions per call
        # otherwise sit in front of the first launch of every timed bracket)
        step_fn = lib.sph2pob_iou_aligned_f32
        step_args = [(G.ptr(b1), G.ptr(b2), G.ptr(shards[k]), ctypes.c_int64(n), 4, variant_c, 0, 0, 0,
                      ctypes.c_void_p(stream.cuda_stream)) for k in range(2)]

        def launch_step(k):
            rc = step_fn(*step_args[k])
            if rc:
                _lib.check(rc, 'sph2pob_iou_aligned_f32')

    def step(i, with_gather):
        k = i & 1
        if with_gather and pending[k] is not None:
            pending[k].wait()          # stream-ordered: the kernel below waits for the collective that read shards[k]
        launch_step(k)
        if with_gather:
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
        if not dry:
            torch.cuda.synchronize(dev)

    def timed(steps, with_gather):
        """K steps bracketed by barrier + synchronize on both sides; max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, with_gather)
        barrier()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    settle = 0 if dry else CLOCK_SETTLE_STEPS
    extras = not args.no_extras and not dry
    unsettled = None
    for i in range(args.warmup):
        step(i, gather)
    if extras:
        # what the literal contract measures: exactly W warm-up steps, then K steps (the clocks are still ramping)
        el = timed(args.steps, gather)
        unsettled = {'ms_per_step': el / args.steps * 1e3, 'value': total * args.steps / el, 'untimed_steps_before_timing': args.warmup}
    for i in range(max(settle - args.warmup - (args.steps if extras else 0), 0)):
        step(i, gather)
    elapsed = timed(args.steps, gather)
    no_gather = gather_only = None
    if gather:
        el = timed(args.steps, False)
        no_gather = {'ms_per_step': el / args.steps * 1e3, 'value': total * args.steps / el}
        # ... and the exchange alone (no kernel): what bounds the step when the collective outlasts the 7 us kernel
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            dist.all_gather_into_tensor(gathered[i & 1], shards[i & 1])
        barrier()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        gather_only = {'ms_per_step': el / args.steps * 1e3, 'bytes_received_per_rank': (world - 1) * n * 4,
                       'note': 'all_gather_into_tensor of the shards alone; the step cannot be shorter than this'}
    checksum = float(shards[0].double().sum().item())
    # the north star's exchange (configs[4]: 8 M pairs split N ways + one all-gather per step), measured beside the default
    north_star = None
    if ((world > 1 and not args.no_extras) or (args.north_star and use_dist)) and scaling == 'weak' and not gather:
        north_star = north_star_exchange(kernel, make_boxes, world, rank, dev, dist, min(args.steps, 500),
                                         min(max(args.warmup, 20), 200), barrier, dry)
    iou_benched = shards[0].clone() if (world == 1 and not dry and not args.no_parity) else None

    # ---- side measurements on rank 0 (the other ranks wait at the final barrier) ----
    kernel_ms = cold = at_8m = one_gpu_ms = two_streams = parity_gpu = None
    if not dry:
        def events_ms(fn, reps):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            ev0.record(stream)
            for r in range(reps):
                fn(r)
            ev1.record(stream)
            torch.cuda.synchronize(dev)
            return ev0.elapsed_time(ev1) / reps
        def median_ms(fn, reps, segments=5):
            """median over `segments` event-timed runs of `reps` launches: one host stall (the launching thread losing
            its CPU for a scheduler period) lands in one segment instead of in the figure"""
            return sorted(events_ms(fn, reps) for _ in range(segments))[segments // 2]
        # dominant-kernel duration: HIP events on the launch stream around back-to-back launches (no collective)
        kernel_ms = median_ms(lambda r: kernel(b1, b2, shards[0], n), max(args.steps // 5, 200))
        if rank == 0 and extras:
            m1 = PAIRS_ONE_GPU
            sets = [(make_boxes(m1, 100 + 2 * k, dev), make_boxes(m1, 101 + 2 * k, dev),
                     torch.empty(m1, dtype=torch.float32, device=dev)) for k in range(COLD_SETS)]
            torch.cuda.synchronize(dev)
            time.sleep(0.2)   # the CPU pool that drew the boxes goes to sleep before the launch loop needs its core
            events_ms(lambda r: kernel(*sets[r % COLD_SETS], m1), 500)
            t = median_ms(lambda r: kernel(*sets[r % COLD_SETS], m1), 500)
            gbs = BYTES_PER_PAIR * m1 / (t * 1e-3) / 1e9
            cold = {'pairs': m1, 'distinct_sets': COLD_SETS, 'working_set_bytes': COLD_SETS * BYTES_PER_PAIR * m1,
                    'kernel_ms': t, 'achieved': gbs, 'frac': gbs / HBM_PEAK_GBS}
            del sets
            # independent steps issued alternately on TWO HIP streams: the ramp-up of one launch (kernel boundary, first
            # data: ~3 us in which no SIMD works) overlaps the other's tail.  A side figure, never `value`: the contract's
            # steps are ordered on one stream
            if world == 1:
                s2 = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
                o2 = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2)]
                for r in range(1000):
                    kernel(b1, b2, o2[r & 1], n, s2[r & 1])
                torch.cuda.synchronize(dev)
                reps2 = max(args.steps, 2000)
                t0 = time.perf_counter()
                for r in range(reps2):
                    kernel(b1, b2, o2[r & 1], n, s2[r & 1])
                torch.cuda.synchronize(dev)
                el2 = time.perf_counter() - t0
                two_streams = {'ms_per_step': el2 / reps2 * 1e3, 'value': n * reps2 / el2, 'streams': 2,
                               'equal_to_one_stream': bool(torch.equal(o2[0], shards[0]) and torch.equal(o2[1], shards[0]))}
                del o2
            m8 = PAIRS_SHARDED
            c1, c2, co = make_boxes(m8, 200, dev), make_boxes(m8, 201, dev), torch.empty(m8, dtype=torch.float32, device=dev)
            events_ms(lambda r: kernel(c1, c2, co, m8), 500)   # the clocks re-settle after the side measurements above
            t = sorted(events_ms(lambda r: kernel(c1, c2, co, m8), 300) for _ in range(3))[1]
            gbs = BYTES_PER_PAIR * m8 / (t * 1e-3) / 1e9
            at_8m = {'pairs': m8, 'kernel_ms': t, 'achieved': gbs, 'frac': gbs / HBM_PEAK_GBS}
            if world > 1 and total == m8:
                one_gpu_ms = t
            del c1, c2, co
        if iou_benched is not None:
            # the reference-order mode on the same boxes: its IoUs and its time per launch (the other side of the trade)
            ref_c = (variant_c & 0xff) | G.FLAG_REFERENCE_ORDER
            ref_out = torch.empty(n, dtype=torch.float32, device=dev)

            def ref_kernel(r):
                rc = lib.sph2pob_iou_aligned_f32(G.ptr(b1), G.ptr(b2), G.ptr(ref_out), ctypes.c_int64(n), 4, ref_c, 0, 0, 0,
                                                 ctypes.c_void_p(stream.cuda_stream))
                if rc:
                    _lib.check(rc, 'sph2pob_iou_aligned_f32 (reference order)')
            events_ms(ref_kernel, 100)
            ref_ms = median_ms(ref_kernel, 100, 3)
            parity_gpu = {args.arithmetic if args.arithmetic != 'robust' else 'fast': (iou_benched.cpu().numpy(), {'kernel_ms': kernel_ms, 'benched': True})}
            if args.arithmetic != 'reference':
                parity_gpu['reference_order'] = (ref_out.cpu().numpy(), {'kernel_ms': ref_ms, 'benched': False})
        if rank == 0 and world > 1 and one_gpu_ms is None and scaling == 'strong':
            c1, c2, co = make_boxes(total, 200, dev), make_boxes(total, 201, dev), torch.empty(total, dtype=torch.float32, device=dev)
            events_ms(lambda r: kernel(c1, c2, co, total), 100)
            one_gpu_ms = events_ms(lambda r: kernel(c1, c2, co, total), 300)
            del c1, c2, co

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        kern = 'iou_aligned_chunk_kernel' if args.arithmetic != 'reference' and args.variant != 'legacy' else 'iou_aligned_compact_kernel'
        out = {
            'metric': 'box-pairs/sec, Sph2Pob spherical IoU (aligned BFoV, fp32), '
                      + (f'{total:,} pairs per launch on one MI355X' if world == 1 else
                         f'{total:,} pairs per step over {world} MI355X'
                         + (' incl. the RCCL all-gather of the shards' if gather else ''))
                      + ('' if dry else f' (clock-settled: {max(settle, args.warmup)} untimed steps before timing)'),
            'value': total * args.steps / elapsed,
            'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms,
            'higher_is_better': True,
            'scaling': scaling,
            'vs_baseline': None,   # BASELINE.json:published is empty (README T_cuda has no stated hardware)
            'dtype': 'f32',
            'data': 'synthetic' if not dry else 'dry-run (gloo, stand-in operator, no kernel)',
            'config': {'workload': (f'{total:,} uniform random BFoV pairs per step, sph2pob_{args.variant}_iou aligned: '
                                    + ('BASELINE configs[1] on one GPU' if world == 1 and total == PAIRS_ONE_GPU else
                                       'BASELINE configs[1] on every GPU, no collective (pairs are independent; the consumers shard '
                                       'the same way)' if scaling == 'weak' and counts[0] == PAIRS_ONE_GPU else
                                       'BASELINE configs[4]' if total == PAIRS_SHARDED else 'custom batch')
                                    + (f', contiguous shards of {counts[0]:,} pairs on {world} GPUs' if world > 1 else '')
                                    + (', one RCCL all-gather of the shards per step, pipelined one step deep' if gather else '')),
                       'total_pairs': total, 'pairs_per_gpu': counts[0], 'variant': args.variant, 'arithmetic': args.arithmetic,
                       'parallelism': f'shard{world}', 'gather': bool(gather),
                       'untimed_steps_before_timing': max(settle, args.warmup)},
            'checksum': checksum,
        }
        if not dry:
            achieved = BYTES_PER_PAIR * n / (kernel_ms * 1e-3) / 1e9
            pmc = pmc_summary(n, args.variant, 'fast' if args.arithmetic == 'robust' else args.arithmetic, kern)
            out['roofline'] = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': achieved / HBM_PEAK_GBS, 'traffic': pmc.get('hbm_bytes_per_launch'),
                               'traffic_source': pmc.get('source') if pmc.get('hbm_bytes_per_launch') else None,
                               'valu_active_frac': pmc.get('valu_active_frac'),
                               # the committed rocprofv3 --kernel-trace --stats average of this command: the tracer brackets
                               # every dispatch (the launches no longer run back to back from warm caches), which costs this
                               # kernel 1-1.5 us per launch; bench.py itself reports as much when it is run under the tracer
                               'rocprof_traced_kernel_ms': (pmc.get('avg_ns') or 0) / 1e6 or None,
                               'kernel': kern, 'kernel_ms': kernel_ms, 'pairs_per_launch': n,
                               'algorithmic_bytes_per_launch': BYTES_PER_PAIR * n,
                               # the same bytes over the driver-visible step and over the committed rocprofv3 average
                               'frac_step': BYTES_PER_PAIR * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if world == 1 else None,
                               'frac_rocprof': (BYTES_PER_PAIR * n / (pmc['avg_ns'] * 1e-9) / 1e9 / HBM_PEAK_GBS) if pmc.get('avg_ns') else None}
            if cold:
                out['roofline']['cold'] = cold
            if at_8m:
                out['roofline']['at_8m'] = at_8m
            out['readme_t_cuda_ratio'] = (n / (kernel_ms * 1e-3)) / (1e6 / 0.0096)
        if unsettled:
            out['unsettled'] = unsettled
        if two_streams is not None:
            out['two_streams'] = two_streams
        if no_gather:
            out['no_gather'] = no_gather
        if gather_only:
            out['gather_only'] = gather_only
        if north_star:
            if at_8m:   # rank 0's one-GPU time of the whole 8 M batch: the speed-ups of the strong-scaling curve
                north_star['one_gpu_ms'] = at_8m['kernel_ms']
                north_star['speedup_with_gather'] = at_8m['kernel_ms'] / north_star['with_gather']['ms_per_step']
                north_star['speedup_no_gather'] = at_8m['kernel_ms'] / north_star['no_gather']['ms_per_step']
            out['north_star_configs4'] = north_star
        if world > 1 and scaling == 'strong':
            ss = {'total_pairs': total, 'one_gpu_ms': one_gpu_ms}
            if one_gpu_ms:
                ss['speedup'] = one_gpu_ms / ms
                if no_gather:
                    ss['speedup_no_gather'] = one_gpu_ms / no_gather['ms_per_step']
            out['strong_scaling'] = ss
        if parity_gpu is not None:
            out['parity'] = parity_block(b1.cpu().numpy(), b2.cpu().numpy(), parity_gpu, args.variant)
        if not args.no_cpu_baseline and not dry and world == 1:   # rank 0 at N = 1 only
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
