#!/bin/bash
# GPU box: dominant-kernel time (bench.py's HIP-event measurement) against SPH2POB_SLICES_PER_WAVE and batch size
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for n in 1000000 4000000; do
  for s in 1 2 3 4 6 8; do
    SPH2POB_SLICES_PER_WAVE=$s python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --pairs $n | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs', $n, 'slices', $s, 'ms_per_step %.5f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])"
  done
done
