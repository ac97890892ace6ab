#!/bin/bash
# GPU box: dominant-kernel time (bench.py's HIP-event measurement) against the number of workgroups per CU and batch size
# usage: tools/sweep_slices.sh [sizes...]   (SPH2POB_WGS_PER_CU forces 256*k workgroups; default = the launcher's rule)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() {
  python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline --pairs $1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs', $1, '$2', 'kernel_us %.3f' % (d['roofline']['kernel_ms']*1e3), 'frac %.3f' % d['roofline']['frac'])"
}
for n in ${@:-125000 250000 500000 750000 1000000 1500000 3000000 8000000}; do
  run $n "default"
  for k in 1 2 3 4 5 6 7; do
    SPH2POB_WGS_PER_CU=$k run $n "wgs_per_cu=$k"
  done
done
