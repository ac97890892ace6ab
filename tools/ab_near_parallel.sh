#!/bin/bash
# GPU box: A/B of the near-parallel branch of the closed-form path on ONE box (boxes differ by a few percent)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() { python3 bench.py --no-cpu-baseline | python3 -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '%.4e'%d['value'], 'kernel_us %.3f'%(d['roofline']['kernel_ms']*1e3))"; }
for rep in 1 2; do
  SPH2POB_EXTRA_HIPCC_FLAGS=-DSPH2POB_NO_NEAR_PARALLEL python3 -c "from sph_retina_amd import _lib; _lib.build(force=True)"; run without
  python3 -c "from sph_retina_amd import _lib; _lib.build(force=True)"; run with
done
