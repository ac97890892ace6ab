#!/bin/bash
# GPU box, round 2, call 23: loss kernels compiled for 4 / 5 / 6 waves per SIMD (config 3: 1 M RBFoV pairs, CIoU fwd + bwd)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02t
L=sph_retina_amd/lib/libsph2pob_hip.so
cp $L /tmp/lib_w4.so
for w in 4 5 6 4 5; do
  if [ $w = 4 ]; then cp /tmp/lib_w4.so $L; else cp build/ab/lib_loss$w.so $L; fi
  touch $L
  timeout -k 10 200 python3 -c "
import sys, json; sys.path.insert(0,'tools')
import bench_configs as B
r = B.config3()
print('loss waves $w:', {k: round(v, 5) for k, v in r.items() if k.endswith('_ms')})
" 2>&1 | grep "loss waves"
done | tee gpurun_out/r02t/loss_waves.log
cp /tmp/lib_w4.so $L
