#!/bin/bash
# GPU box, round 2, call 36: RBFoV rare blocks (wide gamma, unwrapped angle difference) as called functions
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03d
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="r03c=build/ab/lib_r03c.so new=$NEW"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03d/ab_dim5.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --variant efficient --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03d/ab_dim5_eff.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --variant efficient --nearby 8 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03d/ab_dim5_eff_nearby.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03d/ab_1m.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03d/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03d/pytest.log
timeout -k 10 200 python3 -c "
import sys, json; sys.path.insert(0,'tools')
import bench_configs as B
r = B.config3()
print({k: round(v, 5) for k, v in r.items() if k.endswith('_ms')})
" 2>&1 | grep -v amdgpu | tail -1
