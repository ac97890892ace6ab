#!/bin/bash
# GPU box, round 2, call 26: leaner prologue of the chunk kernel; kernel arguments preloaded into SGPRs
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02w
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="r02v=build/ab/lib_r02v.so new=$NEW preload=build/ab/lib_preload.so"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02w/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000,100000,250000,500000,2000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02w/ab_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02w/ab_dim5.log
