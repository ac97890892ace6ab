#!/bin/bash
# GPU box, round 2, call 33: near-parallel form as a called (not inlined) function
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03a
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="inline=$NEW call=build/ab/lib_nearcall.so"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a/ab_nearcall_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03a/ab_nearcall_8m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --nearby 8 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum| tee gpurun_out/r03a/ab_nearcall_nearby.log
