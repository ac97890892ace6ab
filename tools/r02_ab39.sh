#!/bin/bash
# GPU box, round 2, call 39: priority levels of the cull phase (pass at 0 / 1)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03g
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="plain=$NEW c3p0=build/ab/lib_prio12.so c2p0=build/ab/lib_prio8.so c1p0=build/ab/lib_prio4.so c3p1=build/ab/lib_prio13.so"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03g/ab_prio2_1m.log
timeout -k 10 400 python3 tools/ab_kernels.py --pairs 500000,2000000,4000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03g/ab_prio2_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --nearby 8 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03g/ab_prio2_nearby.log
