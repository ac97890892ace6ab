#!/bin/bash
# GPU box, round 2, call 38: wave priority (s_setprio) of the cull phase / of the finishing pass in the chunk kernel
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03g
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="plain=$NEW cull_high=build/ab/lib_prio12.so pass_high=build/ab/lib_prio3.so"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03g/ab_prio_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03g/ab_prio_sizes.log
