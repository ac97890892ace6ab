#!/bin/bash
# GPU box, round 2, call 43: priority falling with the progress through the finishing pass
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03g
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="shipped=$NEW staged=build/ab/lib_staged.so"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03g/ab_staged_1m.log
timeout -k 10 600 python3 tools/ab_kernels.py --pairs 250000,500000,2000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03g/ab_staged_sizes.log
