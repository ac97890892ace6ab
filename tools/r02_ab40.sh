#!/bin/bash
# GPU box, round 2, call 40: cull priority 1 against plain between 2 M and 8 M pairs
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03g
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="plain=$NEW c1p0=build/ab/lib_prio4.so"
timeout -k 10 600 python3 tools/ab_kernels.py --pairs 1500000,2000000,3000000,4000000,5000000,6000000,8000000,16000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03g/ab_prio3_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03g/ab_prio3_dim5.log
