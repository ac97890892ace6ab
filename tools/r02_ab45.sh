#!/bin/bash
# GPU box, round 2, call 45: pooled workgroup stack again, with the cull phase at a raised wave priority
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03j
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="chunk=$NEW pool8=$NEW:SPH2POB_POOL_WAVES=8 pool16=$NEW:SPH2POB_POOL_WAVES=16"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03j/ab_pool_prio_1m.log
timeout -k 10 600 python3 tools/ab_kernels.py --pairs 250000,500000,2000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03j/ab_pool_prio_sizes.log
