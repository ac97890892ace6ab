#!/bin/bash
# GPU box, round 2, call 30: workgroup size of the chunk kernel (4 / 8 / 16 independent waves)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02z
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="w4=$NEW w2=$NEW:SPH2POB_CHUNK_WAVES=2 w1=$NEW:SPH2POB_CHUNK_WAVES=1 w8=$NEW:SPH2POB_CHUNK_WAVES=8"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02z/ab_wg2_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 100000,250000,2000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02z/ab_wg2_sizes.log
