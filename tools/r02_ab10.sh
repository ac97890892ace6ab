#!/bin/bash
# GPU box, round 2, call 10: wave -> SIMD mapping of the dominant launch shape; A/B of the tail-chunk rotation
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02l
timeout -k 5 60 ./build/hwid > gpurun_out/r02l/hwid.log 2>&1; cat gpurun_out/r02l/hwid.log
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="rot0=$NEW rot1=$NEW:SPH2POB_TAIL_ROT=1 rot2=$NEW:SPH2POB_TAIL_ROT=2 rot3=$NEW:SPH2POB_TAIL_ROT=3"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02l/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,4000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02l/ab_sizes.log
