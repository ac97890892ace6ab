#!/bin/bash
# GPU box, round 2, call 13: ablations of the chunk kernel (no finishing arithmetic / no cull arithmetic / neither)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02l
NEW=sph_retina_amd/lib/libsph2pob_hip.so
A=build/ab
E=SPH2POB_CHUNK_SLICES=2
ARMS="chunk2=$NEW:$E nofinish=$A/lib_abl_NOFINISH.so:$E nocull=$A/lib_abl_NOCULL.so:$E neither=$A/lib_abl_NEITHER.so:$E"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02l/ab_abl_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 500000,2000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02l/ab_abl_sizes.log
