#!/bin/bash
# GPU box, round 2, call 11: A/B of the one-round chunk kernel (SPH2POB_CHUNK_SLICES) against the persistent compact kernel
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02l
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="persistent=$NEW chunk2=$NEW:SPH2POB_CHUNK_SLICES=2 chunk1=$NEW:SPH2POB_CHUNK_SLICES=1"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02l/ab_chunk_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,500000,2000000,4000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02l/ab_chunk_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02l/ab_chunk_dim5.log
