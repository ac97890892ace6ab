#!/bin/bash
# GPU box, round 2, call 28: scheduler flags
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02x
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="new=$NEW maxilp=build/ab/lib_max-ilp.so"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02x/ab_flags_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02x/ab_flags_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02x/ab_flags_dim5.log
