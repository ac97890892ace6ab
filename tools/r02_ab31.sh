#!/bin/bash
# GPU box, round 2, call 31: non-temporal loads of the boxes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02z
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="plain=$NEW ntload=build/ab/lib_ntload.so"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02z/ab_ntload_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,2000000,8000000,16000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02z/ab_ntload_sizes.log
