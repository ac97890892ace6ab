#!/bin/bash
# GPU box, round 2, call 17: cold (out-of-line) rare blocks against the previous build
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02n
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="r02n=build/ab/lib_r02n.so new=$NEW"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02n/ab2_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 500000,2000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02n/ab2_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02n/ab2_dim5.log
timeout -k 5 120 ./build/finish_rate > gpurun_out/r02n/finish_rate2.log 2>&1; grep "lean_finish" gpurun_out/r02n/finish_rate2.log | cut -c1-130
