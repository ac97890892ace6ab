#!/bin/bash
# GPU box, round 2, call 18: two-tier finishing pass (straight-line common path + one guard) against the guarded form
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02o
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02o/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02o/pytest.log
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="r02n=build/ab/lib_r02n.so new=$NEW"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02o/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,500000,2000000,4000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02o/ab_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02o/ab_dim5.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --nearby 8 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02o/ab_nearby.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --variant efficient --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02o/ab_eff.log
timeout -k 5 120 ./build/finish_rate > gpurun_out/r02o/finish_rate.log 2>&1; grep "lean_finish" gpurun_out/r02o/finish_rate.log | cut -c1-130
