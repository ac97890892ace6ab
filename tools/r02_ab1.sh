#!/bin/bash
# GPU box, round 2, call 1: GPU test suite, then A/B of the round-1 library against the three-stage pipeline
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02a
NEW=sph_retina_amd/lib/libsph2pob_hip.so
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 r01=build/ab/lib_r01.so new=$NEW new_w4=$NEW:SPH2POB_WGS_PER_CU=4 new_w3=$NEW:SPH2POB_WGS_PER_CU=3 new_nopf=$NEW:SPH2POB_NO_PREFETCH=1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02a/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,4000000,8000000 --rounds 3 --launches 300 --settle 1000 r01=build/ab/lib_r01.so new=$NEW 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02a/ab_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 r01=build/ab/lib_r01.so new=$NEW 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02a/ab_dim5.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --nearby 8 --rounds 3 r01=build/ab/lib_r01.so new=$NEW 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02a/ab_nearby.log
