#!/bin/bash
# GPU box, round 2, call 4: GPU tests + A/B of lean_finish (guarded rare branches) builds
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02d
A=build/ab
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r02d/pytest.log
tail -5 gpurun_out/r02d/pytest.log
ARMS="r01=$A/lib_r01.so lb6pp1=$A/lib_d_lb6_pp1.so lb5pp1=$A/lib_d_lb5_pp1.so lb6pp0=$A/lib_d_lb6_pp0.so lb7pp0=$A/lib_d_lb7_pp0.so lb7pp0w6=$A/lib_d_lb7_pp0.so:SPH2POB_WGS_PER_CU=6"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02d/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,4000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02d/ab_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 r01=$A/lib_r01.so new=$A/lib_d_lb6_pp1.so newpp0=$A/lib_d_lb6_pp0.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02d/ab_dim5.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --nearby 8 --rounds 3 r01=$A/lib_r01.so new=$A/lib_d_lb6_pp1.so newpp0=$A/lib_d_lb6_pp0.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02d/ab_nearby.log
