#!/bin/bash
# GPU box, round 2, call 3: GPU tests + A/B of the fused two-stage kernel with deferred rare lanes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02c
NEW=sph_retina_amd/lib/libsph2pob_hip.so
A=build/ab
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r02c/pytest.log
tail -5 gpurun_out/r02c/pytest.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 r01=$A/lib_r01.so new=$NEW new_w5=$NEW:SPH2POB_WGS_PER_CU=5 new_w4=$NEW:SPH2POB_WGS_PER_CU=4 new_nopf=$NEW:SPH2POB_NO_PREFETCH=1 fused_norare=$A/lib_abl_FUSED_NORARE.so:SPH2POB_WGS_PER_CU=4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02c/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,500000,2000000,4000000,8000000 --rounds 3 --launches 300 --settle 1000 r01=$A/lib_r01.so new=$NEW new_w4=$NEW:SPH2POB_WGS_PER_CU=4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02c/ab_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 r01=$A/lib_r01.so new=$NEW new_w4=$NEW:SPH2POB_WGS_PER_CU=4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02c/ab_dim5.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --nearby 8 --rounds 3 r01=$A/lib_r01.so new=$NEW 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02c/ab_nearby.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --variant efficient --rounds 3 r01=$A/lib_r01.so new=$NEW 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02c/ab_eff.log
