#!/bin/bash
# GPU box, round 2, call 16: trimmed cull / finish against the previous commit; GPU tests; pass cost; VALU instruction count
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02n
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02n/pytest.log
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="r02m=build/ab/lib_r02m.so new=$NEW"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02n/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,500000,2000000,4000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02n/ab_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02n/ab_dim5.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --nearby 8 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02n/ab_nearby.log
timeout -k 5 120 ./build/finish_rate > gpurun_out/r02n/finish_rate.log 2>&1; grep lean_finish gpurun_out/r02n/finish_rate.log
bash tools/profile_pmc2.sh n_new --no-extras 2>&1 | grep -v amdgpu.ids | grep pmc_a | tee gpurun_out/r02n/pmc_new.log
