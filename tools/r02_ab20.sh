#!/bin/bash
# GPU box, round 2, call 20: merged guards (coincident + floors; rotated-jitter decisions + adjustments) against r02p; new chunk tests
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02q
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02q/pytest.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r02q/pytest.log
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="r02p=build/ab/lib_r02p.so new=$NEW"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02q/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 500000,2000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02q/ab_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02q/ab_dim5.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --nearby 8 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02q/ab_nearby.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --variant efficient --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02q/ab_dim5_eff.log
timeout -k 5 120 ./build/finish_rate > gpurun_out/r02q/finish_rate.log 2>&1; grep "lean_finish" gpurun_out/r02q/finish_rate.log | cut -c1-130
