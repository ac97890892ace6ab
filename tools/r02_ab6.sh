#!/bin/bash
# GPU box, round 2, call 6: GPU tests; A/B incl. the one-lane kernel and the chunk kernel; loss host overhead; graph capture
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02f
NEW=sph_retina_amd/lib/libsph2pob_hip.so
A=build/ab
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02f/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r02f/pytest.log
tail -4 gpurun_out/r02f/pytest.log
ARMS="r01=$A/lib_r01.so r02a=$A/lib_r02a.so new=$NEW nocompact=$NEW:SPH2POB_NO_COMPACT=1 chunk=$NEW:SPH2POB_ALIGNED_KERNEL=chunk chunk_w4=$NEW:SPH2POB_ALIGNED_KERNEL=chunk,SPH2POB_WGS_PER_CU=4 chunk_w6=$NEW:SPH2POB_ALIGNED_KERNEL=chunk,SPH2POB_WGS_PER_CU=6"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02f/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 250000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r02f/ab_sizes.log
timeout -k 10 300 python3 tools/host_overhead.py 1000000 > gpurun_out/r02f/host_overhead_1m.log 2>&1; grep "us per call" gpurun_out/r02f/host_overhead_1m.log
timeout -k 10 300 python3 tools/host_overhead.py 1000 > gpurun_out/r02f/host_overhead_1k.log 2>&1; grep "us per call" gpurun_out/r02f/host_overhead_1k.log
timeout -k 10 120 python3 tools/graph_capture_backward.py recipe > gpurun_out/r02f/capture_recipe.log 2>&1; echo "capture recipe rc $?"; tail -2 gpurun_out/r02f/capture_recipe.log
timeout -k 10 120 python3 tools/graph_capture_backward.py stale > gpurun_out/r02f/capture_stale.log 2>&1; echo "capture stale rc $?"; tail -3 gpurun_out/r02f/capture_stale.log | cut -c1-200
