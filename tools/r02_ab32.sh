#!/bin/bash
# GPU box, round 2, call 32: the rotated jitter's angle decisions from sin(a_g - a_p) instead of two atan2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03a
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="atan=build/ab/lib_atan.so sine=$NEW"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a/ab_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a/ab_8m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --nearby 8 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a/ab_nearby.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a/ab_dim5.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --nearby 8 --rounds 2 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03a/ab_dim5_nearby.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03a/pytest.log
timeout -k 10 600 python3 tools/parity_report.py > gpurun_out/r03a/parity_report.jsonl 2> gpurun_out/r03a/parity.err; echo "parity rc $?"
