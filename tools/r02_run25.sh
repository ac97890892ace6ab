#!/bin/bash
# GPU box, round 2, call 25: soaks and stress on the final kernels of the round
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02v
timeout -k 10 500 python3 tools/soak.py 400 3e-3 2>&1 | grep -v amdgpu.ids > gpurun_out/r02v/soak_400M.log; echo "soak rc $?"; tail -3 gpurun_out/r02v/soak_400M.log | cut -c1-400
timeout -k 10 300 python3 tools/soak_loss.py 40 2>&1 | grep -v amdgpu.ids > gpurun_out/r02v/soak_loss_40M.log; echo "soak_loss rc $?"; tail -4 gpurun_out/r02v/soak_loss_40M.log | cut -c1-400
SPH2POB_STRESS_N=1000000 timeout -k 10 300 python3 tools/stress_compare.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02v/stress_2M.log; echo "stress rc $?"; tail -5 gpurun_out/r02v/stress_2M.log | cut -c1-300
