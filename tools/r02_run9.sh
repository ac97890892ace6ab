#!/bin/bash
# GPU box, round 2, call 9: soaks and stress on the round-2 kernels, final bench line, rocprof trace + PMC, configs
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02j
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02j/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r02j/pytest.log; tail -3 gpurun_out/r02j/pytest.log
timeout -k 10 900 python3 tools/soak.py 400 3e-3 2>&1 | grep -v amdgpu.ids > gpurun_out/r02j/soak_400M.log; echo "soak rc $?"; tail -3 gpurun_out/r02j/soak_400M.log | cut -c1-400
timeout -k 10 900 python3 tools/soak_loss.py 40 2>&1 | grep -v amdgpu.ids > gpurun_out/r02j/soak_loss_40M.log; echo "soak_loss rc $?"; tail -4 gpurun_out/r02j/soak_loss_40M.log | cut -c1-400
SPH2POB_STRESS_N=1000000 timeout -k 10 900 python3 tools/stress_compare.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02j/stress_2M.log; echo "stress rc $?"; tail -5 gpurun_out/r02j/stress_2M.log | cut -c1-300
timeout -k 10 600 python3 bench.py > gpurun_out/r02j/bench.json 2> gpurun_out/r02j/bench.err; echo "bench rc $?"; cut -c1-600 gpurun_out/r02j/bench.json
timeout -k 10 900 bash tools/profile.sh r02b > gpurun_out/r02j/profile.log 2>&1; echo "profile rc $?"
timeout -k 10 600 python3 tools/bench_configs.py > gpurun_out/r02j/configs.jsonl 2> gpurun_out/r02j/configs.err; echo "configs rc $?"
