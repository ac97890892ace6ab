#!/bin/bash
# GPU box, round 2, call 7: GPU tests, secondary configurations (configs[2], configs[3] at both grids, coders, operators)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02g
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02g/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r02g/pytest.log
tail -4 gpurun_out/r02g/pytest.log
timeout -k 10 600 python3 tools/bench_configs.py > gpurun_out/r02g/configs.jsonl 2> gpurun_out/r02g/configs.err; echo "configs rc $?"; cut -c1-900 gpurun_out/r02g/configs.jsonl
timeout -k 10 300 python3 bench.py --arithmetic reference --no-cpu-baseline --no-extras --steps 2000 > gpurun_out/r02g/bench_ref.json 2>/dev/null; cut -c1-400 gpurun_out/r02g/bench_ref.json
