#!/bin/bash
# GPU box, round 2, call 47: evidence at the head of round 2: tests, bench line, rocprof trace + PMC,
# parity report, operator table
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r04d
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04d/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r04d/pytest.log; tail -3 gpurun_out/r04d/pytest.log
timeout -k 10 600 python3 bench.py > gpurun_out/r04d/bench.json 2> gpurun_out/r04d/bench.err; echo "bench rc $?"; cut -c1-2500 gpurun_out/r04d/bench.json
timeout -k 10 900 bash tools/profile.sh r04d > gpurun_out/r04d/profile.log 2>&1; echo "profile rc $?"
timeout -k 10 600 python3 tools/parity_report.py > gpurun_out/r04d/parity_report.jsonl 2> gpurun_out/r04d/parity.err; echo "parity rc $?"
timeout -k 10 600 python3 tools/bench_configs.py > gpurun_out/r04d/configs.jsonl 2> gpurun_out/r04d/configs.err; echo "configs rc $?"; cut -c1-300 gpurun_out/r04d/configs.jsonl
