#!/bin/bash
# GPU box, round 2, call 44: evidence for the chunk kernel + trimmed stages: tests, bench line, rocprof trace + PMC,
# parity report, operator table
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03h
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03h/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r03h/pytest.log; tail -3 gpurun_out/r03h/pytest.log
timeout -k 10 600 python3 bench.py > gpurun_out/r03h/bench.json 2> gpurun_out/r03h/bench.err; echo "bench rc $?"; cut -c1-2500 gpurun_out/r03h/bench.json
timeout -k 10 900 bash tools/profile.sh r03h > gpurun_out/r03h/profile.log 2>&1; echo "profile rc $?"
timeout -k 10 600 python3 tools/parity_report.py > gpurun_out/r03h/parity_report.jsonl 2> gpurun_out/r03h/parity.err; echo "parity rc $?"
timeout -k 10 600 python3 tools/bench_configs.py > gpurun_out/r03h/configs.jsonl 2> gpurun_out/r03h/configs.err; echo "configs rc $?"; cut -c1-300 gpurun_out/r03h/configs.jsonl
