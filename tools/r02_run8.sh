#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02h
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02h/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r02h/pytest.log
tail -4 gpurun_out/r02h/pytest.log
timeout -k 10 600 python3 tools/bench_configs.py > gpurun_out/r02h/configs.jsonl 2> gpurun_out/r02h/configs.err; echo "configs rc $?"; head -1 gpurun_out/r02h/configs.jsonl | cut -c1-900
timeout -k 10 300 python3 tools/host_overhead.py 1000 > gpurun_out/r02h/host_overhead_1k.log 2>&1; grep "us per call" gpurun_out/r02h/host_overhead_1k.log
