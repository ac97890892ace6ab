#!/bin/bash
# GPU box, round 2, call 5: full GPU tests, default bench line, rocprof trace + PMC passes, parity report
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02e
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02e/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r02e/pytest.log
tail -5 gpurun_out/r02e/pytest.log
timeout -k 10 600 python3 bench.py > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err; echo "bench rc $?"; cat gpurun_out/r02e/bench.json
timeout -k 10 900 bash tools/profile.sh r02a > gpurun_out/r02e/profile.log 2>&1; echo "profile rc $?"
timeout -k 10 600 python3 tools/parity_report.py 1000000 > gpurun_out/r02e/parity_report.jsonl 2> gpurun_out/r02e/parity.err; echo "parity rc $?"; tail -3 gpurun_out/r02e/parity_report.jsonl
