#!/bin/bash
# GPU box: kernel trace + separate PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_*: the TCC counters do not fit one pass, and
# counters never share a run with the trace statistics) for the kernels of configs[2] / configs[3].
# usage: tools/profile_configs_pmc.sh <tag>   -> gpurun_out/pmc_<tag>/{trace,fetch,write,sq}; then tools/summarize_configs_pmc.py <tag>
set -o pipefail
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/pmc_configs.py" --launches 300 --settle 1500 > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/tools/pmc_configs.py" > "$OUT/fetch.log" 2>&1 || { tail -5 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/tools/pmc_configs.py" > "$OUT/write.log" 2>&1 || { tail -5 "$OUT/write.log"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/sq" -- python3 "$ROOT/tools/pmc_configs.py" > "$OUT/sq.log" 2>&1 || { tail -5 "$OUT/sq.log"; }
python3 "$ROOT/tools/summarize_configs_pmc.py" "$TAG" | tail -60
