#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + separate PMC passes for bench.py (dominant kernel: iou_aligned).
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq}
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the kernel-trace pass runs bench.py's own default step / warm-up counts (the command BENCH numbers come from: the
# clocks only settle after a few thousand back-to-back launches) without the side measurements (--no-extras: the
# cold-HBM rotation and the 8 M-pair launches run the same kernel at other sizes and would mix into its statistics);
# the counter passes use a short run
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras $* > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
ARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-extras $*"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { tail -5 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1 || { tail -5 "$OUT/pmc_write.log"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1 || { tail -5 "$OUT/pmc_sq.log"; }
find "$OUT" -name "*.csv" | head -20
