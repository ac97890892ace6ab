#!/bin/bash
# GPU-side kernel durations (rocprofv3 kernel-trace) of the dominant kernel for several batch sizes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ktime
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for p in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/n$p" -- python3 "$ROOT/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --pairs $p > "$OUT/n$p.log" 2>&1
  f=$(ls $OUT/n$p/*/*_kernel_stats.csv | head -1)
  grep iou_aligned "$f" | awk -F'","' -v p=$p '{gsub(/"/,"",$0); print p, "calls", $2, "avg_ns", $4, "min_ns", $6, "max_ns", $7}'
done
