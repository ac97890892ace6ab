#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of the secondary configurations (tools/bench_configs.py: loss, assigner, NMS,
# coders, unbiased IoU) and of the end-to-end head pipeline (tools/demo_hot_path.py).
# usage: tools/profile_configs.sh <tag>  -> gpurun_out/prof_<tag>_configs/, gpurun_out/prof_<tag>_pipeline/
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_configs" -- python3 "$ROOT/tools/bench_configs.py" > "$OUT/${TAG}_configs.jsonl" 2> "$OUT/${TAG}_configs.err" || tail -3 "$OUT/${TAG}_configs.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_pipeline" -- python3 "$ROOT/tools/demo_hot_path.py" > "$OUT/${TAG}_pipeline.jsonl" 2> "$OUT/${TAG}_pipeline.err" || tail -3 "$OUT/${TAG}_pipeline.err"
for d in configs pipeline; do
  f=$(ls "$OUT"/prof_${TAG}_$d/*/*_kernel_stats.csv | head -1)
  cp "$f" "$OUT/${TAG}_${d}_kernel_stats.csv"
done
head -12 "$OUT/${TAG}_configs_kernel_stats.csv" | cut -c1-160
