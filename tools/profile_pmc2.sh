#!/bin/bash
# extra PMC pass for the dominant kernel: LDS, wait and busy counters (+ GRBM for the effective clock)
set -o pipefail
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_a" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_a.log" 2>&1 || tail -3 "$OUT/pmc_a.log"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d "$OUT/pmc_b" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_b.log" 2>&1 || tail -3 "$OUT/pmc_b.log"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d "$OUT/pmc_c" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_c.log" 2>&1 || tail -3 "$OUT/pmc_c.log"
python3 - <<PY
import csv, glob, collections
for pas in ('pmc_a','pmc_b','pmc_c'):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv" % pas):
        acc = collections.defaultdict(list); dur=[]
        for r in csv.DictReader(open(f)):
            if 'iou_aligned' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value'])); dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
        print(pas, 'avg dur ns', sum(dur)/max(len(dur),1), {k: round(sum(v)/len(v),1) for k,v in acc.items()})
PY
