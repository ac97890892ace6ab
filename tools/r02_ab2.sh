#!/bin/bash
# GPU box, round 2, call 2: stage ablations of the three-stage pipeline + PMC counters, r01 vs new
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02b
NEW=sph_retina_amd/lib/libsph2pob_hip.so
A=build/ab
for P in 1000000 8000000; do
timeout -k 10 300 python3 tools/ab_kernels.py --pairs $P --rounds 3 --launches 400 --settle 2000 r01=$A/lib_r01.so new=$NEW:SPH2POB_WGS_PER_CU=4 norare=$A/lib_abl_NO_RARE.so:SPH2POB_WGS_PER_CU=4 nostage2=$A/lib_abl_NO_STAGE2.so:SPH2POB_WGS_PER_CU=4 nostage1=$A/lib_abl_NO_STAGE1.so:SPH2POB_WGS_PER_CU=4 fused=$A/lib_abl_FUSED.so:SPH2POB_WGS_PER_CU=4 fused_norare=$A/lib_abl_FUSED_NORARE.so:SPH2POB_WGS_PER_CU=4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02b/abl_$P.log
done
cd /tmp && export TMPDIR=/tmp
for arm in r01=$A/lib_r01.so new=$NEW:SPH2POB_WGS_PER_CU=4 norare=$A/lib_abl_NO_RARE.so:SPH2POB_WGS_PER_CU=4; do
  tag=${arm%%=*}
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $ROOT/gpurun_out/r02b/pmc_a_$tag -- python3 $ROOT/tools/ab_kernels.py --pairs 8000000 --rounds 1 --launches 20 --settle 20 $tag=$ROOT/${arm#*=} > $ROOT/gpurun_out/r02b/pmc_a_$tag.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $ROOT/gpurun_out/r02b/pmc_b_$tag -- python3 $ROOT/tools/ab_kernels.py --pairs 8000000 --rounds 1 --launches 20 --settle 20 $tag=$ROOT/${arm#*=} > $ROOT/gpurun_out/r02b/pmc_b_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$ROOT/gpurun_out/r02b/pmc_*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open(f)):
        if 'iou_aligned' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value'])); dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    print(f.split('/')[-3], 'avg dur ns', round(sum(dur)/max(len(dur),1)), {k: round(sum(v)/len(v)) for k,v in acc.items()})
PY
