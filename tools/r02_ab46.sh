#!/bin/bash
# GPU box, round 2, call 46: wave priority in the persistent kernel with the reference-order finish
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03j
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="plain=$NEW prio=build/ab/lib_refprio.so"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --reference-order --rounds 3 --launches 500 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03j/ab_refprio_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --variant legacy --rounds 3 --launches 500 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03j/ab_refprio_legacy.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 8000000 --reference-order --rounds 2 --launches 100 --settle 300 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03j/ab_refprio_8m.log
