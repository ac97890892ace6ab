#!/bin/bash
# GPU box, round 2, call 41: cull-phase priority as shipped (launches of up to ~2 rounds) against SPH2POB_NO_PRIO=1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03g
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="noprio=$NEW:SPH2POB_NO_PRIO=1 shipped=$NEW"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03g/ab_prio_shipped_1m.log
timeout -k 10 600 python3 tools/ab_kernels.py --pairs 100000,250000,500000,2000000,2600000,3000000,8000000 --rounds 3 --launches 300 --settle 1000 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03g/ab_prio_shipped_sizes.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | grep -v checksum | tee gpurun_out/r03g/ab_prio_shipped_dim5.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03g/pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r03g/pytest.log
