#!/bin/bash
# GPU box, round 2, call 22: pairwise kernel (arc instantiation, merged leftovers) and RBFoV chunk kernel at 8 waves per SIMD
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02s
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02s/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02s/pytest.log
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="r02r=build/ab/lib_r02r.so new=$NEW"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02s/ab_dim5.log
timeout -k 10 600 python3 tools/bench_configs.py > gpurun_out/r02s/configs.jsonl 2> gpurun_out/r02s/configs.err; echo "configs rc $?"; grep "configs\[3\]" gpurun_out/r02s/configs.jsonl | cut -c1-330
bash tools/sweep_pw_rows.sh 2>&1 | tail -14
