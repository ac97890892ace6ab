#!/bin/bash
# GPU box, round 2, call 27: the library built with kernel-argument preloading: all GPU tests, A/B, configs
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02w
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r02w/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r02w/pytest.log
NEW=sph_retina_amd/lib/libsph2pob_hip.so
ARMS="r02v=build/ab/lib_r02v.so new=$NEW"
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --rounds 4 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02w/ab2_1m.log
timeout -k 10 300 python3 tools/ab_kernels.py --pairs 1000000 --dim 5 --rounds 3 $ARMS 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02w/ab2_dim5.log
timeout -k 10 600 python3 tools/bench_configs.py > gpurun_out/r02w/configs.jsonl 2> gpurun_out/r02w/configs.err; echo "configs rc $?"; cut -c1-420 gpurun_out/r02w/configs.jsonl
timeout -k 10 300 python3 tools/demo_hot_path.py 2>&1 | grep -v amdgpu | head -1 | cut -c1-400
