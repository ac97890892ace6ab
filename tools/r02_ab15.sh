#!/bin/bash
# GPU box, round 2, call 15: SQ counters of the persistent / chunk / pooled kernels
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r02l
bash tools/profile_pmc2.sh l_persistent --no-extras 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02l/pmc_persistent.log
export SPH2POB_CHUNK_SLICES=2
bash tools/profile_pmc2.sh l_chunk2 --no-extras 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02l/pmc_chunk2.log
unset SPH2POB_CHUNK_SLICES
export SPH2POB_POOL_WAVES=8
bash tools/profile_pmc2.sh l_pool8 --no-extras 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02l/pmc_pool8.log
