#!/bin/bash
# GPU box, round 2, call 42: wave priority in the assigner kernel (cull rows at 1, finishing passes at 0)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r03g
L=sph_retina_amd/lib/libsph2pob_hip.so
cp $L /tmp/plain.so
for v in plain prio plain prio; do
  if [ $v = plain ]; then cp /tmp/plain.so $L; else cp build/ab/lib_pwprio.so $L; fi; touch $L
  timeout -k 10 200 python3 -c "
import sys; sys.path.insert(0,'tools')
import bench_configs as B, torch, sph_retina_amd as S
for hw in ((512,1024),(1024,2048)):
    anchors = B.retina_anchors(*hw)
    g = torch.Generator().manual_seed(0); u = torch.rand((64, 4), generator=g)
    gt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1).cuda()
    calc = S.SphOverlaps2D(backend='sph2pob_standard_iou', box_version=4)
    t = B.timeit(lambda: calc(gt, anchors), reps=200)
    print('$v', hw, 'iou_matrix_us %.2f' % (t*1e6))
" 2>&1 | grep iou_matrix
done | tee gpurun_out/r03g/pw_prio.log
cp /tmp/plain.so $L
