for r in 0 4 8 16 32 64; do SPH2POB_PW_ROWS=$r timeout -k 10 200 python3 -c "
import sys; sys.path.insert(0,'tools')
import bench_configs as B, torch, sph_retina_amd as S
for hw in ((512,1024),(1024,2048)):
    anchors = B.retina_anchors(*hw)
    g = torch.Generator().manual_seed(0); u = torch.rand((64, 4), generator=g)
    gt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1).cuda()
    calc = S.SphOverlaps2D(backend='sph2pob_standard_iou', box_version=4)
    t = B.timeit(lambda: calc(gt, anchors), reps=100)
    print('rows_per_wg', $r, hw, 'iou_matrix_us %.1f' % (t*1e6))
" 2>&1 | grep rows_per; done
