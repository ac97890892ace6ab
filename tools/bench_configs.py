"""GPU box: time BASELINE.json configs[2] (Sph2Pob + CIoU loss fwd+bwd, 1M RBFoV) and configs[3] (MaxIoUAssigner
overlaps 64 GT x ~98k anchors + SphNMS on 5000 boxes / 37 classes) on one MI355X.  One JSON line per config."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sph_retina_amd as S  # noqa: E402
from sph_retina_amd.losses import Sph2PobIoULoss  # noqa: E402
from sph_retina_amd.bbox.nms import SphNMS  # noqa: E402


def timeit(fn, warm=5, reps=30, settle_s=0.03):
    """Seconds per call, HIP events around `reps` back-to-back calls.  After `warm` calls the function is repeated for
    another `settle_s` seconds untimed: the clocks drop while the host prepares inputs and only recover after a few
    thousand launches (a 14 us kernel measured right after an idle gap reads 32 us)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def pmc_roofline(kernel_prefix, seconds, algorithmic_bytes, grid=None):
    """What bounds a kernel, from this round's committed counter passes (profiles/configs_pmc_summary.json, made by
    tools/profile_configs_pmc.sh + tools/summarize_configs_pmc.py): HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, the
    guide's gfx950 correction) and the VALU issue-slot fraction, next to the algorithmic bytes and THIS run's time."""
    out = {'algorithmic_bytes': algorithmic_bytes, 'hbm_frac_algorithmic': algorithmic_bytes / seconds / 8e12 if algorithmic_bytes else None}
    try:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'configs_pmc_summary.json')) as f:
            ks = json.load(f)['kernels']
    except (OSError, ValueError):
        return out
    cands = [e for k, e in ks.items() if e['kernel'].startswith(kernel_prefix) and (grid is None or e['grid_threads'] == grid)]
    if not cands:
        return out
    e = max(cands, key=lambda c: c['grid_threads'])
    out.update(kernel=e['kernel'], counter_bytes=e.get('hbm_bytes'), fetch_calibrated=e.get('fetch_calibrated'),
               hbm_frac_counter_bytes=(e['hbm_bytes'] / seconds / 8e12) if e.get('hbm_bytes') else None,
               valu_issue_frac=e.get('valu_issue_frac'), valu_insts_per_wave=e.get('valu_insts_per_wave'), binds=e.get('binds'),
               profiled_kernel_us=e['avg_ns'] / 1e3, source='profiles/configs_pmc_summary.json')
    return out


def boxes(n, seed, dim=4, alpha=(1, 100), gamma=(-90, 90)):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand((n, 5), generator=g)
    cols = [u[:, 0] * 360, u[:, 1] * 180, u[:, 2] * (alpha[1] - alpha[0]) + alpha[0],
            u[:, 3] * (alpha[1] - alpha[0]) + alpha[0]]
    if dim == 5:
        cols.append(u[:, 4] * (gamma[1] - gamma[0]) + gamma[0])
    return torch.stack(cols, 1).cuda()


def config3(n=1_000_000):
    tgt = boxes(n, 0, 5, alpha=(5, 90), gamma=(-60, 60))
    g = torch.Generator().manual_seed(1)
    pred = tgt + (torch.randn((n, 5), generator=g) * torch.tensor([8., 8., 6., 6., 10.])).cuda()
    pred[:, 0] %= 360
    pred[:, 1].clamp_(1, 179)
    pred[:, 2:4].clamp_(1, 170)
    pred.requires_grad_(True)
    loss = Sph2PobIoULoss(mode='ciou', reduction='mean')

    def step():
        pred.grad = None
        loss(pred, tgt).backward()
    t = timeit(step)
    tf = timeit(lambda: loss(pred.detach(), tgt))
    # the same work through the C ABI directly (no autograd / Python object overhead): fwd kernel + 2-pass sum + bwd kernel
    import ctypes
    from sph_retina_amd import _lib, _torch_glue as G
    lib = _lib.lib()
    p_, t_ = pred.detach().contiguous(), tgt.contiguous()
    elem, out = torch.empty(n, device='cuda'), torch.empty((), device='cuda')
    ws = torch.empty(lib.sph2pob_sum_workspace_floats(), device='cuda')
    gp, one = torch.empty_like(p_), torch.ones((), device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nn, null = ctypes.c_int64(n), ctypes.c_void_p(0)

    ws2 = torch.empty(lib.sph2pob_loss_sum_workspace_floats(n), device='cuda')

    stash = torch.empty_like(p_)

    def abi_step():   # what the autograd Function launches: forward + gradients in one pass, final sum, backward = scale
        lib.sph2pob_loss_fwd_grad_f32(G.ptr(p_), G.ptr(t_), null, 0, ctypes.c_float(1.0 / n), null, ctypes.c_void_p(out.data_ptr()),
                                      G.ptr(ws2), G.ptr(stash), null, nn, 5, 3, ctypes.c_float(1e-6), st)
        lib.sph2pob_loss_grad_scale_f32(G.ptr(stash), ctypes.c_void_p(one.data_ptr()), 0, G.ptr(stash), nn, 5, st)   # in place, g = 1

    def abi_step_two_pass():   # round 1's form: forward (+ sum), then a backward kernel that recomputes the forward
        lib.sph2pob_loss_fwd_sum_f32(G.ptr(p_), G.ptr(t_), null, 0, ctypes.c_float(1.0 / n), ctypes.c_void_p(out.data_ptr()),
                                     G.ptr(ws2), nn, 5, 3, ctypes.c_float(1e-6), st)
        lib.sph2pob_loss_bwd_f32(G.ptr(p_), G.ptr(t_), null, 0, ctypes.c_void_p(one.data_ptr()), 0, ctypes.c_float(1.0 / n),
                                 G.ptr(gp), null, nn, 5, 3, ctypes.c_float(1e-6), st)
    half = torch.full((), 0.5, device='cuda')

    def abi_step_scaled():   # an upstream gradient other than 1 (loss_weight, a loss summed with others and rescaled): the 40 MB scale pass runs
        lib.sph2pob_loss_fwd_grad_f32(G.ptr(p_), G.ptr(t_), null, 0, ctypes.c_float(1.0 / n), null, ctypes.c_void_p(out.data_ptr()),
                                      G.ptr(ws2), G.ptr(stash), null, nn, 5, 3, ctypes.c_float(1e-6), st)
        lib.sph2pob_loss_grad_scale_f32(G.ptr(stash), ctypes.c_void_p(half.data_ptr()), 0, G.ptr(gp), nn, 5, st)
    def abi_step_root():   # the loss is the root of the graph (upstream gradient 1 known on the host: a C / C++ training loop):
        # forward + gradients + final sum, no scale launch; and the same without the scalar loss (gradients only)
        lib.sph2pob_loss_fwd_grad_f32(G.ptr(p_), G.ptr(t_), null, 0, ctypes.c_float(1.0 / n), null, ctypes.c_void_p(out.data_ptr()),
                                      G.ptr(ws2), G.ptr(stash), null, nn, 5, 3, ctypes.c_float(1e-6), st)

    def abi_step_grad_only():
        lib.sph2pob_loss_fwd_grad_f32(G.ptr(p_), G.ptr(t_), null, 0, ctypes.c_float(1.0 / n), null, null, null, G.ptr(stash), null, nn, 5, 3,
                                      ctypes.c_float(1e-6), st)
    tb = timeit(abi_step_two_pass)
    ta = timeit(abi_step)
    tr = timeit(abi_step_root)
    tgo = timeit(abi_step_grad_only)
    ts = timeit(abi_step_scaled)
    # the same Python step with the host out of the way: captured once (torch's whole-network recipe: forward, backward
    # and the accumulation into pred.grad in one hipGraph), replayed
    tg = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        pred.grad = None
        whole = torch.cuda.CUDAGraph()
        with torch.cuda.graph(whole):
            loss(pred, tgt).backward()
        tg = timeit(whole.replay)
    except Exception as e:   # noqa: BLE001  (reported, not fatal: the table's other figures do not depend on it)
        print('graph capture of the loss step failed:', e, file=sys.stderr)
    return {'config': 'configs[2]: 1,000,000 RBFoV pairs, Sph2Pob + CIoU loss forward+backward', 'pairs': n,
            'autograd_fwd_bwd_ms': t * 1e3, 'autograd_fwd_ms': tf * 1e3, 'c_abi_fwd_bwd_ms': ta * 1e3, 'c_abi_root_loss_ms': tr * 1e3, 'c_abi_gradients_only_ms': tgo * 1e3,
            'c_abi_two_pass_fwd_bwd_ms': tb * 1e3, 'c_abi_fwd_bwd_scaled_grad_ms': ts * 1e3, 'graph_replay_fwd_bwd_ms': tg * 1e3 if tg else None,
            'pairs_per_s_c_abi': n / ta, 'pairs_per_s_autograd': n / t, 'pairs_per_s_graph_replay': n / tg if tg else None,
            'algorithmic_bytes_per_pair': 108, 'hbm_GBps_c_abi': 108 * n / ta / 1e9,
            'roofline': pmc_roofline('loss_fwd_grad_kernel', ta, 60.0 * n),
            'hbm_frac_of_8TBps_c_abi': 108 * n / ta / 8e12,
            'note': 'c_abi = loss_fwd_grad (forward + gradients + per-workgroup partial sums in one pass) + final sum + '
                    'grad_scale through the C ABI; two_pass = loss_fwd_sum + final sum + loss_bwd (recomputes the forward); '
                    'autograd = the c_abi launches behind torch.autograd (one Function node), host-bound; graph_replay = that Python '
                    'step (incl. the accumulation into pred.grad) captured once into a hipGraph and replayed'}


def retina_anchors(h=512, w=1024):
    """5-level, 9-anchor RetinaNet grid (strides 8..128, octave_base_scale 4, 3 scales x 3 ratios) mapped pixel->sph
    as sphdet/bbox/box_formator.py:85-92: (x/W*360, y/H*180, w/W*360, h/H*180)."""
    out = []
    for stride in (8, 16, 32, 64, 128):
        fh, fw = math.ceil(h / stride), math.ceil(w / stride)
        ys, xs = torch.meshgrid(torch.arange(fh) * stride, torch.arange(fw) * stride, indexing='ij')
        for sc in (2 ** 0, 2 ** (1 / 3), 2 ** (2 / 3)):
            for ratio in (0.5, 1.0, 2.0):
                size = 4 * stride * sc
                ww, hh = size / math.sqrt(ratio), size * math.sqrt(ratio)
                out.append(torch.stack([xs.flatten() / w * 360, ys.flatten() / h * 180,
                                        torch.full((fh * fw,), ww / w * 360), torch.full((fh * fw,), hh / h * 180)], 1))
    a = torch.cat(out).float()
    a[:, 1].clamp_(0.5, 179.5)
    return a.cuda()


def config4(h=512, w=1024):
    anchors = retina_anchors(h, w)
    g = torch.Generator().manual_seed(0)
    u = torch.rand((64, 4), generator=g)
    gt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1).cuda()
    calc = S.SphOverlaps2D(backend='sph2pob_standard_iou', box_version=4)
    t_iou = timeit(lambda: calc(gt, anchors))
    ov = calc(gt, anchors)

    def assign():
        o = calc(gt, anchors)
        mx, am = o.max(dim=0)
        gmx, gam = o.max(dim=1)
        return mx, am, gmx, gam
    t_assign = timeit(assign)
    from sph_retina_amd.bbox.assigners import SphMaxIoUAssigner
    fused = SphMaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1)
    labels = torch.randint(0, 37, (64,)).cuda()
    t_fused = timeit(lambda: fused.assign(anchors, gt, gt_labels=labels))
    matrix = SphMaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1, fused=False)
    t_matrix = timeit(lambda: matrix.assign(anchors, gt, gt_labels=labels))
    # the same two routes through the C ABI with every buffer allocated once (what a captured graph or a C++ caller pays)
    from sph_retina_amd import _lib, _torch_glue as G
    lib = _lib.lib()
    kk, nn = gt.size(0), anchors.size(0)
    st = G.raw_stream_of(gt.device)
    mo, gi, lab = torch.empty(nn, device='cuda'), torch.empty(nn, dtype=torch.int64, device='cuda'), torch.empty(nn, dtype=torch.int64, device='cuda')
    amo, gm, gam = torch.empty(nn, dtype=torch.int64, device='cuda'), torch.empty(kk, device='cuda'), torch.empty(kk, dtype=torch.int64, device='cuda')
    wsf = torch.empty(lib.sph2pob_iou_assign_workspace_bytes(kk, nn) // 8, dtype=torch.int64, device='cuda')
    stf = torch.zeros(lib.sph2pob_iou_assign_state_bytes(kk, nn) // 8, dtype=torch.int64, device='cuda')
    wsm = torch.empty(lib.sph2pob_assign_workspace_bytes(kk, nn) // 8, dtype=torch.int64, device='cuda')
    ovb = torch.empty((kk, nn), device='cuda')

    def fused_abi():
        lib.sph2pob_iou_assign_f32(G.ptr(gt), kk, G.ptr(anchors), nn, 4, 0, 0, None, None, 0.5, 0.0, 0.4, 0.0, 1, 1, G.ptr(labels),
                                   G.ptr(mo), None, None, None, G.ptr(gi), G.ptr(lab), G.ptr(wsf), G.ptr(stf), st)

    def matrix_abi():
        lib.sph2pob_iou_pairwise_f32(G.ptr(gt), kk, G.ptr(anchors), nn, G.ptr(ovb), 4, 0, 0, 0, 0, st)
        lib.sph2pob_assign_f32(G.ptr(ovb), kk, nn, 0.5, 0.0, 0.4, 0.0, 1, 1, G.ptr(labels), G.ptr(mo), G.ptr(amo), G.ptr(gm),
                               G.ptr(gam), G.ptr(gi), G.ptr(lab), G.ptr(wsm), st)
    t_fused_abi, t_matrix_abi = timeit(fused_abi, reps=200), timeit(matrix_abi, reps=200)
    k = 5000
    rng = np.random.default_rng(4)
    centres = boxes(300, 8, alpha=(5, 60)).cpu().numpy()
    b = centres[rng.integers(0, 300, k)] + rng.standard_normal((k, 4)).astype(np.float32) * 2.0
    b[:, 0] %= 360
    b[:, 1] = b[:, 1].clip(1, 179)
    b[:, 2:] = b[:, 2:].clip(2, 120)
    nb, ns, ni = torch.from_numpy(b).cuda(), torch.rand(k).cuda(), torch.randint(0, 37, (k,)).cuda()
    nms = SphNMS('sph2pob_efficient')
    cfg = dict(type='nms', iou_threshold=0.5, max_num=100)
    t_nms = timeit(lambda: nms(nb, ns, ni, cfg), reps=20)
    t_nms1 = timeit(lambda: nms(nb, ns, torch.zeros_like(ni), cfg), reps=10)
    # the launcher alone (buffers allocated once, no host read of the count): what a captured graph or a C++ caller pays
    wsn = torch.empty(lib.sph2pob_batched_nms_workspace_bytes(k, 4), dtype=torch.uint8, device='cuda')
    kout, dout, stat = torch.empty(100, dtype=torch.int64, device='cuda'), torch.empty((100, 5), device='cuda'), torch.empty(1, dtype=torch.int32, device='cuda')
    ni64 = ni.to(torch.int64)
    t_nms_abi = timeit(lambda: lib.sph2pob_batched_nms_f32(G.ptr(nb), G.ptr(ns), G.ptr(ni64), k, 4, 1, 0.5, 100, G.ptr(wsn), G.ptr(kout),
                                                            G.ptr(dout), G.ptr(stat), st), reps=100)
    kall, dall = torch.empty(k, dtype=torch.int64, device='cuda'), torch.empty((k, 5), device='cuda')
    t_nms1_abi = timeit(lambda: lib.sph2pob_batched_nms_f32(G.ptr(nb), G.ptr(ns), None, k, 4, 1, 0.5, k, G.ptr(wsn), G.ptr(kall),
                                                             G.ptr(dall), G.ptr(stat), st), reps=50)
    m, n = ov.shape
    roof = {'pairwise': pmc_roofline('iou_pairwise_compact_kernel<0, 4, true, 1>', t_iou, 16.0 * (m + n) + 4.0 * m * n, grid=((n + 255) // 256) * 256 * (8 if n < 200000 else 3)),
            'fused_phase1': pmc_roofline('iou_pairwise_compact_kernel<0, 4, true, 2>', t_fused_abi, 16.0 * (m + n), grid=((n + 255) // 256) * 256 * (8 if n < 200000 else 3)),
            'nms_mask': pmc_roofline('nms_mask_compact_kernel', t_nms_abi, None), 'nms_sweep': pmc_roofline('nms_sweep_kernel', t_nms_abi, None)}
    return {'config': 'configs[3]: MaxIoUAssigner overlaps 64 GT x %d anchors (%dx%d ERP grid) + SphNMS 5000 boxes' % (n, h, w),
            'pairs': m * n, 'iou_matrix_ms': t_iou * 1e3, 'pairs_per_s': m * n / t_iou,
            'iou_plus_torch_max_argmax_ms': t_assign * 1e3, 'fused_assign_total_ms': t_fused * 1e3, 'matrix_assign_total_ms': t_matrix * 1e3,
            'fused_assign_c_abi_ms': t_fused_abi * 1e3, 'matrix_assign_c_abi_ms': t_matrix_abi * 1e3, 'frac_pairs_overlapping': float((ov > 0).float().mean()),
            'nms_5000x37cls_ms': t_nms * 1e3, 'nms_5000_single_class_ms': t_nms1 * 1e3,
            'nms_5000x37cls_c_abi_ms': t_nms_abi * 1e3, 'nms_5000_single_class_c_abi_ms': t_nms1_abi * 1e3, 'roofline': roof}


def coder(n=1_000_000):
    """§8f-2: decode (the op in front of loss_bbox) and encode on n RBFoV / BFoV boxes, via the C ABI."""
    import ctypes
    from sph_retina_amd import _lib, _torch_glue as G
    lib = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {'config': 'box coder, %d boxes (sph2pob_coder_*_f32 through the C ABI)' % n}
    for dim in (4, 5):
        a, g = boxes(n, 3, dim).contiguous(), boxes(n, 4, dim).contiguous()
        d = torch.randn((n, dim), device='cuda')
        o, go = torch.empty_like(d), torch.randn((n, dim), device='cuda')
        stds = (ctypes.c_float * dim)(*([0.1, 0.1, 0.2, 0.2, 0.1][:dim]))
        nn, mr = ctypes.c_int64(n), ctypes.c_float(abs(math.log(16 / 1000)))
        te = timeit(lambda: lib.sph2pob_coder_encode_f32(G.ptr(a), G.ptr(g), None, stds, G.ptr(o), nn, dim, st), reps=100)
        td = timeit(lambda: lib.sph2pob_coder_decode_f32(G.ptr(a), G.ptr(d), None, stds, G.ptr(o), nn, 1, dim, mr, 1,
                                                         ctypes.c_float(32), st), reps=100)
        tb = timeit(lambda: lib.sph2pob_coder_decode_bwd_f32(G.ptr(a), G.ptr(d), G.ptr(go), None, stds, G.ptr(o), nn, 1, dim,
                                                             mr, 1, ctypes.c_float(32), st), reps=100)
        bpb = 4 * dim
        out['dim%d' % dim] = {'encode_us': te * 1e6, 'decode_us': td * 1e6, 'decode_bwd_us': tb * 1e6,
                              'bytes_per_box': {'encode': 3 * bpb, 'decode': 3 * bpb, 'decode_bwd': 4 * bpb},
                              'hbm_GBps': {'encode': 3 * bpb * n / te / 1e9, 'decode': 3 * bpb * n / td / 1e9,
                                           'decode_bwd': 4 * bpb * n / tb / 1e9}}
    return out


def unbiased(n=1_000_000):
    """§8f-4: Unbiased IoU (fp64 spherical-polygon area) on the benchmark's uniform pairs, aligned; plus the CPU
    restatement timed on a bounded sample for scale (README quotes 46 s per 1 M pairs for the reference's numpy)."""
    from sph_retina_amd.iou import naive_iou, unbiased_iou
    out = {'config': 'unbiased_iou / naive_iou, %d aligned pairs' % n}
    for dim in (4, 5):
        a, b = boxes(n, 0, dim), boxes(n, 1, dim)
        near = a + torch.randn_like(a) * 4
        near[:, 1].clamp_(1, 179)
        near[:, 2:4].clamp_(1, 170)
        tu = timeit(lambda: unbiased_iou(a, b, is_aligned=True), reps=20)
        tn = timeit(lambda: unbiased_iou(a, near, is_aligned=True), reps=20)
        tv = timeit(lambda: naive_iou(a, b, is_aligned=True), reps=50)
        out['dim%d' % dim] = {'unbiased_uniform_us': tu * 1e6, 'unbiased_uniform_pairs_per_s': n / tu,
                              'unbiased_nearby_us': tn * 1e6, 'unbiased_nearby_pairs_per_s': n / tn,
                              'naive_uniform_us': tv * 1e6}
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import oracle as O
        m = 200_000
        a, b = boxes(m, 0, 4).cpu().numpy(), boxes(m, 1, 4).cpu().numpy()
        O.unbiased_iou(a[:1000], b[:1000])
        t0 = time.perf_counter()
        O.unbiased_iou(a, b)
        dt = time.perf_counter() - t0
        out['cpu_restatement_pairs_per_s'] = m / dt
        out['cpu_threads'] = O.max_threads()
    except Exception as e:  # the oracle is test infrastructure; its absence must not break the GPU numbers
        out['cpu_restatement_error'] = repr(e)
    return out


def variants(n=1_000_000):
    """Aligned IoU, 1 M uniform pairs, every operator of sphdet.iou that the kernels serve (us per call through the
    Python boundary, back-to-back)."""
    from sph_retina_amd import iou as I
    out = {'config': 'aligned IoU operators, %d uniform pairs (us per call)' % n}
    for dim in (4, 5):
        a, b = boxes(n, 0, dim), boxes(n, 1, dim)
        ops = [('sph2pob_standard_iou', I.sph2pob_standard_iou), ('sph2pob_efficient_iou', I.sph2pob_efficient_iou)]
        if dim == 4:
            ops += [('sph2pob_legacy_iou', I.sph2pob_legacy_iou), ('sph_iou', I.sph_iou), ('fov_iou', I.fov_iou)]
        r = {}
        for name, fn in ops:
            r[name] = timeit(lambda: fn(a, b, is_aligned=True), warm=50, reps=300) * 1e6
        out['bfov' if dim == 4 else 'rbfov'] = r
    return out


if __name__ == '__main__':
    # config4 twice: the reference's default 512 x 1024 ERP (98 208 anchors: the "~100k" of BASELINE configs[3]) and the
    # literal 1024 x 2048 grid (392 832 anchors, SURVEY §8d "secondary")
    # (arguments select configurations by name: config3 config4 config4b coder unbiased variants; none = all)
    table = dict(config3=config3, config4=config4, config4b=lambda: config4(1024, 2048), coder=coder, unbiased=unbiased,
                 variants=variants)
    for name in (sys.argv[1:] or list(table)):
        print(json.dumps(table[name]()), flush=True)
