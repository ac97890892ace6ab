"""GPU: adversarial differential test (tools/stress_compare.py at a reduced size) — poles, the theta seam, tiny and
huge boxes, identical / near-identical pairs, integer degrees, |gamma| up to 360: the closed-form core must stay in
range and at least as close to the f64 oracle as the reference-order kernels on every set."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_adversarial_sets_fast_vs_reference_order_vs_truth():
    os.environ['SPH2POB_STRESS_N'] = '40000'
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import stress_compare
    bad, rows = stress_compare.run()
    assert bad == 0
    for dim, name, v, df, dr, dd in rows:
        assert np.quantile(df, 0.999) <= 3 * np.quantile(dr, 0.999) + 1e-4, (dim, name, v)
        assert df.mean() <= 3 * dr.mean() + 2e-6, (dim, name, v, df.mean(), dr.mean())
        assert df.max() <= max(3 * dr.max(), 0.1), (dim, name, v, df.max(), dr.max())


@pytest.mark.parametrize('arith', ['fast', 'reference'])
def test_loss_exactly_parallel_planar_boxes(arith):
    """Regression: after the jitter the two planar angles of this pair are bit-identical in reference order, so the
    planar clip sees sin(delta) == 0 and an edge parallel to (and outside) a slab; the loss and its gradient must
    stay finite and agree with the IoU kernels."""
    import torch
    import sph_retina_amd as S
    from sph_retina_amd.losses import Sph2PobIoULoss
    b1 = torch.tensor([[1., 49.131344, 88.38273, 6.3124614, -83.81888]], device='cuda')
    b2 = torch.tensor([[0., 49.994923, 87.12384, 8.817537, -83.099594]], device='cuda')
    prev = S.get_arithmetic()
    S.set_arithmetic(arith)
    try:
        iou = S.sph2pob_standard_iou(b1, b2, is_aligned=True).item()
        for mode in ('iou', 'giou', 'diou', 'ciou'):
            p = b1.clone().requires_grad_(True)
            loss = Sph2PobIoULoss(mode=mode, reduction='none')(p, b2)
            loss.sum().backward()
            assert torch.isfinite(loss).all() and torch.isfinite(p.grad).all(), (mode, loss, p.grad)
            if mode == 'iou':
                assert abs((1 - loss.item()) - iou) < 1e-4
    finally:
        S.set_arithmetic(prev)


def test_loss_adversarial_sets_finite():
    """tools/stress_loss.py at a reduced size: every loss mode, both arithmetic modes, all adversarial sets —
    no non-finite loss or gradient anywhere."""
    os.environ['SPH2POB_STRESS_N'] = '20000'
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import stress_loss
    assert stress_loss.run() == 0


def test_launchers_are_graph_capturable():
    """The C ABI only enqueues on the given stream (no allocation, no synchronisation): a step made of the IoU kernel,
    the fused loss forward + backward and the coder decode replays from a hipGraph with identical results."""
    import torch
    import sph_retina_amd as S
    from oracle import oracle as O
    n = 50000
    b1 = torch.from_numpy(O.generate_boxes(n, 3)).cuda()
    b2 = torch.from_numpy(O.generate_boxes(n, 4)).cuda()
    anchors = b1.clone()
    deltas = (torch.randn(n, 4, device='cuda') * 0.1).requires_grad_(True)
    coder = S.DeltaXYWHSphBBoxCoder(target_stds=(0.1, 0.1, 0.2, 0.2))
    loss_fn = S.Sph2PobIoULoss(mode='ciou')

    def step():
        iou = S.sph2pob_standard_iou(b1, b2, is_aligned=True)
        loss = loss_fn(coder.decode(anchors, deltas), b1)
        grad, = torch.autograd.grad(loss, deltas)
        return iou, loss.detach(), grad   # keeping the autograd graph alive across iterations breaks torch's capture
    eager = [t.clone() for t in step()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):                       # warm-up on the capture stream (workspace allocation, lazy init)
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    for t in captured:
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, captured):
        assert torch.equal(a, b)
    b2.copy_(b1)                                 # new inputs, same graph
    graph.replay()
    torch.cuda.synchronize()
    assert float(captured[0].min()) > 0.8        # IoU(x, x) after the reference's jitter


def test_default_arithmetic_is_right_on_jitter_cancellation_pairs():
    """Pairs found by the 2 M-pair soak: the reference's two jitter steps cancel and leave the planar boxes parallel to
    ~1e-6 rad, where the plain fp32 boundary integral is off by up to 2e-2.  Such lanes are classified `rare` by
    lean_stage1 and take the general form with the near-parallel safeguard: the DEFAULT arithmetic must be right on
    them (round 1 asserted the opposite and offered an opt-in 'robust' mode; that name is still accepted)."""
    import torch
    import sph_retina_amd as S
    from oracle import oracle as O
    b1 = np.array([[237.02872, 171.37167, 179.9, 179.0], [128.4777, 38.103848, 179.9, 179.9],
                   [88.996124, 86.51253, 120.0, 150.0]], np.float32)
    b2 = np.array([[236.92949, 179.62257, 179.9, 179.9], [128.5839, 14.652416, 179.9, 179.9],
                   [87.48591, 98.4085, 126.54463, 173.79935]], np.float32)
    tru = O.iou_aligned(b1, b2, variant='standard', planar='exact', dtype=np.float64)
    t1, t2 = torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda()
    prev = S.get_arithmetic()
    try:
        got = {}
        for mode in ('fast', 'robust', 'reference'):
            S.set_arithmetic(mode)
            got[mode] = np.abs(S.sph2pob_standard_iou(t1, t2, is_aligned=True).cpu().numpy() - tru)
            pw = S.sph2pob_standard_iou(t1, t2).cpu().numpy()
            assert np.abs(np.diag(pw) - S.sph2pob_standard_iou(t1, t2, is_aligned=True).cpu().numpy()).max() == 0
        assert got['fast'].max() < 2e-5 and got['robust'].max() < 2e-5 and got['reference'].max() < 2e-5, got
        # the loss kernels carry the same safeguard: IoU-mode loss element = 1 - IoU on these pairs
        S.set_arithmetic('fast')
        le = S.Sph2PobIoULoss(mode='iou', reduction='none')(t1, t2).cpu().numpy()
        assert np.abs((1.0 - le) - tru).max() < 2e-5, (1.0 - le, tru)
        # 'robust' is an alias of the default now
        a = torch.from_numpy(O.generate_boxes(200000, 0)).cuda()
        b = torch.from_numpy(O.generate_boxes(200000, 1)).cuda()
        f = S.sph2pob_standard_iou(a, b, is_aligned=True)
        S.set_arithmetic('robust')
        r = S.sph2pob_standard_iou(a, b, is_aligned=True)
        assert torch.equal(f, r)
    finally:
        S.set_arithmetic(prev)
