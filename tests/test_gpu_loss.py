"""GPU: Sph2PobIoULoss (fused forward + hand-derived backward kernels) vs the reference fixtures (values AND
autograd gradients), the oracle, and f64 finite differences; config-3-size smoke with properties."""
import numpy as np
import pytest
import torch

from conftest import err_stats, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    import sph_retina_amd.losses as losses
    assert torch.cuda.is_available()
    return losses


@pytest.fixture(params=['fast', 'reference'])
def arith(request):
    import sph_retina_amd as S
    S.set_arithmetic(request.param)
    yield request.param
    S.set_arithmetic('fast')


def cu(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda().requires_grad_(grad)


@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
@pytest.mark.parametrize('mode', ['iou', 'giou', 'diou', 'ciou'])
def test_values_and_grads_vs_reference_fixture(L, box, mode, arith):
    g = load_golden('loss_' + box)
    pred, target = cu(g['pred'], True), cu(g['target'], True)
    loss = L.Sph2PobIoULoss(mode=mode, reduction='none')(pred, target)
    assert loss.shape == (400,)
    s = err_stats(loss.detach().cpu().numpy(), g['loss_' + mode])
    assert s['mean'] < 2e-6 and s['n4'] <= 2, s
    loss.sum().backward()
    for mine, ref in ((pred.grad, g['gpred_' + mode]), (target.grad, g['gtarget_' + mode])):
        d = np.abs(mine.cpu().numpy() - ref)
        scale = np.abs(ref).max()
        assert np.isfinite(mine.cpu().numpy()).all()
        assert np.median(d) < 1e-6 * scale and np.quantile(d, 0.99) < 2e-4 * scale and d.max() < 5e-3 * scale, \
            (np.median(d), np.quantile(d, 0.99), d.max(), scale)


@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
def test_reductions_weights_avg_factor(L, box, arith):
    g = load_golden('loss_' + box)
    pred, target = cu(g['pred']), cu(g['target'])
    w1, w2 = cu(g['w1']), cu(g['w2'])
    Lm = L.Sph2PobIoULoss(mode='ciou', reduction='mean', loss_weight=2.0)
    rt = dict(rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(Lm(pred, target).item(), g['mean_ciou'], **rt)
    np.testing.assert_allclose(Lm(pred, target, w1).item(), g['mean_ciou_w1'], **rt)
    np.testing.assert_allclose(Lm(pred, target, w2).item(), g['mean_ciou_w2'], **rt)
    np.testing.assert_allclose(Lm(pred, target, w2, avg_factor=123.0).item(), g['mean_ciou_w2_avg'], **rt)
    np.testing.assert_allclose(Lm(pred, target, w1, reduction_override='sum').item(), g['sum_ciou_w1'], **rt)
    with pytest.raises(ValueError):
        Lm(pred, target, w1, avg_factor=3.0, reduction_override='sum')
    with pytest.raises(AssertionError):
        Lm(pred, target, reduction_override='median')
    with pytest.raises(AssertionError):
        L.Sph2PobIoULoss(mode='siou')
    # gradient of the weighted, avg_factor-normalised mean (the RetinaNet loss_bbox call pattern)
    p = cu(g['pred'], True)
    Lm(p, target, w2, avg_factor=123.0).backward()
    ref = g['gpred_mean_ciou_w2_avg']
    d = np.abs(p.grad.cpu().numpy() - ref)
    assert np.median(d) < 1e-6 * np.abs(ref).max() and d.max() < 5e-3 * np.abs(ref).max()
    # all-zero weights: zero loss, zero grads (reference shortcut sph2pob_iou_loss.py:36-39)
    p = cu(g['pred'], True)
    z = Lm(p, target, torch.zeros_like(w1))
    z.backward()
    assert z.item() == 0.0 and float(p.grad.abs().max()) == 0.0
    # deterministic reduction: two evaluations are bitwise equal
    assert Lm(pred, target, w1).item() == Lm(pred, target, w1).item()


def test_registry_build_and_target_grad_optional(L):
    from sph_retina_amd.registry import LOSSES, LOSSES_IS_MMDET
    if not LOSSES_IS_MMDET:
        loss = LOSSES.build(dict(type='Sph2PobIoULoss', mode='ciou', loss_weight=1.0))
        assert isinstance(loss, L.Sph2PobIoULoss)
    g = load_golden('loss_rbfov')
    pred, target = cu(g['pred'], True), cu(g['target'], False)
    L.Sph2PobIoULoss(mode='diou')(pred, target).backward()
    assert pred.grad is not None and target.grad is None and pred.grad.shape == (400, 5)
    e = L.Sph2PobIoULoss()(pred[:0], target[:0], reduction_override='sum')
    assert e.item() == 0.0


@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
def test_grad_vs_fp64_finite_differences(L, oracle, box, arith):
    g = load_golden('loss_' + box)
    sl = slice(0, 150)
    for mode in ('iou', 'ciou'):
        pred, target = cu(g['pred'][sl], True), cu(g['target'][sl], True)
        L.Sph2PobIoULoss(mode=mode, reduction='sum')(pred, target).backward()
        fp, ft = oracle.loss_grad_fd(g['pred'][sl], g['target'][sl], mode=mode)
        for mine, fd in ((pred.grad, fp), (target.grad, ft)):
            d = np.abs(mine.cpu().numpy() - fd)
            scale = np.abs(fd).max()
            assert np.median(d) < 2e-4 * scale and (d > 0.05 * scale).mean() < 0.03


def test_config3_size_1m_rbfov_ciou_fwd_bwd(L, oracle):
    """BASELINE config 3: 1,000,000 RBFoV pairs, Sph2Pob + CIoU loss forward + backward."""
    n = 1_000_000
    tgt = oracle.generate_boxes(n, 0, box='rbfov', alpha=(5, 90), beta=(5, 90), gamma=(-60, 60))
    rng = np.random.default_rng(1)
    prd = tgt + rng.standard_normal(tgt.shape).astype(np.float32) * np.array([8, 8, 6, 6, 10], np.float32)
    prd[:, 0] %= 360
    prd[:, 1] = prd[:, 1].clip(1, 179)
    prd[:, 2:4] = prd[:, 2:4].clip(1, 170)
    pred, target = cu(prd, True), cu(tgt)
    loss = L.Sph2PobIoULoss(mode='ciou', reduction='mean')(pred, target)
    loss.backward()
    assert torch.isfinite(loss) and 0.0 < loss.item() < 2.0
    assert pred.grad.shape == (n, 5) and bool(torch.isfinite(pred.grad).all())
    # value against the oracle on a 50k sample, and a directional-derivative check on the whole batch
    idx = np.arange(0, n, 20)
    ref = oracle.loss_elements(prd[idx], tgt[idx], mode='ciou', nthreads=8)
    got = L.Sph2PobIoULoss(mode='ciou', reduction='none')(pred.detach()[idx], target[idx]).cpu().numpy()
    assert np.abs(got - ref).mean() < 2e-6
    # element-wise directional derivative over the whole batch (CIoU's alpha = [iou > 0.5] makes the loss itself
    # discontinuous, so a handful of elements that cross 0.5 or a jitter threshold within +-h are expected outliers)
    with torch.no_grad():
        v = torch.randn_like(pred)
        h = 2e-2
        Ln = L.Sph2PobIoULoss(mode='ciou', reduction='none')
        fd = ((Ln(pred + h * v, target).double() - Ln(pred - h * v, target).double()) / (2 * h)).cpu().numpy()
        an = (pred.grad.double() * n * v.double()).sum(1).cpu().numpy()   # undo the 1/n of the mean
    err = np.abs(fd - an)
    # fp32 forward noise (~1e-6 per element, see parity report) over 2h = 0.04 degrees => ~3e-5 per element
    tol = 3e-2 * np.abs(an) + 3e-4
    assert (err > tol).mean() < 0.02, ((err > tol).mean(), np.median(err), np.median(np.abs(an)))
    assert np.median(err) < 1e-4, np.median(err)


def test_sph2pob_transform_decorator_is_differentiable(L, oracle):
    """Any torch OBB loss body wrapped by Sph2PobTransfrom (sphdet/losses/sph2pob_transform.py:11-37) back-propagates
    to the spherical inputs through the HIP transform adjoint; checked against the oracle (values) and f64 finite
    differences (gradients)."""
    import torch.nn as nn

    @L.Sph2PobTransfrom()
    class SmoothPlanarL1(nn.Module):
        def forward(self, pred, target, weight=None):
            d = pred - target
            cols = torch.tensor([0, 2, 3, 4], device=d.device)       # y is a constant of the transform
            return torch.sqrt(d[:, cols] ** 2 + 1e-2).sum()

    g = load_golden('loss_rbfov')
    sl = slice(0, 200)
    pred, target = cu(g['pred'][sl], True), cu(g['target'][sl], True)
    loss = SmoothPlanarL1()(pred, target)
    loss.backward()
    p1, p2 = oracle.transform(g['pred'][sl], g['target'][sl], variant='standard', jitter=True, dtype=np.float64)
    d = (p1 - p2)[:, [0, 2, 3, 4]]
    np.testing.assert_allclose(loss.item(), np.sqrt(d ** 2 + 1e-2).sum(), rtol=2e-5)
    gd = np.zeros_like(p1)
    gd[:, [0, 2, 3, 4]] = d / np.sqrt(d ** 2 + 1e-2)
    want = oracle.transform_vjp_fd(g['pred'][sl], g['target'][sl], gd, -gd, variant='standard', jitter=True)
    for mine, ref in ((pred.grad, want[0]), (target.grad, want[1])):
        e = np.abs(mine.cpu().numpy() - ref)
        scale = np.abs(ref).max()
        assert np.median(e) < 2e-4 * scale and (e > 0.02 * scale).mean() < 0.02, (np.median(e), scale)
    # plain transforms are differentiable too (efficient variant, no jitter), degrees output keeps the graph
    from sph_retina_amd.iou import sph2pob_efficient
    p = cu(g['pred'][sl], True)
    o1, o2 = sph2pob_efficient(p, cu(g['target'][sl]), rbb_angle_version='deg')
    (o2[:, 0].sum() + o1[:, 4].sum()).backward()
    assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().sum()) > 0


def test_diff_calculator_is_autograd_capable(L):
    """`calculator='diff'` (a NameError in the reference at HEAD, sph_iou_api.py:14,81) returns the same IoU as
    'common' and carries gradients when asked to."""
    import sph_retina_amd as S
    g = load_golden('loss_bfov')
    p, t = cu(g['pred'], True), cu(g['target'])
    iou = S.sph2pob_standard_iou(p, t, is_aligned=True, calculator='diff')
    common = S.sph2pob_standard_iou(p.detach(), t, is_aligned=True)
    assert float((iou.detach() - common).abs().max()) < 1e-5
    (1 - iou).sum().backward()
    ref = g['gpred_iou']
    d = np.abs(p.grad.cpu().numpy() - ref)
    assert np.median(d) < 1e-6 * np.abs(ref).max() and d.max() < 5e-3 * np.abs(ref).max()


@pytest.mark.parametrize('wdim', [1, 4])
def test_sparse_weights_dense_head_pattern(L, oracle, wdim):
    """RetinaNet passes every anchor with weight 0 on the negatives (sph_retina_head.py:261-264): waves whose weights
    are all zero are skipped by the kernels — results must equal the dense computation times the weights."""
    n = 70001
    t = oracle.generate_boxes(n, 5, box='bfov', alpha=(5, 60), beta=(5, 60))
    p = t + np.random.default_rng(1).normal(0, 3, t.shape).astype(np.float32)
    p[:, 1] = np.clip(p[:, 1], 1, 179)
    p[:, 2:] = np.clip(p[:, 2:], 2, 100)
    w = np.zeros((n, wdim), np.float32)
    w[np.random.default_rng(2).integers(0, n, 150)] = 1.0   # isolated positives
    w[5000:5300] = 0.5                                        # a run of positives spanning several waves
    Ln = L.Sph2PobIoULoss(mode='ciou', reduction='none')
    pd = cu(p, True)
    dense = Ln(pd, cu(t))
    dense.backward(cu(w.mean(1) if wdim > 1 else w[:, 0]))
    ps = cu(p, True)
    sparse = Ln(ps, cu(t), cu(w) if wdim > 1 else cu(w[:, 0]))
    sparse.sum().backward()
    wv = cu(w.mean(1) if wdim > 1 else w[:, 0])
    assert torch.equal(sparse, dense.detach() * wv)
    assert torch.equal(ps.grad, pd.grad)
    assert float(ps.grad[w.reshape(n, -1).sum(1) == 0].abs().max()) == 0.0


@pytest.mark.parametrize('mode', ['log', 'linear', 'square'])
def test_sph_iou_loss_legacy_postmaps(L, mode):
    """SphIoULossLegacy = RotatedIoULoss post-map on the fused IoU (parity of the mmrotate post-map: unpinned)."""
    from sph_retina_amd.losses import SphIoULossLegacy
    g = load_golden('loss_rbfov')
    pred, target = cu(g['pred'], True), cu(g['target'])
    iou_ref = 1.0 - g['loss_iou'] if 'loss_iou' in g else None
    loss = SphIoULossLegacy(mode=mode, reduction='none')(pred, target)
    iou = (1.0 - L.Sph2PobIoULoss(mode='iou', reduction='none')(cu(g['pred']), target)).clamp(min=1e-6)
    want = {'log': -iou.log(), 'linear': 1 - iou, 'square': 1 - iou ** 2}[mode]
    assert torch.allclose(loss, want, rtol=1e-6, atol=1e-7)
    loss.sum().backward()
    assert torch.isfinite(pred.grad).all() and float(pred.grad.abs().max()) > 0
    # chain rule against the fused IoU-mode gradient
    p2 = cu(g['pred'], True)
    L.Sph2PobIoULoss(mode='iou', reduction='none')(p2, target).backward(
        {'log': 1.0 / iou, 'linear': torch.ones_like(iou), 'square': 2 * iou}[mode] * (iou > 1e-6))
    assert torch.allclose(pred.grad, p2.grad, rtol=1e-4, atol=1e-7)
    from sph_retina_amd.registry import build_loss
    assert isinstance(build_loss(dict(type='SphIoULossLegacy')), SphIoULossLegacy)


def test_mmrotate_loss_bodies_are_wrapped_when_mmrotate_is_present(L, monkeypatch):
    """Sph2PobGDLoss / Sph2PobKFLoss = mmrotate's GDLoss / KFLoss behind the Sph2PobTransfrom decorator (reference
    sph2pob_gd_loss.py:7-9, sph2pob_kf_loss.py:8-26).  mmrotate is absent here, so a stand-in module checks the wiring:
    the bodies receive planar (n, 5) boxes in radians, KFLoss gets the swapped decode arguments, gradients reach the
    spherical inputs through the fused transform backward."""
    import importlib
    import sys
    import types
    import torch.nn as nn
    seen = {}

    class GDLoss(nn.Module):
        def __init__(self, loss_type='gwd', **kw):
            super().__init__()
            self.loss_type = loss_type

        def forward(self, pred, target, weight=None, **kw):
            seen['gd'] = (pred.shape, target.shape)
            return ((pred - target) ** 2).sum(-1).mean()

    class KFLoss(nn.Module):
        def forward(self, pred, target, weight=None, pred_decode=None, targets_decode=None, **kw):
            seen['kf'] = (pred_decode is target, targets_decode is pred)
            return (pred - target).abs().sum(-1).mean()
    for name in ('mmrotate', 'mmrotate.models'):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    fake = types.ModuleType('mmrotate.models.losses')
    fake.GDLoss, fake.KFLoss = GDLoss, KFLoss
    monkeypatch.setitem(sys.modules, 'mmrotate.models.losses', fake)
    import sph_retina_amd.losses.sph2pob_mmrotate_losses as M
    M = importlib.reload(M)
    try:
        assert M.__all__ == ['Sph2PobGDLoss', 'Sph2PobKFLoss']
        g = load_golden('loss_bfov')
        pred, target = cu(g['pred'], True), cu(g['target'])
        loss = M.Sph2PobGDLoss(loss_type='kld')(pred, target)
        loss.backward()
        assert seen['gd'] == ((pred.size(0), 5), (pred.size(0), 5))
        assert torch.isfinite(pred.grad).all() and float(pred.grad.abs().max()) > 0
        p2 = cu(g['pred'], True)
        M.Sph2PobKFLoss()(p2, target).backward()
        assert seen['kf'] == (True, True) and torch.isfinite(p2.grad).all()
        from sph_retina_amd.registry import LOSSES
        assert LOSSES.get('Sph2PobGDLoss') is M.Sph2PobGDLoss
    finally:
        monkeypatch.undo()
        importlib.reload(M)


@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
@pytest.mark.parametrize('weighted', [False, True])
def test_c_abi_one_pass_and_two_pass_forms_agree(L, box, weighted):
    """The C ABI offers the training step in two forms: `sph2pob_loss_fwd_grad_f32` (loss + gradients in one pass, then
    `sph2pob_loss_grad_scale_f32`) — what the autograd Function launches — and the round-1 form `sph2pob_loss_fwd_sum_f32` /
    `sph2pob_loss_fwd_f32` + `sph2pob_loss_bwd_f32` (a backward kernel that recomputes the forward).  Same arithmetic:
    losses and gradients must be identical bit for bit, also for a non-unit upstream gradient, weights and 'none'."""
    from sph_retina_amd import _lib
    lib = _lib.lib()
    g = load_golden('loss_' + box)
    p, t = cu(g['pred']), cu(g['target'])
    n, dim = p.shape
    w = (torch.rand(n, device='cuda') > 0.3).float() * torch.rand(n, device='cuda') if weighted else None
    wp, wd = (w.data_ptr(), 1) if weighted else (None, 0)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(int(lib.sph2pob_loss_sum_workspace_floats(n)), device='cuda')
    for mode in range(4):
        s1, s2 = torch.empty((), device='cuda'), torch.empty((), device='cuda')
        e1, e2 = torch.empty(n, device='cuda'), torch.empty(n, device='cuda')
        gp1, gt1, gp2, gt2 = (torch.empty_like(p) for _ in range(4))
        assert lib.sph2pob_loss_fwd_grad_f32(p.data_ptr(), t.data_ptr(), wp, wd, 0.25, e1.data_ptr(), s1.data_ptr(), ws.data_ptr(),
                                             gp1.data_ptr(), gt1.data_ptr(), n, dim, mode, 1e-6, st) == 0
        assert lib.sph2pob_loss_fwd_f32(p.data_ptr(), t.data_ptr(), wp, wd, 0.25, e2.data_ptr(), None, n, dim, mode, 1e-6, st) == 0
        assert lib.sph2pob_loss_fwd_sum_f32(p.data_ptr(), t.data_ptr(), wp, wd, 0.25, s2.data_ptr(), ws.data_ptr(), n, dim, mode,
                                            1e-6, st) == 0
        assert torch.equal(e1, e2) and torch.equal(s1, s2)
        assert abs(float(s1) - float(e1.double().sum())) < 1e-4 * max(1.0, abs(float(s1)))
        # scalar upstream gradient (reduced loss)
        up = torch.full((), 0.7, device='cuda')
        o1, o2 = torch.empty_like(p), torch.empty_like(p)
        assert lib.sph2pob_loss_grad_scale_f32(gp1.data_ptr(), up.data_ptr(), 0, o1.data_ptr(), n, dim, st) == 0
        assert lib.sph2pob_loss_bwd_f32(p.data_ptr(), t.data_ptr(), wp, wd, up.data_ptr(), 0, 0.25, gp2.data_ptr(), gt2.data_ptr(),
                                        n, dim, mode, 1e-6, st) == 0
        assert torch.allclose(o1, gp2, rtol=2e-7, atol=0) and torch.equal(o1 == 0, gp2 == 0)   # (w g) x vs (g w) x: one rounding
        # per-element upstream gradients (reduction 'none')
        ue = torch.rand(n, device='cuda')
        assert lib.sph2pob_loss_grad_scale_f32(gt1.data_ptr(), ue.data_ptr(), 1, o1.data_ptr(), n, dim, st) == 0
        assert lib.sph2pob_loss_bwd_f32(p.data_ptr(), t.data_ptr(), wp, wd, ue.data_ptr(), 1, 0.25, gp2.data_ptr(), gt2.data_ptr(),
                                        n, dim, mode, 1e-6, st) == 0
        assert torch.allclose(o1, gt2, rtol=2e-7, atol=0)
    # argument validation
    assert lib.sph2pob_loss_fwd_grad_f32(p.data_ptr(), t.data_ptr(), None, 0, 1.0, None, None, None, None, None, n, dim, 0, 1e-6, st) == -1
    assert lib.sph2pob_loss_grad_scale_f32(gp1.data_ptr(), up.data_ptr(), 2, o1.data_ptr(), n, dim, st) == -3


@pytest.mark.parametrize('reduction', ['mean', 'none'])
def test_backward_twice_and_non_unit_upstream_gradients(L, reduction):
    """The first backward scales the stashed gradients IN PLACE (a no-op launch for an upstream gradient of exactly 1, the
    plain `loss.backward()`) and hands them to autograd; a second backward through the same node (retain_graph=True)
    recomputes with the two-pass kernel.  Both must equal an independent evaluation for any upstream gradient."""
    g = load_golden('loss_rbfov')
    tgt = cu(g['target'])

    def fresh(up):
        p = cu(g['pred'], True)
        out = L.Sph2PobIoULoss(mode='ciou', reduction=reduction)(p, tgt)
        (out * up).sum().backward()
        return p.grad.clone()
    n = tgt.size(0)
    ups = [1.0, 0.37, torch.linspace(0.5, 2.0, n).cuda() if reduction == 'none' else 3.0]
    want = [fresh(u) for u in ups]
    p = cu(g['pred'], True)
    out = L.Sph2PobIoULoss(mode='ciou', reduction=reduction)(p, tgt)
    for k, up in enumerate(ups):   # three backward passes through ONE forward
        p.grad = None
        (out * up).sum().backward(retain_graph=True)
        assert torch.allclose(p.grad, want[k], rtol=3e-7, atol=1e-12), k
    # the C entry itself, in place: unit gradient leaves the stash untouched, anything else scales it
    from sph_retina_amd import _lib
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    stash = torch.randn(1000, 5, device='cuda')
    keep = stash.clone()
    one, two = torch.ones((), device='cuda'), torch.full((), 2.0, device='cuda')
    assert lib.sph2pob_loss_grad_scale_f32(stash.data_ptr(), one.data_ptr(), 0, stash.data_ptr(), 1000, 5, st) == 0
    assert torch.equal(stash, keep)
    assert lib.sph2pob_loss_grad_scale_f32(stash.data_ptr(), two.data_ptr(), 0, stash.data_ptr(), 1000, 5, st) == 0
    assert torch.equal(stash, keep * 2)


def test_final_sum_round_boundaries_and_weight_means(L):
    """The one workgroup that adds the per-workgroup partials takes them 16 x 256 at a time (masked past the end): batch
    sizes whose partial counts sit on either side of a round (1, 255, 256, 257, 4095, 4096, 4097, 9000 partials) must give
    the float64 sum of the elements within float rounding, twice the same bits; and an (n, 4) weight is its row mean
    (sph2pob_iou_loss.py:48), equal to passing that mean as an (n,) weight."""
    from sph_retina_amd import _lib
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cpu').manual_seed(7)
    nmax = 9000 * 256
    u = torch.rand((nmax, 4), generator=g)
    tgt = torch.stack([u[:, 0] * 360, 10 + u[:, 1] * 160, 5 + u[:, 2] * 60, 5 + u[:, 3] * 60], 1).cuda()
    pred = (tgt + torch.randn((nmax, 4), generator=g).cuda() * 3).contiguous()
    pred[:, 1].clamp_(1, 179)
    pred[:, 2:].clamp_(1, 120)
    ws = torch.empty(int(lib.sph2pob_loss_sum_workspace_floats(nmax)), device='cuda')
    for nb in (1, 255, 256, 257, 4095, 4096, 4097, 9000):
        n = nb * 256 - 3
        e = torch.empty(n, device='cuda')
        s1, s2, s3 = (torch.empty((), device='cuda') for _ in range(3))
        gp = torch.empty((n, 4), device='cuda')
        assert lib.sph2pob_loss_fwd_f32(pred.data_ptr(), tgt.data_ptr(), None, 0, 1.0, e.data_ptr(), None, n, 4, 3, 1e-6, st) == 0
        assert lib.sph2pob_loss_fwd_sum_f32(pred.data_ptr(), tgt.data_ptr(), None, 0, 1.0, s1.data_ptr(), ws.data_ptr(), n, 4, 3, 1e-6, st) == 0
        assert lib.sph2pob_loss_fwd_sum_f32(pred.data_ptr(), tgt.data_ptr(), None, 0, 1.0, s2.data_ptr(), ws.data_ptr(), n, 4, 3, 1e-6, st) == 0
        assert lib.sph2pob_loss_fwd_grad_f32(pred.data_ptr(), tgt.data_ptr(), None, 0, 1.0, None, s3.data_ptr(), ws.data_ptr(),
                                             gp.data_ptr(), None, n, 4, 3, 1e-6, st) == 0
        ref = float(e.double().sum())
        assert torch.equal(s1, s2) and torch.equal(s1, s3), nb
        assert abs(float(s1) - ref) < 2e-6 * ref, (nb, float(s1), ref)
    n = 70001
    w4 = torch.rand((n, 4), device='cuda') * (torch.rand((n, 1), device='cuda') > 0.5)
    # the reference widens an (n, 4) weight to five columns with the row mean and takes the mean again (sph2pob_transform.py:32-34)
    m = w4.mean(1)
    w1 = ((w4.sum(1) + m) / 5.0).contiguous()
    ea, eb = torch.empty(n, device='cuda'), torch.empty(n, device='cuda')
    assert lib.sph2pob_loss_fwd_f32(pred.data_ptr(), tgt.data_ptr(), w4.data_ptr(), 4, 1.0, ea.data_ptr(), None, n, 4, 3, 1e-6, st) == 0
    assert lib.sph2pob_loss_fwd_f32(pred.data_ptr(), tgt.data_ptr(), w1.data_ptr(), 1, 1.0, eb.data_ptr(), None, n, 4, 3, 1e-6, st) == 0
    assert torch.allclose(ea, eb, rtol=1e-6, atol=0) and torch.equal(ea == 0, eb == 0)   # torch's mean adds in another order: a few ulps
