"""Sph2PobL1Loss: fixtures tests/golden/l1.npz come from the reference's real, decorator-wrapped forward
(sphdet/losses/sph2pob_l1_loss.py:28-37 on the vendored mmdet L1Loss), fp32 and fp64 runs, with autograd gradients.

Tolerance: the element losses inherit the fp32 conditioning of the Sph2Pob transform — the reference's own fp32 run is
up to 2.7e-4 away from its fp64 run on these fixtures (values up to 25).  Criteria are therefore: median error at the
last-bit level and every error within the reference's own fp32-vs-fp64 deviation plus 1e-5 of the value scale."""
import numpy as np
import pytest
import torch

from conftest import load_golden

CFGS = [('enc', dict(encode=True, swap=False, angle_modifier='original')),
        ('swapmod', dict(encode=True, swap=True, angle_modifier='modulus')),
        ('raw', dict(encode=False, swap=False, angle_modifier='original'))]


def within_reference_noise(mine, ref32, ref64, what):
    mine, ref32, ref64 = (np.asarray(a, np.float64) for a in (mine, ref32, ref64))
    noise = np.abs(ref32 - ref64)
    scale = np.abs(ref64).max()
    d32, d64 = np.abs(mine - ref32), np.abs(mine - ref64)
    assert np.median(d32) <= 2e-7 * scale, (what, np.median(d32))
    # close to the fp32 fixture, or at least as close to the truth as the fixture's own noise level
    assert (np.minimum(d32, d64) <= 1e-5 * scale + 2 * noise.max()).all(), (what, np.minimum(d32, d64).max(), noise.max())
    assert (d32 > 1e-5 * scale).mean() <= max(2 * (noise > 1e-5 * scale).mean(), 0.01), what


@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
@pytest.mark.parametrize('name,cfg', CFGS)
def test_restatement_matches_reference_elements(oracle, box, name, cfg):
    g = load_golden('l1')
    e = oracle.obb_l1_elements(g[box + '_pred'], g[box + '_target'], cfg['encode'], cfg['swap'], cfg['angle_modifier'])
    within_reference_noise(e, g[f'{box}_{name}_elements'], g[f'{box}_{name}_elements64'], (box, name))
    if not cfg['encode']:   # no float32 cast inside: the float64 run pins the restatement tightly
        e64 = oracle.obb_l1_elements(g[box + '_pred'], g[box + '_target'], False, dtype=np.float64)
        assert np.abs(e64 - g[f'{box}_{name}_elements64']).max() < 1e-10


def test_l1_registry_and_asserts():
    import sph_retina_amd as S
    from sph_retina_amd.losses import Sph2PobL1Loss
    from sph_retina_amd.registry import build_loss
    loss = build_loss(dict(type='Sph2PobL1Loss', loss_weight=2.0))
    assert isinstance(loss, Sph2PobL1Loss) and loss.encode and not loss.swap and loss.loss_weight == 2.0
    with pytest.raises(AssertionError):
        Sph2PobL1Loss(angle_modifier='wrap')         # sph2pob_l1_loss.py:20
    with pytest.raises(AssertionError):
        loss(torch.zeros(2, 4), torch.zeros(2, 4), reduction_override='median')
    cpu = loss(torch.rand(2, 4) * 50 + 20, torch.rand(2, 4) * 50 + 20)     # CPU tensors: the product's host twins
    assert cpu.device.type == 'cpu' and torch.isfinite(cpu)
    assert S.__version__


@pytest.mark.gpu
@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
@pytest.mark.parametrize('name,cfg', CFGS)
def test_gpu_l1_values_grads_reductions(box, name, cfg):
    l1_fixture_on('cuda', box, name, cfg)


@pytest.mark.parametrize('box', ['bfov', 'rbfov'])
@pytest.mark.parametrize('name,cfg', CFGS)
def test_cpu_twin_l1_values_grads_reductions(box, name, cfg):
    """The same criteria on CPU tensors: transform + adjoint and the L1 rows from libsph2pob_host.so."""
    l1_fixture_on('cpu', box, name, cfg)


def l1_fixture_on(device, box, name, cfg):
    from sph_retina_amd.losses import Sph2PobL1Loss

    def cu(a, grad=False):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device).requires_grad_(grad)
    g = load_golden('l1')
    k = f'{box}_{name}_'
    pred, target = cu(g[box + '_pred'], True), cu(g[box + '_target'], True)
    loss = Sph2PobL1Loss(**cfg)
    el = loss(pred, target, reduction_override='none')
    assert el.shape == (pred.size(0), 5)
    within_reference_noise(el.detach().cpu().numpy(), g[k + 'elements'], g[k + 'elements64'], k)
    (el * cu(g[box + '_gout'])).sum().backward()
    for mine, ref in ((pred.grad, g[k + 'gpred']), (target.grad, g[k + 'gtarget'])):
        m = mine.cpu().numpy()
        d, scale = np.abs(m - ref), np.abs(ref).max()
        assert np.isfinite(m).all()
        assert np.median(d) < 2e-6 * scale and np.quantile(d, 0.99) < 1e-3 * scale and d.max() < 3e-2 * scale, \
            (k, np.median(d), np.quantile(d, 0.99), d.max(), scale)
    p, t, w = cu(g[box + '_pred']), cu(g[box + '_target']), cu(g[box + '_weight'])
    rt = dict(rtol=5e-5, atol=1e-6)
    np.testing.assert_allclose(loss(p, t).item(), g[k + 'mean'], **rt)
    np.testing.assert_allclose(Sph2PobL1Loss(loss_weight=2.0, **cfg)(p, t, w, avg_factor=97.0).item(), g[k + 'mean_w_avg'], **rt)
    np.testing.assert_allclose(loss(p, t, w, reduction_override='sum').item(), g[k + 'sum_w'], **rt)
    z = loss(cu(np.zeros((0, p.size(1)), np.float32)), cu(np.zeros((0, p.size(1)), np.float32)))
    assert z.item() == 0.0
