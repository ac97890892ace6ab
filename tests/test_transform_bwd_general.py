"""Backward of the transforms without a closed-form adjoint — sph2pob_legacy and rbb_angle='project' (round-1 VERDICT
missing #3: `Sph2PobTransfrom('sph2pob_legacy')` is accepted by the reference and differentiated by torch autograd,
sphdet/losses/sph2pob_transform.py:12-16) — against the reference's own autograd (tests/golden/transform_bwd.npz, generated
by oracle/gen_goldens.py transform_bwd from the unmodified reference, float32 and float64).

The kernels differentiate the reference-order transform in forward mode on (value, derivative) pairs, so the only
differences from torch are roundings; the derivative of acos(clamp(.)) is 1 / sin(angle), which amplifies them where
an angle is within ~1e-3 of 0 or pi — criteria are therefore quantiles against the float64 gradients, bounded by how far
the reference's own float32 gradients are from them."""
import numpy as np
import pytest

from conftest import load_golden

CASES = [('legacy', 'bfov', 'legacy', 'equator'), ('standard_project', 'bfov', 'standard', 'project'),
         ('standard_project', 'rbfov', 'standard', 'project'), ('efficient_project', 'bfov', 'efficient', 'project'),
         ('efficient_project', 'rbfov', 'efficient', 'project')]


def _check(mine, ref32, ref64, what):
    fin = np.isfinite(ref64) & np.isfinite(ref32)
    assert fin.mean() > 0.95, what
    assert np.isfinite(mine[fin]).all(), what
    scale = np.quantile(np.abs(ref64[fin]), 0.9) + 1e-12
    d, noise = np.abs(mine - ref64)[fin] / scale, np.abs(ref32 - ref64)[fin] / scale
    assert np.median(d) < 1e-5, (what, np.median(d))
    assert np.quantile(d, 0.99) <= max(1e-3, 4 * np.quantile(noise, 0.99)), (what, np.quantile(d, 0.99), np.quantile(noise, 0.99))


@pytest.mark.parametrize('name,box,variant,angle', CASES)
def test_dual_number_adjoint_on_host_vs_reference_autograd(host_harness, name, box, variant, angle):
    g = load_golden('transform_bwd')
    for dist in ('uni', 'near'):
        for edge in ('arc', 'chord'):
            k = f'{name}_{box}_{dist}_{edge}_'
            o1, o2 = host_harness.transform_bwd_general(g[k + 'b1'], g[k + 'b2'], g[k + 'go1'], g[k + 'go2'], variant=variant,
                                                        edge=edge, angle=angle)
            _check(o1, g[k + 'g1'], g[k + 'g1_64'], k + 'g1')
            _check(o2, g[k + 'g2'], g[k + 'g2_64'], k + 'g2')


@pytest.mark.gpu
@pytest.mark.parametrize('name,box,variant,angle', CASES)
def test_gpu_transform_backward_legacy_and_project(name, box, variant, angle):
    import torch
    import sph_retina_amd.iou as I
    fn = {'legacy': I.sph2pob_legacy, 'standard': I.sph2pob_standard, 'efficient': I.sph2pob_efficient}[variant]
    g = load_golden('transform_bwd')
    for dist in ('uni', 'near'):
        for edge in ('arc', 'chord'):
            k = f'{name}_{box}_{dist}_{edge}_'
            b1 = torch.from_numpy(g[k + 'b1']).cuda().requires_grad_(True)
            b2 = torch.from_numpy(g[k + 'b2']).cuda().requires_grad_(True)
            kw = {} if variant == 'legacy' else dict(rbb_angle=angle)
            p1, p2 = fn(b1, b2, rbb_angle_version='rad', rbb_edge=edge, **kw)
            # forward values: the reference's planar boxes
            for mine, ref in ((p1, g[k + 'p1']), (p2, g[k + 'p2'])):
                d = np.abs(mine.detach().cpu().numpy() - ref)
                assert np.nanmedian(d) < 1e-6 and np.nanquantile(d, 0.99) < 1e-3, (k, np.nanmedian(d))
            ((p1 * torch.from_numpy(g[k + 'go1']).cuda()).sum() + (p2 * torch.from_numpy(g[k + 'go2']).cuda()).sum()).backward()
            _check(b1.grad.cpu().numpy(), g[k + 'g1'], g[k + 'g1_64'], k + 'g1')
            _check(b2.grad.cpu().numpy(), g[k + 'g2'], g[k + 'g2_64'], k + 'g2')


def test_jittered_legacy_adjoint_on_host_vs_reference_autograd(host_harness):
    """The form Sph2PobTransfrom('sph2pob_legacy') differentiates: jitter -> transform -> jitter."""
    g = load_golden('transform_bwd')
    for dist in ('uni', 'near'):
        k = f'legacyjit_bfov_{dist}_arc_'
        o1, o2 = host_harness.transform_bwd_general(g[k + 'b1'], g[k + 'b2'], g[k + 'go1'], g[k + 'go2'], variant='legacy',
                                                    jitter=True)
        _check(o1, g[k + 'g1'], g[k + 'g1_64'], k + 'g1')
        _check(o2, g[k + 'g2'], g[k + 'g2_64'], k + 'g2')


@pytest.mark.gpu
def test_gpu_sph2pob_transfrom_decorator_with_legacy_transform():
    """`@Sph2PobTransfrom('sph2pob_legacy')` around a torch loss body: planar boxes as the reference's, gradients reach
    the spherical inputs (round 1 raised NotImplementedError in backward)."""
    import torch
    from sph_retina_amd.losses.sph2pob_transform import Sph2PobTransfrom
    g = load_golden('transform_bwd')

    @Sph2PobTransfrom('sph2pob_legacy')
    class Body(torch.nn.Module):
        def forward(self, pred, target, weight=None):
            return pred, target

    for dist in ('uni', 'near'):
        k = f'legacyjit_bfov_{dist}_arc_'
        b1 = torch.from_numpy(g[k + 'b1']).cuda().requires_grad_(True)
        b2 = torch.from_numpy(g[k + 'b2']).cuda().requires_grad_(True)
        p1, p2 = Body()(b1, b2)
        for mine, ref in ((p1, g[k + 'p1']), (p2, g[k + 'p2'])):
            d = np.abs(mine.detach().cpu().numpy() - ref)
            assert np.nanmedian(d) < 1e-6 and np.nanquantile(d, 0.98) < 1e-3, (k, np.nanmedian(d), np.nanquantile(d, 0.98))
        ((p1 * torch.from_numpy(g[k + 'go1']).cuda()).sum() + (p2 * torch.from_numpy(g[k + 'go2']).cuda()).sum()).backward()
        _check(b1.grad.cpu().numpy(), g[k + 'g1'], g[k + 'g1_64'], k + 'g1')
        _check(b2.grad.cpu().numpy(), g[k + 'g2'], g[k + 'g2_64'], k + 'g2')
