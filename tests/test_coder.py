"""Box coders: CPU half — the numpy restatement against the reference's outputs
(tests/golden/coder.npz, generated from sphdet/bbox/coder/*.py by oracle/gen_goldens.py coder) — and the GPU half —
the HIP kernels through the C ABI against those fixtures and the restatement.

Tolerance: fp32, one rounding of exp / log apart (libm vs Sleef vs ocml): |d| <= 2e-6 * max(1, |value|) for encode and
4e-7 relative to the box range (360 deg) for decode; gradients 2e-6 relative."""
import numpy as np
import pytest
import torch

from conftest import load_golden

CASES = [(4, 'DeltaXYWHSphBBoxCoder'), (5, 'DeltaXYWHASphBBoxCoder')]


def close(a, b, rtol, atol, what=''):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, what
    fin = np.isfinite(b)
    assert (np.isfinite(a) == fin).all(), what
    err = np.abs(a - b)[fin] - rtol * np.abs(b)[fin]
    assert err.size == 0 or err.max() <= atol, (what, err.max())


@pytest.mark.parametrize('dim,cls', CASES)
def test_restatement_matches_reference_outputs(oracle, dim, cls):
    g = load_golden('coder')
    k = f'd{dim}_'
    a, gt, m, s = g[k + 'anchors'], g[k + 'gts'], g[k + 'means'], g[k + 'stds']
    close(oracle.coder_encode(a, gt, m, s), g[k + 'enc'], 2e-6, 2e-6, 'encode')
    dec, gd = oracle.coder_decode(a, g[k + 'deltas'], m, s, grad_boxes=g[k + 'gout'])
    close(dec, g[k + 'dec'], 4e-7, 1e-6, 'decode')
    close(gd, g[k + 'gdeltas'], 2e-6, 1e-6, 'decode grad')
    dec, gd = oracle.coder_decode(a, g[k + 'deltas'], m, s, wh_ratio_clip=0.05, clip_border=False, add_ctr_clamp=True,
                                  ctr_clamp=6, grad_boxes=g[k + 'gout'])
    close(dec, g[k + 'dec_ctr'], 4e-7, 1e-6, 'decode ctr')
    close(gd, g[k + 'gdeltas_ctr'], 2e-6, 1e-6, 'decode ctr grad')
    close(oracle.coder_decode(a[:50], g[k + 'deltas_mc'], m, s, box_dim=dim), g[k + 'dec_mc'], 4e-7, 1e-6, 'multiclass')
    close(oracle.coder_decode(a, g[k + 'deltas']), g[k + 'dec_plain'], 4e-7, 1e-6, 'default means/stds')
    # the clamps really fire in the fixture
    assert (g[k + 'gdeltas'] == 0).any() and (g[k + 'dec'][:, 2:4] >= 179.9).any()


def test_coder_registry_and_asserts():
    import sph_retina_amd as S
    from sph_retina_amd.registry import BBOX_CODERS, build_bbox_coder
    assert 'DeltaXYWHSphBBoxCoder' in BBOX_CODERS or hasattr(BBOX_CODERS, 'get')
    coder = build_bbox_coder(dict(type='DeltaXYWHASphBBoxCoder', target_stds=(1., 1., 1., 1., 1.)))
    assert isinstance(coder, S.DeltaXYWHASphBBoxCoder) and coder.box_dim == 5
    enc = coder.encode(torch.ones(2, 5), torch.ones(2, 5))          # CPU tensors: the product's host twin
    assert enc.device.type == 'cpu' and torch.equal(enc, torch.zeros(2, 5))
    with pytest.raises(AssertionError):
        S.DeltaXYWHSphBBoxCoder().encode(torch.zeros(2, 5), torch.ones(2, 5))
    empty = S.DeltaXYWHSphBBoxCoder().decode(torch.zeros(0, 4), torch.zeros(0, 4))
    assert empty.shape == (0, 4)


def cu(a, device='cuda'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.gpu
@pytest.mark.parametrize('dim,cls', CASES)
def test_gpu_coder_matches_reference_outputs(dim, cls):
    coder_fixture_on(dim, cls, 'cuda')


@pytest.mark.parametrize('dim,cls', CASES)
def test_cpu_twin_coder_matches_reference_outputs(dim, cls):
    """CPU tensors (the reference's coders are plain torch and run wherever the boxes live): libsph2pob_host.so computes the
    rows with the very functions of csrc/sph2pob_coder.hpp the kernels run."""
    coder_fixture_on(dim, cls, 'cpu')


def coder_fixture_on(dim, cls, device):
    import functools
    import sph_retina_amd.bbox.coder as C
    cu_dev = functools.partial(cu, device=device)
    return _coder_fixture(C, cu_dev, dim, cls)


def _coder_fixture(C, cu, dim, cls):
    g = load_golden('coder')
    k = f'd{dim}_'
    a, gt, m, s = cu(g[k + 'anchors']), cu(g[k + 'gts']), tuple(g[k + 'means']), tuple(g[k + 'stds'])
    coder = getattr(C, cls)(target_means=m, target_stds=s)
    a0 = a.clone()
    close(coder.encode(a, gt).cpu().numpy(), g[k + 'enc'], 2e-6, 2e-6, 'encode')
    d = cu(g[k + 'deltas']).requires_grad_(True)
    dec = coder.decode(a, d)
    close(dec.detach().cpu().numpy(), g[k + 'dec'], 4e-7, 1e-6, 'decode')
    (dec * cu(g[k + 'gout'])).sum().backward()
    close(d.grad.cpu().numpy(), g[k + 'gdeltas'], 2e-6, 1e-6, 'decode grad')
    ctr = getattr(C, cls)(target_means=m, target_stds=s, add_ctr_clamp=True, ctr_clamp=6, clip_border=False)
    d2 = cu(g[k + 'deltas']).requires_grad_(True)
    dec2 = ctr.decode(a, d2, wh_ratio_clip=0.05)
    close(dec2.detach().cpu().numpy(), g[k + 'dec_ctr'], 4e-7, 1e-6, 'decode ctr')
    (dec2 * cu(g[k + 'gout'])).sum().backward()
    close(d2.grad.cpu().numpy(), g[k + 'gdeltas_ctr'], 2e-6, 1e-6, 'decode ctr grad')
    close(C.delta2bbox(a[:50], cu(g[k + 'deltas_mc']), m, s, box_dim=dim).cpu().numpy(), g[k + 'dec_mc'], 4e-7, 1e-6, 'mc')
    close(C.delta2bbox(a, cu(g[k + 'deltas'])).cpu().numpy(), g[k + 'dec_plain'], 4e-7, 1e-6, 'plain')
    assert torch.equal(a, a0)                                             # inputs never written


@pytest.mark.gpu
@pytest.mark.parametrize('dim', [4, 5])
def test_gpu_coder_random_vs_restatement_and_roundtrip(oracle, dim):
    import sph_retina_amd.bbox.coder as C
    rng = np.random.default_rng(dim)
    n = 100003                                                            # ragged tail block
    box = 'bfov' if dim == 4 else 'rbfov'
    anchors = oracle.generate_boxes(n, 11, box=box)
    gts = oracle.generate_boxes(n, 12, box=box)
    m = (0.0, 0.0, 0.0, 0.0, 0.0)[:dim]
    s = (0.1, 0.1, 0.2, 0.2, 0.1)[:dim]
    enc = C.bbox2delta(cu(anchors), cu(gts), m, s)
    close(enc.cpu().numpy(), oracle.coder_encode(anchors, gts, m, s), 2e-6, 2e-6, 'encode')
    back = C.delta2bbox(cu(anchors), enc, m, s, clip_border=False, wh_ratio_clip=1e-9)
    close(back.cpu().numpy(), gts, 0, 2e-4, 'round trip')                 # decode(encode(gt)) == gt
    deltas = (rng.standard_normal((n, dim)) * 2).astype(np.float32)
    deltas[::97, 0] = np.nan                                              # NaN propagates like torch.clamp
    gout = rng.standard_normal((n, dim)).astype(np.float32)
    d = cu(deltas).requires_grad_(True)
    dec = C.delta2bbox(cu(anchors), d, m, s)
    ref, gref = oracle.coder_decode(anchors, deltas, m, s, grad_boxes=gout)
    close(dec.detach().cpu().numpy(), ref, 4e-7, 1e-6, 'decode')
    dec.backward(cu(gout))
    close(d.grad.cpu().numpy(), gref, 2e-6, 1e-6, 'grad')


@pytest.mark.gpu
def test_gpu_decode_feeds_loss_and_gradients_reach_deltas(oracle):
    """reg_decoded_bbox=True pattern (sph_retina_head.py:255-264): deltas -> decode -> Sph2PobIoULoss -> backward."""
    import sph_retina_amd as S
    n = 4096
    anchors = oracle.generate_boxes(n, 21, box='bfov', alpha=(5, 60), beta=(5, 60))
    gts = anchors + np.random.default_rng(0).normal(0, 2, anchors.shape).astype(np.float32)
    gts[:, 1] = np.clip(gts[:, 1], 1, 179)
    gts[:, 2:] = np.clip(gts[:, 2:], 2, 100)
    coder = S.DeltaXYWHSphBBoxCoder(target_stds=(0.1, 0.1, 0.2, 0.2))
    deltas = torch.zeros(n, 4, device='cuda', requires_grad=True)
    loss = S.Sph2PobIoULoss(mode='ciou')(coder.decode(cu(anchors), deltas), cu(gts))
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(deltas.grad).all() and deltas.grad.abs().max() > 0
    # one gradient step along -grad lowers the loss
    with torch.no_grad():
        stepped = deltas - 0.5 * deltas.grad / deltas.grad.abs().max()
    assert S.Sph2PobIoULoss(mode='ciou')(coder.decode(cu(anchors), stepped), cu(gts)) < loss


@pytest.mark.gpu
@pytest.mark.parametrize('dim,cls', CASES)
def test_gpu_batched_decode_equals_per_image_decode(dim, cls):
    """(B, N, d) anchors with (B, N, C*d) deltas in one launch (the reference's decode raises there,
    delta_xywh_sph_bbox_coder.py:104): identical to decoding image by image, gradients included."""
    import sph_retina_amd as S
    g = load_golden('coder')
    k = f'd{dim}_'
    coder = getattr(S, cls)(target_means=tuple(g[k + 'means']), target_stds=tuple(g[k + 'stds']))
    a = torch.from_numpy(g[k + 'anchors']).cuda()
    d = torch.from_numpy(g[k + 'deltas']).cuda()
    n = (a.size(0) // 3) * 3
    ab, db = a[:n].reshape(3, n // 3, dim), d[:n].reshape(3, n // 3, dim).clone().requires_grad_(True)
    out = coder.decode(ab, db)
    assert out.shape == (3, n // 3, dim)
    per = torch.stack([coder.decode(ab[i], db[i].detach()) for i in range(3)])
    assert torch.equal(out.detach(), per)
    out.sum().backward()
    d2 = d[:n].clone().requires_grad_(True)
    coder.decode(a[:n], d2).sum().backward()
    assert torch.equal(db.grad.reshape(n, dim), d2.grad)
