"""-m gpu: a NaN coordinate must surface as NaN (the reference's torch.clamp chain propagates it: sph_iou_api.py:86,
:244-260; sph2pob_iou_loss.py through autograd) instead of being sanitised into IoU 0 / loss 1 by v_med3 / fmin / fmax.
+-inf in a clamped column is made finite by the reference's jitter clamps and must give a finite result here too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _boxes(n, dim, seed):
    from oracle import oracle as O
    return O.generate_boxes(n, seed, box='rbfov' if dim == 5 else 'bfov')


@pytest.mark.parametrize('arith', ['fast', 'reference'])
@pytest.mark.parametrize('dim', [4, 5])
def test_nan_rows_give_nan_iou_everywhere_else_unchanged(dim, arith):
    import torch
    import sph_retina_amd as S
    n = 20000
    b1, b2 = _boxes(n, dim, 1), _boxes(n, dim, 2)
    b2[: n // 2] = b1[: n // 2] + np.random.default_rng(0).standard_normal((n // 2, dim)).astype(np.float32)  # overlapping half
    b2[:, 1:4] = b2[:, 1:4].clip(1, 179)
    t1, t2 = torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda()
    prev = S.get_arithmetic()
    S.set_arithmetic(arith)
    try:
        for fn in (S.sph2pob_standard_iou, S.sph2pob_efficient_iou):
            clean = fn(t1, t2, is_aligned=True)
            assert torch.isfinite(clean).all()
            rows = torch.arange(0, n, 37, device='cuda')
            for col in range(dim):
                for which in (0, 1):
                    x1, x2 = t1.clone(), t2.clone()
                    (x1 if which == 0 else x2)[rows, col] = float('nan')
                    got = fn(x1, x2, is_aligned=True)
                    assert torch.isnan(got[rows]).all(), (fn.__name__, col, which)
                    mask = torch.ones(n, dtype=torch.bool, device='cuda')
                    mask[rows] = False
                    assert torch.equal(got[mask], clean[mask])
            # pairwise: a NaN row / a NaN column
            x1, x2 = t1[:8].clone(), t2[:300].clone()
            x1[3, 1] = float('nan')
            x2[17, 0] = float('nan')
            pw = fn(x1, x2)
            assert torch.isnan(pw[3]).all() and torch.isnan(pw[:, 17]).all()
            keep = torch.ones_like(pw, dtype=torch.bool)
            keep[3] = False
            keep[:, 17] = False
            assert torch.isfinite(pw[keep]).all()
        # +-inf in a clamped column: finite, equal to the value at the clamp bound
        x1 = t1.clone()
        x1[:100, 2] = float('inf')
        y1 = t1.clone()
        y1[:100, 2] = 180.0
        assert torch.equal(S.sph2pob_standard_iou(x1, t2, is_aligned=True), S.sph2pob_standard_iou(y1, t2, is_aligned=True))
    finally:
        S.set_arithmetic(prev)


@pytest.mark.parametrize('mode', ['iou', 'ciou'])
def test_nan_prediction_gives_nan_loss_and_gradient(mode):
    import torch
    import sph_retina_amd as S
    n = 4096
    tgt = torch.from_numpy(_boxes(n, 5, 3)).cuda()
    pred = (tgt + torch.randn_like(tgt) * 3).clamp(min=1)
    pred[:, 4] = tgt[:, 4] + 2
    pred[5, 2] = float('nan')
    pred.requires_grad_(True)
    el = S.Sph2PobIoULoss(mode=mode, reduction='none')(pred, tgt)
    assert torch.isnan(el[5]) and torch.isfinite(el[torch.arange(n, device='cuda') != 5]).all()
    tot = S.Sph2PobIoULoss(mode=mode)(pred, tgt)
    assert torch.isnan(tot)                      # the divergence is visible in the scalar the trainer logs
    el.sum().backward()
    assert torch.isnan(pred.grad[5]).all()
    ok = torch.arange(n, device='cuda') != 5
    assert torch.isfinite(pred.grad[ok]).all()
