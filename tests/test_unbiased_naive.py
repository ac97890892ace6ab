"""Unbiased IoU and Naive IoU (SphOverlaps2D's default backend and SphNMS's other two calculators,
sphdet/bbox/nms/sph_nms.py:9-14).

Fixtures: tests/golden/unbiased.npz = the reference's `unbiased_iou` (sphdet/iou/sph_iou_api.py:103-126) on float32
tensors (`*_iou32`, its own mixed float32/float64 numpy arithmetic) and on float64 tensors (`*_iou64`).

What can and cannot be matched.  The reference computes the two box areas in float32 (4*acos(-sin*sin) - 2*pi cancels
catastrophically for small boxes) — its float32 output is up to 4.8e-4 away from its own float64 output on these
fixtures, and its vertex test `np.round(dot, 8) >= 0` is decided by rounding noise whenever two edge planes coincide
(integer-degree boxes that share phi +- beta/2).  So:
  * the C restatement in float64 must reproduce `*_iou64` EXACTLY (it does: max |d| = 0);
  * the HIP kernel (fp32 jitter + deg2rad as the reference applies to fp32 tensors, then double) must agree with the
    float64 fixtures to 5e-6 except on the coincident-plane pairs (<= 2 per 4412), and with the float32 fixtures to
    within the reference's own float32 noise;
  * the reference-arithmetic mode must track the float32 fixtures at least as closely as the reference tracks itself.
Naive IoU: mmcv-full 1.6.0 is absent, `bbox_overlaps` is restated from its published kernel — parity unpinned for the
BFoV branch; the RBFoV branch is checked against the planar restatement that the Sph2Pob goldens pin."""
import numpy as np
import pytest
import torch

from conftest import load_golden

BOXES = ['bfov', 'rbfov']


def structured_pairs(oracle, dim, n=4000, seed=3):
    box = 'bfov' if dim == 4 else 'rbfov'
    b1 = oracle.generate_boxes(n, seed, box=box)
    b2 = b1 + np.random.default_rng(seed).normal(0, 4, b1.shape).astype(np.float32)
    b2[:, 0] %= 360
    b2[:, 1] = np.clip(b2[:, 1], 1, 179)
    b2[:, 2:4] = np.clip(b2[:, 2:4], 1, 170)
    b2[:50] = b1[:50]                                   # identical
    if dim == 5:
        b2[50:150, 4] = b1[50:150, 4]                   # parallel
        b2[150:200, 4] = b1[150:200, 4] - 90            # perpendicular
    far = oracle.generate_boxes(n, seed + 1, box=box)   # mostly disjoint
    return np.concatenate([b1, b1]), np.concatenate([b2, far])


@pytest.mark.parametrize('box', BOXES)
def test_restatement_float64_is_exact_and_float32_modes_within_reference_noise(oracle, box):
    g = load_golden('unbiased')
    b1, b2, r32, r64 = g[box + '_b1'], g[box + '_b2'], g[box + '_iou32'], g[box + '_iou64']
    assert np.array_equal(oracle.unbiased_iou(b1, b2, prec='f64'), r64)          # bit-exact restatement
    assert np.array_equal(oracle.unbiased_iou(g[box + '_pa'], g[box + '_pb'], is_aligned=False, prec='f64'),
                          g[box + '_pw64'])
    noise = np.abs(r32.astype(np.float64) - r64)
    assert noise.max() > 1e-4                                                     # the reference's own fp32 noise
    k = oracle.unbiased_iou(b1, b2, prec='kernel')
    d = np.abs(k - r64)
    assert (d > 5e-6).sum() <= 2 and np.median(d) < 1e-7, ((d > 5e-6).sum(), d.max())
    assert (np.abs(k - r32) > noise + 5e-6).sum() <= 2
    e = oracle.unbiased_iou(b1, b2, prec='reference_f32')
    de = np.abs(e - r32)
    assert (de > 1e-5).sum() < (noise > 1e-5).sum() and np.median(de) < 2.5e-7 and de.max() < 2e-2   # one chaotic pair may land anywhere


@pytest.mark.parametrize('dim', [4, 5])
def test_device_math_on_host_matches_restatement(oracle, host_harness, dim):
    g = load_golden('unbiased')
    box = 'bfov' if dim == 4 else 'rbfov'
    for b1, b2 in ((g[box + '_b1'], g[box + '_b2']), structured_pairs(oracle, dim)):
        h = host_harness.extra_iou(b1, b2, 'unbiased')
        k = oracle.unbiased_iou(b1, b2, prec='kernel')
        d = np.abs(h - k)
        assert (d > 1e-6).sum() <= 1 and np.median(d) == 0, ((d > 1e-6).sum(), d.max())   # cull / division-free test
        assert ((h >= 0) & (h <= 1)).all()
        hr = host_harness.extra_iou(b1, b2, 'unbiased_ref')
        e = oracle.unbiased_iou(b1, b2, prec='reference_f32')
        assert (np.abs(hr - e) > 1e-4).sum() <= 3 and np.median(np.abs(hr - e)) < 2.5e-7
        hn = host_harness.extra_iou(b1, b2, 'naive')
        on = oracle.naive_iou(b1, b2)
        assert np.abs(hn - on).max() < 1e-5, np.abs(hn - on).max()
    # identical boxes: naive exactly 1; unbiased (jittered apart by 3 eps) ~ 1 except where a vertex lands inside the
    # reference's 5e-9 rounding tolerance (np.round(dot, 8) >= 0) and the vertex set becomes inconsistent: 0.4 % of
    # identical pairs in the reference's own float64 arithmetic (measured with the restatement on 100 k pairs)
    b1, b2 = structured_pairs(oracle, dim)
    assert (host_harness.extra_iou(b1[:50], b2[:50], 'unbiased') > 0.999).sum() >= 48
    np.testing.assert_allclose(host_harness.extra_iou(b1[:50], b2[:50], 'naive'), 1.0, atol=1e-6)


def test_api_asserts_and_backends():
    import sph_retina_amd as S
    from sph_retina_amd.iou import naive_iou, unbiased_iou
    a, b = torch.rand(3, 4), torch.rand(3, 4)
    for fn in (unbiased_iou, naive_iou):
        with pytest.raises(AssertionError):              # sph_iou_api.py:104, :180: mode in ['iou']
            fn(a, b, mode='iof')
        out = fn(a * 50 + 20, b * 50 + 20)               # CPU tensors: the product's host twins (round 3), never the oracle
        assert out.shape == (3, 3) and out.device.type == 'cpu' and bool(((out >= 0) & (out <= 1)).all())
        assert fn(torch.zeros(0, 4), b).shape == (0, 3) and fn(torch.zeros(0, 4), torch.zeros(0, 4), is_aligned=True).shape == (0, 1)
    assert S.SphOverlaps2D().backend == 'unbiased_iou'   # the reference's default (sph_iou_calculator.py:12)
    assert S.SphNMS('unbiased_iou').variant == 'unbiased' and S.SphNMS('naive_iou').variant == 'naive'
    with pytest.raises(NotImplementedError):
        S.sph_overlaps(a, b, backend='kent_iou')


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize('box', BOXES)
def test_gpu_unbiased_vs_fixtures_and_restatement(oracle, box):
    import sph_retina_amd as S
    from sph_retina_amd.iou import unbiased_iou
    g = load_golden('unbiased')
    b1, b2, r32, r64 = g[box + '_b1'], g[box + '_b2'], g[box + '_iou32'], g[box + '_iou64']
    noise = np.abs(r32.astype(np.float64) - r64)
    assert S.get_arithmetic() == 'fast'
    t1, t2 = cu(b1), cu(b2)
    out = unbiased_iou(t1, t2, is_aligned=True)
    assert out.shape == (b1.shape[0],) and out.dtype == torch.float32 and torch.equal(t1, cu(b1))
    o = out.cpu().numpy()
    d = np.abs(o - r64)
    assert (d > 5e-6).sum() <= 3 and np.median(d) < 1e-7, ((d > 5e-6).sum(), d.max())
    assert (np.abs(o - r32) > noise + 5e-6).sum() <= 3
    k = oracle.unbiased_iou(b1, b2, prec='kernel')
    assert (np.abs(o - k) > 1e-6).sum() <= 2                  # device libm vs glibc: only the chaotic pairs may move
    pw = unbiased_iou(cu(g[box + '_pa']), cu(g[box + '_pb'])).cpu().numpy()
    assert pw.shape == g[box + '_pw64'].shape and np.abs(pw - g[box + '_pw64']).max() < 5e-6
    # the calculator's default backend, with a trailing score column
    calc = S.SphOverlaps2D(box_version=b1.shape[1])
    withscore = torch.cat([cu(g[box + '_pa']), torch.rand(7, 1, device='cuda')], 1)
    assert torch.equal(calc(withscore, cu(g[box + '_pb'])), cu(pw))
    # reference arithmetic: tracks the float32 fixture at least as well as the reference tracks itself
    S.set_arithmetic('reference')
    try:
        e = unbiased_iou(t1, t2, is_aligned=True).cpu().numpy()
    finally:
        S.set_arithmetic('fast')
    de = np.abs(e - r32)
    assert (de > 1e-5).sum() < (noise > 1e-5).sum() and np.median(de) < 2.5e-7 and de.max() < 2e-2   # one chaotic pair may land anywhere


@pytest.mark.gpu
@pytest.mark.parametrize('dim', [4, 5])
def test_gpu_unbiased_and_naive_random_and_structured(oracle, dim):
    from sph_retina_amd.iou import naive_iou, unbiased_iou
    b1, b2 = structured_pairs(oracle, dim, n=30001)
    o = unbiased_iou(cu(b1), cu(b2), is_aligned=True).cpu().numpy()
    k = oracle.unbiased_iou(b1, b2, prec='kernel')
    d = np.abs(o - k)
    assert (d > 1e-6).sum() <= 3 and ((o >= 0) & (o <= 1)).all(), ((d > 1e-6).sum(), d.max())
    assert (o[:50] > 0.999).sum() >= 48
    n_ = naive_iou(cu(b1), cu(b2), is_aligned=True).cpu().numpy()
    assert np.abs(n_ - oracle.naive_iou(b1, b2)).max() < 1e-5
    np.testing.assert_allclose(n_[:50], 1.0, atol=1e-6)
    # pairwise == aligned on the expanded pairs (rows = bboxes1)
    a, b = b1[:37], b2[100:229]
    pw = unbiased_iou(cu(a), cu(b)).cpu().numpy()
    al = unbiased_iou(cu(np.repeat(a, len(b), 0)), cu(np.tile(b, (len(a), 1))), is_aligned=True).cpu().numpy()
    assert np.array_equal(pw.reshape(-1), al)
    pwn = naive_iou(cu(a), cu(b)).cpu().numpy()
    assert np.abs(pwn - oracle.naive_iou(a, b, is_aligned=False)).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('calc,variant', [('unbiased_iou', 'unbiased'), ('naive_iou', 'naive')])
@pytest.mark.parametrize('dim', [4, 5])
def test_gpu_nms_with_unbiased_and_naive_calculators(oracle, calc, variant, dim):
    import sph_retina_amd as S
    rng = np.random.default_rng(dim)
    box = 'bfov' if dim == 4 else 'rbfov'
    centres = oracle.generate_boxes(40, 9, box=box, alpha=(8, 50), beta=(8, 50))
    k = 700
    b = centres[rng.integers(0, 40, k)] + rng.normal(0, 2.0, (k, dim)).astype(np.float32)
    b[:, 0] %= 360
    b[:, 1] = np.clip(b[:, 1], 1, 179)
    b[:, 2:4] = np.clip(b[:, 2:4], 2, 120)
    scores = rng.random(k).astype(np.float32)
    idxs = rng.integers(0, 3, k)
    thr = 0.45
    dets, keep = S.SphNMS(calc)(cu(b), cu(scores), cu(idxs), dict(type='nms', iou_threshold=thr, max_num=200))
    rd, rk = oracle.batched_nms(b, scores, idxs, thr, 200, variant=variant)
    if not np.array_equal(keep.cpu().numpy(), rk):
        # a decision may only differ where an IoU sits within 2e-6 of the threshold
        iou = oracle.unbiased_iou(b, b, is_aligned=False) if variant == 'unbiased' else oracle.naive_iou(b, b, is_aligned=False)
        assert (np.abs(iou - thr) < 2e-6).any(), 'keep lists differ without a borderline IoU'
    else:
        np.testing.assert_allclose(dets.cpu().numpy(), rd, rtol=0, atol=0)


@pytest.mark.gpu
def test_gpu_unbiased_full_size_properties(oracle):
    """1 M pairs (BASELINE.json's size): range, self-overlap, disjointness, agreement with Sph2Pob where both are accurate
    (README: R_all ~ 0.9989 between Sph2Pob and Unbiased IoU on uniform pairs)."""
    import sph_retina_amd as S
    from sph_retina_amd.iou import unbiased_iou
    n = 1_000_000
    b1 = oracle.generate_boxes(n, 0)
    b2 = oracle.generate_boxes(n, 1)
    t1, t2 = cu(b1), cu(b2)
    u = unbiased_iou(t1, t2, is_aligned=True)
    assert bool(((u >= 0) & (u <= 1)).all())
    s = S.sph2pob_standard_iou(t1, t2, is_aligned=True)
    both = torch.stack([u, s])
    r = torch.corrcoef(both)[0, 1].item()
    assert r > 0.995, r
    assert float((u - s).abs().mean()) < 3e-3
    assert float((unbiased_iou(t1, t1, is_aligned=True) > 0.999).float().mean()) > 0.99   # see the identical-box note above
    sample = slice(0, 20000)
    k = oracle.unbiased_iou(b1[sample], b2[sample], prec='kernel')
    assert (np.abs(u[sample].cpu().numpy() - k) > 1e-6).sum() <= 2


# README "Comprehensive Comparison", column R_all: Pearson correlation with Unbiased-IoU over uniform pairs with
# alpha, beta in [1, 100) (protocol: tests/test_all_ious.py:29-88, :225-241; 10 k pairs there).
README_R_ALL = {'sph_iou': 0.7819, 'fov_iou': 0.9600, 'standard': 0.9989}


def test_readme_consistency_table_with_restatements(oracle):
    n = 30000
    b1, b2 = oracle.generate_boxes(n, 0), oracle.generate_boxes(n, 1)
    u = oracle.unbiased_iou(b1, b2, prec='kernel').astype(np.float64)
    for variant, published in README_R_ALL.items():
        r = np.corrcoef(oracle.iou_aligned(b1, b2, variant=variant).astype(np.float64), u)[0, 1]
        tol = 0.0006 if variant == 'standard' else 0.02      # sampling spread of R at 10 k pairs for the loose methods
        assert abs(r - published) < tol, (variant, r, published)


@pytest.mark.gpu
def test_gpu_readme_consistency_table_at_full_size(oracle):
    import sph_retina_amd.iou as I
    n = 1_000_000
    t1, t2 = cu(oracle.generate_boxes(n, 0)), cu(oracle.generate_boxes(n, 1))
    u = I.unbiased_iou(t1, t2, is_aligned=True).double()
    got = {}
    for variant, fn in (('sph_iou', I.sph_iou), ('fov_iou', I.fov_iou), ('standard', I.sph2pob_standard_iou),
                        ('efficient', I.sph2pob_efficient_iou)):
        got[variant] = torch.corrcoef(torch.stack([fn(t1, t2, is_aligned=True).double(), u]))[0, 1].item()
    assert abs(got['standard'] - README_R_ALL['standard']) < 0.0004, got
    assert abs(got['efficient'] - got['standard']) < 1e-5, got
    assert abs(got['fov_iou'] - README_R_ALL['fov_iou']) < 0.01 and abs(got['sph_iou'] - README_R_ALL['sph_iou']) < 0.02, got


def test_reference_sample_pairs_other_backends(oracle):
    """The 7 hard-coded pairs of the reference's tests/test_all_ious.py:244-261 through unbiased / Sph / FoV IoU."""
    g = load_golden('samples7_backends')
    b1, b2 = g['b1'], g['b2']
    assert np.array_equal(oracle.unbiased_iou(b1, b2, prec='f64'), g['unbiased64'])
    assert np.abs(oracle.unbiased_iou(b1, b2, prec='kernel') - g['unbiased']).max() < 5e-6
    assert np.array_equal(oracle.iou_aligned(b1, b2, variant='sph_iou'), g['sph'])
    assert np.array_equal(oracle.iou_aligned(b1, b2, variant='fov_iou'), g['fov'])


@pytest.mark.gpu
def test_gpu_reference_sample_pairs_other_backends():
    import sph_retina_amd.iou as I
    g = load_golden('samples7_backends')
    t1, t2 = cu(g['b1']), cu(g['b2'])
    assert np.abs(I.unbiased_iou(t1, t2, is_aligned=True).cpu().numpy() - g['unbiased64']).max() < 1e-6
    assert np.abs(I.unbiased_iou(t1, t2).cpu().numpy() - g['unbiased_pw']).max() < 5e-6
    assert np.abs(I.sph_iou(t1, t2, is_aligned=True).cpu().numpy() - g['sph']).max() < 1e-6
    assert np.abs(I.fov_iou(t1, t2, is_aligned=True).cpu().numpy() - g['fov']).max() < 1e-6
    # input immutability (tests/test_all_ious.py:322-332)
    assert torch.equal(t1, cu(g['b1'])) and torch.equal(t2, cu(g['b2']))


# ---- round 3: naive IoU with both planar box formators (Sph2PlanarBoxTransform 'sph2pix' | 'sph2tan') ----
@pytest.mark.parametrize('formator', ['sph2pix', 'sph2tan'])
def test_naive_rbfov_both_formators_vs_reference_fixture_cpu(oracle, formator):
    """tests/golden/naive.npz: the reference's naive_iou on RBFoV pairs (sph_iou_api.py:179-197, box_formator.py:76-106) through
    its vendored planar IoU.  The oracle restatement and the product's host twin (CPU tensors) against it; the BFoV branch
    needs mmcv.ops.bbox_overlaps (absent): parity unpinned, checked against the restatement only."""
    from sph_retina_amd.iou import naive_iou
    g = load_golden('naive')
    ref, ref64 = g['iou_' + formator], g['iou64_' + formator]
    o = oracle.naive_iou(g['b1'], g['b2'], planar='diff', box_formator=formator)
    assert np.abs(o - ref).max() < 1e-5 and np.abs(o - ref).mean() < 3e-7
    got = naive_iou(torch.from_numpy(g['b1']), torch.from_numpy(g['b2']), is_aligned=True, box_formator=formator).numpy()
    assert np.abs(got - ref64).max() < 5e-6, np.abs(got - ref64).max()     # the kernel's rotated stage runs in double
    assert np.abs(got - ref).max() < 2.5e-5
    b4 = g['b1'][:, :4], g['b2'][:, :4]
    got4 = naive_iou(torch.from_numpy(b4[0].copy()), torch.from_numpy(b4[1].copy()), is_aligned=True, box_formator=formator).numpy()
    assert np.abs(got4 - oracle.naive_iou(b4[0], b4[1], box_formator=formator)).max() < 5e-6    # (tanf of the host libm vs numpy's)
    with pytest.raises(AssertionError):
        naive_iou(torch.from_numpy(g['b1']), torch.from_numpy(g['b2']), box_formator='sph2kent')
    if formator == 'sph2tan':
        assert np.abs(ref - g['iou_sph2pix']).max() > 1e-2      # the two formators really differ


@pytest.mark.gpu
@pytest.mark.parametrize('formator', ['sph2pix', 'sph2tan'])
def test_gpu_naive_both_formators_and_planar_nms(oracle, formator):
    from sph_retina_amd.iou import naive_iou
    from sph_retina_amd.bbox.nms import PlanarNMS
    g = load_golden('naive')
    got = naive_iou(cu(g['b1']), cu(g['b2']), is_aligned=True, box_formator=formator).cpu().numpy()
    assert np.abs(got - g['iou64_' + formator]).max() < 5e-6
    pw = naive_iou(cu(g['b1'][:40]), cu(g['b2'][:70]), box_formator=formator).cpu().numpy()
    assert np.abs(pw - oracle.naive_iou(g['b1'][:40], g['b2'][:70], is_aligned=False, planar='exact', box_formator=formator)).max() < 5e-6
    host = naive_iou(torch.from_numpy(g['b1']), torch.from_numpy(g['b2']), is_aligned=True, box_formator=formator).numpy()
    assert np.abs(host - got).max() < 2e-6
    # PlanarNMS(box_formator) = naive-IoU NMS with that formator (sphdet/bbox/nms/planar_nms.py:7-19)
    rng = np.random.default_rng(3)
    b = g['b1'][:600, :4].copy()
    b[:, :2] = np.array([180, 90], np.float32) + rng.standard_normal((600, 2)).astype(np.float32) * 12
    s = rng.random(600).astype(np.float32)
    dets, keep = PlanarNMS(formator)(cu(b), cu(s), cu(np.zeros(600, np.int64)), dict(iou_threshold=0.5))
    iou = oracle.naive_iou(b, b, is_aligned=False, box_formator=formator)
    order = np.argsort(-s, kind='stable')
    want, alive = [], np.ones(600, bool)
    for i in order:
        if alive[i]:
            want.append(i)
            alive &= ~(iou[i] > 0.5)
            alive[i] = False
    assert keep.tolist() == want


@pytest.mark.gpu
@pytest.mark.parametrize('dim', [4, 5])
@pytest.mark.parametrize('thr', [0.5, 0.05, 0.0])
def test_gpu_unbiased_nms_compaction_never_changes_a_decision(dim, thr):
    """Round 3: the Unbiased-IoU NMS culls a row's far columns with the bounding caps and runs the exact fp64 path on the
    survivors only (nms_mask_compact_kernel<VARIANT_UNBIASED>).  The keep list must be that of the reference's loop
    (sph_nms.py:62-74) run on this package's own pairwise unbiased IoU — the one-lane kernel, no cull — including where the
    BFoV form's far value 1e-8 / (A1 + A2) matters: near-degenerate boxes (it exceeds thr there and the reference's loop
    suppresses DISJOINT boxes) and thr = 0 (nothing may be culled)."""
    import sph_retina_amd as S
    from test_gpu_nms import _loop_with
    rng = np.random.default_rng(100 + dim)
    k = 1500
    centres = np.stack([rng.random(60) * 360, 20 + rng.random(60) * 140, 5 + rng.random(60) * 50, 5 + rng.random(60) * 50,
                        -60 + rng.random(60) * 120], 1).astype(np.float32)[:, :dim]
    b = centres[rng.integers(0, 60, k)] + rng.normal(0, 2.0, (k, dim)).astype(np.float32)
    b[:, 0] %= 360
    b[:, 1] = np.clip(b[:, 1], 1, 179)
    b[:, 2:4] = np.clip(b[:, 2:4], 2, 120)
    b[::25, 2:4] = (0.004 + rng.random((len(b[::25]), 1)) * 0.016).astype(np.float32)   # near-degenerate squares of 0.004 ... 0.02 deg: the band in
    # which the BFoV far value 1e-8 / (A1 + A2 - 1e-8) of two DISJOINT boxes is 1.0 ... 0.04 (measured: 0.005 deg -> 1.0, 0.008 -> 0.35,
    # 0.012 -> 0.13, 0.02 -> 0.043): the reference's loop suppresses them, so the cull must not take them
    b[7::90, 2:4] = 0.0
    s = rng.random(k).astype(np.float32)
    idxs = rng.integers(0, 4, k)
    tb, ts, ti = cu(b), cu(s), cu(idxs)
    calc = S.SphOverlaps2D(backend='unbiased_iou', box_version=dim)
    want = _loop_with(lambda x, y: calc(x, y), tb, ts, ti, thr)
    dets, keep = S.SphNMS('unbiased_iou')(tb, ts, ti, dict(iou_threshold=thr))
    assert keep.tolist() == want
    assert len(want) < k   # something was suppressed
    if dim == 4 and thr > 0:   # ... including near-degenerate boxes that touch nothing (the far value alone removed them)
        tiny = set(range(0, k, 25)) - set(range(7, k, 90))
        assert len(tiny - set(want)) > 0
