"""GPU parity tests proper: HIP kernels (through the C ABI, via the Python boundary) vs the CPU oracle, the
committed reference fixtures, and size-independent properties at BASELINE.json's full sizes.

Tolerance model: the north-star bar is |dIoU| <= 1e-5 in fp32.  The reference's own fp32
arithmetic computes centre distance / edge angles as acos(clamp(dot)) of nearly parallel unit vectors, so two
faithful fp32 implementations with different libm last-bits (torch-CPU, CUDA, glibc, ocml) differ by more than
1e-5 on a small, measurable fraction of pairs whose centres are close (14-19 of 1M uniform
pairs, ~0.5 % of detector-like nearby pairs, against the same code in fp64).  Hence:
  * well-conditioned strata (uniform benchmark distribution minus the close-centre tail; hand-picked samples):
    max |d| <= 1e-5 asserted outright;
  * everywhere: mean |d| < 1e-6 (the reference's own criterion, tests/test_sph_iou_loss.py:34) and the number
    of >1e-5 outliers against fp64 truth must not exceed what the reference's fp32 arithmetic (the oracle's
    f32 instantiation) itself produces on the same inputs.

Two arithmetic modes are tested (sph_retina_amd.set_arithmetic): 'reference' = the reference's fp32 operation order
(strict criteria against the oracle's f32 instantiation), 'fast' = the default closed-form core, which is ~10x
closer to f64 truth on close-centre pairs than the reference's own fp32 arithmetic; its distance to the f32 oracle is
therefore bounded by the f32 oracle's own distance to truth.
"""
import numpy as np
import pytest
import torch

from conftest import err_stats, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def S():
    import sph_retina_amd
    assert torch.cuda.is_available()
    return sph_retina_amd


@pytest.fixture(params=['fast', 'reference'])
def arith(request, S):
    S.set_arithmetic(request.param)
    yield request.param
    S.set_arithmetic('fast')


FN = {'standard': 'sph2pob_standard_iou', 'efficient': 'sph2pob_efficient_iou', 'legacy': 'sph2pob_legacy_iou'}


def hip_iou(S, variant, b1, b2, aligned=True, **kw):
    t1, t2 = torch.from_numpy(np.ascontiguousarray(b1)).cuda(), torch.from_numpy(np.ascontiguousarray(b2)).cuda()
    return getattr(S.iou, FN[variant])(t1, t2, is_aligned=aligned, **kw).cpu().numpy()


def nearby(b, seed, sigma=(8, 8, 6, 6, 10)):
    rng = np.random.default_rng(seed)
    p = b + rng.standard_normal(b.shape).astype(np.float32) * np.asarray(sigma, np.float32)[:b.shape[1]]
    p[:, 0] %= 360
    p[:, 1] = p[:, 1].clip(0.5, 179.5)
    p[:, 2:4] = p[:, 2:4].clip(1, 170)
    if b.shape[1] == 5:
        p[:, 4] = p[:, 4].clip(-89, 89)
    return p.astype(np.float32)


def test_samples7_known_answers(S, arith):
    g = load_golden('samples7')
    for v in FN:
        got = hip_iou(S, v, g['b1'], g['b2'])
        np.testing.assert_allclose(got, g['iou_' + v], atol=1e-5, err_msg=v)


@pytest.mark.parametrize('v', list(FN))
def test_edge_cases(S, oracle, v, arith):
    g = load_golden('edge_cases')
    for mode in ('iou', 'iof'):
        got = hip_iou(S, v, g['b1'], g['b2'], mode=mode)
        ref = g[f'{mode}_{v}']
        orc = oracle.iou_aligned(g['b1'], g['b2'], variant=v, mode=mode, planar='mmcv')
        truth = oracle.iou_aligned(g['b1'], g['b2'], variant=v, mode=mode, planar='exact', dtype=np.float64)
        assert np.isfinite(got).all() and (got >= 0).all() and (got <= 1).all()
        well = np.array([1, 2, 3, 4, 5, 6, 10])          # seam, pole, antipodal, contained, swapped, clamped, big
        np.testing.assert_allclose(got[well], ref[well], atol=2e-5 if arith == 'reference' else 4e-5,
                                   err_msg=f'{v}/{mode}')
        # legacy = haversine + asin(sqrt(difference of squares)): its own fp32 noise is ~1e-4 at the theta seam
        if arith == 'reference' or v == 'legacy':
            np.testing.assert_allclose(got[well], orc[well], atol=1e-5 if v != 'legacy' else 1e-4)
        else:  # the closed-form core is held to the 1e-5 bar against the exact value of the reference's formula
            # (or of the f32 oracle where f64 'truth' is not a referee: row 6's clamp bound 180 - eps rounds differently)
            # Row 1 (both boxes ON the equator, across the seam): the bearing is exactly 0 / pi, where the reference takes
            # sign(noise) * acos(clamp) = +-4.88e-4 rad — a 2.6e-5 IoU coin flip in any implementation.
            tol = np.where(np.arange(len(got)) == 1, 4e-5, 1e-5)
            assert (np.minimum(np.abs(got - truth), np.abs(got - orc)) <= tol)[well].all(), (v, mode, got, truth, orc)
            np.testing.assert_allclose(got[well], orc[well], atol=4e-5)
        # rows 0, 7, 8, 9, 11: identical / pole-identical / 1-degree boxes half a degree apart — the reference's
        # fp32 arithmetic is itself 6e-4 .. 3e-3 away from the exact value of its own formula there
        ok = np.isfinite(ref)
        ref_err = np.abs(ref - truth)
        if arith == 'reference' or v == 'legacy':
            np.testing.assert_allclose(got[ok], ref[ok], atol=3e-3, err_msg=f'{v}/{mode} (ill-conditioned rows)')
        else:
            # identical / near-identical boxes live on near-coincident, near-parallel edges: every fp32 evaluation
            # (reference included) carries 1e-4 .. 3e-3 of noise there
            assert (np.abs(got - truth)[ok] <= 1.5 * ref_err[ok] + 3e-4).all(), (v, mode, got, truth, ref)
            assert (np.abs(got - ref)[ok] <= 2 * ref_err[ok] + 3e-4).all(), (v, mode, got, truth, ref)
            np.testing.assert_allclose(got[ok], truth[ok], atol=3e-3)


@pytest.mark.parametrize('name,variants', [('uniform_bfov', list(FN)), ('nearby_bfov', list(FN)),
                                           ('int_bfov', list(FN)), ('uniform_rbfov', ['standard', 'efficient']),
                                           ('nearby_rbfov', ['standard', 'efficient'])])
def test_fixtures_from_reference(S, name, variants, arith):
    g = load_golden(name)
    for v in variants:
        got = hip_iou(S, v, g['b1'], g['b2'])
        ref32, ref64 = g['iou_' + v], g['iou64_' + v]
        ok = np.isfinite(ref32)
        s = err_stats(got[ok], ref32[ok])
        # not further from fp64 truth than the reference's own fp32 run (torch CPU) is
        mine, theirs = err_stats(got[ok], ref64[ok]), err_stats(ref32[ok], ref64[ok])
        assert s['mean'] < max(1e-6, 1.5 * theirs['mean']), (name, v, s, theirs)
        # |hip - ref32| <= |hip - truth| + |truth - ref32|: up to twice the reference's own outlier count
        assert s['n5'] <= max(0.025 * s['n'], (1.5 if arith == 'reference' else 2.0) * theirs['n5'] + 5), (name, v, s, theirs)
        if name.startswith('uniform'):
            assert s['max'] < 1e-4, (name, v, s)
        assert mine['n5'] <= 1.5 * theirs['n5'] + 5, (name, v, mine, theirs)
        assert mine['mean'] <= 1.5 * theirs['mean'] + 1e-7, (name, v, mine, theirs)


def test_options_matrix(S, arith):
    g = load_golden('options')
    for key, ref in g.items():
        if key in ('b1', 'b2', 'r1', 'r2'):
            continue
        box, v, edge, ang, mode = key.split('_')
        b1, b2 = (g['b1'], g['b2']) if box == 'bfov' else (g['r1'], g['r2'])
        kw = dict(mode=mode, rbb_edge=edge)
        if v != 'legacy':
            kw['rbb_angle'] = ang
        got = hip_iou(S, v, b1, b2, **kw)
        s = err_stats(got, ref)
        assert s['mean'] < 3e-6 and s['n4'] <= 4, (key, s)


def test_pairwise_fixture_rows_are_first_argument(S, oracle, arith):
    g = load_golden('pairwise')
    for v in FN:
        got = hip_iou(S, v, g['b1'], g['b2'], aligned=False)
        assert got.shape == (7, 11)
        truth = oracle.iou_pairwise(g['b1'], g['b2'], variant=v, planar='exact', dtype=np.float64)
        tol = 5e-5 + 1.5 * np.abs(g['iou_' + v] - truth)   # the reference's own distance to the exact value
        assert (np.abs(got - g['iou_' + v]) <= tol).all(), (v, np.abs(got - g['iou_' + v]).max())
    for v in ('standard', 'efficient'):
        got = hip_iou(S, v, g['r1'], g['r2'], aligned=False)
        truth = oracle.iou_pairwise(g['r1'], g['r2'], variant=v, planar='exact', dtype=np.float64)
        tol = 5e-5 + 1.5 * np.abs(g['riou_' + v] - truth)
        assert (np.abs(got - g['riou_' + v]) <= tol).all(), (v, np.abs(got - g['riou_' + v]).max())


@pytest.mark.parametrize('v,box', [('standard', 'bfov'), ('efficient', 'bfov'), ('legacy', 'bfov'),
                                   ('standard', 'rbfov'), ('efficient', 'rbfov')])
@pytest.mark.parametrize('dist', ['uniform', 'nearby'])
def test_vs_oracle_200k(S, oracle, v, box, dist, arith):
    n = 200_000
    b1 = oracle.generate_boxes(n, 101, box=box)
    b2 = oracle.generate_boxes(n, 202, box=box) if dist == 'uniform' else nearby(b1, 7)
    got = hip_iou(S, v, b1, b2)
    ref32 = oracle.iou_aligned(b1, b2, variant=v, planar='mmcv', nthreads=8)
    truth = oracle.iou_aligned(b1, b2, variant=v, planar='exact', dtype=np.float64, nthreads=8)
    ok = np.isfinite(ref32) & np.isfinite(truth)
    assert np.isfinite(got).all() and got.min() >= 0 and got.max() <= 1
    keep = int(ok.sum() * 0.9999)  # trimmed means drop the 0.01 % largest: jitter-threshold / NaN flips of either side
    d = np.sort(np.abs(got[ok].astype(np.float64) - ref32[ok]))
    ref_noise = np.sort(np.abs(ref32[ok] - truth[ok]))[:keep].mean()
    mine, theirs = err_stats(got[ok], truth[ok]), err_stats(ref32[ok], truth[ok])
    if arith == 'reference':
        assert d[:keep].mean() < 1e-6, (v, box, dist, d[:keep].mean())
    else:  # distance to the f32 oracle is bounded by the f32 oracle's own distance to truth
        assert d[:keep].mean() < max(1e-6, 1.25 * ref_noise + 1e-7), (v, box, dist, d[:keep].mean(), ref_noise)
        assert err_stats(got[ok], ref32[ok])['n5'] <= 1.5 * theirs['n5'] + 10
    med_ref = np.median(np.abs(ref32[ok] - truth[ok]))
    assert np.median(d) <= (3e-7 if arith == 'reference' else max(3e-7, 1.25 * med_ref + 1e-7)), \
        (v, box, dist, np.median(d), med_ref)  # a few ulps of an IoU near 1
    # the reference's fp32 arithmetic (oracle f32) sets the noise floor; the kernel must not add to it
    assert mine['n5'] <= 1.25 * theirs['n5'] + 10, (v, box, dist, mine, theirs)
    assert mine['n4'] <= 1.25 * theirs['n4'] + 5, (v, box, dist, mine, theirs)
    assert np.sort(np.abs(got[ok] - truth[ok]))[:keep].mean() <= 1.25 * ref_noise + 1e-7
    # exact zeros agree (disjoint pairs are exactly 0 in the reference); the only disagreements are slivers that
    # mmcv's hull drops through its absolute tolerances (points within 1e-4 of each other count as one point)
    dis = ((got == 0) != (ref32 == 0)) & ok
    assert dis.mean() < 5e-4 and (not dis.any() or np.maximum(got, ref32)[dis].max() < 1e-4), \
        (v, box, dist, dis.sum(), np.maximum(got, ref32)[dis].max())
    if dist == 'uniform' and v != 'legacy':
        # the benchmark distribution: the 1e-5 bar holds for all but a handful of close-centre pairs
        # (a jitter-threshold flip between two fp32 realisations shows up as one ~1e-3 outlier per ~1e6 pairs)
        s = err_stats(got[ok], ref32[ok])
        assert s['n5'] <= 10 and s['n4'] <= 2, (v, box, dist, s)


def test_pairwise_equals_aligned_on_expanded(S, oracle, arith):
    m, n = 37, 1531
    b1 = oracle.generate_boxes(m, 5)
    b2 = oracle.generate_boxes(n, 6)
    b2[:m] = nearby(b1, 9)
    for v in FN:
        pw = hip_iou(S, v, b1, b2, aligned=False)
        al = hip_iou(S, v, np.repeat(b1, n, axis=0), np.tile(b2, (m, 1)))
        assert pw.shape == (m, n)
        np.testing.assert_array_equal(pw.reshape(-1), al)  # same device function -> bit-identical


def test_inputs_not_mutated_and_shapes(S):
    # reference tests/test_all_ious.py:322-332
    a = torch.rand(1000, 4, device='cuda') * 90 + 1
    b = a + 1.0
    b[::7] = a[::7]  # identical rows trigger the spherical jitter path
    a0, b0 = a.clone(), b.clone()
    out = S.sph2pob_standard_iou(a, b, is_aligned=True)
    assert out.shape == (1000,) and out.dtype == torch.float32 and out.device == a.device
    assert torch.equal(a, a0) and torch.equal(b, b0)
    out = S.sph2pob_efficient_iou(a[:5], b[:9])
    assert out.shape == (5, 9)
    assert S.sph2pob_standard_iou(a[:0], b[:3]).shape == (0, 3)
    assert S.sph2pob_standard_iou(a[:0], b[:0], is_aligned=True).shape == (0, 1)
    with pytest.raises(AssertionError):
        S.sph2pob_standard_iou(a, b, mode='giou')
    with pytest.raises(AssertionError):
        S.sph2pob_standard_iou(a, b, rbb_edge='secant')
    with pytest.raises(ValueError):
        S.sph2pob_legacy_iou(torch.rand(3, 5, device='cuda'), torch.rand(3, 5, device='cuda'))
    # non-contiguous and fp64 inputs are accepted (converted on the fly), extra score column is dropped
    calc = S.SphOverlaps2D(backend='sph2pob_standard_iou', box_version=4)
    with_scores = torch.cat([a, torch.rand(1000, 1, device='cuda')], 1)
    np.testing.assert_array_equal(calc(with_scores[:64], b[:128]).cpu().numpy(),
                                  S.sph2pob_standard_iou(a[:64], b[:128]).cpu().numpy())
    assert S.sph2pob_standard_iou(a.double(), b.double(), is_aligned=True).dtype == torch.float64


def test_full_size_properties_1m(S, oracle, arith):
    """BASELINE config 2 size (1,000,000 BFoV pairs): size-independent properties."""
    n = 1_000_000
    b1 = torch.from_numpy(oracle.generate_boxes(n, 0)).cuda()
    b2 = torch.from_numpy(oracle.generate_boxes(n, 1)).cuda()
    iou = S.sph2pob_efficient_iou(b1, b2, is_aligned=True)
    assert iou.shape == (n,) and bool(torch.isfinite(iou).all())
    assert float(iou.min()) >= 0 and float(iou.max()) <= 1
    frac = float((iou > 0).float().mean())
    assert 0.24 < frac < 0.28, frac  # 26 % of uniform pairs overlap
    # role swap: IoU is symmetric up to the (asymmetric) jitter constants
    sw = S.sph2pob_efficient_iou(b2, b1, is_aligned=True)
    assert float((iou - sw).abs().max()) < 5e-3 and float((iou - sw).abs().mean()) < 1e-6
    # standard and efficient are the same function of the pair (rigid planar motion)
    st = S.sph2pob_standard_iou(b1, b2, is_aligned=True)
    d = (iou - st).abs()
    assert float(d.mean()) < 1e-6 and int((d > 1e-4).sum()) <= 20
    # IoU(x, x) ~ 1 (jitter keeps it just below), iof >= iou
    same = S.sph2pob_standard_iou(b1[:100000], b1[:100000].clone(), is_aligned=True)
    assert float(same.min()) > 0.8 and float(same.max()) <= 1.0  # thin boxes: the jitter offsets are ~6 % of a 1-degree side
    iof = S.sph2pob_efficient_iou(b1, b2, mode='iof', is_aligned=True)
    assert bool((iof >= iou - 1e-6).all())
    # a random 20k sample against the oracle
    idx = torch.randperm(n, device='cuda')[:20000]
    ref = oracle.iou_aligned(b1[idx].cpu().numpy(), b2[idx].cpu().numpy(), variant='efficient', planar='mmcv')
    assert np.abs(iou[idx].cpu().numpy() - ref).mean() < 1e-6


def test_cheap_backends_sph_iou_fov_iou(S, oracle):
    """Sph-IoU / FoV-IoU (sphdet/iou/approximate_ious.py) served through the same registry: fixtures from the reference."""
    g = load_golden('approx')
    t1, t2 = torch.from_numpy(g['b1']).cuda(), torch.from_numpy(g['b2']).cuda()
    for name in ('sph_iou', 'fov_iou'):
        fn = getattr(S.iou, name)
        got = fn(t1, t2, is_aligned=True).cpu().numpy()
        np.testing.assert_allclose(got, g[name], atol=2e-6)       # element-wise closed forms: a few ulps
        np.testing.assert_allclose(got, oracle.iou_aligned(g['b1'], g['b2'], variant=name), atol=2e-6)
        pw = fn(torch.from_numpy(g['pa']).cuda(), torch.from_numpy(g['pb']).cuda()).cpu().numpy()
        np.testing.assert_allclose(pw, g[name + '_pw'], atol=2e-6)
        calc = S.SphOverlaps2D(backend=name, box_version=4)
        np.testing.assert_array_equal(calc(t1[:50], t2[:70]).cpu().numpy(), fn(t1[:50], t2[:70]).cpu().numpy())
        with pytest.raises(AssertionError):
            fn(t1, t2, mode='iof')
        with pytest.raises(ValueError):
            fn(torch.rand(3, 5, device='cuda'), torch.rand(3, 5, device='cuda'))


# Regression gate for the DEFAULT ('fast', closed-form) arithmetic — what bench.py runs — with FIXED numbers per stratum
# (round-1 VERDICT #2-ii), 1 M pairs each, inputs as tools/parity_report.py draws them (seeds 0 / 1, nearby(., 7)).
# Columns: pairs with |d| > 1e-5 and > 1e-4 against the reference's own fp32 arithmetic (the pinned oracle, mmcv planar
# stage), then the same against f64 truth of the reference's formula.  Measured on MI355X this round
# (profiles/r02a_parity_report.jsonl) in brackets; the bounds leave ~25 % headroom for libm / box differences, so a
# change that moves the default arithmetic away from the reference fails here rather than inside a noise-relative
# criterion.  For scale: the reference's fp32 result is itself > 1e-5 from the exact value of its own formula on
# 7 / 13 / 20 152 / 5 314 / 13 / 18 / 13 590 / 3 151 of these pairs (same order).
FAST_BOUNDS = {
    # (box, dist, variant):      (n5_ref32, n4_ref32, n5_truth, n4_truth)
    ('bfov', 'uniform', 'standard'):   (20, 1, 8, 0),            # [9, 0, 3, 0]
    ('bfov', 'uniform', 'efficient'):  (30, 1, 8, 0),            # [16, 0, 3, 0]
    ('bfov', 'nearby', 'standard'):    (26000, 550, 2700, 45),   # [21 776, 417, 2 027, 27]
    ('bfov', 'nearby', 'efficient'):   (9000, 190, 2700, 45),    # [7 084, 136, 2 005, 24]
    ('rbfov', 'uniform', 'standard'):  (25, 1, 10, 0),           # [12, 0, 4, 0]
    ('rbfov', 'uniform', 'efficient'): (35, 1, 10, 0),           # [18, 0, 4, 0]
    ('rbfov', 'nearby', 'standard'):   (17500, 260, 1100, 20),   # [14 030, 195, 788, 10]
    ('rbfov', 'nearby', 'efficient'):  (4700, 75, 1000, 18),     # [3 613, 52, 730, 9]
}


@pytest.mark.parametrize('box,dist,variant', sorted(FAST_BOUNDS))
def test_default_arithmetic_fixed_bounds_per_stratum_1m(S, oracle, box, dist, variant):
    S.set_arithmetic('fast')
    n = 1_000_000
    b1 = oracle.generate_boxes(n, 0, box=box)
    b2 = oracle.generate_boxes(n, 1, box=box) if dist == 'uniform' else nearby(b1, 7)
    got = hip_iou(S, variant, b1, b2)
    ref = oracle.iou_aligned(b1, b2, variant=variant, planar='mmcv', nthreads=64)
    tru = oracle.iou_aligned(b1, b2, variant=variant, planar='exact', dtype=np.float64, nthreads=64)
    ok = np.isfinite(ref) & np.isfinite(tru)
    r, t = err_stats(got[ok], ref[ok]), err_stats(got[ok], tru[ok])
    n5r, n4r, n5t, n4t = FAST_BOUNDS[(box, dist, variant)]
    assert r['n5'] <= n5r and r['n4'] <= n4r, (r, FAST_BOUNDS[(box, dist, variant)])
    assert t['n5'] <= n5t and t['n4'] <= n4t, (t, FAST_BOUNDS[(box, dist, variant)])
    assert r['mean'] < (1e-7 if dist == 'uniform' else 2.5e-6)
    if dist == 'uniform':   # the benchmark distribution: everything within 1e-4, all but a handful within 1e-5
        assert r['max'] < 1e-4 and t['max'] < 5e-5, (r, t)


def test_full_size_properties_8m_and_sharded_assembly(S, oracle):
    """BASELINE configs[4]'s batch on ONE GPU (8,000,000 BFoV pairs, the default arithmetic): size-independent properties,
    and the multi-GPU contract rehearsed on one device — evaluating the 8 contiguous shards that 8 ranks would own and
    concatenating them (what the all-gather assembles) gives the single-launch result bit for bit."""
    S.set_arithmetic('fast')
    n = 8_000_000
    g = torch.Generator(device='cpu').manual_seed(4)
    u = torch.rand((2, n, 4), generator=g)
    mk = lambda v: torch.stack([v[:, 0] * 360, v[:, 1] * 180, v[:, 2] * 99 + 1, v[:, 3] * 99 + 1], 1).cuda()  # noqa: E731
    b1, b2 = mk(u[0]), mk(u[1])
    iou = S.sph2pob_standard_iou(b1, b2, is_aligned=True)
    assert iou.shape == (n,) and bool(torch.isfinite(iou).all())
    assert float(iou.min()) >= 0 and float(iou.max()) <= 1
    assert 0.255 < float((iou > 0).float().mean()) < 0.265       # 26 % of uniform pairs overlap
    from sph_retina_amd.parallel import shard_bounds
    parts = []
    for r in range(8):
        lo, hi = shard_bounds(n, 8, r)
        parts.append(S.sph2pob_standard_iou(b1[lo:hi], b2[lo:hi], is_aligned=True))
    assert torch.equal(torch.cat(parts), iou)
    # ragged shards (3 ranks) too
    assert torch.equal(torch.cat([S.sph2pob_standard_iou(b1[lo:hi], b2[lo:hi], is_aligned=True)
                                  for lo, hi in (shard_bounds(n, 3, r) for r in range(3))]), iou)
    # The WHOLE population against both oracles with fixed counts (round-2 VERDICT weak #3: a sample that misses the batch's
    # few outliers is not a gate).  Measured on MI355X (tools/parity_population.py, profiles/r05g_parity_population.log), in
    # brackets; bounds = those + ~30 %:   vs ref32: 178 pairs > 1e-5, 5 > 1e-4;   vs f64: 38 > 1e-5, NONE > 1e-4 (max 7.7e-5);
    # the reference's own fp32 arithmetic vs f64: 145 / 5.  The five pairs beyond 1e-4 of ref32 are 211681, 2706828, 2876822,
    # 2991369, 4712741, and on every one of them it is ref32 that is off, not the kernel (|kernel - f64| < 2e-6 on all five):
    #   211681, 2706828, 4712741  mmcv's hull with absolute tolerances loses a vertex of a thin / large intersection
    #                             (ref32 0.0129 for a true 0.0360; the reference-order kernel, whose planar stage is exact,
    #                             agrees with f64 there too) — the planar stage's deliberate difference, DESIGN §3;
    #   2876822, 2991369          fp32 conditioning of the reference's acos(clamp(dot)) form at |a_g - a_p| ~ 2 pi resp. at a
    #                             centre distance of 0.017 rad: the reference-order kernel reproduces ref32 to 2e-7, the
    #                             closed form reproduces f64.
    # None is a jitter-threshold flip (the planar angle differences are nowhere near 1.2345678e-3).
    h1, h2 = b1.cpu().numpy(), b2.cpu().numpy()
    got = iou.cpu().numpy()
    ref = oracle.iou_aligned(h1, h2, variant='standard', planar='mmcv', nthreads=64)
    tru = oracle.iou_aligned(h1, h2, variant='standard', planar='exact', dtype=np.float64, nthreads=64)
    r, t = err_stats(got, ref), err_stats(got, tru)
    assert r['mean'] < 1e-7 and r['n5'] <= 230 and r['n4'] <= 8, r
    assert t['mean'] < 5e-8 and t['n5'] <= 55 and t['n4'] == 0 and t['max'] < 1e-4, t
    far = np.abs(got - ref) > 1e-4
    assert (np.abs(ref[far] - tru[far]) > 1e-4).all()       # wherever the kernel leaves ref32 by 1e-4, ref32 has left f64


@pytest.mark.parametrize('variant,n5_ref,n4_ref', [('standard', 700, 12), ('efficient', 200, 9)])
def test_full_size_assigner_matrix_25m_whole_population(S, oracle, variant, n5_ref, n4_ref):
    """All 25 141 248 pairs of the 64 GT x 392 832-anchor matrix (the literal 1024 x 2048 grid of configs[3]) against both
    oracles (round 2 checked every 97th column).  Measured (profiles/r05g_parity_population.log): standard 562 pairs > 1e-5 and
    8 > 1e-4 against ref32, 13 and 0 against f64 (max 2.3e-5); efficient 153 / 6 and 13 / 0.  As for the 8 M batch, the pairs
    beyond 1e-4 of ref32 — (GT, anchor) = (19, 362045), (23, 376178), (37, 312992), (39, 266155), (40, 12646), (40, 243303),
    (44, 322598), (59, 232336) for `standard` — are all pairs on which ref32 is > 1e-4 from f64 and the kernel is within 1e-6
    of f64: five are slivers / thin intersections that mmcv's tolerance hull drops or truncates (ref32 2e-6 for a true
    7.8e-3), three are the reference's fp32 conditioning at centre distances below 0.14 rad."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from bench_configs import retina_anchors
    S.set_arithmetic('fast')
    anchors = retina_anchors(1024, 2048)
    g = torch.Generator().manual_seed(0)
    u = torch.rand((64, 4), generator=g)
    gt = torch.stack([u[:, 0] * 360, 20 + u[:, 1] * 140, 5 + u[:, 2] * 85, 5 + u[:, 3] * 85], 1)
    fn = S.sph2pob_standard_iou if variant == 'standard' else S.sph2pob_efficient_iou
    ov = fn(gt.cuda(), anchors).cpu().numpy()
    a = anchors.cpu().numpy()
    ref = oracle.iou_pairwise(gt.numpy(), a, variant=variant, planar='mmcv', nthreads=64)
    tru = oracle.iou_pairwise(gt.numpy(), a, variant=variant, planar='exact', dtype=np.float64, nthreads=64)
    r, t = err_stats(ov.ravel(), ref.ravel()), err_stats(ov.ravel(), tru.ravel())
    assert r['mean'] < 2e-8 and r['n5'] <= n5_ref and r['n4'] <= n4_ref, r
    assert t['mean'] < 5e-9 and t['n5'] <= 25 and t['n4'] == 0 and t['max'] < 5e-5, t
    far = np.abs(ov - ref) > 1e-4
    assert (np.abs(ref[far] - tru[far]) > 1e-4).all()
    # exact zeros: the cull and mmcv's hull disagree only on slivers
    dis = (ov == 0) != (ref == 0)
    assert dis.sum() < 3000 and max(float(ov[dis].max()), float(ref[dis].max())) < 1e-2
