"""-m gpu: the one-round chunk kernel (one wave = 128 consecutive pairs: cull, compact on the wave's LDS stack, finish)
at the sizes and survivor counts where its bookkeeping can go wrong: batches that end inside a chunk / a slice / a
workgroup, chunks with no survivor, with exactly 64, with more than 64 (second pass) and with all 128 surviving.  The
referee is the same arithmetic evaluated one lane per pair through the pairwise entry point (`is_aligned=False` on
single rows: no compaction, no chunks) — bit for bit — and the CPU oracle within the parity bound.
Reference: sphdet/iou/sph_iou_api.py:48-86 (aligned mode: row i of bboxes1 against row i of bboxes2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mixed(n, dim, seed, frac_near):
    """uniform boxes (60 % culled) with a controlled share of near-duplicates (always survive)"""
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    b1 = O.generate_boxes(n, seed, box='rbfov' if dim == 5 else 'bfov')
    b2 = O.generate_boxes(n, seed + 1, box='rbfov' if dim == 5 else 'bfov')
    near = rng.random(n) < frac_near
    b2[near] = b1[near] + rng.standard_normal((int(near.sum()), dim)).astype(np.float32) * 2.0
    b2[:, 0] %= 360.0
    b2[:, 1:4] = b2[:, 1:4].clip(1, 179)
    return b1, np.ascontiguousarray(b2)


@pytest.mark.parametrize('dim', [4, 5])
@pytest.mark.parametrize('n', [1, 63, 64, 65, 127, 128, 129, 255, 511, 512, 513, 1023, 4097, 100003])
def test_ragged_sizes_match_one_lane_per_pair_bit_for_bit(n, dim):
    import torch
    import sph_retina_amd as S
    for frac in (0.0, 0.5, 1.0):   # 40 % / 70 % / 100 % survivors: one pass, sometimes two, always two
        b1, b2 = _mixed(n, dim, 11 + n % 7, frac)
        t1, t2 = torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda()
        for fn in (S.sph2pob_standard_iou, S.sph2pob_efficient_iou):
            got = fn(t1, t2, is_aligned=True)
            assert got.shape == (n,)
            # referee: chunks of the same pairs through the pairwise kernel's diagonal (other launch shape, other kernel)
            m = min(n, 300)
            ref = torch.stack([fn(t1[i:i + 1], t2[i:i + 1])[0, 0] for i in range(m)])
            assert torch.equal(got[:m], ref), (n, dim, frac, fn.__name__)
            # and the tail of the batch (the last, partial chunk)
            tail = range(max(n - 70, 0), n)
            ref = torch.stack([fn(t1[i:i + 1], t2[i:i + 1])[0, 0] for i in tail])
            assert torch.equal(got[max(n - 70, 0):], ref)


def test_survivor_counts_around_64_and_output_not_touched_past_n():
    """chunks whose survivor count is exactly 0 / 1 / 63 / 64 / 65 / 128; the output buffer beyond n stays as it was"""
    import torch
    import sph_retina_amd as S
    from oracle import oracle as O
    base = O.generate_boxes(128, 5)
    far = base.copy()
    far[:, 0] = (far[:, 0] + 180.0) % 360.0
    far[:, 1] = 180.0 - far[:, 1]          # antipodal boxes: culled for every size below
    base[:, 2:4] = base[:, 2:4].clip(1, 40)
    far[:, 2:4] = base[:, 2:4]
    for survivors in (0, 1, 63, 64, 65, 128):
        b2 = far.copy()
        b2[:survivors] = base[:survivors] + 0.5   # overlapping partner
        b2[:, 1:4] = b2[:, 1:4].clip(1, 179)
        reps = 5   # five chunks in a row, the same pattern in each
        t1 = torch.from_numpy(np.tile(base, (reps, 1))).cuda()
        t2 = torch.from_numpy(np.tile(b2, (reps, 1))).cuda()
        got = S.sph2pob_standard_iou(t1, t2, is_aligned=True).cpu().numpy().reshape(reps, 128)
        assert (got == got[0]).all()
        assert (got[0, survivors:] == 0).all()
        assert (got[0, :survivors] > 0.2).all()
        want = O.iou_aligned(base[:max(survivors, 1)], b2[:max(survivors, 1)], 'standard', planar='exact', dtype=np.float64)
        assert np.abs(got[0, :survivors] - want[:survivors]).max(initial=0.0) < 2e-5
    # raw C ABI: nothing is written past n
    from sph_retina_amd import _lib, _torch_glue as G
    import ctypes
    n = 1000
    t1 = torch.from_numpy(O.generate_boxes(1024, 1)).cuda()
    t2 = torch.from_numpy(O.generate_boxes(1024, 2)).cuda()
    out = torch.full((1024,), -7.0, device='cuda')
    rc = _lib.lib().sph2pob_iou_aligned_f32(G.ptr(t1), G.ptr(t2), G.ptr(out), ctypes.c_int64(n), 4, 0, 0, 0, 0,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert (out[n:] == -7.0).all() and (out[:n] >= 0).all()


@pytest.mark.parametrize('edge', ['arc', 'chord'])
@pytest.mark.parametrize('dim', [4, 5])
def test_pairwise_tiles_with_merged_leftovers_match_aligned_bit_for_bit(dim, edge):
    """the assigner kernel (rows x columns tiles, survivors stacked per wave, the four waves' leftovers merged behind an
    LDS-only barrier) at row / column counts that end inside a tile, with few and with many survivors per wave"""
    import torch
    import sph_retina_amd as S
    from oracle import oracle as O
    box = 'rbfov' if dim == 5 else 'bfov'
    for m, n in ((1, 1), (1, 257), (7, 255), (8, 256), (9, 1000), (64, 513), (65, 300), (130, 77)):
        b1 = O.generate_boxes(m, 21 + m, box=box)
        b2 = O.generate_boxes(n, 22 + n, box=box)
        k = min(m, n)
        b2[:k] = b1[:k] + 1.0            # a diagonal of certain survivors on top of the ~40 % that pass the cull
        b2[:, 1:4] = b2[:, 1:4].clip(1, 179)
        t1, t2 = torch.from_numpy(b1).cuda(), torch.from_numpy(np.ascontiguousarray(b2)).cuda()
        for fn in (S.sph2pob_standard_iou, S.sph2pob_efficient_iou):
            pw = fn(t1, t2, rbb_edge=edge)
            al = fn(t1.repeat_interleave(n, 0), t2.repeat(m, 1), is_aligned=True, rbb_edge=edge)
            assert pw.shape == (m, n)
            assert torch.equal(pw.reshape(-1), al), (m, n, dim, edge, fn.__name__)
        # every pair survives: columns = copies of the rows' boxes, slightly moved
        if m <= 9:
            t2 = (t1[torch.arange(n) % m] + 0.25).contiguous()
            pw = S.sph2pob_standard_iou(t1, t2, rbb_edge=edge)
            al = S.sph2pob_standard_iou(t1.repeat_interleave(n, 0), t2.repeat(m, 1), is_aligned=True, rbb_edge=edge)
            assert torch.equal(pw.reshape(-1), al)


def test_pairwise_more_than_65535_column_tiles():
    """the assigner kernel walks its column tiles from the last one backwards (a division of the linear workgroup id):
    more than 65 535 tiles, 2 rows x 16 777 300 columns, against the aligned kernel"""
    import torch
    import sph_retina_amd as S
    from bench import make_boxes
    n = 65535 * 256 + 340
    dev = torch.device('cuda', 0)
    cols = make_boxes(n, 5, dev)
    rows = make_boxes(2, 6, dev)
    pw = S.sph2pob_standard_iou(rows, cols)
    assert pw.shape == (2, n)
    for r in range(2):
        al = S.sph2pob_standard_iou(rows[r:r + 1].expand(n, 4).contiguous(), cols, is_aligned=True)
        assert torch.equal(pw[r], al), r
        del al
    assert float((pw > 0).float().mean()) > 0.1


def test_independent_calls_on_two_streams_give_the_same_bits():
    """calls that do not depend on each other may be issued on several HIP streams (their ramp-up and
    tail overlap); the launchers keep no state between calls, so the results are those of one stream"""
    import torch
    import sph_retina_amd as S
    from oracle import oracle as O
    n = 300_000
    sets = [(torch.from_numpy(O.generate_boxes(n, 40 + k)).cuda(), torch.from_numpy(O.generate_boxes(n, 50 + k)).cuda()) for k in range(4)]
    want = [S.sph2pob_standard_iou(a, b, is_aligned=True) for a, b in sets]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [None] * 4
    for rep in range(20):
        for k, (a, b) in enumerate(sets):
            with torch.cuda.stream(streams[k & 1]):
                got[k] = S.sph2pob_standard_iou(a, b, is_aligned=True)
    torch.cuda.synchronize()
    for k in range(4):
        assert torch.equal(got[k], want[k])


@pytest.mark.parametrize('dim', [4, 5])
def test_out_of_range_coordinates_are_clamped_like_the_reference_never_culled_wrongly(dim):
    """The cull tests the coordinates' ranges on their bit patterns and leaves any pair with a coordinate outside the
    range the spherical jitter clamps to (theta [0, 360], phi / alpha / beta [0, 180]; sph_iou_api.py:244-260) to the
    finishing stage, which clamps exactly like the reference: the result must be the one of the pre-clamped boxes."""
    import torch
    import sph_retina_amd as S
    from oracle import oracle as O
    n = 40000
    box = 'rbfov' if dim == 5 else 'bfov'
    b1, b2 = O.generate_boxes(n, 61, box=box), O.generate_boxes(n, 62, box=box)
    b2[: n // 2] = b1[: n // 2] + np.random.default_rng(3).standard_normal((n // 2, dim)).astype(np.float32) * 3
    b2[:, 0] %= 360.0
    b2[:, 1:4] = b2[:, 1:4].clip(1, 179)   # the partner of an out-of-range box is in range (see below)
    rng = np.random.default_rng(4)
    bad1, bad2 = b1.copy(), b2.copy()
    # one box of a pair at a time (even rows: the first, odd rows: the second): were both out of range on the same side
    # they would be EQUAL after clamping, and the jitter's `similar` test — taken on the raw values — would differ
    for arr, parity in ((bad1, 0), (bad2, 1)):
        rows = rng.integers(0, n // 2, 6000) * 2 + parity
        cols = rng.integers(0, 4, 6000)
        vals = rng.choice(np.array([-50.0, -0.5, -1e-3, 180.5, 250.0, 361.0, 720.0, 1e6, -1e6], dtype=np.float32), 6000)
        arr[rows, cols] = vals
    hi = np.array([360.0, 180.0, 180.0, 180.0] + ([np.inf] if dim == 5 else []), dtype=np.float32)
    lo = np.array([0.0, 0.0, 0.0, 0.0] + ([-np.inf] if dim == 5 else []), dtype=np.float32)
    ok1, ok2 = np.clip(bad1, lo, hi), np.clip(bad2, lo, hi)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    for fn in (S.sph2pob_standard_iou, S.sph2pob_efficient_iou):
        got = fn(t(bad1), t(bad2), is_aligned=True)
        want = fn(t(ok1), t(ok2), is_aligned=True)
        assert torch.equal(got, want), fn.__name__
        pw = fn(t(bad1[:9]), t(bad2[:700]))
        assert torch.equal(pw, fn(t(ok1[:9]), t(ok2[:700])))
    # and against the oracle (which restates the reference's clamps) on the out-of-range rows
    rows = np.unique(np.nonzero((bad1 != b1).any(1) | (bad2 != b2).any(1))[0])[:3000]
    want = O.iou_aligned(bad1[rows], bad2[rows], 'standard', planar='exact', dtype=np.float64)
    got = S.sph2pob_standard_iou(t(bad1[rows]), t(bad2[rows]), is_aligned=True).cpu().numpy()
    d = np.abs(got - want)
    assert np.mean(d) < 1e-6 and (d > 1e-4).sum() <= 3


def _big_boxes(n, seed):
    """(n, 4) BFoV boxes drawn on the device (the harness's ranges): the 2^31-pair batches below never exist on the host"""
    import torch
    g = torch.Generator(device='cuda').manual_seed(seed)
    b = torch.rand((n, 4), device='cuda', generator=g)
    b.mul_(torch.tensor([360.0, 180.0, 99.0, 99.0], device='cuda')).add_(torch.tensor([0.0, 0.0, 1.0, 1.0], device='cuda'))
    return b


def test_maximum_sizes_two_billion_aligned_pairs_and_a_ten_gigabyte_matrix():
    """Maximum sizes (the reference has no limit but memory: sph_iou_api.py:48-86 works on whatever the tensors hold).
    (i) 2^31 + 4173 aligned pairs — past every 32-bit index: the launcher leaves the chunk kernel (int indices, < 2^31 - 1024
    pairs) for the one-lane-per-pair kernel with 64-bit indices; (ii) the largest batch the chunk kernel takes; (iii) a
    64 x 40 000 003 pairwise matrix (2.56e9 elements: 64-bit row offsets in the compacting pairwise kernel).  Checked on
    windows at the start, across the 2^31 boundary and at the ragged end against small calls on the same boxes, bit for bit
    (the kernels share one arithmetic), and by the share of zero IoUs (the benchmark distribution's ~74 %)."""
    import torch
    import sph_retina_amd as S
    free, _ = torch.cuda.mem_get_info()
    if free < 100 * 2 ** 30:
        pytest.skip('needs ~80 GB of free device memory')
    n = 2 ** 31 + 4096 + 77
    b1, b2 = _big_boxes(n, 5), _big_boxes(n, 6)
    windows = [(0, 5000), (2 ** 31 - 1024 - 3000, 2 ** 31 - 1024 + 10), (2 ** 31 - 2500, 2 ** 31 + 2500), (n - 4200, n)]

    def check(out, upto):
        assert out.shape == (upto,)
        for lo, hi in windows:
            hi = min(hi, upto)
            if lo >= hi:
                continue
            small = S.sph2pob_standard_iou(b1[lo:hi], b2[lo:hi], is_aligned=True)
            assert torch.equal(out[lo:hi], small), (upto, lo, hi)
        for lo in range(0, upto, 2 ** 29):                     # every part of the batch was written: zeros ~74 %, never NaN
            part = out[lo:lo + 2 ** 22]
            z = float((part == 0).float().mean())
            assert 0.70 < z < 0.78 and bool(torch.isfinite(part).all()) and float(part.max()) <= 1.0, (upto, lo, z)

    out = S.sph2pob_standard_iou(b1, b2, is_aligned=True)
    check(out, n)
    del out
    n2 = 2 ** 31 - 1024 - 1                                     # the largest chunk-kernel batch (ragged: ends inside a chunk)
    out = S.sph2pob_standard_iou(b1[:n2], b2[:n2], is_aligned=True)
    check(out, n2)
    del out, b1, b2
    torch.cuda.empty_cache()
    # (iii) pairwise: 64 rows x 40 000 003 columns
    m, k = 40_000_003, 64
    gt, anchors = _big_boxes(k, 7), _big_boxes(m, 8)
    ov = S.sph2pob_standard_iou(gt, anchors)
    assert ov.shape == (k, m)
    for lo, hi in ((0, 3000), (2 ** 31 // 64 - 1500, 2 ** 31 // 64 + 1500), (m - 2051, m)):
        small = S.sph2pob_standard_iou(gt, anchors[lo:hi])
        assert torch.equal(ov[:, lo:hi], small), (lo, hi)
    col = torch.randint(0, m, (4096,), device='cuda')
    rows = torch.randint(0, k, (4096,), device='cuda')
    assert torch.equal(ov[rows, col], S.sph2pob_standard_iou(gt[rows], anchors[col], is_aligned=True))
    del ov, gt, anchors
    torch.cuda.empty_cache()
