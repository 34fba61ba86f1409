"""GPU parity tests of the Hamming searches (through the C ABI) against the CPU
oracle: distances and (best, second-best) results are integers -> bit-exact."""
import numpy as np
import pytest

from oracle_lib import KP_DTYPE
from orb_slam_amd.synth import random_descriptors, map_descriptors, synthetic_stereo_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ex():
    from orb_slam_amd import ORBextractor
    e = ORBextractor(1000, 1.2, 8, 20, 7)
    yield e
    e.close()


@pytest.fixture(scope="module")
def matcher(ex):
    from orb_slam_amd import ORBmatcher
    return ORBmatcher(ex)


def test_descriptor_distance(matcher, oracle):
    rng = np.random.default_rng(0)
    from orb_slam_amd import ORBmatcher
    z = np.zeros(32, np.uint8)
    o = np.full(32, 255, np.uint8)
    assert ORBmatcher.DescriptorDistance(z, z) == 0
    assert ORBmatcher.DescriptorDistance(z, o) == 256
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        ref = int(np.unpackbits(a ^ b).sum())
        assert ORBmatcher.DescriptorDistance(a, b) == ref == oracle.hamming(a, b)


@pytest.mark.parametrize("Q,T", [(1000, 1000), (1, 1), (257, 3), (64, 5000), (300, 0), (1000, 20000)])
def test_bruteforce(matcher, oracle, Q, T):
    q = random_descriptors(Q, 1)
    t = map_descriptors(T, q, seed=2) if T else np.zeros((0, 32), np.uint8)
    if T > 10:
        t[5] = t[3]          # exact duplicates: ties must go to the lower index
        q[0] = t[3]
    bi, bd, sd = matcher.match_bruteforce(q, t)
    oi, od, os_ = oracle.match_bruteforce(q, t)
    assert np.array_equal(bi, oi) and np.array_equal(bd, od) and np.array_equal(sd, os_)


def test_bruteforce_all_bits_differ(matcher, oracle):
    q = np.zeros((3, 32), np.uint8)
    t = np.full((4, 32), 255, np.uint8)      # distance 256 never beats the initial 256
    bi, bd, sd = matcher.match_bruteforce(q, t)
    oi, od, os_ = oracle.match_bruteforce(q, t)
    assert np.array_equal(bi, oi) and np.array_equal(bd, od) and np.array_equal(sd, os_)
    assert (bi == -1).all() and (bd == 256).all()


def test_bruteforce_extreme_popcounts(matcher, oracle):
    """Sizes that take the matrix-core sweep, with the extreme cases of its distance identity
    |a| + |b| - 2<a,b>: empty and full descriptors, complements (distance 256, never a match),
    identical sets, a train count that is no multiple of the 32-row tile, ragged query count."""
    rng = np.random.default_rng(21)
    Q, T = 97, 333
    q = rng.integers(0, 256, (Q, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (T, 32), dtype=np.uint8)
    q[0] = 0
    q[1] = 255
    t[0] = 255
    t[1] = 0
    t[2] = ~q[2]            # exact complement
    t[40] = q[5]
    t[300] = q[5]           # duplicate exact match: lower index wins, second distance 0
    bi, bd, sd = matcher.match_bruteforce(q, t)
    oi, od, os_ = oracle.match_bruteforce(q, t)
    assert np.array_equal(bi, oi) and np.array_equal(bd, od) and np.array_equal(sd, os_)
    assert bi[5] == 40 and bd[5] == 0 and sd[5] == 0
    # every train the complement of every query: nothing matches
    qq = np.zeros((64, 32), np.uint8)
    tt = np.full((96, 32), 255, np.uint8)
    bi, bd, sd = matcher.match_bruteforce(qq, tt)
    assert (bi == -1).all() and (bd == 256).all() and (sd == 256).all()
    # a set against itself
    bi, bd, sd = matcher.match_bruteforce(t, t)
    oi, od, os_ = oracle.match_bruteforce(t, t)
    assert np.array_equal(bi, oi) and np.array_equal(bd, od) and np.array_equal(sd, os_)


def _random_csr(rng, Q, T, maxc):
    counts = rng.integers(0, maxc + 1, Q)
    counts[rng.integers(0, Q, max(1, Q // 10))] = 0        # ragged: empty lists
    off = np.zeros(Q + 1, np.int32)
    off[1:] = np.cumsum(counts)
    idx = rng.integers(0, T, off[-1]).astype(np.int32)
    return off, idx


def test_windowed_and_csr(matcher, oracle):
    rng = np.random.default_rng(7)
    Q, T = 1500, 1200
    q = random_descriptors(Q, 3)
    t = map_descriptors(T, q, seed=4, planted=0.5)
    t[10] = t[11] = t[12]
    off, idx = _random_csr(rng, Q, T, 40)
    toct = rng.integers(0, 8, T).astype(np.int32)
    got = matcher.match_windowed(q, t, off, idx, toct)
    ref = oracle.match_windowed(q, t, off, idx, toct)
    for g, r, name in zip(got, ref, ["best_idx", "best_dist", "second_dist", "best_lvl", "second_lvl"]):
        assert np.array_equal(g, r), name
    d = matcher.hamming_csr(q, t, off, idx)
    assert np.array_equal(d, oracle.hamming_csr(q, t, off, idx))
    # without octaves
    got = matcher.match_windowed(q, t, off, idx, None)
    ref = oracle.match_windowed(q, t, off, idx, None)
    for g, r in zip(got[:3], ref[:3]):
        assert np.array_equal(g, r)


def test_bruteforce_device_batched(ex, matcher, oracle):
    torch = pytest.importorskip("torch")
    nb, stride = 5, 700
    rng = np.random.default_rng(11)
    nq = rng.integers(1, stride + 1, nb).astype(np.int32)
    nt = rng.integers(1, stride + 1, nb).astype(np.int32)
    q = rng.integers(0, 256, (nb, stride, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nb, stride, 32), dtype=np.uint8)
    dev = torch.device("cuda:0")
    dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    dnq, dnt = torch.from_numpy(nq).to(dev), torch.from_numpy(nt).to(dev)
    outs = [torch.full((nb, stride), -7, dtype=torch.int32, device=dev) for _ in range(3)]
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    matcher.match_bruteforce_device(dq, stride, stride, dnq, dt, stride, stride, dnt, nb, *outs)
    torch.cuda.synchronize()
    ex.set_stream(None)
    for b in range(nb):
        oi, od, os_ = oracle.match_bruteforce(q[b, :nq[b]], t[b, :nt[b]])
        assert np.array_equal(outs[0][b, :nq[b]].cpu().numpy(), oi)
        assert np.array_equal(outs[1][b, :nq[b]].cpu().numpy(), od)
        assert np.array_equal(outs[2][b, :nq[b]].cpu().numpy(), os_)


def test_stereo_on_extracted_pair(ex, matcher, oracle):
    left, right = synthetic_stereo_pair(752, 480, 4321)
    kl, dl = ex(left)
    kr, dr = ex(right)
    assert len(kl) > 500 and len(kr) > 500
    sf = ex.GetScaleFactors()
    mbf, mb = 435.2 * 0.11 * 0 + 47.9, 0.11
    maxD = np.float32(mbf) / np.float32(mb)
    bi, bd, ur, dp, nm = matcher.stereo_match(kl, dl, kr, dr, 480, mbf, mb, 0.0)
    oi, od = oracle.stereo_search(kl, dl, kr, dr, sf, 480, 0.0, float(maxD), 0.0)
    assert np.array_equal(bi, oi) and np.array_equal(bd, od)
    on, our, odp = oracle.stereo_finish(kl, kr, oi, od, mbf, 0.0, float(maxD))
    assert nm == on
    assert np.array_equal(ur, our) and np.array_equal(dp, odp)
    assert (ur >= 0).sum() > 100        # the synthetic pair really has matches


def test_stereo_edge_cases(ex, matcher, oracle):
    sf = ex.GetScaleFactors()
    rng = np.random.default_rng(3)
    n = 300
    kl = np.zeros(n, KP_DTYPE)
    kr = np.zeros(n, KP_DTYPE)
    for k in (kl, kr):
        k["x"] = rng.uniform(0, 752, n).astype(np.float32)
        k["y"] = rng.uniform(-3, 483, n).astype(np.float32)      # some rows outside [0, nRows-1]
        k["octave"] = rng.integers(0, 8, n)
    dl = random_descriptors(n, 5)
    dr = dl.copy()
    rng.shuffle(dr)
    kr["y"][:50] = kl["y"][:50]
    kr["x"][:50] = kl["x"][:50] - rng.uniform(0, 30, 50).astype(np.float32)
    kr["octave"][:50] = kl["octave"][:50]
    dr[:50] = dl[:50]
    mbf, mb = 40.0, 0.1
    maxD = np.float32(mbf) / np.float32(mb)
    bi, bd, ur, dp, nm = matcher.stereo_match(kl, dl, kr, dr, 480, mbf, mb, 100.0)
    oi, od = oracle.stereo_search(kl, dl, kr, dr, sf, 480, 0.0, float(maxD), 100.0)
    assert np.array_equal(bi, oi) and np.array_equal(bd, od)
    on, our, odp = oracle.stereo_finish(kl, kr, oi, od, mbf, 0.0, float(maxD))
    assert nm == on and np.array_equal(ur, our) and np.array_equal(dp, odp)
    # no right keypoints at all
    bi, bd, ur, dp, nm = matcher.stereo_match(kl, dl, kr[:0], dr[:0], 480, mbf, mb, 0.0)
    assert (bi == -1).all() and (ur == -1).all() and nm == 0


def test_stereo_classic_sad_refinement(ex, matcher):
    """Frame::ComputeStereoMatches with the SAD sub-pixel refinement on both extractors' pyramids."""
    from orb_slam_amd import ORBextractor
    from oracle_lib import Oracle
    exr = ORBextractor(1000, 1.2, 8, 20, 7)
    ol, orr = Oracle(1000, 1.2, 8, 20, 7), Oracle(1000, 1.2, 8, 20, 7)
    for seed in (4321, 77):
        left, right = synthetic_stereo_pair(752, 480, seed)
        kl, dl = ex(left)
        kr, dr = exr(right)
        kl_o, dl_o = ol.extract(left)
        kr_o, dr_o = orr.extract(right)
        assert np.array_equal(kl, kl_o) and np.array_equal(kr, kr_o)
        mbf, mb = 47.9, 0.11
        bi, bd, ur, dp, nm = matcher.stereo_match_sad(exr, kl, dl, kr, dr, mbf, mb)
        oi, od, our, odp, onm = ol.stereo_classic(orr, kl, dl, kr, dr, mbf, mb)
        assert np.array_equal(bi, oi) and np.array_equal(bd, od)
        assert nm == onm
        assert np.array_equal(ur, our) and np.array_equal(dp, odp)       # float results bit-identical
        assert (ur >= 0).sum() > 100
    exr.close()


def _grid_queries(rng, kps, Q):
    pick = rng.integers(0, len(kps), Q)
    qx = (kps["x"][pick] + rng.normal(0, 6, Q)).astype(np.float32)
    qy = (kps["y"][pick] + rng.normal(0, 6, Q)).astype(np.float32)
    qx[:8] = [-50, 800, 10, 740, 376, 0, 751.5, 300]          # outside / at the borders
    qy[:8] = [100, 100, -40, 500, 240, 0, 479.5, 479]
    qr = rng.uniform(2, 40, Q).astype(np.float32)
    qmin = rng.integers(-1, 4, Q).astype(np.int32)
    qmax = np.where(rng.random(Q) < 0.3, -1, qmin + rng.integers(0, 4, Q)).astype(np.int32)
    return qx, qy, qr, qmin, qmax, pick


def test_grid_candidates_and_fused_search(ex, matcher, oracle, euroc_left):
    """Frame::GetFeaturesInArea candidate lists (CSR, reference order) and the fused grid search."""
    rng = np.random.default_rng(21)
    kps, desc = ex(euroc_left)
    # grid constants of an undistorted 752x480 frame: src/Frame.cc:134-135
    min_x, min_y = 0.0, 0.0
    inv_w, inv_h = float(np.float32(64) / np.float32(752)), float(np.float32(48) / np.float32(480))
    Q = 700
    qx, qy, qr, qmin, qmax, pick = _grid_queries(rng, kps, Q)
    off_o, idx_o = oracle.features_in_area(kps, min_x, min_y, inv_w, inv_h, qx, qy, qr, qmin, qmax)
    off_g, idx_g = matcher.features_in_area(kps, min_x, min_y, inv_w, inv_h, qx, qy, qr, qmin, qmax)
    assert np.array_equal(off_g, off_o) and np.array_equal(idx_g, idx_o)
    assert off_o[-1] > 3 * Q
    qdesc = desc[pick].copy()
    flip = rng.integers(0, 256, Q)
    qdesc[np.arange(Q), flip // 8] ^= (1 << (flip % 8)).astype(np.uint8)
    got = matcher.grid_search(kps, desc, min_x, min_y, inv_w, inv_h, qx, qy, qr, qmin, qmax, qdesc)
    ref = oracle.match_windowed(qdesc, desc, off_o, idx_o, kps["octave"])
    for g_, r_, name in zip(got[:5], ref, ["best_idx", "best_dist", "second_dist", "best_lvl", "second_lvl"]):
        assert np.array_equal(g_, r_), name
    assert np.array_equal(got[5], np.diff(off_o))
    # keypoints outside the grid (undistortion can push them out) are not listed
    k2 = kps.copy()
    k2["x"][:40] -= 700
    off_o, idx_o = oracle.features_in_area(k2, min_x, min_y, inv_w, inv_h, qx, qy, qr, qmin, qmax)
    off_g, idx_g = matcher.features_in_area(k2, min_x, min_y, inv_w, inv_h, qx, qy, qr, qmin, qmax)
    assert np.array_equal(off_g, off_o) and np.array_equal(idx_g, idx_o)
