"""GPU parity of the two search routines whose queries exclude what earlier queries claimed
(ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:520-633; ORBmatcher::SearchByProjection(Frame&,
vector<MapPoint*>&, th), :155-241) against the CPU oracle: the device forms the candidate lists and every
candidate distance, the library walks them in query order.  Every output array must be bit-identical."""
import numpy as np
import pytest

from orb_slam_amd import KP_DTYPE
from test_oracle_search import GRID, _map_points, _random_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ini():
    """the monocular initialisation extractor of the reference: 2000 features (src/Tracking.cc:279)"""
    from orb_slam_amd import ORBextractor
    ex = ORBextractor(2000, 1.2, 8, 20, 7)
    yield ex
    ex.close()


def _both_init(matcher, oracle, k1, d1, k2, d2, prev, window, ratio, ori):
    from orb_slam_amd import ORBmatcher
    m = ORBmatcher(matcher._ex, nnratio=ratio, checkOri=ori)
    pg, po = prev.copy(), prev.copy()
    nm_g, m_g = m.SearchForInitialization(k1, d1, k2, d2, GRID, pg, window)
    nm_o, m_o = oracle.search_for_initialization(k1, d1, k2, d2, GRID, po, window, ratio, ori)
    assert nm_g == nm_o
    assert np.array_equal(m_g, m_o)
    assert np.array_equal(pg, po)
    return nm_g, m_g, pg


@pytest.fixture(scope="module")
def matcher(ini):
    from orb_slam_amd import ORBmatcher
    return ORBmatcher(ini, nnratio=0.9, checkOri=True)


def test_initialization_search_on_the_euroc_pair(ini, matcher, oracle, euroc_left, euroc_right):
    """Tracking::MonocularInitialization (src/Tracking.cc:1175-1180): matcher(0.9, true), window 100, vbPrevMatched =
    the first frame's keypoint positions; then the next call with the vector the first one updated."""
    k1, d1 = ini(euroc_left)
    k2, d2 = ini(euroc_right)
    assert len(k1) > 1800 and len(k2) > 1800
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    for ori in (True, False):
        nm, m12, prev2 = _both_init(matcher, oracle, k1, d1, k2, d2, prev, 100, 0.9, ori)
        assert nm > 50
        # steals happened: some F2 keypoint was the best of several F1 keypoints, only one holds it
        assert len(set(m12[m12 >= 0].tolist())) == (m12 >= 0).sum()
    # frame after frame: the updated positions, a narrower window, the frames swapped
    _both_init(matcher, oracle, k1, d1, k2, d2, prev2, 30, 0.9, True)
    _both_init(matcher, oracle, k2, d2, k1, d1, np.stack([k2["x"], k2["y"]], 1).astype(np.float32), 100, 0.9, True)


def test_initialization_search_on_a_synthetic_pair(ini, matcher, oracle):
    from orb_slam_amd.synth import synthetic_frame
    a = synthetic_frame(752, 480, seed=9100)
    b = np.roll(a, (3, -7), axis=(0, 1))                 # the same scene moved by (-7, 3)
    k1, d1 = ini(a)
    k2, d2 = ini(b)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    nm, m12, _ = _both_init(matcher, oracle, k1, d1, k2, d2, prev, 100, 0.9, True)
    assert nm > 150
    sel = m12 >= 0
    dx = k2["x"][m12[sel]] - k1["x"][sel]
    assert np.median(dx) == -7.0


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_initialization_search_with_forced_steals_and_ties(matcher, oracle, seed):
    """look-alike descriptors (a few prototypes), crowded windows: equal distances, claims that collide,
    un-matching (:578-582) and the `<=` skip (:559-560) all occur"""
    rng = np.random.default_rng(seed)
    n1, n2 = 1500, 1700
    k1, d1, k2, d2 = _random_frames(rng, n1, n2, clusters=12)
    k2["x"][:n1] = k1["x"] + rng.integers(-6, 7, n1).astype(np.float32)
    k2["y"][:n1] = k1["y"] + rng.integers(-6, 7, n1).astype(np.float32)
    bits = np.unpackbits(d1, axis=1)
    d2[:n1] = np.packbits(bits ^ (rng.random(bits.shape) < 0.02), axis=1)
    # exact duplicates: several F1 keypoints with the same descriptor and position compete for one F2 keypoint
    d1[1::7] = d1[0:-1:7][:len(d1[1::7])]
    k1["x"][1::7], k1["y"][1::7] = k1["x"][0:-1:7][:len(k1["x"][1::7])], k1["y"][0:-1:7][:len(k1["x"][1::7])]
    k1["octave"][1::7] = 0
    k1["octave"][0::7] = 0
    k2["angle"][:n1] = ((k1["angle"] + rng.choice([0.0, 0.0, 2.0, 90.0, 185.0], n1)) % 360).astype(np.float32)
    k2["octave"][:n1] = np.where(rng.random(n1) < 0.8, 0, k2["octave"][:n1])
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    for ratio, ori, window in ((0.9, True, 25), (0.6, False, 60), (1.0, True, 10)):
        nm, m12, _ = _both_init(matcher, oracle, k1, d1, k2, d2, prev, window, ratio, ori)
        assert nm > 50


def test_initialization_search_edge_cases(matcher, oracle):
    rng = np.random.default_rng(3)
    k1, d1, k2, d2 = _random_frames(rng, 40, 50, clusters=3)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    # an empty second frame, an empty first frame, no keypoint on level 0, windows outside the image
    nm, m12 = matcher.SearchForInitialization(k1, d1, k2[:0], d2[:0], GRID, prev.copy(), 100)
    assert nm == 0 and (m12 == -1).all()
    nm, m12 = matcher.SearchForInitialization(k1[:0], d1[:0], k2, d2, GRID, np.zeros((0, 2), np.float32), 100)
    assert nm == 0 and len(m12) == 0
    k1b = k1.copy()
    k1b["octave"] = 2
    _both_init(matcher, oracle, k1b, d1, k2, d2, prev, 100, 0.9, True)
    far = prev + np.float32(5000)
    nm, _, _ = _both_init(matcher, oracle, k1, d1, k2, d2, far, 100, 0.9, True)
    assert nm == 0
    # negative coordinates and keypoints outside the grid (undistortion can push them out)
    k2b = k2.copy()
    k2b["x"][::5] -= 800
    _both_init(matcher, oracle, k1, d1, k2b, d2, prev - np.float32(20), 300, 0.9, False)


def _both_projection(matcher, oracle, kps, desc, u_right, mp, th, ratio, obs0):
    from orb_slam_amd import ORBmatcher
    m = ORBmatcher(matcher._ex, nnratio=ratio)
    res = []
    for side in (m, oracle):
        fo, fm, ms = obs0.copy(), np.full(len(kps), -1, np.int32), np.full(len(kps), -1, np.int32)
        if side is oracle:
            nm = oracle.search_by_projection(kps, desc, u_right, GRID, mp, th, ratio, fo, fm, ms)
        else:
            nm = m.SearchByProjection(kps, desc, u_right, GRID, mp, th, fo, fm, ms)
        res.append((nm, fo, fm, ms))
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert np.array_equal(a, b)
    return res[0]


def test_projection_search_on_the_euroc_pair(oracle, euroc_left, euroc_right):
    """Tracking::SearchLocalPoints (src/Tracking.cc:2437-2441): local map points projected into the current frame.
    Map points = the right image's keypoints seen from the left frame (projection = the keypoint the row-band
    stereo search paired it with, jittered), descriptors from the right image, th = 1 and 3 (:2431-2436)."""
    from orb_slam_amd import ORBextractor, ORBmatcher
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex, nnratio=0.8)
    kl, dl = ex(euroc_left)
    kr, dr = ex(euroc_right)
    bi, bd, ur, dp, _ = m.stereo_match(kl, dl, kr, dr, 480, 47.9, 0.11, 0.0)
    rng = np.random.default_rng(77)
    has = np.nonzero(bi >= 0)[0]
    M = len(has)
    assert M > 300
    mp = {
        "mbTrackInView": (rng.random(M) < 0.95).astype(np.uint8),
        "isBad": (rng.random(M) < 0.03).astype(np.uint8),
        "mTrackProjX": (kl["x"][has] + rng.normal(0, 1.0, M)).astype(np.float32),
        "mTrackProjY": (kl["y"][has] + rng.normal(0, 1.0, M)).astype(np.float32),
        "mTrackProjXR": (kr["x"][bi[has]] + rng.normal(0, 1.0, M)).astype(np.float32),
        "mnTrackScaleLevel": kl["octave"][has].astype(np.int32),
        "mTrackViewCos": rng.choice([0.95, 0.9995], M).astype(np.float32),
        "Observations": rng.choice([0, 2, 5], M).astype(np.int32),
        "descriptor": dr[bi[has]],
    }
    obs0 = np.where(rng.random(len(kl)) < 0.1, 3, -1).astype(np.int32)
    for th in (1.0, 3.0):
        for u_right in (ur, None):
            nm, fo, fm, ms = _both_projection(m, oracle, kl, dl, u_right, mp, th, 0.8, obs0)
            assert nm > 200
    ex.close()


@pytest.mark.parametrize("seed,stereo,th", [(21, False, 1.0), (22, True, 1.0), (23, True, 5.0), (24, False, 3.0)])
def test_projection_search_with_collisions(matcher, oracle, seed, stereo, th):
    """many map points per keypoint neighbourhood: claims by points with and without observations, overwrites,
    the same-level ratio test, the right-image gate"""
    rng = np.random.default_rng(seed)
    kps, desc, _, _ = _random_frames(rng, 3000, 1, clusters=10)
    kps["octave"] = rng.integers(0, 8, len(kps))
    sf = oracle.tables()["scale"]
    mp = _map_points(rng, kps, desc, 4000, sf)
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.random(len(kps)).astype(np.float32) * 30, -1).astype(np.float32)
    obs0 = rng.choice([-1, -1, -1, 0, 2], len(kps)).astype(np.int32)
    nm, fo, fm, ms = _both_projection(matcher, oracle, kps, desc, u_right, mp, th, 0.8, obs0)
    assert nm > 500
    # empty inputs
    e = {k: v[:0] for k, v in mp.items()}
    assert _both_projection(matcher, oracle, kps, desc, u_right, e, th, 0.8, obs0)[0] == 0
    none_in_view = dict(mp, mbTrackInView=np.zeros(len(mp["isBad"]), np.uint8))
    assert _both_projection(matcher, oracle, kps, desc, u_right, none_in_view, th, 0.8, obs0)[0] == 0


# ---- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, numVisibleMpt), src/ORBmatcher.cc:1440-1593 ----
def _both_projection_last(matcher, oracle, kps, desc, u_right, last, th, mbf, fwd, bwd, ori, obs0):
    from orb_slam_amd import ORBmatcher
    m = ORBmatcher(matcher._ex, nnratio=0.9, checkOri=ori)
    res = []
    for side in (m, oracle):
        fo, fm = obs0.copy(), np.full(len(kps), -1, np.int32)
        if side is oracle:
            nm = oracle.search_by_projection_last(kps, desc, u_right, GRID, last, th, mbf, fwd, bwd, ori, fo, fm)
        else:
            nm = m.SearchByProjectionLast(kps, desc, u_right, GRID, last, th, mbf, fwd, bwd, fo, fm)
        res.append((nm, fo, fm))
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    return res[0]


def test_last_frame_projection_search_on_the_euroc_pair(oracle, euroc_left, euroc_right):
    """Tracking::TrackWithMotionModel (src/Tracking.cc:2196-2216): th = 15 (mono) or 7 (stereo), the search repeated
    with 2*th when fewer than 20 matches come back.  The last frame = the right image, its map points projected
    where the row-band stereo search found them in the left image (jittered)."""
    from orb_slam_amd import ORBextractor, ORBmatcher
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex, nnratio=0.9, checkOri=True)
    kl, dl = ex(euroc_left)
    kr, dr = ex(euroc_right)
    bi, bd, ur, dp, _ = m.stereo_match(kl, dl, kr, dr, 480, 47.9, 0.11, 0.0)
    rng = np.random.default_rng(78)
    has = np.nonzero(bi >= 0)[0]
    M = len(has)
    last = {
        "visible": (rng.random(M) < 0.95).astype(np.uint8),
        "u": (kl["x"][has] + rng.normal(0, 2.0, M)).astype(np.float32),
        "v": (kl["y"][has] + rng.normal(0, 2.0, M)).astype(np.float32),
        "invzc": (dp[has] / np.float32(47.9)).astype(np.float32) if (dp[has] > 0).all() else rng.uniform(0.05, 1, M).astype(np.float32),
        "octave": kr["octave"][bi[has]].astype(np.int32),
        "angle": kr["angle"][bi[has]].astype(np.float32),
        "Observations": rng.choice([0, 2, 5], M).astype(np.int32),
        "descriptor": dr[bi[has]],
    }
    obs0 = np.where(rng.random(len(kl)) < 0.05, 3, -1).astype(np.int32)
    for th, u_right in ((15.0, None), (7.0, ur), (14.0, ur)):
        for fwd, bwd in ((False, False), (True, False), (False, True)):
            for ori in (True, False):
                nm, fo, fm = _both_projection_last(m, oracle, kl, dl, u_right, last, th, 47.9, fwd, bwd, ori, obs0)
                assert nm > 80
    ex.close()


@pytest.mark.parametrize("seed,stereo,fwd,bwd,th", [(31, False, False, False, 15.0), (32, True, True, False, 7.0),
                                                   (33, True, False, True, 7.0), (34, True, False, False, 30.0)])
def test_last_frame_projection_search_with_collisions(matcher, oracle, seed, stereo, fwd, bwd, th):
    from test_oracle_search import _last_frame_points
    rng = np.random.default_rng(seed)
    kps, desc, _, _ = _random_frames(rng, 3000, 1, clusters=10)
    kps["octave"] = rng.integers(0, 8, len(kps))
    last = _last_frame_points(rng, kps, desc, 2500)
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.random(len(kps)).astype(np.float32) * 40, -1).astype(np.float32)
    obs0 = rng.choice([-1, -1, -1, 0, 2], len(kps)).astype(np.int32)
    for ori in (True, False):
        nm, fo, fm = _both_projection_last(matcher, oracle, kps, desc, u_right, last, th, 40.0, fwd, bwd, ori, obs0)
        assert nm > 300
    e = {k: v[:0] for k, v in last.items()}
    assert _both_projection_last(matcher, oracle, kps, desc, u_right, e, th, 40.0, fwd, bwd, True, obs0)[0] == 0
    hidden = dict(last, visible=np.zeros(len(last["u"]), np.uint8))
    assert _both_projection_last(matcher, oracle, kps, desc, u_right, hidden, th, 40.0, fwd, bwd, True, obs0)[0] == 0
    assert _both_projection_last(matcher, oracle, kps[:0], desc[:0], None, last, th, 40.0, fwd, bwd, True, obs0[:0])[0] == 0


# ---- the good-feature forms (include/ORBmatcher.h:68-252, src/ORBmatcher.cc:45-153) ----
@pytest.mark.parametrize("seed,stereo,th", [(41, False, 1.0), (42, True, 3.0), (43, True, 1.0)])
def test_candidate_table_and_one_point_matching(matcher, oracle, seed, stereo, th):
    """GetCandidates for all map points in one device call (lists + distances), then MatchCandidates /
    SearchByProjection_OnePoint per point in the order an active-matching heap would draw them
    (src/Observability.cc:1309-1329, :1404-1413): every list, every distance, every return value and the frame's
    assignment after every call equal the oracle's."""
    from orb_slam_amd import ORBmatcher
    from test_oracle_search import _fresh, _gf_case
    rng, kps, desc, sf, mp, u_right, obs0 = _gf_case(oracle, seed, stereo, n=3000, m=2500)
    m = ORBmatcher(matcher._ex, nnratio=0.8)
    M = len(mp["isBad"])
    cands = m.GetCandidates(kps, desc, GRID, mp, th)
    off, idx, dist = cands
    assert off[0] == 0 and off[M] == len(idx) == len(dist) and len(idx) > 2000
    for i in range(0, M, 7):
        want = oracle.get_candidates(kps, GRID, mp, i, th)
        assert np.array_equal(idx[off[i]:off[i + 1]], want)
        if len(want):
            d = np.unpackbits(desc[want] ^ mp["descriptor"][i][None, :], axis=1).sum(1)
            assert np.array_equal(dist[off[i]:off[i + 1]], d.astype(np.uint16))
    order = rng.permutation(M)
    got, ref = _fresh(obs0), _fresh(obs0)
    for n, i in enumerate(order):
        a = m.MatchCandidates(kps, u_right, cands, mp, int(i), th, *got)
        b = oracle.search_by_projection_one_point(kps, desc, u_right, GRID, mp, int(i), th, 0.8, *ref)
        assert a == b, (n, i)
    for x, y in zip(got, ref):
        assert np.array_equal(x, y)
    assert (got[1] >= 0).sum() > 200
    # nothing in view, nothing at all
    hidden = dict(mp, mbTrackInView=np.zeros(M, np.uint8))
    off0, idx0, _ = m.GetCandidates(kps, desc, GRID, hidden, th)
    assert len(idx0) == 0 and (off0 == 0).all()
    assert m.MatchCandidates(kps, u_right, (off0, idx0, idx0.astype(np.uint16)), hidden, 3, th, *got) == -1


@pytest.mark.parametrize("seed,stereo,th", [(51, False, 1.0), (52, True, 3.0)])
def test_budgeted_projection_search(matcher, oracle, seed, stereo, th):
    """SearchByProjection_Budget (src/Tracking.cc:2240): with a budget that is never reached the result of
    SearchByProjection, with none the first map point that reaches the clock; in between any prefix of the loop, which
    is what the wall clock of the reference yields too."""
    from orb_slam_amd import ORBmatcher
    from test_oracle_search import _fresh, _gf_case
    rng, kps, desc, sf, mp, u_right, obs0 = _gf_case(oracle, seed, stereo, n=3000, m=2500)
    m = ORBmatcher(matcher._ex, nnratio=0.8)
    M = len(mp["isBad"])
    mp["mbTrackInView"][0] = 0
    mp["isBad"][1] = 1
    mp["mTrackProjX"][2] = 5000.0
    for budget in (1e12, 0.0):
        got, ref = _fresh(obs0), _fresh(obs0)
        a = m.SearchByProjection_Budget(kps, desc, u_right, GRID, mp, th, budget, *got)
        b = oracle.search_by_projection_budget(kps, desc, u_right, GRID, mp, th, 0.8, budget, *ref)
        assert a == b
        assert a[1] == M if budget > 0 else 3 <= a[1] < M
        for x, y in zip(got, ref):
            assert np.array_equal(x, y)
    # a budget in between: the loop ended after map point `stop`, and the result is that of the loop over [0, stop]
    got, ref = _fresh(obs0), _fresh(obs0)
    nm, stop = m.SearchByProjection_Budget(kps, desc, u_right, GRID, mp, th, 0.3, *got)
    pre = {k: v[:stop + 1] for k, v in mp.items()}
    nm_ref = oracle.search_by_projection(kps, desc, u_right, GRID, pre, th, 0.8, *ref)
    if stop < M:
        assert nm == nm_ref
        for x, y in zip(got, ref):
            assert np.array_equal(x, y)


# ---- ORBmatcher::SearchByBoW (src/ORBmatcher.cc:270-404, :635-768) ----
@pytest.mark.parametrize("seed", [61, 62, 63])
def test_bow_search(matcher, oracle, seed):
    """Both forms on frames of 2000 keypoints over 100 vocabulary nodes (what levelsup = 4 of a 10^6-word vocabulary
    gives a Frame, src/Frame.cc ComputeBoW): every (keypoint, keypoint) pair of a shared node gets its distance on the
    device; matches and nmatches equal the oracle's with and without the rotation check."""
    from orb_slam_amd import ORBmatcher
    from test_oracle_search import _bow_case
    rng = np.random.default_rng(seed)
    k1, d1, h1, fv1, k2, d2, h2, fv2 = _bow_case(rng, 2000, 2100, 100)
    for ratio, ori in ((0.7, True), (0.75, False), (0.9, True)):
        m = ORBmatcher(matcher._ex, nnratio=ratio, checkOri=ori)
        nm_g, m_g = m.SearchByBoW(k1, d1, h1, fv1, k2, d2, fv2)
        nm_o, m_o = oracle.search_by_bow_frame(k1, d1, h1, fv1, k2, d2, fv2, ratio, ori)
        assert nm_g == nm_o and nm_g > 300 and np.array_equal(m_g, m_o)
        nm_g, m_g = m.SearchByBoW_KeyFrames(k1, d1, h1, fv1, k2, d2, h2, fv2)
        nm_o, m_o = oracle.search_by_bow_keyframes(k1, d1, h1, fv1, k2, d2, h2, fv2, ratio, ori)
        assert nm_g == nm_o and nm_g > 250 and np.array_equal(m_g, m_o)
    # nothing in common, nothing at all
    m = ORBmatcher(matcher._ex, nnratio=0.7)
    shifted = {k + 100000: v for k, v in fv2.items()}
    assert m.SearchByBoW(k1, d1, h1, fv1, k2, d2, shifted)[0] == 0
    assert m.SearchByBoW(k1, d1, h1, {}, k2, d2, fv2)[0] == 0
    assert m.SearchByBoW(k1, d1, np.zeros(len(k1), np.uint8), fv1, k2, d2, fv2)[0] == 0
    nm, mm = m.SearchByBoW(k1[:0], d1[:0], h1[:0], {}, k2, d2, fv2)
    assert nm == 0 and (mm == -1).all()


def test_bow_search_on_the_euroc_pair(oracle, euroc_left, euroc_right):
    """Tracking::TrackReferenceKeyFrame (src/Tracking.cc, ORBmatcher matcher(0.7, true)): the left image as the
    keyframe, the right one as the frame; nodes = a coarse hash of the descriptor standing in for the vocabulary
    (equal on both sides for most true pairs)."""
    from orb_slam_amd import ORBextractor, ORBmatcher
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex, nnratio=0.7, checkOri=True)
    kl, dl = ex(euroc_left)
    kr, dr = ex(euroc_right)

    def nodes_of(d):
        bits = np.unpackbits(d, axis=1)
        return (bits[:, 0:64].sum(1) > 32).astype(np.int64) * 4 + (bits[:, 64:128].sum(1) > 32) * 2 + (bits[:, 128:192].sum(1) > 32)

    def fv_of(d):
        nd = nodes_of(d)
        return {int(v): np.nonzero(nd == v)[0].tolist() for v in np.unique(nd)}

    h = np.ones(len(kl), np.uint8)
    h[::9] = 0
    nm_g, m_g = m.SearchByBoW(kl, dl, h, fv_of(dl), kr, dr, fv_of(dr))
    nm_o, m_o = oracle.search_by_bow_frame(kl, dl, h, fv_of(dl), kr, dr, fv_of(dr), 0.7, True)
    assert nm_g == nm_o and nm_g > 100 and np.array_equal(m_g, m_o)
    ex.close()


# ---- ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1595-1721 ----
@pytest.mark.parametrize("seed,th,orb_dist", [(71, 10.0, 100), (72, 3.0, 64), (73, 10.0, 30)])
def test_relocalisation_projection_search(matcher, oracle, seed, th, orb_dist):
    """Tracking::Relocalization (src/Tracking.cc: SearchByProjection(mCurrentFrame, vpCandidateKFs[i], sFound, 10, 100) and
    (…, 3, 64)) on 3000-keypoint frames with 2500 projected keyframe points"""
    from orb_slam_amd import ORBmatcher
    from test_oracle_search import _keyframe_points
    rng = np.random.default_rng(seed)
    kps, desc, _, _ = _random_frames(rng, 3000, 1, clusters=10)
    kps["octave"] = rng.integers(0, 8, len(kps))
    kf = _keyframe_points(rng, kps, desc, 2500)
    obs0 = rng.choice([-1, -1, -1, 0, 2], len(kps)).astype(np.int32)
    for ori in (True, False):
        m = ORBmatcher(matcher._ex, nnratio=0.9, checkOri=ori)
        g = (obs0.copy(), np.full(len(kps), -1, np.int32))
        o = (obs0.copy(), np.full(len(kps), -1, np.int32))
        nm_g = m.SearchByProjectionKeyFrame(kps, desc, GRID, kf, th, orb_dist, *g)
        nm_o = oracle.search_by_projection_keyframe(kps, desc, GRID, kf, th, orb_dist, ori, *o)
        assert nm_g == nm_o and nm_g > 100
        assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
        assert (g[1][obs0 >= 0] == -1).all()                 # a keypoint that held a map point was never given another
    hidden = dict(kf, visible=np.zeros(2500, np.uint8))
    g = (obs0.copy(), np.full(len(kps), -1, np.int32))
    assert ORBmatcher(matcher._ex).SearchByProjectionKeyFrame(kps, desc, GRID, hidden, th, orb_dist, *g) == 0


# ---- the searches of the mapping and loop-closing threads (src/ORBmatcher.cc:406-518, :937-1212, :1214-1438, :770-935) ----
def _big_window_case(oracle, seed, n=3000, m=2500):
    from test_oracle_search import _projected_points
    rng = np.random.default_rng(seed)
    kps, desc, _, _ = _random_frames(rng, n, 1, clusters=10)
    kps["octave"] = rng.integers(0, 8, len(kps))
    return rng, kps, desc, _projected_points(rng, kps, desc, m)


@pytest.mark.parametrize("seed,th", [(81, 10), (82, 4)])
def test_sim3_projection_search(matcher, oracle, seed, th):
    rng, kps, desc, pts = _big_window_case(oracle, seed)
    matched0 = rng.choice([-1, -1, -1, 1], len(kps)).astype(np.int32)
    g = (matched0.copy(), np.full(len(kps), -1, np.int32))
    o = (matched0.copy(), np.full(len(kps), -1, np.int32))
    nm_g = matcher.SearchByProjectionSim3(kps, desc, GRID, pts, th, *g)
    nm_o = oracle.search_by_projection_sim3(kps, desc, GRID, pts, th, *o)
    assert nm_g == nm_o and nm_g > 300
    assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    hidden = dict(pts, visible=np.zeros(len(pts["u"]), np.uint8))
    assert matcher.SearchByProjectionSim3(kps, desc, GRID, hidden, th, *g) == 0


@pytest.mark.parametrize("seed,stereo,gate,th", [(83, True, True, 3.0), (84, False, True, 3.0), (85, False, False, 4.0),
                                                 (86, True, True, 10.0)])
def test_fuse_search(matcher, oracle, seed, stereo, gate, th):
    rng, kps, desc, pts = _big_window_case(oracle, seed)
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.uniform(2, 40, len(kps)).astype(np.float32), -1).astype(np.float32)
    nf_g, bi_g, bd_g = matcher.FuseSearch(kps, desc, u_right, GRID, pts, th, gate)
    nf_o, bi_o, bd_o = oracle.fuse_search(kps, desc, u_right, GRID, pts, th, gate)
    assert nf_g == nf_o and nf_g > 300
    assert np.array_equal(bi_g, bi_o) and np.array_equal(bd_g, bd_o)
    nf, bi, bd = matcher.FuseSearch(kps[:0], desc[:0], None, GRID, pts, th, gate)
    assert nf == 0 and (bi == -1).all() and (bd == 256).all()


@pytest.mark.parametrize("seed", [87, 88])
def test_sim3_search(matcher, oracle, seed):
    rng = np.random.default_rng(seed)
    k1, d1, k2, d2 = _random_frames(rng, 2500, 2600, clusters=10)
    k1["octave"] = rng.integers(0, 8, len(k1))
    k2["octave"] = rng.integers(0, 8, len(k2))
    n = len(k1)
    k2["x"][:n], k2["y"][:n], k2["octave"][:n] = k1["x"] + 1, k1["y"] - 1, k1["octave"]
    bits = np.unpackbits(d1, axis=1)
    d2[:n] = np.packbits(bits ^ (rng.random(bits.shape) < 0.03), axis=1)
    pts1 = {"visible": (rng.random(len(k1)) < 0.85).astype(np.uint8), "u": k2["x"][:n].copy(), "v": k2["y"][:n].copy(),
            "level": k1["octave"].astype(np.int32), "descriptor": d1}
    pts2 = {"visible": (rng.random(len(k2)) < 0.85).astype(np.uint8),
            "u": np.concatenate([k1["x"], k2["x"][n:]]).astype(np.float32),
            "v": np.concatenate([k1["y"], k2["y"][n:]]).astype(np.float32),
            "level": k2["octave"].astype(np.int32), "descriptor": d2}
    nf_g, m_g = matcher.SearchBySim3(k1, d1, k2, d2, GRID, pts1, pts2, 7.5)
    nf_o, m_o = oracle.search_by_sim3(k1, d1, k2, d2, GRID, pts1, pts2, 7.5)
    assert nf_g == nf_o and nf_g > 800 and np.array_equal(m_g, m_o)


@pytest.mark.parametrize("seed", [89, 90, 91])
def test_triangulation_search(matcher, oracle, seed):
    from orb_slam_amd import ORBmatcher
    from test_oracle_search import _triangulation_case
    rng = np.random.default_rng(seed)
    case = _triangulation_case(rng, 2000, 2100, 100)
    for only_stereo, ori in ((False, True), (True, False), (False, False)):
        m = ORBmatcher(matcher._ex, nnratio=0.6, checkOri=ori)     # LocalMapping::CreateNewMapPoints: ORBmatcher matcher(0.6, false)
        nm_g, m_g = m.SearchForTriangulation(*case, only_stereo)
        nm_o, m_o = oracle.search_for_triangulation(*case, only_stereo, ori)
        assert nm_g == nm_o and np.array_equal(m_g, m_o) and nm_g > (60 if only_stereo else 200)


def test_cpp_matcher_calls(oracle, euroc_left, euroc_right, tmp_path):
    """orbx::ORBmatcher from C++ (tests/cpp/matcher_calls.cpp): SearchByBoW (both forms), SearchForTriangulation,
    SearchByProjection with a Sim3, both Fuse searches, SearchBySim3, the relocalisation search and the budgeted
    local-map search on the EuRoC pair, every output array against the CPU restatement fed the same derived inputs."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(os.environ.get("ORBX_CPP_DIR", os.path.join(root, "tests", "cpp")), "matcher_calls")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "matcher_calls"])
    raw, out = tmp_path / "pair.raw", tmp_path / "calls.bin"
    np.stack([euroc_left, euroc_right]).tofile(raw)
    p = subprocess.run([exe, str(raw), "752", "480", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150)
    assert p.returncode == 0, p.stderr.decode()
    data = np.frombuffer(out.read_bytes(), np.int32)
    blocks, pos = [], 0
    while pos < len(data):
        n = int(data[pos])
        blocks.append(data[pos + 1:pos + 1 + n])
        pos += 1 + n
    it = iter(blocks)
    k1, d1 = oracle.extract(euroc_left)
    k2, d2 = oracle.extract(euroc_right)
    n1, n2 = len(k1), len(k2)
    assert next(it).tolist() == [n1, n2]
    grid = (0.0, 0.0, float(np.float32(64) / np.float32(752)), float(np.float32(48) / np.float32(480)))

    def fv_of(d):
        bits = np.unpackbits(d, axis=1)
        nd = (bits[:, 0:64].sum(1) > 32).astype(np.int64) * 4 + (bits[:, 64:128].sum(1) > 32) * 2 + (bits[:, 128:192].sum(1) > 32)
        return {int(v): np.nonzero(nd == v)[0].tolist() for v in np.unique(nd)}

    fv1, fv2 = fv_of(d1), fv_of(d2)
    i1, i2 = np.arange(n1), np.arange(n2)
    has1, has2 = (i1 % 9 != 0).astype(np.uint8), (i2 % 7 != 0).astype(np.uint8)
    held1, held2 = (i1 % 3 == 0).astype(np.uint8), (i2 % 4 == 0).astype(np.uint8)
    st1, st2 = (i1 % 2).astype(np.uint8), ((i2 // 2) % 2).astype(np.uint8)

    def expect(nm, *arrays):
        assert next(it).tolist() == [nm]
        for a in arrays:
            assert np.array_equal(next(it), a)

    nm, m = oracle.search_by_bow_frame(k1, d1, has1, fv1, k2, d2, fv2, 0.7, True)
    assert nm > 100
    expect(nm, m)
    nm, m = oracle.search_by_bow_keyframes(k1, d1, has1, fv1, k2, d2, has2, fv2, 0.7, True)
    expect(nm, m)
    F = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32)
    nm, m = oracle.search_for_triangulation(k1, d1, held1, st1, fv1, k2, d2, held2, st2, fv2, F, (752 * 0.5 + 0.5, 480 * 0.5), False, True)
    expect(nm, m)

    def pts_of(k, d):
        n = len(k)
        return {"visible": (np.arange(n) % 5 != 0).astype(np.uint8), "u": k["x"].copy(), "v": k["y"].copy(),
                "ur": (k["x"] - np.float32(10)).astype(np.float32), "level": k["octave"].astype(np.int32), "descriptor": d}
    p1, p2 = pts_of(k1, d1), pts_of(k2, d2)
    matched = np.where(i2 % 4 == 0, 1, -1).astype(np.int32)
    match = np.full(n2, -1, np.int32)
    nm = oracle.search_by_projection_sim3(k2, d2, grid, p1, 10, matched, match)
    expect(nm, matched, match)
    ur2 = np.where(i2 % 2 == 1, k2["x"] - np.float32(10), np.float32(-1)).astype(np.float32)
    nf, bi, bd = oracle.fuse_search(k2, d2, ur2, grid, p1, 3.0, True)
    expect(nf, bi, bd)
    nf, bi, bd = oracle.fuse_search(k2, d2, None, grid, p1, 4.0, False)
    expect(nf, bi, bd)
    nf, m = oracle.search_by_sim3(k1, d1, k2, d2, grid, p1, p2, 7.5)
    assert nf > 100
    expect(nf, m)
    kf = {"visible": p1["visible"], "u": p1["u"], "v": p1["v"], "level": p1["level"], "angle": k1["angle"].astype(np.float32),
          "descriptor": d1, "Observations": np.ones(n1, np.int32)}
    fobs = np.where(i2 % 6 == 0, 0, -1).astype(np.int32)
    fmatch = np.full(n2, -1, np.int32)
    nm = oracle.search_by_projection_keyframe(k2, d2, grid, kf, 10.0, 64, True, fobs, fmatch)
    expect(nm, fobs, fmatch)
    mp = {"mbTrackInView": np.ones(n1, np.uint8), "isBad": np.zeros(n1, np.uint8), "mTrackProjX": k1["x"].copy(),
          "mTrackProjY": k1["y"].copy(), "mTrackProjXR": (k1["x"] - np.float32(10)).astype(np.float32),
          "mnTrackScaleLevel": k1["octave"].astype(np.int32),
          "mTrackViewCos": np.where(i1 % 2 == 1, np.float32(0.9995), np.float32(0.9)).astype(np.float32),
          "Observations": np.ones(n1, np.int32), "descriptor": d1}
    o2, m2, s2 = (np.full(n2, -1, np.int32) for _ in range(3))
    nm, stop = oracle.search_by_projection_budget(k2, d2, ur2, grid, mp, 3.0, 0.7, 1e12, o2, m2, s2)
    expect(nm)
    assert next(it).tolist() == [stop] and stop == n1
    assert np.array_equal(next(it), m2) and np.array_equal(next(it), s2)
    assert next(it, None) is None
