"""The oracle's restatement of the two search routines with sequential exclusion
(ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:520-633; ORBmatcher::SearchByProjection(Frame&,
vector<MapPoint*>&, th), :155-241): hand-computed known answers for every rule of the loops, and a second,
independent statement of each loop in plain Python (per-query GetFeaturesInArea by brute force over all
keypoints, no CSR lists) that must give the same arrays on random frames.  No GPU."""
import numpy as np
import pytest

from oracle_lib import KP_DTYPE

GRID = (0.0, 0.0, float(np.float32(64) / np.float32(752)), float(np.float32(48) / np.float32(480)))


def _kps(xy, octave=0, angle=0.0):
    k = np.zeros(len(xy), KP_DTYPE)
    k["x"] = [p[0] for p in xy]
    k["y"] = [p[1] for p in xy]
    k["size"] = 31
    k["angle"] = angle
    k["octave"] = octave
    k["class_id"] = -1
    return k


def _desc_with_bits(nbits_list):
    """descriptors with the given number of leading one-bits: distance between two of them = |a - b|"""
    d = np.zeros((len(nbits_list), 32), np.uint8)
    for i, n in enumerate(nbits_list):
        bits = np.zeros(256, np.uint8)
        bits[:n] = 1
        d[i] = np.packbits(bits, bitorder="little")
    return d


def test_three_maxima(oracle):
    # src/ORBmatcher.cc:1723-1764: the three largest bins, the 2nd / 3rd dropped below 10 % of the largest
    assert oracle.three_maxima([0] * 30) == (-1, -1, -1)
    s = [0] * 30
    s[4], s[9], s[20] = 50, 30, 10
    assert oracle.three_maxima(s) == (4, 9, 20)
    s[20] = 4                         # 4 < 0.1 * 50
    assert oracle.three_maxima(s) == (4, 9, -1)
    s[9] = 4
    assert oracle.three_maxima(s) == (4, -1, -1)
    # ties: strict '>' keeps the first bin of equal size in front
    s = [0] * 30
    s[3] = s[7] = s[8] = s[9] = 5
    assert oracle.three_maxima(s) == (3, 7, 8)
    # a later larger bin shifts the earlier ones down
    s = [1, 2, 3, 4] + [0] * 26
    assert oracle.three_maxima(s) == (3, 2, 1)


def test_init_search_rules(oracle):
    # F2: three keypoints close together; F1: queries all looking at the same window
    k2 = _kps([(100, 100), (104, 100), (300, 300)])
    d2 = _desc_with_bits([0, 40, 0])
    # query 0: distances 10 / 30 to the first two -> best 10 (idx 0), second 30; 10 < 0.9 * 30 -> match
    # query 1: distances 5 / 35: candidate 0 is held at 10 > 5 -> steals it, query 0 is un-matched
    # query 2: distance 5 to candidate 0 as well: vMatchedDistance[0] = 5 <= 5 -> candidate skipped; best = 35 (idx 1)
    #          second = INT_MAX -> 35 <= 50 and 35 < 0.9 * INT_MAX -> matches idx 1
    # query 3: on level 1 -> skipped; query 4: window holds no keypoint
    k1 = _kps([(100, 100)] * 5, octave=[0, 0, 0, 1, 0])
    d1 = _desc_with_bits([10, 5, 5, 0, 0])
    prev = np.array([[101, 101]] * 4 + [[600, 50]], np.float32)
    nm, m12 = oracle.search_for_initialization(k1, d1, k2, d2, GRID, prev, 10, nnratio=0.9, check_ori=False)
    assert m12.tolist() == [-1, 0, 1, -1, -1] and nm == 2
    # vbPrevMatched follows the matches (:628-630), the rest keeps its value
    assert prev.tolist() == [[101, 101], [100, 100], [104, 100], [101, 101], [600, 50]]
    # TH_LOW: a best distance of 51 is refused, 50 accepted (:574)
    for bits, want in ((51, -1), (50, 0)):
        prev = np.array([[100, 100]], np.float32)
        nm, m12 = oracle.search_for_initialization(_kps([(100, 100)]), _desc_with_bits([bits]), k2[:1], d2[:1], GRID, prev,
                                                   10, 0.9, False)
        assert m12.tolist() == [want] and nm == (want >= 0)
    # ratio test is strict and in float: best 27, second 30, ratio 0.9 -> 27 < 27.0 fails
    k2b = _kps([(100, 100), (104, 100)])
    prev = np.array([[100, 100]], np.float32)
    nm, m12 = oracle.search_for_initialization(_kps([(100, 100)]), _desc_with_bits([27]), k2b, _desc_with_bits([0, 57]), GRID,
                                               prev, 10, 0.9, False)
    assert m12.tolist() == [-1] and nm == 0
    # equal distances: the first candidate in list order is kept as best, the other becomes the second (:562-571)
    prev = np.array([[100, 100]], np.float32)
    nm, m12 = oracle.search_for_initialization(_kps([(100, 100)]), _desc_with_bits([10]), k2b, _desc_with_bits([0, 20]), GRID,
                                               prev, 10, 0.9, False)
    assert m12.tolist() == [-1]       # 10 < 0.9 * 10 fails
    # window: |dx| < r strictly (src/Frame.cc:715)
    prev = np.array([[110, 100]], np.float32)
    nm, m12 = oracle.search_for_initialization(_kps([(0, 0)]), _desc_with_bits([0]), k2[:1], d2[:1], GRID, prev, 10, 0.9, False)
    assert nm == 0
    prev = np.array([[109.5, 100]], np.float32)
    nm, m12 = oracle.search_for_initialization(_kps([(0, 0)]), _desc_with_bits([0]), k2[:1], d2[:1], GRID, prev, 10, 0.9, False)
    assert nm == 1
    # level filter of GetFeaturesInArea(…, 0, 0): a candidate on level 1 is not seen
    k2c = _kps([(100, 100)], octave=1)
    prev = np.array([[100, 100]], np.float32)
    nm, m12 = oracle.search_for_initialization(_kps([(0, 0)]), _desc_with_bits([0]), k2c, d2[:1], GRID, prev, 10, 0.9, False)
    assert nm == 0


def test_init_search_rotation_histogram(oracle):
    # 10 matches with rotation 0 deg, 2 with 95 deg, 1 with 200 deg, 1 with 300 deg.  bin = round(rot / 30):
    # bins 0 (10), 3 (2), 7 (1), 10 (1).  Three maxima: 0, 3, 7 (first of the equal ones); 1 < 0.1 * 10 is false,
    # so the third stays.  The match in bin 10 is removed.
    n = 14
    xy = [(40 + 40 * i, 60 + 20 * (i % 3)) for i in range(n)]
    rot = [0.0] * 10 + [95.0, 95.0, 200.0, 300.0]
    k2 = _kps(xy, angle=0.0)
    k1 = _kps(xy, angle=rot)
    rng = np.random.default_rng(5)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    prev = np.array(xy, np.float32)
    nm, m12 = oracle.search_for_initialization(k1, d, k2, d, GRID, prev.copy(), 5, 0.9, True)
    assert m12.tolist() == list(range(13)) + [-1] and nm == 13
    nm, m12 = oracle.search_for_initialization(k1, d, k2, d, GRID, prev.copy(), 5, 0.9, False)
    assert m12.tolist() == list(range(14)) and nm == 14
    # with 12 matches in the first bin the third maximum (1 < 0.1 * 12) is dropped as well (:1760-1763)
    xy2 = xy + [(60 + 40 * i, 200) for i in range(2)]
    k2c, k1c = _kps(xy2, angle=0.0), _kps(xy2, angle=rot + [0.0, 0.0])
    d2c = rng.integers(0, 256, (16, 32), dtype=np.uint8)
    nm, m12 = oracle.search_for_initialization(k1c, d2c, k2c, d2c, GRID, np.array(xy2, np.float32), 5, 0.9, True)
    assert m12.tolist() == list(range(12)) + [-1, -1, 14, 15] and nm == 14
    # negative difference wraps by 360 (:591-592): angle1 = 10, angle2 = 350 -> rot = 20 -> bin round(0.667) = 1
    k1b, k2b = _kps(xy, angle=10.0), _kps(xy, angle=350.0)
    k1b["angle"][0] = 350.0      # rot 0 -> bin 0, alone: second maximum, but 1 < 0.1 * 13 drops it
    nm, m12 = oracle.search_for_initialization(k1b, d, k2b, d, GRID, prev.copy(), 5, 0.9, True)
    assert nm == 13 and m12[0] == -1


def _py_features_in_area(kps, grid, x, y, r, min_level, max_level):
    """Frame::GetFeaturesInArea (src/Frame.cc:669-722) by its definition: cells in column-major window
    order, keypoints of a cell in index order (PosInGrid :724-734)"""
    min_x, min_y, inv_w, inv_h = (np.float32(v) for v in grid)
    x, y, r = np.float32(x), np.float32(y), np.float32(r)
    cx0 = max(0, int(np.floor((x - min_x - r) * inv_w)))
    cx1 = min(63, int(np.ceil((x - min_x + r) * inv_w)))
    cy0 = max(0, int(np.floor((y - min_y - r) * inv_h)))
    cy1 = min(47, int(np.ceil((y - min_y + r) * inv_h)))
    if cx0 >= 64 or cx1 < 0 or cy0 >= 48 or cy1 < 0:
        return []
    px = np.round((kps["x"] - min_x) * inv_w).astype(int)      # all test coordinates are away from .5 cell positions
    py = np.round((kps["y"] - min_y) * inv_h).astype(int)
    out = []
    check = min_level > 0 or max_level >= 0
    for ix in range(cx0, cx1 + 1):
        for iy in range(cy0, cy1 + 1):
            for i in np.nonzero((px == ix) & (py == iy))[0]:
                if check and (kps["octave"][i] < min_level or (max_level >= 0 and kps["octave"][i] > max_level)):
                    continue
                if abs(kps["x"][i] - x) < r and abs(kps["y"][i] - y) < r:
                    out.append(int(i))
    return out


def _ham(a, b):
    return int(np.unpackbits(a ^ b).sum())


def _py_search_for_initialization(k1, d1, k2, d2, grid, prev, window, nnratio, check_ori):
    n1, n2 = len(k1), len(k2)
    m12 = [-1] * n1
    m21 = [-1] * n2
    md = [2 ** 31 - 1] * n2
    hist = [[] for _ in range(30)]
    nm = 0
    for i1 in range(n1):
        if k1["octave"][i1] > 0:
            continue
        cands = _py_features_in_area(k2, grid, prev[i1, 0], prev[i1, 1], window, 0, 0)
        best, best2, bi = 2 ** 31 - 1, 2 ** 31 - 1, -1
        for i2 in cands:
            dist = _ham(d1[i1], d2[i2])
            if md[i2] <= dist:
                continue
            if dist < best:
                best2, best, bi = best, dist, i2
            elif dist < best2:
                best2 = dist
        if best <= 50 and np.float32(best) < np.float32(best2) * np.float32(nnratio):
            if m21[bi] >= 0:
                m12[m21[bi]] = -1
                nm -= 1
            m12[i1], m21[bi], md[bi] = bi, i1, best
            nm += 1
            if check_ori:
                rot = np.float32(k1["angle"][i1]) - np.float32(k2["angle"][bi])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360))
                v = float(np.float32(rot * np.float32(1.0 / 30)))
                b = int(np.floor(v + 0.5))          # C round(): half away from zero, v >= 0 here
                if b == 30:
                    b = 0
                hist[b].append(i1)
    if check_ori:
        sizes = [len(h) for h in hist]
        order = sorted(range(30), key=lambda i: (-sizes[i], i))[:3]
        keep = [order[0]] if sizes[order[0]] > 0 else []
        for o in order[1:]:
            if sizes[o] > 0 and not (np.float32(sizes[o]) < np.float32(0.1) * np.float32(sizes[order[0]])):
                keep.append(o)
            else:
                break
        for i in range(30):
            if i in keep:
                continue
            for i1 in hist[i]:
                if m12[i1] >= 0:
                    m12[i1] = -1
                    nm -= 1
    for i1 in range(n1):
        if m12[i1] >= 0:
            prev[i1] = (k2["x"][m12[i1]], k2["y"][m12[i1]])
    return nm, m12


def _random_frames(rng, n1, n2, clusters):
    """two keypoint sets that share structure: F2 = F1 moved a little, descriptors = a few prototypes with
    bit flips, so that windows hold several candidates at close (and equal) distances and claims collide"""
    protos = rng.integers(0, 256, (clusters, 32), dtype=np.uint8)

    def descs(n):
        d = protos[rng.integers(0, clusters, n)].copy()
        bits = np.unpackbits(d, axis=1)
        flip = rng.random(bits.shape) < 0.03
        return np.packbits(bits ^ flip, axis=1)

    def keys(n):
        k = np.zeros(n, KP_DTYPE)
        k["x"] = (rng.integers(0, 7500, n) / 10 + 0.03).astype(np.float32)
        k["y"] = (rng.integers(0, 4780, n) / 10 + 0.03).astype(np.float32)
        k["angle"] = (rng.random(n) * 360).astype(np.float32)
        k["octave"] = rng.choice([0, 0, 0, 1, 2], n)
        k["size"] = 31
        k["class_id"] = -1
        return k

    return keys(n1), descs(n1), keys(n2), descs(n2)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_init_search_against_python_statement(oracle, seed):
    rng = np.random.default_rng(seed)
    k1, d1, k2, d2 = _random_frames(rng, 300, 350, clusters=25)
    # crowd F2 around F1's points so that the windows are busy
    k2["x"][:300] = k1["x"] + rng.integers(-8, 9, 300).astype(np.float32)
    k2["y"][:300] = k1["y"] + rng.integers(-8, 9, 300).astype(np.float32)
    # true correspondences (a few bits flipped, most of them rotated alike) among look-alikes of the same cluster
    bits = np.unpackbits(d1, axis=1)
    d2[:300] = np.packbits(bits ^ (rng.random(bits.shape) < 0.02), axis=1)
    k2["angle"][:300] = ((k1["angle"] + rng.choice([0.0, 0.0, 0.0, 3.0, 100.0, 170.0], 300)) % 360).astype(np.float32)
    k2["octave"][:300] = np.where(rng.random(300) < 0.85, 0, k2["octave"][:300])
    for check_ori in (False, True):
        prev_o = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
        prev_p = prev_o.copy()
        nm_o, m_o = oracle.search_for_initialization(k1, d1, k2, d2, GRID, prev_o, 40, 0.9, check_ori)
        nm_p, m_p = _py_search_for_initialization(k1, d1, k2, d2, GRID, prev_p, 40, 0.9, check_ori)
        assert nm_o == nm_p and m_o.tolist() == m_p
        assert np.array_equal(prev_o, prev_p)
        assert nm_o > 40                         # the case is not vacuous
    # steals did happen: some F1 keypoint with candidates ends unmatched although it was best for someone


def _map_points(rng, kps, desc, M, sf):
    """map points projected near keypoints of the frame, descriptors = those keypoints' with bit flips"""
    pick = rng.integers(0, len(kps), M)
    lvl = np.clip(kps["octave"][pick] + rng.integers(0, 2, M), 0, 7).astype(np.int32)
    mp = {
        "mbTrackInView": (rng.random(M) < 0.9).astype(np.uint8),
        "isBad": (rng.random(M) < 0.05).astype(np.uint8),
        "mTrackProjX": (kps["x"][pick] + rng.normal(0, 1.5, M)).astype(np.float32),
        "mTrackProjY": (kps["y"][pick] + rng.normal(0, 1.5, M)).astype(np.float32),
        "mnTrackScaleLevel": lvl,
        "mTrackViewCos": rng.choice([0.9, 0.9985, 0.999, 0.5], M).astype(np.float32),
        "Observations": rng.choice([0, 1, 3, 7], M).astype(np.int32),
    }
    mp["mTrackProjXR"] = (mp["mTrackProjX"] - rng.random(M).astype(np.float32) * 30).astype(np.float32)
    bits = np.unpackbits(desc[pick], axis=1)
    flip = rng.random(bits.shape) < 0.06
    mp["descriptor"] = np.packbits(bits ^ flip, axis=1)
    return mp


def _py_search_by_projection(kps, desc, u_right, grid, mp, th, nnratio, sf, frame_obs, frame_match, match_score,
                             order=None, budget0=False, extra=None):
    """order: the map points to visit (default: all, ascending -- the loop of SearchByProjection; any other order is a
    sequence of SearchByProjection_OnePoint calls); budget0: end after the first point that reaches the end of the loop
    body (SearchByProjection_Budget with no time at all); extra: dict that receives 'returns' (bestIdx or -1 per visited
    point) and 'stopped_at'"""
    nm = 0
    if extra is not None:
        extra["returns"], extra["stopped_at"] = [], len(mp["mbTrackInView"])
    for i in (range(len(mp["mbTrackInView"])) if order is None else order):
        if extra is not None:
            extra["returns"].append(-1)
        if not mp["mbTrackInView"][i] or mp["isBad"][i]:
            continue
        lvl = int(mp["mnTrackScaleLevel"][i])
        r = np.float32(2.5) if float(mp["mTrackViewCos"][i]) > 0.998 else np.float32(4.0)
        if th != 1.0:
            r = np.float32(r * np.float32(th))
        win = np.float32(r * sf[lvl])
        cands = _py_features_in_area(kps, grid, mp["mTrackProjX"][i], mp["mTrackProjY"][i], win, lvl - 1, lvl)
        if len(cands) == 0:
            continue
        best, best2, bl, bl2, bi = 256, 256, -1, -1, -1
        for idx in cands:
            if frame_obs[idx] > 0:
                continue
            if u_right is not None and u_right[idx] > 0:
                if abs(np.float32(mp["mTrackProjXR"][i]) - np.float32(u_right[idx])) > win:
                    continue
            dist = _ham(mp["descriptor"][i], desc[idx])
            if dist < best:
                best2, best, bl2, bl, bi = best, dist, bl, int(kps["octave"][idx]), idx
            elif dist < best2:
                bl2, best2 = int(kps["octave"][idx]), dist
        if best <= 100:
            if bl == bl2 and np.float32(best) > np.float32(nnratio) * np.float32(best2):
                continue
            frame_match[bi] = i
            frame_obs[bi] = mp["Observations"][i]
            match_score[bi] = best
            nm += 1
            if extra is not None:
                extra["returns"][-1] = int(bi)
        if budget0:
            if extra is not None:
                extra["stopped_at"] = i
            break
    return nm


@pytest.mark.parametrize("seed,stereo,th", [(1, False, 1.0), (2, True, 1.0), (3, True, 3.0), (4, False, 5.0)])
def test_projection_search_against_python_statement(oracle, seed, stereo, th):
    rng = np.random.default_rng(100 + seed)
    kps, desc, _, _ = _random_frames(rng, 400, 1, clusters=8)
    kps["octave"] = rng.integers(0, 8, len(kps))
    sf = oracle.tables()["scale"]
    mp = _map_points(rng, kps, desc, 300, sf)
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.random(len(kps)).astype(np.float32) * 30, -1).astype(np.float32)
    obs0 = rng.choice([-1, -1, -1, 0, 2], len(kps)).astype(np.int32)
    outs = []
    for fn in ("oracle", "python"):
        fo, fm, ms = obs0.copy(), np.full(len(kps), -1, np.int32), np.full(len(kps), -1, np.int32)
        if fn == "oracle":
            nm = oracle.search_by_projection(kps, desc, u_right, GRID, mp, th, 0.8, fo, fm, ms)
        else:
            nm = _py_search_by_projection(kps, desc, u_right, GRID, mp, th, 0.8, sf, fo, fm, ms)
        outs.append((nm, fo, fm, ms))
    assert outs[0][0] == outs[1][0] and outs[0][0] > 30
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)
    # exclusion did matter: fewer keypoints hold a match than matches were counted, or a map point with
    # observations blocks later ones -- at least one keypoint was claimed by a point with Observations() > 0
    assert (outs[0][1][outs[0][2] >= 0] > 0).any()


def test_projection_search_rules(oracle):
    sf = oracle.tables()["scale"]
    kps = _kps([(100, 100), (102, 100), (400, 300)], octave=[1, 1, 1])
    desc = _desc_with_bits([0, 30, 0])

    def run(mp_bits, obs, frame_obs=(-1, -1, -1), u_right=None, level=1, xr=0.0, ratio=0.8, th=1.0, view_cos=0.9):
        M = len(mp_bits)
        mp = {"mbTrackInView": np.ones(M, np.uint8), "isBad": np.zeros(M, np.uint8),
              "mTrackProjX": np.full(M, 101, np.float32), "mTrackProjY": np.full(M, 100, np.float32),
              "mTrackProjXR": np.full(M, xr, np.float32), "mnTrackScaleLevel": np.full(M, level, np.int32),
              "mTrackViewCos": np.full(M, view_cos, np.float32), "Observations": np.array(obs, np.int32),
              "descriptor": _desc_with_bits(mp_bits)}
        fo = np.array(frame_obs, np.int32)
        fm, ms = np.full(3, -1, np.int32), np.full(3, -1, np.int32)
        nm = oracle.search_by_projection(kps, desc, u_right, GRID, mp, th, ratio, fo, fm, ms)
        return nm, fo.tolist(), fm.tolist(), ms.tolist()

    # one map point at distance 5 / 25 from the two candidates (same level): 5 <= 0.8 * 25 -> keypoint 0
    assert run([5], [3]) == (1, [3, -1, -1], [0, -1, -1], [5, -1, -1])
    # ratio test fails only when both are on the same level: 22 > 0.8 * 8?  dist 22 / 8 -> best 8 (kp 1), second 22
    assert run([22], [3])[2] == [-1, 0, -1]
    assert run([16], [3], ratio=0.8)[0] == 0                 # 14 vs 16: 14 > 12.8 -> rejected
    # a second map point cannot take a keypoint claimed by a point with observations (:197-199) ...
    assert run([5, 4], [3, 3]) == (2, [3, 3, -1], [0, 1, -1], [5, 26, -1])
    # ... but overwrites one claimed by a point without observations, and both count (:233-236)
    assert run([5, 4], [0, 3]) == (2, [3, -1, -1], [1, -1, -1], [4, -1, -1])
    # a keypoint that already holds a map point with observations is never a candidate
    assert run([5], [3], frame_obs=(2, -1, -1)) == (1, [2, 3, -1], [-1, 0, -1], [-1, 25, -1])
    # TH_HIGH = 100 (:228)
    kps["x"][1] = 300
    assert run([100], [1])[0] == 1 and run([101], [1])[0] == 0
    kps["x"][1] = 102
    # level window [level - 1, level]: candidates on level 1 are seen from predicted levels 1 and 2 only
    assert run([5], [1], level=2)[0] == 1 and run([5], [1], level=3)[0] == 0 and run([5], [1], level=0)[0] == 0
    # right-image gate (:201-206): |projXR - uR| > r * scale -> skipped; r = 4 (cos 0.9), scale[1] = 1.2 -> 4.8
    ur = np.array([80.0, -1.0, -1.0], np.float32)
    assert run([5], [1], u_right=ur, xr=84.5)[2] == [0, -1, -1]
    assert run([5], [1], u_right=ur, xr=85.0)[2] == [-1, 0, -1]
    # radius: 2.5 for cos > 0.998, times th when th != 1
    kps["x"][0], kps["x"][1] = 97.5, 300
    assert run([5], [1], view_cos=0.999)[0] == 0             # |97.5 - 101| = 3.5 >= 2.5 * 1.2 = 3.0
    assert run([5], [1], view_cos=0.9)[0] == 1               # 3.5 < 4.8
    assert run([5], [1], view_cos=0.999, th=1.5)[0] == 1     # 3.5 < 4.5


# ---- ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, numVisibleMpt), src/ORBmatcher.cc:1440-1593 ----
def _last_frame_points(rng, kps, desc, M):
    """map points of the last frame projected near keypoints of the current one"""
    pick = rng.integers(0, len(kps), M)
    last = {
        "visible": (rng.random(M) < 0.9).astype(np.uint8),
        "u": (kps["x"][pick] + rng.normal(0, 2.0, M)).astype(np.float32),
        "v": (kps["y"][pick] + rng.normal(0, 2.0, M)).astype(np.float32),
        "invzc": rng.uniform(0.02, 1.0, M).astype(np.float32),
        "octave": np.clip(kps["octave"][pick] + rng.integers(-1, 2, M), 0, 7).astype(np.int32),
        "angle": ((kps["angle"][pick] + rng.choice([0.0, 0.0, 0.0, 1.5, 60.0, 200.0], M)) % 360).astype(np.float32),
        "Observations": rng.choice([0, 1, 4], M).astype(np.int32),
    }
    bits = np.unpackbits(desc[pick], axis=1)
    last["descriptor"] = np.packbits(bits ^ (rng.random(bits.shape) < 0.05), axis=1)
    return last


def _py_search_by_projection_last(kps, desc, u_right, grid, last, th, mbf, fwd, bwd, check_ori, sf, frame_obs, frame_match):
    nm = 0
    hist = [[] for _ in range(30)]
    for i in range(len(last["visible"])):
        if not last["visible"][i]:
            continue
        o = int(last["octave"][i])
        radius = np.float32(np.float32(th) * sf[o])
        lv = (o, -1) if fwd else ((0, o) if bwd else (o - 1, o + 1))
        cands = _py_features_in_area(kps, grid, last["u"][i], last["v"][i], radius, lv[0], lv[1])
        best, bi = 256, -1
        for i2 in cands:
            if frame_obs[i2] > 0:
                continue
            if u_right is not None and u_right[i2] > 0:
                ur = np.float32(last["u"][i] - np.float32(np.float32(mbf) * last["invzc"][i]))
                if abs(np.float32(ur - u_right[i2])) > radius:
                    continue
            d = _ham(last["descriptor"][i], desc[i2])
            if d < best:
                best, bi = d, i2
        if best <= 100:
            frame_match[bi] = i
            frame_obs[bi] = last["Observations"][i]
            nm += 1
            if check_ori:
                rot = np.float32(last["angle"][i]) - np.float32(kps["angle"][bi])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360))
                b = int(np.floor(float(np.float32(rot * np.float32(1.0 / 30))) + 0.5))
                hist[0 if b == 30 else b].append(bi)
    if check_ori:
        sizes = [len(h) for h in hist]
        order = sorted(range(30), key=lambda i: (-sizes[i], i))[:3]
        keep = [order[0]] if sizes[order[0]] > 0 else []
        for o_ in order[1:]:
            if sizes[o_] > 0 and not (np.float32(sizes[o_]) < np.float32(0.1) * np.float32(sizes[order[0]])):
                keep.append(o_)
            else:
                break
        for b in range(30):
            if b not in keep:
                for bi in hist[b]:
                    frame_match[bi] = -1
                    frame_obs[bi] = -1
                    nm -= 1
    return nm


@pytest.mark.parametrize("seed,stereo,fwd,bwd,th", [(1, False, False, False, 15.0), (2, True, True, False, 7.0),
                                                   (3, True, False, True, 7.0), (4, True, False, False, 14.0)])
def test_last_frame_projection_search_against_python_statement(oracle, seed, stereo, fwd, bwd, th):
    rng = np.random.default_rng(200 + seed)
    kps, desc, _, _ = _random_frames(rng, 500, 1, clusters=10)
    kps["octave"] = rng.integers(0, 8, len(kps))
    sf = oracle.tables()["scale"]
    last = _last_frame_points(rng, kps, desc, 350)
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.random(len(kps)).astype(np.float32) * 40, -1).astype(np.float32)
    obs0 = rng.choice([-1, -1, -1, 0, 3], len(kps)).astype(np.int32)
    for ori in (True, False):
        outs = []
        for side in ("oracle", "python"):
            fo, fm = obs0.copy(), np.full(len(kps), -1, np.int32)
            if side == "oracle":
                nm = oracle.search_by_projection_last(kps, desc, u_right, GRID, last, th, 40.0, fwd, bwd, ori, fo, fm)
            else:
                nm = _py_search_by_projection_last(kps, desc, u_right, GRID, last, th, 40.0, fwd, bwd, ori, sf, fo, fm)
            outs.append((nm, fo, fm))
        assert outs[0][0] == outs[1][0] and outs[0][0] > 30
        assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def test_last_frame_projection_search_rules(oracle):
    kps = _kps([(100, 100), (103, 100), (400, 300)], octave=[2, 3, 2], angle=[10.0, 10.0, 10.0])
    desc = _desc_with_bits([0, 30, 0])

    def run(bits, obs, octave=2, fwd=False, bwd=False, frame_obs=(-1, -1, -1), u_right=None, invzc=0.5, th=3.0, angle=10.0, ori=False):
        M = len(bits)
        last = {"visible": np.ones(M, np.uint8), "u": np.full(M, 101, np.float32), "v": np.full(M, 100, np.float32),
                "invzc": np.full(M, invzc, np.float32), "octave": np.full(M, octave, np.int32),
                "angle": np.full(M, angle, np.float32), "Observations": np.array(obs, np.int32),
                "descriptor": _desc_with_bits(bits)}
        fo, fm = np.array(frame_obs, np.int32), np.full(3, -1, np.int32)
        nm = oracle.search_by_projection_last(kps, desc, u_right, GRID, last, th, 40.0, fwd, bwd, ori, fo, fm)
        return nm, fo.tolist(), fm.tolist()

    # radius = th * scale[2] = 3 * 1.44 = 4.32: both near keypoints are candidates (levels 1..3); the best one wins, no ratio test
    assert run([12], [2]) == (1, [2, -1, -1], [0, -1, -1])
    assert run([20], [2]) == (1, [-1, 2, -1], [-1, 0, -1])            # 20 vs 10 -> keypoint 1
    assert run([15], [2])[2] == [0, -1, -1]                             # equal distances: the first candidate stays (strict <)
    # forward motion: levels >= the last octave; backward: levels <= it
    assert run([20], [2], octave=3, fwd=True)[2] == [-1, 0, -1]
    assert run([1], [2], octave=3, fwd=True)[2] == [-1, 0, -1]         # keypoint 0 (level 2) is not seen at all
    assert run([29], [2], octave=2, bwd=True)[2] == [0, -1, -1]        # keypoint 1 (level 3) is not seen
    # a second point cannot take a keypoint whose map point has observations, but does overwrite one without
    assert run([12, 11], [2, 2]) == (2, [2, 2, -1], [0, 1, -1])
    assert run([12, 11], [0, 2]) == (2, [2, -1, -1], [1, -1, -1])
    # TH_HIGH
    kps["x"][1] = 300
    assert run([100], [1])[0] == 1 and run([101], [1])[0] == 0
    kps["x"][1] = 103
    # right-image gate: ur = u - mbf * invzc = 101 - 20 = 81; |81 - uR| > radius 4.32 -> skipped
    ur = np.array([86.0, -1.0, -1.0], np.float32)
    assert run([12], [1], u_right=ur)[2] == [-1, 0, -1]
    ur[0] = 85.0
    assert run([12], [1], u_right=ur)[2] == [0, -1, -1]
    # rotation check: two matches in bin 0 and one in bin 4 (rot 120): 1 >= 0.1 * 2 keeps it; with 11 in bin 0 it goes
    kq = _kps([(50 + 30 * i, 60) for i in range(12)], octave=2, angle=0.0)
    dq = np.random.default_rng(1).integers(0, 256, (12, 32), dtype=np.uint8)
    last = {"visible": np.ones(12, np.uint8), "u": kq["x"].copy(), "v": kq["y"].copy(), "invzc": np.full(12, 0.5, np.float32),
            "octave": np.full(12, 2, np.int32), "angle": np.array([0.0] * 11 + [120.0], np.float32),
            "Observations": np.full(12, 1, np.int32), "descriptor": dq}
    fo, fm = np.full(12, -1, np.int32), np.full(12, -1, np.int32)
    assert oracle.search_by_projection_last(kq, dq, None, GRID, last, 3.0, 40.0, False, False, True, fo, fm) == 11
    assert fm.tolist() == list(range(11)) + [-1] and fo.tolist() == [1] * 11 + [-1]


# ---- the good-feature forms: SearchByProjection_OnePoint / GetCandidates + MatchCandidates (include/ORBmatcher.h:68-252),
# ---- SearchByProjection_Budget (src/ORBmatcher.cc:45-153)
def _gf_case(oracle, seed, stereo, n=400, m=300):
    rng = np.random.default_rng(300 + seed)
    kps, desc, _, _ = _random_frames(rng, n, 1, clusters=8)
    kps["octave"] = rng.integers(0, 8, len(kps))
    sf = oracle.tables()["scale"]
    mp = _map_points(rng, kps, desc, m, sf)
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.random(len(kps)).astype(np.float32) * 30, -1).astype(np.float32)
    obs0 = rng.choice([-1, -1, -1, 0, 2], len(kps)).astype(np.int32)
    return rng, kps, desc, sf, mp, u_right, obs0


def _fresh(obs0):
    return obs0.copy(), np.full(len(obs0), -1, np.int32), np.full(len(obs0), -1, np.int32)


@pytest.mark.parametrize("seed,stereo,th", [(1, False, 1.0), (2, True, 3.0), (3, True, 1.0)])
def test_one_point_forms_in_any_order(oracle, seed, stereo, th):
    """The active-matching loop of the reference draws map points in the order of its heap (src/Observability.cc:
    1404-1413): SearchByProjection_OnePoint per drawn point, each seeing what the earlier ones assigned.  In ascending
    order the calls add up to SearchByProjection; in a shuffled order they equal the Python statement run in that
    order; GetCandidates + MatchCandidates equal SearchByProjection_OnePoint."""
    rng, kps, desc, sf, mp, u_right, obs0 = _gf_case(oracle, seed, stereo)
    M = len(mp["isBad"])
    ref = _fresh(obs0)
    nm_ref = oracle.search_by_projection(kps, desc, u_right, GRID, mp, th, 0.8, *ref)
    one = _fresh(obs0)
    rets = [oracle.search_by_projection_one_point(kps, desc, u_right, GRID, mp, i, th, 0.8, *one) for i in range(M)]
    assert sum(r >= 0 for r in rets) == nm_ref and nm_ref > 30
    for a, b in zip(ref, one):
        assert np.array_equal(a, b)
    order = rng.permutation(M)[: M * 2 // 3]
    shuffled, split, py = _fresh(obs0), _fresh(obs0), _fresh(obs0)
    r1 = [oracle.search_by_projection_one_point(kps, desc, u_right, GRID, mp, int(i), th, 0.8, *shuffled) for i in order]
    r2 = []
    for i in order:
        cand = oracle.get_candidates(kps, GRID, mp, int(i), th)
        r2.append(oracle.match_candidates(kps, desc, u_right, cand, mp, int(i), th, 0.8, *split))
    extra = {}
    _py_search_by_projection(kps, desc, u_right, GRID, mp, th, 0.8, sf, *py, order=[int(i) for i in order], extra=extra)
    assert r1 == r2 == extra["returns"]
    for a, b, c in zip(shuffled, split, py):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert not all(np.array_equal(a, b) for a, b in zip(shuffled, ref))      # the order did matter


@pytest.mark.parametrize("seed,stereo,th", [(5, False, 1.0), (6, True, 3.0)])
def test_budgeted_projection_search_at_its_two_deterministic_ends(oracle, seed, stereo, th):
    rng, kps, desc, sf, mp, u_right, obs0 = _gf_case(oracle, seed, stereo)
    M = len(mp["isBad"])
    ref, never = _fresh(obs0), _fresh(obs0)
    nm_ref = oracle.search_by_projection(kps, desc, u_right, GRID, mp, th, 0.8, *ref)
    nm, stop = oracle.search_by_projection_budget(kps, desc, u_right, GRID, mp, th, 0.8, 1e12, *never)
    assert (nm, stop) == (nm_ref, M)
    for a, b in zip(ref, never):
        assert np.array_equal(a, b)
    # no time at all: the loop ends after the first map point that reaches the clock -- not one that is out of view,
    # bad, without candidates or rejected by the ratio test.  Make the first points such cases.
    mp["mbTrackInView"][0] = 0
    mp["isBad"][1] = 1
    mp["mTrackProjX"][2] = 5000.0
    zero, py = _fresh(obs0), _fresh(obs0)
    extra = {}
    nm0, stop0 = oracle.search_by_projection_budget(kps, desc, u_right, GRID, mp, th, 0.8, 0.0, *zero)
    nm_py = _py_search_by_projection(kps, desc, u_right, GRID, mp, th, 0.8, sf, *py, budget0=True, extra=extra)
    assert (nm0, stop0) == (nm_py, extra["stopped_at"]) and 3 <= stop0 < M and nm0 <= 1
    for a, b in zip(zero, py):
        assert np.array_equal(a, b)


# ---- ORBmatcher::SearchByBoW (src/ORBmatcher.cc:270-404 KeyFrame/Frame, :635-768 KeyFrame/KeyFrame) ----
def _bow_case(rng, n1=600, n2=650, nodes=60, flip=0.03):
    """two frames whose first min(n1, n2) keypoints correspond; a feature's vocabulary node is mostly the same on both
    sides (as for a real vocabulary: similar descriptors fall into the same node), some land elsewhere"""
    k1, d1, k2, d2 = _random_frames(rng, n1, n2, clusters=10)
    n = min(n1, n2)
    bits = np.unpackbits(d1[:n], axis=1)
    d2[:n] = np.packbits(bits ^ (rng.random(bits.shape) < flip), axis=1)
    k2["angle"][:n] = ((k1["angle"][:n] + rng.choice([0.0, 0.0, 0.0, 3.0, 100.0], n)) % 360).astype(np.float32)
    node1 = rng.integers(0, nodes, n1) * 7 + 3
    node2 = rng.integers(0, nodes, n2) * 7 + 3
    same = rng.random(n) < 0.85
    node2[:n][same] = node1[:n][same]
    node2[node2 == 3] = 2                       # a node only side 2 has, below all of side 1's ...
    node1[node1 == 10] = 11                     # ... and one only side 1 has: both lower_bound branches are taken
    fv1 = {int(v): np.nonzero(node1 == v)[0][rng.permutation((node1 == v).sum())].tolist() for v in np.unique(node1)}
    fv2 = {int(v): np.nonzero(node2 == v)[0][rng.permutation((node2 == v).sum())].tolist() for v in np.unique(node2)}
    h1 = (rng.random(n1) < 0.8).astype(np.uint8)
    h2 = (rng.random(n2) < 0.8).astype(np.uint8)
    return k1, d1, h1, fv1, k2, d2, h2, fv2


def _py_search_by_bow(k1, d1, h1, fv1, k2, d2, h2, fv2, ratio, ori, keyframes):
    out = np.full(len(k1) if keyframes else len(k2), -1, np.int32)
    taken = np.zeros(len(k2), bool)
    hist = [[] for _ in range(30)]
    nm = 0
    for node in sorted(set(fv1) & set(fv2)):
        for i1 in fv1[node]:
            if not h1[i1]:
                continue
            b1, b2, bi = 256, 256, -1
            for i2 in fv2[node]:
                if taken[i2] or (keyframes and not h2[i2]):
                    continue
                d = _ham(d1[i1], d2[i2])
                if d < b1:
                    b2, b1, bi = b1, d, i2
                elif d < b2:
                    b2 = d
            if (b1 < 50 if keyframes else b1 <= 50) and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                taken[bi] = True
                if keyframes:
                    out[i1] = bi
                else:
                    out[bi] = i1
                nm += 1
                if ori:
                    rot = np.float32(k1["angle"][i1]) - np.float32(k2["angle"][bi])
                    if rot < 0:
                        rot = np.float32(rot + np.float32(360))
                    b = int(np.floor(float(np.float32(rot * np.float32(1.0 / 30))) + 0.5))
                    hist[0 if b == 30 else b].append(i1 if keyframes else bi)
    if ori:
        sizes = [len(h) for h in hist]
        order = sorted(range(30), key=lambda i: (-sizes[i], i))[:3]
        keep = [order[0]] if sizes[order[0]] > 0 else []
        for o_ in order[1:]:
            if sizes[o_] > 0 and not (np.float32(sizes[o_]) < np.float32(0.1) * np.float32(sizes[order[0]])):
                keep.append(o_)
            else:
                break
        for b in range(30):
            if b not in keep:
                for j in hist[b]:
                    out[j] = -1
                    nm -= 1
    return nm, out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bow_search_against_python_statement(oracle, seed):
    rng = np.random.default_rng(400 + seed)
    case = _bow_case(rng)
    k1, d1, h1, fv1, k2, d2, h2, fv2 = case
    for ratio, ori in ((0.7, True), (0.9, False)):
        nm_o, m_o = oracle.search_by_bow_frame(k1, d1, h1, fv1, k2, d2, fv2, ratio, ori)
        nm_p, m_p = _py_search_by_bow(*case, ratio, ori, False)
        assert nm_o == nm_p and nm_o > 100 and np.array_equal(m_o, m_p)
        nm_o, m_o = oracle.search_by_bow_keyframes(*case, ratio, ori)
        nm_p, m_p = _py_search_by_bow(*case, ratio, ori, True)
        assert nm_o == nm_p and nm_o > 80 and np.array_equal(m_o, m_p)


def test_bow_search_rules(oracle):
    k1 = _kps([(10, 10), (20, 20), (30, 30)], angle=[0.0, 0.0, 0.0])
    k2 = _kps([(10, 10), (20, 20), (30, 30), (40, 40)], angle=[0.0, 0.0, 0.0, 0.0])
    d1 = _desc_with_bits([0, 0, 0])
    ones = np.ones(3, np.uint8)

    def frame(bits2, fv1, fv2, h1=ones, ratio=0.7):
        return oracle.search_by_bow_frame(k1, d1, h1, fv1, k2, _desc_with_bits(bits2), fv2, ratio, False)

    def kfs(bits2, fv1, fv2, h2, ratio=0.7):
        return oracle.search_by_bow_keyframes(k1, d1, ones, fv1, k2, _desc_with_bits(bits2), h2, fv2, ratio, False)

    one_node = ({5: [0]}, {5: [0, 1]})
    # best 20 against second 30: 20 < 0.7 * 30 accepts; against 28 it does not (20 >= 19.6)
    assert frame([20, 30, 0, 0], *one_node)[1].tolist() == [0, -1, -1, -1]
    assert frame([20, 28, 0, 0], *one_node)[0] == 0
    # TH_LOW: 50 passes against a frame, not between keyframes (strictly below there)
    assert frame([50, 200, 0, 0], *one_node)[0] == 1
    assert kfs([50, 200, 0, 0], *one_node, np.ones(4, np.uint8))[0] == 0
    assert kfs([49, 200, 0, 0], *one_node, np.ones(4, np.uint8))[1].tolist() == [0, -1, -1]
    # features in different nodes never meet
    assert frame([0, 200, 0, 0], {5: [0]}, {6: [0, 1]})[0] == 0
    # a frame keypoint is taken once: the second keyframe keypoint of the node finds it gone and takes the next best
    nm, m = frame([5, 40, 200, 200], {5: [0, 1]}, {5: [0, 1, 2]}, ratio=0.9)
    assert nm == 2 and m.tolist() == [0, 1, -1, -1]
    # a keyframe keypoint without a good map point does not search; a keypoint of the second keyframe without one is no candidate
    assert frame([5, 200, 0, 0], *one_node, h1=np.array([0, 1, 1], np.uint8))[0] == 0
    assert kfs([5, 30, 0, 0], *one_node, np.array([0, 1, 1, 1], np.uint8), ratio=0.9)[1].tolist() == [1, -1, -1]
    # the node's stored order decides ties (strict <): the first of two equal candidates
    assert frame([20, 20, 0, 0], {5: [0]}, {5: [1, 0]}, ratio=1.1)[1].tolist() == [-1, 0, -1, -1]
    # empty feature vectors
    assert frame([0, 0, 0, 0], {}, {5: [0]})[0] == 0 and frame([0, 0, 0, 0], {5: [0]}, {})[0] == 0


# ---- ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1595-1721 ----
def _keyframe_points(rng, kps, desc, M):
    last = _last_frame_points(rng, kps, desc, M)
    return {"visible": last["visible"], "u": last["u"], "v": last["v"], "level": last["octave"], "angle": last["angle"],
            "descriptor": last["descriptor"], "Observations": rng.choice([0, 1, 4], M).astype(np.int32)}


@pytest.mark.parametrize("seed,th,orb_dist", [(1, 10.0, 100), (2, 3.0, 64)])
def test_relocalisation_projection_search(oracle, seed, th, orb_dist):
    """Against the last-frame search, which differs in three places only (any held map point blocks a keypoint, no
    right-image gate, the caller's distance bound): with keypoints that hold either nothing or a point with
    observations, ORBdist = TH_HIGH and a monocular frame the two must agree; then the differences one by one."""
    rng = np.random.default_rng(500 + seed)
    kps, desc, _, _ = _random_frames(rng, 500, 1, clusters=10)
    kps["octave"] = rng.integers(0, 8, len(kps))
    kf = _keyframe_points(rng, kps, desc, 350)
    kf["Observations"][:] = 2
    last = {"visible": kf["visible"], "u": kf["u"], "v": kf["v"], "invzc": np.ones(350, np.float32), "octave": kf["level"],
            "angle": kf["angle"], "descriptor": kf["descriptor"], "Observations": kf["Observations"]}
    obs0 = rng.choice([-1, -1, -1, 3], len(kps)).astype(np.int32)
    for ori in (True, False):
        a = (obs0.copy(), np.full(len(kps), -1, np.int32))
        b = (obs0.copy(), np.full(len(kps), -1, np.int32))
        nm_a = oracle.search_by_projection_keyframe(kps, desc, GRID, kf, th, 100, ori, *a)
        nm_b = oracle.search_by_projection_last(kps, desc, None, GRID, last, th, 0.0, False, False, ori, *b)
        assert nm_a == nm_b and nm_a > 30
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # a keypoint holding a map point without observations: free for the last-frame search, taken here
    obs1 = np.where(obs0 == 3, 0, obs0).astype(np.int32)
    a = (obs1.copy(), np.full(len(kps), -1, np.int32))
    oracle.search_by_projection_keyframe(kps, desc, GRID, kf, th, 100, False, *a)
    assert (a[1][obs1 == 0] == -1).all()
    # the distance bound
    fo, fm = np.full(len(kps), -1, np.int32), np.full(len(kps), -1, np.int32)
    nm_small = oracle.search_by_projection_keyframe(kps, desc, GRID, kf, th, orb_dist // 4, False, fo, fm)
    m = fm >= 0
    d = np.unpackbits(desc[m] ^ kf["descriptor"][fm[m]], axis=1).sum(1)
    assert nm_small == m.sum() and (d <= orb_dist // 4).all() and 0 < nm_small


# ---- the searches of the mapping and loop-closing threads (src/ORBmatcher.cc:406-518, :937-1212, :1214-1438, :770-935) ----
def _projected_points(rng, kps, desc, M, flip=0.05):
    pick = rng.integers(0, len(kps), M)
    bits = np.unpackbits(desc[pick], axis=1)
    u = (kps["x"][pick] + rng.normal(0, 1.5, M)).astype(np.float32)
    return {"visible": (rng.random(M) < 0.9).astype(np.uint8), "u": u,
            "v": (kps["y"][pick] + rng.normal(0, 1.5, M)).astype(np.float32),
            "ur": (u - rng.uniform(2, 40, M)).astype(np.float32),
            "level": np.clip(kps["octave"][pick] + rng.integers(0, 2, M), 0, 7).astype(np.int32),
            "descriptor": np.packbits(bits ^ (rng.random(bits.shape) < flip), axis=1)}


def _py_window_best(kps, desc, u_right, grid, pts, i, th, sf, inv_s2, gate, blocked):
    lvl = int(pts["level"][i])
    radius = np.float32(np.float32(th) * sf[lvl])
    best, bi = 256, -1
    for idx in _py_features_in_area(kps, grid, pts["u"][i], pts["v"][i], radius, -1, -1):
        if blocked is not None and blocked[idx] >= 0:
            continue
        kl = int(kps["octave"][idx])
        if kl < lvl - 1 or kl > lvl:
            continue
        if gate:
            ex = np.float32(pts["u"][i] - kps["x"][idx])
            ey = np.float32(pts["v"][i] - kps["y"][idx])
            if u_right is not None and u_right[idx] >= 0:
                er = np.float32(pts["ur"][i] - u_right[idx])
                e2 = np.float32(np.float32(np.float32(ex * ex) + np.float32(ey * ey)) + np.float32(er * er))
                if float(np.float32(e2 * inv_s2[kl])) > 7.8:
                    continue
            else:
                e2 = np.float32(np.float32(ex * ex) + np.float32(ey * ey))
                if float(np.float32(e2 * inv_s2[kl])) > 5.99:
                    continue
        d = _ham(pts["descriptor"][i], desc[idx])
        if d < best:
            best, bi = d, idx
    return bi, best


def _window_case(oracle, seed, n=450, m=380):
    rng = np.random.default_rng(600 + seed)
    kps, desc, _, _ = _random_frames(rng, n, 1, clusters=10)
    kps["octave"] = rng.integers(0, 8, len(kps))
    t = oracle.tables()
    return rng, kps, desc, t["scale"], t["inv_sigma2"], _projected_points(rng, kps, desc, m)


@pytest.mark.parametrize("seed,th", [(1, 10), (2, 4)])
def test_sim3_projection_search_against_python_statement(oracle, seed, th):
    rng, kps, desc, sf, is2, pts = _window_case(oracle, seed)
    matched0 = rng.choice([-1, -1, -1, 1], len(kps)).astype(np.int32)
    a = (matched0.copy(), np.full(len(kps), -1, np.int32))
    nm = oracle.search_by_projection_sim3(kps, desc, GRID, pts, th, *a)
    b = (matched0.copy(), np.full(len(kps), -1, np.int32))
    nm_py = 0
    for i in range(len(pts["u"])):
        if not pts["visible"][i]:
            continue
        bi, best = _py_window_best(kps, desc, None, GRID, pts, i, th, sf, is2, False, b[0])
        if best <= 50:
            b[0][bi], b[1][bi] = 1, i
            nm_py += 1
    assert nm == nm_py and nm > 30
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert (a[1][matched0 >= 0] == -1).all()


@pytest.mark.parametrize("seed,stereo,gate,th", [(1, True, True, 3.0), (2, False, True, 3.0), (3, False, False, 4.0)])
def test_fuse_search_against_python_statement(oracle, seed, stereo, gate, th):
    rng, kps, desc, sf, is2, pts = _window_case(oracle, seed)
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.uniform(2, 40, len(kps)).astype(np.float32), -1).astype(np.float32)
    nf, bi, bd = oracle.fuse_search(kps, desc, u_right, GRID, pts, th, gate)
    nf_py = 0
    for i in range(len(pts["u"])):
        want = (-1, 256)
        if pts["visible"][i]:
            want = _py_window_best(kps, desc, u_right, GRID, pts, i, th, sf, is2, gate, None)
        assert bd[i] == want[1] and bi[i] == (want[0] if want[1] <= 50 else -1), i
        nf_py += want[1] <= 50
    assert nf == nf_py and nf > 30
    if gate:          # the gate did reject candidates somewhere: without it more points fuse or some find a closer keypoint
        nf2, bi2, bd2 = oracle.fuse_search(kps, desc, u_right, GRID, pts, th, False)
        assert (bd2 < bd).any()


@pytest.mark.parametrize("seed", [1, 2])
def test_sim3_search_against_python_statement(oracle, seed):
    rng = np.random.default_rng(700 + seed)
    k1, d1, k2, d2 = _random_frames(rng, 400, 420, clusters=8)
    k1["octave"] = rng.integers(0, 8, len(k1))
    k2["octave"] = rng.integers(0, 8, len(k2))
    n = 400
    k2["x"][:n], k2["y"][:n], k2["octave"][:n] = k1["x"] + 1, k1["y"] - 1, k1["octave"]
    bits = np.unpackbits(d1, axis=1)
    d2[:n] = np.packbits(bits ^ (rng.random(bits.shape) < 0.03), axis=1)
    t = oracle.tables()
    # map point i of keyframe 1 carries keypoint i's descriptor and projects onto its partner in keyframe 2, and back
    pts1 = {"visible": (rng.random(len(k1)) < 0.85).astype(np.uint8), "u": k2["x"][:n].copy(), "v": k2["y"][:n].copy(),
            "level": k1["octave"].astype(np.int32), "descriptor": d1}
    pts2 = {"visible": (rng.random(len(k2)) < 0.85).astype(np.uint8),
            "u": np.concatenate([k1["x"], k2["x"][n:]]).astype(np.float32), "v": np.concatenate([k1["y"], k2["y"][n:]]).astype(np.float32),
            "level": k2["octave"].astype(np.int32), "descriptor": d2}
    nf, m12 = oracle.search_by_sim3(k1, d1, k2, d2, GRID, pts1, pts2, 7.5)
    m1 = np.full(len(k1), -1)
    m2 = np.full(len(k2), -1)
    for i in range(len(k1)):
        if pts1["visible"][i]:
            bi, best = _py_window_best(k2, d2, None, GRID, pts1, i, 7.5, t["scale"], None, False, None)
            if best <= 100:
                m1[i] = bi
    for i in range(len(k2)):
        if pts2["visible"][i]:
            bi, best = _py_window_best(k1, d1, None, GRID, pts2, i, 7.5, t["scale"], None, False, None)
            if best <= 100:
                m2[i] = bi
    want = np.array([m1[i] if m1[i] >= 0 and m2[m1[i]] == i else -1 for i in range(len(k1))], np.int32)
    assert np.array_equal(m12, want) and nf == (want >= 0).sum() and nf > 100
    assert ((m1 >= 0) & (want < 0)).any()           # some one-sided choices were dropped


def _py_search_for_triangulation(k1, d1, h1, s1, fv1, k2, d2, h2, s2, fv2, F, ep, only_stereo, ori, sf, s2tab):
    out = np.full(len(k1), -1, np.int32)
    hist = [[] for _ in range(30)]
    nm = 0
    F = np.asarray(F, np.float32).reshape(3, 3)
    f32 = np.float32
    for node in sorted(set(fv1) & set(fv2)):
        for i1 in fv1[node]:
            if h1[i1] or (only_stereo and not s1[i1]):
                continue
            best, bi = 50, -1
            x1, y1 = f32(k1["x"][i1]), f32(k1["y"][i1])
            a = f32(f32(f32(x1 * F[0, 0]) + f32(y1 * F[1, 0])) + F[2, 0])
            b = f32(f32(f32(x1 * F[0, 1]) + f32(y1 * F[1, 1])) + F[2, 1])
            c = f32(f32(f32(x1 * F[0, 2]) + f32(y1 * F[1, 2])) + F[2, 2])
            for i2 in fv2[node]:
                if h2[i2] or (only_stereo and not s2[i2]):
                    continue
                d = _ham(d1[i1], d2[i2])
                if d > 50 or d > best:
                    continue
                x2, y2 = f32(k2["x"][i2]), f32(k2["y"][i2])
                if not s1[i1] and not s2[i2]:
                    dx, dy = f32(f32(ep[0]) - x2), f32(f32(ep[1]) - y2)
                    if f32(f32(dx * dx) + f32(dy * dy)) < f32(f32(100) * sf[int(k2["octave"][i2])]):
                        continue
                num = f32(f32(f32(a * x2) + f32(b * y2)) + c)
                den = f32(f32(a * a) + f32(b * b))
                if den == 0:
                    continue
                dsqr = f32(f32(num * num) / den)
                if float(dsqr) < 3.84 * float(s2tab[int(k2["octave"][i2])]):
                    best, bi = d, i2
            if bi >= 0:
                out[i1] = bi
                nm += 1
                if ori:
                    rot = f32(k1["angle"][i1]) - f32(k2["angle"][bi])
                    if rot < 0:
                        rot = f32(rot + f32(360))
                    bn = int(np.floor(float(f32(rot * f32(1.0 / 30))) + 0.5))
                    hist[0 if bn == 30 else bn].append(i1)
    if ori:
        sizes = [len(h) for h in hist]
        order = sorted(range(30), key=lambda i: (-sizes[i], i))[:3]
        keep = [order[0]] if sizes[order[0]] > 0 else []
        for o_ in order[1:]:
            if sizes[o_] > 0 and not (f32(sizes[o_]) < f32(0.1) * f32(sizes[order[0]])):
                keep.append(o_)
            else:
                break
        for bn in range(30):
            if bn not in keep:
                for j in hist[bn]:
                    out[j] = -1
                    nm -= 1
    return nm, out


def _triangulation_case(rng, n1=600, n2=650, nodes=60):
    k1, d1, h1, fv1, k2, d2, h2, fv2 = _bow_case(rng, n1, n2, nodes)
    n = min(n1, n2)
    # a pure sideways motion: F12 = [t]x with t = (1, 0, 0) up to scale -> epipolar lines are the rows y2 = y1
    k2["y"][:n] = k1["y"][:n] + rng.choice([0.0, 0.0, 0.5, 1.0, 6.0], n).astype(np.float32)
    k2["x"][:n] = k1["x"][:n] - rng.uniform(1, 30, n).astype(np.float32)
    k2["octave"] = rng.integers(0, 8, len(k2))
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)
    s1 = (rng.random(n1) < 0.5).astype(np.uint8)
    s2 = (rng.random(n2) < 0.5).astype(np.uint8)
    h1 = (rng.random(n1) < 0.3).astype(np.uint8)        # here: keypoints that already hold a map point
    h2 = (rng.random(n2) < 0.3).astype(np.uint8)
    ep = (float(k2["x"][5]) + 3.0, float(k2["y"][5]) + 2.0)       # an epipole inside the image, near some keypoints
    return k1, d1, h1, s1, fv1, k2, d2, h2, s2, fv2, F, ep


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_triangulation_search_against_python_statement(oracle, seed):
    rng = np.random.default_rng(800 + seed)
    case = _triangulation_case(rng)
    t = oracle.tables()
    for only_stereo, ori in ((False, True), (True, False), (False, False)):
        nm_o, m_o = oracle.search_for_triangulation(*case, only_stereo, ori)
        nm_p, m_p = _py_search_for_triangulation(*case, only_stereo, ori, t["scale"], t["sigma2"])
        assert nm_o == nm_p and np.array_equal(m_o, m_p) and nm_o > (20 if only_stereo else 60)
    # a keypoint of the second keyframe may serve several of the first (the reference never sets vbMatched2)
    nm, m = oracle.search_for_triangulation(*case, False, False)
    vals = m[m >= 0]
    assert len(vals) == nm
