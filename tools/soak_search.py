#!/usr/bin/env python3
"""Randomised parity run of the whole-routine searches (every ORBmatcher search the library reproduces) against the CPU
restatement: random frame sizes, window sizes, ratios, stereo / mono, orientation check on / off.
Usage: python tools/soak_search.py [cases] [seed0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ORBX_SWITCHES_PER_CALL", "1")
from oracle_lib import Oracle  # noqa: E402
from orb_slam_amd import ORBextractor, ORBmatcher  # noqa: E402
from test_oracle_search import (GRID, _bow_case, _keyframe_points, _last_frame_points, _map_points, _projected_points,  # noqa: E402
                                _random_frames, _triangulation_case)


def same(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return np.array_equal(a, b)
    return a == b


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    oracle = Oracle(1000, 1.2, 8, 20, 7)
    sf = oracle.tables()["scale"]
    bad = 0
    checked = 0
    for case in range(cases):
        rng = np.random.default_rng(seed0 + case)
        n = int(rng.integers(50, 2500))
        M = int(rng.integers(1, 2500))
        ratio = float(rng.choice([0.6, 0.7, 0.75, 0.8, 0.9, 1.0]))
        ori = bool(rng.integers(0, 2))
        stereo = bool(rng.integers(0, 2))
        clusters = int(rng.integers(2, 40))
        m = ORBmatcher(ex, nnratio=ratio, checkOri=ori)
        kps, desc, k2, d2 = _random_frames(rng, n, n + int(rng.integers(0, 200)), clusters=clusters)
        kps["octave"] = rng.integers(0, 8, n)
        k2["octave"] = rng.integers(0, 8, len(k2))
        u_right = None
        if stereo:
            u_right = np.where(rng.random(n) < 0.6, kps["x"] - rng.uniform(0, 40, n).astype(np.float32), -1).astype(np.float32)
        obs0 = rng.choice([-1, -1, -1, 0, 2], n).astype(np.int32)
        fresh = lambda k: tuple(obs0.copy() if i == 0 else np.full(n, -1, np.int32) for i in range(k))
        results = {}

        def check(name, got, want):
            nonlocal bad, checked
            checked += 1
            if not same(got, want):
                bad += 1
                print("MISMATCH case %d (seed %d) %s: n=%d M=%d ratio=%.2f ori=%d stereo=%d" % (case, seed0 + case, name, n, M, ratio, ori, stereo))

        mp = _map_points(rng, kps, desc, M, sf)
        th = float(rng.choice([1.0, 3.0, 5.0]))
        a, b = fresh(3), fresh(3)
        check("SearchByProjection", (m.SearchByProjection(kps, desc, u_right, GRID, mp, th, *a), a),
              (oracle.search_by_projection(kps, desc, u_right, GRID, mp, th, ratio, *b), b))
        a, b = fresh(3), fresh(3)
        check("SearchByProjection_Budget(0)", (m.SearchByProjection_Budget(kps, desc, u_right, GRID, mp, th, 0.0, *a), a),
              (oracle.search_by_projection_budget(kps, desc, u_right, GRID, mp, th, ratio, 0.0, *b), b))
        cands = m.GetCandidates(kps, desc, GRID, mp, th)
        a, b = fresh(3), fresh(3)
        order = rng.permutation(M)[: min(M, 400)]
        check("MatchCandidates", ([m.MatchCandidates(kps, u_right, cands, mp, int(i), th, *a) for i in order], a),
              ([oracle.search_by_projection_one_point(kps, desc, u_right, GRID, mp, int(i), th, ratio, *b) for i in order], b))
        last = _last_frame_points(rng, kps, desc, M)
        fwd, bwd = [(False, False), (True, False), (False, True)][int(rng.integers(0, 3))]
        thl = float(rng.choice([7.0, 14.0, 15.0, 30.0]))
        a, b = fresh(2), fresh(2)
        check("SearchByProjection(last)", (m.SearchByProjectionLast(kps, desc, u_right, GRID, last, thl, 40.0, fwd, bwd, *a), a),
              (oracle.search_by_projection_last(kps, desc, u_right, GRID, last, thl, 40.0, fwd, bwd, ori, *b), b))
        kf = _keyframe_points(rng, kps, desc, M)
        od = int(rng.choice([30, 64, 100]))
        a, b = fresh(2), fresh(2)
        check("SearchByProjection(keyframe)", (m.SearchByProjectionKeyFrame(kps, desc, GRID, kf, thl, od, *a), a),
              (oracle.search_by_projection_keyframe(kps, desc, GRID, kf, thl, od, ori, *b), b))
        k0 = kps.copy()
        k0["octave"] = np.where(rng.random(n) < 0.7, 0, k0["octave"])
        prev = np.stack([k0["x"], k0["y"]], 1).astype(np.float32) + rng.normal(0, 3, (n, 2)).astype(np.float32)
        win = int(rng.choice([10, 30, 100]))
        pa, pb = prev.copy(), prev.copy()
        check("SearchForInitialization", (m.SearchForInitialization(k0, desc, k2, d2, GRID, pa, win), pa),
              (oracle.search_for_initialization(k0, desc, k2, d2, GRID, pb, win, ratio, ori), pb))
        bow = _bow_case(rng, n, len(k2), int(rng.integers(5, 150)))
        check("SearchByBoW(frame)", m.SearchByBoW(bow[0], bow[1], bow[2], bow[3], bow[4], bow[5], bow[7]),
              oracle.search_by_bow_frame(bow[0], bow[1], bow[2], bow[3], bow[4], bow[5], bow[7], ratio, ori))
        check("SearchByBoW(keyframes)", m.SearchByBoW_KeyFrames(*bow), oracle.search_by_bow_keyframes(*bow, ratio, ori))
        tri = _triangulation_case(rng, n, len(k2), int(rng.integers(5, 150)))
        only = bool(rng.integers(0, 2))
        check("SearchForTriangulation", m.SearchForTriangulation(*tri, only), oracle.search_for_triangulation(*tri, only, ori))
        pts = _projected_points(rng, kps, desc, M)
        ths = int(rng.choice([3, 4, 10]))
        matched0 = rng.choice([-1, -1, 1], n).astype(np.int32)
        a = (matched0.copy(), np.full(n, -1, np.int32))
        b = (matched0.copy(), np.full(n, -1, np.int32))
        check("SearchByProjection(Sim3)", (m.SearchByProjectionSim3(kps, desc, GRID, pts, ths, *a), a),
              (oracle.search_by_projection_sim3(kps, desc, GRID, pts, ths, *b), b))
        gate = bool(rng.integers(0, 2))
        check("Fuse", m.FuseSearch(kps, desc, u_right, GRID, pts, float(ths), gate),
              oracle.fuse_search(kps, desc, u_right, GRID, pts, float(ths), gate))
        p2 = _projected_points(rng, k2, d2, len(k2))
        p1 = _projected_points(rng, k2, d2, n)          # points of keyframe 1, projected onto keypoints of keyframe 2
        p2b = _projected_points(rng, kps, desc, len(k2))
        check("SearchBySim3", m.SearchBySim3(kps, desc, k2, d2, GRID, p1, p2b, 7.5),
              oracle.search_by_sim3(kps, desc, k2, d2, GRID, p1, p2b, 7.5))
        del p2
        if (case + 1) % 10 == 0:
            print("case %d: %d comparisons, %d mismatches" % (case + 1, checked, bad), flush=True)
    print("soak_search: %d cases, %d comparisons, %d mismatches" % (cases, checked, bad))
    ex.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
