#!/usr/bin/env python3
"""Wall-clock time of the whole-routine search calls on one frame (host arrays in, host arrays out): what a tracker
that calls them once per frame pays.  Usage: python tools/search_latency.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orb_slam_amd import ORBextractor, ORBmatcher  # noqa: E402
from test_oracle_search import GRID, _bow_case, _last_frame_points, _map_points, _random_frames  # noqa: E402


def clock(fn, reps=200):
    for _ in range(10):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    rng = np.random.default_rng(1)
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    m = ORBmatcher(ex, nnratio=0.8)
    sf = ex.GetScaleFactors()
    for n in (1000, 2000):
        kps, desc, k2, d2 = _random_frames(rng, n, n, clusters=10)
        kps["octave"] = rng.integers(0, 8, n)
        mp = _map_points(rng, kps, desc, n, np.asarray(sf, np.float32))
        last = _last_frame_points(rng, kps, desc, n)
        obs0 = np.full(n, -1, np.int32)

        def proj():
            fo, fm, ms = obs0.copy(), obs0.copy(), obs0.copy()
            m.SearchByProjection(kps, desc, None, GRID, mp, 3.0, fo, fm, ms)

        def lastf():
            fo, fm = obs0.copy(), obs0.copy()
            m.SearchByProjectionLast(kps, desc, None, GRID, last, 15.0, 0.0, False, False, fo, fm)

        prev = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
        k0 = kps.copy()
        k0["octave"] = 0

        def init():
            m.SearchForInitialization(k0, desc, k2, d2, GRID, prev.copy(), 100)

        bow = _bow_case(rng, n, n, 100)

        def bowf():
            m.SearchByBoW(bow[0], bow[1], bow[2], bow[3], bow[4], bow[5], bow[7])

        print("%d keypoints, %d map points: SearchByProjection %.0f us, SearchByProjection(last frame) %.0f us, "
              "SearchForInitialization %.0f us, SearchByBoW %.0f us" % (n, n, clock(proj), clock(lastf), clock(init), clock(bowf)))
    ex.close()


if __name__ == "__main__":
    main()
