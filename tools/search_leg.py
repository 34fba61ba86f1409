#!/usr/bin/env python3
"""Latency of the whole-routine search calls a tracker makes once per frame, on the features of two real extractions
(a synthetic frame and the same scene moved by a few pixels): SearchForInitialization, SearchByProjection against the
last frame, SearchByProjection against local map points.  Host arrays in, host arrays out.
measure(ex) -> dict for bench.py; run as a program it prints the dict."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _clock(fn, reps):
    for _ in range(5):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def measure(ex, frame, reps=100):
    from orb_slam_amd import ORBmatcher
    H, W = frame.shape
    a = frame
    b = np.roll(a, (2, -5), axis=(0, 1))
    k1, d1 = ex(a)
    k2, d2 = ex(b)
    n1, n2 = len(k1), len(k2)
    grid = (0.0, 0.0, float(np.float32(64) / np.float32(W)), float(np.float32(48) / np.float32(H)))
    m = ORBmatcher(ex, nnratio=0.9, checkOri=True)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    out = {"keypoints": [int(n1), int(n2)]}
    res = {}

    def init():
        res["init"] = m.SearchForInitialization(k1, d1, k2, d2, grid, prev.copy(), 100)[0]
    out["SearchForInitialization_ms"] = round(_clock(init, reps), 4)
    # the first frame's keypoints as the last frame's / the local map's points, projected where the scene moved to
    last = {"visible": np.ones(n1, np.uint8), "u": (k1["x"] - 5).astype(np.float32), "v": (k1["y"] + 2).astype(np.float32),
            "invzc": np.ones(n1, np.float32), "octave": k1["octave"].astype(np.int32), "angle": k1["angle"].astype(np.float32),
            "Observations": np.ones(n1, np.int32), "descriptor": d1}

    def lastf():
        fo, fm = np.full(n2, -1, np.int32), np.full(n2, -1, np.int32)
        res["last"] = m.SearchByProjectionLast(k2, d2, None, grid, last, 15.0, 0.0, False, False, fo, fm)
    out["SearchByProjection_last_frame_ms"] = round(_clock(lastf, reps), 4)
    mp = {"mbTrackInView": np.ones(n1, np.uint8), "isBad": np.zeros(n1, np.uint8), "mTrackProjX": last["u"], "mTrackProjY": last["v"],
          "mTrackProjXR": np.zeros(n1, np.float32), "mnTrackScaleLevel": k1["octave"].astype(np.int32),
          "mTrackViewCos": np.full(n1, 0.9995, np.float32), "Observations": np.ones(n1, np.int32), "descriptor": d1}
    m2 = ORBmatcher(ex, nnratio=0.8)

    def local():
        fo, fm, ms = (np.full(n2, -1, np.int32) for _ in range(3))
        res["local"] = m2.SearchByProjection(k2, d2, None, grid, mp, 3.0, fo, fm, ms)
    out["SearchByProjection_local_map_ms"] = round(_clock(local, reps), 4)
    out["matches"] = {"SearchForInitialization": int(res["init"]), "last_frame": int(res["last"]), "local_map": int(res["local"])}
    out["what"] = ("one call per frame through the Python binding, host arrays in and out: the features of a synthetic frame and "
                   "of the same scene moved by (-5, 2) pixels; window 100 / th 15 / th 3")
    return out


if __name__ == "__main__":
    from orb_slam_amd import ORBextractor
    from orb_slam_amd.synth import synthetic_frame
    ex_ = ORBextractor(1000, 1.2, 8, 20, 7)
    print(measure(ex_, synthetic_frame(752, 480, 1234)))
    ex_.close()
