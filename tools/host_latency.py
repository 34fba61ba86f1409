#!/usr/bin/env python3
"""End-to-end latency of the host-buffer entry point (orbx_extract: H2D copy, kernels, D2H copy,
synchronise) for one frame, as the reference's operator() is called.  Usage: host_latency.py [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam_amd import ORBextractor  # noqa: E402
from orb_slam_amd.synth import synthetic_frame  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    imgs = [synthetic_frame(752, 480, 100 + i) for i in range(8)]
    for i in range(20):
        ex(imgs[i % 8])
    t = []
    for i in range(n):
        t0 = time.perf_counter()
        k, d = ex(imgs[i % 8])
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    print("orbx_extract, 752x480, one frame: median %.0f us, p90 %.0f us, min %.0f us (%d keypoints)" %
          (np.median(t), np.percentile(t, 90), t.min(), len(k)))
    ex.set_stage_timing(True)          # a second pass with the per-stage events in the launch sequence
    ex.reset_stage_times()
    for i in range(n):
        ex(imgs[i % 8])
    st = ex.stage_times()
    print("  GPU stages (us): " + "  ".join("%s %.0f" % (s, 1e3 * v[0] / max(1, v[1])) for s, v in st.items()))


if __name__ == "__main__":
    main()
