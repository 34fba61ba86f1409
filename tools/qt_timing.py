#!/usr/bin/env python3
"""Where the quadtree launch spends its time, per (frame, level) workgroup: needs a diagnostic build of the library
(make -C orb_slam_amd/csrc OUT=../lib/liborbx_qt.so EXTRA=-DORBX_QT_TIMING) named by ORBX_LIBRARY.
Usage: ORBX_LIBRARY=... python tools/qt_timing.py [frames] [w h nfeat]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orb_slam_amd import ORBextractor  # noqa: E402
from orb_slam_amd._lib import library_path  # noqa: E402
from orb_slam_amd.synth import synthetic_frame  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    W, H, NF = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160, 8000)
    host = np.stack([synthetic_frame(W, H, 1234 + i) for i in range(B)])
    dev = torch.device("cuda:0")
    d_imgs = torch.from_numpy(host).to(dev)
    ex = ORBextractor(NF, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    for _ in range(3):
        ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
    torch.cuda.synchronize()
    lib = C.CDLL(library_path())
    out = np.zeros(64 * 8 * 8, np.uint64)
    rc = lib.orbx_debug_qt_timing(C.c_void_p(out.ctypes.data))
    assert rc == 0, rc
    t = out.reshape(64, 8, 8)[:B].astype(np.int64)
    t0 = t[:, :, 0].min()
    print("frames %d  %dx%d  nfeat %d   (microseconds; start = after the launch's first workgroup began)" % (B, W, H, NF))
    print("frame level  form   keys  steps   start  ->m1    ->m2    ->m3    ->end   total")
    for f in range(min(B, 1)):
        for lv in range(8):
            m = t[f, lv]
            us = lambda a, b: (m[b] - m[a]) / 100.0
            print("%5d %5d  %-5s %6d  %5d  %6.1f %6.1f  %6.1f  %6.1f  %6.1f  %6.1f" % (
                f, lv, "keys" if m[5] else "count", m[6], m[7], (m[0] - t0) / 100.0, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(0, 4)))
    print("launch span: %.1f us" % ((t[:, :, 4].max() - t0) / 100.0))
    bins(lib, B)
    st = np.zeros(64 * 8 * 8, np.uint64)
    if hasattr(lib, "orbx_debug_qt_steps") and lib.orbx_debug_qt_steps(C.c_void_p(st.ctypes.data)) == 0:
        st = st.reshape(64, 8, 8)[:B].astype(np.int64)
        print("stretches of the steps, summed over the steps of the level (us): flags+scan | order+keys | sweep rank.. | positions scan | build")
        for lv in range(8):
            print("  level %d: " % lv + "  ".join("%6.1f" % (st[0, lv, k] / 100.0) for k in range(5)) +
                  "   shader clock %.0f MHz" % (st[0, lv, 6] / max(1, st[0, lv, 7]) * 100.0))


def bins(lib, B):
    if not hasattr(lib, "orbx_debug_qt_bins"):
        return
    t = np.zeros(2048 * 8, np.uint64)
    if lib.orbx_debug_qt_bins(C.c_void_p(t.ctypes.data)) != 0:
        return
    t = t.reshape(2048, 8).astype(np.int64)
    t = t[t[:, 0] > 0]
    if not len(t):
        return
    t0 = t[:, 0].min()
    print("k_qt_bins, blocks of the first listed level (%d workgroups): start after the first | zero | keys | sums | stores (us)" % len(t))
    for row in t[np.argsort(t[:, 0])][:: max(1, len(t) // 12)]:
        print("   %6.1f | %5.1f %5.1f %5.1f %5.1f" % ((row[0] - t0) / 100.0, (row[1] - row[0]) / 100.0, (row[2] - row[1]) / 100.0,
                                                  (row[3] - row[2]) / 100.0, (row[4] - row[3]) / 100.0))
    print("   span %.1f us" % ((t[:, 4].max() - t0) / 100.0))


if __name__ == "__main__":
    main()
