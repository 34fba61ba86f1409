#!/usr/bin/env python3
"""Rate of orbx_extract_batch fed from host memory (upload, kernels, download, synchronise), as
opposed to the resident rate bench.py reports.  Usage: host_batch_rate.py [frames] [reps] [pinned]   (pinned: the caller's buffers are page-locked)"""
import ctypes as C
import sys
import time

import numpy as np

from orb_slam_amd import ORBextractor
from orb_slam_amd.extractor import KP_DTYPE
from orb_slam_amd.synth import synthetic_batch


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    pinned = len(sys.argv) > 3 and sys.argv[3] == "pinned"
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    imgs = np.ascontiguousarray(synthetic_batch(min(b, 32), 752, 480, seed0=1234))
    imgs = np.ascontiguousarray(np.concatenate([imgs] * ((b + len(imgs) - 1) // len(imgs)))[:b])
    h, w = imgs.shape[1:]
    cap = ex.max_keypoints(w, h)
    kps = np.zeros((b, cap), KP_DTYPE)
    desc = np.zeros((b, cap, 32), np.uint8)
    n = np.zeros(b, np.int32)
    if pinned:
        import torch
        keep = []

        def pin(a):
            t = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
            keep.append(t)
            v = t.numpy().view(a.dtype).reshape(a.shape)
            v[...] = a
            return v
        imgs, kps, desc = pin(imgs), pin(kps), pin(desc)
    ptrs = (C.c_void_p * b)(*[imgs[i].ctypes.data for i in range(b)])

    def call():
        rc = ex._lib.orbx_extract_batch(ex._ctx, ptrs, b, w, h, w, kps.ctypes.data, desc.ctypes.data, cap, n.ctypes.data)
        assert rc == 0, rc

    for _ in range(3):
        call()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    print("orbx_extract_batch from %s host memory, %d x %dx%d: %.3f ms per batch, %.0f frames/s (mean %.0f keypoints)"
          % ("pinned" if pinned else "pageable", b, w, h, dt * 1e3, b / dt, n.mean()))


if __name__ == "__main__":
    main()
