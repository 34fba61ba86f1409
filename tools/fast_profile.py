#!/usr/bin/env python3
"""Where a FAST wavefront spends its cycles, per section, summed over all cells of a 256-frame batch.
Needs the diagnostic build: make -C orb_slam_amd/csrc OUT=$PWD/build/liborbx_prof.so EXTRA=-DORBX_FAST_PROFILE
Usage: ORBX_LIBRARY=build/liborbx_prof.so python tools/fast_profile.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orb_slam_amd import ORBextractor, load_library  # noqa: E402
from orb_slam_amd.synth import synthetic_frame  # noqa: E402

NAMES = ["tile load", "pretest", "strong pass (compact, score, emit)", "full score", "maxima + emit"]


def run(name, host):
    lib = load_library()
    B, H, W = host.shape
    dev = torch.device("cuda:0")
    d_imgs = torch.from_numpy(host).to(dev)
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
    torch.cuda.synchronize()
    out = (C.c_uint64 * 16)()
    lib.orbx_debug_fast_profile(out, 1)
    ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
    torch.cuda.synchronize()
    lib.orbx_debug_fast_profile(out, 1)
    v = [int(x) for x in out]
    cells = max(1, v[6])
    tot = sum(v[:5])
    print("%s: %d cells, %.0f cycles per cell; early exit in %.1f %% of cells; survivors per cell %.1f, strong %.1f"
          % (name, cells, tot / cells, 100.0 * v[7] / cells, v[8] / cells, v[9] / cells))
    for k in range(5):
        print("   %-38s %6.1f %%  %8.0f cycles per cell" % (NAMES[k], 100.0 * v[k] / tot, v[k] / cells))
    ex.set_stream(None)
    ex.close()


def main():
    syn = np.stack([synthetic_frame(752, 480, 1234 + i) for i in range(64)])
    run("synthetic", np.concatenate([syn] * 4))
    l_ = np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_l.npy"))
    r_ = np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_r.npy"))
    run("natural", np.stack([l_, r_] * 128))


if __name__ == "__main__":
    main()
