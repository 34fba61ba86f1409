#!/usr/bin/env python3
"""What overlapping the stages of two half-batches is worth: one context with 256 resident frames on one
stream against two contexts with 128 frames each on two streams (and four with 64).  Usage: python tools/two_stream.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orb_slam_amd import ORBextractor  # noqa: E402
from orb_slam_amd.synth import synthetic_frame  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    W, H, B = 752, 480, 256
    host = np.stack([synthetic_frame(W, H, 1234 + i) for i in range(B)])
    d_imgs = torch.from_numpy(host).to(dev)
    for parts in (1, 2, 4):
        n = B // parts
        exs, streams, outs = [], [], []
        for p in range(parts):
            ex = ORBextractor(1000, 1.2, 8, 20, 7)
            st = torch.cuda.Stream(device=dev)
            ex.set_stream(st.cuda_stream)
            cap = ex.max_keypoints(W, H)
            outs.append((torch.zeros((n, cap, 7), dtype=torch.float32, device=dev),
                         torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev),
                         torch.zeros((n,), dtype=torch.int32, device=dev)))
            exs.append(ex)
            streams.append(st)

        def step():
            for p in range(parts):
                k, d, c = outs[p]
                exs[p].extract_batch_device(d_imgs[p * n:], n, W, H, W, W * H, k, d, cap, c)

        t_end = time.perf_counter() + 0.4
        while time.perf_counter() < t_end:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
        reps = 100
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%d stream(s) x %3d frames: %.4f ms per 256 frames, %.0f frames/s" % (parts, n, dt * 1e3, B / dt))
        for ex in exs:
            ex.set_stream(None)
            ex.close()


if __name__ == "__main__":
    main()
