#!/usr/bin/env python3
"""A resident batch whose pyramid block exceeds 4 GiB (160 frames of 3840 x 2160, 8000 features): eight distinct
frames repeated twenty times, so every frame's output must equal that of the frame eight places before it; any
32-bit offset along the batch would break that beyond the 4 GiB mark.  Usage: python tools/big_batch_check.py [frames]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from orb_slam_amd import ORBextractor  # noqa: E402
from orb_slam_amd.synth import synthetic_frame  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    W, H, NF, P = 3840, 2160, 8000, 8
    base = np.stack([synthetic_frame(W, H, 4000 + i) for i in range(P)])
    dev = torch.device("cuda:0")
    d_imgs = torch.from_numpy(base).to(dev).repeat((B + P - 1) // P, 1, 1)[:B].contiguous()
    ex = ORBextractor(NF, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(W, H)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    for _ in range(2):
        ex.extract_batch_device(d_imgs, B, W, H, W, W * H, d_kps, d_desc, cap, d_n)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    k_bits = d_kps.view(torch.int32)
    bad = 0
    for f in range(P, B):
        # keypoints as bit patterns: class_id = -1 read as a float is a NaN, which equals nothing
        same = n[f] == n[f - P] and torch.equal(k_bits[f, :n[f]], k_bits[f - P, :n[f]]) and torch.equal(d_desc[f, :n[f]], d_desc[f - P, :n[f]])
        if not same:
            bad += 1
            if bad <= 4 and n[f] == n[f - P]:
                a, b = k_bits[f, :n[f]].cpu().numpy(), k_bits[f - P, :n[f]].cpu().numpy()
                da, db = d_desc[f, :n[f]].cpu().numpy(), d_desc[f - P, :n[f]].cpu().numpy()
                rows = np.nonzero((a != b).any(axis=1))[0]
                print("frame %d vs %d: %d keypoint rows differ (first %s), fields %s, %d descriptor rows differ" % (
                    f, f - P, len(rows), rows[:5], (a != b).any(axis=0), int((da != db).any(axis=1).sum())))
                if len(rows):
                    print("   ", a[rows[0]], b[rows[0]])
            elif bad <= 12:
                print("frame %d differs from frame %d (%d / %d keypoints)" % (f, f - P, n[f], n[f - P]))
    print("big batch: %d frames of %dx%d, keypoints per frame %d..%d, %d frames differ from their twin" % (B, W, H, n.min(), n.max(), bad))
    ex.close()
    return 1 if bad or n.min() < NF // 2 else 0


if __name__ == "__main__":
    sys.exit(main())
