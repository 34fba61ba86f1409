#!/usr/bin/env python3
"""Randomised parity soak: many frames of random sizes, contents and extractor settings through
liborbx and through the CPU oracle; every keypoint field and descriptor byte must agree.
Usage: python tools/soak_parity.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402
from orb_slam_amd import ORBextractor, OrbxError  # noqa: E402
from orb_slam_amd.synth import synthetic_frame  # noqa: E402


def make_image(rng, w, h):
    kind = rng.integers(0, 4)
    if kind == 0:
        return synthetic_frame(w, h, int(rng.integers(1 << 30)))
    if kind == 1:                       # noise at a random contrast
        amp = int(rng.integers(4, 128))
        return (128 + rng.integers(-amp, amp + 1, size=(h, w))).astype(np.uint8)
    if kind == 2:                       # blocky texture with saturated regions
        s = int(rng.integers(3, 17))
        small = rng.integers(0, 256, size=(h // s + 2, w // s + 2)).astype(np.uint8)
        img = np.kron(small, np.ones((s, s), np.uint8))[:h, :w]
        return np.clip(img.astype(np.int32) * 2 - 100, 0, 255).astype(np.uint8)
    base = synthetic_frame(w, h, int(rng.integers(1 << 30))).astype(np.int32)
    return np.clip(base + rng.integers(-12, 13, size=(h, w)), 0, 255).astype(np.uint8)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    frames = 0
    for case in range(n_cases):
        w, h = int(rng.integers(120, 1400)), int(rng.integers(100, 900))
        if rng.random() < 0.6:
            w &= ~3                      # rows that are dword-addressable: batches read level 0 from the image itself
        cfg = (int(rng.integers(50, 2500)), float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, 2.0])),
               int(rng.integers(1, 9)), int(rng.integers(8, 40)), int(rng.integers(2, 8)))
        try:
            ex = ORBextractor(*cfg)
            ex.max_keypoints(w, h)
        except OrbxError as e:          # geometries the library documents as unsupported
            print("case %d %dx%d %s: refused (%s)" % (case, w, h, cfg, e))
            continue
        orc = Oracle(*cfg)
        for rep in range(3):
            img = make_image(rng, w, h)
            try:
                kg, dg = ex(img)
            except OrbxError as e:
                print("case %d %dx%d %s: refused (%s)" % (case, w, h, cfg, e))
                break
            ko, do = orc.extract(img, literal=False)
            frames += 1
            ok = len(kg) == len(ko)
            if ok and len(ko):
                ok = all(np.array_equal(kg[f], ko[f]) for f in ko.dtype.names) and np.array_equal(dg, do)
            if not ok:
                bad += 1
                print("MISMATCH case %d rep %d %dx%d %s: %d vs %d keypoints" % (case, rep, w, h, cfg, len(kg), len(ko)))
                np.save(os.path.join(ROOT, "gpurun_out", "soak_mismatch_%d_%d.npy" % (case, rep)), img)
        else:
            # the same geometry as a batch of five (uploaded in one piece, level 0 read in place where the rows allow it)
            imgs = np.stack([make_image(rng, w, h) for _ in range(5)])
            res = ex.extract_batch(imgs)
            for i in range(5):
                ko, do = orc.extract(imgs[i], literal=False)
                kg, dg = res[i]
                frames += 1
                ok = len(kg) == len(ko) and (len(ko) == 0 or (all(np.array_equal(kg[f], ko[f]) for f in ko.dtype.names) and np.array_equal(dg, do)))
                if not ok:
                    bad += 1
                    print("MISMATCH case %d batch frame %d %dx%d %s: %d vs %d keypoints" % (case, i, w, h, cfg, len(kg), len(ko)))
                    np.save(os.path.join(ROOT, "gpurun_out", "soak_mismatch_%d_b%d.npy" % (case, i)), imgs[i])
        ex.close()
    print("soak: %d frames, %d mismatches" % (frames, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
