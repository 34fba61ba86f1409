#!/usr/bin/env python3
"""Generate tests/golden/{euroc,tum}_golden.npz from the CPU oracle (literal
formulations) on the reference's own test images: test/EuRoC_{l,r}.png (752x480, 8 bit,
copied to tests/fixtures/) and test/TUM_{l,r}.png, the pair test/test_Stereo.cpp:94-95
loads (512x512, 16 bit; tests/fixtures/TUM_{l,r}.npy hold the high byte of every
pixel, what imread(..., CV_LOAD_IMAGE_GRAYSCALE) at :156,:163 yields for a 16-bit PNG;
made with `(numpy.array(PIL.Image.open(png)) >> 8).astype(uint8)`).  A square frame
has one quadtree root (nIni = 1).  The reference holds no golden vectors for this
path, so these pin the oracle against its own regressions and give the GPU
tests a fixed target; they are not outputs of the reference binary."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle  # noqa: E402

STEREO_MBF, STEREO_MB = 47.90639384423901, 0.110074   # EuRoC-like bf / baseline


def make(name, fixture, mbf, mb):
    left = np.load(os.path.join(ROOT, "tests", "fixtures", fixture + "_l.npy"))
    right = np.load(os.path.join(ROOT, "tests", "fixtures", fixture + "_r.npy"))
    rows = left.shape[0]
    o = Oracle(1000, 1.2, 8, 20, 7)
    kl, dl = o.extract(left, literal=True)
    counts_l = o.last_level_counts()
    sha = []
    for l in range(8):
        sha.append(hashlib.sha256(o.pyramid_plane(l).tobytes()).hexdigest())
    ncand = [len(o.detect_level(l, True)) for l in range(8)]
    kr, dr = o.extract(right, literal=True)
    counts_r = o.last_level_counts()
    sf = o.tables()["scale"]
    maxD = float(np.float32(mbf) / np.float32(mb))
    bi, bd = o.stereo_search(kl, dl, kr, dr, sf, rows, 0.0, maxD, 0.0)
    nm, ur, dp = o.stereo_finish(kl, kr, bi, bd, mbf, 0.0, maxD)
    bf_i, bf_d, bf_s = o.match_bruteforce(dl, dr)
    out = os.path.join(ROOT, "tests", "golden", name + "_golden.npz")
    np.savez_compressed(out, kps_l=kl, desc_l=dl, counts_l=counts_l, kps_r=kr, desc_r=dr, counts_r=counts_r,
                        pyramid_sha256_l=np.array(sha), ncand_l=np.array(ncand, np.int32),
                        stereo_best_idx=bi, stereo_best_dist=bd, stereo_u_right=ur, stereo_depth=dp,
                        stereo_nmatched=np.int32(nm), bf_best_idx=bf_i, bf_best_dist=bf_d, bf_second_dist=bf_s,
                        stereo_params=np.array([mbf, mb], np.float64))
    print("wrote", out, os.path.getsize(out), "bytes;", len(kl), len(kr), "keypoints;", nm, "stereo matched;",
          int((ur >= 0).sum()), "with depth")


def main():
    make("euroc", "EuRoC", STEREO_MBF, STEREO_MB)
    make("tum", "TUM", 19.06, 0.1009)      # TUM-VI-like bf / baseline (190 px focal length at half resolution)


if __name__ == "__main__":
    main()
