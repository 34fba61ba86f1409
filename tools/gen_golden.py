#!/usr/bin/env python3
"""Generate tests/golden/euroc_golden.npz from the CPU oracle (literal
formulations) on the reference's own test images (test/EuRoC_l.png, test/EuRoC_r.png,
copied to tests/fixtures/).  The reference holds no golden vectors for this
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


def main():
    left = np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_l.npy"))
    right = np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_r.npy"))
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
    maxD = float(np.float32(STEREO_MBF) / np.float32(STEREO_MB))
    bi, bd = o.stereo_search(kl, dl, kr, dr, sf, 480, 0.0, maxD, 0.0)
    nm, ur, dp = o.stereo_finish(kl, kr, bi, bd, STEREO_MBF, 0.0, maxD)
    bf_i, bf_d, bf_s = o.match_bruteforce(dl, dr)
    out = os.path.join(ROOT, "tests", "golden", "euroc_golden.npz")
    np.savez_compressed(out, kps_l=kl, desc_l=dl, counts_l=counts_l, kps_r=kr, desc_r=dr, counts_r=counts_r,
                        pyramid_sha256_l=np.array(sha), ncand_l=np.array(ncand, np.int32),
                        stereo_best_idx=bi, stereo_best_dist=bd, stereo_u_right=ur, stereo_depth=dp,
                        stereo_nmatched=np.int32(nm), bf_best_idx=bf_i, bf_best_dist=bf_d, bf_second_dist=bf_s,
                        stereo_params=np.array([STEREO_MBF, STEREO_MB], np.float64))
    print("wrote", out, os.path.getsize(out), "bytes;", len(kl), len(kr), "keypoints;", nm, "stereo matched;",
          int((ur >= 0).sum()), "with depth")


if __name__ == "__main__":
    main()
