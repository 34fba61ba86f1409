"""GPU path against the committed golden fixtures (tests/golden/{euroc,tum}_golden.npz, written by
tools/gen_golden.py from the reference's two test pairs: EuRoC 752x480, and the 512x512 TUM pair
test/test_Stereo.cpp loads, a square frame with a single quadtree root): no oracle call in this file, only
liborbx through the C ABI and the stored vectors."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=["euroc", "tum"])
def pair(request):
    fix = {"euroc": "EuRoC", "tum": "TUM"}[request.param]
    g = np.load(os.path.join(ROOT, "tests", "golden", request.param + "_golden.npz"))
    left = np.load(os.path.join(ROOT, "tests", "fixtures", fix + "_l.npy"))
    right = np.load(os.path.join(ROOT, "tests", "fixtures", fix + "_r.npy"))
    return g, left, right


def test_extraction_matches_golden(pair):
    from orb_slam_amd import ORBextractor
    golden, euroc_left, euroc_right = pair
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        kl, dl = ex(euroc_left)
        assert np.array_equal(kl, golden["kps_l"]) and np.array_equal(dl, golden["desc_l"])
        assert np.array_equal(ex.debug_level_counts(), golden["counts_l"])
        for l in range(8):
            plane = np.ascontiguousarray(ex.pyramid_level(l, border=19))
            assert hashlib.sha256(plane.tobytes()).hexdigest() == str(golden["pyramid_sha256_l"][l])
            assert len(ex.debug_candidates(l)) == int(golden["ncand_l"][l])
        # single call, then the same pair as one batch
        kr, dr = ex(euroc_right)
        assert np.array_equal(kr, golden["kps_r"]) and np.array_equal(dr, golden["desc_r"])
        assert np.array_equal(ex.debug_level_counts(), golden["counts_r"])
        res = ex.extract_batch(np.stack([euroc_left, euroc_right]))
        assert np.array_equal(res[0][0], golden["kps_l"]) and np.array_equal(res[0][1], golden["desc_l"])
        assert np.array_equal(res[1][0], golden["kps_r"]) and np.array_equal(res[1][1], golden["desc_r"])
    finally:
        ex.close()


def test_matching_matches_golden(pair):
    from orb_slam_amd import ORBextractor, ORBmatcher
    golden, left, _ = pair
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        m = ORBmatcher(ex)
        kl, dl, kr, dr = golden["kps_l"], golden["desc_l"], golden["kps_r"], golden["desc_r"]
        bi, bd, sd = m.match_bruteforce(dl, dr)
        assert np.array_equal(bi, golden["bf_best_idx"]) and np.array_equal(bd, golden["bf_best_dist"])
        assert np.array_equal(sd, golden["bf_second_dist"])
        mbf, mb = (float(v) for v in golden["stereo_params"])
        si, sdist, ur, dp, nm = m.stereo_match(kl, dl, kr, dr, left.shape[0], mbf, mb)
        assert np.array_equal(si, golden["stereo_best_idx"]) and np.array_equal(sdist, golden["stereo_best_dist"])
        assert nm == int(golden["stereo_nmatched"])
        assert np.array_equal(ur, golden["stereo_u_right"]) and np.array_equal(dp, golden["stereo_depth"])
    finally:
        ex.close()
