"""The segment test of FAST-9/16 against an implementation that shares nothing with this repository:
scikit-image's corner_fast (n = 9), where the image has a second Python with scikit-image installed
(/opt/conda/bin/python3.9).  It is run as a child process on crops of the reference's own test
images and of a synthetic frame; its corner mask (before any non-maximum suppression) must equal
"threshold-free score >= t" of the oracle for both thresholds of the extractor.  This pins the ring
offsets, the wrap-around of the 9-arc and the strict inequality; it says nothing about scores."""
import os
import subprocess

import numpy as np
import pytest

OTHER_PYTHON = "/opt/conda/bin/python3.9"

CHILD = r"""
import sys, warnings
warnings.filterwarnings("ignore")
import numpy as np
from skimage.feature import corner_fast
img = np.load(sys.argv[1]).astype(np.float64)          # integers 0..255 as floats: compared as they are
out = []
for t in (7, 20):
    # ring > centre + t  <=>  ring > centre + t + 0.5 on integers: no equality in floating point
    out.append(corner_fast(img, n=9, threshold=t + 0.5) > 0)
np.save(sys.argv[2], np.stack(out))
"""


def _have_other_python():
    if not os.path.exists(OTHER_PYTHON):
        return False
    try:
        r = subprocess.run([OTHER_PYTHON, "-c", "import skimage.feature"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=120)
    except (OSError, subprocess.TimeoutExpired):
        return False
    return r.returncode == 0


@pytest.mark.skipif(not _have_other_python(), reason="no second Python with scikit-image in this image")
def test_segment_test_equals_scikit_image(oracle, euroc_left, euroc_right, tmp_path):
    from orb_slam_amd.synth import synthetic_frame
    crops = [euroc_left[100:260, 200:420], euroc_right[250:400, 380:600], synthetic_frame(240, 180, 4242)]
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    total = corners = 0
    for k, img in enumerate(crops):
        img = np.ascontiguousarray(img)
        src, dst = tmp_path / ("in%d.npy" % k), tmp_path / ("out%d.npy" % k)
        np.save(src, img)
        subprocess.run([OTHER_PYTHON, str(child), str(src), str(dst)], check=True, timeout=300,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        masks = np.load(dst)
        h, w = img.shape
        score = np.zeros((h, w), np.int32)
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                score[y, x] = oracle.fast_score(img, x, y)
        for mask, t in zip(masks, (7, 20)):
            mine = score >= t
            inner = (slice(3, h - 3), slice(3, w - 3))
            assert np.array_equal(mine[inner], mask[inner]), "threshold %d, crop %d: %d pixels differ" % (
                t, k, int((mine[inner] != mask[inner]).sum()))
            corners += int(mask[inner].sum())
        total += (h - 6) * (w - 6)
    assert total > 90000 and corners > 300


RESIZE_CHILD = r"""
import sys, warnings
warnings.filterwarnings("ignore")
import numpy as np
from skimage.transform import resize
img = np.load(sys.argv[1]).astype(np.float64)
dh, dw = int(sys.argv[3]), int(sys.argv[4])
np.save(sys.argv[2], resize(img, (dh, dw), order=1, mode="edge", anti_aliasing=False, preserve_range=True, clip=False))
"""


@pytest.mark.skipif(not _have_other_python(), reason="no second Python with scikit-image in this image")
@pytest.mark.parametrize("dw,dh", [(627, 400), (522, 333)])
def test_resize_sampling_rule_equals_scikit_image(oracle, euroc_left, tmp_path, dw, dh):
    """cv::resize(INTER_LINEAR) samples at (x + 0.5) * scale - 0.5 with edge replication; scikit-image's order-1
    resize without anti-aliasing is the same map evaluated in double precision.  The 11-bit fixed-point restatement
    stays within one grey level of it (a different sampling rule, e.g. corner-aligned, is off by tens of levels
    on a natural image)."""
    src_img = euroc_left if dw == 627 else oracle.resize(euroc_left, 627, 400)
    child = tmp_path / "child_resize.py"
    child.write_text(RESIZE_CHILD)
    src, dst = tmp_path / "in.npy", tmp_path / "out.npy"
    np.save(src, np.ascontiguousarray(src_img))
    subprocess.run([OTHER_PYTHON, str(child), str(src), str(dst), str(dh), str(dw)], check=True, timeout=300,
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    ref = np.load(dst)
    got = oracle.resize(src_img, dw, dh).astype(np.float64)
    err = np.abs(got - ref)
    assert err.max() <= 1.0, err.max()
    assert err.mean() < 0.35


SKIMAGE_PATTERN = "/opt/conda/lib/python3.9/site-packages/skimage/feature/orb_descriptor_positions.txt"


@pytest.mark.skipif(not os.path.exists(SKIMAGE_PATTERN), reason="scikit-image's copy of the rBRIEF pattern is not in this image")
def test_rbrief_pattern_equals_scikit_image_copy(oracle):
    """scikit-image ships its own copy of OpenCV's learned 256-pair test pattern (x0, y0, x1, y1 per row): the 1024
    integers the oracle and the kernels use (data taken from src/ORBextractor.cc:152-410) are the same."""
    theirs = np.loadtxt(SKIMAGE_PATTERN)
    assert theirs.shape == (256, 4)
    assert np.array_equal(theirs, np.rint(theirs))
    assert np.array_equal(oracle.pattern().astype(np.int64), theirs.astype(np.int64))


GAUSS_CHILD = r"""
import sys, warnings
warnings.filterwarnings("ignore")
import numpy as np
from scipy.ndimage import gaussian_filter
img = np.load(sys.argv[1]).astype(np.float64)
# radius int(truncate * sigma + 0.5) = 3: seven taps exp(-x^2 / 8) normalised; 'mirror' = d c b | a b c d | c b a
np.save(sys.argv[2], gaussian_filter(img, sigma=2.0, truncate=1.5, mode="mirror"))
"""


@pytest.mark.skipif(not _have_other_python(), reason="no second Python with scipy in this image")
def test_gaussian_equals_scipy(oracle, euroc_left, tmp_path):
    """GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) against scipy.ndimage.gaussian_filter of the second Python with the
    same support and border rule: within one grey level once the 8.8 fixed-point taps' extra weight (257 / 256 per
    pass) is taken into account."""
    img = np.ascontiguousarray(euroc_left[:240, :320])
    child = tmp_path / "child_gauss.py"
    child.write_text(GAUSS_CHILD)
    src, dst = tmp_path / "in.npy", tmp_path / "out.npy"
    np.save(src, img)
    subprocess.run([OTHER_PYTHON, str(child), str(src), str(dst)], check=True, timeout=300,
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    ref = np.load(dst)
    got = oracle.blur_bordered(np.pad(img, 3, mode="reflect"), 3).astype(np.float64)
    assert got.shape == ref.shape
    err = np.abs(got - np.minimum(ref * (257.0 / 256.0) ** 2, 255.0))
    assert err.max() <= 1.0, err.max()
    assert err.mean() < 0.3


ORIENT_CHILD = r"""
import sys, warnings
warnings.filterwarnings("ignore")
import numpy as np
from skimage.feature import corner_orientations
from skimage.feature.orb import OFAST_MASK
img = np.load(sys.argv[1]).astype(np.float64)
corners = np.load(sys.argv[2])                      # (row, col)
np.save(sys.argv[3], corner_orientations(img, corners, OFAST_MASK))
"""


@pytest.mark.skipif(not _have_other_python(), reason="no second Python with scikit-image in this image")
def test_patch_moments_equal_scikit_image(oracle, euroc_left, tmp_path):
    """IC_Angle's circular 31 x 31 patch (row extents umax[|v|], src/ORBextractor.cc:74-101, :1010-1025): scikit-image's
    corner_orientations with its own oFAST mask gives atan2(m01, m10) of the same patch in double precision; the
    oracle's integer moments must give the same angle."""
    import math
    rng = np.random.default_rng(3)
    img = np.ascontiguousarray(euroc_left)
    h, w = img.shape
    pts = np.stack([rng.integers(20, h - 20, 300), rng.integers(20, w - 20, 300)], axis=1)      # (row, col)
    child = tmp_path / "child_orient.py"
    child.write_text(ORIENT_CHILD)
    a, b, c = tmp_path / "img.npy", tmp_path / "pts.npy", tmp_path / "ang.npy"
    np.save(a, img)
    np.save(b, pts)
    subprocess.run([OTHER_PYTHON, str(child), str(a), str(b), str(c)], check=True, timeout=300,
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    theirs = np.load(c)
    worst = 0.0
    for (y, x), ang in zip(pts, theirs):
        m01, m10 = oracle.ic_moments(img, int(x), int(y))
        mine = math.atan2(m01, m10)
        d = abs(mine - float(ang))
        worst = max(worst, min(d, 2 * math.pi - d))
    assert worst < 1e-9, worst
