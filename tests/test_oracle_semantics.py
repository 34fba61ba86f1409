"""Independent double-precision derivations of the OpenCV operations the oracle restates with
integer arithmetic.  They cannot pin the oracle to OpenCV 3.4.1 bit for bit (nothing here can:
OpenCV is absent and the reference holds no fixtures), but they bound how far a wrong belief
about a rounding rule could move a result: one grey level / 0.3 degrees."""
import math
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import ORACLE_DIR


def _bilinear_double(src, dw, dh):
    """cv::resize(INTER_LINEAR) sampling rule in double: centre-aligned, clamped at the edges"""
    sh, sw = src.shape
    s = src.astype(np.float64)

    def axis(n_dst, n_src):
        scale = n_src / n_dst
        f = (np.arange(n_dst) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        fr = f - i0
        fr[i0 < 0] = 0.0
        i0[i0 < 0] = 0
        over = i0 >= n_src - 1
        i0[over] = n_src - 1
        fr[over] = 0.0
        return i0, np.minimum(i0 + 1, n_src - 1), fr
    x0, x1, fx = axis(dw, sw)
    y0, y1, fy = axis(dh, sh)
    top = s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx
    bot = s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx
    return top * (1 - fy)[:, None] + bot * fy[:, None]


@pytest.mark.parametrize("sw,sh,dw,dh", [(752, 480, 627, 400), (627, 400, 522, 333), (1241, 376, 1034, 313),
                                          (210, 134, 175, 112), (97, 61, 81, 51), (64, 48, 53, 40)])
def test_resize_within_one_level_of_double_bilinear(oracle, sw, sh, dw, dh):
    rng = np.random.default_rng(sw * 7 + dh)
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    src[: sh // 3] = np.clip(np.linspace(0, 255, sw)[None, :] + rng.normal(0, 2, (sh // 3, sw)), 0, 255).astype(np.uint8)
    got = oracle.resize(src, dw, dh).astype(np.float64)
    ref = _bilinear_double(src, dw, dh)
    err = np.abs(got - ref)
    assert err.max() <= 1.0, err.max()
    # the two truncating shifts of the vertical pass lose up to half a level before the final rounding
    assert (err <= 0.5 + 1e-9).mean() > 0.8 and abs((got - ref).mean()) < 0.25
    # extremes are preserved
    assert (oracle.resize(np.full((sh, sw), 255, np.uint8), dw, dh) == 255).all()
    assert (oracle.resize(np.zeros((sh, sw), np.uint8), dw, dh) == 0).all()


def test_resize_exact_cases(oracle):
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (60, 90), dtype=np.uint8)
    assert np.array_equal(oracle.resize(src, 90, 60), src)                      # 1:1 is a copy
    # 2:1: both coefficients are 1024 / 2048 in both directions -> (a + b + c + d + 2) >> 2
    s = src.astype(np.int32)
    expect = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(oracle.resize(src, 45, 30), expect.astype(np.uint8))


def test_gaussian_within_one_level_of_double_convolution(oracle):
    # cv::getGaussianKernel(7, 2): exp(-x^2 / (2 sigma^2)), normalised
    g = np.exp(-(np.arange(7) - 3.0) ** 2 / 8.0)
    g /= g.sum()
    assert [int(round(256 * x)) for x in g] == [18, 34, 49, 55, 49, 34, 18]      # the 8.8 fixed-point taps
    rng = np.random.default_rng(17)
    for k in range(3):
        h, w = 70 + 11 * k, 90 + 7 * k
        plane = rng.integers(0, 256, (h + 6, w + 6), dtype=np.uint8)
        if k == 1:
            plane[:, : w // 2] = 255                                            # saturation: taps sum to 257 / 256
        got = oracle.blur_bordered(plane, 3).astype(np.float64)
        p = plane.astype(np.float64)
        ref = np.zeros((h, w))
        for dy in range(7):
            for dx in range(7):
                ref += g[dy] * g[dx] * p[dy:dy + h, dx:dx + w]
        # the integer taps over-weight by (257 / 256)^2 - 1 = 0.8 %: up to 2 levels at 255 before saturation
        err = np.abs(got - np.minimum(ref * (257.0 / 256.0) ** 2, 255.0))
        assert err.max() <= 1.0, err.max()
        assert np.abs(got - ref).max() <= 3.0


def test_fast_atan2_against_libm(oracle):
    # dense grid of moment pairs (the arguments are integer moments cast to float)
    rng = np.random.default_rng(23)
    worst = 0.0
    ys = np.concatenate([rng.integers(-2000000, 2000000, 20000), rng.integers(-300, 300, 5000)])
    xs = np.concatenate([rng.integers(-2000000, 2000000, 20000), rng.integers(-300, 300, 5000)])
    for y, x in zip(ys, xs):
        if x == 0 and y == 0:
            continue
        a = oracle.fast_atan2(float(y), float(x))
        ref = math.degrees(math.atan2(float(y), float(x))) % 360.0
        d = abs(a - ref)
        worst = max(worst, min(d, 360.0 - d))
        assert 0.0 <= a < 360.0 or a == 360.0
    assert worst <= 0.3, worst
    # the 8 octant boundaries: exact up to the polynomial's value at c = 0 and c = 1
    for y, x, ref in [(0, 1, 0), (1, 1, 45), (1, 0, 90), (1, -1, 135), (0, -1, 180), (-1, -1, 225), (-1, 0, 270), (-1, 1, 315)]:
        a = oracle.fast_atan2(float(y), float(x))
        assert abs(a - ref) <= 0.01, (y, x, a)
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    assert oracle.fast_atan2(0.0, 5.0) == 0.0 and oracle.fast_atan2(7.0, 0.0) == 90.0


def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The whole restatement (extraction literal + derived, matching, both stereo flows, grid,
    undistortion) on one image pair, built with -fsanitize=address,undefined: any out-of-bounds
    access or undefined operation aborts the run."""
    exe = os.path.join(ORACLE_DIR, "_san", "oracle_selftest")
    subprocess.check_call(["make", "-C", ORACLE_DIR, "sanitize"], stdout=subprocess.DEVNULL)
    img_l = os.path.join(os.path.dirname(ORACLE_DIR), "tests", "fixtures", "EuRoC_l.npy")
    img_r = os.path.join(os.path.dirname(ORACLE_DIR), "tests", "fixtures", "EuRoC_r.npy")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([exe, img_l, img_r], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    out = p.stdout.decode()
    assert "selftest ok" in out, out
    assert "runtime error" not in p.stderr.decode()
