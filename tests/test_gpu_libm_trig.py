"""The descriptor criterion against the reference's own trig formulation, measured on the GPU box.
computeOrbDescriptor takes cos / sin of the keypoint angle through libm's float functions
(src/ORBextractor.cc:108-109), which are not correctly rounded; the kernel evaluates correctly rounded values.
Here the GPU output is compared with the oracle in its libm mode -- this host's cosf / sinf, what the reference
computes when it runs on this node: keypoints bit-identical (every field but the angle compared as bits),
angles within 1e-3 degree, at least 99.9 % of the descriptors bit-identical.  Frames: the 256 frames of the
benchmarked batch and the reference's two test pairs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compare_with_libm_oracle(frames, nfeat=1000, batch=64):
    """(keypoints, descriptors that differ, largest angle difference in degrees); asserts the bit-exact parts"""
    from oracle_lib import Oracle
    from orb_slam_amd import ORBextractor
    o = Oracle(nfeat, 1.2, 8, 20, 7, trig="libm")
    ex = ORBextractor(nfeat, 1.2, 8, 20, 7)
    total = differ = 0
    worst = 0.0
    try:
        for b0 in range(0, len(frames), batch):
            res = ex.extract_batch(np.stack(frames[b0:b0 + batch]))
            for i, (kg, dg) in enumerate(res):
                ko, do = o.extract(frames[b0 + i])
                assert len(kg) == len(ko), "frame %d" % (b0 + i)
                for f in ("x", "y", "size", "response", "octave", "class_id"):
                    assert np.array_equal(kg[f], ko[f]), (b0 + i, f)
                worst = max(worst, float(np.abs(kg["angle"] - ko["angle"]).max(initial=0.0)))
                total += len(kg)
                differ += int((dg != do).any(axis=1).sum())
    finally:
        ex.close()
    return total, differ, worst


def test_gpu_against_libm_trig_on_bench_frames_and_reference_pairs():
    from orb_slam_amd.synth import synthetic_batch
    frames = list(synthetic_batch(256, 752, 480, seed0=1234))
    for name in ("EuRoC_l", "EuRoC_r"):
        frames.append(np.load(os.path.join(ROOT, "tests", "fixtures", name + ".npy")))
    total, differ, worst = compare_with_libm_oracle(frames)
    frac = 1.0 - differ / total
    print("\nGPU vs oracle(libm cosf/sinf): %d keypoints on %d frames, %d descriptors differ -> %.6f identical; "
          "largest angle difference %.3g deg" % (total, len(frames), differ, frac, worst))
    assert total > 250000
    assert worst <= 1e-3
    assert frac >= 0.999
    # the square 512 x 512 pair of the reference's stereo test
    tum = [np.load(os.path.join(ROOT, "tests", "fixtures", "TUM_%s.npy" % s)) for s in "lr"]
    total, differ, worst = compare_with_libm_oracle(tum)
    assert worst <= 1e-3 and 1.0 - differ / total >= 0.999
