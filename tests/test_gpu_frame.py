"""GPU parity of the keypoint maps that follow extraction (Frame::UndistortKeyPoints[Stereo],
src/Frame.cc:745-852) against the CPU oracle.  cv::undistortPoints is +,*,/ in double: results
must be bit-identical.  The fisheye model calls tan(): the device and glibc implementations may
differ in the last bit of the double, so the float result is allowed 1 ulp."""
import numpy as np
import pytest

from orb_slam_amd import KP_DTYPE

pytestmark = pytest.mark.gpu

K = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]], np.float64)
D4 = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float64)
D5 = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.011], np.float64)
D8 = np.array([-0.2834, 0.0739, 1.9e-4, 1.7e-5, 0.011, 0.02, -0.01, 0.003], np.float64)
D12 = np.concatenate([D8, [1e-3, -2e-3, 5e-4, 1e-4]])
R = np.array([[0.999966, -0.001422, 0.008111], [0.001365, 0.999974, 0.007073], [-0.008121, -0.007062, 0.999942]])
P = np.array([[435.2, 0, 367.45, 0], [0, 435.2, 252.2, 0], [0, 0, 1, 0]], np.float64)


@pytest.fixture(scope="module")
def gpu():
    from orb_slam_amd import ORBextractor, Frame
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    yield ex, Frame(ex)
    ex.close()


def _kps(n, seed):
    rng = np.random.default_rng(seed)
    k = np.zeros(n, KP_DTYPE)
    k["x"] = (rng.random(n) * 752).astype(np.float32)
    k["y"] = (rng.random(n) * 480).astype(np.float32)
    k["size"] = 31.0
    k["angle"] = (rng.random(n) * 360).astype(np.float32)
    k["response"] = rng.integers(7, 200, n)
    k["octave"] = rng.integers(0, 8, n)
    k["class_id"] = -1
    return k


@pytest.mark.parametrize("dist", [D4, D5, D8, D12])
@pytest.mark.parametrize("rect", [False, True])
def test_undistort_matches_oracle(gpu, oracle, dist, rect):
    from orb_slam_amd import CameraModel
    ex, fr = gpu
    kps = _kps(3000, 11)
    cam = CameraModel(K, dist, R if rect else None, P if rect else K)
    out = fr.UndistortKeyPoints(kps, cam)
    exp = oracle.undistort_points(np.stack([kps["x"], kps["y"]], 1), K, dist, R if rect else None, P if rect else K)
    assert np.array_equal(out["x"], exp[:, 0]) and np.array_equal(out["y"], exp[:, 1])
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(out[f], kps[f])


def test_undistort_shortcut_and_edge_cases(gpu, oracle):
    from orb_slam_amd import CameraModel, OrbxError
    ex, fr = gpu
    kps = _kps(100, 12)
    # first coefficient zero: the reference copies the keypoints (src/Frame.cc:747, :820)
    out = fr.UndistortKeyPoints(kps, CameraModel(K, np.array([0.0, 0.1, 0, 0]), None, K))
    assert np.array_equal(out, kps)
    # empty list
    assert len(fr.UndistortKeyPoints(kps[:0], CameraModel(K, D4, None, K))) == 0
    # unsupported coefficient counts are refused, not guessed
    with pytest.raises(OrbxError):
        fr.UndistortKeyPoints(kps, CameraModel(K, np.ones(14) * 1e-3, None, K))
    with pytest.raises(OrbxError):
        fr.UndistortKeyPoints(kps, CameraModel(K, np.ones(3) * 1e-3, None, K))


def test_undistort_fisheye(gpu, oracle):
    from orb_slam_amd import CameraModel
    ex, fr = gpu
    kps = _kps(3000, 13)
    kf = np.array([-0.01, 0.02, -0.005, 0.001])
    Kf = np.array([[380.0, 0, 376.0], [0, 380.0, 240.0], [0, 0, 1]])
    out = fr.UndistortKeyPoints(kps, CameraModel(Kf, kf, R, P, fisheye=True))
    exp = oracle.undistort_points(np.stack([kps["x"], kps["y"]], 1), Kf, kf, R, P, fisheye=True)
    for got, want in ((out["x"], exp[:, 0]), (out["y"], exp[:, 1])):
        ulp = np.spacing(np.abs(want).astype(np.float32))
        assert (np.abs(got - want) <= ulp).all()
        assert (got == want).mean() > 0.999


def test_undistort_device_batch_on_extraction_output(gpu, oracle):
    """The batched form runs in place on the buffers extract_batch_device fills."""
    import torch
    from orb_slam_amd import CameraModel
    from orb_slam_amd.synth import synthetic_batch
    ex, fr = gpu
    imgs = synthetic_batch(3, 752, 480, seed0=5100)
    dev = torch.device("cuda:0")
    d_imgs = torch.from_numpy(imgs).to(dev)
    cap = ex.max_keypoints(752, 480)
    d_kps = torch.zeros((3, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((3, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((3,), dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_imgs, 3, 752, 480, 752, 752 * 480, d_kps, d_desc, cap, d_n)
    torch.cuda.synchronize()
    before = d_kps.cpu().numpy().copy()
    n = d_n.cpu().numpy()
    fr.undistort_keypoints_device(d_kps, d_n, 0, cap, 3, CameraModel(K, D5, R, P), d_kps)
    torch.cuda.synchronize()
    after = d_kps.cpu().numpy()
    for b in range(3):
        exp = oracle.undistort_points(before[b, :n[b], :2], K, D5, R, P)
        assert np.array_equal(after[b, :n[b], :2], exp)
        # bit patterns: octave / class_id are integers stored in the float view (class_id -1 reads as NaN)
        assert np.array_equal(after[b, :n[b], 2:].view(np.int32), before[b, :n[b], 2:].view(np.int32))
        assert np.array_equal(after[b, n[b]:].view(np.int32), before[b, n[b]:].view(np.int32))   # slots past the count are untouched


@pytest.mark.gpu
def test_detections_written_by_the_gpu_path_replay_bit_identically(tmp_path, euroc_left, euroc_right):
    """The detection-I/O bypass end to end (src/Frame.cc:590-609 writes, include/Util.hpp:400-449 reads): a stereo
    pair extracted on the GPU, exported, read back as the tracker would, equals the oracle's extraction, and the
    stereo matches computed from the replayed lists equal the oracle's."""
    from orb_slam_amd import ORBextractor, ORBmatcher, KP_DTYPE
    from orb_slam_amd.detection_io import ExportToYML, KeyPointLog
    from oracle_lib import Oracle
    exl, exr = ORBextractor(1000, 1.2, 8, 20, 7), ORBextractor(1000, 1.2, 8, 20, 7)
    kl, dl = exl(euroc_left)
    kr, dr = exr(euroc_right)
    p = tmp_path / "frame.yml"
    ExportToYML(p, 1403636579.763555527, kl, dl, kr, dr)
    log = KeyPointLog()
    assert log.ReadFromYAML(p)
    o = Oracle(1000, 1.2, 8, 20, 7)
    kl_o, dl_o = o.extract(euroc_left)
    kr_o, dr_o = o.extract(euroc_right)
    for a, b in ((log.mvKeysLeft, kl_o), (log.mvKeysRight, kr_o)):
        for f in KP_DTYPE.names:
            assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(log.mDiscLeft, dl_o) and np.array_equal(log.mDiscRight, dr_o)
    mbf, mb = 47.9, 0.11
    got = ORBmatcher(exl).stereo_match(log.mvKeysLeft, log.mDiscLeft, log.mvKeysRight, log.mDiscRight, 480, mbf, mb, 0.0)
    sf = o.tables()["scale"]
    f = o.stereo_frame(kl_o, dl_o, kr_o, dr_o, sf, 480, mbf, mb, 0.0)
    nm = f.ComputeStereoMatches_Undistorted(False)
    assert got[4] == nm and nm > 100
    assert np.array_equal(got[2], f.mvuRight[:len(kl_o)]) and np.array_equal(got[3], f.mvDepth[:len(kl_o)])
    exl.close()
    exr.close()


@pytest.mark.gpu
def test_cpp_abi_sweep(tmp_path):
    """tests/cpp/abi_sweep.cpp: the host-array entry points no other C++ program reaches (candidate lists, fused grid
    search, windowed and brute-force matching, the three stereo searches, undistortion, extraction of odd sizes) on
    empty, tiny and ordinary sizes with every output buffer at its exact size.  Results are compared with the CPU
    restatement in the Python tests; this program is what tools/host_sanitizers.sh runs under AddressSanitizer and
    UndefinedBehaviorSanitizer (ORBX_CPP_DIR points at that build)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(os.environ.get("ORBX_CPP_DIR", os.path.join(root, "tests", "cpp")), "abi_sweep")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "abi_sweep"])
    p = subprocess.run([exe, "all", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280)
    assert p.returncode == 0, (p.stdout.decode()[-500:], p.stderr.decode()[-3000:])
    assert b"0 failed expectations" in p.stdout
