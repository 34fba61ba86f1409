"""GPU parity tests of the extraction path: every stage of liborbx (through the
C ABI) against the CPU oracle on the same inputs.  Bit-exact for pixels,
candidates, retained sets and descriptors; orientation within 1e-3 degrees
(it is expected to be bit-identical as well)."""
import numpy as np
import pytest

from oracle_lib import Oracle
from orb_slam_amd.synth import synthetic_frame, synthetic_batch

pytestmark = pytest.mark.gpu

ANGLE_TOL_DEG = 1e-3   # tolerance stated by the north star for orientation


@pytest.fixture(scope="module")
def gpu():
    from orb_slam_amd import ORBextractor
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    yield ex
    ex.close()


def _compare_frame(ex, oracle, img, frame=0, check_stages=True):
    kps_o, desc_o = oracle.extract(img, literal=False)
    counts_o = oracle.last_level_counts()
    if check_stages:
        oracle.compute_pyramid(img)
        for l in range(oracle.nlevels):
            pl = oracle.pyramid_plane(l)
            got = ex.pyramid_level(l, frame=frame, border=19)
            assert got.shape == pl.shape
            assert np.array_equal(got, pl), "pyramid level %d differs" % l
            blur_o = oracle.blur_bordered(pl, 19)
            assert np.array_equal(ex.debug_blurred_level(l, frame=frame), blur_o), "blur level %d differs" % l
            cand_o = oracle.detect_level(l, literal=False)
            cand_g = ex.debug_candidates(l, frame=frame)
            co = np.stack([cand_o["x"], cand_o["y"], cand_o["response"]], 1) if len(cand_o) else np.zeros((0, 3), np.int32)
            assert np.array_equal(cand_g, co), "FAST candidates of level %d differ" % l
        assert np.array_equal(ex.debug_level_counts(frame=frame), counts_o)
    return kps_o, desc_o


def _assert_features(kps_g, desc_g, kps_o, desc_o):
    assert len(kps_g) == len(kps_o)
    if len(kps_o) == 0:
        return
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps_g[f], kps_o[f]), f
    dang = np.abs(kps_g["angle"] - kps_o["angle"])
    dang = np.minimum(dang, 360.0 - dang)
    assert dang.max() <= ANGLE_TOL_DEG
    same = (desc_g == desc_o).all(axis=1)
    assert same.mean() >= 0.999, "only %.4f of descriptors identical" % same.mean()


def test_euroc_frame_all_stages(gpu, oracle, euroc_left):
    kps_g, desc_g = gpu(euroc_left)
    kps_o, desc_o = _compare_frame(gpu, oracle, euroc_left)
    _assert_features(kps_g, desc_g, kps_o, desc_o)
    # stronger than the stated bar: everything is expected to be bit-identical
    assert np.array_equal(kps_g["angle"], kps_o["angle"])
    assert np.array_equal(desc_g, desc_o)


def test_literal_oracle_agrees(gpu, oracle, euroc_left):
    kps_g, desc_g = gpu(euroc_left)
    kps_l, desc_l = oracle.extract(euroc_left, literal=True)
    _assert_features(kps_g, desc_g, kps_l, desc_l)


@pytest.mark.parametrize("seed", [1234, 1235, 99])
def test_synthetic_frames(gpu, oracle, seed):
    img = synthetic_frame(752, 480, seed)
    kps_g, desc_g = gpu(img)
    kps_o, desc_o = _compare_frame(gpu, oracle, img)
    _assert_features(kps_g, desc_g, kps_o, desc_o)
    assert np.array_equal(desc_g, desc_o)


def test_batch_equals_single(gpu, oracle):
    imgs = synthetic_batch(6, 752, 480, seed0=4000)
    res = gpu.extract_batch(imgs)
    for i in (0, 3, 5):
        kps_o, desc_o = _compare_frame(gpu, oracle, imgs[i], frame=i, check_stages=(i == 3))
        _assert_features(res[i][0], res[i][1], kps_o, desc_o)
        assert np.array_equal(res[i][1], desc_o)


def test_large_batch_shapes(gpu, oracle, euroc_left):
    """From 32 frames on every frame's FAST cells and keypoints stay on one XCD and the quadtree runs
    in its batch shape: 40 frames (natural, synthetic and noise) against the oracle."""
    imgs = synthetic_batch(40, 752, 480, seed0=4400)
    imgs[7] = euroc_left
    imgs[33] = np.random.default_rng(8).integers(0, 256, size=(480, 752), dtype=np.uint8)
    res = gpu.extract_batch(imgs)
    for i in (0, 7, 15, 33, 39):
        kps_o, desc_o = _compare_frame(gpu, oracle, imgs[i], frame=i, check_stages=(i in (7, 33)))
        _assert_features(res[i][0], res[i][1], kps_o, desc_o)
        assert np.array_equal(res[i][1], desc_o)


@pytest.mark.parametrize("w,h,nfeat,pad,gap", [(752, 480, 1000, 0, 0), (752, 480, 1000, 12, 4096), (640, 481, 800, 4, 8),
                                                   (1024, 256, 1200, 0, 0), (316, 243, 500, 0, 0), (64, 64, 100, 8, 0),
                                                   (2048, 1100, 3000, 0, 0), (1241, 376, 2000, 0, 0), (318, 243, 500, 0, 0),
                                                   (755, 480, 1000, 0, 3), (642, 300, 800, 5, 1), (65, 67, 100, 0, 0)])
def test_level0_read_from_the_image(w, h, nfeat, pad, gap, monkeypatch):
    """Resident batches read level 0 where it lies instead of copying it into the pyramid block first (FAST cells,
    blur with its own REFLECT_101, resize into level 1, orientation patches); the level-0 plane is made on demand.
    Every stage against the oracle, with a row stride wider than the image and a gap between frames, widths and
    strides that are no multiples of 4 (rows that start at any byte, a last dword column of 1 to 3 pixels), an odd
    base address, and the same call with the copy switched back on."""
    import torch
    from orb_slam_amd import ORBextractor
    o = Oracle(nfeat, 1.2, 8, 20, 7)
    ex = ORBextractor(nfeat, 1.2, 8, 20, 7)
    dev = torch.device("cuda:0")
    B = 3
    imgs = synthetic_batch(B, w, h, seed0=6100 + w)
    imgs[1] = np.random.default_rng(w).integers(0, 256, size=(h, w), dtype=np.uint8)      # noise: every edge pixel matters
    stride, fstride = w + pad, (w + pad) * h + gap
    buf = np.full(B * fstride + 64, 0xA5, np.uint8)
    for b in range(B):
        buf[b * fstride:b * fstride + stride * h].reshape(h, stride)[:, :w] = imgs[b]
    cap = ex.max_keypoints(w, h)
    results = {}
    for mode in ("image", "copy", "image, unaligned base"):
        if mode == "copy":
            monkeypatch.setenv("ORBX_NO_RAW0", "1")
        else:
            monkeypatch.delenv("ORBX_NO_RAW0", raising=False)
        shift = 1 if "unaligned" in mode else 0                  # an odd base address
        d_buf = torch.from_numpy(np.concatenate([np.zeros(shift, np.uint8), buf])).to(dev)
        d_k = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        d_d = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
        ex.extract_batch_device(d_buf.data_ptr() + shift, B, w, h, stride, fstride, d_k, d_d, cap, d_n)
        torch.cuda.synchronize()
        n = d_n.cpu().numpy()
        k = d_k.cpu().numpy().view(np.uint8).reshape(B, cap, 28)
        d = d_d.cpu().numpy()
        for b in range(B):
            kps_o, desc_o = _compare_frame(ex, o, imgs[b], frame=b, check_stages=(mode != "copy" or b == 1))
            assert n[b] == len(kps_o)
            from orb_slam_amd import KP_DTYPE
            kg = k[b, :n[b]].copy().view(KP_DTYPE).reshape(n[b])
            assert np.array_equal(kg, kps_o) and np.array_equal(d[b, :n[b]], desc_o)
        results[mode] = (n.copy(), k.copy(), d.copy())
        del d_buf
    for mode in ("copy", "image, unaligned base"):
        for a, b_ in zip(results["image"], results[mode]):
            assert np.array_equal(a, b_)
    ex.close()


def test_degenerate_frames(gpu, oracle):
    # constant image: no keypoints, descriptors released
    k, d = gpu(np.full((480, 752), 77, np.uint8))
    assert len(k) == 0 and d is None
    # pure noise: candidate-capacity stress
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(480, 752), dtype=np.uint8)
    kps_g, desc_g = gpu(img)
    kps_o, desc_o = _compare_frame(gpu, oracle, img)
    _assert_features(kps_g, desc_g, kps_o, desc_o)
    # empty image: nothing happens
    k, d = gpu(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d is None


@pytest.mark.parametrize("size,nfeat", [((376, 1241), 2000), ((243, 317), 500), ((480, 640), 1000)])
def test_other_geometries(size, nfeat):
    from orb_slam_amd import ORBextractor
    h, w = size
    ex = ORBextractor(nfeat, 1.2, 8, 20, 7)
    orc = Oracle(nfeat, 1.2, 8, 20, 7)
    img = synthetic_frame(w, h, 777)
    kps_g, desc_g = ex(img)
    kps_o, desc_o = _compare_frame(ex, orc, img)
    _assert_features(kps_g, desc_g, kps_o, desc_o)
    ex.close()


def test_quadtree_batch_shape_on_small_batch(gpu, oracle, monkeypatch):
    """Small batches run the quadtree on 1024-thread workgroups, medium ones on 512-thread and large
    ones on 256-thread ones.  Force the other two shapes on a small batch: same keypoints."""
    imgs = synthetic_batch(2, 752, 480, seed0=4300)
    for mid_max in ("0", "1000"):          # the 256-thread shape, then the 512-thread one
        monkeypatch.setenv("ORBX_QT_WIDE_MAX", "0")
        monkeypatch.setenv("ORBX_QT_MID_MAX", mid_max)
        res = gpu.extract_batch(imgs)
        for i in range(2):
            kps_o, desc_o = _compare_frame(gpu, oracle, imgs[i], frame=i)
            _assert_features(res[i][0], res[i][1], kps_o, desc_o)
            assert np.array_equal(res[i][1], desc_o)


def test_fast_queue_overflow_path(oracle, monkeypatch):
    """The per-cell FAST kernel queues pretest survivors in a bounded LDS queue and falls back to
    a dense maxima search when a cell overflows it (noise).  Force the fallback with the smallest
    queue (256 entries: every cell of the noise image and the busy cells of the synthetic one
    overflow it) and require identical results, stage by stage."""
    from orb_slam_amd import ORBextractor
    monkeypatch.setenv("ORBX_FAST_QCAP", "256")
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    for img in (synthetic_frame(752, 480, 4242),
                np.random.default_rng(6).integers(0, 256, size=(480, 752), dtype=np.uint8)):
        kps_g, desc_g = ex(img)
        kps_o, desc_o = _compare_frame(ex, oracle, img)
        _assert_features(kps_g, desc_g, kps_o, desc_o)
    ex.close()


def test_separate_fast_and_blur_launches(gpu, oracle, euroc_left, monkeypatch):
    """By default the blur's work items ride in the FAST launch; ORBX_FUSE_MAX=0 runs the two kernels
    on their own: same results, stage by stage."""
    monkeypatch.setenv("ORBX_FUSE_MAX", "0")
    imgs = synthetic_batch(3, 752, 480, seed0=4900)
    imgs[1] = euroc_left
    res = gpu.extract_batch(imgs)
    for i in range(3):
        kps_o, desc_o = _compare_frame(gpu, oracle, imgs[i], frame=i)
        _assert_features(res[i][0], res[i][1], kps_o, desc_o)
        assert np.array_equal(res[i][1], desc_o)


def test_fast_variable_pitch_kernel(oracle, euroc_left, monkeypatch):
    """Cells wider than 36 pixels run the FAST kernel with run-time LDS pitches instead of the
    compiled-in ones.  Force that kernel on the usual geometry: identical results, stage by stage."""
    from orb_slam_amd import ORBextractor
    monkeypatch.setenv("ORBX_FAST_VAR_PITCH", "1")
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        for img in (euroc_left, synthetic_frame(752, 480, 4343)):
            kps_g, desc_g = ex(img)
            kps_o, desc_o = _compare_frame(ex, oracle, img)
            _assert_features(kps_g, desc_g, kps_o, desc_o)
            assert np.array_equal(desc_g, desc_o)
    finally:
        ex.close()


@pytest.mark.parametrize("cfg", [(1500, 1.5, 5, 25, 10), (800, 1.1, 8, 12, 5), (300, 2.0, 3, 20, 7), (1000, 1.2, 1, 20, 7)])
def test_other_extractor_settings(cfg, euroc_left):
    """The constructor parameters of the reference's settings files vary (ORBextractor.nFeatures,
    scaleFactor, nLevels, iniThFAST, minThFAST): every stage must follow the oracle for them too."""
    from orb_slam_amd import ORBextractor
    ex = ORBextractor(*cfg)
    orc = Oracle(*cfg)
    for img in (euroc_left, synthetic_frame(640, 480, 808)):
        kps_g, desc_g = ex(img)
        kps_o, desc_o = _compare_frame(ex, orc, img)
        _assert_features(kps_g, desc_g, kps_o, desc_o)
        assert np.array_equal(desc_g, desc_o)
    ex.close()


def test_strided_input(gpu, oracle):
    big = synthetic_frame(800, 480, 31)
    view = big[:, 24:24 + 752]           # stride 800, width 752
    kps_g, desc_g = gpu(view)
    kps_o, desc_o = oracle.extract(np.ascontiguousarray(view))
    _assert_features(kps_g, desc_g, kps_o, desc_o)
    # source rows that are not dword-addressable (odd column offset, odd stride)
    big = synthetic_frame(801, 480, 32)
    view = big[:, 13:13 + 752]
    kps_g, desc_g = gpu(view)
    kps_o, desc_o = _compare_frame(gpu, oracle, np.ascontiguousarray(view))
    _assert_features(kps_g, desc_g, kps_o, desc_o)


def test_getters_and_tables(gpu, oracle):
    t = oracle.tables()
    assert gpu.GetLevels() == 8 and gpu.GetInitThres() == 20 and gpu.GetMinThres() == 7
    assert np.array_equal(gpu.GetScaleFactors(), t["scale"])
    assert np.array_equal(gpu.GetInverseScaleFactors(), t["inv_scale"])
    assert np.array_equal(gpu.GetScaleSigmaSquares(), t["sigma2"])
    assert np.array_equal(gpu.GetInverseScaleSigmaSquares(), t["inv_sigma2"])
    assert np.array_equal(gpu.GetFeaturesPerLevel(), t["nfeat"])


def test_compute_pyramid_only(gpu, oracle, euroc_right):
    gpu.ComputePyramid(euroc_right)
    oracle.compute_pyramid(euroc_right)
    for l in range(8):
        assert np.array_equal(gpu.pyramid_level(l, border=19), oracle.pyramid_plane(l))


def test_pyramid_mirror_survives_staging_growth(oracle, euroc_left):
    """The host mirror of the pyramid and the staging area for host buffers are separate
    allocations: a call that makes the staging area grow must not disturb the mirror."""
    from orb_slam_amd import ORBextractor
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        small = np.ascontiguousarray(euroc_left[:240, :320])
        ex(small)
        first = ex.pyramid_level(0, border=19).copy()
        ex.extract_batch(np.stack([euroc_left] * 6))     # far larger uploads and downloads than before
        ex(small)
        assert np.array_equal(ex.pyramid_level(0, border=19), first)
        oracle.compute_pyramid(euroc_left)
        ex(euroc_left)
        for l in range(8):
            assert np.array_equal(ex.pyramid_level(l, border=19), oracle.pyramid_plane(l))
    finally:
        ex.close()


def test_pipelined_host_batch(gpu, oracle, euroc_left):
    """Host-fed batches of 128 frames and more run in chunks of 64 whose uploads overlap the kernels
    of the previous chunk: 130 frames (two chunks and a short one), every chunk checked against
    single-frame extraction, and the pyramids of all chunks stay readable afterwards."""
    base = synthetic_batch(13, 752, 480, seed0=4800)
    imgs = np.ascontiguousarray(np.concatenate([base] * 10))
    imgs[129] = euroc_left
    res = gpu.extract_batch(imgs)
    oracle.compute_pyramid(euroc_left)
    assert np.array_equal(gpu.pyramid_level(2, frame=129, border=19), oracle.pyramid_plane(2))
    first = gpu.pyramid_level(1, frame=5).copy()
    assert np.array_equal(gpu.pyramid_level(1, frame=5 + 65), first)       # same image, another chunk
    for i in (0, 63, 64, 127, 128, 129):
        k1, d1 = gpu(imgs[i])
        assert np.array_equal(res[i][0], k1) and np.array_equal(res[i][1], d1), i
    for i in range(13, 130 - 1):
        assert np.array_equal(res[i][1], res[i % 13][1]), i
    # the same batch from page-locked memory (direct chunk uploads instead of staged ones)
    import torch
    t = torch.empty(imgs.nbytes, dtype=torch.uint8).pin_memory()
    pin = t.numpy().reshape(imgs.shape)
    pin[...] = imgs
    res2 = gpu.extract_batch(pin)
    for i in range(130):
        assert np.array_equal(res2[i][0], res[i][0]) and np.array_equal(res2[i][1], res[i][1]), i


def test_page_locked_caller_buffers(gpu, euroc_left):
    """Caller buffers that are page-locked are copied from / to directly instead of through the
    staging area (frames lying back to back go up as one copy, strided ones as a 2-D copy):
    same results as from ordinary memory."""
    import ctypes as C
    import torch
    from orb_slam_amd.extractor import KP_DTYPE
    keep = []

    def pinned(shape, dtype):
        t = torch.empty(int(np.prod(shape)) * np.dtype(dtype).itemsize, dtype=torch.uint8).pin_memory()
        keep.append(t)
        return t.numpy().view(dtype).reshape(shape)

    imgs = synthetic_batch(5, 752, 480, seed0=4700)
    imgs[2] = euroc_left
    want = gpu.extract_batch(imgs)
    # input pinned and contiguous, outputs pinned (raw call), then input pinned with a row stride
    pin_imgs = pinned(imgs.shape, np.uint8)
    pin_imgs[...] = imgs
    got = gpu.extract_batch(pin_imgs)
    for (kw, dw), (kg, dg) in zip(want, got):
        assert np.array_equal(kw, kg) and np.array_equal(dw, dg)
    b, h, w = imgs.shape
    cap = gpu.max_keypoints(w, h)
    kps, desc, n = pinned((b, cap), KP_DTYPE), pinned((b, cap, 32), np.uint8), np.zeros(b, np.int32)
    ptrs = (C.c_void_p * b)(*[pin_imgs[i].ctypes.data for i in range(b)])
    assert gpu._lib.orbx_extract_batch(gpu._ctx, ptrs, b, w, h, w, kps.ctypes.data, desc.ctypes.data, cap, n.ctypes.data) == 0
    for i, (kw, dw) in enumerate(want):
        assert n[i] == len(kw) and np.array_equal(kps[i, :n[i]], kw) and np.array_equal(desc[i, :n[i]], dw)
    wide = pinned((480, 800), np.uint8)
    wide[...] = 0
    wide[:, :752] = euroc_left
    kg, dg = gpu(wide[:, :752])
    assert np.array_equal(kg, want[2][0]) and np.array_equal(dg, want[2][1])


def test_two_extractors_from_two_threads(euroc_left, euroc_right):
    """The reference runs the left and the right extractor of a stereo frame on two threads at once
    (src/Frame.cc:84-87): two contexts called concurrently give what they give one after the other."""
    import threading
    from orb_slam_amd import ORBextractor
    exl, exr = ORBextractor(1000, 1.2, 8, 20, 7), ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        want = [exl(euroc_left), exr(euroc_right)]
        errors = []

        def work(ex, img, ref):
            try:
                for _ in range(40):
                    k, d = ex(img)
                    if not (np.array_equal(k, ref[0]) and np.array_equal(d, ref[1])):
                        errors.append("result changed under concurrency")
                        return
                    if not np.array_equal(ex.pyramid_level(3, border=19), ex.pyramid_level(3, border=19)):
                        errors.append("pyramid mirror unstable")
                        return
            except Exception as e:      # noqa: BLE001 - reported through the list
                errors.append(repr(e))

        ts = [threading.Thread(target=work, args=(exl, euroc_left, want[0])),
              threading.Thread(target=work, args=(exr, euroc_right, want[1]))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
    finally:
        exl.close()
        exr.close()


def test_cpp_host_program(oracle, euroc_left, tmp_path):
    """The C++ host class (include/orbx_host.hpp) gives the same result as the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(os.environ.get("ORBX_CPP_DIR", os.path.join(root, "tests", "cpp")), "host_extract")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "host_extract"])
    raw = tmp_path / "img.raw"
    out = tmp_path / "out.bin"
    euroc_left.tofile(raw)
    subprocess.check_call([exe, str(raw), "752", "480", str(out)], timeout=100)
    buf = out.read_bytes()
    n, nl = np.frombuffer(buf, np.int32, 2)
    sizes = np.frombuffer(buf, np.int32, 2 * nl, 8).reshape(nl, 2)
    from oracle_lib import KP_DTYPE
    kps = np.frombuffer(buf, KP_DTYPE, n, 8 + 8 * nl)
    desc = np.frombuffer(buf, np.uint8, n * 32, 8 + 8 * nl + 28 * n).reshape(n, 32)
    kps_o, desc_o = oracle.extract(euroc_left, literal=False)
    assert [tuple(s) for s in sizes] == [oracle.level_size(752, 480, l) for l in range(nl)]
    _assert_features(kps, desc, kps_o, desc_o)
    oracle.compute_pyramid(euroc_left)
    top = oracle.pyramid_plane(nl - 1)[19:-19, 19:-19]
    assert buf[-1] == top[top.shape[0] // 2, top.shape[1] // 2]


def test_cpp_sharded_extractor_and_matcher(oracle, euroc_left, euroc_right, tmp_path):
    """orbx::ShardedExtractor (one context + host thread per listed device, contiguous frame ranges, results in the
    caller's arrays in frame order) equals a single context on every frame -- three shards on the one GPU of the
    test box, 7 frames, so the shards get 2, 2 and 3 -- and orbx::ORBmatcher::SearchForInitialization and the
    last-frame SearchByProjection on the first two frames equal the oracle's."""
    import os
    import subprocess
    from orb_slam_amd.synth import synthetic_batch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(os.environ.get("ORBX_CPP_DIR", os.path.join(root, "tests", "cpp")), "sharded_extract")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "sharded_extract"])
    frames = np.concatenate([np.stack([euroc_left, euroc_right]), synthetic_batch(5, 752, 480, seed0=8800)])
    raw = tmp_path / "frames.raw"
    out = tmp_path / "init.bin"
    frames.tofile(raw)
    p = subprocess.run([exe, str(raw), "752", "480", str(len(frames)), "0,0,0", str(out)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=150)
    assert p.returncode == 0, p.stderr.decode()
    assert "sharded ok: 7 frames over 3 shards" in p.stdout.decode()
    k1, d1 = oracle.extract(euroc_left)
    k2, d2 = oracle.extract(euroc_right)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    grid = (0.0, 0.0, float(np.float32(64) / np.float32(752)), float(np.float32(48) / np.float32(480)))
    nm_o, m_o = oracle.search_for_initialization(k1, d1, k2, d2, grid, prev, 100, 0.9, True)
    buf = out.read_bytes()
    n1, nm = np.frombuffer(buf, np.int32, 2)
    assert n1 == len(k1) and nm == nm_o and nm > 20
    assert np.array_equal(np.frombuffer(buf, np.int32, n1, 8), m_o)
    assert np.array_equal(np.frombuffer(buf, np.float32, 2 * n1, 8 + 4 * n1).reshape(n1, 2), prev)
    # orbx::ORBmatcher::SearchByProjection(current, last frame): frame 0's keypoints searched in frame 1
    off = 8 + 12 * n1
    n2, nm2 = np.frombuffer(buf, np.int32, 2, off)
    last = {"visible": np.ones(len(k1), np.uint8), "u": k1["x"].copy(), "v": k1["y"].copy(),
            "invzc": np.ones(len(k1), np.float32), "octave": k1["octave"].astype(np.int32),
            "angle": k1["angle"].astype(np.float32), "Observations": np.ones(len(k1), np.int32), "descriptor": d1}
    fo, fm = np.full(len(k2), -1, np.int32), np.full(len(k2), -1, np.int32)
    nm_o2 = oracle.search_by_projection_last(k2, d2, None, grid, last, 15.0, 0.0, False, False, True, fo, fm)
    assert n2 == len(k2) and nm2 == nm_o2 and nm2 > 20
    assert np.array_equal(np.frombuffer(buf, np.int32, n2, off + 8), fm)
    assert np.array_equal(np.frombuffer(buf, np.int32, n2, off + 8 + 4 * n2), fo)
    # orbx::ORBmatcher::GetCandidates + SearchByProjection_OnePoint, last map point first
    off += 8 + 8 * n2
    ncand, nm3 = np.frombuffer(buf, np.int32, 2, off)
    M = len(k1)
    mp = {"mbTrackInView": np.ones(M, np.uint8), "isBad": np.zeros(M, np.uint8), "mTrackProjX": k1["x"].copy(),
          "mTrackProjY": k1["y"].copy(), "mTrackProjXR": np.zeros(M, np.float32),
          "mnTrackScaleLevel": k1["octave"].astype(np.int32),
          "mTrackViewCos": np.where(np.arange(M) % 2, np.float32(0.9995), np.float32(0.9)).astype(np.float32),
          "Observations": np.ones(M, np.int32), "descriptor": d1}
    fo, fm, ms = (np.full(len(k2), -1, np.int32) for _ in range(3))
    rets = [oracle.search_by_projection_one_point(k2, d2, None, grid, mp, i, 3.0, 0.8, fo, fm, ms) for i in range(M - 1, -1, -1)]
    assert ncand == sum(len(oracle.get_candidates(k2, grid, mp, i, 3.0)) for i in range(M))
    assert nm3 == sum(r >= 0 for r in rets) and nm3 > 20
    assert np.array_equal(np.frombuffer(buf, np.int32, n2, off + 8), fm)
    assert np.array_equal(np.frombuffer(buf, np.int32, n2, off + 8 + 4 * n2), ms)


def test_cpp_adapter_program(euroc_left, tmp_path):
    """The drop-in class of include/orbx_adapter.hpp, driven like the reference's Frame drives its extractors:
    same keypoints / descriptors as the OpenCV-free host class, level views that are regions of interest inside
    their REFLECT_101-bordered planes, ComputePyramid on its own, empty-image behaviour."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(os.environ.get("ORBX_CPP_DIR", os.path.join(root, "tests", "cpp")), "adapter_frame")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "adapter_frame"])
    raw = tmp_path / "img.raw"
    euroc_left.tofile(raw)
    p = subprocess.run([exe, str(raw), "752", "480"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=100)
    assert p.returncode == 0, p.stderr.decode()
    assert "adapter ok: 1009 keypoints, 8 levels" in p.stdout.decode()


@pytest.mark.parametrize("nfeat,size,seed", [(12000, (720, 1280), 5), (30000, (1080, 1920), 6)])
def test_more_than_2048_nodes_per_level(nfeat, size, seed):
    """Levels that are asked for more than 2040 features (nFeatures above about 9400) do not fit the
    quadtree's LDS arrays; they run with the arrays in global memory and must give the same keypoints."""
    from orb_slam_amd import ORBextractor
    h, w = size
    rng = np.random.default_rng(seed)
    img = synthetic_frame(w, h, 300 + seed, nrect=1200)
    img[: h // 2] = rng.integers(0, 256, size=(h // 2, w), dtype=np.uint8)      # noise half: plenty of candidates
    ex = ORBextractor(nfeat, 1.2, 8, 20, 7)
    orc = Oracle(nfeat, 1.2, 8, 20, 7)
    assert ex.GetFeaturesPerLevel()[0] > 2040
    kps_g, desc_g = ex(img)
    kps_o, desc_o = orc.extract(img, literal=True, cap=nfeat + 1000)
    assert len(kps_o) > 0.9 * nfeat
    assert np.array_equal(ex.debug_level_counts(), orc.last_level_counts())
    _assert_features(kps_g, desc_g, kps_o, desc_o)
    assert np.array_equal(desc_g, desc_o)
    # a batch of two through the same path
    res = ex.extract_batch(np.stack([img, img[::-1].copy()]))
    assert np.array_equal(res[0][0], kps_g) and np.array_equal(res[0][1], desc_g)
    k2, d2 = orc.extract(img[::-1].copy(), literal=True, cap=nfeat + 1000)
    _assert_features(res[1][0], res[1][1], k2, d2)
    ex.close()


def test_4k_8000_features():
    """3840x2160, 8000 features: 1737-node quadtree (LDS above 64 KB), > 4096 candidates per
    level (keys beyond the register-resident ones), 8820 cells on level 0."""
    from orb_slam_amd import ORBextractor
    w, h = 3840, 2160
    img = synthetic_frame(w, h, 99, nrect=1500)
    ex = ORBextractor(8000, 1.2, 8, 20, 7)
    orc = Oracle(8000, 1.2, 8, 20, 7)
    kps_g, desc_g = ex(img)
    kps_o, desc_o = orc.extract(img, literal=True, cap=9000)
    assert len(kps_o) > 7900
    _assert_features(kps_g, desc_g, kps_o, desc_o)
    assert np.array_equal(desc_g, desc_o)
    assert np.array_equal(ex.debug_level_counts(), orc.last_level_counts())
    orc.compute_pyramid(img)
    for l in (0, 7):
        assert np.array_equal(ex.pyramid_level(l, border=19), orc.pyramid_plane(l))
        cand_o = orc.detect_level(l, literal=True)
        cand_g = ex.debug_candidates(l)
        assert np.array_equal(cand_g, np.stack([cand_o["x"], cand_o["y"], cand_o["response"]], 1))
    ex.close()


@pytest.mark.parametrize("ratio", ["0", "3"])
def test_quadtree_on_node_counts(monkeypatch, ratio):
    """Levels with very many candidates run DistributeOctTree on node counts (leaf bins filled by all keys in
    parallel); here that form is forced on every level of ordinary geometries.  ratio 0: also on levels that keep
    nearly all their candidates, whose trees grow deeper than the leaves, so that the hand-over to the passes over
    the keys inside the same workgroup runs too.  Shallow leaves (512 bins) make that hand-over frequent."""
    from orb_slam_amd import ORBextractor
    monkeypatch.setenv("ORBX_QT_COUNT_CELLS", "1")
    monkeypatch.setenv("ORBX_QT_COUNT_RATIO", ratio)
    cases = [(752, 480, 1000, 1234, {}), (1241, 376, 2000, 777, {}), (640, 200, 300, 5, {}), (1600, 1200, 500, 8, dict(nrect=900))]
    for bins in ("32768", "512"):
        monkeypatch.setenv("ORBX_QT_LEAF_BINS", bins)
        for w, h, nfeat, seed, kw in cases:
            imgs = np.stack([synthetic_frame(w, h, seed + i, **kw) for i in range(3)])
            imgs[2, : h // 2] = np.random.default_rng(seed).integers(0, 256, size=(h // 2, w), dtype=np.uint8)   # noise: a crowd of candidates
            ex = ORBextractor(nfeat, 1.2, 8, 20, 7)
            orc = Oracle(nfeat, 1.2, 8, 20, 7)
            res = ex.extract_batch(imgs)
            for i in range(3):
                kps_o, desc_o = orc.extract(imgs[i], literal=True, cap=nfeat + 1000)
                assert np.array_equal(ex.debug_level_counts(frame=i), orc.last_level_counts()), (w, h, i)
                _assert_features(res[i][0], res[i][1], kps_o, desc_o)
                assert np.array_equal(res[i][1], desc_o)
            ex.close()


def test_recorded_launch_sequence_follows_its_arguments(oracle, euroc_left, euroc_right, monkeypatch):
    """Small host-fed calls replay their launch sequence as a recorded graph from the third call with the same
    arguments on; any change (image size, number of frames, capacity, a batch on the device in between that grows
    the buffers, a diagnostic switch) must give a fresh sequence, never a stale one."""
    from orb_slam_amd import ORBextractor
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    small = np.ascontiguousarray(euroc_left[:300, :500])
    ko_l, do_l = oracle.extract(euroc_left, literal=False)
    ko_r, do_r = oracle.extract(euroc_right, literal=False)
    ko_s, do_s = oracle.extract(small, literal=False)

    def check(img, ko, do):
        kg, dg = ex(img)
        assert np.array_equal(kg, ko) and np.array_equal(dg, do)

    for _ in range(4):                      # plain, recording, two replays on changing pixels
        check(euroc_left, ko_l, do_l)
        check(euroc_right, ko_r, do_r)
    for _ in range(3):                      # another size, then back
        check(small, ko_s, do_s)
    for _ in range(3):
        check(euroc_left, ko_l, do_l)
    for _ in range(3):                      # two frames per call
        res = ex.extract_batch(np.stack([euroc_left, euroc_right]))
        assert np.array_equal(res[0][0], ko_l) and np.array_equal(res[1][1], do_r)
    big = ex.extract_batch(np.stack([euroc_left, euroc_right] * 20))       # grows the per-frame buffers
    assert np.array_equal(big[39][0], ko_r)
    for _ in range(3):
        check(euroc_right, ko_r, do_r)
    monkeypatch.setenv("ORBX_FUSE_MAX", "0")                                # another launch sequence for the same arguments
    for _ in range(3):
        check(euroc_left, ko_l, do_l)
    monkeypatch.delenv("ORBX_FUSE_MAX")
    check(euroc_left, ko_l, do_l)
    # the pyramid a caller reads after a replay is that call's
    ex(euroc_right)
    oracle.compute_pyramid(euroc_right)
    assert np.array_equal(ex.pyramid_level(3, border=19), oracle.pyramid_plane(3))
    ex.close()


def test_recorded_launch_sequence_survives_geometry_changes_from_other_entry_points(oracle, euroc_left):
    """The recording holds the addresses of the geometry tables.  Device-resident batches of another frame size
    free and rebuild those tables without passing through the small host-fed path, and a device batch of the first
    size then restores the geometry: the next host call must not replay the recording made before."""
    import torch
    from orb_slam_amd import ORBextractor
    from orb_slam_amd.synth import synthetic_batch
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    ko, do = oracle.extract(euroc_left, literal=False)
    dev = torch.device("cuda:0")

    def device_batch(imgs):
        b, h, w = imgs.shape
        cap = ex.max_keypoints(w, h)
        d_imgs = torch.from_numpy(imgs).to(dev)
        d_k = torch.zeros((b, cap, 7), dtype=torch.float32, device=dev)
        d_d = torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev)
        d_n = torch.zeros((b,), dtype=torch.int32, device=dev)
        ex.extract_batch_device(d_imgs, b, w, h, w, w * h, d_k, d_d, cap, d_n)
        torch.cuda.synchronize()
        return d_n.cpu().numpy()

    for _ in range(4):                                           # plain, recording, replays
        kg, dg = ex(euroc_left)
        assert np.array_equal(kg, ko) and np.array_equal(dg, do)
    for wide in (synthetic_batch(2, 1024, 256, seed0=77), synthetic_batch(2, 2048, 1100, seed0=78)):
        assert (device_batch(wide) > 0).all()                    # other geometries: larger tables, reallocated
        n = device_batch(np.stack([euroc_left, euroc_left]))     # the first geometry again, through the device path
        assert n.tolist() == [len(ko), len(ko)]
        for _ in range(3):
            kg, dg = ex(euroc_left)                              # same arguments as the recording
            assert np.array_equal(kg, ko) and np.array_equal(dg, do)
    ex.close()


def test_small_feature_budget_and_tiny_levels():
    """nfeatures = 40 (Tracking::updateORBExtractor re-creates extractors with small budgets):
    N per level below 4*nIni, and a 200x150 image whose top levels are too small to search."""
    from orb_slam_amd import ORBextractor
    for (w, h, nf) in ((752, 480, 40), (200, 150, 300), (97, 83, 100)):
        ex = ORBextractor(nf, 1.2, 8, 20, 7)
        orc = Oracle(nf, 1.2, 8, 20, 7)
        img = synthetic_frame(w, h, 5 + w, nrect=60)
        kps_g, desc_g = ex(img)
        kps_o, desc_o = orc.extract(img, literal=True)
        _assert_features(kps_g, desc_g if desc_g is not None else np.zeros((0, 32), np.uint8), kps_o, desc_o)
        assert np.array_equal(ex.debug_level_counts(), orc.last_level_counts())
        ex.close()


def test_randomised_soak():
    """A short run of tools/soak_parity.py: random sizes, contents and extractor settings, every
    keypoint field and descriptor byte identical to the oracle."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_parity.py"), "25", "3"], env=env,
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


@pytest.mark.parametrize("w,h,nlevels,scale", [(641, 481, 8, 1.2), (333, 207, 8, 1.2), (1241, 376, 8, 1.2), (500, 300, 5, 1.5),
                                                (203, 122, 8, 1.3)])
def test_batch_pyramid_borders_on_odd_sizes(w, h, nlevels, scale):
    """In batches of 8 frames and more the resize computes image rows only and stores the rows near the top / bottom edge a
    second time as the REFLECT_101 border rows (levels lower than 40 rows keep computing their border rows): every plane
    of such a batch, border included, against the oracle, for sizes whose levels fall on both sides of that limit."""
    from orb_slam_amd import ORBextractor
    from oracle_lib import Oracle
    ex = ORBextractor(600, scale, nlevels, 20, 7)
    o = Oracle(600, scale, nlevels, 20, 7)
    imgs = synthetic_batch(9, w, h, seed0=5100 + w)
    res = ex.extract_batch(imgs)
    for i in (0, 8):
        kps_o, desc_o = _compare_frame(ex, o, imgs[i], frame=i, check_stages=True)
        _assert_features(res[i][0], res[i][1], kps_o, desc_o)
        assert np.array_equal(res[i][1], desc_o)
    ex.close()


@pytest.mark.gpu
def test_two_extractors_with_large_feature_counts_take_turns():
    """Two contexts on one device whose quadtrees each need more than 64 KB of LDS, of different sizes (8000 and
    6000 features): the opt-in for large dynamic LDS belongs to the kernel, not to a context, so the second
    context's smaller request must not undo the first one's.  A, B, A again: every call succeeds and A repeats
    its result."""
    from orb_slam_amd import ORBextractor
    img = synthetic_frame(1600, 1200, 31, nrect=1500)
    a = ORBextractor(8000, 1.2, 8, 20, 7)
    ka1, da1 = a(img)
    b = ORBextractor(6000, 1.2, 8, 20, 7)
    kb, db = b(img)
    ka2, da2 = a(img)
    kb2, db2 = b(img)
    assert len(ka1) > 6000 and len(kb) > 4500
    assert np.array_equal(ka1, ka2) and np.array_equal(da1, da2)
    assert np.array_equal(kb, kb2) and np.array_equal(db, db2)
    orc = Oracle(6000, 1.2, 8, 20, 7)
    ko, do = orc.extract(img, literal=False, cap=8000)
    _assert_features(kb, db, ko, do)
    a.close()
    b.close()
