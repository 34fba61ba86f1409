"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).

Test infrastructure: only tests, the smoke check and the CPU-baseline leg of
the benchmark may import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<i4")])

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)


def build(native=False):
    """Compile the oracle with its Makefile if the library is missing."""
    name = "liborb_oracle_native.so" if native else "liborb_oracle.so"
    path = os.path.join(ORACLE_DIR, name)
    src = os.path.join(ORACLE_DIR, "orb_oracle.c")
    if not os.path.exists(path) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, name])
    return path


def _ptr(a, t):
    return a.ctypes.data_as(t)


class StereoState(C.Structure):
    """oo_stereo_state: mvuRight, mvDepth, mvStereoMatched, mvDistIdx of a Frame"""
    _fields_ = [("u_right", _f32p), ("depth", _f32p), ("stereo_matched", _u8p),
                ("dist_idx_dist", _i32p), ("dist_idx_il", _i32p), ("n_dist_idx", C.c_int32)]


TRIG_LIBM, TRIG_ROUNDED = 0, 1


class Oracle:
    """trig: how computeOrbDescriptor's cos / sin are taken.  "rounded" (default here) = correctly
    rounded, the formulation the GPU kernel evaluates, so that descriptor comparisons are exact;
    "libm" = this host's cosf / sinf as the reference calls them (differs in the last bit for
    about 0.1 % of all angles; see oracle/orb_oracle.h)."""

    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, native=False, trig="rounded"):
        self.lib = lib = C.CDLL(build(native))
        lib.oo_create.restype = C.c_void_p
        lib.oo_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        lib.oo_destroy.argtypes = [C.c_void_p]
        lib.oo_get_tables.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _i32p, _i32p]
        lib.oo_pattern.restype = _i32p
        lib.oo_level_size.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _i32p, _i32p]
        lib.oo_compute_pyramid.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int]
        lib.oo_pyramid_plane.restype = _u8p
        lib.oo_pyramid_plane.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p]
        lib.oo_resize_linear_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int]
        lib.oo_fast9_nms.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.oo_fast_score_bruteforce.argtypes = [C.c_void_p, C.c_int]
        lib.oo_detect_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.oo_distribute_octree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int]
        lib.oo_ic_moments.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _i32p]
        lib.oo_fast_atan2.restype = C.c_float
        lib.oo_fast_atan2.argtypes = [C.c_float, C.c_float]
        lib.oo_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]
        lib.oo_orb_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_float, _u8p]
        lib.oo_extract.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, _u8p, C.c_int, C.c_int]
        lib.oo_last_level_counts.argtypes = [C.c_void_p, _i32p]
        lib.oo_hamming.argtypes = [_u32p, _u32p]
        lib.oo_match_bruteforce.argtypes = [_u32p, C.c_int, _u32p, C.c_int, _i32p, _i32p, _i32p]
        lib.oo_match_windowed.argtypes = [_u32p, C.c_int, _u32p, C.c_int, _i32p, _i32p, _i32p,
                                          _i32p, _i32p, _i32p, _i32p, _i32p]
        lib.oo_hamming_csr.argtypes = [_u32p, C.c_int, _u32p, C.c_int, _i32p, _i32p, C.POINTER(C.c_uint16)]
        lib.oo_stereo_search.argtypes = [C.c_void_p, _u32p, C.c_int, C.c_void_p, _u32p, C.c_int, _f32p,
                                         C.c_int, C.c_float, C.c_float, C.c_float, _i32p, _i32p]
        lib.oo_stereo_finish.argtypes = [C.c_void_p, C.c_int, C.c_void_p, _i32p, _i32p,
                                         C.c_float, C.c_float, C.c_float, _f32p, _f32p]
        lib.oo_stereo_classic.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _u32p, C.c_int, C.c_void_p, _u32p, C.c_int,
                                          C.c_float, C.c_float, _i32p, _i32p, _f32p, _f32p]
        lib.oo_features_in_area.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                            _f32p, _f32p, _f32p, _i32p, _i32p, C.c_int, _i32p, _i32p, C.c_int]
        lib.oo_undistort_points.argtypes = [_f32p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_int, _f32p]
        lib.oo_set_trig_mode.argtypes = [C.c_void_p, C.c_int]
        lib.oo_orb_descriptor_mode.argtypes = [C.c_void_p, C.c_int, C.c_float, _u8p, C.c_int]
        lib.oo_count_trig_differences.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.oo_stereo_prepare.argtypes = [C.POINTER(StereoState), C.c_int]
        lib.oo_stereo_undistorted.argtypes = [C.c_void_p, _u32p, C.c_int, C.c_void_p, _u32p, C.c_int, _f32p, C.c_int,
                                              C.c_float, C.c_float, C.c_float, C.c_float, _u8p, _f32p, C.c_int,
                                              C.POINTER(StereoState), _i32p, _i32p]
        lib.oo_stereo_one_point.argtypes = [C.c_void_p, _u32p, C.c_int, C.c_void_p, _u32p, C.c_int, _f32p, C.c_int,
                                            C.c_float, _u8p, C.c_int, C.c_int, C.c_float, C.c_float,
                                            C.POINTER(StereoState), _i32p, _i32p]
        lib.oo_three_maxima.argtypes = [_i32p, C.c_int, _i32p, _i32p, _i32p]
        lib.oo_search_for_initialization.argtypes = [C.c_void_p, _u32p, C.c_int, C.c_void_p, _u32p, C.c_int,
                                                     C.c_float, C.c_float, C.c_float, C.c_float, _f32p, C.c_int,
                                                     C.c_float, C.c_int, _i32p]
        lib.oo_search_by_projection.argtypes = [C.c_void_p, _u32p, _f32p, C.c_int, _f32p, C.c_float, C.c_float,
                                                C.c_float, C.c_float, _u8p, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p,
                                                _u32p, _i32p, C.c_int, C.c_float, C.c_float, _i32p, _i32p, _i32p]
        lib.oo_search_by_projection_last.argtypes = [C.c_void_p, _u32p, _f32p, C.c_int, _f32p, C.c_float, C.c_float, C.c_float,
                                                     C.c_float, _u8p, _f32p, _f32p, _f32p, _i32p, _f32p, _u32p, _i32p,
                                                     C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, _i32p, _i32p]
        _f, _ci = C.c_float, C.c_int
        lib.oo_search_by_projection_budget.argtypes = [C.c_void_p, _u32p, _f32p, _ci, _f32p, _f, _f, _f, _f, _u8p, _u8p, _f32p,
                                                       _f32p, _f32p, _i32p, _f32p, _u32p, _i32p, _ci, _f, _f, C.c_double,
                                                       _i32p, _i32p, _i32p, _i32p]
        lib.oo_get_candidates.argtypes = [C.c_void_p, _ci, _f32p, _f, _f, _f, _f, _ci, _ci, _f, _f, _ci, _f, _f, _i32p, _ci]
        lib.oo_match_candidates.argtypes = [C.c_void_p, _u32p, _f32p, _f32p, _i32p, _ci, _ci, _ci, _f, _ci, _f, _u32p, _ci,
                                            _ci, _f, _f, _i32p, _i32p, _i32p]
        lib.oo_search_by_projection_one_point.argtypes = [C.c_void_p, _u32p, _f32p, _ci, _f32p, _f, _f, _f, _f, _ci, _ci, _f,
                                                          _f, _f, _ci, _f, _u32p, _ci, _ci, _f, _f, _i32p, _i32p, _i32p]
        lib.oo_search_by_bow_frame.argtypes = [C.c_void_p, _u32p, _u8p, _i32p, _i32p, _i32p, _ci, C.c_void_p, _u32p, _ci, _i32p,
                                               _i32p, _i32p, _ci, _f, _ci, _i32p]
        lib.oo_search_by_bow_keyframes.argtypes = [C.c_void_p, _u32p, _u8p, _ci, _i32p, _i32p, _i32p, _ci, C.c_void_p, _u32p,
                                                   _u8p, _ci, _i32p, _i32p, _i32p, _ci, _f, _ci, _i32p]
        lib.oo_search_by_projection_keyframe.argtypes = [C.c_void_p, _u32p, _ci, _f32p, _f, _f, _f, _f, _u8p, _f32p, _f32p, _i32p,
                                                         _f32p, _u32p, _i32p, _ci, _f, _ci, _ci, _i32p, _i32p]
        lib.oo_search_by_projection_sim3.argtypes = [C.c_void_p, _u32p, _ci, _f32p, _f, _f, _f, _f, _u8p, _f32p, _f32p, _i32p, _u32p,
                                                     _ci, _ci, _i32p, _i32p]
        lib.oo_fuse_search.argtypes = [C.c_void_p, _u32p, _f32p, _ci, _f32p, _f32p, _f, _f, _f, _f, _u8p, _f32p, _f32p, _f32p,
                                       _i32p, _u32p, _ci, _f, _ci, _i32p, _i32p]
        lib.oo_search_by_sim3.argtypes = [C.c_void_p, _u32p, _ci, C.c_void_p, _u32p, _ci, _f32p, _f, _f, _f, _f, _u8p, _f32p, _f32p,
                                          _i32p, _u32p, _u8p, _f32p, _f32p, _i32p, _u32p, _f, _i32p]
        lib.oo_search_for_triangulation.argtypes = [C.c_void_p, _u32p, _u8p, _u8p, _ci, _i32p, _i32p, _i32p, _ci, C.c_void_p,
                                                    _u32p, _u8p, _u8p, _i32p, _i32p, _i32p, _ci, _f32p, _f32p, _f32p, _f, _f,
                                                    _ci, _ci, _i32p]
        self.nlevels = nlevels
        self.h = lib.oo_create(nfeatures, scale, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("oo_create failed")
        self.trig = {"rounded": TRIG_ROUNDED, "libm": TRIG_LIBM}[trig]
        lib.oo_set_trig_mode(self.h, self.trig)
        self._img_shape = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.oo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- tables -----------------------------------------------------------
    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.lib.oo_get_tables(self.h, _ptr(sc, _f32p), _ptr(isc, _f32p), _ptr(s2, _f32p), _ptr(is2, _f32p),
                               _ptr(nf, _i32p), _ptr(um, _i32p))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, nfeat=nf, umax=um)

    def pattern(self):
        return np.ctypeslib.as_array(self.lib.oo_pattern(), shape=(256, 4)).copy()

    def level_size(self, w, h, level):
        lw, lh = C.c_int32(), C.c_int32()
        self.lib.oo_level_size(self.h, w, h, level, C.byref(lw), C.byref(lh))
        return lw.value, lh.value

    # ---- pyramid ----------------------------------------------------------
    def compute_pyramid(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        rc = self.lib.oo_compute_pyramid(self.h, _ptr(img, _u8p), w, h, w)
        if rc != 0:
            raise ValueError("oo_compute_pyramid failed")
        self._img_shape = (h, w)

    def pyramid_plane(self, level):
        """bordered plane (border 19) of a level as a numpy copy"""
        bw, bh = C.c_int32(), C.c_int32()
        p = self.lib.oo_pyramid_plane(self.h, level, C.byref(bw), C.byref(bh))
        return np.ctypeslib.as_array(p, shape=(bh.value, bw.value)).copy()

    def resize(self, src, dw, dh):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros((dh, dw), np.uint8)
        self.lib.oo_resize_linear_u8(_ptr(src, _u8p), src.shape[1], src.shape[0], src.shape[1],
                                     _ptr(dst, _u8p), dw, dh, dw)
        return dst

    # ---- FAST ---------------------------------------------------------------
    def fast9_nms(self, img, threshold):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros(max(w * h, 1), CAND_DTYPE)
        n = self.lib.oo_fast9_nms(_ptr(img, _u8p), w, h, w, threshold, out.ctypes.data, out.size)
        return out[:n].copy()

    def fast_score(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8)
        return self.lib.oo_fast_score_bruteforce(img.ctypes.data + y * img.shape[1] + x, img.shape[1])

    def detect_level(self, level, literal):
        h, w = self._img_shape
        lw, lh = self.level_size(w, h, level)
        cap = max((lw * lh) // 4 + 64, 64)
        out = np.zeros(cap, CAND_DTYPE)
        n = self.lib.oo_detect_level(self.h, level, int(literal), out.ctypes.data, cap)
        assert 0 <= n <= cap
        return out[:n].copy()

    def distribute_octree(self, cands, width, height, N, literal):
        cands = np.ascontiguousarray(cands, CAND_DTYPE)
        cap = max(N + 16, 4 * (width // max(height, 1) + 2) + 16)
        out = np.zeros(cap, np.int32)
        n = self.lib.oo_distribute_octree(cands.ctypes.data, len(cands), width, height, N, int(literal),
                                          _ptr(out, _i32p), cap)
        if n < 0:
            raise ValueError("oo_distribute_octree error %d" % n)
        assert n <= cap
        return out[:n].copy()

    # ---- orientation / blur / descriptor ------------------------------------
    def ic_moments(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8)
        um = self.tables()["umax"]
        m01, m10 = C.c_int32(), C.c_int32()
        self.lib.oo_ic_moments(img.ctypes.data + y * img.shape[1] + x, img.shape[1], _ptr(um, _i32p),
                               C.byref(m01), C.byref(m10))
        return m01.value, m10.value

    def fast_atan2(self, y, x):
        return self.lib.oo_fast_atan2(y, x)

    def blur_bordered(self, plane, border):
        """blur the interior of a bordered plane (border >= 3 on every side)"""
        plane = np.ascontiguousarray(plane, np.uint8)
        bh, bw = plane.shape
        w, h = bw - 2 * border, bh - 2 * border
        dst = np.zeros((h, w), np.uint8)
        self.lib.oo_gaussian_blur7(plane.ctypes.data + border * bw + border, w, h, bw, _ptr(dst, _u8p), w)
        return dst

    def descriptor(self, img, x, y, angle_deg):
        img = np.ascontiguousarray(img, np.uint8)
        d = np.zeros(32, np.uint8)
        self.lib.oo_orb_descriptor_mode(img.ctypes.data + y * img.shape[1] + x, img.shape[1], angle_deg, _ptr(d, _u8p),
                                        self.trig)
        return d

    def count_trig_differences(self, first=0, step=1):
        """(cos differences, sin differences, angles tested) between libm cosf / sinf and the correctly
        rounded values over the float angles first, first + step, ... in [0, 360) degrees"""
        counts = (C.c_uint64 * 2)()
        tested = C.c_uint64()
        self.lib.oo_count_trig_differences(first, step, counts, C.byref(tested))
        return int(counts[0]), int(counts[1]), int(tested.value)

    # ---- full extraction ------------------------------------------------------
    def extract(self, img, literal=False, cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        if cap is None:
            cap = int(self.tables()["nfeat"].sum()) + 64 * self.nlevels
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.lib.oo_extract(self.h, _ptr(img, _u8p), w, h, w, kps.ctypes.data, _ptr(desc, _u8p), cap, int(literal))
        if n < 0:
            raise ValueError("oo_extract failed (%d)" % n)
        self._img_shape = (h, w)
        return kps[:n].copy(), desc[:n].copy()

    def last_level_counts(self):
        c = np.zeros(self.nlevels, np.int32)
        self.lib.oo_last_level_counts(self.h, _ptr(c, _i32p))
        return c

    # ---- matching ---------------------------------------------------------------
    def hamming(self, a, b):
        a = np.ascontiguousarray(a).view(np.uint32)
        b = np.ascontiguousarray(b).view(np.uint32)
        return self.lib.oo_hamming(_ptr(a, _u32p), _ptr(b, _u32p))

    def match_bruteforce(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        Q, T = len(q), len(t)
        bi, bd, sd = (np.zeros(max(Q, 1), np.int32) for _ in range(3))
        self.lib.oo_match_bruteforce(_ptr(q.view(np.uint32), _u32p), Q, _ptr(t.view(np.uint32), _u32p), T,
                                     _ptr(bi, _i32p), _ptr(bd, _i32p), _ptr(sd, _i32p))
        return bi[:Q], bd[:Q], sd[:Q]

    def match_windowed(self, q, t, cand_off, cand_idx, t_octave=None):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        cand_off = np.ascontiguousarray(cand_off, np.int32)
        cand_idx = np.ascontiguousarray(cand_idx, np.int32)
        Q = len(q)
        outs = [np.zeros(max(Q, 1), np.int32) for _ in range(5)]
        toct = None if t_octave is None else np.ascontiguousarray(t_octave, np.int32)
        self.lib.oo_match_windowed(_ptr(q.view(np.uint32), _u32p), Q, _ptr(t.view(np.uint32), _u32p), len(t),
                                   _ptr(cand_off, _i32p), _ptr(cand_idx, _i32p),
                                   None if toct is None else _ptr(toct, _i32p),
                                   *[_ptr(o, _i32p) for o in outs])
        return tuple(o[:Q] for o in outs)

    def hamming_csr(self, q, t, cand_off, cand_idx):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        cand_off = np.ascontiguousarray(cand_off, np.int32)
        cand_idx = np.ascontiguousarray(cand_idx, np.int32)
        out = np.zeros(max(len(cand_idx), 1), np.uint16)
        self.lib.oo_hamming_csr(_ptr(q.view(np.uint32), _u32p), len(q), _ptr(t.view(np.uint32), _u32p), len(t),
                                _ptr(cand_off, _i32p), _ptr(cand_idx, _i32p), _ptr(out, C.POINTER(C.c_uint16)))
        return out[:len(cand_idx)]

    def stereo_search(self, kl, dl, kr, dr, scale_factors, nrows, minD, maxD, minX=0.0):
        kl = np.ascontiguousarray(kl, KP_DTYPE)
        kr = np.ascontiguousarray(kr, KP_DTYPE)
        dl = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32)
        dr = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        bi = np.zeros(max(len(kl), 1), np.int32)
        bd = np.zeros(max(len(kl), 1), np.int32)
        self.lib.oo_stereo_search(kl.ctypes.data, _ptr(dl.view(np.uint32), _u32p), len(kl),
                                  kr.ctypes.data, _ptr(dr.view(np.uint32), _u32p), len(kr),
                                  _ptr(sf, _f32p), nrows, minD, maxD, minX, _ptr(bi, _i32p), _ptr(bd, _i32p))
        return bi[:len(kl)], bd[:len(kl)]

    def stereo_finish(self, kl, kr, best_idx, best_dist, mbf, minD, maxD):
        kl = np.ascontiguousarray(kl, KP_DTYPE)
        kr = np.ascontiguousarray(kr, KP_DTYPE)
        bi = np.ascontiguousarray(best_idx, np.int32)
        bd = np.ascontiguousarray(best_dist, np.int32)
        ur = np.zeros(max(len(kl), 1), np.float32)
        dp = np.zeros(max(len(kl), 1), np.float32)
        n = self.lib.oo_stereo_finish(kl.ctypes.data, len(kl), kr.ctypes.data, _ptr(bi, _i32p), _ptr(bd, _i32p),
                                      mbf, minD, maxD, _ptr(ur, _f32p), _ptr(dp, _f32p))
        return n, ur[:len(kl)], dp[:len(kl)]

    def stereo_frame(self, kl, dl, kr, dr, scale_factors, nrows, mbf, mb, min_x=0.0):
        return OracleStereoFrame(self, kl, dl, kr, dr, scale_factors, nrows, mbf, mb, min_x)

    def stereo_classic(self, right_oracle, kl, dl, kr, dr, mbf, mb):
        """self / right_oracle hold the pyramids of the left / right frame"""
        kl = np.ascontiguousarray(kl, KP_DTYPE)
        kr = np.ascontiguousarray(kr, KP_DTYPE)
        dl = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32)
        dr = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
        n = max(len(kl), 1)
        bi, bd = np.zeros(n, np.int32), np.zeros(n, np.int32)
        ur, dp = np.zeros(n, np.float32), np.zeros(n, np.float32)
        nm = self.lib.oo_stereo_classic(self.h, right_oracle.h, kl.ctypes.data, _ptr(dl.view(np.uint32), _u32p), len(kl),
                                        kr.ctypes.data, _ptr(dr.view(np.uint32), _u32p), len(kr), mbf, mb,
                                        _ptr(bi, _i32p), _ptr(bd, _i32p), _ptr(ur, _f32p), _ptr(dp, _f32p))
        return bi[:len(kl)], bd[:len(kl)], ur[:len(kl)], dp[:len(kl)], nm

    def features_in_area(self, kps, minX, minY, invW, invH, qx, qy, qr, qmin, qmax):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (qx, qy, qr))
        qmin, qmax = (np.ascontiguousarray(a, np.int32) for a in (qmin, qmax))
        Q = len(qx)
        off = np.zeros(Q + 1, np.int32)
        args = (kps.ctypes.data, len(kps), minX, minY, invW, invH, _ptr(qx, _f32p), _ptr(qy, _f32p), _ptr(qr, _f32p),
                _ptr(qmin, _i32p), _ptr(qmax, _i32p), Q, _ptr(off, _i32p))
        total = self.lib.oo_features_in_area(*args, None, 0)
        idx = np.zeros(max(total, 1), np.int32)
        self.lib.oo_features_in_area(*args, _ptr(idx, _i32p), total)
        return off, idx[:total]

    def three_maxima(self, sizes):
        sizes = np.ascontiguousarray(sizes, np.int32)
        i1, i2, i3 = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        self.lib.oo_three_maxima(_ptr(sizes, _i32p), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
        return i1.value, i2.value, i3.value

    def search_for_initialization(self, k1, d1, k2, d2, grid, prev_matched, window, nnratio=0.9, check_ori=True):
        """ORBmatcher::SearchForInitialization; prev_matched (n1 x 2 float32) is updated in place.
        Returns (nmatches, vnMatches12)."""
        k1 = np.ascontiguousarray(k1, KP_DTYPE)
        k2 = np.ascontiguousarray(k2, KP_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        assert prev_matched.dtype == np.float32 and prev_matched.flags.c_contiguous and prev_matched.shape == (len(k1), 2)
        m12 = np.full(max(len(k1), 1), -1, np.int32)
        nm = self.lib.oo_search_for_initialization(k1.ctypes.data, _ptr(d1.view(np.uint32), _u32p), len(k1),
                                                   k2.ctypes.data, _ptr(d2.view(np.uint32), _u32p), len(k2),
                                                   grid[0], grid[1], grid[2], grid[3], _ptr(prev_matched, _f32p),
                                                   int(window), nnratio, int(bool(check_ori)), _ptr(m12, _i32p))
        return nm, m12[:len(k1)]

    def search_by_projection(self, kps, desc, u_right, grid, mp, th, nnratio, frame_obs, frame_match, match_score):
        """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th); the three frame arrays are
        updated in place.  Returns nmatches."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        sf = self.tables()["scale"]
        inview = np.ascontiguousarray(mp["mbTrackInView"], np.uint8)
        bad = np.ascontiguousarray(mp["isBad"], np.uint8)
        px, py, pxr, vc = (np.ascontiguousarray(mp[k], np.float32) for k in
                           ("mTrackProjX", "mTrackProjY", "mTrackProjXR", "mTrackViewCos"))
        lvl = np.ascontiguousarray(mp["mnTrackScaleLevel"], np.int32)
        obs = np.ascontiguousarray(mp["Observations"], np.int32)
        md = np.ascontiguousarray(mp["descriptor"], np.uint8).reshape(-1, 32)
        for a in (frame_obs, frame_match, match_score):
            assert a.dtype == np.int32 and a.flags.c_contiguous and len(a) >= len(kps)
        return self.lib.oo_search_by_projection(
            kps.ctypes.data, _ptr(desc.view(np.uint32), _u32p), None if ur is None else _ptr(ur, _f32p), len(kps),
            _ptr(sf, _f32p), grid[0], grid[1], grid[2], grid[3], _ptr(inview, _u8p), _ptr(bad, _u8p), _ptr(px, _f32p),
            _ptr(py, _f32p), _ptr(pxr, _f32p), _ptr(lvl, _i32p), _ptr(vc, _f32p), _ptr(md.view(np.uint32), _u32p),
            _ptr(obs, _i32p), len(inview), th, nnratio, _ptr(frame_obs, _i32p), _ptr(frame_match, _i32p),
            _ptr(match_score, _i32p))

    def search_by_projection_budget(self, kps, desc, u_right, grid, mp, th, nnratio, time_constr, frame_obs, frame_match,
                                    match_score):
        """ORBmatcher::SearchByProjection_Budget; returns (nmatches, stopped_at)"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        sf = self.tables()["scale"]
        inview = np.ascontiguousarray(mp["mbTrackInView"], np.uint8)
        bad = np.ascontiguousarray(mp["isBad"], np.uint8)
        px, py, pxr, vc = (np.ascontiguousarray(mp[k], np.float32) for k in
                           ("mTrackProjX", "mTrackProjY", "mTrackProjXR", "mTrackViewCos"))
        lvl = np.ascontiguousarray(mp["mnTrackScaleLevel"], np.int32)
        obs = np.ascontiguousarray(mp["Observations"], np.int32)
        md = np.ascontiguousarray(mp["descriptor"], np.uint8).reshape(-1, 32)
        stop = np.zeros(1, np.int32)
        nm = self.lib.oo_search_by_projection_budget(
            kps.ctypes.data, _ptr(desc.view(np.uint32), _u32p), None if ur is None else _ptr(ur, _f32p), len(kps),
            _ptr(sf, _f32p), grid[0], grid[1], grid[2], grid[3], _ptr(inview, _u8p), _ptr(bad, _u8p), _ptr(px, _f32p),
            _ptr(py, _f32p), _ptr(pxr, _f32p), _ptr(lvl, _i32p), _ptr(vc, _f32p), _ptr(md.view(np.uint32), _u32p),
            _ptr(obs, _i32p), len(inview), th, nnratio, float(time_constr), _ptr(frame_obs, _i32p), _ptr(frame_match, _i32p),
            _ptr(match_score, _i32p), _ptr(stop, _i32p))
        return nm, int(stop[0])

    def get_candidates(self, kps, grid, mp, i, th):
        """ORBmatcher::GetCandidates for map point i -> pMP->mvMatchCandidates as an int32 array"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        sf = self.tables()["scale"]
        out = np.zeros(max(len(kps), 1), np.int32)
        n = self.lib.oo_get_candidates(kps.ctypes.data, len(kps), _ptr(sf, _f32p), grid[0], grid[1], grid[2], grid[3],
                                       int(mp["mbTrackInView"][i]), int(mp["isBad"][i]), float(mp["mTrackProjX"][i]),
                                       float(mp["mTrackProjY"][i]), int(mp["mnTrackScaleLevel"][i]),
                                       float(mp["mTrackViewCos"][i]), th, _ptr(out, _i32p), len(out))
        return out[:n].copy()

    def match_candidates(self, kps, desc, u_right, cand, mp, i, th, nnratio, frame_obs, frame_match, match_score):
        """ORBmatcher::MatchCandidates for map point i on its stored list -> bestIdx or -1"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        sf = self.tables()["scale"]
        cand = np.ascontiguousarray(cand, np.int32)
        md = np.ascontiguousarray(mp["descriptor"][i], np.uint8).reshape(32)
        return self.lib.oo_match_candidates(
            kps.ctypes.data, _ptr(desc.view(np.uint32), _u32p), None if ur is None else _ptr(ur, _f32p), _ptr(sf, _f32p),
            _ptr(cand, _i32p) if len(cand) else None, len(cand), int(mp["mbTrackInView"][i]), int(mp["isBad"][i]),
            float(mp["mTrackProjXR"][i]), int(mp["mnTrackScaleLevel"][i]), float(mp["mTrackViewCos"][i]),
            _ptr(md.view(np.uint32), _u32p), int(mp["Observations"][i]), int(i), th, nnratio, _ptr(frame_obs, _i32p),
            _ptr(frame_match, _i32p), _ptr(match_score, _i32p))

    def search_by_projection_one_point(self, kps, desc, u_right, grid, mp, i, th, nnratio, frame_obs, frame_match, match_score):
        """ORBmatcher::SearchByProjection_OnePoint for map point i -> bestIdx or -1"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        sf = self.tables()["scale"]
        md = np.ascontiguousarray(mp["descriptor"][i], np.uint8).reshape(32)
        return self.lib.oo_search_by_projection_one_point(
            kps.ctypes.data, _ptr(desc.view(np.uint32), _u32p), None if ur is None else _ptr(ur, _f32p), len(kps),
            _ptr(sf, _f32p), grid[0], grid[1], grid[2], grid[3], int(mp["mbTrackInView"][i]), int(mp["isBad"][i]),
            float(mp["mTrackProjX"][i]), float(mp["mTrackProjY"][i]), float(mp["mTrackProjXR"][i]),
            int(mp["mnTrackScaleLevel"][i]), float(mp["mTrackViewCos"][i]), _ptr(md.view(np.uint32), _u32p),
            int(mp["Observations"][i]), int(i), th, nnratio, _ptr(frame_obs, _i32p), _ptr(frame_match, _i32p),
            _ptr(match_score, _i32p))

    @staticmethod
    def _fv(fv):
        nodes = np.array(sorted(fv), np.int32)
        off = np.zeros(len(nodes) + 1, np.int32)
        for k, n in enumerate(nodes):
            off[k + 1] = off[k] + len(fv[int(n)])
        idx = np.concatenate([np.asarray(fv[int(n)], np.int32) for n in nodes]) if len(nodes) else np.zeros(1, np.int32)
        return nodes if len(nodes) else np.zeros(1, np.int32), off, np.ascontiguousarray(idx, np.int32), len(nodes)

    def search_by_bow_frame(self, kf_kps, kf_desc, kf_has_point, kf_fv, f_kps, f_desc, f_fv, nnratio, check_ori):
        """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) -> (nmatches, matches_f)"""
        k1, k2 = np.ascontiguousarray(kf_kps, KP_DTYPE), np.ascontiguousarray(f_kps, KP_DTYPE)
        d1 = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(f_desc, np.uint8).reshape(-1, 32)
        hp = np.ascontiguousarray(kf_has_point, np.uint8)
        n1, o1, i1, c1 = self._fv(kf_fv)
        n2, o2, i2, c2 = self._fv(f_fv)
        out = np.full(max(len(k2), 1), -1, np.int32)
        nm = self.lib.oo_search_by_bow_frame(k1.ctypes.data, _ptr(d1.view(np.uint32), _u32p), _ptr(hp, _u8p), _ptr(n1, _i32p),
                                             _ptr(o1, _i32p), _ptr(i1, _i32p), c1, k2.ctypes.data,
                                             _ptr(d2.view(np.uint32), _u32p), len(k2), _ptr(n2, _i32p), _ptr(o2, _i32p),
                                             _ptr(i2, _i32p), c2, nnratio, int(check_ori), _ptr(out, _i32p))
        return nm, out[:len(k2)]

    def search_by_bow_keyframes(self, k1, d1, h1, fv1, k2, d2, h2, fv2, nnratio, check_ori):
        """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) -> (nmatches, matches12)"""
        k1, k2 = np.ascontiguousarray(k1, KP_DTYPE), np.ascontiguousarray(k2, KP_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        h1, h2 = np.ascontiguousarray(h1, np.uint8), np.ascontiguousarray(h2, np.uint8)
        n1, o1, i1, c1 = self._fv(fv1)
        n2, o2, i2, c2 = self._fv(fv2)
        out = np.full(max(len(k1), 1), -1, np.int32)
        nm = self.lib.oo_search_by_bow_keyframes(k1.ctypes.data, _ptr(d1.view(np.uint32), _u32p), _ptr(h1, _u8p), len(k1),
                                                 _ptr(n1, _i32p), _ptr(o1, _i32p), _ptr(i1, _i32p), c1, k2.ctypes.data,
                                                 _ptr(d2.view(np.uint32), _u32p), _ptr(h2, _u8p), len(k2), _ptr(n2, _i32p),
                                                 _ptr(o2, _i32p), _ptr(i2, _i32p), c2, nnratio, int(check_ori),
                                                 _ptr(out, _i32p))
        return nm, out[:len(k1)]

    def search_by_projection_keyframe(self, kps, desc, grid, kf, th, orb_dist, check_ori, frame_obs, frame_match):
        """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) from the projected points on;
        kf: dict of per-keypoint arrays of the keyframe (visible, u, v, level, angle, descriptor, Observations)"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        sf = self.tables()["scale"]
        vis = np.ascontiguousarray(kf["visible"], np.uint8)
        u, v, ang = (np.ascontiguousarray(kf[k], np.float32) for k in ("u", "v", "angle"))
        lvl = np.ascontiguousarray(kf["level"], np.int32)
        obs = np.ascontiguousarray(kf["Observations"], np.int32)
        kd = np.ascontiguousarray(kf["descriptor"], np.uint8).reshape(-1, 32)
        return self.lib.oo_search_by_projection_keyframe(
            kps.ctypes.data, _ptr(desc.view(np.uint32), _u32p), len(kps), _ptr(sf, _f32p), grid[0], grid[1], grid[2], grid[3],
            _ptr(vis, _u8p), _ptr(u, _f32p), _ptr(v, _f32p), _ptr(lvl, _i32p), _ptr(ang, _f32p),
            _ptr(kd.view(np.uint32), _u32p), _ptr(obs, _i32p), len(vis), th, int(orb_dist), int(check_ori),
            _ptr(frame_obs, _i32p), _ptr(frame_match, _i32p))

    @staticmethod
    def _pts(pts):
        d = np.ascontiguousarray(pts["descriptor"], np.uint8).reshape(-1, 32)
        return (np.ascontiguousarray(pts["visible"], np.uint8), np.ascontiguousarray(pts["u"], np.float32),
                np.ascontiguousarray(pts["v"], np.float32), np.ascontiguousarray(pts["level"], np.int32), d)

    def search_by_projection_sim3(self, kps, desc, grid, pts, th, kf_matched, kf_match):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        sf = self.tables()["scale"]
        vis, u, v, lvl, pd = self._pts(pts)
        return self.lib.oo_search_by_projection_sim3(
            kps.ctypes.data, _ptr(desc.view(np.uint32), _u32p), len(kps), _ptr(sf, _f32p), grid[0], grid[1], grid[2], grid[3],
            _ptr(vis, _u8p), _ptr(u, _f32p), _ptr(v, _f32p), _ptr(lvl, _i32p), _ptr(pd.view(np.uint32), _u32p), len(vis),
            int(th), _ptr(kf_matched, _i32p), _ptr(kf_match, _i32p))

    def fuse_search(self, kps, desc, u_right, grid, pts, th, gate):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        t = self.tables()
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        vis, u, v, lvl, pd = self._pts(pts)
        pur = None if ur is None else np.ascontiguousarray(pts["ur"], np.float32)
        bi, bd = np.zeros(max(len(vis), 1), np.int32), np.zeros(max(len(vis), 1), np.int32)
        nf = self.lib.oo_fuse_search(
            kps.ctypes.data, _ptr(desc.view(np.uint32), _u32p), None if ur is None else _ptr(ur, _f32p), len(kps),
            _ptr(t["scale"], _f32p), _ptr(t["inv_sigma2"], _f32p), grid[0], grid[1], grid[2], grid[3], _ptr(vis, _u8p),
            _ptr(u, _f32p), _ptr(v, _f32p), None if pur is None else _ptr(pur, _f32p), _ptr(lvl, _i32p),
            _ptr(pd.view(np.uint32), _u32p), len(vis), th, int(gate), _ptr(bi, _i32p), _ptr(bd, _i32p))
        return nf, bi[:len(vis)], bd[:len(vis)]

    def search_by_sim3(self, k1, d1, k2, d2, grid, pts1, pts2, th):
        k1, k2 = np.ascontiguousarray(k1, KP_DTYPE), np.ascontiguousarray(k2, KP_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        sf = self.tables()["scale"]
        a, b = self._pts(pts1), self._pts(pts2)
        out = np.full(max(len(k1), 1), -1, np.int32)

        def args(p):
            return (_ptr(p[0], _u8p), _ptr(p[1], _f32p), _ptr(p[2], _f32p), _ptr(p[3], _i32p), _ptr(p[4].view(np.uint32), _u32p))
        nf = self.lib.oo_search_by_sim3(k1.ctypes.data, _ptr(d1.view(np.uint32), _u32p), len(k1), k2.ctypes.data,
                                        _ptr(d2.view(np.uint32), _u32p), len(k2), _ptr(sf, _f32p), grid[0], grid[1], grid[2],
                                        grid[3], *args(a), *args(b), th, _ptr(out, _i32p))
        return nf, out[:len(k1)]

    def search_for_triangulation(self, k1, d1, h1, s1, fv1, k2, d2, h2, s2, fv2, F12, epipole, only_stereo, check_ori):
        k1, k2 = np.ascontiguousarray(k1, KP_DTYPE), np.ascontiguousarray(k2, KP_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        h1, h2 = np.ascontiguousarray(h1, np.uint8), np.ascontiguousarray(h2, np.uint8)
        s1, s2 = np.ascontiguousarray(s1, np.uint8), np.ascontiguousarray(s2, np.uint8)
        t = self.tables()
        n1, o1, i1, c1 = self._fv(fv1)
        n2, o2, i2, c2 = self._fv(fv2)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        out = np.full(max(len(k1), 1), -1, np.int32)
        nm = self.lib.oo_search_for_triangulation(
            k1.ctypes.data, _ptr(d1.view(np.uint32), _u32p), _ptr(h1, _u8p), _ptr(s1, _u8p), len(k1), _ptr(n1, _i32p),
            _ptr(o1, _i32p), _ptr(i1, _i32p), c1, k2.ctypes.data, _ptr(d2.view(np.uint32), _u32p), _ptr(h2, _u8p),
            _ptr(s2, _u8p), _ptr(n2, _i32p), _ptr(o2, _i32p), _ptr(i2, _i32p), c2, _ptr(t["scale"], _f32p),
            _ptr(t["sigma2"], _f32p), _ptr(F, _f32p), float(epipole[0]), float(epipole[1]), int(only_stereo), int(check_ori),
            _ptr(out, _i32p))
        return nm, out[:len(k1)]

    def search_by_projection_last(self, kps, desc, u_right, grid, last, th, mbf, forward, backward, check_ori,
                                  frame_obs, frame_match):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, ...) from :1496 on; frame arrays updated in place"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        sf = self.tables()["scale"]
        vis = np.ascontiguousarray(last["visible"], np.uint8)
        u, v, iz, ang = (np.ascontiguousarray(last[k], np.float32) for k in ("u", "v", "invzc", "angle"))
        octv = np.ascontiguousarray(last["octave"], np.int32)
        obs = np.ascontiguousarray(last["Observations"], np.int32)
        ld = np.ascontiguousarray(last["descriptor"], np.uint8).reshape(-1, 32)
        for a in (frame_obs, frame_match):
            assert a.dtype == np.int32 and a.flags.c_contiguous and len(a) >= len(kps)
        return self.lib.oo_search_by_projection_last(
            kps.ctypes.data, _ptr(desc.view(np.uint32), _u32p), None if ur is None else _ptr(ur, _f32p), len(kps),
            _ptr(sf, _f32p), grid[0], grid[1], grid[2], grid[3], _ptr(vis, _u8p), _ptr(u, _f32p), _ptr(v, _f32p),
            _ptr(iz, _f32p), _ptr(octv, _i32p), _ptr(ang, _f32p), _ptr(ld.view(np.uint32), _u32p), _ptr(obs, _i32p),
            len(vis), th, mbf, int(bool(forward)), int(bool(backward)), int(bool(check_ori)), _ptr(frame_obs, _i32p),
            _ptr(frame_match, _i32p))

    def undistort_points(self, xy, K, dist=None, R=None, P=None, fisheye=False):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        K = np.ascontiguousarray(K, np.float64)
        d = np.zeros(0, np.float64) if dist is None else np.ascontiguousarray(dist, np.float64).reshape(-1)
        R = None if R is None else np.ascontiguousarray(R, np.float64)
        P = None if P is None else np.ascontiguousarray(P, np.float64)
        out = np.zeros_like(xy)
        self.lib.oo_undistort_points(_ptr(xy, _f32p), len(xy), K.ctypes.data, d.ctypes.data if len(d) else None, len(d),
                                     None if R is None else R.ctypes.data, None if P is None else P.ctypes.data,
                                     0 if P is None else P.shape[1], 1 if fisheye else 0, _ptr(out, _f32p))
        return out


class OracleStereoFrame:
    """State + calls of the reference's delayed stereo matching (include/Frame.h:217-255) on the oracle"""

    def __init__(self, oracle, kl, dl, kr, dr, scale_factors, nrows, mbf, mb, min_x):
        self.o = oracle
        self.kl = np.ascontiguousarray(kl, KP_DTYPE)
        self.kr = np.ascontiguousarray(kr, KP_DTYPE)
        self.dl = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32)
        self.dr = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
        self.sf = np.ascontiguousarray(scale_factors, np.float32)
        self.nrows, self.mbf, self.mb, self.min_x = nrows, mbf, mb, min_x
        n = max(len(self.kl), 1)
        self.N = len(self.kl)
        self.mvuRight = np.zeros(n, np.float32)
        self.mvDepth = np.zeros(n, np.float32)
        self.mvStereoMatched = np.zeros(n, np.uint8)
        self.di_dist = np.zeros(n, np.int32)
        self.di_il = np.zeros(n, np.int32)
        self.has_mappoint = np.zeros(n, np.uint8)
        self.prior_disp = np.zeros(n, np.float32)
        self.best_idx = np.full(n, -1, np.int32)
        self.best_dist = np.full(n, 100, np.int32)
        self.st = StereoState(_ptr(self.mvuRight, _f32p), _ptr(self.mvDepth, _f32p), _ptr(self.mvStereoMatched, _u8p),
                              _ptr(self.di_dist, _i32p), _ptr(self.di_il, _i32p), 0)
        self.PrepareStereoCandidates()

    def PrepareStereoCandidates(self):
        self.o.lib.oo_stereo_prepare(C.byref(self.st), self.N)

    def listed_dist(self):
        """mvDistIdx as an array over left keypoints (-1 = not listed)"""
        out = np.full(max(self.N, 1), -1, np.int32)
        n = self.st.n_dist_idx
        out[self.di_il[:n]] = self.di_dist[:n]
        return out

    def _common(self):
        return (self.kl.ctypes.data, _ptr(self.dl.view(np.uint32), _u32p), self.N,
                self.kr.ctypes.data, _ptr(self.dr.view(np.uint32), _u32p), len(self.kr),
                _ptr(self.sf, _f32p), self.nrows)

    def ComputeStereoMatches_Undistorted(self, isOnline=False, disparity_thres=50.0):
        return self.o.lib.oo_stereo_undistorted(*self._common(), self.mbf, self.mb, self.min_x, disparity_thres,
                                                _ptr(self.has_mappoint, _u8p), _ptr(self.prior_disp, _f32p),
                                                int(isOnline), C.byref(self.st), _ptr(self.best_idx, _i32p),
                                                _ptr(self.best_dist, _i32p))

    def ComputeStereoMatch_OnePoint(self, iL, thOrbDist, minD, maxD):
        return bool(self.o.lib.oo_stereo_one_point(*self._common(), self.mbf, _ptr(self.has_mappoint, _u8p), iL,
                                                   thOrbDist, minD, maxD, C.byref(self.st), _ptr(self.best_idx, _i32p),
                                                   _ptr(self.best_dist, _i32p)))
