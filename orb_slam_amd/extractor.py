"""Host-side mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:71-176)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, OrbxError, check, ptr


class ORBextractor:
    """Same constructor arguments, call semantics and getters as the reference
    class; results come from the gfx950 kernels behind liborbx.so.

    extractor(image, mask=None) -> (keypoints, descriptors) replaces
    operator()(image, mask, keypoints, descriptors): `keypoints` is a
    structured array with the fields of cv::KeyPoint, `descriptors` an (n, 32)
    uint8 array (None when n == 0, as the reference releases the matrix).
    """
    HARRIS_SCORE = 0
    FAST_SCORE = 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0):
        self._lib = _lib.load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.orbx_create(C.byref(self._ctx), int(nfeatures), float(scaleFactor), int(nlevels),
                                   int(iniThFAST), int(minThFAST), int(device))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise OrbxError(rc, "orbx_create failed (needs a gfx950 device; there is no CPU path)")
        self.nfeatures, self.scaleFactor, self.nlevels = int(nfeatures), float(scaleFactor), int(nlevels)
        self.iniThFAST, self.minThFAST = int(iniThFAST), int(minThFAST)
        self._shape = None
        self._nframes = 0

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.orbx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters, include/ORBextractor.h:92-130 --------------------------------
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetInitThres(self):
        return self.iniThFAST

    def GetMinThres(self):
        return self.minThFAST

    def _tables(self):
        n = self.nlevels
        arrs = [np.zeros(n, np.float32) for _ in range(4)]
        nf = np.zeros(n, np.int32)
        check(self._lib, self._ctx, self._lib.orbx_get_tables(
            self._ctx, *[a.ctypes.data_as(C.POINTER(C.c_float)) for a in arrs],
            nf.ctypes.data_as(C.POINTER(C.c_int32))))
        return arrs + [nf]

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def GetFeaturesPerLevel(self):
        return self._tables()[4]

    def set_stream(self, hip_stream):
        check(self._lib, self._ctx, self._lib.orbx_set_stream(self._ctx, hip_stream))

    def max_keypoints(self, w, h):
        n = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_max_keypoints(self._ctx, int(w), int(h), C.byref(n)))
        return n.value

    def level_size(self, w, h, level):
        lw, lh = C.c_int32(), C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_level_size(self._ctx, w, h, level, C.byref(lw), C.byref(lh)))
        return lw.value, lh.value

    # ---- operator() --------------------------------------------------------------
    def __call__(self, image, mask=None):
        """ORBextractor::operator() (src/ORBextractor.cc:1127-1190); mask is ignored."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), None
        if image.dtype != np.uint8 or image.ndim != 2:
            raise AssertionError("image.type() == CV_8UC1")   # the reference asserts (:1134)
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_extract(self._ctx, image.ctypes.data, w, h, image.strides[0],
                                                           kps.ctypes.data, desc.ctypes.data, cap, C.byref(n)))
        self._shape, self._nframes = (h, w), 1
        if n.value == 0:
            return kps[:0], None
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """images: (B, H, W) uint8 host array -> list of (keypoints, descriptors)"""
        images = np.ascontiguousarray(images, np.uint8)
        b, h, w = images.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros((b, cap), KP_DTYPE)
        desc = np.zeros((b, cap, 32), np.uint8)
        n = np.zeros(b, np.int32)
        ptrs = (C.c_void_p * b)(*[images[i].ctypes.data for i in range(b)])
        check(self._lib, self._ctx, self._lib.orbx_extract_batch(self._ctx, ptrs, b, w, h, w, kps.ctypes.data,
                                                                 desc.ctypes.data, cap, n.ctypes.data))
        self._shape, self._nframes = (h, w), b
        return [(kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(b)]

    def extract_batch_device(self, d_imgs, nframes, w, h, stride, frame_stride, d_kps, d_desc, cap, d_n_out):
        """Hot entry: device pointers (ints or torch tensors); asynchronous on the context's stream."""
        check(self._lib, self._ctx, self._lib.orbx_extract_batch_device(
            self._ctx, ptr(d_imgs), nframes, w, h, stride, frame_stride, ptr(d_kps), ptr(d_desc), cap, ptr(d_n_out)))
        self._shape, self._nframes = (h, w), nframes

    # ---- pyramid -------------------------------------------------------------------
    def ComputePyramid(self, image):
        """ORBextractor::ComputePyramid (src/ORBextractor.cc:1192-1216)"""
        image = np.ascontiguousarray(image, np.uint8)
        h, w = image.shape
        check(self._lib, self._ctx, self._lib.orbx_compute_pyramid(self._ctx, image.ctypes.data, w, h, w))
        self._shape, self._nframes = (h, w), 1

    def pyramid_level(self, level, frame=0, border=0):
        """mvImagePyramid[level] of `frame` as a numpy copy; border <= 19 adds the stored REFLECT_101 margin"""
        data = C.c_void_p()
        w, h, s = C.c_int32(), C.c_int32(), C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_pyramid_level(self._ctx, frame, level, C.byref(data),
                                                                 C.byref(w), C.byref(h), C.byref(s)))
        assert 0 <= border <= 19
        rows, cols, stride = h.value + 2 * border, w.value + 2 * border, s.value
        start = data.value - border * stride - border
        buf = (C.c_uint8 * (rows * stride)).from_address(start)
        a = np.frombuffer(buf, np.uint8).reshape(rows, stride)[:, :cols]
        return a.copy()

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    # ---- inspection (parity tests) ----------------------------------------------------
    def debug_candidates(self, level, frame=0):
        h, w = self._shape
        lw, lh = self.level_size(w, h, level)
        cap = max(lw * lh // 4 + 64, 64)
        xyr = np.zeros((cap, 3), np.int32)
        n = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_debug_candidates(self._ctx, frame, level, xyr.ctypes.data, cap, C.byref(n)))
        return xyr[:n.value].copy()

    def debug_level_counts(self, frame=0):
        c = np.zeros(self.nlevels, np.int32)
        check(self._lib, self._ctx, self._lib.orbx_debug_level_counts(self._ctx, frame, c.ctypes.data))
        return c

    def debug_blurred_level(self, level, frame=0):
        h, w = self._shape
        lw, lh = self.level_size(w, h, level)
        out = np.zeros((lh, lw), np.uint8)
        check(self._lib, self._ctx, self._lib.orbx_debug_blurred_level(self._ctx, frame, level, out.ctypes.data, out.size))
        return out

    def stage_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        ln = (C.c_int32 * 16)()
        n = self._lib.orbx_stage_times(self._ctx, names, ms, ln, 16)
        if n < 0:
            raise OrbxError(n)
        return {names[i].decode(): (ms[i], ln[i]) for i in range(n)}

    def reset_stage_times(self):
        check(self._lib, self._ctx, self._lib.orbx_reset_stage_times(self._ctx))

    def set_stage_timing(self, enable):
        check(self._lib, self._ctx, self._lib.orbx_set_stage_timing(self._ctx, int(enable)))   # 2: only the FAST launch
