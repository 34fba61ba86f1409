"""Host-side mirror of the per-keypoint maps ORB_SLAM2::Frame runs right after extraction."""
import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, ptr


def _mat(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError("expected a %s matrix, got %s" % (shape, a.shape))
    return a


class CameraModel:
    """K, distortion coefficients and the optional rectification (R, P) of one camera, as
    Frame keeps them (mK / mK_ori, mDistCoef, mR, mP: src/Frame.cc:745-852)."""

    def __init__(self, K, dist=None, R=None, P=None, fisheye=False):
        self.K = _mat(K, (3, 3))
        self.dist = np.zeros(0, np.float64) if dist is None else np.ascontiguousarray(dist, np.float64).reshape(-1)
        self.R = _mat(R, (3, 3))
        self.P = _mat(P)
        if self.P is not None and self.P.shape not in ((3, 3), (3, 4)):
            raise ValueError("P must be 3x3 or 3x4")
        self.fisheye = bool(fisheye)

    def _args(self):
        d = self.dist
        return (self.K.ctypes.data, d.ctypes.data if len(d) else None, len(d),
                None if self.R is None else self.R.ctypes.data,
                None if self.P is None else self.P.ctypes.data,
                0 if self.P is None else self.P.shape[1], 1 if self.fisheye else 0)


class Frame:
    """The keypoint maps of ORB_SLAM2::Frame that follow extraction.  `extractor` supplies the
    device context."""

    def __init__(self, extractor):
        self._lib = _lib.load_library()
        self._ctx = extractor._ctx

    def UndistortKeyPoints(self, kps, cam):
        """Frame::UndistortKeyPoints (src/Frame.cc:818-852); with cam.R / cam.P set it is one half
        of UndistortKeyPointsStereo (:745-816).  Like the reference, a first distortion coefficient
        of 0 returns the keypoints unchanged (:747, :820)."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        if len(cam.dist) == 0 or cam.dist[0] == 0.0:
            return kps.copy()
        out = np.empty_like(kps)
        check(self._lib, self._ctx, self._lib.orbx_undistort_keypoints(
            self._ctx, kps.ctypes.data if len(kps) else None, len(kps), *cam._args(),
            out.ctypes.data if len(kps) else None))
        return out

    def undistort_keypoints_device(self, d_kps_in, d_n, n, cap, nframes, cam, d_kps_out):
        """Batched, device-resident form on the buffers of extract_batch_device (no shortcut)."""
        check(self._lib, self._ctx, self._lib.orbx_undistort_keypoints_device(
            self._ctx, ptr(d_kps_in), ptr(d_n), n, cap, nframes, *cam._args(), ptr(d_kps_out)))
