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


DISPARITY_THRES = 50.0    # include/Frame.h:47


class StereoFrame:
    """The delayed stereo matching of ORB_SLAM2::Frame in the reference's default build
    (DELAYED_STEREO_MATCHING, include/Frame.h:38-40): the members PrepareStereoCandidates sets up
    and the calls that fill them in several steps.  The searches run on the device through
    orbx_stereo_match_ex; this class keeps the Frame's state and derives, per call, which left
    keypoints take part and with which disparity window, as the reference does on the host.

    mvpMapPoints is represented by two arrays the tracker fills: has_mappoint[iL] (a map point is
    associated) and prior_disp[iL] = mbf / Pc.z for a map point that is not bad and lies in front of
    the camera (src/Frame.cc:1325-1334), <= 0 otherwise."""

    def __init__(self, matcher, kl, dl, kr, dr, nrows, mbf, mb, min_x=0.0):
        self._m = matcher
        self.mvKeysUn = np.ascontiguousarray(kl, KP_DTYPE)
        self.mvKeysRightUn = np.ascontiguousarray(kr, KP_DTYPE)
        self.mDescriptors = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32)
        self.mDescriptorsRight = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
        self.N = len(self.mvKeysUn)
        self.nrows, self.mbf, self.mb, self.mnMinX = int(nrows), float(mbf), float(mb), float(min_x)
        self.has_mappoint = np.zeros(self.N, np.uint8)
        self.prior_disp = np.zeros(self.N, np.float32)
        self.PrepareStereoCandidates()

    def PrepareStereoCandidates(self):
        """include/Frame.h:217-251 (the row table itself is implicit in the device search)"""
        n = max(self.N, 1)
        self.mvuRight = np.full(n, -1.0, np.float32)
        self.mvDepth = np.full(n, -1.0, np.float32)
        self.mvStereoMatched = np.zeros(n, bool)
        self.listed_dist = np.full(n, -1, np.int32)          # mvDistIdx, indexed by left keypoint
        self.best_idx = np.full(n, -1, np.int32)
        self.best_dist = np.full(n, 100, np.int32)

    def disparity_windows(self):
        """[minD, maxD) per left keypoint, src/Frame.cc:1302-1303 and :1325-1338"""
        full = np.float32(self.mbf) / np.float32(self.mb)
        min_d = np.zeros(max(self.N, 1), np.float32)
        max_d = np.full(max(self.N, 1), full, np.float32)
        use = (self.has_mappoint[:self.N] != 0) & (self.prior_disp[:self.N] > 0)
        disp = self.prior_disp[:self.N][use]
        th = np.float32(DISPARITY_THRES)
        min_d[:self.N][use] = np.maximum(disp - th, np.float32(0))
        max_d[:self.N][use] = np.minimum(disp + th, full)
        return min_d, max_d

    def ComputeStereoMatches_Undistorted(self, isOnline=False):
        """src/Frame.cc:1262-1422; returns nmatched"""
        if self.N == 0:
            return 0
        if isOnline:      # :1283-1290
            active = (self.has_mappoint[:self.N] != 0) & ~self.mvStereoMatched[:self.N]
        else:             # :1291-1298
            active = ~self.mvStereoMatched[:self.N]
        min_d, max_d = self.disparity_windows()
        self.mvStereoMatched[:self.N] |= active      # :1304
        return self._m.stereo_match_ex(self.mvKeysUn, self.mDescriptors, self.mvKeysRightUn, self.mDescriptorsRight,
                                       self.nrows, self.mbf, self.mb, self.mnMinX, active.astype(np.uint8), min_d, max_d,
                                       75, not isOnline, self.listed_dist, self.best_idx, self.best_dist,
                                       self.mvuRight, self.mvDepth)

    def ComputeStereoMatch_OnePoint(self, iL, thOrbDist, minD, maxD):
        """src/Frame.cc:1175-1260"""
        if iL < 0 or iL >= self.N:
            return False
        if not self.has_mappoint[iL] or self.mvStereoMatched[iL]:
            return False
        self.mvStereoMatched[iL] = True
        active = np.zeros(self.N, np.uint8)
        active[iL] = 1
        min_d = np.full(self.N, minD, np.float32)
        max_d = np.full(self.N, maxD, np.float32)
        self._m.stereo_match_ex(self.mvKeysUn, self.mDescriptors, self.mvKeysRightUn, self.mDescriptorsRight,
                                self.nrows, self.mbf, self.mb, 0.0, active, min_d, max_d, int(thOrbDist), False,
                                self.listed_dist, self.best_idx, self.best_dist, self.mvuRight, self.mvDepth)
        return bool(self.listed_dist[iL] >= 0)

    def ComputeStereoMatch_Points(self, iLs, thOrbDist, minD, maxD):
        """Several ComputeStereoMatch_OnePoint calls (src/Frame.cc:1175-1260) in one device call: the reference
        calls the one-point form once per tracked map point, and a whole-frame launch per point costs N times the
        work it needs.  iLs: left keypoint indices; minD / maxD: one disparity window per index (or scalars).  A
        one-point call changes nothing another one reads, so the batch equals the calls made one after the other;
        returns the array of their return values."""
        iLs = np.asarray(iLs, np.int64).reshape(-1)
        minD = np.broadcast_to(np.asarray(minD, np.float32), iLs.shape)
        maxD = np.broadcast_to(np.asarray(maxD, np.float32), iLs.shape)
        res = np.zeros(len(iLs), bool)
        if self.N == 0 or len(iLs) == 0:
            return res
        active = np.zeros(self.N, np.uint8)
        min_d = np.zeros(self.N, np.float32)
        max_d = np.zeros(self.N, np.float32)
        taken = []
        for k, iL in enumerate(iLs):
            if iL < 0 or iL >= self.N or not self.has_mappoint[iL] or self.mvStereoMatched[iL]:
                continue                              # :1182-1186 (a second mention of one index finds it matched already)
            self.mvStereoMatched[iL] = True
            active[iL] = 1
            min_d[iL], max_d[iL] = minD[k], maxD[k]
            taken.append(k)
        if taken:
            self._m.stereo_match_ex(self.mvKeysUn, self.mDescriptors, self.mvKeysRightUn, self.mDescriptorsRight,
                                    self.nrows, self.mbf, self.mb, 0.0, active, min_d, max_d, int(thOrbDist), False,
                                    self.listed_dist, self.best_idx, self.best_dist, self.mvuRight, self.mvDepth)
            for k in taken:
                res[k] = self.listed_dist[iLs[k]] >= 0
        return res

