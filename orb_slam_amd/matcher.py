"""Host-side mirror of the Hamming searches of ORB_SLAM2::ORBmatcher."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, ptr


def _desc(a):
    a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
    return a


class ORBmatcher:
    """DescriptorDistance and the best / second-best scans written around it
    (src/ORBmatcher.cc).  `extractor` supplies the device context."""
    TH_LOW = 50     # src/ORBmatcher.cc:37-39
    TH_HIGH = 100
    HISTO_LENGTH = 30

    def __init__(self, extractor, nnratio=0.6, checkOri=True):
        self._lib = _lib.load_library()
        self._ex = extractor
        self._ctx = extractor._ctx
        self.mfNNratio = nnratio
        self.mbCheckOrientation = checkOri

    @staticmethod
    def DescriptorDistance(a, b):
        """ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1768-1784)"""
        lib = _lib.load_library()
        a = np.ascontiguousarray(a, np.uint8).reshape(32)
        b = np.ascontiguousarray(b, np.uint8).reshape(32)
        return lib.orbx_hamming(a.ctypes.data, b.ctypes.data)

    def match_bruteforce(self, q, t):
        q, t = _desc(q), _desc(t)
        Q, T = len(q), len(t)
        outs = [np.zeros(max(Q, 1), np.int32) for _ in range(3)]
        check(self._lib, self._ctx, self._lib.orbx_match_bruteforce(
            self._ctx, q.ctypes.data, Q, t.ctypes.data if T else None, T, *[o.ctypes.data for o in outs]))
        return tuple(o[:Q] for o in outs)

    def match_bruteforce_device(self, d_q, Q, q_stride, d_nq, d_t, T, t_stride, d_nt, nbatch, d_bi, d_bd, d_sd):
        check(self._lib, self._ctx, self._lib.orbx_match_bruteforce_device(
            self._ctx, ptr(d_q), Q, q_stride, ptr(d_nq), ptr(d_t), T, t_stride, ptr(d_nt), nbatch,
            ptr(d_bi), ptr(d_bd), ptr(d_sd)))

    def match_bruteforce_device_static(self, d_q, Q, q_stride, d_nq, d_t, T, t_stride, d_nt, nbatch, d_bi, d_bd, d_sd):
        """the same search on the library's static device scratch: allocates nothing"""
        check(self._lib, self._ctx, self._lib.orbx_match_bruteforce_device_static(
            self._ctx, ptr(d_q), Q, q_stride, ptr(d_nq), ptr(d_t), T, t_stride, ptr(d_nt), nbatch,
            ptr(d_bi), ptr(d_bd), ptr(d_sd)))

    def match_windowed(self, q, t, cand_off, cand_idx, t_octave=None):
        q, t = _desc(q), _desc(t)
        Q, T = len(q), len(t)
        cand_off = np.ascontiguousarray(cand_off, np.int32)
        cand_idx = np.ascontiguousarray(cand_idx, np.int32)
        toct = None if t_octave is None else np.ascontiguousarray(t_octave, np.int32)
        outs = [np.zeros(max(Q, 1), np.int32) for _ in range(5)]
        check(self._lib, self._ctx, self._lib.orbx_match_windowed(
            self._ctx, q.ctypes.data, Q, t.ctypes.data if T else None, T, cand_off.ctypes.data,
            cand_idx.ctypes.data if len(cand_idx) else None, None if toct is None else toct.ctypes.data,
            *[o.ctypes.data for o in outs]))
        return tuple(o[:Q] for o in outs)

    def hamming_csr(self, q, t, cand_off, cand_idx):
        q, t = _desc(q), _desc(t)
        cand_off = np.ascontiguousarray(cand_off, np.int32)
        cand_idx = np.ascontiguousarray(cand_idx, np.int32)
        out = np.zeros(max(len(cand_idx), 1), np.uint16)
        check(self._lib, self._ctx, self._lib.orbx_hamming_csr(
            self._ctx, q.ctypes.data, len(q), t.ctypes.data if len(t) else None, len(t), cand_off.ctypes.data,
            cand_idx.ctypes.data if len(cand_idx) else None, out.ctypes.data))
        return out[:len(cand_idx)]

    def stereo_match(self, kl, dl, kr, dr, nrows, mbf, mb, min_x=0.0):
        """PrepareStereoCandidates + ComputeStereoMatches_Undistorted(false)
        (include/Frame.h:217-251, src/Frame.cc:1262-1422)"""
        kl = np.ascontiguousarray(kl, KP_DTYPE)
        kr = np.ascontiguousarray(kr, KP_DTYPE)
        dl, dr = _desc(dl), _desc(dr)
        nl, nr = len(kl), len(kr)
        bi = np.zeros(max(nl, 1), np.int32)
        bd = np.zeros(max(nl, 1), np.int32)
        ur = np.zeros(max(nl, 1), np.float32)
        dp = np.zeros(max(nl, 1), np.float32)
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_stereo_match(
            self._ctx, kl.ctypes.data, dl.ctypes.data, nl, kr.ctypes.data if nr else None,
            dr.ctypes.data if nr else None, nr, nrows, mbf, mb, min_x,
            bi.ctypes.data, bd.ctypes.data, ur.ctypes.data, dp.ctypes.data, C.byref(nm)))
        return bi[:nl], bd[:nl], ur[:nl], dp[:nl], nm.value

    def stereo_match_device(self, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, nbatch, stride, nrows, mbf, mb, min_x,
                            d_bi, d_bd, d_ur, d_dp, d_nm):
        check(self._lib, self._ctx, self._lib.orbx_stereo_match_device(
            self._ctx, ptr(d_kl), ptr(d_dl), ptr(d_nl), ptr(d_kr), ptr(d_dr), ptr(d_nr), nbatch, stride, nrows,
            mbf, mb, min_x, ptr(d_bi), ptr(d_bd), ptr(d_ur), ptr(d_dp), ptr(d_nm)))

    def stereo_match_ex(self, kl, dl, kr, dr, nrows, mbf, mb, min_x, active, min_d, max_d, th_orb_dist, median_cut,
                        listed_dist, best_idx, best_dist, u_right, depth):
        """orbx_stereo_match_ex: one call of the delayed stereo matching on a Frame's state arrays
        (updated in place); returns nmatched"""
        kl = np.ascontiguousarray(kl, KP_DTYPE)
        kr = np.ascontiguousarray(kr, KP_DTYPE)
        dl, dr = _desc(dl), _desc(dr)
        nl, nr = len(kl), len(kr)
        for a, t in ((listed_dist, np.int32), (best_idx, np.int32), (best_dist, np.int32),
                     (u_right, np.float32), (depth, np.float32)):
            if a.dtype != t or not a.flags.c_contiguous or len(a) < nl:
                raise ValueError("state arrays must be contiguous %s arrays of nl entries" % t)
        act = None if active is None else np.ascontiguousarray(active, np.uint8)
        mn = None if min_d is None else np.ascontiguousarray(min_d, np.float32)
        mx = None if max_d is None else np.ascontiguousarray(max_d, np.float32)
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_stereo_match_ex(
            self._ctx, kl.ctypes.data, dl.ctypes.data, nl, kr.ctypes.data if nr else None,
            dr.ctypes.data if nr else None, nr, nrows, mbf, mb, min_x, ptr(act), ptr(mn), ptr(mx),
            int(th_orb_dist), int(bool(median_cut)), listed_dist.ctypes.data, best_idx.ctypes.data,
            best_dist.ctypes.data, u_right.ctypes.data, depth.ctypes.data, C.byref(nm)))
        return nm.value

    def stereo_match_ex_device(self, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, nbatch, stride, nrows, mbf, mb, min_x,
                               d_active, d_min_d, d_max_d, th_orb_dist, median_cut, d_listed,
                               d_bi, d_bd, d_ur, d_dp, d_nm):
        check(self._lib, self._ctx, self._lib.orbx_stereo_match_ex_device(
            self._ctx, ptr(d_kl), ptr(d_dl), ptr(d_nl), ptr(d_kr), ptr(d_dr), ptr(d_nr), nbatch, stride, nrows,
            mbf, mb, min_x, ptr(d_active), ptr(d_min_d), ptr(d_max_d), int(th_orb_dist), int(bool(median_cut)),
            ptr(d_listed), ptr(d_bi), ptr(d_bd), ptr(d_ur), ptr(d_dp), ptr(d_nm)))

    def stereo_match_sad(self, right_extractor, kl, dl, kr, dr, mbf, mb):
        """Frame::ComputeStereoMatches (src/Frame.cc:984-1173): Hamming row-band search + SAD
        sub-pixel refinement on the pyramids of this matcher's extractor (left) and
        `right_extractor`, which must hold the frames the keypoints came from."""
        kl = np.ascontiguousarray(kl, KP_DTYPE)
        kr = np.ascontiguousarray(kr, KP_DTYPE)
        dl, dr = _desc(dl), _desc(dr)
        nl, nr = len(kl), len(kr)
        bi = np.zeros(max(nl, 1), np.int32)
        bd = np.zeros(max(nl, 1), np.int32)
        ur = np.zeros(max(nl, 1), np.float32)
        dp = np.zeros(max(nl, 1), np.float32)
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_stereo_match_sad(
            self._ctx, right_extractor._ctx, kl.ctypes.data, dl.ctypes.data, nl, kr.ctypes.data if nr else None,
            dr.ctypes.data if nr else None, nr, mbf, mb,
            bi.ctypes.data, bd.ctypes.data, ur.ctypes.data, dp.ctypes.data, C.byref(nm)))
        return bi[:nl], bd[:nl], ur[:nl], dp[:nl], nm.value

    def stereo_match_sad_device(self, right_extractor, d_kl, d_dl, d_nl, d_kr, d_dr, d_nr, nbatch, stride, mbf, mb,
                                d_bi, d_bd, d_ur, d_dp, d_nm):
        check(self._lib, self._ctx, self._lib.orbx_stereo_match_sad_device(
            self._ctx, right_extractor._ctx, ptr(d_kl), ptr(d_dl), ptr(d_nl), ptr(d_kr), ptr(d_dr), ptr(d_nr),
            nbatch, stride, mbf, mb, ptr(d_bi), ptr(d_bd), ptr(d_ur), ptr(d_dp), ptr(d_nm)))

    # ---- frame grid (Frame::GetFeaturesInArea) ---------------------------------------------
    def features_in_area(self, kps, min_x, min_y, inv_w, inv_h, qx, qy, qr, qmin, qmax):
        """CSR candidate lists of Frame::GetFeaturesInArea (src/Frame.cc:669-722) for Q queries"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (qx, qy, qr))
        qmin, qmax = (np.ascontiguousarray(a, np.int32) for a in (qmin, qmax))
        Q = len(qx)
        off = np.zeros(Q + 1, np.int32)
        total = C.c_int32()
        cap = max(len(kps) * 8, 1024)
        while True:
            idx = np.zeros(cap, np.int32)
            rc = self._lib.orbx_features_in_area(self._ctx, kps.ctypes.data, len(kps), min_x, min_y, inv_w, inv_h,
                                                 qx.ctypes.data, qy.ctypes.data, qr.ctypes.data, qmin.ctypes.data,
                                                 qmax.ctypes.data, Q, off.ctypes.data, idx.ctypes.data, cap, C.byref(total))
            if rc == -4 and total.value > cap:
                cap = total.value
                continue
            check(self._lib, self._ctx, rc)
            return off, idx[:total.value]

    def grid_search(self, kps, desc, min_x, min_y, inv_w, inv_h, qx, qy, qr, qmin, qmax, qdesc):
        """GetFeaturesInArea fused with the best / second-best scan (src/ORBmatcher.cc:186-225)"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc, qdesc = _desc(desc), _desc(qdesc)
        qx, qy, qr = (np.ascontiguousarray(a, np.float32) for a in (qx, qy, qr))
        qmin, qmax = (np.ascontiguousarray(a, np.int32) for a in (qmin, qmax))
        Q = len(qx)
        outs = [np.zeros(max(Q, 1), np.int32) for _ in range(6)]
        check(self._lib, self._ctx, self._lib.orbx_grid_search(
            self._ctx, kps.ctypes.data, desc.ctypes.data, len(kps), min_x, min_y, inv_w, inv_h, qx.ctypes.data,
            qy.ctypes.data, qr.ctypes.data, qmin.ctypes.data, qmax.ctypes.data, qdesc.ctypes.data, Q,
            *[o.ctypes.data for o in outs]))
        return tuple(o[:Q] for o in outs)

    def grid_search_device(self, d_kps, d_desc, d_n, N, stride, nbatch, min_x, min_y, inv_w, inv_h,
                           d_qx, d_qy, d_qr, d_qmin, d_qmax, d_qdesc, d_nq, Q, q_stride, d_bi, d_bd, d_sd, d_bl, d_sl, d_nc):
        check(self._lib, self._ctx, self._lib.orbx_grid_search_device(
            self._ctx, ptr(d_kps), ptr(d_desc), ptr(d_n), N, stride, nbatch, min_x, min_y, inv_w, inv_h,
            ptr(d_qx), ptr(d_qy), ptr(d_qr), ptr(d_qmin), ptr(d_qmax), ptr(d_qdesc), ptr(d_nq), Q, q_stride,
            ptr(d_bi), ptr(d_bd), ptr(d_sd), ptr(d_bl), ptr(d_sl), ptr(d_nc)))

    # ---- searches with sequential exclusion ------------------------------------------------
    def SearchForInitialization(self, keys1, desc1, keys2, desc2, grid, vbPrevMatched, windowSize=10):
        """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)
        (src/ORBmatcher.cc:520-633).  keys / desc: mvKeysUn / mDescriptors of the two frames; grid =
        (mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv) of F2; vbPrevMatched: float32
        array n1 x 2, updated in place.  Returns (nmatches, vnMatches12)."""
        k1 = np.ascontiguousarray(keys1, KP_DTYPE)
        k2 = np.ascontiguousarray(keys2, KP_DTYPE)
        d1, d2 = _desc(desc1), _desc(desc2)
        if vbPrevMatched.dtype != np.float32 or not vbPrevMatched.flags.c_contiguous or vbPrevMatched.shape != (len(k1), 2):
            raise ValueError("vbPrevMatched must be a contiguous float32 array of shape (n1, 2)")
        m12 = np.full(max(len(k1), 1), -1, np.int32)
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_for_initialization(
            self._ctx, k1.ctypes.data, d1.ctypes.data, len(k1), k2.ctypes.data if len(k2) else None,
            d2.ctypes.data if len(k2) else None, len(k2), grid[0], grid[1], grid[2], grid[3],
            vbPrevMatched.ctypes.data, int(windowSize), self.mfNNratio, int(bool(self.mbCheckOrientation)),
            m12.ctypes.data, C.byref(nm)))
        return nm.value, m12[:len(k1)]

    def SearchByProjection(self, keys, desc, mvuRight, grid, mp, th, frame_obs, frame_match, match_score):
        """ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:155-241).
        mp: dict of per-map-point arrays (mbTrackInView, isBad, mTrackProjX, mTrackProjY, mTrackProjXR,
        mnTrackScaleLevel, mTrackViewCos, descriptor, Observations); frame_obs / frame_match /
        match_score: int32 arrays over the frame's keypoints, updated in place.  Returns nmatches."""
        kps = np.ascontiguousarray(keys, KP_DTYPE)
        d = _desc(desc)
        ur = None if mvuRight is None else np.ascontiguousarray(mvuRight, np.float32)
        for a in (frame_obs, frame_match, match_score):
            if a.dtype != np.int32 or not a.flags.c_contiguous or len(a) < len(kps):
                raise ValueError("frame arrays must be contiguous int32 arrays of N entries")
        inview = np.ascontiguousarray(mp["mbTrackInView"], np.uint8)
        bad = np.ascontiguousarray(mp["isBad"], np.uint8)
        px, py, pxr, vc = (np.ascontiguousarray(mp[k], np.float32) for k in
                           ("mTrackProjX", "mTrackProjY", "mTrackProjXR", "mTrackViewCos"))
        lvl = np.ascontiguousarray(mp["mnTrackScaleLevel"], np.int32)
        obs = np.ascontiguousarray(mp["Observations"], np.int32)
        md = _desc(mp["descriptor"])
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_by_projection(
            self._ctx, kps.ctypes.data, d.ctypes.data, ptr(ur), len(kps), grid[0], grid[1], grid[2], grid[3],
            inview.ctypes.data, bad.ctypes.data, px.ctypes.data, py.ctypes.data, pxr.ctypes.data, lvl.ctypes.data,
            vc.ctypes.data, md.ctypes.data, obs.ctypes.data, len(inview), float(th), self.mfNNratio,
            frame_obs.ctypes.data, frame_match.ctypes.data, match_score.ctypes.data, C.byref(nm)))
        return nm.value

    def SearchByProjection_Budget(self, keys, desc, mvuRight, grid, mp, th, time_constr, frame_obs, frame_match, match_score):
        """ORBmatcher::SearchByProjection_Budget(F, vpMapPoints, th, time_constr) (src/ORBmatcher.cc:45-153):
        SearchByProjection that stops once time_constr milliseconds have passed since the call was entered.
        Returns (nmatches, index of the map point after which the loop ended, or M)."""
        kps = np.ascontiguousarray(keys, KP_DTYPE)
        d = _desc(desc)
        ur = None if mvuRight is None else np.ascontiguousarray(mvuRight, np.float32)
        for a in (frame_obs, frame_match, match_score):
            if a.dtype != np.int32 or not a.flags.c_contiguous or len(a) < len(kps):
                raise ValueError("frame arrays must be contiguous int32 arrays of N entries")
        inview = np.ascontiguousarray(mp["mbTrackInView"], np.uint8)
        bad = np.ascontiguousarray(mp["isBad"], np.uint8)
        px, py, pxr, vc = (np.ascontiguousarray(mp[k], np.float32) for k in
                           ("mTrackProjX", "mTrackProjY", "mTrackProjXR", "mTrackViewCos"))
        lvl = np.ascontiguousarray(mp["mnTrackScaleLevel"], np.int32)
        obs = np.ascontiguousarray(mp["Observations"], np.int32)
        md = _desc(mp["descriptor"])
        nm, stop = C.c_int32(), C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_by_projection_budget(
            self._ctx, kps.ctypes.data, d.ctypes.data, ptr(ur), len(kps), grid[0], grid[1], grid[2], grid[3],
            inview.ctypes.data, bad.ctypes.data, px.ctypes.data, py.ctypes.data, pxr.ctypes.data, lvl.ctypes.data,
            vc.ctypes.data, md.ctypes.data, obs.ctypes.data, len(inview), float(th), self.mfNNratio, float(time_constr),
            frame_obs.ctypes.data, frame_match.ctypes.data, match_score.ctypes.data, C.byref(nm), C.byref(stop)))
        return nm.value, stop.value

    def GetCandidates(self, keys, desc, grid, mp, th):
        """ORBmatcher::GetCandidates (include/ORBmatcher.h:151-171) for every map point of mp in one device call, with
        the descriptor distance of every candidate.  Returns (cand_off[M + 1], cand_idx, cand_dist): the list of map
        point i is cand_idx[cand_off[i]:cand_off[i + 1]] (pMP->mvMatchCandidates)."""
        kps = np.ascontiguousarray(keys, KP_DTYPE)
        d = _desc(desc)
        inview = np.ascontiguousarray(mp["mbTrackInView"], np.uint8)
        bad = np.ascontiguousarray(mp["isBad"], np.uint8)
        px, py, vc = (np.ascontiguousarray(mp[k], np.float32) for k in ("mTrackProjX", "mTrackProjY", "mTrackViewCos"))
        lvl = np.ascontiguousarray(mp["mnTrackScaleLevel"], np.int32)
        md = _desc(mp["descriptor"])
        M = len(inview)
        off = np.zeros(M + 1, np.int32)
        total = C.c_int32()
        args = (self._ctx, kps.ctypes.data, d.ctypes.data, len(kps), grid[0], grid[1], grid[2], grid[3], inview.ctypes.data,
                bad.ctypes.data, px.ctypes.data, py.ctypes.data, lvl.ctypes.data, vc.ctypes.data, md.ctypes.data, M, float(th),
                off.ctypes.data)
        rc = self._lib.orbx_projection_candidates(*args, None, None, 0, C.byref(total))
        if rc == 0:                                   # no candidate at all
            return off, np.zeros(0, np.int32), np.zeros(0, np.uint16)
        if rc != -4:                                  # ORBX_ERR_CAPACITY is the answer to the sizing call
            check(self._lib, self._ctx, rc)
        idx = np.zeros(total.value, np.int32)
        dist = np.zeros(total.value, np.uint16)
        check(self._lib, self._ctx, self._lib.orbx_projection_candidates(*args, idx.ctypes.data, dist.ctypes.data,
                                                                         total.value, C.byref(total)))
        return off, idx, dist

    def MatchCandidates(self, keys, mvuRight, cands, mp, i, th, frame_obs, frame_match, match_score):
        """ORBmatcher::MatchCandidates (include/ORBmatcher.h:173-252) = SearchByProjection_OnePoint (:68-149) on the
        list GetCandidates made: map point i of mp against the frame's current assignment.  Returns bestIdx or -1."""
        kps = np.ascontiguousarray(keys, KP_DTYPE)
        ur = None if mvuRight is None else np.ascontiguousarray(mvuRight, np.float32)
        off, idx, dist = cands
        a, b = int(off[i]), int(off[i + 1])
        best = C.c_int32()
        li, ld = idx[a:b], dist[a:b]
        check(self._lib, self._ctx, self._lib.orbx_match_candidates(
            self._ctx, kps.ctypes.data, ptr(ur), len(kps), li.ctypes.data if b > a else None,
            ld.ctypes.data if b > a else None, b - a, int(mp["mbTrackInView"][i]), int(mp["isBad"][i]),
            float(mp["mTrackProjXR"][i]), int(mp["mnTrackScaleLevel"][i]), float(mp["mTrackViewCos"][i]), float(th),
            self.mfNNratio, int(i), int(mp["Observations"][i]), frame_obs.ctypes.data, frame_match.ctypes.data,
            match_score.ctypes.data, C.byref(best)))
        return best.value

    @staticmethod
    def feature_vector(fv):
        """A DBoW2::FeatureVector given as {node id: [feature indices]} -> (node ids ascending, offsets, indices)"""
        nodes = np.array(sorted(fv), np.int32)
        off = np.zeros(len(nodes) + 1, np.int32)
        for k, n in enumerate(nodes):
            off[k + 1] = off[k] + len(fv[int(n)])
        idx = np.concatenate([np.asarray(fv[int(n)], np.int32) for n in nodes]) if len(nodes) else np.zeros(0, np.int32)
        return nodes, off, np.ascontiguousarray(idx, np.int32)

    def SearchByBoW(self, kf_keys, kf_desc, kf_has_point, kf_featvec, f_keys, f_desc, f_featvec):
        """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:270-404).  Feature vectors as
        {node id: [feature indices]}; returns (nmatches, matches_f): the keyframe keypoint per frame keypoint or -1."""
        k1, k2 = np.ascontiguousarray(kf_keys, KP_DTYPE), np.ascontiguousarray(f_keys, KP_DTYPE)
        d1, d2 = _desc(kf_desc), _desc(f_desc)
        hp = np.ascontiguousarray(kf_has_point, np.uint8)
        n1, o1, i1 = self.feature_vector(kf_featvec)
        n2, o2, i2 = self.feature_vector(f_featvec)
        out = np.full(len(k2), -1, np.int32)
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_by_bow_frame(
            self._ctx, k1.ctypes.data, d1.ctypes.data, hp.ctypes.data, len(k1), n1.ctypes.data, o1.ctypes.data, i1.ctypes.data,
            len(n1), k2.ctypes.data, d2.ctypes.data, len(k2), n2.ctypes.data, o2.ctypes.data, i2.ctypes.data, len(n2),
            self.mfNNratio, int(bool(self.mbCheckOrientation)), out.ctypes.data, C.byref(nm)))
        return nm.value, out

    def SearchByBoW_KeyFrames(self, keys1, desc1, has_point1, featvec1, keys2, desc2, has_point2, featvec2):
        """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (src/ORBmatcher.cc:635-768); returns (nmatches,
        matches12): the keypoint of the second keyframe per keypoint of the first or -1."""
        k1, k2 = np.ascontiguousarray(keys1, KP_DTYPE), np.ascontiguousarray(keys2, KP_DTYPE)
        d1, d2 = _desc(desc1), _desc(desc2)
        h1, h2 = np.ascontiguousarray(has_point1, np.uint8), np.ascontiguousarray(has_point2, np.uint8)
        n1, o1, i1 = self.feature_vector(featvec1)
        n2, o2, i2 = self.feature_vector(featvec2)
        out = np.full(len(k1), -1, np.int32)
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_by_bow_keyframes(
            self._ctx, k1.ctypes.data, d1.ctypes.data, h1.ctypes.data, len(k1), n1.ctypes.data, o1.ctypes.data, i1.ctypes.data,
            len(n1), k2.ctypes.data, d2.ctypes.data, h2.ctypes.data, len(k2), n2.ctypes.data, o2.ctypes.data, i2.ctypes.data,
            len(n2), self.mfNNratio, int(bool(self.mbCheckOrientation)), out.ctypes.data, C.byref(nm)))
        return nm.value, out

    def SearchByProjectionKeyFrame(self, keys, desc, grid, kf, th, ORBdist, frame_obs, frame_match):
        """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1595-1721) from
        the projected points on.  kf: dict of per-keypoint arrays of the keyframe (visible, u, v, level, angle,
        descriptor, Observations); frame_obs (-1 = the keypoint holds no map point) / frame_match: int32 arrays over the
        current keypoints, updated in place.  Returns nmatches."""
        kps = np.ascontiguousarray(keys, KP_DTYPE)
        d = _desc(desc)
        for a in (frame_obs, frame_match):
            if a.dtype != np.int32 or not a.flags.c_contiguous or len(a) < len(kps):
                raise ValueError("frame arrays must be contiguous int32 arrays of N entries")
        vis = np.ascontiguousarray(kf["visible"], np.uint8)
        u, v, ang = (np.ascontiguousarray(kf[k], np.float32) for k in ("u", "v", "angle"))
        lvl = np.ascontiguousarray(kf["level"], np.int32)
        obs = np.ascontiguousarray(kf["Observations"], np.int32)
        kd = _desc(kf["descriptor"])
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_by_projection_keyframe(
            self._ctx, kps.ctypes.data, d.ctypes.data, len(kps), grid[0], grid[1], grid[2], grid[3], vis.ctypes.data,
            u.ctypes.data, v.ctypes.data, lvl.ctypes.data, ang.ctypes.data, kd.ctypes.data, obs.ctypes.data, len(vis),
            float(th), int(ORBdist), int(bool(self.mbCheckOrientation)), frame_obs.ctypes.data, frame_match.ctypes.data,
            C.byref(nm)))
        return nm.value

    @staticmethod
    def _points(pts):
        """dict of per-point arrays (visible, u, v, level, descriptor) -> contiguous arrays"""
        return (np.ascontiguousarray(pts["visible"], np.uint8), np.ascontiguousarray(pts["u"], np.float32),
                np.ascontiguousarray(pts["v"], np.float32), np.ascontiguousarray(pts["level"], np.int32), _desc(pts["descriptor"]))

    def SearchByProjectionSim3(self, keys, desc, grid, pts, th, kf_matched, kf_match):
        """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:406-518) from the projected
        points on; kf_matched (>= 0: vpMatched[idx] is set) and kf_match (index into pts) are updated in place."""
        kps, d = np.ascontiguousarray(keys, KP_DTYPE), _desc(desc)
        vis, u, v, lvl, pd = self._points(pts)
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_by_projection_sim3(
            self._ctx, kps.ctypes.data, d.ctypes.data, len(kps), grid[0], grid[1], grid[2], grid[3], vis.ctypes.data,
            u.ctypes.data, v.ctypes.data, lvl.ctypes.data, pd.ctypes.data, len(vis), int(th), kf_matched.ctypes.data,
            kf_match.ctypes.data, C.byref(nm)))
        return nm.value

    def FuseSearch(self, keys, desc, mvuRight, grid, pts, th, gate=True):
        """The search of ORBmatcher::Fuse (src/ORBmatcher.cc:937-1087 with gate, :1089-1212 without): returns (nFused,
        best_idx, best_dist) per map point; pts carries "ur" (u - bf * invz) when mvuRight is given."""
        kps, d = np.ascontiguousarray(keys, KP_DTYPE), _desc(desc)
        ur = None if mvuRight is None else np.ascontiguousarray(mvuRight, np.float32)
        vis, u, v, lvl, pd = self._points(pts)
        pur = None if ur is None else np.ascontiguousarray(pts["ur"], np.float32)
        bi, bd = np.zeros(len(vis), np.int32), np.zeros(len(vis), np.int32)
        nf = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_fuse_search(
            self._ctx, kps.ctypes.data, d.ctypes.data, ptr(ur), len(kps), grid[0], grid[1], grid[2], grid[3], vis.ctypes.data,
            u.ctypes.data, v.ctypes.data, ptr(pur), lvl.ctypes.data, pd.ctypes.data, len(vis), float(th), int(bool(gate)),
            bi.ctypes.data, bd.ctypes.data, C.byref(nf)))
        return nf.value, bi, bd

    def SearchBySim3(self, keys1, desc1, keys2, desc2, grid, pts1, pts2, th):
        """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1214-1438) from the projected points on: pts1 = pKF1's map points
        projected into pKF2, pts2 the other way.  Returns (nFound, matches12)."""
        k1, k2 = np.ascontiguousarray(keys1, KP_DTYPE), np.ascontiguousarray(keys2, KP_DTYPE)
        d1, d2 = _desc(desc1), _desc(desc2)
        a, b = self._points(pts1), self._points(pts2)
        out = np.full(len(k1), -1, np.int32)
        nf = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_by_sim3(
            self._ctx, k1.ctypes.data, d1.ctypes.data, len(k1), k2.ctypes.data, d2.ctypes.data, len(k2), grid[0], grid[1],
            grid[2], grid[3], *(x.ctypes.data for x in a), *(x.ctypes.data for x in b), float(th), out.ctypes.data,
            C.byref(nf)))
        return nf.value, out

    def SearchForTriangulation(self, keys1, desc1, has_point1, stereo1, featvec1, keys2, desc2, has_point2, stereo2, featvec2,
                               F12, epipole, bOnlyStereo=False):
        """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:770-935).  Returns (nmatches, vMatches12)."""
        k1, k2 = np.ascontiguousarray(keys1, KP_DTYPE), np.ascontiguousarray(keys2, KP_DTYPE)
        d1, d2 = _desc(desc1), _desc(desc2)
        h1, h2 = np.ascontiguousarray(has_point1, np.uint8), np.ascontiguousarray(has_point2, np.uint8)
        s1, s2 = np.ascontiguousarray(stereo1, np.uint8), np.ascontiguousarray(stereo2, np.uint8)
        n1, o1, i1 = self.feature_vector(featvec1)
        n2, o2, i2 = self.feature_vector(featvec2)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        out = np.full(len(k1), -1, np.int32)
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_for_triangulation(
            self._ctx, k1.ctypes.data, d1.ctypes.data, h1.ctypes.data, s1.ctypes.data, len(k1), n1.ctypes.data, o1.ctypes.data,
            i1.ctypes.data, len(n1), k2.ctypes.data, d2.ctypes.data, h2.ctypes.data, s2.ctypes.data, len(k2), n2.ctypes.data,
            o2.ctypes.data, i2.ctypes.data, len(n2), F.ctypes.data, float(epipole[0]), float(epipole[1]), int(bool(bOnlyStereo)),
            int(bool(self.mbCheckOrientation)), out.ctypes.data, C.byref(nm)))
        return nm.value, out

    def SearchByProjectionLast(self, keys, desc, mvuRight, grid, last, th, mbf, bForward, bBackward, frame_obs, frame_match):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, numVisibleMpt) (src/ORBmatcher.cc:1440-1593)
        from :1496 on.  last: dict of per-keypoint arrays of the last frame (visible, u, v, invzc, octave, angle,
        descriptor, Observations); frame_obs / frame_match: int32 arrays over the current keypoints, updated in
        place.  Returns nmatches."""
        kps = np.ascontiguousarray(keys, KP_DTYPE)
        d = _desc(desc)
        ur = None if mvuRight is None else np.ascontiguousarray(mvuRight, np.float32)
        for a in (frame_obs, frame_match):
            if a.dtype != np.int32 or not a.flags.c_contiguous or len(a) < len(kps):
                raise ValueError("frame arrays must be contiguous int32 arrays of N entries")
        vis = np.ascontiguousarray(last["visible"], np.uint8)
        u, v, iz, ang = (np.ascontiguousarray(last[k], np.float32) for k in ("u", "v", "invzc", "angle"))
        octv = np.ascontiguousarray(last["octave"], np.int32)
        obs = np.ascontiguousarray(last["Observations"], np.int32)
        ld = _desc(last["descriptor"])
        nm = C.c_int32()
        check(self._lib, self._ctx, self._lib.orbx_search_by_projection_last(
            self._ctx, kps.ctypes.data, d.ctypes.data, ptr(ur), len(kps), grid[0], grid[1], grid[2], grid[3],
            vis.ctypes.data, u.ctypes.data, v.ctypes.data, iz.ctypes.data, octv.ctypes.data, ang.ctypes.data,
            ld.ctypes.data, obs.ctypes.data, len(vis), float(th), float(mbf), int(bool(bForward)), int(bool(bBackward)),
            int(bool(self.mbCheckOrientation)), frame_obs.ctypes.data, frame_match.ctypes.data, C.byref(nm)))
        return nm.value

