/*
 * orb_oracle.h -- CPU restatement of the GF-ORB-SLAM2 ORB front-end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped GPU library links, loads or
 * calls this code; it exists so that tests, the smoke check and the CPU-baseline
 * leg of the benchmark have something to compare the GPU results against.
 *
 * PARITY UNPINNED: the reference holds no golden keypoints / descriptors /
 * matches for this path (test/test_Stereo.cpp:230-245 asserts nothing), and its
 * arithmetic lives in OpenCV 3.4.1, which is neither vendored in the reference
 * nor installed here.  This file therefore restates the published OpenCV 3.4.1
 * generic (non-IPP) algorithms at the reference's call sites; every OpenCV
 * semantic choice is in orb_oracle.c under a "OpenCV 3.4.1 semantics" banner.
 * What stands in for the missing pin: double-precision bounds on resize, blur
 * and fastAtan2 (tests/test_oracle_semantics.py) and comparisons with
 * scikit-image / scipy where a second Python of the image has them
 * (tests/test_oracle_skimage.py: FAST segment test, resize sampling rule,
 * Gaussian, patch moments, rBRIEF pattern).
 *
 * Each function cites the reference file:line it follows.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OO_EDGE_THRESHOLD 19   /* src/ORBextractor.cc:72 */
#define OO_HALF_PATCH 15       /* src/ORBextractor.cc:71 */
#define OO_PATCH 31            /* src/ORBextractor.cc:70 */
#define OO_MAX_LEVELS 16

/* layout-identical to cv::KeyPoint (7 x 4 bytes) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} oo_keypoint;

/* FAST candidate in the coordinate frame of ComputeKeyPointsOctTree:
 * relative to (minBorderX, minBorderY) = (16, 16) of the level image. */
typedef struct {
    int32_t x, y, response;
} oo_cand;

typedef struct oo_extractor oo_extractor;

/* ORBextractor::ORBextractor, src/ORBextractor.cc:967-1026 */
oo_extractor *oo_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
void oo_destroy(oo_extractor *ex);

/* tables built by the constructor; arrays of nlevels entries (umax: 16) */
void oo_get_tables(const oo_extractor *ex, float *scale, float *inv_scale, float *sigma2,
                   float *inv_sigma2, int *nfeat_per_level, int *umax16);
const int *oo_pattern(void); /* 256*4 ints */

/* level size for an input w x h: src/ORBextractor.cc:1196-1197 */
void oo_level_size(const oo_extractor *ex, int w, int h, int level, int *lw, int *lh);

/* ORBextractor::ComputePyramid, src/ORBextractor.cc:1192-1216.
 * Planes are kept inside the extractor until the next call. */
int oo_compute_pyramid(oo_extractor *ex, const uint8_t *img, int w, int h, int stride);
/* bordered plane of a level: pointer to its top-left (border) pixel; the level
 * image starts at (+19,+19).  *bw,*bh are the bordered sizes; pitch == *bw. */
const uint8_t *oo_pyramid_plane(const oo_extractor *ex, int level, int *bw, int *bh);

/* cv::resize(INTER_LINEAR, 8UC1) for one plane (used by ComputePyramid) */
void oo_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                         uint8_t *dst, int dw, int dh, int dstride);

/* cv::FAST(img, thr, nonmax=true), TYPE_9_16, on a sub-image (literal OpenCV
 * algorithm: threshold table, run-length test, cornerScore, 3-row NMS).
 * out: keypoints (x, y, response) in sub-image coordinates, emission order. */
int oo_fast9_nms(const uint8_t *img, int w, int h, int stride, int threshold,
                 oo_cand *out, int cap);
/* threshold-free FAST score of one pixel by the brute-force arc definition
 * (independent second derivation, used to cross-check the literal path). */
int oo_fast_score_bruteforce(const uint8_t *center, int stride);

/* Cell loop of ComputeKeyPointsOctTree, src/ORBextractor.cc:1036-1101, on the
 * pyramid held by ex.  literal!=0: two cv::FAST calls per cell as written;
 * literal==0: single score-image formulation (must give the same list). */
int oo_detect_level(const oo_extractor *ex, int level, int literal, oo_cand *out, int cap);

/* ORBextractor::DistributeOctTree, src/ORBextractor.cc:488-711 (+DivideNode :431-486).
 * Coordinates of cands are relative to (minX,minY); width=maxX-minX, height=maxY-minY.
 * Writes the indices (into cands) of the retained keypoints in output order.
 * The reference breaks ties between equally sized nodes by heap address
 * (:633); here: among equal sizes the node created LATER is split first.
 * literal!=0: std::list emulation; literal==0: array passes + sort + scan. */
int oo_distribute_octree(const oo_cand *cands, int n, int width, int height, int N,
                         int literal, int *out_idx, int cap);

/* IC_Angle, src/ORBextractor.cc:74-101 (moments) and cv::fastAtan2 */
void oo_ic_moments(const uint8_t *center, int stride, const int *umax16, int *m01, int *m10);
float oo_fast_atan2(float y, float x);

/* cv::GaussianBlur(7x7, sigma 2, REFLECT_101) on an 8-bit plane whose 3-pixel
 * border is already present around [0,w)x[0,h): src/ORBextractor.cc:1169-1170 */
void oo_gaussian_blur7(const uint8_t *src_origin, int w, int h, int sstride,
                       uint8_t *dst, int dstride);

/* computeOrbDescriptor, src/ORBextractor.cc:104-150.
 * The reference takes cos / sin of a float angle through the float overloads (cosf / sinf of the
 * libm it is linked with, :108-109), which are not correctly rounded and differ between libm
 * versions in the last bit.  OO_TRIG_LIBM calls this host's cosf / sinf, as the reference does
 * (the default of an extractor and of oo_orb_descriptor); OO_TRIG_ROUNDED uses the correctly
 * rounded values, the platform-independent formulation the GPU kernel evaluates.
 * Measured with oo_count_trig_differences on glibc 2.35 (x86-64) over all 1 135 869 952 float
 * angles in [0, 360): cosf differs from the correctly rounded value for 483 807 of them, sinf for
 * 1 001 902 (always by one unit in the last place).  A differing a or b changes a descriptor only
 * when a rotated coordinate lies within that unit of a rounding boundary: tests measure the share
 * of descriptors that differ between the two modes (none on the committed frames; the parity
 * budget is 0.1 %).  PARITY UNPINNED like the rest of this file. */
#define OO_TRIG_LIBM 0
#define OO_TRIG_ROUNDED 1
void oo_set_trig_mode(oo_extractor *ex, int mode);
void oo_orb_descriptor(const uint8_t *center, int stride, float angle_deg, uint8_t desc[32]);
void oo_orb_descriptor_mode(const uint8_t *center, int stride, float angle_deg, uint8_t desc[32], int trig_mode);
void oo_count_trig_differences(uint32_t first, uint32_t step, uint64_t counts[2], uint64_t *tested);

/* ORBextractor::operator(), src/ORBextractor.cc:1127-1190. Returns the number
 * of keypoints (<= cap) or -1 if cap is too small. literal selects the literal
 * or the derived formulations of FAST cells and the quadtree. */
int oo_extract(oo_extractor *ex, const uint8_t *img, int w, int h, int stride,
               oo_keypoint *kps, uint8_t *desc, int cap, int literal);
/* per-level keypoint counts of the last oo_extract call */
void oo_last_level_counts(const oo_extractor *ex, int *counts);

/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1768-1784 */
int oo_hamming(const uint32_t a[8], const uint32_t b[8]);

/* Brute force: for each query, the first two candidates in (distance, index)
 * order -- the result of the reference's sequential best/second scan
 * (src/ORBmatcher.cc:193-225) over the whole train set. */
void oo_match_bruteforce(const uint32_t *q, int Q, const uint32_t *t, int T,
                         int32_t *best_idx, int32_t *best_dist, int32_t *second_dist);

/* Windowed search over CSR candidate lists; literal sequential scan of
 * src/ORBmatcher.cc:186-225 (no exclusion tests). t_octave may be NULL. */
void oo_match_windowed(const uint32_t *q, int Q, const uint32_t *t, int T,
                       const int32_t *cand_off, const int32_t *cand_idx, const int32_t *t_octave,
                       int32_t *best_idx, int32_t *best_dist, int32_t *second_dist,
                       int32_t *best_lvl, int32_t *second_lvl);

void oo_hamming_csr(const uint32_t *q, int Q, const uint32_t *t, int T,
                    const int32_t *cand_off, const int32_t *cand_idx, uint16_t *dist_out);

/* Stereo row-band search: Frame::PrepareStereoCandidates (include/Frame.h:217-251)
 * + the Hamming scan of Frame::ComputeStereoMatches_Undistorted
 * (src/Frame.cc:1341-1375).  best_dist starts at TH_HIGH=100, strict '<'.
 * best_idx = -1 when the row has no candidates / maxU < minX (left kp skipped). */
void oo_stereo_search(const oo_keypoint *kl, const uint32_t *dl, int nl,
                      const oo_keypoint *kr, const uint32_t *dr, int nr,
                      const float *scale_factors, int nrows, float minD, float maxD, float minX,
                      int32_t *best_idx, int32_t *best_dist);
/* Disparity/depth + median cut: src/Frame.cc:1380-1419 (offline variant). */
int oo_stereo_finish(const oo_keypoint *kl, int nl, const oo_keypoint *kr,
                     const int32_t *best_idx, const int32_t *best_dist,
                     float mbf, float minD, float maxD, float *u_right, float *depth);

/* The Frame members the delayed stereo matching of the default build reads and writes
 * (DELAYED_STEREO_MATCHING, include/Frame.h:40): mvuRight, mvDepth, mvStereoMatched and mvDistIdx
 * (pairs (bestDist, iL) in push_back order; sorted in place by the offline call, src/Frame.cc:1402).
 * All arrays are owned by the caller and hold nl entries. */
typedef struct {
    float *u_right, *depth;
    uint8_t *stereo_matched;
    int32_t *dist_idx_dist, *dist_idx_il;
    int32_t n_dist_idx;
} oo_stereo_state;
/* Frame::PrepareStereoCandidates, include/Frame.h:217-251 (state part; the row table is rebuilt
 * from the right keypoints by the calls below) */
void oo_stereo_prepare(oo_stereo_state *st, int nl);
/* Frame::ComputeStereoMatches_Undistorted(isOnline), src/Frame.cc:1262-1422: online = only
 * keypoints with a map point, disparity window narrowed to prior +- DISPARITY_THRES, no median cut;
 * offline = every keypoint not matched yet, then the cut at 2.1 x median over all of mvDistIdx.
 * Returns nmatched as the reference does (scanned keypoints minus cut ones). */
int oo_stereo_undistorted(const oo_keypoint *kl, const uint32_t *dl, int nl,
                          const oo_keypoint *kr, const uint32_t *dr, int nr,
                          const float *scale_factors, int nrows, float mbf, float mb, float minX,
                          float disparity_thres, const uint8_t *has_mappoint, const float *prior_disp,
                          int is_online, oo_stereo_state *st, int32_t *best_idx, int32_t *best_dist);
/* Frame::ComputeStereoMatch_OnePoint, src/Frame.cc:1175-1260; returns 1 where it returns true */
int oo_stereo_one_point(const oo_keypoint *kl, const uint32_t *dl, int nl,
                        const oo_keypoint *kr, const uint32_t *dr, int nr,
                        const float *scale_factors, int nrows, float mbf,
                        const uint8_t *has_mappoint, int iL, int thOrbDist, float minD, float maxD,
                        oo_stereo_state *st, int32_t *best_idx, int32_t *best_dist);

/* Frame::ComputeStereoMatches, src/Frame.cc:984-1173 (the classic variant): row-band Hamming
 * search, then 11x11 SAD sliding-window refinement on pyramid level kpL.octave of the two
 * extractors' last pyramids, parabola sub-pixel fit, disparity gating, and the median cut on the
 * SAD distances.  The reference indexes its row table without clamping (:1007-1011); rows are
 * clamped to [0, nRows-1] here (no effect on keypoints the extractor produces).
 * best_idx/best_dist: result of the Hamming scan (best_idx = -1 if the left keypoint was skipped);
 * u_right/depth: -1 where no stereo match survives. Returns the number of surviving matches. */
int oo_stereo_classic(const oo_extractor *exl, const oo_extractor *exr,
                      const oo_keypoint *kl, const uint32_t *dl, int nl,
                      const oo_keypoint *kr, const uint32_t *dr, int nr,
                      float mbf, float mb,
                      int32_t *best_idx, int32_t *best_dist, float *u_right, float *depth);

/* Frame::AssignFeaturesToGrid + PosInGrid (src/Frame.cc:536-551, :724-734) and
 * Frame::GetFeaturesInArea (:669-722) on a 64 x 48 grid (include/Frame.h:92-93): candidate
 * lists of Q queries in CSR form, each in the reference's order (cell columns, then cell rows,
 * then ascending keypoint index).  Returns the total number of candidates (cand_idx may be NULL
 * to count only); cand_off has Q+1 entries. */
int oo_features_in_area(const oo_keypoint *kps, int n, float minX, float minY, float invW, float invH,
                        const float *qx, const float *qy, const float *qr, const int32_t *qmin,
                        const int32_t *qmax, int Q, int32_t *cand_off, int32_t *cand_idx, int cap);

/* ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1723-1764; sizes[i] = histo[i].size().
 * ind1..3 must be initialised by the caller (the reference passes -1). */
void oo_three_maxima(const int *sizes, int L, int *ind1, int *ind2, int *ind3);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize),
 * src/ORBmatcher.cc:520-633, on plain arrays: k1/d1 = F1.mvKeysUn / mDescriptors, k2/d2 = F2's,
 * minX..invH = F2's grid constants (mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv),
 * prev_matched_xy = vbPrevMatched (n1 x 2 floats, updated in place as :628-630 do),
 * nnratio / check_orientation = the matcher's mfNNratio / mbCheckOrientation.
 * matches12[n1] = vnMatches12.  Returns nmatches. */
int oo_search_for_initialization(const oo_keypoint *k1, const uint32_t *d1, int n1,
                                 const oo_keypoint *k2, const uint32_t *d2, int n2,
                                 float minX, float minY, float invW, float invH,
                                 float *prev_matched_xy, int windowSize, float nnratio, int check_orientation,
                                 int32_t *matches12);

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th),
 * src/ORBmatcher.cc:155-241, on plain arrays.  Frame: kps = mvKeysUn, desc = mDescriptors,
 * u_right = mvuRight (NULL: monocular, every entry -1), scale_factors = mvScaleFactors, grid constants.
 * Map point i: mbTrackInView, isBad(), mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos,
 * GetDescriptor(), Observations().
 * frame_obs[N] (in/out): Observations() of F.mvpMapPoints[idx], -1 where that pointer is NULL; an
 * accepted match stores the map point's own count, as assigning the pointer does (:233).
 * frame_match[N] (in/out): index of the map point assigned to F.mvpMapPoints[idx] by this call
 * (untouched elsewhere); match_score[N] (in/out): F.mvpMatchScore.  Returns nmatches. */
int oo_search_by_projection(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                            const float *scale_factors, float minX, float minY, float invW, float invH,
                            const uint8_t *mp_track_in_view, const uint8_t *mp_is_bad,
                            const float *mp_proj_x, const float *mp_proj_y, const float *mp_proj_xr,
                            const int32_t *mp_level, const float *mp_view_cos, const uint32_t *mp_desc,
                            const int32_t *mp_observations, int M, float th, float nnratio,
                            int32_t *frame_obs, int32_t *frame_match, int32_t *match_score);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono, numVisibleMpt),
 * src/ORBmatcher.cc:1440-1593 (the search of TrackWithMotionModel), from :1496 on: the caller has projected the last
 * frame's map points (u, v, invzc; lp_visible = the point exists, is no outlier and projects inside the image) and
 * decided bForward / bBackward (:1461-1462).  lp_octave = LastFrame.mvKeys[i].octave, lp_angle =
 * LastFrame.mvKeysUn[i].angle, lp_desc = the map point's descriptor, lp_observations = its Observations().
 * Current frame as in oo_search_by_projection; a match removed by the rotation check resets frame_match and frame_obs
 * to -1 (the reference stores NULL, :1585).  Returns nmatches. */
int oo_search_by_projection_last(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                                 const float *scale_factors, float minX, float minY, float invW, float invH,
                                 const uint8_t *lp_visible, const float *lp_u, const float *lp_v, const float *lp_invzc,
                                 const int32_t *lp_octave, const float *lp_angle, const uint32_t *lp_desc,
                                 const int32_t *lp_observations, int M, float th, float mbf, int bForward, int bBackward,
                                 int check_orientation, int32_t *frame_obs, int32_t *frame_match);

/* The good-feature forms of the projection search (one map point per call, or a loop under a time budget); frame
 * state as in oo_search_by_projection.
 * oo_get_candidates: ORBmatcher::GetCandidates, include/ORBmatcher.h:151-171 -> list length, indices in cand_idx.
 * oo_match_candidates: ORBmatcher::MatchCandidates, :173-252 -> bestIdx or -1.
 * oo_search_by_projection_one_point: ORBmatcher::SearchByProjection_OnePoint, :68-149 -> bestIdx or -1.
 * oo_search_by_projection_budget: ORBmatcher::SearchByProjection_Budget, src/ORBmatcher.cc:45-153; time_constr in
 * ms; *stopped_at = the map point after which the budget ended the loop, M when it ran to the end. */
int oo_get_candidates(const oo_keypoint *kps, int N, const float *scale_factors, float minX, float minY,
                      float invW, float invH, int track_in_view, int is_bad, float proj_x, float proj_y,
                      int level, float view_cos, float th, int32_t *cand_idx, int cap);
int oo_match_candidates(const oo_keypoint *kps, const uint32_t *desc, const float *u_right,
                        const float *scale_factors, const int32_t *cand_idx, int ncand,
                        int track_in_view, int is_bad, float proj_xr, int level, float view_cos,
                        const uint32_t *mp_desc, int mp_observations, int mp_index, float th, float nnratio,
                        int32_t *frame_obs, int32_t *frame_match, int32_t *match_score);
int oo_search_by_projection_one_point(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                                      const float *scale_factors, float minX, float minY, float invW, float invH,
                                      int track_in_view, int is_bad, float proj_x, float proj_y, float proj_xr,
                                      int level, float view_cos, const uint32_t *mp_desc, int mp_observations,
                                      int mp_index, float th, float nnratio,
                                      int32_t *frame_obs, int32_t *frame_match, int32_t *match_score);
int oo_search_by_projection_budget(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                                   const float *scale_factors, float minX, float minY, float invW, float invH,
                                   const uint8_t *mp_track_in_view, const uint8_t *mp_is_bad,
                                   const float *mp_proj_x, const float *mp_proj_y, const float *mp_proj_xr,
                                   const int32_t *mp_level, const float *mp_view_cos, const uint32_t *mp_desc,
                                   const int32_t *mp_observations, int M, float th, float nnratio, double time_constr,
                                   int32_t *frame_obs, int32_t *frame_match, int32_t *match_score, int *stopped_at);

/* ORBmatcher::SearchByBoW: KeyFrame against Frame (src/ORBmatcher.cc:270-404) and KeyFrame against KeyFrame (:635-768).
 * A DBoW2::FeatureVector is passed as node ids in ascending order, offsets (nodes + 1) and feature indices;
 * has_point = the keypoint's map point exists and is not bad; the outputs hold keypoint indices of the other side
 * (-1 = NULL).  Both return nmatches. */
int oo_search_by_bow_frame(const oo_keypoint *kf_kps, const uint32_t *kf_desc, const uint8_t *kf_has_point,
                           const int32_t *kf_node, const int32_t *kf_off, const int32_t *kf_idx, int kf_nodes,
                           const oo_keypoint *f_kps, const uint32_t *f_desc, int n_f,
                           const int32_t *f_node, const int32_t *f_off, const int32_t *f_idx, int f_nodes,
                           float nnratio, int check_orientation, int32_t *matches_f);
int oo_search_by_bow_keyframes(const oo_keypoint *k1, const uint32_t *d1, const uint8_t *has_point1, int n1,
                               const int32_t *node1, const int32_t *off1, const int32_t *idx1, int nodes1,
                               const oo_keypoint *k2, const uint32_t *d2, const uint8_t *has_point2, int n2,
                               const int32_t *node2, const int32_t *off2, const int32_t *idx2, int nodes2,
                               float nnratio, int check_orientation, int32_t *matches12);

/* ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1595-1721 (relocalisation),
 * from the projected points on; frame_obs[i2] >= 0 = the keypoint holds a map point.  Returns nmatches. */
int oo_search_by_projection_keyframe(const oo_keypoint *kps, const uint32_t *desc, int N, const float *scale_factors,
                                     float minX, float minY, float invW, float invH,
                                     const uint8_t *kp_visible, const float *kp_u, const float *kp_v,
                                     const int32_t *kp_level, const float *kp_angle, const uint32_t *kp_desc,
                                     const int32_t *kp_observations, int M, float th, int ORBdist, int check_orientation,
                                     int32_t *frame_obs, int32_t *frame_match);

/* The searches of the mapping and loop-closing threads, from the projected points on (visible = every test before the window
 * search passed): SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) src/ORBmatcher.cc:406-518; the search of Fuse (:937-1087
 * with gate != 0, :1089-1212 with gate == 0); SearchBySim3 :1214-1438; SearchForTriangulation :770-935. */
int oo_search_by_projection_sim3(const oo_keypoint *kps, const uint32_t *desc, int N, const float *scale_factors,
                                 float minX, float minY, float invW, float invH, const uint8_t *mp_visible,
                                 const float *mp_u, const float *mp_v, const int32_t *mp_level, const uint32_t *mp_desc,
                                 int M, int th, int32_t *kf_matched, int32_t *kf_match);
int oo_fuse_search(const oo_keypoint *kps, const uint32_t *desc, const float *u_right, int N, const float *scale_factors,
                   const float *inv_sigma2, float minX, float minY, float invW, float invH, const uint8_t *mp_visible,
                   const float *mp_u, const float *mp_v, const float *mp_ur, const int32_t *mp_level,
                   const uint32_t *mp_desc, int M, float th, int gate, int32_t *best_idx, int32_t *best_dist);
int oo_search_by_sim3(const oo_keypoint *k1, const uint32_t *d1, int n1, const oo_keypoint *k2, const uint32_t *d2, int n2,
                      const float *scale_factors, float minX, float minY, float invW, float invH,
                      const uint8_t *p1_visible, const float *p1_u, const float *p1_v, const int32_t *p1_level,
                      const uint32_t *p1_desc, const uint8_t *p2_visible, const float *p2_u, const float *p2_v,
                      const int32_t *p2_level, const uint32_t *p2_desc, float th, int32_t *matches12);
int oo_search_for_triangulation(const oo_keypoint *k1, const uint32_t *d1, const uint8_t *has_point1, const uint8_t *stereo1,
                                int n1, const int32_t *node1, const int32_t *off1, const int32_t *idx1, int nodes1,
                                const oo_keypoint *k2, const uint32_t *d2, const uint8_t *has_point2, const uint8_t *stereo2,
                                const int32_t *node2, const int32_t *off2, const int32_t *idx2, int nodes2,
                                const float *scale_factors, const float *sigma2, const float *F12, float ex, float ey,
                                int bOnlyStereo, int check_orientation, int32_t *matches12);

/* Frame::UndistortKeyPoints[Stereo], src/Frame.cc:745-852: cv::undistortPoints (fisheye = 0)
 * or cv::fisheye::undistortPoints (fisheye != 0) on n points (x, y interleaved).
 * K: 3x3 row-major camera matrix; dist: ndist (0, 4, 5, 8 or 12; fisheye: 4) coefficients;
 * R: 3x3 or NULL; P: 3 x p_cols (p_cols = 3 or 4) or NULL.  The caller applies the reference's
 * "first coefficient is zero: copy" shortcut. */
void oo_undistort_points(const float *xy_in, int n, const double K[9], const double *dist, int ndist,
                         const double *R, const double *P, int p_cols, int fisheye, float *xy_out);
void oo_undistort_rr(const double *R, const double *P, int p_cols, double RR[9]);

#ifdef __cplusplus
}
#endif
#endif
