/*
 * orbx.h -- C ABI of liborbx: the MI355X (gfx950) ORB front-end.
 *
 * The reference (GF-ORB-SLAM2) has no FFI layer: the boundary of this hot path
 * is the C++ class interface ORB_SLAM2::ORBextractor (include/ORBextractor.h:71-176)
 * and the static ORBmatcher::DescriptorDistance (include/ORBmatcher.h:48) plus the
 * best / second-best scans written around it.  Every entry point below names
 * the reference interface it replaces.  A header-only C++ adapter with the
 * reference's class shape on top of this ABI is in include/orbx_adapter.hpp.
 *
 * Conventions: every function returns 0 (ORBX_OK) or a negative error code and
 * never throws; all memory named in a signature is owned by the caller; a
 * context may be used by one thread at a time, distinct contexts concurrently
 * (the reference calls the left and right extractor from two threads,
 * src/Frame.cc:84-87).  There is no CPU path: without a usable gfx950 device
 * orbx_create fails.
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbx_ctx orbx_ctx;

/* layout-identical to cv::KeyPoint (7 x 4 bytes): pt.x, pt.y, size, angle,
 * response, octave, class_id -- what operator() fills (src/ORBextractor.cc:1109-1119,
 * :1123-1124, :1179-1186). */
typedef struct orbx_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbx_keypoint;

enum {
    ORBX_OK = 0,
    ORBX_ERR_INVALID = -1,     /* bad argument */
    ORBX_ERR_NO_DEVICE = -2,   /* no usable HIP device / not gfx950 */
    ORBX_ERR_HIP = -3,         /* a HIP runtime call failed (see orbx_last_error) */
    ORBX_ERR_CAPACITY = -4,    /* an output buffer is too small */
    ORBX_ERR_UNSUPPORTED = -5, /* geometry outside what the kernels support */
    ORBX_ERR_STATE = -6,       /* call order violated (e.g. no pyramid yet) */
    ORBX_ERR_IO = -7           /* a file could not be opened, read or written */
};

const char *orbx_version(void);

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * include/ORBextractor.h:80-81, src/ORBextractor.cc:967-1026. device >= 0. */
int orbx_create(orbx_ctx **out, int nfeatures, float scaleFactor, int nlevels,
                int iniThFAST, int minThFAST, int device);
/* ~ORBextractor, include/ORBextractor.h:83 */
void orbx_destroy(orbx_ctx *ctx);
/* text of the last error on this context ("" if none); never NULL */
const char *orbx_last_error(const orbx_ctx *ctx);
/* Run all work of this context on the caller's hipStream_t (NULL: the
 * context's own stream).  No counterpart in the reference. */
int orbx_set_stream(orbx_ctx *ctx, void *hip_stream);

/* GetLevels / GetScaleFactor / GetInitThres / GetMinThres, include/ORBextractor.h:92-110 */
int orbx_get_params(const orbx_ctx *ctx, int *nfeatures, float *scaleFactor, int *nlevels,
                    int *iniThFAST, int *minThFAST);
/* GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:112-130) and
 * mnFeaturesPerLevel; each array has nlevels entries, NULL to skip. */
int orbx_get_tables(const orbx_ctx *ctx, float *scale, float *inv_scale, float *sigma2,
                    float *inv_sigma2, int32_t *nfeat_per_level);
/* Upper bound of keypoints one w x h frame can yield (sum over levels of
 * max(N_level + 2, 4 * nIni), cf. src/ORBextractor.cc:618-622,679-680). */
int orbx_max_keypoints(orbx_ctx *ctx, int w, int h, int *max_kps);
/* Size of pyramid level `level` for a w x h input (src/ORBextractor.cc:1196-1197). */
int orbx_level_size(const orbx_ctx *ctx, int w, int h, int level, int *lw, int *lh);

/* ORBextractor::operator()(image, mask, keypoints, descriptors),
 * include/ORBextractor.h:88-90, src/ORBextractor.cc:1127-1190; the mask is
 * ignored there and has no parameter here.  img: 8-bit gray, `stride` bytes
 * per row, host memory.  kps[cap], desc[cap*32].  *n_out = number of keypoints;
 * ORBX_ERR_CAPACITY (with *n_out set) if cap is too small.  An empty image
 * (img == NULL or w,h <= 0) returns ORBX_OK with *n_out = 0 (:1130-1131). */
int orbx_extract(orbx_ctx *ctx, const uint8_t *img, int w, int h, int stride,
                 orbx_keypoint *kps, uint8_t *desc, int cap, int *n_out);

/* The same over nframes independent frames of one size in one launch
 * sequence.  Host buffers: kps[nframes*cap], desc[nframes*cap*32], n_out[nframes]. */
int orbx_extract_batch(orbx_ctx *ctx, const uint8_t *const *imgs, int nframes, int w, int h, int stride,
                       orbx_keypoint *kps, uint8_t *desc, int cap_per_frame, int32_t *n_out);

/* Hot entry: everything resident in device memory, asynchronous on the
 * context's stream, no host synchronisation.  d_imgs: nframes frames,
 * `stride` bytes per row, `frame_stride` bytes between frames.  A frame that
 * yields more than cap_per_frame keypoints stores the first cap_per_frame and
 * reports the full count in d_n_out.
 * When d_imgs, stride, frame_stride and w are multiples of 4 (and the image is at least 64 x 64) the
 * kernels read pyramid level 0 from d_imgs itself instead of copying it first (the reference's
 * copyMakeBorder of level 0, src/ORBextractor.cc:1212-1213, is then made only on demand).  Lifetime rule
 * that follows: a caller who goes on to read level 0 of this context's pyramid -- orbx_pyramid_level, or
 * orbx_stereo_match_sad[_device] with this context -- keeps d_imgs unchanged until that call has run;
 * a caller who reads neither may reuse the buffer as soon as the stream has passed this call.  The
 * host-buffer entry points (orbx_extract, orbx_extract_batch, orbx_compute_pyramid) hold their own copy
 * of the image and carry no such rule. */
int orbx_extract_batch_device(orbx_ctx *ctx, const uint8_t *d_imgs, int nframes, int w, int h,
                              int stride, size_t frame_stride, orbx_keypoint *d_kps, uint8_t *d_desc,
                              int cap_per_frame, int32_t *d_n_out);

/* ORBextractor::ComputePyramid(cv::Mat image), include/ORBextractor.h:143,
 * src/ORBextractor.cc:1192-1216 (public: it is called on its own by the
 * detection-I/O path, src/Frame.cc:187-190).  Host image, one frame. */
int orbx_compute_pyramid(orbx_ctx *ctx, const uint8_t *img, int w, int h, int stride);

/* mvImagePyramid[level] (include/ORBextractor.h:138): host-readable view of
 * the un-blurred level image of frame `frame` of the last extract /
 * compute_pyramid call.  *data points at pixel (0,0) of the level; at least
 * 19 valid REFLECT_101 border pixels surround it at the same stride.  Valid
 * until the next call on this context. */
int orbx_pyramid_level(orbx_ctx *ctx, int frame, int level, const uint8_t **data,
                       int *w, int *h, int *stride);

/* --- inspection of intermediate stages (used by the parity tests) -------- */
/* FAST candidates of one level after the per-cell threshold fallback
 * (src/ORBextractor.cc:1052-1101), in the reference's append order;
 * xyr = cap triples (x, y, response), coordinates relative to (16,16). */
int orbx_debug_candidates(orbx_ctx *ctx, int frame, int level, int32_t *xyr, int cap, int *n);
/* keypoints retained per level by the quadtree (src/ORBextractor.cc:1106-1107) */
int orbx_debug_level_counts(orbx_ctx *ctx, int frame, int32_t *counts);
/* blurred level image (src/ORBextractor.cc:1169-1170), w*h bytes, tightly packed */
int orbx_debug_blurred_level(orbx_ctx *ctx, int frame, int level, uint8_t *out, size_t out_size);
/* Per-stage GPU time of the batch calls since the last reset, measured with
 * HIP events on the context's stream. names: static strings; returns the
 * number of stages written (<= cap). Synchronises the stream. The blur's work
 * items normally ride in the FAST launch: their time is then part of "fast_cells" and
 * "blur" stays near zero. */
int orbx_stage_times(orbx_ctx *ctx, const char **names, float *total_ms, int32_t *launches, int cap);
int orbx_reset_stage_times(orbx_ctx *ctx);
/* Stage timing is off by default; enable = 1 adds seven event records to the launch sequence of
 * every batch call (about 10 us for a lone frame) and makes orbx_stage_times report them;
 * enable = 2 brackets only the FAST (+ blur) launch, the longest one (two records). */
int orbx_set_stage_timing(orbx_ctx *ctx, int enable);

/* --- matching -------------------------------------------------------------- */
/* ORBmatcher::DescriptorDistance, include/ORBmatcher.h:48,
 * src/ORBmatcher.cc:1768-1784: 256-bit Hamming distance of one pair, on the
 * calling thread (pure function, as in the reference). */
int orbx_hamming(const uint32_t a[8], const uint32_t b[8]);

/* Brute-force search: for every query the first two train descriptors in
 * (distance, index) order, i.e. the outcome of the reference's sequential
 * "dist < bestDist / else if dist < bestDist2" scan (src/ORBmatcher.cc:186-225)
 * over the whole train set.  T == 0: best_idx = -1, distances 256.
 * Host buffers, synchronous. */
int orbx_match_bruteforce(orbx_ctx *ctx, const uint32_t *q, int Q, const uint32_t *t, int T,
                          int32_t *best_idx, int32_t *best_dist, int32_t *second_dist);
/* Device-resident, asynchronous, batched: batch b uses queries
 * d_q + b*q_stride*8 (count d_nq[b] <= q_stride, or Q if d_nq == NULL) against
 * train descriptors d_t + b*t_stride*8 (count d_nt[b] or T; t_stride = 0
 * shares one train set).  Outputs are [nbatch][q_stride].  May grow two buffers of the
 * context (partial results of the train splits; 132 bytes per train descriptor when 64 or
 * more tiles of 256 queries sweep one train set); the call that grows one waits for the
 * context's earlier work first. */
int orbx_match_bruteforce_device(orbx_ctx *ctx, const uint32_t *d_q, int Q, int q_stride, const int32_t *d_nq,
                                 const uint32_t *d_t, int T, int t_stride, const int32_t *d_nt, int nbatch,
                                 int32_t *d_best_idx, int32_t *d_best_dist, int32_t *d_second_dist);

/* The same search without any allocation.  No counterpart in the reference: it is for callers
 * that cannot allocate device memory -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  orbx_match_bruteforce_device keeps the per-split partial
 * results of a large search in a buffer it allocates (and grows) on demand; this entry point keeps
 * them in a static device array of 4 MiB that every GPU the library runs on holds until the
 * process ends, and calls no hipMalloc / hipFree.  The partial results take
 * nbatch * tsplit * q_stride * 8 bytes (8000 x 100 000 in 64 splits: 3.9 MiB; 128 x 1000 x 1000 in 2: 2.0 MiB);
 * when the split count orbx_match_bruteforce_device would choose does not fit (8 x 8000 queries against
 * 100 000: 16 splits, 7.8 MiB), this entry point takes as many splits as fit (8), down to a single
 * split, which needs no scratch at all: the results are the same, and no problem is refused for
 * the scratch.  The limits of orbx_match_bruteforce_device apply (a train set of 2^22 descriptors
 * or more is rejected with ORBX_ERR_UNSUPPORTED).  All calls on one GPU share
 * that GPU's array: calls that overlap in time on one GPU (different streams or contexts) race;
 * calls ordered on one stream do not, and nothing carries over between calls. */
int orbx_match_bruteforce_device_static(orbx_ctx *ctx, const uint32_t *d_q, int Q, int q_stride, const int32_t *d_nq,
                                        const uint32_t *d_t, int T, int t_stride, const int32_t *d_nt, int nbatch,
                                        int32_t *d_best_idx, int32_t *d_best_dist, int32_t *d_second_dist);

/* Windowed search over per-query candidate lists in CSR form (cand_off[Q+1],
 * cand_idx[]): best / second-best distance, index and octave exactly as the
 * scan at src/ORBmatcher.cc:186-225 leaves them (first candidate in list order
 * wins ties).  t_octave, best_lvl, second_lvl may be NULL. Host buffers. */
int orbx_match_windowed(orbx_ctx *ctx, const uint32_t *q, int Q, const uint32_t *t, int T,
                        const int32_t *cand_off, const int32_t *cand_idx, const int32_t *t_octave,
                        int32_t *best_idx, int32_t *best_dist, int32_t *second_dist,
                        int32_t *best_lvl, int32_t *second_lvl);
/* All candidate distances, so that a caller can replay the reference's
 * sequential exclusions (e.g. src/ORBmatcher.cc:197-199, :559-560) on the host. */
int orbx_hamming_csr(orbx_ctx *ctx, const uint32_t *q, int Q, const uint32_t *t, int T,
                     const int32_t *cand_off, const int32_t *cand_idx, uint16_t *dist_out);

/* Stereo row-band search + disparity + median cut of a Frame that is matched in one piece:
 * Frame::PrepareStereoCandidates (include/Frame.h:217-251) followed by
 * Frame::ComputeStereoMatches_Undistorted(false) (src/Frame.cc:1262-1422) on a Frame without
 * map points.  Host buffers, one pair, any number of keypoints.
 * best_idx[i] = -1 if left keypoint i was skipped before the scan.
 * u_right/depth: -1 where there is no match.  *nmatched as returned there. */
int orbx_stereo_match(orbx_ctx *ctx, const orbx_keypoint *kl, const uint32_t *dl, int nl,
                      const orbx_keypoint *kr, const uint32_t *dr, int nr,
                      int nrows, float mbf, float mb, float min_x,
                      int32_t *best_idx, int32_t *best_dist, float *u_right, float *depth, int *nmatched);
/* Device-resident, asynchronous, batched over stereo pairs: pair b reads
 * d_kl + b*stride (d_nl[b] keypoints), d_dl + b*stride*8, likewise right.
 * Outputs [nbatch][stride]; d_nmatched[nbatch]. */
int orbx_stereo_match_device(orbx_ctx *ctx, const orbx_keypoint *d_kl, const uint32_t *d_dl, const int32_t *d_nl,
                             const orbx_keypoint *d_kr, const uint32_t *d_dr, const int32_t *d_nr,
                             int nbatch, int stride, int nrows, float mbf, float mb, float min_x,
                             int32_t *d_best_idx, int32_t *d_best_dist, float *d_u_right, float *d_depth,
                             int32_t *d_nmatched);

/* The delayed stereo matching of the reference's default build (DELAYED_STEREO_MATCHING,
 * include/Frame.h:38-40), where a Frame is matched in several calls:
 *   Frame::ComputeStereoMatches_Undistorted(isOnline)  src/Frame.cc:1262-1422
 *   Frame::ComputeStereoMatch_OnePoint(iL, thOrbDist, minD, maxD)  src/Frame.cc:1175-1260
 * The Frame's members live in caller arrays of nl entries that this call updates exactly where
 * the reference writes them (in/out):
 *   u_right, depth  mvuRight / mvDepth, -1 after PrepareStereoCandidates (include/Frame.h:220-221);
 *   listed_dist     mvDistIdx as an array: the Hamming distance of a left keypoint that holds a
 *                   stereo match (:1393, :1253), -1 otherwise; -1 everywhere after Prepare;
 *   best_idx, best_dist  result of the scan for the keypoints of this call (best_idx = -1: left
 *                   before the scan, the reference's `continue` / `return false`).
 * active[iL] != 0 selects the left keypoints this call works on (NULL: all); the caller derives it
 * from mvpMapPoints / mvStereoMatched as :1283-1297 and :1182 do, and sets mvStereoMatched for
 * them (:1304, :1187).  min_d / max_d: disparity window per left keypoint (NULL: 0 and mbf / mb);
 * the caller narrows it from the map point's depth as :1325-1338 do.  th_orb_dist: 75 in
 * ComputeStereoMatches_Undistorted (:1264), the argument of _OnePoint.  min_x: mnMinX (:1344), 0 for
 * _OnePoint (:1207).  median_cut != 0 is the !isOnline tail (:1400-1419): every listed keypoint,
 * of this call or an earlier one, whose distance is not below 2.1 x the median loses u_right and
 * depth.  *nmatched: keypoints of this call that reached the scan, minus the cut ones (:1397,
 * :1416) -- for _OnePoint the reference's return value is (listed_dist[iL] >= 0) afterwards. */
int orbx_stereo_match_ex(orbx_ctx *ctx, const orbx_keypoint *kl, const uint32_t *dl, int nl,
                         const orbx_keypoint *kr, const uint32_t *dr, int nr,
                         int nrows, float mbf, float mb, float min_x,
                         const uint8_t *active, const float *min_d, const float *max_d,
                         int th_orb_dist, int median_cut, int32_t *listed_dist,
                         int32_t *best_idx, int32_t *best_dist, float *u_right, float *depth, int *nmatched);
/* The same, device-resident, asynchronous, batched over pairs; the per-keypoint arrays are
 * [nbatch][stride]. */
int orbx_stereo_match_ex_device(orbx_ctx *ctx, const orbx_keypoint *d_kl, const uint32_t *d_dl, const int32_t *d_nl,
                                const orbx_keypoint *d_kr, const uint32_t *d_dr, const int32_t *d_nr,
                                int nbatch, int stride, int nrows, float mbf, float mb, float min_x,
                                const uint8_t *d_active, const float *d_min_d, const float *d_max_d,
                                int th_orb_dist, int median_cut, int32_t *d_listed_dist,
                                int32_t *d_best_idx, int32_t *d_best_dist, float *d_u_right, float *d_depth,
                                int32_t *d_nmatched);

/* Classic stereo matching with sub-pixel refinement: Frame::ComputeStereoMatches,
 * src/Frame.cc:984-1173 -- row-band Hamming search, 11x11 SAD sliding window on pyramid level
 * kpL.octave of the left and right extractors' pyramids (mvImagePyramid, :1089,:1111), parabola
 * fit, disparity gating, cut at 2.1 x the median SAD.  `left` and `right` are the two extractor
 * contexts (same device, same parameters, same frame size) whose last extract call produced the
 * keypoints; pair b uses frame b of both.  Work runs on left's stream after right's stream.
 * Host variant: one pair (frame 0 of both contexts). */
int orbx_stereo_match_sad(orbx_ctx *left, orbx_ctx *right, const orbx_keypoint *kl, const uint32_t *dl, int nl,
                          const orbx_keypoint *kr, const uint32_t *dr, int nr, float mbf, float mb,
                          int32_t *best_idx, int32_t *best_dist, float *u_right, float *depth, int *nmatched);
int orbx_stereo_match_sad_device(orbx_ctx *left, orbx_ctx *right,
                                 const orbx_keypoint *d_kl, const uint32_t *d_dl, const int32_t *d_nl,
                                 const orbx_keypoint *d_kr, const uint32_t *d_dr, const int32_t *d_nr,
                                 int nbatch, int stride, float mbf, float mb,
                                 int32_t *d_best_idx, int32_t *d_best_dist, float *d_u_right, float *d_depth,
                                 int32_t *d_nmatched);

/* --- frame grid: candidate generation for the windowed searches ------------------------ */
/* Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:536-551, :724-734) and
 * Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (:669-722) on the reference's 64 x 48
 * grid (include/Frame.h:92-93).  kps: the frame's (undistorted) keypoints; min_x/min_y =
 * mnMinX/mnMinY, inv_w/inv_h = mfGridElementWidthInv/HeightInv (src/Frame.cc:134-135).
 * Query q asks for the keypoints within the square of half-size qr[q] around (qx[q], qy[q])
 * with octave in [qmin[q], qmax[q]] under the reference's level rules.
 *
 * orbx_features_in_area: the candidate lists themselves, CSR (cand_off[Q+1], cand_idx[cap]),
 * each in the reference's order (cell columns, cell rows, ascending keypoint index).
 * *total = number of candidates; ORBX_ERR_CAPACITY if cap is smaller (cand_off is still valid). */
int orbx_features_in_area(orbx_ctx *ctx, const orbx_keypoint *kps, int N,
                          float min_x, float min_y, float inv_w, float inv_h,
                          const float *qx, const float *qy, const float *qr,
                          const int32_t *qmin, const int32_t *qmax, int Q,
                          int32_t *cand_off, int32_t *cand_idx, int cap, int *total);
/* orbx_grid_search: GetFeaturesInArea fused with the best / second-best scan of
 * src/ORBmatcher.cc:186-225 -- the result orbx_match_windowed gives on those lists, without
 * materialising them.  qdesc: Q query descriptors; desc: the frame's descriptors.
 * ncand (may be NULL): number of candidates per query. Host buffers, one frame. */
int orbx_grid_search(orbx_ctx *ctx, const orbx_keypoint *kps, const uint32_t *desc, int N,
                     float min_x, float min_y, float inv_w, float inv_h,
                     const float *qx, const float *qy, const float *qr,
                     const int32_t *qmin, const int32_t *qmax, const uint32_t *qdesc, int Q,
                     int32_t *best_idx, int32_t *best_dist, int32_t *second_dist,
                     int32_t *best_lvl, int32_t *second_lvl, int32_t *ncand);
/* Device-resident, asynchronous, batched over frames: frame b reads d_kps + b*stride
 * (d_n[b] keypoints, or N if d_n == NULL) and its queries at offset b*q_stride (d_nq[b] of them,
 * or Q).  Outputs are [nbatch][q_stride].  Feeds directly on orbx_extract_batch_device output. */
int orbx_grid_search_device(orbx_ctx *ctx, const orbx_keypoint *d_kps, const uint32_t *d_desc, const int32_t *d_n,
                            int N, int stride, int nbatch, float min_x, float min_y, float inv_w, float inv_h,
                            const float *d_qx, const float *d_qy, const float *d_qr,
                            const int32_t *d_qmin, const int32_t *d_qmax, const uint32_t *d_qdesc,
                            const int32_t *d_nq, int Q, int q_stride,
                            int32_t *d_best_idx, int32_t *d_best_dist, int32_t *d_second_dist,
                            int32_t *d_best_lvl, int32_t *d_second_lvl, int32_t *d_ncand);

/* --- searches whose queries exclude what earlier queries claimed ------------------------ */
/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize),
 * src/ORBmatcher.cc:520-633.  k1 / d1 = F1.mvKeysUn / F1.mDescriptors (n1 rows), k2 / d2 = F2's;
 * min_x .. inv_h = F2's mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv;
 * prev_matched_xy = vbPrevMatched as n1 (x, y) pairs, in/out (:628-630); nnratio / check_orientation =
 * the matcher's mfNNratio / mbCheckOrientation (include/ORBmatcher.h).  matches12[n1] = vnMatches12,
 * *nmatches = the return value.  The device forms F2.GetFeaturesInArea(...) (:540) for every level-0
 * keypoint of F1 with the distance of every candidate; the greedy part -- a candidate already held with
 * a distance <= ours is skipped (:559-560), a better claim un-matches the previous holder (:578-582),
 * the rotation histogram and ComputeThreeMaxima (:588-625) -- is walked in query order on the host. */
int orbx_search_for_initialization(orbx_ctx *ctx, const orbx_keypoint *k1, const uint32_t *d1, int n1,
                                   const orbx_keypoint *k2, const uint32_t *d2, int n2,
                                   float min_x, float min_y, float inv_w, float inv_h,
                                   float *prev_matched_xy, int window_size, float nnratio, int check_orientation,
                                   int32_t *matches12, int *nmatches);
/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th),
 * src/ORBmatcher.cc:155-241, with the MapPoint members it reads as arrays of M entries:
 * mbTrackInView, isBad() (NULL: none is bad), mTrackProjX, mTrackProjY, mTrackProjXR (may be NULL when
 * u_right is), mnTrackScaleLevel, mTrackViewCos, GetDescriptor(), Observations().
 * Frame: kps = mvKeysUn, desc = mDescriptors, u_right = mvuRight (NULL: monocular), the grid constants;
 * mvScaleFactors are the context's.  In/out arrays of N entries:
 *   frame_obs    Observations() of F.mvpMapPoints[idx], -1 where that pointer is NULL (the exclusion
 *                test :197-199); an accepted match stores the map point's own count;
 *   frame_match  index into the M map points of what this call assigned to F.mvpMapPoints[idx] (:233),
 *                untouched elsewhere -- initialise to -1;
 *   match_score  F.mvpMatchScore (:235).
 * *nmatches = the return value. */
int orbx_search_by_projection(orbx_ctx *ctx, const orbx_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                              float min_x, float min_y, float inv_w, float inv_h,
                              const uint8_t *mp_track_in_view, const uint8_t *mp_is_bad,
                              const float *mp_proj_x, const float *mp_proj_y, const float *mp_proj_xr,
                              const int32_t *mp_level, const float *mp_view_cos, const uint32_t *mp_desc,
                              const int32_t *mp_observations, int M, float th, float nnratio,
                              int32_t *frame_obs, int32_t *frame_match, int32_t *match_score, int *nmatches);

/* ORBmatcher::SearchByProjection_Budget(F, vpMapPoints, th, time_constr), src/ORBmatcher.cc:45-153 (the search of
 * Tracking::SearchAdditionalMatchesInFrame, src/Tracking.cc:2240): orbx_search_by_projection under a wall clock, started
 * when the call is entered and read after every map point that was accepted or whose best distance is above TH_HIGH
 * (:140-146; the reference's `continue`s skip the clock, so do these); the loop ends once time_constr_ms have passed.
 * *stopped_at = the map point after which it ended, M when it ran to the end.  The clock makes the result depend on
 * the machine, in the reference as here; with a budget that is never reached the result is orbx_search_by_projection's,
 * with 0 the loop ends after the first map point that reaches the clock. */
int orbx_search_by_projection_budget(orbx_ctx *ctx, const orbx_keypoint *kps, const uint32_t *desc, const float *u_right,
                                     int N, float min_x, float min_y, float inv_w, float inv_h,
                                     const uint8_t *mp_track_in_view, const uint8_t *mp_is_bad,
                                     const float *mp_proj_x, const float *mp_proj_y, const float *mp_proj_xr,
                                     const int32_t *mp_level, const float *mp_view_cos, const uint32_t *mp_desc,
                                     const int32_t *mp_observations, int M, float th, float nnratio, double time_constr_ms,
                                     int32_t *frame_obs, int32_t *frame_match, int32_t *match_score, int *nmatches,
                                     int *stopped_at);
/* ORBmatcher::GetCandidates(F, pMP, th), include/ORBmatcher.h:151-171, for all M map points in one device call, with
 * the descriptor distance of every candidate: cand_off[M + 1], cand_idx / cand_dist[cap] = pMP->mvMatchCandidates of
 * every point (GetFeaturesInArea order; empty for a point that is not in view or bad) and DescriptorDistance(
 * pMP->GetDescriptor(), F.mDescriptors.row(idx)) beside each index.  *total = cand_off[M]; with cap < *total (or NULL
 * buffers) the offsets are filled and ORBX_ERR_CAPACITY comes back.  Src/Observability.cc:1309-1329 runs this loop
 * before it draws map points one at a time. */
int orbx_projection_candidates(orbx_ctx *ctx, const orbx_keypoint *kps, const uint32_t *desc, int N,
                               float min_x, float min_y, float inv_w, float inv_h,
                               const uint8_t *mp_track_in_view, const uint8_t *mp_is_bad,
                               const float *mp_proj_x, const float *mp_proj_y, const int32_t *mp_level,
                               const float *mp_view_cos, const uint32_t *mp_desc, int M, float th,
                               int32_t *cand_off, int32_t *cand_idx, uint16_t *cand_dist, int cap, int *total);
/* ORBmatcher::MatchCandidates(F, pMP, th), include/ORBmatcher.h:173-252, and -- on the list orbx_projection_candidates
 * made for the point -- ORBmatcher::SearchByProjection_OnePoint(F, pMP, th), :68-149 (src/Observability.cc:1413, :1685):
 * one map point against its candidate list and the frame's current assignment.  Host arithmetic only (the distances
 * are in the list), so the caller can interleave it with its own work in any order; frame_obs / frame_match /
 * match_score as in orbx_search_by_projection, mp_index = what frame_match receives.  *best_idx = the return value
 * (-1: no match). */
int orbx_match_candidates(orbx_ctx *ctx, const orbx_keypoint *kps, const float *u_right, int N,
                          const int32_t *cand_idx, const uint16_t *cand_dist, int ncand,
                          int track_in_view, int is_bad, float proj_xr, int level, float view_cos, float th, float nnratio,
                          int mp_index, int mp_observations,
                          int32_t *frame_obs, int32_t *frame_match, int32_t *match_score, int *best_idx);

/* ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), src/ORBmatcher.cc:270-404 (the
 * search of Tracking::TrackReferenceKeyFrame and Relocalization).  A DBoW2::FeatureVector (pKF->mFeatVec, F.mFeatVec) is
 * passed as its node ids in ascending order (the order a std::map iterates in), nodes + 1 offsets and the feature
 * indices of every node in stored order.  kf_kps = pKF->mvKeysUn, kf_has_point[i] = vpMapPointsKF[i] exists and is
 * not bad (:304-308), f_kps = F.mvKeys (only the angles are read, :349).  matches_f[n_f]: the keypoint of the keyframe
 * whose map point vpMapPointMatches[i] receives, -1 for NULL.  The device computes the distance of every (keyframe
 * keypoint, frame keypoint) pair that shares a node; the walk that skips frame keypoints matched earlier (:320-321),
 * the TH_LOW / ratio test (:339-341) and the rotation histogram (:347-357, :383-401) run on the host. */
int orbx_search_by_bow_frame(orbx_ctx *ctx, const orbx_keypoint *kf_kps, const uint32_t *kf_desc,
                             const uint8_t *kf_has_point, int n_kf,
                             const int32_t *kf_node, const int32_t *kf_off, const int32_t *kf_idx, int kf_nodes,
                             const orbx_keypoint *f_kps, const uint32_t *f_desc, int n_f,
                             const int32_t *f_node, const int32_t *f_off, const int32_t *f_idx, int f_nodes,
                             float nnratio, int check_orientation, int32_t *matches_f, int *nmatches);
/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), src/ORBmatcher.cc:635-768 (loop
 * closing): both sides need a good map point (:671-675, :689-693), a keypoint of pKF2 is taken once (vbMatched2), the
 * distance must be strictly below TH_LOW (:711).  matches12[n1]: the keypoint of pKF2 whose map point vpMatches12[i]
 * receives, -1 for NULL. */
int orbx_search_by_bow_keyframes(orbx_ctx *ctx, const orbx_keypoint *k1, const uint32_t *d1, const uint8_t *has_point1, int n1,
                                 const int32_t *node1, const int32_t *off1, const int32_t *idx1, int nodes1,
                                 const orbx_keypoint *k2, const uint32_t *d2, const uint8_t *has_point2, int n2,
                                 const int32_t *node2, const int32_t *off2, const int32_t *idx2, int nodes2,
                                 float nnratio, int check_orientation, int32_t *matches12, int *nmatches);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono, numVisibleMpt),
 * src/ORBmatcher.cc:1440-1593 -- the search of Tracking::TrackWithMotionModel -- from the point where the map points of
 * the last frame have been projected into the current one (:1496 on).  The projection (:1451-1493) stays with the
 * caller, who passes per keypoint i of the last frame: lp_visible (mvpMapPoints[i] exists, is no outlier, invzc >= 0,
 * (u, v) inside the image bounds -- the points :1495 counts into numVisibleMpt), lp_u / lp_v / lp_invzc, lp_octave =
 * LastFrame.mvKeys[i].octave, lp_angle = LastFrame.mvKeysUn[i].angle, lp_desc = the map point's descriptor,
 * lp_observations = its Observations(); forward / backward as :1461-1462 decide them; mbf = CurrentFrame.mbf.
 * Current frame as in orbx_search_by_projection.  frame_match[i2] = index i of the last frame's keypoint whose map
 * point was assigned to CurrentFrame.mvpMapPoints[i2]; the rotation check (:1570-1590) resets what it removes to -1
 * (frame_obs too: the reference stores NULL).  *nmatches = the return value. */
int orbx_search_by_projection_last(orbx_ctx *ctx, const orbx_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                                   float min_x, float min_y, float inv_w, float inv_h,
                                   const uint8_t *lp_visible, const float *lp_u, const float *lp_v, const float *lp_invzc,
                                   const int32_t *lp_octave, const float *lp_angle, const uint32_t *lp_desc,
                                   const int32_t *lp_observations, int M, float th, float mbf, int forward, int backward,
                                   int check_orientation, int32_t *frame_obs, int32_t *frame_match, int *nmatches);
/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist),
 * src/ORBmatcher.cc:1595-1721 (Tracking::Relocalization), from the projected points on: per keypoint i of the keyframe
 * kp_visible (its map point exists, is not bad, is not in sAlreadyFound, projects inside the image bounds and lies
 * inside its scale-invariance distances, :1615-1644), kp_u / kp_v, kp_level = pMP->PredictScale(dist3D, &CurrentFrame),
 * kp_angle = pKF->mvKeysUn[i].angle, kp_desc = pMP->GetDescriptor(), kp_observations = what frame_obs receives (>= 0).
 * A keypoint of the current frame is out when it holds any map point (frame_obs[i2] >= 0, :1664-1665), levels
 * [kp_level - 1, kp_level + 1], accepted with bestDist <= orb_dist (:1678); rotation check as in the call above. */
int orbx_search_by_projection_keyframe(orbx_ctx *ctx, const orbx_keypoint *kps, const uint32_t *desc, int N,
                                       float min_x, float min_y, float inv_w, float inv_h,
                                       const uint8_t *kp_visible, const float *kp_u, const float *kp_v,
                                       const int32_t *kp_level, const float *kp_angle, const uint32_t *kp_desc,
                                       const int32_t *kp_observations, int M, float th, int orb_dist,
                                       int check_orientation, int32_t *frame_obs, int32_t *frame_match, int *nmatches);

/* --- the searches of the mapping and loop-closing threads ---------------------------------------------------------------
 * Each takes the map points from the place where the reference has projected them: visible[i] = every `continue` before
 * the window search passed (bad / already found / behind the camera / outside the image / outside the scale-invariance
 * distances / viewing angle), u / v = the projection, level = PredictScale(...).  A point looks at the keypoints of levels
 * [level - 1, level] inside th * mvScaleFactors[level] (GetFeaturesInArea without levels + the loops' own level test). */
/* ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched,
 * int th), src/ORBmatcher.cc:406-518 (LoopClosing::ComputeSim3).  kf_matched[idx] >= 0 = vpMatched[idx] != NULL, in/out
 * (:491-492, :512: an accepted point stores 1); kf_match[idx] = the point of vpPoints this call put there (initialise to -1). */
int orbx_search_by_projection_sim3(orbx_ctx *ctx, const orbx_keypoint *kps, const uint32_t *desc, int N,
                                   float min_x, float min_y, float inv_w, float inv_h,
                                   const uint8_t *mp_visible, const float *mp_u, const float *mp_v, const int32_t *mp_level,
                                   const uint32_t *mp_desc, int M, int th, int32_t *kf_matched, int32_t *kf_match, int *nmatches);
/* The search of ORBmatcher::Fuse: gate != 0 the form of src/ORBmatcher.cc:937-1087 (LocalMapping::SearchInNeighbors; reprojection
 * test :1026-1050 with u_right = pKF->mvuRight -- NULL for a monocular keyframe -- and mp_ur = u - bf * invz, the context's
 * mvInvLevelSigma2), gate == 0 the Sim3 form of :1089-1212 (LoopClosing::SearchAndFuse).  best_idx[i] = bestIdx where bestDist <=
 * TH_LOW (:1064, :1194), else -1; best_dist[i] = bestDist (256: nothing compared).  The search of a point does not depend on
 * the other points; replacing or adding the observation (:1066-1081, :1196-1206) stays with the caller, in order. */
int orbx_fuse_search(orbx_ctx *ctx, const orbx_keypoint *kps, const uint32_t *desc, const float *u_right, int N,
                     float min_x, float min_y, float inv_w, float inv_h,
                     const uint8_t *mp_visible, const float *mp_u, const float *mp_v, const float *mp_ur,
                     const int32_t *mp_level, const uint32_t *mp_desc, int M, float th, int gate,
                     int32_t *best_idx, int32_t *best_dist, int *nfused);
/* ORBmatcher::SearchBySim3, src/ORBmatcher.cc:1214-1438: p1_* = the map points of pKF1 (per keypoint of pKF1) projected into pKF2
 * with the Sim3 and searched among pKF2's keypoints (k2, d2), p2_* the other way round; both keyframes share the grid
 * constants.  matches12[i1] = idx2 for the pairs that chose each other (:1419-1435), -1 elsewhere; *nfound = the return value. */
int orbx_search_by_sim3(orbx_ctx *ctx, const orbx_keypoint *k1, const uint32_t *d1, int n1,
                        const orbx_keypoint *k2, const uint32_t *d2, int n2,
                        float min_x, float min_y, float inv_w, float inv_h,
                        const uint8_t *p1_visible, const float *p1_u, const float *p1_v, const int32_t *p1_level,
                        const uint32_t *p1_desc,
                        const uint8_t *p2_visible, const float *p2_u, const float *p2_v, const int32_t *p2_level,
                        const uint32_t *p2_desc, float th, int32_t *matches12, int *nfound);
/* ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:770-935 (LocalMapping::CreateNewMapPoints).  has_point = the keypoint
 * already holds a map point (:815, :838), stereo = mvuRight >= 0, feature vectors as in orbx_search_by_bow_frame, F12 = the
 * fundamental matrix row-major as float, (ex, ey) = the epipole in the second image (:776-783); mvScaleFactors and
 * mvLevelSigma2 are the context's.  The device computes the distance of every pair sharing a node; the host applies
 * `dist > TH_LOW || dist > bestDist` (:851: the last of equal candidates wins), the epipole distance (:856-862),
 * CheckDistEpipolarLine (:251-268) and the rotation histogram.  The reference never sets its vbMatched2, so a keypoint of the
 * second keyframe can be matched to several of the first; so here.  matches12[n1] = vMatches12. */
int orbx_search_for_triangulation(orbx_ctx *ctx, const orbx_keypoint *k1, const uint32_t *d1, const uint8_t *has_point1,
                                  const uint8_t *stereo1, int n1,
                                  const int32_t *node1, const int32_t *off1, const int32_t *idx1, int nodes1,
                                  const orbx_keypoint *k2, const uint32_t *d2, const uint8_t *has_point2,
                                  const uint8_t *stereo2, int n2,
                                  const int32_t *node2, const int32_t *off2, const int32_t *idx2, int nodes2,
                                  const float *F12, float ex, float ey, int only_stereo, int check_orientation,
                                  int32_t *matches12, int *nmatches);

/* --- keypoint undistortion --------------------------------------------------- */
/* Frame::UndistortKeyPoints / UndistortKeyPointsStereo, src/Frame.cc:745-852: the keypoint
 * coordinates go through cv::undistortPoints(pts, K, dist, R, P) (fisheye = 0) or
 * cv::fisheye::undistortPoints (fisheye != 0, the reference's USE_FISHEYE_DISTORTION build);
 * every other keypoint field is copied.  K: 3x3 row-major; dist: ndist coefficients
 * (0, 4, 5, 8 or 12 in OpenCV's order k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4; fisheye: 4);
 * R: 3x3 or NULL (identity); P: 3 x p_cols with p_cols 3 or 4, or NULL (identity) -- the mono
 * path passes P = K (:839), the stereo path the rectified projection (:771).  The reference's
 * "first coefficient == 0: plain copy" shortcut (:747, :820) is the caller's.
 * Host buffers, one frame; kps_out may equal kps_in. */
int orbx_undistort_keypoints(orbx_ctx *ctx, const orbx_keypoint *kps_in, int n,
                             const double *K, const double *dist, int ndist,
                             const double *R, const double *P, int p_cols, int fisheye,
                             orbx_keypoint *kps_out);
/* Device-resident, asynchronous, batched: frame b holds d_n[b] keypoints (n if d_n == NULL) at
 * d_kps_in + b*cap -- the layout orbx_extract_batch_device writes.  d_kps_out may equal d_kps_in. */
int orbx_undistort_keypoints_device(orbx_ctx *ctx, const orbx_keypoint *d_kps_in, const int32_t *d_n, int n,
                                    int cap, int nframes, const double *K, const double *dist, int ndist,
                                    const double *R, const double *P, int p_cols, int fisheye,
                                    orbx_keypoint *d_kps_out);

/* --- detection I/O (host only, no device needed) ------------------------------ */
/* Frame::ExportToYML, src/Frame.cc:590-609: one cv::FileStorage YAML document with the keys
 * nNextId, mnId, mTimeStamp, mK (3x3 float, skipped if NULL), mvKeys, mDescriptors (n x 32, dt u),
 * mvKeysRight, mDescriptorsRight.  legacy_flat != 0 writes each keypoint list as one flat
 * sequence of 7 numbers per keypoint (OpenCV before 3.3.1) instead of one flow sequence per
 * keypoint; OpenCV's reader accepts both. */
int orbx_detection_write_yaml(const char *path, double nNextId, double mnId, double mTimeStamp, const float *mK,
                              const orbx_keypoint *keys, const uint8_t *desc, int n,
                              const orbx_keypoint *keys_right, const uint8_t *desc_right, int n_right,
                              int legacy_flat);
/* KeyPointLog::ReadFromYAML, include/Util.hpp:400-449: reads mTimeStamp, mvKeys, mDescriptors,
 * mvKeysRight, mDescriptorsRight.  With keys == NULL (keys_right == NULL) only the count is
 * returned; ORBX_ERR_CAPACITY if a list is longer than its buffer.  A missing file gives
 * ORBX_ERR_IO with *mTimeStamp = -1 and both counts 0, the state the reference leaves (:407-413). */
int orbx_detection_read_yaml(const char *path, double *mTimeStamp,
                             orbx_keypoint *keys, uint8_t *desc, int cap, int *n,
                             orbx_keypoint *keys_right, uint8_t *desc_right, int cap_right, int *n_right);
/* text of the last detection-I/O error on this thread */
const char *orbx_io_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
