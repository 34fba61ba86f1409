// orbx_host.hpp -- header-only C++ host class over the C ABI (orbx.h) with the
// shape of ORB_SLAM2::ORBextractor (include/ORBextractor.h:71-176) but without
// OpenCV types, so that it can be built anywhere.  include/orbx_adapter.hpp
// layers the cv::Mat / cv::KeyPoint signatures on top of it.
//
// Error behaviour: the reference's operator() cannot fail short of an assert;
// a GPU call can, so failures surface as std::runtime_error with the text of
// orbx_last_error().
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "orbx.h"

namespace orbx {

struct ImageView {
    const uint8_t* data = nullptr;
    int cols = 0, rows = 0;
    size_t step = 0;   // bytes per row
    ImageView() = default;
    ImageView(const uint8_t* d, int c, int r, size_t s) : data(d), cols(c), rows(r), step(s) {}
    bool empty() const { return data == nullptr || cols <= 0 || rows <= 0; }
};

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };   // include/ORBextractor.h:74-78

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
        : nlevels_(nlevels), scaleFactor_(scaleFactor), iniThFAST_(iniThFAST), minThFAST_(minThFAST)
    {
        int rc = orbx_create(&ctx_, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device);
        if (rc != ORBX_OK) throw std::runtime_error("orbx_create failed with code " + std::to_string(rc));
        mvScaleFactor.resize(nlevels);
        mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        orbx_get_tables(ctx_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                        mvInvLevelSigma2.data(), mnFeaturesPerLevel.data());
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor() { orbx_destroy(ctx_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // operator()(image, mask, keypoints, descriptors): include/ORBextractor.h:88-90.
    // The mask is ignored, as in the reference.  descriptors: n x 32 bytes, row i
    // belongs to keypoints[i]; cleared when there are no keypoints.
    void operator()(const ImageView& image, const ImageView& /*mask*/, std::vector<orbx_keypoint>& keypoints,
                    std::vector<uint8_t>& descriptors)
    {
        if (image.empty()) return;   // src/ORBextractor.cc:1130-1131: outputs untouched
        int cap = 0;
        check(orbx_max_keypoints(ctx_, image.cols, image.rows, &cap));
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0;
        check(orbx_extract(ctx_, image.data, image.cols, image.rows, (int)image.step, keypoints.data(),
                           descriptors.data(), cap, &n));
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
        if (mirrorPyramid) refresh_pyramid();
    }

    // include/ORBextractor.h:143 (public: the detection-I/O path calls it on its own)
    void ComputePyramid(const ImageView& image)
    {
        check(orbx_compute_pyramid(ctx_, image.data, image.cols, image.rows, (int)image.step));
        refresh_pyramid();
    }

    int GetLevels() { return nlevels_; }
    float GetScaleFactor() { return scaleFactor_; }
    int GetInitThres() { return iniThFAST_; }
    int GetMinThres() { return minThFAST_; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // mvImagePyramid (include/ORBextractor.h:138): host-readable views of the
    // un-blurred levels of the last call, each surrounded by >= 19 REFLECT_101
    // border pixels at the same step; valid until the next call on this object.
    std::vector<ImageView> mvImagePyramid;
    // Only the classic stereo matcher reads mvImagePyramid on the host (src/Frame.cc:991,1089,1111).  A caller that
    // does not (monocular, RGB-D, or stereo through orbx_stereo_match_sad, which reads the pyramid on the device)
    // clears this flag and saves the 1.5 MB download per frame; the views then stay empty.
    bool mirrorPyramid = true;

    orbx_ctx* context() { return ctx_; }

private:
    void check(int rc)
    {
        if (rc != ORBX_OK) throw std::runtime_error(std::string("liborbx: ") + orbx_last_error(ctx_));
    }
    void refresh_pyramid()
    {
        for (int l = 0; l < nlevels_; l++) {
            const uint8_t* d = nullptr;
            int w = 0, h = 0, s = 0;
            check(orbx_pyramid_level(ctx_, 0, l, &d, &w, &h, &s));
            mvImagePyramid[l] = ImageView(d, w, h, (size_t)s);
        }
    }

    orbx_ctx* ctx_ = nullptr;
    int nlevels_;
    float scaleFactor_;
    int iniThFAST_, minThFAST_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int32_t> mnFeaturesPerLevel;
};

// Several GPUs in one process: one extractor (context + stream) per listed device, one host thread per
// extractor during a call -- the pattern of the reference's stereo Frame constructor, which runs its left and right
// extractor on two threads (src/Frame.cc:84-87), widened from 2 cameras to G devices.  Frames are independent, so a
// batch is cut into contiguous frame ranges [g*B/G, (g+1)*B/G) and every shard writes straight into the caller's
// arrays at its frames' places: results arrive in frame order and are identical to one extractor running the whole
// batch.  No collective is involved (one process, host memory).  A device may be listed more than once (two
// contexts on one GPU overlap their transfers and kernels).
class ShardedExtractor {
public:
    ShardedExtractor(const std::vector<int>& devices, int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                     int minThFAST)
    {
        if (devices.empty()) throw std::invalid_argument("ShardedExtractor needs at least one device");
        for (int d : devices)
            shards_.emplace_back(new ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, d));
        for (auto& e : shards_) e->mirrorPyramid = false;
    }

    int shards() const { return (int)shards_.size(); }
    ORBextractor& shard(int g) { return *shards_[g]; }
    // frames [first, last) of shard g of a batch of nframes
    static void frame_range(int g, int nshards, int nframes, int& first, int& last)
    {
        first = (int)((long long)g * nframes / nshards);
        last = (int)((long long)(g + 1) * nframes / nshards);
    }
    int max_keypoints(int w, int h)
    {
        int cap = 0;
        if (orbx_max_keypoints(shards_[0]->context(), w, h, &cap) != ORBX_OK)
            throw std::runtime_error(std::string("liborbx: ") + orbx_last_error(shards_[0]->context()));
        return cap;
    }

    // nframes frames of one size (imgs[f]: `stride` bytes per row) -> kps[nframes * cap], desc[nframes * cap * 32],
    // n_out[nframes], the layout of orbx_extract_batch.  Throws the first shard's error after all shards have finished.
    void ExtractBatch(const uint8_t* const* imgs, int nframes, int w, int h, int stride, orbx_keypoint* kps, uint8_t* desc,
                      int cap, int32_t* n_out)
    {
        const int G = shards();
        std::vector<int> rc(G, ORBX_OK);
        std::vector<std::thread> th;
        auto work = [&](int g) {
            int f0, f1;
            frame_range(g, G, nframes, f0, f1);
            if (f1 > f0)
                rc[g] = orbx_extract_batch(shards_[g]->context(), imgs + f0, f1 - f0, w, h, stride, kps + (size_t)f0 * cap,
                                           desc + (size_t)f0 * cap * 32, cap, n_out + f0);
        };
        for (int g = 1; g < G; g++) th.emplace_back(work, g);
        work(0);                                   // the calling thread is shard 0's
        for (auto& t : th) t.join();
        for (int g = 0; g < G; g++)
            if (rc[g] != ORBX_OK)
                throw std::runtime_error("liborbx (shard " + std::to_string(g) + "): " + orbx_last_error(shards_[g]->context()));
    }

private:
    std::vector<std::unique_ptr<ORBextractor>> shards_;
};

// Frame::UndistortKeyPoints (src/Frame.cc:818-852) and each half of UndistortKeyPointsStereo
// (:745-816) on the extractor's device: K 3x3 row-major, dist = the mDistCoef entries, R / P null
// or the rectification (P 3 x p_cols).  A first coefficient of 0 copies, as the reference does.
inline void UndistortKeyPoints(ORBextractor& ex, const std::vector<orbx_keypoint>& keys, const double* K,
                               const std::vector<double>& dist, const double* R, const double* P, int p_cols,
                               bool fisheye, std::vector<orbx_keypoint>& keysUn)
{
    keysUn = keys;
    if (dist.empty() || dist[0] == 0.0 || keys.empty()) return;
    int rc = orbx_undistort_keypoints(ex.context(), keys.data(), (int)keys.size(), K, dist.data(), (int)dist.size(), R, P,
                                      p_cols, fisheye ? 1 : 0, keysUn.data());
    if (rc != ORBX_OK) throw std::runtime_error(std::string("liborbx: ") + orbx_last_error(ex.context()));
}

// The delayed stereo matching of the reference's default build (DELAYED_STEREO_MATCHING, include/Frame.h:38-40):
// the Frame members PrepareStereoCandidates sets up (include/Frame.h:217-251) and the calls that fill them in
// several steps (src/Frame.cc:1175-1260, :1262-1422), with the searches on the device through
// orbx_stereo_match_ex.  A Frame keeps one of these next to mvKeysUn / mvKeysRightUn / mDescriptors /
// mDescriptorsRight and replaces the bodies of the three member functions by the calls below.
//   has_mappoint[iL] = (mvpMapPoints[iL] != NULL)
//   prior_disp[iL]   = mbf / Pc.z of a map point that is not bad and lies in front of the camera (:1325-1334), else <= 0
struct DelayedStereo {
    static constexpr float kDisparityThres = 50.0f;   // DISPARITY_THRES, include/Frame.h:47

    std::vector<float> mvuRight, mvDepth;
    std::vector<uint8_t> mvStereoMatched, has_mappoint;
    std::vector<float> prior_disp;
    std::vector<int32_t> listed_dist;     // mvDistIdx, indexed by left keypoint (-1: not listed)
    std::vector<int32_t> best_idx, best_dist;

    // Frame::PrepareStereoCandidates (the row table itself is implicit in the device search)
    void PrepareStereoCandidates(int N)
    {
        mvuRight.assign(N, -1.0f);
        mvDepth.assign(N, -1.0f);
        mvStereoMatched.assign(N, 0);
        has_mappoint.resize(N, 0);
        prior_disp.resize(N, 0.0f);
        listed_dist.assign(N, -1);
        best_idx.assign(N, -1);
        best_dist.assign(N, 100);
    }

    // Frame::ComputeStereoMatches_Undistorted(isOnline); returns nmatched
    int ComputeStereoMatches_Undistorted(ORBextractor& ex, bool isOnline, const std::vector<orbx_keypoint>& keysUn,
                                         const uint8_t* desc, const std::vector<orbx_keypoint>& keysRightUn,
                                         const uint8_t* descRight, int nRows, float mbf, float mb, float mnMinX)
    {
        const int N = (int)keysUn.size();
        if (N == 0) return 0;
        std::vector<uint8_t> active(N);
        std::vector<float> minD(N, 0.0f), maxD(N, mbf / mb);
        for (int i = 0; i < N; i++) {
            active[i] = isOnline ? (has_mappoint[i] && !mvStereoMatched[i]) : !mvStereoMatched[i];   // :1283-1297
            if (has_mappoint[i] && prior_disp[i] > 0) {                                              // :1325-1338
                minD[i] = std::max(prior_disp[i] - kDisparityThres, 0.0f);
                maxD[i] = std::min(prior_disp[i] + kDisparityThres, mbf / mb);
            }
            if (active[i]) mvStereoMatched[i] = 1;                                                   // :1304
        }
        int nmatched = 0;
        check(ex, orbx_stereo_match_ex(ex.context(), keysUn.data(), reinterpret_cast<const uint32_t*>(desc), N,
                                       keysRightUn.data(), reinterpret_cast<const uint32_t*>(descRight),
                                       (int)keysRightUn.size(), nRows, mbf, mb, mnMinX, active.data(), minD.data(),
                                       maxD.data(), 75, isOnline ? 0 : 1, listed_dist.data(), best_idx.data(),
                                       best_dist.data(), mvuRight.data(), mvDepth.data(), &nmatched));
        return nmatched;
    }

    // Frame::ComputeStereoMatch_OnePoint(iL, thOrbDist, minD, maxD)
    bool ComputeStereoMatch_OnePoint(ORBextractor& ex, int iL, int thOrbDist, float minD, float maxD,
                                     const std::vector<orbx_keypoint>& keysUn, const uint8_t* desc,
                                     const std::vector<orbx_keypoint>& keysRightUn, const uint8_t* descRight, int nRows,
                                     float mbf, float mb)
    {
        const int N = (int)keysUn.size();
        if (iL < 0 || iL >= N) return false;
        if (!has_mappoint[iL] || mvStereoMatched[iL]) return false;
        mvStereoMatched[iL] = 1;
        std::vector<uint8_t> active(N, 0);
        active[iL] = 1;
        std::vector<float> mn(N, minD), mx(N, maxD);
        int nmatched = 0;
        check(ex, orbx_stereo_match_ex(ex.context(), keysUn.data(), reinterpret_cast<const uint32_t*>(desc), N,
                                       keysRightUn.data(), reinterpret_cast<const uint32_t*>(descRight),
                                       (int)keysRightUn.size(), nRows, mbf, mb, 0.0f, active.data(), mn.data(), mx.data(),
                                       thOrbDist, 0, listed_dist.data(), best_idx.data(), best_dist.data(),
                                       mvuRight.data(), mvDepth.data(), &nmatched));
        return listed_dist[iL] >= 0;
    }

    // Several one-point calls in one device call.  The reference calls the one-point form once per tracked map point
    // (hundreds of times per frame); a whole-frame launch and a synchronisation per point would cost N times the work
    // it needs.  A one-point call changes nothing another one reads, so the batch equals the calls made one after the
    // other.  iL[k] with the window [minD[k], maxD[k]]; result[k] = what the k-th call returns.
    std::vector<uint8_t> ComputeStereoMatch_Points(ORBextractor& ex, const std::vector<int>& iL, int thOrbDist,
                                                   const std::vector<float>& minD, const std::vector<float>& maxD,
                                                   const std::vector<orbx_keypoint>& keysUn, const uint8_t* desc,
                                                   const std::vector<orbx_keypoint>& keysRightUn, const uint8_t* descRight,
                                                   int nRows, float mbf, float mb)
    {
        const int N = (int)keysUn.size();
        std::vector<uint8_t> res(iL.size(), 0), active(N, 0);
        if (minD.size() != iL.size() || maxD.size() != iL.size()) throw std::invalid_argument("one window per index");
        std::vector<float> mn(N, 0.0f), mx(N, 0.0f);
        std::vector<size_t> taken;
        for (size_t k = 0; k < iL.size(); k++) {
            const int i = iL[k];
            if (i < 0 || i >= N || !has_mappoint[i] || mvStereoMatched[i]) continue;     // :1182-1186
            mvStereoMatched[i] = 1;
            active[i] = 1;
            mn[i] = minD[k];
            mx[i] = maxD[k];
            taken.push_back(k);
        }
        if (taken.empty()) return res;
        int nmatched = 0;
        check(ex, orbx_stereo_match_ex(ex.context(), keysUn.data(), reinterpret_cast<const uint32_t*>(desc), N,
                                       keysRightUn.data(), reinterpret_cast<const uint32_t*>(descRight),
                                       (int)keysRightUn.size(), nRows, mbf, mb, 0.0f, active.data(), mn.data(), mx.data(),
                                       thOrbDist, 0, listed_dist.data(), best_idx.data(), best_dist.data(),
                                       mvuRight.data(), mvDepth.data(), &nmatched));
        for (size_t k : taken) res[k] = listed_dist[iL[k]] >= 0;
        return res;
    }

private:
    static void check(ORBextractor& ex, int rc)
    {
        if (rc != ORBX_OK) throw std::runtime_error(std::string("liborbx: ") + orbx_last_error(ex.context()));
    }
};

// The grid constants of a Frame (src/Frame.cc:129-135): mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv
struct FrameGrid {
    float mnMinX = 0, mnMinY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
};

// What ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) reads of each map point
// (src/ORBmatcher.cc:161-184), as parallel arrays of one length
struct MapPointsInView {
    std::vector<uint8_t> mbTrackInView, isBad;
    std::vector<float> mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos;
    std::vector<int32_t> mnTrackScaleLevel, Observations;
    std::vector<uint8_t> descriptors;     // 32 bytes per map point (GetDescriptor())
};

// pMP->mvMatchCandidates of every map point (include/MapPoint.h, filled by ORBmatcher::GetCandidates) with the
// descriptor distance of every candidate: the list of map point i is [cand_off[i], cand_off[i + 1])
struct MatchCandidateTable {
    std::vector<int32_t> cand_off, cand_idx;
    std::vector<uint16_t> cand_dist;
    size_t size(size_t i) const { return (size_t)(cand_off[i + 1] - cand_off[i]); }
};

// A DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) flattened in the map's order: node ids ascending,
// node.size() + 1 offsets, the feature indices of every node as stored
struct FeatureVector {
    std::vector<int32_t> node, off{0}, idx;
    void add(int32_t node_id, const std::vector<unsigned int>& features)     // call in ascending order of node_id
    {
        node.push_back(node_id);
        idx.insert(idx.end(), features.begin(), features.end());
        off.push_back((int32_t)idx.size());
    }
};

// Map points as the searches of the mapping and loop-closing threads see them once projected: visible = every test before
// the window search passed, (u, v) = the projection, ur = u - bf / z (Fuse on a stereo keyframe only), level = PredictScale
struct ProjectedPoints {
    std::vector<uint8_t> visible;
    std::vector<float> u, v, ur;
    std::vector<int32_t> level;
    std::vector<uint8_t> descriptors;     // 32 bytes per map point
    bool consistent() const
    {
        const size_t M = visible.size();
        return u.size() == M && v.size() == M && level.size() == M && descriptors.size() == 32 * M && (ur.empty() || ur.size() == M);
    }
};

// What ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, numVisibleMpt) reads of each map point of
// the last frame after projecting it (src/ORBmatcher.cc:1478-1510): visible = non-null, not an outlier, invzc >= 0
// and (u, v) inside the image bounds; octave / angle = LastFrame.mvKeys[i]; parallel arrays of one length
struct LastFramePoints {
    std::vector<uint8_t> visible;
    std::vector<float> u, v, invzc, angle;
    std::vector<int32_t> octave, Observations;
    std::vector<uint8_t> descriptors;     // 32 bytes per map point (GetDescriptor())
};

// The searches of ORB_SLAM2::ORBmatcher whose queries exclude what earlier queries claimed, with the
// reference's constructor arguments (include/ORBmatcher.h: ORBmatcher(float nnratio = 0.6, bool checkOri = true)).
// Candidate lists and distances come from the extractor's device; the sequential part runs inside the library.
class ORBmatcher {
public:
    static constexpr int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:37-39
    explicit ORBmatcher(ORBextractor& ex, float nnratio = 0.6f, bool checkOri = true)
        : ex_(ex), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), src/ORBmatcher.cc:520-633.
    // keys / descriptors = mvKeysUn / mDescriptors of the two frames, grid2 = F2's; vbPrevMatched holds (x, y) per
    // keypoint of F1 and is updated in place; returns nmatches.
    int SearchForInitialization(const std::vector<orbx_keypoint>& keysUn1, const uint8_t* desc1,
                                const std::vector<orbx_keypoint>& keysUn2, const uint8_t* desc2, const FrameGrid& grid2,
                                std::vector<float>& vbPrevMatched, std::vector<int32_t>& vnMatches12, int windowSize = 10)
    {
        if (vbPrevMatched.size() != 2 * keysUn1.size()) throw std::invalid_argument("vbPrevMatched needs (x, y) per keypoint of F1");
        vnMatches12.assign(keysUn1.size(), -1);
        int nmatches = 0;
        check(orbx_search_for_initialization(ex_.context(), keysUn1.data(), reinterpret_cast<const uint32_t*>(desc1),
                                             (int)keysUn1.size(), keysUn2.data(), reinterpret_cast<const uint32_t*>(desc2),
                                             (int)keysUn2.size(), grid2.mnMinX, grid2.mnMinY, grid2.mfGridElementWidthInv,
                                             grid2.mfGridElementHeightInv, vbPrevMatched.data(), windowSize, mfNNratio,
                                             mbCheckOrientation ? 1 : 0, vnMatches12.data(), &nmatches));
        return nmatches;
    }

    // SearchByProjection(F, vpMapPoints, th), src/ORBmatcher.cc:155-241.  frame_obs[i] = Observations() of
    // F.mvpMapPoints[i] or -1 for a null pointer (in/out); frame_match[i] = index of the map point this call assigned
    // to keypoint i (set to -1 beforehand by the caller); mvpMatchScore as in the reference.  Returns nmatches.
    int SearchByProjection(const std::vector<orbx_keypoint>& keysUn, const uint8_t* desc, const std::vector<float>* mvuRight,
                           const FrameGrid& grid, const MapPointsInView& mp, float th, std::vector<int32_t>& frame_obs,
                           std::vector<int32_t>& frame_match, std::vector<int32_t>& mvpMatchScore)
    {
        const size_t N = keysUn.size(), M = mp.mbTrackInView.size();
        check_lengths(N, M, mp, mvuRight, frame_obs, frame_match, mvpMatchScore);
        int nmatches = 0;
        check(orbx_search_by_projection(ex_.context(), keysUn.data(), reinterpret_cast<const uint32_t*>(desc),
                                        mvuRight ? mvuRight->data() : nullptr, (int)N, grid.mnMinX, grid.mnMinY,
                                        grid.mfGridElementWidthInv, grid.mfGridElementHeightInv, mp.mbTrackInView.data(),
                                        mp.isBad.empty() ? nullptr : mp.isBad.data(), mp.mTrackProjX.data(),
                                        mp.mTrackProjY.data(), mp.mTrackProjXR.empty() ? nullptr : mp.mTrackProjXR.data(),
                                        mp.mnTrackScaleLevel.data(), mp.mTrackViewCos.data(),
                                        reinterpret_cast<const uint32_t*>(mp.descriptors.data()), mp.Observations.data(),
                                        (int)M, th, mfNNratio, frame_obs.data(), frame_match.data(), mvpMatchScore.data(),
                                        &nmatches));
        return nmatches;
    }

    // SearchByProjection_Budget(F, vpMapPoints, th, time_constr), src/ORBmatcher.cc:45-153: the call above under a wall
    // clock (milliseconds since the call was entered).  stopped_at = the map point after which the loop ended, or M.
    int SearchByProjection_Budget(const std::vector<orbx_keypoint>& keysUn, const uint8_t* desc, const std::vector<float>* mvuRight,
                                  const FrameGrid& grid, const MapPointsInView& mp, float th, double time_constr,
                                  std::vector<int32_t>& frame_obs, std::vector<int32_t>& frame_match,
                                  std::vector<int32_t>& mvpMatchScore, int* stopped_at = nullptr)
    {
        const size_t N = keysUn.size(), M = mp.mbTrackInView.size();
        check_lengths(N, M, mp, mvuRight, frame_obs, frame_match, mvpMatchScore);
        int nmatches = 0, stop = 0;
        check(orbx_search_by_projection_budget(ex_.context(), keysUn.data(), reinterpret_cast<const uint32_t*>(desc),
                                               mvuRight ? mvuRight->data() : nullptr, (int)N, grid.mnMinX, grid.mnMinY,
                                               grid.mfGridElementWidthInv, grid.mfGridElementHeightInv, mp.mbTrackInView.data(),
                                               mp.isBad.empty() ? nullptr : mp.isBad.data(), mp.mTrackProjX.data(),
                                               mp.mTrackProjY.data(), mp.mTrackProjXR.empty() ? nullptr : mp.mTrackProjXR.data(),
                                               mp.mnTrackScaleLevel.data(), mp.mTrackViewCos.data(),
                                               reinterpret_cast<const uint32_t*>(mp.descriptors.data()), mp.Observations.data(),
                                               (int)M, th, mfNNratio, time_constr, frame_obs.data(), frame_match.data(),
                                               mvpMatchScore.data(), &nmatches, &stop));
        if (stopped_at) *stopped_at = stop;
        return nmatches;
    }

    // GetCandidates(F, pMP, th), include/ORBmatcher.h:151-171, for every map point at once (the loop of
    // src/Observability.cc:1309-1329), the distances of all candidates included
    MatchCandidateTable GetCandidates(const std::vector<orbx_keypoint>& keysUn, const uint8_t* desc, const FrameGrid& grid,
                                      const MapPointsInView& mp, float th)
    {
        const size_t N = keysUn.size(), M = mp.mbTrackInView.size();
        if (mp.mTrackProjX.size() != M || mp.mTrackProjY.size() != M || mp.mnTrackScaleLevel.size() != M ||
            mp.mTrackViewCos.size() != M || mp.descriptors.size() != 32 * M || (!mp.isBad.empty() && mp.isBad.size() != M))
            throw std::invalid_argument("array lengths do not agree");
        MatchCandidateTable t;
        t.cand_off.assign(M + 1, 0);
        int total = 0;
        for (int pass = 0; pass < 2; pass++) {
            const int rc = orbx_projection_candidates(
                ex_.context(), keysUn.data(), reinterpret_cast<const uint32_t*>(desc), (int)N, grid.mnMinX, grid.mnMinY,
                grid.mfGridElementWidthInv, grid.mfGridElementHeightInv, mp.mbTrackInView.data(),
                mp.isBad.empty() ? nullptr : mp.isBad.data(), mp.mTrackProjX.data(), mp.mTrackProjY.data(),
                mp.mnTrackScaleLevel.data(), mp.mTrackViewCos.data(), reinterpret_cast<const uint32_t*>(mp.descriptors.data()),
                (int)M, th, t.cand_off.data(), t.cand_idx.data(), t.cand_dist.data(), (int)t.cand_idx.size(), &total);
            if (rc == ORBX_ERR_CAPACITY && pass == 0) {       // the first pass asked for the size
                t.cand_idx.resize(total);
                t.cand_dist.resize(total);
                continue;
            }
            check(rc);
            break;
        }
        return t;
    }

    // MatchCandidates(F, pMP, th), include/ORBmatcher.h:173-252, = SearchByProjection_OnePoint(F, pMP, th), :68-149, on the
    // table GetCandidates made: map point i against the frame's assignment as it stands.  Returns bestIdx or -1.
    int MatchCandidates(const std::vector<orbx_keypoint>& keysUn, const std::vector<float>* mvuRight,
                        const MatchCandidateTable& t, const MapPointsInView& mp, size_t i, float th,
                        std::vector<int32_t>& frame_obs, std::vector<int32_t>& frame_match, std::vector<int32_t>& mvpMatchScore)
    {
        const size_t N = keysUn.size();
        if (i + 1 >= t.cand_off.size() || frame_obs.size() != N || frame_match.size() != N || mvpMatchScore.size() != N ||
            (mvuRight && mvuRight->size() != N))
            throw std::invalid_argument("array lengths do not agree");
        int best = -1;
        const int a = t.cand_off[i];
        check(orbx_match_candidates(ex_.context(), keysUn.data(), mvuRight ? mvuRight->data() : nullptr, (int)N,
                                    t.cand_idx.data() + a, t.cand_dist.data() + a, (int)t.size(i), mp.mbTrackInView[i],
                                    mp.isBad.empty() ? 0 : mp.isBad[i], mp.mTrackProjXR.empty() ? 0.0f : mp.mTrackProjXR[i],
                                    mp.mnTrackScaleLevel[i], mp.mTrackViewCos[i], th, mfNNratio, (int)i, mp.Observations[i],
                                    frame_obs.data(), frame_match.data(), mvpMatchScore.data(), &best));
        return best;
    }
    int SearchByProjection_OnePoint(const std::vector<orbx_keypoint>& keysUn, const std::vector<float>* mvuRight,
                                    const MatchCandidateTable& t, const MapPointsInView& mp, size_t i, float th,
                                    std::vector<int32_t>& frame_obs, std::vector<int32_t>& frame_match,
                                    std::vector<int32_t>& mvpMatchScore)
    {
        return MatchCandidates(keysUn, mvuRight, t, mp, i, th, frame_obs, frame_match, mvpMatchScore);
    }

    // SearchByBoW(pKF, F, vpMapPointMatches), src/ORBmatcher.cc:270-404.  kfHasPoint[i] = pKF's map point i exists and is
    // not bad; vpMapPointMatches[i] = index of the keyframe keypoint whose map point the frame keypoint i receives, -1 = NULL.
    int SearchByBoW(const std::vector<orbx_keypoint>& kfKeysUn, const uint8_t* kfDesc, const std::vector<uint8_t>& kfHasPoint,
                    const FeatureVector& kfFeatVec, const std::vector<orbx_keypoint>& fKeys, const uint8_t* fDesc,
                    const FeatureVector& fFeatVec, std::vector<int32_t>& vpMapPointMatches)
    {
        if (kfHasPoint.size() != kfKeysUn.size()) throw std::invalid_argument("array lengths do not agree");
        vpMapPointMatches.assign(fKeys.size(), -1);
        int nmatches = 0;
        check(orbx_search_by_bow_frame(ex_.context(), kfKeysUn.data(), reinterpret_cast<const uint32_t*>(kfDesc), kfHasPoint.data(),
                                       (int)kfKeysUn.size(), kfFeatVec.node.data(), kfFeatVec.off.data(), kfFeatVec.idx.data(),
                                       (int)kfFeatVec.node.size(), fKeys.data(), reinterpret_cast<const uint32_t*>(fDesc),
                                       (int)fKeys.size(), fFeatVec.node.data(), fFeatVec.off.data(), fFeatVec.idx.data(),
                                       (int)fFeatVec.node.size(), mfNNratio, mbCheckOrientation ? 1 : 0,
                                       vpMapPointMatches.data(), &nmatches));
        return nmatches;
    }

    // SearchByBoW(pKF1, pKF2, vpMatches12), src/ORBmatcher.cc:635-768.  vpMatches12[i] = index of the keypoint of pKF2 whose
    // map point keypoint i of pKF1 receives, -1 = NULL.
    int SearchByBoW(const std::vector<orbx_keypoint>& keysUn1, const uint8_t* desc1, const std::vector<uint8_t>& hasPoint1,
                    const FeatureVector& featVec1, const std::vector<orbx_keypoint>& keysUn2, const uint8_t* desc2,
                    const std::vector<uint8_t>& hasPoint2, const FeatureVector& featVec2, std::vector<int32_t>& vpMatches12)
    {
        if (hasPoint1.size() != keysUn1.size() || hasPoint2.size() != keysUn2.size())
            throw std::invalid_argument("array lengths do not agree");
        vpMatches12.assign(keysUn1.size(), -1);
        int nmatches = 0;
        check(orbx_search_by_bow_keyframes(ex_.context(), keysUn1.data(), reinterpret_cast<const uint32_t*>(desc1), hasPoint1.data(),
                                           (int)keysUn1.size(), featVec1.node.data(), featVec1.off.data(), featVec1.idx.data(),
                                           (int)featVec1.node.size(), keysUn2.data(), reinterpret_cast<const uint32_t*>(desc2),
                                           hasPoint2.data(), (int)keysUn2.size(), featVec2.node.data(), featVec2.off.data(),
                                           featVec2.idx.data(), (int)featVec2.node.size(), mfNNratio, mbCheckOrientation ? 1 : 0,
                                           vpMatches12.data(), &nmatches));
        return nmatches;
    }

    // SearchByProjection(CurrentFrame, LastFrame, th, bMono, numVisibleMpt), src/ORBmatcher.cc:1440-1593, from the
    // projected points on: bForward / bBackward as computed at :1468-1469 (both false for a monocular frame), mbf =
    // CurrentFrame.mbf, mvuRight = nullptr for a monocular frame.  frame_obs / frame_match as in the call above; the
    // rotation check sets both to -1 for the matches it removes.  Returns nmatches.
    int SearchByProjection(const std::vector<orbx_keypoint>& keysUn, const uint8_t* desc, const std::vector<float>* mvuRight,
                           const FrameGrid& grid, const LastFramePoints& last, float th, float mbf, bool bForward,
                           bool bBackward, std::vector<int32_t>& frame_obs, std::vector<int32_t>& frame_match)
    {
        const size_t N = keysUn.size(), M = last.visible.size();
        if (frame_obs.size() != N || frame_match.size() != N || last.u.size() != M || last.v.size() != M ||
            last.invzc.size() != M || last.angle.size() != M || last.octave.size() != M || last.Observations.size() != M ||
            last.descriptors.size() != 32 * M || (mvuRight && mvuRight->size() != N))
            throw std::invalid_argument("array lengths do not agree");
        int nmatches = 0;
        check(orbx_search_by_projection_last(ex_.context(), keysUn.data(), reinterpret_cast<const uint32_t*>(desc),
                                             mvuRight ? mvuRight->data() : nullptr, (int)N, grid.mnMinX, grid.mnMinY,
                                             grid.mfGridElementWidthInv, grid.mfGridElementHeightInv, last.visible.data(),
                                             last.u.data(), last.v.data(), last.invzc.data(), last.octave.data(),
                                             last.angle.data(), reinterpret_cast<const uint32_t*>(last.descriptors.data()),
                                             last.Observations.data(), (int)M, th, mbf, bForward ? 1 : 0, bBackward ? 1 : 0,
                                             mbCheckOrientation ? 1 : 0, frame_obs.data(), frame_match.data(), &nmatches));
        return nmatches;
    }

    // SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1595-1721 (relocalisation), from the
    // projected points on: `kf` holds, per keypoint of the keyframe, visible / u / v, octave = the predicted level, angle =
    // pKF->mvKeysUn[i].angle, the map point's descriptor and Observations (invzc is not read).  frame_obs[i2] >= 0 = the
    // keypoint of the current frame holds a map point.
    int SearchByProjection(const std::vector<orbx_keypoint>& keysUn, const uint8_t* desc, const FrameGrid& grid,
                           const LastFramePoints& kf, float th, int ORBdist, std::vector<int32_t>& frame_obs,
                           std::vector<int32_t>& frame_match)
    {
        const size_t N = keysUn.size(), M = kf.visible.size();
        if (frame_obs.size() != N || frame_match.size() != N || kf.u.size() != M || kf.v.size() != M || kf.angle.size() != M ||
            kf.octave.size() != M || kf.Observations.size() != M || kf.descriptors.size() != 32 * M)
            throw std::invalid_argument("array lengths do not agree");
        int nmatches = 0;
        check(orbx_search_by_projection_keyframe(ex_.context(), keysUn.data(), reinterpret_cast<const uint32_t*>(desc), (int)N,
                                                 grid.mnMinX, grid.mnMinY, grid.mfGridElementWidthInv, grid.mfGridElementHeightInv,
                                                 kf.visible.data(), kf.u.data(), kf.v.data(), kf.octave.data(), kf.angle.data(),
                                                 reinterpret_cast<const uint32_t*>(kf.descriptors.data()), kf.Observations.data(),
                                                 (int)M, th, ORBdist, mbCheckOrientation ? 1 : 0, frame_obs.data(),
                                                 frame_match.data(), &nmatches));
        return nmatches;
    }

    // SearchByProjection(pKF, Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:406-518, from the projected points on.
    // kfMatched[idx] >= 0 = vpMatched[idx] is set (in/out); kfMatch[idx] = the point of `pts` this call stored there.
    int SearchByProjection(const std::vector<orbx_keypoint>& kfKeysUn, const uint8_t* kfDesc, const FrameGrid& grid,
                           const ProjectedPoints& pts, int th, std::vector<int32_t>& kfMatched, std::vector<int32_t>& kfMatch)
    {
        if (!pts.consistent() || kfMatched.size() != kfKeysUn.size() || kfMatch.size() != kfKeysUn.size())
            throw std::invalid_argument("array lengths do not agree");
        int nmatches = 0;
        check(orbx_search_by_projection_sim3(ex_.context(), kfKeysUn.data(), reinterpret_cast<const uint32_t*>(kfDesc),
                                             (int)kfKeysUn.size(), grid.mnMinX, grid.mnMinY, grid.mfGridElementWidthInv,
                                             grid.mfGridElementHeightInv, pts.visible.data(), pts.u.data(), pts.v.data(),
                                             pts.level.data(), reinterpret_cast<const uint32_t*>(pts.descriptors.data()),
                                             (int)pts.visible.size(), th, kfMatched.data(), kfMatch.data(), &nmatches));
        return nmatches;
    }

    // The search of Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:937-1087, reprojectionGate = true; mvuRight = nullptr for a
    // monocular keyframe) and of Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:1089-1212, reprojectionGate = false):
    // bestIdx[i] = the keypoint map point i fuses with (-1: none), bestDist[i] = its distance.  Returns nFused; the
    // Replace / AddObservation step stays with the caller.
    int Fuse(const std::vector<orbx_keypoint>& kfKeysUn, const uint8_t* kfDesc, const std::vector<float>* mvuRight,
             const FrameGrid& grid, const ProjectedPoints& pts, float th, bool reprojectionGate, std::vector<int32_t>& bestIdx,
             std::vector<int32_t>& bestDist)
    {
        if (!pts.consistent() || (mvuRight && (mvuRight->size() != kfKeysUn.size() || pts.ur.size() != pts.visible.size())))
            throw std::invalid_argument("array lengths do not agree");
        bestIdx.assign(pts.visible.size(), -1);
        bestDist.assign(pts.visible.size(), 256);
        int nFused = 0;
        check(orbx_fuse_search(ex_.context(), kfKeysUn.data(), reinterpret_cast<const uint32_t*>(kfDesc),
                               mvuRight ? mvuRight->data() : nullptr, (int)kfKeysUn.size(), grid.mnMinX, grid.mnMinY,
                               grid.mfGridElementWidthInv, grid.mfGridElementHeightInv, pts.visible.data(), pts.u.data(),
                               pts.v.data(), mvuRight ? pts.ur.data() : nullptr, pts.level.data(),
                               reinterpret_cast<const uint32_t*>(pts.descriptors.data()), (int)pts.visible.size(), th,
                               reprojectionGate ? 1 : 0, bestIdx.data(), bestDist.data(), &nFused));
        return nFused;
    }

    // SearchBySim3, src/ORBmatcher.cc:1214-1438: pts1 = pKF1's map points (one entry per keypoint of pKF1) projected into
    // pKF2, pts2 the other way round.  vpMatches12[i1] = the keypoint of pKF2 chosen mutually, -1 elsewhere.
    int SearchBySim3(const std::vector<orbx_keypoint>& keysUn1, const uint8_t* desc1, const std::vector<orbx_keypoint>& keysUn2,
                     const uint8_t* desc2, const FrameGrid& grid, const ProjectedPoints& pts1, const ProjectedPoints& pts2,
                     float th, std::vector<int32_t>& vpMatches12)
    {
        if (!pts1.consistent() || !pts2.consistent() || pts1.visible.size() != keysUn1.size() || pts2.visible.size() != keysUn2.size())
            throw std::invalid_argument("array lengths do not agree");
        vpMatches12.assign(keysUn1.size(), -1);
        int nFound = 0;
        check(orbx_search_by_sim3(ex_.context(), keysUn1.data(), reinterpret_cast<const uint32_t*>(desc1), (int)keysUn1.size(),
                                  keysUn2.data(), reinterpret_cast<const uint32_t*>(desc2), (int)keysUn2.size(), grid.mnMinX,
                                  grid.mnMinY, grid.mfGridElementWidthInv, grid.mfGridElementHeightInv, pts1.visible.data(),
                                  pts1.u.data(), pts1.v.data(), pts1.level.data(),
                                  reinterpret_cast<const uint32_t*>(pts1.descriptors.data()), pts2.visible.data(), pts2.u.data(),
                                  pts2.v.data(), pts2.level.data(), reinterpret_cast<const uint32_t*>(pts2.descriptors.data()), th,
                                  vpMatches12.data(), &nFound));
        return nFound;
    }

    // SearchForTriangulation, src/ORBmatcher.cc:770-935.  hasPoint = the keypoint already holds a map point, stereo =
    // mvuRight >= 0, F12 row-major, (ex, ey) = the epipole in the second image.  vMatches12[i1] = partner in pKF2 or -1.
    int SearchForTriangulation(const std::vector<orbx_keypoint>& keysUn1, const uint8_t* desc1, const std::vector<uint8_t>& hasPoint1,
                               const std::vector<uint8_t>& stereo1, const FeatureVector& featVec1,
                               const std::vector<orbx_keypoint>& keysUn2, const uint8_t* desc2, const std::vector<uint8_t>& hasPoint2,
                               const std::vector<uint8_t>& stereo2, const FeatureVector& featVec2, const float F12[9], float ex,
                               float ey, bool bOnlyStereo, std::vector<int32_t>& vMatches12)
    {
        if (hasPoint1.size() != keysUn1.size() || stereo1.size() != keysUn1.size() || hasPoint2.size() != keysUn2.size() ||
            stereo2.size() != keysUn2.size())
            throw std::invalid_argument("array lengths do not agree");
        vMatches12.assign(keysUn1.size(), -1);
        int nmatches = 0;
        check(orbx_search_for_triangulation(ex_.context(), keysUn1.data(), reinterpret_cast<const uint32_t*>(desc1), hasPoint1.data(),
                                            stereo1.data(), (int)keysUn1.size(), featVec1.node.data(), featVec1.off.data(),
                                            featVec1.idx.data(), (int)featVec1.node.size(), keysUn2.data(),
                                            reinterpret_cast<const uint32_t*>(desc2), hasPoint2.data(), stereo2.data(),
                                            (int)keysUn2.size(), featVec2.node.data(), featVec2.off.data(), featVec2.idx.data(),
                                            (int)featVec2.node.size(), F12, ex, ey, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                            vMatches12.data(), &nmatches));
        return nmatches;
    }

private:
    void check(int rc)
    {
        if (rc != ORBX_OK) throw std::runtime_error(std::string("liborbx: ") + orbx_last_error(ex_.context()));
    }
    static void check_lengths(size_t N, size_t M, const MapPointsInView& mp, const std::vector<float>* mvuRight,
                              const std::vector<int32_t>& frame_obs, const std::vector<int32_t>& frame_match,
                              const std::vector<int32_t>& mvpMatchScore)
    {
        if (frame_obs.size() != N || frame_match.size() != N || mvpMatchScore.size() != N ||
            mp.mTrackProjX.size() != M || mp.mTrackProjY.size() != M || mp.mnTrackScaleLevel.size() != M ||
            mp.mTrackViewCos.size() != M || mp.Observations.size() != M || mp.descriptors.size() != 32 * M ||
            (mvuRight && (mvuRight->size() != N || mp.mTrackProjXR.size() != M)) || (!mp.isBad.empty() && mp.isBad.size() != M))
            throw std::invalid_argument("array lengths do not agree");
    }
    ORBextractor& ex_;

public:
    float mfNNratio;
    bool mbCheckOrientation;
};

// ORBmatcher::DescriptorDistance (include/ORBmatcher.h:48) on raw 32-byte rows
inline int DescriptorDistance(const uint8_t* a, const uint8_t* b)
{
    uint32_t wa[8], wb[8];
    for (int i = 0; i < 8; i++) {
        wa[i] = (uint32_t)a[4 * i] | ((uint32_t)a[4 * i + 1] << 8) | ((uint32_t)a[4 * i + 2] << 16) | ((uint32_t)a[4 * i + 3] << 24);
        wb[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    }
    return orbx_hamming(wa, wb);
}

} // namespace orbx
