// The searches of orbx::ORBmatcher (include/orbx_host.hpp) that take projected map points or feature vectors, called
// from C++ on the features of two frames and dumped for the caller to compare with the CPU restatement:
//   matcher_calls <raw_u8_file with two frames> <w> <h> <out_file>
// Inputs are derived from the extraction by fixed rules (the test repeats them): feature vector node of a descriptor =
// three bits, bit k = more than 32 of its bits 64 k .. 64 k + 63 are set; flags by index modulo small numbers; every
// keypoint of frame 0 as a map point that projects to its own position in frame 1, and the other way round.
// out_file: a sequence of blocks, each int32 length followed by that many int32 values (see `dump` calls below).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbx_host.hpp"

static FILE* g_out;
static void dump(const std::vector<int32_t>& v)
{
    if (!g_out) return;      // the calls on cut-down feature sets below are made for their memory accesses only
    const int32_t n = (int32_t)v.size();
    std::fwrite(&n, 4, 1, g_out);
    std::fwrite(v.data(), 4, v.size(), g_out);
}
static void dump1(int32_t x) { dump(std::vector<int32_t>{x}); }

static int node_of(const uint8_t* d)
{
    int node = 0;
    for (int k = 0; k < 3; k++) {
        int bits = 0;
        for (int b = 0; b < 8; b++) bits += __builtin_popcount(d[8 * k + b]);
        node = node * 2 + (bits > 32 ? 1 : 0);
    }
    return node;
}

static orbx::FeatureVector feature_vector(const uint8_t* desc, size_t n)
{
    std::vector<std::vector<unsigned int>> by_node(8);
    for (size_t i = 0; i < n; i++) by_node[node_of(desc + 32 * i)].push_back((unsigned int)i);
    orbx::FeatureVector fv;
    for (int node = 0; node < 8; node++)
        if (!by_node[node].empty()) fv.add(node, by_node[node]);
    return fv;
}

static orbx::ProjectedPoints points_of(const std::vector<orbx_keypoint>& k, const std::vector<uint8_t>& d)
{
    orbx::ProjectedPoints p;
    for (size_t i = 0; i < k.size(); i++) {
        p.visible.push_back(i % 5 != 0);
        p.u.push_back(k[i].x);
        p.v.push_back(k[i].y);
        p.ur.push_back(k[i].x - 10.0f);
        p.level.push_back(k[i].octave);
    }
    p.descriptors = d;
    return p;
}

static void calls(orbx::ORBextractor& ex, int w, int h, const std::vector<orbx_keypoint>& k1, const std::vector<uint8_t>& d1,
                  const std::vector<orbx_keypoint>& k2, const std::vector<uint8_t>& d2)
{
        const size_t n1 = k1.size(), n2 = k2.size();
        orbx::FrameGrid grid;
        grid.mfGridElementWidthInv = 64.0f / (float)w;
        grid.mfGridElementHeightInv = 48.0f / (float)h;
        dump(std::vector<int32_t>{(int32_t)n1, (int32_t)n2});
        const orbx::FeatureVector fv1 = feature_vector(d1.data(), n1), fv2 = feature_vector(d2.data(), n2);
        std::vector<uint8_t> has1(n1), has2(n2), held1(n1), held2(n2), st1(n1), st2(n2);
        for (size_t i = 0; i < n1; i++) { has1[i] = i % 9 != 0; held1[i] = i % 3 == 0; st1[i] = i % 2; }
        for (size_t i = 0; i < n2; i++) { has2[i] = i % 7 != 0; held2[i] = i % 4 == 0; st2[i] = (i / 2) % 2; }
        orbx::ORBmatcher m(ex, 0.7f, true);
        std::vector<int32_t> out;
        // 1, 2: SearchByBoW, keyframe against frame and keyframe against keyframe
        dump1(m.SearchByBoW(k1, d1.data(), has1, fv1, k2, d2.data(), fv2, out));
        dump(out);
        dump1(m.SearchByBoW(k1, d1.data(), has1, fv1, k2, d2.data(), has2, fv2, out));
        dump(out);
        // 3: SearchForTriangulation for a sideways motion (epipolar lines = image rows), epipole near the image centre
        const float F12[9] = {0, 0, 0, 0, 0, -1, 0, 1, 0};
        dump1(m.SearchForTriangulation(k1, d1.data(), held1, st1, fv1, k2, d2.data(), held2, st2, fv2, F12, w * 0.5f + 0.5f,
                                       h * 0.5f, false, out));
        dump(out);
        // 4: frame 0's keypoints as map points seen in frame 1
        const orbx::ProjectedPoints p1 = points_of(k1, d1), p2 = points_of(k2, d2);
        std::vector<int32_t> matched(n2), match(n2, -1);
        for (size_t i = 0; i < n2; i++) matched[i] = i % 4 == 0 ? 1 : -1;
        dump1(m.SearchByProjection(k2, d2.data(), grid, p1, 10, matched, match));
        dump(matched);
        dump(match);
        std::vector<float> ur2(n2);
        for (size_t i = 0; i < n2; i++) ur2[i] = i % 2 ? k2[i].x - 10.0f : -1.0f;
        std::vector<int32_t> bi, bd;
        dump1(m.Fuse(k2, d2.data(), &ur2, grid, p1, 3.0f, true, bi, bd));
        dump(bi);
        dump(bd);
        dump1(m.Fuse(k2, d2.data(), nullptr, grid, p1, 4.0f, false, bi, bd));
        dump(bi);
        dump(bd);
        dump1(m.SearchBySim3(k1, d1.data(), k2, d2.data(), grid, p1, p2, 7.5f, out));
        dump(out);
        // 5: relocalisation search, frame 0 as the keyframe
        orbx::LastFramePoints kf;
        for (size_t i = 0; i < n1; i++) {
            kf.visible.push_back(i % 5 != 0); kf.u.push_back(k1[i].x); kf.v.push_back(k1[i].y); kf.invzc.push_back(1.0f);
            kf.angle.push_back(k1[i].angle); kf.octave.push_back(k1[i].octave); kf.Observations.push_back(1);
        }
        kf.descriptors = d1;
        std::vector<int32_t> fobs(n2), fmatch(n2, -1);
        for (size_t i = 0; i < n2; i++) fobs[i] = i % 6 == 0 ? 0 : -1;
        dump1(m.SearchByProjection(k2, d2.data(), grid, kf, 10.0f, 64, fobs, fmatch));
        dump(fobs);
        dump(fmatch);
        // 6: the budgeted local-map search with a budget that is never reached
        orbx::MapPointsInView mp;
        mp.mbTrackInView.assign(n1, 1);
        mp.Observations.assign(n1, 1);
        for (size_t i = 0; i < n1; i++) {
            mp.mTrackProjX.push_back(k1[i].x); mp.mTrackProjY.push_back(k1[i].y); mp.mTrackProjXR.push_back(k1[i].x - 10.0f);
            mp.mnTrackScaleLevel.push_back(k1[i].octave); mp.mTrackViewCos.push_back(i % 2 ? 0.9995f : 0.9f);
        }
        mp.descriptors = d1;
        std::vector<int32_t> o2(n2, -1), m2(n2, -1), s2(n2, -1);
        int stopped = -1;
        dump1(m.SearchByProjection_Budget(k2, d2.data(), &ur2, grid, mp, 3.0f, 1e12, o2, m2, s2, &stopped));
        dump1(stopped);
        dump(m2);
        dump(s2);
}

int main(int argc, char** argv)
{
    if (argc != 5) { std::fprintf(stderr, "usage: %s raw w h out\n", argv[0]); return 2; }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    std::vector<uint8_t> img((size_t)w * h * 2);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) { std::fprintf(stderr, "cannot read frames\n"); return 2; }
    std::fclose(f);
    try {
        orbx::ORBextractor ex(1000, 1.2f, 8, 20, 7);
        std::vector<orbx_keypoint> k1, k2;
        std::vector<uint8_t> d1, d2;
        const orbx::ImageView none;
        ex(orbx::ImageView(img.data(), w, h, (size_t)w), none, k1, d1);
        ex(orbx::ImageView(img.data() + (size_t)w * h, w, h, (size_t)w), none, k2, d2);
        g_out = std::fopen(argv[4], "wb");
        if (!g_out) { std::fprintf(stderr, "cannot write %s\n", argv[4]); return 2; }
        calls(ex, w, h, k1, d1, k2, d2);
        std::fclose(g_out);
        g_out = nullptr;
        // the same calls on empty, one-element and lopsided feature sets (nothing is dumped: under the sanitizers of
        // tools/host_sanitizers.sh every access outside a buffer is a report)
        const size_t cuts[][2] = {{0, 0}, {0, 5}, {5, 0}, {1, 1}, {3, 700}, {700, 2}};
        for (const auto& c : cuts) {
            const size_t c1 = std::min(c[0], k1.size()), c2 = std::min(c[1], k2.size());
            calls(ex, w, h, std::vector<orbx_keypoint>(k1.begin(), k1.begin() + c1), std::vector<uint8_t>(d1.begin(), d1.begin() + 32 * c1),
                  std::vector<orbx_keypoint>(k2.begin(), k2.begin() + c2), std::vector<uint8_t>(d2.begin(), d2.begin() + 32 * c2));
        }
        const size_t n1 = k1.size(), n2 = k2.size();
        std::printf("matcher calls ok: %zu and %zu keypoints\n", n1, n2);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
