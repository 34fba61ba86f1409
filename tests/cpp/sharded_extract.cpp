// orbx::ShardedExtractor (include/orbx_host.hpp): a batch cut into contiguous frame ranges over several
// contexts, one host thread each, against a single context running the whole batch; then
// orbx::ORBmatcher::SearchForInitialization between the first two frames, dumped for the caller to compare.
//   sharded_extract <raw_u8_file> <w> <h> <nframes> <devices, e.g. 0,0,0> <out_file>
// raw file: nframes frames back to back.  Prints "sharded ok: ..." when sharded == single for every frame.
// out_file: int32 n1, nmatches; n1 x int32 vnMatches12; n1 x 2 float vbPrevMatched; then the last-frame projection
// search of frame 0's keypoints in frame 1: int32 n2, nmatches; n2 x int32 frame_match; n2 x int32 frame_obs; then
// one-point matching of frame 0's keypoints as map points, last first: int32 ncandidates, nmatches; n2 x int32
// frame_match; n2 x int32 mvpMatchScore
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "orbx_host.hpp"

int main(int argc, char** argv)
{
    if (argc != 7) { std::fprintf(stderr, "usage: %s raw w h nframes devices out\n", argv[0]); return 2; }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), B = std::atoi(argv[4]);
    std::vector<int> devices;
    for (const char* p = argv[5]; *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') p++; if (*p) p++; }
    std::vector<uint8_t> img((size_t)w * h * B);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) { std::fprintf(stderr, "cannot read frames\n"); return 2; }
    std::fclose(f);
    std::vector<const uint8_t*> ptrs(B);
    for (int i = 0; i < B; i++) ptrs[i] = img.data() + (size_t)i * w * h;
    try {
        orbx::ShardedExtractor sh(devices, 1000, 1.2f, 8, 20, 7);
        orbx::ORBextractor one(1000, 1.2f, 8, 20, 7, devices[0]);
        const int cap = sh.max_keypoints(w, h);
        std::vector<orbx_keypoint> k1((size_t)B * cap), k2((size_t)B * cap);
        std::vector<uint8_t> d1((size_t)B * cap * 32), d2((size_t)B * cap * 32);
        std::vector<int32_t> n1(B, -1), n2(B, -2);
        for (int rep = 0; rep < 3; rep++) {          // repeated calls reuse the shards' buffers
            std::fill(n1.begin(), n1.end(), -1);
            sh.ExtractBatch(ptrs.data(), B, w, h, w, k1.data(), d1.data(), cap, n1.data());
        }
        if (orbx_extract_batch(one.context(), ptrs.data(), B, w, h, w, k2.data(), d2.data(), cap, n2.data()) != ORBX_OK) {
            std::fprintf(stderr, "single: %s\n", orbx_last_error(one.context()));
            return 1;
        }
        long long total = 0;
        for (int i = 0; i < B; i++) {
            if (n1[i] != n2[i] || n1[i] <= 0) { std::fprintf(stderr, "frame %d: %d vs %d keypoints\n", i, n1[i], n2[i]); return 1; }
            if (std::memcmp(&k1[(size_t)i * cap], &k2[(size_t)i * cap], (size_t)n1[i] * sizeof(orbx_keypoint)) != 0 ||
                std::memcmp(&d1[(size_t)i * cap * 32], &d2[(size_t)i * cap * 32], (size_t)n1[i] * 32) != 0) {
                std::fprintf(stderr, "frame %d differs\n", i);
                return 1;
            }
            total += n1[i];
        }
        // a batch smaller than the number of shards: the empty shards stay idle
        if (B >= 1) {
            std::fill(n1.begin(), n1.end(), -1);
            sh.ExtractBatch(ptrs.data(), 1, w, h, w, k1.data(), d1.data(), cap, n1.data());
            if (n1[0] != n2[0]) { std::fprintf(stderr, "one-frame batch: %d vs %d\n", n1[0], n2[0]); return 1; }
        }
        std::printf("sharded ok: %d frames over %d shards, %lld keypoints\n", B, sh.shards(), total);
        if (B >= 2) {
            orbx::ORBmatcher matcher(one, 0.9f, true);           // Tracking.cc:1175
            std::vector<orbx_keypoint> a(k2.begin(), k2.begin() + n2[0]), b(k2.begin() + cap, k2.begin() + cap + n2[1]);
            std::vector<float> prev(2 * a.size());
            for (size_t i = 0; i < a.size(); i++) { prev[2 * i] = a[i].x; prev[2 * i + 1] = a[i].y; }
            std::vector<int32_t> m12;
            orbx::FrameGrid grid;
            grid.mfGridElementWidthInv = 64.0f / (float)w;       // src/Frame.cc:134-135 with mnMinX = mnMinY = 0
            grid.mfGridElementHeightInv = 48.0f / (float)h;
            const int nm = matcher.SearchForInitialization(a, d2.data(), b, d2.data() + (size_t)cap * 32, grid, prev, m12, 100);
            FILE* o = std::fopen(argv[6], "wb");
            int32_t hdr[2] = {(int32_t)a.size(), nm};
            std::fwrite(hdr, 4, 2, o);
            std::fwrite(m12.data(), 4, m12.size(), o);
            std::fwrite(prev.data(), 4, prev.size(), o);
            // frame-to-frame search (src/Tracking.cc:2204): the first frame's keypoints as the last frame's map points,
            // projected where they were seen, monocular (no right coordinate, no forward / backward), th = 15
            orbx::LastFramePoints last;
            last.visible.assign(a.size(), 1);
            last.invzc.assign(a.size(), 1.0f);
            last.Observations.assign(a.size(), 1);
            for (const orbx_keypoint& k : a) {
                last.u.push_back(k.x); last.v.push_back(k.y); last.angle.push_back(k.angle); last.octave.push_back(k.octave);
            }
            last.descriptors.assign(d2.begin(), d2.begin() + 32 * a.size());
            std::vector<int32_t> fobs(b.size(), -1), fmatch(b.size(), -1);
            const int nm2 = matcher.SearchByProjection(b, d2.data() + (size_t)cap * 32, nullptr, grid, last, 15.0f, 0.0f, false,
                                                       false, fobs, fmatch);
            int32_t hdr2[2] = {(int32_t)b.size(), nm2};
            std::fwrite(hdr2, 4, 2, o);
            std::fwrite(fmatch.data(), 4, fmatch.size(), o);
            std::fwrite(fobs.data(), 4, fobs.size(), o);
            // active matching (src/Observability.cc:1309-1329, :1404-1413): candidate lists of all map points at once, then
            // one point at a time, here from the last map point to the first
            orbx::MapPointsInView mp;
            mp.mbTrackInView.assign(a.size(), 1);
            mp.Observations.assign(a.size(), 1);
            for (size_t i = 0; i < a.size(); i++) {
                mp.mTrackProjX.push_back(a[i].x); mp.mTrackProjY.push_back(a[i].y); mp.mTrackProjXR.push_back(0.0f);
                mp.mnTrackScaleLevel.push_back(a[i].octave); mp.mTrackViewCos.push_back(i % 2 ? 0.9995f : 0.9f);
            }
            mp.descriptors.assign(d2.begin(), d2.begin() + 32 * a.size());
            orbx::ORBmatcher active(one, 0.8f, true);
            const orbx::MatchCandidateTable table = active.GetCandidates(b, d2.data() + (size_t)cap * 32, grid, mp, 3.0f);
            std::vector<int32_t> aobs(b.size(), -1), amatch(b.size(), -1), ascore(b.size(), -1);
            int nm3 = 0;
            for (size_t i = a.size(); i-- > 0;)
                nm3 += active.SearchByProjection_OnePoint(b, nullptr, table, mp, i, 3.0f, aobs, amatch, ascore) >= 0;
            int32_t hdr3[2] = {(int32_t)table.cand_idx.size(), nm3};
            std::fwrite(hdr3, 4, 2, o);
            std::fwrite(amatch.data(), 4, amatch.size(), o);
            std::fwrite(ascore.data(), 4, ascore.size(), o);
            std::fclose(o);
            std::printf("SearchForInitialization: %d matches, SearchByProjection(last frame): %d, one point at a time: %d\n", nm,
                        nm2, nm3);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
