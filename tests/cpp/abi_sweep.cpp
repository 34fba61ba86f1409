// Sweeps entry points of the C ABI that the other C++ programs do not reach, for runs under AddressSanitizer and
// UndefinedBehaviorSanitizer (tools/host_sanitizers.sh).  Results are compared with the CPU restatement elsewhere (the
// Python tests); here the point is that the host side touches no byte outside the caller's buffers, on ordinary,
// empty and ragged sizes.  Every output buffer is allocated at its exact size, so one element too many is a report.
//   abi_sweep yaml <tmpdir>     detection I/O only (needs no GPU): round trip, truncations, random mutations
//   abi_sweep all <tmpdir>      the above and the host-array entry points on device 0
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "orbx.h"

static int g_fail = 0;
static orbx_ctx* g_ctx = nullptr;      // its last error text goes with a failed expectation
#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "%s:%d: %s   [%s]\n", __FILE__, __LINE__, #c, g_ctx ? orbx_last_error(g_ctx) : ""); g_fail++; } } while (0)

static std::vector<orbx_keypoint> random_keypoints(std::mt19937& rng, int n, float w, float h)
{
    std::uniform_real_distribution<float> ux(0.f, w), uy(0.f, h), ua(0.f, 360.f);
    std::uniform_int_distribution<int> uo(0, 7), ur(7, 255);
    std::vector<orbx_keypoint> k((size_t)n);
    for (auto& p : k) {
        p.x = ux(rng); p.y = uy(rng); p.octave = uo(rng); p.size = 31.f * std::pow(1.2f, (float)p.octave);
        p.angle = ua(rng); p.response = (float)ur(rng); p.class_id = -1;
    }
    return k;
}
static std::vector<uint32_t> random_words(std::mt19937& rng, size_t n)
{
    std::vector<uint32_t> v(n);
    for (auto& x : v) x = rng();
    return v;
}

static std::string slurp(const std::string& path)
{
    std::string s;
    if (FILE* f = std::fopen(path.c_str(), "rb")) {
        char buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
        std::fclose(f);
    }
    return s;
}
static void spit(const std::string& path, const std::string& s)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::perror(path.c_str()); std::exit(2); }
    std::fwrite(s.data(), 1, s.size(), f);
    std::fclose(f);
}

// reads `path` into buffers of exactly cap / cap_r entries; any return code is fine, a stray write is not
static int read_into(const std::string& path, int cap, int cap_r, int* n, int* nr)
{
    std::vector<orbx_keypoint> k((size_t)cap), kr((size_t)cap_r);
    std::vector<uint8_t> d((size_t)cap * 32), dr((size_t)cap_r * 32);
    double ts = 0;
    return orbx_detection_read_yaml(path.c_str(), &ts, cap ? k.data() : nullptr, cap ? d.data() : nullptr, cap, n,
                                    cap_r ? kr.data() : nullptr, cap_r ? dr.data() : nullptr, cap_r, nr);
}

static void yaml_sweep(const std::string& dir)
{
    std::mt19937 rng(7);
    const int n = 37, nr = 5;
    auto k = random_keypoints(rng, n, 752.f, 480.f), kr = random_keypoints(rng, nr, 752.f, 480.f);
    std::vector<uint8_t> d((size_t)n * 32), dr((size_t)nr * 32);
    for (auto& b : d) b = (uint8_t)rng();
    for (auto& b : dr) b = (uint8_t)rng();
    const float K[9] = {458.f, 0.f, 367.f, 0.f, 457.f, 248.f, 0.f, 0.f, 1.f};
    for (int legacy = 0; legacy < 2; legacy++) {
        const std::string path = dir + "/det" + std::to_string(legacy) + ".yml";
        EXPECT(orbx_detection_write_yaml(path.c_str(), 3, 2, 1403636579.76, K, k.data(), d.data(), n, kr.data(), dr.data(), nr, legacy) == ORBX_OK);
        int gn = -1, gnr = -1;
        EXPECT(read_into(path, n, nr, &gn, &gnr) == ORBX_OK && gn == n && gnr == nr);
        EXPECT(read_into(path, 0, 0, &gn, &gnr) == ORBX_OK && gn == n && gnr == nr);      // counts only
        EXPECT(read_into(path, n - 1, nr, &gn, &gnr) == ORBX_ERR_CAPACITY);
        EXPECT(read_into(path, n, nr - 1, &gn, &gnr) == ORBX_ERR_CAPACITY);
        const std::string text = slurp(path);
        EXPECT(text.size() > 1000);
        const std::string mut = dir + "/mut.yml";
        // every prefix in steps, dense around the ends of the lists
        for (size_t cut = 0; cut < text.size(); cut += (cut < 400 || cut + 400 > text.size()) ? 1 : 37) {
            spit(mut, text.substr(0, cut));
            if (read_into(mut, n, nr, &gn, &gnr) == ORBX_OK) EXPECT(gn >= 0 && gn <= n && gnr >= 0 && gnr <= nr);
        }
        // random byte edits, deletions and duplications
        const char pool[] = "0123456789-+.e,:[]{}\" \n\tabcdefu!%&*#";
        for (int it = 0; it < 600; it++) {
            std::string s = text;
            const int edits = 1 + (int)(rng() % 6);
            for (int e = 0; e < edits && !s.empty(); e++) {
                const size_t pos = rng() % s.size();
                switch (rng() % 4) {
                case 0: s[pos] = pool[rng() % (sizeof pool - 1)]; break;
                case 1: s.erase(pos, 1 + rng() % 9); break;
                case 2: s.insert(pos, s.substr(pos, 1 + rng() % 40)); break;
                default: s[pos] = (char)(rng() & 0xff); break;
                }
            }
            spit(mut, s);
            read_into(mut, n + (int)(rng() % 3), nr + (int)(rng() % 3), &gn, &gnr);
            read_into(mut, 0, 0, &gn, &gnr);
        }
        // declared sizes that disagree with the data
        for (const char* from : {"rows: 37", "rows: 5", "cols: 32"}) {
            for (const char* to : {"rows: 1000000", "rows: -3", "cols: 0", "rows: 2147483647", "cols: 99999999999999999999"}) {
                std::string s = text;
                const size_t pos = s.find(from);
                if (pos == std::string::npos) continue;
                s.replace(pos, std::strlen(from), to);
                spit(mut, s);
                if (read_into(mut, n, nr, &gn, &gnr) == ORBX_OK) EXPECT(gn >= 0 && gn <= n && gnr >= 0 && gnr <= nr);
            }
        }
    }
    int gn = 5, gnr = 5;
    EXPECT(read_into(dir + "/no_such_file.yml", 4, 4, &gn, &gnr) == ORBX_ERR_IO && gn == 0 && gnr == 0);
    EXPECT(orbx_io_last_error() != nullptr);
}

static void device_sweep()
{
    orbx_ctx* ctx = nullptr;
    if (orbx_create(&ctx, 1000, 1.2f, 8, 20, 7, 0) != ORBX_OK) { std::fprintf(stderr, "orbx_create: %s\n", orbx_last_error(nullptr)); g_fail++; return; }
    g_ctx = ctx;
    std::mt19937 rng(11);
    const float inv_w = 64.f / 752.f, inv_h = 48.f / 480.f;
    for (int N : {0, 1, 63, 1000, 2500}) {
        auto kps = random_keypoints(rng, N, 752.f, 480.f);
        auto desc = random_words(rng, (size_t)N * 8);
        for (int Q : {0, 1, 17, 700}) {
            std::vector<float> qx((size_t)Q), qy((size_t)Q), qr((size_t)Q);
            std::vector<int32_t> qmin((size_t)Q), qmax((size_t)Q);
            for (int i = 0; i < Q; i++) {
                qx[i] = (float)(rng() % 900) - 70.f; qy[i] = (float)(rng() % 600) - 60.f; qr[i] = (float)(rng() % 120);
                qmin[i] = (int)(rng() % 4) - 1; qmax[i] = (rng() & 1) ? -1 : qmin[i] + (int)(rng() % 4);
            }
            auto qdesc = random_words(rng, (size_t)Q * 8);
            // candidate lists: first the total, then buffers of exactly that size
            std::vector<int32_t> off((size_t)Q + 1);
            int total = -1;
            int rc = orbx_features_in_area(ctx, kps.data(), N, 0.f, 0.f, inv_w, inv_h, qx.data(), qy.data(), qr.data(),
                                           qmin.data(), qmax.data(), Q, off.data(), nullptr, 0, &total);
            EXPECT(rc == ORBX_OK || rc == ORBX_ERR_CAPACITY);
            EXPECT(total >= 0);
            std::vector<int32_t> idx((size_t)total);
            rc = orbx_features_in_area(ctx, kps.data(), N, 0.f, 0.f, inv_w, inv_h, qx.data(), qy.data(), qr.data(),
                                       qmin.data(), qmax.data(), Q, off.data(), idx.data(), total, &total);
            EXPECT(rc == ORBX_OK && off[(size_t)Q] == total);
            for (int v : idx) EXPECT(v >= 0 && v < N);
            if (total > 1) {                       // one entry short
                std::vector<int32_t> small((size_t)total - 1);
                int t2 = 0;
                EXPECT(orbx_features_in_area(ctx, kps.data(), N, 0.f, 0.f, inv_w, inv_h, qx.data(), qy.data(), qr.data(),
                                             qmin.data(), qmax.data(), Q, off.data(), small.data(), total - 1, &t2) == ORBX_ERR_CAPACITY);
            }
            std::vector<int32_t> bi((size_t)Q), bd((size_t)Q), sd((size_t)Q), bl((size_t)Q), sl((size_t)Q), nc((size_t)Q);
            EXPECT(orbx_grid_search(ctx, kps.data(), desc.data(), N, 0.f, 0.f, inv_w, inv_h, qx.data(), qy.data(), qr.data(),
                                    qmin.data(), qmax.data(), qdesc.data(), Q, bi.data(), bd.data(), sd.data(), bl.data(),
                                    sl.data(), nc.data()) == ORBX_OK);
            EXPECT(orbx_grid_search(ctx, kps.data(), desc.data(), N, 0.f, 0.f, inv_w, inv_h, qx.data(), qy.data(), qr.data(),
                                    qmin.data(), qmax.data(), qdesc.data(), Q, bi.data(), bd.data(), sd.data(), bl.data(),
                                    sl.data(), nullptr) == ORBX_OK);
            std::vector<int32_t> oct((size_t)N);
            for (int i = 0; i < N; i++) oct[i] = kps[i].octave;
            std::vector<uint16_t> dist((size_t)total);
            EXPECT(orbx_hamming_csr(ctx, qdesc.data(), Q, desc.data(), N, off.data(), idx.data(), dist.data()) == ORBX_OK);
            for (auto v : dist) EXPECT(v <= 256);
            EXPECT(orbx_match_windowed(ctx, qdesc.data(), Q, desc.data(), N, off.data(), idx.data(), oct.data(), bi.data(),
                                       bd.data(), sd.data(), bl.data(), sl.data()) == ORBX_OK);
            EXPECT(orbx_match_windowed(ctx, qdesc.data(), Q, desc.data(), N, off.data(), idx.data(), nullptr, bi.data(),
                                       bd.data(), sd.data(), nullptr, nullptr) == ORBX_OK);
            EXPECT(orbx_match_bruteforce(ctx, qdesc.data(), Q, desc.data(), N, bi.data(), bd.data(), sd.data()) == ORBX_OK);
            for (int i = 0; i < Q; i++) EXPECT(N == 0 ? bi[i] == -1 : (bi[i] >= 0 && bi[i] < N));
        }
        // stereo searches of one pair, host arrays
        for (int nr : {0, 1, N}) {
            auto kr = random_keypoints(rng, nr, 752.f, 480.f);
            auto dr = random_words(rng, (size_t)nr * 8);
            std::vector<int32_t> bi((size_t)N), bd((size_t)N), listed((size_t)N);
            std::vector<float> ur((size_t)N), dep((size_t)N), mind((size_t)N, 0.f), maxd((size_t)N, 90.f);
            std::vector<uint8_t> active((size_t)N);
            for (auto& a : active) a = (uint8_t)(rng() & 1);
            int nm = -1;
            EXPECT(orbx_stereo_match(ctx, kps.data(), desc.data(), N, kr.data(), dr.data(), nr, 480, 47.9f, 0.11f, 0.f,
                                     bi.data(), bd.data(), ur.data(), dep.data(), &nm) == ORBX_OK);
            EXPECT(nm >= 0 && nm <= N);
            for (int cut = 0; cut < 2; cut++)
                EXPECT(orbx_stereo_match_ex(ctx, kps.data(), desc.data(), N, kr.data(), dr.data(), nr, 480, 47.9f, 0.11f, 0.f,
                                            active.data(), mind.data(), maxd.data(), 100, cut, listed.data(), bi.data(),
                                            bd.data(), ur.data(), dep.data(), &nm) == ORBX_OK);
            EXPECT(orbx_stereo_match_ex(ctx, kps.data(), desc.data(), N, kr.data(), dr.data(), nr, 480, 47.9f, 0.11f, 0.f,
                                        nullptr, nullptr, nullptr, 75, 1, listed.data(), bi.data(), bd.data(), ur.data(),
                                        dep.data(), &nm) == ORBX_OK);
            if (N > 0)                      // the Frame's distance list is in/out, never optional
                EXPECT(orbx_stereo_match_ex(ctx, kps.data(), desc.data(), N, kr.data(), dr.data(), nr, 480, 47.9f, 0.11f, 0.f,
                                            nullptr, nullptr, nullptr, 75, 1, nullptr, bi.data(), bd.data(), ur.data(),
                                            dep.data(), &nm) == ORBX_ERR_INVALID);
        }
    }
    // extraction of odd sizes into exact buffers, then the classic stereo search with its SAD refinement on the two
    // contexts' pyramids, undistortion in place
    orbx_ctx* right = nullptr;
    EXPECT(orbx_create(&right, 1000, 1.2f, 8, 20, 7, 0) == ORBX_OK);
    for (auto wh : {std::pair<int, int>{752, 480}, {333, 211}, {97, 64}}) {
        const int w = wh.first, h = wh.second, stride = w + 5;
        std::vector<uint8_t> imgL((size_t)stride * h), imgR((size_t)stride * h);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < stride; x++) {
                const uint8_t v = (uint8_t)(((x / 9) * 37 + (y / 7) * 91) ^ ((x * y) >> 3));
                imgL[(size_t)y * stride + x] = v;
                imgR[(size_t)y * stride + (x >= 6 ? x - 6 : x)] = v;
            }
        int cap = 0;
        EXPECT(orbx_max_keypoints(ctx, w, h, &cap) == ORBX_OK && cap > 0);
        std::vector<orbx_keypoint> kl((size_t)cap), kr((size_t)cap);
        std::vector<uint8_t> dl((size_t)cap * 32), dr((size_t)cap * 32);
        int nl = -1, nr = -1;
        EXPECT(orbx_extract(ctx, imgL.data(), w, h, stride, kl.data(), dl.data(), cap, &nl) == ORBX_OK);
        EXPECT(orbx_extract(right, imgR.data(), w, h, stride, kr.data(), dr.data(), cap, &nr) == ORBX_OK);
        EXPECT(nl >= 0 && nl <= cap && nr >= 0 && nr <= cap);
        if (nl > 2) {
            int n2 = 0;
            std::vector<orbx_keypoint> ks((size_t)nl - 1);
            std::vector<uint8_t> ds(((size_t)nl - 1) * 32);
            EXPECT(orbx_extract(ctx, imgL.data(), w, h, stride, ks.data(), ds.data(), nl - 1, &n2) == ORBX_ERR_CAPACITY);
            EXPECT(orbx_extract(ctx, imgL.data(), w, h, stride, kl.data(), dl.data(), cap, &nl) == ORBX_OK);
        }
        std::vector<int32_t> bi((size_t)nl), bd((size_t)nl);
        std::vector<float> ur((size_t)nl), dep((size_t)nl);
        int nm = -1;
        EXPECT(orbx_stereo_match_sad(ctx, right, kl.data(), reinterpret_cast<const uint32_t*>(dl.data()), nl, kr.data(),
                                     reinterpret_cast<const uint32_t*>(dr.data()), nr, 47.9f, 0.11f, bi.data(), bd.data(),
                                     ur.data(), dep.data(), &nm) == ORBX_OK);
        const double K[9] = {458.654, 0, 367.215, 0, 457.296, 248.375, 0, 0, 1};
        const double dist[5] = {-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0};
        std::vector<orbx_keypoint> ku((size_t)nl);
        EXPECT(orbx_undistort_keypoints(ctx, kl.data(), nl, K, dist, 4, nullptr, K, 3, 0, ku.data()) == ORBX_OK);
        EXPECT(orbx_undistort_keypoints(ctx, kl.data(), nl, K, dist, 5, nullptr, K, 3, 0, kl.data()) == ORBX_OK);
    }
    g_ctx = nullptr;
    orbx_destroy(right);
    orbx_destroy(ctx);
}

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: %s yaml|all tmpdir\n", argv[0]); return 2; }
    yaml_sweep(argv[2]);
    if (std::strcmp(argv[1], "all") == 0) device_sweep();
    std::printf("%s: %d failed expectations\n", argv[1], g_fail);
    return g_fail ? 1 : 0;
}
