// C++ host-side use of liborbx through include/orbx_host.hpp: reads a raw 8-bit
// image, runs the extractor exactly like Frame::ExtractORB calls the
// reference's operator() (src/Frame.cc:553-559), and dumps the results.
//   host_extract <raw_u8_file> <w> <h> <out_file>
// out_file: int32 n, int32 nlevels, nlevels x (int32 w, int32 h), n x 28 bytes keypoints, n x 32 bytes descriptors
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbx_host.hpp"

int main(int argc, char** argv)
{
    if (argc != 5) { std::fprintf(stderr, "usage: %s raw w h out\n", argv[0]); return 2; }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    std::vector<uint8_t> img((size_t)w * h);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) { std::fprintf(stderr, "cannot read image\n"); return 2; }
    std::fclose(f);
    try {
        orbx::ORBextractor* ex = new orbx::ORBextractor(1000, 1.2f, 8, 20, 7);   // heap-allocated like Tracking.cc:271
        std::vector<orbx_keypoint> kps;
        std::vector<uint8_t> desc;
        (*ex)(orbx::ImageView(img.data(), w, h, (size_t)w), orbx::ImageView(), kps, desc);
        FILE* o = std::fopen(argv[4], "wb");
        int32_t n = (int32_t)kps.size(), nl = ex->GetLevels();
        std::fwrite(&n, 4, 1, o);
        std::fwrite(&nl, 4, 1, o);
        for (int l = 0; l < nl; l++) {
            int32_t wh[2] = {ex->mvImagePyramid[l].cols, ex->mvImagePyramid[l].rows};
            std::fwrite(wh, 4, 2, o);
        }
        std::fwrite(kps.data(), sizeof(orbx_keypoint), kps.size(), o);
        std::fwrite(desc.data(), 1, desc.size(), o);
        // a pixel of the top pyramid level must be readable on the host (Frame.cc:1089 reads such windows)
        const orbx::ImageView& top = ex->mvImagePyramid[nl - 1];
        uint8_t px = top.data[(size_t)(top.rows / 2) * top.step + top.cols / 2];
        std::fwrite(&px, 1, 1, o);
        std::fclose(o);
        std::printf("%d keypoints, scale[1]=%.6f\n", n, ex->GetScaleFactors()[1]);
        // per-call latency of the drop-in class, with and without the host mirror of the pyramid
        for (int mirror = 1; mirror >= 0; mirror--) {
            ex->mirrorPyramid = mirror != 0;
            for (int i = 0; i < 20; i++) (*ex)(orbx::ImageView(img.data(), w, h, (size_t)w), orbx::ImageView(), kps, desc);
            const auto t0 = std::chrono::steady_clock::now();
            const int reps = 200;
            for (int i = 0; i < reps; i++) (*ex)(orbx::ImageView(img.data(), w, h, (size_t)w), orbx::ImageView(), kps, desc);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            std::printf("operator() %s pyramid mirror: %.0f us per frame (%d keypoints)\n", mirror ? "with" : "without", us, (int)kps.size());
        }
        delete ex;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
