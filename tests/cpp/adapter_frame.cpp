// Compiles include/orbx_adapter.hpp -- the drop-in ORB_SLAM2::ORBextractor with cv:: types -- and
// drives it the way the reference's Frame does:
//   Frame::ExtractORB           (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors)      src/Frame.cc:553-559
//   Tracking                    new ORBextractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST)  src/Tracking.cc:271-279
//   Frame::ComputeStereoMatches mvImagePyramid[l].rowRange(..).colRange(..) windows            src/Frame.cc:1089, :1111
//   detection I/O               ComputePyramid(im) on its own                                  src/Frame.cc:187-190
// Built against tests/cpp/mini_cv (a stand-in for the handful of OpenCV types), so it proves that the
// header is code and keeps its call sites compiling; results are checked against orbx_host.hpp.
//   adapter_frame <raw_u8_file> <w> <h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbx_adapter.hpp"

using namespace ORB_SLAM2;

struct Frame {      // the members the reference's Frame uses around the extractor
    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    cv::Mat mDescriptors, mDescriptorsRight;
    void ExtractORB(int flag, const cv::Mat& im)
    {
        if (flag == 0) (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
        else (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
    }
};

int main(int argc, char** argv)
{
    if (argc != 4) { std::fprintf(stderr, "usage: %s raw w h\n", argv[0]); return 2; }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    std::vector<uint8_t> buf((size_t)w * h);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fprintf(stderr, "cannot read image\n"); return 2; }
    std::fclose(f);
    cv::Mat im(h, w, CV_8UC1, buf.data(), (size_t)w);
    try {
        ORBextractor::SetDevice(0);
        Frame F;
        F.mpORBextractorLeft = new ORBextractor(1000, 1.2f, 8, 20, 7);
        F.mpORBextractorRight = new ORBextractor(1000, 1.2f, 8, 20, 7);
        F.ExtractORB(0, im);
        F.ExtractORB(1, im);
        const int n = (int)F.mvKeys.size();
        if (n < 500 || F.mDescriptors.rows != n || F.mDescriptors.cols != 32) { std::fprintf(stderr, "bad outputs\n"); return 1; }
        if ((int)F.mvKeysRight.size() != n) { std::fprintf(stderr, "left and right extractor disagree\n"); return 1; }
        // the same through the OpenCV-free host class
        orbx::ORBextractor ref(1000, 1.2f, 8, 20, 7);
        std::vector<orbx_keypoint> kps;
        std::vector<uint8_t> desc;
        ref(orbx::ImageView(buf.data(), w, h, (size_t)w), orbx::ImageView(), kps, desc);
        if ((int)kps.size() != n || std::memcmp(kps.data(), F.mvKeys.data(), (size_t)n * sizeof(orbx_keypoint)) != 0) {
            std::fprintf(stderr, "keypoints differ from orbx_host.hpp\n");
            return 1;
        }
        for (int i = 0; i < n; i++)
            if (std::memcmp(F.mDescriptors.ptr(i), desc.data() + (size_t)i * 32, 32) != 0) { std::fprintf(stderr, "descriptor %d differs\n", i); return 1; }
        if (DescriptorDistance(F.mDescriptors.row(0), F.mDescriptorsRight.row(0)) != 0) { std::fprintf(stderr, "distance of equal rows\n"); return 1; }
        // pyramid views as ComputeStereoMatches reads them (:1089, :1111), and their stored border
        const std::vector<float> sf = F.mpORBextractorLeft->GetScaleFactors();
        long sum = 0;
        for (int l = 0; l < F.mpORBextractorLeft->GetLevels(); l++) {
            const cv::Mat& lvl = F.mpORBextractorLeft->mvImagePyramid[l];
            const int cy = lvl.rows / 2, cx = lvl.cols / 2, r = 5;
            cv::Mat IL = lvl.rowRange(cy - r, cy + r + 1).colRange(cx - r, cx + r + 1);
            for (int y = 0; y < IL.rows; y++) for (int x = 0; x < IL.cols; x++) sum += IL.at<uint8_t>(y, x);
            cv::Size whole; cv::Point ofs;
            lvl.locateROI(whole, ofs);
            if (ofs.x != 19 || ofs.y != 19 || whole.width != lvl.cols + 38 || whole.height != lvl.rows + 38) {
                std::fprintf(stderr, "level %d is not a region of interest inside its bordered plane\n", l);
                return 1;
            }
            // REFLECT_101: the border pixel left of (0, y) equals pixel (1, y)
            if (*(lvl.ptr(3) - 1) != lvl.at<uint8_t>(3, 1) || *(lvl.ptr(0) - (long)lvl.step) != lvl.at<uint8_t>(1, 0)) {
                std::fprintf(stderr, "level %d: stored border is not REFLECT_101\n", l);
                return 1;
            }
        }
        if (F.mpORBextractorLeft->mvImagePyramid[0].at<uint8_t>(7, 9) != buf[7 * (size_t)w + 9]) { std::fprintf(stderr, "level 0 is not the image\n"); return 1; }
        // the detection-I/O path: the pyramid alone
        F.mpORBextractorRight->ComputePyramid(im);
        if (F.mpORBextractorRight->mvImagePyramid[1].cols != F.mpORBextractorLeft->mvImagePyramid[1].cols) return 1;
        // an empty image leaves the outputs untouched (src/ORBextractor.cc:1130-1131)
        (*F.mpORBextractorLeft)(cv::Mat(), cv::Mat(), F.mvKeys, F.mDescriptors);
        if ((int)F.mvKeys.size() != n) return 1;
        std::printf("adapter ok: %d keypoints, %d levels, scale[1]=%.6f, window sum %ld\n", n, F.mpORBextractorLeft->GetLevels(), sf[1], sum);
        delete F.mpORBextractorLeft;
        delete F.mpORBextractorRight;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
