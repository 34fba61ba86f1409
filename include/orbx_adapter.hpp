// orbx_adapter.hpp -- drop-in for the reference's include/ORBextractor.h +
// src/ORBextractor.cc at a site that has OpenCV: class ORB_SLAM2::ORBextractor
// with the reference's exact public interface (include/ORBextractor.h:71-176),
// implemented by liborbx (MI355X).  Frame.cc / Tracking.cc / Observability.cc
// compile against it unmodified.  Also ORBmatcher::DescriptorDistance as a free
// helper (include/ORBmatcher.h:48) for sites that want the library's popcount.
//
// This header needs OpenCV (cv::Mat, cv::KeyPoint) and is therefore not built in
// the OpenCV-less build container; all logic lives in orbx_host.hpp, which is.
#pragma once
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include <cassert>
#include <cstdlib>
#include <cstring>
#include <list>
#include <vector>

#include "orbx_host.hpp"

namespace ORB_SLAM2 {

static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint must be 7 x 4 bytes");

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // The reference's constructor has no device argument: extractors created afterwards run on the
    // GPU named by SetDevice(), initially the one in the environment variable ORBX_DEVICE (default 0).
    static int& Device()
    {
        static int dev = std::getenv("ORBX_DEVICE") ? std::atoi(std::getenv("ORBX_DEVICE")) : 0;
        return dev;
    }
    static void SetDevice(int device) { Device() = device; }

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : impl_(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, Device())
    {
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor() {}

    // Mask is ignored, as in the reference (include/ORBextractor.h:85-90).
    void operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                    cv::OutputArray _descriptors)
    {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1);
        std::vector<orbx_keypoint> kps;
        std::vector<uint8_t> desc;
        impl_(orbx::ImageView(image.data, image.cols, image.rows, image.step), orbx::ImageView(), kps, desc);
        const int n = (int)kps.size();
        if (n == 0)
            _descriptors.release();
        else {
            _descriptors.create(n, 32, CV_8U);
            cv::Mat d = _descriptors.getMat();
            for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), desc.data() + (size_t)i * 32, 32);
        }
        _keypoints.resize(n);
        if (n) std::memcpy(static_cast<void*>(_keypoints.data()), kps.data(), (size_t)n * sizeof(orbx_keypoint));
        refresh();
    }

    int inline GetLevels() { return impl_.GetLevels(); }
    float inline GetScaleFactor() { return impl_.GetScaleFactor(); }
    int inline GetInitThres() { return impl_.GetInitThres(); }
    int inline GetMinThres() { return impl_.GetMinThres(); }
    std::vector<float> inline GetScaleFactors() { return impl_.GetScaleFactors(); }
    std::vector<float> inline GetInverseScaleFactors() { return impl_.GetInverseScaleFactors(); }
    std::vector<float> inline GetScaleSigmaSquares() { return impl_.GetScaleSigmaSquares(); }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return impl_.GetInverseScaleSigmaSquares(); }

    // Host-readable, un-blurred level images of the last call: each is a region of interest inside
    // a matrix that also holds its 19 REFLECT_101 border pixels, as in the reference
    // (src/ORBextractor.cc:1198-1200), so locateROI / adjustROI by up to 19 and a copyMakeBorder
    // without BORDER_ISOLATED see the stored border.
    std::vector<cv::Mat> mvImagePyramid;

    void ComputePyramid(cv::Mat image)
    {
        impl_.ComputePyramid(orbx::ImageView(image.data, image.cols, image.rows, image.step));
        refresh();
    }

private:
    void refresh()
    {
        for (size_t l = 0; l < mvImagePyramid.size(); l++) {
            const orbx::ImageView& v = impl_.mvImagePyramid[l];
            const int B = 19;    // EDGE_THRESHOLD, src/ORBextractor.cc:72; the library stores at least that much
            cv::Mat whole(v.rows + 2 * B, v.cols + 2 * B, CV_8UC1, const_cast<uint8_t*>(v.data) - (size_t)B * v.step - B, v.step);
            mvImagePyramid[l] = whole(cv::Rect(B, B, v.cols, v.rows));
        }
    }
    orbx::ORBextractor impl_;
};

// ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&), src/ORBmatcher.cc:1768-1784
inline int DescriptorDistance(const cv::Mat& a, const cv::Mat& b)
{
    return orbx_hamming(a.ptr<uint32_t>(), b.ptr<uint32_t>());
}

} // namespace ORB_SLAM2
