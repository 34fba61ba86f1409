// Minimal stand-in for the few OpenCV types include/orbx_adapter.hpp touches (cv::Mat as a view
// with a parent extent, cv::KeyPoint, cv::InputArray / cv::OutputArray, cv::Rect, cv::Range).
// It exists only so that the drop-in header and a caller shaped like the reference's Frame.cc can
// be compiled and run in an image without OpenCV; it is not OpenCV and pins no OpenCV result.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float a, float b) : x(a), y(b) {} };
struct Point { int x = 0, y = 0; Point() {} Point(int a, int b) : x(a), y(b) {} };
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() {} Rect(int a, int b, int w, int h) : x(a), y(b), width(w), height(h) {} };
struct Range { int start = 0, end = 0; Range() {} Range(int s, int e) : start(s), end(e) {} };

// same member order and size as cv::KeyPoint: 7 x 4 bytes
struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    KeyPoint() {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    const uint8_t* datastart = nullptr;     // extent of the matrix this view was cut from
    const uint8_t* dataend = nullptr;

    Mat() {}
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8UC1); }
    Mat(int r, int c, int /*type*/, void* d, size_t s = 0)
        : rows(r), cols(c), step(s ? s : (size_t)c), data(static_cast<uint8_t*>(d))
    {
        datastart = data;
        dataend = data + (size_t)(rows - 1) * step + cols;
    }
    void create(int r, int c, int /*type*/)
    {
        if (r == rows && c == cols && owner_ && step == (size_t)c) return;
        owner_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * c + 1](), std::default_delete<uint8_t[]>());
        rows = r; cols = c; step = (size_t)c; data = owner_.get();
        datastart = data; dataend = data + (size_t)r * c;
    }
    void release() { owner_.reset(); rows = cols = 0; step = 0; data = nullptr; datastart = dataend = nullptr; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    Size size() const { return Size(cols, rows); }
    uint8_t* ptr(int r = 0) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r = 0) const { return data + (size_t)r * step; }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
    template <class T> T& at(int r, int c) { return reinterpret_cast<T*>(data + (size_t)r * step)[c]; }
    template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(data + (size_t)r * step)[c]; }
    Mat operator()(const Rect& roi) const
    {
        Mat m = *this;
        m.data = data + (size_t)roi.y * step + roi.x;
        m.rows = roi.height; m.cols = roi.width;
        return m;
    }
    Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
    Mat row(int r) const { return rowRange(r, r + 1); }
    Mat clone() const
    {
        Mat m;
        m.create(rows, cols, CV_8UC1);
        for (int r = 0; r < rows; r++) std::memcpy(m.ptr(r), ptr(r), (size_t)cols);
        return m;
    }
    // size of the matrix this view lies in and the view's offset inside it (cv::Mat::locateROI)
    void locateROI(Size& whole, Point& ofs) const
    {
        const size_t d = (size_t)(data - datastart);
        ofs.y = (int)(d / step); ofs.x = (int)(d % step);
        const size_t span = (size_t)(dataend - datastart);
        whole.height = (int)((span - 1) / step) + 1;
        whole.width = (int)(span - (size_t)(whole.height - 1) * step);
    }

private:
    std::shared_ptr<uint8_t> owner_;
};

// cv::InputArray / cv::OutputArray are references to proxy objects that wrap a Mat
class _InputArray {
public:
    _InputArray() {}
    _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
    bool empty() const { return !m_ || m_->empty(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }
protected:
    Mat* m_ = nullptr;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat& m) { m_ = &m; }
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void release() const { if (m_) m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

} // namespace cv
