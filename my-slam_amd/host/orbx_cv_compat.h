// orbx_cv_compat.h -- the handful of OpenCV 3.1.0 core types the adapters use, for hosts WITHOUT OpenCV
// (this image has none).  With OpenCV present the adapters include the real headers and this file is not
// used.  It is NOT a stand-in for building the reference: it exists so that my-slam_amd/host/*.h can be
// compiled and exercised by this repo's own tests THROUGH THE SAME CODE PATH a maintainer compiles against
// OpenCV: the adapters are written against OpenCV's API (InputArray::getMat(), OutputArray::create(),
// unqualified CV_8U ...), and this header models exactly those calls:
//   CV_8U / CV_8UC1 / CV_32F / CV_32FC1 are macros with OpenCV's values (core/hal/interface.h);
//   cv::InputArray = const cv::_InputArray &, cv::OutputArray = const cv::_OutputArray & (core/mat.hpp:...),
//   both constructible from a cv::Mat as in `extractor(im, cv::Mat(), keys, descriptors)` (src/Frame.cc:250).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#ifndef CV_8U
#define CV_8U 0
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#endif

namespace cv {
template <class T> struct Point_ { T x = 0, y = 0; Point_() {} Point_(T x_, T y_) : x(x_), y(y_) {} };
typedef Point_<float> Point2f;
struct KeyPoint {                      // field order of cv::KeyPoint (28 bytes)
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};

class Mat {                            // single-channel CV_8U / CV_32F, two dimensions
public:
    int rows = 0, cols = 0; size_t step = 0; uint8_t *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t step_ = 0) : rows(r), cols(c), step(step_ ? step_ : (size_t)c * esz(type)), data((uint8_t *)ext), type_(type) {}
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && store_) return;
        rows = r; cols = c; type_ = type; step = (size_t)c * esz(type);
        store_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * step + 16], std::default_delete<uint8_t[]>());
        data = store_.get();
    }
    void release() { store_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return esz(type_); }
    bool isContinuous() const { return step == (size_t)cols * esz(type_) || rows <= 1; }
    template <class T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <class T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }
    template <class T> T &at(int r, int c) { return ((T *)(data + (size_t)r * step))[c]; }
    template <class T> const T &at(int r, int c) const { return ((const T *)(data + (size_t)r * step))[c]; }
    template <class T> T &at(int i) { return cols == 1 ? *(T *)(data + (size_t)i * step) : ((T *)data)[i]; }
    template <class T> const T &at(int i) const { return cols == 1 ? *(const T *)(data + (size_t)i * step) : ((const T *)data)[i]; }
    Mat row(int r) const { Mat m(1, cols, type_, data + (size_t)r * step, step); m.store_ = store_; return m; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; r++) memcpy(m.data + (size_t)r * m.step, data + (size_t)r * step, (size_t)cols * esz(type_));
        return m;
    }
private:
    static size_t esz(int type) { return (type & 7) == CV_32F ? 4 : 1; }
    std::shared_ptr<uint8_t> store_;
    int type_ = CV_8UC1;
};

class _InputArray {                    // core/mat.hpp: the proxy every cv:: function takes its inputs through
public:
    _InputArray() {}
    _InputArray(const Mat &m) : m_(&m) {}
    Mat getMat(int = -1) const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }
    int type(int = -1) const { return m_ ? m_->type() : 0; }
protected:
    const Mat *m_ = nullptr;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat &m) : _InputArray(m), o_(&m) {}
    void create(int rows, int cols, int type) const { if (o_) o_->create(rows, cols, type); }
    void release() const { if (o_) o_->release(); }
    Mat getMat(int = -1) const { return o_ ? *o_ : Mat(); }
private:
    Mat *o_ = nullptr;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
}  // namespace cv
