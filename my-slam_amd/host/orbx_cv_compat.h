// orbx_cv_compat.h -- the handful of cv:: types the adapters need, used ONLY when OpenCV's headers are
// absent (this image has no OpenCV).  With OpenCV present the adapters include the real headers and
// this file is not used.  It is NOT a stand-in for building the reference: it exists so that
// my-slam_amd/host/*.h can be compiled and exercised by this repo's own tests.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace cv {
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct KeyPoint {                      // field order of cv::KeyPoint (28 bytes)
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};
enum { CV_8U = 0, CV_8UC1 = 0 };
class Mat {                            // 8-bit single-channel only
public:
    int rows = 0, cols = 0; size_t step = 0; uint8_t *data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
    Mat(int r, int c, int /*type*/, void *ext, size_t step_) : rows(r), cols(c), step(step_), data((uint8_t *)ext) {}
    void create(int r, int c, int /*type*/) {
        if (r == rows && c == cols && store_) return;
        rows = r; cols = c; step = (size_t)c;
        store_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * c + 1], std::default_delete<uint8_t[]>());
        data = store_.get();
    }
    void release() { store_.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    template <class T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step); }
    template <class T> const T *ptr(int r = 0) const { return (const T *)(data + (size_t)r * step); }
    Mat row(int r) const { return Mat(1, cols, CV_8U, data + (size_t)r * step, step); }
private:
    std::shared_ptr<uint8_t> store_;
};
typedef const Mat &InputArray;
typedef Mat &OutputArray;
}  // namespace cv
