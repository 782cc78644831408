// ORBextractor.h -- drop-in replacement for the reference's include/ORBextractor.h (WChen09/My-SLAM).
// Same namespace, class name, constructor, operator() and getters (include/ORBextractor.h:46-112), so
// src/Frame.cc:247-253 (ExtractORB) and src/Tracking.cc:121-127 compile unchanged; the work is done
// by liborbx.so (hand-written HIP for gfx950) through the C ABI of include/orbx.h.
//
// Written against OpenCV 3.1.0's API (image.getMat(), descriptors.create() + getMat(), the CV_8U / CV_8UC1
// macros); on a host without OpenCV, orbx_cv_compat.h models those same calls so that this exact code is
// what the repo's tests compile.
//
// Differences a maintainer should know:
//   * mvImagePyramid is valid after every operator(), as in the reference (include/ORBextractor.h:85; read by
//     Frame::ComputeStereoMatches, src/Frame.cc:473,563,580): each level is a view of the interior of a
//     (w + 38) x (h + 38) buffer with the 19-px reflect-101 border (src/ORBextractor.cc:1115-1133), downloaded
//     after the extraction in one synchronisation (orbx_download_pyramid).  A caller that never reads it on the
//     host (monocular tracking; stereo through orbx_stereo_matches, which reads the levels on the device)
//     switches the download off with SetHostPyramid(false) and may still call FetchImagePyramid() on demand.
//   * The handle is sized for the largest image seen so far (it starts at 1920x1080 and grows on demand).
//   * Failures of the GPU layer, and an image that is not CV_8UC1 (the reference asserts, :1052), produce an
//     empty result, the reference's only failure mode (src/ORBextractor.cc:1048-1049); the text is in LastError().
#pragma once
#include <cstring>
#include <string>
#include <vector>

#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else
#include "orbx_cv_compat.h"
#endif
#include "../../include/orbx.h"

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };   // include/ORBextractor.h:49 (unused there too)

    // include/ORBextractor.h:51-52; device / maxWidth / maxHeight are this layer's own, defaulted
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int device = 0, int maxWidth = 1920, int maxHeight = 1080)
        : nfeatures_(nfeatures), nlevels_(nlevels), iniTh_(iniThFAST), minTh_(minThFAST), device_(device), scaleFactor_(scaleFactor)
    {
        mvImagePyramid.resize(nlevels > 0 ? nlevels : 0);
        open(maxWidth, maxHeight);
    }
    ~ORBextractor() { orbx_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image; mask is ignored (as in the reference, :1045-1107).
    void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &_keypoints,
                    cv::OutputArray _descriptors)
    {
        _keypoints.clear();
        if (_image.empty()) { _descriptors.release(); return; }                  // :1048-1049
        cv::Mat image = _image.getMat();
        if (image.type() != CV_8UC1) { err_ = "image is not CV_8UC1"; _descriptors.release(); return; }   // :1052 assert
        if (!fits(image.cols, image.rows)) { _descriptors.release(); return; }
        int n = 0;
        const int rc = orbx_extract(h_, image.data, image.cols, image.rows, (int)image.step, kp_.data(), desc_.data(), cap_, &n);
        if (rc != ORBX_OK) { err_ = orbx_last_error(); _descriptors.release(); return; }
        fill(n, _keypoints, _descriptors);
        if (hostPyramid_) FetchImagePyramid();                           // ComputePyramid's side effect, :1055
    }

    // operator() in two halves (orbx_extract_begin / orbx_extract_end): Begin returns as soon as the image is staged and the
    // work is queued; End waits and fills the outputs.  One call in flight per extractor (the stereo pair has two anyway).
    bool Begin(cv::InputArray _image)
    {
        cv::Mat image = _image.getMat();
        if (!image.empty() && image.type() != CV_8UC1) { err_ = "image is not CV_8UC1"; return false; }
        if (!image.empty() && !fits(image.cols, image.rows)) return false;
        if (!h_) return false;
        const int rc = image.empty() ? orbx_extract_begin(h_, nullptr, 0, 0, 0)
                                     : orbx_extract_begin(h_, image.data, image.cols, image.rows, (int)image.step);
        if (rc != ORBX_OK) err_ = orbx_last_error();
        return rc == ORBX_OK;
    }
    void End(std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors)
    {
        _keypoints.clear();
        if (!h_) { _descriptors.release(); return; }
        int n = 0;
        const int rc = orbx_extract_end(h_, kp_.data(), desc_.data(), cap_, &n);
        if (rc != ORBX_OK) { err_ = orbx_last_error(); _descriptors.release(); return; }
        fill(n, _keypoints, _descriptors);
        if (hostPyramid_) FetchImagePyramid();
    }

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return scaleFactor_; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // include/ORBextractor.h:85: valid after every operator() (End()), unless SetHostPyramid(false)
    std::vector<cv::Mat> mvImagePyramid;
    void SetHostPyramid(bool on) { hostPyramid_ = on; }
    bool FetchImagePyramid()
    {
        if (!h_) return false;
        const int B = 19;                                                // EDGE_THRESHOLD, src/ORBextractor.cc:76
        std::vector<unsigned char *> dst(nlevels_);
        std::vector<int> stride(nlevels_);
        if ((int)pyrStore_.size() != nlevels_) pyrStore_.resize(nlevels_);
        for (int l = 0; l < nlevels_; l++) {
            int w = 0, hgt = 0;
            if (orbx_level_size(h_, l, &w, &hgt) != ORBX_OK) { err_ = orbx_last_error(); return false; }
            pyrStore_[l].create(hgt + 2 * B, w + 2 * B, CV_8UC1);        // :1115-1117: temp(sz + 2*EDGE_THRESHOLD)
            dst[l] = pyrStore_[l].data; stride[l] = (int)pyrStore_[l].step;
            // mvImagePyramid[level] = temp(Rect(EDGE_THRESHOLD, EDGE_THRESHOLD, sz.width, sz.height)): a view of the interior
            mvImagePyramid[l] = cv::Mat(hgt, w, CV_8UC1, pyrStore_[l].data + (size_t)B * pyrStore_[l].step + B, pyrStore_[l].step);
        }
        if (orbx_download_pyramid(h_, 0, dst.data(), stride.data(), B) != ORBX_OK) { err_ = orbx_last_error(); return false; }
        return true;
    }
    orbx_extractor *handle() { return h_; }   // for orbx_stereo_matches (Frame::ComputeStereoMatches)
    bool Valid() const { return h_ != nullptr; }
    const std::string &LastError() const { return err_; }

protected:
    bool open(int maxW, int maxH)
    {
        if (h_) { orbx_destroy(h_); h_ = nullptr; }
        if (orbx_create(&h_, nfeatures_, scaleFactor_, nlevels_, iniTh_, minTh_, device_, maxW, maxH, 1) != ORBX_OK) {
            err_ = orbx_last_error();
            h_ = nullptr;
            return false;
        }
        maxW_ = maxW; maxH_ = maxH;
        mvScaleFactor.resize(nlevels_); mvInvScaleFactor.resize(nlevels_);
        mvLevelSigma2.resize(nlevels_); mvInvLevelSigma2.resize(nlevels_);
        orbx_get_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
        cap_ = orbx_capacity(h_);
        kp_.resize(cap_);
        desc_.resize((size_t)cap_ * 32);
        return true;
    }
    bool fits(int w, int h)   // a larger image than any before: re-create the handle for it (the reference has no size limit)
    {
        if (h_ && w <= maxW_ && h <= maxH_) return true;
        return open(w > maxW_ ? w : maxW_, h > maxH_ ? h : maxH_);
    }
    void fill(int n, std::vector<cv::KeyPoint> &_keypoints, cv::OutputArray _descriptors)
    {
        if (n == 0) { _descriptors.release(); return; }                  // :1066-1067
        _descriptors.create(n, 32, CV_8U);                               // :1070
        cv::Mat descriptors = _descriptors.getMat();                     // :1071
        for (int i = 0; i < n; i++) memcpy(descriptors.ptr<unsigned char>(i), &desc_[(size_t)i * 32], 32);
        _keypoints.resize(n);
        static_assert(sizeof(orbx_keypoint) == 28, "orbx_keypoint mirrors cv::KeyPoint");
        for (int i = 0; i < n; i++) {
            cv::KeyPoint &k = _keypoints[i];
            k.pt.x = kp_[i].x; k.pt.y = kp_[i].y; k.size = kp_[i].size; k.angle = kp_[i].angle;
            k.response = kp_[i].response; k.octave = kp_[i].octave; k.class_id = kp_[i].class_id;
        }
    }

    orbx_extractor *h_ = nullptr;
    int nfeatures_, nlevels_, iniTh_, minTh_, device_, cap_ = 0, maxW_ = 0, maxH_ = 0;
    float scaleFactor_;
    std::vector<orbx_keypoint> kp_;
    std::vector<unsigned char> desc_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<cv::Mat> pyrStore_;     // the bordered level buffers mvImagePyramid's views point into
    bool hostPyramid_ = true;
    std::string err_;
};

}  // namespace ORB_SLAM2
