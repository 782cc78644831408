// ORBextractor.h -- drop-in replacement for the reference's include/ORBextractor.h (WChen09/My-SLAM).
// Same namespace, class name, constructor, operator() and getters (include/ORBextractor.h:46-112), so
// src/Frame.cc:247-253 (ExtractORB) and src/Tracking.cc:121-127 compile unchanged; the work is done
// by liborbx.so (hand-written HIP for gfx950) through the C ABI of include/orbx.h.
//
// Differences a maintainer should know:
//   * mvImagePyramid is filled lazily by FetchImagePyramid() (a D2H copy per level); only
//     Frame::ComputeStereoMatches reads it (src/Frame.cc:473,563,580).
//   * Failures of the GPU layer produce an empty result, the reference's only failure mode
//     (src/ORBextractor.cc:1048-1049), and the text is available through LastError().
#pragma once
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

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int device = 0, int maxWidth = 1920, int maxHeight = 1080)
        : nfeatures_(nfeatures), nlevels_(nlevels), scaleFactor_(scaleFactor)
    {
        if (orbx_create(&h_, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, maxWidth, maxHeight, 1) != ORBX_OK) {
            err_ = orbx_last_error();
            h_ = nullptr;
            return;
        }
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        orbx_get_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
        mvImagePyramid.resize(nlevels);
        cap_ = orbx_capacity(h_);
        kp_.resize(cap_);
    }
    ~ORBextractor() { orbx_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image; mask is ignored (as in the reference).
    void operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors)
    {
        const cv::Mat &im = image;
        keypoints.clear();
        if (im.empty() || !h_) { descriptors.release(); return; }        // :1048-1049
        cv::Mat tmp(cap_, 32, cv::CV_8U);
        int n = 0;
        const int rc = orbx_extract(h_, im.data, im.cols, im.rows, (int)im.step, kp_.data(), tmp.data, cap_, &n);
        if (rc != ORBX_OK) { err_ = orbx_last_error(); descriptors.release(); return; }
        fill(tmp, n, keypoints, descriptors);
    }

    // operator() in two halves (orbx_extract_begin / orbx_extract_end): Begin returns as soon as the image is staged and the
    // work is queued; End waits and fills the outputs.  One call in flight per extractor (the stereo pair has two anyway).
    bool Begin(cv::InputArray image)
    {
        const cv::Mat &im = image;
        if (!h_) return false;
        const int rc = im.empty() ? orbx_extract_begin(h_, nullptr, 0, 0, 0) : orbx_extract_begin(h_, im.data, im.cols, im.rows, (int)im.step);
        if (rc != ORBX_OK) err_ = orbx_last_error();
        return rc == ORBX_OK;
    }
    void End(std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors)
    {
        keypoints.clear();
        if (!h_) { descriptors.release(); return; }
        cv::Mat tmp(cap_, 32, cv::CV_8U);
        int n = 0;
        const int rc = orbx_extract_end(h_, kp_.data(), tmp.data, cap_, &n);
        if (rc != ORBX_OK) { err_ = orbx_last_error(); descriptors.release(); return; }
        fill(tmp, n, keypoints, descriptors);
    }

private:
    void fill(const cv::Mat &tmp, int n, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors)
    {
        if (n == 0) { descriptors.release(); return; }                   // :1066-1067
        descriptors.create(n, 32, cv::CV_8U);                            // :1070
        for (int i = 0; i < n; i++) memcpy(descriptors.ptr<uint8_t>(i), tmp.ptr<uint8_t>(i), 32);
        keypoints.resize(n);
        static_assert(sizeof(orbx_keypoint) == 28, "orbx_keypoint mirrors cv::KeyPoint");
        for (int i = 0; i < n; i++) {
            cv::KeyPoint &k = keypoints[i];
            k.pt.x = kp_[i].x; k.pt.y = kp_[i].y; k.size = kp_[i].size; k.angle = kp_[i].angle;
            k.response = kp_[i].response; k.octave = kp_[i].octave; k.class_id = kp_[i].class_id;
        }
    }

public:
    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return scaleFactor_; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // include/ORBextractor.h:85.  Call after operator() when the pyramid is needed on the host.
    std::vector<cv::Mat> mvImagePyramid;
    bool FetchImagePyramid()
    {
        if (!h_) return false;
        for (int l = 0; l < nlevels_; l++) {
            int w = 0, hgt = 0;
            if (orbx_level_size(h_, l, &w, &hgt) != ORBX_OK) return false;
            mvImagePyramid[l].create(hgt, w, cv::CV_8U);
            if (orbx_download_level(h_, 0, l, mvImagePyramid[l].data, (int)mvImagePyramid[l].step, 0) != ORBX_OK) return false;
        }
        return true;
    }
    orbx_extractor *handle() { return h_; }   // for orbx_stereo_matches (Frame::ComputeStereoMatches)
    bool Valid() const { return h_ != nullptr; }
    const std::string &LastError() const { return err_; }

protected:
    orbx_extractor *h_ = nullptr;
    int nfeatures_, nlevels_, cap_ = 0;
    float scaleFactor_;
    std::vector<orbx_keypoint> kp_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::string err_;
};

}  // namespace ORB_SLAM2
