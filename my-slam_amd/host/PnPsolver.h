// PnPsolver.h / Optimizer -- the reference's pose-solver call shapes over include/orbp.h (host code inside liborbx.so).
// Reference: include/PnPsolver.h:66-78 (PnPsolver(const Frame&, const vector<MapPoint*>&), SetRansacParameters, find,
// iterate) and include/Optimizer.h (static int PoseOptimization(Frame*)).  Frame / MapPoint belong to the reference's
// object graph, so the constructor here takes what the reference constructor reads out of them
// (src/PnPsolver.cc:78-106); INTEGRATION.md 3d shows the five-line gather that keeps the original signature.
// Poses are row-major 4x4 float (the layout of the reference's CV_32F Tcw); an empty optional stands for the empty
// cv::Mat the reference returns.
#pragma once
#include <array>
#include <cstdint>
#include <optional>
#include <string>
#include <vector>
#include "../../include/orbp.h"

namespace ORB_SLAM2 {

typedef std::array<float, 16> Pose;

class PnPsolver {
public:
    // p2d[i] = F.mvKeysUn[idx].pt, sigma2[i] = F.mvLevelSigma2[octave], p3d[i] = pMP->GetWorldPos(); keyPointIndices[i] = idx
    PnPsolver(const std::vector<float> &p2d, const std::vector<float> &sigma2, const std::vector<float> &p3d,
              const std::vector<size_t> &keyPointIndices, size_t nFrameFeatures, float fx, float fy, float cx, float cy)
        : mvKeyPointIndices(keyPointIndices), mnFrameFeatures(nFrameFeatures)
    {
        if (orbp_pnp_create(&s_, (int)sigma2.size(), p2d.data(), sigma2.data(), p3d.data(), fx, fy, cx, cy) < 0) { err_ = orbp_last_error(); s_ = nullptr; }
    }
    ~PnPsolver() { orbp_pnp_destroy(s_); }
    PnPsolver(const PnPsolver &) = delete;
    PnPsolver &operator=(const PnPsolver &) = delete;

    void SetRansacParameters(double probability = 0.99, int minInliers = 8, int maxIterations = 300, int minSet = 4,
                             float epsilon = 0.4, float th2 = 5.991)
    {
        if (s_ && orbp_pnp_set_ransac_parameters(s_, probability, minInliers, maxIterations, minSet, epsilon, th2) < 0) err_ = orbp_last_error();
    }

    std::optional<Pose> find(std::vector<bool> &vbInliers, int &nInliers)
    {
        bool flag;
        int maxIts = 0;
        if (s_) orbp_pnp_get_ransac_state(s_, nullptr, &maxIts, nullptr, nullptr);
        return iterate(maxIts, flag, vbInliers, nInliers);
    }

    // vbInliers is indexed like the frame's features (src/PnPsolver.cc:226-231)
    std::optional<Pose> iterate(int nIterations, bool &bNoMore, std::vector<bool> &vbInliers, int &nInliers)
    {
        bNoMore = false; vbInliers.clear(); nInliers = 0;
        if (!s_) { bNoMore = true; return std::nullopt; }
        std::vector<uint8_t> inl(mvKeyPointIndices.size() + 1);
        Pose T;
        int noMore = 0;
        const int got = orbp_pnp_iterate(s_, nIterations, &noMore, inl.data(), &nInliers, T.data());
        bNoMore = noMore != 0;
        if (got <= 0) { if (got < 0) err_ = orbp_last_error(); return std::nullopt; }
        vbInliers.assign(mnFrameFeatures, false);
        for (size_t i = 0; i < mvKeyPointIndices.size(); i++)
            if (inl[i]) vbInliers[mvKeyPointIndices[i]] = true;
        return T;
    }

    bool Valid() const { return s_ != nullptr; }
    const std::string &LastError() const { return err_; }

private:
    orbp_pnp *s_ = nullptr;
    std::vector<size_t> mvKeyPointIndices;
    size_t mnFrameFeatures;
    std::string err_;
};

class Optimizer {
public:
    // Optimizer::PoseOptimization(Frame*) on the arrays it gathers (src/Optimizer.cc:277-355): one entry per feature with
    // a MapPoint.  Tcw is pFrame->mTcw in and the optimised pose out; outlier receives mvbOutlier for those features.
    static int PoseOptimization(const std::vector<float> &obs, const std::vector<float> &uRight, const std::vector<float> &invSigma2,
                                const std::vector<float> &Xw, float fx, float fy, float cx, float cy, float bf, Pose &Tcw,
                                std::vector<bool> &outlier)
    {
        const int n = (int)invSigma2.size();
        std::vector<uint8_t> o(n + 1);
        const int good = orbp_pose_optimization(n, obs.data(), uRight.empty() ? nullptr : uRight.data(), invSigma2.data(), Xw.data(),
                                                fx, fy, cx, cy, bf, Tcw.data(), o.data());
        outlier.assign(o.begin(), o.begin() + n);
        return good;
    }
};

}  // namespace ORB_SLAM2
