// ORBmatcher.h -- GPU-backed Hamming primitives with the reference's ORBmatcher vocabulary
// (include/ORBmatcher.h:41-101 of WChen09/My-SLAM).  The nine Search*/Fuse methods of the reference
// walk the MapPoint/KeyFrame object graph on the host; what they all share is the inner loop
// "best / second-best DescriptorDistance over a candidate list" (SURVEY.md A10).  A maintainer keeps
// those methods and replaces their inner loops with BestTwo()/Distances() below; INTEGRATION.md shows
// the edit for SearchByBoW (src/ORBmatcher.cc:201-232) and SearchByProjection (:1397-1430).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#else
#include "orbx_cv_compat.h"
#endif
#include "../../include/orbm.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0, int maxDescriptors = 8192, int maxPairs = 1 << 22)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri)
    {
        if (orbm_create(&m_, device, maxDescriptors, maxDescriptors, maxPairs) != ORBX_OK) { err_ = orbm_last_error(); m_ = nullptr; }
    }
    ~ORBmatcher() { orbm_destroy(m_); }
    ORBmatcher(const ORBmatcher &) = delete;
    ORBmatcher &operator=(const ORBmatcher &) = delete;

    // Computes the Hamming distance between two ORB descriptors (include/ORBmatcher.h:44)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbm_distance(a.data, b.data); }

    // best / second-best over CSR candidate lists (candOff == nullptr: dense); outputs sized nq
    bool BestTwo(const cv::Mat &queries, const cv::Mat &train, const std::vector<int32_t> *candOff,
                 const std::vector<int32_t> *candIdx, std::vector<int32_t> &bestIdx,
                 std::vector<int32_t> &bestDist, std::vector<int32_t> &secondDist)
    {
        const int nq = queries.rows;
        bestIdx.assign(nq, -1); bestDist.assign(nq, 256); secondDist.assign(nq, 256);
        if (!m_ || nq == 0) return m_ != nullptr;
        const int rc = orbm_best2(m_, queries.data, nq, train.data, train.rows, candOff ? candOff->data() : nullptr,
                                  candIdx ? candIdx->data() : nullptr, bestIdx.data(), bestDist.data(), secondDist.data());
        if (rc != ORBX_OK) err_ = orbm_last_error();
        return rc == ORBX_OK;
    }
    // per-candidate distances, for the variants whose skip predicates depend on earlier matches
    bool Distances(const cv::Mat &queries, const cv::Mat &train, const std::vector<int32_t> &candOff,
                   const std::vector<int32_t> &candIdx, std::vector<int32_t> &dist)
    {
        dist.assign(candIdx.size(), 256);
        if (!m_) return false;
        const int rc = orbm_distances(m_, queries.data, queries.rows, train.data, train.rows, candOff.data(), candIdx.data(), dist.data());
        if (rc != ORBX_OK) err_ = orbm_last_error();
        return rc == ORBX_OK;
    }
    // rotation histogram + ComputeThreeMaxima cull (src/ORBmatcher.cc:236-246,266-284,1601-1642)
    static int RotationFilter(const std::vector<float> &angleQ, const std::vector<float> &angleT, std::vector<int32_t> &match12)
    {
        return orbm_rot_filter(angleQ.data(), angleT.data(), match12.data(), (int)match12.size());
    }

    static const int TH_LOW = ORBM_TH_LOW;
    static const int TH_HIGH = ORBM_TH_HIGH;
    static const int HISTO_LENGTH = ORBM_HISTO_LENGTH;
    bool Valid() const { return m_ != nullptr; }
    const std::string &LastError() const { return err_; }

protected:
    float mfNNratio;
    bool mbCheckOrientation;
    orbm_matcher *m_ = nullptr;
    std::string err_;
};

}  // namespace ORB_SLAM2
