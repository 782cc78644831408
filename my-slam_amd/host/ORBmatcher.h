// ORBmatcher.h -- GPU-backed Hamming primitives with the reference's ORBmatcher vocabulary
// (include/ORBmatcher.h:41-101 of WChen09/My-SLAM).  The nine Search*/Fuse methods of the reference
// walk the MapPoint/KeyFrame object graph on the host; what they all share is the inner loop
// "best / second-best DescriptorDistance over a candidate list" (SURVEY.md A10).  A maintainer keeps
// those methods and replaces their inner loops with BestTwo()/Distances() below; INTEGRATION.md shows
// the edit for SearchByBoW (src/ORBmatcher.cc:201-232) and SearchByProjection (:1397-1430).  The four matchers of the per-frame
// tracking path (SearchForInitialization, SearchByBoW, SearchByProjection x 2) are here whole, over arrays.
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

    // ---- whole matchers (include/orbm.h has the argument conventions; Frame / KeyFrame / MapPoint fields arrive as arrays) ----
    // The Frame grid of the frame that is searched (F2 / CurrentFrame / F): Frame::AssignFeaturesToGrid
    bool BuildGrid(const std::vector<cv::KeyPoint> &keysUn, float minX, float maxX, float minY, float maxY)
    {
        return ok(m_ ? orbm_grid_build(m_, (const orbx_keypoint *)keysUn.data(), (int)keysUn.size(), minX, maxX, minY, maxY) : ORBX_E_INVALID);
    }
    // int SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)
    int SearchForInitialization(const std::vector<cv::KeyPoint> &keysUn1, const cv::Mat &desc1, const std::vector<cv::KeyPoint> &keysUn2,
                                const cv::Mat &desc2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10)
    {
        vnMatches12.assign(keysUn1.size(), -1);
        int n = 0;
        if (!m_ || !ok(orbm_search_for_initialization(m_, (const orbx_keypoint *)keysUn1.data(), desc1.data, (int)keysUn1.size(),
                                                      (const orbx_keypoint *)keysUn2.data(), desc2.data, (int)keysUn2.size(),
                                                      (float *)vbPrevMatched.data(), windowSize, mfNNratio, mbCheckOrientation,
                                                      vnMatches12.data(), &n)))
            return 0;
        return n;
    }
    // int SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches): matchF[i] = key-frame feature or -1
    int SearchByBoW(const std::vector<cv::KeyPoint> &keysUnKF, const cv::Mat &descKF, const std::vector<uint8_t> &validKF,
                    const std::vector<int32_t> &kfNode, const std::vector<int32_t> &kfOff, const std::vector<int32_t> &kfIdx,
                    const std::vector<cv::KeyPoint> &keysF, const cv::Mat &descF, const std::vector<int32_t> &fNode,
                    const std::vector<int32_t> &fOff, const std::vector<int32_t> &fIdx, std::vector<int32_t> &matchF)
    {
        matchF.assign(keysF.size(), -1);
        int n = 0;
        if (!m_ || !ok(orbm_search_by_bow(m_, descKF.data, (const orbx_keypoint *)keysUnKF.data(), (int)keysUnKF.size(),
                                          validKF.empty() ? nullptr : validKF.data(), kfNode.data(), kfOff.data(), kfIdx.data(), (int)kfNode.size(),
                                          descF.data, (const orbx_keypoint *)keysF.data(), (int)keysF.size(), fNode.data(), fOff.data(), fIdx.data(),
                                          (int)fNode.size(), mfNNratio, mbCheckOrientation, matchF.data(), &n)))
            return 0;
        return n;
    }
    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
    struct LastFrameView {          // per feature of LastFrame
        std::vector<uint8_t> hasPoint;      // pMP && !mvbOutlier[i]
        std::vector<float> worldPos;        // 3 per feature
        std::vector<uint8_t> descriptors;   // pMP->GetDescriptor(), 32 per feature
        std::vector<int32_t> observations;  // pMP->Observations()
        const std::vector<cv::KeyPoint> *keys = nullptr;   // octave of mvKeys, angle of mvKeysUn
        float Tcw[16];
    };
    struct FrameView {              // the frame that is searched; its grid must be current (BuildGrid)
        float Tcw[16], fx, fy, cx, cy, mb, mbf, bounds[4];
        const std::vector<float> *scaleFactors = nullptr;
        const std::vector<cv::KeyPoint> *keysUn = nullptr;
        const cv::Mat *descriptors = nullptr;
        const std::vector<float> *uRight = nullptr;      // mvuRight or NULL
        std::vector<int32_t> pointObservations;          // -1 = NULL mvpMapPoints entry, else Observations(); updated
        std::vector<int32_t> assigned;                   // out: index of the feature / MapPoint put into mvpMapPoints[i2], or -1
    };
    int SearchByProjection(FrameView &Cur, const LastFrameView &Last, float th, bool bMono)
    {
        const int nc = (int)Cur.keysUn->size();
        Cur.assigned.assign(nc, -1);
        int n = 0;
        if (!m_ || !ok(orbm_search_by_projection_last(m_, (int)Last.keys->size(), Last.hasPoint.data(), Last.worldPos.data(), Last.descriptors.data(),
                                                      Last.observations.data(), (const orbx_keypoint *)Last.keys->data(), Cur.Tcw, Last.Tcw,
                                                      Cur.fx, Cur.fy, Cur.cx, Cur.cy, Cur.mb, Cur.mbf, Cur.bounds, Cur.scaleFactors->data(),
                                                      (int)Cur.scaleFactors->size(), (const orbx_keypoint *)Cur.keysUn->data(), Cur.descriptors->data,
                                                      Cur.uRight ? Cur.uRight->data() : nullptr, nc, th, bMono, mbCheckOrientation,
                                                      Cur.pointObservations.data(), Cur.assigned.data(), &n)))
            return 0;
        return n;
    }
    // int SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
    struct MapPointsView {          // per MapPoint of the local map
        std::vector<uint8_t> inView;                     // mbTrackInView && !isBad()
        std::vector<float> projX, projY, projXR, viewCos;
        std::vector<int32_t> predictedLevel, observations;
        std::vector<uint8_t> descriptors;
    };
    int SearchByProjection(FrameView &F, const MapPointsView &P, float th = 3)
    {
        const int nc = (int)F.keysUn->size();
        F.assigned.assign(nc, -1);
        int n = 0;
        if (!m_ || !ok(orbm_search_by_projection_map(m_, (int)P.inView.size(), P.inView.data(), P.projX.data(), P.projY.data(),
                                                     P.projXR.empty() ? nullptr : P.projXR.data(), P.predictedLevel.data(), P.viewCos.data(),
                                                     P.descriptors.data(), P.observations.data(), F.scaleFactors->data(), (int)F.scaleFactors->size(),
                                                     (const orbx_keypoint *)F.keysUn->data(), F.descriptors->data,
                                                     (F.uRight && !P.projXR.empty()) ? F.uRight->data() : nullptr, nc, th, mfNNratio,
                                                     F.pointObservations.data(), F.assigned.data(), &n)))
            return 0;
        return n;
    }

    static const int TH_LOW = ORBM_TH_LOW;
    static const int TH_HIGH = ORBM_TH_HIGH;
    static const int HISTO_LENGTH = ORBM_HISTO_LENGTH;
    bool Valid() const { return m_ != nullptr; }
    const std::string &LastError() const { return err_; }

protected:
    bool ok(int rc) { if (rc != ORBX_OK) err_ = orbm_last_error(); return rc == ORBX_OK; }
    float mfNNratio;
    bool mbCheckOrientation;
    orbm_matcher *m_ = nullptr;
    std::string err_;
};

}  // namespace ORB_SLAM2
