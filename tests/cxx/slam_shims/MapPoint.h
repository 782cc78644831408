// MapPoint.h -- repo-authored minimal MapPoint for the adapter tests: only the members ORB_SLAM2::ORBmatcher's methods read or
// call (names and signatures as in the reference's include/MapPoint.h:40-150, bodies written here).  It is test scaffolding for
// my-slam_amd/host/ORBmatcher.h, not a rebuild of the reference's map: Replace() keeps the part the matchers can observe
// (bad flag, observations moved over, key-frame slots rewritten) and leaves out the map bookkeeping.
#pragma once
#include <cmath>
#include <map>
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#else
#include "../../../my-slam_amd/host/orbx_cv_compat.h"
#endif

namespace ORB_SLAM2 {
class KeyFrame;
class Frame;

class MapPoint {
public:
    MapPoint(const cv::Mat &Pos, const cv::Mat &descriptor, int observations, float minDistance, float maxDistance)
        : mTrackProjX(0), mTrackProjY(0), mTrackProjXR(-1), mbTrackInView(false), mnTrackScaleLevel(0), mTrackViewCos(1),
          mnId(nNextId++), mWorldPos(Pos.clone()), mDescriptor(descriptor.clone()), nObs(observations), mbBad(false), mfMinDistance(minDistance),
          mfMaxDistance(maxDistance), mpReplaced(nullptr)
    {
        mNormalVector = cv::Mat(3, 1, CV_32F);
        for (int k = 0; k < 3; k++) mNormalVector.at<float>(k) = 0.f;
    }
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    void SetNormal(float x, float y, float z) { mNormalVector.at<float>(0) = x; mNormalVector.at<float>(1) = y; mNormalVector.at<float>(2) = z; }
    int Observations() { return nObs; }
    bool isBad() { return mbBad; }
    void SetBadFlag() { mbBad = true; }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int PredictScale(const float &currentDist, Frame *pF);      // defined in Frame.h (needs the complete Frame)
    int PredictScale(const float &currentDist, KeyFrame *pKF);  // defined in KeyFrame.h
    // observations (src/MapPoint.cc:98-109, 160-175, 177-215)
    void AddObservation(KeyFrame *pKF, size_t idx);             // defined in KeyFrame.h (reads pKF->mvuRight)
    int GetIndexInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    bool IsInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) != 0; }
    void Replace(MapPoint *pMP);                                // defined in KeyFrame.h
    MapPoint *GetReplaced() { return mpReplaced; }

    // Variables used by the tracking (public in the reference too)
    float mTrackProjX, mTrackProjY, mTrackProjXR;
    bool mbTrackInView;
    int mnTrackScaleLevel;
    float mTrackViewCos;
    long unsigned int mnId;
    inline static long unsigned int nNextId = 0;

    float MaxDistance() const { return mfMaxDistance; }        // test-side access for the direct C-ABI call
protected:
    cv::Mat mWorldPos, mDescriptor, mNormalVector;
    std::map<KeyFrame *, size_t> mObservations;
    int nObs;
    bool mbBad;
    float mfMinDistance, mfMaxDistance;
    MapPoint *mpReplaced;
};
}  // namespace ORB_SLAM2
