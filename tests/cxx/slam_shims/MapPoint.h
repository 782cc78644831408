// MapPoint.h -- repo-authored minimal MapPoint for the adapter tests: only the members ORB_SLAM2::ORBmatcher's Tracking-thread
// methods read (names and signatures as in the reference's include/MapPoint.h:40-150, bodies written here).  It is test
// scaffolding for my-slam_amd/host/ORBmatcher.h, not a rebuild of the reference's map.
#pragma once
#include <cmath>
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
          mWorldPos(Pos.clone()), mDescriptor(descriptor.clone()), nObs(observations), mbBad(false), mfMinDistance(minDistance), mfMaxDistance(maxDistance) {}
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    int Observations() { return nObs; }
    bool isBad() { return mbBad; }
    void SetBadFlag() { mbBad = true; }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int PredictScale(const float &currentDist, Frame *pF);      // defined in Frame.h (needs the complete Frame)

    // Variables used by the tracking (public in the reference too)
    float mTrackProjX, mTrackProjY, mTrackProjXR;
    bool mbTrackInView;
    int mnTrackScaleLevel;
    float mTrackViewCos;

    float MaxDistance() const { return mfMaxDistance; }        // test-side access for the direct C-ABI call
protected:
    cv::Mat mWorldPos, mDescriptor;
    int nObs;
    bool mbBad;
    float mfMinDistance, mfMaxDistance;
};
}  // namespace ORB_SLAM2
