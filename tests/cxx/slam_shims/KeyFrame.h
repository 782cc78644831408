// KeyFrame.h -- repo-authored minimal KeyFrame for the adapter tests (member names and types as in the reference's
// include/KeyFrame.h:48-193; only what ORB_SLAM2::ORBmatcher's methods read or call, bodies written here).
#pragma once
#include <map>
#include <set>
#include <vector>
#include "MapPoint.h"

#ifndef ORBX_SHIM_DBOW2
#define ORBX_SHIM_DBOW2
namespace DBoW2 {              // Thirdparty/DBoW2/DBoW2/FeatureVector.h: a std::map<NodeId, std::vector<unsigned int>>
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
}
#endif

namespace ORB_SLAM2 {
class KeyFrame {
public:
    // the Tracking-thread tests' constructor: keypoints, descriptors, FeatureVector, MapPoints
    KeyFrame(long unsigned int id, const std::vector<cv::KeyPoint> &keysUn, const cv::Mat &descriptors, const DBoW2::FeatureVector &fv,
             const std::vector<MapPoint *> &mapPoints)
        : mnId(id), mfGridElementWidthInv(0), mfGridElementHeightInv(0), fx(0), fy(0), cx(0), cy(0), mbf(0), N((int)keysUn.size()), mvKeysUn(keysUn),
          mvuRight(keysUn.size(), -1.f), mDescriptors(descriptors.clone()), mFeatVec(fv), mnScaleLevels(0), mfLogScaleFactor(0), mnMinX(0), mnMinY(0),
          mnMaxX(0), mnMaxY(0), mvpMapPoints(mapPoints) {}
    // the LocalMapping / LoopClosing tests' constructor: what KeyFrame::KeyFrame(Frame&, ...) copies from the frame (src/KeyFrame.cc:30-46);
    // the float image bounds are truncated to int exactly as the member initialisers mnMinX(F.mnMinX) ... do
    KeyFrame(long unsigned int id, const std::vector<cv::KeyPoint> &keysUn, const std::vector<float> &uRight, const cv::Mat &descriptors,
             const DBoW2::FeatureVector &fv, const std::vector<MapPoint *> &mapPoints, float fx_, float fy_, float cx_, float cy_, float bf,
             float gridWInv, float gridHInv, float minX, float minY, float maxX, float maxY, const std::vector<float> &scaleFactors,
             const std::vector<float> &levelSigma2, const std::vector<float> &invLevelSigma2, float logScaleFactor)
        : mnId(id), mfGridElementWidthInv(gridWInv), mfGridElementHeightInv(gridHInv), fx(fx_), fy(fy_), cx(cx_), cy(cy_), mbf(bf), N((int)keysUn.size()),
          mvKeysUn(keysUn), mvuRight(uRight), mDescriptors(descriptors.clone()), mFeatVec(fv), mnScaleLevels((int)scaleFactors.size()),
          mfLogScaleFactor(logScaleFactor), mvScaleFactors(scaleFactors), mvLevelSigma2(levelSigma2), mvInvLevelSigma2(invLevelSigma2),
          mnMinX(minX), mnMinY(minY), mnMaxX(maxX), mnMaxY(maxY), mvpMapPoints(mapPoints) {}

    void SetPose(const cv::Mat &Tcw_)         // src/KeyFrame.cc:72-92; Ow = -Rwc*tcw is a 3x3 * 3x1 cv::Mat product: the test passes it in
    {
        Tcw = Tcw_.clone();
    }
    void SetCameraCenter(const cv::Mat &Ow_) { Ow = Ow_.clone(); }
    cv::Mat GetRotation() { cv::Mat R(3, 3, CV_32F); for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R.at<float>(r, c) = Tcw.at<float>(r, c); return R; }
    cv::Mat GetTranslation() { cv::Mat t(3, 1, CV_32F); for (int r = 0; r < 3; r++) t.at<float>(r) = Tcw.at<float>(r, 3); return t; }
    cv::Mat GetCameraCenter() { return Ow.clone(); }

    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    std::set<MapPoint *> GetMapPoints()       // src/KeyFrame.cc:236-249
    {
        std::set<MapPoint *> s;
        for (MapPoint *p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
        return s;
    }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
    void EraseMapPointMatch(const size_t &idx) { mvpMapPoints[idx] = static_cast<MapPoint *>(NULL); }
    void ReplaceMapPointMatch(const size_t &idx, MapPoint *pMP) { mvpMapPoints[idx] = pMP; }
    bool IsInImage(const float &x, const float &y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }

    long unsigned int mnId;
    const float mfGridElementWidthInv, mfGridElementHeightInv;
    const float fx, fy, cx, cy, mbf;
    const int N;
    const std::vector<cv::KeyPoint> mvKeysUn;
    const std::vector<float> mvuRight;
    const cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    const int mnScaleLevels;
    const float mfLogScaleFactor;
    const std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    const int mnMinX, mnMinY, mnMaxX, mnMaxY;
protected:
    cv::Mat Tcw, Ow;
    std::vector<MapPoint *> mvpMapPoints;
};

// MapPoint::PredictScale(dist, KeyFrame*) with the reference's arithmetic (src/MapPoint.cc:387-400)
inline int MapPoint::PredictScale(const float &currentDist, KeyFrame *pKF)
{
    float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / pKF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
    return nScale;
}
inline void MapPoint::AddObservation(KeyFrame *pKF, size_t idx)      // src/MapPoint.cc:98-109
{
    if (mObservations.count(pKF)) return;
    mObservations[pKF] = idx;
    if (pKF->mvuRight[idx] >= 0) nObs += 2; else nObs++;
}
inline void MapPoint::Replace(MapPoint *pMP)                          // src/MapPoint.cc:177-215 without the Map / descriptor bookkeeping
{
    if (pMP->mnId == this->mnId) return;
    std::map<KeyFrame *, size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    mpReplaced = pMP;
    for (auto &o : obs) {
        KeyFrame *pKF = o.first;
        if (!pMP->IsInKeyFrame(pKF)) { pKF->ReplaceMapPointMatch(o.second, pMP); pMP->AddObservation(pKF, o.second); }
        else pKF->EraseMapPointMatch(o.second);
    }
}
}  // namespace ORB_SLAM2
