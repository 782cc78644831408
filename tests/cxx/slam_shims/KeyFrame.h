// KeyFrame.h -- repo-authored minimal KeyFrame for the adapter tests (member names as in the reference's include/KeyFrame.h:89,
// 124, 164-171; only what ORBmatcher::SearchByBoW(KF, F) and SearchByProjection(F, KF, ...) read).
#pragma once
#include <map>
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
    KeyFrame(long unsigned int id, const std::vector<cv::KeyPoint> &keysUn, const cv::Mat &descriptors, const DBoW2::FeatureVector &fv,
             const std::vector<MapPoint *> &mapPoints)
        : mnId(id), mvKeysUn(keysUn), mDescriptors(descriptors.clone()), mFeatVec(fv), mvpMapPoints(mapPoints) {}
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    long unsigned int mnId;
    const std::vector<cv::KeyPoint> mvKeysUn;
    const cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
protected:
    std::vector<MapPoint *> mvpMapPoints;
};
}  // namespace ORB_SLAM2
