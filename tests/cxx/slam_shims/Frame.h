// Frame.h -- repo-authored minimal Frame for the adapter tests (member names as in the reference's include/Frame.h:100-190; only
// what ORB_SLAM2::ORBextractor / ORBmatcher read or write).  The monocular constructor follows the shape of src/Frame.cc:172-224:
// scale tables from the extractor's getters, ExtractORB, N, mvKeysUn (no distortion), empty MapPoint slots.
#pragma once
#include <cmath>
#include <vector>
#include "KeyFrame.h"
#include "MapPoint.h"
#include "ORBextractor.h"

namespace ORB_SLAM2 {
class Frame {
public:
    Frame() : mpORBextractorLeft(nullptr), mpORBextractorRight(nullptr), mbf(0), mb(0), N(0), mnId(0), mnScaleLevels(0), mfScaleFactor(1), mfLogScaleFactor(0) {}
    // Constructor for Monocular cameras (src/Frame.cc:172-224)
    Frame(const cv::Mat &imGray, ORBextractor *extractor, float fx_, float fy_, float cx_, float cy_, float bf)
        : mpORBextractorLeft(extractor), mpORBextractorRight(nullptr), mbf(bf), mb(bf / fx_)
    {
        mnId = nNextId++;
        mnScaleLevels = mpORBextractorLeft->GetLevels();
        mfScaleFactor = mpORBextractorLeft->GetScaleFactor();
        mfLogScaleFactor = log(mfScaleFactor);
        mvScaleFactors = mpORBextractorLeft->GetScaleFactors();
        mvInvScaleFactors = mpORBextractorLeft->GetInverseScaleFactors();
        mvLevelSigma2 = mpORBextractorLeft->GetScaleSigmaSquares();
        mvInvLevelSigma2 = mpORBextractorLeft->GetInverseScaleSigmaSquares();
        ExtractORB(0, imGray);
        N = (int)mvKeys.size();
        mvKeysUn = mvKeys;                                       // UndistortKeyPoints with k1 == 0 (src/Frame.cc:406-410)
        mvuRight = std::vector<float>(N, -1);
        mvDepth = std::vector<float>(N, -1);
        mvpMapPoints = std::vector<MapPoint *>(N, static_cast<MapPoint *>(NULL));
        mvbOutlier = std::vector<bool>(N, false);
        fx = fx_; fy = fy_; cx = cx_; cy = cy_;
        mnMinX = 0.0f; mnMaxX = (float)imGray.cols; mnMinY = 0.0f; mnMaxY = (float)imGray.rows;   // ComputeImageBounds, no distortion (:455-461)
    }
    // Extract ORB on the image. 0 for left image and 1 for right image.  (src/Frame.cc:247-253, the call expressions verbatim)
    void ExtractORB(int flag, const cv::Mat &im)
    {
        if (flag == 0)
            (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
        else
            (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
    }
    void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); }

    ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    inline static float fx = 0, fy = 0, cx = 0, cy = 0;
    float mbf, mb;
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    cv::Mat mTcw;
    inline static long unsigned int nNextId = 0;
    long unsigned int mnId;
    int mnScaleLevels;
    float mfScaleFactor, mfLogScaleFactor;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    inline static float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    inline static float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
};

// MapPoint::PredictScale(dist, Frame*) with the reference's arithmetic (src/MapPoint.cc:402-417): float ratio, log of a float, ceil
inline int MapPoint::PredictScale(const float &currentDist, Frame *pF)
{
    float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
    return nScale;
}
}  // namespace ORB_SLAM2
