// ORBmatcher.h -- drop-in replacement for the reference's include/ORBmatcher.h (WChen09/My-SLAM).
//
// Same namespace, class name, constructor, constants and the signatures of all eleven Search* / Fuse methods of
// include/ORBmatcher.h:41-83, so the call sites of src/Tracking.cc (608-609, 774-777, 879-899, 1191-1199, 1364-1403, 1459, 1473),
// src/LocalMapping.cc (270, 491, 516) and src/LoopClosing.cc (266, 324, 376, 600) compile unchanged:
//     ORBmatcher(float nnratio = 0.6, bool checkOri = true)                                              :41
//     static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b)                                  :44
//     int SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th = 3)    :48
//     int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)            :53
//     int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint*> &sAlreadyFound, th, ORBdist) :57
//     int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, vpPoints, vpMatched, int th)                    :61
//     int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches)                :66
//     int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint*> &vpMatches12)               :67
//     int SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, windowSize = 10)     :70
//     int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vMatchedPairs, bOnlyStereo) :73
//     int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12, s12, R12, t12, th)                   :78
//     int Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th = 3.0)                :81
//     int Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, float th, vpReplacePoint)                           :84
// Each method gathers exactly the members its reference body reads (listed above it) from the maintainer's own Frame /
// KeyFrame / MapPoint classes into flat arrays, calls the C ABI of include/orbm.h (cv::Mat algebra on the host with OpenCV's
// arithmetic, window queries, candidate predicates and Hamming distances on the GPU) and writes the result back the way the
// reference does (Frame::mvpMapPoints / vpMapPointMatches / vnMatches12 / vpMatched / vMatchedPairs; for Fuse the reference's own
// Replace / AddObservation / AddMapPoint sequence, in the reference's order, on the caller's objects).  Like the reference header
// it includes "MapPoint.h", "KeyFrame.h" and "Frame.h": in an ORB-SLAM2 tree those are the tree's own; this repo's tests supply
// minimal classes with the same member names (tests/cxx/slam_shims/).
//
// Construction is free after the first: the reference builds a matcher on the stack at every call site, so the GPU handle
// (device buffers + a stream) comes from a thread-local pool and goes back to it in the destructor.  The reference's matcher has
// no size limit; a handle grows when a call brings more descriptors or candidates than it was created for (Configure() only sets
// the starting size).
#pragma once
#include <cstdint>
#include <cstring>
#include <set>
#include <string>
#include <utility>
#include <vector>

#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else
#include "orbx_cv_compat.h"
#endif
#include "../../include/orbm.h"

#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"

namespace ORB_SLAM2 {

namespace orbm_detail {
struct Scratch {                                 // marshalling buffers: they stay with the pooled handle, so a call allocates nothing once warm
    std::vector<uint8_t> u8_, in_, has_, desc_;
    std::vector<float> f0_, f1_, f2_, f3_, f4_, f5_, f6_, g0_, g1_, g2_, g3_;
    std::vector<uint8_t> v8_, w8_, desc2_;
    std::vector<int32_t> i0_, i1_, i2_, j0_, j1_, j2_, obs_, match_;
    std::vector<orbx_keypoint> kp_;
};
struct PooledHandle {
    orbm_matcher *m = nullptr; Scratch *s = nullptr;
    unsigned long gridFrame = ~0ul; const void *gridKeys = nullptr; int gridN = -1; int gridKind = -1;   // which frame's (0) / key frame's (1) grid the handle holds
};
struct HandlePool {                              // one per thread: handles are not re-entrant, threads never share one
    std::vector<PooledHandle> idle;
    ~HandlePool() { for (auto &h : idle) { orbm_destroy(h.m); delete h.s; } }
    static HandlePool &tls() { static thread_local HandlePool p; return p; }
};
inline int &pool_device() { static int d = 0; return d; }
inline int &pool_max_descriptors() { static int n = 8192; return n; }
inline int &pool_max_pairs() { static int n = 1 << 22; return n; }
}  // namespace orbm_detail

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri)
    {
        auto &pool = orbm_detail::HandlePool::tls();
        if (!pool.idle.empty()) { h_ = pool.idle.back(); pool.idle.pop_back(); }
    }
    ~ORBmatcher() { if (h_.m) orbm_detail::HandlePool::tls().idle.push_back(h_); }
    ORBmatcher(const ORBmatcher &o) : mfNNratio(o.mfNNratio), mbCheckOrientation(o.mbCheckOrientation) {}   // a copy takes its own handle on first use
    ORBmatcher &operator=(const ORBmatcher &o) { mfNNratio = o.mfNNratio; mbCheckOrientation = o.mbCheckOrientation; return *this; }

    // device and workspace sizes of handles created from now on (process-wide; call before the first matcher is used)
    static void Configure(int device, int maxDescriptors = 8192, int maxPairs = 1 << 22)
    {
        orbm_detail::pool_device() = device; orbm_detail::pool_max_descriptors() = maxDescriptors; orbm_detail::pool_max_pairs() = maxPairs;
    }

    // Computes the Hamming distance between two ORB descriptors (include/ORBmatcher.h:44)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orbm_distance(a.ptr<unsigned char>(), b.ptr<unsigned char>()); }

    // ---- include/ORBmatcher.h:48, src/ORBmatcher.cc:45-125 (Tracking::SearchLocalPoints) ----
    // reads: pMP->mbTrackInView, isBad(), mnTrackScaleLevel, mTrackViewCos, mTrackProjX / Y / XR, GetDescriptor();
    //        F.mvScaleFactors, F.mvpMapPoints[i]->Observations(), F.mvuRight, F.mDescriptors, F.mvKeysUn (grid, octave)
    // writes: F.mvpMapPoints[bestIdx] = pMP
    int SearchByProjection(Frame &F, const std::vector<MapPoint *> &vpMapPoints, const float th = 3)
    {
        const int n = (int)vpMapPoints.size(), nc = (int)F.mvKeysUn.size();
        if (!ready() || !grid(F)) return 0;
        orbm_detail::Scratch &S = *h_.s;
        S.u8_.assign(n, 0); S.f0_.assign(n, 0.f); S.f1_.assign(n, 0.f); S.f2_.assign(n, 0.f); S.f3_.assign(n, 0.f);
        S.i0_.assign(n, 0); S.i1_.assign(n, 0);
        S.desc_.assign((size_t)n * 32, 0);
        for (int i = 0; i < n; i++) {
            MapPoint *pMP = vpMapPoints[i];
            if (!pMP->mbTrackInView || pMP->isBad()) continue;
            S.u8_[i] = 1;
            S.f0_[i] = pMP->mTrackProjX; S.f1_[i] = pMP->mTrackProjY; S.f2_[i] = pMP->mTrackProjXR; S.f3_[i] = pMP->mTrackViewCos;
            S.i0_[i] = pMP->mnTrackScaleLevel; S.i1_[i] = pMP->Observations();
            copy_desc(pMP->GetDescriptor(), &S.desc_[(size_t)i * 32]);
        }
        cur_obs(F);
        S.match_.assign(nc, -1);
        int nm = 0;
        const bool stereo = (int)F.mvuRight.size() == nc && nc > 0;
        if (!ok(orbm_search_by_projection_map(h_.m, n, S.u8_.data(), S.f0_.data(), S.f1_.data(), stereo ? S.f2_.data() : nullptr, S.i0_.data(), S.f3_.data(),
                                              S.desc_.data(), S.i1_.data(), F.mvScaleFactors.data(), (int)F.mvScaleFactors.size(), kp(F.mvKeysUn),
                                              F.mDescriptors.ptr<unsigned char>(), stereo ? F.mvuRight.data() : nullptr, nc, th, mfNNratio,
                                              S.obs_.data(), S.match_.data(), &nm)))
            return 0;
        for (int i2 = 0; i2 < nc; i2++)
            if (S.match_[i2] >= 0) F.mvpMapPoints[i2] = vpMapPoints[S.match_[i2]];
        return nm;
    }

    // ---- include/ORBmatcher.h:53, src/ORBmatcher.cc:1328-1470 (Tracking::TrackWithMotionModel) ----
    // reads: CurrentFrame.mTcw, mb, mbf, fx, fy, cx, cy, mnMinX..mnMaxY, mvScaleFactors, mvpMapPoints[i]->Observations(), mvuRight,
    //        mDescriptors, mvKeysUn; LastFrame.mTcw, N, mvpMapPoints, mvbOutlier, mvKeys[i].octave, mvKeysUn[i].angle;
    //        pMP->GetWorldPos(), GetDescriptor()
    // writes: CurrentFrame.mvpMapPoints[bestIdx2] = pMP, and NULL for the entries the rotation check removes
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
    {
        const int n = LastFrame.N, nc = (int)CurrentFrame.mvKeysUn.size();
        if (!ready() || !grid(CurrentFrame)) return 0;
        orbm_detail::Scratch &S = *h_.s;
        S.u8_.assign(n, 0); S.f0_.assign((size_t)n * 3, 0.f); S.i1_.assign(n, 0); S.desc_.assign((size_t)n * 32, 0);
        S.kp_.assign(n, orbx_keypoint());
        for (int i = 0; i < n; i++) {
            S.kp_[i].octave = LastFrame.mvKeys[i].octave; S.kp_[i].angle = LastFrame.mvKeysUn[i].angle;
            MapPoint *pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            S.u8_[i] = 1;
            const cv::Mat x3Dw = pMP->GetWorldPos();
            for (int k = 0; k < 3; k++) S.f0_[(size_t)3 * i + k] = x3Dw.at<float>(k);
            S.i1_[i] = pMP->Observations();
            copy_desc(pMP->GetDescriptor(), &S.desc_[(size_t)i * 32]);
        }
        float Tcw[16], Tlw[16];
        pose(CurrentFrame.mTcw, Tcw); pose(LastFrame.mTcw, Tlw);
        const float bounds[4] = {(float)CurrentFrame.mnMinX, (float)CurrentFrame.mnMaxX, (float)CurrentFrame.mnMinY, (float)CurrentFrame.mnMaxY};
        cur_obs(CurrentFrame);
        S.match_.assign(nc, -1);
        int nm = 0;
        const bool stereo = (int)CurrentFrame.mvuRight.size() == nc && nc > 0;
        if (!ok(orbm_search_by_projection_last(h_.m, n, S.u8_.data(), S.f0_.data(), S.desc_.data(), S.i1_.data(), S.kp_.data(), Tcw, Tlw,
                                               CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, CurrentFrame.mb, CurrentFrame.mbf,
                                               bounds, CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(),
                                               kp(CurrentFrame.mvKeysUn), CurrentFrame.mDescriptors.ptr<unsigned char>(),
                                               stereo ? CurrentFrame.mvuRight.data() : nullptr, nc, th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                               S.obs_.data(), S.match_.data(), &nm)))
            return 0;
        for (int i2 = 0; i2 < nc; i2++) {
            if (S.match_[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[S.match_[i2]];
            else if (S.obs_[i2] < 0) CurrentFrame.mvpMapPoints[i2] = static_cast<MapPoint *>(NULL);   // assigned, then removed by the rotation check
        }
        return nm;
    }

    // ---- include/ORBmatcher.h:57, src/ORBmatcher.cc:1472-1599 (Tracking::Relocalization, after PnP) ----
    // reads: CurrentFrame.mTcw, fx, fy, cx, cy, mnMinX..mnMaxY, mvScaleFactors, mvpMapPoints, mDescriptors, mvKeysUn;
    //        pKF->GetMapPointMatches(), pKF->mvKeysUn[i].angle; pMP->isBad(), GetWorldPos(), GetMinDistanceInvariance(),
    //        GetMaxDistanceInvariance(), PredictScale(dist3D, &CurrentFrame), GetDescriptor()
    // writes: CurrentFrame.mvpMapPoints[bestIdx2] = pMP (free slots only), NULL again for what the rotation check removes
    int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint *> &sAlreadyFound, const float th, const int ORBdist)
    {
        const std::vector<MapPoint *> vpMPs = pKF->GetMapPointMatches();
        const int n = (int)vpMPs.size(), nc = (int)CurrentFrame.mvKeysUn.size();
        if (!ready() || !grid(CurrentFrame)) return 0;
        orbm_detail::Scratch &S = *h_.s;
        S.u8_.assign(n, 0); S.f0_.assign((size_t)n * 3, 0.f); S.f1_.assign(n, 0.f); S.f2_.assign(n, 0.f); S.f3_.assign(n, 0.f); S.f4_.assign(n, 0.f);
        S.i0_.assign(n, 0); S.desc_.assign((size_t)n * 32, 0); S.in_.assign(n, 0);
        for (int i = 0; i < n; i++) {
            MapPoint *pMP = vpMPs[i];
            S.f4_[i] = pKF->mvKeysUn[i].angle;
            if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
            S.u8_[i] = 1;
            const cv::Mat x3Dw = pMP->GetWorldPos();
            for (int k = 0; k < 3; k++) S.f0_[(size_t)3 * i + k] = x3Dw.at<float>(k);
        }
        float Tcw[16];
        pose(CurrentFrame.mTcw, Tcw);
        const float bounds[4] = {(float)CurrentFrame.mnMinX, (float)CurrentFrame.mnMaxX, (float)CurrentFrame.mnMinY, (float)CurrentFrame.mnMaxY};
        if (!ok(orbm_project_points(Tcw, CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy, bounds, S.f0_.data(), n,
                                    S.f1_.data(), S.f2_.data(), nullptr, S.f3_.data(), S.in_.data())))
            return 0;
        for (int i = 0; i < n; i++) {
            if (!S.u8_[i]) continue;
            MapPoint *pMP = vpMPs[i];
            const float dist3D = S.f3_[i];
            if (!S.in_[i] || dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) { S.u8_[i] = 0; continue; }   // :1507-1521
            S.i0_[i] = pMP->PredictScale(dist3D, &CurrentFrame);                                                                                 // :1523
            copy_desc(pMP->GetDescriptor(), &S.desc_[(size_t)i * 32]);
        }
        S.has_.assign(nc, 0);
        for (int i2 = 0; i2 < nc; i2++) S.has_[i2] = CurrentFrame.mvpMapPoints[i2] ? 1 : 0;
        S.match_.assign(nc, -1);
        int nm = 0;
        if (!ok(orbm_search_by_projection_kf(h_.m, n, S.u8_.data(), S.f1_.data(), S.f2_.data(), S.i0_.data(), S.desc_.data(), S.f4_.data(),
                                             CurrentFrame.mvScaleFactors.data(), (int)CurrentFrame.mvScaleFactors.size(), kp(CurrentFrame.mvKeysUn),
                                             CurrentFrame.mDescriptors.ptr<unsigned char>(), nc, th, ORBdist, mbCheckOrientation ? 1 : 0,
                                             S.has_.data(), S.match_.data(), &nm)))
            return 0;
        for (int i2 = 0; i2 < nc; i2++)
            if (S.match_[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = vpMPs[S.match_[i2]];
        return nm;
    }

    // ---- include/ORBmatcher.h:70, src/ORBmatcher.cc:159-288 (Tracking::TrackReferenceKeyFrame, Relocalization) ----
    // reads: pKF->GetMapPointMatches(), pKF->mFeatVec, pKF->mDescriptors, pKF->mvKeysUn[i].angle; pMP->isBad();
    //        F.N, F.mFeatVec, F.mDescriptors, F.mvKeys[i].angle
    // writes: vpMapPointMatches (F.N entries, MapPoint* or NULL)
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint *> &vpMapPointMatches)
    {
        const std::vector<MapPoint *> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));
        const int nkf = (int)vpMapPointsKF.size();
        if (!ready() || nkf == 0 || F.N == 0) return 0;
        orbm_detail::Scratch &S = *h_.s;
        S.u8_.assign(nkf, 0);
        for (int i = 0; i < nkf; i++) S.u8_[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
        flatten(pKF->mFeatVec, S.i0_, S.i1_, S.i2_);
        flatten(F.mFeatVec, S.j0_, S.j1_, S.j2_);
        S.match_.assign(F.N, -1);
        int nm = 0;
        if (!ok(orbm_search_by_bow(h_.m, pKF->mDescriptors.ptr<unsigned char>(), kp(pKF->mvKeysUn), nkf, S.u8_.data(),
                                   S.i0_.data(), S.i1_.data(), S.i2_.data(), (int)S.i0_.size(), F.mDescriptors.ptr<unsigned char>(), kp(F.mvKeys), F.N,
                                   S.j0_.data(), S.j1_.data(), S.j2_.data(), (int)S.j0_.size(), mfNNratio, mbCheckOrientation ? 1 : 0, S.match_.data(), &nm)))
            return 0;
        for (int i = 0; i < F.N; i++)
            if (S.match_[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[S.match_[i]];
        return nm;
    }

    // ---- include/ORBmatcher.h:74, src/ORBmatcher.cc:405-520 (Tracking::MonocularInitialization) ----
    // reads: F1.mvKeysUn, F1.mDescriptors, F2.mvKeysUn (grid), F2.mDescriptors, F2.mnMinX..mnMaxY; updates vbPrevMatched (:513-516)
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10)
    {
        vnMatches12 = std::vector<int>(F1.mvKeysUn.size(), -1);
        if (!ready() || F1.mvKeysUn.empty() || !grid(F2)) return 0;
        static_assert(sizeof(cv::Point2f) == 8 && sizeof(int) == sizeof(int32_t), "vbPrevMatched / vnMatches12 are handed over as they are");
        int nm = 0;
        if (!ok(orbm_search_for_initialization(h_.m, kp(F1.mvKeysUn), F1.mDescriptors.ptr<unsigned char>(), (int)F1.mvKeysUn.size(),
                                               kp(F2.mvKeysUn), F2.mDescriptors.ptr<unsigned char>(), (int)F2.mvKeysUn.size(),
                                               reinterpret_cast<float *>(vbPrevMatched.data()), windowSize, mfNNratio, mbCheckOrientation ? 1 : 0,
                                               vnMatches12.data(), &nm)))
            return 0;
        return nm;
    }

    // ---- include/ORBmatcher.h:61, src/ORBmatcher.cc:290-403 (LoopClosing::ComputeSim3, src/LoopClosing.cc:376) ----
    // reads: pKF->fx, fy, cx, cy, mnMinX..mnMaxY, mfGridElementWidthInv / HeightInv, mvScaleFactors, mvKeysUn, mDescriptors; Scw; vpMatched;
    //        pMP->isBad(), GetWorldPos(), GetNormal(), GetMin / MaxDistanceInvariance(), PredictScale(dist, pKF), GetDescriptor()
    // writes: vpMatched[bestIdx] = pMP
    int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, std::vector<MapPoint *> &vpMatched, int th)
    {
        const int n = (int)vpPoints.size(), nk = (int)pKF->mvKeysUn.size();
        if (!ready() || n == 0 || nk == 0 || !gridKF(pKF)) return 0;
        orbm_detail::Scratch &S = *h_.s;
        float Sc[16], T[16], Ow[3];
        pose(Scw, Sc);
        if (!ok(orbm_sim3_decompose(Sc, T, Ow))) return 0;                                     // :299-303
        std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());               // :306-307
        spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
        S.u8_.assign(n, 0);
        for (int i = 0; i < n; i++) S.u8_[i] = !(vpPoints[i]->isBad() || spAlreadyFound.count(vpPoints[i]));   // :317
        if (!kf_project(vpPoints, pKF, T, Ow, S)) return 0;
        kf_gate_and_level(vpPoints, pKF, S);
        S.has_.assign(nk, 0);
        for (int k = 0; k < nk; k++) S.has_[k] = vpMatched[k] ? 1 : 0;
        S.match_.assign(nk, -1);
        int nm = 0;
        if (!ok(orbm_search_by_projection_sim3(h_.m, n, S.u8_.data(), S.f1_.data(), S.f2_.data(), S.i0_.data(), S.desc_.data(), pKF->mvScaleFactors.data(),
                                               (int)pKF->mvScaleFactors.size(), kp(pKF->mvKeysUn), pKF->mDescriptors.ptr<unsigned char>(), nk, th,
                                               S.has_.data(), S.match_.data(), &nm)))
            return 0;
        for (int k = 0; k < nk; k++)
            if (S.match_[k] >= 0) vpMatched[k] = vpPoints[S.match_[k]];
        return nm;
    }

    // ---- include/ORBmatcher.h:67, src/ORBmatcher.cc:522-655 (LoopClosing::ComputeSim3, src/LoopClosing.cc:266) ----
    // reads: pKF1 / pKF2->mvKeysUn[i].angle, mFeatVec, GetMapPointMatches(), mDescriptors; pMP->isBad()
    // writes: vpMatches12 (one entry per feature of pKF1: a MapPoint of pKF2 or NULL)
    int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12)
    {
        const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        vpMatches12 = std::vector<MapPoint *>(vpMapPoints1.size(), static_cast<MapPoint *>(NULL));
        const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
        if (!ready() || n1 == 0 || n2 == 0) return 0;
        orbm_detail::Scratch &S = *h_.s;
        S.u8_.assign(n1, 0); S.v8_.assign(n2, 0);
        for (int i = 0; i < n1; i++) S.u8_[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();
        for (int i = 0; i < n2; i++) S.v8_[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();
        flatten(pKF1->mFeatVec, S.i0_, S.i1_, S.i2_);
        flatten(pKF2->mFeatVec, S.j0_, S.j1_, S.j2_);
        S.match_.assign(n1, -1);
        int nm = 0;
        if (!ok(orbm_search_by_bow_kf(h_.m, pKF1->mDescriptors.ptr<unsigned char>(), kp(pKF1->mvKeysUn), n1, S.u8_.data(), S.i0_.data(), S.i1_.data(),
                                      S.i2_.data(), (int)S.i0_.size(), pKF2->mDescriptors.ptr<unsigned char>(), kp(pKF2->mvKeysUn), n2, S.v8_.data(),
                                      S.j0_.data(), S.j1_.data(), S.j2_.data(), (int)S.j0_.size(), mfNNratio, mbCheckOrientation ? 1 : 0, S.match_.data(), &nm)))
            return 0;
        for (int i = 0; i < n1; i++)
            if (S.match_[i] >= 0) vpMatches12[i] = vpMapPoints2[S.match_[i]];
        return nm;
    }

    // ---- include/ORBmatcher.h:73, src/ORBmatcher.cc:657-823 (LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:270) ----
    // reads: pKF1 / pKF2->mFeatVec, N, GetMapPoint(idx), mvuRight, mvKeysUn, mDescriptors; pKF1->GetCameraCenter(); pKF2->GetRotation(),
    //        GetTranslation(), fx, fy, cx, cy, mvScaleFactors, mvLevelSigma2; F12
    // writes: vMatchedPairs
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo)
    {
        vMatchedPairs.clear();
        const int n1 = pKF1->N, n2 = pKF2->N;
        if (!ready() || n1 == 0 || n2 == 0) return 0;
        orbm_detail::Scratch &S = *h_.s;
        S.u8_.assign(n1, 0); S.v8_.assign(n2, 0);
        for (int i = 0; i < n1; i++) S.u8_[i] = pKF1->GetMapPoint(i) ? 1 : 0;
        for (int i = 0; i < n2; i++) S.v8_[i] = pKF2->GetMapPoint(i) ? 1 : 0;
        flatten(pKF1->mFeatVec, S.i0_, S.i1_, S.i2_);
        flatten(pKF2->mFeatVec, S.j0_, S.j1_, S.j2_);
        float Cw[3], T2w[16], F[9];
        vec3(pKF1->GetCameraCenter(), Cw);
        kf_pose(pKF2, T2w);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F[3 * r + c] = F12.at<float>(r, c);
        S.match_.assign(n1, -1);
        int nm = 0;
        if (!ok(orbm_search_for_triangulation(h_.m, kp(pKF1->mvKeysUn), pKF1->mDescriptors.ptr<unsigned char>(), n1, S.u8_.data(), pKF1->mvuRight.data(),
                                              S.i0_.data(), S.i1_.data(), S.i2_.data(), (int)S.i0_.size(), kp(pKF2->mvKeysUn),
                                              pKF2->mDescriptors.ptr<unsigned char>(), n2, S.v8_.data(), pKF2->mvuRight.data(), S.j0_.data(), S.j1_.data(),
                                              S.j2_.data(), (int)S.j0_.size(), Cw, T2w, pKF2->fx, pKF2->fy, pKF2->cx, pKF2->cy, F, pKF2->mvScaleFactors.data(),
                                              pKF2->mvLevelSigma2.data(), (int)pKF2->mvScaleFactors.size(), bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                              S.match_.data(), &nm)))
            return 0;
        vMatchedPairs.reserve(nm > 0 ? nm : 0);                                               // :812-820
        for (int i = 0; i < n1; i++)
            if (S.match_[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)S.match_[i]));
        return nm;
    }

    // ---- include/ORBmatcher.h:78, src/ORBmatcher.cc:1102-1326 (LoopClosing::ComputeSim3, src/LoopClosing.cc:324) ----
    // reads: pKF1->fx, fy, cx, cy; both key frames' GetRotation(), GetTranslation(), GetMapPointMatches(), mnMinX..mnMaxY, grid members,
    //        mvScaleFactors, mvKeysUn, mDescriptors; vpMatches12, pMP->GetIndexInKeyFrame(pKF2); pMP->isBad(), GetWorldPos(),
    //        GetMin / MaxDistanceInvariance(), PredictScale(dist, pKF), GetDescriptor(); s12, R12, t12
    // writes: vpMatches12[i1] = vpMapPoints2[idx2] where both directions agree
    int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12,
                     const float th)
    {
        const std::vector<MapPoint *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
        if (!ready() || N1 == 0 || N2 == 0) return 0;
        orbm_detail::Scratch &S = *h_.s;
        std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);         // :1129-1142
        for (int i = 0; i < N1; i++) {
            MapPoint *pMP = vpMatches12[i];
            if (pMP) {
                vbAlreadyMatched1[i] = true;
                int idx2 = pMP->GetIndexInKeyFrame(pKF2);
                if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
            }
        }
        float T1w[16], T2w[16], R[9], t[3], sR12[9], sR21[9], t21[3], b1[4], b2[4];
        kf_pose(pKF1, T1w); kf_pose(pKF2, T2w);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = R12.at<float>(r, c);
        vec3(t12, t);
        if (!ok(orbm_sim3_relative(s12, R, t, sR12, sR21, t21))) return 0;                      // :1119-1121
        kf_bounds(pKF1, b1); kf_bounds(pKF2, b2);
        // one side: MapPoints of key frame A searched in key frame B
        auto side = [&](const std::vector<MapPoint *> &pts, const std::vector<bool> &already, const float *TAw, const float *sR, const float *tt,
                        KeyFrame *pKFB, const float *bB, std::vector<uint8_t> &use, std::vector<float> &xw, std::vector<float> &u, std::vector<float> &v,
                        std::vector<float> &d3, std::vector<uint8_t> &okv, std::vector<int32_t> &lv, std::vector<uint8_t> &desc) -> bool {
            const int n = (int)pts.size();
            use.assign(n, 0); xw.assign((size_t)n * 3, 0.f); u.assign(n, 0.f); v.assign(n, 0.f); d3.assign(n, 0.f); okv.assign(n, 0); lv.assign(n, 0);
            desc.assign((size_t)n * 32, 0);
            for (int i = 0; i < n; i++) {
                MapPoint *pMP = pts[i];
                if (!pMP || already[i] || pMP->isBad()) continue;                              // :1152-1156
                use[i] = 1;
                vec3(pMP->GetWorldPos(), &xw[(size_t)3 * i]);
            }
            if (!ok(orbm_project_points_sim3(TAw, sR, tt, pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy, bB, xw.data(), n, u.data(), v.data(), d3.data(), okv.data())))
                return false;
            for (int i = 0; i < n; i++) {
                if (!use[i]) continue;
                MapPoint *pMP = pts[i];
                if (!okv[i] || d3[i] < pMP->GetMinDistanceInvariance() || d3[i] > pMP->GetMaxDistanceInvariance()) { use[i] = 0; continue; }   // :1163-1183
                lv[i] = pMP->PredictScale(d3[i], pKFB);                                        // :1186
                copy_desc(pMP->GetDescriptor(), &desc[(size_t)i * 32]);
            }
            return true;
        };
        if (!side(vpMapPoints1, vbAlreadyMatched1, T1w, sR21, t21, pKF2, b2, S.u8_, S.f0_, S.f1_, S.f2_, S.f3_, S.in_, S.i0_, S.desc_)) return 0;
        if (!side(vpMapPoints2, vbAlreadyMatched2, T2w, sR12, t, pKF1, b1, S.v8_, S.g0_, S.g1_, S.g2_, S.g3_, S.w8_, S.j0_, S.desc2_)) return 0;
        const orbm_kf_grid g1 = kf_grid_of(pKF1), g2 = kf_grid_of(pKF2);
        S.match_.assign(N1, -1);
        int nf = 0;
        h_.gridN = -1;
        if (!ok(orbm_search_by_sim3(h_.m, N1, S.u8_.data(), S.f1_.data(), S.f2_.data(), S.i0_.data(), S.desc_.data(), N2, S.v8_.data(), S.g1_.data(),
                                    S.g2_.data(), S.j0_.data(), S.desc2_.data(), kp(pKF1->mvKeysUn), pKF1->mDescriptors.ptr<unsigned char>(), N1, &g1,
                                    pKF1->mvScaleFactors.data(), (int)pKF1->mvScaleFactors.size(), kp(pKF2->mvKeysUn),
                                    pKF2->mDescriptors.ptr<unsigned char>(), N2, &g2, pKF2->mvScaleFactors.data(), (int)pKF2->mvScaleFactors.size(), th,
                                    S.match_.data(), &nf)))
            return 0;
        if (orbm_grid_count(h_.m) == N1) { h_.gridKind = 1; h_.gridFrame = (unsigned long)pKF1->mnId; h_.gridKeys = (const void *)pKF1->mvKeysUn.data(); h_.gridN = N1; }
        for (int i1 = 0; i1 < N1; i1++)
            if (S.match_[i1] >= 0) vpMatches12[i1] = vpMapPoints2[S.match_[i1]];               // :1319
        return nf;
    }

    // ---- include/ORBmatcher.h:81, src/ORBmatcher.cc:825-975 (LocalMapping::SearchInNeighbors, src/LocalMapping.cc:491, 516) ----
    // reads: pKF->GetRotation(), GetTranslation(), GetCameraCenter(), fx, fy, cx, cy, mbf, bounds, grid members, mvScaleFactors, mvInvLevelSigma2,
    //        mvKeysUn, mvuRight, mDescriptors, GetMapPoint(idx); pMP->isBad(), IsInKeyFrame(pKF), GetWorldPos(), GetNormal(),
    //        GetMin / MaxDistanceInvariance(), PredictScale(dist, pKF), GetDescriptor(), Observations()
    // writes (the reference's own sequence, :954-970): pMP->Replace(pMPinKF) / pMPinKF->Replace(pMP), or pMP->AddObservation + pKF->AddMapPoint
    // The searches do not depend on one another (nothing in :903-949 reads what :954-970 writes), so they run as one GPU launch on the
    // points that are eligible when the call starts; the bookkeeping then walks the points in order and re-evaluates :849 on the live
    // objects, as the reference's loop does (a point can only LOSE eligibility during the call: Replace makes it bad, AddObservation
    // puts it into pKF).
    int Fuse(KeyFrame *pKF, const std::vector<MapPoint *> &vpMapPoints, const float th = 3.0)
    {
        const int n = (int)vpMapPoints.size(), nk = (int)pKF->mvKeysUn.size();
        if (!ready() || n == 0 || nk == 0 || !gridKF(pKF)) return 0;
        orbm_detail::Scratch &S = *h_.s;
        float T[16], Ow[3];
        kf_pose(pKF, T);
        vec3(pKF->GetCameraCenter(), Ow);
        const float bf = pKF->mbf;
        S.u8_.assign(n, 0);
        for (int i = 0; i < n; i++) {
            MapPoint *pMP = vpMapPoints[i];
            S.u8_[i] = pMP && !(pMP->isBad() || pMP->IsInKeyFrame(pKF));                       // :846-850
        }
        if (!kf_project(vpMapPoints, pKF, T, Ow, S)) return 0;
        kf_gate_and_level(vpMapPoints, pKF, S);
        S.f6_.assign(n, 0.f);
        for (int i = 0; i < n; i++) S.f6_[i] = S.f1_[i] - bf * S.f4_[i];                       // ur = u - bf*invz :870
        S.match_.assign(n, -1);
        int cnt = 0;
        if (!ok(orbm_fuse(h_.m, n, S.u8_.data(), S.f1_.data(), S.f2_.data(), S.f6_.data(), S.i0_.data(), S.desc_.data(), pKF->mvScaleFactors.data(),
                          pKF->mvInvLevelSigma2.data(), (int)pKF->mvScaleFactors.size(), kp(pKF->mvKeysUn), pKF->mvuRight.data(),
                          pKF->mDescriptors.ptr<unsigned char>(), nk, th, S.match_.data(), &cnt)))
            return 0;
        int nFused = 0;
        for (int i = 0; i < n; i++) {
            MapPoint *pMP = vpMapPoints[i];
            if (!pMP) continue;
            if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;                              // :849 on the live objects
            const int bestIdx = S.match_[i];
            if (bestIdx < 0) continue;
            MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);                                     // :954-970
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
        return nFused;
    }

    // ---- include/ORBmatcher.h:84, src/ORBmatcher.cc:977-1100 (LoopClosing::SearchAndFuse, src/LoopClosing.cc:600) ----
    // reads: as Fuse above with Scw in place of the key frame's pose, pKF->GetMapPoints(); no stereo gate
    // writes: vpReplacePoint[iMP] = pMPinKF, or pMP->AddObservation + pKF->AddMapPoint (:1084-1094)
    int Fuse(KeyFrame *pKF, cv::Mat Scw, const std::vector<MapPoint *> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint)
    {
        const int n = (int)vpPoints.size(), nk = (int)pKF->mvKeysUn.size();
        if (!ready() || n == 0 || nk == 0 || !gridKF(pKF)) return 0;
        orbm_detail::Scratch &S = *h_.s;
        float Sc[16], T[16], Ow[3];
        pose(Scw, Sc);
        if (!ok(orbm_sim3_decompose(Sc, T, Ow))) return 0;                                     // :986-990
        const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();                       // :993
        S.u8_.assign(n, 0);
        for (int i = 0; i < n; i++) S.u8_[i] = !(vpPoints[i]->isBad() || spAlreadyFound.count(vpPoints[i]));   // :1005
        if (!kf_project(vpPoints, pKF, T, Ow, S)) return 0;
        kf_gate_and_level(vpPoints, pKF, S);
        S.match_.assign(n, -1);
        int cnt = 0;
        if (!ok(orbm_fuse_sim3(h_.m, n, S.u8_.data(), S.f1_.data(), S.f2_.data(), S.i0_.data(), S.desc_.data(), pKF->mvScaleFactors.data(),
                               (int)pKF->mvScaleFactors.size(), kp(pKF->mvKeysUn), pKF->mDescriptors.ptr<unsigned char>(), nk, th, S.match_.data(), &cnt)))
            return 0;
        int nFused = 0;
        for (int iMP = 0; iMP < n; iMP++) {
            const int bestIdx = S.match_[iMP];
            if (bestIdx < 0) continue;
            MapPoint *pMP = vpPoints[iMP];
            MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);                                     // :1084-1095
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
        return nFused;
    }

    // ---- the shared inner loop as primitives, for callers with candidate lists of their own (MapPoint::ComputeDistinctiveDescriptors, src/MapPoint.cc:271-285, is the dense case) ----
    // best / second-best over CSR candidate lists (candOff == nullptr: dense); outputs sized nq
    bool BestTwo(const cv::Mat &queries, const cv::Mat &train, const std::vector<int32_t> *candOff,
                 const std::vector<int32_t> *candIdx, std::vector<int32_t> &bestIdx,
                 std::vector<int32_t> &bestDist, std::vector<int32_t> &secondDist)
    {
        const int nq = queries.rows;
        bestIdx.assign(nq, -1); bestDist.assign(nq, 256); secondDist.assign(nq, 256);
        if (!ready()) return false;
        if (nq == 0) return true;
        return ok(orbm_best2(h_.m, queries.ptr<unsigned char>(), nq, train.ptr<unsigned char>(), train.rows, candOff ? candOff->data() : nullptr,
                             candIdx ? candIdx->data() : nullptr, bestIdx.data(), bestDist.data(), secondDist.data()));
    }
    // per-candidate distances, for the variants whose skip predicates depend on earlier matches
    bool Distances(const cv::Mat &queries, const cv::Mat &train, const std::vector<int32_t> &candOff,
                   const std::vector<int32_t> &candIdx, std::vector<int32_t> &dist)
    {
        dist.assign(candIdx.size(), 256);
        if (!ready()) return false;
        return ok(orbm_distances(h_.m, queries.ptr<unsigned char>(), queries.rows, train.ptr<unsigned char>(), train.rows, candOff.data(), candIdx.data(), dist.data()));
    }
    // rotation histogram + ComputeThreeMaxima cull (src/ORBmatcher.cc:236-246,266-284,1601-1642)
    static int RotationFilter(const std::vector<float> &angleQ, const std::vector<float> &angleT, std::vector<int32_t> &match12)
    {
        return orbm_rot_filter(angleQ.data(), angleT.data(), match12.data(), (int)match12.size());
    }

    static const int TH_LOW = ORBM_TH_LOW;              // src/ORBmatcher.cc:37-39
    static const int TH_HIGH = ORBM_TH_HIGH;
    static const int HISTO_LENGTH = ORBM_HISTO_LENGTH;
    bool Valid() { return ready(); }
    const std::string &LastError() const { return err_; }

protected:
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "orbx_keypoint mirrors cv::KeyPoint");
    static const orbx_keypoint *kp(const std::vector<cv::KeyPoint> &v) { return reinterpret_cast<const orbx_keypoint *>(v.data()); }
    bool ok(int rc) { if (rc != ORBX_OK) err_ = orbm_last_error(); return rc == ORBX_OK; }
    bool ready()
    {
        if (h_.m) return true;
        if (orbm_create(&h_.m, orbm_detail::pool_device(), orbm_detail::pool_max_descriptors(), orbm_detail::pool_max_descriptors(),
                        orbm_detail::pool_max_pairs()) != ORBX_OK) { err_ = orbm_last_error(); h_.m = nullptr; return false; }
        h_.s = new orbm_detail::Scratch();
        return true;
    }
    // Frame::AssignFeaturesToGrid of the searched frame, on the GPU; skipped while the handle still holds this frame's grid
    template <class FrameT> bool grid(const FrameT &F)
    {
        const int n = (int)F.mvKeysUn.size();
        if (h_.gridKind == 0 && h_.gridFrame == (unsigned long)F.mnId && h_.gridKeys == (const void *)F.mvKeysUn.data() && h_.gridN == n &&
            orbm_grid_count(h_.m) == n) return true;
        h_.gridN = -1;
        if (!ok(orbm_grid_build(h_.m, kp(F.mvKeysUn), n, (float)F.mnMinX, (float)F.mnMaxX, (float)F.mnMinY, (float)F.mnMaxY))) return false;
        h_.gridKind = 0; h_.gridFrame = (unsigned long)F.mnId; h_.gridKeys = (const void *)F.mvKeysUn.data(); h_.gridN = n;
        return true;
    }
    // A key frame's grid: its cells were filled by the Frame it was made of (Frame's static float image origin), its queries subtract
    // the key frame's own int mnMinX / mnMinY (include/KeyFrame.h:190-193, src/KeyFrame.cc:48-54, 569-606)
    static orbm_kf_grid kf_grid_of(const KeyFrame *pKF)
    {
        orbm_kf_grid g;
        g.assign_min_x = (float)Frame::mnMinX; g.assign_min_y = (float)Frame::mnMinY;
        g.inv_w = pKF->mfGridElementWidthInv; g.inv_h = pKF->mfGridElementHeightInv;
        g.query_min_x = (float)pKF->mnMinX; g.query_min_y = (float)pKF->mnMinY;
        return g;
    }
    bool gridKF(const KeyFrame *pKF)
    {
        const int n = (int)pKF->mvKeysUn.size();
        if (h_.gridKind == 1 && h_.gridFrame == (unsigned long)pKF->mnId && h_.gridKeys == (const void *)pKF->mvKeysUn.data() && h_.gridN == n &&
            orbm_grid_count(h_.m) == n) return true;
        h_.gridN = -1;
        const orbm_kf_grid g = kf_grid_of(pKF);
        if (!ok(orbm_grid_build_kf(h_.m, kp(pKF->mvKeysUn), n, g.assign_min_x, g.assign_min_y, g.inv_w, g.inv_h, g.query_min_x, g.query_min_y))) return false;
        h_.gridKind = 1; h_.gridFrame = (unsigned long)pKF->mnId; h_.gridKeys = (const void *)pKF->mvKeysUn.data(); h_.gridN = n;
        return true;
    }
    static void kf_bounds(const KeyFrame *pKF, float b[4]) { b[0] = (float)pKF->mnMinX; b[1] = (float)pKF->mnMaxX; b[2] = (float)pKF->mnMinY; b[3] = (float)pKF->mnMaxY; }
    static void kf_pose(KeyFrame *pKF, float T[16])            // [GetRotation() | GetTranslation()] as a row-major 4x4
    {
        const cv::Mat R = pKF->GetRotation(), t = pKF->GetTranslation();
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[4 * r + c] = R.at<float>(r, c); T[4 * r + 3] = t.at<float>(r); }
        T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    }
    static void vec3(const cv::Mat &v, float *out) { for (int k = 0; k < 3; k++) out[k] = v.at<float>(k); }
    // the MapPoint half of the projection block (:343-357, :872-887, :1031-1046): distance-invariance test, PredictScale(dist, pKF), descriptor
    void kf_gate_and_level(const std::vector<MapPoint *> &pts, KeyFrame *pKF, orbm_detail::Scratch &S)
    {
        const int n = (int)pts.size();
        S.i0_.assign(n, 0); S.desc_.assign((size_t)n * 32, 0);
        for (int i = 0; i < n; i++) {
            if (!S.u8_[i]) continue;
            MapPoint *pMP = pts[i];
            const float dist = S.f3_[i];
            if (!S.in_[i] || dist < pMP->GetMinDistanceInvariance() || dist > pMP->GetMaxDistanceInvariance()) { S.u8_[i] = 0; continue; }
            S.i0_[i] = pMP->PredictScale(dist, pKF);
            copy_desc(pMP->GetDescriptor(), &S.desc_[(size_t)i * 32]);
        }
    }
    // world position and normal of the points flagged in S.u8_, then orbm_project_points_kf: u, v -> f1_, f2_; 1/z -> f4_; dist -> f3_; ok -> in_
    bool kf_project(const std::vector<MapPoint *> &pts, KeyFrame *pKF, const float *T, const float *Ow, orbm_detail::Scratch &S)
    {
        const int n = (int)pts.size();
        S.f0_.assign((size_t)n * 3, 0.f); S.f5_.assign((size_t)n * 3, 0.f);
        S.f1_.assign(n, 0.f); S.f2_.assign(n, 0.f); S.f3_.assign(n, 0.f); S.f4_.assign(n, 0.f); S.in_.assign(n, 0);
        for (int i = 0; i < n; i++) {
            if (!S.u8_[i]) continue;
            vec3(pts[i]->GetWorldPos(), &S.f0_[(size_t)3 * i]);
            vec3(pts[i]->GetNormal(), &S.f5_[(size_t)3 * i]);
        }
        float b[4];
        kf_bounds(pKF, b);
        return ok(orbm_project_points_kf(T, Ow, pKF->fx, pKF->fy, pKF->cx, pKF->cy, b, S.f0_.data(), S.f5_.data(), n, S.f1_.data(), S.f2_.data(),
                                         S.f4_.data(), S.f3_.data(), S.in_.data()));
    }
    template <class FrameT> void cur_obs(const FrameT &F)      // -1 = NULL slot, else the point's Observations() (:85-87, :1403-1405)
    {
        orbm_detail::Scratch &S = *h_.s;
        const int nc = (int)F.mvKeysUn.size();
        S.obs_.assign(nc, -1);
        for (int i = 0; i < nc; i++)
            if (F.mvpMapPoints[i]) S.obs_[i] = F.mvpMapPoints[i]->Observations();
    }
    static void pose(const cv::Mat &T, float out[16])
    {
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) out[4 * r + c] = T.at<float>(r, c);
    }
    static void copy_desc(const cv::Mat &d, unsigned char *dst) { memcpy(dst, d.ptr<unsigned char>(), 32); }
    template <class FV> static void flatten(const FV &fv, std::vector<int32_t> &node, std::vector<int32_t> &off, std::vector<int32_t> &idx)
    {
        node.clear(); off.assign(1, 0); idx.clear();             // DBoW2::FeatureVector = std::map<NodeId, std::vector<unsigned int>>: ascending ids
        for (typename FV::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            node.push_back((int32_t)it->first);
            for (size_t k = 0; k < it->second.size(); k++) idx.push_back((int32_t)it->second[k]);
            off.push_back((int32_t)idx.size());
        }
    }

    float mfNNratio;
    bool mbCheckOrientation;
    orbm_detail::PooledHandle h_;
    std::string err_;
};

}  // namespace ORB_SLAM2
