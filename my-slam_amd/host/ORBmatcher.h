// ORBmatcher.h -- drop-in replacement for the reference's include/ORBmatcher.h (WChen09/My-SLAM), Tracking-thread part.
//
// Same namespace, class name, constructor and method signatures as include/ORBmatcher.h:41-73 for the five matchers the
// Tracking thread calls (src/Tracking.cc:608-609, 774-777, 879-899, 1191-1199, 1364-1403, 1459, 1473), so those call sites
// compile unchanged:
//     ORBmatcher(float nnratio = 0.6, bool checkOri = true)                                              :41
//     static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b)                                  :44
//     int SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th = 3)    :48
//     int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)            :53
//     int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const std::set<MapPoint*> &sAlreadyFound, th, ORBdist) :57
//     int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches)                :70
//     int SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, windowSize = 10)     :74
// Each method gathers exactly the members its reference body reads (listed above it) from the maintainer's own Frame /
// KeyFrame / MapPoint classes into flat arrays, calls the C ABI of include/orbm.h (window queries and Hamming distances
// on the GPU, the reference's sequential scan on those distances) and writes the result back the way the reference does
// (Frame::mvpMapPoints / vpMapPointMatches / vnMatches12).  Like the reference header it includes "MapPoint.h",
// "KeyFrame.h" and "Frame.h": in an ORB-SLAM2 tree those are the tree's own; this repo's tests supply minimal classes
// with the same member names (tests/cxx/slam_shims/).  The LocalMapping / LoopClosing matchers (SearchForTriangulation,
// Fuse, SearchBySim3, SearchByBoW(KF, KF), SearchByProjection(KF, Scw, ...)) are outside this path;
// a maintainer keeps the reference bodies for them and may swap their inner loops for BestTwo()/Distances() below.
//
// Construction is free after the first: the reference builds a matcher on the stack at every call site, so the GPU handle
// (device buffers + a stream) comes from a thread-local pool and goes back to it in the destructor.
#pragma once
#include <cstdint>
#include <cstring>
#include <set>
#include <string>
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
    std::vector<float> f0_, f1_, f2_, f3_, f4_;
    std::vector<int32_t> i0_, i1_, i2_, j0_, j1_, j2_, obs_, match_;
    std::vector<orbx_keypoint> kp_;
};
struct PooledHandle {
    orbm_matcher *m = nullptr; Scratch *s = nullptr;
    unsigned long gridFrame = ~0ul; const void *gridKeys = nullptr; int gridN = -1;   // which frame's grid the handle holds
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

    // ---- the shared inner loop as primitives, for the LocalMapping / LoopClosing matchers a maintainer keeps on the host ----
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
        if (h_.gridFrame == (unsigned long)F.mnId && h_.gridKeys == (const void *)F.mvKeysUn.data() && h_.gridN == n) return true;
        if (!ok(orbm_grid_build(h_.m, kp(F.mvKeysUn), n, (float)F.mnMinX, (float)F.mnMaxX, (float)F.mnMinY, (float)F.mnMaxY))) return false;
        h_.gridFrame = (unsigned long)F.mnId; h_.gridKeys = (const void *)F.mvKeysUn.data(); h_.gridN = n;
        return true;
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
