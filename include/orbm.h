/*
 * orbm.h -- C ABI of the MI355X-native ORB descriptor matcher primitives (liborbx.so).
 *
 * Drop-in boundary for the Hamming inner loops of the reference's ORB_SLAM2::ORBmatcher
 * (citations relative to WChen09/My-SLAM):
 *   include/ORBmatcher.h:44, src/ORBmatcher.cc:1647-1663   static DescriptorDistance(a, b)
 *   src/ORBmatcher.cc:201-232 (SearchByBoW), :432-471 (SearchForInitialization), :76-125, :370-398,
 *   :1397-1430, :1535-1558 (SearchByProjection), :715-756 (SearchForTriangulation), :901-949,
 *   :1060-1079 (Fuse), :1199-1219, :1279-1299 (SearchBySim3), src/Frame.cc:522-549 (stereo):
 *       every one is "best / second-best DescriptorDistance over a candidate index list".
 *   src/ORBmatcher.cc:1601-1642 ComputeThreeMaxima + the 30-bin rotation histogram (:236-246).
 * The reference has no whole-frame brute-force matcher: dense N x M matching is the
 * degenerate case "one candidate list = every train descriptor".
 *
 * The geometry / MapPoint bookkeeping around these loops stays host C++ in the caller
 * (my-slam_amd/host/ORBmatcher.h shows the adapter); the variants whose skip predicates depend on
 * earlier matches (e.g. :444-445) take the per-candidate distances from orbm_distances() and run
 * their tiny selection loop on the host, so their results stay identical.
 *
 * Descriptors are rows of 32 bytes (cv::Mat N x 32 CV_8U, contiguous).  All functions return 0 or
 * a negative orbx_status (orbx.h); text via orbm_last_error().  A matcher handle owns a HIP stream
 * and staging buffers; handles are independent, one handle is not re-entrant.
 */
#ifndef ORBM_H
#define ORBM_H

#include <stddef.h>
#include <stdint.h>
#include "orbx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORBM_TH_HIGH 100      /* src/ORBmatcher.cc:37 */
#define ORBM_TH_LOW 50        /* src/ORBmatcher.cc:38 */
#define ORBM_HISTO_LENGTH 30  /* src/ORBmatcher.cc:39 */

typedef struct orbm_matcher orbm_matcher;

int orbm_create(orbm_matcher **out, int device, int max_queries, int max_train, int max_pairs);
void orbm_destroy(orbm_matcher *m);

/* ORBmatcher::DescriptorDistance for one pair (host popcount; one pair is not GPU work). */
int orbm_distance(const uint8_t a[32], const uint8_t b[32]);

/*
 * best / second-best over candidate lists, on the GPU.  Host buffers.
 *   cand_off[nq+1], cand_idx[cand_off[nq]] : CSR lists of train indices per query, scanned in list
 *   order; cand_off == NULL means dense (each query against train 0..nt-1).
 * Semantics of src/ORBmatcher.cc:201-226: bestDist1 = bestDist2 = 256, bestIdx = -1; strict '<'
 * updates (first candidate wins a tie; a tie with the best becomes the second best).
 */
int orbm_best2(orbm_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt,
               const int32_t *cand_off, const int32_t *cand_idx,
               int32_t *best_idx, int32_t *best_d, int32_t *second_d);

/* Per-candidate distances dist[cand_off[nq]] (dense: dist[nq*nt], row-major) for the search
 * variants whose skip predicates interleave with the scan. */
int orbm_distances(orbm_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt,
                   const int32_t *cand_off, const int32_t *cand_idx, int32_t *dist);

/*
 * Device-resident, batched, dense: pair b matches d_q[b] (d_nq[b] rows) against d_t[b] (d_nt[b]
 * rows); descriptor blocks are [nbatch][cap][32] as written by orbx_extract_batch_device, counts
 * live on the device.  Outputs [nbatch][cap].  Asynchronous on hip_stream (NULL = handle stream).
 */
int orbm_best2_batch_device(orbm_matcher *m, const uint8_t *d_q, const int32_t *d_nq,
                            const uint8_t *d_t, const int32_t *d_nt, int cap, int nbatch,
                            int32_t *d_best_idx, int32_t *d_best_d, int32_t *d_second_d,
                            void *hip_stream);

/*
 * Same plus the SearchByBoW acceptance (:228-232: best <= th and best < nnratio * second) and the
 * rotation-consistency filter (:236-246, :266-284) using the keypoints' angles.  d_match12[b][i] =
 * train index or -1; d_nmatches[b] = surviving matches.  BASELINE config 3's "match against the
 * previous frame".
 */
int orbm_match_batch_device(orbm_matcher *m, const uint8_t *d_q, const orbx_keypoint *d_kq,
                            const int32_t *d_nq, const uint8_t *d_t, const orbx_keypoint *d_kt,
                            const int32_t *d_nt, int cap, int nbatch, int th, float nnratio,
                            int check_orientation, int32_t *d_match12, int32_t *d_nmatches,
                            void *hip_stream);

/*
 * ---- "next" row N1: the Frame grid behind every windowed search ----
 * src/Frame.cc:230-245 AssignFeaturesToGrid, :382-392 PosInGrid, :327-380 GetFeaturesInArea, and the
 * scan they feed in ORBmatcher::SearchByProjection (src/ORBmatcher.cc:1397-1430) / SearchForInitialization
 * (:425-457).  FRAME_GRID_COLS x FRAME_GRID_ROWS = 64 x 48 (include/Frame.h:37-38).
 *
 * orbm_grid_build   builds the 64x48 cell lists of one frame on the GPU from its undistorted keypoints
 *                   (mvKeysUn; with the shipped calibration k1 = 0 that is mvKeys, src/Frame.cc:406-410).
 *                   Bounds are mnMinX/mnMaxX/mnMinY/mnMaxY (:436-463).  The grid stays in the handle.
 * orbm_features_in_area   GetFeaturesInArea for nq windows in one call: CSR lists in the reference's
 *                   order (cell columns, cell rows, push_back order).  Returns the total count.
 * orbm_search_area_best2  the fused form: window query + best / second-best DescriptorDistance with
 *                   the strict-'<' tie rules; skip[i] != 0 drops train keypoint i (the reference's
 *                   `continue` predicates).  *_device takes device pointers and does not synchronise.
 */
/*
 * ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520), the monocular initialiser's matcher.  Frame 2's grid
 * must be in the handle (orbm_grid_build on F2.mvKeysUn).  For every level-0 keypoint of frame 1 the window
 * GetFeaturesInArea(vbPrevMatched[i1], windowSize, 0, 0) and the candidates' distances come from the GPU (one
 * orbm_features_in_area + one orbm_distances pass over all windows); the scan itself is sequential in the reference
 * (vMatchedDistance / vnMatches21 let a later keypoint steal an earlier one's match, :444, :463-467) and runs on the host
 * on those distances, as do the rotation histogram -- whose bins keep the entries of stolen matches, as the reference's
 * rotHist does -- and the cull (:489-510).  prev_matched (2 * n1 floats) is vbPrevMatched, updated in place (:513-516);
 * matches12[n1] = vnMatches12; *nmatches = the return value.
 */
int orbm_search_for_initialization(orbm_matcher *m, const orbx_keypoint *kps1, const uint8_t *desc1, int n1,
                                   const orbx_keypoint *kps2, const uint8_t *desc2, int n2,
                                   float *prev_matched, int window_size, float nnratio, int check_orientation,
                                   int32_t *matches12, int *nmatches);

/*
 * ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (src/ORBmatcher.cc:1328-1470), the
 * matcher of Tracking::TrackWithMotionModel.  The current frame's grid must be in the handle (orbm_grid_build on its mvKeysUn).
 *   last frame, per feature i:  has_point[i] = pMP && !mvbOutlier[i];  xw = pMP->GetWorldPos();  mp_desc = pMP->GetDescriptor();
 *                               mp_obs[i] = pMP->Observations();  kps_last[i] = octave of mvKeys[i], angle of mvKeysUn[i]
 *   current frame:              Tcw / Tlw = CurrentFrame.mTcw / LastFrame.mTcw (row-major 4x4 float);  bounds = mnMinX, mnMaxX,
 *                               mnMinY, mnMaxY;  scale_factors = mvScaleFactors;  u_right = mvuRight or NULL
 *   cur_obs[i2]   in/out: -1 where mvpMapPoints[i2] is NULL, else that point's Observations() (> 0 keeps its place, :1403-1405)
 *   cur_match[i2] out: the last-frame feature whose MapPoint this call put into mvpMapPoints[i2], or -1
 * The projections (cv::Mat algebra as cv::gemm's small-matrix path evaluates it: float accumulation, alpha / beta in double) run on the host, all windows and
 * all candidate distances in two GPU passes, and the scan on the host: it is sequential in the reference (an assignment
 * blocks or is overwritten by later ones; *nmatches counts assignments, the rotation cull decrements once per histogram
 * entry, exactly as :1428-1429 and :1458-1462 do).
 */
int orbm_search_by_projection_last(orbm_matcher *m, int n_last, const uint8_t *has_point, const float *xw, const uint8_t *mp_desc,
                                   const int32_t *mp_obs, const orbx_keypoint *kps_last, const float *Tcw, const float *Tlw,
                                   float fx, float fy, float cx, float cy, float mb, float mbf, const float bounds[4],
                                   const float *scale_factors, int nlevels, const orbx_keypoint *kps_cur, const uint8_t *desc_cur,
                                   const float *u_right, int n_cur, float th, int mono, int check_orientation,
                                   int32_t *cur_obs, int32_t *cur_match, int *nmatches);

/*
 * ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (src/ORBmatcher.cc:45-125), the matcher of
 * Tracking::SearchLocalPoints (every frame, against the local map).  Per MapPoint: in_view = mbTrackInView && !isBad(),
 * proj_x / proj_y / proj_xr = mTrackProjX / Y / XR (proj_xr may be NULL when u_right is), pred_level = mnTrackScaleLevel,
 * view_cos = mTrackViewCos (RadiusByViewingCos :127-133), mp_desc = GetDescriptor(), mp_obs = Observations().  Frame side and
 * the cur_obs / cur_match convention as in orbm_search_by_projection_last (cur_match[i2] = MapPoint index).  Best and
 * second-best with their octaves, TH_HIGH, and the ratio test only when both sit on the same level (:115-118).
 */
int orbm_search_by_projection_map(orbm_matcher *m, int n_mp, const uint8_t *in_view, const float *proj_x, const float *proj_y,
                                  const float *proj_xr, const int32_t *pred_level, const float *view_cos, const uint8_t *mp_desc,
                                  const int32_t *mp_obs, const float *scale_factors, int nlevels, const orbx_keypoint *kps_cur,
                                  const uint8_t *desc_cur, const float *u_right, int n_cur, float th, float nnratio,
                                  int32_t *cur_obs, int32_t *cur_match, int *nmatches);

/*
 * ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)
 * (src/ORBmatcher.cc:1472-1599), the matcher Tracking::Relocalization runs after PnP (src/Tracking.cc:1459, 1473: th = 10,
 * ORBdist = 100, then th = 3, ORBdist = 64).  Three entry points, because the level prediction in the middle belongs to the
 * caller's MapPoint (MapPoint::PredictScale reads the protected mfMaxDistance, src/MapPoint.cc:402-417):
 *   orbm_project_points   (host) :1498-1514 for n world points: x3Dc = Rcw x + tcw, u, v, 1 / zc, the image-bounds test
 *                         (in_image[i], :1507-1510) and dist3D = |x - Ow| -- cv::Mat algebra as cv::gemm's small-matrix path does it
 *                         (float accumulation, alpha / beta in double), cv::norm accumulating in double.  invzc and dist3d may be NULL.
 *   orbm_predict_scale    (host) MapPoint::PredictScale(dist, Frame*) for callers that own mfMaxDistance themselves.
 *   orbm_search_by_projection_kf   the search: per key-frame MapPoint i, use[i] = usable (non-NULL, !isBad(), not in sAlreadyFound,
 *                         in_image, minDistance <= dist3D <= maxDistance :1492-1521), window (proj_u, proj_v, th *
 *                         mvScaleFactors[pred_level], levels pred_level -+ 1 :1526-1528), best distance over the free slots
 *                         (cur_has_point[i2] == 0, :1541-1542), accept at <= orb_dist, rotation histogram with
 *                         kf_angle[i] = pKF->mvKeysUn[i].angle and the ComputeThreeMaxima cull (:1577-1596).  The current frame's
 *                         grid must be in the handle.  cur_has_point in/out; cur_match[i2] = i or -1; *nmatches = return value.
 * Windows and candidate distances on the GPU in two passes, the scan on the host (it is sequential in the reference: an
 * assignment blocks the slot for every later MapPoint).
 */
int orbm_project_points(const float *Tcw, float fx, float fy, float cx, float cy, const float bounds[4],
                        const float *xw, int n, float *u, float *v, float *invzc, float *dist3d, uint8_t *in_image);
int orbm_predict_scale(float mf_max_distance, float current_dist, float log_scale_factor, int n_levels);
int orbm_search_by_projection_kf(orbm_matcher *m, int n_mp, const uint8_t *use, const float *proj_u, const float *proj_v,
                                 const int32_t *pred_level, const uint8_t *mp_desc, const float *kf_angle,
                                 const float *scale_factors, int nlevels, const orbx_keypoint *kps_cur, const uint8_t *desc_cur,
                                 int n_cur, float th, int orb_dist, int check_orientation,
                                 uint8_t *cur_has_point, int32_t *cur_match, int *nmatches);

/*
 * Frame::UndistortKeyPoints (src/Frame.cc:404-434) and Frame::ComputeImageBounds (:436-463): host code, they run once per
 * frame on ~10^3 points between orbx_extract and orbm_grid_build.  dist = mDistCoef (k1, k2, p1, p2[, k3]); ndist = 4 or 5.
 * With dist[0] == 0 both are the identity exactly as in the reference (:406-410, :455-461).  Otherwise the points go
 * through cv::undistortPoints(src, dst, K, dist, noArray(), K) of OpenCV 3.1.0, restated: normalise with K, five fixed-point
 * iterations x <- (x0 - deltaX(x)) * icdist(x) of the Brown model in double, reproject with K, store as float.
 * kps_un may alias kps.  bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY}.
 */
int orbm_undistort_keypoints(const orbx_keypoint *kps, int n, float fx, float fy, float cx, float cy,
                             const float *dist, int ndist, orbx_keypoint *kps_un);
int orbm_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float *dist, int ndist,
                      float bounds[4]);

int orbm_grid_build(orbm_matcher *m, const orbx_keypoint *kps_un, int n,
                    float min_x, float max_x, float min_y, float max_y);
int orbm_features_in_area(orbm_matcher *m, const float *x, const float *y, const float *r,
                          const int32_t *min_level, const int32_t *max_level, int nq,
                          int32_t *cand_off, int32_t *cand_idx, int cap_idx);
int orbm_search_area_best2(orbm_matcher *m, const uint8_t *qdesc, const float *x, const float *y, const float *r,
                           const int32_t *min_level, const int32_t *max_level, int nq,
                           const uint8_t *train_desc, const uint8_t *skip,
                           int32_t *best_idx, int32_t *best_d, int32_t *second_d);
int orbm_search_area_best2_device(orbm_matcher *m, const uint8_t *d_qdesc, const float *d_x, const float *d_y,
                                  const float *d_r, const int32_t *d_min_level, const int32_t *d_max_level, int nq,
                                  const uint8_t *d_train_desc, const uint8_t *d_skip,
                                  int32_t *d_best_idx, int32_t *d_best_d, int32_t *d_second_d, void *hip_stream);

/*
 * ---- "next" row N2: ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) src/ORBmatcher.cc:159-288 ----
 * The two FeatureVectors come from orbv_feature_vector (ascending node ids, CSR feature lists).  For every node both
 * frames share, every key-frame feature with a usable MapPoint (valid_kf[i] != 0: the reference's `pMP && !pMP->isBad()`)
 * scans the frame's features of that node in list order, skipping the ones an earlier key-frame feature already took
 * (`if(vpMapPointMatches[realIdxF]) continue;` :209) -- so the scan order matters and is kept.  A frame feature lives in
 * exactly one node, hence nodes are independent: one wave per node walks its key-frame features in order with the node's
 * frame features across the lanes (best / second-best :214-226, TH_LOW and ratio :228-232); the rotation histogram
 * (:236-246, :266-284) runs on the host over the match table.  Nodes with more than 4096 frame features take the general
 * path: all node-mate distances in one GPU call, the same selection scan on the host.  match_f[i] = index of the key-frame feature matched to frame feature i, or -1
 * (the reference's vpMapPointMatches as indices); *nmatches = the function's return value.
 * kps_kf are the key frame's mvKeysUn, kps_f the frame's mvKeys (only .angle is read).
 */
int orbm_search_by_bow(orbm_matcher *m,
                       const uint8_t *desc_kf, const orbx_keypoint *kps_kf, int n_kf, const uint8_t *valid_kf,
                       const int32_t *fv_kf_node, const int32_t *fv_kf_off, const int32_t *fv_kf_idx, int fv_kf_n,
                       const uint8_t *desc_f, const orbx_keypoint *kps_f, int n_f,
                       const int32_t *fv_f_node, const int32_t *fv_f_off, const int32_t *fv_f_idx, int fv_f_n,
                       float nnratio, int check_orientation, int32_t *match_f, int *nmatches);

/*
 * ---- the LocalMapping / LoopClosing matchers (include/ORBmatcher.h:60, 66, 72, 77, 80, 83) ----
 * Same division of labour as the Tracking-thread matchers above: the cv::Mat algebra of a call runs on the host with OpenCV
 * 3.1.0's arithmetic (3x3 * 3x1 products through cv::gemm's small-matrix path: float accumulation, alpha / beta applied in
 * double; Mat / scalar as a float multiplication by (float)(1./s); Mat::dot and cv::norm accumulating in double), MapPoint::
 * PredictScale(dist, pKF) stays with the caller's MapPoint, and the windows (KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:569-606),
 * the per-candidate predicates and the Hamming distances run on the GPU.  Where the reference's inner loop carries no state from
 * one MapPoint / feature to the next (Fuse x 2, SearchBySim3, SearchForTriangulation) the selection runs on the GPU as well.
 *
 * orbm_reserve       grows the handle's workspace (never shrinks it).  The reference's matcher has no size limit, so every entry
 *                    point grows the handle when its inputs are larger than the handle instead of refusing; reserving up front
 *                    keeps allocation out of the calls.  Growing max_train drops the grid in the handle.
 * orbm_grid_build_kf a KeyFrame's grid: the cells were filled by Frame::PosInGrid with Frame's float mnMinX / mnMinY and grid
 *                    element sizes (assign_*, inv_*: src/KeyFrame.cc:48-54 copies mGrid), while KeyFrame::GetFeaturesInArea
 *                    subtracts the key frame's own int mnMinX / mnMinY (query_*: include/KeyFrame.h:190-193).
 */
typedef struct {
    float assign_min_x, assign_min_y;   /* Frame::mnMinX, Frame::mnMinY */
    float inv_w, inv_h;                 /* pKF->mfGridElementWidthInv, mfGridElementHeightInv */
    float query_min_x, query_min_y;     /* (float)pKF->mnMinX, (float)pKF->mnMinY */
} orbm_kf_grid;
int orbm_reserve(orbm_matcher *m, int max_queries, int max_train, int max_pairs);
int orbm_grid_count(const orbm_matcher *m);     /* keypoints in the handle's grid; -1: none (never built, or dropped by a growth) */
int orbm_grid_build_kf(orbm_matcher *m, const orbx_keypoint *kps_un, int n, float assign_min_x, float assign_min_y,
                       float inv_w, float inv_h, float query_min_x, float query_min_y);

/* (host) Scw -> [Rcw|tcw] as a row-major 4x4 and Ow: src/ORBmatcher.cc:299-303 == :986-990. */
int orbm_sim3_decompose(const float *Scw, float *Tcw, float *Ow);
/* (host) SearchBySim3's src/ORBmatcher.cc:1119-1121: sR12 = s12*R12, sR21 = (1.0/s12)*R12.t(), t21 = -sR21*t12 (row-major 3x3, 3). */
int orbm_sim3_relative(float s12, const float *R12, const float *t12, float *sR12, float *sR21, float *t21);
/*
 * (host) the projection block of :320-355, :852-885, :1008-1043 for n world points: p3Dc = Rcw*p3Dw+tcw, u, v, 1/z, dist3D =
 * |p3Dw - Ow|.  ok[i] = depth not negative && KeyFrame::IsInImage(u, v) (bounds = the key frame's mnMinX, mnMaxX, mnMinY, mnMaxY;
 * [min, max), src/KeyFrame.cc:608-611) && (normal == NULL || !(PO.dot(Pn) < 0.5*dist3D)).  Ow == NULL: -Rcw^T tcw; Fuse passes
 * pKF->GetCameraCenter().  The distance-invariance test and PredictScale are the caller's (MapPoint getters).  invz may be NULL.
 */
int orbm_project_points_kf(const float *Tcw, const float *Ow, float fx, float fy, float cx, float cy, const float bounds[4],
                           const float *xw, const float *normal, int n, float *u, float *v, float *invz, float *dist3d, uint8_t *ok);
/* (host) SearchBySim3's :1158-1179 / :1238-1259: p = sR*(R_A x + t_A) + t for the MapPoints of key frame A (pose TAw), projected
 * with pKF1's calibration into key frame B; ok[i] = depth not negative && B->IsInImage(u, v); dist3d = |p|. */
int orbm_project_points_sim3(const float *TAw, const float *sR, const float *t, float fx, float fy, float cx, float cy,
                             const float boundsB[4], const float *xw, int n, float *u, float *v, float *dist3d, uint8_t *ok);

/*
 * ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:290-403; LoopClosing.cc:376).
 * Per candidate MapPoint i: use[i] = every `continue` of :317-355 passed (not bad, not in spAlreadyFound, orbm_project_points_kf's
 * ok, minDistance <= dist <= maxDistance), proj_u / proj_v, pred_level = pMP->PredictScale(dist, pKF), mp_desc = GetDescriptor().
 * The key frame's grid must be in the handle (orbm_grid_build_kf).  kf_matched[idx] in/out = (vpMatched[idx] != NULL);
 * kf_match[idx] out = the MapPoint this call stored in vpMatched[idx], or -1; *nmatches = the return value.  Windows and candidate
 * distances on the GPU, the scan on the host: a match blocks its slot for every later MapPoint (:375, :396).
 */
int orbm_search_by_projection_sim3(orbm_matcher *m, int n_mp, const uint8_t *use, const float *proj_u, const float *proj_v,
                                   const int32_t *pred_level, const uint8_t *mp_desc, const float *scale_factors, int nlevels,
                                   const orbx_keypoint *kps_kf, const uint8_t *desc_kf, int n_kf, int th,
                                   uint8_t *kf_matched, int32_t *kf_match, int *nmatches);

/*
 * ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (src/ORBmatcher.cc:522-655; LoopClosing.cc:266).
 * valid1 / valid2 = `pMP && !pMP->isBad()` per feature; kps = mvKeysUn (angles); FeatureVectors as for orbm_search_by_bow.
 * matches12[idx1] = the feature of KF2 whose MapPoint goes into vpMatches12[idx1], or -1.  Strict `bestDist1 < TH_LOW` (:598).
 */
int orbm_search_by_bow_kf(orbm_matcher *m,
                          const uint8_t *desc1, const orbx_keypoint *kps1, int n1, const uint8_t *valid1,
                          const int32_t *fv1_node, const int32_t *fv1_off, const int32_t *fv1_idx, int fv1_n,
                          const uint8_t *desc2, const orbx_keypoint *kps2, int n2, const uint8_t *valid2,
                          const int32_t *fv2_node, const int32_t *fv2_off, const int32_t *fv2_idx, int fv2_n,
                          float nnratio, int check_orientation, int32_t *matches12, int *nmatches);

/*
 * ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (src/ORBmatcher.cc:657-823; LocalMapping.cc:270).
 * has_mp = (GetMapPoint(idx) != NULL), u_right = mvuRight, kps = mvKeysUn; Cw = pKF1->GetCameraCenter(), T2w = pKF2's [R2w|t2w]
 * (row-major 4x4), the calibration, scale factors and level sigma^2 of pKF2, F12 row-major 3x3.  matches12[idx1] = idx2 or -1:
 * vMatchedPairs is the list of (i, matches12[i]) with matches12[i] >= 0 in ascending i (:812-820).  Entirely on the GPU (one wave
 * per feature of KF1): this reference never sets vbMatched2, so no feature's search depends on another's; the candidate accepted
 * is the last one of minimal distance among those that pass the epipole and epipolar-line tests (:738-755).
 */
int orbm_search_for_triangulation(orbm_matcher *m,
                                  const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const uint8_t *has_mp1, const float *u_right1,
                                  const int32_t *fv1_node, const int32_t *fv1_off, const int32_t *fv1_idx, int fv1_n,
                                  const orbx_keypoint *kps2, const uint8_t *desc2, int n2, const uint8_t *has_mp2, const float *u_right2,
                                  const int32_t *fv2_node, const int32_t *fv2_off, const int32_t *fv2_idx, int fv2_n,
                                  const float *Cw, const float *T2w, float fx2, float fy2, float cx2, float cy2, const float *F12,
                                  const float *scale_factors2, const float *level_sigma2_2, int nlevels2, int only_stereo,
                                  int check_orientation, int32_t *matches12, int *nmatches);

/*
 * ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th) (src/ORBmatcher.cc:825-975; LocalMapping.cc:491, 516),
 * the search half (:889-952).  use[i] = every `continue` of :846-885 passed; proj_ur = u - bf*invz (:870); the key frame's grid in
 * the handle.  best_idx[i] = the key-frame feature the point is fused with (window, octave window, chi-square gate :914-938, best
 * distance <= TH_LOW), or -1; *nfused = their number.  What happens to a fused point (:954-970: Replace, or AddObservation +
 * AddMapPoint) is object-graph work and stays with the caller; no MapPoint's search depends on it, so the searches run on the GPU
 * as one launch.  orbm_fuse_sim3 is Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint) (:977-1100; LoopClosing.cc:600),
 * :1048-1082: the same without the gate.
 */
int orbm_fuse(orbm_matcher *m, int n_mp, const uint8_t *use, const float *proj_u, const float *proj_v, const float *proj_ur,
              const int32_t *pred_level, const uint8_t *mp_desc, const float *scale_factors, const float *inv_level_sigma2,
              int nlevels, const orbx_keypoint *kps_kf, const float *u_right_kf, const uint8_t *desc_kf, int n_kf, float th,
              int32_t *best_idx, int *nfused);
int orbm_fuse_sim3(orbm_matcher *m, int n_mp, const uint8_t *use, const float *proj_u, const float *proj_v,
                   const int32_t *pred_level, const uint8_t *mp_desc, const float *scale_factors, int nlevels,
                   const orbx_keypoint *kps_kf, const uint8_t *desc_kf, int n_kf, float th, int32_t *best_idx, int *nfused);

/*
 * ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (src/ORBmatcher.cc:1102-1326; LoopClosing.cc:324), the two
 * searches and the agreement check (:1188-1323).  One MapPoint slot per key-frame feature (n_mp1 == n1, n_mp2 == n2).
 * use1[i] = pMP && !vbAlreadyMatched1[i] && !isBad() && orbm_project_points_sim3's ok && the distance-invariance test (:1152-1183);
 * proj_u1 / proj_v1 = its projection into key frame 2, pred_level1 = PredictScale(dist3D, pKF2); use2 / proj_*2 / pred_level2 the
 * reverse.  Both grids are built by the call (slot 1 of the handle holds key frame 1's afterwards).  match12[i1] = the feature of
 * KF2 whose MapPoint goes into vpMatches12[i1], or -1; *nfound = the return value.
 */
int orbm_search_by_sim3(orbm_matcher *m,
                        int n_mp1, const uint8_t *use1, const float *proj_u1, const float *proj_v1, const int32_t *pred_level1,
                        const uint8_t *mp_desc1,
                        int n_mp2, const uint8_t *use2, const float *proj_u2, const float *proj_v2, const int32_t *pred_level2,
                        const uint8_t *mp_desc2,
                        const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const orbm_kf_grid *grid1, const float *scale_factors1,
                        int nlevels1,
                        const orbx_keypoint *kps2, const uint8_t *desc2, int n2, const orbm_kf_grid *grid2, const float *scale_factors2,
                        int nlevels2, float th, int32_t *match12, int *nfound);

/* Host helpers: ComputeThreeMaxima (ind[3], -1 = none) and the histogram cull over match12. */
int orbm_three_maxima(const int32_t *hist_sizes, int L, int32_t ind[3]);
int orbm_rot_filter(const float *angle_q, const float *angle_t, int32_t *match12, int nq);

const char *orbm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
