/*
 * orb_oracle.h -- CPU restatement ("oracle") of the reference ORB front-end.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under my-slam_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
 * and as the reported CPU baseline ("kind": "port").
 *
 * PARITY UNPINNED: the reference (WChen09/My-SLAM) ships no tests, fixtures or golden vectors for
 * this path and cannot be built here (needs OpenCV 3.1.0, absent; writing stand-in
 * headers for it is not allowed).  Four of the stages' arithmetic lives in OpenCV 3.1.0 (resize,
 * FAST, GaussianBlur, fastAtan2/cvRound), restated here from its published algorithm; every such
 * decision point is marked "OpenCV 3.1.0:" in orb_oracle.c.  The reference's own logic is restated
 * line by line with file:line citations (paths relative to /root/reference).
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORO_MAX_LEVELS 16
#define ORO_EDGE_THRESHOLD 19   /* src/ORBextractor.cc:76 */
#define ORO_HALF_PATCH 15       /* src/ORBextractor.cc:75 */
#define ORO_PATCH_SIZE 31       /* src/ORBextractor.cc:74 */

/* cv::KeyPoint field layout (28 bytes): Point2f pt; float size, angle, response; int octave, class_id */
typedef struct {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} oro_keypoint;

/* FAST candidate before distribution: integer coordinates relative to (minBorderX,minBorderY) */
typedef struct {
    int32_t x, y;
    int32_t response;
} oro_cand;

enum { ORO_BLUR_SCALAR = 0,   /* OpenCV portable C column pass: (sum + 32768) >> 16           */
       ORO_BLUR_X86_SIMD = 1  /* OpenCV SSE2 column pass for x < (w & ~3): round-half-to-even  */ };

typedef struct {
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast, min_th_fast;
    float scale[ORO_MAX_LEVELS];        /* mvScaleFactor      */
    float inv_scale[ORO_MAX_LEVELS];    /* mvInvScaleFactor   */
    float sigma2[ORO_MAX_LEVELS];       /* mvLevelSigma2      */
    float inv_sigma2[ORO_MAX_LEVELS];   /* mvInvLevelSigma2   */
    int quota[ORO_MAX_LEVELS];          /* mnFeaturesPerLevel */
    int umax[ORO_HALF_PATCH + 1];
    int gauss_k[7];                     /* 7x7 sigma=2 kernel x256, per pass */
    int blur_mode;
} oro_extractor;

/* ---- A1: constructor tables (src/ORBextractor.cc:412-472) ---- */
int oro_extractor_init(oro_extractor *e, int nfeatures, float scale_factor, int nlevels,
                       int ini_th, int min_th);
void oro_level_size(const oro_extractor *e, int W, int H, int level, int *w, int *h);
const signed char *oro_pattern(void);           /* 1024 int8 */

/* ---- OpenCV primitive restatements ---- */
int oro_cv_round(double v);                     /* cvRound: round half to even */
float oro_fast_atan2(float y, float x);         /* cv::fastAtan2, degrees */
void oro_sincos_deg(float angle_deg, float *a_cos, float *b_sin); /* :114-115, correctly rounded */
void oro_sincos_rad_array(const float *theta, float *c, float *s, long long n);
int oro_reflect101(int p, int len);
void oro_resize_linear(const uint8_t *src, int sw, int sh, int sstride,
                       uint8_t *dst, int dw, int dh, int dstride);
void oro_copy_make_border101(const uint8_t *src, int w, int h, int sstride,
                             uint8_t *dst, int dstride, int border);
void oro_gaussian_blur7(const uint8_t *src, int w, int h, int sstride,
                        uint8_t *dst, int dstride, const int k[7], int mode);
/* cv::FAST(img, kps, threshold, nonmax, TYPE_9_16) on a cols x rows view; returns count (<= cap). */
int oro_fast9_16(const uint8_t *img, int stride, int cols, int rows, int threshold, int nonmax,
                 oro_cand *out, int cap);
/* threshold-free corner score of one pixel: max over 9-arcs of min |diff|, minus 1 (may be < 0) */
int oro_fast_score_pixel(const uint8_t *p, int stride);

/* ---- A2..A8 ---- */
/* levels[l] must hold w_l*h_l bytes, contiguous (stride = w_l), interior only */
void oro_compute_pyramid(const oro_extractor *e, const uint8_t *img, int W, int H, int stride,
                         uint8_t *const *levels);
/* A3 cell loop for one level; returns the number of candidates (<= cap), -1 on overflow */
int oro_detect_level(const oro_extractor *e, const uint8_t *lvl, int w, int h, int stride,
                     oro_cand *out, int cap);
/* A4; returns the number kept, written as indices into cands (order = list order) */
int oro_distribute_octree(const oro_cand *cands, int n, int minX, int maxX, int minY, int maxY,
                          int N, int32_t *out_idx, int cap);
float oro_ic_angle(const uint8_t *lvl, int stride, int x, int y, const int umax[16]);
void oro_descriptor(const uint8_t *blurred, int stride, int x, int y, float angle_deg,
                    uint8_t desc[32]);
/*
 * A8 operator().  kps/desc capacity cap; *n receives the count.  level_out (optional, may be NULL)
 * receives nlevels pointers that the caller has sized via oro_level_size.  n_per_level optional.
 * Returns 0, or <0: -1 bad args, -2 capacity, -3 unsupported shape (reference UB).
 */
int oro_extract(const oro_extractor *e, const uint8_t *img, int W, int H, int stride,
                oro_keypoint *kps, uint8_t *desc, int cap, int *n,
                uint8_t *const *level_out, int *n_per_level);

/* ---- A9..A11 matcher ---- */
int oro_descriptor_distance(const uint8_t a[32], const uint8_t b[32]);  /* src/ORBmatcher.cc:1647 */
/*
 * best / second-best over candidate lists (SearchByBoW inner loop, src/ORBmatcher.cc:201-226).
 * cand_off NULL => dense (every query against train 0..nt-1 in order).
 */
void oro_best2(const uint8_t *q, int nq, const uint8_t *t, int nt,
               const int32_t *cand_off, const int32_t *cand_idx,
               int32_t *best_idx, int32_t *best_d, int32_t *second_d);
void oro_three_maxima(const int *hist_sizes, int L, int *ind1, int *ind2, int *ind3); /* :1601 */
int oro_rot_bin(float angle1, float angle2);  /* :236-244 */
/*
 * Rotation-consistency filter (:236-246 fill + :266-284 cull): match12[i] = train index or -1;
 * culled entries are set to -1.  Returns surviving match count.
 */
int oro_rot_filter(const float *angle_q, const float *angle_t, int32_t *match12, int nq);
/*
 * Dense frame-vs-frame matching with the SearchByBoW acceptance (<= th, ratio) on stateless best2,
 * then the rotation filter.  Used for BASELINE config 3.  Returns match count.
 */
int oro_match_dense(const uint8_t *q, const float *angle_q, int nq,
                    const uint8_t *t, const float *angle_t, int nt,
                    int th, float nnratio, int check_ori, int32_t *match12);

/* ---- N1: Frame grid (src/Frame.cc:230-245, 327-392) ---- */
#define ORO_GRID_COLS 64   /* include/Frame.h:38 */
#define ORO_GRID_ROWS 48   /* include/Frame.h:37 */
typedef struct {
    float min_x, max_x, min_y, max_y;       /* mnMinX, mnMaxX, mnMinY, mnMaxY (Frame::ComputeImageBounds) */
    float inv_w, inv_h;                     /* mfGridElementWidthInv / HeightInv (src/Frame.cc:212-213) */
    int n;                                  /* keypoints */
    int cell_start[ORO_GRID_COLS * ORO_GRID_ROWS + 1];   /* CSR over cells, cell = ix*48 + iy */
    int *items;                             /* keypoint indices, push_back order inside a cell */
} oro_grid;
/* Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:404-463) on cv::undistortPoints(src, dst, K, D, noArray(), K)
 * of OpenCV 3.1.0 (5 fixed-point iterations in double).  xy: n (x, y) float pairs in place; dist = k1 k2 p1 p2 k3. */
void oro_undistort_points(float *xy, int n, float fx, float fy, float cx, float cy, const float dist[5]);
void oro_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float dist[5], float bounds[4]);
/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520) on a grid of frame 2.  prev_matched: n1 (x, y) pairs,
 * updated in place (:513-516); matches12[n1].  Returns nmatches. */
int oro_search_for_initialization(const oro_keypoint *kps1, const uint8_t *desc1, int n1,
                                  const oro_grid *g2, const oro_keypoint *kps2, const uint8_t *desc2, int n2,
                                  float *prev_matched, int window_size, float nnratio, int check_orientation, int32_t *matches12);
/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (src/ORBmatcher.cc:1328-1470), the
 * matcher of TrackWithMotionModel.  Last frame, per feature i: has_point = pMP && !mvbOutlier[i], xw = GetWorldPos(),
 * mp_desc = GetDescriptor(), mp_obs = Observations(), kps_last = mvKeys/mvKeysUn (octave, angle).  Current frame: poses as
 * row-major 4x4 float, grid g of kps_cur (mvKeysUn), u_right = mvuRight or NULL, cur_obs[i2] = -1 for a NULL mvpMapPoints
 * entry else that point's Observations() (in/out), cur_match[i2] = last-frame feature assigned by this call or -1 (out).
 * cv::Mat algebra (Rcw * x + tcw) as cv::gemm's small-matrix path does it (oro_gemm_row in orb_oracle.c). */
int oro_search_by_projection_last(int n_last, const uint8_t *has_point, const float *xw, const uint8_t *mp_desc, const int32_t *mp_obs,
                                  const oro_keypoint *kps_last, const float *Tcw, const float *Tlw,
                                  float fx, float fy, float cx, float cy, float mb, float mbf, const float bounds[4],
                                  const float *scale_factors, const oro_grid *g, const oro_keypoint *kps_cur, const uint8_t *desc_cur,
                                  const float *u_right, int n_cur, float th, int mono, int check_orientation,
                                  int32_t *cur_obs, int32_t *cur_match);
/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (src/ORBmatcher.cc:45-125), the matcher of
 * Tracking::SearchLocalPoints.  Per MapPoint: in_view = mbTrackInView && !isBad(), proj_x/y/xr = mTrackProjX/Y/XR,
 * pred_level = mnTrackScaleLevel, view_cos = mTrackViewCos, mp_desc, mp_obs = Observations().  Frame as in
 * oro_search_by_projection_last (grid of mvKeysUn, u_right or NULL, cur_obs in/out, cur_match out = MapPoint index). */
int oro_search_by_projection_map(int n_mp, const uint8_t *in_view, const float *proj_x, const float *proj_y, const float *proj_xr,
                                 const int32_t *pred_level, const float *view_cos, const uint8_t *mp_desc, const int32_t *mp_obs,
                                 const float *scale_factors, const oro_grid *g, const oro_keypoint *kps_cur, const uint8_t *desc_cur,
                                 const float *u_right, int n_cur, float th, float nnratio, int32_t *cur_obs, int32_t *cur_match);
/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)
 * (src/ORBmatcher.cc:1472-1599), Tracking::Relocalization's matcher after PnP (src/Tracking.cc:1459,1473), and
 * MapPoint::PredictScale(dist, Frame*) (src/MapPoint.cc:402-417).  Arguments: see orb_oracle.c. */
int oro_predict_scale(float mf_max_distance, float current_dist, float log_scale_factor, int n_levels);
int oro_search_by_projection_kf(int n_kf, const uint8_t *usable, const float *xw, const float *min_dist_inv, const float *max_dist_inv,
                                const float *mf_max_distance, const uint8_t *mp_desc, const float *kf_angle, const float *Tcw,
                                float fx, float fy, float cx, float cy, const float bounds[4], const float *scale_factors, int nlevels,
                                float log_scale_factor, const oro_grid *g, const oro_keypoint *kps_cur, const uint8_t *desc_cur, int n_cur,
                                float th, int orb_dist, int check_orientation, uint8_t *cur_has_point, int32_t *cur_match);
/* cv::Mat 3x3 * 3x1 algebra of OpenCV 3.1.0 (cv::gemm's small-matrix path; see orb_oracle.c): (R x + t)[row] of a row-major 4x4 [R|t],
 * and the camera centre -R^T t. */
float oro_gemm_row(const float *T, int row, const float *x);
void oro_camera_center(const float *T, float Ow[3]);

/* ---- the LocalMapping / LoopClosing matchers (orb_oracle_kf.c; arguments documented there) ---- */
void oro_grid_build_kf(oro_grid *g, const oro_keypoint *kps_un, int n, float assign_min_x, float assign_min_y,
                       float inv_w, float inv_h, float query_min_x, float query_min_y, int *items);
void oro_sim3_decompose(const float *Scw, float T[16], float Ow[3]);                        /* src/ORBmatcher.cc:299-303 */
int oro_search_by_projection_sim3(int n_mp, const uint8_t *usable, const float *xw, const float *normal, const float *min_dist_inv,
                                  const float *max_dist_inv, const float *mf_max_distance, const uint8_t *mp_desc, const float *Scw,
                                  float fx, float fy, float cx, float cy, const float bounds[4], const float *scale_factors, int nlevels,
                                  float log_scale_factor, const oro_grid *g, const oro_keypoint *kps_kf, const uint8_t *desc_kf, int n_kf,
                                  int th, uint8_t *kf_matched, int32_t *kf_match);          /* :290-403 */
int oro_search_by_bow_kf(const uint8_t *desc1, const float *angle1, int n1, const uint8_t *valid1,
                         const int32_t *node1, const int32_t *off1, const int32_t *idx1v, int nn1,
                         const uint8_t *desc2, const float *angle2, int n2, const uint8_t *valid2,
                         const int32_t *node2, const int32_t *off2, const int32_t *idx2v, int nn2,
                         float nnratio, int check_orientation, int32_t *matches12);         /* :522-655 */
int oro_search_for_triangulation(const oro_keypoint *kps1, const uint8_t *desc1, int n1, const uint8_t *has_mp1, const float *u_right1,
                                 const int32_t *node1, const int32_t *off1, const int32_t *idx1v, int nn1,
                                 const oro_keypoint *kps2, const uint8_t *desc2, int n2, const uint8_t *has_mp2, const float *u_right2,
                                 const int32_t *node2, const int32_t *off2, const int32_t *idx2v, int nn2,
                                 const float *Cw, const float *T2w, float fx2, float fy2, float cx2, float cy2, const float *F12,
                                 const float *scale_factors2, const float *level_sigma2_2, int only_stereo, int check_orientation,
                                 int32_t *matches12);                                       /* :657-823 */
int oro_fuse(int n_mp, const uint8_t *usable, const float *xw, const float *normal, const float *min_dist_inv, const float *max_dist_inv,
             const float *mf_max_distance, const uint8_t *mp_desc, const float *Tcw, const float *Ow, float fx, float fy, float cx, float cy,
             float bf, const float bounds[4], const float *scale_factors, const float *inv_level_sigma2, int nlevels, float log_scale_factor,
             const oro_grid *g, const oro_keypoint *kps_kf, const float *u_right_kf, const uint8_t *desc_kf, int n_kf, float th,
             int32_t *best_idx);                                                            /* :825-975 */
int oro_fuse_sim3(int n_mp, const uint8_t *usable, const float *xw, const float *normal, const float *min_dist_inv, const float *max_dist_inv,
                  const float *mf_max_distance, const uint8_t *mp_desc, const float *Scw, float fx, float fy, float cx, float cy,
                  const float bounds[4], const float *scale_factors, int nlevels, float log_scale_factor, const oro_grid *g,
                  const oro_keypoint *kps_kf, const uint8_t *desc_kf, int n_kf, float th, int32_t *best_idx);   /* :977-1100 */
int oro_search_by_sim3(int n1, const uint8_t *usable1, const float *xw1, const float *min_dist1, const float *max_dist1,
                       const float *mf_max1, const uint8_t *mp_desc1, const float *T1w, const float bounds1[4], const float *scale_factors1,
                       int nlevels1, float log_scale_factor1, const oro_grid *g1, const oro_keypoint *kps1, const uint8_t *desc1,
                       int n2, const uint8_t *usable2, const float *xw2, const float *min_dist2, const float *max_dist2,
                       const float *mf_max2, const uint8_t *mp_desc2, const float *T2w, const float bounds2[4], const float *scale_factors2,
                       int nlevels2, float log_scale_factor2, const oro_grid *g2, const oro_keypoint *kps2, const uint8_t *desc2,
                       float fx, float fy, float cx, float cy, float s12, const float *R12, const float *t12, float th, int32_t *match12);   /* :1102-1326 */
/* AssignFeaturesToGrid + PosInGrid; items must hold n ints */
void oro_grid_build(oro_grid *g, const oro_keypoint *kps_un, int n, float min_x, float max_x, float min_y, float max_y, int *items);
/* GetFeaturesInArea: returns the count written to out (reference order), -1 if cap is too small */
int oro_features_in_area(const oro_grid *g, const oro_keypoint *kps_un, float x, float y, float r,
                         int min_level, int max_level, int32_t *out, int cap);
/* window query + best/second-best scan (SearchByProjection inner loop, src/ORBmatcher.cc:1397-1424, with the
 * second-best kept as in :432-457); skip[i] != 0 drops train keypoint i (the reference's `continue` predicates) */
void oro_search_area_best2(const oro_grid *g, const oro_keypoint *kps_un, const uint8_t *train_desc, const uint8_t *skip,
                           const uint8_t *qdesc, const float *x, const float *y, const float *r,
                           const int32_t *min_level, const int32_t *max_level, int nq,
                           int32_t *best_idx, int32_t *best_d, int32_t *second_d);

/* ---- N3: Frame::ComputeStereoMatches (src/Frame.cc:466-640) ----
 * pyrL/pyrR: nlevels interior level images (stride = width) of the left/right extractor (mvImagePyramid);
 * columns left of 0 are read through the reference's 19-px reflect-101 border.  u_right/depth get N floats. */
void oro_stereo_matches(const oro_extractor *e, const oro_keypoint *kl, const uint8_t *dl, int nl,
                        const oro_keypoint *kr, const uint8_t *dr, int nr,
                        uint8_t *const *pyrL, uint8_t *const *pyrR, const int *lw, const int *lh,
                        float mb, float mbf, float *u_right, float *depth);

/* ---- N2: DBoW2 vocabulary tree (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259) ---- */
typedef struct {
    int k, L, scoring, weighting, nnodes;
    const int32_t *child_off;   /* [nnodes+1] */
    const int32_t *child_ids;   /* children in push_back (file) order */
    const uint8_t *desc;        /* nnodes x 32 */
    const int32_t *word_of;     /* leaf -> word id */
    const double *weight;       /* node weight (idf for leaves) */
} oro_voc;
void oro_voc_transform_features(const oro_voc *v, const uint8_t *feat, int n, int levelsup,
                                int32_t *word_id, int32_t *node_id, double *weight);
int oro_voc_bow_vector(const oro_voc *v, const int32_t *word_id, const double *weight, int n, int32_t *ids, double *vals);
int oro_voc_feature_vector(const int32_t *node_id, const double *weight, int n, int32_t *node_ids, int32_t *off, int32_t *idx);
double oro_voc_score_l1(const int32_t *ids1, const double *vals1, int n1, const int32_t *ids2, const double *vals2, int n2);
/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) src/ORBmatcher.cc:159-288 on feature indices; returns nmatches */
int oro_search_by_bow(const uint8_t *desc_kf, const float *angle_kf, int n_kf, const uint8_t *valid_kf,
                      const int32_t *kf_node, const int32_t *kf_off, const int32_t *kf_idx, int kf_n,
                      const uint8_t *desc_f, const float *angle_f, int n_f,
                      const int32_t *f_node, const int32_t *f_off, const int32_t *f_idx, int f_n,
                      float nnratio, int check_orientation, int32_t *match_f);

#ifdef __cplusplus
}
#endif
#endif
