/*
 * orbp.h -- C ABI of the host-side pose solvers of liborbx.so: "next" row N4 of SURVEY.md 8(f).
 *
 * SURVEY 8(f) N4 keeps these on the host ("tiny dense fp64 solves"): they are plain C++ inside liborbx.so, take the
 * outputs of the device path (orbx_extract -> orbv_* -> orbm_search_by_bow) and close BASELINE config 5's loop.
 *
 * Reference (WChen09/My-SLAM):
 *   src/PnPsolver.cc:66-109    PnPsolver::PnPsolver          -> orbp_pnp_create   (caller drops NULL / bad MapPoints)
 *   src/PnPsolver.cc:120-158   SetRansacParameters           -> orbp_pnp_set_ransac_parameters
 *   src/PnPsolver.cc:160-164   find                          -> orbp_pnp_find
 *   src/PnPsolver.cc:166-258   iterate (RANSAC over 4-point EPnP, Refine :260-309, CheckInliers :312-344)
 *                                                            -> orbp_pnp_iterate
 *   src/PnPsolver.cc:346-1022  EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009): control points by PCA, barycentric
 *                              coordinates, null space of M^T M, the three beta approximations, 5 Gauss-Newton
 *                              steps, absolute orientation, least mean reprojection error wins
 *                                                            -> orbp_epnp
 *   src/Optimizer.cc:239-451   Optimizer::PoseOptimization   -> orbp_pose_optimization
 *       with g2o's EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose
 *       (Thirdparty/g2o/g2o/types/types_six_dof_expmap.{h:143-215,cpp:266-364}), SE3Quat::exp
 *       (types/se3quat.h:223-257), RobustKernelHuber (core/robust_kernel_impl.cpp:78-91) and
 *       OptimizationAlgorithmLevenberg::solve (core/optimization_algorithm_levenberg.cpp:66-170, including this
 *       fork's _nBad early stop :157-164) over a dense 6x6 system.
 *   Call sites: src/Tracking.cc:766-809 (TrackReferenceKeyFrame), :1348-1509 (Relocalization).
 *
 * Dependencies the reference takes from OpenCV 3.1.0 / Eigen3 (neither is in the image) are replaced by a one-sided
 * Jacobi SVD and a 6x6 LDL^T written here.  Where the reference's result depends on those libraries' internals the
 * behaviour is implementation-defined and documented in DESIGN.md ("parity unpinned"): the basis OpenCV returns for
 * the (numerically) null singular vectors of the rank-8 M^T M of a 4-point sample (any orthonormal basis of that null
 * space is a valid SVD; which one comes out is an artefact of the Jacobi sweep order, and EPnP's beta linearisations
 * depend on it), the sign of the PCA axes, and the completion of U for a rank-deficient 3x3.
 * RANSAC draws: DUtils::Random::RandomInt (Thirdparty/DBoW2/DUtils/Random.cpp:47-50) on libc rand(); the default
 * source here is the same rand() with the same formula, so a process that seeds like the reference draws like it.
 */
#ifndef ORBP_H
#define ORBP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orbp_pnp orbp_pnp;

/* n correspondences: p2d = mvKeysUn[i].pt (2n floats), sigma2 = mvLevelSigma2[octave] (n), p3d = world position (3n). */
int orbp_pnp_create(orbp_pnp **out, int n, const float *p2d, const float *sigma2, const float *p3d,
                    float fx, float fy, float cx, float cy);
void orbp_pnp_destroy(orbp_pnp *s);
/* Uniform source for the RANSAC draws: fn(ctx) in [0, rand_max].  NULL restores libc rand() / RAND_MAX. */
void orbp_pnp_set_rand(orbp_pnp *s, int (*fn)(void *), void *ctx, int rand_max);
int orbp_pnp_set_ransac_parameters(orbp_pnp *s, double probability, int min_inliers, int max_iterations, int min_set,
                                   float epsilon, float th2);
/* Effective values after SetRansacParameters' adjustment (any pointer may be NULL). */
void orbp_pnp_get_ransac_state(const orbp_pnp *s, int *min_inliers, int *max_its, float *epsilon, int *iterations_done);
/* Returns 1 and fills Tcw (row-major 4x4 float), inliers[n] (0/1) and *n_inliers when a pose is returned, 0 when the
 * reference returns an empty cv::Mat (inliers untouched, *n_inliers = 0), < 0 on bad arguments. */
int orbp_pnp_iterate(orbp_pnp *s, int n_iterations, int *no_more, uint8_t *inliers, int *n_inliers, float *Tcw);
int orbp_pnp_find(orbp_pnp *s, uint8_t *inliers, int *n_inliers, float *Tcw);

/* PnPsolver::compute_pose on n >= 4 correspondences (pws 3n, us 2n doubles): R row-major, t; returns the mean
 * reprojection error of the chosen solution. */
double orbp_epnp(int n, const double *pws, const double *us, double fu, double fv, double uc, double vc,
                 double *R, double *t);

/* Optimizer::PoseOptimization.  Observation i: obs (2n floats, mvKeysUn[i].pt), u_right[i] < 0 (or u_right == NULL)
 * for a monocular edge, inv_sigma2 = mvInvLevelSigma2[octave], xw = world position (3n floats).  Tcw: in = pFrame->mTcw,
 * out = optimised pose (row-major 4x4 float).  outlier[n] receives mvbOutlier.  Returns nInitialCorrespondences - nBad
 * (0 when fewer than 3 observations: pose untouched), < 0 on bad arguments. */
int orbp_pose_optimization(int n, const float *obs, const float *u_right, const float *inv_sigma2, const float *xw,
                           float fx, float fy, float cx, float cy, float bf, float *Tcw, uint8_t *outlier);

const char *orbp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
