/*
 * orb_oracle_kf.c -- CPU restatement of the six ORBmatcher methods the LocalMapping / LoopClosing threads call
 * (see orb_oracle.h header note: TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at the OpenCV 3.1.0 boundary).
 *
 *   SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)   src/ORBmatcher.cc:290-403   (LoopClosing.cc:376)
 *   SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)                src/ORBmatcher.cc:522-655   (LoopClosing.cc:266)
 *   SearchForTriangulation(pKF1, pKF2, F12, pairs, bOnlyStereo)   src/ORBmatcher.cc:657-823   (LocalMapping.cc:270)
 *   Fuse(KeyFrame*, vpMapPoints, th)                              src/ORBmatcher.cc:825-975   (LocalMapping.cc:491,516)
 *   Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)            src/ORBmatcher.cc:977-1100  (LoopClosing.cc:600)
 *   SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)      src/ORBmatcher.cc:1102-1326 (LoopClosing.cc:324)
 *
 * MapPoint / KeyFrame objects are replaced by flat arrays: one entry per MapPoint with the values its getters return
 * (GetWorldPos, GetNormal, GetMin/MaxDistanceInvariance, mfMaxDistance for PredictScale, GetDescriptor) and a `usable`
 * flag for the object predicates at the head of each loop (NULL, isBad(), membership in a set).  What the reference does
 * to the object graph after a match (Replace, AddObservation, AddMapPoint) is the caller's business and is not restated;
 * the functions return the match tables those actions are taken from.
 *
 * OpenCV 3.1.0 decisions used here (cv::Mat algebra; each is a recollection of that version's sources, like every OpenCV
 * semantic in this oracle):
 *   - 3x3 * 3x1 products: cv::gemm's small-matrix path, float accumulation (oro_gemm_row / oro_camera_center in orb_oracle.c);
 *   - Mat / scalar and scalar * Mat: MatOp_AddEx with alpha = 1/s (resp. s) in double, materialised by convertTo, whose
 *     32f -> 32f kernel multiplies by (float)alpha in float (cvtScale_<float, float, float>);
 *   - Mat::dot of float vectors: products and sum in double (dotProd_32f);
 *   - cv::norm (L2) of a float vector: squares summed in double, one sqrt, rounded to float by the assignment;
 *   - KeyFrame::mnMinX .. mnMaxY are ints (include/KeyFrame.h:190-193), IsInImage is [min, max) (src/KeyFrame.cc:608-611).
 * Build: as orb_oracle.c (strict IEEE fp32, no contraction).  All file:line citations are relative to /root/reference.
 */
#include "orb_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* KeyFrame's grid: mGrid is copied from the Frame that became the key frame (src/KeyFrame.cc:48-54), so it was filled by
 * Frame::PosInGrid with Frame's float mnMinX / mnMinY (src/Frame.cc:382-392), while KeyFrame::GetFeaturesInArea
 * (src/KeyFrame.cc:569-606) subtracts KeyFrame's own int mnMinX / mnMinY; mfGridElementWidthInv / HeightInv are the Frame's.
 * With the shipped calibration (k1 == 0) the two origins are equal. */
void oro_grid_build_kf(oro_grid *g, const oro_keypoint *kps_un, int n, float assign_min_x, float assign_min_y,
                       float inv_w, float inv_h, float query_min_x, float query_min_y, int *items)
{
    g->n = n; g->items = items;
    g->inv_w = inv_w; g->inv_h = inv_h;
    const int NC = ORO_GRID_COLS * ORO_GRID_ROWS;
    int *cnt = (int *)calloc((size_t)NC + 1, sizeof(int));
    int *cell = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    for (int i = 0; i < n; i++) {
        const int px = (int)roundf((kps_un[i].x - assign_min_x) * inv_w);       /* src/Frame.cc:384 */
        const int py = (int)roundf((kps_un[i].y - assign_min_y) * inv_h);       /* :385 */
        cell[i] = (px < 0 || px >= ORO_GRID_COLS || py < 0 || py >= ORO_GRID_ROWS) ? -1 : px * ORO_GRID_ROWS + py;
        if (cell[i] >= 0) cnt[cell[i]]++;
    }
    g->cell_start[0] = 0;
    for (int c = 0; c < NC; c++) g->cell_start[c + 1] = g->cell_start[c] + cnt[c];
    memset(cnt, 0, sizeof(int) * (size_t)NC);
    for (int i = 0; i < n; i++)
        if (cell[i] >= 0) items[g->cell_start[cell[i]] + cnt[cell[i]]++] = i;
    free(cnt); free(cell);
    g->min_x = query_min_x; g->min_y = query_min_y;                              /* what oro_features_in_area subtracts */
    g->max_x = query_min_x + (float)ORO_GRID_COLS / inv_w; g->max_y = query_min_y + (float)ORO_GRID_ROWS / inv_h;   /* informative only */
}

/* Scw -> Rcw, tcw (as a row-major 4x4 [R|t]) and Ow, src/ORBmatcher.cc:299-303 == :986-990 */
void oro_sim3_decompose(const float *Scw, float T[16], float Ow[3])
{
    double dd = 0;                                                              /* sRcw.row(0).dot(sRcw.row(0)): Mat::dot, double */
    for (int k = 0; k < 3; k++) dd += (double)Scw[k] * (double)Scw[k];
    const float scw = (float)sqrt(dd);                                          /* :300 */
    const float inv = (float)(1.0 / (double)scw);                               /* Mat / s: alpha = 1./s, convertTo multiplies by (float)alpha */
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[4 * i + j] = Scw[4 * i + j] * inv + 0.0f; /* :301 */
        T[4 * i + 3] = Scw[4 * i + 3] * inv + 0.0f;                             /* :302 */
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    oro_camera_center(T, Ow);                                                   /* :303 */
}

/* the projection block shared by :320-357, :852-887, :1008-1046 (T = [Rcw|tcw]); returns 0 when one of the `continue`s fires.
 * normal == NULL skips the viewing-angle test (SearchBySim3 has none). */
typedef struct { float u, v, invz, dist; int level; } kf_proj;
static int project_into_kf(const float *T, const float *Ow, const float *X, const float *normal, float fx, float fy, float cx, float cy,
                           const float bounds[4], float min_dist, float max_dist, float mf_max_distance, float log_scale_factor, int nlevels,
                           kf_proj *out)
{
    const float xc = oro_gemm_row(T, 0, X), yc = oro_gemm_row(T, 1, X), zc = oro_gemm_row(T, 2, X);   /* Rcw*p3Dw+tcw */
    if (zc < 0.0f) return 0;                                                    /* Depth must be positive */
    const float invz = 1 / zc;
    const float x = xc * invz, y = yc * invz;
    const float u = fx * x + cx, v = fy * y + cy;
    if (!(u >= bounds[0] && u < bounds[1] && v >= bounds[2] && v < bounds[3])) return 0;     /* KeyFrame::IsInImage */
    float PO[3];
    double nn = 0;
    for (int k = 0; k < 3; k++) { PO[k] = X[k] - Ow[k]; nn += (double)PO[k] * (double)PO[k]; }
    const float dist = (float)sqrt(nn);                                         /* cv::norm(PO) */
    if (dist < min_dist || dist > max_dist) return 0;
    if (normal) {
        double dot = 0;                                                         /* PO.dot(Pn) */
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)normal[k];
        if (dot < 0.5 * dist) return 0;                                         /* Viewing angle must be less than 60 deg */
    }
    out->u = u; out->v = v; out->invz = invz; out->dist = dist;
    out->level = oro_predict_scale(mf_max_distance, dist, log_scale_factor, nlevels);       /* MapPoint::PredictScale(dist, pKF), src/MapPoint.cc:387-400 */
    return 1;
}

/* ---- ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:290-403 ----
 * usable[i] = !pMP->isBad() && !spAlreadyFound.count(pMP) (:317); kf_matched[idx] in/out = (vpMatched[idx] != NULL);
 * kf_match[idx] out = the MapPoint this call stored in vpMatched[idx], or -1.  Returns nmatches. */
int oro_search_by_projection_sim3(int n_mp, const uint8_t *usable, const float *xw, const float *normal, const float *min_dist_inv,
                                  const float *max_dist_inv, const float *mf_max_distance, const uint8_t *mp_desc, const float *Scw,
                                  float fx, float fy, float cx, float cy, const float bounds[4], const float *scale_factors, int nlevels,
                                  float log_scale_factor, const oro_grid *g, const oro_keypoint *kps_kf, const uint8_t *desc_kf, int n_kf,
                                  int th, uint8_t *kf_matched, int32_t *kf_match)
{
    float T[16], Ow[3];
    oro_sim3_decompose(Scw, T, Ow);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_kf > 0 ? n_kf : 1));
    for (int i = 0; i < n_kf; i++) kf_match[i] = -1;
    int nmatches = 0;
    for (int iMP = 0; iMP < n_mp; iMP++) {                                       /* :312 */
        if (!usable[iMP]) continue;
        kf_proj p;
        if (!project_into_kf(T, Ow, xw + 3 * (size_t)iMP, normal + 3 * (size_t)iMP, fx, fy, cx, cy, bounds, min_dist_inv[iMP], max_dist_inv[iMP],
                             mf_max_distance[iMP], log_scale_factor, nlevels, &p)) continue;
        const float radius = th * scale_factors[p.level];                       /* :360 */
        const int nc = oro_features_in_area(g, kps_kf, p.u, p.v, radius, -1, -1, cand, n_kf);     /* KeyFrame::GetFeaturesInArea: no levels */
        if (nc <= 0) continue;
        const uint8_t *dMP = mp_desc + (size_t)iMP * 32;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {                                           /* :372 */
            const int idx = cand[c];
            if (kf_matched[idx]) continue;
            const int kpLevel = kps_kf[idx].octave;
            if (kpLevel < p.level - 1 || kpLevel > p.level) continue;
            const int dist = oro_descriptor_distance(dMP, desc_kf + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= 50) {                                                   /* TH_LOW :394 */
            kf_matched[bestIdx] = 1; kf_match[bestIdx] = iMP;
            nmatches++;
        }
    }
    free(cand);
    return nmatches;
}

/* ---- ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12), src/ORBmatcher.cc:522-655 ----
 * valid1 / valid2 = `pMP && !pMP->isBad()` per feature; angles of mvKeysUn; matches12[idx1] = idx2 or -1.  Returns nmatches. */
int oro_search_by_bow_kf(const uint8_t *desc1, const float *angle1, int n1, const uint8_t *valid1,
                         const int32_t *node1, const int32_t *off1, const int32_t *idx1v, int nn1,
                         const uint8_t *desc2, const float *angle2, int n2, const uint8_t *valid2,
                         const int32_t *node2, const int32_t *off2, const int32_t *idx2v, int nn2,
                         float nnratio, int check_orientation, int32_t *matches12)
{
    int nmatches = 0, nrot = 0, hist[30];
    uint8_t *matched2 = (uint8_t *)calloc((size_t)(n2 > 0 ? n2 : 1), 1);        /* vbMatched2 :535 */
    int *rot = (int *)malloc(sizeof(int) * 2 * (size_t)(n1 > 0 ? n1 : 1));       /* rotHist as (bin, idx1) in push order */
    for (int i = 0; i < 30; i++) hist[i] = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                             /* :534 */
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {                                                 /* :550 */
        if (node1[a] == node2[b]) {
            for (int c1 = off1[a]; c1 < off1[a + 1]; c1++) {
                const int idx1 = idx1v[c1];
                if (!valid1[idx1]) continue;                                     /* :558-562 */
                const uint8_t *d1 = desc1 + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int c2 = off2[b]; c2 < off2[b + 1]; c2++) {
                    const int idx2 = idx2v[c2];
                    if (matched2[idx2] || !valid2[idx2]) continue;               /* :576-580 */
                    const int dist = oro_descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < 50) {                                            /* strict: `bestDist1<TH_LOW` :598 */
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {         /* :600 */
                        matches12[idx1] = bestIdx2;
                        matched2[bestIdx2] = 1;
                        if (check_orientation) {
                            const int bin = oro_rot_bin(angle1[idx1], angle2[bestIdx2]);
                            rot[2 * nrot] = bin; rot[2 * nrot + 1] = idx1; nrot++;
                            hist[bin]++;
                        }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (node1[a] < node2[b]) a++;                                     /* lower_bound on an ascending map == advance */
        else b++;
    }
    if (check_orientation) {                                                     /* :634-652 */
        int ind1, ind2, ind3;
        oro_three_maxima(hist, 30, &ind1, &ind2, &ind3);
        for (int k = 0; k < nrot; k++) {
            const int bin = rot[2 * k];
            if (bin == ind1 || bin == ind2 || bin == ind3) continue;
            matches12[rot[2 * k + 1]] = -1;
            nmatches--;
        }
    }
    free(matched2); free(rot);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine, src/ORBmatcher.cc:140-157.  F12 row-major 3x3. */
static int check_dist_epipolar_line(const oro_keypoint *kp1, const oro_keypoint *kp2, const float *F12, const float *level_sigma2_2)
{
    const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
    const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
    const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * level_sigma2_2[kp2->octave];
}

/* ---- ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo), src/ORBmatcher.cc:657-823 ----
 * has_mp1 / has_mp2 = (GetMapPoint(idx) != NULL); u_right = mvuRight; Cw = pKF1->GetCameraCenter(); T2w = pKF2's [R2w|t2w];
 * matches12[idx1] = idx2 or -1 (vMatchedPairs = the (i, matches12[i]) with matches12[i] >= 0, ascending i).  Returns nmatches.
 * vbMatched2 (:677) is never set in this reference, so it is not modelled. */
int oro_search_for_triangulation(const oro_keypoint *kps1, const uint8_t *desc1, int n1, const uint8_t *has_mp1, const float *u_right1,
                                 const int32_t *node1, const int32_t *off1, const int32_t *idx1v, int nn1,
                                 const oro_keypoint *kps2, const uint8_t *desc2, int n2, const uint8_t *has_mp2, const float *u_right2,
                                 const int32_t *node2, const int32_t *off2, const int32_t *idx2v, int nn2,
                                 const float *Cw, const float *T2w, float fx2, float fy2, float cx2, float cy2, const float *F12,
                                 const float *scale_factors2, const float *level_sigma2_2, int only_stereo, int check_orientation,
                                 int32_t *matches12)
{
    (void)n2;
    const float C2x = oro_gemm_row(T2w, 0, Cw), C2y = oro_gemm_row(T2w, 1, Cw), C2z = oro_gemm_row(T2w, 2, Cw);     /* :667 */
    const float invz = 1.0f / C2z;                                               /* :668 */
    const float ex = fx2 * C2x * invz + cx2, ey = fy2 * C2y * invz + cy2;        /* :669-670 */
    int nmatches = 0, nrot = 0, hist[30];
    int *rot = (int *)malloc(sizeof(int) * 2 * (size_t)(n1 > 0 ? n1 : 1));
    for (int i = 0; i < 30; i++) hist[i] = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                             /* :678 */
    int a = 0, b = 0;
    while (a < nn1 && b < nn2) {                                                 /* :691 */
        if (node1[a] == node2[b]) {
            for (int c1 = off1[a]; c1 < off1[a + 1]; c1++) {
                const int idx1 = idx1v[c1];
                if (has_mp1[idx1]) continue;                                     /* :702 */
                const int bStereo1 = u_right1[idx1] >= 0;
                if (only_stereo && !bStereo1) continue;
                const oro_keypoint *kp1 = &kps1[idx1];
                const uint8_t *d1 = desc1 + (size_t)idx1 * 32;
                int bestDist = 50, bestIdx2 = -1;                                /* TH_LOW :715 */
                for (int c2 = off2[b]; c2 < off2[b + 1]; c2++) {
                    const int idx2 = idx2v[c2];
                    if (has_mp2[idx2]) continue;                                 /* :725 */
                    const int bStereo2 = u_right2[idx2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = oro_descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                    if (dist > 50 || dist > bestDist) continue;                  /* :738 */
                    const oro_keypoint *kp2 = &kps2[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2->x, distey = ey - kp2->y;
                        if (distex * distex + distey * distey < 100 * scale_factors2[kp2->octave]) continue;   /* :747 */
                    }
                    if (check_dist_epipolar_line(kp1, kp2, F12, level_sigma2_2)) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    matches12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        const int bin = oro_rot_bin(kp1->angle, kps2[bestIdx2].angle);
                        rot[2 * nrot] = bin; rot[2 * nrot + 1] = idx1; nrot++;
                        hist[bin]++;
                    }
                }
            }
            a++; b++;
        } else if (node1[a] < node2[b]) a++;
        else b++;
    }
    if (check_orientation) {                                                     /* :791-810 */
        int ind1, ind2, ind3;
        oro_three_maxima(hist, 30, &ind1, &ind2, &ind3);
        for (int k = 0; k < nrot; k++) {
            const int bin = rot[2 * k];
            if (bin == ind1 || bin == ind2 || bin == ind3) continue;
            matches12[rot[2 * k + 1]] = -1;
            nmatches--;
        }
    }
    free(rot);
    return nmatches;
}

/* ---- ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th), src/ORBmatcher.cc:825-975 ----
 * usable[i] = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF) at the time the point is visited (:846-850); Ow = pKF->GetCameraCenter().
 * best_idx[i] = the key-frame feature the point is fused with (:952 passed), or -1.  Returns the number of such points;
 * what happens to them (:954-970: Replace / AddObservation + AddMapPoint) is object-graph work of the caller. */
int oro_fuse(int n_mp, const uint8_t *usable, const float *xw, const float *normal, const float *min_dist_inv, const float *max_dist_inv,
             const float *mf_max_distance, const uint8_t *mp_desc, const float *Tcw, const float *Ow, float fx, float fy, float cx, float cy,
             float bf, const float bounds[4], const float *scale_factors, const float *inv_level_sigma2, int nlevels, float log_scale_factor,
             const oro_grid *g, const oro_keypoint *kps_kf, const float *u_right_kf, const uint8_t *desc_kf, int n_kf, float th,
             int32_t *best_idx)
{
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_kf > 0 ? n_kf : 1));
    int nFused = 0;
    for (int i = 0; i < n_mp; i++) {                                             /* :842 */
        best_idx[i] = -1;
        if (!usable[i]) continue;
        kf_proj p;
        if (!project_into_kf(Tcw, Ow, xw + 3 * (size_t)i, normal + 3 * (size_t)i, fx, fy, cx, cy, bounds, min_dist_inv[i], max_dist_inv[i],
                             mf_max_distance[i], log_scale_factor, nlevels, &p)) continue;
        const float u = p.u, v = p.v;
        const float ur = u - bf * p.invz;                                        /* :870 */
        const float radius = th * scale_factors[p.level];                       /* :890 */
        const int nc = oro_features_in_area(g, kps_kf, u, v, radius, -1, -1, cand, n_kf);
        if (nc <= 0) continue;
        const uint8_t *dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {                                           /* :903 */
            const int idx = cand[c];
            const oro_keypoint *kp = &kps_kf[idx];
            const int kpLevel = kp->octave;
            if (kpLevel < p.level - 1 || kpLevel > p.level) continue;
            if (u_right_kf[idx] >= 0) {                                          /* Check reprojection error in stereo */
                const float ex = u - kp->x, ey = v - kp->y, er = ur - u_right_kf[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - kp->x, ey = v - kp->y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = oro_descriptor_distance(dMP, desc_kf + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= 50) { best_idx[i] = bestIdx; nFused++; }                /* TH_LOW :952 */
    }
    free(cand);
    return nFused;
}

/* ---- ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint), src/ORBmatcher.cc:977-1100 ----
 * usable[i] = !pMP->isBad() && !spAlreadyFound.count(pMP) (:1005).  best_idx / return value as oro_fuse. */
int oro_fuse_sim3(int n_mp, const uint8_t *usable, const float *xw, const float *normal, const float *min_dist_inv, const float *max_dist_inv,
                  const float *mf_max_distance, const uint8_t *mp_desc, const float *Scw, float fx, float fy, float cx, float cy,
                  const float bounds[4], const float *scale_factors, int nlevels, float log_scale_factor, const oro_grid *g,
                  const oro_keypoint *kps_kf, const uint8_t *desc_kf, int n_kf, float th, int32_t *best_idx)
{
    float T[16], Ow[3];
    oro_sim3_decompose(Scw, T, Ow);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_kf > 0 ? n_kf : 1));
    int nFused = 0;
    for (int i = 0; i < n_mp; i++) {                                             /* :1000 */
        best_idx[i] = -1;
        if (!usable[i]) continue;
        kf_proj p;
        if (!project_into_kf(T, Ow, xw + 3 * (size_t)i, normal + 3 * (size_t)i, fx, fy, cx, cy, bounds, min_dist_inv[i], max_dist_inv[i],
                             mf_max_distance[i], log_scale_factor, nlevels, &p)) continue;
        const float radius = th * scale_factors[p.level];                       /* :1049 */
        const int nc = oro_features_in_area(g, kps_kf, p.u, p.v, radius, -1, -1, cand, n_kf);
        if (nc <= 0) continue;
        const uint8_t *dMP = mp_desc + (size_t)i * 32;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int c = 0; c < nc; c++) {                                           /* :1062 */
            const int idx = cand[c];
            const int kpLevel = kps_kf[idx].octave;
            if (kpLevel < p.level - 1 || kpLevel > p.level) continue;
            const int dist = oro_descriptor_distance(dMP, desc_kf + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= 50) { best_idx[i] = bestIdx; nFused++; }                /* TH_LOW :1082 */
    }
    free(cand);
    return nFused;
}

/* one direction of SearchBySim3 (:1148-1225 with (A, B) = (1, 2), :1228-1305 with (A, B) = (2, 1)): the MapPoints of key frame A,
 * moved into camera B by (sRBA, tBA), are searched in key frame B.  match[iA] = feature of B or -1. */
static void sim3_direction(int nA, const uint8_t *usableA, const float *xwA, const float *min_dist, const float *max_dist,
                           const float *mf_max_distance, const uint8_t *mp_desc, const float *TAw, const float sR[9], const float t[3],
                           float fx, float fy, float cx, float cy, const float boundsB[4], const float *scale_factorsB, int nlevelsB,
                           float log_scale_factorB, const oro_grid *gB, const oro_keypoint *kpsB, const uint8_t *descB, int nB, float th,
                           int32_t *match)
{
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nB > 0 ? nB : 1));
    for (int i = 0; i < nA; i++) {
        match[i] = -1;
        if (!usableA[i]) continue;                                               /* :1152-1156 */
        const float *X = xwA + 3 * (size_t)i;
        const float pA[3] = {oro_gemm_row(TAw, 0, X), oro_gemm_row(TAw, 1, X), oro_gemm_row(TAw, 2, X)};       /* :1159 */
        float pB[3];
        for (int r = 0; r < 3; r++) {                                            /* sR21*p3Dc1 + t21 :1160: the same small-matrix path */
            const float t0 = sR[3 * r] * pA[0] + sR[3 * r + 1] * pA[1] + sR[3 * r + 2] * pA[2];
            pB[r] = (float)((double)t0 * 1.0 + (double)t[r] * 1.0);
        }
        if (pB[2] < 0.0f) continue;                                              /* :1163 */
        const float invz = (float)(1.0 / pB[2]);                                 /* `1.0/p3Dc2.at<float>(2)`: a double division, rounded by the float it is assigned to */
        const float x = pB[0] * invz, y = pB[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= boundsB[0] && u < boundsB[1] && v >= boundsB[2] && v < boundsB[3])) continue;               /* :1174 */
        double nn = 0;
        for (int k = 0; k < 3; k++) nn += (double)pB[k] * (double)pB[k];
        const float dist3D = (float)sqrt(nn);                                    /* cv::norm(p3Dc2) :1179 */
        if (dist3D < min_dist[i] || dist3D > max_dist[i]) continue;
        const int nPredictedLevel = oro_predict_scale(mf_max_distance[i], dist3D, log_scale_factorB, nlevelsB);
        const float radius = th * scale_factorsB[nPredictedLevel];              /* :1189 */
        const int nc = oro_features_in_area(gB, kpsB, u, v, radius, -1, -1, cand, nB);
        if (nc <= 0) continue;
        const uint8_t *dMP = mp_desc + (size_t)i * 32;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int c = 0; c < nc; c++) {                                           /* :1201 */
            const int idx = cand[c];
            if (kpsB[idx].octave < nPredictedLevel - 1 || kpsB[idx].octave > nPredictedLevel) continue;
            const int dist = oro_descriptor_distance(dMP, descB + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= 100) match[i] = bestIdx;                                 /* TH_HIGH :1221 */
    }
    free(cand);
}

/* ---- ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), src/ORBmatcher.cc:1102-1326 ----
 * usable1[i] = pMP && !vbAlreadyMatched1[i] && !pMP->isBad() (:1152-1156), usable2 likewise (:1232-1236); T1w / T2w = the key
 * frames' [R|t]; R12 row-major 3x3.  The calibration is pKF1's for both directions (:1105-1108).  match12[i1] = the feature of
 * KF2 whose MapPoint this call writes into vpMatches12[i1], or -1.  Returns nFound. */
int oro_search_by_sim3(int n1, const uint8_t *usable1, const float *xw1, const float *min_dist1, const float *max_dist1,
                       const float *mf_max1, const uint8_t *mp_desc1, const float *T1w, const float bounds1[4], const float *scale_factors1,
                       int nlevels1, float log_scale_factor1, const oro_grid *g1, const oro_keypoint *kps1, const uint8_t *desc1,
                       int n2, const uint8_t *usable2, const float *xw2, const float *min_dist2, const float *max_dist2,
                       const float *mf_max2, const uint8_t *mp_desc2, const float *T2w, const float bounds2[4], const float *scale_factors2,
                       int nlevels2, float log_scale_factor2, const oro_grid *g2, const oro_keypoint *kps2, const uint8_t *desc2,
                       float fx, float fy, float cx, float cy, float s12, const float *R12, const float *t12, float th, int32_t *match12)
{
    float sR12[9], sR21[9], t21[3];
    const float a12 = (float)(double)s12;                                        /* s12*R12: alpha = s12, convertTo multiplies by (float)alpha :1119 */
    const float a21 = (float)(1.0 / (double)s12);                                /* (1.0/s12)*R12.t(): MatOp_T with alpha = 1.0/s12 :1120 */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { sR12[3 * i + j] = R12[3 * i + j] * a12 + 0.0f; sR21[3 * i + j] = R12[3 * j + i] * a21 + 0.0f; }
    for (int r = 0; r < 3; r++) {                                                /* t21 = -sR21*t12 :1121: alpha = -1 */
        const float t0 = sR21[3 * r] * t12[0] + sR21[3 * r + 1] * t12[1] + sR21[3 * r + 2] * t12[2];
        t21[r] = (float)((double)t0 * -1.0 + 0.0 * 0.0);       /* no C operand: c = zerof, beta = 0 (a zero sum comes out as +0) */
    }
    int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1));   /* vnMatch1 :1144 */
    int32_t *m2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1));   /* vnMatch2 :1145 */
    sim3_direction(n1, usable1, xw1, min_dist1, max_dist1, mf_max1, mp_desc1, T1w, sR21, t21, fx, fy, cx, cy, bounds2, scale_factors2, nlevels2,
                   log_scale_factor2, g2, kps2, desc2, n2, th, m1);
    sim3_direction(n2, usable2, xw2, min_dist2, max_dist2, mf_max2, mp_desc2, T2w, sR12, t12, fx, fy, cx, cy, bounds1, scale_factors1, nlevels1,
                   log_scale_factor1, g1, kps1, desc1, n1, th, m2);
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {                                            /* Check agreement :1310-1323 */
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    free(m1); free(m2);
    return nFound;
}
