// orbm_kf.hip -- the ORBmatcher methods of the LocalMapping / LoopClosing threads on gfx950.
// Reference (WChen09/My-SLAM), all in src/ORBmatcher.cc:
//   :290-403   SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)      orbm_search_by_projection_sim3
//   :522-655   SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)                   orbm_search_by_bow_kf (orbm.hip, beside its sibling)
//   :657-823   SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bStereo)  orbm_search_for_triangulation
//   :825-975   Fuse(KeyFrame*, vpMapPoints, th)                                 orbm_fuse
//   :977-1100  Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)               orbm_fuse_sim3
//   :1102-1326 SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)         orbm_search_by_sim3
// Division of labour, as for the Tracking-thread matchers (orbm_grid.hip): the cv::Mat algebra of a call (a handful of 3x3
// products per MapPoint) runs on the host with OpenCV 3.1.0's arithmetic (orbm_sim3_decompose, orbm_project_points_kf, ...);
// MapPoint::PredictScale stays with the caller's MapPoint; the windows (KeyFrame::GetFeaturesInArea), the per-candidate
// predicates and the Hamming distances run on the GPU.  Four of the six inner loops carry no state from one MapPoint / feature
// to the next (Fuse x 2, SearchBySim3's two directions, SearchForTriangulation -- its vbMatched2 is never set), so their
// selection runs on the GPU too, one wave per query; SearchByProjection(KeyFrame*, Scw, ...) blocks a key-frame slot for every
// later MapPoint (:375, :396), so its candidate lists and distances come back and the reference's scan runs on the host.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "orbm_internal.h"
#include "orbx_internal.h"

// -------------------------------------------------------------------------------------------------
// host side: cv::Mat algebra as OpenCV 3.1.0 evaluates it
// -------------------------------------------------------------------------------------------------
// `R*x + t` on a 3x3 and a 3x1 float matrix is cv::gemm(R, x, 1, t, 1, dst, 0), whose small-matrix path (modules/core/src/matmul.cpp:
// flags == 0, 2 <= len <= 4) sums the three float products in float, left to right, and finishes with (float)(t0*alpha + c*beta) in
// double.  `-A.t()*b` materialises the transpose and runs the same path with alpha = -1.
static inline float gemm3(const float *a, int sa, const float *b, double alpha, float c, double beta)
{
    const float t0 = a[0] * b[0] + a[sa] * b[1] + a[2 * sa] * b[2];
    return (float)((double)t0 * alpha + (double)c * beta);
}
static inline void camera_center(const float *T, float Ow[3])           // -Rcw^T tcw
{
    const float t[3] = {T[3], T[7], T[11]};
    for (int k = 0; k < 3; k++) Ow[k] = gemm3(T + k, 4, t, -1.0, 0.f, 0.0);
}

// Scw -> [Rcw|tcw] (row-major 4x4) and Ow: src/ORBmatcher.cc:299-303 and :986-990.
//   scw = sqrt(sRcw.row(0).dot(sRcw.row(0)))   Mat::dot accumulates in double; the sqrt is a double one, stored in a float
//   Rcw = sRcw/scw, tcw = Scw.col(3)/scw       Mat / s is MatOp_AddEx with alpha = 1./s, materialised by convertTo, whose 32f kernel
//                                              multiplies by (float)alpha in float
//   Ow = -Rcw.t()*tcw
extern "C" int orbm_sim3_decompose(const float *Scw, float *Tcw, float *Ow)
{
    if (!Scw || !Tcw || !Ow) return mfail(ORBX_E_INVALID, "NULL argument");
    double dd = 0;
    for (int k = 0; k < 3; k++) dd += (double)Scw[k] * (double)Scw[k];
    const float scw = (float)sqrt(dd);
    const float inv = (float)(1.0 / (double)scw);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) Tcw[4 * i + j] = Scw[4 * i + j] * inv + 0.0f;
    Tcw[12] = 0.f; Tcw[13] = 0.f; Tcw[14] = 0.f; Tcw[15] = 1.f;
    camera_center(Tcw, Ow);
    return ORBX_OK;
}

// SearchBySim3's :1119-1121: sR12 = s12*R12, sR21 = (1.0/s12)*R12.t(), t21 = -sR21*t12 (3x3 row-major, 3-vectors)
extern "C" int orbm_sim3_relative(float s12, const float *R12, const float *t12, float *sR12, float *sR21, float *t21)
{
    if (!R12 || !t12 || !sR12 || !sR21 || !t21) return mfail(ORBX_E_INVALID, "NULL argument");
    const float a12 = (float)(double)s12, a21 = (float)(1.0 / (double)s12);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { sR12[3 * i + j] = R12[3 * i + j] * a12 + 0.0f; sR21[3 * i + j] = R12[3 * j + i] * a21 + 0.0f; }
    for (int r = 0; r < 3; r++) t21[r] = gemm3(sR21 + 3 * r, 1, t12, -1.0, 0.f, 0.0);
    return ORBX_OK;
}

// The projection block of :320-355, :852-885, :1008-1043 for n world points: p3Dc = Rcw*p3Dw+tcw, the depth test, u / v, KeyFrame::
// IsInImage (bounds = the key frame's int mnMinX, mnMaxX, mnMinY, mnMaxY as floats; [min, max)), PO = p3Dw - Ow, dist = cv::norm(PO)
// (double accumulation), and with normals the viewing-angle test PO.dot(Pn) < 0.5*dist (Mat::dot: double).  ok[i] = every one of
// those tests passed; the distance-invariance test and PredictScale belong to the caller's MapPoint.  Ow == NULL: -Rcw^T tcw.
extern "C" int orbm_project_points_kf(const float *Tcw, const float *Ow, float fx, float fy, float cx, float cy, const float bounds[4],
                                      const float *xw, const float *normal, int n, float *u, float *v, float *invz, float *dist3d,
                                      uint8_t *ok)
{
    if (!Tcw || !bounds || n < 0 || (n > 0 && (!xw || !u || !v || !dist3d || !ok))) return mfail(ORBX_E_INVALID, "bad argument");
    float ow[3];
    if (Ow) memcpy(ow, Ow, sizeof ow); else camera_center(Tcw, ow);
    for (int i = 0; i < n; i++) {
        const float *X = xw + 3 * (size_t)i;
        const float xc = gemm3(Tcw, 1, X, 1.0, Tcw[3], 1.0), yc = gemm3(Tcw + 4, 1, X, 1.0, Tcw[7], 1.0), zc = gemm3(Tcw + 8, 1, X, 1.0, Tcw[11], 1.0);
        const float iz = 1 / zc;
        const float x = xc * iz, y = yc * iz;
        u[i] = fx * x + cx; v[i] = fy * y + cy;
        bool good = !(zc < 0.0f) && (u[i] >= bounds[0] && u[i] < bounds[1] && v[i] >= bounds[2] && v[i] < bounds[3]);
        float PO[3];
        double nn = 0;
        for (int k = 0; k < 3; k++) { PO[k] = X[k] - ow[k]; nn += (double)PO[k] * (double)PO[k]; }
        const float dist = (float)sqrt(nn);
        if (normal) {
            double dot = 0;
            for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)normal[3 * (size_t)i + k];
            if (dot < 0.5 * dist) good = false;
        }
        dist3d[i] = dist;
        if (invz) invz[i] = iz;
        ok[i] = good;
    }
    return ORBX_OK;
}

// SearchBySim3's :1158-1179 (and :1238-1259 with the roles swapped): p = sR*(R_A x + t_A) + t, depth, u / v, IsInImage of the other
// key frame, dist3D = cv::norm(p).  `1.0/z` is a double division rounded by the float it initialises.
extern "C" int orbm_project_points_sim3(const float *TAw, const float *sR, const float *t, float fx, float fy, float cx, float cy,
                                        const float boundsB[4], const float *xw, int n, float *u, float *v, float *dist3d, uint8_t *ok)
{
    if (!TAw || !sR || !t || !boundsB || n < 0 || (n > 0 && (!xw || !u || !v || !dist3d || !ok))) return mfail(ORBX_E_INVALID, "bad argument");
    for (int i = 0; i < n; i++) {
        const float *X = xw + 3 * (size_t)i;
        const float pA[3] = {gemm3(TAw, 1, X, 1.0, TAw[3], 1.0), gemm3(TAw + 4, 1, X, 1.0, TAw[7], 1.0), gemm3(TAw + 8, 1, X, 1.0, TAw[11], 1.0)};
        float pB[3];
        for (int r = 0; r < 3; r++) pB[r] = gemm3(sR + 3 * r, 1, pA, 1.0, t[r], 1.0);
        const float iz = (float)(1.0 / pB[2]);
        const float x = pB[0] * iz, y = pB[1] * iz;
        u[i] = fx * x + cx; v[i] = fy * y + cy;
        double nn = 0;
        for (int k = 0; k < 3; k++) nn += (double)pB[k] * (double)pB[k];
        dist3d[i] = (float)sqrt(nn);
        ok[i] = !(pB[2] < 0.0f) && (u[i] >= boundsB[0] && u[i] < boundsB[1] && v[i] >= boundsB[2] && v[i] < boundsB[3]);
    }
    return ORBX_OK;
}

// -------------------------------------------------------------------------------------------------
// kernels
// -------------------------------------------------------------------------------------------------
struct KfGate {                 // Fuse's reprojection test (:914-938); on = 0 for the variants without one
    int on;
    const float *q_ur;          // per query: ur = u - bf*invz
    const float *t_uright;      // per key-frame feature: mvuRight
    float inv_sigma2[ORBX_MAX_LEVELS];
};

// One wave per projected MapPoint: KeyFrame::GetFeaturesInArea(u, v, radius) in the reference's order, the octave window
// [level - 1, level] (:380, :911, :1067, :1207), Fuse's chi-square gate, the Hamming distance, strict '<' (first candidate wins a tie).
__global__ __launch_bounds__(M_THREADS) void k_search_kf(OrbmGrid g, const uint8_t *__restrict__ qdesc, const float *__restrict__ qx,
                                                        const float *__restrict__ qy, const float *__restrict__ qr,
                                                        const int32_t *__restrict__ qlevel, int nq, const uint8_t *__restrict__ tdesc,
                                                        KfGate gate, int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (M_THREADS / 64) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float x = qx[q], y = qy[q], r = qr[q];
    const int lv = qlevel[q];
    const float ur = gate.on ? gate.q_ur[q] : 0.f;
    const uint4 *Q = reinterpret_cast<const uint4 *>(qdesc) + 2 * (long long)q;
    const uint4 q0 = Q[0], q1 = Q[1];
    uint32_t bp = 0xFFFFFFFFu;
    int bidx = -1, n = 0;
    int cx0, cx1, cy0, cy1;
    if (window_cells(g, x, y, r, cx0, cx1, cy0, cy1)) {
        for (int ix = cx0; ix <= cx1; ix++) {
            const int s = g.cell_start[ix * ORBM_GRID_ROWS + cy0], e = g.cell_start[ix * ORBM_GRID_ROWS + cy1 + 1];
            for (int j0 = s; j0 < e; j0 += 64) {
                const int j = j0 + lane;
                int i = -1;
                bool ok = false;
                if (j < e) {
                    i = g.items[j];
                    ok = in_window(g, i, x, y, r, -1, -1);                // the window itself has no level test (src/KeyFrame.cc:569-606)
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);   // positions count every window member, as vIndices does
                if (ok) {
                    const int oct = g.koct[i];
                    bool use = !(oct < lv - 1 || oct > lv);
                    if (use && gate.on) {
                        const float ex = __fsub_rn(x, g.kx[i]), ey = __fsub_rn(y, g.ky[i]);
                        const float kr = gate.t_uright[i];
                        const float inv = gate.inv_sigma2[min(max(oct, 0), ORBX_MAX_LEVELS - 1)];
                        float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                        if (kr >= 0) {
                            const float er = __fsub_rn(ur, kr);
                            e2 = __fadd_rn(e2, __fmul_rn(er, er));
                            use = !((double)__fmul_rn(e2, inv) > 7.8);          // :925
                        } else {
                            use = !((double)__fmul_rn(e2, inv) > 5.99);         // :936
                        }
                    }
                    if (use) {
                        const uint4 *Tj = reinterpret_cast<const uint4 *>(tdesc) + 2 * (long long)i;
                        const int d = hamming256(q0, q1, Tj[0], Tj[1]);
                        const uint32_t p = ((uint32_t)d << 22) | (uint32_t)min(orbx_prefix_cnt(m, n), 0x3FFFFF);
                        if (p < bp) { bp = p; bidx = i; }
                    }
                }
                n += __popcll(m);
            }
        }
    }
    const uint32_t B = wave_min_u32(bp);
    if (B == 0xFFFFFFFFu) {
        if (lane == 0) { best_idx[q] = -1; best_d[q] = 256; }
    } else if (bp == B) {                       // positions are unique: one lane holds the winner
        best_idx[q] = bidx; best_d[q] = (int)(B >> 22);
    }
}

struct TriParams {
    float F12[9];               // row-major
    float ex, ey;               // epipole of camera 1 in image 2 (:664-670)
    float thr_epipole[ORBX_MAX_LEVELS];     // 100*pKF2->mvScaleFactors[level]  (int * float -> float, :747)
    double thr_line[ORBX_MAX_LEVELS];       // 3.84*pKF2->mvLevelSigma2[level]  (double * float -> double, :156)
};
#define TRI_KEY_NONE 0xFFFFFFFFu
// SearchForTriangulation: one wave per feature of key frame 1 that has no MapPoint yet; the lanes walk the features of key frame 2
// in the same vocabulary node.  The reference accepts a candidate when `dist <= TH_LOW && dist <= bestDist` and both epipolar tests
// pass (:738-755), and bestDist only moves when a candidate is accepted: the result is the LAST candidate of minimal distance among
// those that pass the stateless tests -> minimum of (distance, -position).
__global__ __launch_bounds__(M_THREADS) void k_triangulation(const int4 *__restrict__ queries, int nq, const int32_t *__restrict__ idx2v,
                                                            const uint8_t *__restrict__ desc1, const uint8_t *__restrict__ desc2,
                                                            const float2 *__restrict__ xy1, const float2 *__restrict__ xy2,
                                                            const int32_t *__restrict__ oct2, const uint8_t *__restrict__ flags2,
                                                            TriParams P, int32_t *__restrict__ match)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (M_THREADS / 64) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int4 rec = queries[q];            // x = idx1, y / z = [lo, hi) in idx2v, w = bStereo1
    const int idx1 = rec.x;
    const uint4 *Q = reinterpret_cast<const uint4 *>(desc1) + 2 * (long long)idx1;
    const uint4 q0 = Q[0], q1 = Q[1];
    const float2 p1 = xy1[idx1];
    // epipolar line in image 2, l = x1' F12 (:143-145)
    const float a = __fadd_rn(__fadd_rn(__fmul_rn(p1.x, P.F12[0]), __fmul_rn(p1.y, P.F12[3])), P.F12[6]);
    const float b = __fadd_rn(__fadd_rn(__fmul_rn(p1.x, P.F12[1]), __fmul_rn(p1.y, P.F12[4])), P.F12[7]);
    const float c = __fadd_rn(__fadd_rn(__fmul_rn(p1.x, P.F12[2]), __fmul_rn(p1.y, P.F12[5])), P.F12[8]);
    const float den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
    uint32_t best = TRI_KEY_NONE;
    for (int pos = rec.y + lane; pos < rec.z; pos += 64) {
        const int idx2 = idx2v[pos];
        const uint8_t f2 = flags2[idx2];     // bit 0: eligible (no MapPoint, stereo filter passed :725-732), bit 1: bStereo2
        if (!(f2 & 1)) continue;
        const uint4 *Tj = reinterpret_cast<const uint4 *>(desc2) + 2 * (long long)idx2;
        const int d = hamming256(q0, q1, Tj[0], Tj[1]);
        if (d > ORBM_TH_LOW) continue;                                           // :738
        const float2 p2 = xy2[idx2];
        const int o2 = min(max(oct2[idx2], 0), ORBX_MAX_LEVELS - 1);
        if (!rec.w && !(f2 & 2)) {                                               // :743-749
            const float dx = __fsub_rn(P.ex, p2.x), dy = __fsub_rn(P.ey, p2.y);
            if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < P.thr_epipole[o2]) continue;
        }
        if (den == 0) continue;                                                  // CheckDistEpipolarLine :147-156
        const float num = __fadd_rn(__fadd_rn(__fmul_rn(a, p2.x), __fmul_rn(b, p2.y)), c);
        const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
        if (!((double)dsqr < P.thr_line[o2])) continue;
        const uint32_t key = ((uint32_t)d << 20) | (0xFFFFFu - (uint32_t)min(pos - rec.y, 0xFFFFF));
        best = min(best, key);
    }
    const uint32_t B = wave_min_u32(best);
    if (lane == 0) match[q] = B == TRI_KEY_NONE ? -1 : idx2v[rec.y + (int)(0xFFFFFu - (B & 0xFFFFFu))];
}

// -------------------------------------------------------------------------------------------------
// staging: the inputs of one call go up in one copy (pinned arena -> its device mirror), or through a temporary block on the first
// call, before the arena has grown
// -------------------------------------------------------------------------------------------------
struct InBlock {
    orbm_matcher *m;
    std::vector<std::pair<const void *, size_t>> parts;
    std::vector<size_t> offs;
    size_t total = 0;
    uint8_t *dev = nullptr;
    void *tmp_dev = nullptr;
    explicit InBlock(orbm_matcher *m_) : m(m_) {}
    int add(const void *host, size_t bytes)
    {
        parts.emplace_back(host, bytes);
        offs.push_back(total);
        total += (bytes + 63) & ~(size_t)63;
        return (int)parts.size() - 1;
    }
    int upload(hipStream_t s)
    {
        if (total == 0) return ORBX_OK;
        const size_t mark = m->arena_used;
        uint8_t *dst_dev = nullptr;
        if (m->d_arena && m->arena_used + total <= m->arena_cap) {
            uint8_t *h = m->arena + m->arena_used;
            for (size_t i = 0; i < parts.size(); i++)
                if (parts[i].second) memcpy(h + offs[i], parts[i].first, parts[i].second);
            dst_dev = m->d_arena + m->arena_used;
            m->arena_used += total; m->arena_want += total;
            int rc = orbm_flush_in(m, mark, s);
            if (rc != ORBX_OK) return rc;
        } else {
            m->arena_want += total;                      // the next call has room
            std::vector<uint8_t> h(total);
            for (size_t i = 0; i < parts.size(); i++)
                if (parts[i].second) memcpy(h.data() + offs[i], parts[i].first, parts[i].second);
            MHIPCHK(hipMalloc(&tmp_dev, total));
            dst_dev = (uint8_t *)tmp_dev;
            MHIPCHK(hipMemcpyAsync(dst_dev, h.data(), total, hipMemcpyHostToDevice, s));
            MHIPCHK(hipStreamSynchronize(s));            // h goes out of scope
        }
        dev = dst_dev;
        return ORBX_OK;
    }
    template <class T> const T *at(int part) const { return parts[part].second ? reinterpret_cast<const T *>(dev + offs[part]) : nullptr; }
    void release() { if (tmp_dev) { (void)hipFree(tmp_dev); tmp_dev = nullptr; } }
};

// the compacted queries of one stateless window search, and its launch
struct KfQueries {
    std::vector<int> src;                   // original MapPoint index
    std::vector<float> x, y, r, ur;
    std::vector<int32_t> level;
    std::vector<uint8_t> desc;
    int build(int n_mp, const uint8_t *use, const float *u, const float *v, const float *proj_ur, const int32_t *pred_level, const uint8_t *mp_desc,
              const float *scale_factors, int nlevels, float th)
    {
        for (int i = 0; i < n_mp; i++) {
            if (!use[i]) continue;
            const int lv = pred_level[i];
            if (lv < 0 || lv >= nlevels) return mfail(ORBX_E_INVALID, "MapPoint %d predicted on level %d of %d", i, lv, nlevels);
            src.push_back(i); x.push_back(u[i]); y.push_back(v[i]); r.push_back(th * scale_factors[lv]); level.push_back(lv);
            if (proj_ur) ur.push_back(proj_ur[i]);
        }
        desc.resize(src.size() * 32);
        for (size_t k = 0; k < src.size(); k++) memcpy(&desc[k * 32], mp_desc + (size_t)src[k] * 32, 32);
        return ORBX_OK;
    }
};

// queues one k_search_kf over grid slot g; results land in d_res[0 .. nq) (indices) and d_res[nq .. 2 nq) (distances)
static int launch_search_kf(orbm_matcher *m, const OrbmGrid &g, const KfQueries &Q, const uint8_t *desc_kf, int n_kf, const float *u_right_kf,
                            const float *inv_level_sigma2, int nlevels, int32_t *d_res, InBlock &in, hipStream_t s)
{
    const int nq = (int)Q.src.size();
    const int px = in.add(Q.x.data(), (size_t)nq * 4), py = in.add(Q.y.data(), (size_t)nq * 4), pr = in.add(Q.r.data(), (size_t)nq * 4);
    const int pl = in.add(Q.level.data(), (size_t)nq * 4), pd = in.add(Q.desc.data(), (size_t)nq * 32), pt = in.add(desc_kf, (size_t)n_kf * 32);
    const bool gated = u_right_kf != nullptr;
    const int pu = gated ? in.add(Q.ur.data(), (size_t)nq * 4) : -1, pk = gated ? in.add(u_right_kf, (size_t)n_kf * 4) : -1;
    int rc = in.upload(s);
    if (rc != ORBX_OK) return rc;
    KfGate gate = {};
    gate.on = gated ? 1 : 0;
    if (gated) {
        gate.q_ur = in.at<float>(pu); gate.t_uright = in.at<float>(pk);
        for (int l = 0; l < ORBX_MAX_LEVELS; l++) gate.inv_sigma2[l] = l < nlevels ? inv_level_sigma2[l] : 0.f;
    }
    hipLaunchKernelGGL(k_search_kf, dim3((nq + 3) / 4), dim3(M_THREADS), 0, s, g, in.at<uint8_t>(pd), in.at<float>(px), in.at<float>(py),
                       in.at<float>(pr), in.at<int32_t>(pl), nq, in.at<uint8_t>(pt), gate, d_res, d_res + nq);
    MHIPCHK(hipGetLastError());
    return ORBX_OK;
}

static int check_kf_args(orbm_matcher *m, int n_mp, const uint8_t *use, const float *u, const float *v, const int32_t *lvl, const uint8_t *desc,
                         const float *sf, int nlevels, const orbx_keypoint *kps_kf, const uint8_t *desc_kf, int n_kf)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n_mp < 0 || n_kf < 0 || !sf || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || (n_mp > 0 && (!use || !u || !v || !lvl || !desc)) ||
        (n_kf > 0 && (!kps_kf || !desc_kf)))
        return mfail(ORBX_E_INVALID, "bad argument");
    return ORBX_OK;
}

// Fuse's search (:889-952) and Fuse(Scw)'s (:1048-1082) for every usable MapPoint: best_idx[i] = the key-frame feature with the
// smallest distance in the window / octave window (/ chi-square gate), if that distance is <= max_dist, else -1.
static int stateless_search(orbm_matcher *m, int n_mp, const uint8_t *use, const float *proj_u, const float *proj_v, const float *proj_ur,
                            const int32_t *pred_level, const uint8_t *mp_desc, const float *scale_factors, const float *inv_level_sigma2,
                            int nlevels, const float *u_right_kf, const uint8_t *desc_kf, int n_kf, float th, int max_dist,
                            int32_t *best_idx, int *count)
{
    *count = 0;
    for (int i = 0; i < n_mp; i++) best_idx[i] = -1;
    if (n_mp == 0 || n_kf == 0) return ORBX_OK;
    if (!m->grid_ok || m->grid.n != n_kf) return mfail(ORBX_E_INVALID, "orbm_grid_build_kf(key frame) has not been called");
    KfQueries Q;
    int rc = Q.build(n_mp, use, proj_u, proj_v, proj_ur, pred_level, mp_desc, scale_factors, nlevels, th);
    if (rc != ORBX_OK) return rc;
    const int nq = (int)Q.src.size();
    if (nq == 0) return ORBX_OK;
    MHIPCHK(hipSetDevice(m->device));
    rc = orbm_grow(m, nq, 0, 0);
    if (rc != ORBX_OK) return rc;
    rc = orbm_arena_begin(m);
    if (rc != ORBX_OK) return rc;
    hipStream_t s = m->stream;
    InBlock in(m);
    rc = launch_search_kf(m, m->grid, Q, desc_kf, n_kf, proj_ur ? u_right_kf : nullptr, inv_level_sigma2, nlevels, m->d_out, in, s);
    if (rc != ORBX_OK) { in.release(); return rc; }
    std::vector<int32_t> res((size_t)2 * nq);
    rc = orbm_d2h(m, res.data(), m->d_out, (size_t)2 * nq * 4, s);
    if (rc == ORBX_OK) rc = orbm_sync(m, s);
    in.release();
    if (rc != ORBX_OK) return rc;
    int cnt = 0;
    for (int k = 0; k < nq; k++)
        if (res[k] >= 0 && res[(size_t)nq + k] <= max_dist) { best_idx[Q.src[k]] = res[k]; cnt++; }
    *count = cnt;
    return ORBX_OK;
}

// -------------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------------
extern "C" int orbm_fuse(orbm_matcher *m, int n_mp, const uint8_t *use, const float *proj_u, const float *proj_v, const float *proj_ur,
                         const int32_t *pred_level, const uint8_t *mp_desc, const float *scale_factors, const float *inv_level_sigma2,
                         int nlevels, const orbx_keypoint *kps_kf, const float *u_right_kf, const uint8_t *desc_kf, int n_kf, float th,
                         int32_t *best_idx, int *nfused)
{
    int rc = check_kf_args(m, n_mp, use, proj_u, proj_v, pred_level, mp_desc, scale_factors, nlevels, kps_kf, desc_kf, n_kf);
    if (rc != ORBX_OK) return rc;
    if (!best_idx || !nfused || !inv_level_sigma2 || (n_mp > 0 && !proj_ur) || (n_kf > 0 && !u_right_kf)) return mfail(ORBX_E_INVALID, "bad argument");
    return stateless_search(m, n_mp, use, proj_u, proj_v, proj_ur, pred_level, mp_desc, scale_factors, inv_level_sigma2, nlevels, u_right_kf,
                            desc_kf, n_kf, th, ORBM_TH_LOW, best_idx, nfused);
}

extern "C" int orbm_fuse_sim3(orbm_matcher *m, int n_mp, const uint8_t *use, const float *proj_u, const float *proj_v,
                              const int32_t *pred_level, const uint8_t *mp_desc, const float *scale_factors, int nlevels,
                              const orbx_keypoint *kps_kf, const uint8_t *desc_kf, int n_kf, float th, int32_t *best_idx, int *nfused)
{
    int rc = check_kf_args(m, n_mp, use, proj_u, proj_v, pred_level, mp_desc, scale_factors, nlevels, kps_kf, desc_kf, n_kf);
    if (rc != ORBX_OK) return rc;
    if (!best_idx || !nfused) return mfail(ORBX_E_INVALID, "bad argument");
    return stateless_search(m, n_mp, use, proj_u, proj_v, nullptr, pred_level, mp_desc, scale_factors, nullptr, nlevels, nullptr,
                            desc_kf, n_kf, th, ORBM_TH_LOW, best_idx, nfused);
}

extern "C" int orbm_search_by_projection_sim3(orbm_matcher *m, int n_mp, const uint8_t *use, const float *proj_u, const float *proj_v,
                                              const int32_t *pred_level, const uint8_t *mp_desc, const float *scale_factors, int nlevels,
                                              const orbx_keypoint *kps_kf, const uint8_t *desc_kf, int n_kf, int th,
                                              uint8_t *kf_matched, int32_t *kf_match, int *nmatches)
{
    int rc = check_kf_args(m, n_mp, use, proj_u, proj_v, pred_level, mp_desc, scale_factors, nlevels, kps_kf, desc_kf, n_kf);
    if (rc != ORBX_OK) return rc;
    if (!nmatches || (n_kf > 0 && (!kf_matched || !kf_match))) return mfail(ORBX_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n_kf; i++) kf_match[i] = -1;
    if (n_mp == 0 || n_kf == 0) return ORBX_OK;
    if (!m->grid_ok || m->grid.n != n_kf) return mfail(ORBX_E_INVALID, "orbm_grid_build_kf(key frame) has not been called");
    std::vector<int> qi;
    std::vector<float> x, y, r;
    std::vector<int32_t> mn, mx;
    for (int i = 0; i < n_mp; i++) {
        if (!use[i]) continue;
        const int lv = pred_level[i];
        if (lv < 0 || lv >= nlevels) return mfail(ORBX_E_INVALID, "MapPoint %d predicted on level %d of %d", i, lv, nlevels);
        qi.push_back(i); x.push_back(proj_u[i]); y.push_back(proj_v[i]); r.push_back(th * scale_factors[lv]);     // :360
        // the octave test of :380 drops candidates without any other effect, so it rides in the window query: with (lv - 1, lv) the
        // level branch of GetFeaturesInArea's Frame twin is `octave < lv - 1 || octave > lv` for every lv >= 0
        mn.push_back(lv - 1); mx.push_back(lv);
    }
    const int nq = (int)qi.size();
    if (nq == 0) return ORBX_OK;
    { int rc_ = orbm_grow(m, nq, 0, 0); if (rc_ != ORBX_OK) return rc_; }
    std::vector<int32_t> off, idx, dist;
    std::vector<uint8_t> qd((size_t)nq * 32);
    for (int k = 0; k < nq; k++) memcpy(&qd[(size_t)k * 32], mp_desc + (size_t)qi[k] * 32, 32);
    const int total = orbm_area_pairs(m, x.data(), y.data(), r.data(), mn.data(), mx.data(), nq, qd.data(), desc_kf, n_kf, off, idx, dist);
    if (total < 0) return total;
    int nm = 0;
    for (int k = 0; k < nq; k++) {                  // the sequential scan (:370-398): a match blocks its slot for every later MapPoint
        int bestDist = 256, bestIdx = -1;
        for (int c = off[k]; c < off[k + 1]; c++) {
            const int i2 = idx[c];
            if (kf_matched[i2]) continue;           // :375
            const int d = dist[c];
            if (d < bestDist) { bestDist = d; bestIdx = i2; }
        }
        if (bestDist <= ORBM_TH_LOW) { kf_matched[bestIdx] = 1; kf_match[bestIdx] = qi[k]; nm++; }
    }
    *nmatches = nm;
    return ORBX_OK;
}

extern "C" int orbm_search_by_sim3(orbm_matcher *m,
                                   int n_mp1, const uint8_t *use1, const float *proj_u1, const float *proj_v1, const int32_t *pred_level1,
                                   const uint8_t *mp_desc1,
                                   int n_mp2, const uint8_t *use2, const float *proj_u2, const float *proj_v2, const int32_t *pred_level2,
                                   const uint8_t *mp_desc2,
                                   const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const orbm_kf_grid *grid1, const float *scale_factors1,
                                   int nlevels1,
                                   const orbx_keypoint *kps2, const uint8_t *desc2, int n2, const orbm_kf_grid *grid2, const float *scale_factors2,
                                   int nlevels2, float th, int32_t *match12, int *nfound)
{
    int rc = check_kf_args(m, n_mp1, use1, proj_u1, proj_v1, pred_level1, mp_desc1, scale_factors2, nlevels2, kps2, desc2, n2);
    if (rc != ORBX_OK) return rc;
    rc = check_kf_args(m, n_mp2, use2, proj_u2, proj_v2, pred_level2, mp_desc2, scale_factors1, nlevels1, kps1, desc1, n1);
    if (rc != ORBX_OK) return rc;
    if (!grid1 || !grid2 || !nfound || (n_mp1 > 0 && !match12)) return mfail(ORBX_E_INVALID, "bad argument");
    if (n_mp1 != n1 || n_mp2 != n2) return mfail(ORBX_E_INVALID, "one MapPoint slot per key-frame feature: n_mp1 = %d / n1 = %d, n_mp2 = %d / n2 = %d", n_mp1, n1, n_mp2, n2);
    *nfound = 0;
    for (int i = 0; i < n_mp1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    KfQueries Q1, Q2;                                // Q1: MapPoints of key frame 1 searched in key frame 2 (:1148-1225); Q2: the reverse (:1228-1305)
    rc = Q1.build(n_mp1, use1, proj_u1, proj_v1, nullptr, pred_level1, mp_desc1, scale_factors2, nlevels2, th);
    if (rc != ORBX_OK) return rc;
    rc = Q2.build(n_mp2, use2, proj_u2, proj_v2, nullptr, pred_level2, mp_desc2, scale_factors1, nlevels1, th);
    if (rc != ORBX_OK) return rc;
    const int nq1 = (int)Q1.src.size(), nq2 = (int)Q2.src.size();
    if (nq1 == 0 || nq2 == 0) return ORBX_OK;       // a match needs both directions
    MHIPCHK(hipSetDevice(m->device));
    m->grid_ok = false; m->grid2_ok = false;
    rc = orbm_grow(m, 2 * std::max(nq1, nq2), std::max(n1, n2), 0);
    if (rc != ORBX_OK) return rc;
    rc = orbm_arena_begin(m);
    if (rc != ORBX_OK) return rc;
    hipStream_t s = m->stream;
    // both grids, then both searches, one synchronisation: key frame 1 -> slot `grid`, key frame 2 -> slot `grid2`
    rc = orbm_grid_build_into(m, m->grid, kps1, n1, grid1->assign_min_x, grid1->assign_min_y, grid1->inv_w, grid1->inv_h, grid1->query_min_x, grid1->query_min_y);
    if (rc != ORBX_OK) return rc;
    rc = orbm_grid_build_into(m, m->grid2, kps2, n2, grid2->assign_min_x, grid2->assign_min_y, grid2->inv_w, grid2->inv_h, grid2->query_min_x, grid2->query_min_y);
    if (rc != ORBX_OK) return rc;
    // the grid builder's keypoint staging in d_out is consumed by its kernel before the searches run (same stream), so d_out is free
    // again: the two searches share it
    InBlock in1(m), in2(m);
    int32_t *d_r1 = m->d_out, *d_r2 = m->d_out + 2 * (size_t)nq1;
    rc = launch_search_kf(m, m->grid2, Q1, desc2, n2, nullptr, nullptr, nlevels2, d_r1, in1, s);
    if (rc == ORBX_OK) rc = launch_search_kf(m, m->grid, Q2, desc1, n1, nullptr, nullptr, nlevels1, d_r2, in2, s);
    std::vector<int32_t> res((size_t)2 * (nq1 + nq2));
    if (rc == ORBX_OK) rc = orbm_d2h(m, res.data(), m->d_out, res.size() * 4, s);
    if (rc == ORBX_OK) rc = orbm_sync(m, s);
    in1.release(); in2.release();
    if (rc != ORBX_OK) return rc;
    m->grid_ok = true; m->grid2_ok = true;          // the handle's grid is key frame 1's now
    std::vector<int32_t> vnMatch1((size_t)n1, -1), vnMatch2((size_t)n2, -1);
    for (int k = 0; k < nq1; k++)
        if (res[k] >= 0 && res[(size_t)nq1 + k] <= ORBM_TH_HIGH) vnMatch1[Q1.src[k]] = res[k];                               // :1221
    const int32_t *r2 = res.data() + 2 * (size_t)nq1;
    for (int k = 0; k < nq2; k++)
        if (r2[k] >= 0 && r2[(size_t)nq2 + k] <= ORBM_TH_HIGH) vnMatch2[Q2.src[k]] = r2[k];                                  // :1301
    int nf = 0;
    for (int i1 = 0; i1 < n1; i1++) {               // Check agreement :1310-1323
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { match12[i1] = idx2; nf++; }
    }
    *nfound = nf;
    return ORBX_OK;
}

extern "C" int orbm_search_for_triangulation(orbm_matcher *m,
                                             const orbx_keypoint *kps1, const uint8_t *desc1, int n1, const uint8_t *has_mp1, const float *u_right1,
                                             const int32_t *fv1_node, const int32_t *fv1_off, const int32_t *fv1_idx, int fv1_n,
                                             const orbx_keypoint *kps2, const uint8_t *desc2, int n2, const uint8_t *has_mp2, const float *u_right2,
                                             const int32_t *fv2_node, const int32_t *fv2_off, const int32_t *fv2_idx, int fv2_n,
                                             const float *Cw, const float *T2w, float fx2, float fy2, float cx2, float cy2, const float *F12,
                                             const float *scale_factors2, const float *level_sigma2_2, int nlevels2, int only_stereo,
                                             int check_orientation, int32_t *matches12, int *nmatches)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n1 < 0 || n2 < 0 || fv1_n < 0 || fv2_n < 0 || !matches12 || !nmatches || !Cw || !T2w || !F12 || !scale_factors2 || !level_sigma2_2 ||
        nlevels2 < 1 || nlevels2 > ORBX_MAX_LEVELS)
        return mfail(ORBX_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                             // :678
    if (n1 == 0 || n2 == 0 || fv1_n == 0 || fv2_n == 0) return ORBX_OK;
    if (!kps1 || !desc1 || !has_mp1 || !u_right1 || !kps2 || !desc2 || !has_mp2 || !u_right2 || !fv1_node || !fv1_off || !fv1_idx || !fv2_node ||
        !fv2_off || !fv2_idx)
        return mfail(ORBX_E_INVALID, "NULL buffer");
    TriParams P;
    {   // epipole in the second image (:664-670)
        const float C2x = gemm3(T2w, 1, Cw, 1.0, T2w[3], 1.0), C2y = gemm3(T2w + 4, 1, Cw, 1.0, T2w[7], 1.0), C2z = gemm3(T2w + 8, 1, Cw, 1.0, T2w[11], 1.0);
        const float invz = 1.0f / C2z;
        P.ex = fx2 * C2x * invz + cx2; P.ey = fy2 * C2y * invz + cy2;
    }
    for (int k = 0; k < 9; k++) P.F12[k] = F12[k];
    for (int l = 0; l < ORBX_MAX_LEVELS; l++) {
        P.thr_epipole[l] = l < nlevels2 ? 100 * scale_factors2[l] : 0.f;
        P.thr_line[l] = l < nlevels2 ? 3.84 * level_sigma2_2[l] : 0.0;
    }
    // queries: the features of key frame 1 in shared nodes that pass :699-709, in visiting order
    std::vector<int4> qs;
    for (int a = 0, b = 0; a < fv1_n && b < fv2_n;) {
        if (fv1_node[a] == fv2_node[b]) {
            if (fv2_off[b + 1] > fv2_off[b])
                for (int c = fv1_off[a]; c < fv1_off[a + 1]; c++) {
                    const int idx1 = fv1_idx[c];
                    if (idx1 < 0 || idx1 >= n1) return mfail(ORBX_E_INVALID, "feature index %d outside [0,%d)", idx1, n1);
                    if (has_mp1[idx1]) continue;
                    const int st1 = u_right1[idx1] >= 0;
                    if (only_stereo && !st1) continue;
                    qs.push_back(make_int4(idx1, fv2_off[b], fv2_off[b + 1], st1));
                }
            a++; b++;
        } else if (fv1_node[a] < fv2_node[b]) a++;
        else b++;
    }
    const int nq = (int)qs.size(), ni2 = fv2_off[fv2_n];
    if (nq == 0) return ORBX_OK;
    std::vector<uint8_t> flags2((size_t)n2);
    for (int i = 0; i < n2; i++) {
        const int st2 = u_right2[i] >= 0;
        flags2[i] = (uint8_t)(((!has_mp2[i] && (!only_stereo || st2)) ? 1 : 0) | (st2 ? 2 : 0));      // :725-732
    }
    for (int c = 0; c < ni2; c++)
        if (fv2_idx[c] < 0 || fv2_idx[c] >= n2) return mfail(ORBX_E_INVALID, "feature index %d outside [0,%d)", fv2_idx[c], n2);
    std::vector<float> xy1((size_t)2 * n1), xy2((size_t)2 * n2);
    std::vector<int32_t> oct2((size_t)n2);
    for (int i = 0; i < n1; i++) { xy1[2 * (size_t)i] = kps1[i].x; xy1[2 * (size_t)i + 1] = kps1[i].y; }
    for (int i = 0; i < n2; i++) { xy2[2 * (size_t)i] = kps2[i].x; xy2[2 * (size_t)i + 1] = kps2[i].y; oct2[i] = kps2[i].octave; }
    MHIPCHK(hipSetDevice(m->device));
    int rc = orbm_grow(m, nq, 0, 0);
    if (rc != ORBX_OK) return rc;
    rc = orbm_arena_begin(m);
    if (rc != ORBX_OK) return rc;
    hipStream_t s = m->stream;
    InBlock in(m);
    const int pq = in.add(qs.data(), (size_t)nq * 16), pi = in.add(fv2_idx, (size_t)ni2 * 4), pd1 = in.add(desc1, (size_t)n1 * 32), pd2 = in.add(desc2, (size_t)n2 * 32);
    const int p1 = in.add(xy1.data(), (size_t)n1 * 8), p2 = in.add(xy2.data(), (size_t)n2 * 8), po = in.add(oct2.data(), (size_t)n2 * 4), pf = in.add(flags2.data(), (size_t)n2);
    rc = in.upload(s);
    if (rc != ORBX_OK) { in.release(); return rc; }
    hipLaunchKernelGGL(k_triangulation, dim3((nq + 3) / 4), dim3(M_THREADS), 0, s, in.at<int4>(pq), nq, in.at<int32_t>(pi), in.at<uint8_t>(pd1),
                       in.at<uint8_t>(pd2), in.at<float2>(p1), in.at<float2>(p2), in.at<int32_t>(po), in.at<uint8_t>(pf), P, m->d_out);
    MHIPCHK(hipGetLastError());
    std::vector<int32_t> res((size_t)nq);
    rc = orbm_d2h(m, res.data(), m->d_out, (size_t)nq * 4, s);
    if (rc == ORBX_OK) rc = orbm_sync(m, s);
    in.release();
    if (rc != ORBX_OK) return rc;
    // matches, rotation histogram and cull (:758-810) in visiting order
    int32_t hist[ORBM_HISTO_LENGTH] = {0};
    std::vector<std::pair<int, int>> rot;
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int nm = 0;
    for (int k = 0; k < nq; k++) {
        if (res[k] < 0) continue;
        const int idx1 = qs[k].x, idx2 = res[k];
        matches12[idx1] = idx2;
        nm++;
        if (check_orientation) {
            float r_ = kps1[idx1].angle - kps2[idx2].angle;
            if (r_ < 0.0) r_ += 360.0f;
            int bin = (int)roundf(r_ * factor);
            if (bin == ORBM_HISTO_LENGTH) bin = 0;
            if (bin < 0 || bin >= ORBM_HISTO_LENGTH) return mfail(ORBX_E_INVALID, "keypoint angle outside [0, 360)");     // the reference asserts
            rot.emplace_back(bin, idx1);
            hist[bin]++;
        }
    }
    if (check_orientation) {
        int32_t ind[3];
        orbm_three_maxima(hist, ORBM_HISTO_LENGTH, ind);
        for (const auto &e : rot)
            if (e.first != ind[0] && e.first != ind[1] && e.first != ind[2]) { matches12[e.second] = -1; nm--; }
    }
    *nmatches = nm;
    return ORBX_OK;
}
