// orbm_grid.hip -- the 64x48 Frame grid and windowed searches on gfx950.
// Reference (WChen09/My-SLAM): src/Frame.cc:230-245 AssignFeaturesToGrid, :382-392 PosInGrid,
// :327-380 GetFeaturesInArea; consumers: ORBmatcher::SearchByProjection (src/ORBmatcher.cc:1397-1430),
// SearchForInitialization (:425-457).  The window maths is fp32 exactly as written there (no
// contraction: __f*_rn); candidate ORDER is the reference's (cell column, cell row, push_back order),
// because "first candidate wins a tie" depends on it.
#include <algorithm>
#include <vector>

#include <climits>
#include <utility>
#include "orbm_internal.h"
#include "orbx_internal.h"

// -------------------------------------------------------------------------------------------------
// k_grid_build: one 1024-thread workgroup.  Count per cell (LDS atomics) -> scan -> scatter -> each
// cell's short list is sorted ascending, which restores push_back order (keypoint index order).
// -------------------------------------------------------------------------------------------------
#define G_THREADS 1024
__global__ __launch_bounds__(G_THREADS) void k_grid_build(OrbmGrid g, const orbx_keypoint *__restrict__ kps)
{
    __shared__ int cnt[ORBM_GRID_CELLS];
    __shared__ int start[ORBM_GRID_CELLS + 1];
    __shared__ int wsum[G_THREADS / 64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < ORBM_GRID_CELLS; c += G_THREADS) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < g.n; i += G_THREADS) {
        const orbx_keypoint kp = kps[i];
        g.kx[i] = kp.x; g.ky[i] = kp.y; g.koct[i] = kp.octave;
        const int px = (int)roundf(__fmul_rn(__fsub_rn(kp.x, g.min_x), g.inv_w));   // PosInGrid :384
        const int py = (int)roundf(__fmul_rn(__fsub_rn(kp.y, g.min_y), g.inv_h));   // :385
        int cell = -1;
        if (px >= 0 && px < ORBM_GRID_COLS && py >= 0 && py < ORBM_GRID_ROWS) {
            cell = px * ORBM_GRID_ROWS + py;
            atomicAdd(&cnt[cell], 1);
        }
        g.cell_of[i] = cell;
    }
    __syncthreads();
    // exclusive scan of 3072 counts: 3 per thread
    const int c0 = tid * 3;
    const int a = cnt[c0], b = cnt[c0 + 1], c = cnt[c0 + 2];
    int inc = a + b + c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int w = lane < G_THREADS / 64 ? wsum[lane] : 0, wi = w;
#pragma unroll
        for (int o = 1; o < G_THREADS / 64; o <<= 1) {
            const int t = __shfl_up(wi, o);
            if (lane >= o) wi += t;
        }
        if (lane < G_THREADS / 64) wsum[lane] = wi - w;
        if (lane == G_THREADS / 64 - 1) wsum[G_THREADS / 64] = wi;
    }
    __syncthreads();
    const int ex = wsum[wave] + inc - (a + b + c);
    start[c0] = ex; start[c0 + 1] = ex + a; start[c0 + 2] = ex + a + b;
    if (tid == 0) start[ORBM_GRID_CELLS] = wsum[G_THREADS / 64];
    cnt[c0] = 0; cnt[c0 + 1] = 0; cnt[c0 + 2] = 0;
    __syncthreads();
    for (int i = tid; i <= ORBM_GRID_CELLS; i += G_THREADS) g.cell_start[i] = start[i];
    for (int i = tid; i < g.n; i += G_THREADS) {
        const int cell = g.cell_of[i];
        if (cell >= 0) g.items[start[cell] + atomicAdd(&cnt[cell], 1)] = i;
    }
    __syncthreads();
    __threadfence_block();
    for (int cl = tid; cl < ORBM_GRID_CELLS; cl += G_THREADS) {     // insertion sort of a short list
        const int s = start[cl], e = start[cl + 1];
        for (int i = s + 1; i < e; i++) {
            const int v = g.items[i];
            int j = i - 1;
            while (j >= s && g.items[j] > v) { g.items[j + 1] = g.items[j]; j--; }
            g.items[j + 1] = v;
        }
    }
}

// one wave per window; MODE 0 = count, 1 = write the list at off[q] in the reference's order
template <int MODE>
__global__ __launch_bounds__(M_THREADS) void k_area_list(OrbmGrid g, const float *__restrict__ qx, const float *__restrict__ qy,
                                                        const float *__restrict__ qr, const int32_t *__restrict__ minl,
                                                        const int32_t *__restrict__ maxl, int nq,
                                                        int32_t *__restrict__ counts, const int32_t *__restrict__ off,
                                                        int32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (M_THREADS / 64) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float x = qx[q], y = qy[q], r = qr[q];
    const int mn = minl[q], mx = maxl[q];
    int n = 0, cx0, cx1, cy0, cy1;
    if (window_cells(g, x, y, r, cx0, cx1, cy0, cy1)) {
        const int base = MODE == 1 ? off[q] : 0;
        for (int ix = cx0; ix <= cx1; ix++) {              // a cell column is one contiguous item range
            const int s = g.cell_start[ix * ORBM_GRID_ROWS + cy0], e = g.cell_start[ix * ORBM_GRID_ROWS + cy1 + 1];
            for (int j0 = s; j0 < e; j0 += 64) {
                const int j = j0 + lane;
                int i = -1;
                bool ok = false;
                if (j < e) { i = g.items[j]; ok = in_window(g, i, x, y, r, mn, mx); }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
                if (MODE == 1 && ok) out[orbx_prefix_cnt(m, base + n)] = i;
                n += __popcll(m);
            }
        }
    }
    if (MODE == 0 && lane == 0) counts[q] = n;
}

// fused window query + best / second-best (strict '<': first candidate wins ties, a tie with the best
// becomes the second best).  Key = distance << 22 | position in the reference's candidate order.
__global__ __launch_bounds__(M_THREADS) void k_search_area(OrbmGrid g, const uint8_t *__restrict__ qdesc,
                                                          const float *__restrict__ qx, const float *__restrict__ qy,
                                                          const float *__restrict__ qr, const int32_t *__restrict__ minl,
                                                          const int32_t *__restrict__ maxl, int nq,
                                                          const uint8_t *__restrict__ tdesc, const uint8_t *__restrict__ skip,
                                                          int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d,
                                                          int32_t *__restrict__ second_d)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (M_THREADS / 64) + (threadIdx.x >> 6);
    if (q >= nq) return;
    const float x = qx[q], y = qy[q], r = qr[q];
    const int mn = minl[q], mx = maxl[q];
    const uint4 *Q = reinterpret_cast<const uint4 *>(qdesc) + 2 * (long long)q;
    const uint4 q0 = Q[0], q1 = Q[1];
    uint32_t bp = (256u << 22) | 0x3FFFFFu;
    int s2 = 256, bidx = -1, n = 0;
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
                    ok = in_window(g, i, x, y, r, mn, mx) && !(skip && skip[i]);
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
                if (ok) {
                    const uint4 *Tj = reinterpret_cast<const uint4 *>(tdesc) + 2 * (long long)i;
                    const int d = hamming256(q0, q1, Tj[0], Tj[1]);
                    const uint32_t p = ((uint32_t)d << 22) | (uint32_t)min(orbx_prefix_cnt(m, n), 0x3FFFFF);
                    if (p < bp) { s2 = (int)(bp >> 22); bp = p; bidx = i; }
                    else if (d < s2) s2 = d;
                }
                n += __popcll(m);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t op = __shfl_xor(bp, o);
        const int os = __shfl_xor(s2, o);
        const int oi = __shfl_xor(bidx, o);
        const int loser = (int)(max(bp, op) >> 22);
        if (op < bp) bidx = oi;
        bp = min(bp, op);
        s2 = min(min(s2, os), loser);
    }
    if (lane == 0) {
        const int d = (int)(bp >> 22);
        best_d[q] = d;
        second_d[q] = s2;
        best_idx[q] = d < 256 ? bidx : -1;
    }
}

// -------------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------------
static int ensure_grid(orbm_matcher *m, OrbmGrid &g)
{
    if (g.cell_start) return ORBX_OK;
    const size_t n = (size_t)m->max_t;
    MHIPCHK(hipMalloc((void **)&g.kx, n * 4));
    MHIPCHK(hipMalloc((void **)&g.ky, n * 4));
    MHIPCHK(hipMalloc((void **)&g.koct, n * 4));
    MHIPCHK(hipMalloc((void **)&g.items, n * 4));
    MHIPCHK(hipMalloc((void **)&g.cell_of, n * 4));
    MHIPCHK(hipMalloc((void **)&g.cell_start, (ORBM_GRID_CELLS + 1) * 4));
    return ORBX_OK;
}

static int ensure_query_staging(orbm_matcher *m, size_t nq)
{
    if (nq <= m->qf_elems) return ORBX_OK;
    MHIPCHK(hipDeviceSynchronize());
    (void)hipFree(m->d_qf); (void)hipFree(m->d_qi);
    m->d_qf = nullptr; m->d_qi = nullptr; m->qf_elems = 0;
    MHIPCHK(hipMalloc((void **)&m->d_qf, nq * 3 * sizeof(float)));
    MHIPCHK(hipMalloc((void **)&m->d_qi, nq * 4 * sizeof(int32_t)));    // min_level, max_level, counts, offsets
    m->qf_elems = nq;
    return ORBX_OK;
}

// cv::undistortPoints(src, dst, K, distCoeffs, noArray(), K) of OpenCV 3.1.0 for one point (cvUndistortPoints: camera matrix
// and coefficients converted to double, ITERS = 5, no tilt, R = I, P = K).  Called by Frame::UndistortKeyPoints (src/Frame.cc:421)
// and Frame::ComputeImageBounds (:449).
static inline void undistort_point(float xin, float yin, double fx, double fy, double cx, double cy, const double k[5],
                                   float *xout, float *yout)
{
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = xin, y = yin;
    const double x0 = x = (x - cx) * ifx;
    const double y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);     // k4..k6 = 0
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
    *xout = (float)(xx * ww);
    *yout = (float)(yy * ww);
}

extern "C" int orbm_undistort_keypoints(const orbx_keypoint *kps, int n, float fx, float fy, float cx, float cy,
                                        const float *dist, int ndist, orbx_keypoint *kps_un)
{
    if (n < 0 || (n > 0 && (!kps || !kps_un)) || !dist || (ndist != 4 && ndist != 5)) return mfail(ORBX_E_INVALID, "bad argument");
    if (dist[0] == 0.0f) {                     // :406-410
        if (kps_un != kps) memmove(kps_un, kps, sizeof(orbx_keypoint) * (size_t)n);
        return ORBX_OK;
    }
    const double k[5] = {dist[0], dist[1], dist[2], dist[3], ndist == 5 ? dist[4] : 0.0};
    for (int i = 0; i < n; i++) {
        orbx_keypoint kp = kps[i];
        undistort_point(kps[i].x, kps[i].y, fx, fy, cx, cy, k, &kp.x, &kp.y);
        kps_un[i] = kp;
    }
    return ORBX_OK;
}

extern "C" int orbm_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float *dist, int ndist,
                                 float bounds[4])
{
    if (!dist || !bounds || (ndist != 4 && ndist != 5)) return mfail(ORBX_E_INVALID, "bad argument");
    if (dist[0] == 0.0f) { bounds[0] = 0.0f; bounds[1] = (float)width; bounds[2] = 0.0f; bounds[3] = (float)height; return ORBX_OK; }
    const double k[5] = {dist[0], dist[1], dist[2], dist[3], ndist == 5 ? dist[4] : 0.0};
    const float cxs[4] = {0.f, (float)width, 0.f, (float)width}, cys[4] = {0.f, 0.f, (float)height, (float)height};
    float ux[4], uy[4];
    for (int i = 0; i < 4; i++) undistort_point(cxs[i], cys[i], fx, fy, cx, cy, k, &ux[i], &uy[i]);
    bounds[0] = std::min(ux[0], ux[2]); bounds[1] = std::max(ux[1], ux[3]);      // :451-454
    bounds[2] = std::min(uy[0], uy[1]); bounds[3] = std::max(uy[2], uy[3]);
    return ORBX_OK;
}

// Builds a grid slot.  The keypoints are staged in the pinned arena of the current call (the caller has run orbm_arena_begin) and
// go to the handle's d_out (a temporary block when that is too small); the kernel is queued on the handle's stream.
int orbm_grid_build_into(orbm_matcher *m, OrbmGrid &g, const orbx_keypoint *kps_un, int n, float assign_min_x, float assign_min_y,
                         float inv_w, float inv_h, float query_min_x, float query_min_y)
{
    int rc = ensure_grid(m, g);
    if (rc != ORBX_OK) return rc;
    g.min_x = assign_min_x; g.min_y = assign_min_y; g.inv_w = inv_w; g.inv_h = inv_h; g.qmin_x = query_min_x; g.qmin_y = query_min_y;
    g.n = n;
    hipStream_t s = m->stream;
    orbx_keypoint *d_kps = reinterpret_cast<orbx_keypoint *>(m->d_out);
    const size_t need = (size_t)n * sizeof(orbx_keypoint);
    const size_t have = std::max<size_t>((size_t)3 * m->max_q, (size_t)m->max_pairs) * 4;
    void *tmp = nullptr;
    if (need > have) { MHIPCHK(hipMalloc(&tmp, need)); d_kps = reinterpret_cast<orbx_keypoint *>(tmp); }
    if (n > 0) { int rc_ = orbm_h2d(m, d_kps, kps_un, need, s); if (rc_ != ORBX_OK) return rc_; }
    hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(G_THREADS), 0, s, g, d_kps);
    MHIPCHK(hipGetLastError());
    if (tmp) { MHIPCHK(hipStreamSynchronize(s)); (void)hipFree(tmp); }
    return ORBX_OK;
}

extern "C" int orbm_grid_build(orbm_matcher *m, const orbx_keypoint *kps_un, int n,
                               float min_x, float max_x, float min_y, float max_y)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n < 0) return mfail(ORBX_E_INVALID, "n=%d keypoints", n);
    if (n > 0 && !kps_un) return mfail(ORBX_E_INVALID, "NULL keypoints");
    if (!(max_x > min_x) || !(max_y > min_y)) return mfail(ORBX_E_INVALID, "empty image bounds");
    MHIPCHK(hipSetDevice(m->device));
    m->grid_ok = false;
    { int rc_ = orbm_grow(m, 0, n, 0); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_arena_begin(m); if (rc_ != ORBX_OK) return rc_; }
    int rc = orbm_grid_build_into(m, m->grid, kps_un, n, min_x, min_y, (float)ORBM_GRID_COLS / (max_x - min_x),     // src/Frame.cc:212
                                  (float)ORBM_GRID_ROWS / (max_y - min_y), min_x, min_y);                        // :213
    if (rc != ORBX_OK) return rc;
    { int rc_ = orbm_sync(m, m->stream); if (rc_ != ORBX_OK) return rc_; }
    m->grid_ok = true;
    return ORBX_OK;
}

// number of keypoints in the handle's grid, -1 when there is none (never built, or dropped by a workspace growth)
extern "C" int orbm_grid_count(const orbm_matcher *m) { return (m && m->grid_ok) ? m->grid.n : -1; }

// A key frame's grid.  KeyFrame copies mGrid from the Frame it was made of (src/KeyFrame.cc:48-54), so the cells were filled by
// Frame::PosInGrid with Frame's float mnMinX / mnMinY and mfGridElementWidthInv / HeightInv (src/Frame.cc:382-392: assign_*, inv_*),
// while KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:569-606) subtracts the key frame's own mnMinX / mnMinY, which are ints
// (include/KeyFrame.h:190-193: query_*).  With the shipped calibration (k1 == 0) the two origins coincide.
extern "C" int orbm_grid_build_kf(orbm_matcher *m, const orbx_keypoint *kps_un, int n, float assign_min_x, float assign_min_y,
                                  float inv_w, float inv_h, float query_min_x, float query_min_y)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n < 0) return mfail(ORBX_E_INVALID, "n=%d keypoints", n);
    if (n > 0 && !kps_un) return mfail(ORBX_E_INVALID, "NULL keypoints");
    if (!(inv_w > 0.f) || !(inv_h > 0.f)) return mfail(ORBX_E_INVALID, "grid cell sizes must be positive");
    MHIPCHK(hipSetDevice(m->device));
    m->grid_ok = false;
    { int rc_ = orbm_grow(m, 0, n, 0); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_arena_begin(m); if (rc_ != ORBX_OK) return rc_; }
    int rc = orbm_grid_build_into(m, m->grid, kps_un, n, assign_min_x, assign_min_y, inv_w, inv_h, query_min_x, query_min_y);
    if (rc != ORBX_OK) return rc;
    { int rc_ = orbm_sync(m, m->stream); if (rc_ != ORBX_OK) return rc_; }
    m->grid_ok = true;
    return ORBX_OK;
}

static int upload_windows(orbm_matcher *m, const float *x, const float *y, const float *r, const int32_t *mn,
                          const int32_t *mx, int nq, hipStream_t s)
{
    int rc = ensure_query_staging(m, (size_t)nq);
    if (rc != ORBX_OK) return rc;
    { int rc_ = orbm_h2d(m, m->d_qf, x, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_h2d(m, m->d_qf + m->qf_elems, y, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_h2d(m, m->d_qf + 2 * m->qf_elems, r, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_h2d(m, m->d_qi, mn, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_h2d(m, m->d_qi + m->qf_elems, mx, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    return ORBX_OK;
}

extern "C" int orbm_features_in_area(orbm_matcher *m, const float *x, const float *y, const float *r,
                                     const int32_t *min_level, const int32_t *max_level, int nq,
                                     int32_t *cand_off, int32_t *cand_idx, int cap_idx)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (!m->grid_ok) return mfail(ORBX_E_INVALID, "orbm_grid_build has not been called");
    if (nq < 0 || !cand_off) return mfail(ORBX_E_INVALID, "bad argument");
    cand_off[0] = 0;
    if (nq == 0) return 0;
    if (!x || !y || !r || !min_level || !max_level) return mfail(ORBX_E_INVALID, "NULL window array");
    MHIPCHK(hipSetDevice(m->device));
    { int rc_ = orbm_arena_begin(m); if (rc_ != ORBX_OK) return rc_; }
    hipStream_t s = m->stream;
    int rc = upload_windows(m, x, y, r, min_level, max_level, nq, s);
    if (rc != ORBX_OK) return rc;
    const size_t Q = m->qf_elems;
    int32_t *d_cnt = m->d_qi + 2 * Q, *d_off = m->d_qi + 3 * Q;
    const dim3 grid((nq + 3) / 4);
    hipLaunchKernelGGL(k_area_list<0>, grid, dim3(M_THREADS), 0, s, m->grid, m->d_qf, m->d_qf + Q, m->d_qf + 2 * Q, m->d_qi, m->d_qi + Q,
                       nq, d_cnt, (const int32_t *)nullptr, (int32_t *)nullptr);
    MHIPCHK(hipGetLastError());
    std::vector<int32_t> cnt(nq);
    { int rc_ = orbm_d2h(m, cnt.data(), d_cnt, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_sync(m, s); if (rc_ != ORBX_OK) return rc_; }
    for (int i = 0; i < nq; i++) cand_off[i + 1] = cand_off[i] + cnt[i];
    const int total = cand_off[nq];
    if (total > cap_idx) return mfail(ORBX_E_CAPACITY, "%d candidates, caller capacity %d", total, cap_idx);
    if (total == 0) return 0;
    if (!cand_idx) return mfail(ORBX_E_INVALID, "cand_idx is NULL");
    if (total > m->max_pairs) { int rc_ = orbm_grow(m, 0, 0, total); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_h2d(m, d_off, cand_off, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    hipLaunchKernelGGL(k_area_list<1>, grid, dim3(M_THREADS), 0, s, m->grid, m->d_qf, m->d_qf + Q, m->d_qf + 2 * Q, m->d_qi, m->d_qi + Q,
                       nq, (int32_t *)nullptr, d_off, m->d_idx);
    MHIPCHK(hipGetLastError());
    { int rc_ = orbm_d2h(m, cand_idx, m->d_idx, (size_t)total * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_sync(m, s); if (rc_ != ORBX_OK) return rc_; }
    return total;
}

extern "C" int orbm_search_area_best2_device(orbm_matcher *m, const uint8_t *d_qdesc, const float *d_x, const float *d_y,
                                             const float *d_r, const int32_t *d_min_level, const int32_t *d_max_level, int nq,
                                             const uint8_t *d_train_desc, const uint8_t *d_skip,
                                             int32_t *d_best_idx, int32_t *d_best_d, int32_t *d_second_d, void *hip_stream)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (!m->grid_ok) return mfail(ORBX_E_INVALID, "orbm_grid_build has not been called");
    if (nq <= 0) return nq == 0 ? ORBX_OK : mfail(ORBX_E_INVALID, "nq < 0");
    if (!d_qdesc || !d_x || !d_y || !d_r || !d_min_level || !d_max_level || !d_train_desc || !d_best_idx || !d_best_d || !d_second_d)
        return mfail(ORBX_E_INVALID, "NULL device pointer");
    MHIPCHK(hipSetDevice(m->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : m->stream;
    hipLaunchKernelGGL(k_search_area, dim3((nq + 3) / 4), dim3(M_THREADS), 0, s, m->grid, d_qdesc, d_x, d_y, d_r, d_min_level,
                       d_max_level, nq, d_train_desc, d_skip, d_best_idx, d_best_d, d_second_d);
    MHIPCHK(hipGetLastError());
    return ORBX_OK;
}

extern "C" int orbm_search_area_best2(orbm_matcher *m, const uint8_t *qdesc, const float *x, const float *y, const float *r,
                                      const int32_t *min_level, const int32_t *max_level, int nq,
                                      const uint8_t *train_desc, const uint8_t *skip,
                                      int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (!m->grid_ok) return mfail(ORBX_E_INVALID, "orbm_grid_build has not been called");
    if (nq < 0) return mfail(ORBX_E_INVALID, "nq=%d", nq);
    if (nq == 0) return ORBX_OK;
    { int rc_ = orbm_grow(m, nq, 0, 0); if (rc_ != ORBX_OK) return rc_; }
    if (!qdesc || !x || !y || !r || !min_level || !max_level || !best_idx || !best_d || !second_d || (m->grid.n > 0 && !train_desc))
        return mfail(ORBX_E_INVALID, "NULL buffer");
    MHIPCHK(hipSetDevice(m->device));
    { int rc_ = orbm_arena_begin(m); if (rc_ != ORBX_OK) return rc_; }
    hipStream_t s = m->stream;
    int rc;
    int32_t *o_bi = m->d_out, *o_bd = m->d_out + nq, *o_sd = m->d_out + 2 * (size_t)nq;
    // usual path: every input of the call is staged in the pinned arena and goes up in one copy to its device mirror
    const size_t mark = m->arena_used;
    const size_t nb4 = (size_t)nq * 4;
    const float *sx = (const float *)orbm_stage_in(m, x, nb4), *sy = (const float *)orbm_stage_in(m, y, nb4), *sr = (const float *)orbm_stage_in(m, r, nb4);
    const int32_t *smn = (const int32_t *)orbm_stage_in(m, min_level, nb4), *smx = (const int32_t *)orbm_stage_in(m, max_level, nb4);
    const uint8_t *sq = (const uint8_t *)orbm_stage_in(m, qdesc, (size_t)nq * 32);
    const uint8_t *st = m->grid.n > 0 ? (const uint8_t *)orbm_stage_in(m, train_desc, (size_t)m->grid.n * 32) : m->d_t;
    const uint8_t *ss = (skip && m->grid.n > 0) ? (const uint8_t *)orbm_stage_in(m, skip, (size_t)m->grid.n) : nullptr;
    if (sx && sy && sr && smn && smx && sq && st && (ss || !(skip && m->grid.n > 0))) {
        rc = orbm_flush_in(m, mark, s);
        if (rc != ORBX_OK) return rc;
        rc = orbm_search_area_best2_device(m, sq, sx, sy, sr, smn, smx, nq, st, ss, o_bi, o_bd, o_sd, s);
        if (rc != ORBX_OK) return rc;
    } else {     // first call / arena too small: per-array copies into the fixed buffers (the arena grows for the next call)
        rc = upload_windows(m, x, y, r, min_level, max_level, nq, s);
        if (rc != ORBX_OK) return rc;
        const size_t Q = m->qf_elems;
        { int rc_ = orbm_h2d(m, m->d_q, qdesc, (size_t)nq * 32, s); if (rc_ != ORBX_OK) return rc_; }
        if (m->grid.n > 0) { int rc_ = orbm_h2d(m, m->d_t, train_desc, (size_t)m->grid.n * 32, s); if (rc_ != ORBX_OK) return rc_; }
        const uint8_t *d_skip = nullptr;
        if (skip && m->grid.n > 0) {
            if (!m->d_skip) MHIPCHK(hipMalloc((void **)&m->d_skip, (size_t)m->max_t));
            { int rc_ = orbm_h2d(m, m->d_skip, skip, (size_t)m->grid.n, s); if (rc_ != ORBX_OK) return rc_; }
            d_skip = m->d_skip;
        }
        rc = orbm_search_area_best2_device(m, m->d_q, m->d_qf, m->d_qf + Q, m->d_qf + 2 * Q, m->d_qi, m->d_qi + Q, nq, m->d_t, d_skip,
                                           o_bi, o_bd, o_sd, s);
        if (rc != ORBX_OK) return rc;
    }
    {
        void *const hosts[3] = {best_idx, best_d, second_d};
        const size_t parts[3] = {nb4, nb4, nb4};
        int rc_ = orbm_d2h_split(m, hosts, parts, 3, o_bi, s);
        if (rc_ != ORBX_OK) return rc_;
    }
    { int rc_ = orbm_sync(m, s); if (rc_ != ORBX_OK) return rc_; }
    return ORBX_OK;
}

// GetFeaturesInArea for nq windows AND the distance of every candidate to its window's descriptor, for the matchers whose scan
// is sequential on the host (SearchForInitialization, SearchByProjection(Frame, Frame)): inputs go up in one copy, counts come
// back, offsets go up, lists and distances come back -- two synchronisations.  (One pass into fixed per-window slots was
// measured: the fullest windows need > 128 slots, and copying nq x slots back costs more than the second round trip.)
// Falls back to the two public calls when the arena has no room yet (first call).
int orbm_area_pairs(orbm_matcher *m, const float *x, const float *y, const float *r, const int32_t *mn, const int32_t *mx, int nq,
                      const uint8_t *qdesc, const uint8_t *train_desc, int n_train,
                      std::vector<int32_t> &off, std::vector<int32_t> &idx, std::vector<int32_t> &dist)
{
    off.assign((size_t)nq + 1, 0);
    MHIPCHK(hipSetDevice(m->device));
    { int rc_ = orbm_arena_begin(m); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = ensure_query_staging(m, (size_t)nq); if (rc_ != ORBX_OK) return rc_; }
    hipStream_t s = m->stream;
    const size_t nb4 = (size_t)nq * 4, mark = m->arena_used;
    const float *sx = (const float *)orbm_stage_in(m, x, nb4), *sy = (const float *)orbm_stage_in(m, y, nb4), *sr = (const float *)orbm_stage_in(m, r, nb4);
    const int32_t *smn = (const int32_t *)orbm_stage_in(m, mn, nb4), *smx = (const int32_t *)orbm_stage_in(m, mx, nb4);
    const uint8_t *sq = (const uint8_t *)orbm_stage_in(m, qdesc, (size_t)nq * 32), *st = (const uint8_t *)orbm_stage_in(m, train_desc, (size_t)n_train * 32);
    int32_t *soff = (int32_t *)orbm_stage_in(m, off.data(), ((size_t)nq + 1) * 4);      // place holder, filled after the counts are known
    if (!(sx && sy && sr && smn && smx && sq && st && soff)) {
        idx.assign((size_t)std::max<long long>(std::min<long long>(m->max_pairs, (long long)nq * n_train), 1), 0);
        int total = orbm_features_in_area(m, x, y, r, mn, mx, nq, off.data(), idx.data(), (int)idx.size());
        if (total == ORBX_E_CAPACITY) {       // more candidates than the handle was created for: every window can hold at most n_train
            idx.assign((size_t)std::max<long long>(std::min<long long>((long long)nq * n_train, INT_MAX), 1), 0);
            total = orbm_features_in_area(m, x, y, r, mn, mx, nq, off.data(), idx.data(), (int)idx.size());
        }
        if (total < 0) return total;
        idx.resize((size_t)std::max(total, 1));
        dist.assign((size_t)std::max(total, 1), 0);
        if (total > 0) { int rc = orbm_distances(m, qdesc, nq, train_desc, n_train, off.data(), idx.data(), dist.data()); if (rc != ORBX_OK) return rc; }
        return total;
    }
    { int rc_ = orbm_flush_in(m, mark, s); if (rc_ != ORBX_OK) return rc_; }
    const size_t Q = m->qf_elems;
    int32_t *d_cnt = m->d_qi + 2 * Q;
    const dim3 grid((nq + 3) / 4);
    hipLaunchKernelGGL(k_area_list<0>, grid, dim3(M_THREADS), 0, s, m->grid, sx, sy, sr, smn, smx, nq, d_cnt, (const int32_t *)nullptr, (int32_t *)nullptr);
    MHIPCHK(hipGetLastError());
    const int32_t *cnt = (const int32_t *)orbm_d2h_tmp(m, d_cnt, nb4, s);
    std::vector<int32_t> cnt_plain;
    if (!cnt) {           // no room left in the arena this call (it grows for the next one): a plain copy
        cnt_plain.resize((size_t)nq);
        MHIPCHK(hipMemcpyAsync(cnt_plain.data(), d_cnt, nb4, hipMemcpyDeviceToHost, s));
        cnt = cnt_plain.data();
    }
    { int rc_ = orbm_sync(m, s); if (rc_ != ORBX_OK) return rc_; }
    for (int i = 0; i < nq; i++) off[i + 1] = off[i] + cnt[i];
    const int total = off[nq];
    idx.assign((size_t)std::max(total, 1), 0); dist.assign((size_t)std::max(total, 1), 0);
    if (total == 0) return 0;
    if (total > m->max_pairs) { int rc_ = orbm_grow(m, 0, 0, total); if (rc_ != ORBX_OK) return rc_; }     // d_idx / d_out hold nothing yet
    uint8_t *h_off = m->arena + ((uint8_t *)soff - m->d_arena);
    memcpy(h_off, off.data(), ((size_t)nq + 1) * 4);
    MHIPCHK(hipMemcpyAsync(soff, h_off, ((size_t)nq + 1) * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_area_list<1>, grid, dim3(M_THREADS), 0, s, m->grid, sx, sy, sr, smn, smx, nq, (int32_t *)nullptr, (const int32_t *)soff, m->d_idx);
    MHIPCHK(hipGetLastError());
    orbm_launch_dist_csr(sq, nq, st, soff, m->d_idx, total, m->d_out, s);
    MHIPCHK(hipGetLastError());
    const void *hi = orbm_d2h_tmp(m, m->d_idx, (size_t)total * 4, s), *hd = orbm_d2h_tmp(m, m->d_out, (size_t)total * 4, s);
    if (!hi || !hd) {     // no room for the lists this call (the arena grows for the next one): plain copies
        MHIPCHK(hipMemcpyAsync(idx.data(), m->d_idx, (size_t)total * 4, hipMemcpyDeviceToHost, s));
        MHIPCHK(hipMemcpyAsync(dist.data(), m->d_out, (size_t)total * 4, hipMemcpyDeviceToHost, s));
        MHIPCHK(hipStreamSynchronize(s));
        return total;
    }
    { int rc_ = orbm_sync(m, s); if (rc_ != ORBX_OK) return rc_; }
    memcpy(idx.data(), hi, (size_t)total * 4);
    memcpy(dist.data(), hd, (size_t)total * 4);
    return total;
}

// ---- ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520) ----
extern "C" int orbm_search_for_initialization(orbm_matcher *m, const orbx_keypoint *kps1, const uint8_t *desc1, int n1,
                                              const orbx_keypoint *kps2, const uint8_t *desc2, int n2,
                                              float *prev_matched, int window_size, float nnratio, int check_orientation,
                                              int32_t *matches12, int *nmatches)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n1 < 0 || n2 < 0 || !matches12 || !nmatches || (n1 > 0 && (!kps1 || !desc1 || !prev_matched)) || (n2 > 0 && (!kps2 || !desc2)))
        return mfail(ORBX_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                       // :408
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    if (!m->grid_ok || m->grid.n != n2) return mfail(ORBX_E_INVALID, "orbm_grid_build(frame 2) has not been called (grid holds %d keypoints, n2 = %d)", m->grid_ok ? m->grid.n : -1, n2);
    // the queries: keypoints of frame 1 on level 0 (:421-423), their windows and descriptors
    std::vector<int> qi;
    for (int i = 0; i < n1; i++)
        if (kps1[i].octave <= 0) qi.push_back(i);
    const int nq = (int)qi.size();
    if (nq == 0) return ORBX_OK;
    { int rc_ = orbm_grow(m, nq, 0, 0); if (rc_ != ORBX_OK) return rc_; }
    std::vector<float> x(nq), y(nq), r(nq, (float)window_size);
    std::vector<int32_t> lv(nq), off, idx, dist;
    std::vector<uint8_t> qd((size_t)nq * 32);
    for (int k = 0; k < nq; k++) {
        const int i = qi[k];
        x[k] = prev_matched[2 * i]; y[k] = prev_matched[2 * i + 1]; lv[k] = kps1[i].octave;
        memcpy(&qd[(size_t)k * 32], desc1 + (size_t)i * 32, 32);
    }
    const int total = orbm_area_pairs(m, x.data(), y.data(), r.data(), lv.data(), lv.data(), nq, qd.data(), desc2, n2, off, idx, dist);
    if (total < 0) return total;
    // the sequential scan (:418-487)
    std::vector<int> matched_dist((size_t)n2, INT_MAX), matches21((size_t)n2, -1);
    std::vector<std::pair<int, int>> rot;           // rotHist as (bin, i1) in push order
    int32_t hist[ORBM_HISTO_LENGTH] = {0};
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int nm = 0;
    for (int k = 0; k < nq; k++) {
        const int i1 = qi[k];
        if (off[k + 1] == off[k]) continue;          // :427
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = off[k]; c < off[k + 1]; c++) {
            const int i2 = idx[c], d = dist[c];
            if (matched_dist[i2] <= d) continue;     // :444
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
            else if (d < bestDist2) bestDist2 = d;
        }
        if (bestDist <= ORBM_TH_LOW && (float)bestDist < (float)bestDist2 * nnratio) {     // :459-461
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nm--; }
            matches12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matched_dist[bestIdx2] = bestDist;
            nm++;
            if (check_orientation) {
                float rot_ = kps1[i1].angle - kps2[bestIdx2].angle;
                if (rot_ < 0.0) rot_ += 360.0f;
                int bin = (int)roundf(rot_ * factor);
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
                if (bin < 0 || bin >= ORBM_HISTO_LENGTH) return mfail(ORBX_E_INVALID, "keypoint angle outside [0, 360)");   // the reference asserts
                rot.emplace_back(bin, i1);
                hist[bin]++;
            }
        }
    }
    if (check_orientation) {                         // :489-510
        int32_t ind[3];
        orbm_three_maxima(hist, ORBM_HISTO_LENGTH, ind);
        for (const auto &e : rot) {
            if (e.first == ind[0] || e.first == ind[1] || e.first == ind[2]) continue;
            if (matches12[e.second] >= 0) { matches12[e.second] = -1; nm--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)                  // :513-516
        if (matches12[i1] >= 0) { prev_matched[2 * i1] = kps2[matches12[i1]].x; prev_matched[2 * i1 + 1] = kps2[matches12[i1]].y; }
    *nmatches = nm;
    return ORBX_OK;
}

// ---- ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) (src/ORBmatcher.cc:1328-1470) ----
// (R x + t)[row] as OpenCV 3.1.0 evaluates `R*x + t` on a 3x3 and a 3x1 float matrix: one cv::gemm(R, x, 1, t, 1, dst, 0) call, whose
// small-matrix path (modules/core/src/matmul.cpp: flags == 0, 2 <= len <= 4) sums the three products in float, left to right, and
// finishes with (float)(t0*alpha + c*beta) in double.  `-R.t()*t` materialises the transpose and takes the same path with alpha = -1.
static inline float gemm_row(const float *T, int row, const float *x)
{
    const float t0 = T[4 * row] * x[0] + T[4 * row + 1] * x[1] + T[4 * row + 2] * x[2];
    return (float)((double)t0 * 1.0 + (double)T[4 * row + 3] * 1.0);
}
static inline void camera_center(const float *T, float Ow[3])
{
    for (int k = 0; k < 3; k++) {
        const float t0 = T[k] * T[3] + T[4 + k] * T[7] + T[8 + k] * T[11];
        Ow[k] = (float)((double)t0 * -1.0 + 0.0 * 0.0);       // no C operand: c = zerof, beta = 0 (a zero sum comes out as +0)
    }
}
extern "C" int orbm_search_by_projection_last(orbm_matcher *m, int n_last, const uint8_t *has_point, const float *xw, const uint8_t *mp_desc,
                                              const int32_t *mp_obs, const orbx_keypoint *kps_last, const float *Tcw, const float *Tlw,
                                              float fx, float fy, float cx, float cy, float mb, float mbf, const float bounds[4],
                                              const float *scale_factors, int nlevels, const orbx_keypoint *kps_cur, const uint8_t *desc_cur,
                                              const float *u_right, int n_cur, float th, int mono, int check_orientation,
                                              int32_t *cur_obs, int32_t *cur_match, int *nmatches)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n_last < 0 || n_cur < 0 || !Tcw || !Tlw || !bounds || !scale_factors || nlevels < 1 || !nmatches ||
        (n_last > 0 && (!has_point || !xw || !mp_desc || !mp_obs || !kps_last)) || (n_cur > 0 && (!kps_cur || !desc_cur || !cur_obs || !cur_match)))
        return mfail(ORBX_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n_cur; i++) cur_match[i] = -1;
    if (n_last == 0 || n_cur == 0) return ORBX_OK;
    if (!m->grid_ok || m->grid.n != n_cur) return mfail(ORBX_E_INVALID, "orbm_grid_build(current frame) has not been called");
    float twc[3];                                   // :1342-1350
    camera_center(Tcw, twc);
    const float tlc2 = gemm_row(Tlw, 2, twc);
    const bool forward = tlc2 > mb && !mono, backward = -tlc2 > mb && !mono;
    // projections (:1352-1394): one window per last-frame feature that survives the checks
    struct Qr { int i; float u, invzc, radius; };
    std::vector<Qr> qs;
    std::vector<float> x, y, r;
    std::vector<int32_t> mn, mx;
    for (int i = 0; i < n_last; i++) {
        if (!has_point[i]) continue;
        const float *X = xw + 3 * (size_t)i;
        const float xc = gemm_row(Tcw, 0, X), yc = gemm_row(Tcw, 1, X), zc = gemm_row(Tcw, 2, X);
        const float invzc = (float)(1.0 / zc);
        if (invzc < 0) continue;
        const float u = fx * xc * invzc + cx, v = fy * yc * invzc + cy;
        if (u < bounds[0] || u > bounds[1]) continue;
        if (v < bounds[2] || v > bounds[3]) continue;
        const int oct = kps_last[i].octave;
        if (oct < 0 || oct >= nlevels) return mfail(ORBX_E_INVALID, "last-frame keypoint %d on octave %d of %d", i, oct, nlevels);
        const float radius = th * scale_factors[oct];
        qs.push_back({i, u, invzc, radius});
        x.push_back(u); y.push_back(v); r.push_back(radius);
        if (forward) { mn.push_back(oct); mx.push_back(-1); }
        else if (backward) { mn.push_back(0); mx.push_back(oct); }
        else { mn.push_back(oct - 1); mx.push_back(oct + 1); }
    }
    const int nq = (int)qs.size();
    if (nq == 0) return ORBX_OK;
    { int rc_ = orbm_grow(m, nq, 0, 0); if (rc_ != ORBX_OK) return rc_; }
    std::vector<int32_t> off, idx, dist;
    std::vector<uint8_t> qd((size_t)nq * 32);
    for (int k = 0; k < nq; k++) memcpy(&qd[(size_t)k * 32], mp_desc + (size_t)qs[k].i * 32, 32);
    const int total = orbm_area_pairs(m, x.data(), y.data(), r.data(), mn.data(), mx.data(), nq, qd.data(), desc_cur, n_cur, off, idx, dist);
    if (total < 0) return total;
    // the sequential scan (:1396-1444)
    std::vector<std::pair<int, int>> rot;
    int32_t hist[ORBM_HISTO_LENGTH] = {0};
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int nm = 0;
    for (int k = 0; k < nq; k++) {
        if (off[k + 1] == off[k]) continue;
        const Qr &q = qs[k];
        int bestDist = 256, bestIdx2 = -1;
        for (int c = off[k]; c < off[k + 1]; c++) {
            const int i2 = idx[c];
            if (cur_obs[i2] > 0) continue;
            if (u_right && u_right[i2] > 0) {
                const float ur = q.u - mbf * q.invzc;
                const float er = fabsf(ur - u_right[i2]);
                if (er > q.radius) continue;
            }
            const int d = dist[c];
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= ORBM_TH_HIGH) {
            cur_obs[bestIdx2] = mp_obs[q.i];
            cur_match[bestIdx2] = q.i;
            nm++;
            if (check_orientation) {
                float rot_ = kps_last[q.i].angle - kps_cur[bestIdx2].angle;
                if (rot_ < 0.0) rot_ += 360.0f;
                int bin = (int)roundf(rot_ * factor);
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
                if (bin < 0 || bin >= ORBM_HISTO_LENGTH) return mfail(ORBX_E_INVALID, "keypoint angle outside [0, 360)");
                rot.emplace_back(bin, bestIdx2);
                hist[bin]++;
            }
        }
    }
    if (check_orientation) {                         // :1447-1466
        int32_t ind[3];
        orbm_three_maxima(hist, ORBM_HISTO_LENGTH, ind);
        for (const auto &e : rot)
            if (e.first != ind[0] && e.first != ind[1] && e.first != ind[2]) { cur_obs[e.second] = -1; cur_match[e.second] = -1; nm--; }
    }
    *nmatches = nm;
    return ORBX_OK;
}

// ---- ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)
// (src/ORBmatcher.cc:1472-1599), in three pieces so that a caller holding real MapPoint objects can keep calling its own
// MapPoint::PredictScale between the projection and the search (mfMaxDistance is not reachable from outside the class) ----
extern "C" int orbm_project_points(const float *Tcw, float fx, float fy, float cx, float cy, const float bounds[4],
                                   const float *xw, int n, float *u, float *v, float *invzc, float *dist3d, uint8_t *in_image)
{
    if (!Tcw || !bounds || n < 0 || (n > 0 && (!xw || !u || !v || !in_image))) return mfail(ORBX_E_INVALID, "bad argument");
    float Ow[3];                                    // Ow = -Rcw^T tcw (:1478)
    camera_center(Tcw, Ow);
    for (int i = 0; i < n; i++) {
        const float *X = xw + 3 * (size_t)i;
        const float xc = gemm_row(Tcw, 0, X), yc = gemm_row(Tcw, 1, X), zc = gemm_row(Tcw, 2, X);   // :1498
        const float iz = (float)(1.0 / zc);                                                            // :1502
        u[i] = fx * xc * iz + cx; v[i] = fy * yc * iz + cy;                                            // :1504-1505
        in_image[i] = !(u[i] < bounds[0] || u[i] > bounds[1] || v[i] < bounds[2] || v[i] > bounds[3]); // :1507-1510
        if (invzc) invzc[i] = iz;
        if (dist3d) {                               // cv::norm(x3Dw - Ow) (:1513-1514): float difference, double accumulation
            double nn = 0;
            for (int k = 0; k < 3; k++) { const float po = X[k] - Ow[k]; nn += (double)po * (double)po; }
            dist3d[i] = (float)sqrt(nn);
        }
    }
    return ORBX_OK;
}

extern "C" int orbm_predict_scale(float mf_max_distance, float current_dist, float log_scale_factor, int n_levels)
{
    const float ratio = mf_max_distance / current_dist;                // src/MapPoint.cc:407
    int nScale = (int)ceilf(logf(ratio) / log_scale_factor);          // :410 (log of a float: logf)
    if (nScale < 0) nScale = 0;
    else if (nScale >= n_levels) nScale = n_levels - 1;
    return nScale;
}

extern "C" int orbm_search_by_projection_kf(orbm_matcher *m, int n_mp, const uint8_t *use, const float *proj_u, const float *proj_v,
                                            const int32_t *pred_level, const uint8_t *mp_desc, const float *kf_angle,
                                            const float *scale_factors, int nlevels, const orbx_keypoint *kps_cur, const uint8_t *desc_cur,
                                            int n_cur, float th, int orb_dist, int check_orientation,
                                            uint8_t *cur_has_point, int32_t *cur_match, int *nmatches)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n_mp < 0 || n_cur < 0 || !scale_factors || nlevels < 1 || !nmatches ||
        (n_mp > 0 && (!use || !proj_u || !proj_v || !pred_level || !mp_desc || (check_orientation && !kf_angle))) ||
        (n_cur > 0 && (!kps_cur || !desc_cur || !cur_has_point || !cur_match)))
        return mfail(ORBX_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n_cur; i++) cur_match[i] = -1;
    if (n_mp == 0 || n_cur == 0) return ORBX_OK;
    if (!m->grid_ok || m->grid.n != n_cur) return mfail(ORBX_E_INVALID, "orbm_grid_build(current frame) has not been called");
    std::vector<int> qi;
    std::vector<float> x, y, r;
    std::vector<int32_t> mn, mx;
    for (int i = 0; i < n_mp; i++) {
        if (!use[i]) continue;
        const int lv = pred_level[i];
        if (lv < 0 || lv >= nlevels) return mfail(ORBX_E_INVALID, "MapPoint %d predicted on level %d of %d", i, lv, nlevels);
        qi.push_back(i); x.push_back(proj_u[i]); y.push_back(proj_v[i]); r.push_back(th * scale_factors[lv]);   // :1526
        mn.push_back(lv - 1); mx.push_back(lv + 1);                                                              // :1528
    }
    const int nq = (int)qi.size();
    if (nq == 0) return ORBX_OK;
    { int rc_ = orbm_grow(m, nq, 0, 0); if (rc_ != ORBX_OK) return rc_; }
    std::vector<int32_t> off, idx, dist;
    std::vector<uint8_t> qd((size_t)nq * 32);
    for (int k = 0; k < nq; k++) memcpy(&qd[(size_t)k * 32], mp_desc + (size_t)qi[k] * 32, 32);
    const int total = orbm_area_pairs(m, x.data(), y.data(), r.data(), mn.data(), mx.data(), nq, qd.data(), desc_cur, n_cur, off, idx, dist);
    if (total < 0) return total;
    // the sequential scan (:1538-1575): an assignment blocks the slot for every later MapPoint
    std::vector<std::pair<int, int>> rot;
    int32_t hist[ORBM_HISTO_LENGTH] = {0};
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int nm = 0;
    for (int k = 0; k < nq; k++) {
        if (off[k + 1] == off[k]) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = off[k]; c < off[k + 1]; c++) {
            const int i2 = idx[c];
            if (cur_has_point[i2]) continue;
            const int d = dist[c];
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist && bestIdx2 >= 0) {
            cur_has_point[bestIdx2] = 1;
            cur_match[bestIdx2] = qi[k];
            nm++;
            if (check_orientation) {
                float rot_ = kf_angle[qi[k]] - kps_cur[bestIdx2].angle;
                if (rot_ < 0.0) rot_ += 360.0f;
                int bin = (int)roundf(rot_ * factor);
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
                if (bin < 0 || bin >= ORBM_HISTO_LENGTH) return mfail(ORBX_E_INVALID, "keypoint angle outside [0, 360)");
                rot.emplace_back(bin, bestIdx2);
                hist[bin]++;
            }
        }
    }
    if (check_orientation) {                         // :1577-1596
        int32_t ind[3];
        orbm_three_maxima(hist, ORBM_HISTO_LENGTH, ind);
        for (const auto &e : rot)
            if (e.first != ind[0] && e.first != ind[1] && e.first != ind[2]) { cur_has_point[e.second] = 0; cur_match[e.second] = -1; nm--; }
    }
    *nmatches = nm;
    return ORBX_OK;
}

// ---- ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:45-125) ----
extern "C" int orbm_search_by_projection_map(orbm_matcher *m, int n_mp, const uint8_t *in_view, const float *proj_x, const float *proj_y,
                                             const float *proj_xr, const int32_t *pred_level, const float *view_cos, const uint8_t *mp_desc,
                                             const int32_t *mp_obs, const float *scale_factors, int nlevels, const orbx_keypoint *kps_cur,
                                             const uint8_t *desc_cur, const float *u_right, int n_cur, float th, float nnratio,
                                             int32_t *cur_obs, int32_t *cur_match, int *nmatches)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n_mp < 0 || n_cur < 0 || !scale_factors || nlevels < 1 || !nmatches || (u_right && !proj_xr) ||
        (n_mp > 0 && (!in_view || !proj_x || !proj_y || !pred_level || !view_cos || !mp_desc || !mp_obs)) ||
        (n_cur > 0 && (!kps_cur || !desc_cur || !cur_obs || !cur_match)))
        return mfail(ORBX_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n_cur; i++) cur_match[i] = -1;
    if (n_mp == 0 || n_cur == 0) return ORBX_OK;
    if (!m->grid_ok || m->grid.n != n_cur) return mfail(ORBX_E_INVALID, "orbm_grid_build(frame) has not been called");
    const bool bFactor = th != 1.0;
    std::vector<int> qi;
    std::vector<float> x, y, r;
    std::vector<int32_t> mn, mx;
    for (int i = 0; i < n_mp; i++) {
        if (!in_view[i]) continue;
        const int lv = pred_level[i];
        if (lv < 0 || lv >= nlevels) return mfail(ORBX_E_INVALID, "MapPoint %d predicted on level %d of %d", i, lv, nlevels);
        float rr = view_cos[i] > 0.998 ? 2.5f : 4.0f;         // RadiusByViewingCos
        if (bFactor) rr *= th;
        qi.push_back(i); x.push_back(proj_x[i]); y.push_back(proj_y[i]); r.push_back(rr * scale_factors[lv]);
        mn.push_back(lv - 1); mx.push_back(lv);
    }
    const int nq = (int)qi.size();
    if (nq == 0) return ORBX_OK;
    { int rc_ = orbm_grow(m, nq, 0, 0); if (rc_ != ORBX_OK) return rc_; }
    std::vector<int32_t> off, idx, dist;
    std::vector<uint8_t> qd((size_t)nq * 32);
    for (int k = 0; k < nq; k++) memcpy(&qd[(size_t)k * 32], mp_desc + (size_t)qi[k] * 32, 32);
    const int total = orbm_area_pairs(m, x.data(), y.data(), r.data(), mn.data(), mx.data(), nq, qd.data(), desc_cur, n_cur, off, idx, dist);
    if (total < 0) return total;
    int nm = 0;
    for (int k = 0; k < nq; k++) {                  // :73-122
        if (off[k + 1] == off[k]) continue;
        const int iMP = qi[k];
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = off[k]; c < off[k + 1]; c++) {
            const int i2 = idx[c];
            if (cur_obs[i2] > 0) continue;
            if (u_right && u_right[i2] > 0) {
                const float er = fabsf(proj_xr[iMP] - u_right[i2]);
                if (er > r[k]) continue;
            }
            const int d = dist[c];
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = kps_cur[i2].octave; bestIdx = i2; }
            else if (d < bestDist2) { bestLevel2 = kps_cur[i2].octave; bestDist2 = d; }
        }
        if (bestDist <= ORBM_TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            cur_obs[bestIdx] = mp_obs[iMP];
            cur_match[bestIdx] = iMP;
            nm++;
        }
    }
    *nmatches = nm;
    return ORBX_OK;
}
