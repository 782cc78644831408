// orbx_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the ORB extractor.
//
// Stages (reference: src/ORBextractor.cc of WChen09/My-SLAM; OpenCV 3.1.0 semantics per DESIGN.md):
//   k_resize_linear / k_resize_area2   ComputePyramid                         :1109-1137
//   k_fast_cells                       per-cell FAST-9/16 + NMS + 20->7 fallback :767-831
//   k_octree                           DistributeOctTree + DivideNode          :483-765
//   k_describe                         IC_Angle + GaussianBlur 7x7 + rBRIEF     :79-149,1087-1103
// All image arithmetic is integer; the fp32 in fastAtan2 and in the sample rotation is written with
// explicit __f*_rn intrinsics so no FMA contraction can happen, and the file is also
// built with -ffp-contract=off.
#include "orbx_internal.h"
#include "orb_pattern_data.h"

__constant__ int c_umax[16];
__constant__ int c_gauss[7];
__constant__ signed char c_pattern[1024];

int orbx_upload_constants(const int umax[16], const int gauss_k[7])
{
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax, sizeof(int) * 16) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), gauss_k, sizeof(int) * 7) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), ORBX_PATTERN, 1024) != hipSuccess) return -1;
    return 0;
}

// -------------------------------------------------------------------------------------------------
// Pyramid: cv::resize INTER_LINEAR, CV_8UC1 fixed point (11-bit coefficients).  The per-column and
// per-row source offsets / coefficients are planned on the host exactly as OpenCV does (double
// arithmetic there), so the kernel is integer only.  One thread = 4 consecutive destination pixels.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_linear(
    const uint8_t *__restrict__ src, int sw, int sh, int sstride, long long sframe,
    uint8_t *__restrict__ dst, int dw, int dh, int dstride, long long dframe, ResizeTab tab)
{
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (y >= dh || x4 >= dw) return;
    const uint8_t *S = src + (long long)blockIdx.z * sframe;
    uint8_t *D = dst + (long long)blockIdx.z * dframe + (long long)y * dstride;
    const int sy = tab.yofs[y];
    const short2 b = tab.beta[y];
    const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
    const uint8_t *R0 = S + (long long)sy0 * sstride, *R1 = S + (long long)sy1 * sstride;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int dx = min(x4 + i, dw - 1);
        const int sx = tab.xofs[dx];
        const short2 a = tab.alpha[dx];
        const int sx1 = min(sx + 1, sw - 1);
        const int h0 = R0[sx] * a.x + R0[sx1] * a.y;
        const int h1 = R1[sx] * a.x + R1[sx1] * a.y;
        const int v = (((b.x * (h0 >> 4)) >> 16) + ((b.y * (h1 >> 4)) >> 16) + 2) >> 2;
        out |= (uint32_t)(v & 255) << (8 * i);
    }
    if (x4 + 3 < dw) {
        *reinterpret_cast<uint32_t *>(D + x4) = out;
    } else {
        for (int i = 0; x4 + i < dw; i++) D[x4 + i] = (uint8_t)(out >> (8 * i));
    }
}

// exact 2x decimation: OpenCV reroutes INTER_LINEAR to INTER_AREA, (s00+s01+s10+s11+2)>>2
__global__ __launch_bounds__(256) void k_resize_area2(
    const uint8_t *__restrict__ src, int sstride, long long sframe,
    uint8_t *__restrict__ dst, int dw, int dh, int dstride, long long dframe)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (y >= dh || x >= dw) return;
    const uint8_t *s0 = src + (long long)blockIdx.z * sframe + (long long)(2 * y) * sstride + 2 * x;
    const uint8_t *s1 = s0 + sstride;
    dst[(long long)blockIdx.z * dframe + (long long)y * dstride + x] =
        (uint8_t)((s0[0] + s0[1] + s1[0] + s1[1] + 2) >> 2);
}

void orbx_launch_resize(const OrbxLevel &src, const OrbxLevel &dst, const ResizeTab &tab, int area2,
                        int nframes, hipStream_t s)
{
    dim3 block(64, 4);
    if (area2) {
        dim3 grid((dst.w + 63) / 64, (dst.h + 3) / 4, nframes);
        hipLaunchKernelGGL(k_resize_area2, grid, block, 0, s, src.base, src.stride, src.frame_stride,
                           dst.base, dst.w, dst.h, dst.stride, dst.frame_stride);
    } else {
        dim3 grid((dst.w + 255) / 256, (dst.h + 3) / 4, nframes);
        hipLaunchKernelGGL(k_resize_linear, grid, block, 0, s, src.base, src.w, src.h, src.stride,
                           src.frame_stride, dst.base, dst.w, dst.h, dst.stride, dst.frame_stride, tab);
    }
}

// -------------------------------------------------------------------------------------------------
// FAST-9/16 per cell.  One 256-thread workgroup = one 30x30-ish cell of one level of one frame:
//   tile (cell + 3 px ring) -> LDS; stage 1 wave-ballot compaction of the pixels that pass the
//   4-compass-point reject at the lower threshold; stage 2 full 16-pixel arc test (bit masks,
//   rotate-AND) and corner score (max over the 16 nine-arcs of the min |diff|, minus 1) into an LDS
//   score map; stage 3 strict 8-neighbour NMS inside the cell; then the reference's per-cell
//   threshold rule: keep score >= iniThFAST if any such maximum exists, else score >= minThFAST.
// Identity used: corner at threshold t <=> score >= t, and a pixel that survives NMS
// at threshold t is a strict maximum of the t-independent score map, so one map serves both.
// Candidates are appended to the level's buffer with one global atomic per cell; their order is
// irrelevant because the quadtree selection recomputes the reference's scan order from (x, y).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_append(bool pass, int *counter)
{
    const unsigned long long m = __ballot(pass);
    if (m == 0) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader);
    return pass ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

__global__ __launch_bounds__(FAST_THREADS) void k_fast_cells(OrbxPlan plan, OrbxWork wk)
{
    __shared__ uint8_t tile[FAST_TILE_MAX * FAST_TILE_STRIDE];
    __shared__ uint8_t smap[(FAST_ZONE_MAX + 2) * (FAST_ZONE_MAX + 2)];
    __shared__ uint16_t lst1[FAST_ZONE_MAX * FAST_ZONE_MAX];
    __shared__ uint16_t lst2[FAST_ZONE_MAX * FAST_ZONE_MAX];
    __shared__ int s_n1, s_n2, s_n3, s_nini, s_nmin, s_gbase, s_nout;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cell = blockIdx.x, f = blockIdx.y;
    int l = 0;
    for (int i = 1; i < plan.nlevels; i++)
        if (cell >= plan.lv[i].cell_begin) l = i;
    const OrbxLevel &L = plan.lv[l];
    const int c = cell - L.cell_begin;
    const int ci = c / L.nCols, cj = c - ci * L.nCols;
    const int iniX = ORBX_MINB + cj * L.wCell, iniY = ORBX_MINB + ci * L.hCell;
    const int tw = min(iniX + L.wCell + 6, L.maxBX) - iniX;
    const int th = min(iniY + L.hCell + 6, L.maxBY) - iniY;
    const int zw = tw - 6, zh = th - 6;
    if (zw <= 0 || zh <= 0) return;
    const int t_ini = plan.ini_th, t_min = plan.min_th;
    const int t_lo = min(t_ini, t_min);
    const int sms = zw + 2;

    if (tid == 0) { s_n1 = 0; s_n2 = 0; s_n3 = 0; s_nini = 0; s_nmin = 0; s_nout = 0; }
    const uint8_t *img = L.base + (long long)f * L.frame_stride + (long long)iniY * L.stride + iniX;
    for (int y = wave; y < th; y += FAST_THREADS / 64)
        for (int x = lane; x < tw; x += 64) tile[y * FAST_TILE_STRIDE + x] = img[(long long)y * L.stride + x];
    for (int i = tid; i < sms * (zh + 2); i += FAST_THREADS) smap[i] = 0;
    __syncthreads();

    // stage 1: compass reject (a 9-arc contains at least one pixel of every opposite pair)
    for (int y = wave; y < zh; y += FAST_THREADS / 64) {
        for (int x0 = 0; x0 < zw; x0 += 64) {
            const int x = x0 + lane;
            bool pass = false;
            if (x < zw) {
                const uint8_t *p = &tile[(y + 3) * FAST_TILE_STRIDE + x + 3];
                const int v = p[0], lo = v - t_lo, hi = v + t_lo;
                const int a = p[3 * FAST_TILE_STRIDE], b = p[-3 * FAST_TILE_STRIDE];
                int d = ((a < lo) ? 1 : 0) | ((a > hi) ? 2 : 0) | ((b < lo) ? 1 : 0) | ((b > hi) ? 2 : 0);
                if (d) {
                    const int e = p[3], w = p[-3];
                    d &= ((e < lo) ? 1 : 0) | ((e > hi) ? 2 : 0) | ((w < lo) ? 1 : 0) | ((w > hi) ? 2 : 0);
                }
                pass = d != 0;
            }
            const int slot = wave_append(pass, &s_n1);
            if (slot >= 0) lst1[slot] = (uint16_t)((y << 6) | x);
        }
    }
    __syncthreads();

    // stage 2: full arc test + score
    const int n1 = s_n1;
    for (int i0 = wave * 64; i0 < n1; i0 += FAST_THREADS) {
        const int i = i0 + lane;
        bool corner = false;
        int pos = 0;
        if (i < n1) {
            pos = lst1[i];
            const int y = pos >> 6, x = pos & 63;
            const uint8_t *p = &tile[(y + 3) * FAST_TILE_STRIDE + x + 3];
            const int v = p[0];
            int d[16];
            d[0] = v - p[3 * FAST_TILE_STRIDE];
            d[1] = v - p[3 * FAST_TILE_STRIDE + 1];
            d[2] = v - p[2 * FAST_TILE_STRIDE + 2];
            d[3] = v - p[1 * FAST_TILE_STRIDE + 3];
            d[4] = v - p[3];
            d[5] = v - p[-1 * FAST_TILE_STRIDE + 3];
            d[6] = v - p[-2 * FAST_TILE_STRIDE + 2];
            d[7] = v - p[-3 * FAST_TILE_STRIDE + 1];
            d[8] = v - p[-3 * FAST_TILE_STRIDE];
            d[9] = v - p[-3 * FAST_TILE_STRIDE - 1];
            d[10] = v - p[-2 * FAST_TILE_STRIDE - 2];
            d[11] = v - p[-1 * FAST_TILE_STRIDE - 3];
            d[12] = v - p[-3];
            d[13] = v - p[1 * FAST_TILE_STRIDE - 3];
            d[14] = v - p[2 * FAST_TILE_STRIDE - 2];
            d[15] = v - p[3 * FAST_TILE_STRIDE - 1];
            uint32_t dm = 0, bm = 0;   // darker: p_k < v - t  <=> d > t ; brighter: d < -t
#pragma unroll
            for (int k = 0; k < 16; k++) {
                dm |= (d[k] > t_lo ? 1u : 0u) << k;
                bm |= (d[k] < -t_lo ? 1u : 0u) << k;
            }
            uint32_t m2 = dm | (dm << 16), r = m2 & (m2 >> 1);
            r &= r >> 2; r &= r >> 4; r &= m2 >> 8;
            uint32_t n2 = bm | (bm << 16), q = n2 & (n2 >> 1);
            q &= q >> 2; q &= q >> 4; q &= n2 >> 8;
            corner = ((r | q) & 0xFFFFu) != 0;
            if (corner) {
                int mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
                for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
                for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
                int A = -256, B = 256;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
                    const int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
                    A = max(A, mn9);
                    B = min(B, mx9);
                }
                const int score = max(A, -B) - 1;   // >= t_lo for a corner, <= 254
                smap[(y + 1) * sms + x + 1] = (uint8_t)score;
            }
        }
        const int slot = wave_append(corner, &s_n2);
        if (slot >= 0) lst2[slot] = (uint16_t)pos;
    }
    __syncthreads();

    // stage 3: NMS (strictly greater than all 8 neighbours; outside the cell zone counts as 0)
    const int n2c = s_n2;
    for (int i0 = wave * 64; i0 < n2c; i0 += FAST_THREADS) {
        const int i = i0 + lane;
        bool ismax = false;
        int pos = 0, s = 0;
        if (i < n2c) {
            pos = lst2[i];
            const int y = pos >> 6, x = pos & 63;
            const uint8_t *q = &smap[(y + 1) * sms + x + 1];
            s = q[0];
            ismax = s > q[-1] && s > q[1] && s > q[-sms - 1] && s > q[-sms] && s > q[-sms + 1] &&
                    s > q[sms - 1] && s > q[sms] && s > q[sms + 1];
        }
        const int slot = wave_append(ismax, &s_n3);
        if (slot >= 0) lst1[slot] = (uint16_t)pos;
        const unsigned long long mi = __ballot(ismax && s >= t_ini), mm = __ballot(ismax && s >= t_min);
        if (lane == 0) {
            if (mi) atomicAdd(&s_nini, __popcll(mi));
            if (mm) atomicAdd(&s_nmin, __popcll(mm));
        }
    }
    __syncthreads();

    // reference :811-818: FAST at iniThFAST; only if that yields nothing, FAST at minThFAST
    const int t_use = s_nini > 0 ? t_ini : t_min;
    const int total = s_nini > 0 ? s_nini : s_nmin;
    if (total == 0) return;
    if (tid == 0) s_gbase = (int)atomicAdd(&wk.cand_count[f * plan.nlevels + l], (uint32_t)total);
    __syncthreads();
    const int gbase = s_gbase, n3 = s_n3;
    OrbxCand *out = wk.cand + (long long)f * plan.cand_frame + L.cand_off;
    for (int i = tid; i < n3; i += FAST_THREADS) {
        const int pos = lst1[i];
        const int y = pos >> 6, x = pos & 63;
        const int s = smap[(y + 1) * sms + x + 1];
        if (s >= t_use) {
            const int o = gbase + atomicAdd(&s_nout, 1);
            if (o < L.cand_cap) {
                OrbxCand cnd;
                cnd.xy = (uint32_t)(iniX + 3 + x) | ((uint32_t)(iniY + 3 + y) << 16);
                cnd.resp = (uint32_t)s;
                out[o] = cnd;
            } else {
                atomicOr(&wk.errflags[f], (uint32_t)ERRF_CAND_OVERFLOW);
            }
        }
    }
}

void orbx_launch_fast(const OrbxPlan &plan, const OrbxWork &wk, int nframes, hipStream_t s)
{
    if (plan.ncells <= 0) return;
    dim3 grid(plan.ncells, nframes);
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(FAST_THREADS), 0, s, plan, wk);
}

// -------------------------------------------------------------------------------------------------
// Quadtree distribution.  One workgroup = one (frame, level).  The reference's std::list algorithm
// is restated on arrays:
//   * keys never move: owner[k] is the arena id of the node that currently holds candidate k;
//   * a pass expands a set E of nodes in a processing order pi and the new list is
//       reverse(children of E in creation order) ++ (old list minus E),
//     which is what push_front + erase produce (:619-665, :689-730);
//   * phase A (:596-667): E = every node with > 1 key, pi = list order;
//   * phase B (:675-739): pi = nodes sorted by (size, creation) descending, E = the shortest prefix
//     after which the list has >= N nodes (the reference's break at :732), found with a prefix sum;
//   * the final "best response, first wins" (:746-762) uses the reference's scan order recomputed
//     from (x, y): cells row-major, then rows, then columns.
// The reference's pointer tie-break in the sort (:629,:686) is taken as creation order.
// -------------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ int block_excl_scan(int v, int *total, int *wsum)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int w = lane < T / 64 ? wsum[lane] : 0;
        int winc = w;
#pragma unroll
        for (int o = 1; o < T / 64; o <<= 1) {
            const int t = __shfl_up(winc, o);
            if (lane >= o) winc += t;
        }
        if (lane < T / 64) wsum[lane] = winc - w;
        if (lane == T / 64 - 1) wsum[T / 64] = winc;
    }
    __syncthreads();
    const int res = wsum[wave] + inc - v;
    *total = wsum[T / 64];
    __syncthreads();
    return res;
}

struct OctShared {
    int m, prevM, arenaN, lastBase, lastC, nAlive, nE, C, nToExpand, phaseB, done, cutoff, err, firstPass;
    int wsum[OCT_THREADS / 64 + 2];
};

__device__ __forceinline__ void oct_child_box(const OrbxNode &p, int q, OrbxNode &c)
{
    const int halfX = (p.x1 - p.x0 + 1) >> 1;   // ceil((UR.x-UL.x)/2), :485
    const int halfY = (p.y1 - p.y0 + 1) >> 1;   // ceil((BR.y-UL.y)/2), :486
    c.x0 = (q & 1) ? p.x0 + halfX : p.x0;
    c.x1 = (q & 1) ? p.x1 : p.x0 + halfX;
    c.y0 = (q & 2) ? p.y0 + halfY : p.y0;
    c.y1 = (q & 2) ? p.y1 : p.y0 + halfY;
}

__global__ __launch_bounds__(OCT_THREADS) void k_octree(OrbxPlan plan, OrbxWork wk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char oct_lds[];
    __shared__ OctShared sh;
    constexpr int T = OCT_THREADS;
    const int tid = threadIdx.x;
    const int l = blockIdx.x, f = blockIdx.y;
    const OrbxLevel &L = plan.lv[l];
    const int cap = L.list_cap;
    const int N = L.quota;

    uint32_t *cnt = reinterpret_cast<uint32_t *>(oct_lds);                 // [4*cap]  counts, then child ids
    uint32_t *listA = cnt + 4 * cap;                                       // [cap]
    uint32_t *listB = listA + cap;                                         // [cap]
    uint32_t *slotNode = listB + cap;                                      // [cap]
    int *childBase = reinterpret_cast<int *>(slotNode + cap);              // [cap]
    unsigned long long *sortbuf = reinterpret_cast<unsigned long long *>(cnt);   // alias, phase B
    unsigned long long *best = reinterpret_cast<unsigned long long *>(cnt);      // alias, final

    const OrbxCand *cand = wk.cand + (long long)f * plan.cand_frame + L.cand_off;
    uint32_t *owner = wk.owner + (long long)f * plan.cand_frame + L.cand_off;
    OrbxNode *arena = wk.arena + (long long)f * plan.arena_frame + L.arena_off;
    OrbxCand *sel = wk.sel + (long long)f * plan.list_frame + L.list_off;
    const int n = (int)min(wk.cand_count[f * plan.nlevels + l], (uint32_t)L.cand_cap);

    if (n == 0 || L.nIni <= 0) {
        if (tid == 0) wk.nk[f * plan.nlevels + l] = 0;
        return;
    }
    const int nIni = L.nIni;
    const int boxH = L.maxBY - ORBX_MINB;

    // ---- roots (:554-572) ----
    for (int i = tid; i < nIni; i += T) cnt[i] = 0;
    if (tid == 0) {
        sh.err = 0; sh.phaseB = 0; sh.done = 0; sh.firstPass = 1; sh.nToExpand = 0;
    }
    __syncthreads();
    for (int k = tid; k < n; k += T) {
        const int xr = (int)(cand[k].xy & 0xFFFFu) - ORBX_MINB;
        int b = (int)__fdiv_rn((float)xr, L.hX);
        b = min(max(b, 0), nIni - 1);
        owner[k] = (uint32_t)b;
        atomicAdd(&cnt[b], 1u);
    }
    __syncthreads();
    for (int i = tid; i < nIni; i += T) {
        OrbxNode nd;
        nd.x0 = (int16_t)(int)__fmul_rn(L.hX, (float)i);
        nd.x1 = (int16_t)(int)__fmul_rn(L.hX, (float)(i + 1));
        nd.y0 = 0;
        nd.y1 = (int16_t)boxH;
        nd.count = (int)cnt[i];
        nd.slot = i;   // creation index
        arena[i] = nd;
    }
    __syncthreads();
    if (tid == 0) {   // initial list: non-empty roots in order (:574-587); nIni is small
        int m = 0;
        for (int i = 0; i < nIni; i++)
            if (cnt[i] > 0) listA[m++] = (uint32_t)i;
        sh.m = m; sh.arenaN = nIni; sh.lastBase = 0; sh.lastC = nIni;
    }
    __syncthreads();
    uint32_t *cur = listA, *nxt = listB;

    // ---- expansion passes ----
    while (true) {
        const int m = sh.m, lastBase = sh.lastBase, lastC = sh.lastC, arenaN = sh.arenaN;
        const int phaseB = sh.phaseB, firstPass = sh.firstPass;
        __syncthreads();
        // S1: slots for the alive nodes (all of them were created by the previous pass)
        int nAlive;
        if (!phaseB) {
            int carry = 0;
            for (int base = 0; base < lastC; base += T) {
                const int j = base + tid;
                int id = -1, alive = 0;
                if (j < lastC) {
                    const int cidx = firstPass ? j : lastC - 1 - j;   // list order of the last children
                    id = lastBase + cidx;
                    alive = arena[id].count > 1;
                }
                int tot;
                const int ex = block_excl_scan<T>(alive, &tot, sh.wsum);
                if (alive) {
                    slotNode[carry + ex] = (uint32_t)id;
                    arena[id].slot = carry + ex;
                }
                carry += tot;
            }
            nAlive = carry;
        } else {
            int P = 1;
            while (P < lastC) P <<= 1;
            for (int j = tid; j < P; j += T) {
                unsigned long long key = 0;
                if (j < lastC) {
                    const int c = arena[lastBase + j].count;
                    if (c > 1) key = ((unsigned long long)(uint32_t)c << 32) | (uint32_t)j;
                }
                sortbuf[j] = key;
            }
            __syncthreads();
            for (int k2 = 2; k2 <= P; k2 <<= 1) {          // bitonic, descending
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    for (int i = tid; i < P; i += T) {
                        const int ixj = i ^ j2;
                        if (ixj > i) {
                            const unsigned long long a = sortbuf[i], b = sortbuf[ixj];
                            const bool up = (i & k2) == 0;   // descending block
                            if (up ? (a < b) : (a > b)) { sortbuf[i] = b; sortbuf[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            }
            if (tid == 0) sh.nAlive = 0;
            __syncthreads();
            for (int j = tid; j < P; j += T)
                if (sortbuf[j] != 0 && (j + 1 == P || sortbuf[j + 1] == 0)) sh.nAlive = j + 1;
            __syncthreads();
            nAlive = sh.nAlive;
            // slotNode is not aliased with sortbuf/cnt: copy the sorted order out before cnt is zeroed
            for (int j = tid; j < nAlive; j += T) {
                const int id = lastBase + (int)(sortbuf[j] & 0xFFFFFFFFull);
                slotNode[j] = (uint32_t)id;
                arena[id].slot = j;
            }
        }
        __syncthreads();
        // S2: zero the quadrant counters
        for (int i = tid; i < 4 * nAlive; i += T) cnt[i] = 0;
        if (tid == 0) { sh.cutoff = 0x7FFFFFFF; sh.nToExpand = 0; }
        __threadfence_block();
        __syncthreads();
        // S3: quadrant of every key held by an alive node (:513-528)
        for (int k = tid; k < n; k += T) {
            const uint32_t id = owner[k] & OCT_ID_MASK;
            const OrbxNode nd = arena[id];
            if (nd.count > 1) {
                const uint32_t xy = cand[k].xy;
                const int xr = (int)(xy & 0xFFFFu) - ORBX_MINB, yr = (int)(xy >> 16) - ORBX_MINB;
                const int midx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1), midy = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
                const uint32_t q = (xr < midx ? 0u : 1u) | (yr < midy ? 0u : 2u);
                atomicAdd(&cnt[4 * nd.slot + q], 1u);
                owner[k] = id | (q << 30);
            }
        }
        __syncthreads();
        // S4: children per slot, prefix sums, phase-B cut-off
        int carry = 0;
        for (int base = 0; base < nAlive; base += T) {
            const int s = base + tid;
            int ne = 0;
            if (s < nAlive) ne = (cnt[4 * s] > 0) + (cnt[4 * s + 1] > 0) + (cnt[4 * s + 2] > 0) + (cnt[4 * s + 3] > 0);
            int tot;
            const int ex = block_excl_scan<T>(ne, &tot, sh.wsum);
            if (s < nAlive) {
                childBase[s] = carry + ex;
                if (phaseB && m + carry + ex + ne - (s + 1) >= N) atomicMin(&sh.cutoff, s);
            }
            carry += tot;
        }
        __syncthreads();
        int nE = nAlive, C = carry;
        if (phaseB && sh.cutoff != 0x7FFFFFFF) {
            nE = sh.cutoff + 1;
            const int s = nE - 1;
            C = childBase[s] + (cnt[4 * s] > 0) + (cnt[4 * s + 1] > 0) + (cnt[4 * s + 2] > 0) + (cnt[4 * s + 3] > 0);
        }
        const int survivors = m - nE;
        if (arenaN + C > L.arena_cap || C + survivors > cap) {
            if (tid == 0) { atomicOr(&wk.errflags[f], (uint32_t)ERRF_TREE_OVERFLOW); wk.nk[f * plan.nlevels + l] = 0; }
            return;
        }
        __syncthreads();
        // S5: create the children (:489-537) in creation order; cnt becomes the child-id table
        int myExp = 0;
        for (int s = tid; s < nE; s += T) {
            const OrbxNode p = arena[slotNode[s]];
            int r = 0;
            for (int q = 0; q < 4; q++) {
                const int c = (int)cnt[4 * s + q];
                if (c > 0) {
                    const int cidx = childBase[s] + r;
                    OrbxNode ch;
                    oct_child_box(p, q, ch);
                    ch.count = c;
                    ch.slot = cidx;
                    arena[arenaN + cidx] = ch;
                    cnt[4 * s + q] = (uint32_t)(arenaN + cidx);
                    nxt[C - 1 - cidx] = (uint32_t)(arenaN + cidx);   // push_front => reversed
                    myExp += c > 1;
                    r++;
                } else {
                    cnt[4 * s + q] = 0xFFFFFFFFu;
                }
            }
        }
        if (myExp) atomicAdd(&sh.nToExpand, myExp);
        __threadfence_block();
        __syncthreads();
        // S6: move the keys of expanded nodes to their children
        for (int k = tid; k < n; k += T) {
            const uint32_t w = owner[k];
            const uint32_t id = w & OCT_ID_MASK;
            const OrbxNode nd = arena[id];
            if (nd.count > 1 && nd.slot < nE) owner[k] = cnt[4 * nd.slot + (w >> 30)];
            else owner[k] = id;
        }
        // S7: survivors keep their relative order behind the new children
        int scarry = 0;
        for (int base = 0; base < m; base += T) {
            const int i = base + tid;
            int keep = 0;
            uint32_t id = 0;
            if (i < m) {
                id = cur[i];
                const OrbxNode nd = arena[id];
                keep = !(nd.count > 1 && nd.slot < nE);
            }
            int tot;
            const int ex = block_excl_scan<T>(keep, &tot, sh.wsum);
            if (keep) nxt[C + scarry + ex] = id;
            scarry += tot;
        }
        __syncthreads();
        // S8: bookkeeping + termination (:671-675, :736)
        if (tid == 0) {
            const int newM = C + scarry;
            sh.prevM = m;
            sh.m = newM;
            sh.lastBase = arenaN;
            sh.lastC = C;
            sh.arenaN = arenaN + C;
            sh.firstPass = 0;
            if (newM >= N || newM == m) sh.done = 1;
            else if (!phaseB && newM + 3 * sh.nToExpand > N) sh.phaseB = 1;
        }
        __syncthreads();
        uint32_t *t2 = cur; cur = nxt; nxt = t2;
        if (sh.done) break;
    }

    // ---- final selection (:743-762) ----
    const int m = sh.m;
    __syncthreads();
    for (int i = tid; i < m; i += T) {
        arena[cur[i]].slot = i;
        best[i] = 0ull;   // aliases cnt; child ids are no longer needed
    }
    __threadfence_block();
    __syncthreads();
    for (int k = tid; k < n; k += T) {
        const uint32_t id = owner[k] & OCT_ID_MASK;
        const int pos = arena[id].slot;
        const OrbxCand c = cand[k];
        const int xa = (int)(c.xy & 0xFFFFu) - ORBX_EDGE, ya = (int)(c.xy >> 16) - ORBX_EDGE;
        const int cr = ya / L.hCell, cc = xa / L.wCell;
        const unsigned long long order = ((unsigned long long)cr << 24) | ((unsigned long long)cc << 12) |
                                         ((unsigned long long)(ya - cr * L.hCell) << 6) |
                                         (unsigned long long)(xa - cc * L.wCell);
        const unsigned long long pack = ((unsigned long long)c.resp << 56) |
                                        ((~order & 0xFFFFFFFFFull) << 20) | (unsigned long long)k;
        atomicMax(&best[pos], pack);
    }
    __syncthreads();
    for (int i = tid; i < m; i += T) sel[i] = cand[(int)(best[i] & 0xFFFFFull)];
    if (tid == 0) wk.nk[f * plan.nlevels + l] = (uint32_t)m;
}

size_t orbx_octree_lds_bytes(int list_cap_max) { return (size_t)list_cap_max * 32; }

void orbx_launch_octree(const OrbxPlan &plan, const OrbxWork &wk, int nframes, size_t lds_bytes, hipStream_t s)
{
    dim3 grid(plan.nlevels, nframes);
    if (lds_bytes > 32 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(k_octree, grid, dim3(OCT_THREADS), lds_bytes, s, plan, wk);
}

// -------------------------------------------------------------------------------------------------
// Orientation + blur + descriptor, one wave per keypoint.
//   raw 43x43 tile (reflect-101 at the image edge) -> LDS
//   IC_Angle moments over the radius-15 disc (integer), cv::fastAtan2 polynomial (fp32, no FMA)
//   7x7 sigma=2 fixed-point Gaussian of the inner 37x37 (row pass exact, column pass rounds once)
//   256 rotated comparisons, 4 wave ballots -> 32 bytes
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// cos/sin of an fp32 angle in [0, 2pi], rounded to fp32 from a double-precision evaluation
// (Cody-Waite reduction by pi/2 with a 33-bit head, Taylor kernels to r^19 / r^20 on |r| <= pi/4).
// Canonical semantics = correctly rounded cosf/sinf; tools/verify_sincos.py checks this
// routine against the x87 long-double libm over every fp32 input of the domain.
__device__ __forceinline__ void sincos_cr(float theta, float *cs, float *sn)
{
    const double x = (double)theta;
    const double kd = rint(x * 0.6366197723675814);
    const int k = (int)kd;
    double r = fma(-kd, 1.5707963267341256, x);          // exact: 33-bit head times k <= 4
    r = fma(-kd, 6.077100506506192e-11, r);
    const double z = r * r;
    double ps = -8.22063524662433e-18;
    ps = fma(ps, z, 2.8114572543455206e-15);
    ps = fma(ps, z, -7.647163731819816e-13);
    ps = fma(ps, z, 1.6059043836821613e-10);
    ps = fma(ps, z, -2.505210838544172e-08);
    ps = fma(ps, z, 2.7557319223985893e-06);
    ps = fma(ps, z, -0.0001984126984126984);
    ps = fma(ps, z, 0.008333333333333333);
    ps = fma(ps, z, -0.16666666666666666);
    const double s = fma(r * z, ps, r);
    double pc = 4.110317623312165e-19;
    pc = fma(pc, z, -1.5619206968586225e-16);
    pc = fma(pc, z, 4.779477332387385e-14);
    pc = fma(pc, z, -1.1470745597729725e-11);
    pc = fma(pc, z, 2.08767569878681e-09);
    pc = fma(pc, z, -2.755731922398589e-07);
    pc = fma(pc, z, 2.48015873015873e-05);
    pc = fma(pc, z, -0.001388888888888889);
    pc = fma(pc, z, 0.041666666666666664);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    double cv, sv;
    switch (k & 3) {
    case 0: cv = c; sv = s; break;
    case 1: cv = -s; sv = c; break;
    case 2: cv = -c; sv = -s; break;
    default: cv = s; sv = -c; break;
    }
    *cs = (float)cv;
    *sn = (float)sv;
}

// exported for the exhaustive sincos check (tools/verify_sincos.py)
__global__ void k_sincos_probe(const float *theta, float *cs, float *sn, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincos_cr(theta[i], &cs[i], &sn[i]);
}
extern "C" int orbx_debug_sincos(const float *h_theta, float *h_cos, float *h_sin, int n)
{
    float *d = nullptr;
    if (hipMalloc(&d, sizeof(float) * 3 * (size_t)n) != hipSuccess) return ORBX_E_HIP;
    hipMemcpy(d, h_theta, sizeof(float) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sincos_probe, dim3((n + 255) / 256), dim3(256), 0, 0, d, d + n, d + 2 * (size_t)n, n);
    hipMemcpy(h_cos, d + n, sizeof(float) * n, hipMemcpyDeviceToHost);
    hipMemcpy(h_sin, d + 2 * (size_t)n, sizeof(float) * n, hipMemcpyDeviceToHost);
    const hipError_t e = hipDeviceSynchronize();
    hipFree(d);
    return e == hipSuccess ? ORBX_OK : ORBX_E_HIP;
}

__global__ __launch_bounds__(DESC_THREADS) void k_describe(
    OrbxPlan plan, OrbxWork wk, orbx_keypoint *__restrict__ kps, uint8_t *__restrict__ desc,
    int32_t *__restrict__ counts, int32_t *__restrict__ status)
{
    __shared__ uint8_t raw[DESC_RAW * 44];
    __shared__ uint16_t rb[DESC_RAW * 38];
    __shared__ uint8_t bl[DESC_BL * 40];
    const int g = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;

    int total = 0, l = -1, idx = 0;
    for (int i = 0; i < plan.nlevels; i++) {
        const int c = (int)wk.nk[f * plan.nlevels + i];
        if (l < 0 && g < total + c) { l = i; idx = g - total; }
        total += c;
    }
    if (g == 0 && lane == 0) {
        const uint32_t e = wk.errflags[f];
        counts[f] = min(total, plan.out_cap);
        status[f] = (e & ERRF_CAND_OVERFLOW) ? ORBX_E_CAND_OVERFLOW
                  : (e & ERRF_TREE_OVERFLOW) ? ORBX_E_TREE_OVERFLOW
                  : (total > plan.out_cap)   ? ORBX_E_CAPACITY : ORBX_OK;
    }
    if (l < 0 || g >= plan.out_cap) return;
    const OrbxLevel &L = plan.lv[l];
    const OrbxCand kc = wk.sel[(long long)f * plan.list_frame + L.list_off + idx];
    const int x = (int)(kc.xy & 0xFFFFu), y = (int)(kc.xy >> 16);
    const uint8_t *img = L.base + (long long)f * L.frame_stride;

    for (int i = lane; i < DESC_RAW * DESC_RAW; i += DESC_THREADS) {
        const int r = i / DESC_RAW, c = i - r * DESC_RAW;
        const int gy = reflect101(y - 21 + r, L.h), gx = reflect101(x - 21 + c, L.w);
        raw[r * 44 + c] = img[(long long)gy * L.stride + gx];
    }
    __syncthreads();

    // IC_Angle (:79-106): m10 = sum u*I, m01 = sum v*I over |u| <= umax[|v|]
    int m10 = 0, m01 = 0;
    for (int i = lane; i < 31 * 31; i += DESC_THREADS) {
        const int vr = i / 31, v = vr - 15, u = i - vr * 31 - 15;
        if (abs(u) <= c_umax[abs(v)]) {
            const int I = raw[(21 + v) * 44 + 21 + u];
            m10 += u * I;
            m01 += v * I;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // 7x7 Gaussian, row pass (exact, <= 65535)
    const int k0 = c_gauss[0], k1 = c_gauss[1], k2 = c_gauss[2], k3 = c_gauss[3];
    for (int i = lane; i < DESC_RAW * DESC_BL; i += DESC_THREADS) {
        const int r = i / DESC_BL, c = i - r * DESC_BL;
        const uint8_t *p = &raw[r * 44 + c];
        rb[r * 38 + c] = (uint16_t)(k0 * (p[0] + p[6]) + k1 * (p[1] + p[5]) + k2 * (p[2] + p[4]) + k3 * p[3]);
    }
    __syncthreads();
    // column pass: (sum + 32768) >> 16 saturated (OpenCV C path) or round-half-even (OpenCV SSE2 path)
    const int simd_cols = plan.blur_mode == 1 ? (L.w & ~3) : 0;
    for (int i = lane; i < DESC_BL * DESC_BL; i += DESC_THREADS) {
        const int r = i / DESC_BL, c = i - r * DESC_BL;
        const uint16_t *p = &rb[r * 38 + c];
        const int s = k0 * (p[0] + p[6 * 38]) + k1 * (p[38] + p[5 * 38]) + k2 * (p[2 * 38] + p[4 * 38]) + k3 * p[3 * 38];
        int v;
        if (x - DESC_R + c < simd_cols) {
            v = s >> 16;
            const int rem = s & 0xFFFF;
            if (rem > 0x8000 || (rem == 0x8000 && (v & 1))) v++;
        } else {
            v = (s + 32768) >> 16;
        }
        bl[r * 40 + c] = (uint8_t)min(v, 255);
    }
    __syncthreads();

    // rBRIEF (:110-149)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float a, b;
    sincos_cr(__fmul_rn(angle, factorPI), &a, &b);
    unsigned long long bits[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int pair = j * 64 + lane;
        const signed char *pp = &c_pattern[pair * 4];
        const float x0 = (float)pp[0], y0 = (float)pp[1], x1 = (float)pp[2], y1 = (float)pp[3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int q0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int q1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bl[(DESC_R + r0) * 40 + DESC_R + q0];
        const int t1 = bl[(DESC_R + r1) * 40 + DESC_R + q1];
        bits[j] = __ballot(t0 < t1);
    }
    const long long o = (long long)f * plan.out_cap + g;
    if (lane < 4) {
        const unsigned long long wv = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        reinterpret_cast<unsigned long long *>(desc + o * 32)[lane] = wv;
    }
    if (lane == 0) {
        orbx_keypoint kp;
        kp.x = (float)x;
        kp.y = (float)y;
        if (l != 0) {   // :1097-1103 keypoint->pt *= scale
            kp.x = __fmul_rn(kp.x, L.scale);
            kp.y = __fmul_rn(kp.y, L.scale);
        }
        kp.size = L.kp_size;
        kp.angle = angle;
        kp.response = (float)kc.resp;
        kp.octave = l;
        kp.class_id = -1;
        kps[o] = kp;
    }
}

void orbx_launch_describe(const OrbxPlan &plan, const OrbxWork &wk, int nframes,
                          orbx_keypoint *d_kps, uint8_t *d_desc, int32_t *d_counts,
                          int32_t *d_status, hipStream_t s)
{
    dim3 grid(plan.out_cap, nframes);
    hipLaunchKernelGGL(k_describe, grid, dim3(DESC_THREADS), 0, s, plan, wk, d_kps, d_desc, d_counts, d_status);
}
