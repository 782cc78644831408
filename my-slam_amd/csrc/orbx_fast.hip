// orbx_fast.hip -- per-cell FAST-9/16 + NMS + threshold fallback (src/ORBextractor.cc:767-831) on gfx950.
//
// One WAVE (64 lanes) = one 30x30-ish cell of one level of one frame; four independent waves per
// workgroup, no workgroup barrier anywhere -- a cell is ~1 k pixels, so a whole workgroup per cell is
// latency-bound on its own barriers.  Per wave, all in its private LDS slice:
//   tile (cell + 3 px ring) loaded as aligned dwords -> LDS
//   stage 1  8-point reject at the lower threshold (every 9-arc holds one pixel of each opposite
//            pair, all of one class), ballot-compacted into a 128-entry queue
//   stage 2  whenever 64 survivors are queued: 16-pixel arc test (bit masks, rotate-AND) and corner
//            score (max over the 16 nine-arcs of the min |diff|, minus 1) into an LDS score map
//   stage 3  strict 8-neighbour NMS on the score map, one ballot mask per 64 pixels
//   output   the reference's per-cell rule (:811-818): keep score >= iniThFAST if any such maximum
//            exists, else score >= minThFAST; one global atomic per cell reserves the output range.
// Identity used: corner at threshold t <=> score >= t, and an NMS survivor at threshold
// t is a strict maximum of the t-independent score map, so one map serves both thresholds.
// Candidate order is irrelevant: the quadtree recomputes the reference's scan order from (x, y).
#include "orbx_internal.h"

#define WSYNC()                                                \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

template <int TS, int TH, int ZS>   // tile row stride (bytes, multiple of 4), tile rows, score-map stride/rows
struct FastLds {
    uint8_t tile[TS * TH];
    uint8_t smap[ZS * ZS];
    uint16_t queue[128];
    unsigned long long masks[(ZS * ZS + 63) / 64];
};

__device__ __forceinline__ int cls8(int a, int lo, int hi) { return ((a < lo) ? 1 : 0) | ((a > hi) ? 2 : 0); }

// stage 2 for up to 64 queued survivors: full 16-pixel arc test and corner score
template <int TS>
__device__ __forceinline__ void fast_stage2(const uint8_t *T0, const uint16_t *queue, uint8_t *smap, int cnt,
                                            int lane, int t_lo, int sms)
{
    if (lane >= cnt) return;
    const int pos = queue[lane];
    const int y = pos >> 6, x = pos & 63;
    const uint8_t *p = T0 + (y + 3) * TS + x + 3;
    const int v = p[0];
    int d[16];
    d[0] = v - p[3 * TS];       d[1] = v - p[3 * TS + 1];   d[2] = v - p[2 * TS + 2];   d[3] = v - p[1 * TS + 3];
    d[4] = v - p[3];            d[5] = v - p[-1 * TS + 3];  d[6] = v - p[-2 * TS + 2];  d[7] = v - p[-3 * TS + 1];
    d[8] = v - p[-3 * TS];      d[9] = v - p[-3 * TS - 1];  d[10] = v - p[-2 * TS - 2]; d[11] = v - p[-1 * TS - 3];
    d[12] = v - p[-3];          d[13] = v - p[1 * TS - 3];  d[14] = v - p[2 * TS - 2];  d[15] = v - p[3 * TS - 1];
    uint32_t dm = 0, bm = 0;   // darker: p_k < v - t <=> d > t ; brighter: d < -t
#pragma unroll
    for (int k = 0; k < 16; k++) {
        dm |= (d[k] > t_lo ? 1u : 0u) << k;
        bm |= (d[k] < -t_lo ? 1u : 0u) << k;
    }
    uint32_t m2 = dm | (dm << 16), r = m2 & (m2 >> 1);   // 9 contiguous set bits on the circular mask
    r &= r >> 2; r &= r >> 4; r &= m2 >> 8;
    uint32_t n2 = bm | (bm << 16), q = n2 & (n2 >> 1);
    q &= q >> 2; q &= q >> 4; q &= n2 >> 8;
    if (((r | q) & 0xFFFFu) == 0) return;
    // dark arcs: max over the 16 starts of min(d[s..s+8]); bright arcs: min over starts of max(...)
    int A = -256, B = 256;
    {
        int a2[16], a4[16];
#pragma unroll
        for (int k = 0; k < 16; k++) a2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) a4[k] = min(a2[k], a2[(k + 2) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) A = max(A, min(min(a4[k], a4[(k + 4) & 15]), d[(k + 8) & 15]));
    }
    {
        int b2[16], b4[16];
#pragma unroll
        for (int k = 0; k < 16; k++) b2[k] = max(d[k], d[(k + 1) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) b4[k] = max(b2[k], b2[(k + 2) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) B = min(B, max(max(b4[k], b4[(k + 4) & 15]), d[(k + 8) & 15]));
    }
    smap[(y + 1) * sms + x + 1] = (uint8_t)(max(A, -B) - 1);   // >= t_lo for a corner, <= 254
}

template <int TS, int TH, int ZS>
__global__ __launch_bounds__(FAST_THREADS) void k_fast_cells(OrbxPlan plan, OrbxWork wk, int l0_aligned)
{
    __shared__ __attribute__((aligned(16))) FastLds<TS, TH, ZS> lds[FAST_THREADS / 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cell = blockIdx.x * (FAST_THREADS / 64) + wave, f = blockIdx.y;
    if (cell >= plan.ncells) return;
    FastLds<TS, TH, ZS> &S = lds[wave];

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
    const int npx = zw * zh;
    const uint32_t rcp = (1u << 20) / (uint32_t)zw + 1u;   // idx / zw == (idx * rcp) >> 20 for idx < 4096, zw < 64

    // ---- tile -> LDS.  Rows start at a 4-byte aligned address of the level (levels >= 1 always;
    // level 0 when the caller's base/strides are 4-byte multiples), so whole dwords are moved. ----
    const uint8_t *img = L.base + (long long)f * L.frame_stride + (long long)iniY * L.stride;
    const int xoff = (l == 0 && !l0_aligned) ? 0 : (iniX & 3);
    if (l == 0 && !l0_aligned) {
        for (int i = lane; i < tw * th; i += 64) {
            const int y = i / tw, x = i - y * tw;
            S.tile[y * TS + x] = img[(long long)y * L.stride + iniX + x];
        }
    } else {
        constexpr int NDW = TS / 4, RPP = 64 / NDW;   // dwords per tile row, rows per pass
        const int ndw = (xoff + tw + 3) >> 2;
        const int r_in = lane / NDW, cdw = lane - r_in * NDW;
        const uint8_t *src = img + (iniX & ~3) + 4 * cdw;
        if (r_in < RPP && cdw < ndw) {
            for (int r = r_in; r < th; r += RPP)
                *reinterpret_cast<uint32_t *>(&S.tile[r * TS + 4 * cdw]) =
                    *reinterpret_cast<const uint32_t *>(src + (long long)r * L.stride);
        }
    }
    {   // zero the score map (1-px zero ring = "outside the cell counts as 0")
        uint32_t *z = reinterpret_cast<uint32_t *>(S.smap);
        const int nz = (sms * (zh + 2) + 3) >> 2;
        for (int i = lane; i < nz; i += 64) z[i] = 0;
    }
    WSYNC();
    const uint8_t *T0 = S.tile + xoff;   // tile origin (cell column 0)

    // ---- stages 1+2 ----
    int qn = 0;
    for (int base = 0; base < npx; base += 64) {
        const int idx = base + lane;
        bool pass = false;
        int y = 0, x = 0;
        if (idx < npx) {
            y = (int)(((uint32_t)idx * rcp) >> 20);
            x = idx - y * zw;
            const uint8_t *p = T0 + (y + 3) * TS + x + 3;
            const int v = p[0], lo = v - t_lo, hi = v + t_lo;
            int d = cls8(p[3 * TS], lo, hi) | cls8(p[-3 * TS], lo, hi);                 // pixels 0, 8
            d &= cls8(p[3], lo, hi) | cls8(p[-3], lo, hi);                               // 4, 12
            d &= cls8(p[2 * TS + 2], lo, hi) | cls8(p[-2 * TS - 2], lo, hi);             // 2, 10
            d &= cls8(p[-2 * TS + 2], lo, hi) | cls8(p[2 * TS - 2], lo, hi);             // 6, 14
            pass = d != 0;
        }
        const unsigned long long m = __ballot(pass);
        if (pass) S.queue[qn + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((y << 6) | x);
        qn += __popcll(m);
        if (qn >= 64) {
            WSYNC();
            fast_stage2<TS>(T0, S.queue, S.smap, 64, lane, t_lo, sms);
            // move the tail of the queue to the front
            const int rest = qn - 64;
            uint16_t tmpq = 0;
            if (lane < rest) tmpq = S.queue[64 + lane];
            WSYNC();
            if (lane < rest) S.queue[lane] = tmpq;
            qn = rest;
            WSYNC();
        }
    }
    WSYNC();
    fast_stage2<TS>(T0, S.queue, S.smap, qn, lane, t_lo, sms);   // the remaining survivors
    WSYNC();

    // ---- stage 3: NMS (strictly greater than all 8 neighbours; outside the cell zone counts as 0) ----
    int n_ini = 0, n_min = 0;
    const int niter = (npx + 63) >> 6;
    for (int it = 0; it < niter; it++) {
        const int idx = it * 64 + lane;
        bool ismax = false;
        int s = 0;
        if (idx < npx) {
            const int y = (int)(((uint32_t)idx * rcp) >> 20), x = idx - y * zw;
            const uint8_t *q = &S.smap[(y + 1) * sms + x + 1];
            s = q[0];
            if (s > 0)
                ismax = s > q[-1] && s > q[1] && s > q[-sms - 1] && s > q[-sms] && s > q[-sms + 1] &&
                        s > q[sms - 1] && s > q[sms] && s > q[sms + 1];
        }
        const unsigned long long mm = __ballot(ismax);
        if (lane == 0) S.masks[it] = mm;
        n_ini += __popcll(__ballot(ismax && s >= t_ini));
        n_min += __popcll(__ballot(ismax && s >= t_min));
    }
    // reference :811-818: FAST at iniThFAST; only if that yields nothing, FAST at minThFAST
    const int t_use = n_ini > 0 ? t_ini : t_min;
    const int total = n_ini > 0 ? n_ini : n_min;
    if (total == 0) return;
    int gbase = 0;
    if (lane == 0) gbase = (int)atomicAdd(&wk.cand_count[f * plan.nlevels + l], (uint32_t)total);
    gbase = __shfl(gbase, 0);
    WSYNC();
    OrbxCand *out = wk.cand + (long long)f * plan.cand_frame + L.cand_off;
    int written = 0;
    for (int it = 0; it < niter; it++) {
        const unsigned long long mm = S.masks[it];
        if (mm == 0) continue;
        const int idx = it * 64 + lane;
        bool emit = false;
        int y = 0, x = 0, s = 0;
        if ((mm >> lane) & 1ull) {
            y = (int)(((uint32_t)idx * rcp) >> 20);
            x = idx - y * zw;
            s = S.smap[(y + 1) * sms + x + 1];
            emit = s >= t_use;
        }
        const unsigned long long em = __ballot(emit);
        if (emit) {
            const int o = gbase + written + __popcll(em & ((1ull << lane) - 1ull));
            if (o < L.cand_cap) {
                OrbxCand cnd;
                cnd.xy = (uint32_t)(iniX + 3 + x) | ((uint32_t)(iniY + 3 + y) << 16);
                cnd.resp = (uint32_t)s;
                out[o] = cnd;
            } else {
                atomicOr(&wk.errflags[f], (uint32_t)ERRF_CAND_OVERFLOW);
            }
        }
        written += __popcll(em);
    }
}

void orbx_launch_fast(const OrbxPlan &plan, const OrbxWork &wk, int nframes, hipStream_t s)
{
    if (plan.ncells <= 0) return;
    int maxcell = 0;
    for (int l = 0; l < plan.nlevels; l++)
        if (plan.lv[l].nCols > 0) maxcell = max(maxcell, max(plan.lv[l].wCell, plan.lv[l].hCell));
    const OrbxLevel &L0 = plan.lv[0];
    const int l0_aligned = (((uintptr_t)L0.base | (uintptr_t)L0.stride | (uintptr_t)L0.frame_stride) & 3) == 0;
    dim3 grid((plan.ncells + FAST_THREADS / 64 - 1) / (FAST_THREADS / 64), nframes);
    if (maxcell <= 38)   // tile <= 44x44, zone <= 38x38
        hipLaunchKernelGGL((k_fast_cells<48, 44, 40>), grid, dim3(FAST_THREADS), 0, s, plan, wk, l0_aligned);
    else                 // cells of tiny levels: tile <= 66x66, zone <= 60x60
        hipLaunchKernelGGL((k_fast_cells<72, 66, 64>), grid, dim3(FAST_THREADS), 0, s, plan, wk, l0_aligned);
}
