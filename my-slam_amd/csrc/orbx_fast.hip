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

#define FAST_PADL 4      // left pad (bytes) of every tile row so that dword g-1 exists for every group
#ifndef FAST_CLIST
#define FAST_CLIST 224   // corners listed per cell before NMS falls back to scanning the whole score map
#endif

template <int TS, int TH, int ZS>   // tile row stride (bytes, multiple of 4), tile rows, score-map stride/rows
struct FastLds {
    uint8_t tile[TS * TH];
    uint8_t smap[ZS * ZS];
    uint16_t queue[384];   // < 128 pending + <= 256 appended per stage-1 step (entries 0..382); [383] = trash slot.
                           // Dead once stage 2 has drained it: the NMS ballot masks ((ZS * ZS + 63) / 64 x 8 B) reuse it.
    uint16_t clist[FAST_CLIST];
};
// 52-byte tile rows, 384-entry queue and 224-entry corner list: 5104 B per wave, 20416 B per workgroup -> 8 workgroups
// (32 waves) per CU by LDS instead of 7.

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ uint32_t as_u32(us2 v) { return __builtin_bit_cast(uint32_t, v); }

// Stage-1 reject for two pixels at once (u16 lanes).  A 9-arc holds one pixel of each opposite pair and all its pixels are
// on one side, so a corner needs  v - t > max_k min(p_k, p_k+8)  (every pair has a darker member) or
// v + t < min_k max(p_k, p_k+8).  Returns non-zero u16 lanes where it holds.
// Pairs (0,8) and (4,12) only: testing all four even pairs passes ~20 % fewer pixels to stage 2 but costs twice as much;
// measured slower overall on MI355X (the kernel is VALU-bound).
template <bool SCALED>   // SCALED: the values are bytes << 8; c + t can pass 65535 and must saturate (no pixel is brighter than that)
__device__ __forceinline__ uint32_t reject4(us2 c, us2 t, us2 a0, us2 a8, us2 a4, us2 a12)
{
    const us2 mlo = __builtin_elementwise_max(__builtin_elementwise_min(a0, a8), __builtin_elementwise_min(a4, a12));
    const us2 mhi = __builtin_elementwise_min(__builtin_elementwise_max(a0, a8), __builtin_elementwise_max(a4, a12));
    const us2 dark = __builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(c, t), mlo);
    const us2 bright = __builtin_elementwise_sub_sat(mhi, SCALED ? __builtin_elementwise_add_sat(c, t) : c + t);
    return as_u32(dark) | as_u32(bright);
}

// stage 2 for up to 128 queued survivors, TWO pixels per lane in the two fp16 halves of every register:
// corner score = max over the 16 nine-arcs of min(d) / min(-d), minus 1 (d = centre - circle pixel).
// A pixel is a corner at threshold t  <=>  score >= t, so no separate arc test is needed.
// The values are small integers (|d| <= 255), exact in fp16, and gfx950 has three-input packed fp16 min/max
// (v_pk_minimum3_f16 / v_pk_maximum3_f16): a 9-window is min3 of three min3's, so one polarity costs
// 36 instructions (see below) instead of the 79 of a two-input tree.  A byte b in a 16-bit half IS the fp16 subnormal
// b * 2^-24 (gfx950 kernels run with fp16 subnormals enabled, .amdhsa_float_denorm_mode_16_64 3), differences and
// min/max of such values are exact, and a non-negative result read back as an integer is the value again: no
// conversion in either direction.
typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hh2 pkh(uint32_t lo, uint32_t hi) { return __builtin_bit_cast(hh2, lo | (hi << 16)); }
__device__ __forceinline__ hh2 hmin3(hh2 a, hh2 b, hh2 c) { return __builtin_elementwise_minimum(__builtin_elementwise_minimum(a, b), c); }
__device__ __forceinline__ hh2 hmax3(hh2 a, hh2 b, hh2 c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }

template <int TS, int ZS>
__device__ __forceinline__ void fast_stage2(const uint8_t *T0, const uint16_t *queue, uint8_t *smap, int cnt,
                                            int lane, int t_lo, bool *c0, bool *c1, int *pos0, int *pos1)
{
    *c0 = *c1 = false;
    if (lane >= cnt) return;
    const bool two = lane + 64 < cnt;
    const int pa = queue[lane], pb = two ? queue[lane + 64] : pa;
    *pos0 = pa; *pos1 = pb;
    const uint8_t *p = T0 + __umul24((uint32_t)(pa >> 6) + 3u, (uint32_t)TS) + (pa & 63) + 3;
    const uint8_t *q = T0 + __umul24((uint32_t)(pb >> 6) + 3u, (uint32_t)TS) + (pb & 63) + 3;
    const hh2 v = pkh(p[0], q[0]);
    hh2 d[16];
    d[0] = v - pkh(p[3 * TS], q[3 * TS]);            d[1] = v - pkh(p[3 * TS + 1], q[3 * TS + 1]);
    d[2] = v - pkh(p[2 * TS + 2], q[2 * TS + 2]);    d[3] = v - pkh(p[1 * TS + 3], q[1 * TS + 3]);
    d[4] = v - pkh(p[3], q[3]);                      d[5] = v - pkh(p[-1 * TS + 3], q[-1 * TS + 3]);
    d[6] = v - pkh(p[-2 * TS + 2], q[-2 * TS + 2]);  d[7] = v - pkh(p[-3 * TS + 1], q[-3 * TS + 1]);
    d[8] = v - pkh(p[-3 * TS], q[-3 * TS]);          d[9] = v - pkh(p[-3 * TS - 1], q[-3 * TS - 1]);
    d[10] = v - pkh(p[-2 * TS - 2], q[-2 * TS - 2]); d[11] = v - pkh(p[-1 * TS - 3], q[-1 * TS - 3]);
    d[12] = v - pkh(p[-3], q[-3]);                   d[13] = v - pkh(p[1 * TS - 3], q[1 * TS - 3]);
    d[14] = v - pkh(p[2 * TS - 2], q[2 * TS - 2]);   d[15] = v - pkh(p[3 * TS - 1], q[3 * TS - 1]);
    // With m2[j] = min(d[k], d[k+1]) for odd k = 2 j + 1, the 8-window starting at k is min(m2[j..j+3]); the two 9-arcs that
    // contain it add d[k-1] or d[k+8], and max(min(a, x), min(a, y)) = min(a, max(x, y)):
    //   A = max over odd k of min(window_k, max(d[k-1], d[k+8]))        (36 instructions; 40 as min3 of min3 over all 16 arcs)
    // and the same with min and max exchanged for the bright polarity.
    hh2 A, B;
    {
        hh2 m2[8], e[8], r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) m2[j] = __builtin_elementwise_minimum(d[2 * j + 1], d[(2 * j + 2) & 15]);
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = __builtin_elementwise_maximum(d[2 * j], d[(2 * j + 9) & 15]);
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = hmin3(hmin3(m2[j], m2[(j + 1) & 7], m2[(j + 2) & 7]), m2[(j + 3) & 7], e[j]);
        A = hmax3(hmax3(r[0], r[1], r[2]), hmax3(r[3], r[4], r[5]), __builtin_elementwise_maximum(r[6], r[7]));
    }
    {
        hh2 m2[8], e[8], r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) m2[j] = __builtin_elementwise_maximum(d[2 * j + 1], d[(2 * j + 2) & 15]);
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = __builtin_elementwise_minimum(d[2 * j], d[(2 * j + 9) & 15]);
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = hmax3(hmax3(m2[j], m2[(j + 1) & 7], m2[(j + 2) & 7]), m2[(j + 3) & 7], e[j]);
        B = hmin3(hmin3(r[0], r[1], r[2]), hmin3(r[3], r[4], r[5]), __builtin_elementwise_minimum(r[6], r[7]));
    }
    const hh2 zero = {(_Float16)0.0f, (_Float16)0.0f};
    const uint32_t scb = __builtin_bit_cast(uint32_t, hmax3(A, -B, zero));   // (score + 1) * 2^-24 per half, clamped at 0
    const int s0 = (int)(scb & 0xFFFFu) - 1, s1 = (int)(scb >> 16) - 1;
    if (s0 >= t_lo) { smap[__umul24((uint32_t)(pa >> 6) + 1u, (uint32_t)ZS) + (pa & 63) + 1] = (uint8_t)s0; *c0 = true; }
    if (two && s1 >= t_lo) { smap[__umul24((uint32_t)(pb >> 6) + 1u, (uint32_t)ZS) + (pb & 63) + 1] = (uint8_t)s1; *c1 = true; }
}

// (1 << 20) / n + 1 for n = 1..64 ([0] unused): division of small indices by a wave-uniform n without the
// ~25-instruction integer-division expansion
__constant__ uint32_t c_rcp20[65] = {
    0, 1048577, 524289, 349526, 262145, 209716, 174763, 149797, 131073, 116509, 104858, 95326, 87382, 80660, 74899, 69906,
    65537, 61681, 58255, 55189, 52429, 49933, 47663, 45591, 43691, 41944, 40330, 38837, 37450, 36158, 34953, 33826,
    32769, 31776, 30841, 29960, 29128, 28340, 27595, 26887, 26215, 25576, 24967, 24386, 23832, 23302, 22796, 22311,
    21846, 21400, 20972, 20561, 20165, 19785, 19419, 19066, 18725, 18397, 18079, 17773, 17477, 17190, 16913, 16645, 16385};

ORBX_TRACE_DEFINE(g_fast_trace, orbx_debug_fast_trace)
#define FT_DECL ORBX_TRACE_DECL
#define FT(i) ORBX_TRACE_STAMP(i)
#define FT_FLUSH ORBX_TRACE_FLUSH(g_fast_trace)

template <int TS, int TH, int ZS>
__global__ __launch_bounds__(FAST_THREADS) void k_fast_cells(OrbxPlan plan, OrbxWork wk, int l0_aligned, int cell_lo, int cell_hi,
                                                             int wg_per_frame, int nwg, uint32_t wg_rcp)
{
    __shared__ __attribute__((aligned(16))) FastLds<TS, TH, ZS> lds[FAST_THREADS / 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): give XCD x the x-th contiguous eighth
    // of the (frame, cell) list, so that neighbouring cells -- which share their 6-pixel halos -- meet in one L2.
    // The grid is padded to a multiple of 8, which makes the map a bijection.  Placement is for speed only.
    const int lb = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (lb >= nwg) return;
    // lb / wg_per_frame without the integer-division expansion: multiply by floor(2^32 / d) + 1, one correction step
    int f = wg_rcp ? (int)__umulhi((uint32_t)lb, wg_rcp) : lb;   // wg_rcp == 0: one workgroup per frame
    f -= (f * wg_per_frame > lb) ? 1 : 0;
    // A frame's cells are taken last level first: the cells of the small levels are the slow ones (corner-dense, often
    // both threshold passes), and a kernel that ends on them drains with most of the GPU idle.  Ending on the uniform
    // level-0 cells took 7 % off the launch.
    const int cell = cell_hi - 1 - ((lb - f * wg_per_frame) * (FAST_THREADS / 64) + wave);
    if (cell < cell_lo) return;
    FastLds<TS, TH, ZS> &S = lds[wave];
    FT_DECL;

    const uint32_t ce = __builtin_amdgcn_readfirstlane(plan.cell_tab[cell]);   // one scalar load instead of a level search and a division
    const int l = (int)(ce & 15u), ci = (int)((ce >> 4) & 0xFFFu), cj = (int)(ce >> 16);
    const OrbxLevel &L = plan.lv[l];
    const int iniX = ORBX_MINB + cj * L.wCell, iniY = ORBX_MINB + ci * L.hCell;
    const int tw = min(iniX + L.wCell + 6, L.maxBX) - iniX;
    const int th = min(iniY + L.hCell + 6, L.maxBY) - iniY;
    const int zw = tw - 6, zh = th - 6;
    if (zw <= 0 || zh <= 0) return;
    const int npx = zw * zh;
    const uint32_t rcp = c_rcp20[zw];   // (1 << 20) / zw + 1:  idx / zw == (idx * rcp) >> 20 for idx < 4096, zw < 64

    // ---- tile -> LDS.  Rows start at a 4-byte aligned address of the level (levels >= 1 always;
    // level 0 when the caller's base/strides are 4-byte multiples), so whole dwords are moved. ----
    const uint8_t *img = L.base + (long long)f * L.frame_stride + (long long)iniY * L.stride;
    // Every tile is stored with its column 0 at LDS byte FAST_PADL + 1 of the row, whatever the alignment of iniX in
    // memory: zone column 0 (tile column 3) then starts a dword, a zone row is exactly ceil(zw / 4) stage-1 groups
    // (8 for the usual 31..32-px cells instead of 9..10) and no group hangs over the left edge.  The re-alignment is
    // one v_alignbyte_b32 per dword moved.
    // Rows of levels >= 1 (and of level 0 when the caller's base and strides are multiples of 4) start on a dword, so the
    // byte shift is the same for every row; a level 0 with an odd pitch (e.g. 1241-px rows handed over as they are) gets
    // its shift per row.  Either way whole dwords move; the over-read of <= 7 bytes stays inside the image row (>= 10 bytes
    // follow a tile).
    {
        constexpr int NDW = (TH + 7) / 4, RPP = 64 / NDW;   // payload dwords per tile row (LDS dwords 1 .. NDW), rows per pass
        static_assert(TS % 4 == 0 && TS >= 4 + 4 * NDW, "tile row too short");
        const int ndw = (tw + 4) >> 2;                    // LDS dwords 1 .. ndw hold tile columns -1 .. tw-1
        static_assert(NDW == 12 || NDW == 13 || NDW == 18, "lane / NDW below is written for these");
        const int r_in = NDW == 12 ? (int)(__umul24((uint32_t)lane, 43u) >> 9)
                       : NDW == 13 ? (int)(__umul24((uint32_t)lane, 79u) >> 10)
                                   : (int)(__umul24((uint32_t)lane, 57u) >> 10);   // lane / NDW, lane < 64
        const int cdw = lane - r_in * NDW;
        constexpr int NPASS = (TH + RPP - 1) / RPP;
        if (l == 0 && !l0_aligned) {
            const uint8_t *src = img + (iniX - 1) + 4 * cdw;
            if (r_in < RPP && cdw < ndw) {
                uint32_t lo[NPASS], hi[NPASS], shv[NPASS];
#pragma unroll
                for (int i = 0; i < NPASS; i++) {
                    const int r = r_in + i * RPP;
                    lo[i] = hi[i] = shv[i] = 0u;
                    if (r < th) {
                        const uintptr_t pa = reinterpret_cast<uintptr_t>(src + (long long)r * L.stride);
                        const uint32_t *g2 = reinterpret_cast<const uint32_t *>(pa & ~(uintptr_t)3);
                        lo[i] = g2[0]; hi[i] = g2[1]; shv[i] = (uint32_t)pa & 3u;
                    }
                }
#pragma unroll
                for (int i = 0; i < NPASS; i++) {
                    const int r = r_in + i * RPP;
                    if (r < th) *reinterpret_cast<uint32_t *>(&S.tile[r * TS + FAST_PADL + 4 * cdw]) = __builtin_amdgcn_alignbyte(hi[i], lo[i], shv[i]);
                }
            }
        } else if (NDW <= 13 && ((iniX - 1) & ~3) + (NDW == 13 ? 56 : 52) <= (l == 0 ? L.w : L.stride)) {
            // Rows start on a dword and 52 (56) bytes from the tile's aligned start lie inside the image row: 16 rows per pass, FOUR lanes
            // per row.  Lanes 0..2 of a row fetch source dwords 4j .. 4j + 3 in one 16-byte load, lane 3 the 16 bytes that end with the 13th (14th); the
            // dword behind a lane's four comes from its quad neighbour (one DPP move) and the lane funnel-shifts its four output dwords.
            // Three passes instead of nine, scalar row bases with a 32-bit lane offset (the nine 64-bit multiply-adds of the 12-lanes-
            // per-row form below are gone), all loads issued before the first is used: 55 -> 30 vector instructions for the tile.
            const uint32_t sh = (uint32_t)(iniX - 1) & 3u;
            const int rr = lane >> 2, j = lane & 3;
            typedef const __attribute__((address_space(1))) uint8_t *gp8;
            // lane 3's 16 bytes END with the row's 13th (14th) source dword: the same load instruction for every lane, no byte past the 52 (56)
            constexpr int L3OFF = NDW == 13 ? 40 : 36;
            const uint32_t coff = (uint32_t)(j < 3 ? 16 * j : L3OFF);
            constexpr int NP3 = (TH + 15) / 16;
            typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
            typedef u32x4v __attribute__((aligned(4))) U4a4;
            // the tile's aligned first byte is wave-uniform: pinned to scalar registers (global_load saddr form, 32-bit lane offsets)
            const uint64_t b = reinterpret_cast<uint64_t>(img + ((iniX - 1) & ~3));
            // (the builtin returns int: without the unsigned temporaries the low word would be SIGN-extended into the high one)
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b), bhi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
            const gp8 rb = (gp8)(((uint64_t)bhi << 32) | blo);
            u32x4v gv[NP3];
#pragma unroll
            for (int p = 0; p < NP3; p++) {
                // every lane loads (rows past the tile re-read its last row: no divergent loads, nothing to merge afterwards)
                const uint32_t row = (uint32_t)min(16 * p + rr, th - 1);
                gv[p] = *(const __attribute__((address_space(1))) U4a4 *)(rb + (__umul24(row, (uint32_t)L.stride) + coff));
            }
            uint32_t *dst = reinterpret_cast<uint32_t *>(&S.tile[rr * TS + FAST_PADL + 16 * j]);
#pragma unroll
            for (int p = 0; p < NP3; p++) {
                // the first source dword a lane holds for its left neighbour: dword 12 of the row for lane 3
                const uint32_t first = j < 3 ? gv[p].x : (NDW == 13 ? gv[p].z : gv[p].w);
                const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)first, 0xF9, 0xf, 0xf, false);   // quad_perm:[1,2,3,3]
                if (16 * p + rr < th) {
                    uint32_t *d = dst + p * (16 * TS / 4);
                    if (j < 3) {
                        d[0] = __builtin_amdgcn_alignbyte(gv[p].y, gv[p].x, sh);
                        d[1] = __builtin_amdgcn_alignbyte(gv[p].z, gv[p].y, sh);
                        d[2] = __builtin_amdgcn_alignbyte(gv[p].w, gv[p].z, sh);
                        d[3] = __builtin_amdgcn_alignbyte(nx, gv[p].w, sh);
                    } else if (NDW == 13) {
                        d[0] = __builtin_amdgcn_alignbyte(gv[p].w, gv[p].z, sh);
                    }
                }
            }
        } else {
            // (cells at the right edge of a caller-owned level 0, and the 66-px tiles of tiny levels)
            // every pass's load is issued before the first one is used: one memory round trip for the tile instead of one per pass
            // (a loop of load / wait / store spent 8.5 k of a wave's 27.8 k clocks here, -DORBX_TRACE)
            const uint32_t sh = (uint32_t)(iniX - 1) & 3u;
            const uint8_t *src = img + ((iniX - 1) & ~3) + 4 * cdw;
            if (r_in < RPP && cdw < ndw) {
                uint32_t lo[NPASS], hi[NPASS];
#pragma unroll
                for (int i = 0; i < NPASS; i++) {
                    const int r = r_in + i * RPP;
                    lo[i] = hi[i] = 0u;
                    if (r < th) {
                        const uint32_t *g2 = reinterpret_cast<const uint32_t *>(src + (long long)r * L.stride);
                        lo[i] = g2[0]; hi[i] = g2[1];
                    }
                }
#pragma unroll
                for (int i = 0; i < NPASS; i++) {
                    const int r = r_in + i * RPP;
                    if (r < th) *reinterpret_cast<uint32_t *>(&S.tile[r * TS + FAST_PADL + 4 * cdw]) = __builtin_amdgcn_alignbyte(hi[i], lo[i], sh);
                }
            }
        }
    }
    FT(0);
    const uint8_t *T0 = S.tile + FAST_PADL + 1;      // tile origin (cell column 0)
    constexpr int cb = FAST_PADL + 1 + 3;            // tile byte column of zone column 0: a dword boundary
    static_assert((cb & 3) == 0, "zone column 0 must start a dword");
    constexpr int g0 = cb >> 2;
    const int ng = (zw + 3) >> 2;
    const uint32_t rcpg = (c_rcp20[ng] + 15u) >> 4;          // (1 << 16) / ng + 1 (or one more): t / ng for t < 1024, ng <= 16
    const int ntask = ng * zh;
    const int qlast = zw - 4 * (ng - 1);                     // pixels of a zone row's last group that are inside the zone (1..4)

    // The reference calls FAST at iniThFAST and, only when that cell yields nothing, again at
    // minThFAST (:811-818).  Same here: the first pass thresholds at iniThFAST; a cell with no NMS
    // survivor is redone at minThFAST (when that is lower -- a higher one cannot add anything).
    for (int pass = 0; pass < 2; pass++) {
        const int t_lo = pass == 0 ? plan.ini_th : plan.min_th;
        if (pass == 1 && plan.min_th >= plan.ini_th) { FT_FLUSH; return; }
        {   // zero the score map (1-px zero ring = "outside the cell counts as 0")
            // 16 bytes per lane and store (the map starts on a 16-byte boundary and its size is a multiple of 16): two passes for a 30-row cell
            static_assert((TS * TH) % 16 == 0 && (ZS * ZS) % 16 == 0 && sizeof(FastLds<TS, TH, ZS>) % 16 == 0, "score map alignment");
            uint4 *z = reinterpret_cast<uint4 *>(__builtin_assume_aligned(S.smap, 16));
            const int nz = (ZS * (zh + 2) + 15) >> 4;
            for (int i = lane; i < nz; i += 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        WSYNC();

        // ---- stages 1+2.  Stage-1 task = (zone row, dword group): 4 horizontally adjacent pixels ----
        const us2 tt = as_us2((uint32_t)t_lo * 0x00010001u), ttO = as_us2((uint32_t)t_lo * 0x01000100u);
        int qn = 0, ncl = 0;
        bool cl_over = false;
#define DRAIN(CNT)                                                                                                   \
    do {                                                                                                             \
        bool c0_, c1_; int p0_ = 0, p1_ = 0;                                                                         \
        fast_stage2<TS, ZS>(T0, S.queue, S.smap, (CNT), lane, t_lo, &c0_, &c1_, &p0_, &p1_);                         \
        const unsigned long long m0_ = __builtin_amdgcn_ballot_w64(c0_), m1_ = __builtin_amdgcn_ballot_w64(c1_);                                           \
        const int n0_ = __popcll(m0_), n1_ = __popcll(m1_);                                                          \
        if (ncl + n0_ + n1_ <= FAST_CLIST) {                                                                         \
            if (c0_) S.clist[orbx_prefix_cnt(m0_, ncl)] = (uint16_t)p0_;                         \
            if (c1_) S.clist[orbx_prefix_cnt(m1_, ncl + n0_)] = (uint16_t)p1_;                   \
            ncl += n0_ + n1_;                                                                                        \
        } else cl_over = true;                                                                                       \
    } while (0)
        for (int base = 0; base < ntask; base += 64) {
            // Which pixels of a step count is decided on the scalar unit: lanes past the last task (a bit-field mask) and, in
            // the last group of a zone row, the pixels past the zone's right edge (qlast of its 4 are inside) -- one vector
            // compare per step instead of five and a select.  Lanes past the last task compute on whatever LDS holds there.
            const int t = base + lane;
            const int zy = (int)(__umul24((uint32_t)t, rcpg) >> 16);
            const int gi = t - (int)__umul24((uint32_t)zy, (uint32_t)ng);
            const int g = g0 + gi;
            const int zx0 = 4 * gi;                                  // zone column of byte 0 of this group
            const int nvalid = min(ntask - base, 64);
            const unsigned long long VALID = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
            const unsigned long long NOTLAST = ~__builtin_amdgcn_ballot_w64(gi == ng - 1);
            uint32_t fE, fO;   // non-zero u16 halves = survivors: fE pixels (0, 2), fO pixels (1, 3)
            {
                const uint32_t *r0 = reinterpret_cast<const uint32_t *>(&S.tile[(zy + 3) * TS]) + g;
                const uint32_t dD = *(reinterpret_cast<const uint32_t *>(&S.tile[(zy + 6) * TS]) + g);   // pixel 0  (0,+3)
                const uint32_t dU = *(reinterpret_cast<const uint32_t *>(&S.tile[(zy + 0) * TS]) + g);   // pixel 8  (0,-3)
                const uint32_t cL = r0[-1], cC = r0[0], cR = r0[1];
                const uint32_t d4 = __builtin_amdgcn_alignbyte(cR, cC, 3);    // pixel 4  (+3, 0)
                const uint32_t d12 = __builtin_amdgcn_alignbyte(cC, cL, 1);   // pixel 12 (-3, 0)
                // Even bytes (pixels 0 and 2) and odd bytes (pixels 1 and 3, left scaled by 256: order, saturating
                // differences and the non-zero tests do not care) of a dword are two u16 pairs after ONE full-rate AND each
                // (a byte permute is a half-rate instruction).
#define EV(X) as_us2((X) & 0x00FF00FFu)
                fE = reject4<false>(EV(cC), tt, EV(dD), EV(dU), EV(d4), EV(d12));
                // The odd pixels are compared with the EVEN byte left in place underneath them (no AND): every u16 is odd << 8 | noise.
                // min / max pick by the high byte first, so the high byte of every intermediate is exact, and a saturating difference
                // of such values is non-zero whenever the difference of the high bytes is positive: no corner is lost.  When the high
                // bytes are EQUAL the noise can make the difference non-zero -- a pixel more for stage 2, whose score is exact.
                fO = reject4<true>(as_us2(cC), ttO, as_us2(dD), as_us2(dU), as_us2(d4), as_us2(d12));
#undef EV
            }
            // append the survivors of the 4 pixels: four ballots, one queue update (entry order is free)
            // (each ballot is taken straight from a compare and combined on the scalar unit; a ballot of a combined
            // predicate costs two more vector instructions)
            const bool f0 = (fE & 0x0000FFFFu) != 0, f1 = (fO & 0x0000FFFFu) != 0;
            const bool f2 = (fE & 0xFFFF0000u) != 0, f3 = (fO & 0xFFFF0000u) != 0;
#define BAL(P) __builtin_amdgcn_ballot_w64(P)
            const unsigned long long b0 = BAL(f0) & VALID, b1 = BAL(f1) & (qlast > 1 ? VALID : VALID & NOTLAST);
            const unsigned long long b2 = BAL(f2) & (qlast > 2 ? VALID : VALID & NOTLAST), b3 = BAL(f3) & (qlast > 3 ? VALID : VALID & NOTLAST);
#undef BAL
            // the lane's own bit of each (uniform) mask as a select operand: no vector instruction
            const bool k0 = __builtin_amdgcn_inverse_ballot_w64(b0), k1 = __builtin_amdgcn_inverse_ballot_w64(b1);
            const bool k2 = __builtin_amdgcn_inverse_ballot_w64(b2), k3 = __builtin_amdgcn_inverse_ballot_w64(b3);
            const int n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2);
            // queue in the zone's row-major pixel order: entry = survivors of lower lanes (all four pixels) + own lower pixels.
            // Neighbouring lanes of stage 2 then gather from neighbouring LDS addresses (bank conflicts: DESIGN.md §4).
            const int e0 = orbx_prefix_cnt(b3, orbx_prefix_cnt(b2, orbx_prefix_cnt(b1, orbx_prefix_cnt(b0, qn))));
            const int e1 = e0 + (k0 ? 1 : 0), e2 = e1 + (k1 ? 1 : 0), e3 = e2 + (k2 ? 1 : 0);
            const int pz = (zy << 6) + zx0;          // (zy << 6) | zx for every pixel with 0 <= zx < 64
            S.queue[k0 ? e0 : 383] = (uint16_t)pz;
            S.queue[k1 ? e1 : 383] = (uint16_t)(pz + 1);
            S.queue[k2 ? e2 : 383] = (uint16_t)(pz + 2);
            S.queue[k3 ? e3 : 383] = (uint16_t)(pz + 3);
            qn += n0 + n1 + n2 + __popcll(b3);
            while (qn >= 128) {
                WSYNC();
                FT(1);
                DRAIN(128);
                FT(2);
                const int rest = qn - 128;            // move the tail of the queue to the front
                WSYNC();
                for (int i0 = 0; i0 < rest; i0 += 64) {   // rest < 256: ascending blocks never overwrite unread entries
                    uint16_t tmpq = 0;
                    if (i0 + lane < rest) tmpq = S.queue[128 + i0 + lane];
                    WSYNC();
                    if (i0 + lane < rest) S.queue[i0 + lane] = tmpq;
                    WSYNC();
                }
                qn = rest;
            }
        }
        WSYNC();
        FT(1);
        DRAIN(qn);   // the remaining survivors (qn < 128)
#undef DRAIN
        WSYNC();
        FT(2);

        // ---- stage 3: NMS (strictly greater than all 8 neighbours; outside the cell zone counts as 0).
        // Items are the listed corners, or every zone pixel if the list overflowed. ----
        unsigned long long *nms_masks = reinterpret_cast<unsigned long long *>(S.queue);   // the queue is empty and idle from here on
        static_assert(sizeof(S.queue) >= ((ZS * ZS + 63) / 64) * 8 && (TS * TH + ZS * ZS) % 8 == 0, "mask alias");
        OrbxCand *out = wk.cand + (long long)f * plan.cand_frame + L.cand_off;
        // The corner list.  Usually the <= FAST_CLIST corners listed during stage 2.  When that list overflowed (a corner-dense cell: common
        // on the small pyramid levels), the corners are listed again from the score map -- every non-zero byte is one -- into the drained
        // queue and the old list together (FAST_BIG entries, contiguous), four pixels per lane and step like stage 1; that costs about a
        // third of the per-pixel NMS over the whole zone it replaces (which remains for cells with more than FAST_BIG corners).
        constexpr int FAST_BIG = 384 + FAST_CLIST - 1;   // the last entry is the trash slot of the compaction
        typedef FastLds<TS, TH, ZS> LdsT;
        static_assert(offsetof(LdsT, clist) == offsetof(LdsT, queue) + 384 * sizeof(uint16_t), "queue and list are contiguous");
        const uint16_t *list = S.clist;
        int nlist = ncl;
        bool scan_all = false;
        if (cl_over) {
            uint16_t *big = S.queue;
            constexpr int ND = ZS / 4;                              // dwords per score-map row
            static_assert(ZS % 4 == 0, "score-map rows are whole dwords");
            const uint32_t *sm32 = reinterpret_cast<const uint32_t *>(S.smap);
            const int nt2 = zh * ND;
            int cnt = 0;
            for (int base = 0; base < nt2; base += 64) {
                const int t = base + lane;
                const int row = t / ND, jd = t - row * ND;          // division by a constant
                const uint32_t w = t < nt2 ? sm32[(row + 1) * ND + jd] : 0u;   // bytes 4 jd .. 4 jd + 3 of map row row + 1: zone columns 4 jd - 1 ..
                const unsigned long long b0 = __builtin_amdgcn_ballot_w64((w & 0x000000FFu) != 0u), b1 = __builtin_amdgcn_ballot_w64((w & 0x0000FF00u) != 0u);
                const unsigned long long b2 = __builtin_amdgcn_ballot_w64((w & 0x00FF0000u) != 0u), b3 = __builtin_amdgcn_ballot_w64((w & 0xFF000000u) != 0u);
                const int n = __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
                if (cnt + n > FAST_BIG) { scan_all = true; break; }   // wave-uniform
                const bool k0 = __builtin_amdgcn_inverse_ballot_w64(b0), k1 = __builtin_amdgcn_inverse_ballot_w64(b1);
                const bool k2 = __builtin_amdgcn_inverse_ballot_w64(b2), k3 = __builtin_amdgcn_inverse_ballot_w64(b3);
                const int e0 = orbx_prefix_cnt(b3, orbx_prefix_cnt(b2, orbx_prefix_cnt(b1, orbx_prefix_cnt(b0, cnt))));
                const int e1 = e0 + (k0 ? 1 : 0), e2 = e1 + (k1 ? 1 : 0), e3 = e2 + (k2 ? 1 : 0);
                const int pz = (row << 6) + 4 * jd - 1;             // (y << 6) | x of byte 0 (x = -1 only for the zero ring's byte, never listed)
                big[k0 ? e0 : FAST_BIG] = (uint16_t)pz;
                big[k1 ? e1 : FAST_BIG] = (uint16_t)(pz + 1);
                big[k2 ? e2 : FAST_BIG] = (uint16_t)(pz + 2);
                big[k3 ? e3 : FAST_BIG] = (uint16_t)(pz + 3);
                cnt += n;
            }
            WSYNC();
            list = big;
            nlist = cnt;
        }
        if (!scan_all) {
            // <= 4 rounds of 64 corners at a time.  A round's position and score stay in registers and its mask of maxima in a scalar pair, so
            // that the output pass neither re-reads the list and the score map nor goes through LDS masks; one atomic per chunk of 256
            // reserves the output range (the order of a level's candidates is free: the quadtree sorts).
            constexpr int NR = 4;
            int total_all = 0;
            for (int c0 = 0; c0 < nlist; c0 += 64 * NR) {
                uint32_t cxy[NR], crs[NR];
                unsigned long long mk[NR];
                int total = 0;
#pragma unroll
                for (int it = 0; it < NR; it++) {
                    mk[it] = 0ull; cxy[it] = 0u; crs[it] = 0u;
                    if (c0 + it * 64 < nlist) {
                        const int idx = c0 + it * 64 + lane;
                        bool ismax = false;
                        if (idx < nlist) {
                            const int pos = list[idx];
                            const int y = pos >> 6, x = pos & 63;
                            const uint8_t *q = &S.smap[__umul24((uint32_t)y + 1u, (uint32_t)ZS) + x + 1];
                            const uint32_t s = q[0];
                            const uint32_t n0 = q[-1], n1 = q[1], n2 = q[-ZS - 1], n3 = q[-ZS], n4 = q[-ZS + 1], n5 = q[ZS - 1], n6 = q[ZS], n7 = q[ZS + 1];
                            const uint32_t nm = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
                            ismax = s > nm;      // s > nm >= 0 implies s > 0
                            cxy[it] = (uint32_t)(iniX + 3 + x) | ((uint32_t)(iniY + 3 + y) << 16);
                            crs[it] = s;
                        }
                        mk[it] = __builtin_amdgcn_ballot_w64(ismax);
                        total += __popcll(mk[it]);
                    }
                }
                if (total == 0) continue;
                total_all += total;
                int gbase = 0;
                if (lane == 0) gbase = (int)atomicAdd(&ORBX_CNT(wk, plan, f, l), (uint32_t)total);
                gbase = __shfl(gbase, 0);
                int written = 0;
#pragma unroll
                for (int it = 0; it < NR; it++) {
                    if (mk[it] == 0ull) continue;
                    if ((mk[it] >> lane) & 1ull) {
                        const int o = orbx_prefix_cnt(mk[it], gbase + written);
                        if (o < L.cand_cap) {
                            OrbxCand cnd;
                            cnd.xy = cxy[it];
                            cnd.resp = crs[it];
                            out[o] = cnd;
                        } else {
                            atomicOr(&wk.errflags[f], (uint32_t)ERRF_CAND_OVERFLOW);
                        }
                    }
                    written += __popcll(mk[it]);
                }
            }
            FT(3);
            if (total_all == 0) continue;   // nothing at this threshold: fall back to the lower one
            FT(4);
            FT_FLUSH;
            return;
        }
        // more corners than the big list holds (noise-like cells): every zone pixel is an item
        int total = 0;
        const int nitem = npx;
        const int niter = (nitem + 63) >> 6;
        for (int it = 0; it < niter; it++) {
            const int idx = it * 64 + lane;
            bool ismax = false;
            if (idx < nitem) {
                const int y = (int)(__umul24((uint32_t)idx, rcp) >> 20), x = idx - (int)__umul24((uint32_t)y, (uint32_t)zw);
                const uint8_t *q = &S.smap[__umul24((uint32_t)y + 1u, (uint32_t)ZS) + x + 1];
                // all nine reads at once and one comparison against the neighbours' maximum (v_max3_u32): a chain of &&
                // would turn into eight divergent branches
                const uint32_t s = q[0];
                const uint32_t n0 = q[-1], n1 = q[1], n2 = q[-ZS - 1], n3 = q[-ZS], n4 = q[-ZS + 1], n5 = q[ZS - 1], n6 = q[ZS], n7 = q[ZS + 1];
                const uint32_t nm = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
                ismax = s > nm;      // s > nm >= 0 implies s > 0
            }
            const unsigned long long mm = __builtin_amdgcn_ballot_w64(ismax);
            if (lane == 0) nms_masks[it] = mm;
            total += __popcll(mm);
        }
        FT(3);
        if (total == 0) continue;   // nothing at this threshold: fall back to the lower one

        int gbase = 0;
        if (lane == 0) gbase = (int)atomicAdd(&ORBX_CNT(wk, plan, f, l), (uint32_t)total);
        gbase = __shfl(gbase, 0);
        WSYNC();
        int written = 0;
        for (int it = 0; it < niter; it++) {
            const unsigned long long mm = nms_masks[it];
            if (mm == 0) continue;
            const int idx = it * 64 + lane;
            if ((mm >> lane) & 1ull) {
                const int y = (int)(__umul24((uint32_t)idx, rcp) >> 20), x = idx - (int)__umul24((uint32_t)y, (uint32_t)zw);
                const int o = orbx_prefix_cnt(mm, gbase + written);
                if (o < L.cand_cap) {
                    OrbxCand cnd;
                    cnd.xy = (uint32_t)(iniX + 3 + x) | ((uint32_t)(iniY + 3 + y) << 16);
                    cnd.resp = (uint32_t)S.smap[__umul24((uint32_t)y + 1u, (uint32_t)ZS) + x + 1];
                    out[o] = cnd;
                } else {
                    atomicOr(&wk.errflags[f], (uint32_t)ERRF_CAND_OVERFLOW);
                }
            }
            written += __popcll(mm);
        }
        FT(4);
        FT_FLUSH;
        return;
    }
    FT_FLUSH;
}

// The stage-2 score tree treats bytes as fp16 subnormals: this needs the kernels to run with fp16 subnormals enabled
// (the compiler's default, .amdhsa_float_denorm_mode_16_64 3).  Checked once per handle at create time, on the device:
// differences, three-input min/max and the integer read-back of a handful of byte pairs must be exact.
__global__ void k_fp16_probe(uint32_t *out)
{
    const uint32_t a = 200u + threadIdx.x, b = 3u, c = 255u, z = 0u;
    const hh2 d0 = pkh(a & 255u, b) - pkh(b, a & 255u);     // (a-b, b-a)
    const hh2 d1 = pkh(c, z) - pkh(z, c);                   // (255, -255)
    const hh2 d2 = pkh(b, b) - pkh(b, b);                   // (0, 0)
    const hh2 mn = hmin3(d0, d1, d2), mx = hmax3(d0, d1, d2);
    const hh2 zero = {(_Float16)0.0f, (_Float16)0.0f};
    out[2 * threadIdx.x] = __builtin_bit_cast(uint32_t, hmax3(mn, -mx, zero));
    out[2 * threadIdx.x + 1] = __builtin_bit_cast(uint32_t, hmax3(mx, -mn, zero));
}
int orbx_selftest_fp16(void)
{
    uint32_t *d = nullptr, h[16];
    if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_fp16_probe, dim3(1), dim3(8), 0, 0, d);
    const bool ok = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if (!ok) return -1;
    for (int t = 0; t < 8; t++) {
        const int a = (200 + t) & 255, b = 3;
        // lo halves: min3(a-b, 255, 0) = 0, max3 = 255; hi halves: min3(b-a, -255, 0) = -255, max3(b-a, -255, 0) = 0
        // out0 = max3(mn, -mx, 0)   = (max(0, -255, 0), max(-255, 0, 0))   = (0, 0)
        // out1 = max3(mx, -mn, 0)   = (max(255, 0, 0), max(0, 255, 0))     = (255, 255)
        (void)a; (void)b;
        if (h[2 * t] != 0u || h[2 * t + 1] != (255u | (255u << 16))) return 1;
    }
    return 0;
}

void orbx_launch_fast(const OrbxPlan &plan, const OrbxWork &wk, int nframes, int cell_lo, int cell_hi, hipStream_t s)
{
    cell_hi = min(cell_hi, plan.ncells);
    if (cell_hi <= cell_lo) return;
    int maxcell = 0;
    for (int l = 0; l < plan.nlevels; l++)
        if (plan.lv[l].nCols > 0) maxcell = max(maxcell, max(plan.lv[l].wCell, plan.lv[l].hCell));
    const OrbxLevel &L0 = plan.lv[0];
    const int l0_aligned = (((uintptr_t)L0.base | (uintptr_t)L0.stride | (uintptr_t)L0.frame_stride) & 3) == 0;
    const int wg_per_frame = (cell_hi - cell_lo + FAST_THREADS / 64 - 1) / (FAST_THREADS / 64);
    const int nwg = wg_per_frame * nframes;
    dim3 grid((nwg + 7) & ~7);
    if (maxcell <= 38)   // tile <= 44x44, zone <= 38x38; row = 4 pad + 1 + 44 + over-read (3) -> 52 B
        hipLaunchKernelGGL((k_fast_cells<52, 44, 40>), grid, dim3(FAST_THREADS), 0, s, plan, wk, l0_aligned, cell_lo, cell_hi, wg_per_frame, nwg, wg_per_frame > 1 ? (uint32_t)((1ull << 32) / (unsigned)wg_per_frame + 1) : 0u);
    else if (maxcell <= 42)   // a level whose detection zone is 90..120 px in one direction has cells of 39..42 (e.g. the flat upper
                              // levels of 1241x376): tile <= 48x48, 23.4 KB of LDS per workgroup instead of the 42 KB below
        hipLaunchKernelGGL((k_fast_cells<56, 48, 44>), grid, dim3(FAST_THREADS), 0, s, plan, wk, l0_aligned, cell_lo, cell_hi, wg_per_frame, nwg, wg_per_frame > 1 ? (uint32_t)((1ull << 32) / (unsigned)wg_per_frame + 1) : 0u);
    else                 // cells of tiny levels: tile <= 66x66, zone <= 60x60
        hipLaunchKernelGGL((k_fast_cells<80, 66, 64>), grid, dim3(FAST_THREADS), 0, s, plan, wk, l0_aligned, cell_lo, cell_hi, wg_per_frame, nwg, wg_per_frame > 1 ? (uint32_t)((1ull << 32) / (unsigned)wg_per_frame + 1) : 0u);
}
