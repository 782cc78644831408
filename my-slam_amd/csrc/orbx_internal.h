// orbx_internal.h -- shared between the host-side planner (orbx_capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/orbx.h"

#define ORBX_EDGE 19          // EDGE_THRESHOLD, src/ORBextractor.cc:76
#define ORBX_MINB 16          // minBorderX/Y = EDGE_THRESHOLD-3, src/ORBextractor.cc:775
#define ORBX_HALF_PATCH 15    // src/ORBextractor.cc:75

// FAST: one wave per cell, 4 waves per workgroup (cell zone <= 59x59: wCell = ceil(width/floor(width/30)) < 60)
#ifndef FAST_THREADS
#define FAST_THREADS 64
#endif

#define ORBX_CNT_STRIDE 32
#define ORBX_CNT(wk, plan, f, l) ((wk).cand_count[((f) * (plan).nlevels + (l)) * ORBX_CNT_STRIDE])

#ifndef OCT_THREADS
#define OCT_THREADS 512

#ifdef __HIPCC__
// base + number of set bits of mask below this lane: v_mbcnt_lo + v_mbcnt_hi (two instructions, base folded in)
__device__ __forceinline__ int orbx_prefix_cnt(unsigned long long mask, int base)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, (uint32_t)base));
}
// Inclusive prefix sum over the 64 lanes in six DPP additions (row shifts 1, 2, 4, 8, then the row totals carried across with
// row_bcast:15 / row_bcast:31) -- registers only; __shfl_up() is a ds_bpermute round trip through the LDS unit per step.
__device__ __forceinline__ int orbx_wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1, lanes without a source add 0
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // every row of 16 holds its own inclusive scan
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
    return v;
}
// value of the previous lane (lane 0: 0): DPP wave_shr:1
__device__ __forceinline__ int orbx_lane_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
#endif

// -DORBX_TRACE: per-phase shader-clock totals summed over all waves of a kernel (tools/dbg/phase_trace.py); slot 7 counts waves.
#ifdef ORBX_TRACE
#define ORBX_TRACE_DEFINE(SYM, FN)                                                                                      \
    __device__ unsigned long long SYM[256 * 16];   /* 256 copies, 128 B apart: the flush must not serialise in L2 */   \
    extern "C" int FN(unsigned long long *out, int reset)                                                               \
    {                                                                                                                   \
        static unsigned long long h[256 * 16];                                                                          \
        if (reset) { for (int i = 0; i < 256 * 16; i++) h[i] = 0; return (int)hipMemcpyToSymbol(HIP_SYMBOL(SYM), h, sizeof(h)); }           \
        const int rc = (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(SYM), sizeof(h));                                         \
        for (int i = 0; i < 8; i++) { out[i] = 0; for (int c = 0; c < 256; c++) out[i] += h[c * 16 + i]; }              \
        return rc;                                                                                                      \
    }
#define ORBX_TRACE_DECL unsigned long long ft_last = __builtin_readcyclecounter(), ft_acc[6] = {0, 0, 0, 0, 0, 0}
#define ORBX_TRACE_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); ft_acc[i] += n_ - ft_last; ft_last = n_; } while (0)
#define ORBX_TRACE_FLUSH(SYM) do { if (lane == 0) { unsigned long long *t_ = &SYM[((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255) * 16]; for (int i_ = 0; i_ < 6; i_++) atomicAdd(&t_[i_], ft_acc[i_]); atomicAdd(&t_[7], 1ull); } } while (0)
#else
#define ORBX_TRACE_DEFINE(SYM, FN)
#define ORBX_TRACE_DECL
#define ORBX_TRACE_STAMP(i) do { } while (0)
#define ORBX_TRACE_FLUSH(SYM) do { } while (0)
#endif

#endif
#define OCT_ID_MASK 0x3FFFFFFFu

#define DESC_R 18             // rotated rBRIEF sample radius (max |p| = 18.38)
#define DESC_BL 37            // blurred tile edge  (2*18+1)
#define DESC_RAW 43           // raw tile edge      (37 + 2*3 for the 7x7 blur)

enum { ERRF_CAND_OVERFLOW = 1, ERRF_TREE_OVERFLOW = 2 };

// One FAST candidate: xy = x | y << 16 (absolute level-interior coordinates), resp = corner score
struct __attribute__((aligned(8))) OrbxCand { uint32_t xy; uint32_t resp; };

// Quadtree node in the global arena (DivideNode boxes, relative to (minBorderX,minBorderY))
struct __attribute__((aligned(16))) OrbxNode {
    int16_t x0, y0, x1, y1;   // UL.x, UL.y, UR.x, BR.y
    int32_t count;            // vKeys.size()
    int32_t slot;             // per-pass slot (alive nodes) / final list position
};

struct OrbxLevel {
    int w, h, stride;             // interior size and row stride in bytes
    long long frame_stride;       // bytes between frames of this level
    uint8_t *base;                // frame 0 of this level (level 0: the caller's image)
    int maxBX, maxBY;             // w-16, h-16
    int nCols, nRows, wCell, hCell;
    uint32_t rcpW, rcpH;          // floor(2^32 / wCell) + 1, same for hCell (0 when the cell size is 1): n / cell = umulhi(n, rcp), n < 2^16
    int cell_begin;               // index of this level's first cell in the per-frame cell list
    int quota;                    // mnFeaturesPerLevel[level]
    int nIni;                     // DistributeOctTree root count
    int fastD;                    // quadtree: depth to which the early passes run on a key histogram instead of key loops (0 = off)
    float hX;                     // root width
    int cand_cap;  long long cand_off;    // per-frame candidate capacity / element offset
    int list_cap;  long long list_off;    // quadtree list capacity / element offset (sel buffer)
    int arena_cap; long long arena_off;   // node arena capacity / element offset
    float scale;                  // mvScaleFactor[level]
    float kp_size;                // (float)(int)(31*scale)
};

struct OrbxPlan {
    int nlevels;
    int ini_th, min_th;
    int ncells;                   // cells per frame over all levels
    int blur_mode;
    int oct_ft;                   // quadtree fast-forward table entries (max over levels of nIni * (4^(fastD+1) - 1) / 3)
    int oct_map;                  // quadtree coordinate -> path-bits tables: u16 entries (max over levels of box width + box height, padded)
    int oct_cap_max;              // largest list capacity the launch's LDS is sized for (the handle's maximum shape)
    int oct_big;                  // some level is 1080p-class (tens of thousands of candidates): quadtree workgroups of 1024 threads
    int out_cap;                  // per-frame output capacity
    long long cand_frame;         // candidates per frame (elements)
    long long list_frame;         // sel entries per frame
    long long arena_frame;        // arena nodes per frame
    const uint32_t *cell_tab;     // [ncells] level | cell row << 4 | cell column << 16 (device)
    OrbxLevel lv[ORBX_MAX_LEVELS];
};

struct OrbxWork {                 // device workspace pointers (per handle)
    OrbxCand *cand;               // [B][cand_frame]
    uint32_t *cand_count;         // [B][L] counters, ORBX_CNT_STRIDE dwords apart (one 128-B line each:
                                  // atomics that share a line serialise in the L2 channel)
    uint32_t *owner;              // [B][cand_frame]  quadtree: key -> node id (| quadrant << 30)
    OrbxNode *arena;              // [B][arena_frame]
    OrbxCand *sel;                // [B][list_frame]  selected keypoints per level, list order
    uint32_t *ncand;              // [B][L]           candidates per level of the last call (copy for the debug tap)
    uint32_t *nk;                 // [B][L]           selected per level
    uint32_t *errflags;           // [B]
};

struct ResizeTab {                // per destination level
    const int *xofs; const short2 *alpha; const int *yofs; const short2 *beta;
};

// Several pyramid levels in one launch (orbx_pyramid.hip, k_resize_fused): levels a+1 .. b from level a, one workgroup per
// (band of rows of level b, frame).  bands[band * (b - a + 1) + (l - a)] = rows of level l the band owns (x, y) and needs (z, w).
#define ORBX_FUSE_MAX 12
#define ORBX_FUSE_YTAB 1024      // rows a band needs, summed over its fused levels (checked at plan time)
struct FuseLevel { uint8_t *base; int w, h, stride; long long frame; ResizeTab tab; int nbx; uint32_t rcp_nbx; int pitch; };
struct FuseArgs { FuseLevel lv[ORBX_FUSE_MAX]; int a, b; const int4 *bands; int nbands; int buf0_bytes; };
void orbx_launch_resize_fused(const FuseArgs &A, int nframes, size_t lds_bytes, hipStream_t s);

// Several pyramid levels in one launch, one WAVE per 2-D tile of level b (orbx_pyramid.hip, k_resize_tiles).  Per axis and per
// (tile, level l = a .. b): (x, y) = the range of level l this tile writes to memory, (z, w) = the range it computes (multiples of 4).
#define ORBX_TILE_SLACK 16       // bytes behind a level's region in LDS (the 12-byte source windows of its last row read past it)
struct TileLevel { uint8_t *base; int w, h, stride; long long frame; ResizeTab tab; };
struct TileArgs { TileLevel lv[ORBX_FUSE_MAX]; int lds_off[ORBX_FUSE_MAX], tab_off[ORBX_FUSE_MAX]; int a, b, ntx, nty; const int4 *xr, *yr; };
void orbx_launch_resize_tiles(const TileArgs &A, int nframes, size_t lds_bytes, hipStream_t s);

// ---- launchers (orbx_pyramid.hip, orbx_fast.hip, orbx_octree.hip, orbx_describe.hip) ----
enum { RESIZE_FAST = 0, RESIZE_AREA2 = 1, RESIZE_GENERIC = 2, RESIZE_FAST6 = 3 };   // FAST6: FAST, and a 4-row block touches <= 6 source rows
static inline bool resize_is_fast(int mode) { return mode == RESIZE_FAST || mode == RESIZE_FAST6; }
// src_end != NULL: the source is caller-owned memory; one past its last valid byte (fast path guard)
void orbx_launch_resize(const OrbxLevel &src, const OrbxLevel &dst, const ResizeTab &tab, int mode,
                        int nframes, const uint8_t *src_end, hipStream_t s);
// cells [cell_lo, cell_hi) of the per-frame cell list (level 0 comes first)
void orbx_launch_fast(const OrbxPlan &plan, const OrbxWork &wk, int nframes, int cell_lo, int cell_hi, hipStream_t s);
void orbx_launch_octree(const OrbxPlan &plan, const OrbxWork &wk, int nframes, size_t lds_bytes,
                        hipStream_t s);
void orbx_launch_describe(const OrbxPlan &plan, const OrbxWork &wk, int nframes,
                          orbx_keypoint *d_kps, uint8_t *d_desc, int32_t *d_counts,
                          int32_t *d_status, hipStream_t s);
size_t orbx_octree_lds_bytes(int list_cap_max, int ft_entries, int map_entries);
int orbx_upload_constants(const int umax[16], const int gauss_k[7]);
int orbx_selftest_fp16(void);   // 0 = fp16 subnormal arithmetic behaves as k_fast_cells needs
