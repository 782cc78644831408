// orbm_internal.h -- shared by orbm.hip and orbm_grid.hip
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>
#include <hip/hip_runtime.h>
#include "../../include/orbm.h"

#define M_THREADS 256

int mfail(int code, const char *fmt, ...);
#define MHIPCHK(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return mfail(ORBX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

__device__ __forceinline__ int hamming256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    int d = __popc(a0.x ^ b0.x);
    d += __popc(a0.y ^ b0.y);
    d += __popc(a0.z ^ b0.z);
    d += __popc(a0.w ^ b0.w);
    d += __popc(a1.x ^ b1.x);
    d += __popc(a1.y ^ b1.y);
    d += __popc(a1.z ^ b1.z);
    d += __popc(a1.w ^ b1.w);
    return d;
}

#define ORBM_GRID_COLS 64   // FRAME_GRID_COLS, include/Frame.h:38
#define ORBM_GRID_ROWS 48   // FRAME_GRID_ROWS, include/Frame.h:37
#define ORBM_GRID_CELLS (ORBM_GRID_COLS * ORBM_GRID_ROWS)

struct OrbmGrid {               // device-resident Frame grid of the train frame
    float min_x, min_y, inv_w, inv_h;   // PosInGrid's origin and cell sizes (Frame::mnMinX / mnMinY, mfGridElementWidthInv / HeightInv)
    float qmin_x, qmin_y;               // GetFeaturesInArea's origin: the same for a Frame; a KeyFrame subtracts its own int mnMinX / mnMinY
    int n;
    float *kx, *ky; int32_t *koct;      // SoA copy of the undistorted keypoints
    int32_t *cell_start;                // [ORBM_GRID_CELLS + 1]
    int32_t *items;                     // [n] keypoint indices, push_back order inside a cell
    int32_t *cell_of;                   // [n] scratch
};

#ifdef __HIPCC__
// cell range of a window, src/Frame.cc:332-346.  Returns false when the window misses the grid.
__device__ __forceinline__ bool window_cells(const OrbmGrid &g, float x, float y, float r,
                                             int &cx0, int &cx1, int &cy0, int &cy1)
{
    cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, g.qmin_x), r), g.inv_w)));
    if (cx0 >= ORBM_GRID_COLS) return false;
    cx1 = min(ORBM_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, g.qmin_x), r), g.inv_w)));
    if (cx1 < 0) return false;
    cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, g.qmin_y), r), g.inv_h)));
    if (cy0 >= ORBM_GRID_ROWS) return false;
    cy1 = min(ORBM_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, g.qmin_y), r), g.inv_h)));
    if (cy1 < 0) return false;
    return true;
}

__device__ __forceinline__ bool in_window(const OrbmGrid &g, int i, float x, float y, float r, int minl, int maxl)
{
    if ((minl > 0) || (maxl >= 0)) {                       // bCheckLevels :348
        const int oct = g.koct[i];
        if (oct < minl) return false;
        if (maxl >= 0 && oct > maxl) return false;
    }
    return fabsf(__fsub_rn(g.kx[i], x)) < r && fabsf(__fsub_rn(g.ky[i], y)) < r;   // :368-372
}

// wave-wide minimum, every lane gets it: butterfly inside each row of 16 lanes with DPP (quad swaps, half-row and row mirror),
// then the four row results through v_readlane.  ~10 instructions; six ds_bpermute steps are several hundred cycles.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false));    // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xf, 0xf, false));    // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}

#endif

struct orbm_matcher {
    int device = 0, max_q = 0, max_t = 0, max_pairs = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_q = nullptr, *d_t = nullptr;
    int32_t *d_off = nullptr, *d_idx = nullptr, *d_out = nullptr;   // d_out: max(3*max_q, max_pairs) ints
    uint2 *d_part = nullptr; size_t part_elems = 0;                  // train-split partials (lazy)
    int dense_popcount = 0;                                          // ORBM_DENSE=popcount: the VALU kernel instead of the matrix cores (A/B record)
    OrbmGrid grid = {};  bool grid_ok = false;                       // N1: Frame grid of the last orbm_grid_build
    OrbmGrid grid2 = {}; bool grid2_ok = false;                      // second slot: orbm_search_by_sim3 searches two key frames
    float *d_qf = nullptr; int32_t *d_qi = nullptr; uint8_t *d_skip = nullptr;   // window-query staging (lazy)
    size_t qf_elems = 0;
    uint8_t *h_pin = nullptr; size_t h_pin_bytes = 0;   // pinned staging of orbm_search_by_bow (lazy)
    // pinned bump arena for the host-buffer entry points: pageable hipMemcpyAsync is a staged, synchronous copy of tens of
    // microseconds each; through pinned memory the copies of one call queue up behind each other and cost one round trip
    uint8_t *arena = nullptr; size_t arena_cap = 0, arena_used = 0, arena_want = 0;
    uint8_t *d_arena = nullptr;     // device mirror of the arena: inputs staged with orbm_stage_in() go up in ONE copy
    struct Pend { void *dst; const void *src; size_t bytes; };
    Pend pend[8]; int npend = 0;
};
// workspace growth (orbm.hip): the reference's matcher has no size limit, so entry points grow the handle instead of refusing
int orbm_grow(orbm_matcher *m, long long need_q, long long need_t, long long need_pairs);
void orbm_grid_free(OrbmGrid &g);
// orbm_grid.hip: builds a grid slot (asynchronous on the handle's stream unless it had to allocate a staging block), and the
// windows + candidate distances pass the host-scanned matchers share (against grid slot g: NULL = m->grid)
int orbm_grid_build_into(orbm_matcher *m, OrbmGrid &g, const orbx_keypoint *kps_un, int n, float assign_min_x, float assign_min_y,
                         float inv_w, float inv_h, float query_min_x, float query_min_y);
#ifdef __cplusplus
#include <vector>
int orbm_area_pairs(orbm_matcher *m, const float *x, const float *y, const float *r, const int32_t *mn, const int32_t *mx, int nq,
                    const uint8_t *qdesc, const uint8_t *train_desc, int n_train,
                    std::vector<int32_t> &off, std::vector<int32_t> &idx, std::vector<int32_t> &dist);
#endif
int orbm_arena_begin(orbm_matcher *m);                                                    // start of a host-API call
int orbm_h2d(orbm_matcher *m, void *dev, const void *host, size_t bytes, hipStream_t s);   // staged host -> device copy
int orbm_d2h(orbm_matcher *m, void *host, const void *dev, size_t bytes, hipStream_t s);   // staged; lands in host at orbm_sync()
int orbm_sync(orbm_matcher *m, hipStream_t s);
// orbm_mfma.hip: dense best / second-best partials on the matrix cores (same partial format as k_best2_dense)
int orbm_mfma_splits(int nq_cap, int nt_cap, int nbatch);
int orbm_launch_dense_mfma(orbm_matcher *m, const uint8_t *d_q, const int32_t *d_nq, int nq_fixed, const uint8_t *d_t, const int32_t *d_nt,
                           int nt_fixed, long long qstride, long long tstride, int cap_q, int cap_t, int nbatch, int out_stride, int S,
                           uint2 *part, hipStream_t s);
// k_dist_csr (orbm.hip) for callers in other files: dist[c] of every CSR candidate, off has nq + 1 entries
void orbm_launch_dist_csr(const uint8_t *d_q, int nq, const uint8_t *d_t, const int32_t *d_off, const int32_t *d_idx, int total,
                          int32_t *d_dist, hipStream_t s);
// Staged input: copies into the pinned arena and returns where it will be in the device mirror after orbm_flush_in()
// (every hipMemcpyAsync costs ~7 us of host time, so the inputs of one call travel together).  NULL: no room this call.
void *orbm_stage_in(orbm_matcher *m, const void *host, size_t bytes);
int orbm_flush_in(orbm_matcher *m, size_t from, hipStream_t s);                            // uploads arena[from, used)
// device -> pinned arena; returns where the bytes are after orbm_sync() (NULL: no room this call)
void *orbm_d2h_tmp(orbm_matcher *m, const void *dev, size_t bytes, hipStream_t s);
// one device block -> up to four host arrays, one copy (parts[i] bytes each, consecutive in the block)
int orbm_d2h_split(orbm_matcher *m, void *const *host, const size_t *parts, int nparts, const void *dev, hipStream_t s);                                            // synchronise + deliver the D2H copies

