// orbx_capi.hip -- host side of liborbx.so: constructor tables, shape planning, workspace, C ABI.
// The arithmetic here restates the reference constructor and OpenCV's resize planning on the host
// (citations: src/ORBextractor.cc of WChen09/My-SLAM); all pixel work is in the per-stage kernel files (orbx_pyramid/fast/octree/describe.hip).
#include <cfloat>
#include <chrono>
#include <cstdarg>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "orbx_internal.h"

#include "stage_pool.h"

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ORBX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char *orbx_last_error(void) { return g_err.c_str(); }
extern "C" const char *orbx_version(void) { return "orbx 0.1 (gfx950)"; }

static inline int cv_round(double v) { return (int)lrint(v); }   // cvRound: half to even
static inline int cv_floor(double v) { int i = (int)v; return i - (v < i); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (v > i); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define ORBX_MAX_SUB 4
struct orbx_extractor {
    int nfeatures = 0; float scale_factor = 0; int nlevels = 0, ini_th = 0, min_th = 0, device = 0;
    int max_w = 0, max_h = 0, max_batch = 0;
    float scale[ORBX_MAX_LEVELS], inv_scale[ORBX_MAX_LEVELS], sigma2[ORBX_MAX_LEVELS], inv_sigma2[ORBX_MAX_LEVELS];
    int quota[ORBX_MAX_LEVELS];
    int umax[16]; int gauss_k[7];
    int blur_mode = 0;
    bool need_clear = true;
    hipStream_t stream = nullptr;
    hipStream_t aux[ORBX_MAX_SUB - 1] = {}; hipEvent_t ev_fork = nullptr, ev_join[ORBX_MAX_SUB - 1] = {}; int nsub = 1; int overlap_pyr = 0;
    // current plan
    int cur_w = 0, cur_h = 0; int last_batch = 0;
    const uint8_t *last_input = nullptr; int last_in_stride = 0; long long last_in_frame = 0;
    uint8_t *h_pyr = nullptr; size_t h_pyr_bytes = 0;      // page-locked staging of orbx_download_pyramid (lazy)
    const uint8_t *pin_ptr = nullptr; int pin_n = 0; size_t pin_stride = 0, pin_bytes = 0; bool pin_is = false;   // last is_pinned_host() answer
    OrbxPlan plan; OrbxWork work; ResizeTab tabs[ORBX_MAX_LEVELS]; int area2[ORBX_MAX_LEVELS];
    // several pyramid levels per launch (k_resize_fused): one plan per band height (16 rows for batches, 8 for a few frames)
    struct FusePlan { bool ok = false; int a = 0, b = 0, nbands = 0, buf0 = 0, lds = 0, bh = 0; size_t off = 0; } fuse[2];
    int4 *d_bands = nullptr; size_t bands_cap = 0; int fuse_on = 1; int oct_fast = 1; int oct_cap_max = 0;
    // upper pyramid levels in one launch, one wave per 2-D tile (k_resize_tiles): levels tile.a + 1 .. nlevels - 1
    struct TilePlan { bool ok = false; int a = 0, b = 0, ntx = 0, nty = 0, lds = 0; int lds_off[ORBX_FUSE_MAX] = {}, tab_off[ORBX_FUSE_MAX] = {}; size_t offx = 0, offy = 0; } tile;
    int4 *d_tiles = nullptr; size_t tiles_cap = 0; int tile_a = 0, tile_w = 32, tile_h = 32, tile_min_frames = 8;   // off by default: measured slower than the launches it replaces
    size_t oct_lds = 0;
    // allocations (sized for the max shape)
    OrbxPlan max_plan; size_t pyr_bytes = 0; size_t pyr_level_off[ORBX_MAX_LEVELS];
    uint8_t *d_input = nullptr; int in_stride = 0; size_t in_frame = 0;
    uint8_t *d_pyr = nullptr;
    int *d_tab_i = nullptr; short2 *d_tab_s = nullptr; size_t tab_elems = 0;
    uint32_t *d_cells = nullptr; int cells_cap = 0;   // per-cell (level, row, column) table of the current plan
    orbx_keypoint *d_kps = nullptr; uint8_t *d_desc = nullptr; int32_t *d_counts = nullptr, *d_status = nullptr;
    uint8_t *h_in = nullptr;
    orbx_keypoint *h_kps = nullptr; uint8_t *h_desc = nullptr; int32_t *h_counts = nullptr, *h_status = nullptr;
    uint8_t *d_out = nullptr, *h_out = nullptr; size_t out_hdr = 0, out_kps_bytes = 0, out_bytes = 0;   // the block the eight pointers above point into
    std::vector<size_t> chunk_off;   // orbx_extract_batch: byte offset of every chunk's own [counts | status | keypoints | descriptors] block in d_out / h_out
    int inflight = 0, inflight_frames = 0;      // orbx_extract_begin / orbx_extract_end
    // orbx_extract_begin replays one HIP graph per shape (upload, ~10 kernels, download) instead of ~12 launches
    hipGraphExec_t graph_exec = nullptr; int graph_w = 0, graph_h = 0, graph_seen_w = 0, graph_seen_h = 0; bool graph_off = false; int graph_fails = 0;
    // orbx_extract_batch in chunks: staging threads, two streams, one HIP graph per chunk (kernels + download) per shape
    StagePool *pool = nullptr; int batch_chunk = 16;
    std::vector<hipGraphExec_t> bgraph; int bg_w = 0, bg_h = 0, bg_n = 0, bg_chunk = 0; bool bg_off = false;
    std::vector<hipEvent_t> ev_up, ev_done;
    int profiling = 0; hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; float stage_ms[4] = {0, 0, 0, 0};
    // profiling == 2: the stage events of the last ORBX_PROF_RING calls are recorded and never waited for by the library
    hipEvent_t evr[ORBX_PROF_RING][5] = {}; long long ring_calls = 0;
};

// ---- A1: ORBextractor::ORBextractor tables (:412-472) ----
static void build_tables(orbx_extractor *h)
{
    const int L = h->nlevels;
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < L; i++) {
        h->scale[i] = h->scale[i - 1] * h->scale_factor;
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < L; i++) {
        h->inv_scale[i] = 1.0f / h->scale[i];
        h->inv_sigma2[i] = 1.0f / h->sigma2[i];
    }
    float factor = 1.0f / h->scale_factor;
    float nDesired = h->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int level = 0; level < L - 1; level++) {
        h->quota[level] = cv_round(nDesired);
        sum += h->quota[level];
        nDesired *= factor;
    }
    h->quota[L - 1] = std::max(h->nfeatures - sum, 0);

    int v, v0, vmax = cv_floor(ORBX_HALF_PATCH * sqrtf(2.f) / 2 + 1);
    int vmin = cv_ceil(ORBX_HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = ORBX_HALF_PATCH * ORBX_HALF_PATCH;
    for (v = 0; v <= vmax; ++v) h->umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = ORBX_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
        h->umax[v] = v0;
        ++v0;
    }
    // OpenCV getGaussianKernel(7, 2, CV_32F) -> x256 fixed point (cv::GaussianBlur u8 path)
    float cf[7]; double s = 0;
    for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(-0.5 / 4.0 * x * x); s += cf[i]; }
    s = 1. / s;
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * s); h->gauss_k[i] = cv_round((double)cf[i] * 256.0); }
}

// ---- shape planning: level sizes (:1113-1114), cell grid (:775-789), quadtree roots (:545-547) ----
static int make_plan(const orbx_extractor *h, int W, int H, OrbxPlan *P, std::string *why)
{
    memset(P, 0, sizeof(*P));
    P->nlevels = h->nlevels; P->ini_th = h->ini_th; P->min_th = h->min_th; P->blur_mode = h->blur_mode;
    long long cand_off = 0, list_off = 0, arena_off = 0;
    int cells = 0;
    for (int l = 0; l < h->nlevels; l++) {
        OrbxLevel &L = P->lv[l];
        L.w = cv_round((float)W * h->inv_scale[l]);
        L.h = cv_round((float)H * h->inv_scale[l]);
        if (L.w < 1 || L.h < 1 || L.w > 65535 || L.h > 65535) { *why = "level size out of range"; return ORBX_E_SHAPE; }
        L.maxBX = L.w - ORBX_MINB; L.maxBY = L.h - ORBX_MINB;
        const float width = (float)(L.maxBX - ORBX_MINB), height = (float)(L.maxBY - ORBX_MINB);
        L.nCols = width > 0 ? (int)(width / 30.f) : 0;
        L.nRows = height > 0 ? (int)(height / 30.f) : 0;
        if (L.nCols <= 0 || L.nRows <= 0) { L.nCols = L.nRows = 0; L.wCell = L.hCell = 1; }   // no cell => no keypoint
        else { L.wCell = (int)ceilf(width / L.nCols); L.hCell = (int)ceilf(height / L.nRows); }
        L.rcpW = L.wCell > 1 ? (uint32_t)((1ull << 32) / (unsigned)L.wCell + 1) : 0u;
        L.rcpH = L.hCell > 1 ? (uint32_t)((1ull << 32) / (unsigned)L.hCell + 1) : 0u;
        L.cell_begin = cells;
        cells += L.nCols * L.nRows;
        L.quota = h->quota[l];
        L.nIni = 0; L.hX = 1.f;
        if (L.nCols > 0) {
            L.nIni = (int)roundf(width / (float)(L.maxBY - ORBX_MINB));
            if (L.nIni <= 0) { *why = "portrait level (quadtree root count 0): undefined in the reference"; return ORBX_E_SHAPE; }
            L.hX = width / L.nIni;
        }
        // Candidate capacity = the most NMS survivors a level can have, so that no image overflows it (the reference has no
        // such limit): survivors are strict 8-neighbour maxima inside a cell's zone (cv::FAST nonmax, per cell :811-817), no
        // two of them are adjacent, so a zw x zh zone holds at most ceil(zw/2) * ceil(zh/2); the zones of a level's cells tile
        // [19, w-19) x [19, h-19), hence sum <= ceil((w-38+nCols)/2) * ceil((h-38+nRows)/2) (monotone in w and h, so a
        // smaller frame always fits the workspace planned for the handle's maximum).  The quadtree packs a candidate index
        // into 20 bits: only a level beyond ~4.1 M pixels can still report ORBX_E_CAND_OVERFLOW.
        const long long zw_all = std::max(L.w - 2 * ORBX_EDGE, 0), zh_all = std::max(L.h - 2 * ORBX_EDGE, 0);
        const long long zone = zw_all * zh_all;
        const long long nmax = ((zw_all + L.nCols + 1) / 2) * ((zh_all + L.nRows + 1) / 2);
        L.cand_cap = L.nCols > 0 ? (int)std::min<long long>(nmax + 64, (1 << 20) - 1) : 0;
        if (L.nCols > 0 && zone / 8 + 256 >= 100000) P->oct_big = 1;   // 1080p-class level: the quadtree runs 1024-thread workgroups
        // quadtree fast-forward depth (k_octree): 4 levels of the tree from one key histogram, 5 for 1080p-class levels; fewer when
        // many roots (a wide level) would make the tables large.  ORBX_OCT_FAST=0 turns it off (A/B measurements).
        L.fastD = 0;
        if (L.nCols > 0 && h->oct_fast) {
            int d = (zone / 8 + 256 >= 100000) ? 5 : 4;
            while (d > 0 && (long long)L.nIni * (((1ll << (2 * (d + 1))) - 1) / 3) > 2800) d--;
            L.fastD = d;
            P->oct_ft = std::max(P->oct_ft, (int)(L.nIni * (((1ll << (2 * (d + 1))) - 1) / 3)));
            // per-coordinate path tables of the fast-forward (k_octree): one u16 per column and per row of the level's box
            if (d > 0) P->oct_map = std::max(P->oct_map, (int)align_up((size_t)std::max(L.maxBX - ORBX_MINB, 1), 8) + (int)align_up((size_t)std::max(L.maxBY - ORBX_MINB, 1), 8));
        }
        L.cand_off = cand_off; cand_off += (L.cand_cap + 15) / 16 * 16;
        L.list_cap = L.nCols > 0 ? (std::max(L.quota + 3, 4 * L.nIni) + 1 + 3) / 4 * 4 : 0;
        L.list_off = list_off; list_off += L.list_cap;
        L.arena_cap = L.nCols > 0 ? 24 * L.list_cap + 256 : 0;
        L.arena_off = arena_off; arena_off += L.arena_cap;
        L.scale = h->scale[l];
        L.kp_size = (float)(int)(31 * h->scale[l]);   // :839,:848
    }
    P->ncells = cells;
    P->cand_frame = cand_off; P->list_frame = list_off; P->arena_frame = arena_off;
    P->out_cap = (int)list_off;
    return ORBX_OK;
}

// cv::resize INTER_LINEAR planning for one level pair (OpenCV 3.1.0 imgwarp.cpp)
static void plan_resize(int sw, int sh, int dw, int dh, int *xofs, short2 *alpha, int *yofs, short2 *beta, int *mode)
{
    const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    const double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    const int isx = cv_round(scale_x), isy = cv_round(scale_y);
    const bool area2 = fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON && isx == 2 && isy == 2;
    auto sat = [](float v) { int i = cv_round(v); return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i); };
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[dx] = make_short2(sat((1.f - fx) * 2048), sat(fx * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[dy] = make_short2(sat((1.f - fy) * 2048), sat(fy * 2048));
    }
    // the 4x4 kernel moves 8 source bytes per 4 destination columns: needs xofs[x+3]+1 - xofs[x] <= 7
    int span = 0;
    for (int dx = 0; dx + 3 < dw; dx += 4) span = std::max(span, xofs[dx + 3] + 1 - xofs[dx]);
    *mode = area2 ? RESIZE_AREA2 : (span <= 7 && sw >= 12 ? RESIZE_FAST : RESIZE_GENERIC);
    // the shared-row kernel (k_resize_linear_4x4s / k_resize_tiles): rows y4 .. y4+3 of every block of four destination rows
    // start r or r + 1 source rows below the block's first one (true for scale factors up to 4/3) and nothing reflects at the top
    if (*mode == RESIZE_FAST) {
        static const bool on = [] { const char *e = getenv("ORBX_RESIZE6"); return !e || atoi(e) != 0; }();   // A/B switch
        bool six = on && sh >= 2;
        for (int y4 = 0; y4 < dh && six; y4++) {                 // any first row: k_resize_tiles starts its blocks where a tile's region starts
            six = yofs[y4] >= 0;
            for (int r = 1; r < 4 && y4 + r < dh; r++) { const int o = yofs[y4 + r] - yofs[y4]; six = six && (o == r || o == r + 1); }
        }
        for (int dx = 0; dx + 3 < dw && six; dx++) six = xofs[dx + 3] + 1 - xofs[dx] <= 7;     // and any first column (the 8-byte window)
        if (six) *mode = RESIZE_FAST6;
    }
}

// Stream captures against the rest of the process.  A capture is begun in RELAXED mode (this library issues nothing unsafe inside one,
// and other threads' calls must not be judged against it), and the phases in which a handle uses synchronous runtime calls -- creation,
// destruction, the table upload of a shape change -- exclude every capture of this library through one process-wide lock: on this runtime
// a synchronous copy in one thread has been seen to fail, and to invalidate the capture of ANOTHER thread's handle, even in thread-local
// mode (tests/test_threads_gpu.py, once in a dozen runs).  A capture that is invalidated all the same is not an error: the call runs
// plainly and the capture is tried again on a later call (three times at most).
static std::recursive_mutex &capture_mutex()
{
    static std::recursive_mutex m;
    return m;
}
static void free_all(orbx_extractor *h)
{
    std::lock_guard<std::recursive_mutex> lk_(capture_mutex());
    if (!h) return;
    hipSetDevice(h->device);
    (void)hipHostFree(h->h_pyr);
    hipFree(h->d_input); hipFree(h->d_pyr); hipFree(h->d_tab_i); hipFree(h->d_tab_s); hipFree(h->d_cells); hipFree(h->d_bands); hipFree(h->d_tiles);
    hipFree(h->work.cand); hipFree(h->work.cand_count); hipFree(h->work.owner); hipFree(h->work.arena);
    hipFree(h->work.sel); hipFree(h->work.nk); hipFree(h->work.ncand); hipFree(h->work.errflags);
    hipFree(h->d_out);
    hipHostFree(h->h_in); hipHostFree(h->h_out);
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    for (auto &g : h->bgraph) if (g) hipGraphExecDestroy(g);
    delete h->pool;
    for (auto &e : h->ev_up) if (e) hipEventDestroy(e);
    for (auto &e : h->ev_done) if (e) hipEventDestroy(e);
    for (auto &e : h->ev) if (e) hipEventDestroy(e);
    for (auto &set : h->evr) for (auto &e : set) if (e) hipEventDestroy(e);
    for (auto &e : h->ev_join) if (e) hipEventDestroy(e);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    for (auto &a : h->aux) if (a) hipStreamDestroy(a);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int orbx_create(orbx_extractor **out, int nfeatures, float scale_factor, int nlevels,
                           int ini_th, int min_th, int device, int max_width, int max_height, int max_batch)
{
    std::lock_guard<std::recursive_mutex> lk_(capture_mutex());
    if (!out) return fail(ORBX_E_INVALID, "out is NULL");
    *out = nullptr;
    if (nfeatures < 0 || nlevels < 1 || nlevels > ORBX_MAX_LEVELS || !(scale_factor > 1.0f) ||
        max_width < 1 || max_height < 1 || max_batch < 1)
        return fail(ORBX_E_INVALID, "bad constructor argument (nfeatures=%d scale=%g nlevels=%d max=%dx%dx%d)",
                    nfeatures, scale_factor, nlevels, max_width, max_height, max_batch);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(ORBX_E_HIP, "no HIP device: liborbx has no CPU path");
    if (device < 0 || device >= ndev) return fail(ORBX_E_INVALID, "device %d of %d", device, ndev);
    HIPCHK(hipSetDevice(device));

    orbx_extractor *h = new orbx_extractor();
    h->nfeatures = nfeatures; h->scale_factor = scale_factor; h->nlevels = nlevels;
    h->ini_th = std::min(std::max(ini_th, 0), 255); h->min_th = std::min(std::max(min_th, 0), 255);
    h->device = device; h->max_w = max_width; h->max_h = max_height; h->max_batch = max_batch;
    memset(&h->work, 0, sizeof(h->work));
    { const char *e = getenv("ORBX_OCT_FAST"); if (e) h->oct_fast = atoi(e); }
    build_tables(h);

    std::string why;
    int rc = make_plan(h, max_width, max_height, &h->max_plan, &why);
    if (rc != ORBX_OK) { delete h; return fail(rc, "max shape %dx%d: %s", max_width, max_height, why.c_str()); }
    int max_list = 0;
    for (int l = 0; l < nlevels; l++) max_list = std::max(max_list, h->max_plan.lv[l].list_cap);
    h->oct_cap_max = std::max(max_list, 8);
    h->oct_lds = orbx_octree_lds_bytes(h->oct_cap_max, 0, 0);
    if (h->oct_lds > 150 * 1024) { delete h; return fail(ORBX_E_INVALID, "nfeatures=%d needs %zu B of LDS for the quadtree (max 153600)", nfeatures, h->oct_lds); }

#define ALLOC(ptr, bytes)                                                                       \
    do {                                                                                        \
        hipError_t e_ = hipMalloc((void **)&(ptr), std::max<size_t>((bytes), 256));             \
        if (e_ != hipSuccess) { free_all(h); return fail(ORBX_E_HIP, "hipMalloc(%s, %zu): %s", #ptr, (size_t)(bytes), hipGetErrorString(e_)); } \
    } while (0)
    const size_t B = (size_t)max_batch;
    h->in_stride = (int)align_up(max_width, 64);
    h->in_frame = align_up((size_t)h->in_stride * max_height, 256);
    ALLOC(h->d_input, B * h->in_frame + 256);
    size_t off = 0, tab_e = 0;
    for (int l = 1; l < nlevels; l++) {
        const OrbxLevel &L = h->max_plan.lv[l];
        h->pyr_level_off[l] = off;
        off += B * align_up(align_up(L.w, 64) * (size_t)L.h, 256);
        tab_e += align_up((size_t)L.w + 4, 4) + align_up((size_t)L.h + 4, 4);
    }
    h->pyr_bytes = off; h->tab_elems = tab_e;
    ALLOC(h->d_pyr, off + 256);   // slack: the 4x4 resize reads whole dwords around a row segment
    ALLOC(h->d_tab_i, tab_e * sizeof(int));
    ALLOC(h->d_tab_s, tab_e * sizeof(short2));
    h->bands_cap = 2 * ((size_t)max_height / 8 + 4) * ORBX_MAX_LEVELS;
    ALLOC(h->d_bands, h->bands_cap * sizeof(int4));
    h->tiles_cap = 8192;
    ALLOC(h->d_tiles, h->tiles_cap * sizeof(int4));
    {   // ORBX_PYRAMID_TILES = "a[,tile width[,tile height[,min frames]]]": levels a + 1 .. last in one launch (0 = off = default).
        // Bit-exact, and at 64 x 640x480 slower than the per-level launches (levels 3..7: 40 us against 27; without any store 30): the
        // tiles' halos make it compute 1.8x the pixels, two waves per SIMD are all the 1920 tiles give.  Kept as an A/B switch.
        const char *e = getenv("ORBX_PYRAMID_TILES");
        if (e) { int a = 2, tw = 32, th = 32, mf = 8; const int n = sscanf(e, "%d,%d,%d,%d", &a, &tw, &th, &mf); if (n >= 1) h->tile_a = a; if (n >= 2) h->tile_w = tw; if (n >= 3) h->tile_h = th; if (n >= 4) h->tile_min_frames = mf; }
        if (h->tile_w < 8 || h->tile_w > 128 || (h->tile_w & 3) || h->tile_h < 8 || h->tile_h > 128 || (h->tile_h & 3)) h->tile_a = 0;
    }
    { const char *e = getenv("ORBX_PYRAMID_FUSE"); if (e) h->fuse_on = atoi(e); }
    { const char *e = getenv("ORBX_OVERLAP_PYRAMID"); if (e) h->overlap_pyr = atoi(e) != 0; }   // A/B switch for ORBX_OPT_OVERLAP_PYRAMID
    h->cells_cap = h->max_plan.ncells + 64 * nlevels;   // a smaller frame never has more cells; slack for rounding
    ALLOC(h->d_cells, (size_t)h->cells_cap * sizeof(uint32_t));
    const OrbxPlan &M = h->max_plan;
    ALLOC(h->work.cand, B * M.cand_frame * sizeof(OrbxCand));
    ALLOC(h->work.owner, B * M.cand_frame * sizeof(uint32_t));
    ALLOC(h->work.arena, B * M.arena_frame * sizeof(OrbxNode));
    ALLOC(h->work.sel, B * M.list_frame * sizeof(OrbxCand));
    ALLOC(h->work.cand_count, B * ORBX_MAX_LEVELS * ORBX_CNT_STRIDE * sizeof(uint32_t));
    ALLOC(h->work.nk, B * ORBX_MAX_LEVELS * sizeof(uint32_t));
    ALLOC(h->work.ncand, B * ORBX_MAX_LEVELS * sizeof(uint32_t));
    ALLOC(h->work.errflags, B * sizeof(uint32_t));
    // the host-buffer entry points' staging outputs live in ONE block, [counts B | status B | keypoints B x cap | descriptors
    // B x cap x 32], mirrored in pinned memory: a full batch (or a max_batch = 1 handle) comes back with a single copy
    h->out_hdr = align_up(2 * B * sizeof(int32_t), 256);
    h->out_kps_bytes = B * M.out_cap * sizeof(orbx_keypoint);
    h->out_bytes = h->out_hdr + h->out_kps_bytes + B * M.out_cap * 32;
    h->out_bytes += 256 * (B + 2);   // per-chunk blocks of orbx_extract_batch: one aligned header per chunk instead of one per batch
    ALLOC(h->d_out, h->out_bytes);
#undef ALLOC
    if (hipHostMalloc((void **)&h->h_in, B * h->in_frame + 256) != hipSuccess ||
        hipHostMalloc((void **)&h->h_out, h->out_bytes) != hipSuccess) {
        free_all(h);
        return fail(ORBX_E_HIP, "hipHostMalloc failed");
    }
    h->d_counts = reinterpret_cast<int32_t *>(h->d_out); h->d_status = h->d_counts + B;
    h->d_kps = reinterpret_cast<orbx_keypoint *>(h->d_out + h->out_hdr); h->d_desc = h->d_out + h->out_hdr + h->out_kps_bytes;
    h->h_counts = reinterpret_cast<int32_t *>(h->h_out); h->h_status = h->h_counts + B;
    h->h_kps = reinterpret_cast<orbx_keypoint *>(h->h_out + h->out_hdr); h->h_desc = h->h_out + h->out_hdr + h->out_kps_bytes;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { free_all(h); return fail(ORBX_E_HIP, "hipStreamCreate failed"); }
    for (auto &e : h->ev) if (hipEventCreate(&e) != hipSuccess) { free_all(h); return fail(ORBX_E_HIP, "hipEventCreate failed"); }
    for (auto &a : h->aux) if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess) { free_all(h); return fail(ORBX_E_HIP, "hipStreamCreate failed"); }
    for (auto &e : h->ev_join) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { free_all(h); return fail(ORBX_E_HIP, "hipEventCreate failed"); }
    if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) { free_all(h); return fail(ORBX_E_HIP, "hipEventCreate failed"); }
    if (orbx_upload_constants(h->umax, h->gauss_k) != 0) { free_all(h); return fail(ORBX_E_HIP, "constant upload failed"); }
    if (orbx_selftest_fp16() != 0) { free_all(h); return fail(ORBX_E_HIP, "fp16 subnormal self-test failed: the FAST score tree needs fp16 subnormals enabled on this device"); }
    *out = h;
    return ORBX_OK;
}

extern "C" void orbx_destroy(orbx_extractor *h) { free_all(h); }

extern "C" int orbx_set_option(orbx_extractor *h, int option, int value)
{
    if (!h) return fail(ORBX_E_INVALID, "NULL handle");
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; h->graph_w = h->graph_h = 0; }   // options are baked into the graph
    for (auto &g : h->bgraph) if (g) (void)hipGraphExecDestroy(g);
    h->bgraph.clear(); h->bg_w = h->bg_h = h->bg_n = 0;
    if (option == ORBX_OPT_BATCH_CHUNK && value >= 0 && value <= 4096) { h->batch_chunk = value; return ORBX_OK; }
    if (option == ORBX_OPT_BLUR_ROUNDING && (value == 0 || value == 1)) { h->blur_mode = value; h->plan.blur_mode = value; return ORBX_OK; }
    if (option == ORBX_OPT_SUBBATCHES && value >= 1 && value <= ORBX_MAX_SUB) { h->nsub = value; return ORBX_OK; }
    if (option == ORBX_OPT_OVERLAP_PYRAMID && (value == 0 || value == 1)) { h->overlap_pyr = value; return ORBX_OK; }
    return fail(ORBX_E_INVALID, "unknown option %d=%d", option, value);
}
extern "C" int orbx_get_levels(const orbx_extractor *h) { return h ? h->nlevels : 0; }
extern "C" float orbx_get_scale_factor(const orbx_extractor *h) { return h ? h->scale_factor : 0.f; }
extern "C" int orbx_get_tables(const orbx_extractor *h, float *sf, float *isf, float *s2, float *is2)
{
    if (!h) return fail(ORBX_E_INVALID, "NULL handle");
    for (int i = 0; i < h->nlevels; i++) {
        if (sf) sf[i] = h->scale[i];
        if (isf) isf[i] = h->inv_scale[i];
        if (s2) s2[i] = h->sigma2[i];
        if (is2) is2[i] = h->inv_sigma2[i];
    }
    return ORBX_OK;
}
extern "C" int orbx_get_features_per_level(const orbx_extractor *h, int *q)
{
    if (!h || !q) return fail(ORBX_E_INVALID, "NULL argument");
    for (int i = 0; i < h->nlevels; i++) q[i] = h->quota[i];
    return ORBX_OK;
}
extern "C" int orbx_capacity(const orbx_extractor *h) { return h ? h->max_plan.out_cap : 0; }
extern "C" int orbx_set_profiling(orbx_extractor *h, int mode)
{
    if (!h || mode < 0 || mode > 2) return fail(ORBX_E_INVALID, "profiling mode %d", mode);
    if (mode == 2 && !h->evr[0][0]) {
        HIPCHK(hipSetDevice(h->device));
        for (auto &set : h->evr) for (auto &e : set) HIPCHK(hipEventCreate(&e));
    }
    h->profiling = mode; h->ring_calls = 0;
    return ORBX_OK;
}
extern "C" int orbx_stage_ms_ring(orbx_extractor *h, float *ms, int max_calls)
{
    if (!h || !ms || max_calls < 0) return fail(ORBX_E_INVALID, "bad argument");
    if (h->profiling != 2) return fail(ORBX_E_INVALID, "ring profiling is off");
    const int n = (int)std::min<long long>(std::min<long long>(h->ring_calls, ORBX_PROF_RING), max_calls);
    for (int i = 0; i < n; i++) {                           // newest first
        hipEvent_t *e = h->evr[(h->ring_calls - 1 - i) % ORBX_PROF_RING];
        for (int k = 0; k < 4; k++) HIPCHK(hipEventElapsedTime(&ms[4 * i + k], e[k], e[k + 1]));   // fails if the caller has not synchronised
    }
    return n;
}
extern "C" int orbx_last_stage_ms(orbx_extractor *h, float ms[4])
{
    if (!h || !ms) return fail(ORBX_E_INVALID, "NULL argument");
    memcpy(ms, h->stage_ms, sizeof(float) * 4);
    return ORBX_OK;
}

// One axis of the tile plan of k_resize_tiles.  n[l] = extent of level l, ofs[l] = level l's source-offset table (level-l coordinate ->
// level l-1 coordinate, monotone), T = tile extent at level b.  out[i * (b - a + 1) + (l - a)] = (own0, own1, comp0, comp1) of tile i at
// level l: the owned ranges of a level are cut at the images of the level-b tile boundaries , so they partition the level; the computed range is the hull of the owned range
// and the bilinear footprint of what the tile computes one level down, lengthened to a multiple of 4 (the kernel works in 4x4 blocks
// counted from the range's first pixel).  align_own (the x axis): the computed range starts a multiple of 4 before the owned one, so
// that a block is owned from its first column on or not at all (the range may then start at -1 .. -3: the kernel's table slices repeat column 0).
static int plan_tile_axis(int a, int b, int T, const int *n, const int *const *ofs, bool align_own, std::vector<int4> &out, int *max_comp)
{
    const int nt = (n[b] + T - 1) / T, nl = b - a + 1;
    T = std::min(T, (((n[b] + nt - 1) / nt) + 3) & ~3);       // equal tiles: the kernel ends with its largest tile, and a sliver of a tile carries a full halo
    out.assign((size_t)nt * nl, make_int4(0, 0, 0, 0));
    std::vector<std::vector<int>> B((size_t)nl, std::vector<int>((size_t)nt + 1, 0));
    for (int i = 0; i <= nt; i++) B[(size_t)(b - a)][(size_t)i] = std::min(i * T, n[b]);
    for (int l = b - 1; l >= a; l--)
        for (int i = 0; i <= nt; i++) {
            int v = 0;
            if (i == nt) v = n[l];
            else if (i > 0) {
                const int d = B[(size_t)(l + 1 - a)][(size_t)i];
                if (d >= n[l + 1]) v = n[l];
                else v = std::min(std::max(ofs[l + 1][d], 0), n[l] - 1);
                v = std::max(v, B[(size_t)(l - a)][(size_t)i - 1]);
            }
            B[(size_t)(l - a)][(size_t)i] = v;
        }
    for (int l = a; l <= b; l++) max_comp[l] = 0;
    for (int i = 0; i < nt; i++) {
        int c0 = B[(size_t)(b - a)][(size_t)i], c1 = c0 + ((B[(size_t)(b - a)][(size_t)i + 1] - c0 + 3) & ~3);
        out[(size_t)i * nl + (size_t)(b - a)] = make_int4(B[(size_t)(b - a)][(size_t)i], B[(size_t)(b - a)][(size_t)i + 1], c0, c1);
        max_comp[b] = std::max(max_comp[b], c1 - c0);
        for (int l = b - 1; l >= a; l--) {
            const int v1 = std::min(c1, n[l + 1]);                     // valid coordinates of the computed range one level down
            if (v1 <= c0) return -1;
            const int need0 = std::min(std::max(ofs[l + 1][std::max(c0, 0)], 0), n[l] - 1);
            const int need1 = std::min(std::max(ofs[l + 1][v1 - 1], 0) + 1, n[l] - 1) + 1;
            const int o0 = B[(size_t)(l - a)][(size_t)i], o1 = B[(size_t)(l - a)][(size_t)i + 1];
            c0 = o0 < o1 ? std::min(o0, need0) : need0;
            if (align_own && o0 < o1) c0 = o0 - ((o0 - c0 + 3) & ~3);                // the owned part starts on a block boundary (c0 may be -1 .. -3)
            c1 = c0 + (((o0 < o1 ? std::max(o1, need1) : need1) - c0 + 3) & ~3);     // whole 4x4 blocks from the range's first pixel on
            out[(size_t)i * nl + (size_t)(l - a)] = make_int4(o0, o1, c0, c1);
            max_comp[l] = std::max(max_comp[l], c1 - c0);
        }
    }
    return nt;
}

// (re)plan for a frame shape; buffers stay those sized at create
static int ensure_plan(orbx_extractor *h, int W, int H)
{
    if (W == h->cur_w && H == h->cur_h) return ORBX_OK;
    if (W > h->max_w || H > h->max_h) return fail(ORBX_E_SHAPE, "frame %dx%d exceeds the handle's max %dx%d", W, H, h->max_w, h->max_h);
    OrbxPlan P;
    std::string why;
    int rc = make_plan(h, W, H, &P, &why);
    if (rc != ORBX_OK) return fail(rc, "frame %dx%d: %s", W, H, why.c_str());
    const OrbxPlan &M = h->max_plan;
    // keep the allocation layout of the max plan (offsets/capacities) so every shape fits
    for (int l = 0; l < h->nlevels; l++) {
        OrbxLevel &L = P.lv[l];
        const OrbxLevel &X = M.lv[l];
        if (L.cand_cap > X.cand_cap || L.list_cap > X.list_cap || L.arena_cap > X.arena_cap || L.w > X.w || L.h > X.h)
            return fail(ORBX_E_SHAPE, "frame %dx%d level %d does not fit the workspace planned for %dx%d", W, H, l, h->max_w, h->max_h);
        L.cand_off = X.cand_off; L.list_off = X.list_off; L.arena_off = X.arena_off;
        if (L.nCols > 0) { L.cand_cap = X.cand_cap; L.arena_cap = X.arena_cap; }
    }
    P.cand_frame = M.cand_frame; P.list_frame = M.list_frame; P.arena_frame = M.arena_frame; P.out_cap = M.out_cap;

    std::vector<int> ti(h->tab_elems ? h->tab_elems : 1);
    std::vector<short2> ts(h->tab_elems ? h->tab_elems : 1);
    size_t e = 0;
    size_t yofs_at[ORBX_MAX_LEVELS] = {}, xofs_at[ORBX_MAX_LEVELS] = {};
    for (int l = 1; l < h->nlevels; l++) {
        OrbxLevel &L = P.lv[l];
        L.stride = (int)align_up(L.w, 64);
        L.frame_stride = (long long)align_up((size_t)L.stride * L.h, 256);
        L.base = h->d_pyr + h->pyr_level_off[l];
        const OrbxLevel &S = P.lv[l - 1];
        const size_t ex = align_up((size_t)L.w + 4, 4), ey = align_up((size_t)L.h + 4, 4);
        plan_resize(S.w, S.h, L.w, L.h, &ti[e], &ts[e], &ti[e + ex], &ts[e + ex], &h->area2[l]);
        for (size_t dy = (size_t)L.h; dy < ey; dy++) { ti[e + ex + dy] = ti[e + ex + L.h - 1]; ts[e + ex + dy] = ts[e + ex + L.h - 1]; }   // k_resize_linear_4x4 reads rows in fours
        yofs_at[l] = e + ex; xofs_at[l] = e;
        h->tabs[l].xofs = h->d_tab_i + e; h->tabs[l].alpha = h->d_tab_s + e;
        h->tabs[l].yofs = h->d_tab_i + e + ex; h->tabs[l].beta = h->d_tab_s + e + ex;
        e += ex + ey;
    }
    if (P.ncells > h->cells_cap) return fail(ORBX_E_SHAPE, "frame %dx%d has more FAST cells than the workspace planned for %dx%d", W, H, h->max_w, h->max_h);
    std::vector<uint32_t> cells((size_t)std::max(P.ncells, 1));
    for (int l = 0; l < h->nlevels; l++) {
        const OrbxLevel &L = P.lv[l];
        if (L.nRows >= 4096 || L.nCols >= 4096) return fail(ORBX_E_SHAPE, "level %d has too many FAST cells", l);
        for (int i = 0; i < L.nRows; i++)
            for (int j = 0; j < L.nCols; j++) cells[(size_t)L.cell_begin + (size_t)i * L.nCols + j] = (uint32_t)l | ((uint32_t)i << 4) | ((uint32_t)j << 16);
    }
    // ---- fused upper levels: row-band ownership / footprint tables (k_resize_fused) ----
    std::vector<int4> bands;
    for (int v = 0; v < 2; v++) {
        orbx_extractor::FusePlan &F = h->fuse[v];
        F = orbx_extractor::FusePlan();
        F.bh = v == 0 ? 16 : 8;
        const int b = h->nlevels - 1;
        for (int a = 1; h->fuse_on && b - a >= 2 && b < ORBX_FUSE_MAX; a++) {
            bool fast = true;
            for (int l = a + 1; l <= b; l++) fast = fast && resize_is_fast(h->area2[l]) && (P.lv[l].w + 3) / 4 <= 512;
            if (!fast) continue;
            const int nl = b - a + 1, nb = (P.lv[b].h + F.bh - 1) / F.bh;
            std::vector<int4> t((size_t)nb * nl);
            int need_rows[ORBX_MAX_LEVELS] = {};
            for (int j = 0; j < nb; j++) {
                int o0 = j * F.bh, o1 = std::min((j + 1) * F.bh, P.lv[b].h), n0 = o0, n1 = o1;
                t[(size_t)j * nl + (b - a)] = make_int4(o0, o1, n0, n1);
                need_rows[b] = std::max(need_rows[b], n1 - n0);
                for (int l = b - 1; l >= a; l--) {
                    const int *yo = &ti[yofs_at[l + 1]];
                    const int hl = P.lv[l].h, hu = P.lv[l + 1].h;
                    auto cl = [&](int v2) { return std::min(std::max(v2, 0), hl - 1); };
                    const int p0 = o0 == 0 ? 0 : cl(yo[o0]), p1 = o1 == hu ? hl : cl(yo[o1]);
                    const int q0 = std::min(cl(yo[n0]), p0), q1 = std::max(cl(yo[n1 - 1] + 1) + 1, p1);
                    o0 = p0; o1 = p1; n0 = q0; n1 = q1;
                    t[(size_t)j * nl + (l - a)] = make_int4(o0, o1, n0, n1);
                    need_rows[l] = std::max(need_rows[l], n1 - n0);
                }
            }
            int buf[2] = {0, 0};
            for (int l = a + 1; l < b; l++) {
                const int pitch = (int)align_up((size_t)P.lv[l].w + 12, 16);
                buf[(l - a) & 1] = std::max(buf[(l - a) & 1], need_rows[l] * pitch);
            }
            int ysum = 0;
            for (int l = a + 1; l <= b; l++) ysum += need_rows[l];
            if (buf[0] + buf[1] > 64 * 1024 || ysum > ORBX_FUSE_YTAB) continue;      // too much for LDS from this level on: start the fusion one level up
            if (bands.size() + t.size() > h->bands_cap) break;
            F.ok = true; F.a = a; F.b = b; F.nbands = nb; F.buf0 = (int)align_up((size_t)buf[0], 16); F.lds = F.buf0 + (int)align_up((size_t)buf[1], 16) + 16;
            F.off = bands.size();
            bands.insert(bands.end(), t.begin(), t.end());
            break;
        }
    }
    // ---- fused upper levels, one wave per 2-D tile (k_resize_tiles) ----
    std::vector<int4> tiles;
    h->tile = orbx_extractor::TilePlan();
    {
        orbx_extractor::TilePlan &T = h->tile;
        const int b = h->nlevels - 1, a = h->tile_a;
        bool ok = a >= 1 && b - a >= 2 && b < ORBX_FUSE_MAX;
        for (int l = a + 1; l <= b && ok; l++) ok = h->area2[l] == RESIZE_FAST6;
        if (ok) {
            int nw[ORBX_MAX_LEVELS], nh[ORBX_MAX_LEVELS], mcx[ORBX_MAX_LEVELS], mcy[ORBX_MAX_LEVELS];
            const int *ox[ORBX_MAX_LEVELS] = {}, *oy[ORBX_MAX_LEVELS] = {};
            for (int l = a; l <= b; l++) { nw[l] = P.lv[l].w; nh[l] = P.lv[l].h; if (l > a) { ox[l] = &ti[xofs_at[l]]; oy[l] = &ti[yofs_at[l]]; } }
            std::vector<int4> tx, ty;
            const int ntx = plan_tile_axis(a, b, h->tile_w, nw, ox, true, tx, mcx), nty = plan_tile_axis(a, b, h->tile_h, nh, oy, false, ty, mcy);
            ok = ntx >= 1 && nty >= 1 && ntx <= 64 && ntx * nty < 4096 && tx.size() + ty.size() <= h->tiles_cap;
            int off = 0;
            for (int l = a + 1; l <= b && ok; l++) {                             // every fused level has its own region in the wave's LDS
                ok = mcx[l] / 4 <= 64 && (mcx[l] / 4) * mcy[l] < 4096;           // the lane -> block / lane -> dword division table of the kernel
                T.lds_off[l] = off;
                off += (int)align_up((size_t)mcy[l] * mcx[l] + ORBX_TILE_SLACK, 16);
            }
            for (int l = a + 1; l <= b && ok; l++) { T.tab_off[l] = off; off += 8 * (mcx[l] + mcy[l]); }    // the tile's table slices (mcx, mcy: multiples of 4)
            if (ok && off <= 60 * 1024) {
                T.ok = true; T.a = a; T.b = b; T.ntx = ntx; T.nty = nty; T.lds = off;
                T.offx = 0; T.offy = tx.size();
                tiles = tx; tiles.insert(tiles.end(), ty.begin(), ty.end());
            }
        }
    }
    P.cell_tab = h->d_cells;
    // a shape change rewrites tables that kernels of an earlier call may still be reading -- on the handle's stream, its aux
    // streams or a caller's stream (orbx_extract_batch_device): wait for the device, not only for h->stream (shape changes are rare)
    std::lock_guard<std::recursive_mutex> lk_(capture_mutex());
    HIPCHK(hipDeviceSynchronize());
    if (!bands.empty()) HIPCHK(hipMemcpy(h->d_bands, bands.data(), bands.size() * sizeof(int4), hipMemcpyHostToDevice));
    if (!tiles.empty()) HIPCHK(hipMemcpy(h->d_tiles, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_cells, cells.data(), (size_t)P.ncells * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_tab_i, ti.data(), e * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_tab_s, ts.data(), e * sizeof(short2), hipMemcpyHostToDevice));
    h->plan = P;
    h->cur_w = W; h->cur_h = H;
    return ORBX_OK;
}

// enqueue the whole pipeline for `nframes` frames already resident in HBM
static int enqueue(orbx_extractor *h, const uint8_t *d_images, int nframes, int W, int H, int row_stride,
                   long long frame_stride, orbx_keypoint *d_kps, uint8_t *d_desc, int32_t *d_counts,
                   int32_t *d_status, hipStream_t s, int work_frame0 = 0)   // work_frame0: first frame slot of the workspace / pyramid
{
    int rc = ensure_plan(h, W, H);
    if (rc != ORBX_OK) return rc;
    OrbxPlan &P = h->plan;
    P.blur_mode = h->blur_mode;
    P.lv[0].base = const_cast<uint8_t *>(d_images);
    P.lv[0].stride = row_stride;
    P.lv[0].frame_stride = frame_stride;
    h->last_input = d_images; h->last_in_stride = row_stride; h->last_in_frame = frame_stride; h->last_batch = nframes;

    const bool prof = h->profiling == 1;          // mode 1 times one call in isolation; mode 2 only drops events into the stream
    hipEvent_t *pe = h->profiling == 2 ? h->evr[h->ring_calls % ORBX_PROF_RING] : (prof ? h->ev : nullptr);
    if (h->need_clear) {   // the kernels leave the counters and flags zeroed; only the first call (or one after an error) clears
        HIPCHK(hipMemsetAsync(h->work.cand_count, 0, (size_t)h->max_batch * ORBX_MAX_LEVELS * ORBX_CNT_STRIDE * sizeof(uint32_t), s));
        HIPCHK(hipMemsetAsync(h->work.errflags, 0, (size_t)h->max_batch * sizeof(uint32_t), s));
        h->need_clear = false;
    }
    struct DirtyGuard { orbx_extractor *h; bool ok = false; ~DirtyGuard() { if (!ok) h->need_clear = true; } } guard{h};
    const uint8_t *src_end = nullptr;   // level 0 in caller memory has no slack behind its last byte
    if (d_images < h->d_input || d_images >= h->d_input + (size_t)h->max_batch * h->in_frame)   // the handle's own input block has slack
        src_end = d_images + (long long)(nframes - 1) * frame_stride + (long long)(H - 1) * row_stride + W;

    // Sub-batches on separate streams: the quadtree and the small pyramid levels are latency-bound
    // (few, long workgroups), so one half-batch's latency-bound kernels run beside the other half's
    // VALU-bound ones.  Frames are independent, so a sub-batch is only a pointer offset.
    int nsub = (h->profiling != 0 || nframes < 16) ? 1 : std::min(h->nsub, ORBX_MAX_SUB);
    hipStream_t st[ORBX_MAX_SUB];
    OrbxPlan sp[ORBX_MAX_SUB];
    OrbxWork sw[ORBX_MAX_SUB];
    int f0[ORBX_MAX_SUB + 1];
    for (int i = 0; i <= nsub; i++) f0[i] = (int)((long long)nframes * i / nsub);
    for (int i = 0; i < nsub; i++) {
        st[i] = i == 0 ? s : h->aux[i - 1];
        sp[i] = P;
        sw[i] = h->work;
        const long long o = (long long)work_frame0 + f0[i];
        sp[i].lv[0].base = P.lv[0].base + (long long)f0[i] * P.lv[0].frame_stride;      // d_images is this call's first frame already
        for (int l = 1; l < h->nlevels; l++) sp[i].lv[l].base = P.lv[l].base + o * P.lv[l].frame_stride;
        sw[i].cand += o * P.cand_frame; sw[i].owner += o * P.cand_frame; sw[i].arena += o * P.arena_frame;
        sw[i].sel += o * P.list_frame; sw[i].nk += o * h->nlevels; sw[i].ncand += o * h->nlevels; sw[i].errflags += o;
        sw[i].cand_count += o * h->nlevels * ORBX_CNT_STRIDE;
    }
    if (nsub > 1) {
        HIPCHK(hipEventRecord(h->ev_fork, s));
        for (int i = 1; i < nsub; i++) HIPCHK(hipStreamWaitEvent(st[i], h->ev_fork, 0));
    }
    if (pe) HIPCHK(hipEventRecord(pe[0], s));
    // FAST on level 0 needs only the input, so with one sub-batch the resize chain (seven small, latency-bound
    // launches) runs on a side stream underneath it; the remaining levels' cells wait for the chain.
    const bool overlap = h->profiling == 0 && nsub == 1 && h->overlap_pyr && h->nlevels > 1 && P.lv[1].cell_begin > 0;
    if (overlap) {
        hipStream_t sa = h->aux[ORBX_MAX_SUB - 2];
        HIPCHK(hipEventRecord(h->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(sa, h->ev_fork, 0));
        for (int l = 1; l < h->nlevels; l++)
            orbx_launch_resize(sp[0].lv[l - 1], sp[0].lv[l], h->tabs[l], h->area2[l], nframes, l == 1 ? src_end : nullptr, sa);
        HIPCHK(hipEventRecord(h->ev_join[ORBX_MAX_SUB - 2], sa));
        orbx_launch_fast(sp[0], sw[0], nframes, 0, P.lv[1].cell_begin, s);
        HIPCHK(hipStreamWaitEvent(s, h->ev_join[ORBX_MAX_SUB - 2], 0));
        orbx_launch_fast(sp[0], sw[0], nframes, P.lv[1].cell_begin, P.ncells, s);
    } else {
        for (int i = 0; i < nsub; i++) {
            const int nf = f0[i + 1] - f0[i];
            // The upper levels in one launch -- for a FEW frames only.  The resize arithmetic is ~22 vector instructions per pixel
            // and the per-level kernels of a large batch are bound by that, not by their launches (rocprofv3 + SQ_INSTS_VALU:
            // 47 % VALU-busy over the six upper levels at 64 x 640 x 480); the fused kernel recomputes the band overlaps
            // (+25 % pixels) and measured 70 us against 57 for the chain there.  With one frame the launches dominate and
            // fusing wins (28.6 -> 24.6 us at 640 x 480).  ORBX_PYRAMID_FUSE=2 forces it for A/B measurements.
            const orbx_extractor::FusePlan *F = nullptr;
            const orbx_extractor::FusePlan *cand = h->fuse[1].ok ? &h->fuse[1] : (h->fuse[0].ok ? &h->fuse[0] : nullptr);
            if (h->fuse_on == 2 && h->fuse[0].ok && (long long)h->fuse[0].nbands * nf >= 384) cand = &h->fuse[0];
            if (cand && (h->fuse_on == 2 || (long long)nf * sp[i].lv[cand->a + 1].w * sp[i].lv[cand->a + 1].h <= 1200000)) F = cand;
            // many frames: the upper levels as one wave per 2-D tile (k_resize_tiles); it takes precedence over the band kernel
            const bool use_tiles = h->tile.ok && h->fuse_on != 2 && nf >= h->tile_min_frames;
            if (use_tiles) F = nullptr;
            const int last_single = use_tiles ? h->tile.a : (F ? F->a : h->nlevels - 1);
            for (int l = 1; l <= last_single; l++)
                orbx_launch_resize(sp[i].lv[l - 1], sp[i].lv[l], h->tabs[l], h->area2[l], nf, l == 1 ? src_end : nullptr, st[i]);
            if (use_tiles) {
                TileArgs A;
                memset(&A, 0, sizeof(A));
                A.a = h->tile.a; A.b = h->tile.b; A.ntx = h->tile.ntx; A.nty = h->tile.nty;
                for (int l = 0; l < ORBX_FUSE_MAX; l++) { A.lds_off[l] = h->tile.lds_off[l]; A.tab_off[l] = h->tile.tab_off[l]; }
                A.xr = h->d_tiles + h->tile.offx; A.yr = h->d_tiles + h->tile.offy;
                for (int l = A.a; l <= A.b; l++) {
                    const OrbxLevel &L = sp[i].lv[l];
                    TileLevel &U = A.lv[l];
                    U.base = L.base; U.w = L.w; U.h = L.h; U.stride = L.stride; U.frame = L.frame_stride; U.tab = h->tabs[l];
                }
                orbx_launch_resize_tiles(A, nf, (size_t)h->tile.lds, st[i]);
            }
            if (F) {
                FuseArgs A;
                memset(&A, 0, sizeof(A));
                A.a = F->a; A.b = F->b; A.bands = h->d_bands + F->off; A.nbands = F->nbands; A.buf0_bytes = F->buf0;
                for (int l = F->a; l <= F->b; l++) {
                    const OrbxLevel &L = sp[i].lv[l];
                    FuseLevel &U = A.lv[l];
                    U.base = L.base; U.w = L.w; U.h = L.h; U.stride = L.stride; U.frame = L.frame_stride; U.tab = h->tabs[l];
                    U.nbx = (L.w + 3) / 4;
                    U.rcp_nbx = U.nbx > 1 ? (uint32_t)((1ull << 32) / (unsigned)U.nbx + 1) : 0u;
                    U.pitch = (int)align_up((size_t)L.w + 12, 16);
                }
                orbx_launch_resize_fused(A, nf, (size_t)F->lds, st[i]);
            }
        }
        if (pe) HIPCHK(hipEventRecord(pe[1], s));
        for (int i = 0; i < nsub; i++) orbx_launch_fast(sp[i], sw[i], f0[i + 1] - f0[i], 0, P.ncells, st[i]);
    }
    if (pe) HIPCHK(hipEventRecord(pe[2], s));
    {   // dynamic LDS of the launch: the list arrays for the handle's largest list + this shape's fast-forward tables
        size_t lds = orbx_octree_lds_bytes(h->oct_cap_max, P.oct_ft, P.oct_map);
        if (lds > 150 * 1024) {              // no room for the tables beside very long lists: plain passes
            for (int i = 0; i < nsub; i++) { sp[i].oct_ft = 0; sp[i].oct_map = 0; for (int l = 0; l < h->nlevels; l++) sp[i].lv[l].fastD = 0; }
            lds = orbx_octree_lds_bytes(h->oct_cap_max, 0, 0);
        }
        for (int i = 0; i < nsub; i++) { sp[i].oct_cap_max = h->oct_cap_max; orbx_launch_octree(sp[i], sw[i], f0[i + 1] - f0[i], lds, st[i]); }
    }
    if (pe) HIPCHK(hipEventRecord(pe[3], s));
    for (int i = 0; i < nsub; i++)
        orbx_launch_describe(sp[i], sw[i], f0[i + 1] - f0[i], d_kps + (long long)f0[i] * P.out_cap,
                             d_desc + (long long)f0[i] * P.out_cap * 32, d_counts + f0[i], d_status + f0[i], st[i]);
    if (pe) HIPCHK(hipEventRecord(pe[4], s));
    for (int i = 1; i < nsub; i++) {
        HIPCHK(hipEventRecord(h->ev_join[i - 1], st[i]));
        HIPCHK(hipStreamWaitEvent(s, h->ev_join[i - 1], 0));
    }
    HIPCHK(hipGetLastError());
    if (h->profiling == 2) h->ring_calls++;
    guard.ok = true;
    return ORBX_OK;
}

static int finish_profile(orbx_extractor *h)
{
    if (h->profiling != 1) return ORBX_OK;
    HIPCHK(hipEventSynchronize(h->ev[4]));
    for (int i = 0; i < 4; i++) HIPCHK(hipEventElapsedTime(&h->stage_ms[i], h->ev[i], h->ev[i + 1]));
    return ORBX_OK;
}

extern "C" int orbx_extract_batch_device(orbx_extractor *h, const uint8_t *d_images, int nframes, int width,
                                         int height, int row_stride, size_t frame_stride,
                                         orbx_keypoint *d_keypoints, uint8_t *d_descriptors, int cap,
                                         int32_t *d_counts, int32_t *d_status, void *hip_stream)
{
    if (!h) return fail(ORBX_E_INVALID, "NULL handle");
    if (h->inflight) return fail(ORBX_E_INVALID, "an orbx_extract_begin call is in flight on this handle (its workspace would be overwritten)");
    if (!d_images || !d_keypoints || !d_descriptors || !d_counts || !d_status) return fail(ORBX_E_INVALID, "NULL device pointer");
    if (nframes < 1 || nframes > h->max_batch) return fail(ORBX_E_INVALID, "nframes=%d (max_batch=%d)", nframes, h->max_batch);
    if (width < 1 || height < 1 || row_stride < width) return fail(ORBX_E_INVALID, "bad frame geometry %dx%d stride %d", width, height, row_stride);
    // the kernels index one frame with 31-bit byte offsets and 24-bit row strides
    if (row_stride >= (1 << 23) || (long long)height * row_stride >= (1ll << 31))
        return fail(ORBX_E_SHAPE, "frame %dx%d with row stride %d: one frame must be below 2 GiB, the stride below 8 MiB", width, height, row_stride);
    if (cap != h->max_plan.out_cap) return fail(ORBX_E_CAPACITY, "device outputs must be laid out with cap == orbx_capacity() == %d (got %d)", h->max_plan.out_cap, cap);
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    int rc = enqueue(h, d_images, nframes, width, height, row_stride, (long long)frame_stride, d_keypoints,
                     d_descriptors, d_counts, d_status, s);
    if (rc != ORBX_OK) return rc;
    if (h->profiling == 1) return finish_profile(h);
    return ORBX_OK;
}

// repack frames [k0, k1) into the pinned staging buffer with the aligned pitch (a pitched hipMemcpy2D of a 1241-byte-wide image
// costs ~3 ms; this and ONE contiguous copy cost ~0.1 ms); rows are dealt to the staging threads in blocks
// Is [p, p + bytes) page-locked host memory (hipHostMalloc / hipHostRegister)?  Then the DMA engine can read it where it lies and
// the repack into the handle's pinned staging block -- 0.23 ms of the calling thread for 64 VGA frames -- is skipped.
static bool is_pinned_range(const void *p, size_t bytes)
{
    hipPointerAttribute_t a0, a1;
    if (hipPointerGetAttributes(&a0, p) != hipSuccess || hipPointerGetAttributes(&a1, (const uint8_t *)p + bytes - 1) != hipSuccess) {
        (void)hipGetLastError();   // an ordinary (pageable) pointer is "invalid value" to the runtime: not an error of this call
        return false;
    }
    return a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost;
}
// every frame's first and last byte is checked: frames carved out of several page-locked allocations with pageable gaps between
// them would pass a test of the block's two ends only (the copies are issued per frame or per chunk, never across such a gap
// unless the frames are contiguous -- and contiguous frames inside one registered range are what the per-frame test confirms)
static bool is_pinned_host(const uint8_t *images, int nframes, size_t frame_stride, size_t frame_bytes)
{
    for (int k = 0; k < nframes; k++)
        if (!is_pinned_range(images + (size_t)k * frame_stride, frame_bytes)) return false;
    return true;
}

// upload frames [k0, k1) straight from the caller's page-locked buffer into the handle's input block (rows are re-pitched by the copy)
static int upload_pinned(orbx_extractor *h, const uint8_t *images, int k0, int k1, int width, int height, int row_stride, size_t frame_stride, hipStream_t s)
{
    const bool tall = frame_stride == (size_t)row_stride * height && h->in_frame == (size_t)h->in_stride * height;   // the chunk is one tall image
    if (tall) {
        HIPCHK(hipMemcpy2DAsync(h->d_input + (size_t)k0 * h->in_frame, (size_t)h->in_stride, images + (size_t)k0 * frame_stride, (size_t)row_stride,
                                (size_t)width, (size_t)height * (k1 - k0), hipMemcpyHostToDevice, s));
    } else {
        for (int k = k0; k < k1; k++)
            HIPCHK(hipMemcpy2DAsync(h->d_input + (size_t)k * h->in_frame, (size_t)h->in_stride, images + (size_t)k * frame_stride, (size_t)row_stride,
                                    (size_t)width, (size_t)height, hipMemcpyHostToDevice, s));
    }
    return ORBX_OK;
}

static void stage_frames(orbx_extractor *h, const uint8_t *images, int k0, int k1, int width, int height, int row_stride, size_t frame_stride)
{
    const int RB = 64, nblk = (height + RB - 1) / RB;
    auto unit = [&](int u) {
        const int k = k0 + u / nblk, y0 = (u % nblk) * RB, y1 = std::min(height, y0 + RB);
        uint8_t *dst = h->h_in + (size_t)k * h->in_frame;
        const uint8_t *src = images + (size_t)k * frame_stride;
        if (row_stride == h->in_stride) memcpy(dst + (size_t)y0 * row_stride, src + (size_t)y0 * row_stride, (size_t)row_stride * (y1 - y0 - 1) + width);
        else for (int y = y0; y < y1; y++) memcpy(dst + (size_t)y * h->in_stride, src + (size_t)y * row_stride, (size_t)width);
    };
    const int n = (k1 - k0) * nblk;
    if (h->pool) h->pool->parallel_for(n, unit);
    else for (int u = 0; u < n; u++) unit(u);
}

// hand frames [k0, k1) over to the caller; hc / hs / hk / hd = counts, status, keypoints, descriptors of frame k0 in pinned memory
static int deliver_batch(orbx_extractor *h, int k0, int k1, orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *counts,
                         const int32_t *hc, const int32_t *hs, const orbx_keypoint *hk, const uint8_t *hd)
{
    const int ocap = h->max_plan.out_cap;
    for (int k = k0; k < k1; k++) {
        const int st = hs[k - k0];
        if (st != ORBX_OK)
            return fail(st, "frame %d: device status %d (%s)", k, st,
                        st == ORBX_E_CAND_OVERFLOW ? "FAST candidate buffer overflow" :
                        st == ORBX_E_TREE_OVERFLOW ? "quadtree arena overflow" : "capacity");
        if (hc[k - k0] > cap) return fail(ORBX_E_CAPACITY, "frame %d produced %d keypoints, caller capacity %d (use orbx_capacity())", k, hc[k - k0], cap);
    }
    auto unit = [&](int u) {
        const int k = k0 + u, n = hc[u];
        memcpy(keypoints + (size_t)k * cap, hk + (size_t)u * ocap, sizeof(orbx_keypoint) * n);
        memcpy(descriptors + (size_t)k * cap * 32, hd + (size_t)u * ocap * 32, (size_t)32 * n);
        counts[k] = n;
    };
    if (h->pool && k1 - k0 >= 8) h->pool->parallel_for(k1 - k0, unit);
    else for (int u = 0; u < k1 - k0; u++) unit(u);
    return ORBX_OK;
}

// The whole batch in one piece: upload, kernels, download, strictly one after the other (first call of a shape, profiling,
// small batches).
static int extract_batch_simple(orbx_extractor *h, const uint8_t *images, int nframes, int width, int height, int row_stride,
                                size_t frame_stride, orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *counts)
{
    hipStream_t s = h->stream;
    const int ocap = h->max_plan.out_cap;
    stage_frames(h, images, 0, nframes, width, height, row_stride, frame_stride);
    HIPCHK(hipMemcpyAsync(h->d_input, h->h_in, (size_t)(nframes - 1) * h->in_frame + (size_t)h->in_stride * height,
                          hipMemcpyHostToDevice, s));
    int rc = enqueue(h, h->d_input, nframes, width, height, h->in_stride, (long long)h->in_frame, h->d_kps, h->d_desc,
                     h->d_counts, h->d_status, s);
    if (rc != ORBX_OK) return rc;
    if (nframes == h->max_batch) {
        HIPCHK(hipMemcpyAsync(h->h_out, h->d_out, h->out_bytes, hipMemcpyDeviceToHost, s));
    } else {
        HIPCHK(hipMemcpyAsync(h->h_out, h->d_out, h->out_hdr, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h->h_kps, h->d_kps, sizeof(orbx_keypoint) * (size_t)ocap * nframes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h->h_desc, h->d_desc, (size_t)32 * ocap * nframes, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    rc = finish_profile(h);
    if (rc != ORBX_OK) return rc;
    return deliver_batch(h, 0, nframes, keypoints, descriptors, cap, counts, h->h_counts, h->h_status, h->h_kps, h->h_desc);
}

// Chunk c's outputs live in a block of their own, [counts nf | status nf | (pad to 256) | keypoints nf x cap | descriptors nf x cap x 32]
// at chunk_off[c] of d_out / h_out, so that they come back with ONE copy (a device-to-host copy on a compute stream is ~8 us of
// stream time whatever its size; a chunk had four).
struct ChunkBlock { int32_t *counts, *status; orbx_keypoint *kps; uint8_t *desc; size_t bytes; };
static ChunkBlock chunk_block(const orbx_extractor *h, uint8_t *base, int c, int nf)
{
    const int ocap = h->max_plan.out_cap;
    uint8_t *b = base + h->chunk_off[c];
    const size_t hdr = align_up((size_t)2 * nf * sizeof(int32_t), 256);
    ChunkBlock B;
    B.counts = reinterpret_cast<int32_t *>(b); B.status = B.counts + nf;
    B.kps = reinterpret_cast<orbx_keypoint *>(b + hdr);
    B.desc = b + hdr + (size_t)nf * ocap * sizeof(orbx_keypoint);
    B.bytes = hdr + (size_t)nf * ocap * (sizeof(orbx_keypoint) + 32);
    return B;
}

// kernels + download of frames [k0, k1) = chunk c of the handle's own input block, on stream s
static int enqueue_chunk(orbx_extractor *h, int c, int k0, int k1, int width, int height, hipStream_t s)
{
    const int nf = k1 - k0;
    const ChunkBlock D = chunk_block(h, h->d_out, c, nf);
    int rc = enqueue(h, h->d_input + (size_t)k0 * h->in_frame, nf, width, height, h->in_stride, (long long)h->in_frame,
                     D.kps, D.desc, D.counts, D.status, s, k0);
    if (rc != ORBX_OK) return rc;
    HIPCHK(hipMemcpyAsync(h->h_out + h->chunk_off[c], h->d_out + h->chunk_off[c], D.bytes, hipMemcpyDeviceToHost, s));
    return ORBX_OK;
}

extern "C" int orbx_extract_batch(orbx_extractor *h, const uint8_t *images, int nframes, int width, int height,
                                  int row_stride, size_t frame_stride, orbx_keypoint *keypoints,
                                  uint8_t *descriptors, int cap, int *counts)
{
    if (!h) return fail(ORBX_E_INVALID, "NULL handle");
    if (!counts) return fail(ORBX_E_INVALID, "counts is NULL");
    for (int k = 0; k < std::max(nframes, 0); k++) counts[k] = 0;
    if (h->inflight) return fail(ORBX_E_INVALID, "an orbx_extract_begin call is in flight on this handle");
    if (!images || width <= 0 || height <= 0 || nframes <= 0) return ORBX_OK;   // :1048 empty image: silent return
    if (!keypoints || !descriptors) return fail(ORBX_E_INVALID, "NULL output buffer");
    if (nframes > h->max_batch) return fail(ORBX_E_INVALID, "nframes=%d (max_batch=%d)", nframes, h->max_batch);
    if (row_stride < width) return fail(ORBX_E_INVALID, "row_stride %d < width %d", row_stride, width);
    if (width > h->max_w || height > h->max_h) return fail(ORBX_E_SHAPE, "frame %dx%d exceeds the handle's max %dx%d", width, height, h->max_w, h->max_h);
    HIPCHK(hipSetDevice(h->device));
    if (!h->pool && nframes >= 8) {                         // staging threads: up to 6, leaving cores to the caller
        const unsigned hc = (unsigned)StagePool::usable_cpus();       // affinity mask and cgroup quota, not the machine's core count
        h->pool = new StagePool((int)std::min<unsigned>(6u, hc > 2 ? hc / 2 - 1 : 0u));
    }
    // Chunked pipeline ("host-buffer batches"): while chunk c is uploaded and extracted, chunk c + 1 is repacked by the
    // staging threads; chunks alternate between two streams, so the upload of one runs under the kernels of the other, and each
    // chunk's ten kernels + four downloads are ONE graph launch (a launch is ~5 us of host time, a chunk would be ~70).  Every
    // chunk works in its own slice of the workspace.  The first call of a shape, profiling runs and small batches take the
    // plain path.
    const int chunk = h->batch_chunk;
    const bool chunked = chunk > 0 && nframes >= 2 * chunk && h->profiling == 0 && !h->need_clear && width == h->cur_w && height == h->cur_h;
    if (!chunked) return extract_batch_simple(h, images, nframes, width, height, row_stride, frame_stride, keypoints, descriptors, cap, counts);

    // chunk boundaries: a half-size first chunk (the pipeline starts sooner) and a half-size last one (the tail that nothing hides
    // is shorter) around full-size ones
    std::vector<int> cut(1, 0);
    static const bool equal_chunks = getenv("ORBX_BATCH_EQUAL") != nullptr;   // A/B switch
    if (nframes >= 3 * chunk && chunk >= 2 && !equal_chunks) {
        cut.push_back(chunk / 2);
        while (nframes - cut.back() > chunk + chunk / 2) cut.push_back(cut.back() + chunk);
        if (nframes - cut.back() > chunk / 2) cut.push_back(nframes - chunk / 2);
    } else {
        while (nframes - cut.back() > chunk) cut.push_back(cut.back() + chunk);
    }
    cut.push_back(nframes);
    const int nch = (int)cut.size() - 1;
    {   // every chunk's output block (the graphs below are captured with these addresses; they depend on (nframes, chunk) only)
        h->chunk_off.assign(nch, 0);
        size_t off = 0;
        for (int c = 0; c < nch; c++) {
            const int nf = cut[c + 1] - cut[c];
            h->chunk_off[c] = off;
            off += align_up((size_t)2 * nf * sizeof(int32_t), 256) + (size_t)nf * h->max_plan.out_cap * (sizeof(orbx_keypoint) + 32);
        }
        if (off > h->out_bytes) return fail(ORBX_E_INVALID, "internal: chunk output blocks exceed the staging block");
    }
    hipStream_t st[3] = {h->stream, h->aux[0], h->aux[2]};
    static const int nst = [] { const char *e = getenv("ORBX_BATCH_STREAMS"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : v > 3 ? 3 : v; }();   // compute streams the chunks rotate over
    const bool have_graphs = !h->bg_off && h->bg_w == width && h->bg_h == height && h->bg_n == nframes && h->bg_chunk == chunk && (int)h->bgraph.size() == nch;
    if (!have_graphs && !h->bg_off) {                       // (re)build the per-chunk graphs for this shape
        for (auto &g : h->bgraph) if (g) (void)hipGraphExecDestroy(g);
        h->bgraph.assign(nch, nullptr);
        bool ok = true;
        std::lock_guard<std::recursive_mutex> lk_(capture_mutex());
        for (int c = 0; c < nch && ok; c++) {
            hipStream_t s = st[c % nst];
            ok = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (!ok) break;
            const int rc = enqueue_chunk(h, c, cut[c], cut[c + 1], width, height, s);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(s, &g);
            ok = rc == ORBX_OK && e == hipSuccess && g && hipGraphInstantiate(&h->bgraph[c], g, nullptr, nullptr, 0) == hipSuccess;
            if (g) (void)hipGraphDestroy(g);
        }
        if (ok) { h->bg_w = width; h->bg_h = height; h->bg_n = nframes; h->bg_chunk = chunk; }
        else {
            (void)hipGetLastError();
            for (auto &g : h->bgraph) if (g) (void)hipGraphExecDestroy(g);
            h->bgraph.clear(); h->bg_off = true; h->bg_w = h->bg_h = h->bg_n = 0;
        }
    }
    const bool graphs = !h->bg_off && (int)h->bgraph.size() == nch;
    static const bool trace = getenv("ORBX_BATCH_TRACE") != nullptr;      // host-side time split of a call, to stderr
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // uploads go back to back on a stream of their own (on a chunk's compute stream the upload of chunk c + 2 would wait for
    // chunk c's kernels and downloads); each chunk's kernels wait for its own upload only
    hipStream_t up = h->aux[1];
    while ((int)h->ev_up.size() < nch) {
        hipEvent_t a = nullptr, b = nullptr;
        HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming)); h->ev_up.push_back(a);
        HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming)); h->ev_done.push_back(b);
    }
    // page-locked caller memory is uploaded where it lies (ORBX_BATCH_PINNED=0 keeps the staging copy: A/B switch)
    static const bool pinned_ok = [] { const char *e = getenv("ORBX_BATCH_PINNED"); return !e || atoi(e) != 0; }();
    // the answer for one buffer is remembered (a capture pipeline hands over the same page-locked block again and again; should it
    // have been unregistered meanwhile, the copies below are still correct -- the runtime stages them -- only slower)
    const size_t frame_bytes = (size_t)(height - 1) * row_stride + width;
    if (pinned_ok && !(h->pin_ptr == images && h->pin_n == nframes && h->pin_stride == frame_stride && h->pin_bytes == frame_bytes)) {
        h->pin_ptr = images; h->pin_n = nframes; h->pin_stride = frame_stride; h->pin_bytes = frame_bytes;
        h->pin_is = is_pinned_host(images, nframes, frame_stride, frame_bytes);
    }
    const bool pinned_in = pinned_ok && h->pin_is;
    double t_stage = 0, t_launch = 0;
    const double t_begin = trace ? now() : 0;
    for (int c = 0; c < nch; c++) {
        const int k0 = cut[c], k1 = cut[c + 1];
        hipStream_t s = st[c % nst];
        const double ta = trace ? now() : 0;
        if (!pinned_in) stage_frames(h, images, k0, k1, width, height, row_stride, frame_stride);
        const double tb = trace ? now() : 0;
        if (pinned_in) { int rcu = upload_pinned(h, images, k0, k1, width, height, row_stride, frame_stride, up); if (rcu != ORBX_OK) return rcu; }
        else
            HIPCHK(hipMemcpyAsync(h->d_input + (size_t)k0 * h->in_frame, h->h_in + (size_t)k0 * h->in_frame,
                                  (size_t)(k1 - k0 - 1) * h->in_frame + (size_t)h->in_stride * height, hipMemcpyHostToDevice, up));
        HIPCHK(hipEventRecord(h->ev_up[c], up));
        HIPCHK(hipStreamWaitEvent(s, h->ev_up[c], 0));
        if (graphs) HIPCHK(hipGraphLaunch(h->bgraph[c], s));
        else { int rc = enqueue_chunk(h, c, k0, k1, width, height, s); if (rc != ORBX_OK) return rc; }
        HIPCHK(hipEventRecord(h->ev_done[c], s));
        if (trace) { t_stage += tb - ta; t_launch += now() - tb; }
    }
    const double t_issued = trace ? now() : 0;
    // hand the chunks over as they finish: the copy-out of chunk c runs under the kernels of the chunks behind it
    double t_wait = 0;
    int rcd = ORBX_OK;
    for (int c = 0; c < nch; c++) {
        const int k0 = cut[c], k1 = cut[c + 1];
        const double tw = trace ? now() : 0;
        HIPCHK(hipEventSynchronize(h->ev_done[c]));
        if (trace) t_wait += now() - tw;
        if (rcd == ORBX_OK) {
            const ChunkBlock Hb = chunk_block(h, h->h_out, c, k1 - k0);
            rcd = deliver_batch(h, k0, k1, keypoints, descriptors, cap, counts, Hb.counts, Hb.status, Hb.kps, Hb.desc);
        }
    }
    HIPCHK(hipStreamSynchronize(up));
    h->last_input = h->d_input; h->last_in_stride = h->in_stride; h->last_in_frame = (long long)h->in_frame; h->last_batch = nframes;
    if (trace)
        fprintf(stderr, "orbx_extract_batch %d frames, %d chunks%s%s: stage %.3f ms, launch %.3f, wait %.3f, deliver %.3f, total %.3f\n", nframes, nch,
                graphs ? " (graphs)" : "", pinned_in ? " (page-locked input)" : "", t_stage, t_launch, t_wait, now() - t_issued - t_wait, now() - t_begin);
    return rcd;
}

extern "C" int orbx_extract_begin(orbx_extractor *h, const uint8_t *image, int width, int height, int stride)
{
    if (!h) return fail(ORBX_E_INVALID, "NULL handle");
    if (h->inflight) return fail(ORBX_E_INVALID, "orbx_extract_begin: a call is already in flight on this handle");
    h->inflight_frames = 0;
    if (!image || width <= 0 || height <= 0) { h->inflight = 1; return ORBX_OK; }     // :1048 empty image
    if (stride < width) return fail(ORBX_E_INVALID, "row_stride %d < width %d", stride, width);
    if (width > h->max_w || height > h->max_h) return fail(ORBX_E_SHAPE, "frame %dx%d exceeds the handle's max %dx%d", width, height, h->max_w, h->max_h);
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (stride == h->in_stride) memcpy(h->h_in, image, (size_t)stride * height);
    else for (int y = 0; y < height; y++) memcpy(h->h_in + (size_t)y * h->in_stride, image + (size_t)y * stride, (size_t)width);
    // Same shape as the last call, nothing to clear, no profiling: the upload, the kernels and the download are one HIP graph
    // (captured on the second call of a shape, replayed from then on); every pointer in it belongs to the handle.
    const bool graphable = !h->graph_off && h->profiling == 0 && h->max_batch == 1 && !h->need_clear && width == h->cur_w && height == h->cur_h;
    if (graphable && h->graph_exec && h->graph_w == width && h->graph_h == height) {
        HIPCHK(hipGraphLaunch(h->graph_exec, s));
        h->inflight = 1; h->inflight_frames = 1;
        return ORBX_OK;
    }
    bool capturing = false;
    std::unique_lock<std::recursive_mutex> caplk(capture_mutex(), std::defer_lock);   // see capture_mutex()
    if (graphable && h->graph_seen_w == width && h->graph_seen_h == height) {
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
        caplk.lock();
        capturing = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess;
        if (!capturing) { (void)hipGetLastError(); caplk.unlock(); if (++h->graph_fails >= 3) h->graph_off = true; }
    }
    h->graph_seen_w = width; h->graph_seen_h = height;
    int rc = ORBX_OK;
    hipError_t e1 = hipMemcpyAsync(h->d_input, h->h_in, (size_t)h->in_stride * height, hipMemcpyHostToDevice, s);
    if (e1 == hipSuccess) rc = enqueue(h, h->d_input, 1, width, height, h->in_stride, (long long)h->in_frame, h->d_kps, h->d_desc, h->d_counts, h->d_status, s);
    if (capturing) {
        hipError_t e2 = (e1 == hipSuccess && rc == ORBX_OK) ? hipMemcpyAsync(h->h_out, h->d_out, h->out_bytes, hipMemcpyDeviceToHost, s) : hipErrorUnknown;
        hipGraph_t g = nullptr;
        hipError_t e3 = hipStreamEndCapture(s, &g);
        if (e2 == hipSuccess && e3 == hipSuccess && g && hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
            h->graph_w = width; h->graph_h = height;
        } else {
            // a capture that failed or was invalidated (another thread's synchronous call) is not an error of this call: nothing has
            // run yet, the call runs plainly below; the capture is tried again on a later call, three times at most
            h->graph_exec = nullptr; (void)hipGetLastError();
            if (++h->graph_fails >= 3) h->graph_off = true; else h->graph_seen_w = h->graph_seen_h = 0;
        }
        if (g) (void)hipGraphDestroy(g);
        caplk.unlock();
        // nothing ran during the capture: run this call now, through the graph or (if that failed) plainly
        if (h->graph_exec) { HIPCHK(hipGraphLaunch(h->graph_exec, s)); h->inflight = 1; h->inflight_frames = 1; return ORBX_OK; }
        e1 = hipMemcpyAsync(h->d_input, h->h_in, (size_t)h->in_stride * height, hipMemcpyHostToDevice, s);
        rc = ORBX_OK;
        if (e1 == hipSuccess) rc = enqueue(h, h->d_input, 1, width, height, h->in_stride, (long long)h->in_frame, h->d_kps, h->d_desc, h->d_counts, h->d_status, s);
    }
    if (e1 != hipSuccess) return fail(ORBX_E_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e1));
    if (rc != ORBX_OK) return rc;
    const int ocap = h->max_plan.out_cap;
    if (h->max_batch == 1) {
        HIPCHK(hipMemcpyAsync(h->h_out, h->d_out, h->out_bytes, hipMemcpyDeviceToHost, s));
    } else {
        HIPCHK(hipMemcpyAsync(h->h_out, h->d_out, h->out_hdr, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h->h_kps, h->d_kps, sizeof(orbx_keypoint) * (size_t)ocap, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h->h_desc, h->d_desc, (size_t)32 * ocap, hipMemcpyDeviceToHost, s));
    }
    h->inflight = 1; h->inflight_frames = 1;
    return ORBX_OK;
}

extern "C" int orbx_extract_end(orbx_extractor *h, orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *n)
{
    if (!h || !n) return fail(ORBX_E_INVALID, "NULL argument");
    *n = 0;
    if (!h->inflight) return fail(ORBX_E_INVALID, "orbx_extract_end without orbx_extract_begin");
    h->inflight = 0;
    if (h->inflight_frames == 0) return ORBX_OK;
    if (!keypoints || !descriptors) return fail(ORBX_E_INVALID, "NULL output buffer");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    int rc = finish_profile(h);
    if (rc != ORBX_OK) return rc;
    if (h->h_status[0] != ORBX_OK)
        return fail(h->h_status[0], "device status %d (%s)", h->h_status[0],
                    h->h_status[0] == ORBX_E_CAND_OVERFLOW ? "FAST candidate buffer overflow" :
                    h->h_status[0] == ORBX_E_TREE_OVERFLOW ? "quadtree arena overflow" : "capacity");
    const int cnt = h->h_counts[0];
    if (cnt > cap) return fail(ORBX_E_CAPACITY, "%d keypoints, caller capacity %d (use orbx_capacity())", cnt, cap);
    memcpy(keypoints, h->h_kps, sizeof(orbx_keypoint) * cnt);
    memcpy(descriptors, h->h_desc, (size_t)32 * cnt);
    *n = cnt;
    return ORBX_OK;
}

extern "C" int orbx_extract(orbx_extractor *h, const uint8_t *image, int width, int height, int stride,
                            orbx_keypoint *keypoints, uint8_t *descriptors, int cap, int *n)
{
    if (!n) return fail(ORBX_E_INVALID, "n is NULL");
    if (h && h->max_batch == 1 && h->profiling == 0 && !h->inflight) {      // the two halves back to back: one HIP graph per shape
        *n = 0;
        int rc = orbx_extract_begin(h, image, width, height, stride);
        if (rc != ORBX_OK) return rc;
        if (!keypoints || !descriptors) { h->inflight = 0; if (h->inflight_frames) (void)hipStreamSynchronize(h->stream); return fail(ORBX_E_INVALID, "NULL output buffer"); }
        return orbx_extract_end(h, keypoints, descriptors, cap, n);
    }
    return orbx_extract_batch(h, image, 1, width, height, stride, (size_t)stride * (size_t)std::max(height, 0),
                              keypoints, descriptors, cap, n);
}

extern "C" int orbx_level_size(const orbx_extractor *h, int level, int *width, int *height)
{
    if (!h || level < 0 || level >= h->nlevels || h->cur_w == 0) return fail(ORBX_E_INVALID, "no plan / bad level");
    if (width) *width = h->plan.lv[level].w;
    if (height) *height = h->plan.lv[level].h;
    return ORBX_OK;
}

// ---- the batched-frames mode sharded over several handles (BASELINE north_star: "a batched-frames mode shards
// independent frames across the GPUs of one node") behind the C ABI: ONE process, one host thread and one set of streams per
// handle.  Handle i (normally one per device; several on one device also work) takes the i-th contiguous block of the batch --
// the split of my-slam_amd/shard.py's shard_range, sizes differ by at most one -- and runs orbx_extract_batch on it; every block's
// outputs land in the caller's flat arrays at its frames' positions, so the result is the one-handle result whatever the split.
// No collective: the frames are independent and the outputs are host arrays.  (RCCL carries only the device-resident gather of
// bench.py's N-rank mode, where results stay in HBM.) ----
extern "C" int orbx_extract_batch_multi(orbx_extractor *const *handles, int nhandles, const uint8_t *images, int nframes, int width,
                                        int height, int row_stride, size_t frame_stride, orbx_keypoint *keypoints,
                                        uint8_t *descriptors, int cap, int *counts)
{
    if (!handles || nhandles < 1) return fail(ORBX_E_INVALID, "no handles");
    for (int i = 0; i < nhandles; i++) {
        if (!handles[i]) return fail(ORBX_E_INVALID, "handle %d is NULL", i);
        for (int j = 0; j < i; j++)
            if (handles[j] == handles[i]) return fail(ORBX_E_INVALID, "handle %d is handle %d again: a handle is not re-entrant", i, j);
    }
    if (!counts) return fail(ORBX_E_INVALID, "counts is NULL");
    for (int k = 0; k < std::max(nframes, 0); k++) counts[k] = 0;
    if (!images || width <= 0 || height <= 0 || nframes <= 0) return ORBX_OK;
    if (!keypoints || !descriptors) return fail(ORBX_E_INVALID, "NULL output buffer");
    const int nh = std::min(nhandles, nframes);
    std::vector<int> lo(nh + 1);
    for (int i = 0; i <= nh; i++) lo[i] = i * (nframes / nh) + std::min(i, nframes % nh);
    std::vector<int> rc(nh, ORBX_OK);
    std::vector<std::string> msg(nh);
    auto run = [&](int i) {
        const int a = lo[i], n = lo[i + 1] - a;
        rc[i] = orbx_extract_batch(handles[i], images + (size_t)a * frame_stride, n, width, height, row_stride, frame_stride,
                                   keypoints + (size_t)a * cap, descriptors + (size_t)a * cap * 32, cap, counts + a);
        if (rc[i] != ORBX_OK) msg[i] = orbx_last_error();          // the error text is thread-local: carry it over
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nh; i++) th.emplace_back(run, i);
    run(0);
    for (auto &t : th) t.join();
    for (int i = 0; i < nh; i++)
        if (rc[i] != ORBX_OK) return fail(rc[i], "block %d (frames %d..%d, device %d): %s", i, lo[i], lo[i + 1] - 1, handles[i]->device, msg[i].c_str());
    return ORBX_OK;
}

static inline int reflect101_host(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

extern "C" int orbx_download_level(orbx_extractor *h, int frame, int level, uint8_t *dst, int dst_stride, int border)
{
    if (!h || !dst || level < 0 || level >= h->nlevels || h->cur_w == 0 || frame < 0 || frame >= h->last_batch || border < 0)
        return fail(ORBX_E_INVALID, "bad download_level argument");
    HIPCHK(hipSetDevice(h->device));
    OrbxLevel L = h->plan.lv[level];
    if (level == 0) { L.base = const_cast<uint8_t *>(h->last_input); L.stride = h->last_in_stride; L.frame_stride = h->last_in_frame; }
    if (dst_stride < L.w + 2 * border) return fail(ORBX_E_INVALID, "dst_stride too small");
    HIPCHK(hipDeviceSynchronize());   // the last call may have run on a caller stream and the aux streams
    uint8_t *interior = dst + (size_t)border * dst_stride + border;
    HIPCHK(hipMemcpy2D(interior, dst_stride, L.base + (size_t)frame * L.frame_stride, L.stride, L.w, L.h, hipMemcpyDeviceToHost));
    if (border > 0) {   // copyMakeBorder(..., BORDER_REFLECT_101), :1127-1133
        for (int y = -border; y < L.h + border; y++) {
            const uint8_t *srow = interior + (ptrdiff_t)reflect101_host(y, L.h) * dst_stride;
            uint8_t *drow = interior + (ptrdiff_t)y * dst_stride;
            if (y < 0 || y >= L.h) memcpy(drow, srow, (size_t)L.w);
            for (int x = 1; x <= border; x++) {
                drow[-x] = srow[reflect101_host(-x, L.w)];
                drow[L.w - 1 + x] = srow[reflect101_host(L.w - 1 + x, L.w)];
            }
        }
    }
    return ORBX_OK;
}

// Every level of frame `frame` of the last call in one go: one asynchronous copy per level into a page-locked staging block, ONE
// synchronisation, then the rows are laid out in the caller's buffers and the reflect-101 border is rebuilt on the host
// (src/ORBextractor.cc:1115-1133).  This is what keeps ORBextractor::mvImagePyramid valid after every operator() in the adapter.
extern "C" int orbx_download_pyramid(orbx_extractor *h, int frame, uint8_t *const *dst, const int *dst_stride, int border)
{
    if (!h || !dst || !dst_stride || h->cur_w == 0 || frame < 0 || frame >= h->last_batch || border < 0)
        return fail(ORBX_E_INVALID, "bad download_pyramid argument");
    HIPCHK(hipSetDevice(h->device));
    size_t need = 0, off[ORBX_MAX_LEVELS];
    for (int l = 0; l < h->nlevels; l++) {
        const OrbxLevel &L = h->plan.lv[l];
        if (!dst[l] || dst_stride[l] < L.w + 2 * border) return fail(ORBX_E_INVALID, "level %d: NULL buffer or dst_stride too small", l);
        off[l] = need;
        need += ((size_t)L.w * L.h + 255) & ~(size_t)255;
    }
    if (need > h->h_pyr_bytes) {
        (void)hipHostFree(h->h_pyr); h->h_pyr = nullptr; h->h_pyr_bytes = 0;
        HIPCHK(hipHostMalloc((void **)&h->h_pyr, need + need / 4));
        h->h_pyr_bytes = need + need / 4;
    }
    HIPCHK(hipDeviceSynchronize());   // the last call may have run on a caller stream and the aux streams
    for (int l = 0; l < h->nlevels; l++) {
        OrbxLevel L = h->plan.lv[l];
        if (l == 0) { L.base = const_cast<uint8_t *>(h->last_input); L.stride = h->last_in_stride; L.frame_stride = h->last_in_frame; }
        HIPCHK(hipMemcpy2DAsync(h->h_pyr + off[l], (size_t)L.w, L.base + (size_t)frame * L.frame_stride, L.stride, L.w, L.h, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int l = 0; l < h->nlevels; l++) {
        const OrbxLevel &L = h->plan.lv[l];
        const int ds = dst_stride[l];
        uint8_t *interior = dst[l] + (size_t)border * ds + border;
        for (int y = 0; y < L.h; y++) memcpy(interior + (size_t)y * ds, h->h_pyr + off[l] + (size_t)y * L.w, (size_t)L.w);
        if (border > 0) {             // copyMakeBorder(..., BORDER_REFLECT_101), :1127-1133
            for (int y = 0; y < L.h; y++) {
                uint8_t *row = interior + (size_t)y * ds;
                for (int x = 1; x <= border; x++) { row[-x] = row[reflect101_host(-x, L.w)]; row[L.w - 1 + x] = row[reflect101_host(L.w - 1 + x, L.w)]; }
            }
            for (int y = 1; y <= border; y++) {
                memcpy(interior - (ptrdiff_t)y * ds - border, interior + (ptrdiff_t)reflect101_host(-y, L.h) * ds - border, (size_t)L.w + 2 * border);
                memcpy(interior + (ptrdiff_t)(L.h - 1 + y) * ds - border, interior + (ptrdiff_t)reflect101_host(L.h - 1 + y, L.h) * ds - border, (size_t)L.w + 2 * border);
            }
        }
    }
    return ORBX_OK;
}

extern "C" int orbx_download_candidates(orbx_extractor *h, int frame, int level, int32_t *xyr, int cap)
{
    if (!h || !xyr || level < 0 || level >= h->nlevels || h->cur_w == 0 || frame < 0 || frame >= h->last_batch)
        return fail(ORBX_E_INVALID, "bad download_candidates argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    uint32_t n = 0;
    HIPCHK(hipMemcpy(&n, h->work.ncand + (size_t)frame * h->nlevels + level, sizeof(n), hipMemcpyDeviceToHost));
    const OrbxLevel &L = h->plan.lv[level];
    n = std::min<uint32_t>(n, (uint32_t)L.cand_cap);
    if ((int)n > cap) return fail(ORBX_E_CAPACITY, "%u candidates, capacity %d", n, cap);
    std::vector<OrbxCand> tmp(n ? n : 1);
    HIPCHK(hipMemcpy(tmp.data(), h->work.cand + (size_t)frame * h->plan.cand_frame + L.cand_off, sizeof(OrbxCand) * n, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
        xyr[3 * i] = (int32_t)(tmp[i].xy & 0xFFFFu);
        xyr[3 * i + 1] = (int32_t)(tmp[i].xy >> 16);
        xyr[3 * i + 2] = (int32_t)tmp[i].resp;
    }
    return (int)n;
}

// pyramid description of the last call (frame 0) for the stereo matcher (orbx_stereo.hip)
int orbx_internal_levels(orbx_extractor *h, const uint8_t **base, int *w, int *hh, int *stride, float *scale, float *inv_scale, int *nlevels, int *device)
{
    if (!h || h->cur_w == 0 || h->last_batch < 1) return ORBX_E_INVALID;
    *nlevels = h->nlevels; *device = h->device;
    for (int l = 0; l < h->nlevels; l++) {
        const OrbxLevel &L = h->plan.lv[l];
        base[l] = l == 0 ? h->last_input : L.base;
        stride[l] = l == 0 ? h->last_in_stride : L.stride;
        w[l] = L.w; hh[l] = L.h;
        scale[l] = h->scale[l]; inv_scale[l] = h->inv_scale[l];
    }
    return ORBX_OK;
}
