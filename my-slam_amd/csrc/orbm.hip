// orbm.hip -- gfx950 Hamming matcher primitives + their C ABI (include/orbm.h).
// Reference: src/ORBmatcher.cc of WChen09/My-SLAM (DescriptorDistance :1647-1663, best/second-best
// loops :201-232 and siblings, ComputeThreeMaxima :1601-1642).  Integer/bitwise only: v_xor_b32 +
// v_bcnt_u32_b32 (popcount with accumulate); the train set goes through LDS tiles and is read back as
// wave-uniform broadcasts, so the kernel is VALU-bound, not HBM-bound (16 M pairs
// touch 256 KB).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "orbm_internal.h"
#include "orbm_accept.h"

static thread_local std::string g_merr;
int mfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_merr = buf;
    return code;
}
extern "C" const char *orbm_last_error(void) { return g_merr.c_str(); }

// ---- pinned staging arena (see orbm_internal.h) ----
int orbm_arena_begin(orbm_matcher *m)
{
    if (m->arena_want > m->arena_cap) {          // grow between calls only: nothing is in flight here
        MHIPCHK(hipStreamSynchronize(m->stream));
        (void)hipHostFree(m->arena); m->arena = nullptr; m->arena_cap = 0;
        const size_t cap = m->arena_want + m->arena_want / 2 + (64u << 10);
        MHIPCHK(hipHostMalloc((void **)&m->arena, cap, hipHostMallocDefault));
        (void)hipFree(m->d_arena); m->d_arena = nullptr;
        if (hipMalloc((void **)&m->d_arena, cap) != hipSuccess) { m->d_arena = nullptr; (void)hipGetLastError(); }
        m->arena_cap = cap;
    }
    m->arena_used = 0; m->arena_want = 0; m->npend = 0;
    return ORBX_OK;
}
static void *arena_take(orbm_matcher *m, size_t bytes)
{
    const size_t a = (bytes + 63) & ~(size_t)63;
    m->arena_want += a;
    if (m->arena_used + a > m->arena_cap) return nullptr;     // this call falls back to a pageable copy; the next one has room
    void *p = m->arena + m->arena_used;
    m->arena_used += a;
    return p;
}
int orbm_h2d(orbm_matcher *m, void *dev, const void *host, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return ORBX_OK;
    void *p = arena_take(m, bytes);
    if (p) { memcpy(p, host, bytes); host = p; }
    MHIPCHK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
    return ORBX_OK;
}
int orbm_d2h(orbm_matcher *m, void *host, const void *dev, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return ORBX_OK;
    void *p = m->npend < 8 ? arena_take(m, bytes) : nullptr;
    if (p) { m->pend[m->npend++] = {host, p, bytes}; host = p; }
    MHIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
    return ORBX_OK;
}
void *orbm_stage_in(orbm_matcher *m, const void *host, size_t bytes)
{
    if (!m->d_arena) { m->arena_want += (bytes + 63) & ~(size_t)63; return nullptr; }
    void *p = arena_take(m, bytes);
    if (!p) return nullptr;
    memcpy(p, host, bytes);
    return m->d_arena + ((uint8_t *)p - m->arena);
}
int orbm_flush_in(orbm_matcher *m, size_t from, hipStream_t s)
{
    if (m->arena_used > from)
        MHIPCHK(hipMemcpyAsync(m->d_arena + from, m->arena + from, m->arena_used - from, hipMemcpyHostToDevice, s));
    return ORBX_OK;
}
void *orbm_d2h_tmp(orbm_matcher *m, const void *dev, size_t bytes, hipStream_t s)
{
    void *p = arena_take(m, bytes);
    if (!p) return nullptr;
    if (hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return nullptr;
    return p;
}
int orbm_d2h_split(orbm_matcher *m, void *const *host, const size_t *parts, int nparts, const void *dev, hipStream_t s)
{
    size_t total = 0;
    for (int i = 0; i < nparts; i++) total += parts[i];
    if (total == 0) return ORBX_OK;
    uint8_t *p = (m->npend + nparts <= 8) ? (uint8_t *)arena_take(m, total) : nullptr;
    if (!p) {     // no room in the arena this call: one copy per part
        size_t o = 0;
        for (int i = 0; i < nparts; i++) {
            int rc = orbm_d2h(m, host[i], (const uint8_t *)dev + o, parts[i], s);
            if (rc != ORBX_OK) return rc;
            o += parts[i];
        }
        return ORBX_OK;
    }
    MHIPCHK(hipMemcpyAsync(p, dev, total, hipMemcpyDeviceToHost, s));
    size_t o = 0;
    for (int i = 0; i < nparts; i++) { m->pend[m->npend++] = {host[i], p + o, parts[i]}; o += parts[i]; }
    return ORBX_OK;
}
int orbm_sync(orbm_matcher *m, hipStream_t s)
{
    MHIPCHK(hipStreamSynchronize(s));
    for (int i = 0; i < m->npend; i++) memcpy(m->pend[i].dst, m->pend[i].src, m->pend[i].bytes);
    m->npend = 0;
    return ORBX_OK;
}

// ---- dense best/second-best: one query per thread (8 VGPRs); the train range of a workgroup is
// staged through double-buffered 4 KiB LDS tiles and read back as wave-uniform broadcasts
// (2 x ds_read_b128 per pair against 22 VALU ops).  gridDim.z splits the train range for occupancy; every split writes a partial
// (best key, second key) with key = distance << 22 | train index, so "strictly smaller wins, first
// index wins a tie, a tie with the best becomes the second best" (src/ORBmatcher.cc:214-223) is
// min / median on keys and partials merge exactly (k_merge_best2 / k_accept_rot). ----
#define M_TILE 128   // train descriptors staged per LDS tile (4 KiB)
__global__ __launch_bounds__(M_THREADS) void k_best2_dense(
    const uint8_t *__restrict__ q, const int32_t *__restrict__ nqv, int nq_fixed,
    const uint8_t *__restrict__ t, const int32_t *__restrict__ ntv, int nt_fixed,
    long long qstride, long long tstride, int out_stride, uint2 *__restrict__ part)
{
    __shared__ uint4 tile[2][M_TILE * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nq = nqv ? nqv[b] : nq_fixed;
    const int nt = ntv ? ntv[b] : nt_fixed;
    if ((int)(blockIdx.x * M_THREADS) >= nq) return;
    const int qi = blockIdx.x * M_THREADS + tid;
    const int S = gridDim.z, chunk = (nt + S - 1) / S;
    const int j0 = blockIdx.z * chunk, j1 = min(nt, j0 + chunk);
    const uint4 *Q = reinterpret_cast<const uint4 *>(q + (long long)b * qstride);
    const uint4 *T = reinterpret_cast<const uint4 *>(t + (long long)b * tstride);
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (qi < nq) { q0 = Q[2 * qi]; q1 = Q[2 * qi + 1]; }
    uint32_t bk = M_KEY_NONE, sk = M_KEY_NONE;
    // double-buffered tiles: the loads of tile k+1 are in flight while tile k is scanned
    int cnt = min(M_TILE, j1 - j0);
    if (tid < cnt * 2) tile[0][tid] = T[2 * j0 + tid];
    int buf = 0;
    for (int t0 = j0; t0 < j1; t0 += M_TILE) {
        const int ncnt = min(M_TILE, j1 - (t0 + M_TILE));
        uint4 pre = make_uint4(0, 0, 0, 0);
        if (tid < ncnt * 2) pre = T[2 * (t0 + M_TILE) + tid];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < cnt; j++) {
            const uint32_t key = ((uint32_t)hamming256(q0, q1, tile[buf][2 * j], tile[buf][2 * j + 1]) << 22) | (uint32_t)(t0 + j);
            sk = med3u(bk, sk, key);                            // second smallest of {bk <= sk, key}
            bk = min(bk, key);
        }
        if (tid < ncnt * 2) tile[buf ^ 1][tid] = pre;
        buf ^= 1;
        cnt = ncnt;
    }
    if (qi < nq) part[((long long)blockIdx.z * gridDim.y + b) * out_stride + qi] = make_uint2(bk, sk);
}

__global__ __launch_bounds__(M_THREADS) void k_merge_best2(const uint2 *__restrict__ part, int S, int nq,
                                                          int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d,
                                                          int32_t *__restrict__ second_d)
{
    const int i = blockIdx.x * M_THREADS + threadIdx.x;
    if (i >= nq) return;
    int bi, bd, sd;
    merge_partials(part, S, nq, i, bi, bd, sd);
    best_idx[i] = bi; best_d[i] = bd; second_d[i] = sd;
}

// ---- CSR best/second-best: one wave per query ----
__global__ __launch_bounds__(M_THREADS) void k_best2_csr(
    const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
    const int32_t *__restrict__ off, const int32_t *__restrict__ idx,
    int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d, int32_t *__restrict__ second_d)
{
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * (M_THREADS / 64) + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const uint4 *Q = reinterpret_cast<const uint4 *>(q) + 2 * (long long)qi;
    const uint4 q0 = Q[0], q1 = Q[1];
    const int lo = off[qi], hi = off[qi + 1];
    // pack = dist << 22 | position: min() picks the smallest distance, earliest position
    uint32_t bp = (256u << 22) | 0x3FFFFFu;
    int s = 256;
    for (int c = lo + lane; c < hi; c += 64) {
        const uint4 *Tj = reinterpret_cast<const uint4 *>(t) + 2 * (long long)idx[c];
        const int d = hamming256(q0, q1, Tj[0], Tj[1]);
        const uint32_t p = ((uint32_t)d << 22) | (uint32_t)min(c - lo, 0x3FFFFF);
        if (p < bp) { s = (int)(bp >> 22); bp = p; }
        else if (d < s) s = d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t op = __shfl_xor(bp, o);
        const int os = __shfl_xor(s, o);
        const int loser = (int)(max(bp, op) >> 22);
        bp = min(bp, op);
        s = min(min(s, os), loser);
    }
    if (lane == 0) {
        const int d = (int)(bp >> 22);
        best_d[qi] = d;
        second_d[qi] = s;
        best_idx[qi] = d < 256 ? idx[lo + (int)(bp & 0x3FFFFFu)] : -1;
    }
}

// ---- per-candidate distances ----
__global__ __launch_bounds__(M_THREADS) void k_dist_csr(
    const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t,
    const int32_t *__restrict__ off, const int32_t *__restrict__ idx, int total, int32_t *__restrict__ dist)
{
    const int c = blockIdx.x * M_THREADS + threadIdx.x;
    if (c >= total) return;
    int lo = 0, hi = nq;   // largest i with off[i] <= c
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= c) lo = mid; else hi = mid;
    }
    const uint4 *Q = reinterpret_cast<const uint4 *>(q) + 2 * (long long)lo;
    const uint4 *Tj = reinterpret_cast<const uint4 *>(t) + 2 * (long long)idx[c];
    dist[c] = hamming256(Q[0], Q[1], Tj[0], Tj[1]);
}

void orbm_launch_dist_csr(const uint8_t *d_q, int nq, const uint8_t *d_t, const int32_t *d_off, const int32_t *d_idx, int total,
                          int32_t *d_dist, hipStream_t s)
{
    if (total > 0)
        hipLaunchKernelGGL(k_dist_csr, dim3((unsigned)((total + M_THREADS - 1) / M_THREADS)), dim3(M_THREADS), 0, s, d_q, nq, d_t, d_off, d_idx, total, d_dist);
}

// pairs given explicitly as (query << 16 | train), both below 65536: no per-thread binary search over off[]
__global__ __launch_bounds__(M_THREADS) void k_dist_pairs16(const uint8_t *__restrict__ q, const uint8_t *__restrict__ t,
                                                            const uint32_t *__restrict__ pairs, int total, int32_t *__restrict__ dist)
{
    const int c = blockIdx.x * M_THREADS + threadIdx.x;
    if (c >= total) return;
    const uint32_t p = pairs[c];
    const uint4 *Q = reinterpret_cast<const uint4 *>(q) + 2 * (long long)(p >> 16);
    const uint4 *Tj = reinterpret_cast<const uint4 *>(t) + 2 * (long long)(p & 0xFFFFu);
    dist[c] = hamming256(Q[0], Q[1], Tj[0], Tj[1]);
}

__global__ __launch_bounds__(M_THREADS) void k_dist_dense(
    const uint8_t *__restrict__ q, int nq, const uint8_t *__restrict__ t, int nt, int32_t *__restrict__ dist)
{
    const long long c = (long long)blockIdx.x * M_THREADS + threadIdx.x;
    if (c >= (long long)nq * nt) return;
    const int i = (int)(c / nt), j = (int)(c - (long long)i * nt);
    const uint4 *Q = reinterpret_cast<const uint4 *>(q) + 2 * (long long)i;
    const uint4 *Tj = reinterpret_cast<const uint4 *>(t) + 2 * (long long)j;
    dist[c] = hamming256(Q[0], Q[1], Tj[0], Tj[1]);
}

// Many splits (few, large frame pairs): k_accept_rot is one workgroup per pair and would pull S partials per query through
// one CU.  This grid-wide pre-merge leaves one (best, second) key pair per query, which merges like a single partial.
__global__ __launch_bounds__(M_THREADS) void k_merge_keys(const uint2 *__restrict__ part, int S, const int32_t *__restrict__ nqv,
                                                         int cap, uint2 *__restrict__ merged)
{
    const int b = blockIdx.y, i = blockIdx.x * M_THREADS + threadIdx.x;
    if (i >= nqv[b] || i >= cap) return;
    const long long o = (long long)b * cap + i;
    uint32_t bk, sk;
    merge_partial_keys(part, S, (long long)gridDim.y * cap, o, bk, sk);
    merged[o] = make_uint2(bk, sk);
}

// ---- acceptance (:228-232) + rotation histogram (:236-246) + ComputeThreeMaxima + cull (:266-284): orbm_accept.h ----
__global__ __launch_bounds__(ACC_THREADS) void k_accept_rot(
    const int32_t *__restrict__ nqv, const orbx_keypoint *__restrict__ kq, const orbx_keypoint *__restrict__ kt,
    int cap, const uint2 *__restrict__ part, int S, int th, float nnratio, int check_ori,
    int32_t *__restrict__ match12, int32_t *__restrict__ nmatches,
    int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d, int32_t *__restrict__ second_d)
{
    __shared__ AcceptShared sh;
    accept_rot_body<ACC_THREADS>(sh, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x, nqv, kq, kt, cap, part, S, th, nnratio, check_ori,
                                 match12, nmatches, best_idx, best_d, second_d);
}

// -------------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------------
extern "C" int orbm_distance(const uint8_t a[32], const uint8_t b[32])
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        dist += __builtin_popcount(pa ^ pb);
    }
    return dist;
}

extern "C" void orbm_destroy(orbm_matcher *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipFree(m->d_q); (void)hipFree(m->d_t); (void)hipFree(m->d_off); (void)hipFree(m->d_idx);
    (void)hipFree(m->d_out); (void)hipFree(m->d_part);
    orbm_grid_free(m->grid); orbm_grid_free(m->grid2);
    (void)hipFree(m->d_qf); (void)hipFree(m->d_qi); (void)hipFree(m->d_skip);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    (void)hipHostFree(m->h_pin); (void)hipHostFree(m->arena); (void)hipFree(m->d_arena);
    delete m;
}

extern "C" int orbm_create(orbm_matcher **out, int device, int max_queries, int max_train, int max_pairs)
{
    if (!out) return mfail(ORBX_E_INVALID, "out is NULL");
    *out = nullptr;
    if (max_queries < 1 || max_train < 1 || max_pairs < 0) return mfail(ORBX_E_INVALID, "bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return mfail(ORBX_E_HIP, "no HIP device: liborbx has no CPU path");
    if (device < 0 || device >= ndev) return mfail(ORBX_E_INVALID, "device %d of %d", device, ndev);
    MHIPCHK(hipSetDevice(device));
    orbm_matcher *m = new orbm_matcher();
    m->device = device; m->max_q = max_queries; m->max_t = max_train; m->max_pairs = max_pairs;
    { const char *e = getenv("ORBM_DENSE"); m->dense_popcount = e && !strcmp(e, "popcount"); }
    const size_t outn = std::max<size_t>((size_t)3 * max_queries, (size_t)max_pairs);
    if (hipMalloc((void **)&m->d_q, (size_t)max_queries * 32) != hipSuccess ||
        hipMalloc((void **)&m->d_t, (size_t)max_train * 32) != hipSuccess ||
        hipMalloc((void **)&m->d_off, ((size_t)max_queries + 1) * 4) != hipSuccess ||
        hipMalloc((void **)&m->d_idx, std::max<size_t>((size_t)max_pairs, 1) * 4) != hipSuccess ||
        hipMalloc((void **)&m->d_out, outn * 4) != hipSuccess ||
        hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
        orbm_destroy(m);
        return mfail(ORBX_E_HIP, "matcher workspace allocation failed");
    }
    *out = m;
    return ORBX_OK;
}

void orbm_grid_free(OrbmGrid &g)
{
    (void)hipFree(g.kx); (void)hipFree(g.ky); (void)hipFree(g.koct); (void)hipFree(g.cell_start); (void)hipFree(g.items); (void)hipFree(g.cell_of);
    g.kx = g.ky = nullptr; g.koct = g.cell_start = g.items = g.cell_of = nullptr;
}

// Grows the workspace (never shrinks it).  The reference's matcher has no size limit (it works on std::vectors), so every entry
// point that finds its inputs larger than the handle grows the handle instead of refusing; a caller that knows its sizes calls this
// once up front and no call allocates.  Growing max_train drops the Frame grid in the handle (orbm_grid_build again).
extern "C" int orbm_reserve(orbm_matcher *m, int max_queries, int max_train, int max_pairs)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    const int nq = std::max(m->max_q, max_queries), nt = std::max(m->max_t, max_train), np = std::max(m->max_pairs, max_pairs);
    if (nq == m->max_q && nt == m->max_t && np == m->max_pairs) return ORBX_OK;
    MHIPCHK(hipSetDevice(m->device));
    MHIPCHK(hipStreamSynchronize(m->stream));
    if (nq > m->max_q) {
        (void)hipFree(m->d_q); (void)hipFree(m->d_off); m->d_q = nullptr; m->d_off = nullptr;
        MHIPCHK(hipMalloc((void **)&m->d_q, (size_t)nq * 32));
        MHIPCHK(hipMalloc((void **)&m->d_off, ((size_t)nq + 1) * 4));
    }
    if (nt > m->max_t) {
        (void)hipFree(m->d_t); m->d_t = nullptr;
        MHIPCHK(hipMalloc((void **)&m->d_t, (size_t)nt * 32));
        orbm_grid_free(m->grid); orbm_grid_free(m->grid2); m->grid_ok = false; m->grid2_ok = false;     // sized by max_train
        (void)hipFree(m->d_skip); m->d_skip = nullptr;
    }
    if (np > m->max_pairs) {
        (void)hipFree(m->d_idx); m->d_idx = nullptr;
        MHIPCHK(hipMalloc((void **)&m->d_idx, std::max<size_t>((size_t)np, 1) * 4));
    }
    const size_t out_old = std::max<size_t>((size_t)3 * m->max_q, (size_t)m->max_pairs), out_new = std::max<size_t>((size_t)3 * nq, (size_t)np);
    if (out_new > out_old) {
        (void)hipFree(m->d_out); m->d_out = nullptr;
        MHIPCHK(hipMalloc((void **)&m->d_out, out_new * 4));
    }
    m->max_q = nq; m->max_t = nt; m->max_pairs = np;
    return ORBX_OK;
}
int orbm_grow(orbm_matcher *m, long long need_q, long long need_t, long long need_pairs)
{
    if (need_q > (1ll << 28) || need_t > (1ll << 28) || need_pairs > (1ll << 30)) return mfail(ORBX_E_CAPACITY, "request beyond 2^28 descriptors / 2^30 pairs");
    auto up = [](long long need, int have) { return need > have ? (int)std::min<long long>(need + need / 2, 1ll << 30) : have; };
    return orbm_reserve(m, up(need_q, m->max_q), up(need_t, m->max_t), up(need_pairs, m->max_pairs));
}

static int check_csr(const int32_t *off, const int32_t *idx, int nq, int nt, int max_pairs, int *total)
{
    if (off[0] != 0) return mfail(ORBX_E_INVALID, "cand_off[0] must be 0");
    for (int i = 0; i < nq; i++)
        if (off[i + 1] < off[i]) return mfail(ORBX_E_INVALID, "cand_off not monotone at %d", i);
    *total = off[nq];
    (void)max_pairs;
    if (*total > 0 && !idx) return mfail(ORBX_E_INVALID, "cand_idx is NULL");
    for (int c = 0; c < *total; c++)
        if (idx[c] < 0 || idx[c] >= nt) return mfail(ORBX_E_INVALID, "cand_idx[%d]=%d outside [0,%d)", c, idx[c], nt);
    return ORBX_OK;
}

// train-range splits so that the launch has >= ~4 waves per SIMD (1024 SIMDs); <= 64
#define ORBM_PREMERGE_SPLITS 16
static int pick_splits(int nq_cap, int nbatch, int nt_hint)
{
    const long long waves = (long long)nbatch * ((nq_cap + M_THREADS - 1) / M_THREADS) * (M_THREADS / 64);
    // ~12 waves per SIMD over the launch (measured on MI355X: 2048..16384 waves -> 76, 69, 64, 62, 57, 58 us for 63 x 1007^2
    // pairs): more, shorter waves hide the LDS broadcast latency and even out the tail
    int S = (int)((12288 + waves - 1) / waves);
    S = std::max(1, std::min(S, 64));
    while (S > 1 && nt_hint / S < 16) S--;       // keep >= 16 train descriptors per split
    return S;
}
static int ensure_partials(orbm_matcher *m, size_t need, hipStream_t s = nullptr)
{
    if (need <= m->part_elems) return ORBX_OK;
    {   // growing means synchronise + free + allocate: not inside a stream capture (warm the handle up with the same arguments first)
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (s && hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive)
            return mfail(ORBX_E_INVALID, "the matcher's partial buffer must grow (%zu -> %zu pairs) while the stream is being captured: run the call once outside the capture", m->part_elems, need);
        (void)hipGetLastError();
    }
    MHIPCHK(hipDeviceSynchronize());
    (void)hipFree(m->d_part);
    m->d_part = nullptr; m->part_elems = 0;
    MHIPCHK(hipMalloc((void **)&m->d_part, need * sizeof(uint2)));
    m->part_elems = need;
    return ORBX_OK;
}

extern "C" int orbm_best2(orbm_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt,
                          const int32_t *cand_off, const int32_t *cand_idx,
                          int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (nq < 0 || nt < 0) return mfail(ORBX_E_INVALID, "nq=%d nt=%d", nq, nt);
    { int rc_ = orbm_grow(m, nq, nt, cand_off && nq > 0 ? cand_off[nq] : 0); if (rc_ != ORBX_OK) return rc_; }
    if (nq == 0) return ORBX_OK;
    if (!q || !best_idx || !best_d || !second_d || (nt > 0 && !t)) return mfail(ORBX_E_INVALID, "NULL buffer");
    MHIPCHK(hipSetDevice(m->device));
    { int rc_ = orbm_arena_begin(m); if (rc_ != ORBX_OK) return rc_; }
    hipStream_t s = m->stream;
    { int rc_ = orbm_h2d(m, m->d_q, q, (size_t)nq * 32, s); if (rc_ != ORBX_OK) return rc_; }
    if (nt > 0) { int rc_ = orbm_h2d(m, m->d_t, t, (size_t)nt * 32, s); if (rc_ != ORBX_OK) return rc_; }
    int32_t *o_bi = m->d_out, *o_bd = m->d_out + nq, *o_sd = m->d_out + 2 * (size_t)nq;
    if (cand_off) {
        int total = 0;
        int rc = check_csr(cand_off, cand_idx, nq, nt, m->max_pairs, &total);
        if (rc != ORBX_OK) return rc;
        { int rc_ = orbm_h2d(m, m->d_off, cand_off, ((size_t)nq + 1) * 4, s); if (rc_ != ORBX_OK) return rc_; }
        if (total > 0) { int rc_ = orbm_h2d(m, m->d_idx, cand_idx, (size_t)total * 4, s); if (rc_ != ORBX_OK) return rc_; }
        hipLaunchKernelGGL(k_best2_csr, dim3((nq + 3) / 4), dim3(M_THREADS), 0, s, m->d_q, nq, m->d_t, m->d_off, m->d_idx, o_bi, o_bd, o_sd);
    } else {
        const int S = m->dense_popcount ? pick_splits(nq, 1, nt) : orbm_mfma_splits(nq, nt, 1);
        int rc = ensure_partials(m, (size_t)S * nq);
        if (rc != ORBX_OK) return rc;
        if (m->dense_popcount)
            hipLaunchKernelGGL(k_best2_dense, dim3((nq + M_THREADS - 1) / M_THREADS, 1, S), dim3(M_THREADS), 0, s,
                               m->d_q, (const int32_t *)nullptr, nq, m->d_t, (const int32_t *)nullptr, nt, 0LL, 0LL, nq, m->d_part);
        else if ((rc = orbm_launch_dense_mfma(m, m->d_q, nullptr, nq, m->d_t, nullptr, nt, 0LL, 0LL, nq, std::max(nt, 1), 1, nq, S, m->d_part, s)) != ORBX_OK)
            return rc;
        hipLaunchKernelGGL(k_merge_best2, dim3((nq + M_THREADS - 1) / M_THREADS), dim3(M_THREADS), 0, s, m->d_part, S, nq, o_bi, o_bd, o_sd);
    }
    MHIPCHK(hipGetLastError());
    { int rc_ = orbm_d2h(m, best_idx, o_bi, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_d2h(m, best_d, o_bd, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_d2h(m, second_d, o_sd, (size_t)nq * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_sync(m, s); if (rc_ != ORBX_OK) return rc_; }
    return ORBX_OK;
}

extern "C" int orbm_distances(orbm_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt,
                              const int32_t *cand_off, const int32_t *cand_idx, int32_t *dist)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (nq < 0 || nt < 0) return mfail(ORBX_E_INVALID, "nq=%d nt=%d", nq, nt);
    { int rc_ = orbm_grow(m, nq, nt, cand_off && nq > 0 ? cand_off[nq] : (long long)nq * nt); if (rc_ != ORBX_OK) return rc_; }
    if (nq == 0 || nt == 0) return ORBX_OK;
    if (!q || !t || !dist) return mfail(ORBX_E_INVALID, "NULL buffer");
    MHIPCHK(hipSetDevice(m->device));
    { int rc_ = orbm_arena_begin(m); if (rc_ != ORBX_OK) return rc_; }
    hipStream_t s = m->stream;
    { int rc_ = orbm_h2d(m, m->d_q, q, (size_t)nq * 32, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_h2d(m, m->d_t, t, (size_t)nt * 32, s); if (rc_ != ORBX_OK) return rc_; }
    long long total;
    if (cand_off) {
        int tot = 0;
        int rc = check_csr(cand_off, cand_idx, nq, nt, m->max_pairs, &tot);
        if (rc != ORBX_OK) return rc;
        total = tot;
        if (total == 0) return ORBX_OK;
        { int rc_ = orbm_h2d(m, m->d_off, cand_off, ((size_t)nq + 1) * 4, s); if (rc_ != ORBX_OK) return rc_; }
        { int rc_ = orbm_h2d(m, m->d_idx, cand_idx, (size_t)total * 4, s); if (rc_ != ORBX_OK) return rc_; }
        hipLaunchKernelGGL(k_dist_csr, dim3((unsigned)((total + M_THREADS - 1) / M_THREADS)), dim3(M_THREADS), 0, s,
                           m->d_q, nq, m->d_t, m->d_off, m->d_idx, (int)total, m->d_out);
    } else {
        total = (long long)nq * nt;
        if (total > (long long)std::max<size_t>((size_t)3 * m->max_q, (size_t)m->max_pairs))
            return mfail(ORBX_E_CAPACITY, "dense distances need %lld ints, matcher sized for %d pairs", total, m->max_pairs);
        hipLaunchKernelGGL(k_dist_dense, dim3((unsigned)((total + M_THREADS - 1) / M_THREADS)), dim3(M_THREADS), 0, s,
                           m->d_q, nq, m->d_t, nt, m->d_out);
    }
    MHIPCHK(hipGetLastError());
    { int rc_ = orbm_d2h(m, dist, m->d_out, (size_t)total * 4, s); if (rc_ != ORBX_OK) return rc_; }
    { int rc_ = orbm_sync(m, s); if (rc_ != ORBX_OK) return rc_; }
    return ORBX_OK;
}

static int launch_dense_batch(orbm_matcher *m, const uint8_t *d_q, const int32_t *d_nq, const uint8_t *d_t,
                              const int32_t *d_nt, int cap, int nbatch, hipStream_t s, int *S_out)
{
    const int S = m->dense_popcount ? pick_splits(cap, nbatch, cap) : orbm_mfma_splits(cap, cap, nbatch);
    int rc = ensure_partials(m, (size_t)(S + 1) * nbatch * cap, s);     // + one slot per query for k_merge_keys
    if (rc != ORBX_OK) return rc;
    if (m->dense_popcount)
        hipLaunchKernelGGL(k_best2_dense, dim3((cap + M_THREADS - 1) / M_THREADS, nbatch, S), dim3(M_THREADS), 0, s,
                           d_q, d_nq, 0, d_t, d_nt, 0, (long long)cap * 32, (long long)cap * 32, cap, m->d_part);
    else if ((rc = orbm_launch_dense_mfma(m, d_q, d_nq, 0, d_t, d_nt, 0, (long long)cap * 32, (long long)cap * 32, cap, cap, nbatch, cap, S, m->d_part, s)) != ORBX_OK)
        return rc;
    MHIPCHK(hipGetLastError());
    *S_out = S;
    return ORBX_OK;
}

// all-ones acceptance (th = 256 never rejects a found match; nnratio huge) is not what best2 wants:
// k_accept_rot also exports the merged (best_idx, best_d, second_d) when given the arrays.
extern "C" int orbm_match_batch_device(orbm_matcher *m, const uint8_t *d_q, const orbx_keypoint *d_kq,
                                       const int32_t *d_nq, const uint8_t *d_t, const orbx_keypoint *d_kt,
                                       const int32_t *d_nt, int cap, int nbatch, int th, float nnratio,
                                       int check_orientation, int32_t *d_match12, int32_t *d_nmatches, void *hip_stream)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (!d_q || !d_nq || !d_t || !d_nt || !d_kq || !d_kt || !d_match12 || !d_nmatches) return mfail(ORBX_E_INVALID, "NULL device pointer");
    if (cap < 1 || nbatch < 1 || cap > 0x3FFFFF) return mfail(ORBX_E_INVALID, "cap=%d nbatch=%d", cap, nbatch);
    MHIPCHK(hipSetDevice(m->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : m->stream;
    int S = 1;
    // (The acceptance as the tail of the LAST workgroup of a pair inside the match launch -- arrival counter, agent-scope release /
    // acquire -- was built and measured in round 3: identical tables, 76.7 us against 22.9 + 6.6: every workgroup's release is an L2
    // write-back on this multi-XCD part.  It stays a launch of its own.)
    int rc = launch_dense_batch(m, d_q, d_nq, d_t, d_nt, cap, nbatch, s, &S);
    if (rc != ORBX_OK) return rc;
    const uint2 *part = m->d_part;
    if (S > ORBM_PREMERGE_SPLITS) {
        uint2 *merged = m->d_part + (size_t)S * nbatch * cap;
        hipLaunchKernelGGL(k_merge_keys, dim3((cap + M_THREADS - 1) / M_THREADS, nbatch), dim3(M_THREADS), 0, s, m->d_part, S, d_nq, cap, merged);
        part = merged; S = 1;
    }
    hipLaunchKernelGGL(k_accept_rot, dim3(nbatch), dim3(ACC_THREADS), 0, s, d_nq, d_kq, d_kt, cap, part, S,
                       th, nnratio, check_orientation, d_match12, d_nmatches,
                       (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
    MHIPCHK(hipGetLastError());
    return ORBX_OK;
}

__global__ __launch_bounds__(M_THREADS) void k_merge_batch(const uint2 *__restrict__ part, int S, const int32_t *__restrict__ nqv,
                                                          int cap, int32_t *__restrict__ best_idx,
                                                          int32_t *__restrict__ best_d, int32_t *__restrict__ second_d)
{
    const int b = blockIdx.y, i = blockIdx.x * M_THREADS + threadIdx.x;
    if (i >= cap) return;
    const long long o = (long long)b * cap + i;
    int bi = -1, bd = 256, sd = 256;
    if (i < nqv[b]) merge_partials(part, S, (long long)gridDim.y * cap, o, bi, bd, sd);
    best_idx[o] = bi; best_d[o] = bd; second_d[o] = sd;
}

extern "C" int orbm_best2_batch_device(orbm_matcher *m, const uint8_t *d_q, const int32_t *d_nq,
                                       const uint8_t *d_t, const int32_t *d_nt, int cap, int nbatch,
                                       int32_t *d_best_idx, int32_t *d_best_d, int32_t *d_second_d, void *hip_stream)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (!d_q || !d_nq || !d_t || !d_nt || !d_best_idx || !d_best_d || !d_second_d) return mfail(ORBX_E_INVALID, "NULL device pointer");
    if (cap < 1 || nbatch < 1 || cap > 0x3FFFFF) return mfail(ORBX_E_INVALID, "cap=%d nbatch=%d", cap, nbatch);
    MHIPCHK(hipSetDevice(m->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : m->stream;
    int S = 1;
    int rc = launch_dense_batch(m, d_q, d_nq, d_t, d_nt, cap, nbatch, s, &S);
    if (rc != ORBX_OK) return rc;
    hipLaunchKernelGGL(k_merge_batch, dim3((cap + M_THREADS - 1) / M_THREADS, nbatch), dim3(M_THREADS), 0, s,
                       m->d_part, S, d_nq, cap, d_best_idx, d_best_d, d_second_d);
    MHIPCHK(hipGetLastError());
    return ORBX_OK;
}

// ---- N2: SearchByBoW's selection on the GPU (src/ORBmatcher.cc:199-232).  One wave per pair of equal vocabulary nodes: the
// node's key-frame features are visited in order (the loop is sequential in the reference because a frame feature taken by an
// earlier key-frame feature is skipped, :209), the lanes hold the node's frame features (position = chunk * 64 + lane, a
// "taken" bit per chunk in a register), key = distance << 16 | position so that min() is "smallest distance, first in list".
// A frame feature lives in exactly one node, so nodes do not interact. ----
#define BOW_NONE ((256u << 16) | 0xFFFFu)
#define BOW_MAX_NODE_FEATURES 4096      // 64 chunks of 64 lanes
#define BOW_REG_CHUNKS 4                // frame features 0..255 of a node stay in registers

__global__ __launch_bounds__(M_THREADS) void k_bow_select(
    const uint8_t *__restrict__ q, const uint8_t *__restrict__ t, const int32_t *__restrict__ kf_idx,
    const int32_t *__restrict__ f_idx, const int4 *__restrict__ pairs, int npairs, const uint8_t *__restrict__ valid,
    float nnratio, int th, int32_t *__restrict__ match_f)
{
    // the serial loop over a node's key-frame features must not wait for global memory: their descriptors are staged in LDS
    // 64 at a time (lane j fetches feature j), and the first 256 frame features of the node stay in registers
    __shared__ uint4 s_q[M_THREADS / 64][64][2];
    __shared__ int s_ikf[M_THREADS / 64][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * (M_THREADS / 64) + wv;
    if (w >= npairs) return;                 // whole waves leave; nothing below is a block-wide barrier
    const int4 p = pairs[w];                 // key-frame range [x, y) of kf_idx, frame range [z, w) of f_idx
    const int nF = p.w - p.z;
    const int nch = (nF + 63) >> 6;
    uint4 ta[BOW_REG_CHUNKS], tb[BOW_REG_CHUNKS];
#pragma unroll
    for (int ch = 0; ch < BOW_REG_CHUNKS; ch++) {
        const int pos = ch * 64 + lane;
        const int fi = pos < nF ? f_idx[p.z + pos] : 0;
        const uint4 *T = reinterpret_cast<const uint4 *>(t) + 2 * (long long)fi;
        ta[ch] = T[0]; tb[ch] = T[1];
    }
    for (int pos = lane; pos < nF; pos += 64) match_f[f_idx[p.z + pos]] = -1;     // this wave owns these entries
    unsigned long long taken = 0;
    for (int c0 = p.x; c0 < p.y; c0 += 64) {
        const int nb = min(64, p.y - c0);
        {
            int ikf = -1;
            if (lane < nb) {
                ikf = kf_idx[c0 + lane];
                if (valid && !valid[ikf]) ikf = -1;      // :193-197
            }
            const uint4 *Q = reinterpret_cast<const uint4 *>(q) + 2 * (long long)max(ikf, 0);
            s_q[wv][lane][0] = Q[0]; s_q[wv][lane][1] = Q[1];
            s_ikf[wv][lane] = ikf;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS writes have landed
        for (int j = 0; j < nb; j++) {
            const int ikf = s_ikf[wv][j];
            if (ikf < 0) continue;               // wave-uniform
            const uint4 q0 = s_q[wv][j][0], q1 = s_q[wv][j][1];
            uint32_t bk = BOW_NONE, sk = BOW_NONE;
#pragma unroll
            for (int ch = 0; ch < BOW_REG_CHUNKS; ch++) {
                const int pos = ch * 64 + lane;
                const uint32_t key = ((uint32_t)hamming256(q0, q1, ta[ch], tb[ch]) << 16) | (uint32_t)pos;
                const uint32_t k2 = (pos < nF && !((taken >> ch) & 1ull)) ? key : BOW_NONE;
                sk = med3u(bk, sk, k2); bk = min(bk, k2);
            }
            for (int ch = BOW_REG_CHUNKS; ch < nch; ch++) {
                const int pos = ch * 64 + lane;
                if (pos < nF && !((taken >> ch) & 1ull)) {
                    const uint4 *Tj = reinterpret_cast<const uint4 *>(t) + 2 * (long long)f_idx[p.z + pos];
                    const uint32_t key = ((uint32_t)hamming256(q0, q1, Tj[0], Tj[1]) << 16) | (uint32_t)pos;
                    sk = med3u(bk, sk, key); bk = min(bk, key);
                }
            }
            const uint32_t B = wave_min_u32(bk);
            const uint32_t S2 = wave_min_u32((bk == B) ? sk : bk);   // the winner's position is unique: every other lane offers its best
            const int best1 = (int)(B >> 16), best2 = (int)(S2 >> 16);
            if (best1 <= th && (float)best1 < __fmul_rn(nnratio, (float)best2)) {     // :228-232 (th = TH_LOW), :598-600 (th = TH_LOW - 1)
                const int pos = (int)(B & 0xFFFFu);
                if (lane == (pos & 63)) {
                    taken |= 1ull << (pos >> 6);
                    match_f[f_idx[p.z + pos]] = ikf;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();         // the next batch overwrites the staging
    }
}

// ---- host helpers (ComputeThreeMaxima :1601-1642, histogram fill/cull :236-246,:266-284) ----
// rotation histogram + ComputeThreeMaxima cull of SearchByBoW (:236-246, :266-284) on a finished match table
static int bow_rotation_cull(const orbx_keypoint *kps_kf, const orbx_keypoint *kps_f, int n_f, int check_orientation,
                             int32_t *match_f, int *nmatches)
{
    int32_t hist[ORBM_HISTO_LENGTH] = {0};
    std::vector<int> bin_of((size_t)n_f, -1);
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int nm = 0;
    for (int i = 0; i < n_f; i++) {
        if (match_f[i] < 0) continue;
        nm++;
        if (!check_orientation) continue;
        float rot = kps_kf[match_f[i]].angle - kps_f[i].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == ORBM_HISTO_LENGTH) bin = 0;
        if (bin < 0 || bin >= ORBM_HISTO_LENGTH) return mfail(ORBX_E_INVALID, "keypoint angle outside [0, 360)");   // the reference asserts
        bin_of[i] = bin;
        hist[bin]++;
    }
    if (check_orientation) {
        int32_t ind[3];
        orbm_three_maxima(hist, ORBM_HISTO_LENGTH, ind);
        for (int i = 0; i < n_f; i++)
            if (bin_of[i] >= 0 && bin_of[i] != ind[0] && bin_of[i] != ind[1] && bin_of[i] != ind[2]) { match_f[i] = -1; nm--; }
    }
    *nmatches = nm;
    return ORBX_OK;
}

// Returns 1 when the GPU selection does not apply (a node with more than BOW_MAX_NODE_FEATURES frame features, staging larger
// than the handle's buffers, ORBM_BOW_HOST_SELECT=1 in the environment): the caller then takes the host-selection path.
static int search_by_bow_device(orbm_matcher *m,
                                const uint8_t *desc_kf, const orbx_keypoint *kps_kf, int n_kf, const uint8_t *valid_kf,
                                const int32_t *fv_kf_node, const int32_t *fv_kf_off, const int32_t *fv_kf_idx, int fv_kf_n,
                                const uint8_t *desc_f, const orbx_keypoint *kps_f, int n_f,
                                const int32_t *fv_f_node, const int32_t *fv_f_off, const int32_t *fv_f_idx, int fv_f_n,
                                float nnratio, int th, int check_orientation, int32_t *match_f, int *nmatches)
{
    static const bool force_host = [] { const char *e = getenv("ORBM_BOW_HOST_SELECT"); return e && e[0] == '1'; }();
    if (force_host) return 1;
    const int nki = fv_kf_off[fv_kf_n], nfi = fv_f_off[fv_f_n];
    if (nki < 0 || nfi < 0 || n_kf > m->max_q) return 1;
    std::vector<int4> pairs;
    for (int a = 0, b = 0; a < fv_kf_n && b < fv_f_n;) {       // merge-join of the ascending node lists (:178-262)
        if (fv_kf_node[a] == fv_f_node[b]) {
            if (fv_f_off[b + 1] - fv_f_off[b] > BOW_MAX_NODE_FEATURES) return 1;
            if (fv_kf_off[a + 1] > fv_kf_off[a] && fv_f_off[b + 1] > fv_f_off[b])
                pairs.push_back(make_int4(fv_kf_off[a], fv_kf_off[a + 1], fv_f_off[b], fv_f_off[b + 1]));
            a++; b++;
        } else if (fv_kf_node[a] < fv_f_node[b]) a++;
        else b++;
    }
    const int np = (int)pairs.size();
    if (np == 0) return ORBX_OK;
    for (int c = 0; c < nki; c++)
        if (fv_kf_idx[c] < 0 || fv_kf_idx[c] >= n_kf) return mfail(ORBX_E_INVALID, "key-frame feature index %d outside [0,%d)", fv_kf_idx[c], n_kf);
    for (int c = 0; c < nfi; c++)
        if (fv_f_idx[c] < 0 || fv_f_idx[c] >= n_f) return mfail(ORBX_E_INVALID, "frame feature index %d outside [0,%d)", fv_f_idx[c], n_f);
    // device ints: [kf_idx | f_idx | pairs (16-byte aligned) | valid bytes] in d_idx, match table in d_out
    const size_t i_kf = 0, i_f = i_kf + (size_t)nki, i_pairs = (i_f + (size_t)nfi + 3) & ~(size_t)3, i_valid = i_pairs + 4 * (size_t)np;
    const size_t n_ints = i_valid + (valid_kf ? ((size_t)n_kf + 3) / 4 : 0);
    const size_t out_ints = std::max<size_t>((size_t)3 * m->max_q, (size_t)m->max_pairs);
    if (n_ints > (size_t)std::max(m->max_pairs, 1) || (size_t)n_f > out_ints) return 1;
    MHIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    {   // usual path: everything staged in the pinned arena, one copy up, one kernel, one copy back
        int rc_ = orbm_arena_begin(m);
        if (rc_ != ORBX_OK) return rc_;
        const size_t mark = m->arena_used;
        const uint8_t *dq = (const uint8_t *)orbm_stage_in(m, desc_kf, (size_t)n_kf * 32), *dt = (const uint8_t *)orbm_stage_in(m, desc_f, (size_t)n_f * 32);
        const int32_t *dki = (const int32_t *)orbm_stage_in(m, fv_kf_idx, (size_t)nki * 4), *dfi = (const int32_t *)orbm_stage_in(m, fv_f_idx, (size_t)nfi * 4);
        const int4 *dp = (const int4 *)orbm_stage_in(m, pairs.data(), (size_t)np * 16);
        const uint8_t *dv = valid_kf ? (const uint8_t *)orbm_stage_in(m, valid_kf, (size_t)n_kf) : nullptr;
        if (dq && dt && dki && dfi && dp && (dv || !valid_kf)) {
            rc_ = orbm_flush_in(m, mark, s);
            if (rc_ != ORBX_OK) return rc_;
            hipLaunchKernelGGL(k_bow_select, dim3((np + M_THREADS / 64 - 1) / (M_THREADS / 64)), dim3(M_THREADS), 0, s,
                               dq, dt, dki, dfi, dp, np, dv, nnratio, th, m->d_out);
            MHIPCHK(hipGetLastError());
            const int32_t *out = (const int32_t *)orbm_d2h_tmp(m, m->d_out, (size_t)n_f * 4, s);
            if (out) {
                rc_ = orbm_sync(m, s);
                if (rc_ != ORBX_OK) return rc_;
                for (int k = 0; k < np; k++)         // the kernel wrote the entries of the paired nodes only
                    for (int c = pairs[k].z; c < pairs[k].w; c++) match_f[fv_f_idx[c]] = out[fv_f_idx[c]];
                return bow_rotation_cull(kps_kf, kps_f, n_f, check_orientation, match_f, nmatches);
            }
            MHIPCHK(hipStreamSynchronize(s));      // no room for the result this call: take the slower path below
        }
    }
    const size_t o_q = 0, o_t = o_q + (size_t)n_kf * 32, o_i = o_t + (size_t)n_f * 32, o_m = o_i + n_ints * 4, need = o_m + (size_t)n_f * 4;
    if (need > m->h_pin_bytes) {
        MHIPCHK(hipStreamSynchronize(s));
        (void)hipHostFree(m->h_pin); m->h_pin = nullptr; m->h_pin_bytes = 0;
        MHIPCHK(hipHostMalloc((void **)&m->h_pin, need + need / 2, hipHostMallocDefault));
        m->h_pin_bytes = need + need / 2;
    }
    memcpy(m->h_pin + o_q, desc_kf, (size_t)n_kf * 32);
    memcpy(m->h_pin + o_t, desc_f, (size_t)n_f * 32);
    int32_t *hi = reinterpret_cast<int32_t *>(m->h_pin + o_i);
    memcpy(hi + i_kf, fv_kf_idx, (size_t)nki * 4);
    memcpy(hi + i_f, fv_f_idx, (size_t)nfi * 4);
    memcpy(hi + i_pairs, pairs.data(), (size_t)np * 16);
    if (valid_kf) memcpy(hi + i_valid, valid_kf, (size_t)n_kf);
    MHIPCHK(hipMemcpyAsync(m->d_q, m->h_pin + o_q, (size_t)n_kf * 32, hipMemcpyHostToDevice, s));
    MHIPCHK(hipMemcpyAsync(m->d_t, m->h_pin + o_t, (size_t)n_f * 32, hipMemcpyHostToDevice, s));
    MHIPCHK(hipMemcpyAsync(m->d_idx, hi, n_ints * 4, hipMemcpyHostToDevice, s));
    MHIPCHK(hipMemsetAsync(m->d_out, 0xFF, (size_t)n_f * 4, s));        // match table = -1
    hipLaunchKernelGGL(k_bow_select, dim3((np + M_THREADS / 64 - 1) / (M_THREADS / 64)), dim3(M_THREADS), 0, s,
                       m->d_q, m->d_t, m->d_idx + i_kf, m->d_idx + i_f, reinterpret_cast<const int4 *>(m->d_idx + i_pairs), np,
                       valid_kf ? reinterpret_cast<const uint8_t *>(m->d_idx + i_valid) : (const uint8_t *)nullptr, nnratio, th, m->d_out);
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipMemcpyAsync(m->h_pin + o_m, m->d_out, (size_t)n_f * 4, hipMemcpyDeviceToHost, s));
    MHIPCHK(hipStreamSynchronize(s));
    memcpy(match_f, m->h_pin + o_m, (size_t)n_f * 4);
    return bow_rotation_cull(kps_kf, kps_f, n_f, check_orientation, match_f, nmatches);
}

// ---- N2: SearchByBoW (src/ORBmatcher.cc:159-288).  Usual path: search_by_bow_device above (selection on the GPU).  The rest
// of this function is the general fallback: every node-mate distance on the GPU, the order-dependent selection on the host ----
static int search_by_bow_impl(orbm_matcher *m,
                              const uint8_t *desc_kf, const orbx_keypoint *kps_kf, int n_kf, const uint8_t *valid_kf,
                              const int32_t *fv_kf_node, const int32_t *fv_kf_off, const int32_t *fv_kf_idx, int fv_kf_n,
                              const uint8_t *desc_f, const orbx_keypoint *kps_f, int n_f,
                              const int32_t *fv_f_node, const int32_t *fv_f_off, const int32_t *fv_f_idx, int fv_f_n,
                              float nnratio, int th, int check_orientation, int32_t *match_f, int *nmatches)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n_kf < 0 || n_f < 0 || fv_kf_n < 0 || fv_f_n < 0 || !match_f || !nmatches) return mfail(ORBX_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n_f; i++) match_f[i] = -1;
    if (n_kf == 0 || n_f == 0 || fv_kf_n == 0 || fv_f_n == 0) return ORBX_OK;
    if (!desc_kf || !kps_kf || !desc_f || !kps_f || !fv_kf_node || !fv_kf_off || !fv_kf_idx || !fv_f_node || !fv_f_off || !fv_f_idx)
        return mfail(ORBX_E_INVALID, "NULL buffer");
    { int rc_ = orbm_grow(m, n_kf, n_f, 0); if (rc_ != ORBX_OK) return rc_; }
    {   // selection on the GPU when every matched node fits a wave's registers and the staging fits the handle's buffers
        int rc = search_by_bow_device(m, desc_kf, kps_kf, n_kf, valid_kf, fv_kf_node, fv_kf_off, fv_kf_idx, fv_kf_n, desc_f, kps_f, n_f,
                                      fv_f_node, fv_f_off, fv_f_idx, fv_f_n, nnratio, th, check_orientation, match_f, nmatches);
        if (rc != 1) return rc;    // 1 = not applicable: distances on the GPU, selection on the host (below)
    }
    // merge-join of the two ascending node lists (:178-262); queries = usable key-frame features in visiting order
    struct Q { int kf, f_node; };
    std::vector<Q> qs;
    std::vector<int32_t> off;
    qs.reserve((size_t)n_kf); off.reserve((size_t)n_kf + 1);
    off.push_back(0);
    long long pairs = 0;
    for (int a = 0, b = 0; a < fv_kf_n && b < fv_f_n;) {
        if (fv_kf_node[a] == fv_f_node[b]) {
            const int nb = fv_f_off[b + 1] - fv_f_off[b];
            for (int c = fv_kf_off[a]; c < fv_kf_off[a + 1]; c++) {
                const int ikf = fv_kf_idx[c];
                if (ikf < 0 || ikf >= n_kf) return mfail(ORBX_E_INVALID, "key-frame feature index %d outside [0,%d)", ikf, n_kf);
                if (valid_kf && !valid_kf[ikf]) continue;
                qs.push_back({ikf, b});
                pairs += nb;
                off.push_back((int32_t)pairs);
            }
            a++; b++;
        } else if (fv_kf_node[a] < fv_f_node[b]) a++;   // lower_bound on an ascending list == advance
        else b++;
    }
    const int nq = (int)qs.size();
    if (nq == 0 || pairs == 0) return ORBX_OK;
    { int rc_ = orbm_grow(m, 0, 0, pairs); if (rc_ != ORBX_OK) return rc_; }
    // Usual case (both frames below 65536 features): the key frame's descriptor block goes up as it is and every pair is
    // (key-frame feature << 16 | frame feature).  Otherwise the query descriptors are compacted and the kernel finds a
    // pair's query by binary search over the offsets.
    const bool packed16 = n_kf < 65536 && n_f < 65536;
    const int nq_up = packed16 ? n_kf : nq;
    { int rc_ = orbm_grow(m, nq_up, 0, 0); if (rc_ != ORBX_OK) return rc_; }
    // pinned staging block: [query descriptors | n_f x 32 frame descriptors | nq+1 offsets | pairs indices | pairs distances]
    MHIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const size_t o_q = 0, o_t = o_q + (size_t)nq_up * 32, o_off = o_t + (size_t)n_f * 32, o_idx = o_off + ((size_t)nq + 1) * 4;
    const size_t o_dist = o_idx + (size_t)pairs * 4, need = o_dist + (size_t)pairs * 4;
    if (need > m->h_pin_bytes) {
        MHIPCHK(hipStreamSynchronize(s));
        (void)hipHostFree(m->h_pin); m->h_pin = nullptr; m->h_pin_bytes = 0;
        MHIPCHK(hipHostMalloc((void **)&m->h_pin, need + need / 2, hipHostMallocDefault));
        m->h_pin_bytes = need + need / 2;
    }
    if (packed16) memcpy(m->h_pin + o_q, desc_kf, (size_t)n_kf * 32);
    else for (int i = 0; i < nq; i++) memcpy(m->h_pin + o_q + (size_t)i * 32, desc_kf + (size_t)qs[i].kf * 32, 32);
    memcpy(m->h_pin + o_t, desc_f, (size_t)n_f * 32);
    int32_t *idx = reinterpret_cast<int32_t *>(m->h_pin + o_idx);
    const int32_t *dist = reinterpret_cast<const int32_t *>(m->h_pin + o_dist);
    for (int i = 0; i < nq; i++) {
        const int b = qs[i].f_node;
        const uint32_t hi = packed16 ? (uint32_t)qs[i].kf << 16 : 0u;
        int32_t *dst = idx + off[i];
        for (int c = fv_f_off[b]; c < fv_f_off[b + 1]; c++) {
            const int fi = fv_f_idx[c];
            if (fi < 0 || fi >= n_f) return mfail(ORBX_E_INVALID, "frame feature index %d outside [0,%d)", fi, n_f);
            *dst++ = (int32_t)(hi | (uint32_t)fi);
        }
    }
    MHIPCHK(hipMemcpyAsync(m->d_q, m->h_pin + o_q, (size_t)nq_up * 32, hipMemcpyHostToDevice, s));
    MHIPCHK(hipMemcpyAsync(m->d_t, m->h_pin + o_t, (size_t)n_f * 32, hipMemcpyHostToDevice, s));
    MHIPCHK(hipMemcpyAsync(m->d_idx, m->h_pin + o_idx, (size_t)pairs * 4, hipMemcpyHostToDevice, s));
    if (packed16) {
        hipLaunchKernelGGL(k_dist_pairs16, dim3((unsigned)((pairs + M_THREADS - 1) / M_THREADS)), dim3(M_THREADS), 0, s,
                           m->d_q, m->d_t, reinterpret_cast<const uint32_t *>(m->d_idx), (int)pairs, m->d_out);
    } else {
        memcpy(m->h_pin + o_off, off.data(), ((size_t)nq + 1) * 4);
        MHIPCHK(hipMemcpyAsync(m->d_off, m->h_pin + o_off, ((size_t)nq + 1) * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_dist_csr, dim3((unsigned)((pairs + M_THREADS - 1) / M_THREADS)), dim3(M_THREADS), 0, s,
                           m->d_q, nq, m->d_t, m->d_off, m->d_idx, (int)pairs, m->d_out);
    }
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipMemcpyAsync(m->h_pin + o_dist, m->d_out, (size_t)pairs * 4, hipMemcpyDeviceToHost, s));
    MHIPCHK(hipStreamSynchronize(s));
    // sequential selection (:199-246)
    int32_t hist[ORBM_HISTO_LENGTH] = {0};
    std::vector<int> bin_of((size_t)n_f, -1);
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    int nm = 0;
    for (int i = 0; i < nq; i++) {
        int best1 = 256, best2 = 256, bestF = -1;
        for (int c = off[i]; c < off[i + 1]; c++) {
            const int fi = packed16 ? (int)((uint32_t)idx[c] & 0xFFFFu) : idx[c];
            if (match_f[fi] >= 0) continue;                 // :209
            const int d = dist[c];
            if (d < best1) { best2 = best1; best1 = d; bestF = fi; }
            else if (d < best2) best2 = d;
        }
        if (best1 <= th && (float)best1 < nnratio * (float)best2) {
            match_f[bestF] = qs[i].kf;
            if (check_orientation) {
                float rot = kps_kf[qs[i].kf].angle - kps_f[bestF].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == ORBM_HISTO_LENGTH) bin = 0;
                if (bin < 0 || bin >= ORBM_HISTO_LENGTH) return mfail(ORBX_E_INVALID, "keypoint angle outside [0, 360)");   // the reference asserts
                bin_of[bestF] = bin;
                hist[bin]++;
            }
            nm++;
        }
    }
    if (check_orientation) {
        int32_t ind[3];
        orbm_three_maxima(hist, ORBM_HISTO_LENGTH, ind);
        for (int i = 0; i < n_f; i++)
            if (bin_of[i] >= 0 && bin_of[i] != ind[0] && bin_of[i] != ind[1] && bin_of[i] != ind[2]) { match_f[i] = -1; nm--; }
    }
    *nmatches = nm;
    return ORBX_OK;
}

extern "C" int orbm_search_by_bow(orbm_matcher *m,
                                  const uint8_t *desc_kf, const orbx_keypoint *kps_kf, int n_kf, const uint8_t *valid_kf,
                                  const int32_t *fv_kf_node, const int32_t *fv_kf_off, const int32_t *fv_kf_idx, int fv_kf_n,
                                  const uint8_t *desc_f, const orbx_keypoint *kps_f, int n_f,
                                  const int32_t *fv_f_node, const int32_t *fv_f_off, const int32_t *fv_f_idx, int fv_f_n,
                                  float nnratio, int check_orientation, int32_t *match_f, int *nmatches)
{
    return search_by_bow_impl(m, desc_kf, kps_kf, n_kf, valid_kf, fv_kf_node, fv_kf_off, fv_kf_idx, fv_kf_n, desc_f, kps_f, n_f,
                              fv_f_node, fv_f_off, fv_f_idx, fv_f_n, nnratio, ORBM_TH_LOW, check_orientation, match_f, nmatches);
}

// ---- ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (src/ORBmatcher.cc:522-655) ----
// The same node-by-node scan as the key-frame / frame variant above with three differences: features of BOTH key frames need a
// usable MapPoint (:558-562, :576-580), the acceptance is the strict `bestDist1 < TH_LOW` (:598), and the table is indexed by the
// OUTER key frame's features (vpMatches12[idx1]).  A feature of KF2 that can never be taken (no MapPoint, bad MapPoint) is dropped
// from KF2's node lists before the scan -- skipping it inside the scan is the same thing --, the scan runs with th = TH_LOW - 1,
// and the table it returns (inner feature -> outer feature, one-to-one by vbMatched2) is inverted.
extern "C" int orbm_search_by_bow_kf(orbm_matcher *m,
                                     const uint8_t *desc1, const orbx_keypoint *kps1, int n1, const uint8_t *valid1,
                                     const int32_t *fv1_node, const int32_t *fv1_off, const int32_t *fv1_idx, int fv1_n,
                                     const uint8_t *desc2, const orbx_keypoint *kps2, int n2, const uint8_t *valid2,
                                     const int32_t *fv2_node, const int32_t *fv2_off, const int32_t *fv2_idx, int fv2_n,
                                     float nnratio, int check_orientation, int32_t *matches12, int *nmatches)
{
    if (!m) return mfail(ORBX_E_INVALID, "NULL handle");
    if (n1 < 0 || n2 < 0 || fv1_n < 0 || fv2_n < 0 || !matches12 || !nmatches) return mfail(ORBX_E_INVALID, "bad argument");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                             // :534
    if (n1 == 0 || n2 == 0 || fv1_n == 0 || fv2_n == 0) return ORBX_OK;
    if (!fv2_node || !fv2_off || !fv2_idx || !valid1 || !valid2) return mfail(ORBX_E_INVALID, "NULL buffer");
    std::vector<int32_t> off2((size_t)fv2_n + 1, 0), idx2;
    idx2.reserve((size_t)std::max(fv2_off[fv2_n], 0));
    for (int b = 0; b < fv2_n; b++) {
        for (int c = fv2_off[b]; c < fv2_off[b + 1]; c++) {
            const int i2 = fv2_idx[c];
            if (i2 < 0 || i2 >= n2) return mfail(ORBX_E_INVALID, "feature index %d outside [0,%d)", i2, n2);
            if (valid2[i2]) idx2.push_back(i2);
        }
        off2[(size_t)b + 1] = (int32_t)idx2.size();
    }
    if (idx2.empty()) return ORBX_OK;
    std::vector<int32_t> match2((size_t)n2, -1);
    int rc = search_by_bow_impl(m, desc1, kps1, n1, valid1, fv1_node, fv1_off, fv1_idx, fv1_n, desc2, kps2, n2,
                                fv2_node, off2.data(), idx2.data(), fv2_n, nnratio, ORBM_TH_LOW - 1, check_orientation, match2.data(), nmatches);
    if (rc != ORBX_OK) return rc;
    for (int i2 = 0; i2 < n2; i2++)
        if (match2[i2] >= 0) matches12[match2[i2]] = i2;
    return ORBX_OK;
}

extern "C" int orbm_three_maxima(const int32_t *histo, int L, int32_t ind[3])
{
    if (!histo || !ind || L < 0) return mfail(ORBX_E_INVALID, "bad argument");
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    ind[0] = ind1; ind[1] = ind2; ind[2] = ind3;
    return ORBX_OK;
}

extern "C" int orbm_rot_filter(const float *angle_q, const float *angle_t, int32_t *match12, int nq)
{
    if (nq < 0 || (nq > 0 && (!angle_q || !angle_t || !match12))) return mfail(ORBX_E_INVALID, "bad argument");
    int32_t hist[ORBM_HISTO_LENGTH] = {0};
    std::vector<int> bins(nq, -1);
    int nmatches = 0;
    const float factor = 1.0f / ORBM_HISTO_LENGTH;
    for (int i = 0; i < nq; i++) {
        if (match12[i] < 0) continue;
        float rot = angle_q[i] - angle_t[match12[i]];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == ORBM_HISTO_LENGTH) bin = 0;
        bins[i] = bin;
        hist[bin]++;
        nmatches++;
    }
    int32_t ind[3];
    orbm_three_maxima(hist, ORBM_HISTO_LENGTH, ind);
    for (int i = 0; i < nq; i++)
        if (bins[i] >= 0 && bins[i] != ind[0] && bins[i] != ind[1] && bins[i] != ind[2]) { match12[i] = -1; nmatches--; }
    return nmatches;
}
