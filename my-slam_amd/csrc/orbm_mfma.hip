// orbm_mfma.hip -- dense best / second-best DescriptorDistance (src/ORBmatcher.cc:1647-1663 inside the loops :201-226 and
// siblings) on the gfx950 matrix cores.
//
// For 256-bit descriptors a, b:  hamming(a, b) = (256 - a'.b') / 2  with a'_k = +1 / -1 for bit k set / clear, so all
// nq x nt distances of a frame pair are one integer GEMM.  v_mfma_i32_32x32x32_i8 is exact (i8 x i8 -> i32), so nothing about
// the result changes; what changes is where the work runs: 16 popcount-VALU instructions per pair become 1/128 of an MFMA
// plus two VALU instructions of selection.  The selection rules of the reference (strict '<': the lowest train index wins a
// tie, a tie with the best becomes the second best) are folded into the product itself:
//   operands are -+64 (so a'.b' arrives scaled by 4096) and the accumulator of output row i (train descriptor) starts at
//   -(train index):      acc = 4096 * (256 - 2 * dist) - index
//   a larger acc is a smaller distance and, among equal distances, a smaller index -- all accs of a query are distinct, so the
//   two largest accs ARE the reference's best and second best (v_max_i32 + v_med3_i32 per element), and (dist, index) decode
//   from the winner by shift and mask.  12 bits of index: the train range is walked in chunks of 4096 descriptors.
// Layout: train descriptors are the A operand (output rows), queries the B operand (output columns): a lane's 16 accumulators
// are 16 train rows of ONE query column (C/D map of the 32x32 shapes: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)),
// so the running (best, second) of a query lives in two registers of its lane and only lanes l / l + 32 merge at the end.
// A and B use the same lane -> k map (the instruction is symmetric in it), so any consistent choice gives the dot product.
// The operand bytes are made from the descriptor bits inside the kernel (mf_expand16): nothing but the 32-byte descriptors is read.
// The partial (best key, second key) pairs have the format of the popcount kernel (key = distance << 22 | train index), so
// k_merge_best2 / k_merge_keys / k_accept_rot are shared.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "orbm_internal.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define MF_KEY_NONE ((256u << 22) | 0x3FFFFFu)
#define MF_ACC_NONE (-(1 << 30))        // "no candidate yet"; a real acc is >= -(4096 * 256 + 4095)
#define MF_ROW_NONE (-(1 << 29))        // start value of a row beyond the train count: stays below every real acc
#ifndef MF_QB
#define MF_QB 2                         // query blocks (32 queries each) per wave
#endif
#ifndef MF_OCC
#define MF_OCC 3                        // workgroups per CU the register budget is held to
#endif
#define MF_CHUNK 4096                   // train descriptors per index chunk (12 bits)

// 16 descriptor bits -> the 16 operand bytes of one lane and k-step: -64 for a set bit, +64 for a clear one (both operands are
// made this way, and flipping the sign of both leaves every product as it is; this polarity is one instruction shorter).
// Operand order: byte j of lane (r, h) in k-step s is bit 32 s + 16 h + j of descriptor r -- for the A and the B operand alike.
__device__ __forceinline__ uint4 mf_expand16(uint32_t bits)
{
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t nib = (bits >> (4 * k)) & 0xFu;
        const uint32_t spread = __umul24(nib, 0x00204081u) & 0x01010101u;     // bit i of the nibble -> bit 0 of byte i
        o[k] = (spread << 7) | 0x40404040u;                                   // clear -> 0x40 (+64), set -> 0xC0 (-64)
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ uint32_t mf_med3u(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }
__device__ __forceinline__ int mf_med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// (best acc, second acc) of a chunk -> the running (best key, second key)
__device__ __forceinline__ void mf_fold(int &ba, int &sa, uint32_t &bk, uint32_t &sk, int c0)
{
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int x = k == 0 ? ba : sa;
        if (x > -(1 << 22)) {
            const int v = -x;                                    // index - 4096 * (256 - 2 dist)
            const uint32_t idx = (uint32_t)v & (MF_CHUNK - 1);
            const int two_d = 256 + (v >> 12);                   // arithmetic shift: (v - idx) / 4096 = -(256 - 2 dist)
            const uint32_t key = ((uint32_t)(two_d >> 1) << 22) | (uint32_t)(c0 + (int)idx);
            sk = mf_med3u(bk, sk, key);
            bk = min(bk, key);
        }
    }
    ba = MF_ACC_NONE; sa = MF_ACC_NONE;
}

// Workgroup = 4 waves x MF_QB query blocks (256 queries) against one PART (1 / S) of the train descriptors of one frame pair.
// The part is walked in stages of MF_STG tiles (32 descriptors each) through two LDS buffers that hold the tiles already
// expanded to operand bytes.  What is read from memory is only the 32-byte descriptors (one dword per thread and tile for the
// train side, each wave's 64 query descriptors once): the expansion bits -> bytes happens here, through a 2 KiB table in LDS
// (a pre-expanded copy in memory is 8x the bytes; re-reading it per query block and per part made the kernel memory-bound:
// 163 MB per launch, 21 us with the MFMA pipe 23 % busy).  The words of stage s + 1 are requested before the MFMAs of stage s
// start and expanded after them, so after the first stage no memory latency is exposed (a workgroup that loaded its whole part
// up front spent 3.4 of its 8 us waiting, in lockstep with its neighbours: two rounds of that were 24 us).  Three workgroups
// per CU (register budget): MFMAs of one wave run beside the selection VALU of another.  The 1-D grid is ordered pair-major and
// dealt to the XCDs in contiguous eighths (placement only, as in k_fast_cells), so the workgroups of a pair share an L2.
#ifndef MF_STG
#define MF_STG 3
#endif
#ifdef MF_TRACE
__device__ unsigned long long g_mf_trace[4 * 4096];
extern "C" int orbm_debug_mf_trace(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mf_trace), sizeof(g_mf_trace)); }
#define MF_STAMP(i) do { if (threadIdx.x == 0 && lb < 4096) g_mf_trace[4 * lb + (i)] = wall_clock64(); } while (0)
#else
#define MF_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(256, MF_OCC) void k_best2_mfma(const uint8_t *__restrict__ q, const int32_t *__restrict__ nqv, int nq_fixed,
                                                       const uint8_t *__restrict__ t, const int32_t *__restrict__ ntv, int nt_fixed,
                                                       long long qstride, long long tstride, int cap_q, int cap_t, int out_stride,
                                                       uint2 *__restrict__ part, int nbx, int S, int nbatch, int total)
{
    __shared__ uint4 lds[2][MF_STG * 512];          // two stages of train tiles as operand bytes (32 descriptors x 256 B = 8 KiB per tile)
    __shared__ uint2 lut[256];                      // 8 descriptor bits -> 8 operand bytes
    const int lb = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (lb >= total) return;
    MF_STAMP(0);
    const int b = lb / (nbx * S), rem = lb - b * (nbx * S), bz = rem / nbx, bx = rem - bz * nbx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qb0 = (bx * 4 + wave) * MF_QB;
    const uint8_t *Q = q + (long long)b * qstride, *T = t + (long long)b * tstride;
    // The first loads wait for nothing: they go by the CAPACITY of the arrays (rows up to cap_q / cap_t are allocated, whatever
    // the counts say) while the counts themselves are still on their way; what lies beyond the counts is masked afterwards.
    const int ttiles_cap = (cap_t + 31) >> 5, per = (ttiles_cap + S - 1) / S;
    const int t0 = bz * per, t1c = min(ttiles_cap, t0 + per);
    // Train: thread (r = tid & 31, s = tid >> 5) takes word s of descriptor r of every tile of a stage.
    const int tr = tid & 31, ts = tid >> 5;
    uint32_t tw[MF_STG];
    auto load_stage = [&](int tt, int tend) {
#pragma unroll
        for (int j = 0; j < MF_STG; j++) {
            const int row = (tt + j) * 32 + tr;
            tw[j] = (tt + j < tend && row < cap_t) ? *reinterpret_cast<const uint32_t *>(T + (long long)row * 32 + 4 * ts) : 0u;
        }
    };
    load_stage(t0, t1c);
    // Queries: lane (r, h) of a wave takes descriptor r of each of its MF_QB blocks whole (two 16-byte loads)
    const int qr = lane & 31, qh = lane >> 5;
    uint4 qw[MF_QB][2];
#pragma unroll
    for (int u = 0; u < MF_QB; u++) {
        const int row = (qb0 + u) * 32 + qr;
        qw[u][0] = qw[u][1] = make_uint4(0, 0, 0, 0);
        if (row < cap_q) {
            const uint4 *p = reinterpret_cast<const uint4 *>(Q + (long long)row * 32);
            qw[u][0] = p[0]; qw[u][1] = p[1];
        }
    }
    const int nq = nqv ? nqv[b] : nq_fixed, nt = ntv ? ntv[b] : nt_fixed;
    const int qblocks = (nq + 31) >> 5;
    if (bx * 4 * MF_QB >= qblocks) return;                                   // nothing for this workgroup (uniform)
    const bool active = qb0 < qblocks;                                       // wave-uniform; idle waves still help staging
    const int t1 = min((nt + 31) >> 5, t1c);
    {
        const uint4 e = mf_expand16((uint32_t)tid);            // bytes 0..7 of the result belong to the low byte of the argument
        lut[tid] = make_uint2(e.x, e.y);
    }
    __syncthreads();
    auto expand16 = [&](uint32_t bits) -> uint4 {
        const uint2 a = lut[bits & 255u], b2 = lut[(bits >> 8) & 255u];
        return make_uint4(a.x, a.y, b2.x, b2.y);
    };
    auto expand_stage = [&](int tt, int buf) {                 // tw[] -> LDS: both lane halves of the word's k-step
#pragma unroll
        for (int j = 0; j < MF_STG; j++)
            if (tt + j < t1) {
                const bool valid = (tt + j) * 32 + tr < nt;    // rows beyond the count: zero operand bytes (they never win: MF_ROW_NONE)
                lds[buf][j * 512 + ts * 64 + tr] = valid ? expand16(tw[j] & 0xFFFFu) : make_uint4(0, 0, 0, 0);
                lds[buf][j * 512 + ts * 64 + 32 + tr] = valid ? expand16(tw[j] >> 16) : make_uint4(0, 0, 0, 0);
            }
    };
    expand_stage(t0, 0);
    v4i bq[MF_QB][8];
#pragma unroll
    for (int u = 0; u < MF_QB; u++) {
        const uint32_t w8[8] = {qw[u][0].x, qw[u][0].y, qw[u][0].z, qw[u][0].w, qw[u][1].x, qw[u][1].y, qw[u][1].z, qw[u][1].w};
#pragma unroll
        for (int s = 0; s < 8; s++) bq[u][s] = __builtin_bit_cast(v4i, expand16((w8[s] >> (16 * qh)) & 0xFFFFu));
    }
    __syncthreads();
    MF_STAMP(1);

    int ba[MF_QB], sa[MF_QB];
    uint32_t bk[MF_QB], sk[MF_QB];
#pragma unroll
    for (int u = 0; u < MF_QB; u++) { ba[u] = sa[u] = MF_ACC_NONE; bk[u] = sk[u] = MF_KEY_NONE; }
    const int h4 = 4 * (lane >> 5);
    int c0 = (t0 * 32) & ~(MF_CHUNK - 1);
    int buf = 0;
    for (int tt = t0; tt < t1; tt += MF_STG) {
        const bool more = tt + MF_STG < t1;
        if (more) load_stage(tt + MF_STG, t1);                  // in flight during this stage's MFMAs
        if (active) {
#pragma unroll
            for (int j = 0; j < MF_STG; j++) {
                if (tt + j >= t1) break;
                const int base = (tt + j) * 32;
                if ((base & ~(MF_CHUNK - 1)) != c0) {           // next index chunk: bank the finished one
#pragma unroll
                    for (int u = 0; u < MF_QB; u++) mf_fold(ba[u], sa[u], bk[u], sk[u], c0);
                    c0 = base & ~(MF_CHUNK - 1);
                }
                v16i init;
                const int top = c0 - base - h4;                 // -(index of row 0 of this lane's half) relative to the chunk
#pragma unroll
                for (int r = 0; r < 16; r++) init[r] = top - ((r & 3) + 8 * (r >> 2));
                if (base + 32 > nt) {                           // last, partial tile: rows beyond the train count never win
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        if (base + h4 + (r & 3) + 8 * (r >> 2) >= nt) init[r] = MF_ROW_NONE;
                }
                const uint4 *A = &lds[buf][j * 512];
                v16i acc[MF_QB];
                const v4i a0 = __builtin_bit_cast(v4i, A[lane]);
#pragma unroll
                for (int u = 0; u < MF_QB; u++) acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[u][0], init, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 8; s++) {
                    const v4i a = __builtin_bit_cast(v4i, A[s * 64 + lane]);
#pragma unroll
                    for (int u = 0; u < MF_QB; u++) acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[u][s], acc[u], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < MF_QB; u++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int x = acc[u][r];
                        sa[u] = mf_med3i(ba[u], sa[u], x);       // second largest of {ba >= sa, x}
                        ba[u] = max(ba[u], x);
                    }
            }
        }
        if (more) {
            expand_stage(tt + MF_STG, buf ^ 1);                 // nobody reads that buffer: its stage was finished before the last barrier
            __syncthreads();
        }
        buf ^= 1;
    }
    if (!active) return;
    {
#pragma unroll
        for (int u = 0; u < MF_QB; u++) {
            mf_fold(ba[u], sa[u], bk[u], sk[u], c0);
            // lanes l and l + 32 hold the two halves of the rows of one query column
            const uint32_t obk = __shfl_xor(bk[u], 32), osk = __shfl_xor(sk[u], 32);
            uint32_t k1 = bk[u], k2 = sk[u];
            k2 = mf_med3u(k1, k2, obk); k1 = min(k1, obk);
            k2 = mf_med3u(k1, k2, osk); k1 = min(k1, osk);
            const int qi = (qb0 + u) * 32 + (lane & 31);
            if (lane < 32 && qi < nq) part[((long long)bz * nbatch + b) * out_stride + qi] = make_uint2(k1, k2);
        }
    }
    MF_STAMP(2);
}

// -------------------------------------------------------------------------------------------------
// Software-pipelined form (round 3).  k_best2_mfma above issues a tile's 16 MFMAs back to back and then the 64 selection
// instructions on their results: inside one wave the matrix pipe idles during the selection and the vector pipe during the MFMAs,
// and three co-resident waves per SIMD did not interleave well enough to hide it (MFMA pipe ~30 % busy, 21.8 us at 63 x 1007 x 1007
// against 6.7 us of matrix time; splits, occupancy, stage depth: no effect).  Here ONE wave per SIMD carries both
// streams itself: the accumulators are double-buffered and the selection of tile j-1 is scheduled into the gaps between the MFMAs of
// tile j (an MFMA occupies the SIMD's vector issue for 8 of its 32 cycles: ~5 single-issue instructions fit per gap,
// MI355X_MICROARCH.md), placed with __builtin_amdgcn_sched_group_barrier.  What that needs:
//   * the selection as 3 instructions per TWO elements (v_max3_i32, v_med3_i32, v_max_i32) -- 48 per tile and wave instead of 64;
//   * NO per-tile accumulator seed: the seed is the loop-invariant -(row inside the tile), and instead of re-basing 16 new values per
//     tile the running (best, second) of a query are kept RELATIVE to the tile in hand (+32 per tile: 4 scalar-operand adds).
//     acc = 4096 * (256 - 2 dist) + o = 8192 * (128 - dist) + o with o = (first row of the frame's tile) - (row of the element):
//     o is in [-31, 0] for the tile in hand and grows by 32 per tile for older elements; with o + 31 < 8192 the distance field is
//     never touched, so larger acc <=> smaller distance, then smaller row -- the reference's order (the rows of one query are all
//     distinct).  The range is walked in chunks of 4096 rows, banked into (distance << 22 | row) keys at each chunk end.
//   * one workgroup (4 waves x 2 query blocks = 256 queries) per CU and no train split when there are enough pairs: 63 pairs x 4
//     workgroups = 252 for the 256 CUs.
// Same operands, same partial format, same tie rules as above; tests/test_matcher_gpu.py runs both.
// -------------------------------------------------------------------------------------------------
#ifdef SP_TRACE      // per-phase shader-clock totals over the workgroups (wave 0): 0 prologue, 1 tiles, 2 expand (+ the wait for the stage's loads), 3 barrier, 4 epilogue, 7 = workgroups
__device__ unsigned long long g_sp_trace[8];
__device__ unsigned long long g_sp_span[2 * 1024];      // start / end of workgroup lb on the 100 MHz wall clock
extern "C" int orbm_debug_sp_span(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp_span), sizeof(g_sp_span)); }
extern "C" int orbm_debug_sp_trace(unsigned long long *out, int reset)
{
    static unsigned long long z[8];
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sp_trace), z, sizeof(z));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp_trace), sizeof(z));
}
#define SP_T(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); spt_acc[i] += n_ - spt_last; spt_last = n_; } while (0)
#else
#define SP_T(i) do { } while (0)
#endif
#ifndef SP_STG
#define SP_STG 4
#endif                                    // tiles per LDS stage (even: the accumulator parity of a tile is then a compile-time fact)
#define SP_LDS_BYTES (2 * SP_STG * 512 * 16 + 256 * 8)

// second largest of {ba >= sa, x, y} and the largest, 3 instructions for two elements
__device__ __forceinline__ void sp_select2(int &ba, int &sa, int x, int y)
{
    const int m = mf_med3i(ba, x, y);
    ba = max(max(ba, x), y);
    sa = max(sa, m);
}
// (best, second) in the frame whose tile starts at row fbase -> keys
__device__ __forceinline__ void sp_fold(int &ba, int &sa, uint32_t &bk, uint32_t &sk, int fbase)
{
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int x = k == 0 ? ba : sa;
        if (x > -(1 << 22)) {
            const int w = x + 31;
            const int dist = 128 - (w >> 13);                    // arithmetic shift: floor
            const int o = (w & 8191) - 31;
            const uint32_t key = ((uint32_t)dist << 22) | (uint32_t)(fbase - o);
            sk = mf_med3u(bk, sk, key);
            bk = min(bk, key);
        }
    }
    ba = MF_ACC_NONE; sa = MF_ACC_NONE;
}

__global__ __launch_bounds__(256, 2) void k_best2_mfma_sp(const uint8_t *__restrict__ q, const int32_t *__restrict__ nqv, int nq_fixed,
                                                          const uint8_t *__restrict__ t, const int32_t *__restrict__ ntv, int nt_fixed,
                                                          long long qstride, long long tstride, int cap_q, int cap_t, int out_stride,
                                                          uint2 *__restrict__ part, int nbx, int S, int nbatch, int total)
{
    extern __shared__ __attribute__((aligned(16))) uint4 sp_lds[];      // two stages of SP_STG tiles as operand bytes, then the 2 KiB table
    uint2 *lut = reinterpret_cast<uint2 *>(sp_lds + 2 * SP_STG * 512);
    const int lb = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (lb >= total) return;
#ifdef SP_TRACE
    unsigned long long spt_last = __builtin_readcyclecounter(), spt_acc[5] = {0, 0, 0, 0, 0};
    if (threadIdx.x == 0 && lb < 1024) g_sp_span[2 * lb] = wall_clock64();
#endif
    const int b = lb / (nbx * S), rem = lb - b * (nbx * S), bz = rem / nbx, bx = rem - bz * nbx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qb0 = (bx * 4 + wave) * 2;
    const uint8_t *Q = q + (long long)b * qstride, *T = t + (long long)b * tstride;
    const int ttiles_cap = (cap_t + 31) >> 5, per = (ttiles_cap + S - 1) / S;
    const int t0 = bz * per, t1c = min(ttiles_cap, t0 + per);
    const int tr = tid & 31, ts = tid >> 5;
    uint32_t tw[SP_STG];
    auto load_stage = [&](int tt, int tend) {
#pragma unroll
        for (int j = 0; j < SP_STG; j++) {
            const int row = (tt + j) * 32 + tr;
            tw[j] = (tt + j < tend && row < cap_t) ? *reinterpret_cast<const uint32_t *>(T + (long long)row * 32 + 4 * ts) : 0u;
        }
    };
    load_stage(t0, t1c);
    const int qr = lane & 31, qh = lane >> 5;
    uint4 qw[2][2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int row = (qb0 + u) * 32 + qr;
        qw[u][0] = qw[u][1] = make_uint4(0, 0, 0, 0);
        if (row < cap_q) {
            const uint4 *p = reinterpret_cast<const uint4 *>(Q + (long long)row * 32);
            qw[u][0] = p[0]; qw[u][1] = p[1];
        }
    }
    const int nq = nqv ? nqv[b] : nq_fixed, nt = ntv ? ntv[b] : nt_fixed;
    const int qblocks = (nq + 31) >> 5;
    if (bx * 8 >= qblocks) return;                                          // nothing for this workgroup (uniform)
    const bool active = qb0 < qblocks;                                       // wave-uniform; idle waves still help staging
    const int t1 = min((nt + 31) >> 5, t1c);
    {
        const uint4 e = mf_expand16((uint32_t)tid);
        lut[tid] = make_uint2(e.x, e.y);
    }
    __syncthreads();
    auto expand16 = [&](uint32_t bits) -> uint4 {
        const uint2 a = lut[bits & 255u], b2 = lut[(bits >> 8) & 255u];
        return make_uint4(a.x, a.y, b2.x, b2.y);
    };
    auto expand_stage = [&](int tt, int buf) {
#pragma unroll
        for (int j = 0; j < SP_STG; j++)
            if (tt + j < t1) {
                const bool valid = (tt + j) * 32 + tr < nt;
                uint4 *L = sp_lds + (size_t)buf * (SP_STG * 512) + j * 512 + ts * 64 + tr;
                L[0] = valid ? expand16(tw[j] & 0xFFFFu) : make_uint4(0, 0, 0, 0);
                L[32] = valid ? expand16(tw[j] >> 16) : make_uint4(0, 0, 0, 0);
            }
    };
    expand_stage(t0, 0);
    v4i bq[2][8];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint32_t w8[8] = {qw[u][0].x, qw[u][0].y, qw[u][0].z, qw[u][0].w, qw[u][1].x, qw[u][1].y, qw[u][1].z, qw[u][1].w};
#pragma unroll
        for (int s = 0; s < 8; s++) bq[u][s] = __builtin_bit_cast(v4i, expand16((w8[s] >> (16 * qh)) & 0xFFFFu));
    }
    __syncthreads();

    const int h4 = 4 * (lane >> 5);
    v16i init0;
#pragma unroll
    for (int r = 0; r < 16; r++) init0[r] = -(h4 + (r & 3) + 8 * (r >> 2));
    int ba[2] = {MF_ACC_NONE, MF_ACC_NONE}, sa[2] = {MF_ACC_NONE, MF_ACC_NONE};
    uint32_t bk[2] = {MF_KEY_NONE, MF_KEY_NONE}, sk[2] = {MF_KEY_NONE, MF_KEY_NONE};
    int fbase = t0 * 32, chunk0 = t0 * 32;           // first row of the tile (ba, sa) are relative to; first row of the chunk

    // One tile: the eight A operands of the NEXT tile are requested first (AN), then this tile's 16 MFMAs run on operands that
    // arrived during the previous tile (AC), then its selection.  (Read two at a time right before their MFMAs -- what the
    // compiler makes of a plain loop -- every tile pays the LDS latency four times: 2.4 k clocks per tile and wave against 512 of MFMA.)
#define SP_TILE(AC, AN, J)                                                                                               \
    do {                                                                                                                 \
        const int base_ = (tt + (J)) * 32;                                                                               \
        if ((J) + 1 < SP_STG && tt + (J) + 1 < t1) {                                                                     \
            const uint4 *N_ = sp_lds + (size_t)buf * (SP_STG * 512) + ((J) + 1) * 512;                                   \
            _Pragma("unroll") for (int s = 0; s < 8; s++) AN[s] = __builtin_bit_cast(v4i, N_[s * 64 + lane]);            \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (base_ - chunk0 >= MF_CHUNK) {                      /* rare: > 4096 train rows */                             \
            sp_fold(ba[0], sa[0], bk[0], sk[0], fbase); sp_fold(ba[1], sa[1], bk[1], sk[1], fbase);                      \
            chunk0 = base_;                                                                                              \
        }                                                                                                                \
        const int delta_ = base_ - fbase;                                                                                \
        ba[0] += delta_; sa[0] += delta_; ba[1] += delta_; sa[1] += delta_;                                              \
        fbase = base_;                                                                                                   \
        v16i ini_ = init0;                                                                                               \
        if (base_ + 32 > nt) {                                 /* last, partial tile: rows beyond the train count never win */ \
            _Pragma("unroll") for (int r = 0; r < 16; r++)                                                               \
                if (base_ + h4 + (r & 3) + 8 * (r >> 2) >= nt) ini_[r] = MF_ROW_NONE;                                    \
        }                                                                                                                \
        v16i acc0_ = __builtin_amdgcn_mfma_i32_32x32x32_i8(AC[0], bq[0][0], ini_, 0, 0, 0);                              \
        v16i acc1_ = __builtin_amdgcn_mfma_i32_32x32x32_i8(AC[0], bq[1][0], ini_, 0, 0, 0);                              \
        _Pragma("unroll") for (int s = 1; s < 8; s++) {                                                                  \
            acc0_ = __builtin_amdgcn_mfma_i32_32x32x32_i8(AC[s], bq[0][s], acc0_, 0, 0, 0);                              \
            acc1_ = __builtin_amdgcn_mfma_i32_32x32x32_i8(AC[s], bq[1][s], acc1_, 0, 0, 0);                              \
        }                                                                                                                \
        _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                                              \
            sp_select2(ba[0], sa[0], acc0_[r], acc0_[r + 1]);                                                            \
            sp_select2(ba[1], sa[1], acc1_[r], acc1_[r + 1]);                                                            \
        }                                                                                                                \
    } while (0)

    int buf = 0;
    v4i aE[8], aO[8];
    SP_T(0);
    for (int tt = t0; tt < t1; tt += SP_STG) {
        const bool more = tt + SP_STG < t1;
        if (more) load_stage(tt + SP_STG, t1);                  // in flight during this stage's MFMAs
        if (active) {
            const uint4 *F_ = sp_lds + (size_t)buf * (SP_STG * 512);     // the stage's first tile: nothing to overlap its reads with
#pragma unroll
            for (int s = 0; s < 8; s++) aE[s] = __builtin_bit_cast(v4i, F_[s * 64 + lane]);
            SP_TILE(aE, aO, 0);
            if (tt + 1 < t1) SP_TILE(aO, aE, 1);
#if SP_STG == 4
            if (tt + 2 < t1) SP_TILE(aE, aO, 2);
            if (tt + 3 < t1) SP_TILE(aO, aE, 3);
#endif
            static_assert(SP_STG == 4 || SP_STG == 2, "the tile sequence above is written out for two or four tiles per stage");
        }
        SP_T(1);
        if (more) {
            expand_stage(tt + SP_STG, buf ^ 1);
            SP_T(2);
            __syncthreads();
            SP_T(3);
        }
        buf ^= 1;
    }
#undef SP_TILE
    if (!active) return;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        sp_fold(ba[u], sa[u], bk[u], sk[u], fbase);
        const uint32_t obk = __shfl_xor(bk[u], 32), osk = __shfl_xor(sk[u], 32);
        uint32_t k1 = bk[u], k2 = sk[u];
        k2 = mf_med3u(k1, k2, obk); k1 = min(k1, obk);
        k2 = mf_med3u(k1, k2, osk); k1 = min(k1, osk);
        const int qi = (qb0 + u) * 32 + (lane & 31);
        if (lane < 32 && qi < nq) part[((long long)bz * nbatch + b) * out_stride + qi] = make_uint2(k1, k2);
    }
#ifdef SP_TRACE
    SP_T(4);
    if (tid == 0) { for (int i = 0; i < 5; i++) atomicAdd(&g_sp_trace[i], spt_acc[i]); atomicAdd(&g_sp_trace[7], 1ull); if (lb < 1024) g_sp_span[2 * lb + 1] = wall_clock64(); }
#endif
}

static bool orbm_mfma_sp_on()
{
    static const bool on = [] { const char *e = getenv("ORBM_MFMA_SP"); return e && atoi(e) != 0; }();   // A/B switch, read once; off: measured slower (header comment)
    return on;
}

int orbm_mfma_splits(int nq_cap, int nt_cap, int nbatch)
{
    // parts of the train range per frame pair: enough workgroups for three per CU (one round), a part not shorter than two stages
    const int ttiles = std::max((nt_cap + 31) >> 5, 1);
    if (orbm_mfma_sp_on()) {      // the pipelined kernel: one workgroup per CU; a part not shorter than two stages
        const long long wgs = (long long)nbatch * ((((nq_cap + 31) >> 5) + 7) / 8);
        int S = (int)std::max<long long>((512 + wgs / 2) / std::max<long long>(wgs, 1), 1);     // two workgroups per CU
        static const int forced = [] { const char *e = getenv("ORBM_MFMA_SPLITS"); return e ? std::max(atoi(e), 1) : 0; }();
        if (forced) S = forced;
        return std::min(std::min(S, std::max(ttiles / (2 * SP_STG), 1)), 64);
    }
    const long long wgs = (long long)nbatch * ((((nq_cap + 31) >> 5) + 4 * MF_QB - 1) / (4 * MF_QB));
    int S = (int)std::max<long long>((256 * MF_OCC + wgs - 1) / std::max<long long>(wgs, 1), 1);
    static const int forced = [] { const char *e = getenv("ORBM_MFMA_SPLITS"); return e ? std::max(atoi(e), 1) : 0; }();   // tuning switch, read once
    if (forced) S = forced;
    S = std::min(S, std::max(ttiles / (2 * MF_STG), 1));
    return std::min(S, 64);
}

// Partials of nbatch dense pairs -> part[S][nbatch][out_stride]; counts per pair from d_nq / d_nt (device) or the fixed values.
int orbm_launch_dense_mfma(orbm_matcher *m, const uint8_t *d_q, const int32_t *d_nq, int nq_fixed, const uint8_t *d_t, const int32_t *d_nt,
                           int nt_fixed, long long qstride, long long tstride, int cap_q, int cap_t, int nbatch, int out_stride, int S,
                           uint2 *part, hipStream_t s)
{
    (void)m;
    const int qtiles = (cap_q + 31) >> 5;
    if (orbm_mfma_sp_on()) {
        static const bool attr = [] { return hipFuncSetAttribute(reinterpret_cast<const void *>(k_best2_mfma_sp), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS_BYTES) == hipSuccess; }();
        if (!attr) return ORBX_E_HIP;
        const int nbx8 = (qtiles + 7) / 8, total8 = nbx8 * S * nbatch;
        hipLaunchKernelGGL(k_best2_mfma_sp, dim3((unsigned)((total8 + 7) & ~7)), dim3(256), SP_LDS_BYTES, s,
                           d_q, d_nq, nq_fixed, d_t, d_nt, nt_fixed, qstride, tstride, cap_q, cap_t, out_stride, part, nbx8, S, nbatch, total8);
        MHIPCHK(hipGetLastError());
        return ORBX_OK;
    }
    const int nbx = (qtiles + 4 * MF_QB - 1) / (4 * MF_QB), total = nbx * S * nbatch;
    hipLaunchKernelGGL(k_best2_mfma, dim3((unsigned)((total + 7) & ~7)), dim3(256), 0, s,
                       d_q, d_nq, nq_fixed, d_t, d_nt, nt_fixed, qstride, tstride, cap_q, cap_t, out_stride, part, nbx, S, nbatch, total);
    MHIPCHK(hipGetLastError());
    return ORBX_OK;
}
