// orbx_pyramid.hip -- ComputePyramid (src/ORBextractor.cc:1109-1137 of WChen09/My-SLAM) on gfx950.
#include "orbx_internal.h"
#ifndef RESIZE_THREADS
#define RESIZE_THREADS 256   // threads per workgroup of the 4x4-block resize kernels (independent threads)
#endif

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

// -------------------------------------------------------------------------------------------------
// Fast path: one thread = a 4x4 block of destination pixels.  The four columns' source offsets and
// 11-bit coefficients are loaded once (two 16-byte table reads) and reused for four rows; each source
// row segment (<= 8 bytes for scale factors up to 2) arrives as three aligned dwords that are
// funnel-shifted into place, so there are 1.5 memory instructions per destination pixel instead of
// 6, and the horizontal blend is one v_dot2_u32_u16.  CHECK guards the last bytes of a caller-owned
// level-0 buffer, which has no slack behind it.
// -------------------------------------------------------------------------------------------------
template <bool CHECK>
__global__ __launch_bounds__(RESIZE_THREADS) void k_resize_linear_4x4(
    const uint8_t *__restrict__ src, int sw, int sh, int sstride, long long sframe,
    uint8_t *__restrict__ dst, int dw, int dh, int dstride, long long dframe, ResizeTab tab,
    const uint8_t *src_end, int nbx, int nblk, uint32_t rcp_nbx)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    // The 4x4 blocks of a frame are numbered row-major and dealt to threads linearly: a wave is 64 consecutive
    // blocks (a 256-px strip, wrapping at the row end), so no lane is lost to tile quantisation -- with 2-D tiles of
    // 128 x 32 px a 533 x 400 level launched 25 % more waves than it has work for, a 179 x 134 one 70 % more.
    const int t = blockIdx.x * RESIZE_THREADS + threadIdx.x;
    if (t >= nblk) return;
    int by = rcp_nbx ? (int)__umulhi((uint32_t)t, rcp_nbx) : t;   // t / nbx (multiply-high by floor(2^32 / nbx) + 1, one correction; 0: nbx == 1)
    by -= (by * nbx > t) ? 1 : 0;
    const int x4 = (t - by * nbx) * 4, y4 = by * 4;
    const uint8_t *S = src + (long long)blockIdx.z * sframe;
    uint8_t *D = dst + (long long)blockIdx.z * dframe;

    const int4 sxv = *reinterpret_cast<const int4 *>(tab.xofs + x4);          // tables are padded to 4
    const uint4 alv = *reinterpret_cast<const uint4 *>(tab.alpha + x4);        // (a0 | a1 << 16) per column
    int sx[4] = {sxv.x, sxv.y, sxv.z, sxv.w};
    uint32_t al[4] = {alv.x, alv.y, alv.z, alv.w};
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (x4 + i >= dw) { sx[i] = sx[0]; al[i] = al[0]; }
    const int base = sx[0];
    // v_perm_b32 selectors: source bytes (sx[i] - base) and (sx[i] - base) + 1 of the 8-byte window into the two u16
    // halves (0x0c = zero byte); the window offsets are < 7 by the planner's span check
    uint32_t sel[4];
#pragma unroll
    for (int i = 0; i < 4; i++) sel[i] = 0x0c010c00u + (uint32_t)(sx[i] - base) * 0x00010001u;

    // phase 1: issue every load of the 4x4 block (8 row-table words, 24 source dwords) before any use,
    // so one memory round trip covers the whole block instead of one per row.  Addresses are the uniform frame base
    // plus an unsigned 32-bit byte offset (one frame is below 2 GiB), which keeps the address arithmetic 32-bit.
    const uint32_t s_lo = (uint32_t)(uintptr_t)S;
    const uint32_t endoff = CHECK ? (uint32_t)(src_end - S) : 0u;   // only meaningful (and only used) for the last frame
    int b0v[4], b1v[4];
    uint32_t sh8[4][2], wv[4][2][3];
    // the row tables of the block's four rows in two 16-byte loads (y4 is a multiple of 4; the planner pads both tables to a
    // multiple of 4 rows with copies of the last row, so no clamp is needed here)
    const int4 syv = *reinterpret_cast<const int4 *>(tab.yofs + y4);
    const uint4 bwv = *reinterpret_cast<const uint4 *>(tab.beta + y4);
    const int sy4[4] = {syv.x, syv.y, syv.z, syv.w};
    const uint32_t bw4[4] = {bwv.x, bwv.y, bwv.z, bwv.w};
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    typedef u32x3 __attribute__((aligned(4))) U3a4;   // three dwords at a 4-byte aligned address: one global_load_dwordx3
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int sy = sy4[r];
        const uint32_t bw = bw4[r];
        b0v[r] = (int)(short)(bw & 0xFFFFu);
        b1v[r] = (int)(short)(bw >> 16);
        const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint32_t off = (uint32_t)(__mul24(rr ? sy1 : sy0, sstride) + base);   // 24-bit multiply is full rate
            sh8[r][rr] = (s_lo + off) & 3u;
            const uint32_t offa = off - sh8[r][rr];
            if (CHECK && blockIdx.z == gridDim.z - 1) {   // only the last frame can end at the end of the caller's buffer
                const uint32_t *q = reinterpret_cast<const uint32_t *>(S + offa);
                wv[r][rr][0] = q[0];
                wv[r][rr][1] = (offa + 8u <= endoff) ? q[1] : 0u;
                wv[r][rr][2] = (offa + 12u <= endoff) ? q[2] : 0u;
            } else {
                const U3a4 q = *reinterpret_cast<const U3a4 *>(S + offa);
                wv[r][rr][0] = q.x; wv[r][rr][1] = q.y; wv[r][rr][2] = q.z;
            }
        }
    }
    // phase 2: blend
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = y4 + r;
        if (y >= dh) break;
        const int b0 = b0v[r], b1 = b1v[r];
        int hv[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint32_t lo = __builtin_amdgcn_alignbyte(wv[r][rr][1], wv[r][rr][0], sh8[r][rr]);
            const uint32_t hi = __builtin_amdgcn_alignbyte(wv[r][rr][2], wv[r][rr][1], sh8[r][rr]);
            // (S[sx] | S[sx+1] << 16) . (a0 | a1 << 16)
#pragma unroll
            for (int i = 0; i < 4; i++)
                hv[rr][i] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(hi, lo, sel[i])), __builtin_bit_cast(us2, al[i]), 0u, false);
        }
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // beta <= 2048 and (horizontal sum >> 4) <= 32640: exact in the full-rate 24-bit multiply
            const int v = ((__mul24(b0, hv[0][i] >> 4) >> 16) + (__mul24(b1, hv[1][i] >> 4) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 255) << (8 * i);
        }
        uint8_t *Dr = D + (uint32_t)(__mul24(y, dstride) + x4);
        if (x4 + 3 < dw) *reinterpret_cast<uint32_t *>(Dr) = out;
        else {   // the last block of a row: 1..3 pixels (three predicated byte stores; a loop here gets vectorised into a mess)
            Dr[0] = (uint8_t)out;
            if (x4 + 1 < dw) Dr[1] = (uint8_t)(out >> 8);
            if (x4 + 2 < dw) Dr[2] = (uint8_t)(out >> 16);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Shared-row 4x4 block (round 3).  For scale factors in (1, 4/3] the four destination rows y4 .. y4+3 of a block touch at most SIX
// source rows sy0 .. sy0+5 (sy0 = yofs[y4]; yofs[y4 + r] - sy0 is r or r + 1: checked by the planner, RESIZE_FAST6), where the
// kernel above blends EIGHT row segments horizontally (two per destination row).  Here every source row is blended once:
//   HH[j][i] = (S[row j][sx_i] * a0_i + S[row j][sx_i + 1] * a1_i) >> 4        6 x 4 values, <= 32640
// and destination row r takes rows (r, r+1) or (r+1, r+2) of them.  Which pair is a per-lane fact (the blocks of a wave wrap
// around the row end), so instead of selecting registers the SELECTION RIDES IN THE WEIGHTS: with k = (yofs[y4+r] - sy0 != r),
//   (w0, w1, w2) = k ? (0, b0, b1) : (b0, b1, 0),   v = ((w0 HH[r] + w2 HH[r+2] + 2 << 16) >> 16) + (w1 HH[r+1] >> 16)
// is OpenCV's ((b0 * S0 >> 16) + (b1 * S1 >> 16) + 2) exactly (one of w0, w2 is zero; the rounding constant rides in the first
// product).  The two >> 16, the sum and the final >> 2 are done on u16 PAIRS: v_perm_b32 picks the high halves of two products,
// v_pk_add_u16, v_pk_lshrrev_b16, and one more v_perm_b32 packs the four bytes of a row: 21 vector instructions per row of four
// pixels after the 14 per source row, ~230 per block where the kernel above issues ~350.
// SRC = 0: the source is global memory (S = frame base, absolute rows / columns); SRC = 1: an LDS tile (pitch, rows and
// columns relative to the tile origin: k_resize_tiles below).
// -------------------------------------------------------------------------------------------------
typedef unsigned short rs_us2 __attribute__((ext_vector_type(2)));
typedef uint32_t rs_u32x3 __attribute__((ext_vector_type(3)));
typedef rs_u32x3 __attribute__((aligned(4))) rs_U3a4;
typedef const __attribute__((address_space(1))) uint8_t *rs_gptr;   // explicitly global
__device__ __forceinline__ uint32_t rs_mad24(uint32_t a, uint32_t b, uint32_t c)   // a * b + c on the 24-bit multiplier (left alone the compiler emits mul + add3)
{
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ rs_gptr rs_scalar_ptr(const uint8_t *p)     // a wave-uniform pointer pinned to scalar registers
{
    const uint64_t b = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return (rs_gptr)(((uint64_t)hi << 32) | lo);
}
template <int SRC, bool CHECK>
__device__ __forceinline__ void resize_block6(const uint8_t *S, int pitch, int row0, int rmax, int col, const uint32_t sel[4], const uint32_t al[4],
                                              const int4 syv, const uint4 bwv, uint32_t endoff, bool last_frame, uint32_t out[4])
{
    uint32_t HH[6][4];
    {
        uint32_t wv[6][3], sh8[6];
        // row j of the block = source row min(row0 + j, rmax): one add and one min per row on the byte offset
        const uint32_t off0 = (uint32_t)(__mul24(row0, pitch) + col), offmax = (uint32_t)(__mul24(rmax, pitch) + col);
        const rs_gptr Sg = rs_scalar_ptr(S);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const uint32_t off = min(off0 + (uint32_t)(j * pitch), offmax);
            if (SRC == 0) {
                sh8[j] = ((uint32_t)(uintptr_t)S + off) & 3u;
                const uint32_t offa = off - sh8[j];
                if (CHECK && last_frame) {     // only the last frame can end at the end of the caller's buffer
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(S + offa);
                    wv[j][0] = q[0];
                    wv[j][1] = (offa + 8u <= endoff) ? q[1] : 0u;
                    wv[j][2] = (offa + 12u <= endoff) ? q[2] : 0u;
                } else {
                    // (scalar frame base) + (32-bit lane offset): the global_load saddr form, no 64-bit vector arithmetic
                    const rs_U3a4 q = *(const __attribute__((address_space(1))) rs_U3a4 *)(Sg + offa);
                    wv[j][0] = q.x; wv[j][1] = q.y; wv[j][2] = q.z;
                }
            } else {
                sh8[j] = off & 3u;
                const uint32_t *q = reinterpret_cast<const uint32_t *>(S + (off & ~3u));
                wv[j][0] = q[0]; wv[j][1] = q[1]; wv[j][2] = q[2];
            }
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const uint32_t lo = __builtin_amdgcn_alignbyte(wv[j][1], wv[j][0], sh8[j]);
            const uint32_t hi = __builtin_amdgcn_alignbyte(wv[j][2], wv[j][1], sh8[j]);
#pragma unroll
            for (int i = 0; i < 4; i++)
                HH[j][i] = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_us2, __builtin_amdgcn_perm(hi, lo, sel[i])), __builtin_bit_cast(rs_us2, al[i]), 0u, false) >> 4;
        }
    }
    const int sy4[4] = {syv.x, syv.y, syv.z, syv.w};
    const uint32_t bw4[4] = {bwv.x, bwv.y, bwv.z, bwv.w};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t b0 = bw4[r] & 0xFFFFu, b1 = bw4[r] >> 16;     // 0 .. 2048 (a reduction: both weights are non-negative)
        uint32_t p[4], q[4];
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) { p[i] = rs_mad24(b0, HH[0][i], 0x20000u); q[i] = __umul24(b1, HH[1][i]); }
        } else {
            const uint32_t m = (uint32_t)(r - (sy4[r] - sy4[0]));     // 0: rows (r, r+1); 0xFFFFFFFF: rows (r+1, r+2)
            const uint32_t w0 = b0 & ~m, w2 = b1 & m, w1 = b1 ^ ((b0 ^ b1) & m);
#pragma unroll
            for (int i = 0; i < 4; i++) { p[i] = rs_mad24(w2, HH[r + 2][i], rs_mad24(w0, HH[r][i], 0x20000u)); q[i] = __umul24(w1, HH[r + 1][i]); }
        }
        // (p >> 16) + (q >> 16) on u16 pairs, then >> 2, then the four low bytes
        const rs_us2 z01 = __builtin_bit_cast(rs_us2, __builtin_amdgcn_perm(p[1], p[0], 0x07060302u)) + __builtin_bit_cast(rs_us2, __builtin_amdgcn_perm(q[1], q[0], 0x07060302u));
        const rs_us2 z23 = __builtin_bit_cast(rs_us2, __builtin_amdgcn_perm(p[3], p[2], 0x07060302u)) + __builtin_bit_cast(rs_us2, __builtin_amdgcn_perm(q[3], q[2], 0x07060302u));
        const rs_us2 two = {2, 2};
        out[r] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, z23 >> two), __builtin_bit_cast(uint32_t, z01 >> two), 0x06040200u);
    }
}

template <bool CHECK>
__global__ __launch_bounds__(RESIZE_THREADS) void k_resize_linear_4x4s(
    const uint8_t *__restrict__ src, int sw, int sh, int sstride, long long sframe,
    uint8_t *__restrict__ dst, int dw, int dh, int dstride, long long dframe, ResizeTab tab,
    const uint8_t *src_end, int nbx, int nblk, uint32_t rcp_nbx)
{
    const int t = blockIdx.x * RESIZE_THREADS + threadIdx.x;     // blocks numbered row-major, dealt linearly (see k_resize_linear_4x4)
    if (t >= nblk) return;
    int by = rcp_nbx ? (int)__umulhi((uint32_t)t, rcp_nbx) : t;
    by -= (by * nbx > t) ? 1 : 0;
    const int x4 = (t - by * nbx) * 4, y4 = by * 4;
    const uint8_t *S = src + (long long)blockIdx.z * sframe;
    uint8_t *D = dst + (long long)blockIdx.z * dframe;
    const int4 sxv = *reinterpret_cast<const int4 *>(tab.xofs + x4);          // tables are padded to 4
    const uint4 alv = *reinterpret_cast<const uint4 *>(tab.alpha + x4);
    const int4 syv = *reinterpret_cast<const int4 *>(tab.yofs + y4);
    const uint4 bwv = *reinterpret_cast<const uint4 *>(tab.beta + y4);
    int sx[4] = {sxv.x, sxv.y, sxv.z, sxv.w};
    uint32_t al[4] = {alv.x, alv.y, alv.z, alv.w}, sel[4], out[4];
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (x4 + i >= dw) { sx[i] = sx[0]; al[i] = al[0]; }
#pragma unroll
    for (int i = 0; i < 4; i++) sel[i] = 0x0c010c00u + (uint32_t)(sx[i] - sx[0]) * 0x00010001u;
    const uint32_t endoff = CHECK ? (uint32_t)(src_end - S) : 0u;
    resize_block6<0, CHECK>(S, sstride, max(syv.x, 0), sh - 1, sx[0], sel, al, syv, bwv, endoff, CHECK && blockIdx.z == gridDim.z - 1, out);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = y4 + r;
        if (y >= dh) break;
        uint8_t *Dr = D + (uint32_t)(__mul24(y, dstride) + x4);
        if (x4 + 3 < dw) *reinterpret_cast<uint32_t *>(Dr) = out[r];
        else {
            Dr[0] = (uint8_t)out[r];
            if (x4 + 1 < dw) Dr[1] = (uint8_t)(out[r] >> 8);
            if (x4 + 2 < dw) Dr[2] = (uint8_t)(out[r] >> 16);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Levels a+1 .. b in ONE launch.  The upper levels of the pyramid are small (level 7 of a 640 x 480 frame is 179 x 134): as
// launches of their own each costs the ~4.5 us launch floor for 0.3 .. 4 us of work, six of them in a row.  Level l + 1 is a pure
// function of level l's integer output (:1117-1123), so a workgroup that holds a band of rows of level l in LDS can produce the
// band of level l + 1 under it without going back to memory: one workgroup = one band of BH rows of level b of one frame; it walks
// down from level a (read from memory) to level b, keeping the rows of each level it NEEDS (the bilinear footprint of what the
// next level needs) in one of two LDS buffers and writing the rows it OWNS to memory.  The owned row ranges of the bands of a
// level partition that level (they are the images of the bands of level b under the monotone yofs maps), the needed ranges
// overlap by the footprint: the overlap is computed twice (~28 % more pixels for 16-row bands over six levels) and written once.
// Bands are full-width, so rows are read and written whole and only the row ranges need planning (host, at plan time).
// The arithmetic per pixel is k_resize_linear_4x4's: aligned dwords funnel-shifted, v_perm_b32 pairs, v_dot2_u32_u16 for the
// horizontal blend, 24-bit multiplies for the vertical one.  Thread work item = 4 destination pixels of one row.
// -------------------------------------------------------------------------------------------------
// One row of 4 destination pixels: the source dwords of the two source rows are loaded first (issue), blended later (finish),
// so that the loads of several rows are in flight together.  The row tables (yofs, beta) of every level's band were copied to
// LDS when the workgroup started: inside a level nothing but the source pixels is waited for.
struct FuseRow { uint32_t w[2][3], sh[2]; int b0, b1; };
template <bool MEM>
__device__ __forceinline__ void fuse_issue(FuseRow &R, const FuseLevel &S, const uint8_t *Sg, const uint8_t *sbuf, int need0_src,
                                           const int2 yt, int base)
{
    const int sy = yt.x;
    R.b0 = (int)(short)((uint32_t)yt.y & 0xFFFFu); R.b1 = (int)(short)((uint32_t)yt.y >> 16);
    const int sy0 = min(max(sy, 0), S.h - 1), sy1 = min(max(sy + 1, 0), S.h - 1);
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int srow = rr ? sy1 : sy0;
        if (MEM) {
            // pointer arithmetic on the typed pointer (an integer round trip would turn these into flat loads)
            const uint32_t off = (uint32_t)(__mul24(srow, S.stride) + base);
            R.sh[rr] = ((uint32_t)(uintptr_t)Sg + off) & 3u;
            const uint32_t *q = reinterpret_cast<const uint32_t *>(Sg + (off - R.sh[rr]));
            R.w[rr][0] = q[0]; R.w[rr][1] = q[1]; R.w[rr][2] = q[2];
        } else {
            const uint32_t off = (uint32_t)(__mul24(srow - need0_src, S.pitch) + base);
            const uint32_t *q = reinterpret_cast<const uint32_t *>(sbuf + (off & ~3u));
            R.sh[rr] = off & 3u;
            R.w[rr][0] = q[0]; R.w[rr][1] = q[1]; R.w[rr][2] = q[2];
        }
    }
}
__device__ __forceinline__ uint32_t fuse_finish(const FuseRow &R, const uint32_t sel[4], const uint32_t al[4])
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    int hv[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const uint32_t lo = __builtin_amdgcn_alignbyte(R.w[rr][1], R.w[rr][0], R.sh[rr]), hi = __builtin_amdgcn_alignbyte(R.w[rr][2], R.w[rr][1], R.sh[rr]);
#pragma unroll
        for (int i = 0; i < 4; i++)
            hv[rr][i] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(hi, lo, sel[i])), __builtin_bit_cast(us2, al[i]), 0u, false);
    }
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = ((__mul24(R.b0, hv[0][i] >> 4) >> 16) + (__mul24(R.b1, hv[1][i] >> 4) >> 16) + 2) >> 2;
        out |= (uint32_t)(v & 255) << (8 * i);
    }
    return out;
}

// a thread keeps ONE block of 4 columns for a whole level (its column tables live in registers) and walks down the rows
struct FuseCols { int rg, G, x4; int4 sxv; uint4 alv; };
__device__ __forceinline__ void fuse_cols(FuseCols &C, const FuseLevel &D)
{
    const int tid = threadIdx.x;
    int rg = D.rcp_nbx ? (int)__umulhi((uint32_t)tid, D.rcp_nbx) : tid;       // tid / nbx
    rg -= (rg * D.nbx > tid) ? 1 : 0;
    C.rg = rg; C.G = 512 / D.nbx;                                             // row groups that fit the workgroup (nbx <= 512: checked on the host)
    C.x4 = (tid - rg * D.nbx) * 4;
    C.sxv = make_int4(0, 0, 0, 0); C.alv = make_uint4(0, 0, 0, 0);
    if (rg < C.G) {
        C.sxv = *reinterpret_cast<const int4 *>(D.tab.xofs + C.x4);           // tables are padded to 4
        C.alv = *reinterpret_cast<const uint4 *>(D.tab.alpha + C.x4);         // (a0 | a1 << 16) per column
    }
}

template <bool MEM>
__device__ __forceinline__ void fuse_level(const FuseArgs &A, int l, const int4 rs, const int4 rd, int f, uint8_t *lds, const int2 *ytab,
                                           const FuseCols &C)
{
    const FuseLevel &S = A.lv[l - 1], &D = A.lv[l];
    const uint8_t *sbuf = lds + (((l - 1 - A.a) & 1) ? A.buf0_bytes : 0);
    uint8_t *dbuf = lds + (((l - A.a) & 1) ? A.buf0_bytes : 0);
    const uint8_t *Sg = S.base + (long long)f * S.frame;
    uint8_t *Dg = D.base + (long long)f * D.frame;
    const int rg = C.rg, G = C.G, x4 = C.x4;
    if (rg >= G) return;
    int sx[4] = {C.sxv.x, C.sxv.y, C.sxv.z, C.sxv.w};
    uint32_t al[4] = {C.alv.x, C.alv.y, C.alv.z, C.alv.w}, sel[4];
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (x4 + i >= D.w) { sx[i] = sx[0]; al[i] = al[0]; }
    const int base = sx[0];
#pragma unroll
    for (int i = 0; i < 4; i++) sel[i] = 0x0c010c00u + (uint32_t)(sx[i] - base) * 0x00010001u;
    const int rows = rd.w - rd.z;
    auto emit = [&](int ry, uint32_t out) {
        const int y = rd.z + ry;
        if (l < A.b) *reinterpret_cast<uint32_t *>(dbuf + (uint32_t)(__mul24(ry, D.pitch) + x4)) = out;
        if (y >= rd.x && y < rd.y) {
            uint8_t *Dr = Dg + (uint32_t)(__mul24(y, D.stride) + x4);
            if (x4 + 3 < D.w) *reinterpret_cast<uint32_t *>(Dr) = out;
            else
                for (int i = 0; x4 + i < D.w; i++) Dr[i] = (uint8_t)(out >> (8 * i));
        }
    };
    int ry = rg;
    for (; ry + 3 * G < rows; ry += 4 * G) {                                  // four rows in flight
        FuseRow R0, R1, R2, R3;
        fuse_issue<MEM>(R0, S, Sg, sbuf, rs.z, ytab[ry], base);
        fuse_issue<MEM>(R1, S, Sg, sbuf, rs.z, ytab[ry + G], base);
        fuse_issue<MEM>(R2, S, Sg, sbuf, rs.z, ytab[ry + 2 * G], base);
        fuse_issue<MEM>(R3, S, Sg, sbuf, rs.z, ytab[ry + 3 * G], base);
        emit(ry, fuse_finish(R0, sel, al));
        emit(ry + G, fuse_finish(R1, sel, al));
        emit(ry + 2 * G, fuse_finish(R2, sel, al));
        emit(ry + 3 * G, fuse_finish(R3, sel, al));
    }
    for (; ry < rows; ry += G) {
        FuseRow R0;
        fuse_issue<MEM>(R0, S, Sg, sbuf, rs.z, ytab[ry], base);
        emit(ry, fuse_finish(R0, sel, al));
    }
}

// -DFUSE_TRACE: per-level wall-clock (100 MHz) totals over all workgroups (tools/dbg/fuse_trace.py); slot 15 counts workgroups
#ifdef FUSE_TRACE
__device__ unsigned long long g_fuse_trace[16];
extern "C" int orbx_debug_fuse_trace(unsigned long long *out, int reset)
{
    static unsigned long long z[16];
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fuse_trace), z, sizeof(z));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fuse_trace), sizeof(z));
}
#define FUSE_T(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&g_fuse_trace[i], n_ - ft_); ft_ = n_; } } while (0)
#else
#define FUSE_T(i) do { } while (0)
#endif

__global__ __launch_bounds__(512) void k_resize_fused(FuseArgs A)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t fuse_lds[];
    __shared__ int2 ytab[ORBX_FUSE_YTAB];          // (yofs, beta) of the rows this band needs, level after level
    __shared__ int ybase[ORBX_FUSE_MAX];
    const int band = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const int nl = A.b - A.a + 1;
    const int4 *bd = A.bands + band * nl;
#ifdef FUSE_TRACE
    unsigned long long ft_ = wall_clock64();
    if (threadIdx.x == 0) atomicAdd(&g_fuse_trace[15], 1ull);
#endif
    // every table read of the workgroup up front: the row tables of all its levels into LDS, the column tables of the first level
    // into registers (those of level l + 1 are fetched while level l is computed)
    int yb = 0;
    for (int l = A.a + 1; l <= A.b; l++) {
        const int4 rd = bd[l - A.a];
        const int rows = rd.w - rd.z;
        if (tid == 0) ybase[l] = yb;
        for (int r = tid; r < rows; r += 512) {
            const FuseLevel &D = A.lv[l];
            ytab[yb + r] = make_int2(D.tab.yofs[rd.z + r], *reinterpret_cast<const int *>(D.tab.beta + rd.z + r));
        }
        yb += rows;
    }
    FuseCols C, Cn;
    fuse_cols(C, A.lv[A.a + 1]);
    __syncthreads();
    for (int l = A.a + 1; l <= A.b; l++) {
        if (l < A.b) fuse_cols(Cn, A.lv[l + 1]);
        if (l == A.a + 1) fuse_level<true>(A, l, bd[l - 1 - A.a], bd[l - A.a], f, fuse_lds, ytab + ybase[l], C);     // (own0, own1, need0, need1), workgroup-uniform
        else fuse_level<false>(A, l, bd[l - 1 - A.a], bd[l - A.a], f, fuse_lds, ytab + ybase[l], C);
        C = Cn;
        __syncthreads();
        FUSE_T(l - A.a - 1);
    }
}

void orbx_launch_resize_fused(const FuseArgs &A, int nframes, size_t lds_bytes, hipStream_t s)
{
    if (lds_bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_resize_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(k_resize_fused, dim3(A.nbands, nframes), dim3(512), lds_bytes, s, A);
}

// -------------------------------------------------------------------------------------------------
// Levels a+1 .. b in ONE launch, second form (round 3): ONE WAVE = one 2-D tile of level b and everything above it.
// The band kernel above makes a workgroup of 8 waves walk six barrier-separated phases over full-width bands; its waves wait for
// each other at every level and it measured slower than the launches it replaces at batch 64.  Here a wave owns a TW x TH tile of
// level b of one frame and the region of every level a+1 .. b-1 that the tile's bilinear footprints reach; it computes level l+1
// of its region from level l of its region in its PRIVATE LDS (two buffers, alternating), so nothing but the wave's own LDS
// instructions orders the levels: no workgroup barrier, no other wave to wait for, and the ~2000 waves of a 64-frame batch drift
// through their levels independently, one wave's LDS latency under another's arithmetic.  Every level is written to memory
// by exactly one tile: the x and y OWNED ranges of the tiles partition each level (images of the level-b tile grid under the
// monotone xofs / yofs maps), the COMPUTED ranges (owned + footprint of the next level's computed range, lengthened to a multiple
// of 4: the arithmetic works in 4x4 blocks counted from the range's own first pixel) overlap and are computed twice (+30 % pixels on
// these small levels; rounding the ranges outwards to multiples of 4 in level coordinates, the first version, made it +90 %).  Planned on the host per axis (plan_tile_axis in orbx_capi.hip).
// Arithmetic: resize_block6 on 4x4 blocks, blocks of the tile's region dealt to the 64 lanes.  Needs RESIZE_FAST6 on every fused level
// and a >= 1 (the source of the first fused level is a pyramid level with slack behind its rows, never the caller's level 0).
// -------------------------------------------------------------------------------------------------
__constant__ uint32_t c_rs_rcp20[65] = {      // (1 << 20) / n + 1: idx / n == (idx * rcp) >> 20 for idx < 4096, n <= 64
    0, 1048577, 524289, 349526, 262145, 209716, 174763, 149797, 131073, 116509, 104858, 95326, 87382, 80660, 74899, 69906,
    65537, 61681, 58255, 55189, 52429, 49933, 47663, 45591, 43691, 41944, 40330, 38837, 37450, 36158, 34953, 33826,
    32769, 31776, 30841, 29960, 29128, 28340, 27595, 26887, 26215, 25576, 24967, 24386, 23832, 23302, 22796, 22311,
    21846, 21400, 20972, 20561, 20165, 19785, 19419, 19066, 18725, 18397, 18079, 17773, 17477, 17190, 16913, 16645, 16385};

#define RS_WSYNC()                                             \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        asm volatile("" ::: "memory");                         \
        __builtin_amdgcn_wave_barrier();                       \
        asm volatile("" ::: "memory");                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

typedef int rs_i32x4 __attribute__((ext_vector_type(4)));

// The tile's slices of level l's four tables, copied to the wave's LDS once (all levels, before any arithmetic): xofs and yofs RELATIVE to
// the source region's origin, columns past the level's last one (and before its first: a region may start up to three columns left
// of 0 so that its owned part starts on a block boundary) as copies of the nearest valid column.  With the tables in LDS the loops of
// the LDS-to-LDS levels hold no vector memory LOAD at all, so their stores to memory are never waited for.
// Layout at tab: xofs[cw] | alpha[cw] | yofs[ch] | beta[ch]  (cw, ch multiples of 4: every block reads four entries with one ds_read_b128).
__device__ __forceinline__ void tile_stage_tabs(const TileLevel &D, const int4 X, const int4 Y, int orgx, int orgy, int lane, int *tab)
{
    const int cw = X.w - X.z, ch = Y.w - Y.z;
    for (int i = lane; i < cw; i += 64) {
        const int x = min(max(X.z + i, 0), D.w - 1);
        tab[i] = D.tab.xofs[x] - orgx;
        tab[cw + i] = *reinterpret_cast<const int *>(D.tab.alpha + x);
    }
    for (int i = lane; i < ch; i += 64) {
        const int y = min(max(Y.z + i, 0), D.h - 1);
        tab[2 * cw + i] = D.tab.yofs[y] - orgy;
        tab[2 * cw + ch + i] = *reinterpret_cast<const int *>(D.tab.beta + y);
    }
}

// level l of the tile's region from level l - 1 (memory for the first fused level, the wave's LDS otherwise) into the wave's LDS, and
// -- STORE -- the owned pixels to memory straight from the registers.  The first fused level reads its source from memory: a store
// between two loads of its loop would put its acknowledgement (~1 us) on the path of the next load (vector memory operations retire
// in order), so that level's owned pixels go out in a pass of their own after the loop (tile_writeout).
template <int SRC, bool STORE>
__device__ __forceinline__ void tile_level(const TileArgs &A, int l, const int4 Xs, const int4 Ys, const int4 X, const int4 Y, int f, int lane,
                                           const uint8_t *sbuf, uint8_t *dbuf, const int *tab)
{
    const TileLevel &Sv = A.lv[l - 1], &D = A.lv[l];
    const int cw = X.w - X.z, ch = Y.w - Y.z;
    const int nbx = cw >> 2, nblk = nbx * (ch >> 2);
    const uint32_t rcp = c_rs_rcp20[nbx];
    const int ps = SRC == 0 ? Sv.stride : (Xs.w - Xs.z), pd = cw;
    const uint8_t *S = SRC == 0 ? Sv.base + (long long)f * Sv.frame : sbuf;
    const int orgy = SRC == 0 ? 0 : Ys.z;
    // last source row a block may touch: the level's last row, and in LDS also the region's last row (a block reads six rows whatever it
    // uses; the ones past the region carry zero weights)
    const int rmax = SRC == 0 ? Sv.h - 1 : min(Sv.h - 1 - orgy, (Ys.w - Ys.z) - 1);
    const int x1 = min(X.y, D.w);                                   // one past the last column this tile stores
    uint8_t *Dg = D.base + (long long)f * D.frame;
    for (int it = lane; it < nblk; it += 64) {
        const int by = (int)(__umul24((uint32_t)it, rcp) >> 20), bx = it - by * nbx;
        const rs_i32x4 a = *reinterpret_cast<const rs_i32x4 *>(tab + 4 * bx), b = *reinterpret_cast<const rs_i32x4 *>(tab + cw + 4 * bx);
        const rs_i32x4 c = *reinterpret_cast<const rs_i32x4 *>(tab + 2 * cw + 4 * by), d = *reinterpret_cast<const rs_i32x4 *>(tab + 2 * cw + ch + 4 * by);
        const uint32_t al[4] = {(uint32_t)b.x, (uint32_t)b.y, (uint32_t)b.z, (uint32_t)b.w};
        uint32_t sel[4], out[4];
        sel[0] = 0x0c010c00u;
        sel[1] = 0x0c010c00u + (uint32_t)(a.y - a.x) * 0x00010001u;
        sel[2] = 0x0c010c00u + (uint32_t)(a.z - a.x) * 0x00010001u;
        sel[3] = 0x0c010c00u + (uint32_t)(a.w - a.x) * 0x00010001u;
        resize_block6<SRC, false>(S, ps, c.x, rmax, a.x, sel, al, make_int4(c.x, c.y, c.z, c.w), make_uint4((uint32_t)d.x, (uint32_t)d.y, (uint32_t)d.z, (uint32_t)d.w), 0u, false, out);
        uint32_t *dp = reinterpret_cast<uint32_t *>(dbuf + (uint32_t)(__mul24(4 * by, pd) + 4 * bx));
#pragma unroll
        for (int r = 0; r < 4; r++) dp[r * (pd >> 2)] = out[r];
        if (STORE) {
            const int x4 = X.z + 4 * bx, y4 = Y.z + 4 * by;
            if (x4 >= X.x && x4 < x1) {                             // the owned columns start on a block boundary (planner); they end anywhere
                typedef uint32_t __attribute__((aligned(1))) u32a1;
                uint8_t *Dr = Dg + (uint32_t)(__mul24(y4, D.stride) + x4);
                if (x4 + 3 < x1) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (y4 + r >= Y.x && y4 + r < Y.y) *reinterpret_cast<u32a1 *>(Dr + r * D.stride) = out[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (y4 + r >= Y.x && y4 + r < Y.y) {
                            uint8_t *q = Dr + r * D.stride;
                            q[0] = (uint8_t)out[r];
                            if (x4 + 1 < x1) q[1] = (uint8_t)(out[r] >> 8);
                            if (x4 + 2 < x1) q[2] = (uint8_t)(out[r] >> 16);
                        }
                }
            }
        }
    }
}

// the owned part of level l: LDS -> memory, rows of dwords from the first owned column on (neither the LDS offset nor the address
// in memory is a multiple of 4 in general: two aligned LDS dwords funnel-shifted, one unaligned store); four dwords per lane in
// flight so that the LDS latency is paid once per four stores
__device__ __forceinline__ void tile_writeout(const TileLevel &D, const int4 X, const int4 Y, int f, int lane, const uint8_t *buf)
{
    const int x1 = min(X.y, D.w);
    const int nd = (x1 - X.x + 3) >> 2, rows = Y.y - Y.x;
    if (nd <= 0 || rows <= 0) return;
    const int n = nd * rows, pd = X.w - X.z;
    const uint32_t rcp = c_rs_rcp20[nd];
    uint8_t *Dg = D.base + (long long)f * D.frame;
    for (int i0 = lane; i0 < n; i0 += 256) {
        uint32_t v[4]; int xs[4], ys[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = min(i0 + 64 * k, n - 1);
            const int r = (int)(__umul24((uint32_t)i, rcp) >> 20), d = i - r * nd;
            ys[k] = Y.x + r; xs[k] = X.x + 4 * d;
            const uint32_t o = (uint32_t)(__mul24(ys[k] - Y.z, pd) + (xs[k] - X.z));
            const uint32_t *q = reinterpret_cast<const uint32_t *>(buf + (o & ~3u));
            v[k] = __builtin_amdgcn_alignbyte(q[1], q[0], o & 3u);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + 64 * k >= n) break;
            uint8_t *Dr = Dg + (uint32_t)(__mul24(ys[k], D.stride) + xs[k]);
            typedef uint32_t __attribute__((aligned(1))) u32a1;
            if (xs[k] + 3 < x1) *reinterpret_cast<u32a1 *>(Dr) = v[k];
            else {
                Dr[0] = (uint8_t)v[k];
                if (xs[k] + 1 < x1) Dr[1] = (uint8_t)(v[k] >> 8);
                if (xs[k] + 2 < x1) Dr[2] = (uint8_t)(v[k] >> 16);
            }
        }
    }
}

ORBX_TRACE_DEFINE(g_tile_trace, orbx_debug_tile_trace)
#ifdef ORBX_TRACE          // start / end of every wave on the 100 MHz wall clock + the XCC / CU it ran on (tools/dbg/tile_trace.py)
__device__ unsigned long long g_tile_span[3 * 8192];
extern "C" int orbx_debug_tile_span(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_span), sizeof(g_tile_span)); }
#define TILE_SPAN(i, v) do { if (lane == 0 && lb < 8192) g_tile_span[3 * lb + (i)] = (v); } while (0)
#else
#define TILE_SPAN(i, v) do { } while (0)
#endif
__global__ __launch_bounds__(64) void k_resize_tiles(TileArgs A, int ntiles, int total, uint32_t rcp_ntiles)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tile_lds[];
    const int lane = threadIdx.x;
    ORBX_TRACE_DECL;
    // contiguous eighths of the (frame, tile) list per XCD: the tiles of a frame share their source halos in one L2 (placement only)
    const int lb = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (lb >= total) return;
    TILE_SPAN(0, wall_clock64());
#ifdef ORBX_TRACE
    { uint32_t hwid, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); TILE_SPAN(2, ((unsigned long long)xcc << 32) | hwid); }
#endif
    int f = rcp_ntiles ? (int)__umulhi((uint32_t)lb, rcp_ntiles) : lb;
    f -= (f * ntiles > lb) ? 1 : 0;
    const int t = lb - f * ntiles;
    // the tile's ranges are wave-uniform: pinned to scalar registers so that they arrive by scalar loads (a vector load here would
    // wait behind every store the wave has issued: vector memory operations retire in order)
    const int tj = __builtin_amdgcn_readfirstlane((int)(__umul24((uint32_t)t, c_rs_rcp20[A.ntx]) >> 20));
    const int ti = __builtin_amdgcn_readfirstlane(t) - tj * A.ntx;
    const int nl = A.b - A.a + 1;
    // (constant address space + uniform index = s_load_dwordx4)
    typedef const __attribute__((address_space(4))) rs_i32x4 *rs_cptr4;
    const rs_cptr4 xrc = (rs_cptr4)(uintptr_t)(A.xr + ti * nl), yrc = (rs_cptr4)(uintptr_t)(A.yr + tj * nl);
    auto xr = [&](int k) { const rs_i32x4 v = xrc[k]; return make_int4(v.x, v.y, v.z, v.w); };
    auto yr = [&](int k) { const rs_i32x4 v = yrc[k]; return make_int4(v.x, v.y, v.z, v.w); };
    // every level's table slices -> LDS (one batch of loads, one wait)
    for (int l = A.a + 1; l <= A.b; l++) {
        const int4 Xp = xr(l - 1 - A.a), Yp = yr(l - 1 - A.a);
        tile_stage_tabs(A.lv[l], xr(l - A.a), yr(l - A.a), l == A.a + 1 ? 0 : Xp.z, l == A.a + 1 ? 0 : Yp.z, lane, reinterpret_cast<int *>(tile_lds + A.tab_off[l]));
    }
    RS_WSYNC();
    int4 Xs = xr(0), Ys = yr(0);
    for (int l = A.a + 1; l <= A.b; l++) {
        const int4 X = xr(l - A.a), Y = yr(l - A.a);
        const uint8_t *sbuf = tile_lds + A.lds_off[l - 1];
        uint8_t *dbuf = tile_lds + A.lds_off[l];
        const int *tab = reinterpret_cast<const int *>(tile_lds + A.tab_off[l]);
        if (l == A.a + 1) tile_level<0, false>(A, l, Xs, Ys, X, Y, f, lane, sbuf, dbuf, tab);
        else tile_level<1, true>(A, l, Xs, Ys, X, Y, f, lane, sbuf, dbuf, tab);
        Xs = X; Ys = Y;
        RS_WSYNC();
        if (l == A.a + 1) tile_writeout(A.lv[l], X, Y, f, lane, dbuf);      // no vector load follows in this wave: nothing waits for these stores
        ORBX_TRACE_STAMP(min(l - A.a - 1, 4));
    }
    ORBX_TRACE_STAMP(5);
    TILE_SPAN(1, wall_clock64());
    ORBX_TRACE_FLUSH(g_tile_trace);
}

void orbx_launch_resize_tiles(const TileArgs &A, int nframes, size_t lds_bytes, hipStream_t s)
{
    const int ntiles = A.ntx * A.nty, total = ntiles * nframes;
    const uint32_t rcp = ntiles > 1 ? (uint32_t)((1ull << 32) / (unsigned)ntiles + 1) : 0u;
    hipLaunchKernelGGL(k_resize_tiles, dim3((unsigned)((total + 7) & ~7)), dim3(64), lds_bytes, s, A, ntiles, total, rcp);
}

void orbx_launch_resize(const OrbxLevel &src, const OrbxLevel &dst, const ResizeTab &tab, int mode,
                        int nframes, const uint8_t *src_end, hipStream_t s)
{
    if (mode == RESIZE_AREA2) {
        dim3 grid((dst.w + 63) / 64, (dst.h + 3) / 4, nframes);
        hipLaunchKernelGGL(k_resize_area2, grid, dim3(64, 4), 0, s, src.base, src.stride, src.frame_stride,
                           dst.base, dst.w, dst.h, dst.stride, dst.frame_stride);
    } else if (mode == RESIZE_GENERIC) {
        dim3 grid((dst.w + 255) / 256, (dst.h + 3) / 4, nframes);
        hipLaunchKernelGGL(k_resize_linear, grid, dim3(64, 4), 0, s, src.base, src.w, src.h, src.stride,
                           src.frame_stride, dst.base, dst.w, dst.h, dst.stride, dst.frame_stride, tab);
    } else {
        const int nbx = (dst.w + 3) / 4, nblk = nbx * ((dst.h + 3) / 4);
        const uint32_t rcp = nbx > 1 ? (uint32_t)((1ull << 32) / (unsigned)nbx + 1) : 0u;
        dim3 grid((nblk + RESIZE_THREADS - 1) / RESIZE_THREADS, 1, nframes);
        if (mode == RESIZE_FAST6) {
            if (src_end)
                hipLaunchKernelGGL(k_resize_linear_4x4s<true>, grid, dim3(RESIZE_THREADS), 0, s, src.base, src.w, src.h, src.stride,
                                   src.frame_stride, dst.base, dst.w, dst.h, dst.stride, dst.frame_stride, tab, src_end, nbx, nblk, rcp);
            else
                hipLaunchKernelGGL(k_resize_linear_4x4s<false>, grid, dim3(RESIZE_THREADS), 0, s, src.base, src.w, src.h, src.stride,
                                   src.frame_stride, dst.base, dst.w, dst.h, dst.stride, dst.frame_stride, tab, src_end, nbx, nblk, rcp);
        } else if (src_end)
            hipLaunchKernelGGL(k_resize_linear_4x4<true>, grid, dim3(RESIZE_THREADS), 0, s, src.base, src.w, src.h, src.stride,
                               src.frame_stride, dst.base, dst.w, dst.h, dst.stride, dst.frame_stride, tab, src_end, nbx, nblk, rcp);
        else
            hipLaunchKernelGGL(k_resize_linear_4x4<false>, grid, dim3(RESIZE_THREADS), 0, s, src.base, src.w, src.h, src.stride,
                               src.frame_stride, dst.base, dst.w, dst.h, dst.stride, dst.frame_stride, tab, src_end, nbx, nblk, rcp);
    }
}
