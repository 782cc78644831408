// orbx_pyramid.hip -- ComputePyramid (src/ORBextractor.cc:1109-1137 of WChen09/My-SLAM) on gfx950.
#include "orbx_internal.h"

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
__global__ __launch_bounds__(256) void k_resize_linear_4x4(
    const uint8_t *__restrict__ src, int sw, int sh, int sstride, long long sframe,
    uint8_t *__restrict__ dst, int dw, int dh, int dstride, long long dframe, ResizeTab tab,
    const uint8_t *src_end, int nbx, int nblk, uint32_t rcp_nbx)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    // The 4x4 blocks of a frame are numbered row-major and dealt to threads linearly: a wave is 64 consecutive
    // blocks (a 256-px strip, wrapping at the row end), so no lane is lost to tile quantisation -- with 2-D tiles of
    // 128 x 32 px a 533 x 400 level launched 25 % more waves than it has work for, a 179 x 134 one 70 % more.
    const int t = blockIdx.x * 256 + threadIdx.x;
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
        dim3 grid((nblk + 255) / 256, 1, nframes);
        if (src_end)
            hipLaunchKernelGGL(k_resize_linear_4x4<true>, grid, dim3(256), 0, s, src.base, src.w, src.h, src.stride,
                               src.frame_stride, dst.base, dst.w, dst.h, dst.stride, dst.frame_stride, tab, src_end, nbx, nblk, rcp);
        else
            hipLaunchKernelGGL(k_resize_linear_4x4<false>, grid, dim3(256), 0, s, src.base, src.w, src.h, src.stride,
                               src.frame_stride, dst.base, dst.w, dst.h, dst.stride, dst.frame_stride, tab, src_end, nbx, nblk, rcp);
    }
}
