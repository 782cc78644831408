// orbx_describe.hip -- IC_Angle + GaussianBlur 7x7 + rBRIEF (src/ORBextractor.cc:79-149,1087-1103) on gfx950.
// All image arithmetic is integer; the fp32 in fastAtan2 and in the sample rotation is written with
// explicit __f*_rn intrinsics so no FMA contraction can happen; the file is also built
// with -ffp-contract=off.
#include "orbx_internal.h"
#include "orb_pattern_data.h"
#include <algorithm>
#include <math.h>
#ifndef DESC_RAW_STRIDE
#define DESC_RAW_STRIDE 44
#endif
#define DW_RAW_STRIDE_H DESC_RAW_STRIDE   // = DW_RAW_STRIDE below (host-side table builder)
#define DESC_R_H 18          // = DESC_R

__constant__ int c_gauss[7];
// Per-task constants of the IC_Angle dword tasks (31 rows x dwords 1..9 of the raw row that touch the disc: 213, padded to 256):
//   x = weights (u + 32 per byte, 0 outside the disc), y = 1 per byte inside the disc, z = byte offset of the dword in
//   the raw tile, w = v (row offset, -15..15).  Replaces ~20 instructions of mask arithmetic per task.
__constant__ uint4 c_mom_tab[256];
// Blur task lists.  Only blurred pixels a rotated sample can land on are produced: |rotated p| <= |p|max and each
// coordinate is rounded, so integer (r, q) with r*r + q*q <= (|p|max + 0.72)^2 (1125 of the 37 x 37 = 1369 positions;
// |p|max = 18.38 for the ORB pattern).  Row task = (row pair, 4-column group) -> 189 of 220; column task = 2 x 2 output
// block -> 300 of 361.  Entries: row: raw byte offset | P dword offset << 16; column: P BYTE offset | bl byte offset
// << 16 (two halves the kernel uses as they are; a task's first column is its bl offset mod 80).  0xFFFFFFFF = no task.
__constant__ uint32_t c_row_task[192];
__constant__ uint32_t c_col_task[320];
// The same eight words per lane as two 16-byte entries: [lane] = (row task 0..2, column task 0), [64 + lane] = (column task 1..4).
// The L1 takes a 64-lane load as 16 four-lane accesses whatever its width: two wide loads cost a quarter of eight narrow ones.
__constant__ uint4 c_task8[128];
// rBRIEF sample pairs as floats: (x0, x1, y0, y1) of bit b -- the int8 -> float conversions done once on the host
__constant__ float4 c_pat_f[256];

int orbx_upload_constants(const int umax[16], const int gauss_k[7])
{
    for (int v = 0; v < 16; v++)
        if (umax[v] < 0 || umax[v] > 15) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), gauss_k, sizeof(int) * 7) != hipSuccess) return -1;
    static uint4 mt[256];
    int nmt = 0;
    for (int t = 0; t < 256; t++) mt[t] = make_uint4(0u, 0u, 0u, 0u);
    for (int t = 0; t < 31 * 9; t++) {
        const int vr = t / 9, dw = t - vr * 9 + 1, v = vr - 15;
        const int um = umax[v < 0 ? -v : v];
        const int lo = 21 - um, hi = 21 + um, c0 = 4 * dw;                  // valid raw columns; first column of this dword
        const int nlo = std::min(std::max(lo - c0, 0), 4), nhi = std::min(std::max(c0 + 3 - hi, 0), 4);
        const uint32_t mlo = nlo >= 4 ? 0u : (0xFFFFFFFFu << (8 * nlo));
        const uint32_t mhi = nhi >= 4 ? 0u : (0xFFFFFFFFu >> (8 * nhi));
        const uint32_t msk = mlo & mhi;
        // byte j holds u + 32 = (c0 + j - 21) + 32 = c0 + j + 11  (15..50: no carry between bytes)
        const uint32_t wfull = (uint32_t)(c0 + 11) * 0x01010101u + 0x03020100u;
        if (msk == 0) continue;          // dword entirely outside the disc: 213 tasks remain (4 wave passes)
        if (nmt >= 256) return -1;
        mt[nmt++] = make_uint4(wfull & msk, 0x01010101u & msk, (uint32_t)((6 + vr) * DW_RAW_STRIDE_H + c0), (uint32_t)v);
    }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_mom_tab), mt, sizeof(mt)) != hipSuccess) return -1;
    {
        int maxr2 = 0;
        for (int i = 0; i < 512; i++) maxr2 = std::max(maxr2, (int)ORBX_PATTERN[2 * i] * ORBX_PATTERN[2 * i] + (int)ORBX_PATTERN[2 * i + 1] * ORBX_PATTERN[2 * i + 1]);
        const double R = sqrt((double)maxr2) + 0.72;
        if (R > DESC_R_H + 1.5) return -1;   // the 37 x 37 tile would not hold the pattern
        bool need[37][37];
        for (int i = 0; i < 37; i++)
            for (int j = 0; j < 37; j++) need[i][j] = (double)((i - 18) * (i - 18) + (j - 18) * (j - 18)) <= R * R;
        static uint32_t rt[192], ct[320];
        int nr = 0, nc = 0;
        for (int rp = 21; rp >= 0; rp--)           // highest row pair first: see DescLds (the row pass runs in place)
            for (int gq = 0; gq < 10; gq++) {
                bool ok = false;
                for (int row = 2 * rp; row <= 2 * rp + 1; row++)
                    for (int col = 4 * gq; col < 4 * gq + 4 && col < 37; col++)
                        for (int i = std::max(row - 6, 0); i <= std::min(row, 36); i++) ok = ok || need[i][col];
                if (ok) { if (nr >= 192) return -1; rt[nr++] = (uint32_t)((2 * rp) * DW_RAW_STRIDE_H + 4 * gq) | ((uint32_t)(rp * 40 + 4 * gq) << 16); }
            }
        for (int q = 0; q < 19; q++)
            for (int cp = 0; cp < 19; cp++) {
                bool ok = false;
                for (int i = 2 * q; i <= 2 * q + 1 && i < 37; i++)
                    for (int j = 2 * cp; j <= 2 * cp + 1 && j < 37; j++) ok = ok || need[i][j];
                if (ok) { if (nc >= 320) return -1; ct[nc++] = (uint32_t)(4 * (q * 40 + 2 * cp)) | ((uint32_t)(80 * q + 2 * cp) << 16); }   // P byte offset | bl byte offset << 16
            }
        for (; nr < 192; nr++) rt[nr] = 0xFFFFFFFFu;
        for (; nc < 320; nc++) ct[nc] = 0xFFFFFFFFu;
        if (hipMemcpyToSymbol(HIP_SYMBOL(c_row_task), rt, sizeof(rt)) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(c_col_task), ct, sizeof(ct)) != hipSuccess) return -1;
        static uint4 t8[128];
        for (int ln = 0; ln < 64; ln++) {
            t8[ln] = make_uint4(rt[ln], rt[64 + ln], rt[128 + ln], ct[ln]);
            t8[64 + ln] = make_uint4(ct[64 + ln], ct[128 + ln], ct[192 + ln], ct[256 + ln]);
        }
        if (hipMemcpyToSymbol(HIP_SYMBOL(c_task8), t8, sizeof(t8)) != hipSuccess) return -1;
    }
    static float4 pf[256];
    for (int b = 0; b < 256; b++)
        pf[b] = make_float4((float)ORBX_PATTERN[4 * b], (float)ORBX_PATTERN[4 * b + 2], (float)ORBX_PATTERN[4 * b + 1], (float)ORBX_PATTERN[4 * b + 3]);
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_pat_f), pf, sizeof(pf)) != hipSuccess) return -1;
    return 0;
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
// Canonical semantics = correctly rounded cosf/sinf; tests/tools/verify_sincos.py checks this
// routine against the x87 long-double libm over every fp32 input of the domain.
// A double constant held in a SCALAR register pair (two s_mov_b32) and opaque to the optimiser.  Left to itself the compiler puts each of the
// 21 reduction / polynomial constants into a VECTOR register pair right before its use (v_fmac_f64 wants the addend in its destination):
// 36 v_mov_b32 per keypoint, a third of this routine's vector instructions, all 64 lanes moving the same words.  v_fma_f64 takes the scalar
// pair as its one scalar operand.  The value is unchanged: results are bit-identical (tests/tools/verify_sincos.py).
__device__ __forceinline__ double KC(double c)
{
    unsigned long long u = __builtin_bit_cast(unsigned long long, c);
    asm volatile("" : "+s"(u));
    return __builtin_bit_cast(double, u);
}
__device__ __forceinline__ void sincos_cr(float theta, float *cs, float *sn)
{
    const double x = (double)theta;
    const double kd = rint(x * KC(0.6366197723675814));
    const int k = (int)kd;
    double r = fma(-kd, KC(1.5707963267341256), x);          // exact: 33-bit head times k <= 4
    r = fma(-kd, KC(6.077100506506192e-11), r);
    const double z = r * r;
    double ps = KC(-8.22063524662433e-18);
    ps = fma(ps, z, KC(2.8114572543455206e-15));
    ps = fma(ps, z, KC(-7.647163731819816e-13));
    ps = fma(ps, z, KC(1.6059043836821613e-10));
    ps = fma(ps, z, KC(-2.505210838544172e-08));
    ps = fma(ps, z, KC(2.7557319223985893e-06));
    ps = fma(ps, z, KC(-0.0001984126984126984));
    ps = fma(ps, z, KC(0.008333333333333333));
    ps = fma(ps, z, KC(-0.16666666666666666));
    const double s = fma(r * z, ps, r);
    double pc = KC(4.110317623312165e-19);
    pc = fma(pc, z, KC(-1.5619206968586225e-16));
    pc = fma(pc, z, KC(4.779477332387385e-14));
    pc = fma(pc, z, KC(-1.1470745597729725e-11));
    pc = fma(pc, z, KC(2.08767569878681e-09));
    pc = fma(pc, z, KC(-2.755731922398589e-07));
    pc = fma(pc, z, KC(2.48015873015873e-05));
    pc = fma(pc, z, KC(-0.001388888888888889));
    pc = fma(pc, z, KC(0.041666666666666664));
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

// exported for the exhaustive sincos check (tests/tools/verify_sincos.py)
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

// One wave per keypoint, four independent waves per workgroup (no workgroup barrier).  LDS per wave (all three in ONE buffer, below):
//   raw  44 x 44 B   the 43x43 tile around the keypoint, column 0 at byte 0 (re-aligned at load time); 11 dwords per row: the 5 x 11
//                    lanes of a tile-store pass and the (row pair, column group) tasks of the row pass fall on distinct banks
//   P    22 x 160 B  row-blurred values, two vertically adjacent rows packed per dword (u16 | u16 << 16)
//   bl   37 x 40 B   blurred 37x37
// Row pass: 4 outputs from 3 dword reads, v_dot4_u32_u8 against shifted tap constants (taps are u8).
// Column pass: the vertical pairs make every output 4 x v_dot2_u32_u16; exact integer sums, one
// rounding at the end -- identical to row-then-column on u8 -> int32 -> u8 (OpenCV's fixed-point path).
#define DW_RAW_STRIDE DESC_RAW_STRIDE
#define DW_RAW_ROWS 44
#define DW_P_STRIDE 40      // dwords per row pair
#define DW_P_ROWS 22
#define DW_BL_STRIDE 40

// raw, P and bl share ONE buffer (3520 B per wave instead of 5632: 11 workgroups per CU by LDS instead of 7).  raw (2112 B) and bl
// (1520 B) start at byte 0 like P.  That works because the passes walk it in the right direction and a wave's LDS instructions
// execute in order (all loads of a pass iteration are issued before its stores):
//   row pass     reads raw row pair r (bytes 96 r ..) and writes P row pair r (bytes 160 r ..), HIGHEST pair first: what an
//                iteration writes lies above every raw row the later iterations still read (160 r >= 96 (r' + 1) for r' < r);
//   column pass  reads P pairs q .. q + 3 and writes bl rows 2q, 2q + 1 (bytes 80 q .. 80 q + 120), LOWEST q first: below every P
//                pair a later iteration reads (160 q' for q' >= q, q >= 2; the first iteration holds q = 0 .. 3 entirely).
struct __attribute__((aligned(16))) DescLds {
    uint8_t buf[DW_P_STRIDE * DW_P_ROWS * 4];   // 3520 B
};


// a wave-uniform pointer pinned to scalar registers (keeps "scalar base + 32-bit lane offset" from being re-associated into
// 64-bit vector arithmetic)
typedef const __attribute__((address_space(1))) uint8_t *gptr_u8;   // explicitly global: the integer round trip would leave a flat pointer
__device__ __forceinline__ gptr_u8 scalar_ptr(const uint8_t *p)
{
    const uint64_t b = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return (gptr_u8)(((uint64_t)hi << 32) | lo);
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((may_alias)) u32x2_ma;
typedef u32x2 __attribute__((aligned(4))) U2a4;   // two dwords at a 4-byte aligned address (global_load_dwordx2)

// sum over the 64 lanes in six DPP additions (quad swaps, half-row and row mirrors, row broadcasts); every lane of the
// result is the same scalar
__device__ __forceinline__ int wave_sum_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);     // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);     // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);    // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);    // row_mirror: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Between two passes over the wave's LDS buffer.  The hardware needs nothing (a wave's LDS instructions execute in order); the
// COMPILER must not move an access across: the wavefront-scope fences alone did not stop it from hoisting the column pass's
// first loads above the row pass's last stores once the two used types that type-based alias analysis tells apart (round 3), so
// an empty asm with a memory clobber stands next to them.
#define DSYNC()                                                \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        asm volatile("" ::: "memory");                         \
        __builtin_amdgcn_wave_barrier();                       \
        asm volatile("" ::: "memory");                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

ORBX_TRACE_DEFINE(g_desc_trace, orbx_debug_desc_trace)

// 8 waves per SIMD: 63 VGPRs (68 without the bound) and 14 KB of LDS per workgroup; this kernel hides its latencies with resident waves
#ifndef DESC_WAVES
#define DESC_WAVES 1   // waves (keypoints) per workgroup
#endif
__global__ __launch_bounds__(64 * DESC_WAVES, 8) void k_describe(
    OrbxPlan plan, OrbxWork wk, orbx_keypoint *__restrict__ kps, uint8_t *__restrict__ desc,
    int32_t *__restrict__ counts, int32_t *__restrict__ status, int l0_aligned, int wg_per_frame, int nwg, uint32_t wg_rcp)
{
    __shared__ DescLds lds[DESC_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD x (workgroups x, x + 8, ...) takes the x-th contiguous eighth of the (frame, keypoint) list: the patches of
    // one frame overlap heavily, and its whole pyramid fits the XCD's L2.  Speed only; the padded grid keeps it a bijection.
    const int lb = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (lb >= nwg) return;
    // lb / wg_per_frame without the integer-division expansion: multiply by floor(2^32 / d) + 1, one correction step
    int f = wg_rcp ? (int)__umulhi((uint32_t)lb, wg_rcp) : lb;   // wg_rcp == 0: one workgroup per frame
    f -= (f * wg_per_frame > lb) ? 1 : 0;
    const int g = (lb - f * wg_per_frame) * DESC_WAVES + wave;
    DescLds &S = lds[wave];
    ORBX_TRACE_DECL;

    // which level's list holds keypoint g of this frame: lane i loads level i's count (one memory round trip, not
    // nlevels dependent scalar loads), an inclusive scan over the first 16 lanes (DPP row shifts), a ballot
    int total, l, idx;
    {
        const int nl = plan.nlevels;                                            // <= ORBX_MAX_LEVELS = 16: one DPP row
        int incl = lane < nl ? (int)wk.nk[f * nl + lane] : 0;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);   // row_shr:1, lanes without a source add 0
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
        total = __builtin_amdgcn_readlane(incl, nl - 1);
        l = __popcll(__builtin_amdgcn_ballot_w64(lane < nl && g >= incl));      // levels that end at or before g
        idx = g - (l > 0 ? __builtin_amdgcn_readlane(incl, l - 1) : 0);
        if (l >= nl) l = -1;
    }
    if (g == 0 && lane == 0) {
        const uint32_t e = wk.errflags[f];
        wk.errflags[f] = 0;   // self-cleaning for the next call
        counts[f] = min(total, plan.out_cap);
        status[f] = (e & ERRF_CAND_OVERFLOW) ? ORBX_E_CAND_OVERFLOW
                  : (e & ERRF_TREE_OVERFLOW) ? ORBX_E_TREE_OVERFLOW
                  : (total > plan.out_cap)   ? ORBX_E_CAPACITY : ORBX_OK;
    }
    if (l < 0 || g >= plan.out_cap) return;
    const OrbxLevel &L = plan.lv[l];
    const OrbxCand kc = wk.sel[(long long)f * plan.list_frame + L.list_off + idx];
    const int x = __builtin_amdgcn_readfirstlane((int)(kc.xy & 0xFFFFu));
    const int y = __builtin_amdgcn_readfirstlane((int)(kc.xy >> 16));
    const uint8_t *img = L.base + (long long)f * L.frame_stride;

    // this lane's four sample pairs (bits lane, lane+64, lane+128, lane+192); issued early
    uint4 mom[4];
#pragma unroll
    for (int it = 0; it < 4; it++) mom[it] = c_mom_tab[it * 64 + lane];
    uint32_t rtask[3], ctask[5];
#ifndef DESC_NO_TASK8
    {
        const uint4 ta = c_task8[lane], tb = c_task8[64 + lane];
        rtask[0] = ta.x; rtask[1] = ta.y; rtask[2] = ta.z;
        ctask[0] = ta.w; ctask[1] = tb.x; ctask[2] = tb.y; ctask[3] = tb.z; ctask[4] = tb.w;
    }
#else
#pragma unroll
    for (int it = 0; it < 3; it++) rtask[it] = c_row_task[it * 64 + lane];
#pragma unroll
    for (int it = 0; it < 5; it++) ctask[it] = c_col_task[it * 64 + lane];
#endif

    // ---- raw tile -> LDS ----
    const int x0 = x - 21, y0 = y - 21;
    const int row_bytes = (l == 0) ? L.w : L.stride;   // bytes of a row that may be touched
    const bool fast = (x0 >= 0) && (y0 >= 0) && (y + 21 < L.h) && (x + 21 < L.w) &&   // no reflection needed
                      (x + 27 < row_bytes);                                            // dword over-read stays in the row
#ifndef DESC_NO_TILE3
    if (fast && (l != 0 || l0_aligned)) {
        // Rows start on a dword.  16 rows per pass, FOUR lanes per row: lane j of a row fetches dwords 3j .. 3j + 2 of the row segment
        // in one 12-byte load; the dword that follows them comes from the next lane of the quad (one DPP move), and the lane
        // funnel-shifts its three output dwords.  A quad of lanes is what the L1 serves per clock (it takes any 64-lane load as 16
        // four-lane accesses): the 43 rows are 43 accesses in 3 loads instead of 135 in 9 (8-byte loads, 12 lanes per row), and
        // the L1 was this kernel's busiest unit (profiles/r03_mem_counters.txt).
        const int rr = lane >> 2, j = lane & 3;
        const uint32_t voff = (uint32_t)(__mul24(y0 + rr, L.stride) + (x0 & ~3) + 12 * j);
        const uint32_t xo = (uint32_t)x0 & 3u;
        const int step16 = 16 * L.stride;
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        typedef u32x3 __attribute__((aligned(4))) U3a4;
        u32x3 gv[3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            gv[p] = (u32x3){0u, 0u, 0u};
            if (p < 2 || rr < DESC_RAW - 32) gv[p] = *(const __attribute__((address_space(1))) U3a4 *)(scalar_ptr(img + (long long)(p * step16)) + voff);
        }
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gv[p].x, 0xF9, 0xf, 0xf, false);   // quad_perm:[1,2,3,3]
            if (p < 2 || rr < DESC_RAW - 32) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(&S.buf[(16 * p + rr) * DW_RAW_STRIDE + 12 * j]);
                dst[0] = __builtin_amdgcn_alignbyte(gv[p].y, gv[p].x, xo);
                dst[1] = __builtin_amdgcn_alignbyte(gv[p].z, gv[p].y, xo);
                if (j < 3) dst[2] = __builtin_amdgcn_alignbyte(nx, gv[p].z, xo);   // the row's 12th dword lies past the 44-byte LDS row
            }
        }
    } else
#endif
    if (fast) {
        // 5 rows per pass, 12 lanes per row; lane d < 11 of a row fetches dwords d and d + 1 of the row segment in one
        // 8-byte load and funnel-shifts its own output dword out of them (no cross-lane traffic).
        const int rr = lane / 12, d = lane - rr * 12;
        const int step5 = 5 * L.stride;
        if (lane < 60 && d < 11) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(&S.buf[rr * DW_RAW_STRIDE + 4 * d]);
            U2a4 gv[9];
            if (l != 0 || l0_aligned) {   // rows start on a dword: one shift for the whole tile
                // byte offsets inside one frame fit 31 bits; one 24-bit multiply per lane, the row steps are scalar
                // (scalar row base) + (32-bit unsigned lane offset): the global_load saddr form, no 64-bit vector arithmetic
                const uint32_t voff = (uint32_t)(__mul24(y0 + rr, L.stride) + (x0 & ~3) + 4 * d);
                const uint32_t xo = (uint32_t)x0 & 3u;
#pragma unroll
                for (int p = 0; p < 8; p++) gv[p] = *(const __attribute__((address_space(1))) U2a4 *)(scalar_ptr(img + (long long)(p * step5)) + voff);
                if (rr < 3) gv[8] = *(const __attribute__((address_space(1))) U2a4 *)(scalar_ptr(img + (long long)(8 * step5)) + voff);
#pragma unroll
                for (int p = 0; p < 8; p++) dst[p * (5 * DW_RAW_STRIDE / 4)] = __builtin_amdgcn_alignbyte(gv[p].y, gv[p].x, xo);
                if (rr < 3) dst[8 * (5 * DW_RAW_STRIDE / 4)] = __builtin_amdgcn_alignbyte(gv[8].y, gv[8].x, xo);
            } else {                      // caller-owned level 0 with an odd pitch: the shift differs from row to row
                const uint8_t *b0 = img + (__mul24(y0 + rr, L.stride) + x0) + 4 * d;
                uint32_t xo[9];
#pragma unroll
                for (int p = 0; p < 9; p++) {
                    const uint8_t *pa = b0 + p * step5;
                    xo[p] = (uint32_t)reinterpret_cast<uintptr_t>(pa) & 3u;
                    if (p < 8 || rr < 3) gv[p] = *reinterpret_cast<const U2a4 *>(pa - xo[p]);
                }
#pragma unroll
                for (int p = 0; p < 9; p++)
                    if (p < 8 || rr < 3) dst[p * (5 * DW_RAW_STRIDE / 4)] = __builtin_amdgcn_alignbyte(gv[p].y, gv[p].x, xo[p]);
            }
        }
    } else {   // tile crosses the image border (reflect-101): byte path, 2-7 % of a level's keypoints (those within 21 px of its edge)
        // lane = tile column, its reflected source column computed once; one tile row per load with the reflected row index and the row's
        // base address on the SCALAR unit (global_load_ubyte, scalar base + lane offset): about ten vector instructions for the tile.
        // (The byte-per-lane loop over all 1849 pixels this replaces spent ~1000 on index arithmetic -- twice a whole keypoint.)
        if (lane < DESC_RAW) {
            const uint32_t gx = (uint32_t)reflect101(x0 + lane, L.w);
            uint8_t *dst = &S.buf[lane];
#pragma unroll 1
            for (int r0 = 0; r0 < DESC_RAW; r0 += 11) {   // 43 = 3 x 11 + 10: eleven loads in flight
                uint8_t v[11];
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    const int gy = reflect101(min(y0 + r0 + k, y0 + DESC_RAW - 1), L.h);   // wave-uniform
                    v[k] = *(scalar_ptr(img + (long long)gy * L.stride) + gx);
                }
#pragma unroll
                for (int k = 0; k < 11; k++)
                    if (r0 + k < DESC_RAW) dst[(r0 + k) * DW_RAW_STRIDE] = v[k];
            }
        }
    }
    DSYNC();
    ORBX_TRACE_STAMP(0);

    // ---- IC_Angle (:79-106): m10 = sum u*I, m01 = sum v*I over |u| <= umax[|v|] ----
    // dword tasks (row 0..30, dword 1..9 of the raw row); weights (u+32) keep the dot product unsigned
    int m10 = 0, m01 = 0;
    {
        static_assert(DW_RAW_STRIDE == DW_RAW_STRIDE_H, "c_mom_tab offsets");
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const uint4 e = mom[it];   // zero past the last task
            const uint32_t pix = *reinterpret_cast<const uint32_t *>(&S.buf[e.z]);
            const int sA = (int)__builtin_amdgcn_udot4(pix, e.x, 0u, false);
            const int sB = (int)__builtin_amdgcn_udot4(pix, e.y, 0u, false);
            m10 += sA - 32 * sB;
            m01 += __mul24((int)e.w, sB);
        }
        m10 = wave_sum_dpp(m10);
        m01 = wave_sum_dpp(m01);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    ORBX_TRACE_STAMP(1);
    // ---- 7x7 Gaussian, row pass (exact, <= 65535), two rows per task, packed vertically ----
    const uint32_t k0 = (uint32_t)c_gauss[0], k1 = (uint32_t)c_gauss[1], k2 = (uint32_t)c_gauss[2], k3 = (uint32_t)c_gauss[3];
    // Output j of a group is the 7-tap window at bytes j .. j+6 of the 12 bytes (w0, w1, w2).  Instead of shifting the
    // window into place (v_alignbyte_b32) the taps are shifted: one packed-tap constant per (j, dword) -- 2 + 2 + 3 + 3
    // v_dot4_u32_u8 for the four outputs and no byte shuffles.  Taps are symmetric: k4 = k2, k5 = k1, k6 = k0.
    const uint32_t T0a = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), T0b = k2 | (k1 << 8) | (k0 << 16);
    const uint32_t T1a = (k0 << 8) | (k1 << 16) | (k2 << 24),      T1b = k3 | (k2 << 8) | (k1 << 16) | (k0 << 24);
    const uint32_t T2a = (k0 << 16) | (k1 << 24),                  T2b = k2 | (k3 << 8) | (k2 << 16) | (k1 << 24), T2c = k0;
    const uint32_t T3a = (k0 << 24),                               T3b = k1 | (k2 << 8) | (k3 << 16) | (k2 << 24), T3c = k1 | (k0 << 8);
    static_assert(DW_P_STRIDE == 40 && DW_BL_STRIDE == 40 && DESC_R == DESC_R_H, "task tables");
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const uint32_t te = rtask[it];
        if (te != 0xFFFFFFFFu) {
            const uint32_t *ra = reinterpret_cast<const uint32_t *>(&S.buf[te & 0xFFFFu]);
            const uint32_t *rbp = reinterpret_cast<const uint32_t *>(&S.buf[(te & 0xFFFFu) + DW_RAW_STRIDE]);
            const uint32_t a0 = ra[0], a1 = ra[1], a2 = ra[2];
            const uint32_t b0 = rbp[0], b1 = rbp[1], b2 = rbp[2];
            uint32_t o[4];
#define DOT4(A, K, ACC) __builtin_amdgcn_udot4((A), (K), (ACC), false)
#define ROW4(W0, W1, W2, R0, R1, R2, R3)                                   \
    R0 = DOT4(W0, T0a, DOT4(W1, T0b, 0u));                                 \
    R1 = DOT4(W0, T1a, DOT4(W1, T1b, 0u));                                 \
    R2 = DOT4(W0, T2a, DOT4(W1, T2b, DOT4(W2, T2c, 0u)));                  \
    R3 = DOT4(W0, T3a, DOT4(W1, T3b, DOT4(W2, T3c, 0u)));
            uint32_t ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
            ROW4(a0, a1, a2, ra0, ra1, ra2, ra3)
            ROW4(b0, b1, b2, rb0, rb1, rb2, rb3)
#undef ROW4
#undef DOT4
            // (row a | row b << 16), every sum <= 65535: one byte permute per output
            o[0] = __builtin_amdgcn_perm(rb0, ra0, 0x05040100u); o[1] = __builtin_amdgcn_perm(rb1, ra1, 0x05040100u);
            o[2] = __builtin_amdgcn_perm(rb2, ra2, 0x05040100u); o[3] = __builtin_amdgcn_perm(rb3, ra3, 0x05040100u);
            // one ds_write_b128 (the P offset is a multiple of four dwords and the buffer is 16-byte aligned; without the promise the
            // compiler splits the store into two ds_write2_b32, whose four dword stores at a 16-byte lane pitch are 4-way bank
            // conflicts: 120 LDS cycles per keypoint instead of 39, tools/lds_model_describe.py)
#ifndef DESC_NO_B128
            *reinterpret_cast<uint4 *>(__builtin_assume_aligned(&reinterpret_cast<uint32_t *>(S.buf)[te >> 16], 16)) = make_uint4(o[0], o[1], o[2], o[3]);
#else
            *reinterpret_cast<uint4 *>(&reinterpret_cast<uint32_t *>(S.buf)[te >> 16]) = make_uint4(o[0], o[1], o[2], o[3]);
#endif
        }
    }
    DSYNC();
    ORBX_TRACE_STAMP(2);

    // this lane's sample pairs for the descriptor (bits lane, lane + 64, lane + 128, lane + 192): in flight during the column pass
    float4 pf[4];
#pragma unroll
    for (int j = 0; j < 4; j++) pf[j] = c_pat_f[j * 64 + lane];

    // ---- column pass: lane = (column pair, block of row pairs); 4 x dot2 per output ----
    uint8_t *bl = S.buf;   // raw is dead from here on; bl grows from byte 0 under the P rows the column pass has finished with
    {
        const uint32_t K01 = k0 | (k1 << 16), K23 = k2 | (k3 << 16), K45 = k2 | (k1 << 16), K6_ = k0;       // even rows
        const uint32_t K_0 = k0 << 16, K12 = k1 | (k2 << 16), K34 = k3 | (k2 << 16), K56 = k1 | (k0 << 16); // odd rows
        const int simd_cols = plan.blur_mode == 1 ? (L.w & ~3) : 0;
        // task = 2 x 2 block of outputs from four 8-byte reads (c_col_task)
#pragma unroll
        for (int it = 0; it < 5; it++) {
            const uint32_t te = ctask[it];
            if (te != 0xFFFFFFFFu) {
                const uint32_t *pin = reinterpret_cast<const uint32_t *>(&S.buf[te & 0xFFFFu]);
                uint8_t *pout = &bl[te >> 16];
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                // volatile: four ds_read_b64 (two lane groups of 32, banks mod 64: 2 LDS cycles each when conflict-free).  Left alone the
                // compiler pairs them into ds_read2_b64, which is serviced as 2 x 4 groups of 16 lanes on 32 banks: 148 instead of 76
                // LDS cycles per keypoint for this pass (tools/lds_model_describe.py).  The LDS address space is spelled out (a volatile
                // access through a generic pointer would be a flat load) and the type may alias anything.
                // (w stays a HIP uint2 -- real members: __builtin_bit_cast of an ext-vector ELEMENT, `bit_cast<us2>(v.y)`, read the first
                // four bytes of the whole vector with this compiler, i.e. v.x)
                uint2 w[4];
#ifndef DESC_NO_B64
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32x2 t_ = *(const volatile __attribute__((address_space(3))) u32x2_ma *)(pin + j * DW_P_STRIDE);
                    w[j].x = t_.x; w[j].y = t_.y;
                }
#else
#pragma unroll
                for (int j = 0; j < 4; j++) w[j] = *reinterpret_cast<const uint2 *>(pin + j * DW_P_STRIDE);
#endif
#define D2(A, K, ACC) __builtin_amdgcn_udot2(__builtin_bit_cast(us2, (A)), __builtin_bit_cast(us2, (uint32_t)(K)), (ACC), false)
                // the rounding constant rides in the accumulator: every sum below is (exact sum + 32768) <= 0x01017FFF
                const uint32_t e0 = D2(w[0].x, K01, D2(w[1].x, K23, D2(w[2].x, K45, D2(w[3].x, K6_, 32768u))));
                const uint32_t e1 = D2(w[0].y, K01, D2(w[1].y, K23, D2(w[2].y, K45, D2(w[3].y, K6_, 32768u))));
                const uint32_t o0 = D2(w[0].x, K_0, D2(w[1].x, K12, D2(w[2].x, K34, D2(w[3].x, K56, 32768u))));
                const uint32_t o1 = D2(w[0].y, K_0, D2(w[1].y, K12, D2(w[2].y, K34, D2(w[3].y, K56, 32768u))));
#undef D2
                // (sum >> 16) of two neighbouring outputs as one u16 pair (a byte permute); saturation to 255 (a sum of bright
                // pixels reaches 257: the taps add up to 257) and the packing into two bytes are one v_sat_pk_u8_i16
                uint32_t pe = __builtin_amdgcn_perm(e1, e0, 0x07060302u), po = __builtin_amdgcn_perm(o1, o0, 0x07060302u);
                if (simd_cols != 0) {   // wave-uniform; x86 SSE2 path for columns < simd_cols: an exact .5 tie rounds to even
                    const int c = (int)((te >> 16) % 80u);   // first column of the task: bl offset = 80 q + column
                    const uint32_t s4[4] = {e0, e1, o0, o1};
                    uint32_t dec[4];
#pragma unroll
                    for (int z = 0; z < 4; z++)
                        dec[z] = ((x - DESC_R + c + (z & 1) < simd_cols) && ((s4[z] & 0xFFFFu) == 0u) && ((s4[z] >> 16) & 1u)) ? 1u : 0u;
                    pe -= dec[0] | (dec[1] << 16);   // a decremented value is odd, hence >= 1: no borrow between the halves
                    po -= dec[2] | (dec[3] << 16);
                }
                uint32_t be, bo;
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(be) : "v"(pe));
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(bo) : "v"(po));
                *reinterpret_cast<uint16_t *>(pout) = (uint16_t)be;
                // the last row pair has no second row: its store lands in row 37 of the 44-row area, which nothing reads
                *reinterpret_cast<uint16_t *>(pout + DW_BL_STRIDE) = (uint16_t)bo;
            }
        }
    }
    DSYNC();
    ORBX_TRACE_STAMP(3);

    // ---- rBRIEF (:110-149) ----
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float a, b;
    sincos_cr(__fmul_rn(angle, factorPI), &a, &b);
    unsigned long long bits[4];
    // Both samples of a pair in one packed-fp32 lane pair (v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations, two per
    // instruction; no FMA is formed, the file is built with -ffp-contract=off).  cvRound = round-half-even = adding
    // 1.5 * 2^23: the integer then sits in the low mantissa bits (|value| <= 19), so index = mad24(ri, 40, qi) - const
    // (unsigned wrap-around arithmetic on the index, never on the pointer).
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 A = {a, a}, Bv = {b, b}, MAGIC = {12582912.f, 12582912.f};
    // bits(MAGIC + r) = 0x4B400000 + r; mad24 sees its low 24 bits, 0x400000 + r
    const uint32_t IDX_BIAS = 0x400000u * (uint32_t)DW_BL_STRIDE + 0x4B400000u - (uint32_t)(DESC_R * DW_BL_STRIDE + DESC_R);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const f2 PX = {pf[j].x, pf[j].y}, PY = {pf[j].z, pf[j].w};      // (x0, x1), (y0, y1)
        const f2 R = (PX * Bv + PY * A) + MAGIC;                          // rows:    x * sin + y * cos
        const f2 Q = (PX * A - PY * Bv) + MAGIC;                          // columns: x * cos - y * sin
        const uint32_t i0 = __umul24(__float_as_uint(R.x), (uint32_t)DW_BL_STRIDE) + __float_as_uint(Q.x) - IDX_BIAS;
        const uint32_t i1 = __umul24(__float_as_uint(R.y), (uint32_t)DW_BL_STRIDE) + __float_as_uint(Q.y) - IDX_BIAS;
        const int t0 = bl[i0];
        const int t1 = bl[i1];
        bits[j] = __builtin_amdgcn_ballot_w64(t0 < t1);
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
    ORBX_TRACE_STAMP(4);
    ORBX_TRACE_FLUSH(g_desc_trace);
}

void orbx_launch_describe(const OrbxPlan &plan, const OrbxWork &wk, int nframes,
                          orbx_keypoint *d_kps, uint8_t *d_desc, int32_t *d_counts,
                          int32_t *d_status, hipStream_t s)
{
    const OrbxLevel &L0 = plan.lv[0];
    const int l0_aligned = (((uintptr_t)L0.base | (uintptr_t)L0.stride | (uintptr_t)L0.frame_stride) & 3) == 0;
    const int wg_per_frame = (plan.out_cap + DESC_WAVES - 1) / DESC_WAVES;
    const int nwg = wg_per_frame * nframes;
    hipLaunchKernelGGL(k_describe, dim3((nwg + 7) & ~7), dim3(64 * DESC_WAVES), 0, s, plan, wk, d_kps, d_desc, d_counts, d_status, l0_aligned,
                       wg_per_frame, nwg, wg_per_frame > 1 ? (uint32_t)((1ull << 32) / (unsigned)wg_per_frame + 1) : 0u);
}
