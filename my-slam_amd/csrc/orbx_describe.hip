// orbx_describe.hip -- IC_Angle + GaussianBlur 7x7 + rBRIEF (src/ORBextractor.cc:79-149,1087-1103) on gfx950.
// All image arithmetic is integer; the fp32 in fastAtan2 and in the sample rotation is written with
// explicit __f*_rn intrinsics so no FMA contraction can happen; the file is also built
// with -ffp-contract=off.
#include "orbx_internal.h"
#include "orb_pattern_data.h"

__constant__ int c_umax[16];
__constant__ int c_gauss[7];
__constant__ signed char c_pattern[1024];

int orbx_upload_constants(const int umax[16], const int gauss_k[7])
{
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax, sizeof(int) * 16) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), gauss_k, sizeof(int) * 7) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), ORBX_PATTERN, 1024) != hipSuccess) return -1;
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
// Canonical semantics = correctly rounded cosf/sinf; tools/verify_sincos.py checks this
// routine against the x87 long-double libm over every fp32 input of the domain.
__device__ __forceinline__ void sincos_cr(float theta, float *cs, float *sn)
{
    const double x = (double)theta;
    const double kd = rint(x * 0.6366197723675814);
    const int k = (int)kd;
    double r = fma(-kd, 1.5707963267341256, x);          // exact: 33-bit head times k <= 4
    r = fma(-kd, 6.077100506506192e-11, r);
    const double z = r * r;
    double ps = -8.22063524662433e-18;
    ps = fma(ps, z, 2.8114572543455206e-15);
    ps = fma(ps, z, -7.647163731819816e-13);
    ps = fma(ps, z, 1.6059043836821613e-10);
    ps = fma(ps, z, -2.505210838544172e-08);
    ps = fma(ps, z, 2.7557319223985893e-06);
    ps = fma(ps, z, -0.0001984126984126984);
    ps = fma(ps, z, 0.008333333333333333);
    ps = fma(ps, z, -0.16666666666666666);
    const double s = fma(r * z, ps, r);
    double pc = 4.110317623312165e-19;
    pc = fma(pc, z, -1.5619206968586225e-16);
    pc = fma(pc, z, 4.779477332387385e-14);
    pc = fma(pc, z, -1.1470745597729725e-11);
    pc = fma(pc, z, 2.08767569878681e-09);
    pc = fma(pc, z, -2.755731922398589e-07);
    pc = fma(pc, z, 2.48015873015873e-05);
    pc = fma(pc, z, -0.001388888888888889);
    pc = fma(pc, z, 0.041666666666666664);
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

// exported for the exhaustive sincos check (tools/verify_sincos.py)
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

__global__ __launch_bounds__(DESC_THREADS) void k_describe(
    OrbxPlan plan, OrbxWork wk, orbx_keypoint *__restrict__ kps, uint8_t *__restrict__ desc,
    int32_t *__restrict__ counts, int32_t *__restrict__ status)
{
    __shared__ uint8_t raw[DESC_RAW * 44];
    __shared__ uint16_t rb[DESC_RAW * 38];
    __shared__ uint8_t bl[DESC_BL * 40];
    const int g = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;

    int total = 0, l = -1, idx = 0;
    for (int i = 0; i < plan.nlevels; i++) {
        const int c = (int)wk.nk[f * plan.nlevels + i];
        if (l < 0 && g < total + c) { l = i; idx = g - total; }
        total += c;
    }
    if (g == 0 && lane == 0) {
        const uint32_t e = wk.errflags[f];
        counts[f] = min(total, plan.out_cap);
        status[f] = (e & ERRF_CAND_OVERFLOW) ? ORBX_E_CAND_OVERFLOW
                  : (e & ERRF_TREE_OVERFLOW) ? ORBX_E_TREE_OVERFLOW
                  : (total > plan.out_cap)   ? ORBX_E_CAPACITY : ORBX_OK;
    }
    if (l < 0 || g >= plan.out_cap) return;
    const OrbxLevel &L = plan.lv[l];
    const OrbxCand kc = wk.sel[(long long)f * plan.list_frame + L.list_off + idx];
    const int x = (int)(kc.xy & 0xFFFFu), y = (int)(kc.xy >> 16);
    const uint8_t *img = L.base + (long long)f * L.frame_stride;

    for (int i = lane; i < DESC_RAW * DESC_RAW; i += DESC_THREADS) {
        const int r = i / DESC_RAW, c = i - r * DESC_RAW;
        const int gy = reflect101(y - 21 + r, L.h), gx = reflect101(x - 21 + c, L.w);
        raw[r * 44 + c] = img[(long long)gy * L.stride + gx];
    }
    __syncthreads();

    // IC_Angle (:79-106): m10 = sum u*I, m01 = sum v*I over |u| <= umax[|v|]
    int m10 = 0, m01 = 0;
    for (int i = lane; i < 31 * 31; i += DESC_THREADS) {
        const int vr = i / 31, v = vr - 15, u = i - vr * 31 - 15;
        if (abs(u) <= c_umax[abs(v)]) {
            const int I = raw[(21 + v) * 44 + 21 + u];
            m10 += u * I;
            m01 += v * I;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // 7x7 Gaussian, row pass (exact, <= 65535)
    const int k0 = c_gauss[0], k1 = c_gauss[1], k2 = c_gauss[2], k3 = c_gauss[3];
    for (int i = lane; i < DESC_RAW * DESC_BL; i += DESC_THREADS) {
        const int r = i / DESC_BL, c = i - r * DESC_BL;
        const uint8_t *p = &raw[r * 44 + c];
        rb[r * 38 + c] = (uint16_t)(k0 * (p[0] + p[6]) + k1 * (p[1] + p[5]) + k2 * (p[2] + p[4]) + k3 * p[3]);
    }
    __syncthreads();
    // column pass: (sum + 32768) >> 16 saturated (OpenCV C path) or round-half-even (OpenCV SSE2 path)
    const int simd_cols = plan.blur_mode == 1 ? (L.w & ~3) : 0;
    for (int i = lane; i < DESC_BL * DESC_BL; i += DESC_THREADS) {
        const int r = i / DESC_BL, c = i - r * DESC_BL;
        const uint16_t *p = &rb[r * 38 + c];
        const int s = k0 * (p[0] + p[6 * 38]) + k1 * (p[38] + p[5 * 38]) + k2 * (p[2 * 38] + p[4 * 38]) + k3 * p[3 * 38];
        int v;
        if (x - DESC_R + c < simd_cols) {
            v = s >> 16;
            const int rem = s & 0xFFFF;
            if (rem > 0x8000 || (rem == 0x8000 && (v & 1))) v++;
        } else {
            v = (s + 32768) >> 16;
        }
        bl[r * 40 + c] = (uint8_t)min(v, 255);
    }
    __syncthreads();

    // rBRIEF (:110-149)
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float a, b;
    sincos_cr(__fmul_rn(angle, factorPI), &a, &b);
    unsigned long long bits[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int pair = j * 64 + lane;
        const signed char *pp = &c_pattern[pair * 4];
        const float x0 = (float)pp[0], y0 = (float)pp[1], x1 = (float)pp[2], y1 = (float)pp[3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int q0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int q1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bl[(DESC_R + r0) * 40 + DESC_R + q0];
        const int t1 = bl[(DESC_R + r1) * 40 + DESC_R + q1];
        bits[j] = __ballot(t0 < t1);
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
}

void orbx_launch_describe(const OrbxPlan &plan, const OrbxWork &wk, int nframes,
                          orbx_keypoint *d_kps, uint8_t *d_desc, int32_t *d_counts,
                          int32_t *d_status, hipStream_t s)
{
    dim3 grid(plan.out_cap, nframes);
    hipLaunchKernelGGL(k_describe, grid, dim3(DESC_THREADS), 0, s, plan, wk, d_kps, d_desc, d_counts, d_status);
}
