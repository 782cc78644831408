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

void orbx_launch_resize(const OrbxLevel &src, const OrbxLevel &dst, const ResizeTab &tab, int area2,
                        int nframes, hipStream_t s)
{
    dim3 block(64, 4);
    if (area2) {
        dim3 grid((dst.w + 63) / 64, (dst.h + 3) / 4, nframes);
        hipLaunchKernelGGL(k_resize_area2, grid, block, 0, s, src.base, src.stride, src.frame_stride,
                           dst.base, dst.w, dst.h, dst.stride, dst.frame_stride);
    } else {
        dim3 grid((dst.w + 255) / 256, (dst.h + 3) / 4, nframes);
        hipLaunchKernelGGL(k_resize_linear, grid, block, 0, s, src.base, src.w, src.h, src.stride,
                           src.frame_stride, dst.base, dst.w, dst.h, dst.stride, dst.frame_stride, tab);
    }
}

