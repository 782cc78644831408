// orbx_stereo.hip -- Frame::ComputeStereoMatches (src/Frame.cc:466-640 of WChen09/My-SLAM) on gfx950
// (SURVEY.md 8(f) row N3).  One wave per left keypoint:
//   1. candidates = right keypoints whose row band [floor(y-r), ceil(y+r)], r = 2*scale[octave], holds the
//      left keypoint's row, octave within +-1, u in [uL - maxD, uL]; the reference visits them in ascending
//      right index (its row table is filled in that order), so a dense scan with key = dist << 22 | iR and
//      a strict-'<' minimum below TH_HIGH gives the same best match (no row table needed);
//   2. if best < (TH_HIGH+TH_LOW)/2: 11 SAD windows of 11x11 on the two extractors' pyramid levels
//      (centre-subtracted, exact integers), parabola fit in fp32 (no contraction), disparity/depth.
// The final median filter (:627-639) is a sort of <= N ints and runs on the host.
#include <algorithm>
#include <cmath>
#include <vector>

#include "orbx_internal.h"

struct StereoLevels {
    const uint8_t *L[ORBX_MAX_LEVELS], *R[ORBX_MAX_LEVELS];
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS], strideL[ORBX_MAX_LEVELS], strideR[ORBX_MAX_LEVELS];
    float scale[ORBX_MAX_LEVELS], inv_scale[ORBX_MAX_LEVELS];
    int nlevels;
};

__device__ __forceinline__ int refl(int p, int len)
{
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ int ham32(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ __launch_bounds__(256) void k_stereo(StereoLevels lv, const orbx_keypoint *__restrict__ kl, const uint8_t *__restrict__ dl,
                                                int N, const orbx_keypoint *__restrict__ kr, const uint8_t *__restrict__ dr, int Nr,
                                                int nRows, float mb, float mbf, float *__restrict__ uRight,
                                                float *__restrict__ depth, int32_t *__restrict__ sad)
{
    __shared__ int part[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int iL = blockIdx.x * 4 + wave;
    if (iL >= N) return;
    const orbx_keypoint kpL = kl[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const float maxD = __fdiv_rn(mbf, mb);                    // :497-499 (minZ = mb, minD = 0)
    const float minU = __fsub_rn(uL, maxD), maxU = uL;
    float outU = -1.0f, outD = -1.0f;
    int outS = -1;
    const int row = (int)vL;
    uint32_t bk = (100u << 22);                                // bestDist = TH_HIGH; only dist < 100 can win
    if (!(maxU < 0) && row >= 0 && row < nRows) {
        const uint4 *QL = reinterpret_cast<const uint4 *>(dl) + 2 * (long long)iL;
        const uint4 q0 = QL[0], q1 = QL[1];
        for (int iR = lane; iR < Nr; iR += 64) {
            const orbx_keypoint kpR = kr[iR];
            const float r = __fmul_rn(2.0f, lv.scale[kpR.octave]);
            const int maxr = (int)ceilf(__fadd_rn(kpR.y, r)), minr = (int)floorf(__fsub_rn(kpR.y, r));
            if (row < minr || row > maxr) continue;
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            if (!(kpR.x >= minU && kpR.x <= maxU)) continue;
            const uint4 *TR = reinterpret_cast<const uint4 *>(dr) + 2 * (long long)iR;
            const uint32_t key = ((uint32_t)ham32(q0, q1, TR[0], TR[1]) << 22) | (uint32_t)iR;
            bk = min(bk, key);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bk = min(bk, (uint32_t)__shfl_xor((int)bk, o));
    const int bestDist = (int)(bk >> 22);
    if (bestDist < (100 + 50) / 2) {                           // thOrbDist :471,:549
        const int bestIdxR = (int)(bk & 0x3FFFFFu);
        const float uR0 = kr[bestIdxR].x;
        const float sf = lv.inv_scale[levelL];
        const float scaleduL = roundf(__fmul_rn(kpL.x, sf));
        const float scaledvL = roundf(__fmul_rn(kpL.y, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int W = lv.w[levelL], H = lv.h[levelL];
        const float iniu = scaleduR0, endu = __fadd_rn(scaleduR0, 11.0f);   // +L-w, +L+w+1 with L = w = 5
        if (!(iniu < 0 || endu >= (float)W)) {
            const uint8_t *IL = lv.L[levelL], *IR = lv.R[levelL];
            const int sL = lv.strideL[levelL], sR = lv.strideR[levelL];
            const int cvL = (int)scaledvL, cuL = (int)scaleduL, cuR = (int)scaleduR0;
            const int cL = IL[(long long)refl(cvL, H) * sL + refl(cuL, W)];
            // 121 tasks: (incR index a, patch row b) -> sum over the 11 columns
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int t = lane + 64 * k;
                if (t < 121) {
                    const int a = t / 11, b = t - a * 11;
                    const int cR = IR[(long long)refl(cvL, H) * sR + refl(cuR + a - 5, W)];
                    const uint8_t *rl = IL + (long long)refl(cvL + b - 5, H) * sL;
                    const uint8_t *rr = IR + (long long)refl(cvL + b - 5, H) * sR;
                    int s = 0;
#pragma unroll
                    for (int dx = -5; dx <= 5; dx++)
                        s += abs(((int)rl[refl(cuL + dx, W)] - cL) - ((int)rr[refl(cuR + a - 5 + dx, W)] - cR));
                    part[wave][t] = s;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int dsum = 0;
            if (lane < 11)
                for (int b = 0; b < 11; b++) dsum += part[wave][lane * 11 + b];
            // every lane learns the 11 distances; first minimum wins (strict '<' from INT_MAX, :583-595)
            int bestD = 0x7FFFFFFF, bestinc = 0;
            float vd[11];
#pragma unroll
            for (int a = 0; a < 11; a++) {
                const int d = __shfl(dsum, a);
                vd[a] = (float)d;
                if (d < bestD) { bestD = d; bestinc = a - 5; }
            }
            if (!(bestinc == -5 || bestinc == 5)) {
                float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
                for (int a = 1; a < 10; a++)
                    if (a == bestinc + 5) { dist1 = vd[a - 1]; dist2 = vd[a]; dist3 = vd[a + 1]; }
                const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3),
                                               __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(lv.scale[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= 0.f && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = (float)0.01;
                            bestuR = (float)((double)uL - 0.01);
                        }
                        outD = __fdiv_rn(mbf, disparity);
                        outU = bestuR;
                        outS = bestD;
                    }
                }
            }
        }
    }
    if (lane == 0) { uRight[iL] = outU; depth[iL] = outD; sad[iL] = outS; }
}

#include <mutex>
struct StereoScratch { void *p = nullptr; size_t bytes = 0; std::mutex mu; };
static StereoScratch g_scr[16];   // per device; a call holds the device's scratch lock (stereo pairs are matched one at a time per device)

// host-side description of a handle's pyramid (orbx_capi.hip)
int orbx_internal_levels(orbx_extractor *h, const uint8_t **base, int *w, int *hh, int *stride, float *scale, float *inv_scale, int *nlevels, int *device);

extern "C" int orbx_stereo_matches(orbx_extractor *left, orbx_extractor *right,
                                   const orbx_keypoint *kl, const uint8_t *dl, int nl,
                                   const orbx_keypoint *kr, const uint8_t *dr, int nr,
                                   float mb, float mbf, float *u_right, float *depth)
{
    if (!left || !right || nl < 0 || nr < 0) return ORBX_E_INVALID;
    if (nl == 0) return ORBX_OK;
    if (!kl || !dl || !u_right || !depth || (nr > 0 && (!kr || !dr))) return ORBX_E_INVALID;
    StereoLevels lv;
    int devL = 0, devR = 0, nlv = 0, nlvR = 0;
    int wR[ORBX_MAX_LEVELS], hR[ORBX_MAX_LEVELS];
    float scR[ORBX_MAX_LEVELS], iscR[ORBX_MAX_LEVELS];
    if (orbx_internal_levels(left, lv.L, lv.w, lv.h, lv.strideL, lv.scale, lv.inv_scale, &nlv, &devL) != ORBX_OK) return ORBX_E_INVALID;
    if (orbx_internal_levels(right, lv.R, wR, hR, lv.strideR, scR, iscR, &nlvR, &devR) != ORBX_OK) return ORBX_E_INVALID;
    if (devL != devR || nlv != nlvR) return ORBX_E_INVALID;
    for (int l = 0; l < nlv; l++)
        if (wR[l] != lv.w[l] || hR[l] != lv.h[l]) return ORBX_E_SHAPE;     // rectified pair: same size
    lv.nlevels = nlv;
    if (hipSetDevice(devL) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return ORBX_E_HIP;
    const size_t bk = (size_t)nl * sizeof(orbx_keypoint), bd = (size_t)nl * 32, brk = (size_t)nr * sizeof(orbx_keypoint), brd = (size_t)nr * 32;
    const size_t bo = (size_t)nl * 4;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t need = al(bk) + al(bd) + al(brk) + al(brd) + 3 * al(bo) + 256;
    StereoScratch &sc = g_scr[devL & 15];
    std::lock_guard<std::mutex> lock(sc.mu);
    if (need > sc.bytes) {
        (void)hipFree(sc.p); sc.p = nullptr; sc.bytes = 0;
        if (hipMalloc(&sc.p, need) != hipSuccess) return ORBX_E_HIP;
        sc.bytes = need;
    }
    uint8_t *p = (uint8_t *)sc.p;
    orbx_keypoint *d_kl = (orbx_keypoint *)p; p += al(bk);
    uint8_t *d_dl = p; p += al(bd);
    orbx_keypoint *d_kr = (orbx_keypoint *)p; p += al(brk);
    uint8_t *d_dr = p; p += al(brd);
    float *d_u = (float *)p; p += al(bo);
    float *d_d = (float *)p; p += al(bo);
    int32_t *d_s = (int32_t *)p;
    if (hipMemcpy(d_kl, kl, bk, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_dl, dl, bd, hipMemcpyHostToDevice) != hipSuccess) return ORBX_E_HIP;
    if (nr > 0 && (hipMemcpy(d_kr, kr, brk, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_dr, dr, brd, hipMemcpyHostToDevice) != hipSuccess)) return ORBX_E_HIP;
    hipLaunchKernelGGL(k_stereo, dim3((nl + 3) / 4), dim3(256), 0, 0, lv, d_kl, d_dl, nl, d_kr, d_dr, nr, lv.h[0], mb, mbf, d_u, d_d, d_s);
    if (hipGetLastError() != hipSuccess) return ORBX_E_HIP;
    std::vector<int32_t> sad(nl);
    if (hipMemcpy(u_right, d_u, bo, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(depth, d_d, bo, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(sad.data(), d_s, bo, hipMemcpyDeviceToHost) != hipSuccess) return ORBX_E_HIP;
    // :627-639 sort by (SAD, index), median, cull >= 1.5*1.4*median
    std::vector<std::pair<int, int>> v;
    for (int i = 0; i < nl; i++) if (sad[i] >= 0) v.push_back(std::make_pair(sad[i], i));
    if (!v.empty()) {
        std::sort(v.begin(), v.end());
        const float median = (float)v[v.size() / 2].first;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = (int)v.size() - 1; i >= 0; i--) {
            if ((float)v[i].first < thDist) break;
            u_right[v[i].second] = -1;
            depth[v[i].second] = -1;
        }
    }
    return ORBX_OK;
}
