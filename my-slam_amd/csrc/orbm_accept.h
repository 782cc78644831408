// orbm_accept.h -- the matcher's second half, shared by k_accept_rot (orbm.hip) and the fused tail of k_best2_mfma (orbm_mfma.hip):
// merge of the train-range partials, acceptance (src/ORBmatcher.cc:228-232), rotation histogram (:236-246), ComputeThreeMaxima and
// the cull (:266-284).
#ifndef ORBM_ACCEPT_H
#define ORBM_ACCEPT_H
#include "orbm_internal.h"

#define M_KEY_NONE ((256u << 22) | 0x3FFFFFu)
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }

// merge the train-range partials of one query: the two smallest keys of the union
__device__ __forceinline__ void merge_partial_keys(const uint2 *__restrict__ part, int S, long long stride_z, long long o,
                                                   uint32_t &bk, uint32_t &sk)
{
    bk = M_KEY_NONE; sk = M_KEY_NONE;
    for (int z0 = 0; z0 < S; z0 += 8) {   // eight partials per round trip (a load per iteration would be one memory latency each)
        uint2 p[8];
#pragma unroll
        for (int u = 0; u < 8; u++) p[u] = z0 + u < S ? part[(long long)(z0 + u) * stride_z + o] : make_uint2(M_KEY_NONE, M_KEY_NONE);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            sk = med3u(bk, sk, p[u].x); bk = min(bk, p[u].x);
            sk = med3u(bk, sk, p[u].y); bk = min(bk, p[u].y);
        }
    }
}
__device__ __forceinline__ void merge_partials(const uint2 *__restrict__ part, int S, long long stride_z, long long o,
                                               int &bi, int &bd, int &sd)
{
    uint32_t bk, sk;
    merge_partial_keys(part, S, stride_z, o, bk, sk);
    bd = (int)(bk >> 22);
    sd = (int)(sk >> 22);
    bi = bd < 256 ? (int)(bk & 0x3FFFFFu) : -1;
}

// ---- acceptance (:228-232) + rotation histogram (:236-246) + ComputeThreeMaxima + cull (:266-284) ----
__device__ __forceinline__ int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / 30;
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == 30) bin = 0;
    return bin;
}

// One workgroup of NT threads per frame pair; every thread keeps its (up to ACC_PER_THREAD) queries in registers across the histogram
// barrier, so the body is two dependent global round trips (partials -> matched keypoint angle) instead of a chain per loop
// iteration.  ComputeThreeMaxima runs on the lanes of one wave: the reference's scan (strict '>' in index order) ranks the bins by
// (count descending, index ascending) and ignores empty ones, which is three wave-wide maxima of count << 5 | (31 - index).
#define ACC_THREADS 1024
#define ACC_PER_THREAD 4
struct AcceptShared { int hist[32]; int ind[3]; int count; };

__device__ __forceinline__ int acc_wave_max(int v)       // maximum over the 64 lanes (DPP steps as in the extractor's wave sums), every lane gets it
{
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));     // quad_perm:[1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));     // quad_perm:[2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));    // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));    // row_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));   // row_bcast:15
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));   // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}
// wave 0 of the workgroup: sh.ind[0..2] from sh.hist[0..29]  (all values are counts >= 0; update_dpp's 0 for absent lanes is neutral)
__device__ __forceinline__ void accept_three_maxima_wave(AcceptShared &sh, int lane)
{
    const int c = lane < 30 ? sh.hist[lane] : 0;
    int key = c > 0 ? ((c << 5) | (31 - lane)) : 0;
    int idx[3], cnt[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int m = acc_wave_max(key);
        idx[k] = m > 0 ? 31 - (m & 31) : -1;
        cnt[k] = m >> 5;
        if (key == m) key = 0;
    }
    if ((float)cnt[1] < 0.1f * (float)cnt[0]) { idx[1] = -1; idx[2] = -1; }
    else if ((float)cnt[2] < 0.1f * (float)cnt[0]) idx[2] = -1;
    if (lane == 0) { sh.ind[0] = idx[0]; sh.ind[1] = idx[1]; sh.ind[2] = idx[2]; }
}

template <int NT>
__device__ __forceinline__ void accept_rot_body(AcceptShared &sh, int b, int nbatch, int tid,
    const int32_t *__restrict__ nqv, const orbx_keypoint *__restrict__ kq, const orbx_keypoint *__restrict__ kt,
    int cap, const uint2 *__restrict__ part, int S, int th, float nnratio, int check_ori,
    int32_t *__restrict__ match12, int32_t *__restrict__ nmatches,
    int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d, int32_t *__restrict__ second_d)
{
    const int nq = nqv[b];
    const long long base = (long long)b * cap;
    const long long stride_z = (long long)nbatch * cap;
    if (tid < 32) sh.hist[tid] = 0;
    if (tid == 0) sh.count = 0;
    __syncthreads();
    int cnt = 0;
    int i0 = 0;
    do {            // the first sweep's loads go by the CAPACITY of the arrays, not by the count that is still on its way (rows beyond it are masked)
        int mm[ACC_PER_THREAD], bins[ACC_PER_THREAD];
        float aq[ACC_PER_THREAD];
#pragma unroll
        for (int u = 0; u < ACC_PER_THREAD; u++) {
            const int i = i0 + u * NT + tid;
            mm[u] = -1; bins[u] = -1; aq[u] = 0.f;
            if (i < cap) {
                int bi, bd, sd;
                merge_partials(part, S, stride_z, base + i, bi, bd, sd);
                if (check_ori) aq[u] = kq[base + i].angle;
                if (i < nq) {
                    if (best_idx) { best_idx[base + i] = bi; best_d[base + i] = bd; second_d[base + i] = sd; }
                    if (bd <= th && (float)bd < __fmul_rn(nnratio, (float)sd)) mm[u] = bi;     // :228-232
                }
            }
        }
#pragma unroll
        for (int u = 0; u < ACC_PER_THREAD; u++)
            if (mm[u] >= 0 && check_ori) {
                bins[u] = rot_bin(aq[u], kt[base + mm[u]].angle);
                atomicAdd(&sh.hist[bins[u]], 1);
            }
        if (nq > NT * ACC_PER_THREAD) {
            // more queries than one sweep holds in registers: park (match, bin) in match12 and redo below
#pragma unroll
            for (int u = 0; u < ACC_PER_THREAD; u++) {
                const int i = i0 + u * NT + tid;
                if (i < nq) match12[base + i] = mm[u] >= 0 ? (mm[u] | (max(bins[u], 0) << 24)) : -1;
            }
        } else {
            __syncthreads();
            if (tid < 64) accept_three_maxima_wave(sh, tid);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < ACC_PER_THREAD; u++) {
                const int i = i0 + u * NT + tid;
                if (i < nq) {
                    int m = mm[u];
                    if (m >= 0 && check_ori && bins[u] != sh.ind[0] && bins[u] != sh.ind[1] && bins[u] != sh.ind[2]) m = -1;
                    match12[base + i] = m;
                    cnt += m >= 0;
                }
            }
        }
        i0 += NT * ACC_PER_THREAD;
    } while (i0 < nq);
    if (nq > NT * ACC_PER_THREAD) {   // large-frame path (train index < 2^22: cap <= 0x3FFFFF is checked on the host)
        __syncthreads();
        if (tid < 64) accept_three_maxima_wave(sh, tid);
        __syncthreads();
        for (int i = tid; i < nq; i += NT) {
            const int pk = match12[base + i];
            int m = -1;
            if (pk >= 0) {
                m = pk & 0xFFFFFF;
                const int bin = pk >> 24;
                if (check_ori && bin != sh.ind[0] && bin != sh.ind[1] && bin != sh.ind[2]) m = -1;
            }
            match12[base + i] = m;
            cnt += m >= 0;
        }
    }
    for (int i = nq + tid; i < cap; i += NT) match12[base + i] = -1;
    if (cnt) atomicAdd(&sh.count, cnt);
    __syncthreads();
    if (tid == 0) nmatches[b] = sh.count;
}

#endif
