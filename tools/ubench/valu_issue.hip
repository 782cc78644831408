// valu_issue.hip -- issue-rate microbenchmark for the vector instructions the ORB kernels are made of (gfx950).
//
// For every instruction: a dependency-free stream (8 destination registers written round-robin from two or three
// never-written source registers), 128 instructions per loop iteration, run with 1, 2, 4 and 8 waves per SIMD
// (k workgroups of 256 threads per CU, one wave per SIMD each, on every CU).  Reported per instruction and
// occupancy: SIMD clocks per wave64 instruction = (shader clocks the slowest wave of a CU was inside the loop)
// / (instructions per wave * waves per SIMD), from s_memtime; and the same from the launch's wall time at the
// clock s_memtime / s_memrealtime gives.  Output: one JSON object on stdout (profiles/r02_valu_issue.json).
//
// build: hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct Stamp { unsigned long long clk0, clk1, rt0, rt1; };

// 8 independent instructions; OP2: "op dst, a, b"; OP3: "op dst, a, b, c"
#define R8_2(OP) asm volatile(OP " %0, %8, %9\n" OP " %1, %9, %8\n" OP " %2, %8, %9\n" OP " %3, %9, %8\n" \
                              OP " %4, %8, %9\n" OP " %5, %9, %8\n" OP " %6, %8, %9\n" OP " %7, %9, %8\n" \
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b))
#define R8_3(OP) asm volatile(OP " %0, %8, %9, %10\n" OP " %1, %9, %8, %10\n" OP " %2, %8, %9, %10\n" OP " %3, %9, %8, %10\n" \
                              OP " %4, %8, %9, %10\n" OP " %5, %9, %8, %10\n" OP " %6, %8, %9, %10\n" OP " %7, %9, %8, %10\n" \
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "v"(c))
// compare into vcc (VOPC) followed by nothing: 8 compares
#define R8_C(OP) asm volatile(OP " vcc, %0, %1\n" OP " vcc, %1, %0\n" OP " vcc, %0, %1\n" OP " vcc, %1, %0\n" \
                              OP " vcc, %0, %1\n" OP " vcc, %1, %0\n" OP " vcc, %0, %1\n" OP " vcc, %1, %0\n" : : "v"(a), "v"(b) : "vcc")
// v_cndmask with vcc as the selector
#define R8_M(OP) asm volatile(OP " %0, %8, %9, vcc\n" OP " %1, %9, %8, vcc\n" OP " %2, %8, %9, vcc\n" OP " %3, %9, %8, vcc\n" \
                              OP " %4, %8, %9, vcc\n" OP " %5, %9, %8, vcc\n" OP " %6, %8, %9, vcc\n" OP " %7, %9, %8, vcc\n" \
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc")
// DPP move (row_shr:1) and wave-wide shift
#define R8_D(MOD) asm volatile("v_mov_b32_dpp %0, %8 " MOD "\nv_mov_b32_dpp %1, %9 " MOD "\nv_mov_b32_dpp %2, %8 " MOD "\nv_mov_b32_dpp %3, %9 " MOD "\n" \
                               "v_mov_b32_dpp %4, %8 " MOD "\nv_mov_b32_dpp %5, %9 " MOD "\nv_mov_b32_dpp %6, %8 " MOD "\nv_mov_b32_dpp %7, %9 " MOD "\n" \
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b))
// 64-bit packed fp32
#define R8_P(OP) asm volatile(OP " %0, %8, %9\n" OP " %1, %9, %8\n" OP " %2, %8, %9\n" OP " %3, %9, %8\n" \
                              OP " %4, %8, %9\n" OP " %5, %9, %8\n" OP " %6, %8, %9\n" OP " %7, %9, %8\n" \
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(qa), "v"(qb))
#define X16(S) S; S; S; S; S; S; S; S; S; S; S; S; S; S; S; S

typedef float f2 __attribute__((ext_vector_type(2)));

#define KERNEL(NAME, BODY)                                                                                       \
    __global__ __launch_bounds__(256) void NAME(Stamp *st, uint32_t *sink, int iters, uint32_t seed)            \
    {                                                                                                            \
        uint32_t a = seed + threadIdx.x, b = seed * 3u + 7u * threadIdx.x, c = 0x05040100u;                      \
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;                                 \
        f2 qa = {1.0f + threadIdx.x, 2.0f}, qb = {0.5f, 0.25f};                                                  \
        f2 q0 = qa, q1 = qa, q2 = qa, q3 = qa, q4 = qa, q5 = qa, q6 = qa, q7 = qa;                               \
        const unsigned long long msk = 0x5555aaaa3333ccccull ^ seed; (void)msk;                                     \
        (void)c; (void)qb;                                                                                       \
        __syncthreads();                                                                                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();       \
        for (int i = 0; i < iters; i++) { X16(BODY); }                                                           \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();       \
        if ((threadIdx.x & 63) == 0) {                                                                           \
            Stamp s = {t0, t1, w0, w1};                                                                          \
            st[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;                                                         \
        }                                                                                                        \
        sink[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^                           \
            __float_as_uint(q0.x + q1.x + q2.x + q3.x + q4.y + q5.y + q6.y + q7.y);                              \
    }

KERNEL(k_add_u32, R8_2("v_add_u32"))
KERNEL(k_xor_b32, R8_2("v_xor_b32"))
KERNEL(k_pk_min_u16, R8_2("v_pk_min_u16"))
KERNEL(k_pk_max_u16, R8_2("v_pk_max_u16"))
KERNEL(k_pk_sub_u16, R8_2("v_pk_sub_u16"))
KERNEL(k_pk_add_f16, R8_2("v_pk_add_f16"))
KERNEL(k_pk_min_f16, R8_2("v_pk_min_f16"))
KERNEL(k_pk_minimum3_f16, R8_3("v_pk_minimum3_f16"))
KERNEL(k_pk_maximum3_f16, R8_3("v_pk_maximum3_f16"))
KERNEL(k_perm_b32, R8_3("v_perm_b32"))
KERNEL(k_alignbyte_b32, R8_3("v_alignbyte_b32"))
KERNEL(k_lshl_or_b32, R8_3("v_lshl_or_b32"))
KERNEL(k_max3_u32, R8_3("v_max3_u32"))
KERNEL(k_med3_u32, R8_3("v_med3_u32"))
KERNEL(k_min_u32, R8_2("v_min_u32"))
KERNEL(k_dot4_u32_u8, R8_3("v_dot4_u32_u8"))
KERNEL(k_dot2_u32_u16, R8_3("v_dot2_u32_u16"))
KERNEL(k_mul_u32_u24, R8_2("v_mul_u32_u24"))
KERNEL(k_mad_u32_u24, R8_3("v_mad_u32_u24"))
KERNEL(k_mul_lo_u32, R8_2("v_mul_lo_u32"))
KERNEL(k_bcnt_u32_b32, R8_2("v_bcnt_u32_b32"))
KERNEL(k_mbcnt_lo, R8_2("v_mbcnt_lo_u32_b32"))
KERNEL(k_cmp_lt_u32, R8_C("v_cmp_lt_u32"))
KERNEL(k_cndmask_b32, R8_M("v_cndmask_b32"))
KERNEL(k_mov_dpp_row_shr1, R8_D("row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_mov_dpp_wave_shr1, R8_D("wave_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(k_pk_mul_f32, R8_P("v_pk_mul_f32"))
KERNEL(k_pk_add_f32, R8_P("v_pk_add_f32"))
KERNEL(k_add_f32, R8_2("v_add_f32"))
KERNEL(k_mul_f32, R8_2("v_mul_f32"))
KERNEL(k_and_b32, R8_2("v_and_b32"))
KERNEL(k_or_b32, R8_2("v_or_b32"))
KERNEL(k_sub_u32, R8_2("v_sub_u32"))
KERNEL(k_lshlrev_b32, R8_2("v_lshlrev_b32"))
KERNEL(k_lshrrev_b32, R8_2("v_lshrrev_b32"))
KERNEL(k_max_u32, R8_2("v_max_u32"))
KERNEL(k_max_i32, R8_2("v_max_i32"))
KERNEL(k_min_f32, R8_2("v_min_f32"))
KERNEL(k_max_f32, R8_2("v_max_f32"))
KERNEL(k_max3_f32, R8_3("v_max3_f32"))
KERNEL(k_fma_f32, R8_3("v_fma_f32"))
KERNEL(k_add3_u32, R8_3("v_add3_u32"))
KERNEL(k_lshl_add_u32, R8_3("v_lshl_add_u32"))
KERNEL(k_and_or_b32, R8_3("v_and_or_b32"))
KERNEL(k_or3_b32, R8_3("v_or3_b32"))
KERNEL(k_bfe_u32, R8_3("v_bfe_u32"))
KERNEL(k_bfi_b32, R8_3("v_bfi_b32"))
KERNEL(k_sad_u8, R8_3("v_sad_u8"))
KERNEL(k_sad_u16, R8_3("v_sad_u16"))
KERNEL(k_mad_i32_i24, R8_3("v_mad_i32_i24"))
KERNEL(k_mul_hi_u32, R8_2("v_mul_hi_u32"))
KERNEL(k_pk_add_u16, R8_2("v_pk_add_u16"))
KERNEL(k_pk_mul_lo_u16, R8_2("v_pk_mul_lo_u16"))
KERNEL(k_pk_fma_f16, R8_3("v_pk_fma_f16"))
KERNEL(k_add_f16, R8_2("v_add_f16"))
KERNEL(k_max_f16, R8_2("v_max_f16"))
KERNEL(k_min_u16, R8_2("v_min_u16"))
KERNEL(k_cvt_f32_ubyte0, asm volatile("v_cvt_f32_ubyte0 %0, %8\nv_cvt_f32_ubyte1 %1, %9\nv_cvt_f32_ubyte2 %2, %8\nv_cvt_f32_ubyte3 %3, %9\nv_cvt_f32_ubyte0 %4, %8\nv_cvt_f32_ubyte1 %5, %9\nv_cvt_f32_ubyte2 %6, %8\nv_cvt_f32_ubyte3 %7, %9\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)))
KERNEL(k_mov_b32, asm volatile("v_mov_b32 %0, %8\nv_mov_b32 %1, %9\nv_mov_b32 %2, %8\nv_mov_b32 %3, %9\nv_mov_b32 %4, %8\nv_mov_b32 %5, %9\nv_mov_b32 %6, %8\nv_mov_b32 %7, %9\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)))
// v_cndmask with an SGPR-pair selector written once outside the loop (the vcc form above re-reads an unwritten vcc)
#define R8_MS(OP) asm volatile(OP " %0, %8, %9, %10\n" OP " %1, %9, %8, %10\n" OP " %2, %8, %9, %10\n" OP " %3, %9, %8, %10\n" \
                               OP " %4, %8, %9, %10\n" OP " %5, %9, %8, %10\n" OP " %6, %8, %9, %10\n" OP " %7, %9, %8, %10\n" \
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(msk))
KERNEL(k_cndmask_sgpr, R8_MS("v_cndmask_b32"))
KERNEL(k_addc_co_u32, asm volatile("v_addc_co_u32 %0, vcc, %8, %9, %10\nv_addc_co_u32 %1, vcc, %9, %8, %10\nv_addc_co_u32 %2, vcc, %8, %9, %10\nv_addc_co_u32 %3, vcc, %9, %8, %10\nv_addc_co_u32 %4, vcc, %8, %9, %10\nv_addc_co_u32 %5, vcc, %9, %8, %10\nv_addc_co_u32 %6, vcc, %8, %9, %10\nv_addc_co_u32 %7, vcc, %9, %8, %10\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(msk) : "vcc"))
// the MFMA matcher's selection step: integer max + median on the accumulator
#define MIX_SELECT R8_2("v_max_i32"); R8_3("v_med3_i32")
KERNEL(k_mix_select, MIX_SELECT; MIX_SELECT; MIX_SELECT; MIX_SELECT; MIX_SELECT; MIX_SELECT; MIX_SELECT; MIX_SELECT)
// the mix stage 2 of k_fast_cells is made of: 2 x (16 + 16 + 8) three-input fp16 min/max per 16 packed subtractions and
// 17 byte merges (v_lshl_or_b32)
#define MIX_FAST2 R8_3("v_pk_minimum3_f16"); R8_3("v_pk_maximum3_f16"); R8_2("v_pk_add_f16"); R8_3("v_lshl_or_b32")
KERNEL(k_mix_fast_stage2, MIX_FAST2; MIX_FAST2; MIX_FAST2; MIX_FAST2)
// the matcher's inner loop: xor + popcount-accumulate
#define MIX_HAMMING R8_2("v_xor_b32"); R8_2("v_bcnt_u32_b32")
KERNEL(k_mix_hamming, MIX_HAMMING; MIX_HAMMING; MIX_HAMMING; MIX_HAMMING; MIX_HAMMING; MIX_HAMMING; MIX_HAMMING; MIX_HAMMING)

typedef void (*kern_t)(Stamp *, uint32_t *, int, uint32_t);
struct Entry { const char *name; kern_t fn; int per_iter; };

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    Stamp *d_st; uint32_t *d_sink;
    const int maxblocks = ncu * 8;
    CK(hipMalloc((void **)&d_st, sizeof(Stamp) * maxblocks * 4));
    CK(hipMalloc((void **)&d_sink, sizeof(uint32_t) * maxblocks * 256));
    std::vector<Stamp> h(maxblocks * 4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define E(K) {&#K[2], K, 128}
    const Entry tab[] = {E(k_add_u32), E(k_xor_b32), E(k_min_u32), E(k_pk_min_u16), E(k_pk_max_u16), E(k_pk_sub_u16), E(k_pk_add_f16), E(k_pk_min_f16),
                         E(k_pk_minimum3_f16), E(k_pk_maximum3_f16), E(k_perm_b32), E(k_alignbyte_b32), E(k_lshl_or_b32), E(k_max3_u32), E(k_med3_u32),
                         E(k_dot4_u32_u8), E(k_dot2_u32_u16), E(k_mul_u32_u24), E(k_mad_u32_u24), E(k_mul_lo_u32), E(k_bcnt_u32_b32), E(k_mbcnt_lo),
                         E(k_cmp_lt_u32), E(k_cndmask_b32), E(k_mov_dpp_row_shr1), E(k_mov_dpp_wave_shr1), E(k_pk_mul_f32), E(k_pk_add_f32),
                         E(k_add_f32), E(k_mul_f32),
                         E(k_and_b32), E(k_or_b32), E(k_sub_u32), E(k_lshlrev_b32), E(k_lshrrev_b32), E(k_max_u32), E(k_max_i32), E(k_min_f32), E(k_max_f32),
                         E(k_max3_f32), E(k_fma_f32), E(k_add3_u32), E(k_lshl_add_u32), E(k_and_or_b32), E(k_or3_b32), E(k_bfe_u32), E(k_bfi_b32),
                         E(k_sad_u8), E(k_sad_u16), E(k_mad_i32_i24), E(k_mul_hi_u32), E(k_pk_add_u16), E(k_pk_mul_lo_u16), E(k_pk_fma_f16), E(k_add_f16),
                         E(k_max_f16), E(k_min_u16), E(k_cvt_f32_ubyte0), E(k_mov_b32), E(k_cndmask_sgpr), E(k_addc_co_u32),
                         {"mix_select", k_mix_select, 16 * 8 * 16},
                         {"mix_fast_stage2", k_mix_fast_stage2, 16 * 8 * 16}, {"mix_hamming", k_mix_hamming, 16 * 8 * 16}};
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"note\": \"clk = SIMD clocks per wave64 instruction = loop clocks of a wave / (instructions per wave * waves per SIMD); wall_clk = the same from the launch's wall time\", \"results\": {\n",
           prop.gcnArchName, ncu, iters);
    bool first = true;
    for (const Entry &en : tab) {
        printf("%s  \"%s\": {", first ? "" : ",\n", en.name);
        first = false;
        const int occs[4] = {1, 2, 4, 8};
        for (int oi = 0; oi < 4; oi++) {
            const int k = occs[oi], blocks = ncu * k;
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, iters / 10 + 1, 1u);   // warm-up
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, d_st, d_sink, iters, 1u);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * blocks * 4, hipMemcpyDeviceToHost));
            std::vector<double> clk, ghz;
            for (int i = 0; i < blocks * 4; i++) {
                clk.push_back((double)(h[i].clk1 - h[i].clk0));
                const double rt = (double)(h[i].rt1 - h[i].rt0);   // 100 MHz ticks
                if (rt > 0) ghz.push_back((double)(h[i].clk1 - h[i].clk0) / rt * 0.1);
            }
            std::sort(clk.begin(), clk.end()); std::sort(ghz.begin(), ghz.end());
            const double ninstr = (double)iters * en.per_iter;
            const double med = clk[clk.size() / 2], mx = clk.back();
            const double g = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
            // wall: every SIMD issued k waves * ninstr instructions in ms
            const double wall_clk = g > 0 ? (ms * 1e-3 * g * 1e9) / (ninstr * k) : 0.0;
            printf("%s\"w%d\": {\"clk\": %.3f, \"clk_max\": %.3f, \"wall_clk\": %.3f, \"ghz\": %.3f, \"ms\": %.4f}", oi ? ", " : "", k,
                   med / (ninstr * k), mx / (ninstr * k), wall_clk, g, ms);
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n}}\n");
    return 0;
}
