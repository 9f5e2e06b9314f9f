// issue_calibration.hip — what do the SQ issue counters read when a pipe is PROVABLY full?  
// Microkernels of independent instructions of one kind, written in inline assembly so that nothing is packed, merged or
// removed, run at exactly 1, 2, 4 and 8 waves per SIMD (one workgroup of 256 x wps threads per CU — its LDS lets no second
// one in — so the waves of a SIMD are resident together; 8: two workgroups of 1024):
//   valu      8 independent v_fma_f32 accumulators, UNROLL x 8 instructions per loop turn
//   valu_dep  ONE accumulator: a dependent v_fma_f32 chain (issue-to-issue latency of a dependent VALU instruction)
//   trans     independent v_rcp_f32 (quarter-rate class)
//   salu      8 independent s_add_u32
//   mix       v_fma_f32 and s_add_u32 alternating 1 : 1 (does the scalar stream ride in the shadow of the vector stream?)
//   mix3      2 v_fma_f32 : 1 s_add_u32 : 0.3 ds_read_b32 — roughly the heavy kernels' instruction mix
//   v_*, ds_bpermute_b32: one instruction form each (packed single precision, compares into scalar pairs, DPP, integer
//             multiplies, conversions, transcendental, double precision): which forms take more than one issue slot
// Every wave stamps s_memtime around its loop (ground truth: shader cycles per instruction, independent of any counter);
// the same launches are then profiled with `rocprofv3 --pmc` (tools/profile_r04.sh), and tools/summarize_r04.py puts the
// two side by side in profiles/r04_issue_calibration.json.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_calibration.hip -o tools/bin/issue_calibration && tools/bin/issue_calibration
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int TURNS = 4096;  // loop turns per wave
constexpr int PER_TURN = 64; // instructions of the measured kind per loop turn (8 accumulators x 8)

#define FMA8 \
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define RCP8 \
    "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n" \
    "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
#define SADD8 \
    "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n" \
    "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"

struct Stamp { unsigned long long cycles; };

__device__ __forceinline__ void finish(float acc, unsigned long long t0, float* sink, Stamp* stamps) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) stamps[wave].cycles = t1 - t0;
    if (acc == 12345.678f) sink[0] = acc;  // keeps the accumulators alive
}

__global__ __launch_bounds__(1024) void k_valu(float* sink, Stamp* stamps, float m, float a) {
    extern __shared__ float lds[];
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < TURNS; ++t)
        asm volatile(FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(m), "v"(a));
    finish(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7, t0, sink, stamps);
}
__global__ __launch_bounds__(1024) void k_valu_dep(float* sink, Stamp* stamps, float m, float a) {
    extern __shared__ float lds[];
    float v0 = threadIdx.x;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < TURNS; ++t)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                     : "+v"(v0) : "v"(m), "v"(a));
    finish(v0, t0, sink, stamps);
}
__global__ __launch_bounds__(1024) void k_trans(float* sink, Stamp* stamps, float m, float a) {
    extern __shared__ float lds[];
    float v0 = threadIdx.x + 1.5f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < TURNS; ++t)
        asm volatile(RCP8 RCP8 RCP8 RCP8 RCP8 RCP8 RCP8 RCP8
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    finish(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7, t0, sink, stamps);
}
__global__ __launch_bounds__(1024) void k_salu(float* sink, Stamp* stamps, float m, float a) {
    extern __shared__ float lds[];
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < TURNS; ++t)
        asm volatile(SADD8 SADD8 SADD8 SADD8 SADD8 SADD8 SADD8 SADD8
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");
    finish((float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7), t0, sink, stamps);
}
// 32 v_fma_f32 and 32 s_add_u32 per turn, alternating
#define MIX2(a, b, sa, sb) "v_fma_f32 %" #a ", %" #a ", %12, %13\n s_add_u32 %" #sa ", %" #sa ", 1\n v_fma_f32 %" #b ", %" #b ", %12, %13\n s_add_u32 %" #sb ", %" #sb ", 1\n"
#define MIX8 MIX2(0, 1, 8, 9) MIX2(2, 3, 10, 11) MIX2(4, 5, 8, 9) MIX2(6, 7, 10, 11)
__global__ __launch_bounds__(1024) void k_mix(float* sink, Stamp* stamps, float m, float a) {
    extern __shared__ float lds[];
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < TURNS; ++t)
        asm volatile(MIX8 MIX8 MIX8 MIX8
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                     : "v"(m), "v"(a) : "scc");
    finish(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(s0 + s1 + s2 + s3), t0, sink, stamps);
}
// per turn: 32 v_fma_f32, 16 s_add_u32, 5 ds_read_b32 (conflict-free, results never waited for inside the turn)
#define M3A "v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n s_add_u32 %8, %8, 1\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n s_add_u32 %9, %9, 1\n"
#define M3B "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n s_add_u32 %10, %10, 1\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n s_add_u32 %11, %11, 1\n"
__global__ __launch_bounds__(1024) void k_mix3(float* sink, Stamp* stamps, float m, float a) {
    extern __shared__ float lds[];
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    lds[threadIdx.x] = v0;
    const unsigned addr = threadIdx.x * 4;
    float l0, l1, l2, l3, l4;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < TURNS; ++t) {
        asm volatile("ds_read_b32 %0, %5\n ds_read_b32 %1, %5 offset:1024\n ds_read_b32 %2, %5 offset:2048\n ds_read_b32 %3, %5 offset:3072\n ds_read_b32 %4, %5 offset:4096\n"
                     : "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3), "=v"(l4) : "v"(addr) : "memory");
        asm volatile(M3A M3B M3A M3B M3A M3B M3A M3B
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                     : "v"(m), "v"(a) : "scc");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        v0 += l0 + l1 + l2 + l3 + l4;
    }
    finish(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(s0 + s1 + s2 + s3), t0, sink, stamps);
}

// One more family: a single vector instruction form, eight independent copies per group, 64 per loop turn — which forms
// take more than one issue slot.  OPS is the asm text of 8 instructions over the 32-bit accumulators
// %0..%7, the scalar pair %8 (a lane mask the instruction may read or write) and the inputs %9, %10.
#define DEF_FORM(NAME, OPS, CLOBBER...)                                                                                        \
    __global__ __launch_bounds__(1024) void NAME(float* sink, Stamp* stamps, float m, float a) {                              \
        extern __shared__ float lds[];                                                                                         \
        float v0 = threadIdx.x + 1.5f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
        unsigned long long sm = 0x5555555555555555ull;                                                                         \
        __builtin_amdgcn_s_barrier();                                                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                            \
        for (int t = 0; t < TURNS; ++t)                                                                                        \
            asm volatile(OPS OPS OPS OPS OPS OPS OPS OPS                                                                       \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(sm) : "v"(m), "v"(a) : CLOBBER); \
        finish(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)sm, t0, sink, stamps);                                        \
    }
// ... and over 64-bit register pairs (packed single precision, double precision)
#define DEF_FORM64(NAME, OPS)                                                                                                  \
    __global__ __launch_bounds__(1024) void NAME(float* sink, Stamp* stamps, float m, float a) {                              \
        extern __shared__ float lds[];                                                                                         \
        double v0 = threadIdx.x + 1.5, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
        double mm = m;                                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                            \
        for (int t = 0; t < TURNS; ++t)                                                                                        \
            asm volatile(OPS OPS OPS OPS OPS OPS OPS OPS                                                                       \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(mm));         \
        finish((float)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7), t0, sink, stamps);                                          \
    }
#define E8(pre, post) pre "%0" post "\n" pre "%1" post "\n" pre "%2" post "\n" pre "%3" post "\n" pre "%4" post "\n" pre "%5" post "\n" pre "%6" post "\n" pre "%7" post "\n"
DEF_FORM(k_f_mul, "v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %9\n v_mul_f32 %3, %3, %9\n v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n v_mul_f32 %7, %7, %9\n", "vcc")
DEF_FORM(k_f_cmp64, E8("v_cmp_lt_f32_e64 %8, ", ", %9"), "vcc")
DEF_FORM(k_f_cmpvcc, E8("v_cmp_lt_f32_e32 vcc, ", ", %9"), "vcc")
DEF_FORM(k_f_cndmask, "v_cndmask_b32_e64 %0, %0, %9, %8\n v_cndmask_b32_e64 %1, %1, %9, %8\n v_cndmask_b32_e64 %2, %2, %9, %8\n v_cndmask_b32_e64 %3, %3, %9, %8\n v_cndmask_b32_e64 %4, %4, %9, %8\n v_cndmask_b32_e64 %5, %5, %9, %8\n v_cndmask_b32_e64 %6, %6, %9, %8\n v_cndmask_b32_e64 %7, %7, %9, %8\n", "vcc")
DEF_FORM(k_f_dpp, "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n", "vcc")
DEF_FORM(k_f_mullo, "v_mul_lo_u32 %0, %0, %9\n v_mul_lo_u32 %1, %1, %9\n v_mul_lo_u32 %2, %2, %9\n v_mul_lo_u32 %3, %3, %9\n v_mul_lo_u32 %4, %4, %9\n v_mul_lo_u32 %5, %5, %9\n v_mul_lo_u32 %6, %6, %9\n v_mul_lo_u32 %7, %7, %9\n", "vcc")
DEF_FORM(k_f_mul24, "v_mul_u32_u24 %0, %0, %9\n v_mul_u32_u24 %1, %1, %9\n v_mul_u32_u24 %2, %2, %9\n v_mul_u32_u24 %3, %3, %9\n v_mul_u32_u24 %4, %4, %9\n v_mul_u32_u24 %5, %5, %9\n v_mul_u32_u24 %6, %6, %9\n v_mul_u32_u24 %7, %7, %9\n", "vcc")
DEF_FORM(k_f_cvt, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7\n", "vcc")
DEF_FORM(k_f_sqrt, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n", "vcc")
DEF_FORM(k_f_readlane, "v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s23, %3, 5\n v_readlane_b32 s20, %4, 5\n v_readlane_b32 s21, %5, 5\n v_readlane_b32 s22, %6, 5\n v_readlane_b32 s23, %7, 5\n", "s20", "s21", "s22", "s23")
DEF_FORM(k_f_mov, "v_mov_b32 %0, %9\n v_mov_b32 %1, %9\n v_mov_b32 %2, %9\n v_mov_b32 %3, %9\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n", "vcc")
DEF_FORM(k_f_bperm, "ds_bpermute_b32 %0, %9, %0\n ds_bpermute_b32 %1, %9, %1\n ds_bpermute_b32 %2, %9, %2\n ds_bpermute_b32 %3, %9, %3\n ds_bpermute_b32 %4, %9, %4\n ds_bpermute_b32 %5, %9, %5\n ds_bpermute_b32 %6, %9, %6\n ds_bpermute_b32 %7, %9, %7\n s_waitcnt lgkmcnt(0)\n", "vcc")
#define V2(op) op " %0, %0, %9\n" op " %1, %1, %9\n" op " %2, %2, %9\n" op " %3, %3, %9\n" op " %4, %4, %9\n" op " %5, %5, %9\n" op " %6, %6, %9\n" op " %7, %7, %9\n"
#define V3(op) op " %0, %0, %9, %10\n" op " %1, %1, %9, %10\n" op " %2, %2, %9, %10\n" op " %3, %3, %9, %10\n" op " %4, %4, %9, %10\n" op " %5, %5, %9, %10\n" op " %6, %6, %9, %10\n" op " %7, %7, %9, %10\n"
DEF_FORM(k_f_addf, V2("v_add_f32"), "vcc")
DEF_FORM(k_f_maxf, V2("v_max_f32"), "vcc")
DEF_FORM(k_f_med3, V3("v_med3_f32"), "vcc")
DEF_FORM(k_f_max3, V3("v_max3_f32"), "vcc")
DEF_FORM(k_f_addu, V2("v_add_u32"), "vcc")
DEF_FORM(k_f_and, V2("v_and_b32"), "vcc")
DEF_FORM(k_f_lshl, V2("v_lshlrev_b32"), "vcc")
DEF_FORM(k_f_lshladd, V3("v_lshl_add_u32"), "vcc")
DEF_FORM(k_f_mad24, V3("v_mad_u32_u24"), "vcc")
DEF_FORM(k_f_bfe, V3("v_bfe_u32"), "vcc")
DEF_FORM(k_f_cndvcc, V2("v_cndmask_b32_e32") , "vcc")
DEF_FORM(k_f_fmac, V2("v_fmac_f32"), "vcc")
DEF_FORM(k_f_mbcnt, V2("v_mbcnt_lo_u32_b32"), "vcc")
DEF_FORM(k_f_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n", "vcc")
DEF_FORM64(k_f_pkfma, "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n")
DEF_FORM64(k_f_pkmul, "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
DEF_FORM64(k_f_fma64, "v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n")
DEF_FORM64(k_f_mul64, "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n")
DEF_FORM64(k_f_add64, "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
typedef void (*Kern)(float*, Stamp*, float, float);
struct Variant { const char* name; Kern k; double valu, salu, lds; const char* fn; };  // instructions per loop turn per wave (the loop's own s_add / s_cmp / s_cbranch: +3 scalar)

int main(int argc, char** argv) {
    const Variant variants[] = {{"valu", k_valu, PER_TURN, 0, 0, "k_valu"}, {"valu_dep", k_valu_dep, 16, 0, 0, "k_valu_dep"}, {"trans", k_trans, PER_TURN, 0, 0, "k_trans"},
                                {"salu", k_salu, 0, PER_TURN, 0, "k_salu"}, {"mix", k_mix, 32, 32, 0, "k_mix"}, {"mix3", k_mix3, 32, 16, 5, "k_mix3"},
                                {"v_mul_f32", k_f_mul, 64, 0, 0, "k_f_mul"}, {"v_pk_fma_f32", k_f_pkfma, 64, 0, 0, "k_f_pkfma"}, {"v_pk_mul_f32", k_f_pkmul, 64, 0, 0, "k_f_pkmul"},
                                {"v_cmp_e64_sgpr", k_f_cmp64, 64, 0, 0, "k_f_cmp64"}, {"v_cmp_e32_vcc", k_f_cmpvcc, 64, 0, 0, "k_f_cmpvcc"}, {"v_cndmask_e64", k_f_cndmask, 64, 0, 0, "k_f_cndmask"},
                                {"v_max_i32_dpp", k_f_dpp, 64, 0, 0, "k_f_dpp"}, {"v_mul_lo_u32", k_f_mullo, 64, 0, 0, "k_f_mullo"}, {"v_mul_u32_u24", k_f_mul24, 64, 0, 0, "k_f_mul24"},
                                {"v_cvt", k_f_cvt, 64, 0, 0, "k_f_cvt"}, {"v_sqrt_f32", k_f_sqrt, 64, 0, 0, "k_f_sqrt"}, {"v_readlane_b32", k_f_readlane, 64, 0, 0, "k_f_readlane"},
                                {"v_mov_b32", k_f_mov, 64, 0, 0, "k_f_mov"}, {"v_add_f32", k_f_addf, 64, 0, 0, "k_f_addf"}, {"v_max_f32", k_f_maxf, 64, 0, 0, "k_f_maxf"},
                                {"v_med3_f32", k_f_med3, 64, 0, 0, "k_f_med3"}, {"v_max3_f32", k_f_max3, 64, 0, 0, "k_f_max3"}, {"v_add_u32", k_f_addu, 64, 0, 0, "k_f_addu"},
                                {"v_and_b32", k_f_and, 64, 0, 0, "k_f_and"}, {"v_lshlrev_b32", k_f_lshl, 64, 0, 0, "k_f_lshl"}, {"v_lshl_add_u32", k_f_lshladd, 64, 0, 0, "k_f_lshladd"},
                                {"v_mad_u32_u24", k_f_mad24, 64, 0, 0, "k_f_mad24"}, {"v_bfe_u32", k_f_bfe, 64, 0, 0, "k_f_bfe"}, {"v_cndmask_e32_vcc", k_f_cndvcc, 64, 0, 0, "k_f_cndvcc"},
                                {"v_fmac_f32", k_f_fmac, 64, 0, 0, "k_f_fmac"}, {"v_mbcnt_lo", k_f_mbcnt, 64, 0, 0, "k_f_mbcnt"}, {"ds_bpermute_b32", k_f_bperm, 0, 0, 64, "k_f_bperm"}, {"v_fma_f64", k_f_fma64, 64, 0, 0, "k_f_fma64"},
                                {"v_mul_f64", k_f_mul64, 64, 0, 0, "k_f_mul64"}, {"v_add_f64", k_f_add64, 64, 0, 0, "k_f_add64"}};
    const char* only = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* sink;
    Stamp* stamps;
    const int max_waves = cus * 8 * 4;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&stamps, sizeof(Stamp) * max_waves));
    std::vector<Stamp> host(max_waves);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"turns\": %d, \"runs\": [\n", prop.name, cus, prop.clockRate / 1000, TURNS);
    bool first = true;
    for (const Variant& v : variants) {
        if (only && strcmp(only, v.name)) continue;
        for (int wps : {1, 2, 4, 8}) {
            // exactly `wps` waves on every SIMD, all resident together: ONE workgroup of 256 x wps threads per CU (its 120 KB of
            // LDS let no second one in); 8 per SIMD: two workgroups of 1024 threads (72 KB each).  (The first version launched
            // wps workgroups of 256 threads per CU and trusted the dispatcher to spread them evenly: it did not.)
            const int per_wg = wps > 4 ? 4 : wps;
            const size_t lds = wps > 4 ? 72 * 1024 : 120 * 1024;
            CHECK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * (wps > 4 ? 2 : 1), block = 256 * per_wg;
            float ms = 0.f;
            for (int rep = 0; rep < 3; ++rep) {  // (the last of three launches is reported: clocks up, code object loaded)
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(v.k, dim3(grid), dim3(block), lds, 0, sink, stamps, 1.0001f, 0.5f);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const int waves = grid * block / 64;
            CHECK(hipMemcpy(host.data(), stamps, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> c(waves);
            for (int w = 0; w < waves; ++w) c[w] = host[w].cycles;
            std::sort(c.begin(), c.end());
            const double med = (double)c[waves / 2], total = v.valu + v.salu + v.lds;
            // (s_memtime ticks are shader cycles per MI355X_MICROARCH.md; the event time of the whole launch is printed beside
            // them so that the tick length can be checked: ticks / launch_us = MHz)
            printf("%s{\"kind\": \"%s\", \"kernel\": \"%s\", \"waves_per_simd\": %d, \"waves\": %d, \"valu_per_wave\": %.0f, \"salu_per_wave\": %.0f, \"lds_per_wave\": %.0f, "
                   "\"memtime_ticks_median\": %.0f, \"memtime_ticks_min\": %llu, \"memtime_ticks_max\": %llu, \"ticks_per_instruction_per_wave\": %.4f, \"launch_us\": %.2f}",
                   first ? "" : ",\n", v.name, v.fn, wps, waves, v.valu * TURNS, (v.salu + 3) * TURNS, v.lds * TURNS, med, c.front(), c.back(), med / (total * TURNS), ms * 1e3);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
