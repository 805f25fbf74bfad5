// Issue cost of the vector instructions of the block kernel's coarse step on gfx950, one and two waves per SIMD
// (diagnostic, not part of the library):  hipcc --offload-arch=gfx950 -O2 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define BODY_LOOP(ASM_BLOCK, N_PER_ITER)                                                   \
    for (int it = 0; it < iters; ++it) { ASM_BLOCK }                                        \
    n_instr = (long)iters * (N_PER_ITER);

template <int CASE>
__global__ __launch_bounds__(512) void k(float *out, int iters, long *n_out) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    f2 c = {0.999f, 1e-3f}, d = {1e-3f, -1e-3f};
    float cc = 0.999f, dd = 1e-3f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    unsigned msk = 0xffff0000u + (threadIdx.x >> 10), sel = 0x07060302u + (threadIdx.x >> 10);
    __shared__ unsigned lds_area[16384];
    unsigned ldsa = (unsigned)(size_t)(lds_area) + 4 * threadIdx.x;
    if (iters < 0) lds_area[threadIdx.x] = 1;
    long n_instr = 0;
    if constexpr (CASE == 0) {          // v_fma_f32, 8 independent chains
        BODY_LOOP(REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                       "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cc), "v"(dd));), 64)
    } else if constexpr (CASE == 1) {   // v_fma_f32, one dependent chain
        BODY_LOOP(REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                       "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2"
                       : "+v"(a0) : "v"(cc), "v"(dd));), 64)
    } else if constexpr (CASE == 2) {   // v_fma_f32, two interleaved chains
        BODY_LOOP(REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                       "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3"
                       : "+v"(a0), "+v"(a1) : "v"(cc), "v"(dd));), 64)
    } else if constexpr (CASE == 3) {   // v_pk_fma_f32, 8 independent
        BODY_LOOP(REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                       "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(d));), 64)
    } else if constexpr (CASE == 4) {   // v_pk_fma_f32, dependent
        BODY_LOOP(REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                       "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2"
                       : "+v"(p0) : "v"(c), "v"(d));), 64)
    } else if constexpr (CASE == 5) {   // v_pk_add_f32 independent
        BODY_LOOP(REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                       "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(d));), 64)
    } else if constexpr (CASE == 6) {   // v_cvt_pk_bf16_f32 independent
        BODY_LOOP(REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %6\n v_cvt_pk_bf16_f32 %2, %6, %7\n v_cvt_pk_bf16_f32 %3, %7, %4\n"
                       "v_cvt_pk_bf16_f32 %0, %5, %4\n v_cvt_pk_bf16_f32 %1, %6, %5\n v_cvt_pk_bf16_f32 %2, %7, %6\n v_cvt_pk_bf16_f32 %3, %4, %7"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));), 64)
    } else if constexpr (CASE == 7) {   // v_lshlrev / v_and independent
        BODY_LOOP(REP8(asm volatile("v_lshlrev_b32 %0, 16, %4\n v_and_b32 %1, 0xffff0000, %5\n v_lshlrev_b32 %2, 16, %6\n v_and_b32 %3, 0xffff0000, %7\n"
                       "v_lshlrev_b32 %0, 16, %5\n v_and_b32 %1, 0xffff0000, %6\n v_lshlrev_b32 %2, 16, %7\n v_and_b32 %3, 0xffff0000, %4"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));), 64)
    } else if constexpr (CASE == 8) {   // the step as the compiler emitted it: state v[100:101], serial (11 instructions)
        BODY_LOOP(REP8(asm volatile(
            "v_cvt_pk_bf16_f32 v102, v100, v101\n v_lshlrev_b32 v104, 16, v102\n v_and_b32 v105, 0xffff0000, v102\n"
            "v_pk_add_f32 v[104:105], v[100:101], v[104:105] neg_lo:[0,1] neg_hi:[0,1]\n v_cvt_pk_bf16_f32 v103, v104, v105\n"
            "v_mul_f32 v106, v100, %1\n v_fmac_f32 v100, %0, v100\n v_fmac_f32 v100, %2, v101\n v_fmac_f32 v106, %3, v101\n v_mov_b32 v101, v106"
            : : "v"(dd), "v"(cc), "v"(a4), "v"(a5) : "v100", "v101", "v102", "v103", "v104", "v105", "v106");), 80)
    } else if constexpr (CASE == 9) {   // the packed step, chains interleaved (8 instructions)
        BODY_LOOP(REP8(asm volatile(
            "v_cvt_pk_bf16_f32 v102, v100, v101\n v_pk_mul_f32 v[106:107], v[100:101], %0 op_sel_hi:[0,1]\n v_lshlrev_b32 v104, 16, v102\n v_and_b32 v105, 0xffff0000, v102\n"
            "v_pk_fma_f32 v[106:107], v[100:101], %1, v[106:107] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n v_pk_add_f32 v[104:105], v[100:101], v[104:105] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_pk_add_f32 v[100:101], v[100:101], v[106:107]\n v_cvt_pk_bf16_f32 v103, v104, v105"
            : : "v"(c), "v"(d) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");), 64)
    } else if constexpr (CASE == 12) {  // the scalar step, chains interleaved, fma instead of mov + fmac (9 instructions)
        BODY_LOOP(REP8(asm volatile(
            "v_cvt_pk_bf16_f32 v102, v100, v101\n v_mul_f32 v106, v100, %1\n v_fma_f32 v107, %0, v100, v100\n v_lshlrev_b32 v104, 16, v102\n v_and_b32 v105, 0xffff0000, v102\n"
            "v_fmac_f32 v106, %3, v101\n v_fmac_f32 v107, %2, v101\n v_pk_add_f32 v[104:105], v[100:101], v[104:105] neg_lo:[0,1] neg_hi:[0,1]\n"
            "v_mov_b32 v100, v107\n v_mov_b32 v101, v106\n v_cvt_pk_bf16_f32 v103, v104, v105"
            : : "v"(dd), "v"(cc), "v"(a4), "v"(a5) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");), 88)
    } else if constexpr (CASE == 13) {  // v_mov_b32 x8
        BODY_LOOP(REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n v_mov_b32 %0, %5\n v_mov_b32 %1, %6\n v_mov_b32 %2, %7\n v_mov_b32 %3, %4"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));), 64)
    } else if constexpr (CASE == 14) {  // v_and_b32 with the mask in a register
        BODY_LOOP(REP8(asm volatile("v_and_b32 %0, %8, %4\n v_and_b32 %1, %8, %5\n v_and_b32 %2, %8, %6\n v_and_b32 %3, %8, %7\n v_and_b32 %0, %8, %5\n v_and_b32 %1, %8, %6\n v_and_b32 %2, %8, %7\n v_and_b32 %3, %8, %4"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(msk));), 64)
    } else if constexpr (CASE == 15) {  // v_and_b32 with a 32-bit literal
        BODY_LOOP(REP8(asm volatile("v_and_b32 %0, 0xffff0000, %4\n v_and_b32 %1, 0xffff0000, %5\n v_and_b32 %2, 0xffff0000, %6\n v_and_b32 %3, 0xffff0000, %7\n v_and_b32 %0, 0xffff0000, %5\n v_and_b32 %1, 0xffff0000, %6\n v_and_b32 %2, 0xffff0000, %7\n v_and_b32 %3, 0xffff0000, %4"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));), 64)
    } else if constexpr (CASE == 16) {  // v_lshlrev_b32 by an inline constant
        BODY_LOOP(REP8(asm volatile("v_lshlrev_b32 %0, 16, %4\n v_lshlrev_b32 %1, 16, %5\n v_lshlrev_b32 %2, 16, %6\n v_lshlrev_b32 %3, 16, %7\n v_lshlrev_b32 %0, 16, %5\n v_lshlrev_b32 %1, 16, %6\n v_lshlrev_b32 %2, 16, %7\n v_lshlrev_b32 %3, 16, %4"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));), 64)
    } else if constexpr (CASE == 17) {  // v_perm_b32, selector in a register
        BODY_LOOP(REP8(asm volatile("v_perm_b32 %0, %4, %5, %8\n v_perm_b32 %1, %5, %6, %8\n v_perm_b32 %2, %6, %7, %8\n v_perm_b32 %3, %7, %4, %8\n v_perm_b32 %0, %5, %4, %8\n v_perm_b32 %1, %6, %5, %8\n v_perm_b32 %2, %7, %6, %8\n v_perm_b32 %3, %4, %7, %8"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(msk));), 64)
    } else if constexpr (CASE == 18) {  // v_sub_f32
        BODY_LOOP(REP8(asm volatile("v_sub_f32 %0, %4, %5\n v_sub_f32 %1, %5, %6\n v_sub_f32 %2, %6, %7\n v_sub_f32 %3, %7, %4\n v_sub_f32 %0, %5, %4\n v_sub_f32 %1, %6, %5\n v_sub_f32 %2, %7, %6\n v_sub_f32 %3, %4, %7"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));), 64)
    } else if constexpr (CASE == 19) {  // v_and_or_b32 (VOP3, three registers)
        BODY_LOOP(REP8(asm volatile("v_and_or_b32 %0, %4, %8, %5\n v_and_or_b32 %1, %5, %8, %6\n v_and_or_b32 %2, %6, %8, %7\n v_and_or_b32 %3, %7, %8, %4\n v_and_or_b32 %0, %5, %8, %4\n v_and_or_b32 %1, %6, %8, %5\n v_and_or_b32 %2, %7, %8, %6\n v_and_or_b32 %3, %4, %8, %7"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(msk));), 64)
    } else if constexpr (CASE == 20) {  // v_fma_f32 with a literal-free VOP3 but 3 distinct sources + v_mul_f32 VOP2 mix
        BODY_LOOP(REP8(asm volatile("v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %5, %6\n v_mul_f32 %2, %6, %7\n v_mul_f32 %3, %7, %4\n v_mul_f32 %0, %5, %4\n v_mul_f32 %1, %6, %5\n v_mul_f32 %2, %7, %6\n v_mul_f32 %3, %4, %7"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));), 64)
    } else if constexpr (CASE == 21) {  // step, variant A: mask hi (truncated), perm pack, scalar subs, cvt lo, scalar recurrence via fma (10 instructions, no pairs)
        BODY_LOOP(REP8(asm volatile(
            "v_and_b32 v104, %4, v100\n v_and_b32 v105, %4, v101\n v_mul_f32 v106, v100, %1\n v_fma_f32 v107, %0, v100, v100\n v_perm_b32 v102, v101, v100, %5\n"
            "v_sub_f32 v104, v100, v104\n v_sub_f32 v105, v101, v105\n v_fma_f32 v100, %2, v101, v107\n v_fma_f32 v101, %3, v101, v106\n v_cvt_pk_bf16_f32 v103, v104, v105"
            : : "v"(dd), "v"(cc), "v"(a4), "v"(a5), "v"(msk), "v"(sel) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");), 80)
    } else if constexpr (CASE == 22) {  // step, variant B: as A with the lo part packed by v_perm_b32 as well (truncated)
        BODY_LOOP(REP8(asm volatile(
            "v_and_b32 v104, %4, v100\n v_and_b32 v105, %4, v101\n v_mul_f32 v106, v100, %1\n v_fma_f32 v107, %0, v100, v100\n v_perm_b32 v102, v101, v100, %5\n"
            "v_sub_f32 v104, v100, v104\n v_sub_f32 v105, v101, v105\n v_fma_f32 v100, %2, v101, v107\n v_fma_f32 v101, %3, v101, v106\n v_perm_b32 v103, v105, v104, %5"
            : : "v"(dd), "v"(cc), "v"(a4), "v"(a5), "v"(msk), "v"(sel) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");), 80)
    } else if constexpr (CASE == 23) {  // variant A + the two LDS stores of the step (ds_write2st64_b32 hi / lo planes)
        BODY_LOOP(REP8(asm volatile(
            "v_and_b32 v104, %4, v100\n v_and_b32 v105, %4, v101\n v_mul_f32 v106, v100, %1\n v_fma_f32 v107, %0, v100, v100\n v_perm_b32 v102, v101, v100, %5\n"
            "v_sub_f32 v104, v100, v104\n v_sub_f32 v105, v101, v105\n v_fma_f32 v100, %2, v101, v107\n v_fma_f32 v101, %3, v101, v106\n v_cvt_pk_bf16_f32 v103, v104, v105\n"
            "ds_write2st64_b32 %6, v102, v103 offset0:1 offset1:19"
            : : "v"(dd), "v"(cc), "v"(a4), "v"(a5), "v"(msk), "v"(sel), "v"(ldsa) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");), 88)
    } else if constexpr (CASE == 24) {  // the current step + its LDS store
        BODY_LOOP(REP8(asm volatile(
            "v_cvt_pk_bf16_f32 v102, v100, v101\n v_lshlrev_b32 v104, 16, v102\n v_and_b32 v105, 0xffff0000, v102\n"
            "v_pk_add_f32 v[104:105], v[100:101], v[104:105] neg_lo:[0,1] neg_hi:[0,1]\n v_cvt_pk_bf16_f32 v103, v104, v105\n ds_write2st64_b32 %4, v102, v103 offset0:1 offset1:19\n"
            "v_mul_f32 v106, v100, %1\n v_fmac_f32 v100, %0, v100\n v_fmac_f32 v100, %2, v101\n v_fmac_f32 v106, %3, v101\n v_mov_b32 v101, v106"
            : : "v"(dd), "v"(cc), "v"(a4), "v"(a5), "v"(ldsa) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "memory");), 88)
    } else if constexpr (CASE == 25) {  // v_mov_b32_sdwa word insert (dst_unused:UNUSED_PRESERVE)
        BODY_LOOP(REP8(asm volatile("v_mov_b32_sdwa %0, %4 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_mov_b32_sdwa %1, %5 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
                       "v_mov_b32_sdwa %2, %6 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_mov_b32_sdwa %3, %7 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
                       "v_mov_b32_sdwa %0, %5 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_mov_b32_sdwa %1, %6 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
                       "v_mov_b32_sdwa %2, %7 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_mov_b32_sdwa %3, %4 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));), 64)
    } else if constexpr (CASE == 26) {  // v_add_u32 with a literal
        BODY_LOOP(REP8(asm volatile("v_add_u32 %0, 0x8000, %4\n v_add_u32 %1, 0x8000, %5\n v_add_u32 %2, 0x8000, %6\n v_add_u32 %3, 0x8000, %7\n v_add_u32 %0, 0x8000, %5\n v_add_u32 %1, 0x8000, %6\n v_add_u32 %2, 0x8000, %7\n v_add_u32 %3, 0x8000, %4"
                       : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));), 64)
    } else if constexpr (CASE == 27) {  // step C: integer split, SDWA packs (12 instructions, all full rate if SDWA is)
        BODY_LOOP(REP8(asm volatile(
            "v_add_u32 v104, 0x8000, v100\n v_add_u32 v105, 0x8000, v101\n v_and_b32 v104, 0xffff0000, v104\n v_and_b32 v105, 0xffff0000, v105\n"
            "v_sub_f32 v102, v100, v104\n v_sub_f32 v103, v101, v105\n v_mov_b32_sdwa v105, v104 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
            "v_mov_b32_sdwa v103, v102 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n"
            "v_mul_f32 v106, v100, %1\n v_fmac_f32 v100, %0, v100\n v_fmac_f32 v100, %2, v101\n v_fmac_f32 v106, %3, v101\n v_mov_b32 v101, v106"
            : : "v"(dd), "v"(cc), "v"(a4), "v"(a5) : "v100", "v101", "v102", "v103", "v104", "v105", "v106");), 104)
    } else if constexpr (CASE == 28) {  // step C': integer split, v_perm_b32 packs (what the int-split build emits)
        BODY_LOOP(REP8(asm volatile(
            "v_add_u32 v104, 0x8000, v100\n v_add_u32 v105, 0x8000, v101\n v_and_b32 v104, 0xffff0000, v104\n v_and_b32 v105, 0xffff0000, v105\n"
            "v_perm_b32 v107, v105, v104, %4\n v_sub_f32 v102, v100, v104\n v_sub_f32 v103, v101, v105\n v_perm_b32 v103, v103, v102, %4\n"
            "v_mul_f32 v106, v100, %1\n v_fmac_f32 v100, %0, v100\n v_fmac_f32 v100, %2, v101\n v_fmac_f32 v106, %3, v101\n v_mov_b32 v101, v106"
            : : "v"(dd), "v"(cc), "v"(a4), "v"(a5), "v"(sel) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");), 104)
    } else if constexpr (CASE == 10) {  // v_fmac_f32 (VOP2) 8 independent
        BODY_LOOP(REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                       "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cc), "v"(dd));), 64)
    } else if constexpr (CASE == 11) {  // v_fmac_f32 dependent chain
        BODY_LOOP(REP8(asm volatile("v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n"
                       "v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0"
                       : "+v"(a0) : "v"(dd));), 64)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.y + p3.x + p4.x + p5.x + p6.x + p7.x + u0 + u1 + u2 + u3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *n_out = n_instr;
}

template <int CASE>
static void run(const char *name, float *out, long *n_dev) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {          // waves per SIMD: blocks of 256 * wps threads, one block per CU (256 CUs)
        k<CASE><<<256, 256 * (wps > 2 ? 2 : wps), 0, 0>>>(out, 10, n_dev);
        hipDeviceSynchronize();
        const int blocks = wps > 2 ? 512 : 256;
        hipEventRecord(e0, 0);
        k<CASE><<<blocks, 256 * (wps > 2 ? 2 : wps), 0, 0>>>(out, iters, n_dev);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long n = 0;
        hipMemcpy(&n, n_dev, sizeof(long), hipMemcpyDeviceToHost);
        // ns per instruction per SIMD = time / (instructions per wave * waves per SIMD)
        printf("%-44s waves/SIMD %d: %8.3f ms  %6.3f ns per instr per wave, %6.3f ns per instr per SIMD\n", name, wps, ms, ms * 1e6 / n, ms * 1e6 / (n * wps));
    }
}

int main() {
    float *out;
    long *n_dev;
    hipMalloc(&out, 1024 * 1024 * sizeof(float));
    hipMalloc(&n_dev, sizeof(long));
    run<0>("v_fma_f32 x8 independent", out, n_dev);
    run<1>("v_fma_f32 dependent chain", out, n_dev);
    run<2>("v_fma_f32 two chains", out, n_dev);
    run<10>("v_fmac_f32 x8 independent", out, n_dev);
    run<11>("v_fmac_f32 dependent chain", out, n_dev);
    run<3>("v_pk_fma_f32 x8 independent", out, n_dev);
    run<4>("v_pk_fma_f32 dependent chain", out, n_dev);
    run<5>("v_pk_add_f32 x8 independent", out, n_dev);
    run<6>("v_cvt_pk_bf16_f32 x4 independent", out, n_dev);
    run<7>("v_lshlrev/v_and x4 independent", out, n_dev);
    run<8>("scalar step (10 instr, serial)", out, n_dev);
    run<9>("packed step (8 instr, interleaved)", out, n_dev);
    run<12>("scalar step (11 instr, interleaved)", out, n_dev);
    run<13>("v_mov_b32 x4 independent", out, n_dev);
    run<14>("v_and_b32 register mask", out, n_dev);
    run<15>("v_and_b32 literal mask", out, n_dev);
    run<16>("v_lshlrev_b32 inline 16", out, n_dev);
    run<17>("v_perm_b32", out, n_dev);
    run<18>("v_sub_f32", out, n_dev);
    run<19>("v_and_or_b32", out, n_dev);
    run<20>("v_mul_f32", out, n_dev);
    run<21>("step A (mask/perm/sub/cvt, 10 instr)", out, n_dev);
    run<22>("step B (mask/perm/sub/perm, 10 instr)", out, n_dev);
    run<23>("step A + ds_write2st64 (11 instr)", out, n_dev);
    run<24>("current step + ds_write2st64 (11 instr)", out, n_dev);
    run<25>("v_mov_b32_sdwa word insert", out, n_dev);
    run<26>("v_add_u32 literal", out, n_dev);
    run<27>("step C (int split, SDWA packs, 13 instr)", out, n_dev);
    run<28>("step C' (int split, perm packs, 13 instr)", out, n_dev);
    return 0;
}
