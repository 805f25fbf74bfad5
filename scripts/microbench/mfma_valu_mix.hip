// Does an f32-input MFMA leave the vector ALU free?  (diagnostic, not part of the library)
//     hipcc --offload-arch=gfx950 -O2 mfma_valu_mix.hip -o mfma_valu_mix && ./mfma_valu_mix
// One loop body = 2 MFMAs on two accumulators, each followed by K independent v_fma_f32; one or two waves per SIMD.
// Reports shader cycles per (MFMA + K fillers) and SIMD: the span of the workgroup (first start to last end, s_memtime), median over workgroups.
// If the f32 MFMA shares the vector ALU's datapath, cycles grow by ~2..4 per filler from K = 0 on (additive);
// if it runs beside it (as the bf16 MFMA does), fillers hide until the issue slots of the gap are used up.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define F1(i) "v_fma_f32 %[x" #i "], %[x" #i "], %[cc], %[dd]\n"
#define FILL0 ""
#define FILL1 F1(0)
#define FILL2 F1(0) F1(1)
#define FILL3 F1(0) F1(1) F1(2)
#define FILL4 F1(0) F1(1) F1(2) F1(3)
#define FILL6 F1(0) F1(1) F1(2) F1(3) F1(4) F1(5)
#define FILL8 F1(0) F1(1) F1(2) F1(3) F1(4) F1(5) F1(6) F1(7)
#define FILL12 FILL8 FILL4
#define FILL16 FILL8 FILL8

#define OPERANDS                                                                                                        \
    : [c0] "+v"(c0), [c1] "+v"(c1), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), \
      [x6] "+v"(x6), [x7] "+v"(x7)                                                                                       \
    : [a] "v"(a), [b] "v"(b), [ah] "v"(ah), [bh] "v"(bh), [cc] "v"(cc), [dd] "v"(dd)

#define BODY_F32(FILL)                                                                      \
    asm volatile("v_mfma_f32_16x16x4_f32 %[c0], %[a], %[b], %[c0]\n" FILL                  \
                 "v_mfma_f32_16x16x4_f32 %[c1], %[a], %[b], %[c1]\n" FILL OPERANDS);
#define BODY_BF16(FILL)                                                                     \
    asm volatile("v_mfma_f32_16x16x32_bf16 %[c0], %[ah], %[bh], %[c0]\n" FILL              \
                 "v_mfma_f32_16x16x32_bf16 %[c1], %[ah], %[bh], %[c1]\n" FILL OPERANDS);
#define BODY_NONE(FILL) asm volatile(FILL FILL OPERANDS);
// K ds_read_b32 fillers per MFMA (the block kernel's operand reads): does the LDS pipe disturb back-to-back MFMA issue?
#define L1(i) "ds_read_b32 %[x" #i "], %[la] offset:" #i "*256\n"
#define LFILL1 L1(0)
#define LFILL2 L1(0) L1(1)
#define LFILL4 L1(0) L1(1) L1(2) L1(3)
#define BODY_LDS(FILL)                                                                      \
    asm volatile("v_mfma_f32_16x16x4_f32 %[c0], %[a], %[b], %[c0]\n" FILL                  \
                 "v_mfma_f32_16x16x4_f32 %[c1], %[a], %[b], %[c1]\n" FILL "s_waitcnt lgkmcnt(0)\n"                                    \
                 : [c0] "+v"(c0), [c1] "+v"(c1), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3)        \
                 : [a] "v"(a), [b] "v"(b), [la] "v"(la));

template <int KIND, int K>
__global__ __launch_bounds__(512) void kern(float *out, unsigned long long *cyc, int iters) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    float a = 1e-3f * tid, b = 1.f - 1e-3f * tid, cc = 0.999f, dd = 1e-3f;
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(a + i); bh[i] = (__bf16)(b - i); }
    f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    const unsigned la = (unsigned)(size_t)lds + 4 * tid;
    lds[tid] = 1.f; lds[tid + 512] = 2.f; lds[tid + 1024] = 3.f; lds[tid + 1536] = 4.f;
    if (iters < 0) lds[tid] = 1;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define REP4(X) X X X X
        if constexpr (KIND == 0) {
            if constexpr (K == 0) { REP4(BODY_F32(FILL0)) } else if constexpr (K == 1) { REP4(BODY_F32(FILL1)) }
            else if constexpr (K == 2) { REP4(BODY_F32(FILL2)) } else if constexpr (K == 4) { REP4(BODY_F32(FILL4)) }
            else if constexpr (K == 6) { REP4(BODY_F32(FILL6)) } else if constexpr (K == 8) { REP4(BODY_F32(FILL8)) }
            else if constexpr (K == 12) { REP4(BODY_F32(FILL12)) } else { REP4(BODY_F32(FILL16)) }
        } else if constexpr (KIND == 1) {
            if constexpr (K == 0) { REP4(BODY_BF16(FILL0)) } else if constexpr (K == 1) { REP4(BODY_BF16(FILL1)) }
            else if constexpr (K == 2) { REP4(BODY_BF16(FILL2)) } else if constexpr (K == 4) { REP4(BODY_BF16(FILL4)) }
            else if constexpr (K == 6) { REP4(BODY_BF16(FILL6)) } else if constexpr (K == 8) { REP4(BODY_BF16(FILL8)) }
            else if constexpr (K == 12) { REP4(BODY_BF16(FILL12)) } else { REP4(BODY_BF16(FILL16)) }
        } else if constexpr (KIND == 3) {
            if constexpr (K == 1) { REP4(BODY_LDS(LFILL1)) } else if constexpr (K == 2) { REP4(BODY_LDS(LFILL2)) } else { REP4(BODY_LDS(LFILL4)) }
        } else if constexpr (KIND == 4) {
            // roles: waves 0..3 of the workgroup issue MFMAs only, waves 4..7 the K fillers per group only (two waves per SIMD)
            if (tid < 256) { REP4(BODY_F32(FILL0)) }
            else if constexpr (K == 2) { REP4(BODY_NONE(FILL2)) } else if constexpr (K == 4) { REP4(BODY_NONE(FILL4)) }
            else if constexpr (K == 8) { REP4(BODY_NONE(FILL8)) } else { REP4(BODY_NONE(FILL16)) }
        } else {
            if constexpr (K == 1) { REP4(BODY_NONE(FILL1)) } else if constexpr (K == 2) { REP4(BODY_NONE(FILL2)) }
            else if constexpr (K == 4) { REP4(BODY_NONE(FILL4)) } else if constexpr (K == 6) { REP4(BODY_NONE(FILL6)) }
            else if constexpr (K == 8) { REP4(BODY_NONE(FILL8)) } else if constexpr (K == 12) { REP4(BODY_NONE(FILL12)) }
            else { REP4(BODY_NONE(FILL16)) }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // every wave reports (start, end): the arbiter serves the older wave of a SIMD first, so the span of the WORKGROUP is the measure
    if ((tid & 63) == 0) { cyc[(blockIdx.x * 8 + (tid >> 6)) * 2] = t0; cyc[(blockIdx.x * 8 + (tid >> 6)) * 2 + 1] = t1; }
    out[blockIdx.x * blockDim.x + tid] = c0.x + c1.y + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND, int K>
void run(const char *name, int waves_per_simd, float *d_out, unsigned long long *d_cyc, int n_cu) {
    const int iters = 2000;
    const int threads = 256 * waves_per_simd;
    const size_t lds = 100 * 1024;                      // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern<KIND, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kern<KIND, K>), dim3(n_cu), dim3(threads), lds, 0, d_out, d_cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(n_cu * 16);
    hipMemcpy(h.data(), d_cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::vector<double> span, own;
    const int nw = 4 * waves_per_simd;
    for (int b = 0; b < n_cu; ++b) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < nw; ++w) {
            lo = std::min(lo, h[(b * 8 + w) * 2]);
            hi = std::max(hi, h[(b * 8 + w) * 2 + 1]);
            own.push_back((double)(h[(b * 8 + w) * 2 + 1] - h[(b * 8 + w) * 2]));
        }
        span.push_back((double)(hi - lo));
    }
    std::sort(span.begin(), span.end());
    std::sort(own.begin(), own.end());
    // 8 (MFMA + K fillers) groups per iteration and wave; a SIMD runs waves_per_simd waves
    const double per_simd = span[span.size() / 2] / (iters * 8.0 * waves_per_simd);
    std::printf("%-26s K=%2d fillers  %d wave(s)/SIMD : %7.2f cycles per (MFMA + K VALU) group per SIMD   (a wave's own loop: median %.1f, min %.1f, max %.1f per group)\n",
                name, K, waves_per_simd, per_simd, own[own.size() / 2] / (iters * 8.0), own.front() / (iters * 8.0), own.back() / (iters * 8.0));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    float *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_out, (size_t)n_cu * 512 * 4);
    hipMalloc(&d_cyc, (size_t)n_cu * 16 * 8);
    std::printf("device %s, %d CUs; one loop group = 1 MFMA + K independent v_fma_f32 (wave 0..3 of each workgroup timed by s_memtime)\n", prop.name, n_cu);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("v_mfma_f32_16x16x4_f32", w, d_out, d_cyc, n_cu);
        run<0, 1>("v_mfma_f32_16x16x4_f32", w, d_out, d_cyc, n_cu);
        run<0, 2>("v_mfma_f32_16x16x4_f32", w, d_out, d_cyc, n_cu);
        run<0, 4>("v_mfma_f32_16x16x4_f32", w, d_out, d_cyc, n_cu);
        run<0, 6>("v_mfma_f32_16x16x4_f32", w, d_out, d_cyc, n_cu);
        run<0, 8>("v_mfma_f32_16x16x4_f32", w, d_out, d_cyc, n_cu);
        run<0, 12>("v_mfma_f32_16x16x4_f32", w, d_out, d_cyc, n_cu);
        run<0, 16>("v_mfma_f32_16x16x4_f32", w, d_out, d_cyc, n_cu);
        run<1, 0>("v_mfma_f32_16x16x32_bf16", w, d_out, d_cyc, n_cu);
        run<1, 1>("v_mfma_f32_16x16x32_bf16", w, d_out, d_cyc, n_cu);
        run<1, 2>("v_mfma_f32_16x16x32_bf16", w, d_out, d_cyc, n_cu);
        run<1, 4>("v_mfma_f32_16x16x32_bf16", w, d_out, d_cyc, n_cu);
        run<1, 6>("v_mfma_f32_16x16x32_bf16", w, d_out, d_cyc, n_cu);
        run<1, 8>("v_mfma_f32_16x16x32_bf16", w, d_out, d_cyc, n_cu);
        run<3, 1>("f32 MFMA + K ds_read_b32", w, d_out, d_cyc, n_cu);
        run<3, 2>("f32 MFMA + K ds_read_b32", w, d_out, d_cyc, n_cu);
        run<3, 4>("f32 MFMA + K ds_read_b32", w, d_out, d_cyc, n_cu);
        if (w == 2) {
            run<4, 2>("roles: MFMA wave | VALU wave", w, d_out, d_cyc, n_cu);
            run<4, 4>("roles: MFMA wave | VALU wave", w, d_out, d_cyc, n_cu);
            run<4, 8>("roles: MFMA wave | VALU wave", w, d_out, d_cyc, n_cu);
            run<4, 16>("roles: MFMA wave | VALU wave", w, d_out, d_cyc, n_cu);
        }
        run<2, 4>("(no MFMA)", w, d_out, d_cyc, n_cu);
        run<2, 8>("(no MFMA)", w, d_out, d_cyc, n_cu);
        run<2, 16>("(no MFMA)", w, d_out, d_cyc, n_cu);
    }
    return 0;
}
