// VGPR bank microbenchmark: does v_fma_f32 / v_mul_f32 slow down when its source
// operands sit in the same register bank (register index mod 4)?  Independent
// instructions, explicit registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; } } while (0)

#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17"

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters) {
    asm volatile(
        "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n"
        "v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n"
        "v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n"
        "v_mov_b32 v24, 0.25\n v_mov_b32 v25, 0.25\n v_mov_b32 v26, 0.25\n v_mov_b32 v27, 0.25\n" ::
        : CLOB, "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {        // dst/src2, src0, src1 all in one bank
                asm volatile(
                    "v_fma_f32 v10, v22, v26, v10\n v_fma_f32 v11, v23, v27, v11\n v_fma_f32 v12, v20, v24, v12\n v_fma_f32 v13, v21, v25, v13\n"
                    "v_fma_f32 v14, v22, v26, v14\n v_fma_f32 v15, v23, v27, v15\n v_fma_f32 v16, v20, v24, v16\n v_fma_f32 v17, v21, v25, v17\n" ::: CLOB);
            } else if (MODE == 1) { // three different banks
                asm volatile(
                    "v_fma_f32 v10, v23, v24, v10\n v_fma_f32 v11, v20, v25, v11\n v_fma_f32 v12, v21, v26, v12\n v_fma_f32 v13, v22, v27, v13\n"
                    "v_fma_f32 v14, v23, v24, v14\n v_fma_f32 v15, v20, v25, v15\n v_fma_f32 v16, v21, v26, v16\n v_fma_f32 v17, v22, v27, v17\n" ::: CLOB);
            } else if (MODE == 2) { // v_mul (VOP2), both sources in one bank
                asm volatile(
                    "v_mul_f32 v10, v22, v10\n v_mul_f32 v11, v23, v11\n v_mul_f32 v12, v20, v12\n v_mul_f32 v13, v21, v13\n"
                    "v_mul_f32 v14, v22, v14\n v_mul_f32 v15, v23, v15\n v_mul_f32 v16, v20, v16\n v_mul_f32 v17, v21, v17\n" ::: CLOB);
            } else if (MODE == 3) { // v_mul, different banks
                asm volatile(
                    "v_mul_f32 v10, v23, v10\n v_mul_f32 v11, v20, v11\n v_mul_f32 v12, v21, v12\n v_mul_f32 v13, v22, v13\n"
                    "v_mul_f32 v14, v23, v14\n v_mul_f32 v15, v20, v15\n v_mul_f32 v16, v21, v16\n v_mul_f32 v17, v22, v17\n" ::: CLOB);
            } else if (MODE == 4) { // v_fmac (VOP2) dst+src in 3 different banks
                asm volatile(
                    "v_fmac_f32 v10, v23, v24\n v_fmac_f32 v11, v20, v25\n v_fmac_f32 v12, v21, v26\n v_fmac_f32 v13, v22, v27\n"
                    "v_fmac_f32 v14, v23, v24\n v_fmac_f32 v15, v20, v25\n v_fmac_f32 v16, v21, v26\n v_fmac_f32 v17, v22, v27\n" ::: CLOB);
            } else {                // v_fmac all one bank
                asm volatile(
                    "v_fmac_f32 v10, v22, v26\n v_fmac_f32 v11, v23, v27\n v_fmac_f32 v12, v20, v24\n v_fmac_f32 v13, v21, v25\n"
                    "v_fmac_f32 v14, v22, v26\n v_fmac_f32 v15, v23, v27\n v_fmac_f32 v16, v20, v24\n v_fmac_f32 v17, v21, v25\n" ::: CLOB);
            }
        }
    }
    float r;
    asm volatile("v_add_f32 %0, v10, v11\n v_add_f32 %0, %0, v12\n v_add_f32 %0, %0, v13\n v_add_f32 %0, %0, v14" : "=v"(r) :: CLOB);
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
static int run(const char *name) {
    float *d;
    CHECK(hipMalloc(&d, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps : {2, 4, 8}) {
        int threads = 64 * 4 * wps, blocks = 256, bs = threads;
        if (threads > 1024) { bs = 1024; blocks = 256 * (threads / 1024); }
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(bs), 0, 0, d, 100);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(bs), 0, 0, d, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-36s waves/SIMD=%d  nominal cyc/instr/SIMD = %.2f\n", name, wps, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
    }
    return 0;
}
int main() {
    if (run<0>("v_fma_f32 all operands same bank")) return 1;
    if (run<1>("v_fma_f32 three different banks")) return 1;
    if (run<2>("v_mul_f32 both sources same bank")) return 1;
    if (run<3>("v_mul_f32 different banks")) return 1;
    if (run<5>("v_fmac_f32 all operands same bank")) return 1;
    if (run<4>("v_fmac_f32 three different banks")) return 1;
    return 0;
}
