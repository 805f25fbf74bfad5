// Does a small fp32 MFMA issued between VALU instructions cost VALU issue time?
// (Idea under test: accumulate qnorm's q*q on the otherwise idle matrix pipe with
// v_mfma_f32_4x4x1_16b_f32 D, q, q, D: the diagonal of each 4x4 block is q_lane^2.)
// Stream of independent v_fmac_f32 with MF MFMAs per 8 of them, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ unsigned long long g_clk[2];

template <int MF>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ck0 = __builtin_amdgcn_s_memtime();
    asm volatile(
        "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n"
        "v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n"
        "v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n"
        "v_mov_b32 v24, 0.25\n v_mov_b32 v25, 0.25\n v_mov_b32 v26, 0.25\n v_mov_b32 v27, 0.25\n" ::
        : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = 0.5f + threadIdx.x * 1e-4f;
    asm volatile("" : "+v"(x), "+v"(y));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile(
                "v_fmac_f32 v10, v23, v24\n v_fmac_f32 v11, v20, v25\n v_fmac_f32 v12, v21, v26\n v_fmac_f32 v13, v22, v27\n" ::
                : "v10", "v11", "v12", "v13");
            if (MF >= 1) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile(
                "v_fmac_f32 v14, v23, v24\n v_fmac_f32 v15, v20, v25\n v_fmac_f32 v16, v21, v26\n v_fmac_f32 v17, v22, v27\n" ::
                : "v14", "v15", "v16", "v17");
            if (MF >= 2) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r;
    asm volatile("v_add_f32 %0, v10, v11\n v_add_f32 %0, %0, v12\n v_add_f32 %0, %0, v13\n v_add_f32 %0, %0, v14" : "=v"(r)::"v10", "v11", "v12", "v13", "v14");
    r += acc0.x + acc0.y + acc0.z + acc0.w + acc1.x + acc1.y + acc1.z + acc1.w;
    if (r == 12345.678f) out[0] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_clk[0] = __builtin_amdgcn_s_memrealtime() - rt0;
        g_clk[1] = __builtin_amdgcn_s_memtime() - ck0;
    }
}

// correctness of the idea: diagonal of the 4x4 blocks accumulates q^2 per lane
__global__ void diag_check(float *out) {
    const float q = 1.0f + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(q, q, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(2 * q, 2 * q, acc, 0, 0, 0);
    const int i = threadIdx.x & 3;
    out[threadIdx.x] = i == 0 ? acc.x : i == 1 ? acc.y : i == 2 ? acc.z : acc.w;
}

template <int MF>
static void run(float *d) {
    const int iters = 20000, wps = 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MF>, dim3(256 * wps), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MF>, dim3(256 * wps), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2];
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    const double mhz = (double)clk[1] / (double)clk[0] * 100.0;
    printf("8 v_fmac_f32 + %d v_mfma_f32_4x4x1: clock %4.0f MHz, real cycles per group per SIMD = %.2f (VALU only would be %.2f per fmac)\n", MF, mhz,
           ms * 1e-3 * mhz * 1e6 / ((double)iters * 8 * wps), ms * 1e-3 * mhz * 1e6 / ((double)iters * 64 * wps));
}

int main() {
    float *d;
    hipMalloc(&d, 4096);
    hipLaunchKernelGGL(diag_check, dim3(1), dim3(64), 0, 0, d);
    float h[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += h[l] != 5.0f * (1.0f + l) * (1.0f + l);
    printf("diagonal check: %d lanes wrong (lane 5 -> %g, want %g)\n", bad, h[5], 5.0f * 36);
    run<0>(d); run<1>(d); run<2>(d);
    return 0;
}
