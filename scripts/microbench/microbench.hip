// VALU issue-rate microbenchmark for gfx950: decides how the oscillator bank
// is written.  Measures, per SIMD, cycles per wave-instruction of
//   v_fma_f32 (VGPR operands), v_fma_f32 with an SGPR operand, v_pk_fma_f32,
//   v_pk_fma_f32 with SGPR pair, and ds_write_b32 / ds_read_b128 in the K1 shape,
// at 1, 2, 3, 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void valu_kernel(float *out, int iters, float s0, float s1) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 1.0f + i * 1e-4f; }
    float2v pa[4], pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { pa[i] = {a[2 * i], a[2 * i + 1]}; pb[i] = {b[2 * i], b[2 * i + 1]}; }
    const float2v ps = {s0, s1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(s0), "v"(b[i]));
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pa[i]) : "v"(pb[i]));
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[i]) : "s"(ps), "v"(pb[i]));
            } else if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb[i]));
            } else if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[i]) : "v"(pb[i]));
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) r += pa[i].x + pa[i].y;
    if (r == 12345.678f) out[0] = r;
}

// K1-shaped LDS traffic: per "sample" one ds_write_b32 per lane; per 57 samples 16 ds_read_b128
__global__ __launch_bounds__(1024) void lds_kernel(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = lds + wave * (57 * 68);
    float acc = 0, v = lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 57; ++k) { tile[k * 68 + lane] = v; v += 1.0f; }
        __builtin_amdgcn_wave_barrier();
        if (lane < 57) {
            const float4 *row = (const float4 *)(tile + lane * 68);
#pragma unroll
            for (int j = 0; j < 16; ++j) { float4 x = row[j]; acc += x.x + x.y + x.z + x.w; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
static int run_valu(const char *name, int per_iter_instr, float lanes_per_instr) {
    float *d;
    CHECK(hipMalloc(&d, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4, 8}) {
        // one workgroup per CU with 4*wps waves -> wps waves per SIMD
        const int threads = 64 * 4 * wps;
        int blocks = 256;
        int bs = threads;
        if (threads > 1024) { bs = 1024; blocks = 256 * (threads / 1024); }
        hipLaunchKernelGGL(valu_kernel<MODE>, dim3(blocks), dim3(bs), 0, 0, d, 100, 1.0f, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(valu_kernel<MODE>, dim3(blocks), dim3(bs), 0, 0, d, iters, 1.0f, 1.0f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)iters * 8 * per_iter_instr * wps;
        const double cyc = ms * 1e-3 * 2.4e9;
        const double gflops = (double)iters * 8 * per_iter_instr * lanes_per_instr * 2.0 * 64 * 4 * wps * 256 / (ms * 1e-3) * 1e-12;
        printf("%-28s waves/SIMD=%d  ms=%.3f  cyc/instr/SIMD(@2.4GHz)=%.2f  TFLOP/s=%.1f\n", name, wps, ms,
               cyc / instr_per_simd, gflops);
    }
    CHECK(hipFree(d));
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    if (run_valu<0>("v_fma_f32 vgpr", 8, 1)) return 1;
    if (run_valu<1>("v_fma_f32 sgpr-operand", 8, 1)) return 1;
    if (run_valu<2>("v_pk_fma_f32 vgpr", 4, 2)) return 1;
    if (run_valu<3>("v_pk_fma_f32 sgpr-pair", 4, 2)) return 1;
    if (run_valu<4>("v_pk_mul_f32", 4, 1)) return 1;
    if (run_valu<5>("v_pk_add_f32", 4, 1)) return 1;
    // LDS
    float *d;
    CHECK(hipMalloc(&d, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int waves : {4, 8, 10}) {
        const size_t lds = (size_t)waves * 57 * 68 * 4;
        CHECK(hipFuncSetAttribute((const void *)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int iters = 2000;
        hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(64 * waves), lds, 0, d, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(64 * waves), lds, 0, d, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc_per_wave_sample = ms * 1e-3 * 2.4e9 / ((double)iters * 57 * waves);
        printf("lds transpose tile: waves/CU=%d  ms=%.3f  CU-cycles per wave-sample=%.2f\n", waves, ms, cyc_per_wave_sample);
    }
    return 0;
}
