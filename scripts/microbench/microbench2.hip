// Second microbenchmark: what limits the oscillator-bank inner loop?
//  (a) dependent-chain issue rate of v_fma_f32 / v_pk_fma_f32 at ILP 1, 2, 4 and 1..8 waves/SIMD
//  (b) the K1 sample body itself (velocity form + output + qnorm, R = 2), register-only,
//      in three instruction orders, then with the LDS tile write, then with the row reads.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int ILP, bool PK>
__global__ __launch_bounds__(1024) void chain_kernel(float *out, int iters, float c) {
    float a[4]; v2f pa[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 1e-3f + i; pa[i] = (v2f){a[i], a[i] + 0.5f}; }
    const v2f pc = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (PK) pa[i] = __builtin_elementwise_fma(pa[i], pc, pc);
                else a[i] = fmaf(a[i], c, c);
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) r += a[i] + pa[i].x + pa[i].y;
    if (r == 12345.678f) out[0] = r;
}

// K1 body, R = 2 packed.  ORDER 0: as the compiler orders the naive source;
// ORDER 1: output/qnorm of the previous sample interleaved with the recurrence of this one.
__device__ unsigned long long g_clk[4];
template <int ORDER, int LDS>
__global__ __launch_bounds__(256) void body_kernel(float *out, int tiles, float c0, float c1) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ck0 = __builtin_amdgcn_s_memtime();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *tile = lds + wave * (27 * 68);
    v2f ca = {c0, c0 * 0.999f}, cb = {c1, c1 * 1.001f}, t = {1.0f, 0.5f};
    v2f q = {threadIdx.x * 1e-3f, 0.25f}, d = {1e-3f, 2e-3f}, qn = {0, 0}, qprev = q;
    float acc = 0;
    const unsigned m0 = wave * 27 * 68 * 4;
    const float4 *rsrc = (const float4 *)(tile + (lane >> 1) * 68 + (lane & 1) * 32);
    for (int tl = 0; tl < tiles; ++tl) {
        if (LDS) asm volatile("s_mov_b32 m0, %0" ::"s"(m0) : "memory");
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            float p;
            if (ORDER == 0) {
                v2f a = ca * d;
                a = __builtin_elementwise_fma(-cb, q, a);
                d = a;
                q = q + a;
                p = t.x * q.x;
                p = fmaf(t.y, q.y, p);
                qn = __builtin_elementwise_fma(q, q, qn);
            } else {
                v2f a = ca * d;
                p = t.x * qprev.x;
                a = __builtin_elementwise_fma(-cb, q, a);
                p = fmaf(t.y, qprev.y, p);
                qn = __builtin_elementwise_fma(qprev, qprev, qn);
                d = a;
                q = q + a;
                qprev = q;
            }
            if (LDS) asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(p), "n"(0) : "memory");
            else acc += p;
            asm volatile("" : "+v"(qn));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LDS >= 2) {
            __builtin_amdgcn_wave_barrier();
            float s = 0;
            if (lane < 54) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float4 v = rsrc[j]; s += v.x; s += v.y; s += v.z; s += v.w; }
            }
            acc += s;
        }
    }
    if (acc + qn.x + qn.y + q.x + d.y == 12345.678f) out[0] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_clk[0] = __builtin_amdgcn_s_memrealtime() - rt0;
        g_clk[1] = __builtin_amdgcn_s_memtime() - ck0;
    }
}

template <class K>
static float time_kernel(K launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int ILP, bool PK>
static void run_chain(float *d) {
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        int threads = 64 * 4 * wps, blocks = 256, bs = threads;
        if (threads > 1024) { bs = 1024; blocks = 256 * (threads / 1024); }
        float ms = time_kernel([&] { hipLaunchKernelGGL((chain_kernel<ILP, PK>), dim3(blocks), dim3(bs), 0, 0, d, iters, 0.999f); });
        double instr = (double)iters * 32 * ILP * wps;           // per SIMD
        printf("%s dependent ILP=%d waves/SIMD=%d: cyc/instr/SIMD=%.2f  (per-wave interval %.1f cyc)\n", PK ? "v_pk_fma" : "v_fma   ", ILP,
               wps, ms * 1e-3 * 2.4e9 / instr, ms * 1e-3 * 2.4e9 / ((double)iters * 32 * ILP));
    }
}

template <int ORDER, int LDS>
static void run_body(float *d, const char *name) {
    const int tiles = 8000;
    for (int wps : {2, 4, 5}) {
        // wps workgroups of 256 threads per CU -> wps waves per SIMD
        const size_t lds = 4 * 27 * 68 * 4;
        hipFuncSetAttribute((const void *)body_kernel<ORDER, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        float ms = time_kernel([&] { hipLaunchKernelGGL((body_kernel<ORDER, LDS>), dim3(256 * wps), dim3(256), lds, 0, d, tiles, 0.9995f, 2e-4f); });
        double ws = (double)tiles * 27 * wps;                    // wave-samples per SIMD
        unsigned long long clk[4];
        hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
        const double mhz = (double)clk[1] / (double)clk[0] * 100.0;
        printf("%-34s waves/SIMD=%d: nominal(2.4GHz) cycles per wave-sample per SIMD = %.1f; in-kernel clock %.0f MHz -> real cycles %.1f\n",
               name, wps, ms * 1e-3 * 2.4e9 / ws, mhz, ms * 1e-3 * mhz * 1e6 / ws);
    }
}

int main() {
    float *d;
    CHECK(hipMalloc(&d, 4096));
    run_chain<1, false>(d); run_chain<2, false>(d); run_chain<4, false>(d);
    run_chain<1, true>(d);  run_chain<2, true>(d);
    run_body<0, 0>(d, "body naive order, registers only");
    run_body<1, 0>(d, "body pipelined order, regs only");
    run_body<0, 1>(d, "body naive + addtid write");
    run_body<1, 1>(d, "body pipelined + addtid write");
    run_body<0, 2>(d, "body naive + write + row reads");
    run_body<1, 2>(d, "body pipelined + write + row reads");
    return 0;
}
