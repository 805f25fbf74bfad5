// Issue-rate ceiling for the K1 sample body in its shipped (scalar, VOP2) form, in REAL
// shader cycles (s_memtime / s_memrealtime give the in-kernel clock):
//   (a) independent v_fmac_f32 stream, VGPR operands only        -> the VALU roof
//   (b) the velocity-form body for R modes per lane, registers only
//   (c) (b) + ds_write_addtid_b32 per sample
//   (d) (c) + the lagged row sums (8 ds_read_b128 per tile, 32 adds spread over the samples)
//   (e) (b) + one ds_write_b64 per TWO samples: no cheaper than (c), the cost is not the instruction count
// Each at 4 waves per SIMD (256-thread workgroups, 4 per CU) like the headline launch.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ unsigned long long g_clk[2];

template <int R, int LDS, int ORDER>
__global__ __launch_bounds__(256) void body_kernel(float *out, int tiles, float c0, float c1) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ck0 = __builtin_amdgcn_s_memtime();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *tile = lds + wave * (27 * 68);
    float ca[R], cb[R], t[R], q[R], d[R], qn[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ca[r] = c0 * (1.0f - 1e-3f * r); cb[r] = -c1 * (1.0f + 1e-3f * r); t[r] = 1.0f / (1 + r);
        q[r] = threadIdx.x * 1e-3f + r; d[r] = 1e-3f * (r + 1); qn[r] = 0;
        asm volatile("" : "+v"(ca[r]), "+v"(cb[r]), "+v"(t[r]));
    }
    float acc = 0, rs[8 * 4], pprev = 0;
    const unsigned waddr = (unsigned)(wave * 27 * 68 * 4 + lane * 8);
#pragma unroll
    for (int j = 0; j < 32; ++j) rs[j] = 0;
    const unsigned m0 = wave * 27 * 68 * 4;
    const float4 *rsrc = (const float4 *)(tile + (lane >> 1) * 68 + (lane & 1) * 32);
    if (LDS) asm volatile("s_mov_b32 m0, %0" ::"s"(m0) : "memory");
    for (int tl = 0; tl < tiles; ++tl) {
        if (LDS == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float4 v = rsrc[j]; rs[4 * j] = v.x; rs[4 * j + 1] = v.y; rs[4 * j + 2] = v.z; rs[4 * j + 3] = v.w; }
        }
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            float p = 0;
            if (ORDER == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float a = ca[r] * d[r];
                    a = fmaf(cb[r], q[r], a);
                    d[r] = a;
                    q[r] = q[r] + a;
                    p = r == 0 ? t[r] * q[r] : fmaf(t[r], q[r], p);
                    qn[r] = fmaf(q[r], q[r], qn[r]);
                    asm volatile("" : "+v"(qn[r]));
                }
            } else {
                // software-pipelined: output / qnorm of the PREVIOUS sample between the stages of
                // this sample's recurrence; every dependent pair at least R instructions apart
#define SB __builtin_amdgcn_sched_barrier(0)
                float a[R];
#pragma unroll
                for (int r = 0; r < R; ++r) { a[r] = ca[r] * d[r]; SB; }
#pragma unroll
                for (int r = 0; r < R; ++r) { qn[r] = fmaf(q[r], q[r], qn[r]); SB; }
#pragma unroll
                for (int r = 0; r < R; ++r) { a[r] = fmaf(cb[r], q[r], a[r]); SB; }
#pragma unroll
                for (int r = 0; r < R; ++r) { p = r == 0 ? t[r] * q[r] : fmaf(t[r], q[r], p); SB; }
#pragma unroll
                for (int r = 0; r < R; ++r) { d[r] = a[r]; q[r] = q[r] + a[r]; SB; }
            }
            if (LDS == 3) {
                // two samples per LDS instruction: ds_write_b64 with a per-lane address
                if (k & 1) {
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(waddr), "v"(__builtin_bit_cast(double, (float __attribute__((ext_vector_type(2)))){pprev, p})), "n"(0) : "memory");
                } else {
                    pprev = p;
                }
            } else if (LDS) asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(p), "n"(0) : "memory");
            else acc += p;
            if (LDS == 2) {
                if (k < 16) { acc += rs[2 * k]; asm volatile("" : "+v"(acc)); acc += rs[2 * k + 1]; asm volatile("" : "+v"(acc)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = acc;
#pragma unroll
    for (int r = 0; r < R; ++r) s += qn[r] + q[r] + d[r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_clk[0] = __builtin_amdgcn_s_memrealtime() - rt0;
        g_clk[1] = __builtin_amdgcn_s_memtime() - ck0;
    }
}

__global__ __launch_bounds__(256) void fmac_kernel(float *out, int iters) {
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ck0 = __builtin_amdgcn_s_memtime();
    asm volatile(
        "v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n"
        "v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n"
        "v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n"
        "v_mov_b32 v24, 0.25\n v_mov_b32 v25, 0.25\n v_mov_b32 v26, 0.25\n v_mov_b32 v27, 0.25\n" ::
        : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile(
                "v_fmac_f32 v10, v23, v24\n v_fmac_f32 v11, v20, v25\n v_fmac_f32 v12, v21, v26\n v_fmac_f32 v13, v22, v27\n"
                "v_fmac_f32 v14, v23, v24\n v_fmac_f32 v15, v20, v25\n v_fmac_f32 v16, v21, v26\n v_fmac_f32 v17, v22, v27\n" ::
                : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    }
    float r;
    asm volatile("v_add_f32 %0, v10, v11\n v_add_f32 %0, %0, v12\n v_add_f32 %0, %0, v13\n v_add_f32 %0, %0, v14" : "=v"(r)::"v10", "v11", "v12", "v13", "v14");
    if (r == 12345.678f) out[0] = r;
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

static double clock_mhz() {
    unsigned long long clk[2];
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
    return (double)clk[1] / (double)clk[0] * 100.0;      // s_memrealtime ticks at 100 MHz
}

template <int R, int LDS, int ORDER>
static void run_body(float *d, const char *name) {
    const int tiles = 8000, wps = 4;
    const size_t lds = 4 * 27 * 68 * 4;
    float ms = time_kernel([&] { hipLaunchKernelGGL((body_kernel<R, LDS, ORDER>), dim3(256 * wps), dim3(256), lds, 0, d, tiles, 0.9995f, 2e-4f); });
    const double ws = (double)tiles * 27 * wps, mhz = clock_mhz();
    const double valu = 5.0 * R + (LDS == 2 ? 32.0 / 27 : (LDS ? 0 : 1));
    const double cyc = ms * 1e-3 * mhz * 1e6 / ws;
    printf("R=%d order=%d %-34s clock %4.0f MHz  real cycles per wave-sample per SIMD = %5.1f  (%.1f VALU -> %.2f cycles per VALU instruction)\n", R, ORDER, name, mhz, cyc, valu,
           cyc / valu);
}

int main() {
    float *d;
    hipMalloc(&d, 4096);
    {
        const int iters = 20000, wps = 4;
        float ms = time_kernel([&] { hipLaunchKernelGGL(fmac_kernel, dim3(256 * wps), dim3(256), 0, 0, d, iters); });
        const double mhz = clock_mhz();
        printf("independent v_fmac_f32 (VGPR operands), 4 waves/SIMD: clock %4.0f MHz  real cycles per instruction per SIMD = %.2f\n", mhz,
               ms * 1e-3 * mhz * 1e6 / ((double)iters * 64 * wps));
    }
    run_body<2, 0, 0>(d, "body, registers only");
    run_body<2, 0, 1>(d, "body, registers only");
    run_body<2, 1, 0>(d, "body + addtid write");
    run_body<2, 1, 1>(d, "body + addtid write");
    run_body<2, 2, 0>(d, "body + write + lagged row sums");
    run_body<2, 2, 1>(d, "body + write + lagged row sums");
    run_body<2, 3, 0>(d, "body + ds_write_b64 per 2 samples");
    run_body<4, 3, 0>(d, "body + ds_write_b64 per 2 samples");
    run_body<4, 0, 0>(d, "body, registers only");
    run_body<4, 0, 1>(d, "body, registers only");
    run_body<4, 1, 0>(d, "body + addtid write");
    run_body<4, 1, 1>(d, "body + addtid write");
    run_body<4, 2, 0>(d, "body + write + lagged row sums");
    run_body<4, 2, 1>(d, "body + write + lagged row sums");
    return 0;
}
