// Side work on CUs of its own?  (diagnostic, not part of the library)
//     hipcc --offload-arch=gfx950 -O2 cu_mask.hip -o cu_mask && ./cu_mask
// (1) Which CUs does bit i of a hipExtStreamCreateWithCUMask mask name?  A probe kernel reports (XCC_ID, CU id of HW_ID) of
//     every workgroup that ran on a stream with bits [0, k) set, and with every 8th bit set.
// (2) A "bank" -- workgroups of two 256-register waves and 33 KB of LDS, as many as fill the CUs it may use, each spinning T us --
//     and, beside it, a chain of six small dependent kernels on another stream (the preparation of the next launch: 64
//     workgroups of 256 threads, ~5 us each alone).  When does the chain end, measured from the bank's start,
//       a) no masks (today: the chain's workgroups find no registers until bank workgroups retire),
//       b) the chain's stream on k CUs per XCD, the bank unmasked (the bank still takes those CUs' registers first),
//       c) the chain's stream on k CUs per XCD and the bank's stream on the others (the bank sized for them)?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(unsigned *out) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;
        out[blockIdx.x] = (xcc << 16) | (((hw >> 13) & 7u) << 8) | ((hw >> 8) & 0xFu);      // XCC | SE | CU within the SE's array
    }
    // long enough that every CU the stream may use is handed a workgroup
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) {}
}

// two waves of 256 registers + 33 KB of LDS per workgroup: four per CU, nothing else fits beside them
__global__ __launch_bounds__(128) void bank(float *out, unsigned long long *t_start, int ticks) {
    extern __shared__ float lds[];
    float v = threadIdx.x;
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    lds[threadIdx.x] = v;
    if (blockIdx.x == 0 && threadIdx.x == 0) *t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) v = v * 1.0001f + lds[threadIdx.x];
    out[blockIdx.x * 128 + threadIdx.x] = v;
}

__global__ __launch_bounds__(256) void small(float *buf, unsigned long long *t_end, int ticks, int last) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float v = buf[blockIdx.x * 256 + threadIdx.x];
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) v = v * 1.0001f + 1.f;
    buf[blockIdx.x * 256 + threadIdx.x] = v;
    if (last && threadIdx.x == 0) atomicMax(t_end, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

static int where(hipStream_t s, unsigned *d_out, int n_wg, const char *label) {
    hipMemset(d_out, 0xff, n_wg * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(n_wg), dim3(64), 0, s, d_out);
    CHECK(hipStreamSynchronize(s));
    std::vector<unsigned> h(n_wg);
    hipMemcpy(h.data(), d_out, n_wg * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, int>> per_xcc;
    for (unsigned x : h) per_xcc[x >> 16][x & 0xFFFF]++;
    std::printf("%s: ", label);
    for (auto &kv : per_xcc) std::printf("xcc %u: %zu CUs  ", kv.first, kv.second.size());
    std::printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::printf("device %s, %d CUs\n", prop.name, n_cu);
    unsigned *d_probe;
    float *d_out, *d_buf;
    unsigned long long *d_t;
    CHECK(hipMalloc(&d_probe, 4096 * sizeof(unsigned)));
    CHECK(hipMalloc(&d_out, (size_t)4096 * 128 * 4));
    CHECK(hipMalloc(&d_buf, (size_t)64 * 256 * 4));
    CHECK(hipMalloc(&d_t, 2 * sizeof(unsigned long long)));
    CHECK(hipMemset(d_buf, 0, (size_t)64 * 256 * 4));
    const int words = (n_cu + 31) / 32;
    auto masked = [&](hipStream_t *s, const std::vector<uint32_t> &m) { return hipExtStreamCreateWithCUMask(s, (uint32_t)m.size(), m.data()); };
    // (1) what the bits name
    {
        std::vector<uint32_t> m(words, 0);
        for (int i = 0; i < 16; ++i) m[i / 32] |= 1u << (i % 32);
        hipStream_t s;
        CHECK(masked(&s, m));
        if (where(s, d_probe, 2048, "bits 0..15 set          ")) return 1;
        hipStreamDestroy(s);
        std::fill(m.begin(), m.end(), 0u);
        for (int i = 0; i < n_cu; i += 8) m[i / 32] |= 1u << (i % 32);
        CHECK(masked(&s, m));
        if (where(s, d_probe, 2048, "every 8th bit set       ")) return 1;
        hipStreamDestroy(s);
        std::fill(m.begin(), m.end(), 0u);
        for (int i = 0; i < n_cu; ++i) if (i >= 16) m[i / 32] |= 1u << (i % 32);
        CHECK(masked(&s, m));
        if (where(s, d_probe, 4096, "all but bits 0..15      ")) return 1;
        hipStreamDestroy(s);
    }
    // (2) a bank and a chain of small kernels beside it
    hipFuncSetAttribute(reinterpret_cast<const void *>(bank), hipFuncAttributeMaxDynamicSharedMemorySize, 33 * 1024);
    const int bank_ticks = 30000;                       // 300 us (100 MHz)
    const int small_ticks = 500;                        // 5 us
    for (int k : {0, 16, 32, 64}) {                     // bits given to the chain's stream (0: no masks)
        for (int bank_masked = 0; bank_masked <= (k ? 1 : 0); ++bank_masked) {
            hipStream_t sb, sc;
            std::vector<uint32_t> mc(words, 0), mb(words, 0);
            for (int i = 0; i < n_cu; ++i) (i < k ? mc : mb)[i / 32] |= 1u << (i % 32);
            if (k) CHECK(masked(&sc, mc)); else CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
            if (bank_masked) CHECK(masked(&sb, mb)); else CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
            const int bank_wgs = 4 * (bank_masked ? n_cu - k : n_cu);
            std::vector<double> ends, banks;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipMemset(d_t, 0, 2 * sizeof(unsigned long long)));
                CHECK(hipDeviceSynchronize());
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, sb);
                hipLaunchKernelGGL(bank, dim3(bank_wgs), dim3(128), 33 * 1024, sb, d_out, d_t, bank_ticks);
                hipEventRecord(e1, sb);
                for (int j = 0; j < 6; ++j) hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, sc, d_buf, d_t + 1, small_ticks, j == 5);
                CHECK(hipDeviceSynchronize());
                unsigned long long t[2];
                hipMemcpy(t, d_t, sizeof(t), hipMemcpyDeviceToHost);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                ends.push_back(((double)t[1] - (double)t[0]) / 100.0);
                banks.push_back(ms * 1e3);
                hipEventDestroy(e0); hipEventDestroy(e1);
            }
            std::sort(ends.begin(), ends.end());
            std::sort(banks.begin(), banks.end());
            std::printf("chain's stream on %3d mask bits, bank %-8s (%4d workgroups): the chain of six 5-us kernels ends %7.1f us after the bank's start (median of 5; min %.1f); bank %.1f us\n",
                        k, bank_masked ? "masked" : "unmasked", bank_wgs, ends[2], ends[0], banks[2]);
            hipStreamDestroy(sb); hipStreamDestroy(sc);
        }
    }
    return 0;
}
