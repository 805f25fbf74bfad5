// Unit test of pbso::wave_sum16 (openpbso_amd/csrc/wave_ops.h): sixteen sums over the 64 lanes in one butterfly, used for the
// FIR taps of the forced block path.  Built and run by tests/test_gpu_wave_ops.py on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "wave_ops.h"
__global__ void k(const float *in, float *out) {
    __shared__ float scr[64];
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i];
    auto ws = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    const float t = pbso::wave_sum16(v, threadIdx.x, scr, ws);
    out[threadIdx.x] = t;
    out[64 + threadIdx.x] = (float)pbso::taps_index(threadIdx.x);
}
int main() {
    float h[1024], o[128], *di, *dout;
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 7919) % 101) - 50.f + 0.25f * (i % 3);
    if (hipMalloc(&di, sizeof(h)) != hipSuccess || hipMalloc(&dout, sizeof(o)) != hipSuccess) return 2;
    if (hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    if (hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int idx = (int)o[64 + l];
        double want = 0; for (int t = 0; t < 64; ++t) want += h[t * 16 + idx];
        if (fabs(want - o[l]) > 1e-3) { ++bad; if (bad < 5) printf("lane %d idx %d got %f want %f\n", l, idx, o[l], want); }
    }
    printf("wave_sum16: %s (%d bad)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
