// Does a stream wait on a VALUE in memory (hipStreamWaitValue32) release a kernel while the kernel that wrote the value is still
// running -- i.e. can a launch be ordered behind the START of another stream's kernel?  (Events only order behind its end.)
//   stream A: busy kernel: workgroup 0 writes the flag when it starts, then every workgroup spins `spin_us`
//   stream B: hipStreamWaitValue32(flag >= seq); probe kernel (stamps the wall clock)
// prints, per round: probe start relative to the busy kernel's start and end, and the same with an event wait instead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void busy(unsigned *flag, unsigned seq, unsigned long long *stamps, int spin_ticks) {
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stamps[0] = t0;
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[1] = wall_clock64();
}
// a kernel that leaves dirty lines in the L2s: every workgroup writes `per_wg` floats, stamps its end
// MODE 0: plain stores; 1: __builtin_nontemporal_store; 2: global_store ... sc0 sc1 (write-through, system scope); 3: ... sc1; 4: ... nt sc0 sc1
template <int MODE>
__global__ void writer(float *out, size_t per_wg, unsigned long long *stamps) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 *dst = reinterpret_cast<f4 *>(out + (size_t)blockIdx.x * per_wg);
    for (size_t i = threadIdx.x; i < per_wg / 4; i += blockDim.x) {
        const f4 v = {(float)i, 1.f, 2.f, 3.f};
        if (MODE == 0) dst[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, &dst[i]);
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(&dst[i]), "v"(v) : "memory");
        else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(&dst[i]), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(&dst[i]), "v"(v) : "memory");
    }
    if (threadIdx.x == 0) atomicMax(&stamps[1], wall_clock64());
}
__global__ void probe(unsigned long long *stamps) {
    if (threadIdx.x == 0) stamps[2] = wall_clock64();
}

int main(int argc, char **argv) {
    const int spin_us = argc > 1 ? std::atoi(argv[1]) : 150;
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned *flag = nullptr;
    CK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
    CK(hipMemset(flag, 0, 8));
    unsigned long long *stamps = nullptr;
    CK(hipHostMalloc((void **)&stamps, 64));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int round = 1; round <= 6; ++round) {
        stamps[0] = stamps[1] = stamps[2] = 0;
        // value wait: the probe is queued BEFORE the busy kernel is launched
        CK(hipStreamWaitValue32(b, flag, (unsigned)round, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, b, stamps);
        hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, a, flag, (unsigned)round, stamps, spin_us * 100);
        CK(hipStreamSynchronize(a));
        CK(hipStreamSynchronize(b));
        std::printf("value wait: busy ran %.1f us; probe started %.1f us after the busy kernel's start (%.1f us before its end)\n",
                    (stamps[1] - stamps[0]) / 100.0, ((long long)stamps[2] - (long long)stamps[0]) / 100.0, ((long long)stamps[1] - (long long)stamps[2]) / 100.0);
    }
    for (int round = 7; round <= 10; ++round) {
        stamps[0] = stamps[1] = stamps[2] = 0;
        hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, a, flag, (unsigned)round, stamps, spin_us * 100);
        CK(hipEventRecord(ev, a));
        CK(hipStreamWaitEvent(b, ev, 0));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, b, stamps);
        CK(hipStreamSynchronize(a));
        CK(hipStreamSynchronize(b));
        std::printf("event wait: probe started %.1f us after the busy kernel's END\n", ((long long)stamps[2] - (long long)stamps[1]) / 100.0);
    }
    for (int round = 11; round <= 14; ++round) {
        stamps[0] = stamps[1] = stamps[2] = 0;
        hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, a, flag, (unsigned)round, stamps, spin_us * 100);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, a, stamps);
        CK(hipStreamSynchronize(a));
        std::printf("same stream: probe started %.1f us after the busy kernel's END\n", ((long long)stamps[2] - (long long)stamps[1]) / 100.0);
    }
    // ---- what the packets between two kernels of ONE stream cost (the engine's bank stream: wait for the preparation's event,
    //      bank, record the bank's event)
    hipEvent_t done_ev;                                  // an event of stream b that completed long ago
    CK(hipEventCreateWithFlags(&done_ev, hipEventDisableTiming));
    CK(hipEventRecord(done_ev, b));
    CK(hipStreamSynchronize(b));
    unsigned *flag2 = nullptr;
    CK(hipExtMallocWithFlags((void **)&flag2, 8, hipMallocSignalMemory));
    CK(hipMemset(flag2, 0, 8));
    CK(hipDeviceSynchronize());
    const char *names[] = {"nothing", "hipEventRecord (timing disabled)", "hipStreamWaitEvent on a completed event", "record + wait (the engine today)",
                           "hipStreamWriteValue32", "hipStreamWaitValue32, satisfied", "write value + wait value"};
    for (int mode = 0; mode < 7; ++mode) {
        double sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            stamps[0] = stamps[1] = stamps[2] = 0;
            hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, a, flag, 100u + mode, stamps, 5000);
            if (mode == 1 || mode == 3) CK(hipEventRecord(ev, a));
            if (mode == 2 || mode == 3) CK(hipStreamWaitEvent(a, done_ev, 0));
            if (mode == 4 || mode == 6) CK(hipStreamWriteValue32(a, flag2, 7u, 0));
            if (mode == 5 || mode == 6) CK(hipStreamWaitValue32(a, flag, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, a, stamps);
            CK(hipStreamSynchronize(a));
            if (rep) sum += ((long long)stamps[2] - (long long)stamps[1]) / 100.0;
        }
        std::printf("same stream, between the kernels: %-44s gap %.1f us\n", names[mode], sum / 4);
    }
    // ---- the release fence of an event behind a kernel that wrote a lot (the bank writes 181 MB of samples per launch)
    {
        float *big = nullptr;
        const size_t per_wg = 64 << 10, n_wg = 2048;                     // 512 MB written
        CK(hipMalloc((void **)&big, per_wg * n_wg * sizeof(float)));
        hipEvent_t ev_dev, ev_nofence;
        CK(hipEventCreateWithFlags(&ev_dev, hipEventDisableTiming | hipEventReleaseToDevice));
        CK(hipEventCreateWithFlags(&ev_nofence, hipEventDisableTiming | hipEventDisableSystemFence));
        const char *wn[] = {"nothing", "hipEventRecord (timing disabled)", "hipEventRecord (+ hipEventReleaseToDevice)", "hipEventRecord (+ hipEventDisableSystemFence)"};
        for (int mode = 0; mode < 4; ++mode) {
            double sum = 0;
            for (int rep = 0; rep < 5; ++rep) {
                stamps[0] = stamps[1] = stamps[2] = 0;
                hipLaunchKernelGGL(writer<0>, dim3(n_wg), dim3(256), 0, a, big, per_wg, stamps);
                if (mode == 1) CK(hipEventRecord(ev, a));
                if (mode == 2) CK(hipEventRecord(ev_dev, a));
                if (mode == 3) CK(hipEventRecord(ev_nofence, a));
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, a, stamps);
                CK(hipStreamSynchronize(a));
                if (rep) sum += ((long long)stamps[2] - (long long)stamps[1]) / 100.0;
            }
            std::printf("behind a kernel that wrote 512 MB, same stream: %-46s gap %.1f us\n", wn[mode], sum / 4);
        }
    }
    // ---- the same with stores that do not stay dirty in the L2
    {
        float *big = nullptr;
        const size_t per_wg = 64 << 10, n_wg = 2048;
        CK(hipMalloc((void **)&big, per_wg * n_wg * sizeof(float)));
        const char *sn[] = {"plain stores", "__builtin_nontemporal_store", "global_store sc0 sc1", "global_store sc1", "global_store sc0 sc1 nt"};
        for (int mode = 0; mode < 5; ++mode) {
            double sum = 0, dur = 0;
            for (int rep = 0; rep < 5; ++rep) {
                stamps[0] = stamps[1] = stamps[2] = 0;
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, a, stamps + 4);        // stamps[6] = start
                if (mode == 0) hipLaunchKernelGGL(writer<0>, dim3(n_wg), dim3(256), 0, a, big, per_wg, stamps);
                if (mode == 1) hipLaunchKernelGGL(writer<1>, dim3(n_wg), dim3(256), 0, a, big, per_wg, stamps);
                if (mode == 2) hipLaunchKernelGGL(writer<2>, dim3(n_wg), dim3(256), 0, a, big, per_wg, stamps);
                if (mode == 3) hipLaunchKernelGGL(writer<3>, dim3(n_wg), dim3(256), 0, a, big, per_wg, stamps);
                if (mode == 4) hipLaunchKernelGGL(writer<4>, dim3(n_wg), dim3(256), 0, a, big, per_wg, stamps);
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, a, stamps);
                CK(hipStreamSynchronize(a));
                if (rep) { sum += ((long long)stamps[2] - (long long)stamps[1]) / 100.0; dur += ((long long)stamps[1] - (long long)stamps[6]) / 100.0; }
            }
            std::printf("a kernel that writes 512 MB with %-30s runs %.1f us; the next kernel starts %.1f us after its last store\n", sn[mode], dur / 4, sum / 4);
        }
    }
    // ---- a dependency ACROSS streams: stream a's kernel, then (1) event record / wait, (2) stream write value / wait value,
    //      (3) the kernel's own last workgroup writes the value (a completion counter)
    for (int mode = 0; mode < 2; ++mode) {
        double sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            stamps[0] = stamps[1] = stamps[2] = 0;
            const unsigned seq = 1000u + 10 * mode + rep;
            CK(hipStreamWaitValue32(b, flag2, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, b, stamps);
            hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, a, flag, seq, stamps, 5000);
            if (mode == 0) CK(hipStreamWriteValue32(a, flag2, seq, 0));
            else hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, a, flag2, seq, stamps + 4, 0);      // (a one-wave kernel behind it writes the value)
            CK(hipStreamSynchronize(a));
            CK(hipStreamSynchronize(b));
            if (rep) sum += ((long long)stamps[2] - (long long)stamps[1]) / 100.0;
        }
        std::printf("across streams, %-52s gap %.1f us\n", mode == 0 ? "hipStreamWriteValue32 behind the kernel + wait value:" : "a one-wave kernel behind it writes the value + wait value:", sum / 4);
    }
    return 0;
}
