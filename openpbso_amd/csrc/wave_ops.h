// Wave-level helpers shared by the block kernels (gfx950, wave64).
#pragma once
#include <type_traits>

namespace pbso {

template <int CTRL, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float old, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, 0xF, BANK_MASK, false));
}

// Sixteen sums over the 64 lanes at once: v[i] summed over the wave, i = 0..15.  A butterfly that halves the number of
// values a lane carries at every stage (lane bit s decides which half it keeps and which it sends to lane ^ (1 << s)):
// 50 vector instructions for the four stages inside a row of 16 lanes instead of 16 x 8 for sixteen separate sums; the
// four rows' partial sums meet in LDS (`scratch`: 64 floats).  Returns, in lanes 0..15 (every row alike), the total of
// value bitrev4(lane & 15) -- taps_index() gives that index.
__device__ __forceinline__ int taps_index(int lane) {
    return ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
}
template <class Sync>
__device__ __forceinline__ float wave_sum16(const float (&v)[16], int lane, float *scratch, Sync &&wave_sync) {
    float a[8], b[4], c[2];
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 8; ++i)                      // lane ^ 1: quad_perm [1,0,3,2]
        a[i] = (b0 ? v[i + 8] : v[i]) + dpp_mov<0xB1>(0.f, b0 ? v[i] : v[i + 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i)                      // lane ^ 2: quad_perm [2,3,0,1]
        b[i] = (b1 ? a[i + 4] : a[i]) + dpp_mov<0x4E>(0.f, b1 ? a[i] : a[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {                    // lane ^ 4: row_shl:4 into banks 0 and 2, row_shr:4 into banks 1 and 3
        const float send = b2 ? b[i] : b[i + 2];
        c[i] = (b2 ? b[i + 2] : b[i]) + dpp_mov<0x114, 0xA>(dpp_mov<0x104, 0x5>(0.f, send), send);
    }
    const float send = b3 ? c[0] : c[1];
    const float d = (b3 ? c[1] : c[0]) + dpp_mov<0x128>(0.f, send);          // lane ^ 8: row_ror:8
    scratch[lane] = d;
    wave_sync();
    const int l = lane & 15;
    const float tot = (scratch[l] + scratch[16 + l]) + (scratch[32 + l] + scratch[48 + l]);
    return tot;
}

}  // namespace pbso
