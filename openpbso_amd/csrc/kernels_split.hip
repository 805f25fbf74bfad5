// K1s: the block state-space oscillator bank for an UNDER-FILLED chip (gfx950, wave64, f32 MFMA).
//
// Replaces the same reference code as K1 / K1b: the hot loop of ModalSolver::step (modal_solver.h:262-272) around
// ModalIntegrator::Step (modal_integrator.h:103-113).  Same formulation as kernels_block.hip (read that first): a buffer
// is sample 0 + 2 groups of 16 blocks of 16 samples; block-start states are parked in LDS and projected on the f32
// matrix pipe with the per-mode table W = (a_j, b_j).
//
// When a scene has fewer than one wave of oscillators per SIMD (BASELINE configs[1], [2], [4]: 8 .. 512 waves for 1024
// SIMDs), a launch is bound by the LATENCY of one buffer in one wave -- a lone wave issues one vector instruction per
// four cycles whatever the chip could do -- and most SIMDs idle.  K1s spends the idle SIMDs on time: a team is TWO waves
// that own the SAME 64 modes (one mode per lane), and wave g projects group g of every buffer.
//   * force-free buffers: wave 0 steps sample 0 and the 32 coarse steps, parking t x; both waves project their 256
//     samples side by side (32 MFMAs each instead of 64 in one wave);
//   * buffers with a dense force profile (Gaussian, AR: forces.h:92-128), no qnorm rows: the response is linear, so wave
//     1 steps group 1 FROM ZERO under the profile's second half while wave 0 steps group 0 from the real state; when
//     wave 0 hands over the state at the group boundary, wave 1 adds its free response (16 coarse steps) to its parked
//     block states (superposition) and returns the end state.  256 dependent samples per buffer instead of 512;
//   * dense buffers with qnorm rows (getQBufferNorm needs the true state of every sample): wave 0 steps all 512 samples,
//     the projection is still shared.
// The registers hold the state UNSCALED (the parked values carry the transfer weight: two products per block, nothing
// in a latency-bound launch), so there is no literal fallback for unusable weights and no rescaling when the listener
// moves.  Two workgroup barriers per buffer; every output sample is produced by exactly one wave and stored straight
// from the MFMA's result registers.
#include <type_traits>

#include "kernels.h"
#include "wave_ops.h"

namespace pbso {
namespace iir_split {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int BJ = BLOCK_J, BN = BLOCK_N, GROUP = BJ * BN;

constexpr int ST_ROW = 130;                          // as K1b: [16 blocks][64 lanes][Q, D], row stride 130 floats
constexpr int ST_AREA = BN * ST_ROW + 32 + 132;      // + the wave's FIR taps h_0 .. h_16 + one row for the taps' virtual state
constexpr int U_ROW = 36;                            // increments: [64 modes][16 Q | 16 D] + 4 floats of padding (as K1b's FTM)

struct SplitDims {
    int nb, m_pad, b_pad, frames;
    long long audio_stride, plane;
    int qn_nb, qn_b0;
};

__device__ __forceinline__ float wave_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});
    v += dpp(v, std::integral_constant<int, 0x4E>{});
    v += dpp(v, std::integral_constant<int, 0x141>{});
    v += dpp(v, std::integral_constant<int, 0x140>{});
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

template <int K0, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        f(std::integral_constant<int, K0>{});
        static_for<K0 + 1, N - 1>(f);
    }
}

template <int QNM>
__global__ __launch_bounds__(128) void iir_split_kernel(
    const float *__restrict__ p_ca, const float *__restrict__ p_cb, float *__restrict__ p_sq, float *__restrict__ p_sd,
    float *__restrict__ p_ss, const BufDesc *__restrict__ p_desc, const float *__restrict__ p_grows,
    const float *__restrict__ p_tprof, const double *__restrict__ p_xfer_rows, const int *__restrict__ p_xfer_init,
    float *__restrict__ p_audio, float *__restrict__ p_qnorm, const float *__restrict__ p_gq, const float *__restrict__ p_pc,
    const float *__restrict__ p_wtab, const TeamDesc *__restrict__ p_teams, float *__restrict__ p_audio_parts,
    const float *__restrict__ p_ftab, unsigned long long *__restrict__ p_census, const SplitDims p) {
    constexpr bool QN = QNM != 0;
    __shared__ __attribute__((aligned(16))) float lds_stage[2][2][ST_AREA];       // [buffer parity][group]
    __shared__ __attribute__((aligned(16))) float lds_incr[QN ? 1 : 2][QN ? 4 : 64 * U_ROW];     // [group]: the blocks' state increments on their way back to lane = mode
    __shared__ f2 hand16[64], hand32[64];            // the state at the group boundary (wave 0 -> 1) and at the buffer's end (1 -> 0)
    const TeamDesc team = p_teams[blockIdx.x];
    const int obj = team.obj;
    const int lane = threadIdx.x & 63;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);             // this wave projects group grp
    const size_t ubase = (size_t)obj * p.m_pad + team.col0;
    const unsigned ul = (unsigned)lane;

    // per-mode constants (one mode per lane)
    const float nca = (p_ca + ubase)[ul], ncb = (p_cb + ubase)[ul];                // eps^2, -e  (velocity form, as K1)
    const float p11 = (p_pc + ubase)[ul], p12 = (p_pc + p.plane + ubase)[ul];     // P = A^16: P11 - 1, P12, P21, P22
    const float p21 = (p_pc + 2 * p.plane + ubase)[ul], p22 = (p_pc + 3 * p.plane + ubase)[ul];
    const bool dead = nca == 0.f && ncb == 0.f;                                   // a padding column
    float g11 = 0.f, g12 = 0.f, g22 = 0.f;
    if (QN) {
        g11 = (p_gq + ubase)[ul];
        g12 = (p_gq + p.plane + ubase)[ul];
        g22 = (p_gq + 2 * p.plane + ubase)[ul];
    }
    float wreg[32];                                  // the MFMA A operand of the 32 pairs of columns (as K1b)
    {
        const float *__restrict__ wsrc = p_wtab + (ubase / 2) * 64;
#pragma unroll
        for (int s = 0; s < 32; ++s) wreg[s] = wsrc[s * 64 + lane];
    }
    // Dense profiles without qnorm rows: a block's state increment is F . T_n, F = [A^15 u .. A u, u] (the table of K1b's forced
    // block path; absent -- PBSO_FORCED_BLOCK=0 -- the group is stepped sample by sample).  The increments of a group's 16
    // blocks are a [16 blocks x 16 taps] . [16 taps x 16 modes] product per tile of 16 modes and state component: 32 MFMAs
    // whose A operand is the profile as the FIR's B operand holds it and whose B operand is F, resident here.
    const bool ft = !QN && p_ftab != nullptr;
    float fB[QN ? 1 : 4][2][4];
    if (!QN && ft) {
#pragma unroll
        for (int tl = 0; tl < 4; ++tl)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    fB[QN ? 0 : tl][c][ks] = (p_ftab + (size_t)(2 * (4 * ks + (lane >> 4)) + c) * p.plane + ubase)[16 * tl + (lane & 15)];
    }
    // FIR taps of a dense profile (kernels_block.hip, "forced block path"): h_d = sum over modes of t g phi_d, phi_d = e1' A^d u the
    // mode's response d samples after a unit force sample (u = (1, 1)': d += f, q += d).  phi is sixteen constants per mode,
    // stepped here once per launch (fp64 from the f32 coefficients the per-sample kernels use); the sixteen sums over the wave
    // are one butterfly (wave_ops.h) -- the projection of a virtual block-start state on the matrix pipe that this replaces
    // (32 operand reads + 32 MFMAs for one useful column) was 2 K of a dense buffer's 6 K cycles on the critical path.
    float phi[16];
    {
        double vq = 1.0, vd = 1.0;
        phi[0] = 1.f;
#pragma unroll
        for (int d = 1; d < 16; ++d) {
            vd = (double)nca * vd + (double)ncb * vq;
            vq = vq + vd;
            phi[d] = dead ? 0.f : (float)vq;
        }
    }
    // state, unscaled (the arrays hold scale x state, kernels_iir.hip "scaled state")
    f2 x;
    {
        const float s0 = (p_ss + ubase)[ul];
        x.x = (p_sq + ubase)[ul] / s0;
        x.y = (p_sd + ubase)[ul] / s0;
    }
    float t;
    {
        const int row0 = p_xfer_init[obj];
        const float tr = row0 >= 0 ? (float)(p_xfer_rows + (size_t)row0 * p.m_pad + team.col0)[ul] : 1e7f;
        t = dead ? 1.f : tr;
    }
    const BufDesc *__restrict__ dsc = p_desc + (size_t)obj * p.nb;
    float *__restrict__ aout = team.part_row >= 0 ? p_audio_parts + (size_t)team.part_row * p.audio_stride
                                                  : p_audio + (size_t)obj * p.audio_stride;
    float *__restrict__ b_qn = p_qnorm + ((size_t)obj * p.qn_nb + p.qn_b0) * p.m_pad + team.col0;
    const int B = p.frames;

    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto coarse = [&](f2 v) {                        // v <- P v
        const float qa = fmaf(p11, v.x, v.x);
        const float da = p21 * v.x;
        return f2{fmaf(p12, v.y, qa), fmaf(p22, v.y, da)};
    };

    // A buffer's inputs from memory -- its descriptor, the force gain row, a new transfer row -- are fetched while the
    // previous buffer runs: three dependent round trips to L2 / HBM (~1 us) would otherwise be half of a buffer's time.
    float g_next = 0.f, t_next = 0.f;
    auto fetch_rows = [&](const BufDesc &nd) {
        if (nd.frow >= 0) g_next = (p_grows + (size_t)nd.frow * p.m_pad + team.col0)[ul];
        if (nd.trow >= 0) t_next = (float)(p_xfer_rows + (size_t)nd.trow * p.m_pad + team.col0)[ul];
    };
    auto prefetch = [&](const BufDesc &nd) { fetch_rows(nd); };
    BufDesc next = dsc[0];
    prefetch(next);
    // diagnostics (PBSO_CENSUS=1): where the two waves' shader cycles go -- words 0..5 wave 0, 6..11 wave 1:
    // head + taps | first stepping phase | wait at A | second phase (stepping / free response / projection of group 0) | wait at B | projection
    unsigned long long cy[6] = {0, 0, 0, 0, 0, 0}, cy_mark = 0;
    auto lap = [&](int k) {
        if (p_census) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            cy[k] += now - cy_mark;
            cy_mark = now;
        }
    };
    if (p_census) cy_mark = __builtin_amdgcn_s_memtime();
    int par = 0;                                     // staging parity: flips with every buffer that is stepped (skipped ones pass no barrier)
    for (int b = 0; b < p.nb; ++b) {
        const BufDesc cur = next;
        next = dsc[b + 1 < p.nb ? b + 1 : b];
        const float g_cur = g_next, t_cur = t_next;
        float *__restrict__ ao = aout + (size_t)b * B;
        if (cur.flags & DESC_SKIP) {
            // the reference's step() returned before stepping: no samples, state untouched (no barrier in this buffer: both waves skip)
            for (int i = threadIdx.x; i < B; i += 128) ao[i] = 0.f;
            if (QN && grp == 0) (b_qn + (size_t)b * p.m_pad)[ul] = 0.f;
            prefetch(next);
            continue;
        }
        if (cur.trow != XFER_KEEP) {
            const float tr = cur.trow >= 0 ? t_cur : 1e7f;
            t = dead ? 1.f : tr;
        }
        const int frow = cur.frow;
        const float g = frow >= 0 ? g_cur : 0.f;
        const bool impulse = (cur.flags & DESC_IMPULSE) != 0;
        const bool dense = frow >= 0 && !impulse;
        par ^= 1;
        float *stage0 = lds_stage[par][0], *stage1 = lds_stage[par][1];
        float *stage = grp == 0 ? stage0 : stage1;
        float *taps = stage + BN * ST_ROW;
        const float *__restrict__ tprow = p_tprof + (size_t)(cur.prow >= 0 && dense ? cur.prow : 0) * p.b_pad;
        auto park = [&](float *st, int n, f2 v) { *reinterpret_cast<f2 *>(st + n * ST_ROW + 2 * lane) = f2{t * v.x, t * v.y}; };
        const float *bsrc = stage + (lane & 15) * ST_ROW + 2 * (lane >> 5) + ((lane >> 4) & 1);
        float breg[32];

        // per-sample stepping of 16 blocks under the dense profile, parking every block-start state; T values: one
        // s_load_dwordx16 per block issued a block ahead (scalar loads return out of order: first use, then the next load)
        auto step_group = [&](float *st, int first_sample, f2 &v, float &qacc) {
            float ta[BJ], tb[BJ];
            auto load_t = [&](float (&dst)[BJ], int n) {
                const float *__restrict__ tk = tprow + first_sample + BJ * n;
#pragma unroll
                for (int k = 0; k < BJ; ++k) dst[k] = tk[k];
            };
            // Unit-force form (as K1b): with z = v / g the forcing term is the profile value itself -- an operand of the FMA, no
            // product g T per sample: four vector instructions per sample with the qnorm sum, two of them a dependent chain.
            // Floating point is scale-invariant but for its range: the wave takes this form when every lane's z stays well
            // inside it (a zero / tiny gain on some mode -- the dummy start message of a sustained contact -- takes the general form).
            const float gi = __builtin_amdgcn_rcpf(g);
            f2 w = f2{v.x * gi, v.y * gi};
            const bool z_ok = g != 0.f && fabsf(gi) < 0x1p100f && fabsf(w.x) < 0x1p50f && fabsf(w.y) < 0x1p50f;      // (NaN / inf fail)
            auto run = [&](auto unit_c) {
                constexpr bool unit = decltype(unit_c)::value;
                if (!unit) w = v;
                const float tsc = unit ? t * g : t;
                float qz = 0.f;
                auto samples = [&](const float (&tv)[BJ]) {
#pragma unroll
                    for (int k = 0; k < BJ; ++k) {
                        const float in = unit ? fmaf(nca, w.y, tv[k]) : fmaf(nca, w.y, g * tv[k]);
                        w.y = fmaf(ncb, w.x, in);
                        w.x = w.x + w.y;
                        if (QN) qz = fmaf(w.x, w.x, qz);
                    }
                };
                auto park_w = [&](int n) { *reinterpret_cast<f2 *>(st + n * ST_ROW + 2 * lane) = f2{tsc * w.x, tsc * w.y}; };
                load_t(ta, 0);
                for (int n = 0; n < BN; n += 2) {
                    park_w(n);
                    const float probe = ta[0];
                    asm volatile("" :: "s"(probe));
                    __builtin_amdgcn_sched_barrier(0);
                    load_t(tb, n + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    samples(ta);
                    park_w(n + 1);
                    const float probe2 = tb[0];
                    asm volatile("" :: "s"(probe2));
                    __builtin_amdgcn_sched_barrier(0);
                    load_t(ta, n + 2 < BN ? n + 2 : n + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    samples(tb);
                }
                v = unit ? f2{g * w.x, g * w.y} : w;
                if (QN) qacc = unit ? fmaf(g * g, qz, qacc) : qacc + qz;
            };
            if (__all(z_ok)) run(std::true_type{});
            else run(std::false_type{});
        };

        float fir_a[4] = {0.f, 0.f, 0.f, 0.f};
        // the profile of this wave's group as the FIR's B operand, B[i][n] = T[1 + 256 grp + 16 n + i] (fetched under the taps)
        float fir_b[4] = {0.f, 0.f, 0.f, 0.f};
        if (dense) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fir_b[kk] = tprow[1 + GROUP * grp + BJ * (lane & 15) + 4 * kk + (lane >> 4)];
        }
        // 16 blocks of this wave's group a block at a time: v_{n+1} = P v_n + g (F . T_n), parking every block-start state
        auto step_group_ft = [&](float *st, f2 &v) {
            if constexpr (!QN) {
                float *ua = lds_incr[grp];
                static_for<0, 4>([&](auto tc) {
                    constexpr int tl = decltype(tc)::value;
                    f4 dq = f4{0.f, 0.f, 0.f, 0.f}, dd = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_b[ks], fB[tl][0][ks], dq, 0, 0, 0);
                        dd = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_b[ks], fB[tl][1][ks], dd, 0, 0, 0);
                    }
                    float *dst = ua + (16 * tl + (lane & 15)) * U_ROW + 4 * (lane >> 4);       // D[block 4 (l >> 4) + v][mode 16 tl + (l & 15)]
                    *reinterpret_cast<f4 *>(dst) = dq;
                    *reinterpret_cast<f4 *>(dst + BN) = dd;
                });
                wave_sync();
                f4 uq[4], ud[4];
                {
                    const f4 *src = reinterpret_cast<const f4 *>(ua + lane * U_ROW);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { uq[i] = src[i]; ud[i] = src[4 + i]; }
                }
                static_for<0, BN>([&](auto nc) {
                    constexpr int n = decltype(nc)::value;
                    park(st, n, v);
                    const f2 w = coarse(v);
                    v = f2{fmaf(g, uq[n / 4][n % 4], w.x), fmaf(g, ud[n / 4][n % 4], w.y)};
                });
            }
        };
        // projection of group gi from its parked block-start states (+ the profile's FIR, operand fb): 256 samples, stored straight
        // from the MFMA's result registers
        auto project = [&](int gi, const float (&fb)[4]) {
            const float *bs = (gi == 0 ? stage0 : stage1) + (lane & 15) * ST_ROW + 2 * (lane >> 5) + ((lane >> 4) & 1);
            wave_sync();
#pragma unroll
            for (int s = 0; s < 32; ++s) breg[s] = bs[4 * s];
            f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = f4{0.f, 0.f, 0.f, 0.f};
            static_for<0, 32>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], breg[s], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], breg[s], acc0, 0, 0, 0);
            });
            if (dense) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (kk & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_a[kk], fb[kk], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_a[kk], fb[kk], acc0, 0, 0, 0);
                }
            }
            const f4 acc = acc0 + acc1;              // D[i = 4 (l >> 4) + v][n = l & 15] = sample 16 n + i of the group
            float *o = ao + 1 + GROUP * gi + 16 * (lane & 15) + 4 * (lane >> 4);
            o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
        };
        bool fetched = false;
        bool late_b = false;                         // wave 0, dense buffer without qnorm rows: barrier B comes after its projection
        if (!dense) {
            // ================= force-free buffer (or an impulse at sample 0) =================
            lap(0);
            if (grp == 0) {
                const bool hit0 = frow >= 0 && (cur.tile_mask & 1u);
                float a = nca * x.y;
                a = fmaf(ncb, x.x, a);
                if (hit0) a = fmaf(g, cur.amp, a);
                x.y = a;
                x.x = x.x + a;
                if (QN) {
                    // sum_{k=0}^{B-1} q_k^2 = x0' G x0, x0 = state after sample 0 (the rest of the buffer is force-free)
                    float e = g22 * x.y * x.y;
                    e = fmaf(g12 * x.x, x.y, e);
                    e = fmaf(g11 * x.x, x.x, e);
                    (b_qn + (size_t)b * p.m_pad)[ul] = __builtin_amdgcn_sqrtf(fmaxf(e, 0.f));
                }
                const float p0 = wave_sum(t * x.x);
                if (lane == 0) ao[0] = p0;
#pragma unroll
                for (int n = 0; n < BN; ++n) {
                    park(stage0, n, x);
                    x = coarse(x);
                }
#pragma unroll
                for (int n = 0; n < BN; ++n) {
                    park(stage1, n, x);
                    x = coarse(x);
                }
            }
            // (handing the state over after sample 0 and letting wave 1 jump over group 0 with P^16 -- 17 coarse steps on the
            //  critical path instead of 32 -- ran 17 % SLOWER: 0.173 against 0.148 ms for 86 buffers of 1 x 512 modes; a hand-over
            //  through LDS and a workgroup barrier costs more than sixteen coarse steps)
            lap(1);
            __syncthreads();                         // A: both groups' block-start states are parked
            lap(2);
            __syncthreads();                         // B (every buffer passes both barriers)
            lap(4);
        } else {
            // ================= dense force profile =================
            // FIR taps of this wave (both waves need them: the forced response inside a block, see kernels_block.hip):
            // h_0 = sum t g, h_1 .. h_16 = projection of the virtual block-start state t g u, u = (1, 1)'
            if (!QN || grp == 1) {
                // (scratch and taps behind this wave's staging rows: in a qnorm build wave 0 parks into BOTH groups' staging areas
                //  while wave 1 -- the only one that projects dense buffers there -- is here)
                const float tg = t * g;
                float pv[16];
#pragma unroll
                for (int d = 0; d < 16; ++d) pv[d] = tg * phi[d];
                const float hd = wave_sum16(pv, lane, taps + 32, wave_sync);
                wave_sync();                             // (the scratch reads are done before anything parks there again)
                if (lane < 16) taps[taps_index(lane)] = hd;
                wave_sync();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int idx = (lane & 15) - 4 * kk - (lane >> 4);
                    fir_a[kk] = idx >= 0 ? taps[idx] : 0.f;
                }
            }
            lap(0);
            float qacc = 0.f;
            if (grp == 0) {
                // sample 0
                float a = nca * x.y;
                a = fmaf(ncb, x.x, a);
                a = fmaf(g, tprow[0], a);
                x.y = a;
                x.x = x.x + a;
                if (QN) qacc = x.x * x.x;
                const float p0 = wave_sum(t * x.x);
                if (lane == 0) ao[0] = p0;
                if (ft) step_group_ft(stage0, x);
                else step_group(stage0, 1, x, qacc);
                if (QN) {
                    // qnorm rows: the true state of every sample is needed -- this wave steps the second group as well, and wave 1
                    // projects BOTH groups: group 0 while this wave steps group 1, group 1 while this wave is already in the next
                    // buffer (the staging areas alternate with the buffer parity).  The buffer costs the stepping alone.
                    lap(1);
                    __syncthreads();                 // A: group 0's block-start states are parked
                    lap(2);
                    prefetch(next);                  // (this wave has no projection to fetch the next buffer's rows under)
                    fetched = true;
                    step_group(stage1, 1 + GROUP, x, qacc);
                    (b_qn + (size_t)b * p.m_pad)[ul] = sqrtf(qacc);
                    lap(3);
                    __syncthreads();                 // B: group 1's
                    lap(4);
                } else {
                    hand16[lane] = x;
                    late_b = true;
                }
            } else if (!QN) {
                // group 1 from a zero state under the second half of the profile (its response to the force alone)
                f2 z = f2{0.f, 0.f};
                if (ft) step_group_ft(stage1, z);
                else step_group(stage1, 1 + GROUP, z, qacc);
                lap(1);
                __syncthreads();                     // A: wave 0 has reached the group boundary
                lap(2);
                // + the free response of the state wave 0 handed over: 16 coarse steps, added to the parked states
                f2 xf = hand16[lane];
#pragma unroll
                for (int n = 0; n < BN; ++n) {
                    f2 *slot = reinterpret_cast<f2 *>(stage1 + n * ST_ROW + 2 * lane);
                    const f2 cur_v = *slot;
                    *slot = f2{fmaf(t, xf.x, cur_v.x), fmaf(t, xf.y, cur_v.y)};
                    xf = coarse(xf);
                }
                hand32[lane] = f2{z.x + xf.x, z.y + xf.y};
                lap(3);
                __syncthreads();                     // B: the buffer's end state is handed back
                lap(4);
            }
            if (QN && grp == 1) {
                float fir_b0[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fir_b0[kk] = tprow[1 + BJ * (lane & 15) + 4 * kk + (lane >> 4)];
                lap(1);
                __syncthreads();                     // A
                lap(2);
                project(0, fir_b0);
                lap(3);
                __syncthreads();                     // B
                lap(4);
            }
            if (!QN && grp == 0) { lap(1); __syncthreads(); lap(2); }    // A (B comes after this wave's projection: late_b)
        }

        // ---- projection of this wave's group: 32 MFMAs over the 32 pairs of columns (+ the profile's FIR)
        if (!fetched) prefetch(next);                // (the next descriptor has long arrived: its rows are fetched under the MFMAs)
        if (!(QN && dense && grp == 0)) project(grp, fir_b);
        if (late_b) {
            lap(5);
            __syncthreads();                         // B: wave 1 has added the free response and hands the end state back
            lap(4);
            x = hand32[lane];
        }
        if (!late_b) lap(5);
    }

    if (p_census && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p_census[(size_t)blockIdx.x * CENSUS_WORDS + 6 * grp + k] = cy[k];
    }
    if (grp == 0) {
        (p_sq + ubase)[ul] = x.x;
        (p_sd + ubase)[ul] = x.y;
        (p_ss + ubase)[ul] = 1.f;
    }
}

int launch_iir_split(const IirParams &p, int n_teams, int qnorm_mode, hipStream_t stream) {
    if (n_teams <= 0) return 0;
    if (p.frames != 1 + 2 * GROUP) return (int)hipErrorInvalidValue;
    const SplitDims dims = {p.nb, p.m_pad, p.b_pad, p.frames, p.audio_stride, p.gq_plane, p.qn_nb, p.qn_b0};
    if (qnorm_mode != 0)
        hipLaunchKernelGGL(iir_split_kernel<2>, dim3(n_teams), dim3(128), 0, stream, p.ca, p.cb, p.sq, p.sd, p.ss, p.desc, p.grows, p.tprof,
                           p.xfer_rows, p.xfer_init, p.audio, p.qnorm, p.gq, p.pc, p.wtab, p.teams, p.audio_parts, p.ftab, p.census, dims);
    else
        hipLaunchKernelGGL(iir_split_kernel<0>, dim3(n_teams), dim3(128), 0, stream, p.ca, p.cb, p.sq, p.sd, p.ss, p.desc, p.grows, p.tprof,
                           p.xfer_rows, p.xfer_init, p.audio, p.qnorm, p.gq, p.pc, p.wtab, p.teams, p.audio_parts, p.ftab, p.census, dims);
    return (int)hipGetLastError();
}

}  // namespace iir_split
}  // namespace pbso
