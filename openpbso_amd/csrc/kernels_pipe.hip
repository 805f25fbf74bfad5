// K1p: the block state-space oscillator bank for an UNDER-FILLED chip as a PIPELINE of waves (gfx950, wave64, f32 MFMA).
//
// Replaces the same reference code as K1 / K1b: the hot loop of ModalSolver::step (modal_solver.h:262-272) around
// ModalIntegrator::Step (modal_integrator.h:103-113).  Same formulation as kernels_block.hip (read that first): a buffer
// is sample 0 + 2 groups of 16 blocks of 16 samples; block-start states are parked in LDS and projected on the f32
// matrix pipe with the per-mode table W = (a_j, b_j); a dense force profile adds a 16-tap FIR of the profile.
//
// A scene with fewer waves of oscillators than SIMDs is bound by the LATENCY of one buffer in one wave, and a wave of K1b
// alternates between stepping and projecting.  Here the two kinds of work never share a wave: a team of 1 + NC waves owns
// 64 modes,
//   wave 0          the PRODUCER: steps buffer b -- sample 0, the 32 coarse steps (a dense profile: a block at a time
//                   with the increments F . T on the matrix pipe, or every sample when qnorm rows are asked for) -- from
//                   the true state, parks the 32 block-start states in the staging area of parity b & 1, writes sample 0
//                   and the qnorm row;
//   waves 1 .. NC   the CONSUMERS: project buffer b - 1 from the staging area of the other parity (NC = 2: one group each;
//                   NC = 1: both groups), add the profile's FIR, store 256 samples per group straight from the MFMA's
//                   result registers.
// One workgroup barrier per buffer; the buffer costs max(stepping, projection) instead of their sum, and the state never
// changes hands.  The registers hold the state UNSCALED (the parked values carry the transfer weight).
#include <type_traits>

#include "kernels.h"
#include "wave_ops.h"

namespace pbso {
namespace iir_pipe {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int BJ = BLOCK_J, BN = BLOCK_N, GROUP = BJ * BN;
constexpr int ST_ROW = 130;                          // as K1b: [16 blocks][64 lanes][Q, D], row stride 130 floats
constexpr int ST_AREA = BN * ST_ROW + 32 + 132;      // + a consumer's FIR taps h_0 .. h_15 + 64 floats of scratch for the taps' butterfly
constexpr int U_ROW = 36;                            // increments: [64 modes][16 Q | 16 D] + 4 floats of padding (as K1b's FTM)

struct PipeDims {
    int nb, m_pad, b_pad, frames;
    long long audio_stride, plane;
    int qn_nb, qn_b0;
    unsigned long long *start_flag;     // see IirParams::start_flag
    unsigned long long start_seq;
};

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(0.f, v);
    v += dpp_mov<0x4E>(0.f, v);
    v += dpp_mov<0x141>(0.f, v);
    v += dpp_mov<0x140>(0.f, v);
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
__global__ __launch_bounds__(256) void iir_pipe_kernel(
    const float *__restrict__ p_ca, const float *__restrict__ p_cb, float *__restrict__ p_sq, float *__restrict__ p_sd,
    float *__restrict__ p_ss, const BufDesc *__restrict__ p_desc, const float *__restrict__ p_grows,
    const float *__restrict__ p_tprof, const double *__restrict__ p_xfer_rows, const int *__restrict__ p_xfer_init,
    float *__restrict__ p_audio, float *__restrict__ p_qnorm, const float *__restrict__ p_gq, const float *__restrict__ p_pc,
    const float *__restrict__ p_wtab, const TeamDesc *__restrict__ p_teams, float *__restrict__ p_audio_parts,
    const float *__restrict__ p_ftab, unsigned long long *__restrict__ p_census, const float *__restrict__ p_g32,
    const long long *__restrict__ p_g32_off, const PipeDims p) {
    constexpr bool QN = QNM != 0;
    if (p.start_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(p.start_flag, p.start_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ __attribute__((aligned(16))) float lds_stage[2][2][ST_AREA];       // [buffer parity][group]
    __shared__ __attribute__((aligned(16))) float lds_incr[64 * U_ROW];           // the producer's increments on their way back to lane = mode
    // qnorm rows of dense buffers (block path): the UNWEIGHTED state at the start of every fourth block (0, 4, 8, 12 | 16, 20, 24, 28), from
    // which the consumers re-step the samples for the sum of q^2 only; and consumer 1's half of the sum on its way to consumer 0
    __shared__ f2 lds_raw[QN ? 2 : 1][8][64];
    __shared__ float lds_qsum[QN ? 2 : 1][4][64];    // [parity][group, pair of chains]: the same four partial sums whoever steps them
    __shared__ float lds_taps[QN ? 2 : 1][16];       // (qnorm rows by the consumers: they are the longer stage there, and the producer computes the FIR taps)
    // a dense buffer's profile for the producer's per-sample loop: [buffer parity][T_1 .. T_512 | T_0], staged by consumer 0 a buffer
    // ahead (the loop's 16 values per block come from LDS in ~100 cycles whatever the memory system is busy with; one scalar
    // load per block from L2 / HBM left 100 .. 250 of every block's 290 cycles waiting)
    __shared__ __attribute__((aligned(16))) float lds_t[2][2 * GROUP + 8];
    const TeamDesc team = p_teams[blockIdx.x];
    const int obj = team.obj;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NC = (int)(blockDim.x >> 6) - 1;                                    // consumers: 1 or 2; 3: two that project + a helper for the qnorm chains
    // (A CU holds two teams -- six waves on four SIMDs -- and the dispatcher deals waves to the SIMDs in a fixed cyclic order that
    //  continues from one workgroup to the next: one SIMD ends up with two consumers, one with a lone producer
    //  (scripts/debug/census_placement.py).  Letting the other team of a CU put its producer last gives every SIMD at most one
    //  consumer -- verified with the same census -- and is SLOWER: 8 x 4096 scraping 3220 -> 2890 x without qnorm rows (the
    //  producer, the longer role there, now always shares its SIMD), 1980 -> 1940 x with them.  Not kept.)
    const bool producer = wave == 0;
    const size_t ubase = (size_t)obj * p.m_pad + team.col0;
    const unsigned ul = (unsigned)lane;
    const BufDesc *__restrict__ dsc = p_desc + (size_t)obj * p.nb;
    float *__restrict__ aout = team.part_row >= 0 ? p_audio_parts + (size_t)team.part_row * p.audio_stride
                                                  : p_audio + (size_t)obj * p.audio_stride;
    const int B = p.frames;
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // diagnostics (PBSO_CENSUS=1): words 0..5 the producer (head | stepping | wait), 6..11 consumer 0 (head + taps | projection | wait)
    unsigned long long cy[3] = {0, 0, 0}, cy_mark = 0;
    auto lap = [&](int k) {
        if (p_census) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            cy[k] += now - cy_mark;
            cy_mark = now;
        }
    };
    const float nca = (p_ca + ubase)[ul], ncb = (p_cb + ubase)[ul];                // eps^2, -e  (velocity form, as K1)
    const bool dead = nca == 0.f && ncb == 0.f;                                   // a padding column
    // the transfer weight in force (both roles follow it: the producer parks t x state, a consumer's taps are sums of t g phi)
    float t;
    {
        const int row0 = p_xfer_init[obj];
        const float tr = row0 >= 0 ? (float)(p_xfer_rows + (size_t)row0 * p.m_pad + team.col0)[ul] : 1e7f;
        t = dead ? 1.f : tr;
    }
    // a buffer's rows from memory -- force gain, new transfer weights -- are fetched a buffer ahead
    float g_next = 0.f, t_next = 0.f;
    const float *__restrict__ g32_obj = p_g32 + (size_t)p_g32_off[obj] * p.m_pad + team.col0;
    auto fetch_rows = [&](const BufDesc &nd) {
        if (nd.frow >= 0 && (nd.flags & DESC_DIRECT)) {
            // the hit of a plain PointForce at a vertex (kernels.h): g = n . (three rows of the object's (float)(c3 * shape) table)
            const float *__restrict__ r0 = g32_obj + (size_t)nd.frow * p.m_pad;
            float gv = __builtin_bit_cast(float, nd.prow) * r0[ul];
            gv = fmaf(__builtin_bit_cast(float, nd.tile_mask), (r0 + p.m_pad)[ul], gv);
            g_next = fmaf(__builtin_bit_cast(float, nd.pad[0]), (r0 + 2 * (size_t)p.m_pad)[ul], gv);
        } else if (nd.frow >= 0) g_next = (p_grows + (size_t)nd.frow * p.m_pad + team.col0)[ul];
        if (nd.trow >= 0) t_next = (float)(p_xfer_rows + (size_t)nd.trow * p.m_pad + team.col0)[ul];
    };
    BufDesc next = dsc[0];
    fetch_rows(next);
    if (p_census) cy_mark = __builtin_amdgcn_s_memtime();

    auto is_dense = [](const BufDesc &d) { return !(d.flags & DESC_SKIP) && d.frow >= 0 && !(d.flags & DESC_IMPULSE) && d.prow >= 0; };
    auto stage_profile = [&](const BufDesc &d, int parity) {          // one wave: 2 KB, two 16-byte loads per lane
        if (!is_dense(d)) return;
        const float *__restrict__ row = p_tprof + (size_t)d.prow * p.b_pad;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = row[1 + 8 * lane + i];
        float *dst = lds_t[parity] + 8 * lane;
        *reinterpret_cast<f4 *>(dst) = f4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f4 *>(dst + 4) = f4{v[4], v[5], v[6], v[7]};
    };
    float *__restrict__ b_qn = p_qnorm + ((size_t)obj * p.qn_nb + p.qn_b0) * p.m_pad + team.col0;
    // qnorm rows of dense buffers come from the consumers when the producer takes the block path (the F table is there)
    const bool qn_by_cons = QN && p_ftab != nullptr;
    // FIR taps of a dense profile (kernels_block.hip, "forced block path"): h_d = sum over modes of t g phi_d, phi_d = e1' A^d u
    // the mode's response d samples after a unit force sample (u = (1, 1)': d += f, q += d).  phi: sixteen constants per
    // mode, stepped here once per launch (fp64 from the f32 coefficients the per-sample kernels use); the sixteen sums over
    // the wave are one butterfly (wave_ops.h).  The consumers compute them -- or, when they also re-step samples for qnorm rows
    // and are the longer stage, the producer does.
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
    if (wave == 1) stage_profile(next, 0);
    __syncthreads();
    if (producer) {
        // ================================================= PRODUCER =================================================
        const float p11 = (p_pc + ubase)[ul], p12 = (p_pc + p.plane + ubase)[ul];     // P = A^16: P11 - 1, P12, P21, P22
        const float p21 = (p_pc + 2 * p.plane + ubase)[ul], p22 = (p_pc + 3 * p.plane + ubase)[ul];
        float g11 = 0.f, g12 = 0.f, g22 = 0.f;
        if (QN) {
            g11 = (p_gq + ubase)[ul];
            g12 = (p_gq + p.plane + ubase)[ul];
            g22 = (p_gq + 2 * p.plane + ubase)[ul];
        }
        // Dense profiles without qnorm rows: a block's state increment is F . T_n, F = [A^15 u .. A u, u] (the table of K1b's
        // forced block path; absent -- PBSO_FORCED_BLOCK=0 -- every sample is stepped).  The increments of a group's 16 blocks
        // are a [16 blocks x 16 taps] . [16 taps x 16 modes] product per tile of 16 modes and state component: 32 MFMAs whose
        // A operand is the profile as the FIR's B operand holds it and whose B operand is F, resident here.
        const bool ft = p_ftab != nullptr;
        float fB[4][2][4];
        if (ft) {
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        fB[tl][c][ks] = (p_ftab + (size_t)(2 * (4 * ks + (lane >> 4)) + c) * p.plane + ubase)[16 * tl + (lane & 15)];
        }
        f2 x;                                        // state, unscaled (the arrays hold scale x state, kernels_iir.hip "scaled state")
        {
            const float s0 = (p_ss + ubase)[ul];
            x.x = (p_sq + ubase)[ul] / s0;
            x.y = (p_sd + ubase)[ul] / s0;
        }
        auto coarse = [&](f2 v) {                    // v <- P v
            const float qa = fmaf(p11, v.x, v.x);
            const float da = p21 * v.x;
            return f2{fmaf(p12, v.y, qa), fmaf(p22, v.y, da)};
        };
        for (int it = 0; it <= p.nb; ++it) {
            if (it < p.nb) {
                const int b = it;
                const BufDesc cur = next;
                next = dsc[b + 1 < p.nb ? b + 1 : b];
                const float g_cur = g_next, t_cur = t_next;
                fetch_rows(next);
                float *__restrict__ ao = aout + (size_t)b * B;
                if (cur.flags & DESC_SKIP) {
                    // the reference's step() returned before stepping: no samples (the consumers write the zeros), state untouched
                    if (QN) (b_qn + (size_t)b * p.m_pad)[ul] = 0.f;
                } else {
                    if (cur.trow != XFER_KEEP) {
                        const float tr = cur.trow >= 0 ? t_cur : 1e7f;
                        t = dead ? 1.f : tr;
                    }
                    const int frow = cur.frow;
                    const float g = frow >= 0 ? g_cur : 0.f;
                    const bool dense = frow >= 0 && !(cur.flags & DESC_IMPULSE);
                    float *stage0 = lds_stage[b & 1][0], *stage1 = lds_stage[b & 1][1];
                    const float *__restrict__ tprow = p_tprof + (size_t)(cur.prow >= 0 && dense ? cur.prow : 0) * p.b_pad;
                    auto park = [&](float *st, int n, f2 v) { *reinterpret_cast<f2 *>(st + n * ST_ROW + 2 * lane) = f2{t * v.x, t * v.y}; };
                    if (!dense) {
                        // ---- force-free buffer (or an impulse at sample 0)
                        const bool hit0 = frow >= 0 && ((cur.flags & DESC_DIRECT) || (cur.tile_mask & 1u));
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
                        lap(0);
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
                    } else {
                        // ---- dense force profile (Gaussian, AR: forces.h:92-128)
                        // the profile as an MFMA operand, lane l: T[1 + 256 grp + 16 (l & 15) + 4 kk + (l >> 4)] (fetched under sample 0)
                        float tb[2][4];
                        if (ft) {
#pragma unroll
                            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                                for (int kk = 0; kk < 4; ++kk) tb[gi][kk] = tprow[1 + GROUP * gi + BJ * (lane & 15) + 4 * kk + (lane >> 4)];
                        }
                        float qacc = 0.f;
                        {
                            float a = nca * x.y;
                            a = fmaf(ncb, x.x, a);
                            a = fmaf(g, tprow[0], a);
                            x.y = a;
                            x.x = x.x + a;
                            if (QN) qacc = x.x * x.x;
                            const float p0 = wave_sum(t * x.x);
                            if (lane == 0) ao[0] = p0;
                        }
                        lap(0);
                        if (QN && qn_by_cons) {
                            const float tg = t * g;
                            float pv[16];
#pragma unroll
                            for (int d = 0; d < 16; ++d) pv[d] = tg * phi[d];
                            const float hd = wave_sum16(pv, lane, lds_incr, wave_sync);      // (scratch: the increments' area, not yet in use)
                            if (lane < 16) lds_taps[QN ? (b & 1) : 0][taps_index(lane)] = hd;
                            wave_sync();
                        }
                        if (ft) {
                            // 16 blocks a block at a time: v_{n+1} = P v_n + g (F . T_n), parking every block-start state (with qnorm rows:
                            // also the unweighted state every 4 blocks -- the consumers re-step the samples from there for the sum of q^2)
                            auto step_group_ft = [&](float *st, const float (&tbg)[4], int gi) {
                                {
                                    static_for<0, 4>([&](auto tc) {
                                        constexpr int tl = decltype(tc)::value;
                                        f4 dq = f4{0.f, 0.f, 0.f, 0.f}, dd = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                        for (int ks = 0; ks < 4; ++ks) {
                                            dq = __builtin_amdgcn_mfma_f32_16x16x4f32(tbg[ks], fB[tl][0][ks], dq, 0, 0, 0);
                                            dd = __builtin_amdgcn_mfma_f32_16x16x4f32(tbg[ks], fB[tl][1][ks], dd, 0, 0, 0);
                                        }
                                        float *dst = lds_incr + (16 * tl + (lane & 15)) * U_ROW + 4 * (lane >> 4);   // D[block 4 (l >> 4) + v][mode 16 tl + (l & 15)]
                                        *reinterpret_cast<f4 *>(dst) = dq;
                                        *reinterpret_cast<f4 *>(dst + BN) = dd;
                                    });
                                    wave_sync();
                                    f4 uq[4], ud[4];
                                    {
                                        const f4 *src = reinterpret_cast<const f4 *>(lds_incr + lane * U_ROW);
#pragma unroll
                                        for (int i = 0; i < 4; ++i) { uq[i] = src[i]; ud[i] = src[4 + i]; }
                                    }
                                    wave_sync();                 // (the next group's tiles overwrite the area)
                                    static_for<0, BN>([&](auto nc) {
                                        constexpr int n = decltype(nc)::value;
                                        park(st, n, x);
                                        if constexpr (QN && n % 4 == 0) lds_raw[QN ? (b & 1) : 0][4 * gi + n / 4][lane] = x;
                                        const f2 w = coarse(x);
                                        x = f2{fmaf(g, uq[n / 4][n % 4], w.x), fmaf(g, ud[n / 4][n % 4], w.y)};
                                    });
                                }
                            };
                            step_group_ft(stage0, tb[0], 0);
                            step_group_ft(stage1, tb[1], 1);
                        } else {
                            // every sample (qnorm rows need each sample's true state).  T values: one s_load_dwordx16 per block
                            // issued a block ahead (scalar loads return out of order: first use, then the next load).  Unit-force
                            // form (as K1b): with z = x / g the forcing term is the profile value itself -- an operand of the FMA --
                            // while every lane's z stays well inside the floating-point range (a zero / tiny gain on some mode,
                            // e.g. the dummy start message of a sustained contact, takes the general form).
                            const float gi = __builtin_amdgcn_rcpf(g);
                            const f2 z0 = f2{x.x * gi, x.y * gi};
                            const bool z_ok = g != 0.f && fabsf(gi) < 0x1p100f && fabsf(z0.x) < 0x1p50f && fabsf(z0.y) < 0x1p50f;      // (NaN / inf fail)
                            auto run = [&](auto unit_c) {
                                constexpr bool unit = decltype(unit_c)::value;
                                f2 w = unit ? z0 : x;
                                const float tsc = unit ? t * g : t;
                                float qz = 0.f;
                                // 32 blocks of 16 samples; the profile values of a block: four broadcast ds_read_b128 (one address for
                                // the whole wave), three blocks in flight in rotating registers (LDS operations return in order: a
                                // block waits for ITS values only)
                                const float *tl = lds_t[b & 1] + (ul >> 31);          // (+ 0, opaque: vector registers)
                                f4 q0[4], q1[4], q2[4], q3[4];
                                auto ld = [&](f4 (&d)[4], int n) {
                                    const f4 *src = reinterpret_cast<const f4 *>(tl + BJ * (n < 2 * BN ? n : 2 * BN - 1));
#pragma unroll
                                    for (int i = 0; i < 4; ++i) d[i] = src[i];
                                };
                                auto runb = [&](const f4 (&d)[4], int n) {
                                    float *st = n < BN ? stage0 : stage1;
                                    *reinterpret_cast<f2 *>(st + (n & (BN - 1)) * ST_ROW + 2 * lane) = f2{tsc * w.x, tsc * w.y};
                                    const float tv[BJ] = {d[0].x, d[0].y, d[0].z, d[0].w, d[1].x, d[1].y, d[1].z, d[1].w,
                                                          d[2].x, d[2].y, d[2].z, d[2].w, d[3].x, d[3].y, d[3].z, d[3].w};
#pragma unroll
                                    for (int k = 0; k < BJ; ++k) {
                                        const float in = unit ? fmaf(nca, w.y, tv[k]) : fmaf(nca, w.y, g * tv[k]);
                                        w.y = fmaf(ncb, w.x, in);
                                        w.x = w.x + w.y;
                                        if (QN) qz = fmaf(w.x, w.x, qz);
                                    }
                                };
                                ld(q0, 0); ld(q1, 1); ld(q2, 2);
                                for (int n = 0; n < 2 * BN; n += 4) {
                                    ld(q3, n + 3);
                                    runb(q0, n);
                                    ld(q0, n + 4);
                                    runb(q1, n + 1);
                                    ld(q1, n + 5);
                                    runb(q2, n + 2);
                                    ld(q2, n + 6);
                                    runb(q3, n + 3);
                                }
                                x = unit ? f2{g * w.x, g * w.y} : w;
                                if (QN) qacc = unit ? fmaf(g * g, qz, qacc) : qacc + qz;
                            };
                            if (__all(z_ok)) run(std::true_type{});
                            else run(std::false_type{});
                            if (QN) (b_qn + (size_t)b * p.m_pad)[ul] = sqrtf(qacc);
                        }
                    }
                }
                lap(1);
            }
            __syncthreads();                         // buffer `it` is parked; the consumers have projected buffer it - 1
            lap(2);
        }
        (p_sq + ubase)[ul] = x.x;
        (p_sd + ubase)[ul] = x.y;
        (p_ss + ubase)[ul] = 1.f;
    } else {
        // ================================================= CONSUMERS =================================================
        const int cidx = wave - 1;
        float wreg[32];                              // the MFMA A operand of the 32 pairs of columns (as K1b)
        {
            const float *__restrict__ wsrc = p_wtab + (ubase / 2) * 64;
#pragma unroll
            for (int s = 0; s < 32; ++s) wreg[s] = wsrc[s * 64 + lane];
        }
        const int ctid = (int)threadIdx.x - 64, cthreads = 64 * NC;
        unsigned long long n_unit = 0;                // (census: groups whose qnorm chains took the unit-force form)
        int pend_b = -1;                             // consumer 0: the dense buffer whose four partial sums of q^2 are on their way
        auto finish_row = [&]() {                    // (after the barrier that followed the buffer's projection)
            if (QN && pend_b >= 0) {
                // (a fixed order: the row does not depend on how many waves shared the chains -- launches of one step may differ in that)
                const float (*ps)[64] = lds_qsum[QN ? (pend_b & 1) : 0];
                const float tot = (ps[0][lane] + ps[1][lane]) + (ps[2][lane] + ps[3][lane]);
                (b_qn + (size_t)pend_b * p.m_pad)[ul] = sqrtf(tot);
                pend_b = -1;
            }
        };
        for (int it = 0; it <= p.nb; ++it) {
            // (only a producer that steps every sample itself -- no F table -- wants the profile of ITS next buffer staged: the load
            //  of that descriptor, fetched under the projection and used after it, is not made otherwise)
            const bool stage_ahead = p_ftab == nullptr && cidx == 0;
            BufDesc ahead = next;
            if (stage_ahead) ahead = dsc[it + 1 < p.nb ? it + 1 : p.nb - 1];
            if (cidx == 0) finish_row();
            if (it >= 1) {
                const int b = it - 1;
                const BufDesc cur = next;
                next = dsc[b + 1 < p.nb ? b + 1 : b];
                const float g_cur = g_next, t_cur = t_next;
                fetch_rows(next);
                float *__restrict__ ao = aout + (size_t)b * B;
                if (cur.flags & DESC_SKIP) {
                    for (int i = ctid; i < B; i += cthreads) ao[i] = 0.f;
                } else {
                    if (cur.trow != XFER_KEEP) {
                        const float tr = cur.trow >= 0 ? t_cur : 1e7f;
                        t = dead ? 1.f : tr;
                    }
                    const int frow = cur.frow;
                    const float g = frow >= 0 ? g_cur : 0.f;
                    const bool dense = frow >= 0 && !(cur.flags & DESC_IMPULSE);
                    const float *__restrict__ tprow = p_tprof + (size_t)(cur.prow >= 0 && dense ? cur.prow : 0) * p.b_pad;
                    float fir_a[4] = {0.f, 0.f, 0.f, 0.f};
                    if (dense && QN && qn_by_cons) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const int idx = (lane & 15) - 4 * kk - (lane >> 4);
                            fir_a[kk] = idx >= 0 ? lds_taps[QN ? (b & 1) : 0][idx] : 0.f;
                        }
                    } else if (dense && cidx < 2) {
                        float *taps = lds_stage[b & 1][cidx] + BN * ST_ROW;       // (behind this consumer's first group's rows)
                        const float tg = t * g;
                        float pv[16];
#pragma unroll
                        for (int d = 0; d < 16; ++d) pv[d] = tg * phi[d];
                        const float hd = wave_sum16(pv, lane, taps + 32, wave_sync);
                        if (lane < 16) taps[taps_index(lane)] = hd;
                        wave_sync();
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const int idx = (lane & 15) - 4 * kk - (lane >> 4);
                            fir_a[kk] = idx >= 0 ? taps[idx] : 0.f;
                        }
                    }
                    if (QN && qn_by_cons && dense) {
                        // The sum of q^2 over the buffer's samples: eight chains of 64 samples (four per group), each from the unweighted
                        // state the producer left at its first block, stepped two at a time side by side (independent: their
                        // instructions fill each other's dependency stalls); profile values from LDS (four broadcast ds_read_b128 per
                        // chain and block, fetched a block ahead).  Unit-force form as everywhere (z = x / g) while the range allows.
                        // Two consumers: each its group's two pairs.  Three: the projecting consumers their group's first pair, the
                        // helper the second pair of both groups.
                        auto chain_pair = [&](int gi, int ca_) {          // chains ca_, ca_ + 1 of group gi -> partial sum 2 gi + ca_ / 2
                            float sq = 0.f;
                            const float *tl = lds_t[b & 1] + GROUP * gi + 64 * ca_ + (ul >> 31);      // (+ 0, opaque: vector registers)
                            const f2 xa = lds_raw[QN ? (b & 1) : 0][4 * gi + ca_][lane], xb = lds_raw[QN ? (b & 1) : 0][4 * gi + ca_ + 1][lane];
                            if (gi == 0 && ca_ == 0) sq = fmaf(xa.x, xa.x, sq);       // sample 0 left the state block 0 starts from
                            const float gr = __builtin_amdgcn_rcpf(g);
                            const bool z_ok = g != 0.f && fabsf(gr) < 0x1p100f && fabsf(xa.x * gr) < 0x1p50f && fabsf(xa.y * gr) < 0x1p50f &&
                                              fabsf(xb.x * gr) < 0x1p50f && fabsf(xb.y * gr) < 0x1p50f;      // (NaN / inf fail)
                            auto run = [&](auto unit_c) {
                                constexpr bool unit = decltype(unit_c)::value;
                                f2 wa = unit ? f2{xa.x * gr, xa.y * gr} : xa, wb = unit ? f2{xb.x * gr, xb.y * gr} : xb;
                                float qa = 0.f, qb = 0.f;
                                f4 ca[4], cb[4], na[4], nb4[4];
                                auto ld = [&](f4 (&da)[4], f4 (&db)[4], int blk) {
                                    const f4 *sa = reinterpret_cast<const f4 *>(tl + BJ * (blk < 4 ? blk : 3));
                                    const f4 *sb = reinterpret_cast<const f4 *>(tl + 64 + BJ * (blk < 4 ? blk : 3));
#pragma unroll
                                    for (int i = 0; i < 4; ++i) { da[i] = sa[i]; db[i] = sb[i]; }
                                };
                                auto stepb = [&](const f4 (&da)[4], const f4 (&db)[4]) {
                                    const float ta[BJ] = {da[0].x, da[0].y, da[0].z, da[0].w, da[1].x, da[1].y, da[1].z, da[1].w,
                                                          da[2].x, da[2].y, da[2].z, da[2].w, da[3].x, da[3].y, da[3].z, da[3].w};
                                    const float tb[BJ] = {db[0].x, db[0].y, db[0].z, db[0].w, db[1].x, db[1].y, db[1].z, db[1].w,
                                                          db[2].x, db[2].y, db[2].z, db[2].w, db[3].x, db[3].y, db[3].z, db[3].w};
#pragma unroll
                                    for (int k = 0; k < BJ; ++k) {
                                        const float ia = unit ? fmaf(nca, wa.y, ta[k]) : fmaf(nca, wa.y, g * ta[k]);
                                        const float ib = unit ? fmaf(nca, wb.y, tb[k]) : fmaf(nca, wb.y, g * tb[k]);
                                        wa.y = fmaf(ncb, wa.x, ia);
                                        wb.y = fmaf(ncb, wb.x, ib);
                                        wa.x = wa.x + wa.y;
                                        wb.x = wb.x + wb.y;
                                        qa = fmaf(wa.x, wa.x, qa);
                                        qb = fmaf(wb.x, wb.x, qb);
                                    }
                                };
                                ld(ca, cb, 0);
                                for (int blk = 0; blk < 4; blk += 2) {
                                    ld(na, nb4, blk + 1);
                                    stepb(ca, cb);
                                    ld(ca, cb, blk + 2);
                                    stepb(na, nb4);
                                }
                                sq += unit ? g * g * (qa + qb) : qa + qb;
                            };
                            if (__all(z_ok)) { run(std::true_type{}); n_unit += 1; }
                            else run(std::false_type{});
                            lds_qsum[QN ? (b & 1) : 0][2 * gi + ca_ / 2][lane] = sq;
                        };
                        if (NC == 3) {
                            if (cidx < 2) chain_pair(cidx, 0);
                            else { chain_pair(0, 2); chain_pair(1, 2); }
                        } else {
                            for (int gi = cidx; gi < 2; gi += NC) { chain_pair(gi, 0); chain_pair(gi, 2); }
                        }
                        if (cidx == 0) pend_b = b;
                    }
                    lap(0);
                    for (int gi = cidx; gi < 2; gi += NC) {
                        // projection of group gi from its parked block-start states (+ the profile's FIR): 256 samples, stored
                        // straight from the MFMA's result registers
                        const float *bs = lds_stage[b & 1][gi] + (lane & 15) * ST_ROW + 2 * (lane >> 5) + ((lane >> 4) & 1);
                        float breg[32];
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
                                const float fb = tprow[1 + GROUP * gi + BJ * (lane & 15) + 4 * kk + (lane >> 4)];
                                if (kk & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_a[kk], fb, acc1, 0, 0, 0);
                                else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_a[kk], fb, acc0, 0, 0, 0);
                            }
                        }
                        const f4 acc = acc0 + acc1;  // D[i = 4 (l >> 4) + v][n = l & 15] = sample 16 n + i of the group
                        float *o = ao + 1 + GROUP * gi + 16 * (lane & 15) + 4 * (lane >> 4);
                        o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
                    }
                }
                lap(1);
            }
            // the profile of the buffer the producer steps NEXT (it + 1), into the other parity's row -- or, when the consumers are
            // the ones that step samples (qnorm rows of dense buffers), of the buffer they take up next (it: the one being parked now)
            if (cidx == 0) {
                if (qn_by_cons) { if (it >= 1 && it < p.nb) stage_profile(next, it & 1); }
                else if (stage_ahead && it + 1 < p.nb) stage_profile(ahead, (it + 1) & 1);
            }
            __syncthreads();
            lap(2);
        }
        if (cidx == 0) finish_row();
        if (p_census && lane == 0 && cidx == 0) p_census[(size_t)blockIdx.x * CENSUS_WORDS + 11] = n_unit;
    }
    if (p_census && lane == 0) {                     // where the team's waves sit: HW_ID (SIMD 5:4, CU 11:8, SH 12, SE 15:13) | XCC_ID << 32
        const unsigned long long hw = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                      ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 32);
        // (words 3, 9, 10: the producer's and the two projecting consumers'; word 11 is consumer 0's n_unit: a helper wave keeps quiet)
        if (wave < 3) p_census[(size_t)blockIdx.x * CENSUS_WORDS + (wave == 0 ? 3 : 8 + wave)] = hw;
    }
    if (p_census && lane == 0 && wave < 2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) p_census[(size_t)blockIdx.x * CENSUS_WORDS + 6 * wave + k] = cy[k];
    }
}

// ---------------------------------------------------------------------------------------------------------
// K1p, five roles (round 5): launches that are mostly DENSE-profile buffers (sustained contact: forces.h:107-128,
// modal_solver.h:222-240).  In the three-wave team above the producer of such a launch is the long stage -- it evaluates the
// block increments F . T_n itself (64 MFMAs per buffer and their way back to "lane = mode") before it can take its 32 coarse
// steps (8 x 4096 scraping: producer 5.9 K cycles per buffer, consumers 4.8 K of which 1.2 K waiting).  Nothing in the
// increments depends on the state, so they move to waves of their own, a buffer AHEAD of the producer:
//   wave 0      P  steps buffer b: sample 0, then x <- P x + g U_n thirty-two times with U from LDS; parks t x, and the raw
//                  state every fourth block for the qnorm chains.  The only sequential part of the launch.
//   waves 1, 2  A, B  project buffer b - 1, one group of 256 samples each (32 MFMAs + the FIR's 4), store the samples.
//   waves 3, 4  C, D  the increments of buffer b + 1, one group each (32 MFMAs, tiles to LDS [64 modes][U_ROW]); C also the
//                  FIR taps of that buffer (sixteen wave sums in one butterfly).
//   qnorm rows of dense buffers: eight chains of 64 samples re-stepped from the raw states, one pair per wave A, B, C, D.
// One workgroup barrier per buffer.  The increments' area is single-buffered: P copies its row into registers first thing
// and says so in an LDS word; C / D, whose MFMAs come first, look at that word before they store their tiles.
constexpr int ST5_AREA = BN * ST_ROW;                // [16 blocks][64 lanes][Q, D], row stride 130 floats

template <int QNM>
__global__ __launch_bounds__(640) __attribute__((amdgpu_waves_per_eu(3, 3))) void iir_pipe5_kernel(
    const float *__restrict__ p_ca, const float *__restrict__ p_cb, float *__restrict__ p_sq, float *__restrict__ p_sd,
    float *__restrict__ p_ss, const BufDesc *__restrict__ p_desc, const float *__restrict__ p_grows,
    const float *__restrict__ p_tprof, const double *__restrict__ p_xfer_rows, const int *__restrict__ p_xfer_init,
    float *__restrict__ p_audio, float *__restrict__ p_qnorm, const float *__restrict__ p_gq, const float *__restrict__ p_pc,
    const float *__restrict__ p_wtab, const TeamDesc *__restrict__ p_teams, float *__restrict__ p_audio_parts,
    const float *__restrict__ p_ftab, unsigned long long *__restrict__ p_census, const float *__restrict__ p_g32,
    const long long *__restrict__ p_g32_off, const int n_teams, const PipeDims p) {
    constexpr bool QN = QNM != 0;
    if (p.start_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(p.start_flag, p.start_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // A workgroup holds TWO teams (slices of 64 modes), ten waves: the hardware places a workgroup only where EVERY SIMD has room for
    // ceil(waves / 4) of its waves, so two workgroups of five never share a CU at three waves per SIMD (this kernel's register budget
    // with qnorm rows) -- one workgroup of ten does.  Waves are dealt to the SIMDs in turn (w, w + 4, w + 8 share one:
    // scripts/debug/r05_pipe5_placement.py); in the order P0 A0 B0 C0 | P1 A1 B1 C1 | D0 D1 both P -- light, latency-bound -- share
    // their SIMD with ONE matrix wave.  The matrix work then sits 32 | 104 | 72 | 64 MFMAs per buffer on the four SIMDs and the second
    // one paces the kernel (5.2 K cycles per buffer); P0 C0 A1 B1 | P1 D0 B0 C1 | A0 D1 -- 36 | 96 | 72 | 68 -- was slower (5.6 K: the
    // increment waves carry more than their MFMAs).  An even split needs finer roles than one group per wave: not built.
    __shared__ __attribute__((aligned(16))) float lds_stage_[2][2][2][ST5_AREA];  // [slice][buffer parity][group]: parked t x
    __shared__ __attribute__((aligned(16))) float lds_u_[2][2][64 * U_ROW];       // [slice][group]: the increments of the buffer P steps next
    __shared__ f2 lds_raw_[2][QN ? 2 : 1][8][64];    // [slice][parity][every fourth block]: the unweighted state (unscaled buffers' qnorm chains)
    __shared__ float lds_qsum_[2][QN ? 2 : 1][4][64];// [slice][parity][wave A, B, C, D]: partial sums of q^2
    __shared__ float lds_taps_[2][4][16];            // [slice][buffer & 3]: FIR taps h_0 .. h_15 (bit-reversed order: taps_index)
    __shared__ float lds_scr_[2][64];                // the taps' butterfly
    __shared__ __attribute__((aligned(16))) float lds_t_[2][QN ? 4 : 1][2 * GROUP];   // [slice][buffer & 3]: T_1 .. T_512 for the qnorm chains
    __shared__ int lds_flag_[2];                     // buffers whose increments P has taken into registers
    const int wave_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifndef PBSO_PIPE5_ORDER
#define PBSO_PIPE5_ORDER 2
#endif
#if PBSO_PIPE5_ORDER == 1        // P0 A0 B0 C0 | P1 A1 B1 C1 | D0 D1
    const int slice = wave_wg < 8 ? wave_wg >> 2 : wave_wg - 8;
    const int wave = wave_wg < 8 ? (wave_wg & 3) : 4;                             // the role: 0 P, 1 A, 2 B, 3 C, 4 D
#else                            // P0 P1 B0 B1 | A1 A0 C0 C1 | D1 D0: every SIMD two matrix waves (A + D = B + C = 68 MFMAs), the P on top
    //                                  w:  0  1  2  3  4  5  6  7  8  9
    const int slice = (0x19Au >> wave_wg) & 1;                                    // 0  1  0  1  1  0  0  1  1  0
    const int wave = (int)((0x4433112200ull >> (4 * wave_wg)) & 15);             // 0  0  2  2  1  1  3  3  4  4
#endif
    float (*lds_stage)[2][ST5_AREA] = lds_stage_[slice];
    float (*lds_u)[64 * U_ROW] = lds_u_[slice];
    f2 (*lds_raw)[8][64] = lds_raw_[slice];
    float (*lds_qsum)[4][64] = lds_qsum_[slice];
    float (*lds_taps)[16] = lds_taps_[slice];
    float *lds_scr = lds_scr_[slice];
    float (*lds_t)[2 * GROUP] = lds_t_[slice];
    int &lds_flag = lds_flag_[slice];
    const int team_idx = 2 * (int)blockIdx.x + slice;
    if (team_idx >= n_teams) {                       // (an odd number of teams: the last workgroup's second half only keeps the barriers)
        __syncthreads();
        for (int it = 0; it <= p.nb + 1; ++it) __syncthreads();
        return;
    }
    const TeamDesc team = p_teams[team_idx];
    const int obj = team.obj;
    const int lane = threadIdx.x & 63;
    const size_t ubase = (size_t)obj * p.m_pad + team.col0;
    const unsigned ul = (unsigned)lane;
    const BufDesc *__restrict__ dsc = p_desc + (size_t)obj * p.nb;
    float *__restrict__ aout = team.part_row >= 0 ? p_audio_parts + (size_t)team.part_row * p.audio_stride
                                                  : p_audio + (size_t)obj * p.audio_stride;
    const int B = p.frames;
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // diagnostics (PBSO_CENSUS=1): words 0..2 P (head | stepping | wait), 6..8 A (head + chains | projection | wait), as the three-wave kernel
    unsigned long long cy[3] = {0, 0, 0}, cy_mark = 0;
    auto lap = [&](int k) {
        if (p_census) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            cy[k] += now - cy_mark;
            cy_mark = now;
        }
    };
    const float nca = (p_ca + ubase)[ul], ncb = (p_cb + ubase)[ul];                // eps^2, -e  (velocity form, as K1)
    const bool dead = nca == 0.f && ncb == 0.f;                                   // a padding column
    float t;                                         // the transfer weight in force (every role follows it for ITS buffer)
    {
        const int row0 = p_xfer_init[obj];
        const float tr = row0 >= 0 ? (float)(p_xfer_rows + (size_t)row0 * p.m_pad + team.col0)[ul] : 1e7f;
        t = dead ? 1.f : tr;
    }
    float g_next = 0.f, t_next = 0.f;
    const float *__restrict__ g32_obj = p_g32 + (size_t)p_g32_off[obj] * p.m_pad + team.col0;
    auto fetch_rows = [&](const BufDesc &nd) {
        if (nd.frow >= 0 && (nd.flags & DESC_DIRECT)) {
            const float *__restrict__ r0 = g32_obj + (size_t)nd.frow * p.m_pad;
            float gv = __builtin_bit_cast(float, nd.prow) * r0[ul];
            gv = fmaf(__builtin_bit_cast(float, nd.tile_mask), (r0 + p.m_pad)[ul], gv);
            g_next = fmaf(__builtin_bit_cast(float, nd.pad[0]), (r0 + 2 * (size_t)p.m_pad)[ul], gv);
        } else if (nd.frow >= 0) g_next = (p_grows + (size_t)nd.frow * p.m_pad + team.col0)[ul];
        if (nd.trow >= 0) t_next = (float)(p_xfer_rows + (size_t)nd.trow * p.m_pad + team.col0)[ul];
    };
    auto is_dense = [](const BufDesc &d) { return !(d.flags & DESC_SKIP) && d.frow >= 0 && !(d.flags & DESC_IMPULSE) && d.prow >= 0; };
    // Scaled state (as K1 / K1b): while every weight of the wave is usable P holds t x, so that a park is a plain store and a block
    // step six vector instructions (a lone wave beside matrix bursts issues one every ~10 cycles: the instruction COUNT of its 32
    // steps is the launch's critical path); C / D then deliver the increments times t g, the qnorm chains divide by t at the end.
    // Every role evaluates the same predicate on the same weights, so they agree without talking.
    auto usable = [](float v) { return v >= 0x1p-20f && v <= 0x1p40f; };
    BufDesc next = dsc[0];
    fetch_rows(next);
    float *__restrict__ b_qn = p_qnorm + ((size_t)obj * p.qn_nb + p.qn_b0) * p.m_pad + team.col0;
    if (wave == 0 && lane == 0) lds_flag = 0;
    if (p_census) cy_mark = __builtin_amdgcn_s_memtime();
    __syncthreads();

    // The sum of q^2 over 128 samples of a dense buffer: chains ca_, ca_ + 1 of group gi (64 samples each), re-stepped side by
    // side from the raw state P left at their first block; profile values from LDS (broadcast ds_read_b128, a block ahead);
    // unit-force form (z = x / g) while the range allows.  -> lds_qsum[b & 1][slot]
    auto chain_pair = [&](int b, float g, float inv_scale2, int gi, int ca_, int slot) {     // g: the gain as the parked state sees it (t g when scaled); the sum times inv_scale2
        __builtin_amdgcn_s_setprio(2);               // (a dependent chain: ahead of the other waves' matrix bursts, behind P)
        float sq = 0.f;
        const float *tl = lds_t[QN ? (b & 3) : 0] + GROUP * gi + 64 * ca_ + (ul >> 31);      // (+ 0, opaque: vector registers)
        const f2 xa = lds_raw[QN ? (b & 1) : 0][4 * gi + ca_][lane], xb = lds_raw[QN ? (b & 1) : 0][4 * gi + ca_ + 1][lane];
        if (gi == 0 && ca_ == 0) sq = fmaf(xa.x, xa.x, sq);       // sample 0 left the state block 0 starts from
        const float gr = __builtin_amdgcn_rcpf(g);
        const bool z_ok = g != 0.f && fabsf(gr) < 0x1p100f && fabsf(xa.x * gr) < 0x1p50f && fabsf(xa.y * gr) < 0x1p50f &&
                          fabsf(xb.x * gr) < 0x1p50f && fabsf(xb.y * gr) < 0x1p50f;      // (NaN / inf fail)
        auto run = [&](auto unit_c) {
            constexpr bool unit = decltype(unit_c)::value;
            f2 wa = unit ? f2{xa.x * gr, xa.y * gr} : xa, wb = unit ? f2{xb.x * gr, xb.y * gr} : xb;
            float qa = 0.f, qb = 0.f;
            f4 ca[4], cb[4], na[4], nb4[4];
            auto ld = [&](f4 (&da)[4], f4 (&db)[4], int blk) {
                const f4 *sa = reinterpret_cast<const f4 *>(tl + BJ * (blk < 4 ? blk : 3));
                const f4 *sb = reinterpret_cast<const f4 *>(tl + 64 + BJ * (blk < 4 ? blk : 3));
#pragma unroll
                for (int i = 0; i < 4; ++i) { da[i] = sa[i]; db[i] = sb[i]; }
            };
            auto stepb = [&](const f4 (&da)[4], const f4 (&db)[4]) {
                const float ta[BJ] = {da[0].x, da[0].y, da[0].z, da[0].w, da[1].x, da[1].y, da[1].z, da[1].w,
                                      da[2].x, da[2].y, da[2].z, da[2].w, da[3].x, da[3].y, da[3].z, da[3].w};
                const float tb[BJ] = {db[0].x, db[0].y, db[0].z, db[0].w, db[1].x, db[1].y, db[1].z, db[1].w,
                                      db[2].x, db[2].y, db[2].z, db[2].w, db[3].x, db[3].y, db[3].z, db[3].w};
#pragma unroll
                for (int k = 0; k < BJ; ++k) {
                    const float ia = unit ? fmaf(nca, wa.y, ta[k]) : fmaf(nca, wa.y, g * ta[k]);
                    const float ib = unit ? fmaf(nca, wb.y, tb[k]) : fmaf(nca, wb.y, g * tb[k]);
                    wa.y = fmaf(ncb, wa.x, ia);
                    wb.y = fmaf(ncb, wb.x, ib);
                    wa.x = wa.x + wa.y;
                    wb.x = wb.x + wb.y;
                    qa = fmaf(wa.x, wa.x, qa);
                    qb = fmaf(wb.x, wb.x, qb);
                }
            };
            ld(ca, cb, 0);
            for (int blk = 0; blk < 4; blk += 2) {
                ld(na, nb4, blk + 1);
                stepb(ca, cb);
                ld(ca, cb, blk + 2);
                stepb(na, nb4);
            }
            sq += unit ? g * g * (qa + qb) : qa + qb;
        };
        if (__all(z_ok)) run(std::true_type{});
        else run(std::false_type{});
        lds_qsum[QN ? (b & 1) : 0][slot][lane] = sq * inv_scale2;
        __builtin_amdgcn_s_setprio(0);
    };

    // Scaled buffers: the parked block-start states ARE the chains' starts (t x, the gain t g, the sum over t^2), so a wave takes its
    // 128 samples as FOUR chains of 32 side by side -- half the dependent levels of two chains of 64 (beside matrix bursts a level
    // costs ~35 cycles whatever it holds).  half: blocks 8 half .. 8 half + 7 of group gi.
    auto chain_quad = [&](int b, float g, float inv_scale2, int gi, int half, int slot) {
        __builtin_amdgcn_s_setprio(2);
        const float *tl = lds_t[QN ? (b & 3) : 0] + GROUP * gi + 128 * half + (ul >> 31);       // (+ 0, opaque: vector registers)
        const float *st = lds_stage[b & 1][gi] + (8 * half) * ST_ROW + 2 * lane;
        f2 z[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) z[c] = *reinterpret_cast<const f2 *>(st + 2 * c * ST_ROW);
        float sq0 = (gi == 0 && half == 0) ? z[0].x * z[0].x : 0.f;      // sample 0 left the state block 0 starts from
        const float gr = __builtin_amdgcn_rcpf(g);
        bool ok = g != 0.f && fabsf(gr) < 0x1p100f;
#pragma unroll
        for (int c = 0; c < 4; ++c) ok = ok && fabsf(z[c].x * gr) < 0x1p50f && fabsf(z[c].y * gr) < 0x1p50f;      // (NaN / inf fail)
        auto run = [&](auto unit_c) {
            constexpr bool unit = decltype(unit_c)::value;
            if (unit) {
#pragma unroll
                for (int c = 0; c < 4; ++c) z[c] = f2{z[c].x * gr, z[c].y * gr};
            }
            float q[4] = {0.f, 0.f, 0.f, 0.f};
            // eight samples of every chain at a time: two broadcast ds_read_b128 per chain, the next eight in flight
            f4 ta[4][2], tb[4][2];
            auto ld = [&](f4 (&d)[4][2], int k8) {
                const int kk = k8 < 4 ? k8 : 3;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f4 *src = reinterpret_cast<const f4 *>(tl + 32 * c + 8 * kk);
                    d[c][0] = src[0];
                    d[c][1] = src[1];
                }
            };
            auto step8 = [&](const f4 (&d)[4][2]) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float tk = d[c][k / 4][k % 4];
                        const float in = unit ? fmaf(nca, z[c].y, tk) : fmaf(nca, z[c].y, g * tk);
                        z[c].y = fmaf(ncb, z[c].x, in);
                        z[c].x = z[c].x + z[c].y;
                        q[c] = fmaf(z[c].x, z[c].x, q[c]);
                    }
                }
            };
            ld(ta, 0);
            ld(tb, 1);
            step8(ta);
            ld(ta, 2);
            step8(tb);
            ld(tb, 3);
            step8(ta);
            step8(tb);
            const float tot = (q[0] + q[1]) + (q[2] + q[3]);
            return unit ? g * g * tot : tot;
        };
        const float sq = sq0 + (__all(ok) ? run(std::true_type{}) : run(std::false_type{}));
        lds_qsum[QN ? (b & 1) : 0][slot][lane] = sq * inv_scale2;
        __builtin_amdgcn_s_setprio(0);
    };

    if (wave == 0) {
        // ================================================= P: the state =================================================
        const float p11 = (p_pc + ubase)[ul], p12 = (p_pc + p.plane + ubase)[ul];     // P = A^16: P11 - 1, P12, P21, P22
        const float p21 = (p_pc + 2 * p.plane + ubase)[ul], p22 = (p_pc + 3 * p.plane + ubase)[ul];
        float g11 = 0.f, g12 = 0.f, g22 = 0.f;
        if (QN) {
            g11 = (p_gq + ubase)[ul];
            g12 = (p_gq + p.plane + ubase)[ul];
            g22 = (p_gq + 2 * p.plane + ubase)[ul];
        }
        f2 x;                                        // state times ts (the arrays hold scale x state, kernels_iir.hip "scaled state")
        float ts = (p_ss + ubase)[ul];
        x.x = (p_sq + ubase)[ul];
        x.y = (p_sd + ubase)[ul];
        float t0_next = is_dense(next) ? (p_tprof + (size_t)next.prow * p.b_pad)[0] : 0.f;      // the next dense buffer's first profile sample
        // P's thirty-two dependent steps are the launch's only sequential part, and it shares its SIMD with waves that issue
        // matrix instructions back to back (each holds the f32 datapath for 32 cycles): whenever P has an instruction ready, it goes first
        __builtin_amdgcn_s_setprio(3);
        for (int it = 0; it <= p.nb + 1; ++it) {
            const int b = it - 1;
            if (b >= 0 && b < p.nb) {
                const BufDesc cur = next;
                next = dsc[b + 1 < p.nb ? b + 1 : b];
                const float g_cur = g_next, t_cur = t_next, t0_cur = t0_next;
                fetch_rows(next);
                t0_next = is_dense(next) ? (p_tprof + (size_t)next.prow * p.b_pad)[0] : 0.f;
                float *__restrict__ ao = aout + (size_t)b * B;
                const bool skip = (cur.flags & DESC_SKIP) != 0;
                const int frow = cur.frow;
                const bool dense = !skip && frow >= 0 && !(cur.flags & DESC_IMPULSE);
                // this buffer's increments, lane = mode: [group][16 Q | 16 D]; then C / D may store the next buffer's
                f4 uq[2][4], ud[2][4];
                if (dense) {
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi) {
                        const f4 *src = reinterpret_cast<const f4 *>(lds_u[gi] + lane * U_ROW);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { uq[gi][i] = src[i]; ud[gi][i] = src[4 + i]; }
                    }
                }
                __hip_atomic_store(&lds_flag, b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (skip) {
                    // the reference's step() returned before stepping: no samples (A / B write the zeros), state untouched
                    if (QN) (b_qn + (size_t)b * p.m_pad)[ul] = 0.f;
                } else {
                    if (cur.trow != XFER_KEEP) {
                        const float tr = cur.trow >= 0 ? t_cur : 1e7f;
                        t = dead ? 1.f : tr;
                    }
                    // the registers' scale follows the weight while the whole wave's weights are usable, 1 otherwise
                    const bool scaled = __all(usable(t));
                    {
                        const float tn = scaled ? t : 1.f;
                        if (!__all(tn == ts)) {
                            const float f = tn / ts;
                            x.x *= f;
                            x.y *= f;
                        }
                        ts = tn;
                    }
                    const float g = frow >= 0 ? g_cur * ts : 0.f;        // the gain as the registers see it
                    float *stage0 = lds_stage[b & 1][0], *stage1 = lds_stage[b & 1][1];
                    // sample 0, literal (velocity form): d = eps^2 d - e q + g T_0 ; q += d
                    {
                        const bool hit0 = !dense && frow >= 0 && ((cur.flags & DESC_DIRECT) || (cur.tile_mask & 1u));
                        float a = nca * x.y;
                        a = fmaf(ncb, x.x, a);
                        if (dense) a = fmaf(g, t0_cur, a);
                        else if (hit0) a = fmaf(g, cur.amp, a);
                        x.y = a;
                        x.x = x.x + a;
                    }
                    if (QN && !dense) {
                        // sum_{k=0}^{B-1} q_k^2 = x0' G x0, x0 = state after sample 0 (the rest of the buffer is force-free)
                        float e = g22 * x.y * x.y;
                        e = fmaf(g12 * x.x, x.y, e);
                        e = fmaf(g11 * x.x, x.x, e);
                        (b_qn + (size_t)b * p.m_pad)[ul] = __builtin_amdgcn_sqrtf(fmaxf(e, 0.f)) * __builtin_amdgcn_rcpf(ts);
                    }
                    const float p0 = wave_sum(scaled ? x.x : t * x.x);
                    if (lane == 0) ao[0] = p0;
                    lap(0);
                    // 32 blocks: park, x <- P x (+ U_n, which C / D deliver times the gain): six vector instructions per block.  (A wave
                    // alone issues one every ~5.5 cycles whatever their dependencies, so their NUMBER is P's length: stepping four
                    // blocks at a time with P^4 -- 16 dependent levels on the chain instead of 64, half as many instructions again
                    // for the states in between -- was no faster: 2.8 K cycles per buffer both ways beside a matrix wave.)
                    auto walk = [&](auto scaled_c, auto dense_c) {
                        constexpr bool SC = decltype(scaled_c)::value, DN = decltype(dense_c)::value;
                        static_for<0, 2>([&](auto gc) {
                            constexpr int gi = decltype(gc)::value;
                            float *st = gi == 0 ? stage0 : stage1;
                            static_for<0, BN>([&](auto nc) {
                                constexpr int n = decltype(nc)::value;
                                *reinterpret_cast<f2 *>(st + n * ST_ROW + 2 * lane) = SC ? x : f2{t * x.x, t * x.y};
                                if constexpr (QN && DN && !SC && n % 4 == 0) lds_raw[QN ? (b & 1) : 0][4 * gi + n / 4][lane] = x;
                                const float qa = fmaf(p11, x.x, x.x), qb = DN ? fmaf(p12, x.y, uq[gi][n / 4][n % 4]) : p12 * x.y;
                                const float da = p21 * x.x, db = DN ? fmaf(p22, x.y, ud[gi][n / 4][n % 4]) : p22 * x.y;
                                x.x = qa + qb;
                                x.y = da + db;
                            });
                        });
                    };
                    if (scaled) { if (dense) walk(std::true_type{}, std::true_type{}); else walk(std::true_type{}, std::false_type{}); }
                    else { if (dense) walk(std::false_type{}, std::true_type{}); else walk(std::false_type{}, std::false_type{}); }
                }
                lap(1);
            }
            __syncthreads();
            lap(2);
        }
        (p_sq + ubase)[ul] = x.x;
        (p_sd + ubase)[ul] = x.y;
        (p_ss + ubase)[ul] = ts;
    } else if (wave <= 2) {
        // ================================================= A, B: the samples =================================================
        const int gi = wave - 1;
        float wreg[32];                              // the MFMA A operand of the 32 pairs of columns (as K1b)
        {
            const float *__restrict__ wsrc = p_wtab + (ubase / 2) * 64;
#pragma unroll
            for (int s = 0; s < 32; ++s) wreg[s] = wsrc[s * 64 + lane];
        }
        const int ctid = 64 * gi + lane, cthreads = 128;
        int pend_b = -1;                             // A: the dense buffer whose four partial sums of q^2 are on their way
        auto finish_row = [&]() {                    // (after the barrier that followed the buffer's chains)
            if (QN && pend_b >= 0) {
                const float (*ps)[64] = lds_qsum[QN ? (pend_b & 1) : 0];
                const float tot = (ps[0][lane] + ps[1][lane]) + (ps[2][lane] + ps[3][lane]);
                (b_qn + (size_t)pend_b * p.m_pad)[ul] = sqrtf(tot);
                pend_b = -1;
            }
        };
        for (int it = 0; it <= p.nb + 1; ++it) {
            const int b = it - 2;
            if (gi == 0) finish_row();
            if (b >= 0 && b < p.nb) {
                const BufDesc cur = next;
                next = dsc[b + 1 < p.nb ? b + 1 : b];
                const float g_cur = g_next, t_cur = t_next;
                fetch_rows(next);
                float *__restrict__ ao = aout + (size_t)b * B;
                if (cur.flags & DESC_SKIP) {
                    for (int i = ctid; i < B; i += cthreads) ao[i] = 0.f;
                } else {
                    if (cur.trow != XFER_KEEP) {
                        const float tr = cur.trow >= 0 ? t_cur : 1e7f;
                        t = dead ? 1.f : tr;
                    }
                    const int frow = cur.frow;
                    const float g = frow >= 0 ? g_cur : 0.f;
                    const bool dense = frow >= 0 && !(cur.flags & DESC_IMPULSE);
                    float fir_a[4] = {0.f, 0.f, 0.f, 0.f}, fir_b[4] = {0.f, 0.f, 0.f, 0.f};
                    if (dense) {
                        // FIR operands: A[j][i] = h_{j-i} (i <= j) from C's taps, B = the profile (fetched here, under the projection)
                        const float *__restrict__ tprow = p_tprof + (size_t)(cur.prow >= 0 ? cur.prow : 0) * p.b_pad;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            fir_b[kk] = tprow[1 + GROUP * gi + BJ * (lane & 15) + 4 * kk + (lane >> 4)];
                            const int idx = (lane & 15) - 4 * kk - (lane >> 4);
                            fir_a[kk] = idx >= 0 ? lds_taps[b & 3][taps_index(idx)] : 0.f;
                        }
                    }
                    lap(0);
                    {
                        // projection of group gi from its parked block-start states (+ the profile's FIR): 256 samples, stored
                        // straight from the MFMA's result registers
                        const float *bs = lds_stage[b & 1][gi] + (lane & 15) * ST_ROW + 2 * (lane >> 5) + ((lane >> 4) & 1);
                        float breg[32];
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
                                if (kk & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_a[kk], fir_b[kk], acc1, 0, 0, 0);
                                else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_a[kk], fir_b[kk], acc0, 0, 0, 0);
                            }
                        }
                        const f4 acc = acc0 + acc1;  // D[i = 4 (l >> 4) + v][n = l & 15] = sample 16 n + i of the group
                        float *o = ao + 1 + GROUP * gi + 16 * (lane & 15) + 4 * (lane >> 4);
                        o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
                    }
                    lap(1);
                    if (QN && dense) {
                        const bool scaled = __all(usable(t));
                        const float rt = __builtin_amdgcn_rcpf(t);
                        // this group's first 128 samples (the other 128: C / D)
                        if (scaled) chain_quad(b, t * g, rt * rt, gi, 0, gi);
                        else chain_pair(b, g, 1.f, gi, 0, gi);
                        if (gi == 0) pend_b = b;
                    }
                }
                lap(0);
            }
            __syncthreads();
            lap(2);
        }
        if (gi == 0) finish_row();
    } else {
        // ================================================= C, D: the increments =================================================
        const int gi = wave - 3;
        float fB[4][2][4];                           // B operand: F[tap 4 ks + (l >> 4)][mode 16 tl + (l & 15)], both components
#pragma unroll
        for (int tl = 0; tl < 4; ++tl)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    fB[tl][c][ks] = (p_ftab + (size_t)(2 * (4 * ks + (lane >> 4)) + c) * p.plane + ubase)[16 * tl + (lane & 15)];
        // FIR taps of a dense profile: h_d = sum over modes of t g phi_d, phi_d = e1' A^d u (see the three-wave kernel)
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
        float g_h1 = 0.f, g_h2 = 0.f, r_h1 = 1.f, r_h2 = 1.f;      // gain and 1 / scale^2 of the two buffers before this one (qnorm chains run two behind)
        bool dense_h1 = false, dense_h2 = false, sc_h1 = false, sc_h2 = false;
        for (int it = 0; it <= p.nb + 1; ++it) {
            const int b = it;
            float g = 0.f, r2 = 1.f;
            bool dense = false, sc = false;
            if (b < p.nb) {
                const BufDesc cur = next;
                next = dsc[b + 1 < p.nb ? b + 1 : b];
                const float g_cur = g_next, t_cur = t_next;
                fetch_rows(next);
                const bool skip = (cur.flags & DESC_SKIP) != 0;
                if (!skip && cur.trow != XFER_KEEP) {
                    const float tr = cur.trow >= 0 ? t_cur : 1e7f;
                    t = dead ? 1.f : tr;
                }
                const float g_raw = cur.frow >= 0 ? g_cur : 0.f;
                const bool scaled = __all(usable(t));
                sc = scaled;
                g = scaled ? t * g_raw : g_raw;                          // the gain as P's registers (and the parked raw states) see it
                {
                    const float rt = __builtin_amdgcn_rcpf(t);
                    r2 = scaled ? rt * rt : 1.f;
                }
                dense = !skip && cur.frow >= 0 && !(cur.flags & DESC_IMPULSE);
                if (dense) {
                    const float *__restrict__ tprow = p_tprof + (size_t)(cur.prow >= 0 ? cur.prow : 0) * p.b_pad;
                    // the profile as the A operand, A[block l & 15][tap 4 ks + (l >> 4)] = T[1 + 256 gi + 16 block + tap]
                    float fa[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) fa[ks] = tprow[1 + GROUP * gi + BJ * (lane & 15) + 4 * ks + (lane >> 4)];
                    if (QN) {
                        // ... and in sample order for the qnorm chains, two iterations from now
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = tprow[1 + GROUP * gi + 4 * lane + i];
                        *reinterpret_cast<f4 *>(lds_t[QN ? (b & 3) : 0] + GROUP * gi + 4 * lane) = f4{v[0], v[1], v[2], v[3]};
                    }
                    f4 dq[4], dd[4];
                    static_for<0, 4>([&](auto tc) {
                        constexpr int tl = decltype(tc)::value;
                        dq[tl] = f4{0.f, 0.f, 0.f, 0.f};
                        dd[tl] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            dq[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks], fB[tl][0][ks], dq[tl], 0, 0, 0);
                            dd[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks], fB[tl][1][ks], dd[tl], 0, 0, 0);
                        }
                    });
                    // times the gain of the tile's modes (lane = mode -> the MFMA's column layout): P adds them as they are
                    float gs[4];
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) gs[tl] = __shfl(g, 16 * tl + (lane & 15), 64);
                    // (P has copied the previous buffer's increments into its registers)
                    while (__hip_atomic_load(&lds_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < b) __builtin_amdgcn_s_sleep(1);
                    static_for<0, 4>([&](auto tc) {
                        constexpr int tl = decltype(tc)::value;
                        float *dst = lds_u[gi] + (16 * tl + (lane & 15)) * U_ROW + 4 * (lane >> 4);   // D[block 4 (l >> 4) + v][mode 16 tl + (l & 15)]
                        *reinterpret_cast<f4 *>(dst) = dq[tl] * gs[tl];
                        *reinterpret_cast<f4 *>(dst + BN) = dd[tl] * gs[tl];
                    });
                    if (gi == 0) {
                        const float tg = t * g_raw;
                        float pv[16];
#pragma unroll
                        for (int d = 0; d < 16; ++d) pv[d] = tg * phi[d];
                        const float hd = wave_sum16(pv, lane, lds_scr, wave_sync);
                        if (lane < 16) lds_taps[b & 3][lane] = hd;       // (lane l holds the total of value taps_index(l): bit-reversed order)
                    }
                }
            }
            if (QN && dense_h2 && b >= 2) {          // the second 128 samples of buffer b - 2's group gi
                if (sc_h2) chain_quad(b - 2, g_h2, r_h2, gi, 1, 2 + gi);
                else chain_pair(b - 2, g_h2, r_h2, gi, 2, 2 + gi);
            }
            sc_h2 = sc_h1; sc_h1 = sc;
            g_h2 = g_h1; g_h1 = g;
            r_h2 = r_h1; r_h1 = r2;
            dense_h2 = dense_h1; dense_h1 = dense;
            __syncthreads();
        }
    }
    if (p_census && lane == 0) {                     // where the team's waves sit: HW_ID (SIMD 5:4, CU 11:8, SH 12, SE 15:13) | XCC_ID << 32
        const unsigned long long hw = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                      ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 32);
        if (wave < 3) p_census[(size_t)team_idx * CENSUS_WORDS + (wave == 0 ? 3 : 8 + wave)] = hw;
    }
    if (p_census && lane == 0 && wave < 2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) p_census[(size_t)team_idx * CENSUS_WORDS + 6 * wave + k] = cy[k];
    }
}

int launch_iir_pipe(const IirParams &p, int n_teams, int n_consumers, int qnorm_mode, hipStream_t stream) {
    if (n_teams <= 0) return 0;
    if (p.frames != 1 + 2 * GROUP || n_consumers < 1 || n_consumers > 4) return (int)hipErrorInvalidValue;
    if (n_consumers == 4 && p.ftab == nullptr) n_consumers = 3;                            // (the five-role team steps dense buffers by increments only)
    if (n_consumers == 3 && (qnorm_mode == 0 || p.ftab == nullptr)) n_consumers = 2;      // (the third only steps qnorm chains)
    const PipeDims dims = {p.nb, p.m_pad, p.b_pad, p.frames, p.audio_stride, p.gq_plane, p.qn_nb, p.qn_b0, p.start_flag, p.start_seq};
    const dim3 block(64 * (1 + n_consumers));
    if (n_consumers == 4) {
        const dim3 grid5((n_teams + 1) / 2), block5(640);        // two teams of five waves per workgroup
        if (qnorm_mode != 0)
            hipLaunchKernelGGL(iir_pipe5_kernel<2>, grid5, block5, 0, stream, p.ca, p.cb, p.sq, p.sd, p.ss, p.desc, p.grows, p.tprof,
                               p.xfer_rows, p.xfer_init, p.audio, p.qnorm, p.gq, p.pc, p.wtab, p.teams, p.audio_parts, p.ftab, p.census, p.g32, p.g32_off, n_teams, dims);
        else
            hipLaunchKernelGGL(iir_pipe5_kernel<0>, grid5, block5, 0, stream, p.ca, p.cb, p.sq, p.sd, p.ss, p.desc, p.grows, p.tprof,
                               p.xfer_rows, p.xfer_init, p.audio, p.qnorm, p.gq, p.pc, p.wtab, p.teams, p.audio_parts, p.ftab, p.census, p.g32, p.g32_off, n_teams, dims);
        return (int)hipGetLastError();
    }
    if (qnorm_mode != 0)
        hipLaunchKernelGGL(iir_pipe_kernel<2>, dim3(n_teams), block, 0, stream, p.ca, p.cb, p.sq, p.sd, p.ss, p.desc, p.grows, p.tprof,
                           p.xfer_rows, p.xfer_init, p.audio, p.qnorm, p.gq, p.pc, p.wtab, p.teams, p.audio_parts, p.ftab, p.census, p.g32, p.g32_off, dims);
    else
        hipLaunchKernelGGL(iir_pipe_kernel<0>, dim3(n_teams), block, 0, stream, p.ca, p.cb, p.sq, p.sd, p.ss, p.desc, p.grows, p.tprof,
                           p.xfer_rows, p.xfer_init, p.audio, p.qnorm, p.gq, p.pc, p.wtab, p.teams, p.audio_parts, p.ftab, p.census, p.g32, p.g32_off, dims);
    return (int)hipGetLastError();
}

}  // namespace iir_pipe
}  // namespace pbso
