// K1: damped-IIR oscillator bank + per-sample mode reduction (gfx950, wave64).
//
// Replaces the hot loop of ModalSolver::step (modal_solver.h:262-272) and
// ModalIntegrator::Step (modal_integrator.h:103-113) for a whole batch of
// objects and buffers in one launch.
//
// Mapping.  One workgroup = one "team" of W waves stepping the columns
// [col0, col0 + 64 W R) of one object's SoA rows (a whole object, or one part of
// an object that needs more waves than a team may have -- kernels.h TeamDesc).  A
// lane owns R oscillators: column = col0 + r * (64 W) + tid, so every r-slice is
// one contiguous, coalesced row.  Coefficients, state, the force gain, the
// transfer weights and the qnorm accumulators live in VGPRs for the whole
// launch; descriptors and force time profiles are uniform over the team and
// are read with scalar loads.  The loop is bound by the fp32 vector ALU issue
// rate, not by HBM: per oscillator-sample it issues
//   velocity form:  v_mul, v_fmac [, v_fmac force], v_add, [v_fmac qnorm,] and 1/R..1 op for the output
//   direct form:    v_mul [, v_fmac force], v_fmac, ...
// The registers hold the state multiplied by the transfer weight ("scaled state", see the
// kernel body), which turns the output into a plain sum over the lane's modes.
//
// Per-sample reduction over modes.  Each lane first sums its own R modes
// (p = sum_r t_r q_r; with the scaled state simply sum_r Q_r), then the 64 lane partials of TILE = 27 consecutive
// samples are transposed through a per-wave LDS tile P[27][68]: lane l writes
// P[k][l] while stepping sample k (ds_write_addtid_b32: no address VGPR, half
// the issue cost of ds_write_b32).  The row sums are taken ONE TILE LATER:
// at the start of the next tile lanes 2k and 2k+1 issue 8 ds_read_b128 each
// for the two halves of row k (bank-conflict free with the 68-float stride;
// LDS ops of a wave execute in order, so the reads see the old tile although
// the new tile's writes follow), and the 32 adds are spread over the next
// tile's sample bodies where they fill dependency stalls; one DPP add joins
// the halves.  ~1.2 VALU instructions per wave-sample, fixed order, no LDS
// latency exposed.  Row sums go to a small per-wave LDS ring; a team of W
// waves meets at a workgroup barrier only once per audio buffer, when all
// waves add their rings and store the buffer's 513 samples coalesced.
#include <type_traits>

#include "kernels.h"

// Built with -fno-slp-vectorize so that the instruction selection is the one written here (plain
// VOP2 v_mul / v_fmac / v_add).  A float2 variant on v_pk_*_f32 measured 2-3 % slower (the packed
// ops issue at half rate) and was dropped.

namespace pbso {
namespace iir_scalar {

typedef float f4 __attribute__((ext_vector_type(4)));

template <int K0, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        f(std::integral_constant<int, K0>{});
        static_for<K0 + 1, N - 1>(f);
    }
}

// FMODE: 0 force-free, 1 dense time profile tp[0..TILE), 2 impulse (amp at sample 0 only)
// SCALED: the state registers hold (transfer weight x state), so the lane's output is a
// plain sum of its modes (one add instead of a multiply + FMA per pair of modes).
template <int R, int FORM, bool QN, int FMODE, bool SCALED>
__device__ __forceinline__ void step_tile(float (&q)[R], float (&d)[R], const float (&ca)[R], const float (&cb)[R],
                                          const float (&g)[R], const float (&t)[R], float (&qn)[R],
                                          const float *__restrict__ tp, float amp, f4 (&rv)[8],
                                          float &rsum) {
    // dense profile: fetch the tile's 27 values into SGPRs up front (one wait)
    float tt[TILE];
    if (FMODE == 1) {
#pragma unroll
        for (int k = 0; k < TILE; ++k) tt[k] = tp[k];
    }
    static_for<0, TILE>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr bool forced = FMODE == 1 || (FMODE == 2 && k == 0);
        float tk = 0.f;
        if (FMODE == 1) tk = tt[k];
        if (FMODE == 2 && k == 0) tk = amp;
        float p = 0.f;
#pragma unroll
        for (int v = 0; v < R; ++v) {
            if (FORM == 0) {
                // d_k = eps^2 d_{k-1} - e q_{k-1} + g T_k ;  q_k = q_{k-1} + d_k   (cb holds -e)
                float a = ca[v] * d[v];
                a = fmaf(cb[v], q[v], a);
                if (forced) a = fmaf(g[v], tk, a);
                d[v] = a;
                q[v] = q[v] + a;
            } else {
                // q_k = c1 q_{k-1} + c2 q_{k-2} + g T_k   (d holds q_{k-2})
                float a = cb[v] * d[v];
                if (forced) a = fmaf(g[v], tk, a);
                const float qk = fmaf(ca[v], q[v], a);
                d[v] = q[v];
                q[v] = qk;
            }
            if (SCALED) {
                p = (v == 0) ? q[v] : p + q[v];
            } else {
                p = (v == 0) ? t[v] * q[v] : fmaf(t[v], q[v], p);
            }
            if (QN) qn[v] = fmaf(q[v], q[v], qn[v]);
        }
        // LDS address = M0 (this wave's tile, set by the caller) + offset + 4 * lane
        asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(p), "n"(k * LDS_ROW * 4) : "memory");
        if (QN) {
            // pin the qnorm accumulators here: without it the q^2 FMAs of a whole
            // tile are sunk to the tile's end and TILE*R q values stay live.
#pragma unroll
            for (int v = 0; v < R; ++v) asm volatile("" : "+v"(qn[v]));
        }
        // two adds of the previous tile's row sum per sample (samples 1..16): they
        // are independent of the recurrence and fill its dependency stalls
        if constexpr (k >= 1 && k <= 16) {
            constexpr int i0 = 2 * (k - 1), i1 = i0 + 1;
            const f4 va = rv[i0 / 4], vb = rv[i1 / 4];
            rsum += (i0 % 4 == 0) ? va.x : va.z;
            rsum += (i1 % 4 == 1) ? vb.y : vb.w;
            asm volatile("" : "+v"(rsum));           // keep the adds here (they would be sunk past the tile)
        }
        // keep the scheduler from interleaving whole samples: the other waves of
        // the SIMD fill the issue slots, and register pressure stays bounded.
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Pointers are separate __restrict__ kernel arguments (not struct members) so
// that the uniform loads of descriptors and force profiles are provably
// un-clobbered by the audio/qnorm stores and lower to s_load (SGPR operands).
struct IirDims {
    int nb, n_tiles, m_pad, b_pad;
    long long audio_stride;
    int rotate_prio;
    long long gq_plane;          // elements between the G11 / 2 G12 / G22 planes
    int qn_nb, qn_b0;            // qnorm is [n_obj][qn_nb][m_pad]; this launch fills buffers qn_b0 ..
    unsigned launch_seq;
    unsigned long long *start_flag;     // see IirParams::start_flag
    unsigned long long start_seq;
};

// QNM: 0 no qnorm; 1 per-sample accumulation (the reference's loop, modal_solver.h:270);
//      2 closed form: in a buffer that is force-free after its first sample,
//        sum_k q_k^2 = x0' G x0 with x0 = (q_0, q_0 - q_-1) after sample 0 and G = sum_k (A^k)' e1 e1' A^k
//        precomputed per mode in fp64; buffers with a dense profile accumulate per sample.
template <int R, int FORM, int QNM, int MAXT>
__global__ __launch_bounds__(MAXT) void iir_bank_kernel(
    const float *__restrict__ p_ca, const float *__restrict__ p_cb, float *__restrict__ p_sq,
    float *__restrict__ p_sd, float *__restrict__ p_ss, const BufDesc *__restrict__ p_desc, const float *__restrict__ p_grows,
    const float *__restrict__ p_g32, const long long *__restrict__ p_g32_off,
    const float *__restrict__ p_tprof, const double *__restrict__ p_xfer_rows,
    const int *__restrict__ p_xfer_init, float *__restrict__ p_audio, float *__restrict__ p_qnorm,
    const float *__restrict__ p_gq, const TeamDesc *__restrict__ p_teams, float *__restrict__ p_audio_parts, unsigned *__restrict__ p_board,
    unsigned long long *__restrict__ p_census, const IirDims p) {
    constexpr bool QN = QNM != 0;
    if (p.start_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(p.start_flag, p.start_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TeamDesc team = p_teams[blockIdx.x];
    const int obj = team.obj;
    const int col = team.col0 + threadIdx.x;          // this lane's first column of the object's rows
    unsigned long long census_t0 = 0, census_c0 = 0;
    if (p_census) {
        census_t0 = __builtin_amdgcn_s_memrealtime();
        census_c0 = __builtin_amdgcn_s_memtime();
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = blockDim.x >> 6;
    const int rowlen = blockDim.x;
    float *tile = lds + wave * (TILE * LDS_ROW);
    const unsigned tile_m0 = (unsigned)wave * (unsigned)(TILE * LDS_ROW * sizeof(float));
    const size_t mbase = (size_t)obj * p.m_pad + col;

    float ca[R], cb[R], q[R], d[R], g_[R], t[R], qn[R];   // R oscillators per lane, slice r at column col + r * rowlen
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ca[r] = (p_ca[mbase + r * rowlen]);
        cb[r] = (p_cb[mbase + r * rowlen]);
        q[r] = (p_sq[mbase + r * rowlen]);
        d[r] = (p_sd[mbase + r * rowlen]);
        g_[r] = (0.f);
        qn[r] = (0.f);
    }
    // Scaled state.  The output of a mode is t * q with a transfer weight t that only changes
    // between buffers, and the recurrence is linear: the registers hold Q = t q, D = t d, the
    // lane's output is then a plain sum, the force gain becomes t g and qnorm sqrt(sum Q^2) / t.
    // The state arrays keep (Q, D) together with the scale they carry (p_ss; 1 = unscaled), so a
    // launch boundary changes nothing.  A wave leaves the scaled representation (wave-uniform
    // `scaled`) while any of its modes has a weight it cannot divide by or that would push the
    // state out of fp32 range; it then steps exactly like the literal t * q form.  Padding
    // oscillators (all-zero coefficients, state always 0) take weight 1.
    bool dead[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        dead[r] = ca[r] == 0.f && cb[r] == 0.f;
    bool scaled = false;
    auto usable = [](float x) { return x >= 0x1p-20f && x <= 0x1p40f; };
    // move the state from scale `from` (per mode) to the weights tn, or to scale 1 if some tn is unusable
    auto rescale = [&](const float (&from)[R], const float (&tn)[R]) {
        bool ok = true, same = true, unit = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ok = ok && usable(tn[r]);
            same = same && tn[r] == from[r];
            unit = unit && from[r] == 1.f;
        }
        ok = __all(ok);
        if (ok) {
            if (!__all(same)) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float f = tn[r] / from[r];
                    q[r] = (q[r] * f);
                    d[r] = (d[r] * f);
                }
            }
        } else if (!__all(unit)) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                q[r] = (q[r] / from[r]);
                d[r] = (d[r] / from[r]);
            }
        }
        scaled = ok;
    };
    {
        const int row0 = p_xfer_init[obj];
        float s0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s0[r] = (p_ss[mbase + r * rowlen]);
            const float tr = row0 >= 0 ? (float)p_xfer_rows[(size_t)row0 * p.m_pad + col + r * rowlen] : 1e7f;
            t[r] = (dead[r] ? 1.f : tr);
        }
        rescale(s0, t);
    }

    float g11[R], g12[R], g22[R];                     // QNM == 2: G11, 2 G12, G22 of every mode
    if (QNM == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            g11[r] = (p_gq[mbase + r * rowlen]);
            g12[r] = (p_gq[p.gq_plane + mbase + r * rowlen]);
            g22[r] = (p_gq[2 * p.gq_plane + mbase + r * rowlen]);
        }
    }

    const BufDesc *__restrict__ dsc = p_desc + (size_t)obj * p.nb;
    float *__restrict__ aout = team.part_row >= 0 ? p_audio_parts + (size_t)team.part_row * p.audio_stride
                                                  : p_audio + (size_t)obj * p.audio_stride;
    const int NT = p.n_tiles;
    const int B = NT * TILE;
    const int ring = NT + 1;                         // tile slots in the row-sum ring
    float *pbuf = lds + W * (TILE * LDS_ROW);        // [W][ring * TILE] row sums per wave
    float *mybuf = pbuf + wave * (ring * TILE);
    // row-sum phase: lanes 2k, 2k+1 own the two 32-float halves of row k
    const int rrow = lane >> 1;
    const bool owner = (lane & 1) == 0 && lane < 2 * TILE;
    // lanes 54..63 own no row: they read rows 27..31 (inside the workgroup's LDS, values unused) so
    // that every ds_read_b128 lane group stays on 16 distinct bank slots (clamping them to row 0 cost
    // 15 % of the LDS cycles in bank conflicts)
    const f4 *rsrc = reinterpret_cast<const f4 *>(tile + rrow * LDS_ROW + (lane & 1) * 32);

    // The SIMD arbiter serves equal-priority waves oldest first: left alone, the
    // four teams resident on a CU finish at ~57/66/83/100 % of the kernel and the
    // youngest runs the tail at one-wave issue efficiency (scripts/census.py).
    // Rotating s_setprio by (wave slot + tile counter) gives every resident wave
    // each priority level equally often, so all teams progress at the same pace.
    // Performance hint only: results do not depend on it.
    const int wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 0xF;   // HW_REG_HW_ID.WAVE_ID
    const bool ROTATE = p.rotate_prio != 0;
    // progress feedback (rotate_prio == 2): the waves of a CU publish their tile count; a wave that is
    // behind the leader of its CU raises its priority
    const bool FEEDBACK = p.rotate_prio == 2;
    unsigned *board = p_board + (((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u) << 8) | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xFFu));
    unsigned seen = 0;                                // what the board held one buffer ago (lane 0)
    int boost = 0;

    // state of the one-tile lag (all wave-uniform)
    bool have_prev = false, prev_last = false, settle = false;
    int g = 0, prev_slot = 0, prev_b = 0, prev_g0 = 0;

    // all waves add their ring entries of buffer bb (tiles g0 .. g0 + NT - 1) and store it
    auto combine = [&](int bb, int g0) {
        for (int sidx = tid; sidx < B; sidx += blockDim.x) {
            const int tl = sidx / TILE, row = sidx - tl * TILE;
            const int off = ((g0 + tl) % ring) * TILE + row;
            float acc = pbuf[off];
            for (int w = 1; w < W; ++w) acc += pbuf[w * (ring * TILE) + off];
            aout[(size_t)bb * B + sidx] = acc;
        }
    };
    // finish the row sum held in rsum (lane halves -> row), park it in the ring, and
    // when it completes a buffer, combine that buffer
    auto retire = [&](float rsum) {
        // quad_perm [1,0,3,2]: the partner lane's half (a + b == b + a bitwise)
        rsum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rsum), 0xB1, 0xF, 0xF, false));
        if (owner) mybuf[prev_slot * TILE + rrow] = rsum;
        if (prev_last) {
            __syncthreads();                         // every wave's ring holds buffer prev_b
            combine(prev_b, prev_g0);
            settle = true;
        } else if (settle) {
            __syncthreads();                         // combine done everywhere: slots may be reused
            settle = false;
        }
    };
    // plain (compiler-visible) LDS loads: its s_waitcnt placement cannot see the
    // inline-asm tile writes and is therefore conservative, never wrong.  (Hand-placed
    // asm reads + counted waits measured no faster and break if rv is ever spilled.)
    auto load_rows = [&](f4 (&rv)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rv[j] = rsrc[j];
    };
    // the pending tile has no successor to hide behind: sum it now
    auto flush = [&]() {
        if (!have_prev) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f4 rv[8];
        load_rows(rv);
        float rsum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            rsum += rv[j].x;
            rsum += rv[j].y;
            rsum += rv[j].z;
            rsum += rv[j].w;
        }
        retire(rsum);
        __syncthreads();
        settle = false;
        have_prev = false;
    };

    // descriptors are fetched one buffer ahead (scalar loads, ~1-2 us from HBM)
    BufDesc next = dsc[0];
    for (int b = 0; b < p.nb; ++b) {
        const BufDesc cur = next;
        next = dsc[b + 1 < p.nb ? b + 1 : b];
        const int frow = cur.frow;
        const uint32_t flags = cur.flags;
        // DESC_DIRECT (the hit of a plain PointForce the bank projects itself): prow / tile_mask / pad[0] hold the normal
        const bool direct = (flags & DESC_DIRECT) != 0;
        const int prow = direct ? -1 : cur.prow;
        const uint32_t mask = direct ? 1u : cur.tile_mask;
        const float amp = cur.amp;
        const int trow = cur.trow;

        if (flags & DESC_SKIP) {
            // the reference's step() returned before stepping: no samples, state untouched
            flush();
            for (int i = tid; i < B; i += blockDim.x) aout[(size_t)b * B + i] = 0.f;
            if (QN) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    p_qnorm[((size_t)obj * p.qn_nb + p.qn_b0 + b) * p.m_pad + col + r * rowlen] = 0.f;
            }
            continue;
        }
        if (trow != XFER_KEEP) {
            float tn[R], from[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float tr = trow >= 0 ? (float)p_xfer_rows[(size_t)trow * p.m_pad + col + r * rowlen] : 1e7f;
                tn[r] = (dead[r] ? 1.f : tr);
                from[r] = (scaled ? t[r] : 1.f);
            }
            rescale(from, tn);
#pragma unroll
            for (int v = 0; v < R; ++v) t[v] = tn[v];
        }
        if (frow >= 0) {
            const float *__restrict__ gsrc = direct ? p_g32 + ((size_t)p_g32_off[obj] + frow) * p.m_pad : p_grows + (size_t)frow * p.m_pad;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float gr = gsrc[col + r * rowlen];
                if (direct) {
                    gr = __builtin_bit_cast(float, cur.prow) * gr;
                    gr = fmaf(__builtin_bit_cast(float, cur.tile_mask), (gsrc + p.m_pad)[col + r * rowlen], gr);
                    gr = fmaf(__builtin_bit_cast(float, cur.pad[0]), (gsrc + 2 * (size_t)p.m_pad)[col + r * rowlen], gr);
                }
                g_[r] = (scaled ? gr * t[r] : gr);
            }
        }
        const bool impulse = (flags & DESC_IMPULSE) != 0;
        const bool dense = frow >= 0 && !impulse;
        const bool accum = QNM == 1 || (QNM == 2 && dense);     // per-sample q^2 in this buffer?
        const float *__restrict__ tprow = p_tprof + (size_t)(prow >= 0 ? prow : 0) * p.b_pad;
        if (QN) {
#pragma unroll
            for (int r = 0; r < R; ++r) qn[r] = (0.f);
        }
        if (QNM == 2 && !dense) {
            // x0 = state after sample 0, with exactly the arithmetic sample 0 will use
            const float f0 = (frow >= 0 && (mask & 1u)) ? amp : 0.f;
#pragma unroll
            for (int v = 0; v < R; ++v) {
                float q0, s0;
                if (FORM == 0) {
                    float a = ca[v] * d[v];
                    a = fmaf(cb[v], q[v], a);
                    if (f0 != 0.f) a = fmaf(g_[v], f0, a);
                    s0 = a;
                    q0 = q[v] + a;
                } else {
                    float a = cb[v] * d[v];
                    if (f0 != 0.f) a = fmaf(g_[v], f0, a);
                    q0 = fmaf(ca[v], q[v], a);
                    s0 = q0 - q[v];             // G is kept in the (q, q - q_prev) basis for both forms
                }
                float e = g22[v] * s0 * s0;
                e = fmaf(g12[v] * q0, s0, e);
                qn[v] = fmaf(g11[v] * q0, q0, e);
            }
        }
        const int g0 = g;

        for (int tl = 0; tl < NT; ++tl) {
            // previous tile's rows: issued before this tile's writes (in-order LDS), used during it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            f4 rv[8];
            load_rows(rv);
            float rsum = 0.f;
            if (FEEDBACK && tl == 0) {
                // once per buffer: read what the CU's leader had published a buffer ago, publish own progress
                const unsigned prev = __builtin_amdgcn_readfirstlane(seen);
                const int lead = (prev >> 20) == (p.launch_seq & 0xFFFu) ? (int)(prev & 0xFFFFFu) - (b - 1) : 0;
                boost = lead >= 2 ? 2 : (lead >= 1 ? 1 : 0);
                if (lane == 0) seen = atomicMax(board, ((p.launch_seq & 0xFFFu) << 20) | (unsigned)b);
            }
            if (ROTATE) {
                int pr = ((wave_slot + g) & 3) + boost;
                pr = pr > 3 ? 3 : pr;
                switch (pr) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
                }
            }
            asm volatile("s_mov_b32 m0, %0" ::"s"(tile_m0) : "memory");
            const bool hit = frow >= 0 && ((mask >> tl) & 1u);
            auto run_tile = [&](auto sc) {
                constexpr bool SC = decltype(sc)::value;
                if (hit && impulse) {
                    step_tile<R, FORM, QNM == 1, 2, SC>(q, d, ca, cb, g_, t, qn, nullptr, amp, rv, rsum);
                } else if (hit) {
                    step_tile<R, FORM, QN, 1, SC>(q, d, ca, cb, g_, t, qn, tprow + tl * TILE, 0.f, rv, rsum);
                } else if (QNM == 2) {
                    if (accum) step_tile<R, FORM, true, 0, SC>(q, d, ca, cb, g_, t, qn, nullptr, 0.f, rv, rsum);
                    else step_tile<R, FORM, false, 0, SC>(q, d, ca, cb, g_, t, qn, nullptr, 0.f, rv, rsum);
                } else {
                    step_tile<R, FORM, QN, 0, SC>(q, d, ca, cb, g_, t, qn, nullptr, 0.f, rv, rsum);
                }
            };
            if (scaled) run_tile(std::true_type{});
            else run_tile(std::false_type{});

            if (have_prev) retire(rsum);
            have_prev = true;
            prev_slot = g % ring;
            prev_b = b;
            prev_g0 = g0;
            prev_last = tl == NT - 1;
            ++g;
        }

        if (QN) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // (the closed form can round a tiny sum below zero)
                const float nrm = sqrtf(fmaxf(qn[r], 0.f));
                p_qnorm[((size_t)obj * p.qn_nb + p.qn_b0 + b) * p.m_pad + col + r * rowlen] = scaled ? nrm / t[r] : nrm;
            }
        }
    }
    flush();
    if (p_census && tid == 0) {
        // where and when this workgroup ran (placement / residency diagnostics)
        p_census[(size_t)team.id * CENSUS_WORDS + 0] = census_t0;
        p_census[(size_t)team.id * CENSUS_WORDS + 1] = __builtin_amdgcn_s_memrealtime();
        p_census[(size_t)team.id * CENSUS_WORDS + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        p_census[(size_t)team.id * CENSUS_WORDS + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        p_census[(size_t)team.id * CENSUS_WORDS + 4] = census_c0;                                    // shader clock at start
        p_census[(size_t)team.id * CENSUS_WORDS + 5] = __builtin_amdgcn_s_memtime();                 // ... and at the end
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        p_sq[mbase + r * rowlen] = q[r];
        p_sd[mbase + r * rowlen] = d[r];
        p_ss[mbase + r * rowlen] = scaled ? t[r] : 1.f;
    }
}

template <int R, int FORM, int QNM, int MAXT>
static int launch_one(const IirParams &p, int n_obj, int W, hipStream_t stream) {
    const size_t lds = iir_lds_bytes(W, p.n_tiles);
    auto kern = iir_bank_kernel<R, FORM, QNM, MAXT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const IirDims dims = {p.nb, p.n_tiles, p.m_pad, p.b_pad, p.audio_stride, p.rotate_prio, p.gq_plane, p.qn_nb, p.qn_b0, p.launch_seq, p.start_flag, p.start_seq};
    hipLaunchKernelGGL(kern, dim3(n_obj), dim3(64 * W), lds, stream, p.ca, p.cb, p.sq, p.sd, p.ss, p.desc,
                       p.grows, p.g32, p.g32_off, p.tprof, p.xfer_rows, p.xfer_init, p.audio, p.qnorm, p.gq, p.teams, p.audio_parts, p.board, p.census, dims);
    return (int)hipGetLastError();
}

template <int R, int MAXT>
static int launch_r(const IirParams &p, int n_obj, int W, int form, int qnm, hipStream_t s) {
    switch ((form ? 3 : 0) + qnm) {
    case 0: return launch_one<R, 0, 0, MAXT>(p, n_obj, W, s);
    case 1: return launch_one<R, 0, 1, MAXT>(p, n_obj, W, s);
    case 2: return launch_one<R, 0, 2, MAXT>(p, n_obj, W, s);
    case 3: return launch_one<R, 1, 0, MAXT>(p, n_obj, W, s);
    case 4: return launch_one<R, 1, 1, MAXT>(p, n_obj, W, s);
    default: return launch_one<R, 1, 2, MAXT>(p, n_obj, W, s);
    }
}

// teams of up to 4 waves use the 256-thread build (no VGPR cap in practice);
// larger teams (objects with more than 256 R modes) the 1024-thread build.
// R in {1,2,3,4,8} for both.
int launch_iir_bank(const IirParams &p, int n_obj, int R, int W, int form, int qnm, hipStream_t s) {
    if (n_obj <= 0) return 0;
    if (qnm < 0 || qnm > 2) return (int)hipErrorInvalidValue;
    if (W < 1 || W > MAX_WAVES_PER_TEAM) return (int)hipErrorInvalidValue;
    if (W <= 4) {
        switch (R) {
        case 1: return launch_r<1, 256>(p, n_obj, W, form, qnm, s);
        case 2: return launch_r<2, 256>(p, n_obj, W, form, qnm, s);
        case 3: return launch_r<3, 256>(p, n_obj, W, form, qnm, s);
        case 4: return launch_r<4, 256>(p, n_obj, W, form, qnm, s);
        case 8: return launch_r<8, 256>(p, n_obj, W, form, qnm, s);
        }
    } else {
        switch (R) {
        case 1: return launch_r<1, 1024>(p, n_obj, W, form, qnm, s);
        case 2: return launch_r<2, 1024>(p, n_obj, W, form, qnm, s);
        case 3: return launch_r<3, 1024>(p, n_obj, W, form, qnm, s);
        case 4: return launch_r<4, 1024>(p, n_obj, W, form, qnm, s);
        case 8: return launch_r<8, 1024>(p, n_obj, W, form, qnm, s);
        }
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace iir_scalar
}  // namespace pbso
