// K5: time parallelism ACROSS audio buffers -- the per-mode scan of buffer-start states (gfx950, wave64).
//
// Replaces, together with the chunked launches of K1b (kernels_block.hip), the serial walk over a launch's buffers in
// the hot loop of ModalSolver::step (modal_solver.h:262-272) around ModalIntegrator::Step (modal_integrator.h:103-113).
//
// Why it is allowed.  The reference admits a new force only at the first sample of a buffer (modal_solver.h:184-205; a
// PointForce is ONE sample, forces.h:81-90) and a mode's recurrence is linear with constant coefficients
// (modal_integrator.h:103-113).  With x = (q, d = q - q_prev), A the one-sample matrix in that basis and u = (1, 1)' the
// direction a force sample enters the state (d += f, q += d), a buffer of B = 513 samples maps its start state to
//     x_{b+1} = A^513 x_b                                   force-free buffer
//     x_{b+1} = A^513 x_b + (g amp) A^512 u                 PointForce at sample 0 (g = c3 * S of the hit, modal_solver.h:266)
//     x_{b+1} = x_b                                         step() returned early (clearAllForces, modal_solver.h:186-189)
// so the states at the start of EVERY buffer of a launch follow from an 86-step scan per mode (4 FMA + 2 per step),
// after which the buffers no longer depend on each other: K1b runs the launch as (team, chunk of buffers) workgroups,
// each starting from the state this kernel left for its first buffer -- the chip fills even when a scene has fewer
// waves of oscillators than SIMDs (BASELINE configs[1], [2], [4] and the 128 x 512 share of configs[3] on 8 GPUs).
// A buffer with a DENSE force profile (Gaussian / AR, forces.h:92-128) has no closed form: the scan steps its 513 samples
// literally (velocity form, the per-sample kernels' arithmetic); the engine sends launches that are mostly such buffers
// to the kernels that walk the buffers in order (K1p / K1b unchunked) instead.
//
// A^513 and A^512 u are per-mode constants, fp64 on the host, rounded once (Engine::finalize); A^513's first entry is
// stored minus one, as K1b's coarse step P = A^16.  One wave = 64 consecutive columns of one object, one mode per lane.
// Two passes per 64 buffers.  (1) lane = BUFFER: every lane decodes one descriptor (two coalesced 16-byte loads per 64
// buffers) into what the step needs -- the address of the row(s) the hit's gain comes from (g, or the three g32 rows of a
// DESC_DIRECT hit; a buffer without a hit points at a finite dummy row), their weights (0 without a hit), kind bits.
// (2) lane = MODE: the buffers in order, eight at a time; their fields reach scalar registers by v_readlane, their rows
// are loaded UNCONDITIONALLY and one group ahead, so the compiler counts the loads exactly and a step never waits for a
// round trip to L2 / HBM.  A group that holds a dense buffer (or the ragged tail) takes a generic loop that exists once:
// the first version unrolled all 64 buffer positions with the dense loop inside each -- 90 KB of code per kernel, 45 us
// per launch in instruction fetches alone.
#include <type_traits>

#include "kernels.h"

namespace pbso {
namespace iir_scan {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));

constexpr int G = 8;             // buffers per group of prefetched rows

template <int K0, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        f(std::integral_constant<int, K0>{});
        static_for<K0 + 1, N - 1>(f);
    }
}

struct ScanDims {
    int nb, cb, n_chunks, m_pad, b_pad, frames;
    long long plane;             // elements between the planes of p_sc / p_pc
};

#ifndef PBSO_SCAN_FAST
#define PBSO_SCAN_FAST 1
#endif
constexpr unsigned K_SKIP = 1u, K_DENSE = 2u;

template <bool DIRECT>
__global__ __launch_bounds__(64) void iir_scan_kernel(
    const float *__restrict__ p_ca, const float *__restrict__ p_cb, float *__restrict__ p_sq, float *__restrict__ p_sd,
    float *__restrict__ p_ss, const float *__restrict__ p_sc, const BufDesc *__restrict__ p_desc,
    const float *__restrict__ p_grows, const float *__restrict__ p_g32, const long long *__restrict__ p_g32_off,
    const float *__restrict__ p_tprof, const int *__restrict__ p_xfer_init, float *__restrict__ p_xs,
    int *__restrict__ p_xtrow, const ScanDims p) {
    const int obj = blockIdx.y;
    const int col0 = 64 * blockIdx.x;
    const unsigned lane = threadIdx.x;
    const size_t ubase = (size_t)obj * p.m_pad + col0;
    const float nca = (p_ca + ubase)[lane], ncb = (p_cb + ubase)[lane];               // eps^2, -e (velocity form)
    const float s11 = (p_sc + ubase)[lane], s12 = (p_sc + p.plane + ubase)[lane];     // A^513: P11 - 1, P12, P21, P22
    const float s21 = (p_sc + 2 * p.plane + ubase)[lane], s22 = (p_sc + 3 * p.plane + ubase)[lane];
    const float hq = (p_sc + 4 * p.plane + ubase)[lane], hd = (p_sc + 5 * p.plane + ubase)[lane];      // A^512 u
    f2 x;
    {
        const float s0 = (p_ss + ubase)[lane];       // the arrays hold scale x state (kernels_iir.hip, "scaled state")
        x.x = (p_sq + ubase)[lane] / s0;
        x.y = (p_sd + ubase)[lane] / s0;
    }
    const BufDesc *__restrict__ dsc = p_desc + (size_t)obj * p.nb;
    const float *__restrict__ dummy = p_ca + ubase;                                   // a finite row for buffers without a hit
    const float *__restrict__ g32_obj = DIRECT ? p_g32 + (size_t)p_g32_off[obj] * p.m_pad + col0 : nullptr;
    int cur_row = p_xfer_init[obj];
    f2 *__restrict__ xs = reinterpret_cast<f2 *>(p_xs) + (size_t)obj * p.n_chunks * p.m_pad + col0;
    constexpr int NR = DIRECT ? 3 : 1;
    int next_mark = 0, chunk_i = 0;
    auto rl = [](int v, int j) { return __builtin_amdgcn_readlane(v, j); };
    auto rlf = [](float v, int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); };
    auto rlp = [&](unsigned long long v, int j) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, j), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), j);
        return reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
    };

    for (int base = 0; base < p.nb; base += 64) {
        // ---- pass 1, lane = buffer base + lane: what its step needs
        unsigned long long ptr[NR];
        float w[NR];
        unsigned kind;
        int trow, prow;
        {
            const int bi = base + (int)lane;
            const bool in = bi < p.nb;
            const i4 *src = reinterpret_cast<const i4 *>(dsc + (in ? bi : p.nb - 1));
            const i4 dlo = src[0];                   // frow, prow, tile_mask, amp
            const i4 dhi = src[1];                   // trow, flags, pad[0], pad[1]
            // (scalars first: __builtin_bit_cast of a vector ELEMENT expression reads element 0 with this compiler)
            const int frow = dlo.x, w_prow = dlo.y, w_mask = dlo.z, w_amp = dlo.w, w_pad0 = dhi.z;
            const unsigned flags = (unsigned)dhi.y;
            const bool skip = (flags & DESC_SKIP) != 0;
            const bool live = in && frow >= 0 && !skip;
            const bool impulse = (flags & DESC_IMPULSE) != 0;
            const bool direct = DIRECT && (flags & DESC_DIRECT) != 0;
            const bool hit0 = direct || (w_mask & 1);
            const float a = !live ? 0.f : (impulse ? (hit0 ? __builtin_bit_cast(float, w_amp) : 0.f) : 1.f);      // (a dense buffer wants g itself)
            const bool dl = live && direct;
            // a DESC_DIRECT hit: g = n . (three rows of the object's (float)(c3 * shape) table), the normal in the descriptor's
            // spare words (kernels.h)
            const float *r0 = dl ? g32_obj + (size_t)frow * p.m_pad : (live ? p_grows + (size_t)frow * p.m_pad + col0 : dummy);
            ptr[0] = (unsigned long long)r0;
            w[0] = dl ? a * __builtin_bit_cast(float, w_prow) : a;
            if constexpr (DIRECT) {
                ptr[1] = (unsigned long long)(dl ? g32_obj + (size_t)(frow + 1) * p.m_pad : dummy);
                ptr[2] = (unsigned long long)(dl ? g32_obj + (size_t)(frow + 2) * p.m_pad : dummy);
                w[1] = dl ? a * __builtin_bit_cast(float, w_mask) : 0.f;
                w[2] = dl ? a * __builtin_bit_cast(float, w_pad0) : 0.f;
            }
            kind = (skip ? K_SKIP : 0u) | (live && !impulse ? K_DENSE : 0u);
            trow = dhi.x;
            prow = direct ? -1 : w_prow;
        }
        const unsigned long long slow_mask = __ballot((kind & K_DENSE) != 0);
        const int nd = p.nb - base < 64 ? p.nb - base : 64;

        // ---- pass 2, lane = mode
        struct Rows { float r[G][NR]; };
        auto fetch = [&](Rows &R, int j0) {          // rows of buffers j0 .. j0 + G - 1 (lanes beyond the launch hold the last buffer's)
            static_for<0, G>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int j = j0 + i < 64 ? j0 + i : 63;
                static_for<0, NR>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    R.r[i][k] = rlp(ptr[k], j)[lane];
                });
            });
        };
        auto mark = [&](int b) {                     // the first buffer of a chunk: its start state and the transfer row in force
            if (b == next_mark) {
                (xs + (size_t)chunk_i * p.m_pad)[lane] = x;
                if (col0 == 0 && lane == 0) p_xtrow[(size_t)obj * p.n_chunks + chunk_i] = cur_row;
                next_mark += p.cb;
                chunk_i += 1;
            }
        };
        auto coarse = [&](float gv) {                // x <- A^513 x (+ the impulse's share): q' = q + (P11 - 1) q + P12 d, the small terms last
            const float qa = fmaf(s11, x.x, x.x);
            const float da = s21 * x.x;
            float qn = fmaf(s12, x.y, qa);
            float dn = fmaf(s22, x.y, da);
            qn = fmaf(gv, hq, qn);
            dn = fmaf(gv, hd, dn);
            x.x = qn;
            x.y = dn;
        };
        auto step_fast = [&](const Rows &R, int j0) {    // a full group without a dense buffer
            static_for<0, G>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int j = j0 + i;
                mark(base + j);
                const unsigned kd = (unsigned)rl((int)kind, j);
                if (kd & K_SKIP) return;             // step() returned before stepping: state (and transfer) untouched
                const int tr = rl(trow, j);
                if (tr != XFER_KEEP) cur_row = tr;
                float gv = rlf(w[0], j) * R.r[i][0];
                if constexpr (DIRECT) {
                    gv = fmaf(rlf(w[NR > 1 ? 1 : 0], j), R.r[i][NR > 1 ? 1 : 0], gv);
                    gv = fmaf(rlf(w[NR > 2 ? 2 : 0], j), R.r[i][NR > 2 ? 2 : 0], gv);
                }
                coarse(gv);
            });
        };
        auto step_slow = [&](int j0, int n) {        // any buffers, one at a time (rows loaded on demand)
            for (int j = j0; j < j0 + n; ++j) {
                mark(base + j);
                const unsigned kd = (unsigned)rl((int)kind, j);
                if (kd & K_SKIP) continue;
                const int tr = rl(trow, j);
                if (tr != XFER_KEEP) cur_row = tr;
                float gv = rlf(w[0], j) * rlp(ptr[0], j)[lane];
                if constexpr (DIRECT) {
                    gv = fmaf(rlf(w[NR > 1 ? 1 : 0], j), rlp(ptr[NR > 1 ? 1 : 0], j)[lane], gv);
                    gv = fmaf(rlf(w[NR > 2 ? 2 : 0], j), rlp(ptr[NR > 2 ? 2 : 0], j)[lane], gv);
                }
                if (kd & K_DENSE) {
                    // dense force profile: every sample, literally (d = eps^2 d - e q + g T_k ; q += d); the row 64 samples at a
                    // time, one per lane and a batch ahead; sample k reaches the FMA as a scalar operand
                    const int pr = rl(prow, j);
                    const float *__restrict__ tprow = p_tprof + (size_t)(pr >= 0 ? pr : 0) * p.b_pad;
                    auto ldt = [&](int kb) { const int k = kb + (int)lane; return tprow[k < p.frames ? k : p.frames - 1]; };
                    float tv = ldt(0);
                    for (int kb = 0; kb < p.frames; kb += 64) {
                        const float tn = ldt(kb + 64 < p.frames ? kb + 64 : kb);
                        const int nk = p.frames - kb < 64 ? p.frames - kb : 64;
                        for (int k = 0; k < nk; ++k) {
                            const float tk = rlf(tv, k);
                            float a = nca * x.y;
                            a = fmaf(ncb, x.x, a);
                            a = fmaf(gv, tk, a);
                            x.y = a;
                            x.x = x.x + a;
                        }
                        tv = tn;
                    }
                } else {
                    coarse(gv);
                }
            }
        };
        auto step = [&](const Rows &R, int j0) {
            if (j0 >= nd) return;
            const int n = nd - j0 < G ? nd - j0 : G;
            if (PBSO_SCAN_FAST && n == G && ((slow_mask >> j0) & ((1ull << G) - 1)) == 0) step_fast(R, j0);
            else step_slow(j0, n);
        };
        Rows Ra, Rb;
        fetch(Ra, 0);
        for (int j0 = 0; j0 < nd; j0 += 2 * G) {
            fetch(Rb, j0 + G);
            step(Ra, j0);
            fetch(Ra, j0 + 2 * G);
            step(Rb, j0 + G);
        }
    }
    (p_sq + ubase)[lane] = x.x;
    (p_sd + ubase)[lane] = x.y;
    (p_ss + ubase)[lane] = 1.f;
}

}  // namespace iir_scan

int launch_iir_scan(const IirParams &p, int n_obj, const float *sc, int cb, int n_chunks, float *xs, int *xtrow, bool direct,
                    hipStream_t stream) {
    if (n_obj <= 0 || p.nb <= 0) return 0;
    if (cb <= 0 || n_chunks != (p.nb + cb - 1) / cb || p.m_pad % 64) return (int)hipErrorInvalidValue;
    const iir_scan::ScanDims dims = {p.nb, cb, n_chunks, p.m_pad, p.b_pad, p.frames, p.gq_plane};
    const dim3 grid(p.m_pad / 64, n_obj), block(64);
    if (direct)
        hipLaunchKernelGGL(iir_scan::iir_scan_kernel<true>, grid, block, 0, stream, p.ca, p.cb, p.sq, p.sd, p.ss, sc, p.desc, p.grows,
                           p.g32, p.g32_off, p.tprof, p.xfer_init, xs, xtrow, dims);
    else
        hipLaunchKernelGGL(iir_scan::iir_scan_kernel<false>, grid, block, 0, stream, p.ca, p.cb, p.sq, p.sd, p.ss, sc, p.desc, p.grows,
                           p.g32, p.g32_off, p.tprof, p.xfer_init, xs, xtrow, dims);
    return (int)hipGetLastError();
}

}  // namespace pbso
