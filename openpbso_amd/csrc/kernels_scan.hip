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
// A buffer with a DENSE force profile T (Gaussian / AR, forces.h:92-128; the sustained branch modal_solver.h:222-240) is
// linear in its start state too (modal_integrator.h:103-113):
//     x_{b+1} = A^513 x_b + g V_b,      V_b = sum_k A^(512-k) u T_k          (g = c3 * S, modal_solver.h:266)
// and V_b -- the state a UNIT force gain leaves behind from rest -- depends on the mode and the profile only, not on the state:
// dense_increment_kernel (below) evaluates it for every dense (object, buffer) of a launch AT ONCE, a block of 16 samples at
// a time on the matrix pipe (the increments F . T_n of kernels_block.hip's forced block path, then a 32-step Horner in
// P = A^16), and the scan's DENSE build takes g V_b where an impulse buffer takes (g amp) A^512 u.  Round 5: before that the
// scan stepped such buffers sample by sample (86 x 513 dependent steps for a second of sustained scraping), and launches that
// were mostly dense kept the kernels that walk the buffers in order.
//
// A^513 and A^512 u are per-mode constants, fp64 on the host, rounded once (Engine::finalize); A^513's first entry is
// stored minus one, as K1b's coarse step P = A^16.  One wave = 64 consecutive columns of one object, one mode per lane.
// The only part that has to be sequential is the state update itself, so everything else is taken out of it -- per 64
// buffers:
//   (1) lane = BUFFER: every lane decodes one descriptor (two coalesced 16-byte loads per 64 buffers, fetched a batch
//       ahead) into what its step needs: the address of the row(s) its gain comes from (g, or the three g32 rows of a
//       DESC_DIRECT hit), their weights, kind bits; wave ballots turn those into 64-bit masks (hit / skip / dense), and the
//       transfer row in force at every chunk start is a lane-parallel "last row set before me" (no walk);
//   (2) lane = MODE, buffers with a hit only (a quarter of them in a Poisson train), sixteen at a time: rows loaded back to
//       back, gain g amp written to LDS [buffer][mode] (zero-filled first);
//   (3) lane = MODE, the scan: per buffer one LDS read (issued eight buffers ahead) and eight vector instructions two
//       dependent operations deep; a scalar bit test sends chunk starts to their store and skipped / dense buffers to a
//       generic path that exists once.
// The first version walked descriptors, loads and bookkeeping inside the sequential loop: 45 us per launch (90 KB of
// unrolled code), 22 us after compaction, 13 us in this form (scripts/debug/r04_scan_abl.sh times its stages) -- and since
// only the scan itself hands the state from launch to launch, the engine runs it on the PREPARATION stream, beside the
// previous launch's oscillator bank.
#include <type_traits>

#include "kernels.h"

namespace pbso {
namespace iir_scan {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

constexpr int G = 8;             // buffers per unrolled group of the scan
constexpr int HB = 16;           // hits whose rows are in flight together
typedef const __attribute__((address_space(1))) float *gptr;      // (a pointer rebuilt from two scalar halves stays a GLOBAL pointer)

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

#ifndef PBSO_SCAN_STOP
#define PBSO_SCAN_STOP 9         // (ablation builds for timing only: scripts/debug/r04_scan_abl.sh)
#endif

// One body, two kernels.  SERIAL (iir_scan_kernel): one wave per 64 columns of an object walks ALL buffers of the launch, batches
// of 64, and stores the state at every chunk start.  SEG (iir_scan_seg_kernel, round 5): the scan cut along the time axis itself.
// A chunk of buffers is an AFFINE map of the state,
//     x_end = M x_start + v,      M = (A^513)^(buffers stepped),  v = what the chunk's forces leave behind from rest,
// so one wave per CHUNK (a workgroup = the chunks of one 64-column tile, at most SEG_MAX) scans the zero-state response v of its own
// buffers only -- the same chain from x = 0, batches of 32 buffers (16 with the pairs of a DENSE launch: 8 KB of LDS per wave) --
// and raises A^513 to the number of buffers it stepped (square-and-multiply in fp64, rounded once), the waves leave (M, v) in LDS,
// and every wave composes the maps of the chunks before its own: n_chunks - 1 steps of six FMAs.  860 buffers in 8 chunks: the
// depth of a 108-buffer scan for the work of the serial one.  One buffer per chunk keeps the serial scan, whose arithmetic does
// not depend on where a step is cut.
//
// DENSE: the launch has buffers with a dense force profile; p_vinc holds their increments V [profile row][m_pad] pairs (q, d)
// and the batch's gains are kept as pairs g V (dense) / (g amp) A^512 u (impulse): twice the LDS per buffer.
constexpr int SEG_MAX = 8;       // chunks (= waves per workgroup) of the segmented form
constexpr int SEG_ROWS = 32;     // rows of 64 floats of LDS per wave of the segmented form: 32 gains, or the pairs of 16 buffers

template <bool DIRECT, bool DENSE, bool SEG>
__device__ __forceinline__ void scan_body(
    float *__restrict__ p_sq, float *__restrict__ p_sd,
    float *__restrict__ p_ss, const float *__restrict__ p_sc, const BufDesc *__restrict__ p_desc,
    const float *__restrict__ p_grows, const float *__restrict__ p_g32, const long long *__restrict__ p_g32_off,
    const float *__restrict__ p_vinc, const int *__restrict__ p_xfer_init, float *__restrict__ p_xs,
    int *__restrict__ p_xtrow, const ScanDims &p) {
    constexpr int BATCH = SEG ? (DENSE ? SEG_ROWS / 2 : SEG_ROWS) : 64;       // buffers per batch (lane = buffer for the decode)
    typedef float row64[64];
    // [buffer of the batch][mode]: the hit's gain g amp (0: no hit); DENSE: the pair it adds to the state.  (Dynamic: the
    // segmented form sizes it by the launch's chunk count -- a workgroup of two waves takes 20 KB, not the 78 KB of eight)
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int n_waves = SEG ? (int)(blockDim.x >> 6) : 1;
    const int wv = SEG ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;        // SEG: the chunk
    row64 *lds_g = reinterpret_cast<row64 *>(lds_dyn) + (size_t)wv * (SEG ? SEG_ROWS : 0);
    row64 *lds_map_all = reinterpret_cast<row64 *>(lds_dyn) + (size_t)n_waves * SEG_ROWS;      // SEG, per chunk: M e1, M e2, v
    int *lds_row_all = reinterpret_cast<int *>(lds_map_all + (size_t)n_waves * 6);            // SEG, per chunk: did a buffer set the transfer row, and the last one set
    // (a flat grid: grid.y is capped at 65535 objects)
    const int tiles = p.m_pad / 64;
    const int obj = blockIdx.x / tiles;
    const int col0 = 64 * (blockIdx.x % tiles);
    const unsigned lane = threadIdx.x & 63u;
    const size_t ubase = (size_t)obj * p.m_pad + col0;
    const float s11 = (p_sc + ubase)[lane], s12 = (p_sc + p.plane + ubase)[lane];     // A^513: P11 - 1, P12, P21, P22
    const float s21 = (p_sc + 2 * p.plane + ubase)[lane], s22 = (p_sc + 3 * p.plane + ubase)[lane];
    const float hq = (p_sc + 4 * p.plane + ubase)[lane], hd = (p_sc + 5 * p.plane + ubase)[lane];      // A^512 u
    // the wave's buffers: all of the launch, or its chunk
    const int b_lo = SEG ? wv * p.cb : 0;
    const int b_hi = SEG ? (b_lo + p.cb < p.nb ? b_lo + p.cb : p.nb) : p.nb;
    f2 x = f2{0.f, 0.f};                             // SEG: the zero-state response
    if constexpr (!SEG) {
        const float s0 = (p_ss + ubase)[lane];       // the arrays hold scale x state (kernels_iir.hip, "scaled state")
        x.x = (p_sq + ubase)[lane] / s0;
        x.y = (p_sd + ubase)[lane] / s0;
    }
    const BufDesc *__restrict__ dsc = p_desc + (size_t)obj * p.nb;
    const float *__restrict__ g32_obj = DIRECT ? p_g32 + (size_t)p_g32_off[obj] * p.m_pad + col0 : nullptr;
    int cur_row = SEG ? XFER_KEEP : p_xfer_init[obj];        // SEG: the last row a buffer of the chunk set (XFER_KEEP: none did)
    int n_stepped = 0;                               // SEG: the buffers that stepped the state
    f2 *__restrict__ xs = reinterpret_cast<f2 *>(p_xs) + (size_t)obj * p.n_chunks * p.m_pad + col0;
    constexpr int NR = DIRECT ? 3 : 1;
    auto rl = [](int v, int j) { return __builtin_amdgcn_readlane(v, j); };
    auto rlf = [](float v, int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); };
    auto rlp = [&](unsigned long long v, int j) {
        const unsigned lo = (unsigned)rl((int)(unsigned)v, j), hi = (unsigned)rl((int)(unsigned)(v >> 32), j);
        return (gptr)(((unsigned long long)hi << 32) | lo);
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // the descriptors of a batch, one per lane, fetched a batch ahead
    auto load_descs = [&](int base, i4 &lo, i4 &hi) {
        const int bi = base + (int)lane;
        const i4 *src = reinterpret_cast<const i4 *>(dsc + (bi < b_hi ? bi : b_hi - 1));
        lo = src[0];                                 // frow, prow, tile_mask, amp
        hi = src[1];                                 // trow, flags, pad[0], pad[1]
    };
    i4 nlo = i4{0, 0, 0, 0}, nhi = i4{0, 0, 0, 0};
    if (b_lo < b_hi) load_descs(b_lo, nlo, nhi);
    f2 *__restrict__ xs_next = xs;                   // where the next chunk's start state goes (chunks start in order)

    // Software pipeline over the batches (round 5): batch k + 1 is DECODED and the rows of its first HB hits are REQUESTED before
    // the chain of batch k runs, so the rows' way from HBM (half of a lone scan's time: scripts/debug/r05_scan_abl.sh, stop 3
    // against stop 2) passes beside the chain instead of in front of it.  Same operations on the same values as the staged form.
    struct Batch {
        unsigned long long ptr;                      // lane = buffer: address of its (first) row, this wave's columns
        float w[NR];
        int prow;
        unsigned long long hit_mask, skip_mask, dense_mask, mark_mask, direct_mask;
    };
    // ---- (1) lane = buffer base + lane
    auto decode = [&](int base, Batch &d) {
        const int nd = b_hi - base < BATCH ? b_hi - base : BATCH;
        const bool in = (int)lane < nd;
        const i4 dlo = nlo, dhi = nhi;
        if (base + BATCH < b_hi) load_descs(base + BATCH, nlo, nhi);
        // (scalars first: __builtin_bit_cast of a vector ELEMENT expression reads element 0 with this compiler)
        const int frow = dlo.x, w_prow = dlo.y, w_mask = dlo.z, w_amp = dlo.w, w_pad0 = dhi.z;
        const unsigned flags = (unsigned)dhi.y;
        const bool skip = in && (flags & DESC_SKIP) != 0;
        const bool live = in && frow >= 0 && !(flags & DESC_SKIP);
        const bool impulse = (flags & DESC_IMPULSE) != 0;
        const bool direct = DIRECT && (flags & DESC_DIRECT) != 0;
        const bool hit0 = direct || (w_mask & 1);
        const float a = impulse ? (hit0 ? __builtin_bit_cast(float, w_amp) : 0.f) : 1.f;      // (a dense buffer wants g itself)
        const bool dl = live && direct;
        // a DESC_DIRECT hit: g = n . (three rows of the object's (float)(c3 * shape) table), the normal in the descriptor's
        // spare words (kernels.h); any other hit has ONE row, g = c3 * S from the combine kernel (its other two reads repeat
        // it with weight 0)
        const float *r0 = dl ? g32_obj + (size_t)frow * p.m_pad : p_grows + (size_t)(live ? frow : 0) * p.m_pad + col0;
        d.ptr = (unsigned long long)r0;
        d.w[0] = dl ? a * __builtin_bit_cast(float, w_prow) : a;
        d.direct_mask = 0;
        if constexpr (DIRECT) {
            d.direct_mask = __ballot(dl);            // (its rows 1 and 2 follow row 0 at m_pad floats: a scalar stride, no lane reads)
            d.w[1] = dl ? a * __builtin_bit_cast(float, w_mask) : 0.f;
            d.w[2] = dl ? a * __builtin_bit_cast(float, w_pad0) : 0.f;
        }
        const int trow = dhi.x;
        d.prow = direct ? -1 : w_prow;
        d.hit_mask = __ballot(live && (a != 0.f || !impulse));
        d.skip_mask = __ballot(skip);
        d.dense_mask = __ballot(live && !impulse);
        d.mark_mask = 0;
        const unsigned long long set_mask = __ballot(in && !skip && trow != XFER_KEEP);
        if constexpr (!SEG) {
            // chunk starts in this batch (a handful: scalar), and the transfer row in force when one starts: the last row a
            // non-skipped buffer before it switched to
            const int c0 = (base + p.cb - 1) / p.cb;              // the first chunk that starts at or behind `base`
            for (int b = c0 * p.cb; b < base + nd; b += p.cb) d.mark_mask |= 1ull << (b - base);
            const unsigned long long below = (1ull << lane) - 1ull;
            const unsigned long long before = set_mask & below;
            const int src_lane = before ? 63 - __builtin_clzll(before) : 0;
            const int got = __shfl(trow, src_lane, 64);
            if (((d.mark_mask >> lane) & 1) && col0 == 0)
                p_xtrow[(size_t)obj * p.n_chunks + c0 + __builtin_popcountll(d.mark_mask & below)] = before ? got : cur_row;
        }
        if (set_mask) cur_row = rl(trow, 63 - __builtin_clzll(set_mask));
    };
    // the next (at most HB) hits of a mask: their buffers in jj[0 .. n), n returned; `m` loses them
    auto take_hits = [](unsigned long long &m, int (&jj)[HB]) {
        const int total = __builtin_popcountll(m);
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            jj[h] = m ? __builtin_ctzll(m) : 0;
            m &= m - 1;                              // (0 stays 0)
        }
        return total < HB ? total : HB;
    };
    // ---- (2a) lane = mode: the rows of the hits, requested back to back (a uniform branch per hit: a batch of the segmented
    //      form has eight on average, and the sixteen slots' worth of lane reads and loads was a third of its instructions)
    auto request = [&](const Batch &d, const int (&jj)[HB], int nh, float (&r)[HB][NR], f2 (&vv)[DENSE ? HB : 1]) {
        static_for<0, HB>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            if (h < nh) {
                const gptr row0 = rlp(d.ptr, jj[h]);
                const size_t stride = DIRECT && ((d.direct_mask >> jj[h]) & 1) ? (size_t)p.m_pad : 0;
                static_for<0, NR>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    r[h][k] = (row0 + k * stride)[lane];
                });
                if constexpr (DENSE) {
                    // what a unit gain adds to the state over this buffer: its increment row (dense), A^512 u (impulse)
                    vv[h] = f2{hq, hd};
                    if ((d.dense_mask >> jj[h]) & 1) vv[h] = (reinterpret_cast<const f2 *>(p_vinc) + (size_t)rl(d.prow, jj[h]) * p.m_pad + col0)[lane];
                }
            }
        });
    };
    // ---- (2b) their gains g amp to LDS [buffer][mode]
    auto deposit = [&](const Batch &d, const int (&jj)[HB], int nh, const float (&r)[HB][NR], const f2 (&vv)[DENSE ? HB : 1]) {
        static_for<0, HB>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            if (h < nh) {
                const int j = jj[h];
                float gv = rlf(d.w[0], j) * r[h][0];
                if constexpr (DIRECT) {
                    gv = fmaf(rlf(d.w[NR > 1 ? 1 : 0], j), r[h][NR > 1 ? 1 : 0], gv);
                    gv = fmaf(rlf(d.w[NR > 2 ? 2 : 0], j), r[h][NR > 2 ? 2 : 0], gv);
                }
                if constexpr (DENSE) {
                    lds_g[2 * j][lane] = gv * vv[h].x;
                    lds_g[2 * j + 1][lane] = gv * vv[h].y;
                } else {
                    lds_g[j][lane] = gv;
                }
            }
        });
    };
    // x <- A^513 x + gv A^512 u, two dependent operations deep (a lone wave issues a dependent instruction every ~8 cycles):
    // q' = (q + (P11 - 1) q) + (P12 d + gv hq), d' = P21 q + (P22 d + gv hd); the impulse's share is off the chain
    // (DENSE: the pair (gq, gd) comes ready from LDS -- g V of a dense buffer, (g amp) A^512 u of an impulse)
    auto coarse2 = [&](float gq, float gd) {
        const float qa = fmaf(s11, x.x, x.x), qb = fmaf(s12, x.y, gq);
        const float da = s21 * x.x, db = fmaf(s22, x.y, gd);
        x.x = qa + qb;
        x.y = da + db;
    };
    auto coarse = [&](float gv) { coarse2(gv * hq, gv * hd); };
    auto mark = [&]() {                              // the first buffer of a chunk: its start state
        xs_next[lane] = x;
        xs_next += p.m_pad;
    };

    Batch cur, nxt;
    float r[HB][NR];
    f2 vv[DENSE ? HB : 1];
    bool requested = false;                          // r / vv hold the rows of the first HB hits of `cur`
    if (b_lo < b_hi && PBSO_SCAN_STOP > 1) {
        decode(b_lo, cur);
        if (cur.hit_mask && PBSO_SCAN_STOP > 2) {
            int jj[HB];
            unsigned long long m = cur.hit_mask;
            const int nh = take_hits(m, jj);
            request(cur, jj, nh, r, vv);
            requested = true;
        }
    }
    for (int base = b_lo; base < b_hi && PBSO_SCAN_STOP > 1; base += BATCH) {
        const int nd = b_hi - base < BATCH ? b_hi - base : BATCH;
        const bool more = base + BATCH < b_hi;
        if (PBSO_SCAN_STOP <= 2) {
            x.x += (float)(cur.hit_mask ^ cur.skip_mask ^ cur.dense_mask ^ cur.mark_mask) + cur.w[0] + (float)cur.prow + (float)cur.ptr;
            if (more) decode(base + BATCH, cur);
            continue;
        }
        // ---- (2) the gains of the batch's hits, [buffer][mode] in LDS
        wave_sync();                                 // (the previous batch's reads are done)
        {
            f4 *z = reinterpret_cast<f4 *>(&lds_g[0][0]) + lane;
            const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < (DENSE ? 2 : 1) * ((nd + 3) / 4); ++i) z[64 * i] = zero;
        }
        wave_sync();
        for (unsigned long long m = cur.hit_mask; m;) {
            int jj[HB];
            const int nh = take_hits(m, jj);
            if (!requested) request(cur, jj, nh, r, vv);     // (a batch's hits behind its first HB: requested here)
            requested = false;
            deposit(cur, jj, nh, r, vv);
        }
        wave_sync();
        // ---- the next batch: decoded, its first rows on their way while this batch's chain runs
        if (more) {
            decode(base + BATCH, nxt);
            if (nxt.hit_mask) {
                int jj[HB];
                unsigned long long m = nxt.hit_mask;
                const int nh = take_hits(m, jj);
                request(nxt, jj, nh, r, vv);
                requested = true;
            }
        }
        if (PBSO_SCAN_STOP <= 3) {
            x.x += lds_g[lane & (BATCH - 1)][lane];
            if (more) cur = nxt;
            continue;
        }

        // ---- (3) the scan, lane = mode: the gains of a group of G buffers are read from LDS TWO groups ahead (two register sets)
        const unsigned long long slow_mask = cur.skip_mask, mark_mask = cur.mark_mask;
        if constexpr (SEG) n_stepped += nd - __builtin_popcountll(slow_mask);
        auto fetch = [&](int j0, float (&gv)[G], float (&gw)[DENSE ? G : 1]) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                if constexpr (DENSE) {
                    gv[i] = lds_g[2 * (j0 + i)][lane];
                    gw[i] = lds_g[2 * (j0 + i) + 1][lane];
                } else {
                    gv[i] = lds_g[j0 + i][lane];
                }
            }
        };
        auto generic = [&](int j0, int n) {          // chunk starts, skipped buffers, the batch's tail
#pragma unroll 1
            for (int j = j0; j < j0 + n; ++j) {
                if ((mark_mask >> j) & 1) mark();
                if ((slow_mask >> j) & 1) continue;          // step() returned before stepping: state untouched
                if constexpr (DENSE) coarse2(lds_g[2 * j][lane], lds_g[2 * j + 1][lane]);
                else coarse(lds_g[j][lane]);
            }
        };
        // a full group from registers; then the registers take the group two ahead
        auto group = [&](int j0, float (&gv)[G], float (&gw)[DENSE ? G : 1]) {
            if (((slow_mask >> j0) & ((1ull << G) - 1)) == 0) {
                auto step = [&](int i) {
                    if constexpr (DENSE) coarse2(gv[i], gw[i]);
                    else coarse(gv[i]);
                };
                const unsigned mm = (unsigned)(mark_mask >> j0) & ((1u << G) - 1u);
                if (mm == 0) {
#pragma unroll
                    for (int i = 0; i < G; ++i) step(i);
                } else {
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        if (mm & (1u << i)) mark();
                        step(i);
                    }
                }
            } else {
                generic(j0, G);
            }
            if (j0 + 3 * G <= nd) fetch(j0 + 2 * G, gv, gw);
        };
        float ga[G], gb[G], wa[DENSE ? G : 1], wb[DENSE ? G : 1];
        if (nd >= G) fetch(0, ga, wa);
        if (nd >= 2 * G) fetch(G, gb, wb);
        int j0 = 0;
#pragma unroll 1
        for (; j0 + 2 * G <= nd; j0 += 2 * G) {
            group(j0, ga, wa);
            group(j0 + G, gb, wb);
        }
        if (j0 + G <= nd) {
            group(j0, ga, wa);
            j0 += G;
        }
        if (j0 < nd) generic(j0, nd - j0);
        if (more) cur = nxt;
    }
    if constexpr (!SEG) {
        (p_sq + ubase)[lane] = x.x;
        (p_sd + ubase)[lane] = x.y;
        (p_ss + ubase)[lane] = 1.f;
    } else {
        row64 *lds_map = lds_map_all + (size_t)wv * 6;
        {
            // M = (A^513)^(buffers stepped): square-and-multiply in fp64 from the f32 constants the chain uses (a chunk of 430 buffers: nine
            // squarings; two more chains beside the first tripled the kernel's work and made a throughput-bound scan slower)
            double b00 = 1.0 + (double)s11, b01 = (double)s12, b10 = (double)s21, b11 = (double)s22;      // the running power of A^513
            double m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;
            for (int e = n_stepped; e > 0; e >>= 1) {
                if (e & 1) {
                    const double t00 = b00 * m00 + b01 * m10, t01 = b00 * m01 + b01 * m11;
                    const double t10 = b10 * m00 + b11 * m10, t11 = b10 * m01 + b11 * m11;
                    m00 = t00; m01 = t01; m10 = t10; m11 = t11;
                }
                const double q00 = b00 * b00 + b01 * b10, q01 = b00 * b01 + b01 * b11;
                const double q10 = b10 * b00 + b11 * b10, q11 = b10 * b01 + b11 * b11;
                b00 = q00; b01 = q01; b10 = q10; b11 = q11;
            }
            lds_map[0][lane] = (float)m00; lds_map[1][lane] = (float)m10;        // M e1
            lds_map[2][lane] = (float)m01; lds_map[3][lane] = (float)m11;        // M e2
            lds_map[4][lane] = x.x; lds_map[5][lane] = x.y;
        }
        if (lane == 0) { lds_row_all[2 * wv] = cur_row != XFER_KEEP; lds_row_all[2 * wv + 1] = cur_row; }
        __syncthreads();
        // ---- compose the maps of the chunks in front of this one
        {
            const float s0 = (p_ss + ubase)[lane];       // the arrays hold scale x state (kernels_iir.hip, "scaled state")
            x.x = (p_sq + ubase)[lane] / s0;
            x.y = (p_sd + ubase)[lane] / s0;
        }
        auto apply = [&](int c, f2 v) {
            const row64 *mp = lds_map_all + (size_t)c * 6;
            const float q = fmaf(mp[0][lane], v.x, fmaf(mp[2][lane], v.y, mp[4][lane]));
            const float d = fmaf(mp[1][lane], v.x, fmaf(mp[3][lane], v.y, mp[5][lane]));
            return f2{q, d};
        };
        int row = p_xfer_init[obj];
        for (int c = 0; c < wv; ++c) {
            x = apply(c, x);
            if (lds_row_all[2 * c]) row = lds_row_all[2 * c + 1];
        }
        (reinterpret_cast<f2 *>(p_xs) + ((size_t)obj * p.n_chunks + wv) * p.m_pad + col0)[lane] = x;
        if (col0 == 0 && lane == 0) p_xtrow[(size_t)obj * p.n_chunks + wv] = row;
        __syncthreads();                                 // (every wave has read the launch's start state)
        if (wv == p.n_chunks - 1) {
            x = apply(wv, x);
            (p_sq + ubase)[lane] = x.x;
            (p_sd + ubase)[lane] = x.y;
            (p_ss + ubase)[lane] = 1.f;
        }
    }
}

#define PBSO_SCAN_ARGS                                                                                                          \
    float *__restrict__ p_sq, float *__restrict__ p_sd, float *__restrict__ p_ss, const float *__restrict__ p_sc,               \
        const BufDesc *__restrict__ p_desc, const float *__restrict__ p_grows, const float *__restrict__ p_g32,                 \
        const long long *__restrict__ p_g32_off, const float *__restrict__ p_vinc, const int *__restrict__ p_xfer_init,         \
        float *__restrict__ p_xs, int *__restrict__ p_xtrow, const ScanDims p

template <bool DIRECT, bool DENSE>
__global__ __launch_bounds__(64) void iir_scan_kernel(PBSO_SCAN_ARGS) {
    prep_prio();
    scan_body<DIRECT, DENSE, false>(p_sq, p_sd, p_ss, p_sc, p_desc, p_grows, p_g32, p_g32_off, p_vinc, p_xfer_init, p_xs, p_xtrow, p);
}

template <bool DIRECT, bool DENSE>
__global__ __launch_bounds__(64 * SEG_MAX) void iir_scan_seg_kernel(PBSO_SCAN_ARGS) {
    prep_prio();
    scan_body<DIRECT, DENSE, true>(p_sq, p_sd, p_ss, p_sc, p_desc, p_grows, p_g32, p_g32_off, p_vinc, p_xfer_init, p_xs, p_xtrow, p);
}
#undef PBSO_SCAN_ARGS

// ---------------------------------------------------------------------------------------------------------
// dense_increment_kernel: V[row][mode] = sum_{k=0..512} A^(512-k) u T_row[k], the state a unit force gain with the dense time
// profile of row `row` (one (object, buffer) of the launch: ProfRow, kernels.h) leaves behind from rest -- for ALL dense rows of
// a launch at once: grid (groups of RB rows, tiles of NW x 64 columns) -- the rows on x: a launch of a large all-dense scene has
// more row groups than grid.y may count (65535) --, one wave per (64 columns, group of rows).
//   per group of 16 blocks of 16 samples:  U[16 blocks][16 modes] = T[16 blocks x 16 taps] . F[16 taps x 16 modes]  per tile of
//   16 modes and state component on v_mfma_f32_16x16x4_f32 (F = A^(15-i) u, the table of the forced block path; the A operand
//   is the profile as the FIR of kernels_block.hip holds it), back to "lane = mode" through LDS, then
//   v <- P v + U_n, P = A^16, sixteen times; the buffer's first sample enters as v = T_0 u.
// 64 MFMAs + 32 coarse steps per (64 modes, buffer): the matrix pipe's half of what the oscillator bank spends on the same
// buffer, and none of it sequential across buffers.
constexpr int U_ROW = 36;        // LDS: [64 modes][16 Q increments | 16 D increments] + 4 floats of padding (kernels_block.hip: FTM)
constexpr int INC_NW = 4;        // waves per workgroup
constexpr int INC_RB = 4;        // rows per wave (the mode constants -- 32 operand registers -- are loaded once per object)

struct IncDims {
    int n_rows, m_pad, b_pad, frames;
    long long plane;             // elements between the planes of p_pc / p_ftab
};

__global__ __launch_bounds__(64 * INC_NW) void dense_increment_kernel(
    const float *__restrict__ p_pc, const float *__restrict__ p_ftab, const float *__restrict__ p_tprof,
    const int *__restrict__ p_row_obj, const int *__restrict__ p_n_modes, float *__restrict__ p_vinc, const IncDims p) {
    prep_prio();
    __shared__ __attribute__((aligned(16))) float lds_u[INC_NW][64 * U_ROW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col0 = 64 * (blockIdx.y * INC_NW + wave);
    if (col0 >= p.m_pad) return;                     // (no workgroup barrier in this kernel: a wave may leave)
    float *ua = lds_u[wave];
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    int cur_obj = -1;
    bool tile_dead = false;
    float fB[4][2][4];                               // B operand: F[tap 4 ks + (l >> 4)][mode 16 tl + (l & 15)], both components
    f2 c1 = f2{0.f, 0.f}, c2 = f2{0.f, 0.f};         // P = A^16 as (P11 - 1, P21), (P12, P22)
    const int r_end = (int)(blockIdx.x + 1) * INC_RB < p.n_rows ? (int)(blockIdx.x + 1) * INC_RB : p.n_rows;
    for (int row = blockIdx.x * INC_RB; row < r_end; ++row) {
        const int obj = p_row_obj[row];
        f2 *__restrict__ vdst = reinterpret_cast<f2 *>(p_vinc) + (size_t)row * p.m_pad + col0;
        if (obj != cur_obj) {
            cur_obj = obj;
            tile_dead = col0 >= p_n_modes[obj];      // columns behind the object's modes: zero coefficients, zero gains
            const size_t ubase = (size_t)obj * p.m_pad + col0;
            if (!tile_dead) {
#pragma unroll
                for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
                            fB[tl][c][ks] = (p_ftab + (size_t)(2 * (4 * ks + (lane >> 4)) + c) * p.plane + ubase)[16 * tl + (lane & 15)];
                c1.x = (p_pc + ubase)[lane];
                c2.x = (p_pc + p.plane + ubase)[lane];
                c1.y = (p_pc + 2 * p.plane + ubase)[lane];
                c2.y = (p_pc + 3 * p.plane + ubase)[lane];
            }
        }
        if (tile_dead) {
            vdst[lane] = f2{0.f, 0.f};
            continue;
        }
        const float *__restrict__ tprow = p_tprof + (size_t)row * p.b_pad;
        // the profile as the A operand, A[block l & 15][tap 4 ks + (l >> 4)] = T[1 + 256 grp + 16 block + tap], both groups
        float fa[2][4];
#pragma unroll
        for (int grp = 0; grp < 2; ++grp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[grp][ks] = tprow[1 + 256 * grp + 16 * (lane & 15) + 4 * ks + (lane >> 4)];
        const float t0 = tprow[0];
        f2 v = f2{t0, t0};                           // sample 0: d += T_0, q += d from rest
        static_for<0, 2>([&](auto gc) {
            constexpr int grp = decltype(gc)::value;
            static_for<0, 4>([&](auto tc) {
                constexpr int tl = decltype(tc)::value;
                f4 dq = f4{0.f, 0.f, 0.f, 0.f}, dd = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    dq = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[grp][ks], fB[tl][0][ks], dq, 0, 0, 0);
                    dd = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[grp][ks], fB[tl][1][ks], dd, 0, 0, 0);
                }
                // D[block 4 (l >> 4) + v][mode 16 tl + (l & 15)]
                float *dst = ua + (16 * tl + (lane & 15)) * U_ROW + 4 * (lane >> 4);
                *reinterpret_cast<f4 *>(dst) = dq;
                *reinterpret_cast<f4 *>(dst + 16) = dd;
            });
            wave_sync();
            f4 uq[4], ud[4];
            {
                const f4 *src = reinterpret_cast<const f4 *>(ua + lane * U_ROW);
#pragma unroll
                for (int i = 0; i < 4; ++i) { uq[i] = src[i]; ud[i] = src[4 + i]; }
            }
            wave_sync();                             // (read before the next group's tiles land)
            static_for<0, 16>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                const float qa = fmaf(c1.x, v.x, v.x);
                const float da = c1.y * v.x;
                const float qn = fmaf(c2.x, v.y, qa);
                const float dn = fmaf(c2.y, v.y, da);
                v.x = qn + uq[n / 4][n % 4];
                v.y = dn + ud[n / 4][n % 4];
            });
        });
        vdst[lane] = v;
    }
}

}  // namespace iir_scan

int launch_dense_increments(const float *pc, const float *ftab, long long plane, const float *tprof, const int *row_obj,
                            const int *n_modes, int n_rows, int m_pad, int b_pad, int frames, float *vinc, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    if (frames != 513 || m_pad % 64 || !ftab || !pc || !vinc) return (int)hipErrorInvalidValue;
    const iir_scan::IncDims dims = {n_rows, m_pad, b_pad, frames, plane};
    const dim3 grid((n_rows + iir_scan::INC_RB - 1) / iir_scan::INC_RB, (m_pad / 64 + iir_scan::INC_NW - 1) / iir_scan::INC_NW);
    if (grid.y > 65535u) return (int)hipErrorInvalidValue;        // (m_pad > 16 M columns)
    hipLaunchKernelGGL(iir_scan::dense_increment_kernel, grid, dim3(64 * iir_scan::INC_NW), 0, stream, pc, ftab, tprof, row_obj, n_modes,
                       vinc, dims);
    return (int)hipGetLastError();
}

int launch_iir_scan(const IirParams &p, int n_obj, const float *sc, int cb, int n_chunks, float *xs, int *xtrow, bool direct,
                    const float *vinc, bool segmented, hipStream_t stream) {
    if (n_obj <= 0 || p.nb <= 0) return 0;
    if (cb <= 0 || n_chunks != (p.nb + cb - 1) / cb || p.m_pad % 64) return (int)hipErrorInvalidValue;
    const iir_scan::ScanDims dims = {p.nb, cb, n_chunks, p.m_pad, p.b_pad, p.frames, p.gq_plane};
    const dim3 grid((unsigned)((size_t)(p.m_pad / 64) * n_obj)), block(64);
    if (segmented) {
        if (n_chunks < 2 || n_chunks > iir_scan::SEG_MAX) return (int)hipErrorInvalidValue;
        const dim3 bseg(64 * n_chunks);
        const size_t lds_seg = (size_t)n_chunks * ((iir_scan::SEG_ROWS + 6) * 64 * sizeof(float) + 2 * sizeof(int));
#define PBSO_SCAN_SEG(DIRECT, DENSE)                                                                                             \
    if (lds_seg > 64 * 1024) {                                                                                                  \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(iir_scan::iir_scan_seg_kernel<DIRECT, DENSE>),       \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_seg);                          \
        if (e_ != hipSuccess) return (int)e_;                                                                                   \
    }                                                                                                                           \
    hipLaunchKernelGGL((iir_scan::iir_scan_seg_kernel<DIRECT, DENSE>), grid, bseg, lds_seg, stream, p.sq, p.sd, p.ss, sc, p.desc, p.grows, p.g32, \
                       p.g32_off, vinc, p.xfer_init, xs, xtrow, dims)
        if (vinc) {
            if (direct) { PBSO_SCAN_SEG(true, true); }
            else { PBSO_SCAN_SEG(false, true); }
        } else {
            if (direct) { PBSO_SCAN_SEG(true, false); }
            else { PBSO_SCAN_SEG(false, false); }
        }
#undef PBSO_SCAN_SEG
        return (int)hipGetLastError();
    }
#define PBSO_SCAN_LAUNCH(DIRECT, DENSE)                                                                                          \
    hipLaunchKernelGGL((iir_scan::iir_scan_kernel<DIRECT, DENSE>), grid, block, (DENSE ? 128 : 64) * 64 * sizeof(float), stream, p.sq, p.sd, p.ss, sc, p.desc, p.grows, p.g32, \
                       p.g32_off, vinc, p.xfer_init, xs, xtrow, dims)
    if (vinc) {
        if (direct) PBSO_SCAN_LAUNCH(true, true);
        else PBSO_SCAN_LAUNCH(false, true);
    } else {
        if (direct) PBSO_SCAN_LAUNCH(true, false);
        else PBSO_SCAN_LAUNCH(false, false);
    }
#undef PBSO_SCAN_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace pbso
