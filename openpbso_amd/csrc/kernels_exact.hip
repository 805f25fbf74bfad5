// fp64 helper kernels that feed the oscillator bank (gfx950).  Built with
// -ffp-contract=off: they repeat the reference's double arithmetic in the
// reference's operation order, so their results are bit-comparable with the
// CPU oracle (the reference's own build sets no FMA/arch flags).
//
//   K3 modal_project   GetModalForceVertex / GetModalForceFace
//                      (tools/real_time_modal_sound.cpp:268-280, 236-252)
//   scatter_rows       ForceMessage::data handed over by the host
//   force_combine      S = sum of active data (modal_solver.h:207-221, 238-239),
//                      g = (float)(c3 * S)      (modal_integrator.h:110)
//   K4 ffat_lookup     FFAT_Map<T,3>::GetMapVal (ffat_solver.h:1180-1206) ->
//                      Intersect (:676-712), Interpolate (:736-803),
//                      GetDataQuadStride (:141-144), Reconstruct (:899-906)
//   copy_rows          _latest_transfer = trans (modal_solver.h:251)
#include <cstring>
#include "kernels.h"

namespace pbso {

// ---------------------------------------------------------------------------
// K3.  Mode shapes are stored vertex-major on the device: U[(3 v + c) * m_pad + m]
// (the reference is mode-major, ModeData.h:24) so that the three (or nine) rows
// a hit touches are contiguous over modes: coalesced 8-B loads.
// grid = (ceil(m_pad / 256), n_events)
// GetModalForceVertex / GetModalForceFace for one mode (tools/real_time_modal_sound.cpp:276-280, 244-251),
// the reference's operation order (this file is built without FMA contraction)
__device__ __forceinline__ double project_one(const ProjectEvent &ev, const double *__restrict__ U, int m_pad, int m) {
    if (ev.kind == 1) {                           // vertex
        const double *u = U + (size_t)(ev.vids[0] * 3) * m_pad + m;
        return ev.vn[0] * u[0] + ev.vn[1] * u[m_pad] + ev.vn[2] * u[2 * (size_t)m_pad];
    }
    double acc = 0.0;                             // face
    for (int jj = 0; jj < 3; ++jj) {
        const double *u = U + (size_t)(ev.vids[jj] * 3) * m_pad + m;
        acc += ev.vn[0] * u[0] * ev.coords[jj]
             + ev.vn[1] * u[m_pad] * ev.coords[jj]
             + ev.vn[2] * u[2 * (size_t)m_pad] * ev.coords[jj];
    }
    return acc;
}

// Row-parallel kernels below: one workgroup per (row, tile of 256 columns), the workgroups FLAT on grid.x with the tiles of a row
// adjacent -- rows are events, forced (object, buffer) pairs, transfer rows: a ten-second step of a large scene has more of them
// than grid.y may count (65535).
struct RowTile { unsigned row, tile; };
__device__ __forceinline__ RowTile row_tile_of(unsigned block, int m_pad) {
    const unsigned tiles = (unsigned)(m_pad + 255) / 256u;
    return RowTile{block / tiles, block % tiles};
}
__device__ __forceinline__ RowTile row_tile(int m_pad) { return row_tile_of(blockIdx.x, m_pad); }
static inline bool flat_grid(int m_pad, long long n_rows, dim3 *grid) {
    const long long wgs = (long long)((m_pad + 255) / 256) * n_rows;
    if (wgs > 0x7fffffffLL) return false;
    *grid = dim3((unsigned)wgs);
    return true;
}

__global__ __launch_bounds__(256) void modal_project_kernel(
    const ProjectEvent *__restrict__ events, const double *__restrict__ shapes,
    const long long *__restrict__ shape_off, const int *__restrict__ n_modes,
    double *__restrict__ slots, int m_pad) {
    prep_prio();
    const RowTile rt = row_tile(m_pad);
    const ProjectEvent ev = events[rt.row];
    const int m = rt.tile * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    double out = 0.0;
    if (m < n_modes[ev.obj]) out = project_one(ev, shapes + shape_off[ev.obj], m_pad, m);
    slots[(size_t)ev.slot * m_pad + m] = out;
}

int launch_modal_project(const ProjectEvent *events, int n_events, const double *shapes,
                         const long long *shape_off, const int *n_modes, double *slots,
                         int m_pad, hipStream_t stream) {
    if (n_events <= 0) return 0;
    dim3 grid;
    if (!flat_grid(m_pad, n_events, &grid)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(modal_project_kernel, grid, dim3(256), 0, stream, events, shapes, shape_off,
                       n_modes, slots, m_pad);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scatter_rows_kernel(const double *__restrict__ src,
                                                           const int *__restrict__ dst_slot,
                                                           double *__restrict__ slots, int m_pad) {
    const RowTile rt = row_tile(m_pad);
    const int m = rt.tile * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    slots[(size_t)dst_slot[rt.row] * m_pad + m] = src[(size_t)rt.row * m_pad + m];
}

int launch_scatter_rows(const double *src, const int *dst_slot, int n_rows, double *slots,
                        int m_pad, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    dim3 grid;
    if (!flat_grid(m_pad, n_rows, &grid)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(scatter_rows_kernel, grid, dim3(256), 0, stream, src, dst_slot, slots, m_pad);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// one forced (object, buffer) row: S = 0 + data_0 + data_1 + ... in list order
// (setZero then += , modal_solver.h:209,218), then g = (float)(c3 * S).
// A negative slot index -(e + 1) stands for projection event e evaluated here (the hit of a PointForce lives
// for one buffer: writing its row to the pool only to read it back once costs a kernel and 170 MB of traffic).
// Round 6 (launches of ONE buffer: every slot filled now is read by exactly one row): the entries -(e + 1) with e >= n_events stand for
// row e - n_events of the staged explicit data, and an event with slot >= 0 is a projection whose vector outlives the buffer -- the
// row takes the value AND leaves it in the slot pool, which is what scatter_rows_kernel / modal_project_kernel did in launches of
// their own in front of this one (two hand-overs of ~7 us in the real-time step).  Same values, same order of additions.
// (the body as a device function: the kernel below calls it with its block index, and the one launch of a real-time buffer under
//  sustained contact -- force_rows_combine_kernel, behind K2 -- with the index of a block behind the profile rows')
__device__ __forceinline__ void force_combine_body(
    unsigned block, const int *__restrict__ row_ptr, const int *__restrict__ slot_idx,
    const int *__restrict__ row_obj, const double *__restrict__ slots,
    const double *__restrict__ c3, float *__restrict__ grows, const ProjectEvent *__restrict__ direct,
    const double *__restrict__ shapes, const long long *__restrict__ shape_off, const int *__restrict__ n_modes,
    int m_pad, int n_events, const double *__restrict__ stage, const int *__restrict__ stage_slot, double *slots_w) {
    const RowTile rt = row_tile_of(block, m_pad);
    const int row = (int)rt.row;
    const int m = rt.tile * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    const int obj = row_obj[row];
    double S = 0.0;
    for (int j = row_ptr[row]; j < row_ptr[row + 1]; ++j) {
        const int si = slot_idx[j];
        if (si >= 0) {
            S += slots[(size_t)si * m_pad + m];
        } else if (-si - 1 < n_events) {
            const ProjectEvent ev = direct[-si - 1];
            const double v = m < n_modes[obj] ? project_one(ev, shapes + shape_off[obj], m_pad, m) : 0.0;
            if (ev.slot >= 0) slots_w[(size_t)ev.slot * m_pad + m] = v;
            S += v;
        } else {
            const int r = -si - 1 - n_events;
            const double v = stage[(size_t)r * m_pad + m];
            slots_w[(size_t)stage_slot[r] * m_pad + m] = v;
            S += v;
        }
    }
    grows[(size_t)row * m_pad + m] = (float)(c3[(size_t)obj * m_pad + m] * S);
}
__global__ __launch_bounds__(256) void force_combine_kernel(
    const int *__restrict__ row_ptr, const int *__restrict__ slot_idx,
    const int *__restrict__ row_obj, const double *__restrict__ slots,
    const double *__restrict__ c3, float *__restrict__ grows, const ProjectEvent *__restrict__ direct,
    const double *__restrict__ shapes, const long long *__restrict__ shape_off, const int *__restrict__ n_modes,
    int m_pad, int n_events, const double *__restrict__ stage, const int *__restrict__ stage_slot, double *slots_w) {
    prep_prio();
    force_combine_body(blockIdx.x, row_ptr, slot_idx, row_obj, slots, c3, grows, direct, shapes, shape_off, n_modes, m_pad, n_events, stage,
                       stage_slot, slots_w);
}

int launch_force_combine(const int *row_ptr, const int *slot_idx, const int *row_obj, int n_rows,
                         double *slots, const double *c3, float *grows, const ProjectEvent *direct,
                         const double *shapes, const long long *shape_off, const int *n_modes, int m_pad,
                         int n_events, const double *stage, const int *stage_slot, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    dim3 grid;
    if (!flat_grid(m_pad, n_rows, &grid)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(force_combine_kernel, grid, dim3(256), 0, stream, row_ptr, slot_idx, row_obj,
                       static_cast<const double *>(slots), c3, grows, direct, shapes, shape_off, n_modes, m_pad, n_events, stage, stage_slot, slots);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// K2.  Force::Add for Gaussian and AR(2) forces (forces.h:92-128) in fp64, and the
// libstdc++ objects AutoregressiveForce leans on (forces.h:71-72): minstd_rand0,
// generate_canonical<double,53> (two engine draws, GCC 11 bits/random.tcc) and
// normal_distribution's Marsaglia polar method.  Uniforms, products and the
// rejection test are exact IEEE operations (this file is built without FMA
// contraction), so the accepted pairs are the host's; only log/sqrt may differ
// from glibc in the last place, far below the fp32 rounding of the profile.
// a * b mod (2^31 - 1) for a, b < 2^31 (2^31 = 1 mod M: fold the 62-bit product twice)
__device__ __forceinline__ uint32_t mulmod31(uint32_t a, uint32_t b) {
    const unsigned long long p = (unsigned long long)a * b;
    unsigned long long r = (p & 0x7FFFFFFFull) + (p >> 31);
    r = (r & 0x7FFFFFFFull) + (r >> 31);
    uint32_t q = (uint32_t)r;
    if (q >= 2147483647u) q -= 2147483647u;
    return q;
}

// One workgroup of K2_THREADS per chain (= object with dense rows in this launch, rows in buffer
// order).  The serial parts of Force::Add are kept serial where the arithmetic demands it (the
// AR(2) recurrence, thread 0) and spread over the threads where it does not:
//  * normal variates: candidate pair c of a row uses engine draws 4c+1..4c+4, and the LCG
//    jumps ahead in closed form (x_n = a^n x_0 mod M): thread j of a batch evaluates candidate
//    K2_THREADS k + j, per-wave ballots plus the waves' counts order the accepted pairs exactly as
//    the sequential rejection loop would meet them, and the engine state after the row is the
//    state after the last pair used;
//  * Gaussian samples and the final fp32 stores are lane-parallel.
// LDS: nrm[frames] doubles (normal variates, then mu-tilde) + acc[frames] doubles (several
// forces in one buffer) + the new engine state.
constexpr int K2_THREADS = 256;           // candidate pairs evaluated per batch

__global__ __launch_bounds__(K2_THREADS) void force_profile_kernel(
    const int *__restrict__ chain_ptr, int n_chains, const ProfRow *__restrict__ rows,
    const ProfEntry *__restrict__ entries, ArState *__restrict__ states, float *__restrict__ tprof,
    int frames, int b_pad, int ar_serial, int high_prio) {
    extern __shared__ __attribute__((aligned(16))) double k2_lds[];
    switch (high_prio) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
    }
    double *nrm = k2_lds;
    double *acc = k2_lds + frames;
    double *sh_saved = k2_lds + 2 * frames;                    // [0] saved variate
    uint32_t *sh_u = reinterpret_cast<uint32_t *>(k2_lds + 2 * frames + 1);   // [0] engine x, [1] saved_available
    uint32_t *sh_cnt = sh_u + 2;                               // [2][waves] accepted pairs per wave, by batch parity
    const int c = blockIdx.x;
    if (c >= n_chains) return;
    // (raising the wave priority here -- s_setprio 3 -- takes the chains from 0.71 to 0.57 ms for 86 rows beside the oscillator
    //  bank of the 8 x 4096 scraping scene, and the bank's slowest team from 0.74 to 0.83 ms: nothing gained)
    const int lane = threadIdx.x;                              // 0 .. K2_THREADS-1: one candidate pair per batch
    const int wv = threadIdx.x >> 6;
    constexpr int NWV = K2_THREADS / 64;
    // a^(4 lane) and a^(4 K2_THREADS) (mod M), a = 16807
    uint32_t a4 = 16807u;
    a4 = mulmod31(a4, a4);
    a4 = mulmod31(a4, a4);                                     // a^4
    uint32_t pj = 1u, q64 = 1u;                               // a4^lane and a4^K2_THREADS by square-and-multiply
    {
        uint32_t b = a4;
        for (unsigned e = (unsigned)lane, f = (unsigned)K2_THREADS; e | f; e >>= 1, f >>= 1) {
            if (e & 1u) pj = mulmod31(pj, b);
            if (f & 1u) q64 = mulmod31(q64, b);
            b = mulmod31(b, b);
        }
    }
    const double R = 2147483646.0;                             // max - min + 1
    const double R2 = 4611686009837453316.0;                   // (double)((long double)R * R), as libstdc++ forms it

    // The AutoregressiveForce a chain keeps using stays in registers between rows (global
    // round trips per row cost more than the row's arithmetic): every thread holds the engine
    // state and parameters, thread 0 also the AR history; written back when another state
    // is needed and at the end of the chain.
    int cached = -1;
    ArState s;
    double mpow[6][4];                                         // AR(2) scan: powers of the L-sample matrix (valid while a0, a1 stand)
    bool pow_valid = false;
    // (a row's descriptor and its first entry are fetched a row ahead: two dependent misses -- the plan has just
    //  arrived from the host -- cost more than the row's arithmetic)
    const int ri_end = chain_ptr[c + 1];
    if (chain_ptr[c] >= ri_end) return;
    ProfRow row_next = rows[chain_ptr[c]];
    ProfEntry e_next = entries[row_next.entry_begin < row_next.entry_end ? row_next.entry_begin : 0];
    for (int ri = chain_ptr[c]; ri < ri_end; ++ri) {
        const ProfRow row = row_next;
        const ProfEntry e_first = e_next;
        if (ri + 1 < ri_end) {
            row_next = rows[ri + 1];
            e_next = entries[row_next.entry_begin < row_next.entry_end ? row_next.entry_begin : 0];
        }
        float *out = tprof + (size_t)row.prow * b_pad;
        for (int i = lane; i < frames; i += K2_THREADS) acc[i] = 0.0;                  // setZero, modal_solver.h:206
        __syncthreads();
        for (int ei = row.entry_begin; ei < row.entry_end; ++ei) {
            const ProfEntry e = ei == row.entry_begin ? e_first : entries[ei];
            if (e.kind == 0) {                                                 // PointForce, forces.h:81-90
                if (lane == 0) acc[0] += 1.;
            } else if (e.kind == 1) {                                          // GaussianForce, forces.h:92-105
                for (int ii = lane; ii < frames; ii += K2_THREADS) {
                    const double z = (double)(e.count + ii - e.center) / (double)e.width_samples;
                    acc[ii] += exp(-0.5 * (z * z));
                }
            } else {                                                           // AutoregressiveForce, :107-128
                if (e.state != cached) {                                       // uniform
                    if (cached >= 0 && lane == 0) states[cached] = s;
                    __syncthreads();
                    s = states[e.state];
                    cached = e.state;
                    pow_valid = false;
                }
                if (e.flags & 3) pow_valid = false;
                if (e.flags & 1) {                                             // default-constructed, forces.h:73-76
                    s.x = 1u; s.saved_available = 0; s.saved = 0.0;
                    s.buf[0] = s.buf[1] = s.buf[2] = 0.0; s.buf_idx = 0;
                    s.a[0] = 0.783; s.a[1] = 0.116; s.sigma = 0.00148; s.mu = 0.142;
                }
                if (e.flags & 2) {                                             // SetParam, forces.h:130-137
                    s.buf[0] = s.buf[1] = s.buf[2] = 0.0;
                    s.a[0] = e.a0; s.a[1] = e.a1; s.sigma = e.sigma; s.mu = e.mu;
                }
                // ---- the row's normal variates, in the order distribution(generator) returns them
                const int sa = s.saved_available ? 1 : 0;
                if (sa && lane == 0) nrm[0] = s.saved;
                const int from_pairs = frames - sa;
                const int pairs_needed = (from_pairs + 1) / 2;
                if (lane == 0) { sh_u[0] = s.x; sh_u[1] = 0u; sh_saved[0] = 0.0; }
                __syncthreads();
                uint32_t xb = s.x;                                              // state before candidate K2_THREADS k
                int acc_pairs = 0;
#ifdef PBSO_K2_ABLATE_RNG
                acc_pairs = pairs_needed;
                for (int i = lane; i < frames; i += K2_THREADS) nrm[i] = 0.25;
#endif
                for (int batch = 0; acc_pairs < pairs_needed; ++batch) {
                    uint32_t st = mulmod31(xb, pj);                             // state before candidate K2_THREADS k + lane
                    const uint32_t d1 = (st = mulmod31(st, 16807u));
                    const uint32_t d2 = (st = mulmod31(st, 16807u));
                    const uint32_t d3 = (st = mulmod31(st, 16807u));
                    const uint32_t d4 = (st = mulmod31(st, 16807u));
                    // generate_canonical<double,53>: two draws each
                    double s1 = (double)(d1 - 1u) * 1.0;
                    s1 += (double)(d2 - 1u) * R;
                    double c1 = s1 / R2;
                    if (c1 >= 1.0) c1 = 0.99999999999999988898;
                    double s2 = (double)(d3 - 1u) * 1.0;
                    s2 += (double)(d4 - 1u) * R;
                    double c2 = s2 / R2;
                    if (c2 >= 1.0) c2 = 0.99999999999999988898;
                    const double x = 2.0 * c1 - 1.0;
                    const double y = 2.0 * c2 - 1.0;
                    const double r2 = x * x + y * y;
                    const bool ok = !(r2 > 1.0 || r2 == 0.0);
                    // accepted pairs in candidate order: waves in order, lanes in order within a wave
                    const unsigned long long m = __ballot(ok);
                    uint32_t *cnt = sh_cnt + (batch & 1) * NWV;
                    if ((lane & 63) == 0) cnt[wv] = (uint32_t)__popcll(m);
                    __syncthreads();
                    int before = 0, total = 0;
                    for (int w = 0; w < NWV; ++w) {
                        const int cw = (int)cnt[w];
                        if (w < wv) before += cw;
                        total += cw;
                    }
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    const int idx = acc_pairs + before + rank;
                    if (ok && idx < pairs_needed) {
                        const double mult = sqrt(-2 * log(r2) / r2);
                        const int p0 = sa + 2 * idx;
                        nrm[p0] = y * mult;                                     // returned first
                        if (p0 + 1 < frames) nrm[p0 + 1] = x * mult;            // the cached second variate
                        if (idx == pairs_needed - 1) {
                            sh_u[0] = st;                                       // engine state after this pair's draws
                            if (p0 + 1 >= frames) { sh_u[1] = 1u; sh_saved[0] = x * mult; }
                        }
                    }
                    acc_pairs += total;
                    xb = mulmod31(xb, q64);
                }
                __syncthreads();
                // ---- AR(2), strictly sequential (forces.h:107-117): lane 0.  The products
                // sigma * n_k are formed lane-parallel first (the same rounded product the loop would
                // add); the serial part then runs in blocks of 8 with the next block's inputs already
                // in registers, so only the four dependent fp64 operations of a step remain on the
                // critical path (LDS latency and the _buf index arithmetic are off it).
                for (int ii = lane; ii < frames; ii += K2_THREADS) nrm[ii] = s.sigma * nrm[ii];
                __syncthreads();
#ifdef PBSO_K2_ABLATE_SCAN
                if (false) {
#else
                if (!ar_serial && wv == 0) {
#endif
                    // ---- AR(2) as a parallel scan (the default).  x_k = a0 x_{k-1} + a1 x_{k-2} + c_k is linear with
                    // constant coefficients within a row: lane l of wave 0 runs the recurrence over its L consecutive
                    // samples from a zero state (lane 0: from the force's history), a Kogge-Stone scan over the lanes
                    // composes the end states (every lane's map is the same matrix M = A^L, A = [[a0, a1], [1, 0]]), and the
                    // state entering a lane is added back through row 0 of A^(j+1).  All in fp64; the SAME real-number
                    // sequence as forces.h:107-117, rounded in another order -- it differs from the serial loop by a few
                    // fp64 ulp (1e-16 relative), i.e. in none but ~1e-8 of the fp32 profile samples.  The serial loop
                    // (PBSO_AR_SERIAL=1: three dependent fp64 operations per sample, ~75 shader cycles each step, 16 us per
                    // 513-sample row) bounds a sustained-contact scene; this form takes ~1 us.
                    const int l64 = lane;
                    const int L = (frames + 63) / 64;
                    const int k0 = l64 * L;
                    const double a0 = s.a[0], a1 = s.a[1];
                    const int bidx = s.buf_idx;
                    const double h1 = bidx == 0 ? s.buf[2] : (bidx == 1 ? s.buf[0] : s.buf[1]);     // _buf[(idx + 3 - 1) % 3]
                    const double h2 = bidx == 0 ? s.buf[1] : (bidx == 1 ? s.buf[2] : s.buf[0]);     // _buf[(idx + 3 - 2) % 3]
                    // (explicit fma: this file is built without contraction for the kernels that match the oracle bit for bit)
                    double y1 = l64 == 0 ? h1 : 0.0, y2 = l64 == 0 ? h2 : 0.0;
                    for (int j = 0; j < L; ++j) {
                        const int k = k0 + j;
                        const double c = k < frames ? nrm[k] : 0.0;
                        const double v = fma(a0, y1, fma(a1, y2, c));
                        if (k < frames) nrm[k] = v;
                        y2 = y1;
                        y1 = v;
                    }
                    // M^(2^i), M = A^L: the same for every lane and for every row until the parameters change
                    if (!pow_valid) {
                        double m00 = a0, m01 = a1, m10 = 1.0, m11 = 0.0;
                        for (int i = 1; i < L; ++i) {
                            const double n00 = fma(m00, a0, m01), n01 = m00 * a1, n10 = fma(m10, a0, m11), n11 = m10 * a1;
                            m00 = n00; m01 = n01; m10 = n10; m11 = n11;
                        }
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            mpow[i][0] = m00; mpow[i][1] = m01; mpow[i][2] = m10; mpow[i][3] = m11;
                            const double n00 = fma(m00, m00, m01 * m10), n01 = fma(m00, m01, m01 * m11);
                            const double n10 = fma(m10, m00, m11 * m10), n11 = fma(m10, m01, m11 * m11);
                            m00 = n00; m01 = n01; m10 = n10; m11 = n11;
                        }
                        pow_valid = true;
                    }
                    double b1 = y1, b2 = y2;                       // (x at the lane's last sample, the one before)
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const int d = 1 << i;
                        const double t1 = __shfl_up(b1, d, 64), t2 = __shfl_up(b2, d, 64);
                        if (l64 >= d) {
                            b1 = fma(mpow[i][0], t1, fma(mpow[i][1], t2, b1));
                            b2 = fma(mpow[i][2], t1, fma(mpow[i][3], t2, b2));
                        }
                    }
                    const double s1 = __shfl_up(b1, 1, 64), s2 = __shfl_up(b2, 1, 64);      // the state entering this lane
                    if (l64 > 0) {
                        double al = a0, be = a1;                   // row 0 of A^(j+1)
                        for (int j = 0; j < L; ++j) {
                            const int k = k0 + j;
                            if (k < frames) nrm[k] = fma(al, s1, fma(be, s2, nrm[k]));
                            const double na = fma(al, a0, be), nb = al * a1;
                            al = na; be = nb;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (l64 == 0) {
                        // value k of this row went to _buf[(idx + k) % 3]: the last three, back in place
                        double b[3] = {s.buf[0], s.buf[1], s.buf[2]};
                        for (int k = frames >= 3 ? frames - 3 : 0; k < frames; ++k) b[(bidx + k) % 3] = nrm[k];
                        s.buf[0] = b[0]; s.buf[1] = b[1]; s.buf[2] = b[2];
                        s.buf_idx = (bidx + frames) % 3;
                    }
                }
                if (ar_serial && lane == 0) {
                    double b0 = s.buf[0], b1 = s.buf[1], b2 = s.buf[2];
                    int idx = s.buf_idx;
                    int ii = 0;
                    if (frames >= 8) {
                        double p1 = idx == 0 ? b2 : (idx == 1 ? b0 : b1);       // _buf[(idx + 3 - 1) % 3]
                        double p2 = idx == 0 ? b1 : (idx == 1 ? b2 : b0);       // _buf[(idx + 3 - 2) % 3]
                        double p3 = 0.0;
                        double cur[8], nxt[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) cur[j] = nrm[j];
                        for (; ii + 8 <= frames; ii += 8) {
                            const bool more = ii + 16 <= frames;
                            if (more) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) nxt[j] = nrm[ii + 8 + j];
                            }
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                // forces.h:110-114 starts from 0.0: 0.0 + x == x except for the sign of
                                // a zero, which no sample can see (T += mu + mu_tilde)
                                double mu_tilde = s.a[0] * p1;
                                mu_tilde += s.a[1] * p2;
                                mu_tilde += cur[j];
                                p3 = p2; p2 = p1; p1 = mu_tilde;
                                cur[j] = mu_tilde;
                            }
#pragma unroll
                            for (int j = 0; j < 8; ++j) nrm[ii + j] = cur[j];
                            if (more) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
                            }
                        }
                        // value k of this row went to _buf[(idx + k) % 3]: the last three, back in place
                        const int s1 = (idx + ii - 1) % 3, s2 = (idx + ii - 2) % 3, s3 = (idx + ii - 3) % 3;
                        if (s1 == 0) b0 = p1; else if (s1 == 1) b1 = p1; else b2 = p1;
                        if (s2 == 0) b0 = p2; else if (s2 == 1) b1 = p2; else b2 = p2;
                        if (s3 == 0) b0 = p3; else if (s3 == 1) b1 = p3; else b2 = p3;
                        idx = (idx + ii) % 3;
                    }
                    for (; ii < frames; ++ii) {
                        const double p1 = idx == 0 ? b2 : (idx == 1 ? b0 : b1);
                        const double p2 = idx == 0 ? b1 : (idx == 1 ? b2 : b0);
                        double mu_tilde = 0.0;
                        mu_tilde += s.a[0] * p1;
                        mu_tilde += s.a[1] * p2;
                        mu_tilde += nrm[ii];
                        if (idx == 0) b0 = mu_tilde; else if (idx == 1) b1 = mu_tilde; else b2 = mu_tilde;
                        idx = idx == 2 ? 0 : idx + 1;
                        nrm[ii] = mu_tilde;
                    }
                    s.buf[0] = b0; s.buf[1] = b1; s.buf[2] = b2;
                    s.buf_idx = idx;
                }
                __syncthreads();
                if (pairs_needed > 0) {                                         // every thread: the engine after this row
                    s.x = sh_u[0];
                    s.saved_available = (int)sh_u[1];
                    s.saved = sh_saved[0];
                } else {
                    s.saved_available = 0;                                      // only the cached variate was used
                }
                for (int ii = lane; ii < frames; ii += K2_THREADS) acc[ii] += s.mu + nrm[ii];   // forceSpread(ii) += mu_e
            }
            __syncthreads();
        }
        for (int i = lane; i < b_pad; i += K2_THREADS) out[i] = i < frames ? (float)acc[i] : 0.f;
        __syncthreads();
    }
    if (cached >= 0 && lane == 0) states[cached] = s;
}

int launch_force_profiles(const int *chain_ptr, int n_chains, const ProfRow *rows, const ProfEntry *entries,
                          ArState *states, float *tprof, int frames, int b_pad, int ar_serial, int high_prio, hipStream_t stream) {
    if (n_chains <= 0) return 0;
    const size_t lds = sizeof(double) * (2 * (size_t)frames + 2) + sizeof(uint32_t) * 2 * (K2_THREADS / 64);
    hipLaunchKernelGGL(force_profile_kernel, dim3(n_chains), dim3(K2_THREADS), lds, stream, chain_ptr, n_chains, rows,
                       entries, states, tprof, frames, b_pad, ar_serial, high_prio);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// K2, row-parallel form (the default).  The chain kernel above walks an object's rows one after another because three
// things are carried from row to row: the engine state (how many candidate pairs the rejection loop consumed), the cached
// second variate and the AR(2) history.  None of them needs the walk:
//  * candidate pair j of a force's engine is draws 4j+1..4j+4 of an LCG that jumps ahead in closed form, so whether it is
//    accepted -- and the two variates it yields -- depends on j alone: `ar_variates_kernel` evaluates a launch's candidates
//    K2_SEG per workgroup, all streams and segments side by side, and compacts the accepted ones per segment in order;
//    variate n of the stream is then component (n - saved) & 1 of accepted pair (n - saved) >> 1, found through the prefix
//    sums of the segments' counts;
//  * the AR(2) recurrence is linear: the state entering use u is M^(u - e) s_e + sum_v M^(u-1-v) z_v (M = A^frames, z_v = the
//    state use v reaches from rest, e = the use at which SetParam / construction last cleared the history):
//    `ar_zero_state_kernel` (one workgroup per AR use) gathers the use's variates and produces z_v and M,
//    `force_rows_kernel` (one workgroup per profile row) folds them (Horner, <= uses-1 steps of four fma) and then runs the
//    SAME wave-level scan as the chain kernel with that history -- the real-number sequence of forces.h:107-117 once more.
// The host sizes a stream's candidate range with a margin of many standard deviations over pairs / (pi/4); should the
// accepted pairs still fall short, the workgroup that needs more continues the candidate sequence past the last segment by
// itself (the chain kernel's batch loop) -- exercised by PBSO_K2_MARGIN_PCT < 100 in the tests, never met otherwise.
struct Candidate {
    bool ok;
    double n0, n1;       // distribution(generator) returns n0 first, caches n1
    uint32_t st;         // engine state after the candidate's four draws
};
// x = engine state before the candidate (generate_canonical<double,53> twice, Marsaglia polar: GCC 11 bits/random.tcc)
__device__ __forceinline__ Candidate eval_candidate(uint32_t st) {
    const double R = 2147483646.0;                             // max - min + 1
    const double R2 = 4611686009837453316.0;                   // (double)((long double)R * R), as libstdc++ forms it
    const uint32_t d1 = (st = mulmod31(st, 16807u));
    const uint32_t d2 = (st = mulmod31(st, 16807u));
    const uint32_t d3 = (st = mulmod31(st, 16807u));
    const uint32_t d4 = (st = mulmod31(st, 16807u));
    double s1 = (double)(d1 - 1u) * 1.0;
    s1 += (double)(d2 - 1u) * R;
    double c1 = s1 / R2;
    if (c1 >= 1.0) c1 = 0.99999999999999988898;
    double s2 = (double)(d3 - 1u) * 1.0;
    s2 += (double)(d4 - 1u) * R;
    double c2 = s2 / R2;
    if (c2 >= 1.0) c2 = 0.99999999999999988898;
    const double x = 2.0 * c1 - 1.0;
    const double y = 2.0 * c2 - 1.0;
    const double r2 = x * x + y * y;
    Candidate c;
    c.ok = !(r2 > 1.0 || r2 == 0.0);
    c.st = st;
    c.n0 = 0.0; c.n1 = 0.0;
    if (c.ok) {
        const double mult = sqrt(-2 * log(r2) / r2);
        c.n0 = y * mult;
        c.n1 = x * mult;
    }
    return c;
}
__device__ __forceinline__ uint32_t powmod31(uint32_t b, unsigned e) {
    uint32_t r = 1u;
    for (; e; e >>= 1) {
        if (e & 1u) r = mulmod31(r, b);
        b = mulmod31(b, b);
    }
    return r;
}
__device__ __forceinline__ uint32_t lcg_a4() {
    uint32_t a4 = mulmod31(16807u, 16807u);
    return mulmod31(a4, a4);
}
// order of the accepted candidates of one batch of K2_THREADS: (index of this thread's among them, their number)
__device__ __forceinline__ void batch_rank(bool ok, uint32_t *cnt /*[NWV] of this batch's parity*/, int lane, int &rank, int &total) {
    constexpr int NWV = K2_THREADS / 64;
    const unsigned long long m = __ballot(ok);
    const int wv = lane >> 6;
    if ((lane & 63) == 0) cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    int before = 0;
    total = 0;
    for (int w = 0; w < NWV; ++w) {
        const int cw = (int)cnt[w];
        if (w < wv) before += cw;
        total += cw;
    }
    rank = before + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ void ar_defaults(ArState &s) {          // forces.h:73-76
    s.x = 1u; s.saved_available = 0; s.saved = 0.0;
    s.buf[0] = s.buf[1] = s.buf[2] = 0.0; s.buf_idx = 0;
    s.a[0] = 0.783; s.a[1] = 0.116; s.sigma = 0.00148; s.mu = 0.142;
}

// (bodies as device functions: the three kernels of the row-parallel form call them with their block index, and the fused kernel of
//  single-use launches -- force_rows_fused_kernel, round 6 -- calls them one after the other from the workgroup of a row)
__device__ __forceinline__ void ar_variates_body(
    int seg, int si, uint32_t (*cnt)[K2_THREADS / 64], const ArStream *__restrict__ streams, const ArState *__restrict__ states,
    ArState *__restrict__ snaps, double *__restrict__ vnorm, uint32_t *__restrict__ vstate, int *__restrict__ seg_count) {
    const int lane = threadIdx.x;
    const ArStream S = streams[si];
    const int k = seg - S.seg_base;
    if (k == 0 && lane == 0) {                     // the force as this launch finds it (read by the two kernels that follow)
        ArState s = states[S.state];
        if (S.reset) ar_defaults(s);
        snaps[si] = s;
    }
    const uint32_t a4 = lcg_a4();
    const uint32_t pj = powmod31(a4, (unsigned)lane), q = powmod31(a4, (unsigned)K2_THREADS);
    uint32_t xb = mulmod31(S.reset ? 1u : states[S.state].x, powmod31(a4, (unsigned)k * (unsigned)K2_SEG));
    double *vn = vnorm + (size_t)seg * 2 * K2_SEG;
    uint32_t *vs = vstate + (size_t)seg * K2_SEG;
    int acc = 0;
    for (int batch = 0; batch < K2_SEG / K2_THREADS; ++batch) {
        const Candidate c = eval_candidate(mulmod31(xb, pj));
        int rank, total;
        batch_rank(c.ok, cnt[batch & 1], lane, rank, total);
        if (c.ok) {
            vn[2 * (acc + rank)] = c.n0;
            vn[2 * (acc + rank) + 1] = c.n1;
            vs[acc + rank] = c.st;
        }
        acc += total;
        xb = mulmod31(xb, q);
    }
    if (lane == 0) seg_count[seg] = acc;
}

__global__ __launch_bounds__(K2_THREADS) void ar_variates_kernel(
    const int *__restrict__ seg_stream, const ArStream *__restrict__ streams, const ArState *__restrict__ states,
    ArState *__restrict__ snaps, double *__restrict__ vnorm, uint32_t *__restrict__ vstate, int *__restrict__ seg_count) {
    prep_prio();
    __shared__ uint32_t cnt[2][K2_THREADS / 64];
    ar_variates_body((int)blockIdx.x, seg_stream[blockIdx.x], cnt, streams, states, snaps, vnorm, vstate, seg_count);
}

// x_k = a0 x_{k-1} + a1 x_{k-2} + c_k over nrm[0 .. frames) in place, history (h1, h2) = (x_{-1}, x_{-2}): ONE wave, the scan
// of the chain kernel (lane l owns L consecutive samples; Kogge-Stone over the lanes' end states; the entering state
// added back through row 0 of A^(j+1))
__device__ __forceinline__ void ar_scan_wave(double *nrm, int frames, int l64, double a0, double a1, double h1, double h2) {
    const int L = (frames + 63) / 64;
    const int k0 = l64 * L;
    double y1 = l64 == 0 ? h1 : 0.0, y2 = l64 == 0 ? h2 : 0.0;
    for (int j = 0; j < L; ++j) {
        const int k = k0 + j;
        const double c = k < frames ? nrm[k] : 0.0;
        const double v = fma(a0, y1, fma(a1, y2, c));
        if (k < frames) nrm[k] = v;
        y2 = y1;
        y1 = v;
    }
    double m00 = a0, m01 = a1, m10 = 1.0, m11 = 0.0;           // A^L
    for (int i = 1; i < L; ++i) {
        const double n00 = fma(m00, a0, m01), n01 = m00 * a1, n10 = fma(m10, a0, m11), n11 = m10 * a1;
        m00 = n00; m01 = n01; m10 = n10; m11 = n11;
    }
    double b1 = y1, b2 = y2;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int d = 1 << i;
        const double t1 = __shfl_up(b1, d, 64), t2 = __shfl_up(b2, d, 64);
        if (l64 >= d) {
            b1 = fma(m00, t1, fma(m01, t2, b1));
            b2 = fma(m10, t1, fma(m11, t2, b2));
        }
        const double n00 = fma(m00, m00, m01 * m10), n01 = fma(m00, m01, m01 * m11);
        const double n10 = fma(m10, m00, m11 * m10), n11 = fma(m10, m01, m11 * m11);
        m00 = n00; m01 = n01; m10 = n10; m11 = n11;
    }
    const double s1 = __shfl_up(b1, 1, 64), s2 = __shfl_up(b2, 1, 64);      // the state entering this lane
    if (l64 > 0) {
        double al = a0, be = a1;                   // row 0 of A^(j+1)
        for (int j = 0; j < L; ++j) {
            const int k = k0 + j;
            if (k < frames) nrm[k] = fma(al, s1, fma(be, s2, nrm[k]));
            const double na = fma(al, a0, be), nb = al * a1;
            al = na; be = nb;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the parameters in force at a use: the launch's entry that last set them, or the force as the launch found it
__device__ __forceinline__ void ar_params(const ArUse &U, const ProfEntry *__restrict__ entries, const ArState &snap,
                                          double &a0, double &a1, double &sigma, double &mu) {
    a0 = snap.a[0]; a1 = snap.a[1]; sigma = snap.sigma; mu = snap.mu;          // (a constructed force: the defaults, see ar_variates_kernel)
    if (U.param_entry >= 0) {
        const ProfEntry pe = entries[U.param_entry];
        if (pe.flags & 2) { a0 = pe.a0; a1 = pe.a1; sigma = pe.sigma; mu = pe.mu; }
        else { a0 = 0.783; a1 = 0.116; sigma = 0.00148; mu = 0.142; }
    }
}

// one workgroup per AR use: c_k = sigma n_k into cbuf, the state the use reaches from rest and M = A^frames into recs;
// the last use of a stream also says where the engine stands after the launch (fins)
__device__ __forceinline__ void ar_zero_state_body(
    int use, double *k2_lds, uint32_t (*cnt)[K2_THREADS / 64], int &carry,
    const ArUse *__restrict__ uses, const ArStream *__restrict__ streams, const ProfEntry *__restrict__ entries,
    const ArState *__restrict__ snaps, const double *__restrict__ vnorm, const uint32_t *__restrict__ vstate,
    const int *__restrict__ seg_count, double *__restrict__ cbuf, ArRec *__restrict__ recs, ArFin *__restrict__ fins,
    int frames, int c_pitch) {
    double *nrm = k2_lds;
    int *pre = reinterpret_cast<int *>(k2_lds + frames);        // [n_seg + 1] accepted pairs before segment i
    const int lane = threadIdx.x, wv = lane >> 6, l64 = lane & 63;
    const ArUse U = uses[use];
    const ArStream S = streams[U.stream];
    const ArState snap = snaps[U.stream];
    double a0, a1, sigma, mu;
    ar_params(U, entries, snap, a0, a1, sigma, mu);
    if (wv == 0) {                                              // exclusive prefix sums of the stream's segment counts
        int run = 0;
        for (int base = 0; base < S.n_seg; base += 64) {
            const int i = base + l64;
            int v = i < S.n_seg ? seg_count[S.seg_base + i] : 0;
            const int own = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(v, d, 64);
                if (l64 >= d) v += t;
            }
            if (i < S.n_seg) pre[i] = run + v - own;
            run += __shfl(v, 63, 64);
        }
        if (l64 == 0) { pre[S.n_seg] = run; carry = run; }
    }
    __syncthreads();
    const int total = carry;
    const int sa0 = snap.saved_available ? 1 : 0;
    const long long n_lo = (long long)U.u * frames;            // the use's variates: n_lo .. n_lo + frames - 1 of the stream
    const long long last_n = n_lo + frames - 1 - sa0;          // (past the cached one)
    const int p_hi = last_n >= 0 ? (int)(last_n >> 1) : -1;     // last accepted pair this use touches
    const double *vn = vnorm + (size_t)S.seg_base * 2 * K2_SEG;
    auto seg_of = [&](int pair) {                               // segment holding accepted pair `pair` (< total)
        int lo = 0, hi = S.n_seg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pre[mid] <= pair) lo = mid; else hi = mid - 1;
        }
        return lo;
    };
    for (int k = lane; k < frames; k += K2_THREADS) {
        const long long n = n_lo + k - sa0;
        double v = 0.0;
        if (n < 0) v = snap.saved;
        else if ((int)(n >> 1) < total) {
            const int pair = (int)(n >> 1), sg = seg_of(pair);
            v = vn[(size_t)sg * 2 * K2_SEG + 2 * (pair - pre[sg]) + (int)(n & 1)];
        }
        nrm[k] = sigma * v;
    }
    uint32_t fin_x = snap.x;
    double fin_saved = 0.0;
    bool fin_here = false;                                      // this thread evaluated the last pair itself
    if (p_hi >= total) {                                        // (uniform) short of accepted pairs: continue the candidate sequence
        const uint32_t a4 = lcg_a4();
        const uint32_t pj = powmod31(a4, (unsigned)lane), q = powmod31(a4, (unsigned)K2_THREADS);
        uint32_t xb = mulmod31(snap.x, powmod31(a4, (unsigned)S.n_seg * (unsigned)K2_SEG));
        int accp = total;
        for (int batch = 0; accp <= p_hi; ++batch) {
            const Candidate c = eval_candidate(mulmod31(xb, pj));
            int rank, tb;
            batch_rank(c.ok, cnt[batch & 1], lane, rank, tb);
            const int idx = accp + rank;
            if (c.ok && idx <= p_hi) {
                const long long k0 = 2ll * idx + sa0 - n_lo;
                if (k0 >= 0 && k0 < frames) nrm[k0] = sigma * c.n0;
                if (k0 + 1 >= 0 && k0 + 1 < frames) nrm[k0 + 1] = sigma * c.n1;
                if (idx == p_hi) { fin_x = c.st; fin_saved = c.n1; fin_here = true; }
            }
            accp += tb;
            xb = mulmod31(xb, q);
        }
    }
    __syncthreads();
    double *crow = cbuf + (size_t)(S.use0 + U.u) * c_pitch;
    for (int k = lane; k < frames; k += K2_THREADS) crow[k] = nrm[k];
    if (U.last) {                                               // the engine after the launch: after the last pair used
        if (p_hi >= 0 && p_hi < total && lane == 0) {
            const int sg = seg_of(p_hi);
            fin_x = vstate[(size_t)(S.seg_base + sg) * K2_SEG + (p_hi - pre[sg])];
            fin_saved = vn[(size_t)sg * 2 * K2_SEG + 2 * (p_hi - pre[sg]) + 1];
            fin_here = true;
        }
        if (p_hi < 0 && lane == 0) fin_here = true;            // only the cached variate was used: the engine stands
        if (fin_here) {
            ArFin f;
            f.x = fin_x;
            f.saved_available = (p_hi >= 0 && ((last_n & 1) == 0)) ? 1 : 0;       // the pair's second variate is still cached
            f.saved = f.saved_available ? fin_saved : 0.0;
            fins[U.stream] = f;
        }
    }
    __syncthreads();
    if (wv == 0) {
        ar_scan_wave(nrm, frames, l64, a0, a1, 0.0, 0.0);
        if (l64 == 0) {
            ArRec r;
            r.z1 = nrm[frames - 1];
            r.z2 = frames >= 2 ? nrm[frames - 2] : 0.0;
            recs[S.use0 + U.u].z1 = r.z1;
            recs[S.use0 + U.u].z2 = r.z2;
        }
    } else if (lane == 64) {                                    // M = A^frames by square and multiply
        double r00 = 1.0, r01 = 0.0, r10 = 0.0, r11 = 1.0, b00 = a0, b01 = a1, b10 = 1.0, b11 = 0.0;
        for (unsigned e = (unsigned)frames; e; e >>= 1) {
            if (e & 1u) {
                const double n00 = fma(r00, b00, r01 * b10), n01 = fma(r00, b01, r01 * b11);
                const double n10 = fma(r10, b00, r11 * b10), n11 = fma(r10, b01, r11 * b11);
                r00 = n00; r01 = n01; r10 = n10; r11 = n11;
            }
            const double n00 = fma(b00, b00, b01 * b10), n01 = fma(b00, b01, b01 * b11);
            const double n10 = fma(b10, b00, b11 * b10), n11 = fma(b10, b01, b11 * b11);
            b00 = n00; b01 = n01; b10 = n10; b11 = n11;
        }
        ArRec *r = recs + S.use0 + U.u;
        r->m00 = r00; r->m01 = r01; r->m10 = r10; r->m11 = r11;
    }
}

__global__ __launch_bounds__(K2_THREADS) void ar_zero_state_kernel(
    const ArUse *__restrict__ uses, const ArStream *__restrict__ streams, const ProfEntry *__restrict__ entries,
    const ArState *__restrict__ snaps, const double *__restrict__ vnorm, const uint32_t *__restrict__ vstate,
    const int *__restrict__ seg_count, double *__restrict__ cbuf, ArRec *__restrict__ recs, ArFin *__restrict__ fins,
    int frames, int c_pitch) {
    prep_prio();
    extern __shared__ __attribute__((aligned(16))) double k2_lds[];
    __shared__ uint32_t cnt[2][K2_THREADS / 64];
    __shared__ int carry;
    ar_zero_state_body((int)blockIdx.x, k2_lds, cnt, carry, uses, streams, entries, snaps, vnorm, vstate, seg_count, cbuf, recs, fins, frames, c_pitch);
}

// one workgroup per profile row: Force::Add of every live force of the (object, buffer) in list order (forces.h:81-128)
__device__ __forceinline__ void force_rows_body(
    int row_index, double *k2_lds,
    const ProfRow *__restrict__ rows, const ProfEntry *__restrict__ entries, const ArUse *__restrict__ uses,
    const ArStream *__restrict__ streams, const ArState *__restrict__ snaps, const ArRec *__restrict__ recs,
    const ArFin *__restrict__ fins, const double *__restrict__ cbuf, ArState *__restrict__ states, float *__restrict__ tprof,
    int frames, int b_pad, int c_pitch) {
    double *nrm = k2_lds;
    double *acc = k2_lds + frames;
    const int lane = threadIdx.x, wv = lane >> 6, l64 = lane & 63;
    const ProfRow row = rows[row_index];
    float *out = tprof + (size_t)row.prow * b_pad;
    for (int i = lane; i < frames; i += K2_THREADS) acc[i] = 0.0;                  // setZero, modal_solver.h:206
    __syncthreads();
    for (int ei = row.entry_begin; ei < row.entry_end; ++ei) {
        const ProfEntry e = entries[ei];
        if (e.kind == 0) {                                                 // PointForce, forces.h:81-90
            if (lane == 0) acc[0] += 1.;
        } else if (e.kind == 1) {                                          // GaussianForce, forces.h:92-105
            for (int ii = lane; ii < frames; ii += K2_THREADS) {
                const double z = (double)(e.count + ii - e.center) / (double)e.width_samples;
                acc[ii] += exp(-0.5 * (z * z));
            }
        } else {                                                           // AutoregressiveForce, :107-128
            const ArUse U = uses[e.count];
            const ArStream S = streams[U.stream];
            const ArState snap = snaps[U.stream];
            double a0, a1, sigma, mu;
            ar_params(U, entries, snap, a0, a1, sigma, mu);
            const double *crow = cbuf + (size_t)(S.use0 + U.u) * c_pitch;
            for (int k = lane; k < frames; k += K2_THREADS) nrm[k] = crow[k];
            // the history entering this use
            const int bidx0 = snap.buf_idx;                                // _bufIdx when the launch began (0 for a constructed force)
            double h1 = 0.0, h2 = 0.0;
            int v = U.epoch_u;
            if (v < 0) {                                                   // the force's own history carries on
                h1 = bidx0 == 0 ? snap.buf[2] : (bidx0 == 1 ? snap.buf[0] : snap.buf[1]);     // _buf[(idx + 3 - 1) % 3]
                h2 = bidx0 == 0 ? snap.buf[1] : (bidx0 == 1 ? snap.buf[2] : snap.buf[0]);     // _buf[(idx + 3 - 2) % 3]
                v = 0;
            }
            for (; v < U.u; ++v) {
                const ArRec r = recs[S.use0 + v];
                const double n1 = fma(r.m00, h1, fma(r.m01, h2, r.z1));
                const double n2 = fma(r.m10, h1, fma(r.m11, h2, r.z2));
                h1 = n1; h2 = n2;
            }
            __syncthreads();
            if (wv == 0) {
                ar_scan_wave(nrm, frames, l64, a0, a1, h1, h2);
                if (U.last && l64 == 0) {                                  // the force after this launch
                    ArState s = snap;
                    const ArFin f = fins[U.stream];
                    s.x = f.x; s.saved_available = f.saved_available; s.saved = f.saved;
                    s.a[0] = a0; s.a[1] = a1; s.sigma = sigma; s.mu = mu;
                    // value k of this use went to _buf[(idx + k) % 3]: the last three, back in place
                    const int bidx = (int)(((long long)bidx0 + (long long)U.u * frames) % 3);
                    double b[3] = {0.0, 0.0, 0.0};
                    for (int k = frames - 3; k < frames; ++k) b[(bidx + k) % 3] = nrm[k];
                    s.buf[0] = b[0]; s.buf[1] = b[1]; s.buf[2] = b[2];
                    s.buf_idx = (bidx + frames) % 3;
                    states[S.state] = s;
                }
            }
            __syncthreads();
            for (int ii = lane; ii < frames; ii += K2_THREADS) acc[ii] += mu + nrm[ii];   // forceSpread(ii) += mu_e
        }
        __syncthreads();
    }
    for (int i = lane; i < b_pad; i += K2_THREADS) out[i] = i < frames ? (float)acc[i] : 0.f;
}

__global__ __launch_bounds__(K2_THREADS) void force_rows_kernel(
    const ProfRow *__restrict__ rows, const ProfEntry *__restrict__ entries, const ArUse *__restrict__ uses,
    const ArStream *__restrict__ streams, const ArState *__restrict__ snaps, const ArRec *__restrict__ recs,
    const ArFin *__restrict__ fins, const double *__restrict__ cbuf, ArState *__restrict__ states, float *__restrict__ tprof,
    int frames, int b_pad, int c_pitch) {
    prep_prio();
    extern __shared__ __attribute__((aligned(16))) double k2_lds[];
    force_rows_body((int)blockIdx.x, k2_lds, rows, entries, uses, streams, snaps, recs, fins, cbuf, states, tprof, frames, b_pad, c_pitch);
}

// Round 6: the three kernels above as ONE launch, for launches in which every AutoregressiveForce adds its samples ONCE (one use per
// stream: every launch of one buffer -- the real-time facade's step, tools/real_time_modal_sound.cpp:527-536 with a sustained contact
// on, :1127-1160).  Nothing then crosses from one profile row to another: the workgroup of a row evaluates the candidate segments of
// each of its AR forces, their zero-state uses, and then the row, through the same global scratch arrays and with the same bodies --
// bit-identical profiles, two launch hand-overs less (~7 us each in a chain of dependent launches, profiles/r04_stream_sync.txt).
__device__ __forceinline__ void force_rows_fused_body(
    double *k2_lds, uint32_t (*cnt)[K2_THREADS / 64], int &carry,
    const ProfRow *__restrict__ rows, const ProfEntry *__restrict__ entries, const ArUse *__restrict__ uses,
    const ArStream *__restrict__ streams, ArState *__restrict__ states, ArState *__restrict__ snaps, double *__restrict__ vnorm,
    uint32_t *__restrict__ vstate, int *__restrict__ seg_count, double *__restrict__ cbuf, ArRec *__restrict__ recs,
    ArFin *__restrict__ fins, float *__restrict__ tprof, int frames, int b_pad, int c_pitch) {
    const ProfRow row = rows[blockIdx.x];
    for (int ei = row.entry_begin; ei < row.entry_end; ++ei) {
        const ProfEntry e = entries[ei];
        if (e.kind < 2) continue;                                          // (Point / Gaussian: nothing to prepare)
        const int use = e.count;
        const int si = uses[use].stream;
        const ArStream S = streams[si];
        for (int k = 0; k < S.n_seg; ++k) {
            ar_variates_body(S.seg_base + k, si, cnt, streams, states, snaps, vnorm, vstate, seg_count);
            __syncthreads();                                               // (cnt is reused by the next segment)
        }
        __threadfence();                                                   // snaps, variates and counts: written by some threads, read by all
        __syncthreads();
        ar_zero_state_body(use, k2_lds, cnt, carry, uses, streams, entries, snaps, vnorm, vstate, seg_count, cbuf, recs, fins, frames, c_pitch);
        __threadfence();
        __syncthreads();
    }
    force_rows_body((int)blockIdx.x, k2_lds, rows, entries, uses, streams, snaps, recs, fins, cbuf, states, tprof, frames, b_pad, c_pitch);
}
__global__ __launch_bounds__(K2_THREADS) void force_rows_fused_kernel(
    const ProfRow *__restrict__ rows, const ProfEntry *__restrict__ entries, const ArUse *__restrict__ uses,
    const ArStream *__restrict__ streams, ArState *__restrict__ states, ArState *__restrict__ snaps, double *__restrict__ vnorm,
    uint32_t *__restrict__ vstate, int *__restrict__ seg_count, double *__restrict__ cbuf, ArRec *__restrict__ recs,
    ArFin *__restrict__ fins, float *__restrict__ tprof, int frames, int b_pad, int c_pitch) {
    prep_prio();
    extern __shared__ __attribute__((aligned(16))) double k2_lds[];
    __shared__ uint32_t cnt[2][K2_THREADS / 64];
    __shared__ int carry;
    force_rows_fused_body(k2_lds, cnt, carry, rows, entries, uses, streams, states, snaps, vnorm, vstate, seg_count, cbuf, recs, fins, tprof, frames,
                          b_pad, c_pitch);
}

// The profile rows of a one-buffer launch and its combine rows in ONE launch (round 6, second half): the two touch different arrays --
// time profiles and AR state here, spatial gains and the slot pool there -- so the workgroups [0, n_rows) take a profile row each
// (force_rows_fused_body) and the workgroups behind them a tile of a combine row (force_combine_body).  The real-time step under a
// sustained contact was upload -> profiles 14 us -> combine 3.4 us -> bank: the combine now runs beside the profiles, and one launch
// hand-over (~4 us) is gone.  Same bodies, same values.
struct CombineArgs {
    const int *row_ptr, *slot_idx, *row_obj;
    double *slots;
    const double *c3;
    float *grows;
    const ProjectEvent *direct;
    const double *shapes;
    const long long *shape_off;
    const int *n_modes;
    int m_pad, n_events;
    const double *stage;
    const int *stage_slot;
};
__global__ __launch_bounds__(K2_THREADS) void force_rows_combine_kernel(
    int n_rows, const ProfRow *__restrict__ rows, const ProfEntry *__restrict__ entries, const ArUse *__restrict__ uses,
    const ArStream *__restrict__ streams, ArState *__restrict__ states, ArState *__restrict__ snaps, double *__restrict__ vnorm,
    uint32_t *__restrict__ vstate, int *__restrict__ seg_count, double *__restrict__ cbuf, ArRec *__restrict__ recs,
    ArFin *__restrict__ fins, float *__restrict__ tprof, int frames, int b_pad, int c_pitch, CombineArgs c) {
    prep_prio();
    extern __shared__ __attribute__((aligned(16))) double k2_lds[];
    __shared__ uint32_t cnt[2][K2_THREADS / 64];
    __shared__ int carry;
    if ((int)blockIdx.x < n_rows) {
        force_rows_fused_body(k2_lds, cnt, carry, rows, entries, uses, streams, states, snaps, vnorm, vstate, seg_count, cbuf, recs, fins, tprof,
                              frames, b_pad, c_pitch);
    } else {
        force_combine_body(blockIdx.x - (unsigned)n_rows, c.row_ptr, c.slot_idx, c.row_obj, c.slots, c.c3, c.grows, c.direct, c.shapes,
                           c.shape_off, c.n_modes, c.m_pad, c.n_events, c.stage, c.stage_slot, c.slots);
    }
}
static_assert(K2_THREADS == 256, "the combine body's tiles are 256 modes wide");

int launch_force_rows_combine(const ProfRow *rows, int n_rows, const ProfEntry *entries, const ArUse *uses, const ArStream *streams,
                              int max_segs_per_stream, ArState *states, ArState *snaps, double *vnorm, uint32_t *vstate, int *seg_count,
                              double *cbuf, ArRec *recs, ArFin *fins, float *tprof, int frames, int b_pad, int c_pitch,
                              const int *row_ptr, const int *slot_idx, const int *row_obj, int n_frows, double *slots, const double *c3,
                              float *grows, const ProjectEvent *direct, const double *shapes, const long long *shape_off, const int *n_modes,
                              int m_pad, int n_events, const double *stage, const int *stage_slot, hipStream_t stream) {
    if (n_rows <= 0 || n_frows <= 0) return (int)hipErrorInvalidValue;
    const long long wgs = (long long)n_rows + (long long)((m_pad + 255) / 256) * n_frows;
    if (wgs > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    const size_t lds = std::max(sizeof(double) * (size_t)frames + sizeof(int) * ((size_t)max_segs_per_stream + 2), sizeof(double) * 2 * (size_t)frames);
    CombineArgs c{row_ptr, slot_idx, row_obj, slots, c3, grows, direct, shapes, shape_off, n_modes, m_pad, n_events, stage, stage_slot};
    hipLaunchKernelGGL(force_rows_combine_kernel, dim3((unsigned)wgs), dim3(K2_THREADS), lds, stream, n_rows, rows, entries, uses, streams, states,
                       snaps, vnorm, vstate, seg_count, cbuf, recs, fins, tprof, frames, b_pad, c_pitch, c);
    return (int)hipGetLastError();
}

int launch_force_rows(const ProfRow *rows, int n_rows, const ProfEntry *entries, const ArUse *uses, int n_uses,
                      const ArStream *streams, const int *seg_stream, int n_segs, int max_segs_per_stream, ArState *states,
                      ArState *snaps, double *vnorm, uint32_t *vstate, int *seg_count, double *cbuf, ArRec *recs, ArFin *fins,
                      float *tprof, int frames, int b_pad, int c_pitch, bool fused, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    if (fused && n_uses > 0) {
        const size_t lds = std::max(sizeof(double) * (size_t)frames + sizeof(int) * ((size_t)max_segs_per_stream + 2), sizeof(double) * 2 * (size_t)frames);
        hipLaunchKernelGGL(force_rows_fused_kernel, dim3(n_rows), dim3(K2_THREADS), lds, stream, rows, entries, uses, streams, states, snaps,
                           vnorm, vstate, seg_count, cbuf, recs, fins, tprof, frames, b_pad, c_pitch);
        return (int)hipGetLastError();
    }
    if (n_segs > 0)
        hipLaunchKernelGGL(ar_variates_kernel, dim3(n_segs), dim3(K2_THREADS), 0, stream, seg_stream, streams, states, snaps, vnorm,
                           vstate, seg_count);
    if (n_uses > 0) {
        const size_t lds = sizeof(double) * (size_t)frames + sizeof(int) * ((size_t)max_segs_per_stream + 2);
        hipLaunchKernelGGL(ar_zero_state_kernel, dim3(n_uses), dim3(K2_THREADS), lds, stream, uses, streams, entries, snaps, vnorm,
                           vstate, seg_count, cbuf, recs, fins, frames, c_pitch);
    }
    hipLaunchKernelGGL(force_rows_kernel, dim3(n_rows), dim3(K2_THREADS), sizeof(double) * 2 * (size_t)frames, stream, rows, entries,
                       uses, streams, snaps, recs, fins, cbuf, states, tprof, frames, b_pad, c_pitch);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// K4.  One thread per (listener event, mode).
__device__ __forceinline__ double dmin_(double x, double y) { return (y < x) ? y : x; }  // std::min
__device__ __forceinline__ double dmax_(double x, double y) { return (x < y) ? y : x; }  // std::max

// Where a position falls in a map (Intersect + the cell corners and weights of Interpolate) and its distance from the map's
// centre: everything of GetMapVal that does not touch Psi or k.  Split out in round 6 so that the kernel of objects whose modes
// share one geometry (ffat_lookup_shared_kernel) evaluates it once per event; the same expressions in the same order either way
// (this file is built with -ffp-contract=off: no expression is contracted differently after the split).
struct FfatLoc {
    int i00, i10, i01, i11;          // indices into the mode's Psi: (x, y), (xp, y), (x, yp), (xp, yp)
    double c0, c1, c2, c3;           // the bilinear weights, in GetMapVal's order
    double r;                        // |p - center|
};
__device__ __forceinline__ FfatLoc ffat_locate(const FfatGeom &g, const double p[3]) {
    // Intersect, ffat_solver.h:681-686
    double d[3], t_enter[3], surf[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        d[i] = g.center[i] - p[i];
        const double tmin = (g.bbox_low[i] - p[i]) / d[i];
        const double tmax = (g.bbox_top[i] - p[i]) / d[i];
        t_enter[i] = dmin_(tmin, tmax);
    }
    double t_en = t_enter[0];
    if (t_enter[1] > t_en) t_en = t_enter[1];
    if (t_enter[2] > t_en) t_en = t_enter[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) surf[i] = p[i] + t_en * d[i];
    // nearest plane, low tested before top per axis, strict '<' (:688-698)
    double min_dist = 1.7976931348623157e308;
    int face = 0;
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) {
        const double dl = fabs(g.bbox_low[dd] - surf[dd]);
        if (dl < min_dist) { min_dist = dl; face = dd * 2 + 1; }
        const double dt = fabs(g.bbox_top[dd] - surf[dd]);
        if (dt < min_dist) { min_dist = dt; face = dd * 2; }
    }
    const int dk = face / 2, di = (dk + 1) % 3, dj = (dk + 2) % 3;
    // Interpolate, :750-802 (the nearest-cell index of Intersect :705-710 is not used by GetMapVal)
    const int Nx = g.n_elements[face][0], Ny = g.n_elements[face][1];
    const double h = g.cell_size;
    const double sdi = di == 0 ? surf[0] : (di == 1 ? surf[1] : surf[2]);
    const double sdj = dj == 0 ? surf[0] : (dj == 1 ? surf[1] : surf[2]);
    const double x_float = (sdi - (g.low_corners[face][di] + 0.5 * h)) / h;
    const double y_float = (sdj - (g.low_corners[face][dj] + 0.5 * h)) / h;
    int x = (int)floor(x_float), y = (int)floor(y_float), xp, yp;
    double tx, ty;
    if (x < 0) { x = 0; xp = 0; tx = 0; }
    else if (x >= 0 && x < Nx - 1) { xp = x + 1; tx = x_float - (double)x; }
    else { x = Nx - 1; xp = Nx - 1; tx = 0; }
    if (y < 0) { y = 0; yp = 0; ty = 0; }
    else if (y >= 0 && y < Ny - 1) { yp = y + 1; ty = y_float - (double)y; }
    else { y = Ny - 1; yp = Ny - 1; ty = 0; }
    tx = dmin_(dmax_(tx, 0.0), 1.0);
    ty = dmin_(dmax_(ty, 0.0), 1.0);
    FfatLoc L;
    L.c0 = (1.0 - tx) * (1.0 - ty); L.c1 = tx * (1.0 - ty); L.c2 = (1.0 - tx) * ty; L.c3 = tx * ty;
    const int base = g.strides[face];
    L.i00 = base + x * Ny + y; L.i10 = base + xp * Ny + y; L.i01 = base + x * Ny + yp; L.i11 = base + xp * Ny + yp;
    // Reconstruct :899-906; (p - _center).norm() = sqrt(x^2 + (y^2 + z^2)) (Eigen's unrolled redux)
    const double dx = p[0] - g.center3[0], dy = p[1] - g.center3[1], dz = p[2] - g.center3[2];
    L.r = sqrt(dx * dx + (dy * dy + dz * dz));
    return L;
}
// GetMapVal :1198-1204 with GetDataQuadStride :141-144 on the four corners' values, then Reconstruct's 1 / (k r)
__device__ __forceinline__ double ffat_combine(const FfatLoc &L, double p00, double p10, double p01, double p11, double k) {
    double psi0 = 0.0;
    psi0 += L.c0 * p00;
    psi0 += L.c1 * p10;
    psi0 += L.c2 * p01;
    psi0 += L.c3 * p11;
    const double kr = k * L.r;
    return fabs(psi0 / kr);
}
__device__ double ffat_get_map_val(const FfatGeom &g, const double *__restrict__ psi, const double p[3]) {
    const FfatLoc L = ffat_locate(g, p);
    return ffat_combine(L, psi[L.i00], psi[L.i10], psi[L.i01], psi[L.i11], g.k);
}

// grid = n_events x ceil(m_pad / 128) workgroups, flat on x (tiles of one event adjacent): grid.y counts to 65535 only
__global__ __launch_bounds__(128) void ffat_lookup_kernel(
    const FfatEvent *__restrict__ events, const FfatGeom *__restrict__ geom,
    const long long *__restrict__ geom_off, const int *__restrict__ n_modes,
    const double *__restrict__ psi, double *__restrict__ rows, int m_pad) {
    const unsigned tiles = (unsigned)(m_pad + 127) / 128u;
    const FfatEvent ev = events[blockIdx.x / tiles];
    const int m = (blockIdx.x % tiles) * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    double out = 0.0;
    if (m < n_modes[ev.obj]) {
        const FfatGeom &g = geom[geom_off[ev.obj] + m];
        // computeTransfer wraps GetMapVal in another std::abs (modal_solver.h:295)
        if (g.valid) out = fabs(ffat_get_map_val(g, psi + g.psi_off, ev.pos));
    }
    rows[(size_t)ev.row * m_pad + m] = out;
}

int launch_ffat_lookup(const FfatEvent *events, int n_events, const FfatGeom *geom,
                       const long long *geom_off, const int *n_modes, const double *psi,
                       double *rows, int m_pad, hipStream_t stream) {
    if (n_events <= 0) return 0;
    const long long wgs = (long long)((m_pad + 127) / 128) * n_events;
    if (wgs > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)wgs);
    hipLaunchKernelGGL(ffat_lookup_kernel, grid, dim3(128), 0, stream, events, geom, geom_off,
                       n_modes, psi, rows, m_pad);
    return (int)hipGetLastError();
}

// K4 for a launch's listener events, the planner's list cut into RUNS of consecutive events of one object (it lists them
// object by object).  One workgroup per (run, mode): the mode's geometry is workgroup-uniform -- scalar loads, once -- and the
// threads are the run's events.  The per-event kernel above reads the mode's 232-byte geometry per LOOKUP: 325 MB through L2
// for the 1.4 M lookups of 64 x 256 x 86 with a listener move per buffer, 75 us, the longest kernel of that step.  Same
// arithmetic (ffat_get_map_val), bit-exact with the oracle.
__global__ __launch_bounds__(128) void ffat_lookup_runs_kernel(
    const FfatEvent *__restrict__ events, const FfatRun *__restrict__ runs, const FfatGeom *__restrict__ geom,
    const long long *__restrict__ geom_off, const int *__restrict__ n_modes,
    const double *__restrict__ psi, double *__restrict__ rows, int m_pad) {
    const FfatRun run = runs[blockIdx.x / (unsigned)m_pad];      // (flat grid, the modes of one run adjacent: grid.y counts to 65535 only)
    const int m = blockIdx.x % (unsigned)m_pad;
    const bool live = m < n_modes[run.obj];
    const FfatGeom &g = geom[geom_off[run.obj] + (live ? m : 0)];
    const bool on = live && g.valid;
    for (int i = threadIdx.x; i < run.count; i += blockDim.x) {
        const FfatEvent ev = events[run.first + i];
        // computeTransfer wraps GetMapVal in another std::abs (modal_solver.h:295)
        rows[(size_t)ev.row * m_pad + m] = on ? fabs(ffat_get_map_val(g, psi + g.psi_off, ev.pos)) : 0.0;
    }
}

int launch_ffat_lookup_runs(const FfatEvent *events, const FfatRun *runs, int n_runs, const FfatGeom *geom,
                            const long long *geom_off, const int *n_modes, const double *psi,
                            double *rows, int m_pad, hipStream_t stream) {
    if (n_runs <= 0) return 0;
    if ((long long)n_runs * m_pad > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(ffat_lookup_runs_kernel, dim3((unsigned)((long long)n_runs * m_pad)), dim3(128), 0, stream, events, runs, geom, geom_off,
                       n_modes, psi, rows, m_pad);
    return (int)hipGetLastError();
}

// K4 for objects whose modes all carry the SAME map geometry (bounding box, centre, cell size, face layout: what a map file's
// header holds; only k and Psi differ from mode to mode -- the engine checks it field by field at finalize, Engine::finalize).
// Round 6.  Then where a listener position falls on the cube, its four cells and their weights, is one evaluation per EVENT instead of
// one per (event, mode) -- eight of GetMapVal's nine fp64 divisions and its square root -- and with the object's maps also held
// TRANSPOSED, psi_t[cell][mode], the four corner reads of a wave are four runs of consecutive doubles and so is its write: lane =
// mode.  64 x 256 x 86 with a listener move per buffer (BASELINE configs[2]): the per-(run, mode) kernel above gathers from 201 MB of
// maps and scatters 1.4 M eight-byte writes, 35 us -- a third of that step's device time, in the window between two oscillator
// banks where nothing else can run (the bank holds every register); this one moves 10 KB per event, 55 MB.
// One workgroup per (group of FFAT_SH_EVENTS events, tile of 256 modes): the first lanes locate one event each -- together, in the
// time of one -- and leave cells, weights and distance in LDS; then every lane, a mode, walks the group's events (sixteen corner
// reads in flight).  With a workgroup per event each of its four waves located the position again: 19 us for the 5504 events of that
// step, most of it the divisions.  Events of other objects are left to the kernels above (skipped here).
// Same arithmetic (ffat_locate + ffat_combine), bit-exact with the oracle.
constexpr int FFAT_SH_EVENTS = 4;
__global__ __launch_bounds__(256) void ffat_lookup_shared_kernel(
    const FfatEvent *__restrict__ events, int n_events, const FfatShared *__restrict__ shared, const FfatGeom *__restrict__ geom,
    const long long *__restrict__ geom_off, const int *__restrict__ n_modes, const double *__restrict__ mode_k,
    const int *__restrict__ mode_valid, const double *__restrict__ psi_t, double *__restrict__ rows, int m_pad) {
    __shared__ FfatLoc loc[FFAT_SH_EVENTS];
    __shared__ int ev_obj[FFAT_SH_EVENTS], ev_row[FFAT_SH_EVENTS];
    const unsigned tiles = (unsigned)(m_pad + 255) / 256u;
    const int e0 = (int)(blockIdx.x / tiles) * FFAT_SH_EVENTS;
    if (threadIdx.x < FFAT_SH_EVENTS) {
        const int e = e0 + (int)threadIdx.x;
        int obj = -1, row = 0;
        if (e < n_events) {
            const FfatEvent ev = events[e];
            const FfatShared sh = shared[ev.obj];
            if (sh.pitch != 0) {
                obj = ev.obj;
                row = ev.row;
                loc[threadIdx.x] = ffat_locate(geom[geom_off[ev.obj] + sh.first_valid], ev.pos);
            }
        }
        ev_obj[threadIdx.x] = obj;
        ev_row[threadIdx.x] = row;
    }
    __syncthreads();
    const int m = (int)(blockIdx.x % tiles) * 256 + (int)threadIdx.x;
    if (m >= m_pad) return;
#pragma unroll
    for (int j = 0; j < FFAT_SH_EVENTS; ++j) {
        const int obj = ev_obj[j];
        if (obj < 0) continue;                      // (workgroup-uniform)
        const FfatShared sh = shared[obj];
        const long long go = geom_off[obj];
        double out = 0.0;
        if (m < n_modes[obj] && mode_valid[go + m]) {
            const FfatLoc L = loc[j];
            const double *__restrict__ pt = psi_t + sh.psit_off + m;
            const size_t pitch = (size_t)sh.pitch;
            // computeTransfer wraps GetMapVal in another std::abs (modal_solver.h:295)
            out = fabs(ffat_combine(L, pt[(size_t)L.i00 * pitch], pt[(size_t)L.i10 * pitch], pt[(size_t)L.i01 * pitch], pt[(size_t)L.i11 * pitch],
                                    mode_k[go + m]));
        }
        rows[(size_t)ev_row[j] * m_pad + m] = out;
    }
}

int launch_ffat_lookup_shared(const FfatEvent *events, int n_events, const FfatShared *shared, const FfatGeom *geom, const long long *geom_off,
                              const int *n_modes, const double *mode_k, const int *mode_valid, const double *psi_t, double *rows, int m_pad,
                              hipStream_t stream) {
    if (n_events <= 0) return 0;
    const long long wgs = (long long)((m_pad + 255) / 256) * ((n_events + FFAT_SH_EVENTS - 1) / FFAT_SH_EVENTS);
    if (wgs > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(ffat_lookup_shared_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, events, n_events, shared, geom, geom_off, n_modes,
                       mode_k, mode_valid, psi_t, rows, m_pad);
    return (int)hipGetLastError();
}

// K4, many positions of ONE object (computeTransfer(pos, T*), modal_solver.h:302-315; the HUD sphere of the tool,
// tools/real_time_modal_sound.cpp:916-927).  One workgroup per (mode, chunk of positions): the mode's map -- all six
// faces of Psi -- is staged in LDS once and every bilinear interpolation of the chunk gathers from there; the mode's
// geometry is workgroup-uniform (scalar loads).  The per-event kernel above reads 300 bytes of geometry per lookup and
// gathers Psi from L2 / HBM, which a 1024-mode object overflows (12 MB of maps against 4 MB of L2 per XCD).
// Same arithmetic (ffat_get_map_val), bit-exact with the oracle.  Maps that do not fit the LDS window are read in place.
constexpr int FFAT_LDS_DOUBLES = 7168;             // 56 KB: a 32 x 32-cell cube map is 6144 doubles
constexpr int FFAT_POS_PER_BLOCK = 1024;
__global__ __launch_bounds__(256) void ffat_batch_kernel(const double *__restrict__ pos, int n_pos, const FfatGeom *__restrict__ geom,
                                                         const double *__restrict__ psi, double *__restrict__ rows, int m_pad) {
    extern __shared__ __attribute__((aligned(16))) double lpsi[];
    const int m = blockIdx.x;
    const FfatGeom &g = geom[m];
    const bool staged = g.valid && g.n_psi <= FFAT_LDS_DOUBLES;
    if (staged)
        for (int i = threadIdx.x; i < g.n_psi; i += blockDim.x) lpsi[i] = psi[g.psi_off + i];
    __syncthreads();
    const double *src = staged ? lpsi : psi + g.psi_off;
    const int p0 = blockIdx.y * FFAT_POS_PER_BLOCK;
    const int p1 = p0 + FFAT_POS_PER_BLOCK < n_pos ? p0 + FFAT_POS_PER_BLOCK : n_pos;
    for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const double P[3] = {pos[3 * (size_t)p], pos[3 * (size_t)p + 1], pos[3 * (size_t)p + 2]};
        // computeTransfer wraps GetMapVal in another std::abs (modal_solver.h:312)
        rows[(size_t)p * m_pad + m] = g.valid ? fabs(ffat_get_map_val(g, src, P)) : 0.0;
    }
}

int launch_ffat_batch(const double *pos, int n_pos, const FfatGeom *geom_of_object, int n_maps, const double *psi,
                      double *rows, int m_pad, hipStream_t stream) {
    if (n_pos <= 0 || n_maps <= 0) return 0;
    dim3 grid(n_maps, (n_pos + FFAT_POS_PER_BLOCK - 1) / FFAT_POS_PER_BLOCK);
    hipLaunchKernelGGL(ffat_batch_kernel, grid, dim3(256), FFAT_LDS_DOUBLES * sizeof(double), stream, pos, n_pos, geom_of_object,
                       psi, rows, m_pad);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// Objects stepped by several teams: add the teams' partial sample sums, team 0 first.
// the rows of one sample, added in row order with eight loads in flight (one load at a time made the kernel latency-bound:
// 30 us for the 512 part rows of 8 x 4096 under the pipeline kernel, 3 TB/s)
__device__ __forceinline__ float sum_rows(const float *__restrict__ p, int n_rows, long long stride) {
    float acc = p[0];
    int r = 1;
    for (; r + 8 <= n_rows; r += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(r + k) * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    for (; r < n_rows; ++r) acc += p[(size_t)r * stride];
    return acc;
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const SplitObj *__restrict__ split,
                                                        const float *__restrict__ parts,
                                                        float *__restrict__ audio, long long stride, long long n) {
    const SplitObj so = split[blockIdx.y];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    audio[(size_t)so.obj * stride + i] = sum_rows(parts + (size_t)so.first_row * stride + i, so.n_rows, stride);
}

int launch_sum_parts(const SplitObj *split, int n_split, const float *parts, float *audio, long long stride, long long n,
                     hipStream_t stream) {
    if (n_split <= 0 || n <= 0) return 0;
    if (n_split > 65535) return (int)hipErrorInvalidValue;      // (objects stepped by several teams: each has more than 1024 modes)
    dim3 grid((unsigned)((n + 255) / 256), n_split);
    hipLaunchKernelGGL(sum_parts_kernel, grid, dim3(256), 0, stream, split, parts, audio, stride, n);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_rows_kernel(const int *__restrict__ src_row,
                                                        const int *__restrict__ dst_row,
                                                        double *rows, int m_pad) {
    const RowTile rt = row_tile(m_pad);
    const int m = rt.tile * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    rows[(size_t)dst_row[rt.row] * m_pad + m] = rows[(size_t)src_row[rt.row] * m_pad + m];
}

int launch_copy_rows(const int *src_row, const int *dst_row, int n, double *rows, int m_pad,
                     hipStream_t stream) {
    if (n <= 0) return 0;
    dim3 grid;
    if (!flat_grid(m_pad, n, &grid)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(copy_rows_kernel, grid, dim3(256), 0, stream, src_row, dst_row, rows, m_pad);
    return (int)hipGetLastError();
}

// The two jobs that follow an oscillator-bank launch in its stream, in ONE launch (a launch hand-over costs the stream 5 - 8 us,
// which is 3 - 4 % of a step of the small scenes): blockIdx.y < n_split adds an object's partial sums, the other rows of the
// grid copy transfer rows (_latest_transfer = trans, modal_solver.h:251).  They touch different arrays.
__global__ __launch_bounds__(256) void sum_parts_copy_rows_kernel(const SplitObj *__restrict__ split, int n_split,
                                                                  const float *__restrict__ parts, float *__restrict__ audio,
                                                                  long long stride, long long n, const int *__restrict__ src_row,
                                                                  const int *__restrict__ dst_row, double *rows, int m_pad) {
    if ((int)blockIdx.y < n_split) {
        const SplitObj so = split[blockIdx.y];
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n) return;
        audio[(size_t)so.obj * stride + i] = sum_rows(parts + (size_t)so.first_row * stride + i, so.n_rows, stride);
    } else {
        const int c = (int)blockIdx.y - n_split;
        const int m = blockIdx.x * blockDim.x + threadIdx.x;
        if (m >= m_pad) return;
        rows[(size_t)dst_row[c] * m_pad + m] = rows[(size_t)src_row[c] * m_pad + m];
    }
}

int launch_sum_parts_copy_rows(const SplitObj *split, int n_split, const float *parts, float *audio, long long stride, long long n,
                               const int *src_row, const int *dst_row, int n_copy, double *rows, int m_pad, hipStream_t stream) {
    if (n_split <= 0 || n <= 0) return launch_copy_rows(src_row, dst_row, n_copy, rows, m_pad, stream);
    if (n_copy <= 0) return launch_sum_parts(split, n_split, parts, audio, stride, n, stream);
    if (n_split + (long long)n_copy > 65535) {        // (grid.y: two launches then; the copy's grid is flat)
        const int e = launch_sum_parts(split, n_split, parts, audio, stride, n, stream);
        return e ? e : launch_copy_rows(src_row, dst_row, n_copy, rows, m_pad, stream);
    }
    dim3 grid((unsigned)std::max<long long>((n + 255) / 256, (m_pad + 255) / 256), n_split + n_copy);
    hipLaunchKernelGGL(sum_parts_copy_rows_kernel, grid, dim3(256), 0, stream, split, n_split, parts, audio, stride, n, src_row, dst_row,
                       rows, m_pad);
    return (int)hipGetLastError();
}

// ---- sum over objects of a step's audio (pbso_mix_objects): out[i] = sum_o audio[o][i], two stages of fixed order --
// groups of MIX_GROUP consecutive objects are summed side by side (one workgroup per group and 1024 samples), then the
// groups' partial rows in group order: the result does not depend on the launch shape.  HBM-bound: reads the step's audio once.
constexpr int MIX_GROUP = 32;
__global__ __launch_bounds__(256) void mix_objects_stage1(const float *__restrict__ audio, int n_obj, long long stride, long long n,
                                                         float *__restrict__ parts) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const int o0 = blockIdx.y * MIX_GROUP, o1 = o0 + MIX_GROUP < n_obj ? o0 + MIX_GROUP : n_obj;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // rows are 8-byte aligned when the row length is even (n_buffers * 513 with an even n_buffers): two 8-byte loads per object,
    // eight objects in flight; the adds stay in object order
    const bool wide = (stride & 1) == 0 && i + 3 < n;
    if (wide) {
        int o = o0;
        for (; o + 8 <= o1; o += 8) {
            f2 v[8][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f2 *row = reinterpret_cast<const f2 *>(audio + (long long)(o + k) * stride + i);
                v[k][0] = row[0];
                v[k][1] = row[1];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[0] += v[k][0].x; acc[1] += v[k][0].y; acc[2] += v[k][1].x; acc[3] += v[k][1].y;
            }
        }
        for (; o < o1; ++o) {
            const f2 *row = reinterpret_cast<const f2 *>(audio + (long long)o * stride + i);
            const f2 a = row[0], b2 = row[1];
            acc[0] += a.x; acc[1] += a.y; acc[2] += b2.x; acc[3] += b2.y;
        }
    } else {
        for (int o = o0; o < o1; ++o) {
            const float *row = audio + (long long)o * stride + i;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < n) acc[k] += row[k];
        }
    }
    float *dst = parts + (long long)blockIdx.y * n + i;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i + k < n) dst[k] = acc[k];
}
__global__ __launch_bounds__(256) void mix_objects_stage2(const float *__restrict__ parts, int n_groups, long long n, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int g = 0; g < n_groups; ++g) acc += parts[(long long)g * n + i];
    out[i] = acc;
}
int mix_objects_groups(int n_obj) { return (n_obj + MIX_GROUP - 1) / MIX_GROUP; }
int launch_mix_objects(const float *audio, int n_obj, long long stride, long long n, float *parts, float *out, hipStream_t stream) {
    if (n_obj <= 0 || n <= 0) return 0;
    const int groups = mix_objects_groups(n_obj);
    hipLaunchKernelGGL(mix_objects_stage1, dim3((unsigned)((n + 1023) / 1024), groups), dim3(256), 0, stream, audio, n_obj, stride, n, parts);
    hipLaunchKernelGGL(mix_objects_stage2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, parts, groups, n, out);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(64) void signal_value_kernel(unsigned long long *sig, unsigned long long value) {
    if (threadIdx.x == 0) __hip_atomic_store(sig, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int launch_signal_value(unsigned long long *sig, unsigned long long value, hipStream_t stream) {
    hipLaunchKernelGGL(signal_value_kernel, dim3(1), dim3(64), 0, stream, sig, value);
    return (int)hipGetLastError();
}

}  // namespace pbso
