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
#include "kernels.h"

namespace pbso {

// ---------------------------------------------------------------------------
// K3.  Mode shapes are stored vertex-major on the device: U[(3 v + c) * m_pad + m]
// (the reference is mode-major, ModeData.h:24) so that the three (or nine) rows
// a hit touches are contiguous over modes: coalesced 8-B loads.
// grid = (ceil(m_pad / 256), n_events)
__global__ __launch_bounds__(256) void modal_project_kernel(
    const ProjectEvent *__restrict__ events, const double *__restrict__ shapes,
    const long long *__restrict__ shape_off, const int *__restrict__ n_modes,
    double *__restrict__ slots, int m_pad) {
    const ProjectEvent ev = events[blockIdx.y];
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    double out = 0.0;
    if (m < n_modes[ev.obj]) {
        const double *U = shapes + shape_off[ev.obj];
        if (ev.kind == 1) {                       // vertex, tools/...:276-280
            const double *u = U + (size_t)(ev.vids[0] * 3) * m_pad + m;
            out = ev.vn[0] * u[0] + ev.vn[1] * u[m_pad] + ev.vn[2] * u[2 * (size_t)m_pad];
        } else {                                  // face, tools/...:244-251
            double acc = 0.0;
            for (int jj = 0; jj < 3; ++jj) {
                const double *u = U + (size_t)(ev.vids[jj] * 3) * m_pad + m;
                acc += ev.vn[0] * u[0] * ev.coords[jj]
                     + ev.vn[1] * u[m_pad] * ev.coords[jj]
                     + ev.vn[2] * u[2 * (size_t)m_pad] * ev.coords[jj];
            }
            out = acc;
        }
    }
    slots[(size_t)ev.slot * m_pad + m] = out;
}

int launch_modal_project(const ProjectEvent *events, int n_events, const double *shapes,
                         const long long *shape_off, const int *n_modes, double *slots,
                         int m_pad, hipStream_t stream) {
    if (n_events <= 0) return 0;
    dim3 grid((m_pad + 255) / 256, n_events);
    hipLaunchKernelGGL(modal_project_kernel, grid, dim3(256), 0, stream, events, shapes, shape_off,
                       n_modes, slots, m_pad);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scatter_rows_kernel(const double *__restrict__ src,
                                                           const int *__restrict__ dst_slot,
                                                           double *__restrict__ slots, int m_pad) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    slots[(size_t)dst_slot[blockIdx.y] * m_pad + m] = src[(size_t)blockIdx.y * m_pad + m];
}

int launch_scatter_rows(const double *src, const int *dst_slot, int n_rows, double *slots,
                        int m_pad, hipStream_t stream) {
    if (n_rows <= 0) return 0;
    dim3 grid((m_pad + 255) / 256, n_rows);
    hipLaunchKernelGGL(scatter_rows_kernel, grid, dim3(256), 0, stream, src, dst_slot, slots, m_pad);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// one forced (object, buffer) row: S = 0 + data_0 + data_1 + ... in list order
// (setZero then += , modal_solver.h:209,218), then g = (float)(c3 * S).
__global__ __launch_bounds__(256) void force_combine_kernel(
    const int *__restrict__ row_ptr, const int *__restrict__ slot_idx,
    const int *__restrict__ row_obj, const double *__restrict__ slots,
    const double *__restrict__ c3, float *__restrict__ grows, int m_pad) {
    const int row = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    double S = 0.0;
    for (int j = row_ptr[row]; j < row_ptr[row + 1]; ++j) S += slots[(size_t)slot_idx[j] * m_pad + m];
    grows[(size_t)row * m_pad + m] = (float)(c3[(size_t)row_obj[row] * m_pad + m] * S);
}

int launch_force_combine(const int *row_ptr, const int *slot_idx, const int *row_obj, int n_rows,
                         const double *slots, const double *c3, float *grows, int m_pad,
                         hipStream_t stream) {
    if (n_rows <= 0) return 0;
    dim3 grid((m_pad + 255) / 256, n_rows);
    hipLaunchKernelGGL(force_combine_kernel, grid, dim3(256), 0, stream, row_ptr, slot_idx, row_obj,
                       slots, c3, grows, m_pad);
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
__device__ __forceinline__ uint32_t minstd_next(uint32_t &x) {
    // x * 16807 mod (2^31 - 1) without a 64-bit division: 2^31 = 1 (mod M)
    const unsigned long long p = (unsigned long long)x * 16807ull;
    uint32_t r = (uint32_t)(p & 0x7FFFFFFFull) + (uint32_t)(p >> 31);
    if (r >= 2147483647u) r -= 2147483647u;
    x = r;
    return x;
}
__device__ __forceinline__ double canonical53(uint32_t &x) {
    const double R = 2147483646.0;                      // max - min + 1
    const double R2 = 4611686009837453316.0;            // (double)((long double)R * R), as libstdc++ forms it
    double sum = (double)(minstd_next(x) - 1u) * 1.0;
    sum += (double)(minstd_next(x) - 1u) * R;
    double ret = sum / R2;
    if (ret >= 1.0) ret = 0.99999999999999988898;        // nextafter(1, 0)
    return ret;
}
__device__ double normal01(ArState &s) {
    if (s.saved_available) {
        s.saved_available = 0;
        return s.saved;
    }
    double x, y, r2;
    do {
        x = 2.0 * canonical53(s.x) - 1.0;
        y = 2.0 * canonical53(s.x) - 1.0;
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    const double mult = sqrt(-2 * log(r2) / r2);
    s.saved = x * mult;
    s.saved_available = 1;
    return y * mult;
}

// one thread per chain (= object with dense rows in this launch); scratch[chain][frames] fp64
__global__ __launch_bounds__(64) void force_profile_kernel(
    const int *__restrict__ chain_ptr, int n_chains, const ProfRow *__restrict__ rows,
    const ProfEntry *__restrict__ entries, ArState *__restrict__ states, double *__restrict__ scratch,
    float *__restrict__ tprof, int frames, int b_pad) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    double *T = scratch + (size_t)c * frames;
    for (int ri = chain_ptr[c]; ri < chain_ptr[c + 1]; ++ri) {
        const ProfRow row = rows[ri];
        float *out = tprof + (size_t)row.prow * b_pad;
        // one contributing force (the common case): samples go straight to the fp32 row,
        // fire-and-forget stores.  Several: accumulate in fp64 in list order (modal_solver.h:206-218).
        const bool single = row.entry_end - row.entry_begin == 1;
        if (!single)
            for (int i = 0; i < frames; ++i) T[i] = 0.0;
        for (int ei = row.entry_begin; ei < row.entry_end; ++ei) {
            const ProfEntry e = entries[ei];
            if (e.kind == 0) {                                           // PointForce, forces.h:81-90
                if (single) {
                    out[0] = 1.f;
                    for (int i = 1; i < frames; ++i) out[i] = 0.f;
                } else {
                    T[0] += 1.;
                }
            } else if (e.kind == 1) {                                    // GaussianForce, forces.h:92-105
                for (int ii = 0; ii < frames; ++ii) {
                    const double z = (double)(e.count + ii - e.center) / (double)e.width_samples;
                    const double v = exp(-0.5 * (z * z));
                    if (single) out[ii] = (float)(0.0 + v);
                    else T[ii] += v;
                }
            } else {                                                     // AutoregressiveForce, :107-128
                ArState s = states[e.state];
                if (e.flags & 1) {                                       // default-constructed, forces.h:73-76
                    s.x = 1u; s.saved_available = 0; s.saved = 0.0;
                    s.buf[0] = s.buf[1] = s.buf[2] = 0.0; s.buf_idx = 0;
                    s.a[0] = 0.783; s.a[1] = 0.116; s.sigma = 0.00148; s.mu = 0.142;
                }
                if (e.flags & 2) {                                       // SetParam, forces.h:130-137
                    s.buf[0] = s.buf[1] = s.buf[2] = 0.0;
                    s.a[0] = e.a0; s.a[1] = e.a1; s.sigma = e.sigma; s.mu = e.mu;
                }
                double b0 = s.buf[0], b1 = s.buf[1], b2 = s.buf[2];
                int idx = s.buf_idx;
                for (int ii = 0; ii < frames; ++ii) {
                    // _buf[(idx + 3 - jj - 1) % 3] for jj = 0, 1
                    const double p1 = idx == 0 ? b2 : (idx == 1 ? b0 : b1);
                    const double p2 = idx == 0 ? b1 : (idx == 1 ? b2 : b0);
                    double mu_tilde = 0.0;
                    mu_tilde += s.a[0] * p1;
                    mu_tilde += s.a[1] * p2;
                    mu_tilde += s.sigma * normal01(s);
                    if (idx == 0) b0 = mu_tilde; else if (idx == 1) b1 = mu_tilde; else b2 = mu_tilde;
                    idx = idx == 2 ? 0 : idx + 1;
                    const double v = s.mu + mu_tilde;
                    if (single) out[ii] = (float)(0.0 + v);
                    else T[ii] += v;
                }
                s.buf[0] = b0; s.buf[1] = b1; s.buf[2] = b2;
                s.buf_idx = idx;
                states[e.state] = s;
            }
        }
        if (!single)
            for (int i = 0; i < frames; ++i) out[i] = (float)T[i];
        for (int i = frames; i < b_pad; ++i) out[i] = 0.f;
    }
}

int launch_force_profiles(const int *chain_ptr, int n_chains, const ProfRow *rows, const ProfEntry *entries,
                          ArState *states, double *scratch, float *tprof, int frames, int b_pad,
                          hipStream_t stream) {
    if (n_chains <= 0) return 0;
    hipLaunchKernelGGL(force_profile_kernel, dim3((n_chains + 63) / 64), dim3(64), 0, stream, chain_ptr,
                       n_chains, rows, entries, states, scratch, tprof, frames, b_pad);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
// K4.  One thread per (listener event, mode).
__device__ __forceinline__ double dmin_(double x, double y) { return (y < x) ? y : x; }  // std::min
__device__ __forceinline__ double dmax_(double x, double y) { return (x < y) ? y : x; }  // std::max

__device__ double ffat_get_map_val(const FfatGeom &g, const double *__restrict__ psi,
                                   const double p[3]) {
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
    const double c0 = (1.0 - tx) * (1.0 - ty), c1 = tx * (1.0 - ty), c2 = (1.0 - tx) * ty, c3 = tx * ty;
    const int base = g.strides[face];
    // GetMapVal :1198-1204 with GetDataQuadStride :141-144
    double psi0 = 0.0;
    psi0 += c0 * psi[base + x * Ny + y];
    psi0 += c1 * psi[base + xp * Ny + y];
    psi0 += c2 * psi[base + x * Ny + yp];
    psi0 += c3 * psi[base + xp * Ny + yp];
    // Reconstruct :899-906; (p - _center).norm() = sqrt(x^2 + (y^2 + z^2)) (Eigen's unrolled redux)
    const double dx = p[0] - g.center3[0], dy = p[1] - g.center3[1], dz = p[2] - g.center3[2];
    const double r = sqrt(dx * dx + (dy * dy + dz * dz));
    const double kr = g.k * r;
    return fabs(psi0 / kr);
}

// grid = (ceil(m_pad / 128), n_events)
__global__ __launch_bounds__(128) void ffat_lookup_kernel(
    const FfatEvent *__restrict__ events, const FfatGeom *__restrict__ geom,
    const long long *__restrict__ geom_off, const int *__restrict__ n_modes,
    const double *__restrict__ psi, double *__restrict__ rows, int m_pad) {
    const FfatEvent ev = events[blockIdx.y];
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
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
    dim3 grid((m_pad + 127) / 128, n_events);
    hipLaunchKernelGGL(ffat_lookup_kernel, grid, dim3(128), 0, stream, events, geom, geom_off,
                       n_modes, psi, rows, m_pad);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_rows_kernel(const int *__restrict__ src_row,
                                                        const int *__restrict__ dst_row,
                                                        double *rows, int m_pad) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_pad) return;
    rows[(size_t)dst_row[blockIdx.y] * m_pad + m] = rows[(size_t)src_row[blockIdx.y] * m_pad + m];
}

int launch_copy_rows(const int *src_row, const int *dst_row, int n, double *rows, int m_pad,
                     hipStream_t stream) {
    if (n <= 0) return 0;
    dim3 grid((m_pad + 255) / 256, n);
    hipLaunchKernelGGL(copy_rows_kernel, grid, dim3(256), 0, stream, src_row, dst_row, rows, m_pad);
    return (int)hipGetLastError();
}

}  // namespace pbso
