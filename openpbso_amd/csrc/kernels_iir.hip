// K1: damped-IIR oscillator bank + per-sample mode reduction (gfx950, wave64).
//
// Replaces the hot loop of ModalSolver::step (modal_solver.h:262-272) and
// ModalIntegrator::Step (modal_integrator.h:103-113) for a whole batch of
// objects and buffers in one launch.
//
// Mapping.  One workgroup = one object ("team" of W waves).  A lane owns R
// oscillators: mode m = r * (64 W) + tid, so every r-slice of the SoA arrays
// is one contiguous, coalesced row.  Coefficients, state, g = c3*S, the
// transfer weights and the qnorm accumulators live in VGPRs for the whole
// launch; the force time profile of the current tile is read with scalar
// loads (it is uniform over the team).  This loop is bound by the fp32 vector
// ALU issue rate, not HBM: per oscillator-sample it issues
//   velocity form:  v_mul, v_fma [, v_fma force], v_add, v_fma out [, v_fma qnorm]
//   direct form:    v_mul [, v_fma force], v_fma, v_fma out [, v_fma qnorm]
//
// Per-sample reduction over modes.  Each lane first sums its own R modes
// (p = sum_r t_r q_r), then the 64 lane partials of TILE consecutive samples
// are transposed through a per-wave LDS tile P[TILE][LDS_ROW]: lane l writes
// P[k][l] while stepping sample k (conflict-free), then lane k reads row k with
// 16 ds_read_b128 and adds the 64 values in a fixed order.  That costs one
// v_add per wave-sample instead of a 6-step cross-lane reduction, and the
// result is deterministic.  Teams of W > 1 waves add their row sums through a
// small double-buffered LDS array, one workgroup barrier per tile.
#include "kernels.h"

// built twice (Makefile): PBSO_IIR_NS = iir_slp (SLP vectoriser on: mode pairs
// become v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and iir_noslp
// (-fno-slp-vectorize: plain v_fma_f32).  The engine picks one at run time.
#ifndef PBSO_IIR_NS
#define PBSO_IIR_NS iir_slp
#endif

namespace pbso {
namespace PBSO_IIR_NS {

template <int R, int FORM, bool QN, bool FORCED>
__device__ __forceinline__ void step_tile(float (&q)[R], float (&d)[R], const float (&ca)[R],
                                          const float (&cb)[R], const float (&g)[R],
                                          const float (&t)[R], float (&qn)[R],
                                          const float *__restrict__ tp, float *__restrict__ col) {
#pragma unroll
    for (int k = 0; k < TILE; ++k) {
        float tk = 0.f;
        if (FORCED) tk = tp[k];
        float p = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (FORM == 0) {
                // d_k = eps^2 d_{k-1} - e q_{k-1} + g T_k ;  q_k = q_{k-1} + d_k
                float a = ca[r] * d[r];
                a = fmaf(-cb[r], q[r], a);
                if (FORCED) a = fmaf(g[r], tk, a);
                d[r] = a;
                q[r] = q[r] + a;
            } else {
                // q_k = c1 q_{k-1} + c2 q_{k-2} + g T_k   (d holds q_{k-2})
                float a = cb[r] * d[r];
                if (FORCED) a = fmaf(g[r], tk, a);
                const float qk = fmaf(ca[r], q[r], a);
                d[r] = q[r];
                q[r] = qk;
            }
            p = (r == 0) ? t[r] * q[r] : fmaf(t[r], q[r], p);
            if (QN) qn[r] = fmaf(q[r], q[r], qn[r]);
        }
        col[k * LDS_ROW] = p;
        if (QN) {
            // pin the qnorm accumulators here: without it the q^2 FMAs of a whole
            // tile are sunk to the tile's end and 57*R q values stay live.
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(qn[r]));
        }
        // keep the scheduler from interleaving whole samples (it otherwise keeps
        // hundreds of q values live to batch the qnorm chain): one wave issues a
        // VALU every 4 cycles anyway, the R modes of one sample are ILP enough.
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Pointers are separate __restrict__ kernel arguments (not struct members) so
// that the uniform loads of descriptors and force profiles are provably
// un-clobbered by the audio/qnorm stores and lower to s_load (SGPR operands).
struct IirDims {
    int nb, n_tiles, m_pad, b_pad;
    long long audio_stride;
};

template <int R, int FORM, bool QN>
__global__ __launch_bounds__(256) void iir_bank_kernel(
    const float *__restrict__ p_ca, const float *__restrict__ p_cb, float *__restrict__ p_sq,
    float *__restrict__ p_sd, const BufDesc *__restrict__ p_desc, const float *__restrict__ p_grows,
    const float *__restrict__ p_tprof, const double *__restrict__ p_xfer_rows,
    const int *__restrict__ p_xfer_init, float *__restrict__ p_audio, float *__restrict__ p_qnorm,
    const IirDims p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int obj = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = blockDim.x >> 6;
    const int rowlen = blockDim.x;
    float *tile = lds + wave * (TILE * LDS_ROW);
    float *xw = lds + W * (TILE * LDS_ROW);          // [2][W-1][64] cross-wave partials
    float *col = tile + lane;
    const size_t mbase = (size_t)obj * p.m_pad + tid;

    float ca[R], cb[R], q[R], d[R], g[R], t[R], qn[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ca[r] = p_ca[mbase + r * rowlen];
        cb[r] = p_cb[mbase + r * rowlen];
        q[r] = p_sq[mbase + r * rowlen];
        d[r] = p_sd[mbase + r * rowlen];
        g[r] = 0.f;
        qn[r] = 0.f;
    }
    {
        const int row0 = p_xfer_init[obj];
#pragma unroll
        for (int r = 0; r < R; ++r)
            t[r] = row0 >= 0 ? (float)p_xfer_rows[(size_t)row0 * p.m_pad + tid + r * rowlen] : 1e7f;
    }

    const BufDesc *__restrict__ dsc = p_desc + (size_t)obj * p.nb;
    float *__restrict__ aout = p_audio + (size_t)obj * p.audio_stride;
    const int B = p.n_tiles * TILE;
    int par = 0;

    for (int b = 0; b < p.nb; ++b) {
        const int frow = __builtin_amdgcn_readfirstlane(dsc[b].frow);
        const uint32_t mask = __builtin_amdgcn_readfirstlane(dsc[b].tile_mask);
        const int trow = __builtin_amdgcn_readfirstlane(dsc[b].trow);
        const uint32_t flags = __builtin_amdgcn_readfirstlane(dsc[b].flags);

        if (flags & DESC_SKIP) {
            // the reference's step() returned before stepping: no samples, state untouched
            for (int i = tid; i < B; i += blockDim.x) aout[(size_t)b * B + i] = 0.f;
            if (QN) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    p_qnorm[((size_t)obj * p.nb + b) * p.m_pad + tid + r * rowlen] = 0.f;
            }
            continue;
        }
        if (trow != XFER_KEEP) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                t[r] = trow >= 0 ? (float)p_xfer_rows[(size_t)trow * p.m_pad + tid + r * rowlen] : 1e7f;
        }
        if (frow >= 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) g[r] = p_grows[(size_t)frow * p.m_pad + tid + r * rowlen];
        }
        const float *__restrict__ tprow = p_tprof + (size_t)(frow >= 0 ? frow : 0) * p.b_pad;
        if (QN) {
#pragma unroll
            for (int r = 0; r < R; ++r) qn[r] = 0.f;
        }

        for (int tl = 0; tl < p.n_tiles; ++tl) {
            if (frow >= 0 && ((mask >> tl) & 1u))
                step_tile<R, FORM, QN, true>(q, d, ca, cb, g, t, qn, tprow + tl * TILE, col);
            else
                step_tile<R, FORM, QN, false>(q, d, ca, cb, g, t, qn, nullptr, col);

            // wave-local hand-off: LDS ops of one wave execute in order; the fence
            // only stops the compiler from moving the row reads above the writes.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            float s = 0.f;
            if (lane < TILE) {
                const float4 *row = reinterpret_cast<const float4 *>(tile + lane * LDS_ROW);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float4 v = row[j];
                    s += v.x;
                    s += v.y;
                    s += v.z;
                    s += v.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            if (W > 1) {
                float *x = xw + par * ((W - 1) * 64);
                if (wave > 0 && lane < TILE) x[(wave - 1) * 64 + lane] = s;
                __syncthreads();
                if (wave == 0 && lane < TILE) {
                    for (int w = 1; w < W; ++w) s += x[(w - 1) * 64 + lane];
                }
                par ^= 1;
            }
            if (wave == 0 && lane < TILE) aout[(size_t)b * B + tl * TILE + lane] = s;
        }

        if (QN) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                p_qnorm[((size_t)obj * p.nb + b) * p.m_pad + tid + r * rowlen] = sqrtf(qn[r]);
        }
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        p_sq[mbase + r * rowlen] = q[r];
        p_sd[mbase + r * rowlen] = d[r];
    }
}

template <int R, int FORM, bool QN>
static int launch_one(const IirParams &p, int n_obj, int W, hipStream_t stream) {
    const size_t lds = iir_lds_bytes(W);
    auto kern = iir_bank_kernel<R, FORM, QN>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const IirDims dims = {p.nb, p.n_tiles, p.m_pad, p.b_pad, p.audio_stride};
    hipLaunchKernelGGL(kern, dim3(n_obj), dim3(64 * W), lds, stream, p.ca, p.cb, p.sq, p.sd, p.desc,
                       p.grows, p.tprof, p.xfer_rows, p.xfer_init, p.audio, p.qnorm, dims);
    return (int)hipGetLastError();
}

template <int R>
static int launch_r(const IirParams &p, int n_obj, int W, int form, bool qn, hipStream_t s) {
    if (form == 0) return qn ? launch_one<R, 0, true>(p, n_obj, W, s) : launch_one<R, 0, false>(p, n_obj, W, s);
    return qn ? launch_one<R, 1, true>(p, n_obj, W, s) : launch_one<R, 1, false>(p, n_obj, W, s);
}

int launch_iir_bank(const IirParams &p, int n_obj, int R, int W, int form, bool qn, hipStream_t s) {
    if (n_obj <= 0) return 0;
    switch (R) {
    case 1: return launch_r<1>(p, n_obj, W, form, qn, s);
    case 2: return launch_r<2>(p, n_obj, W, form, qn, s);
    case 4: return launch_r<4>(p, n_obj, W, form, qn, s);
    case 8: return launch_r<8>(p, n_obj, W, form, qn, s);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace PBSO_IIR_NS
}  // namespace pbso
