// K1b: damped-IIR oscillator bank in BLOCK STATE-SPACE form (gfx950, wave64, f32 MFMA).
//
// Replaces the same reference code as K1 (kernels_iir.hip): the hot loop of ModalSolver::step
// (modal_solver.h:262-272) around ModalIntegrator::Step (modal_integrator.h:103-113).
//
// Idea.  A mode's recurrence q_k = c1 q_{k-1} + c2 q_{k-2} (+ force) is linear and, between
// forces, autonomous.  With the state x = (q, d = q - q_prev) and A the one-sample matrix in that
// basis, the J = 16 samples after a state x are e1' A^j x, j = 1..16 -- two products per mode and
// sample with coefficients (a_j, b_j) that depend on the mode only -- and the state 16 samples
// later is P x, P = A^16.  The reference's forces arrive at the first sample of an audio buffer
// (modal_solver.h:184) and 513 = 1 + 2 * 16 * 16, so a buffer is
//     sample 0      literal per-sample step (velocity form, exactly K1's arithmetic) incl. the impulse,
//     2 groups of 16 blocks of 16 samples:
//        VALU   the lane that owns a mode steps x <- P x sixteen times (4 FMA each) and parks the
//               sixteen block-start states in LDS,
//        MFMA   Y[16 samples][16 blocks] += W[16][4] . X[4][16]  per pair of modes
//               (v_mfma_f32_16x16x4_f32: exact f32, a k-ordered fmaf chain; W = (a_j, b_j) of two
//               modes, constant per object, resident in VGPRs for the whole launch).
// The sum over modes -- the reference's q.dot(transfer), modal_solver.h:267-269 -- IS the
// contraction index: with the scaled state (Q = transfer x q, see K1) no weights are left.
// Per mode-sample this costs 4 flop on the matrix pipe and 0.25 VALU instead of K1's ~5 VALU
// (the reference: 10 flop), and the rounding error is smaller than K1's (33 state updates per
// buffer instead of 513; measured 5e-6 of peak against 5e-5, scripts/debug/block_numerics.py).
//
// Buffers with a dense force time profile (Gaussian, AR: forces.h:92-128) and waves whose
// transfer weights are outside the scaled-state range step every sample literally (velocity form)
// with a simple LDS row-sum reduction.
//
// qnorm (getQBufferNorm, modal_solver.h:270-273): closed form x0' G x0 of the state after sample 0
// in block buffers (the per-sample state never exists here), per-sample accumulation in literal ones.
//
// Layout.  Team / column mapping as K1: workgroup = W waves, lane owns R modes, slice r at column
// col0 + r * 64 W + tid.  An MFMA contracts four (mode, component) values; lane l holds
//   A[j = l & 15][k = l >> 4]   = W-table entry (a_{j+1} or b_{j+1} of one mode)
//   B[k = l >> 4][n = l & 15]   = one state component of block n, read back from the LDS staging area
//   D[i = 4 (l >> 4) + v][n]    = output sample 16 n + i of the group, v = 0..3
// Which four: in the W table as the host lays it out (and in the split-bf16 projection, kernels_pipe.hip, the listener mix) an MFMA
// takes a PAIR of adjacent columns, k = 2 * (mode of the pair) + component.  The f32 pipeline of this file (round 6) takes ONE
// component of four modes a quarter of the wave apart -- MFMA 4 i + jj of a slice: component jj & 1 of the modes 16 k + 2 i + (jj >> 1)
// -- so that a lane's 32 B operands of a slice are 128 contiguous bytes of the staging image (ST_PAIR below) and come in with eight
// ds_read_b128; the W table is read through that permutation once per launch.
#include <type_traits>

#include "kernels.h"
#include "wave_ops.h"

namespace pbso {
namespace iir_block {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BJ = BLOCK_J, BN = BLOCK_N;
constexpr int GROUP = BJ * BN;                     // 256 samples
// staging: [16 blocks][64 lanes][Q, D] (one ds_write_b64 per coarse step); the row stride of 130 floats
// (= 2 mod 32) puts the 32 lanes of every ds_read_b32 lane group (2 components x 16 blocks) on 32 distinct banks
constexpr int ST_ROW = 130;
// Round 6: the f32 projection's staging image is laid out for ds_read_b128.  The MFMA's k index is the QUARTER of the wave's 64
// modes (k = mode >> 4), so the lane (k, block n) of a B operand needs, for its 32 MFMAs, the 16 (Q, D) pairs of quarter k at
// block n: 32 contiguous floats = 8 reads of 16 bytes, where the old image (k = 2 (mode & 1) + component) took 32 ds_read_b32
// (scripts/microbench/gen_burst_shapes.py, profiles/r06_burst_shapes.txt: beside a partner wave a slice costs 1344 cycles with
// one ds_read_b32 behind every MFMA, 1284 with one ds_read_b128 behind every fourth, 1281 with no reads at all).  Blocks n and
// n + 8 share a row of ST_PAIR floats -- [quarter k: 64 floats][half n >> 3: 32 floats] -- and ST_PAIR = 4 mod 64 puts the 16
// lanes of every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) on 16 distinct 4-bank slots; the parking lanes of a quarter
// write 32 contiguous dwords (conflict-free as before).  Same size as the old image: 8 x 260 = 16 x 130 floats.
constexpr int ST_PAIR = 260;
__host__ __device__ constexpr int park_off(int n) { return (n & 7) * ST_PAIR + (n >> 3) * 32; }      // floats, block n's row
constexpr int LIT_ROW = 68;                        // literal path: [16 samples][64 lanes] tile, 16-B aligned rows
constexpr int ST_FLOATS = BLOCK_STAGE_FLOATS;
constexpr int RING = BLOCK_RING_FLOATS;            // per wave and parity: samples 1..512 at [0..511], sample 0 at [512]
// split-bf16 projection (PROJ == 1): staging planes "hi" and "lo", [16 blocks][64 lanes] dwords = (Q, D) as two
// bf16; the row stride of 72 dwords makes the ds_read_b128 of (4 modes of block n) conflict-free in all four
// 16-lane groups
constexpr int FTM_U_ROW = 36;                      // FTM: [64 modes][16 Q increments | 16 D increments] + 4 floats of padding
constexpr int H_ROW = 72;
constexpr int H_PLANE = BN * H_ROW;
static_assert(BN * ST_ROW <= ST_FLOATS && 8 * ST_PAIR <= BN * ST_ROW && BJ * LIT_ROW <= ST_FLOATS && 2 * H_PLANE <= ST_FLOATS, "staging area");

struct BlkDims {
    int nb, m_pad, b_pad, frames, n_groups;
    long long audio_stride;
    long long plane;             // elements between the planes of p_gq / p_pc
    int qn_nb, qn_b0;
    int rotate;                  // != 0: the teams of a CU take turns at the highest wave priority, one buffer each
    int forced_block;            // != 0 (f32 projection only): buffers with a dense force profile run in block form too
    // time-chunked launch (kernels_scan.hip): grid (team, chunk); workgroup (t, c) runs buffers c * cb .. of its team from the
    // state the scan left at p_xs[obj][c] (unscaled) under the transfer row p_xtrow[obj][c]; it does NOT write the state back
    // (the scan has).  cb == 0: one workgroup per team walks all nb buffers from the state arrays.
    int cb, n_chunks, census_stride;
    unsigned long long *start_flag;     // see IirParams::start_flag
    unsigned long long start_seq;
};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


// sum over the 64 lanes without LDS traffic: four DPP adds inside every row of 16, then the four row sums
__device__ __forceinline__ float wave_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// The samples leave the kernel for good (181 MB per launch at 1024 x 512 x 86): nontemporal stores keep them from pushing the
// operand tables and coefficient rows out of the L2 (kernel 0.944 -> 0.939 ms; write-through `sc0 sc1` stores: 0.954;
// scripts/debug/r04_nt.sh).  -DPBSO_AUDIO_STORE=0: plain stores, 2: write-through.
#if defined(PBSO_AUDIO_STORE) && PBSO_AUDIO_STORE == 0
#define AUDIO_STORE(p, v) (*(p) = (v))
#elif defined(PBSO_AUDIO_STORE) && PBSO_AUDIO_STORE == 2
#define AUDIO_STORE(p, v) asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory")
#else
#define AUDIO_STORE(p, v) __builtin_nontemporal_store((v), (p))
#endif

template <int K0, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        f(std::integral_constant<int, K0>{});
        static_for<K0 + 1, N - 1>(f);
    }
}

// DUMP: objects with a dump row (multi-listener mix, pbso_listeners_enable) also write every block-start state
// (scaled, as the registers hold it) and the scale of the buffer to memory; other builds carry no trace of it.
// FORCED (f32 projection only): buffers with a dense force profile run in block form too ("forced block path" below); a
// launch without such buffers uses the build without it (its registers are the kernel's peak at R = 4).
// CHUNKED: the launch is cut along the time axis (kernels_scan.hip): grid (team, chunk), start states from the scan.  A build of
// its own -- the walk in buffer order keeps the code it had, and a profile tells the two apart by the kernel's name.
template <int R, int QNM, int PROJ, bool DUMP, int MAXT, bool FORCED, bool CHUNKED = false>
__global__ __launch_bounds__(MAXT) void iir_block_kernel(
    const float *__restrict__ p_ca, const float *__restrict__ p_cb, float *__restrict__ p_sq,
    float *__restrict__ p_sd, float *__restrict__ p_ss, const BufDesc *__restrict__ p_desc,
    const float *__restrict__ p_grows, const float *__restrict__ p_g32, const long long *__restrict__ p_g32_off,
    const float *__restrict__ p_tprof, const double *__restrict__ p_xfer_rows,
    const int *__restrict__ p_xfer_init, float *__restrict__ p_audio, float *__restrict__ p_qnorm,
    const float *__restrict__ p_gq, const float *__restrict__ p_pc, const float *__restrict__ p_wtab,
    const TeamDesc *__restrict__ p_teams, float *__restrict__ p_audio_parts, unsigned long long *__restrict__ p_census,
    float *__restrict__ p_xdump, float *__restrict__ p_xscale, const int *__restrict__ p_dump_row, unsigned *__restrict__ p_board,
    const float *__restrict__ p_ftab, const float *__restrict__ p_xs, const int *__restrict__ p_xtrow, const BlkDims p) {
    constexpr bool QN = QNM != 0;
    constexpr int NG = 2;                              // groups per buffer (513 = 1 + 2 * 256; checked at launch)
    if (p.start_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(p.start_flag, p.start_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    constexpr bool chunked = CHUNKED;
    const int chunk = chunked ? (int)blockIdx.y : 0;
    const int b_begin = chunk * p.cb;
    const int b_end = chunked ? (b_begin + p.cb < p.nb ? b_begin + p.cb : p.nb) : p.nb;
    constexpr int U = NG * R;                          // slices per buffer: (group, r)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const TeamDesc team = p_teams[blockIdx.x];
    const int obj = team.obj;
    unsigned long long census_t0 = 0, census_c0 = 0;
    if (p_census) {
        census_t0 = __builtin_amdgcn_s_memrealtime();
        census_c0 = __builtin_amdgcn_s_memtime();
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = blockDim.x >> 6;
    const unsigned rowlen = blockDim.x;
    const unsigned utid = threadIdx.x;
    // uniform (SGPR) row bases + an UNSIGNED 32-bit lane offset: every access is saddr + voffset, no 64-bit
    // address VGPRs per array and slice (they cost 16 R registers when the index is a signed int)
    const size_t ubase = (size_t)obj * p.m_pad + team.col0;
    const float *__restrict__ b_ca = p_ca + ubase;
    const float *__restrict__ b_cb = p_cb + ubase;
    const float *__restrict__ b_gq = p_gq + ubase;
    float *__restrict__ b_qn = p_qnorm + ((size_t)obj * p.qn_nb + p.qn_b0) * p.m_pad + team.col0;
    // HALF (one mode per lane): the ring holds ONE group of 256 samples per parity and the team combines at the end of every
    // group -- two barriers per buffer instead of one, 2 KB of LDS per wave less: a CU then holds twelve waves of such teams
    // (three per SIMD) instead of eight, and these builds are bound by what a wave does BETWEEN its matrix bursts (head, taps,
    // combine: 8 x 4096 sustained scraping, time-chunked: per-wave pace unchanged at 10 waves per CU, profiles/r05_*census*)
    constexpr bool HALF = R == 1;
    constexpr int RINGF = HALF ? BLOCK_HALF_RING_FLOATS : RING;
    constexpr int WAVE_FLOATS = ST_FLOATS + 2 * RINGF;
    float *stage = lds + wave * WAVE_FLOATS;
    float *ring = stage + ST_FLOATS;                   // [2][RINGF]

    // per-mode registers for the whole launch: scaled state, coarse step P - I / P, transfer weight
    // state x = (Q, D) and the coarse-step matrix as register PAIRS: c1 = (P11 - 1, P21), c2 = (P12, P22)
    f2 x2[R], c1[R], c2[R];
    float t[R];
    bool dead[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned k = r * rowlen + utid;
        if (chunked) {
            const float *__restrict__ xsrc = p_xs + 2 * (((size_t)obj * p.n_chunks + chunk) * p.m_pad + team.col0);
            x2[r] = reinterpret_cast<const f2 *>(xsrc)[k];
        } else {
            x2[r].x = (p_sq + ubase)[k];
            x2[r].y = (p_sd + ubase)[k];
        }
        c1[r].x = (p_pc + ubase)[k];
        c2[r].x = (p_pc + p.plane + ubase)[k];
        c1[r].y = (p_pc + 2 * p.plane + ubase)[k];
        c2[r].y = (p_pc + 3 * p.plane + ubase)[k];
        dead[r] = b_ca[k] == 0.f && b_cb[k] == 0.f;
    }
    // W table: one VGPR per pair of columns, [pair][64 lanes]; this wave's slice r covers the pairs
    // (team.col0 + r * rowlen + 64 * wave) / 2 + s, s = 0..31
    // (PROJ == 1: the same 32 dwords are eight MFMA operands of four registers each -- hi and lo parts of four groups
    //  of 16 modes -- and are kept as quads from the start, so that the allocator never has to re-pack them)
    f4 wreg4[PROJ == 0 ? R : 1][8];                   // (quads, as the B operands: the allocator packs the two big arrays alike)
    u4 wq[PROJ == 1 ? R : 1][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float *__restrict__ wsrc = p_wtab + ((ubase + r * rowlen + 64 * wave) / 2) * 64;
        if constexpr (PROJ == 0) {
            // the table is [pair of columns][lane 16 (2 (mode & 1) + component) + j - 1]; MFMA 4 i + jj of a slice contracts
            // component jj & 1 of the modes 16 k + 2 i + (jj >> 1), k = 0 .. 3 (the staging image above): its A operand at lane
            // (k, j - 1) is the entry of pair 8 k + i at lane 16 jj + j - 1.  Read once per launch.
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
#ifdef PBSO_ABL_W_COALESCED        // (ablation, wrong results: what the permuted read of the table costs a time-chunked launch's preamble)
                for (int jj = 0; jj < 4; ++jj) wreg4[r][i][jj] = wsrc[(4 * i + jj) * 64 + lane];
#else
                for (int jj = 0; jj < 4; ++jj) wreg4[r][i][jj] = wsrc[(8 * (lane >> 4) + i) * 64 + 16 * jj + (lane & 15)];
#endif
        } else {
            const unsigned *__restrict__ wu = reinterpret_cast<const unsigned *>(wsrc);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                wq[r][i] = u4{wu[(4 * i + 0) * 64 + lane], wu[(4 * i + 1) * 64 + lane], wu[(4 * i + 2) * 64 + lane], wu[(4 * i + 3) * 64 + lane]};
        }
    }

    // Forced block path without qnorm rows (FT): the state increment of a block under a dense force profile is
    // F . T_n with F = [A^15 u .. A u, u] (2 x 16 per mode, u = (1, 1)'): 32 constants per mode, resident for the launch
    constexpr bool FT = FORCED && PROJ == 0 && !QN && R <= 2;
    // FTM (one mode per lane): the 16 increments of a group's blocks are a [16 blocks x 16 taps] . [16 taps x 16 modes] product
    // per tile of 16 modes and state component -- 32 MFMAs per group whose A operand is the profile exactly as the FIR's B
    // operand holds it; the tiles come back to "lane = mode" through [64 modes][U_ROW] floats of LDS (the staging area itself,
    // between two slices; per block the vector ALU is left with the coarse step and two FMAs: 8.5 K -> 4 K cycles per buffer).
    constexpr bool FTM = FT && R == 1;
    constexpr int U_ROW = FTM_U_ROW;
    float fB[FTM ? 4 : 1][2][4];
    if constexpr (FTM) {
#pragma unroll
        for (int tl = 0; tl < 4; ++tl)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    fB[tl][c][ks] = (p_ftab + (size_t)(2 * (4 * ks + (lane >> 4)) + c) * p.plane + ubase)[64 * wave + 16 * tl + (lane & 15)];
    }
    float fq[FT && !FTM ? R : 1][BJ], fd[FT && !FTM ? R : 1][BJ];
    if constexpr (FT && !FTM) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int i = 0; i < BJ; ++i) {
                fq[r][i] = (p_ftab + (size_t)(2 * i) * p.plane + ubase)[r * rowlen + utid];
                fd[r][i] = (p_ftab + (size_t)(2 * i + 1) * p.plane + ubase)[r * rowlen + utid];
            }
        }
    }

    // Forced block path, one mode per lane: the FIR taps h_d = sum over modes of g phi_d, phi_d = e1' A^d u, as sixteen wave sums
    // in one butterfly (wave_ops.h) instead of projecting the virtual state g u on the matrix pipe (32 operand reads + 32 MFMAs
    // for one useful column: 2 K cycles per buffer against 0.5 K).  phi: sixteen constants per mode, stepped here once per
    // launch in fp64 from the per-sample coefficients.
    constexpr bool TAPS_VALU = FORCED && PROJ == 0 && R == 1;
    float phi[TAPS_VALU ? 16 : 1];
    if constexpr (TAPS_VALU) {
        const double ea = (double)b_ca[utid], eb = (double)b_cb[utid];
        double vq = 1.0, vd = 1.0;
        phi[0] = 1.f;
#pragma unroll
        for (int d = 1; d < 16; ++d) {
            vd = ea * vd + eb * vq;
            vq = vq + vd;
            phi[d] = dead[0] ? 0.f : (float)vq;
        }
    }

    // ---- scaled state (as K1): registers hold Q = t q, D = t d while every weight of the wave is usable
    bool scaled = false;
    auto usable = [](float x) { return x >= 0x1p-20f && x <= 0x1p40f; };
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
                    x2[r].x = x2[r].x * f;
                    x2[r].y = x2[r].y * f;
                }
            }
        } else if (!__all(unit)) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                x2[r].x = x2[r].x / from[r];
                x2[r].y = x2[r].y / from[r];
            }
        }
        scaled = ok;
    };
    {
        const int row0 = chunked ? p_xtrow[(size_t)obj * p.n_chunks + chunk] : p_xfer_init[obj];
        float s0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            s0[r] = chunked ? 1.f : (p_ss + ubase)[r * rowlen + utid];
            const float tr = row0 >= 0 ? (float)(p_xfer_rows + (size_t)row0 * p.m_pad + team.col0)[r * rowlen + utid] : 1e7f;
            t[r] = dead[r] ? 1.f : tr;
        }
        rescale(s0, t);
    }

    const BufDesc *__restrict__ dsc = p_desc + (size_t)obj * p.nb;
    float *__restrict__ aout = team.part_row >= 0 ? p_audio_parts + (size_t)team.part_row * p.audio_stride
                                                  : p_audio + (size_t)obj * p.audio_stride;
    const int B = p.frames;

    // B-operand read base (f32 projection): lane (k = lane >> 4, block n = lane & 15) reads quarter k of block n's row, 16 bytes
    // at a time; the lane that owns mode m = lane parks (Q, D) at pair m & 15 of quarter m >> 4
    const f4 *bsrc = reinterpret_cast<const f4 *>(stage + park_off(lane & 15) + (lane >> 4) * 64);
    f2 *wdst = reinterpret_cast<f2 *>(stage + (lane >> 4) * 64 + 2 * (lane & 15));
    auto park = [&](int n, f2 v) { wdst[park_off(n) / 2] = v; };
    // literal path: row sums of the [16][64] tile: lane -> row (lane & 15), quarter (lane >> 4)
    const f4 *lsrc = reinterpret_cast<const f4 *>(stage + (lane & 15) * LIT_ROW + (lane >> 4) * 16);

    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };


    // Inputs of a buffer that come from memory -- per-sample coefficients (sample 0 and literal buffers),
    // qnorm matrices, the force gain row, the new transfer row -- are fetched while the previous buffer's
    // last slice runs on the matrix pipe, so that a buffer never starts with a round trip to L2 / HBM.
    float nca[R], ncb[R], ng11[R], ng12[R], ng22[R], ngr[R];
    const float *__restrict__ g32_obj = p_g32 + (size_t)p_g32_off[obj] * p.m_pad + team.col0;
    // landing area of a DESC_DIRECT hit's three table rows: [3][R][64] floats per wave, behind the waves' staging areas;
    // filled by LDS-DMA loads (no registers between the prefetch and the buffer's head)
    float *gland = lds + blockDim.x / 64 * WAVE_FLOATS + wave * (3 * R * 64);
    float ntr[R];
    // (lane offsets used inside the buffer loop are laundered through an empty asm: otherwise LICM hoists one
    //  64-bit address per array and slice out of the loop and keeps 16 R VGPRs alive for the whole kernel)
    auto lane_off = [&]() {
        unsigned v = utid;
        asm volatile("" : "+v"(v));
        return v;
    };
    // DUMP: xdump[row][qn_nb][32 blocks][m_pad] pairs (Q, D), xscale[row][qn_nb][m_pad] (0 marks a buffer that was
    // stepped per sample: no block states)
    const int dump_row = DUMP ? p_dump_row[obj] : -1;
    int dump_b = 0;                                    // buffer (within the step) the pipeline is working on
    auto dump_state = [&](int r, int blk) {
        if (DUMP && dump_row >= 0) {
            f2 *dst = reinterpret_cast<f2 *>(p_xdump) + (((size_t)dump_row * p.qn_nb + dump_b) * 32 + blk) * p.m_pad + team.col0;
            dst[r * rowlen + lane_off()] = x2[r];
        }
    };
    // code: 1 = block buffer (the scale t), 0 = stepped per sample (no block states: the mix refuses the step),
    // -1 = the reference's step() returned early (DESC_SKIP: no samples; the mix emits silence)
    auto dump_scale = [&](int code) {
        if (DUMP && dump_row >= 0) {
            float *dst = p_xscale + ((size_t)dump_row * p.qn_nb + dump_b) * p.m_pad + team.col0;
            const unsigned utid = lane_off();
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r * rowlen + utid] = code > 0 ? t[r] : (float)code;
        }
    };
    // (the qnorm matrices are a fetch of their own: in the f32 pipeline they are requested BEHIND the last matrix burst, when its
    //  operand registers are free -- with them in flight under the burst the headline build spilled -- and land under the
    //  barrier and the combine)
    auto prefetch_gq = [&]() {
        if (QN) {
            const unsigned utid = lane_off();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned k = r * rowlen + utid;
                ng11[r] = b_gq[k];
                ng12[r] = (b_gq + p.plane)[k];
                ng22[r] = (b_gq + 2 * p.plane)[k];
            }
        }
    };
    auto prefetch = [&](const BufDesc &nd, bool with_gq = true) {
        const unsigned utid = lane_off();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned k = r * rowlen + utid;
            nca[r] = b_ca[k];
            ncb[r] = b_cb[k];
        }
        if (with_gq) prefetch_gq();
        if (nd.frow >= 0 && !(nd.flags & DESC_DIRECT)) {
            const float *__restrict__ gsrc = p_grows + (size_t)nd.frow * p.m_pad + team.col0;
#pragma unroll
            for (int r = 0; r < R; ++r) ngr[r] = gsrc[r * rowlen + utid];
        }
        if (nd.trow >= 0) {
            const double *__restrict__ tsrc = p_xfer_rows + (size_t)nd.trow * p.m_pad + team.col0;
#pragma unroll
            for (int r = 0; r < R; ++r) ntr[r] = (float)tsrc[r * rowlen + utid];     // (fp64 rows: the FFAT kernel is bit-exact with the oracle)
        }
    };

    // The hit of a plain PointForce (DESC_DIRECT): three rows of the object's (float)(c3 * shape) table go straight to
    // the landing area by LDS-DMA -- issued a whole buffer ahead, right after the head has read the previous hit's
    // rows, so that even a cold (HBM) row is there when the next head dots it with the normal; no registers in between.
    // (Inline asm: through the builtin the compiler, which cannot tell the landing area from the staging area, waits
    //  for the DMA before the pipeline's first operand read.)
    const unsigned gland_addr = (unsigned)(size_t)gland;
    auto prefetch_direct = [&](const BufDesc &nd) {
        if (nd.frow >= 0 && (nd.flags & DESC_DIRECT)) {
            const unsigned voff = 4u * lane_off();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float *__restrict__ row = g32_obj + (size_t)(nd.frow + k) * p.m_pad + r * rowlen;      // uniform
                    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2"
                                 :: "s"(gland_addr + (unsigned)((k * R + r) * 256)), "v"(voff), "s"(row) : "memory", "m0");
                }
            }
        }
    };

    // Two coarse steps of slice r: x <- P x twice, the block-start states of blocks n and n + 1 parked by ONE ds_write2_b64 between
    // them.  All of it volatile asm, in this order: left to the compiler the parks are paired (n, n + 8) -- neighbours in the
    // image -- which keeps eight states in registers and issues the eight stores back to back at the end of the burst, where
    // nothing overlaps their way to the LDS (round 6: 140 cycles per slice).  A step is four instructions: the three-operand
    // v_fma_f32 (through the builtin the compiler emits v_mov + v_fmac for q + (P11 - 1) q, a fifth vector instruction per step).
    const unsigned park_base = (unsigned)(size_t)wdst;
    unsigned park_row[4];                              // byte address of row pairs 0-1, 2-3, 4-5, 6-7 (ds_write2_b64 offsets reach 2040 B)
#pragma unroll
    for (int j = 0; j < 4; ++j) park_row[j] = park_base + j * 2 * ST_PAIR * 4;
    auto step_p = [&](int r) {
        float qa, da, qn, dn;
        asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(qa) : "v"(c1[r].x), "v"(x2[r].x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(da) : "v"(c1[r].y), "v"(x2[r].x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(qn) : "v"(c2[r].x), "v"(x2[r].y), "v"(qa));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(dn) : "v"(c2[r].y), "v"(x2[r].y), "v"(da));
        x2[r].x = qn;
        x2[r].y = dn;
    };
    // The vector burst as ONE asm statement (round 6; builds without DUMP): the step on the (Q, D) register PAIR as two v_pk_fma_f32,
    //     t = (P11 - 1, P21) * (Q, Q) + (Q, D);    x' = (P12, P22 - 1) * (D, D) + t
    // (the same map: D' = P21 Q + D + (P22 - 1) D; P22 - 1 is exact in f32 for P22 in [0.5, 2]).  Every operand is a 64-bit pair, so
    // the sixteen steps and their eight ds_write2_b64 fit one statement: no s_nop between statements (the compiler puts one behind
    // every inline-asm definition that the next statement reads: 32 per burst), half the instructions to issue, the stores between the
    // steps.  scripts/microbench/gen_burst_shapes.py, beside a partner wave and the b128 matrix burst: 1264 cycles per slice against
    // 1336 for the four-instruction step with its s_nops (profiles/r06_burst_shapes.txt).
#define PBSO_PK_PAIR(J, O0, O1)                                                                   \
    "v_pk_fma_f32 %[t], %[c1], %[x], %[x] op_sel_hi:[1,0,1]\n\t"                                   \
    "v_pk_fma_f32 %[y], %[c2], %[x], %[t] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"                    \
    "ds_write2_b64 %[a" #J "], %[x], %[y] offset0:" #O0 " offset1:" #O1 "\n\t"                      \
    "v_pk_fma_f32 %[t], %[c1], %[y], %[y] op_sel_hi:[1,0,1]\n\t"                                   \
    "v_pk_fma_f32 %[x], %[c2], %[y], %[t] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
    static_assert(ST_PAIR / 2 == 130, "offsets of the ds_write2_b64 pairs below (8-byte units): blocks n, n + 1 are rows 130 apart, n + 8 is 16 further");
    auto vburst_pk = [&](int r) {
        f2 t, y;
        const f2 c2m = f2{c2[r].x, c2[r].y - 1.f};
        asm volatile(PBSO_PK_PAIR(0, 0, 130) PBSO_PK_PAIR(1, 0, 130) PBSO_PK_PAIR(2, 0, 130) PBSO_PK_PAIR(3, 0, 130)
                     PBSO_PK_PAIR(0, 16, 146) PBSO_PK_PAIR(1, 16, 146) PBSO_PK_PAIR(2, 16, 146) PBSO_PK_PAIR(3, 16, 146)
                     : [x] "+v"(x2[r]), [t] "=&v"(t), [y] "=&v"(y)
                     : [c1] "v"(c1[r]), [c2] "v"(c2m), [a0] "v"(park_row[0]), [a1] "v"(park_row[1]), [a2] "v"(park_row[2]), [a3] "v"(park_row[3])
                     : "memory");
    };
#undef PBSO_PK_PAIR
#ifndef PBSO_VB_FORM
#define PBSO_VB_FORM 1           // 1: vburst_pk; 0: the four-instruction steps (A/B builds)
#endif
#ifndef PBSO_REFILL
#define PBSO_REFILL 0            // 0: one ds_read_b128 behind every fourth MFMA; 1: all eight behind the burst's last MFMA (A/B builds)
#endif
    auto coarse2 = [&](int r, auto nc, int blk) {
        constexpr int n = decltype(nc)::value;         // even
        static_assert((n & 1) == 0 && park_off(n + 1) - park_off(n) == ST_PAIR, "a pair of blocks = two rows of the image");
        dump_state(r, blk);
        const f2 xa = x2[r];
        step_p(r);
        dump_state(r, blk + 1);
        asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4"
                     :: "v"(park_row[(n & 7) / 2]), "v"(xa), "v"(x2[r]), "n"((n >> 3) * 16), "n"((n >> 3) * 16 + ST_PAIR / 2) : "memory");
        step_p(r);
    };

    // diagnostics (PBSO_CENSUS=1): where wave 0's shader cycles go
    unsigned long long cy_head = 0, cy_pipe = 0, cy_bar = 0, cy_comb = 0, cy_mark = 0, cy_taps = 0, cy_step = 0;
#ifdef PBSO_WAVE_TRACE
    // diagnostics build (scripts/debug/r06_wave_trace.py): EVERY wave of a team of <= 2 keeps the shader-clock times at which it
    // left the head, the pipeline, the barrier and the combine of TRACE_NB buffers in the middle of the launch, and its HW_ID:
    // do the two waves of a SIMD (they belong to different teams) take their heads together or one under the other's pipeline?
    int trace_b = -1;
    unsigned long long *trace_row = (p_census && wave < 2)
        ? p_census + ((size_t)team.id + (size_t)chunk * p.census_stride) * CENSUS_WORDS + 12 + wave * (2 + TRACE_NB * TRACE_K) : nullptr;
    if (trace_row && lane == 0) {
        trace_row[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        trace_row[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    auto lap = [&](unsigned long long &acc, int k = -1) {
        if (p_census) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            acc += now - cy_mark;
            cy_mark = now;
#ifdef PBSO_WAVE_TRACE
            if (trace_row && k >= 0 && trace_b >= TRACE_B0 && trace_b < TRACE_B0 + TRACE_NB && lane == 0)
                trace_row[2 + (trace_b - TRACE_B0) * TRACE_K + k] = now;
#endif
        }
    };
    BufDesc next = dsc[b_begin];
    prefetch(next);
    prefetch_direct(next);
    if (p_census) cy_mark = __builtin_amdgcn_s_memtime();
    // The arbiters of a CU (instruction issue, LDS) serve equal-priority waves oldest first: left alone, the team that
    // was dispatched to a CU first finishes at 84 % of the kernel and the other three share the rest at the efficiency
    // of three (scripts/census.py).  Every team takes the order in which it arrived on its CU (an atomic counter per
    // CU; never reset: only the order matters) and the four rotate through the wave priorities, one buffer each.
    // A scheduling hint only: results do not depend on it.
    int prio_rank = 0;
    if (p.rotate) {
        unsigned *word = p_board + 2048 + (((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u) << 8) | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xFFu));
        if (tid == 0) ring[0] = __builtin_bit_cast(float, atomicAdd(word, 1u));
        __syncthreads();
        prio_rank = (int)(__builtin_bit_cast(unsigned, lds[ST_FLOATS]) & 3u);       // wave 0's ring[0]
        __syncthreads();
        // (starting the teams of a CU half a slice period apart -- one wave's vector burst under its SIMD partner's matrix
        //  burst -- changed nothing: 0.948 ms with any stagger of 1, 2 or 4 K cycles on either rank bit, 0.948 without)
#ifdef PBSO_STAGGER_BIT
        // (round 5, scripts/debug/r05_stagger.sh: half a BUFFER apart -- one team's head and combine under its SIMD partner's
        //  pipeline -- 8 or 16 K cycles on rank bit 1 or 2: 9.147 / 9.148 / 9.257 / 9.276 ms per 860 buffers against 9.18 - 9.27
        //  of the product on the same box: nothing either; kept as a build option for the next idea)
        if (prio_rank & PBSO_STAGGER_BIT)
            for (int i = 0; i < PBSO_STAGGER_N; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    }
    for (int b = b_begin; b < b_end; ++b) {
        const BufDesc cur = next;
        next = dsc[b + 1 < p.nb ? b + 1 : b];
#ifndef PBSO_HPRIO
#define PBSO_HPRIO 0             // A/B builds: 1 = a wave outside its pipeline (head, barrier, combine) runs at priority 3, its pipeline at 0 .. 2 in rotation
#endif
        auto set_pipe_prio = [&]() {
            if (p.rotate) {
                switch ((prio_rank + b - b_begin) & 3) {       // (the team's OWN buffer count: the chunks of a CU start together at different b)
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(PBSO_HPRIO ? 1 : 3); break;
                }
            }
        };
        if (PBSO_HPRIO) { if (p.rotate) __builtin_amdgcn_s_setprio(3); }
        else set_pipe_prio();

        const int frow = cur.frow;
        const int prow = (cur.flags & DESC_DIRECT) ? -1 : cur.prow;     // (a direct hit keeps its normal there)
        const float amp = cur.amp;
        const int trow = cur.trow;
        const uint32_t flags = cur.flags;
        float *rg = ring + (b & 1) * RING;             // (HALF: unused)
        // where a group's 256 partial sums and the buffer's sample 0 go
        auto ring_grp = [&](int grp) { return HALF ? ring + grp * RINGF : rg + GROUP * grp; };
        auto ring_s0 = [&]() { return HALF ? ring + GROUP : rg + GROUP * NG; };
        // HALF: the team's waves add their rings of group `grp` and store its samples (sample 0 with group 0); the ring of a group
        // is written again a whole buffer later, behind the OTHER group's barrier, so one barrier per group is enough.  The wave
        // that adds rotates (the others go on).
        auto combine_half = [&](int grp) {
            if constexpr (HALF) {
                lap(cy_pipe);
                __syncthreads();
                lap(cy_bar);
                if (wave == (2 * (b - b_begin) + grp) % W) {
                    const float *r0 = lds + ST_FLOATS + grp * RINGF;
                    float *__restrict__ ao = aout + (size_t)b * B + GROUP * grp;
                    const unsigned j = lane_off() & 63u;
                    f4 acc = *reinterpret_cast<const f4 *>(r0 + 4 * j);
                    for (int w = 1; w < W; ++w) acc += *reinterpret_cast<const f4 *>(r0 + w * WAVE_FLOATS + 4 * j);
                    AUDIO_STORE(&ao[4 * j + 1], acc.x);
                    AUDIO_STORE(&ao[4 * j + 2], acc.y);
                    AUDIO_STORE(&ao[4 * j + 3], acc.z);
                    AUDIO_STORE(&ao[4 * j + 4], acc.w);
                    if (grp == 0 && j == 0) {
                        float a0 = r0[GROUP];
                        for (int w = 1; w < W; ++w) a0 += r0[w * WAVE_FLOATS + GROUP];
                        AUDIO_STORE(&ao[0], a0);
                    }
                }
                lap(cy_comb);
            }
        };
        dump_b = p.qn_b0 + b;
#ifdef PBSO_WAVE_TRACE
        trace_b = b - b_begin;
#endif

        if (flags & DESC_SKIP) {
            dump_scale(-1);
            // the reference's step() returned before stepping: no samples, state untouched
            for (int i = tid; i < B; i += blockDim.x) aout[(size_t)b * B + i] = 0.f;
            if (QN) {
                const unsigned utid = lane_off();
#pragma unroll
                for (int r = 0; r < R; ++r) (b_qn + (size_t)b * p.m_pad)[r * rowlen + utid] = 0.f;
            }
            prefetch(next);
            prefetch_direct(next);
            continue;
        }
        if (trow != XFER_KEEP) {
            float tn[R], from[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float tr = trow >= 0 ? ntr[r] : 1e7f;
                tn[r] = dead[r] ? 1.f : tr;
                from[r] = scaled ? t[r] : 1.f;
            }
            rescale(from, tn);
#pragma unroll
            for (int v = 0; v < R; ++v) t[v] = tn[v];
        }
        float g_[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float gv = ngr[r];
            if (flags & DESC_DIRECT) {
                __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the DMA was issued a buffer ago (normally long done)
                const float *gl = gland + r * 64 + lane;
                gv = __builtin_bit_cast(float, cur.prow) * gl[0];
                gv = fmaf(__builtin_bit_cast(float, cur.tile_mask), gl[R * 64], gv);
                gv = fmaf(__builtin_bit_cast(float, cur.pad[0]), gl[2 * R * 64], gv);
            }
            g_[r] = frow >= 0 ? (scaled ? gv * t[r] : gv) : 0.f;
        }
        prefetch_direct(next);                             // (the landing area is free again: the dot above has read it)
#ifdef PBSO_WAVE_TRACE
        lap(cy_head, 4);
#endif
        const bool impulse = (flags & DESC_IMPULSE) != 0;
        const bool dense = frow >= 0 && !impulse;

        dump_scale(scaled && !dense ? 1 : 0);
        const bool forced_block = FORCED && PROJ == 0 && scaled && dense;
        if (scaled && !dense) {
            // ================= block path =================
            // sample 0, literal: d_0 = eps^2 d - e q + g T_0 ; q_0 = q + d_0   (nca = eps^2, ncb = -e)
            const bool hit0 = frow >= 0 && ((flags & DESC_DIRECT) || (cur.tile_mask & 1u));
            float p0 = 0.f;
#pragma unroll
            for (int v = 0; v < R; ++v) {
                float a = nca[v] * x2[v].y;
                a = fmaf(ncb[v], x2[v].x, a);
                if (hit0) a = fmaf(g_[v], amp, a);
                x2[v].y = a;
                x2[v].x = x2[v].x + a;
                p0 = (v == 0) ? x2[v].x : p0 + x2[v].x;
            }
#ifdef PBSO_WAVE_TRACE
            lap(cy_head, 5);
#endif
            if (QN) {
                // sum_{k=0}^{B-1} q_k^2 = x0' G x0, x0 = state after sample 0 (the rest of the buffer is force-free)
                const unsigned utid = lane_off();
#pragma unroll
                for (int v = 0; v < R; ++v) {
                    float e = ng22[v] * x2[v].y * x2[v].y;
                    e = fmaf(ng12[v] * x2[v].x, x2[v].y, e);
                    e = fmaf(ng11[v] * x2[v].x, x2[v].x, e);
                    // v_sqrt_f32 / v_rcp_f32 (1 ulp each): the closed form itself is good to ~1e-5 only
                    const float nrm = __builtin_amdgcn_sqrtf(fmaxf(e, 0.f));     // (it can round a tiny sum below zero)
                    (b_qn + (size_t)b * p.m_pad)[v * rowlen + utid] = nrm * __builtin_amdgcn_rcpf(t[v]);
                }
            }
#ifdef PBSO_WAVE_TRACE
            lap(cy_head, 6);
#endif
            p0 = wave_sum(p0);
            if (lane == 0) *ring_s0() = p0;

            if constexpr (PROJ == 0) {
            // ---- software pipeline over the U = NG * R slices of the buffer, in BURSTS.  Slice u's 32 MFMAs take their B
            // operands from registers (breg).  A wave alternates a vector burst -- the 16 coarse steps of slice u + 1, which
            // overwrite the staging area (its reads for slice u were issued during the previous matrix burst) -- and a
            // matrix burst: the 32 MFMAs of slice u back to back, each followed by the LDS read that refills the operand
            // register it has just consumed with slice u + 1's.  Why bursts: on gfx950 an instruction issued behind an
            // f32-input MFMA of the SAME wave waits 12 .. 16 cycles for it (MFMA + 1 v_fma_f32: 48 cycles, not 36), while
            // another wave's vector instructions issue beside it (profiles/r03_mfma_valu_mix.txt, "roles": one wave MFMAs
            // only at 38 cycles each while its SIMD partner issues a v_fma_f32 every 4).  With one step (4 VALU + 1 LDS)
            // between every two MFMAs each of the two waves of a SIMD paid that wait 16 times per slice; in bursts the
            // partner's vector burst runs under this wave's matrix burst.
            f4 bq[8];                                  // the slice's 32 B operands: MFMA s takes bq[s >> 2][s & 3]
            lap(cy_head, 0);
            if (PBSO_HPRIO) set_pipe_prio();
#ifdef PBSO_ABL_PIPE_REPEAT
            // ablation (wrong results on purpose, scripts/debug/r06_ablate.sh): the pipeline of a buffer runs N times -- what one more
            // pass costs is the pipeline's own pace, the rest of the kernel's time is what sits around it
            int abl_reps = PBSO_ABL_PIPE_REPEAT;
            asm volatile("" : "+s"(abl_reps));
            for (int abl_rep = 0; abl_rep < abl_reps; ++abl_rep) {
#endif
            wave_sync();                               // the previous buffer's staging reads are issued
            if constexpr (!DUMP && PBSO_VB_FORM == 1) vburst_pk(0);
            else static_for<0, BN / 2>([&](auto hc) { coarse2(0, std::integral_constant<int, 2 * decltype(hc)::value>{}, 2 * decltype(hc)::value); });
            wave_sync();
#pragma unroll
            for (int i = 0; i < 8; ++i) bq[i] = bsrc[i];
#ifdef PBSO_WAVE_TRACE
            lap(cy_pipe, 7);
#endif
            f4 acc0, acc1;
            static_for<0, U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int r = u % R, grp = u / R;
                constexpr int rn = (u + 1) % R;        // the slice whose states are prepared meanwhile
                constexpr bool more = u + 1 < U;
                if constexpr (r == 0) {
                    acc0 = f4{0.f, 0.f, 0.f, 0.f};
                    acc1 = f4{0.f, 0.f, 0.f, 0.f};
                }
                wave_sync();                           // this slice's operand reads are issued: the staging area is free
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (more) {
                    if constexpr (!DUMP && PBSO_VB_FORM == 1) vburst_pk(rn);
                    else static_for<0, BN / 2>([&](auto hc) {
                        constexpr int n = 2 * decltype(hc)::value;
                        coarse2(rn, std::integral_constant<int, n>{}, BN * ((u + 1) / R) + n);
                    });
                    wave_sync();
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef PBSO_MB_PRIO
                __builtin_amdgcn_s_setprio(PBSO_MB_PRIO);          // (A/B: the matrix burst above / below the partner's vector instructions)
#endif
                static_for<0, 32>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg4[r][s >> 2][s & 3], bq[s >> 2][s & 3], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg4[r][s >> 2][s & 3], bq[s >> 2][s & 3], acc0, 0, 0, 0);
                    // refill: one ds_read_b128 behind every fourth MFMA, for the four operands consumed BEFORE these (the MFMA reads its
                    // operands in its first pass).  A/B builds: PBSO_REFILL == 2 refills the four just consumed, 1 issues all eight
                    // behind the burst's last MFMA (kernel 9.03 ms per 860 buffers against 8.83: scripts/debug/r06_ablate.sh)
                    if constexpr (PBSO_REFILL == 0 && more && (s & 3) == 3 && s >= 7) bq[(s >> 2) - 1] = bsrc[(s >> 2) - 1];
                    if constexpr (PBSO_REFILL == 2 && more && (s & 3) == 3) bq[s >> 2] = bsrc[s >> 2];
                    if constexpr (PBSO_REFILL == 3 && more && (s & 7) == 7 && s >= 15) {      // (two reads behind every eighth MFMA)
                        bq[(s >> 2) - 3] = bsrc[(s >> 2) - 3];
                        bq[(s >> 2) - 2] = bsrc[(s >> 2) - 2];
                    }
                    // last slice: nothing to refill -- half way through the burst the first operand registers are free and the
                    // next buffer's inputs land in them
                    if constexpr (!more && s == 15) prefetch(next, false);
#ifndef PBSO_NO_MB_SCHED_BARRIER
                    __builtin_amdgcn_sched_barrier(0);
#endif
                });
#ifdef PBSO_MB_PRIO
                __builtin_amdgcn_s_setprio(PBSO_MB_PRIO_VB);
#endif
                if constexpr (more) {
                    if constexpr (PBSO_REFILL == 0) bq[7] = bsrc[7];
                    else if constexpr (PBSO_REFILL == 3) { bq[6] = bsrc[6]; bq[7] = bsrc[7]; }
                    else if constexpr (PBSO_REFILL == 1) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) bq[i] = bsrc[i];
                    }
                } else {
                    prefetch_gq();
                    if (PBSO_HPRIO && p.rotate) __builtin_amdgcn_s_setprio(3);
                }
                if constexpr (r == R - 1) {
                    const f4 acc = acc0 + acc1;
                    const unsigned l = lane_off() & 63u;       // (recomputed: not worth two registers across the pipeline)
                    *reinterpret_cast<f4 *>(ring_grp(grp) + 16 * (l & 15u) + 4 * (l >> 4)) = acc;
                    combine_half(grp);
                }
            });
#ifdef PBSO_ABL_PIPE_REPEAT
            }
#endif
            } else {
                // ---- split-bf16 projection.  The lane that owns a mode parks every block-start state twice: hi =
                // the top 8 significant bits of x (rounded) and lo = the next 8 of x - hi, both bf16 bit patterns; the
                // projection of 16 modes x 16 blocks is three v_mfma_f32_16x16x32_bf16 (Whi.Xhi + Whi.Xlo + Wlo.Xhi,
                // f32 accumulation), 48 matrix cycles instead of 256.  The recurrence itself (coarse steps) is
                // unchanged f32.
                unsigned *hdst = reinterpret_cast<unsigned *>(stage) + lane;
                const u4 *hsrc = reinterpret_cast<const u4 *>(reinterpret_cast<const unsigned *>(stage) + (lane & 15) * H_ROW + 4 * (lane >> 4));
                // one coarse step of slice r: park block n's start state (hi and lo parts), then x <- P x
                auto coarse16 = [&](int r, int n, int blk) {
                    dump_state(r, blk);
                    const f2 x = x2[r];
                    // hi = the top 8 significant bits of x, lo = the next 8 (of x - hi), both TRUNCATED: a mask, a subtraction
                    // and a v_perm_b32 that packs (Q, D) into one dword, per part.  v_and_b32 / v_sub_f32 issue at full rate
                    // on gfx950; v_cvt_pk_bf16_f32, v_pk_add_f32 and the 16-bit shift of the obvious form (hp = cvt_pk(x);
                    // hi = hp << 16, hp & mask; lo = cvt_pk(x - hi)) at half rate (profiles/r02_valu_issue.txt), and the (q, d)
                    // pair no longer has to sit in an aligned register pair.  Truncating twice loses (7.2 +- 6.4)e-6 of every
                    // value, whatever its distribution (2^-16 . 2/3 . E[1/mantissa]): the mean is folded into the operand
                    // table by the host (TRUNC_SPLIT_GAIN), the spread is below the f32 recurrence's own rounding.
                    const float xq = x.x, xd = x.y;          // (scalars first: a bit cast of a vector ELEMENT expression reads element 0)
                    const unsigned hq = __builtin_bit_cast(unsigned, xq) & 0xFFFF0000u;
                    const unsigned hd = __builtin_bit_cast(unsigned, xd) & 0xFFFF0000u;
                    const unsigned hp = __builtin_amdgcn_perm(hd, hq, 0x07060302u);
                    const float lq = xq - __builtin_bit_cast(float, hq), ld = xd - __builtin_bit_cast(float, hd);
                    const unsigned lp = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, ld), __builtin_bit_cast(unsigned, lq), 0x07060302u);
                    hdst[n * H_ROW] = hp;
                    hdst[H_PLANE + n * H_ROW] = lp;
                    // t = ((P11 - 1) q + P12 d, P21 q + P22 d);  q' = q + t.x (the small term last), d' = t.y
                    // (plain f32 ops: v_pk_fma_f32 / v_pk_mul_f32 issue at half rate, a packed step is no shorter; spelling the
                    //  three-operand v_fma_f32 out in asm as the f32 projection's step does made THIS kernel 4 % slower)
                    const float qa = fmaf(c1[r].x, x.x, x.x);
                    const float da = c1[r].y * x.x;
                    x2[r].x = fmaf(c2[r].x, x.y, qa);
                    x2[r].y = fmaf(c2[r].y, x.y, da);
                };
                // Software pipeline as above: a slice's 8 operand reads (4 groups of 16 modes x hi / lo) sit in
                // registers; its 12 MFMAs ride on the coarse steps 4..15 of the NEXT slice (the matrix pipe co-executes
                // with the vector ALU for bf16), whose steps 0..3 cover the LDS latency of the reads.
                u4 breg[8];
                lap(cy_head);
                wave_sync();
#pragma unroll
                for (int n = 0; n < BN; ++n) coarse16(0, n, n);
                wave_sync();
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    breg[2 * g] = hsrc[4 * g];
                    breg[2 * g + 1] = hsrc[H_PLANE / 4 + 4 * g];
                }
                f4 acc0, acc1;
                static_for<0, U>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    constexpr int r = u % R, grp = u / R;
                    constexpr int rn = (u + 1) % R;
                    constexpr bool more = u + 1 < U;
                    if constexpr (r == 0) {
                        acc0 = f4{0.f, 0.f, 0.f, 0.f};
                        acc1 = f4{0.f, 0.f, 0.f, 0.f};
                    }
                    wave_sync();                       // this slice's operand reads are issued: the staging area is free
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, BN>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        if constexpr (more) coarse16(rn, n, BN * ((u + 1) / R) + n);
                        if constexpr (n >= 4) {
                            constexpr int k = n - 4, g = k / 3, which = k % 3;      // group g: Whi.Xhi, Whi.Xlo, Wlo.Xhi
                            const u4 wh = wq[r][2 * g], wl = wq[r][2 * g + 1];
                            if constexpr (which == 0)
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, breg[2 * g]), acc0, 0, 0, 0);
                            else if constexpr (which == 1)
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, breg[2 * g + 1]), acc1, 0, 0, 0);
                            else
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl), __builtin_bit_cast(bf16x8, breg[2 * g]), acc0, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    if constexpr (more) {
                        wave_sync();
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            breg[2 * g] = hsrc[4 * g];
                            breg[2 * g + 1] = hsrc[H_PLANE / 4 + 4 * g];
                        }
                    } else {
                        prefetch(next);                // behind it: ring write, barrier, combine
                    }
                    if constexpr (r == R - 1) {
                        const f4 acc = acc0 + acc1;
                        const unsigned l = lane_off() & 63u;
                        *reinterpret_cast<f4 *>(ring_grp(grp) + 16 * (l & 15u) + 4 * (l >> 4)) = acc;
                        combine_half(grp);
                    }
                });
            }
        } else if (forced_block) {
            if constexpr (PROJ == 0 && FORCED) {
            // ================= forced block path: a dense force profile (Gaussian, AR: forces.h:92-128) in block form =================
            // The lane that owns a mode still steps it through every sample (velocity form, exactly the literal path's
            // arithmetic: the profile value T_k is wave-uniform and comes from scalar loads) -- but only to carry the state
            // and the qnorm sum.  The per-sample sum over modes, which is what makes the literal path slow (a transpose
            // tile and row sums per 16 samples), is not formed: with X_n the state at the start of block n and
            // u = (1, 1)' the direction a force sample enters the state (d += f, q += d),
            //     x_{16n+j} = A^j X_n + sum_{i=1..j} A^{j-i} u f_{16n+i},        f_k = g T_k,
            // so the output of block n is the usual projection W . X_n (matrix pipe, as in the force-free path) plus a
            // 16-tap FIR of the profile with taps h_d = sum_m g_m e1' A_m^d u -- h_0 = sum g, and h_1..h_16 are the
            // projection of the virtual block-start state g u (one more column of the same W table: 32 MFMAs per
            // slice and buffer).  The FIR itself is a [16 x 16 lower-triangular Toeplitz(h)] . [16 x 16 blocks of T]
            // product: 4 MFMAs per group of 256 samples, accumulated into the projection's own accumulators.
            const float *__restrict__ tprow = p_tprof + (size_t)(prow >= 0 ? prow : 0) * p.b_pad;
            float *taps = stage + BN * ST_ROW;                   // [17] h_0 .. h_16 (behind the staging rows)
            float qn[R];
            lap(cy_head);
            // the profile as the FIR's B operand, B[i][n] = T[1 + 256 grp + 16 n + i], both groups: fetched here, under the
            // taps' 2 K cycles (at the head of its group the first MFMA waited out an L2 round trip)
            // (engines of one or two modes per lane: the builds of four and eight have no registers to spare)
            constexpr bool FIR_EARLY = R <= 2;
            float fir_ball[FIR_EARLY ? NG : 1][4];
            if constexpr (FIR_EARLY) {
#pragma unroll
                for (int grp = 0; grp < NG; ++grp)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) fir_ball[grp][kk] = tprow[1 + GROUP * grp + BJ * (lane & 15) + 4 * kk + (lane >> 4)];
            }
            {
                // sample 0 (literal) and h_0
                const float tk0 = tprow[0];
                float p0 = 0.f, gs = 0.f;
#pragma unroll
                for (int v = 0; v < R; ++v) {
                    float a = nca[v] * x2[v].y;
                    a = fmaf(ncb[v], x2[v].x, a);
                    a = fmaf(g_[v], tk0, a);
                    x2[v].y = a;
                    x2[v].x = x2[v].x + a;
                    p0 = (v == 0) ? x2[v].x : p0 + x2[v].x;
                    gs = (v == 0) ? g_[v] : gs + g_[v];
                    qn[v] = x2[v].x * x2[v].x;
                }
                p0 = wave_sum(p0);
                if (!TAPS_VALU) gs = wave_sum(gs);
                if (lane == 0) *ring_s0() = p0;
                wave_sync();                                    // the previous buffer's staging reads are issued
                if (!TAPS_VALU && lane == 0) taps[0] = gs;
            }
            f4 bq[8];
            if constexpr (TAPS_VALU) {
                float pv[16];
#pragma unroll
                for (int d = 0; d < 16; ++d) pv[d] = g_[0] * phi[d];
                const float hd = wave_sum16(pv, lane, stage + ST_ROW, wave_sync);        // (scratch: staging row 1, free until the first park)
                if (lane < 16) taps[taps_index(lane)] = hd;
                wave_sync();
            } else {
            // ---- taps h_1 .. h_16: project the virtual state (g, g) of every slice (block row 0 of the staging area)
                f4 ah0 = f4{0.f, 0.f, 0.f, 0.f}, ah1 = f4{0.f, 0.f, 0.f, 0.f};
                static_for<0, R>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (r > 0) wave_sync();
                    park(0, f2{g_[r], g_[r]});
                    wave_sync();
#pragma unroll
                    for (int i = 0; i < 8; ++i) bq[i] = bsrc[i];
                    static_for<0, 32>([&](auto sc) {
                        constexpr int s2 = decltype(sc)::value;
                        if constexpr (s2 & 1) ah1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg4[r][s2 >> 2][s2 & 3], bq[s2 >> 2][s2 & 3], ah1, 0, 0, 0);
                        else ah0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg4[r][s2 >> 2][s2 & 3], bq[s2 >> 2][s2 & 3], ah0, 0, 0, 0);
                    });
                });
                const f4 ah = ah0 + ah1;                        // column 0 (lanes 0, 16, 32, 48): rows 4 (l >> 4) + v = tap index - 1
                if ((lane & 15) == 0) {
                    float *td = taps + 1 + 4 * (lane >> 4);
                    td[0] = ah.x; td[1] = ah.y; td[2] = ah.z; td[3] = ah.w;
                }
                wave_sync();
            }
            lap(cy_taps);
            // FIR operand A[j][i] = h_{j-i} (i <= j), k-step kk: lane holds row l & 15, column 4 kk + (l >> 4)
            float fir_a[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int idx = (lane & 15) - 4 * kk - (lane >> 4);
                fir_a[kk] = idx >= 0 ? taps[idx] : 0.f;
            }
            static_for<0, NG>([&](auto gc) {
                constexpr int grp = decltype(gc)::value;
                f4 acc0 = f4{0.f, 0.f, 0.f, 0.f}, acc1 = f4{0.f, 0.f, 0.f, 0.f};
                float fir_b[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    fir_b[kk] = FIR_EARLY ? fir_ball[FIR_EARLY ? grp : 0][kk] : tprow[1 + GROUP * grp + BJ * (lane & 15) + 4 * kk + (lane >> 4)];
                static_for<0, R>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    wave_sync();                                // the staging area is free (operand reads of the previous slice issued)
                    // The block's 16 profile values are wave-uniform.  Block-at-a-time stepping of a one-mode-per-lane engine (an
                    // under-filled chip, where a buffer's latency is what counts): every lane loads them itself (one address for
                    // the whole wave: a broadcast), four blocks in flight in rotating registers -- vector loads return in order,
                    // so a block waits for ITS values only (per-sample stepping ran 11 % slower that way: 13.2 K against 11.9 K
                    // cycles per buffer).  Otherwise: one s_load_dwordx16 per block, issued a block ahead; scalar loads return out of
                    // order, so a use waits for EVERYTHING outstanding: the next block's load is issued right after the first
                    // use of this block's values (the scheduling barriers pin that order).
                    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                    auto for_each_block = [&](auto &&body) {
                        if constexpr (R == 1 && FT) {
                            const float *tl = tprow + 1 + GROUP * grp + (lane_off() >> 31);     // (+ 0, opaque: vector loads)
                            f4u q0[4], q1[4], q2[4], q3[4];
                            auto ld = [&](f4u (&d)[4], int n) {
                                const float *src = tl + BJ * (n < BN ? n : BN - 1);
#pragma unroll
                                for (int i = 0; i < 4; ++i) d[i] = *reinterpret_cast<const f4u *>(src + 4 * i);
                            };
                            auto run = [&](const f4u (&d)[4], int n) {
                                const float tv[BJ] = {d[0].x, d[0].y, d[0].z, d[0].w, d[1].x, d[1].y, d[1].z, d[1].w,
                                                      d[2].x, d[2].y, d[2].z, d[2].w, d[3].x, d[3].y, d[3].z, d[3].w};
                                body(tv, n);
                            };
                            ld(q0, 0); ld(q1, 1); ld(q2, 2);
                            for (int n = 0; n < BN; n += 4) {
                                ld(q3, n + 3);
                                run(q0, n);
                                ld(q0, n + 4);
                                run(q1, n + 1);
                                ld(q1, n + 5);
                                run(q2, n + 2);
                                ld(q2, n + 6);
                                run(q3, n + 3);
                            }
                        } else {
                            float ta[BJ], tb[BJ];
                            auto load_t = [&](float (&dst)[BJ], int n) {
                                const float *__restrict__ tk = tprow + 1 + GROUP * grp + BJ * n;
#pragma unroll
                                for (int k = 0; k < BJ; ++k) dst[k] = tk[k];
                            };
                            load_t(ta, 0);
                            for (int n = 0; n < BN; n += 2) {
                                const float probe = ta[0];
                                asm volatile("" :: "s"(probe));                 // (first use of ta: the wait; then the next block's load)
                                __builtin_amdgcn_sched_barrier(0);
                                load_t(tb, n + 1);
                                __builtin_amdgcn_sched_barrier(0);
                                body(ta, n);
                                const float probe2 = tb[0];
                                asm volatile("" :: "s"(probe2));
                                __builtin_amdgcn_sched_barrier(0);
                                load_t(ta, n + 2 < BN ? n + 2 : n + 1);
                                __builtin_amdgcn_sched_barrier(0);
                                body(tb, n + 1);
                            }
                        }
                    };
                    // Unit-force form: with z = x / g the forcing term is the profile value itself (an operand of the fma: no
                    // product g T per sample); the parked block states and the qnorm sum are scaled back by g.  Floating point
                    // is scale-invariant but for its range: the wave takes this form when every lane's z stays well inside it
                    // (a zero / tiny force gain on some mode -- e.g. the dummy start message of a sustained contact, data = 0
                    // -- takes the general form).  Per sample the dependent chain is two operations:
                    // d' = (eps^2 d + T) - e q, q' = q + d'.
                    lap(cy_pipe);
                    const float gr = g_[r];
                    if constexpr (FTM) {
                        // (the tiles land in the wave's own staging area -- [64 modes][U_ROW] floats are exactly its size: the previous
                        //  group's operand reads are issued, the taps behind the staging rows are in registers, and the lane reads its row
                        //  back before it parks the first state; an area of its own cost 9 KB per wave: teams of eight did not fit the CU)
                        static_assert(64 * U_ROW <= ST_FLOATS, "FTM tiles alias the staging area");
                        float *ua = stage;
                        static_for<0, 4>([&](auto tc) {
                            constexpr int tl = decltype(tc)::value;
                            f4 dq = f4{0.f, 0.f, 0.f, 0.f}, dd = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) {
                                dq = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_b[ks], fB[tl][0][ks], dq, 0, 0, 0);
                                dd = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_b[ks], fB[tl][1][ks], dd, 0, 0, 0);
                            }
                            // D[block 4 (l >> 4) + v][mode 16 tl + (l & 15)]
                            float *dst = ua + (16 * tl + (lane & 15)) * U_ROW + 4 * (lane >> 4);
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
                            park(n, x2[r]);
                            const float qa = fmaf(c1[r].x, x2[r].x, x2[r].x);
                            const float da = c1[r].y * x2[r].x;
                            const float qn_ = fmaf(c2[r].x, x2[r].y, qa);
                            const float dn_ = fmaf(c2[r].y, x2[r].y, da);
                            x2[r].x = fmaf(gr, uq[n / 4][n % 4], qn_);
                            x2[r].y = fmaf(gr, ud[n / 4][n % 4], dn_);
                        });
                    } else if constexpr (FT) {
                        // No qnorm rows asked for: nothing needs the state of every sample.  One block at a time,
                        //     x_{n+1} = P x_n + g (F . T_n)        (32 independent FMAs with the profile values as operands + the
                        // coarse step) -- 36 vector instructions per 16 samples instead of 48 in a dependent chain.
                        for_each_block([&](const float (&tv)[BJ], int n) {
                            park(n, x2[r]);
                            float uq0 = fq[r][0] * tv[0], uq1 = fq[r][1] * tv[1], ud0 = fd[r][0] * tv[0], ud1 = fd[r][1] * tv[1];
#pragma unroll
                            for (int i = 2; i < BJ; i += 2) {
                                uq0 = fmaf(fq[r][i], tv[i], uq0);
                                uq1 = fmaf(fq[r][i + 1], tv[i + 1], uq1);
                                ud0 = fmaf(fd[r][i], tv[i], ud0);
                                ud1 = fmaf(fd[r][i + 1], tv[i + 1], ud1);
                            }
                            const float qa = fmaf(c1[r].x, x2[r].x, x2[r].x);
                            const float da = c1[r].y * x2[r].x;
                            const float qn_ = fmaf(c2[r].x, x2[r].y, qa);
                            const float dn_ = fmaf(c2[r].y, x2[r].y, da);
                            x2[r].x = fmaf(gr, uq0 + uq1, qn_);
                            x2[r].y = fmaf(gr, ud0 + ud1, dn_);
                        });
                    } else {
                        const float gi = __builtin_amdgcn_rcpf(gr);
                        f2 z = f2{x2[r].x * gi, x2[r].y * gi};
                        const bool z_ok = gr != 0.f && fabsf(gi) < 0x1p100f && fabsf(z.x) < 0x1p50f && fabsf(z.y) < 0x1p50f;      // (NaN / inf fail)
                        if (__all(z_ok)) {
                            float qz = 0.f;
                            for_each_block([&](const float (&tv)[BJ], int n) {
                                park(n, f2{gr * z.x, gr * z.y});
#pragma unroll
                                for (int k = 0; k < BJ; ++k) {
                                    const float in = fmaf(nca[r], z.y, tv[k]);
                                    z.y = fmaf(ncb[r], z.x, in);
                                    z.x = z.x + z.y;
                                    if (QN) qz = fmaf(z.x, z.x, qz);
                                }
                            });
                            x2[r] = f2{gr * z.x, gr * z.y};
                            if (QN) qn[r] = fmaf(gr * gr, qz, qn[r]);
                        } else {
                            for_each_block([&](const float (&tv)[BJ], int n) {
                                park(n, x2[r]);
#pragma unroll
                                for (int k = 0; k < BJ; ++k) {
                                    const float gt = g_[r] * tv[k];
                                    const float in = fmaf(nca[r], x2[r].y, gt);
                                    x2[r].y = fmaf(ncb[r], x2[r].x, in);
                                    x2[r].x = x2[r].x + x2[r].y;
                                    if (QN) qn[r] = fmaf(x2[r].x, x2[r].x, qn[r]);
                                }
                            });
                        }
                    }
                    lap(cy_step);
                    wave_sync();
#pragma unroll
                    for (int i = 0; i < 8; ++i) bq[i] = bsrc[i];
                    static_for<0, 32>([&](auto sc) {
                        constexpr int s2 = decltype(sc)::value;
                        if constexpr (s2 & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg4[r][s2 >> 2][s2 & 3], bq[s2 >> 2][s2 & 3], acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg4[r][s2 >> 2][s2 & 3], bq[s2 >> 2][s2 & 3], acc0, 0, 0, 0);
                    });
                });
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (kk & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_a[kk], fir_b[kk], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fir_a[kk], fir_b[kk], acc0, 0, 0, 0);
                }
                const f4 acc = acc0 + acc1;
                const unsigned l = lane_off() & 63u;
                *reinterpret_cast<f4 *>(ring_grp(grp) + 16 * (l & 15u) + 4 * (l >> 4)) = acc;
                combine_half(grp);
            });
            if (QN) {
                const unsigned utid = lane_off();
#pragma unroll
                for (int r = 0; r < R; ++r) (b_qn + (size_t)b * p.m_pad)[r * rowlen + utid] = sqrtf(qn[r]) / t[r];
            }
            prefetch(next);
            }
        } else {
            // ================= literal path: every sample stepped (velocity form), as K1 =================
            const float *__restrict__ tprow = p_tprof + (size_t)(prow >= 0 ? prow : 0) * p.b_pad;
            float qn[R];
#pragma unroll
            for (int r = 0; r < R; ++r) qn[r] = 0.f;
            auto step1 = [&](float tk, bool forced) {
                float pp = 0.f;
#pragma unroll
                for (int v = 0; v < R; ++v) {
                    float a = nca[v] * x2[v].y;
                    a = fmaf(ncb[v], x2[v].x, a);
                    if (forced) a = fmaf(g_[v], tk, a);
                    x2[v].y = a;
                    x2[v].x = x2[v].x + a;
                    if (scaled) pp = (v == 0) ? x2[v].x : pp + x2[v].x;
                    else pp = (v == 0) ? t[v] * x2[v].x : fmaf(t[v], x2[v].x, pp);
                    if (QN) qn[v] = fmaf(x2[v].x, x2[v].x, qn[v]);
                }
                return pp;
            };
            {
                const float tk0 = dense ? tprow[0] : ((frow >= 0 && ((flags & DESC_DIRECT) || (cur.tile_mask & 1u))) ? amp : 0.f);
                float p0 = step1(tk0, frow >= 0);
                p0 = wave_sum(p0);
                if (lane == 0) *ring_s0() = p0;
            }
            for (int c = 0; c < NG * BN; ++c) {
                wave_sync();                           // the previous chunk's row reads are issued
#pragma unroll 4                                       // (fully unrolled this path, not the pipeline, sets the kernel's register peak)
                for (int k = 0; k < BJ; ++k) {
                    const float tk = dense ? tprow[1 + c * BJ + k] : 0.f;
                    const float pp = step1(tk, dense);
                    stage[k * LIT_ROW + lane] = pp;
                }
                wave_sync();
                float rs = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4 v = lsrc[j];
                    rs += v.x;
                    rs += v.y;
                    rs += v.z;
                    rs += v.w;
                }
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                if (lane < BJ) ring_grp(c / BN)[(c % BN) * BJ + lane] = rs;
                if (HALF && (c % BN) == BN - 1) combine_half(c / BN);
            }
            if (QN) {
                const unsigned utid = lane_off();
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float nrm = sqrtf(qn[r]);
                    (b_qn + (size_t)b * p.m_pad)[r * rowlen + utid] = scaled ? nrm / t[r] : nrm;
                }
            }
            prefetch(next);                            // (not hidden here: literal buffers are the slow path anyway)
        }

        // ---- the team's waves add their rings and store the buffer; one barrier per buffer (the ring
        //      alternates by buffer parity, so a wave can be one buffer ahead of the slowest reader)
#ifdef PBSO_ABL_NO_COMBINE
        if constexpr (false) {             // ablation: no barrier, no ring sums, no sample stores
#else
        if constexpr (!HALF) {
#endif
        lap(cy_pipe, 1);
        __syncthreads();
        lap(cy_bar, 2);
        {
            // thread j adds the waves' ring entries 4j .. 4j+3 (samples 4j+1 .. 4j+4); thread 0 also sample 0
            const float *r0 = lds + ST_FLOATS + (b & 1) * RING;
            float *__restrict__ ao = aout + (size_t)b * B;
            // (the thread index laundered: otherwise LICM keeps the ring address of thread j alive across the whole pipeline, and
            //  the headline build, which has no register to spare, spills it)
            for (int j = (int)lane_off(); j < GROUP * NG / 4; j += blockDim.x) {
                f4 acc = *reinterpret_cast<const f4 *>(r0 + 4 * j);
                for (int w = 1; w < W; ++w) acc += *reinterpret_cast<const f4 *>(r0 + w * WAVE_FLOATS + 4 * j);
                AUDIO_STORE(&ao[4 * j + 1], acc.x);
                AUDIO_STORE(&ao[4 * j + 2], acc.y);
                AUDIO_STORE(&ao[4 * j + 3], acc.z);
                AUDIO_STORE(&ao[4 * j + 4], acc.w);
            }
            if (tid == 0) {
                float acc = r0[GROUP * NG];
                for (int w = 1; w < W; ++w) acc += r0[w * WAVE_FLOATS + GROUP * NG];
                AUDIO_STORE(&ao[0], acc);
            }
        }
        lap(cy_comb, 3);
        }
    }

    if (p_census && tid == 0) {
        // where and when this workgroup ran, and the shader clock it held (diagnostics, PBSO_CENSUS=1)
        const size_t census_row = (size_t)team.id + (size_t)chunk * p.census_stride;
        p_census[census_row * CENSUS_WORDS + 0] = census_t0;
        p_census[census_row * CENSUS_WORDS + 1] = __builtin_amdgcn_s_memrealtime();
        p_census[census_row * CENSUS_WORDS + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        p_census[census_row * CENSUS_WORDS + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        p_census[census_row * CENSUS_WORDS + 4] = census_c0;
        p_census[census_row * CENSUS_WORDS + 5] = __builtin_amdgcn_s_memtime();
        p_census[census_row * CENSUS_WORDS + 6] = cy_head;
        p_census[census_row * CENSUS_WORDS + 7] = cy_pipe;
        p_census[census_row * CENSUS_WORDS + 8] = cy_bar;
        p_census[census_row * CENSUS_WORDS + 9] = cy_comb;
        p_census[census_row * CENSUS_WORDS + 10] = cy_taps;       // forced block path: sample 0 + FIR taps
        p_census[census_row * CENSUS_WORDS + 11] = cy_step;       // forced block path: per-sample state stepping
    }
    if (chunked) return;                               // (the scan wrote the launch's end state)
    const unsigned utid_end = lane_off();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned k = r * rowlen + utid_end;
        (p_sq + ubase)[k] = x2[r].x;
        (p_sd + ubase)[k] = x2[r].y;
        (p_ss + ubase)[k] = scaled ? t[r] : 1.f;
    }
}

template <int R, int QNM, int PROJ, bool DUMP, bool FORCED = false, bool CHUNKED = false>
static int launch_one(const IirParams &p, int n_teams, int W, hipStream_t stream) {
    if constexpr (!CHUNKED && !DUMP) {
        if (p.tc_cb > 0) return launch_one<R, QNM, PROJ, DUMP, FORCED, true>(p, n_teams, W, stream);
    }
    if (!CHUNKED && p.tc_cb > 0) return (int)hipErrorInvalidValue;       // (no chunked build of this shape: the engine never asks)
    const size_t lds = block_lds_bytes(W, R) + (CHUNKED ? (size_t)p.lds_pad : 0);
    constexpr int MAXT = 64 * MAX_WAVES_PER_BLOCK_TEAM;
    if (64 * W > MAXT) return (int)hipErrorInvalidValue;
    auto kern = iir_block_kernel<R, QNM, PROJ, DUMP, MAXT, FORCED, CHUNKED>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    const int frames = p.frames;
    const int n_chunks = p.tc_cb > 0 ? (p.nb + p.tc_cb - 1) / p.tc_cb : 1;
    const BlkDims dims = {p.nb, p.m_pad, p.b_pad, frames, (frames - 1) / GROUP, p.audio_stride, p.gq_plane, p.qn_nb, p.qn_b0, p.rotate_prio, p.forced_block,
                          p.tc_cb, n_chunks, p.census_stride, p.start_flag, p.start_seq};
    hipLaunchKernelGGL(kern, dim3(n_teams, n_chunks), dim3(64 * W), lds, stream, p.ca, p.cb, p.sq, p.sd, p.ss, p.desc, p.grows, p.g32, p.g32_off,
                       p.tprof, p.xfer_rows, p.xfer_init, p.audio, p.qnorm, p.gq, p.pc, p.wtab, p.teams, p.audio_parts, p.census, p.xdump, p.xscale, p.dump_row, p.board, p.ftab,
                       p.tc_xs, p.tc_xtrow, dims);
    return (int)hipGetLastError();
}

template <int R>
static int launch_r(const IirParams &p, int n_teams, int W, bool qn, int proj, hipStream_t s) {
    if (p.xdump) {          // some object keeps its block-start states (multi-listener mix): qnorm rows always on in that build
        return proj ? launch_one<R, 2, 1, true>(p, n_teams, W, s) : launch_one<R, 2, 0, true>(p, n_teams, W, s);
    }
    if (proj) return qn ? launch_one<R, 2, 1, false>(p, n_teams, W, s) : launch_one<R, 0, 1, false>(p, n_teams, W, s);
    if (p.forced_block) return qn ? launch_one<R, 2, 0, false, true>(p, n_teams, W, s) : launch_one<R, 0, 0, false, true>(p, n_teams, W, s);
    return qn ? launch_one<R, 2, 0, false>(p, n_teams, W, s) : launch_one<R, 0, 0, false>(p, n_teams, W, s);
}

// The builds of one, two and four modes per lane are TWO translation units (the Makefile compiles this file twice, in parallel:
// -DPBSO_BLOCK_PART=0 the four-modes-per-lane builds + everything else in this file, =1 the builds of one and two); without the
// macro -- tests/test_kernel_asm_guards.py, the A/B variants of scripts/debug/r06_variant.sh with -DPBSO_ONLY_R4 -- one unit.
int launch_block_r1(const IirParams &p, int n_teams, int W, bool qn, int proj, hipStream_t s);
int launch_block_r2(const IirParams &p, int n_teams, int W, bool qn, int proj, hipStream_t s);
#if !defined(PBSO_ONLY_R4) && (!defined(PBSO_BLOCK_PART) || PBSO_BLOCK_PART == 1)
int launch_block_r1(const IirParams &p, int n_teams, int W, bool qn, int proj, hipStream_t s) { return launch_r<1>(p, n_teams, W, qn, proj, s); }
int launch_block_r2(const IirParams &p, int n_teams, int W, bool qn, int proj, hipStream_t s) { return launch_r<2>(p, n_teams, W, qn, proj, s); }
#endif

#if !defined(PBSO_BLOCK_PART) || PBSO_BLOCK_PART == 0
int launch_iir_block(const IirParams &p, int n_teams, int R, int W, int qnm, int proj, hipStream_t s) {
    if (n_teams <= 0) return 0;
    if (W < 1 || W > MAX_WAVES_PER_BLOCK_TEAM) return (int)hipErrorInvalidValue;
    if (p.frames != 1 + 2 * GROUP) return (int)hipErrorInvalidValue;      // the ring holds two groups (513 samples)
    const bool qn = qnm != 0;
    switch (R) {
#ifndef PBSO_ONLY_R4          // (tests/test_kernel_asm_guards.py compiles the R = 4 builds alone: the headline shape's)
    case 1: return launch_block_r1(p, n_teams, W, qn, proj, s);
    case 2: return launch_block_r2(p, n_teams, W, qn, proj, s);
#endif
    case 4: return launch_r<4>(p, n_teams, W, qn, proj, s);
    }
    return (int)hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------
// Multi-listener output mix: the sound of ONE object at L listener positions from the block-start
// states a DUMP launch kept.  For listener l the per-sample dot q . transfer_l (modal_solver.h:267-269, with the
// transfer of computeTransfer(pos, T*), :302-315) is  Y_l[16 samples][16 blocks] = sum over modes of
// (T_l[m] / scale[m]) . W[16][(m, comp)] . X[(m, comp)][16 blocks]:  the B operand (states) is shared by all
// listeners, the A operand is the object's (a_j, b_j) table times the listener's weight -- an
// [L x 16 samples x 2M] . [2M x 16 blocks] contraction on v_mfma_f32_16x16x4_f32 per group of 256 samples.
// One wave per (buffer, group); listeners in tiles of 8 accumulators.
constexpr int MIX_TILE = 8;
__global__ __launch_bounds__(64) void listener_mix_kernel(
    const float *__restrict__ xdump, const float *__restrict__ xscale, const float *__restrict__ wtab32,
    const double *__restrict__ trows, float *__restrict__ out, int nb, int m_pad, int n_modes, int n_listeners,
    long long out_stride) {
    __shared__ __attribute__((aligned(16))) float st[BN * ST_ROW];         // [16 blocks][64 modes][Q, D], stride 130
    __shared__ float tls[MIX_TILE][64];                                    // listener weight / scale of the 64 modes
    const int lane = threadIdx.x;
    const int grp = blockIdx.x, b = blockIdx.y;
    const f2 *xs = reinterpret_cast<const f2 *>(xdump) + ((size_t)b * 32 + grp * BN) * m_pad;
    const float *sc = xscale + (size_t)b * m_pad;
    const float *bsrc = st + (lane & 15) * ST_ROW + 2 * (lane >> 5) + ((lane >> 4) & 1);
    for (int l0 = 0; l0 < n_listeners; l0 += MIX_TILE) {
        f4 acc[MIX_TILE];
        float y0[MIX_TILE];
#pragma unroll
        for (int l = 0; l < MIX_TILE; ++l) {
            acc[l] = f4{0.f, 0.f, 0.f, 0.f};
            y0[l] = 0.f;
        }
        for (int slab = 0; slab < (n_modes + 63) / 64; ++slab) {     // (columns behind the object's waves are never written: m_pad may be wider than its teams)
            const int m = slab * 64 + lane;
            __syncthreads();
            // scale < 0: the buffer was skipped (clearAllForces, modal_solver.h:186-189): silence, and the state rows were
            // never written; scale 0 (a buffer stepped per sample) does not get here: Engine::mix_listeners refuses the step
            const float s = sc[m];
            const bool live = s > 0.f;
#pragma unroll
            for (int n = 0; n < BN; ++n) *reinterpret_cast<f2 *>(st + n * ST_ROW + 2 * lane) = live ? xs[(size_t)n * m_pad + m] : f2{0.f, 0.f};
#pragma unroll
            for (int l = 0; l < MIX_TILE; ++l) {
                const bool on = live && l0 + l < n_listeners && m < n_modes;
                tls[l][lane] = on ? (float)trows[(size_t)(l0 + l) * m_pad + m] / s : 0.f;
            }
            __syncthreads();
            if (grp == 0) {                            // sample 0 of the buffer = the Q component of block 0's start state
                const float q0 = st[2 * lane];
#pragma unroll
                for (int l = 0; l < MIX_TILE; ++l) y0[l] = fmaf(tls[l][lane], q0, y0[l]);
            }
            for (int pr = 0; pr < 32; ++pr) {
                const float bop = bsrc[4 * pr];
                const float wv = wtab32[((size_t)slab * 32 + pr) * 64 + lane];
#pragma unroll
                for (int l = 0; l < MIX_TILE; ++l)
                    acc[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv * tls[l][2 * pr + (lane >> 5)], bop, acc[l], 0, 0, 0);
            }
        }
#pragma unroll
        for (int l = 0; l < MIX_TILE; ++l) {
            if (l0 + l >= n_listeners) break;
            float *o = out + (size_t)(l0 + l) * out_stride + (size_t)b * (1 + 2 * GROUP);
            float *og = o + 1 + GROUP * grp + 16 * (lane & 15) + 4 * (lane >> 4);
            og[0] = acc[l].x;
            og[1] = acc[l].y;
            og[2] = acc[l].z;
            og[3] = acc[l].w;
            if (grp == 0) {
                const float tot = wave_sum(y0[l]);
                if (lane == 0) o[0] = tot;
            }
        }
    }
}

int launch_listener_mix(const float *xdump, const float *xscale, const float *wtab32, const double *trows, float *out,
                        int nb, int m_pad, int n_modes, int n_listeners, long long out_stride, hipStream_t stream) {
    if (nb <= 0 || n_listeners <= 0) return 0;
    hipLaunchKernelGGL(listener_mix_kernel, dim3(2, nb), dim3(64), 0, stream, xdump, xscale, wtab32, trows, out, nb, m_pad,
                       n_modes, n_listeners, out_stride);
    return (int)hipGetLastError();
}

#endif      // PBSO_BLOCK_PART == 0

}  // namespace iir_block
}  // namespace pbso
