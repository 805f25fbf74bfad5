// Shared host/device declarations of the HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pbso {

// Wave priority of the preparation's kernels (profiles, projection, combine, increments, scan): they are small, they run BESIDE the
// oscillator bank, and the next bank waits for them (build option for A/B runs: scripts/debug/r05_prep_prio.sh)
#ifndef PBSO_PREP_PRIO
#define PBSO_PREP_PRIO 0
#endif
#if defined(__HIPCC__)
__device__ __forceinline__ void prep_prio() {
    if (PBSO_PREP_PRIO > 0) __builtin_amdgcn_s_setprio(PBSO_PREP_PRIO);
}
#endif


// one audio buffer is processed as tiles of TILE samples; 513 = 19 * 27.
// 27 rows x 68 floats = 7.3 KB of LDS per wave: 16+ waves per CU fit, which the
// VALU issue rate needs (one wave alone issues a v_fma_f32 every ~5.5 cycles,
// four per SIMD every ~2.4-2.8: profiles/r01_microbench.txt).
constexpr int TILE = 27;
// LDS row stride (floats) of the per-wave [TILE][64] transpose tile: 68 keeps
// 16-B alignment for ds_read_b128 and puts the 16 lanes of every b128 lane
// group on 16 distinct 4-bank slots (68 mod 64 = 4).
constexpr int LDS_ROW = 68;
constexpr int MAX_TILES = 32;          // tile mask is 32 bits
#ifdef PBSO_WAVE_TRACE
constexpr int TRACE_B0 = 40, TRACE_NB = 12, TRACE_K = 8;          // K1b diagnostics build: per-wave stamps of buffers 40 .. 51 behind the row
constexpr int CENSUS_WORDS = 12 + 2 * (2 + TRACE_NB * TRACE_K);
#else
constexpr int CENSUS_WORDS = 12;       // diagnostics row per workgroup (PBSO_CENSUS=1)
#endif

// transfer-row codes in BufDesc::trow
constexpr int XFER_KEEP = -1;          // keep _latest_transfer
constexpr int XFER_UNIT = -2;          // TransMessage::setToUnit, 1e7 (modal_solver.h:89-92)

constexpr uint32_t DESC_SKIP = 1u;     // step() returned early (modal_solver.h:186-189)
constexpr uint32_t DESC_IMPULSE = 2u;  // time profile is amp * delta[0] (PointForce): no profile row
// DESC_IMPULSE whose spatial vector the oscillator bank evaluates itself (the hit of a plain PointForce at a vertex,
// tools/real_time_modal_sound.cpp:276-280): frow = row 3 * vertex of the object's g32 table, the three floats
// behind amp's neighbours (prow, tile_mask, pad[0]) are the hit's normal.  No g row, no combine work for it.
constexpr uint32_t DESC_DIRECT = 4u;

// what ModalSolver::step's bookkeeping (modal_solver.h:184-256) decided for one
// (object, buffer); written by the host planner, read with scalar loads.
struct BufDesc {
    int32_t frow;        // row of g = c3 * S; -1: force-free buffer
    int32_t prow;        // row of the dense time profile (unused with DESC_IMPULSE)
    uint32_t tile_mask;  // bit t set: profile has a non-zero sample in tile t
    float amp;           // DESC_IMPULSE: profile value at sample 0
    int32_t trow;        // transfer row to switch to before this buffer, or XFER_*
    uint32_t flags;
    int32_t pad[2];
};
static_assert(sizeof(BufDesc) == 32, "BufDesc is read with one s_load_dwordx8");

// One workgroup ("team") of the oscillator bank: W waves stepping the columns
// [col0, col0 + 64 W R) of one object's SoA rows.  Objects that need more than
// MAX_WAVES_PER_OBJECT waves -- or that are worth spreading over several CUs -- are cut into
// several teams; each then writes its partial per-sample sums to row `part_row` of
// IirParams::audio_parts and sum_parts_kernel adds the rows in a fixed order.
struct TeamDesc {
    int obj;
    int col0;
    int part_row;                // -1: the team is the whole object and writes the audio itself
    int id;                      // global team index (census)
};

struct IirParams {
    const float *ca, *cb;        // [n_obj][m_pad] coefficients (form dependent)
    float *sq, *sd;              // [n_obj][m_pad] state (form dependent), times the per-mode scale in ss
    float *ss;                   // [n_obj][m_pad] scale carried by the stored state (1 = unscaled; kernels_iir.hip)
    const BufDesc *desc;         // [n_obj][nb]
    const float *grows;          // [n_frows][m_pad]  g = (float)(c3 * S)
    const float *g32;            // [sum of n_dof][m_pad]  (float)(c3[m] * shape[dof][m]): DESC_DIRECT hits take their g from three of its rows
    const long long *g32_off;    // [n_obj] first row of an object's table
    const float *tprof;          // [n_prows][b_pad]  dense force time profiles
    const double *xfer_rows;     // [n_rows][m_pad]   FFAT transfer rows (fp64)
    const int *xfer_init;        // [n_obj] row (or XFER_UNIT) in effect when the launch starts
    const TeamDesc *teams;       // [grid] workgroup -> (object, first column, output row); one launch per team size
    int qn_nb, qn_b0;            // qnorm is [n_obj][qn_nb][m_pad]; this launch fills buffers qn_b0 .. qn_b0 + nb - 1
    float *audio_parts;          // [n_part_rows][audio_stride] partial sums of objects split over several teams
    float *audio;                // [n_obj][audio_stride]
    float *qnorm;                // [n_obj][nb][m_pad] or nullptr
    const float *gq;             // closed-form qnorm: planes G11, 2*G12, G22, each [n_obj][m_pad]; or nullptr
    long long gq_plane;          // elements per plane
    unsigned long long *census;  // diagnostics: [n_teams][CENSUS_WORDS] = start, end (100 MHz), HW_ID, XCC_ID, clk0, clk1, (block form) cycles in head / pipeline / barrier / combine; or nullptr
    int nb, n_tiles, m_pad, b_pad;
    long long audio_stride;
    int rotate_prio;             // 1: rotate s_setprio per tile (fair progress of resident teams); 2: + progress feedback per CU
    unsigned *board;             // [4096] per-CU progress words for rotate_prio == 2
    unsigned launch_seq;
    // "this launch's bank has started": workgroup 0 stores start_seq here (signal memory) before anything else, and the engine's
    // preparation stream holds the NEXT launch's kernels back until it has (hipStreamWaitValue64) -- a preparation kernel that
    // starts together with a bank takes slots the bank's workgroups then wait for (128 x 512 x 86: 136 -> 234 us, once per
    // host hiccup: profiles/r04_timeline_share_128x512.txt).  nullptr: no gate.
    unsigned long long *start_flag = nullptr;
    unsigned long long start_seq = 0;
    // block state-space form (kernels_block.hip)
    const float *pc;             // planes P11 - 1, P12, P21, P22 of P = A^16 in the (q, q - q_prev) basis, each [n_obj][m_pad] (stride gq_plane)
    const float *wtab;           // [n_obj * m_pad / 2][64]: MFMA A operand per pair of columns (a_j, b_j of both modes, j = 1..16)
    int frames;                  // samples per buffer
    const float *ftab;           // forced block path: 32 planes [n_obj][m_pad] (stride gq_plane): A^(15-i) u, i = 0..15, components (q, d)
    int forced_block;            // block form, f32 projection: dense-profile buffers run in block form too (kernels_block.hip)
    // multi-listener mix: objects with dump_row[obj] >= 0 keep their block-start states (nullptr: nobody does)
    float *xdump;                // [n_dump][qn_nb][32][m_pad] pairs (Q, D), scaled as the registers hold them
    float *xscale;               // [n_dump][qn_nb][m_pad] the scale (transfer weight) of that buffer; 0: stepped per sample
    const int *dump_row;         // [n_obj]
    // time-chunked launch of the block form (kernels_scan.hip): tc_cb > 0 buffers per chunk, grid (team, chunk); the states at
    // the chunks' first buffers [n_obj][n_chunks][m_pad] pairs (q, d), unscaled, and the transfer rows in force there
    int tc_cb = 0;
    int lds_pad = 0;             // diagnostics (PBSO_TC_LDS_PAD): extra dynamic LDS per workgroup of a time-chunked block launch -- fewer teams per CU
    const float *tc_xs = nullptr;
    const int *tc_xtrow = nullptr;
    int census_stride = 0;       // census rows per chunk (= the launch's teams, all size classes)
};

// launches the oscillator bank for n_teams teams of waves_per_team waves; returns hipError_t as int.
// Supported shapes: waves_per_team <= MAX_WAVES_PER_TEAM, R in {1,2,4,8}.
// qnorm_mode: 0 off, 1 per-sample, 2 closed form (needs IirParams::gq).
namespace iir_scalar {
int launch_iir_bank(const IirParams &p, int n_teams, int modes_per_lane, int waves_per_team,
                    int form, int qnorm_mode, hipStream_t stream);
}
// per wave: one [TILE][LDS_ROW] transpose tile + a ring of (n_tiles + 1) tiles of row sums
// (+ 5 rows: lanes 54..63 of the LAST wave read rows 27..31 behind its tile for bank spreading -- values unused --
//  and with a one-wave team and few tiles the ring alone would not cover them)
inline size_t iir_lds_bytes(int W, int n_tiles) {
    return sizeof(float) * ((size_t)W * (TILE * LDS_ROW + (size_t)(n_tiles + 1) * TILE) + 5 * LDS_ROW);
}
constexpr int MAX_WAVES_PER_TEAM = 16;     // 1024 threads; larger objects are cut into several teams

// ---- K1b: block state-space form of the oscillator bank on the f32 matrix pipe (kernels_block.hip)
constexpr int BLOCK_J = 16;                // samples per block = rows of v_mfma_f32_16x16x4_f32
constexpr int BLOCK_N = 16;                // blocks per group = its columns; a buffer is 1 + n_groups * 256 samples
constexpr int BLOCK_STAGE_FLOATS = 2304;   // (split-bf16 projection: two planes of [16 blocks][72 dwords])
constexpr int BLOCK_STAGE_FLOATS_F32 = 2080;   // per wave: block-start states of one slice, [16 blocks][64 lanes][Q, D] + 2 per row
constexpr int BLOCK_RING_FLOATS = 516;     // per wave and buffer parity: the wave's partial sums of one buffer
// one mode per lane: a ring per GROUP of 256 samples (+ sample 0), combined at the end of every group -- half the LDS per wave
// (12 KB instead of 14: twelve waves of such teams fit a CU, three per SIMD, where eight did)
constexpr int BLOCK_HALF_RING_FLOATS = 260;
constexpr int MAX_WAVES_PER_BLOCK_TEAM = 8;
// split-bf16 projection: the kernel splits every block-start state into two TRUNCATED 8-bit parts, which loses
// 7.2e-6 of its value on average (measured on normal, log-normal and uniform data: 7.0 .. 7.3e-6); the operand table
// carries the inverse
constexpr double TRUNC_SPLIT_GAIN = 1.0 + 7.2e-6;
// per wave: the staging area, two rings, and the landing area of a direct hit's three g32 rows ([3][R][64] floats)
inline size_t block_lds_bytes(int W, int R) {
    return sizeof(float) * (size_t)W * (BLOCK_STAGE_FLOATS + 2 * (R == 1 ? BLOCK_HALF_RING_FLOATS : BLOCK_RING_FLOATS) + 3 * R * 64);
}
namespace iir_pipe {
// K1p (kernels_pipe.hip): teams of one producer wave (steps, parks block-start states) and n_consumers (1 or 2) consumer waves
// (project the previous buffer); one team per 64 modes
int launch_iir_pipe(const IirParams &p, int n_teams, int n_consumers, int qnorm_mode, hipStream_t stream);
}
namespace iir_block {
// modes_per_lane in {1,2,4}; qnorm_mode 0 off, otherwise closed form (+ per-sample in literal buffers)
// proj: 0 = f32 MFMA projection, 1 = split-bf16 projection (wtab holds the split table)
int launch_iir_block(const IirParams &p, int n_teams, int modes_per_lane, int waves_per_team, int qnorm_mode, int proj,
                     hipStream_t stream);
}
namespace iir_block {
// multi-listener mix of ONE object from the states a dump launch kept: xdump / xscale are that object's rows,
// wtab32 its f32 (a_j, b_j) table [m_pad / 2][64], trows [n_listeners][m_pad] transfer values, out [n_listeners][out_stride]
int launch_listener_mix(const float *xdump, const float *xscale, const float *wtab32, const double *trows, float *out,
                        int nb, int m_pad, int n_modes, int n_listeners, long long out_stride, hipStream_t stream);
}

// ---- K5: the scan of buffer-start states that makes a launch's buffers independent (kernels_scan.hip).  sc: 6 planes
// [n_obj][m_pad] (stride gq_plane): A^513 as (P11 - 1, P12, P21, P22), then A^512 u.  Writes the state at the first buffer of
// every chunk of cb buffers to xs, the transfer row in force there to xtrow, and the launch's end state to sq / sd / ss.
// vinc: the increments of the launch's dense-profile buffers (launch_dense_increments), or nullptr when it has none
// segmented (2 <= n_chunks <= SCAN_SEG_MAX): one wave per chunk scans its own buffers as an affine map of the state, the maps are
// composed in LDS -- the depth of the longest chunk instead of the launch's
constexpr int SCAN_SEG_MAX = 8;
int launch_iir_scan(const IirParams &p, int n_obj, const float *sc, int cb, int n_chunks, float *xs, int *xtrow, bool direct,
                    const float *vinc, bool segmented, hipStream_t stream);
// vinc[row][m_pad] pairs (q, d): what a unit force gain with the dense time profile tprof[row] leaves in every mode's state over one
// buffer, from rest (row_obj[row] = the object the row belongs to; pc / ftab: planes of P = A^16 and of A^(15-i) u, stride `plane`)
int launch_dense_increments(const float *pc, const float *ftab, long long plane, const float *tprof, const int *row_obj,
                            const int *n_modes, int n_rows, int m_pad, int b_pad, int frames, float *vinc, hipStream_t stream);

// ---- exact fp64 helper kernels (kernels_exact.hip, built with -ffp-contract=off)
struct ProjectEvent {
    int obj;
    int kind;            // PBSO_DATA_VERTEX / PBSO_DATA_FACE
    int vids[3];
    int slot;            // destination row of the data-slot pool
    double coords[3];
    double vn[3];
};
int launch_modal_project(const ProjectEvent *events, int n_events, const double *shapes,
                         const long long *shape_off, const int *n_modes, double *slots,
                         int m_pad, hipStream_t stream);
int launch_scatter_rows(const double *src, const int *dst_slot, int n_rows, double *slots,
                        int m_pad, hipStream_t stream);
// slot_idx >= 0: a row of the slot pool; < 0: event -(idx + 1) of `direct`, projected on the fly (and left in the pool when the
// event names a slot); an index past n_events: row (index - n_events) of the staged explicit data, copied to stage_slot[row] as well
int launch_force_combine(const int *row_ptr, const int *slot_idx, const int *row_obj, int n_rows,
                         double *slots, const double *c3, float *grows, const ProjectEvent *direct,
                         const double *shapes, const long long *shape_off, const int *n_modes, int m_pad,
                         int n_events, const double *stage, const int *stage_slot, hipStream_t stream);

// ---- K2: force time profiles on the device (forces.h:81-137)
struct ArState {         // AutoregressiveForce members that evolve (forces.h:62-72), one per live AR force
    uint32_t x;          // std::default_random_engine (minstd_rand0) state
    int32_t saved_available;
    double saved;        // std::normal_distribution's cached second variate
    double buf[3];
    int32_t buf_idx;
    int32_t pad;
    double a[2], sigma, mu;
};
struct ProfEntry {       // one active force contributing to one forced buffer's time profile
    int32_t kind;        // PBSO_POINT_FORCE / PBSO_GAUSSIAN_FORCE / PBSO_AUTOREGRESSIVE_FORCE
    int32_t state;       // AR: ArState slot
    int32_t flags;       // bit 0: default-construct the AR state first; bit 1: SetParam first
    int32_t count, center, width_samples;     // Gaussian: _count at the start of this buffer
    double a0, a1, sigma, mu;                 // SetParam values (bit 1)
};
struct ProfRow {         // one dense profile row = one (object, buffer)
    int32_t prow;        // row of the tprof array to write
    int32_t entry_begin, entry_end;
};
// chains: rows of one object in buffer order are generated by ONE wave (the AR state is sequential)
// ar_serial != 0: the AR(2) recurrence as the reference's serial loop (forces.h:107-117) instead of a parallel scan
int launch_force_profiles(const int *chain_ptr, int n_chains, const ProfRow *rows, const ProfEntry *entries,
                          ArState *states, float *tprof, int frames, int b_pad, int ar_serial, int high_prio, hipStream_t stream);

// ---- K2, row-parallel form: every row of a launch at once (kernels_exact.hip)
constexpr int K2_SEG = 1024;     // candidate pairs of a force's engine evaluated by one workgroup of ar_variates_kernel
struct ArStream {        // one AR force that adds samples in this launch
    int32_t state;       // ArState slot
    int32_t reset;       // its first use constructs it (engine at the default seed)
    int32_t n_uses;      // rows it contributes to
    int32_t use0;        // its uses' records: recs / cbuf rows use0 .. use0 + n_uses - 1
    int32_t seg_base, n_seg;     // its candidate segments
    int32_t pad[2];
};
struct ArUse {           // one AR entry of the launch (ProfEntry::count of an AR entry = its index here)
    int32_t entry;       // index in the launch's entries
    int32_t stream;
    int32_t u;           // use number within the stream: variates u * frames ... of the launch
    int32_t epoch_u;     // the use at which construction / SetParam last cleared the history; -1: the force's own history
    int32_t param_entry; // the entry that last set the parameters (flags != 0); -1: the force's own
    int32_t last;        // the stream's last use in this launch: writes the ArState back
    int32_t pad[2];
};
struct ArRec { double z1, z2, m00, m01, m10, m11; };     // per use: the state reached from rest, M = A^frames
struct ArFin { uint32_t x; int32_t saved_available; double saved; };      // per stream: the engine after the launch
int launch_force_rows(const ProfRow *rows, int n_rows, const ProfEntry *entries, const ArUse *uses, int n_uses,
                      const ArStream *streams, const int *seg_stream, int n_segs, int max_segs_per_stream, ArState *states,
                      ArState *snaps, double *vnorm, uint32_t *vstate, int *seg_count, double *cbuf, ArRec *recs, ArFin *fins,
                      float *tprof, int frames, int b_pad, int c_pitch, bool fused, hipStream_t stream);      // fused: every stream has ONE use -> one launch
// ... and, for a one-buffer launch, the fused profile rows AND the combine rows (launch_force_combine's arguments) in one launch
int launch_force_rows_combine(const ProfRow *rows, int n_rows, const ProfEntry *entries, const ArUse *uses, const ArStream *streams,
                              int max_segs_per_stream, ArState *states, ArState *snaps, double *vnorm, uint32_t *vstate, int *seg_count,
                              double *cbuf, ArRec *recs, ArFin *fins, float *tprof, int frames, int b_pad, int c_pitch,
                              const int *row_ptr, const int *slot_idx, const int *row_obj, int n_frows, double *slots, const double *c3,
                              float *grows, const struct ProjectEvent *direct, const double *shapes, const long long *shape_off, const int *n_modes,
                              int m_pad, int n_events, const double *stage, const int *stage_slot, hipStream_t stream);

struct FfatGeom {        // FFAT_Map<double,3> runtime fields, one per (object, mode)
    double k;
    double center3[3];
    double cell_size;
    double low_corners[6][3];
    double center[3];
    double bbox_low[3];
    double bbox_top[3];
    int n_elements[6][2];
    int strides[6];
    int n_psi;
    int valid;
    long long psi_off;   // offset into the psi pool (doubles)
};
struct FfatEvent {
    int obj;
    int row;             // destination transfer row
    double pos[3];
};
struct FfatRun { int obj, first, count, pad; };          // consecutive events of one object in the launch's list
// an object whose modes share one map geometry (kernels_exact.hip, ffat_lookup_shared_kernel): its maps transposed, psi_t[cell][mode]
struct FfatShared {
    long long psit_off;  // offset into the transposed pool (doubles)
    int pitch;           // doubles per cell row (>= n_modes, a multiple of 16); 0: the object's modes do not share a geometry
    int first_valid;     // a mode whose FfatGeom stands for all of them
};
int launch_ffat_lookup_shared(const FfatEvent *events, int n_events, const FfatShared *shared, const FfatGeom *geom, const long long *geom_off,
                              const int *n_modes, const double *mode_k, const int *mode_valid, const double *psi_t, double *rows, int m_pad,
                              hipStream_t stream);
int launch_ffat_lookup_runs(const FfatEvent *events, const FfatRun *runs, int n_runs, const FfatGeom *geom,
                            const long long *geom_off, const int *n_modes, const double *psi,
                            double *rows, int m_pad, hipStream_t stream);
int launch_ffat_lookup(const FfatEvent *events, int n_events, const FfatGeom *geom,
                       const long long *geom_off, const int *n_modes, const double *psi,
                       double *rows, int m_pad, hipStream_t stream);
// the same lookups for n_pos positions of ONE object (geom_of_object = its first FfatGeom), Psi staged in LDS per mode;
// writes rows[p][0 .. n_maps)
int launch_ffat_batch(const double *pos, int n_pos, const FfatGeom *geom_of_object, int n_maps, const double *psi,
                      double *rows, int m_pad, hipStream_t stream);
// audio[obj][i] = sum over the object's teams, in team order (deterministic), i < n; rows are `stride` apart
struct SplitObj { int obj, first_row, n_rows, pad; };
int launch_sum_parts(const SplitObj *split, int n_split, const float *parts, float *audio, long long stride, long long n,
                     hipStream_t stream);
// ... and, in the same launch, n_copy transfer rows copied (rows[dst_row[c]] = rows[src_row[c]]): the two jobs behind a bank launch
int launch_sum_parts_copy_rows(const SplitObj *split, int n_split, const float *parts, float *audio, long long stride, long long n,
                               const int *src_row, const int *dst_row, int n_copy, double *rows, int m_pad, hipStream_t stream);
int launch_copy_rows(const int *src_row, const int *dst_row, int n, double *rows, int m_pad,
                     hipStream_t stream);

// pbso_mix_objects: out[i] = sum over the n_obj rows of audio, i < n (rows `stride` apart); parts: mix_objects_groups(n_obj) x n floats
int mix_objects_groups(int n_obj);
int launch_mix_objects(const float *audio, int n_obj, long long stride, long long n, float *parts, float *out, hipStream_t stream);

// One wave that stores `value` (system scope, release) into signal memory: behind the last kernel of a stream's batch it tells a
// hipStreamWaitValue64 of another stream that the batch is done -- half the latency of an event (scripts/microbench/wait_value.hip)
int launch_signal_value(unsigned long long *sig, unsigned long long value, hipStream_t stream);

}  // namespace pbso
