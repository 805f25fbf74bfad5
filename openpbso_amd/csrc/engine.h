// Host side of the MI355X modal sound engine: one Engine = a batch of
// independent ModalSolver<double> instances ("objects") on one GPU.
//
// The host does what ModalSolver::step does before its hot loop
// (modal_solver.h:184-256: dequeue <=1 force message, active-force list,
// sustained forces, AR parameter updates, transfer selection) for every
// (object, buffer) of a batch and compiles the outcome into a table of
// BufDesc plus small work lists; the device then runs projection, force
// combination, FFAT lookups and the oscillator bank for the whole batch.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <atomic>
#include <deque>
#include <random>
#include <string>
#include <vector>

#include "../../include/openpbso_amd.h"
#include "kernels.h"

namespace pbso {
class SubmitQueue;       // submit_queue.h

// ---- growable device / pinned-host buffers ---------------------------------
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n, bool keep = false, hipStream_t s = nullptr);
    void release();
};
template <class T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n);
    hipError_t ensure_keep(size_t n, size_t keep);
    void release();
};
inline size_t arena_align(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- forces.h: time profile of one force (Point / Gaussian / AR(2)) ---------
struct ForceProfile {
    int type = PBSO_POINT_FORCE;
    bool used = false;                                   // PointForce, forces.h:28
    double width = 0;                                    // GaussianForce, forces.h:35-40
    int width_samples = 1, count = 0, center = 0, cutoff = 5;
    double buf[3] = {0, 0, 0};                           // AutoregressiveForce, forces.h:62-72
    int buf_idx = 0;
    double a[2] = {0.783, 0.116}, sigma = 0.00148, mu = 0.142;
    std::default_random_engine generator;                // default seed, copied with the message
    std::normal_distribution<double> distribution;
    static ForceProfile make(int type, double gaussian_width_us, int sample_rate);
    // Force::Add, forces.h:81-128.  *extent grows to the number of leading samples it touched.
    bool add(double *t, int frames, int *extent);
    void set_param(const double a_[2], double sigma_, double mu_);   // forces.h:130-137
};

// ForceMessage, modal_solver.h:27-77, as it waits in an object's queue: ONE cache line.  (A scene's queues hold a
// step's worth of messages -- megabytes that enqueue writes and the planner reads back: the 120-byte message
// with a std::vector inside cost two lines each way.)  What only some messages carry -- barycentric coordinates of
// a face hit, the Gaussian width, explicit modal data -- lives in a heap block owned by the message.
struct MsgExt {
    double coords[3];
    double gaussian_width_us;                            // GaussianForce(width); a queued message carries a pristine Force
    int n_data;
    double data[1];                                      // n_data doubles (ForceMessage::data as the GUI built it)
};
struct alignas(64) HostForceMsg {
    int64_t not_before = 0;
    double vn[3] = {0, 0, 0};
    int vids[3] = {0, 0, 0};
    int8_t force_type = PBSO_POINT_FORCE, data_kind = PBSO_DATA_ZERO;
    bool sustained_start = false, sustained_end = false, clear_all = false;
    MsgExt *ext = nullptr;                               // owned; nullptr for a plain vertex hit
    double coord(int j) const { return ext ? ext->coords[j] : 0.0; }
    double gaussian_width_us() const { return ext ? ext->gaussian_width_us : 0.0; }
};
static_assert(sizeof(HostForceMsg) == 64, "one cache line per queued message");

// _queue_force (modal_solver.h:105): FIFO of at most 1023 messages.  A ring that grows by doubling and
// never shrinks: no allocation per message (a node-based queue filled by the caller's thread and drained
// by the planner's threads makes every message a cross-thread free).
class ForceQueue {
public:
    bool empty() const { return n_ == 0; }
    size_t size() const { return n_; }
    HostForceMsg &front() { return buf_[head_]; }
    const HostForceMsg &front() const { return buf_[head_]; }
    HostForceMsg &back() { return buf_[(head_ + n_ - 1) & (buf_.size() - 1)]; }
    void pop_front() { head_ = (head_ + 1) & (buf_.size() - 1); --n_; }
    void push_back(HostForceMsg &&m) {
        if (n_ == buf_.size()) grow();
        buf_[(head_ + n_) & (buf_.size() - 1)] = std::move(m);
        ++n_;
    }

private:
    void grow() {
        std::vector<HostForceMsg> nb(buf_.empty() ? 8 : 2 * buf_.size());
        for (size_t i = 0; i < n_; ++i) nb[i] = std::move(buf_[(head_ + i) & (buf_.size() - 1)]);
        buf_.swap(nb);
        head_ = 0;
    }
    std::vector<HostForceMsg> buf_;                      // capacity is a power of two
    size_t head_ = 0, n_ = 0;
};

struct ActiveForce {                                     // one entry of _activeForces
    int slot = -1;                                       // row of the device data-slot pool
    int ar_state = -1;                                   // device ArState slot of an AutoregressiveForce
    int force_type = PBSO_POINT_FORCE;
    ForceProfile force;
};

struct TimedEvent {
    enum Kind { ARPRM, TRANSFER, USE_TRANSFER } kind;
    int64_t not_before;
    double v[4];                                         // arprm: a0,a1,sigma,mu; transfer: pos
    int flag;
};

struct Object {
    int n_modes = 0;
    std::vector<double> c1, c2, c3;                      // modal_integrator.h:95-99 (fp64)
    int n_dof = 0;
    std::vector<double> shapes;                          // mode-major until finalize
    bool have_maps = false;                              // _ffat_maps non-null
    bool maps_cover_modes = false;                       // ... and holds a map for every modeId 0 .. n_modes - 1 (set by finalize)
    int n_maps = 0;
    std::vector<FfatGeom> geom;                          // index = modeId
    std::vector<double> psi;
    // run-time state of the ModalSolver this object stands for
    ForceQueue force_q;                                  // _queue_force (1023 usable slots)
    std::vector<ActiveForce> active;                     // _activeForces
    bool sustained = false;                              // _sustainedForces
    bool arprm_full = false;                             // _queue_arprm (1 slot)
    double arprm[4] = {0, 0, 0, 0};
    bool trans_full = false;                             // _queue_trans (1 slot)
    int trans_row = XFER_UNIT;
    bool use_transfer = true;                            // _useTransfer
    int latest_row = XFER_UNIT;                          // where _latest_transfer lives (XFER_UNIT or own row)
    std::deque<TimedEvent> pending;                      // stamped arprm / transfer / use-transfer calls
    // Round 6: a listener PATH given in one pbso_compute_transfer_path call (this object's positions, stamps ascending) stays an
    // array with a cursor; the planner turns the positions of an otherwise quiet object straight into lookup events and transfer
    // rows (Engine::consume_path), and any call that orders itself against pending events moves the rest into `pending` first
    struct PathEv { int64_t stamp; double pos[3]; };
    std::vector<PathEv> path;
    size_t path_head = 0;
    bool path_left() const { return path_head < path.size(); }
};

// Everything one planning pass over a contiguous range of objects produces.  The planner runs one
// context per host thread (objects are independent, modal_solver.h:100-126) and merges them in object
// order, so forced-row / profile-row numbering does not depend on the thread count; only the ids of
// the pooled data slots do, and those are storage locations without meaning.
struct PlanCtx {
    std::vector<int> row_ptr, slot_idx, row_obj, stage_slot, chain_ptr;   // row_ptr: END offset of each forced row in slot_idx
    std::vector<int> prow_obj;                           // the object of every dense profile row (K5: dense_increment_kernel)
    std::vector<float> tprof;
    std::vector<ProfEntry> prof_entries;
    std::vector<ProfRow> prof_rows;
    std::vector<double> stage;
    std::vector<ProjectEvent> proj, proj_direct;
    std::vector<FfatEvent> ffat;
    std::vector<BufDesc *> forced;                       // descriptors holding context-local frow / prow numbers
    std::vector<int> free_slots, freed_this_plan, free_ar, freed_ar;     // this context's share of the slot / AR-state pools
    std::vector<double> tbuf;
    int t_extent = 0, n_frows = 0, n_prows = 0, n_xfer = 0, xfer_base = 0, chain_obj = -1;
    int rc = 0;
    std::string err;
    void begin();
};

class PlanPool;

class Engine {
public:
    explicit Engine(const pbso_engine_desc &d);
    ~Engine();
    int init();
    int add_object(const pbso_object_desc &d, int *id);
    int set_ffat_maps(int obj, const pbso_ffat_map *maps, int n);
    int finalize();
    int build_gq();                                      // closed-form qnorm matrices (once)
    int enqueue_force(int obj, const pbso_force_msg &m, int64_t not_before);
    int enqueue_force_batch(int n, const int *objs, const pbso_force_msg *msgs, const int64_t *stamps, unsigned char *accepted);
    int enqueue_vertex_hits(int n, const int *objs, const int *vids, const double *vn, const int64_t *stamps);
    int enqueue_arprm(int obj, const double a[2], double sigma, double mu, int64_t not_before);
    int arprm_pending(int obj);
    int compute_transfer(int obj, const double pos[3], int64_t not_before);
    int compute_transfer_path(int n, const int *objs, const double *pos, const int64_t *stamps, unsigned char *accepted);
    int compute_transfer_batch(int obj, const double *pos, int n_pos, double *out, int out_cols);
    int listeners_enable(int obj);
    int mix_listeners(int obj, const double *pos, int n_listeners, float *out, size_t n_out);
    int mix_objects(void *d_out);
    int object_n_maps(int obj);
    int set_use_transfer(int obj, int use, int64_t not_before);
    int get_latest_transfer(int obj, double *out);
    int step(int n_buffers, void *d_audio);
    int step_to_host(int n_buffers, float *host_out, size_t n);
    int host_wait();
    int step_chunk(int nb, int b0, int nb_total, float *audio, int64_t step_id);   // one launch of at most chunk_buffers_ buffers
    int sync();
    int drain_submit();                                  // the submitting thread has made every recorded call (not a device sync)
    int read_audio(float *out, size_t n);
    int read_audio_rows(const int *rows, int n_rows, float *out);
    int read_emitted(unsigned char *out, size_t n);
    int read_qnorm(int obj, int buffer, float *out, int n);
    int read_state(int obj, double *q1, double *q2, int n);
    int write_state(int obj, const double *q1, const double *q2, int n);
    void *audio_ptr() { return last_audio_; }
    int read_census(unsigned long long *out, size_t n);
    int info(pbso_engine_info *out);
    const char *last_error() const { return err_.c_str(); }
    int n_objects() const { return (int)objs_.size(); }
    int object_modes(int obj) const { return objs_[obj].n_modes; }

private:
    int fail(int code, const std::string &msg);
    int hip_fail(hipError_t e, const char *what);
    bool valid_obj(int obj) const { return obj >= 0 && obj < (int)objs_.size(); }
    int enqueue_force_impl(int obj, const pbso_force_msg &m, int64_t not_before, const char **why);
    int alloc_slot(PlanCtx &c);
    void release(PlanCtx &c, ActiveForce &af);
    int plan(int nb);                                    // host bookkeeping for one batch
    int plan_object(PlanCtx &c, int o, int b, int nb, int64_t t);
    int plan_object_span(PlanCtx &c, int o, int nb);
    // the borrowed script of plain vertex hits (pbso_enqueue_vertex_hits): arrays of the caller, valid until the next step
    // has planned; hit_off_[o] .. hit_off_[o + 1] are object o's hits
    struct HitScript { int n = 0; const int *objs = nullptr, *vids = nullptr; const double *vn = nullptr; const int64_t *stamps = nullptr; } script_;
    std::vector<int> hit_off_;
    std::atomic<int64_t> dropped_hits_{0};               // hits of a script that found their object's queue full (rejected as try_enqueue would)
    int script_to_queue(int oi, int h0, int h1, const char **why);     // hits h0 .. h1 - 1 of object oi enter its queue, in order
    int flush_script();                                                 // all of it (another enqueue call came before the step)
    int consume_script(PlanCtx &c, int oi, int nb);                     // planner: the object's hits of this launch
    // the listener's twin of the hit script (Object::path)
    void path_to_pending(Object &o, int64_t before);     // positions stamped below `before` enter the pending list, in order
    int consume_path(PlanCtx &c, int oi, int nb);
    static int cfail(PlanCtx &c, int code, const char *msg) { c.err = msg; return code; }

    pbso_engine_desc desc_;
    int B_ = PBSO_FRAMES_PER_BUFFER, rate_ = PBSO_SAMPLE_RATE, n_tiles_ = 9, b_pad_ = 528;
    int R_ = 0, W_ = 0, m_pad_ = 0;
    bool finalized_ = false, own_stream_ = false;
    hipStream_t stream_ = nullptr;                       // oscillator bank (caller's stream if given)
    hipStream_t prep_stream_ = nullptr;                  // plan upload + projection + FFAT + force profiles + combine + scan
    // round 5: launches with many dense-profile rows fork the preparation in two -- force profiles (K2) and the dense increments
    // stay on prep_stream_, projection + FFAT + combine go to aux_stream_ and join before the scan / the bank (step_chunk)
    hipStream_t aux_stream_ = nullptr;
    int prep_priority_ = 0;
    int prep_split_ = 1;                                 // PBSO_PREP_SPLIT (diagnostic): 0 never, 1 policy, 2 whenever there is anything to fork
    long long tot_prep_splits_ = 0;
    // The hand-over preparation -> bank (desc.stream_sync).  An event costs the waiting stream 10 - 12 us after the preparation's
    // last kernel (scripts/microbench/wait_value.hip, profiles/r04_stream_sync.txt); a value in signal memory, written by a
    // one-wave kernel behind that kernel, and a hipStreamWaitValue64 in front of the bank: 5 - 6 us.  Opt-in: the bank then starts
    // while the preparation's last workgroups still hold slots, and the step gains 1 % (128 x 512 x 86), not 4.
    unsigned long long *sig_prep_ = nullptr;             // hipMallocSignalMemory: the number of the last prepared launch
    unsigned long long prep_seq_ = 0;
    bool sync_values_ = false;
    // The start gate (policy: launches of >= 256 buffers; desc.stream_sync = 3: every launch): the bank kernel's first workgroup stores the launch's number into signal memory, and the preparation
    // kernels of the NEXT launch wait for that value -- they never start together with a bank (whose workgroups would then wait
    // for the slots they hold), only beside one that is resident, i.e. in the slots its workgroups free when they retire.
    unsigned long long *sig_start_ = nullptr;
    unsigned long long *host_start_ = nullptr;          // the same word in pinned host memory: the host form of the gate (policy)
    unsigned long long *host_start_dev_ = nullptr;      // ... as the device sees it
    bool host_gate_used_ = false;
    int gate_choice_ = 0;                                // pbso_engine_info::start_gate
    long long gate_timeouts_ = 0;
    unsigned long long bank_seq_ = 0, last_bank_seq_ = 0;
    bool start_gate_ = false;
    // Plan sets: the host plans and uploads step k while the device still runs step k - N_SETS + 1.  Two sets are enough for
    // a host that never stalls; with three, a hiccup of the host thread (the boxes of this pool stall it for a millisecond now
    // and then) is absorbed by the steps already queued instead of idling the device.  Not four: with three steps of preparation
    // kernels queued ahead, the 64 x 256 scene with a listener move per buffer loses the overlap between its steps (0.24 -> 0.36 ms
    // per step, scripts/debug/r03_sets.sh; the headline and the scraping scene do not care).
#ifndef PBSO_N_SETS
#define PBSO_N_SETS 3
#endif
    static constexpr int N_SETS = PBSO_N_SETS;
    hipEvent_t ev_prep_done_[N_SETS] = {}, ev_k1_done_[N_SETS] = {};
    hipEvent_t ev_aux_fork_[N_SETS] = {}, ev_aux_join_[N_SETS] = {};      // the preparation's fork to aux_stream_ and its way back
    // engines with several team sizes: the size classes are launched side by side on these streams
    // (one class alone rarely fills the chip), forked from and joined into stream_ with events
    static constexpr int N_CLASS_STREAMS = 3;
    hipStream_t class_stream_[N_CLASS_STREAMS] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork_ = nullptr, ev_join_[N_CLASS_STREAMS] = {nullptr, nullptr, nullptr};
    int xfer_cap_ = 0;                                   // scratch transfer rows per plan set
    hipEvent_t ev_set_[N_SETS] = {};
    struct EvQuad { hipEvent_t k0, k1, p0, p1, f0, f1; int64_t step_id; bool has_k2; double h_enter, h_prep, h_copy, h_bank, h_done; unsigned long long batch = 0; };      // bank, pipeline, force-profile kernel
    std::vector<EvQuad> ev_free_, ev_pending_;           // one quad per step, harvested in info()
    int harvest_timing(bool blocking);
    double tot_kernel_ms_ = 0, tot_device_ms_ = 0, tot_plan_ms_ = 0, last_kernel_ms_ = 0, last_device_ms_ = 0;
    int64_t tot_steps_ = 0, tot_block_launches_ = 0, tot_sample_launches_ = 0, tot_timed_launches_ = 0;
    int timing_every_ = 1;                               // PBSO_TIMING_EVERY=n: HIP-event pairs around every n-th launch only (0: none)
    std::string err_;
    std::vector<Object> objs_;
    int64_t buffers_done_ = 0;
    int cur_set_ = 0;

    // persistent device state
    DevBuf<float> d_ca_, d_cb_, d_sq_, d_sd_, d_ss_;
    DevBuf<double> d_c3_;
    struct SizeClass { int W, first, count; };           // teams of W waves: d_teams_[first, first+count)
    std::vector<SizeClass> classes_;
    DevBuf<TeamDesc> d_teams_;
    DevBuf<SplitObj> d_split_;                           // objects stepped by more than one team
    DevBuf<float> d_audio_parts_;                        // [n_part_rows_][nb * B] their partial sample sums
    int n_teams_ = 0, n_split_ = 0, n_part_rows_ = 0;
    // K1p (kernels_pipe.hip): the team table of the pipeline kernel -- one team per 64 columns -- for engines with less than
    // a wave of oscillators per SIMD (f32 block form, one mode per lane; desc.bank_kernel = BLOCK: never)
    bool split_ok_ = false, split_always_ = false;      // bank_kernel = PIPE: every launch it can run, not only the dense ones
    DevBuf<TeamDesc> d_ts_teams_;
    DevBuf<SplitObj> d_ts_split_;
    int n_ts_teams_ = 0, n_ts_split_ = 0, n_ts_part_rows_ = 0;
    int64_t tot_split_launches_ = 0;
    bool use_split() const { return split_ok_ && n_dump_ == 0; }
    // K5 (kernels_scan.hip): launches whose buffers are made independent by a scan of the buffer-start states run the block
    // kernel K1b as (team, chunk of buffers) workgroups.  The chunked launches pick their team shape per launch -- R modes per
    // lane, whole objects as teams of up to 8 waves -- from three tables (the state arrays are indexed by column: any shape
    // reads them), so a short launch of a small scene takes many small teams and a long one few large ones.
    struct TcSet {
        int R = 0;
        std::vector<SizeClass> classes;
        DevBuf<TeamDesc> d_teams;
        DevBuf<SplitObj> d_split;
        int n_teams = 0, n_split = 0, n_part_rows = 0;
        long long waves = 0, cover = 0;                  // waves of all teams; columns they cover (padding included)
        int waves_per_cu = 8;                            // of this shape's build, resident at once (registers, LDS)
    } tc_[3];                                            // R = 1, 2, 4
    bool tc_ok_ = false;
    int tc_mode_ = 0;                                    // pbso_engine_desc::time_chunks: 0 auto, < 0 never, n > 0 chunks of n buffers always
    DevBuf<float> d_scan_;                               // 6 planes [n_obj][m_pad]: A^513 (P11 - 1, P12, P21, P22), A^512 u
    // per plan set (the scan of launch k + 1 runs beside the bank of launch k):
    DevBuf<float> d_xs_[N_SETS];                         // [n_obj][n_chunks][m_pad] pairs: the state at the first buffer of every chunk
    DevBuf<int> d_xtrow_[N_SETS];                        // [n_obj][n_chunks] the transfer row in force there
    DevBuf<float> d_vinc_[N_SETS];                       // [n_prows][m_pad] pairs: the increments of the launch's dense-profile buffers (dense_increment_kernel)
    int tc_shape_ = 0;                                   // pbso_engine_desc::time_chunk_shape: 0 policy, 1 / 2 / 4 modes per lane in time-chunked launches
    int64_t tot_tc_dense_launches_ = 0, tot_seg_scans_ = 0;
    int last_tc_shape_ = 0, last_tc_cb_ = 0, last_tc_teams_ = 0;
    bool last_launch_tc_ = false;                        // the previous launch was time-chunked (its bank did not write the state)
    int last_set_ = -1;
    bool latency_path_ = true, last_one_stream_ = false;  // desc.latency_path; the previous launch prepared on the bank's stream
    int64_t tot_one_stream_launches_ = 0;
    int64_t tot_tc_launches_ = 0;
    bool choose_time_chunks(int nb, int n_dense_rows, int *set, int *cb) const;
    int n_cus_ = 256;                                     // hipDeviceProp_t::multiProcessorCount of the engine's device
    long long total_team_waves_ = 0;
    DevBuf<float> d_gq_;                                 // closed-form qnorm: G11, 2 G12, G22 planes
    // multi-listener mix: objects that keep their block-start states (row per object), the f32 (a_j, b_j) tables of
    // those objects, and whether the last step's states are usable (block launches only, no dense-profile buffer)
    std::vector<int> dump_row_;
    std::vector<char> dump_valid_;
    int n_dump_ = 0, dump_nb_ = 0;
    bool dump_rows_dirty_ = false;
    DevBuf<int> d_dump_row_;
    DevBuf<float> d_xdump_, d_xscale_, d_wtab32_;
    DevBuf<float> d_ftab_;                               // A^(15-i) u per mode: the forced block path (engines with <= 2 modes per lane) and K5's dense increments
    bool ftab_forced_ = false;                           // ... the bank's forced block path may use it
    DevBuf<float> d_pc_, d_wtab_;                        // block form: P = A^16 planes and the MFMA W table (kernels_block.hip)
    bool is_block() const { return form_ == PBSO_FORM_BLOCK || form_ == PBSO_FORM_BLOCK_BF16; }
    int form_ = PBSO_FORM_BLOCK;                         // the form that runs (block falls back to velocity for odd buffer lengths)
    bool dense_to_k1_ = true;                            // dense-heavy launches run on the per-sample kernel K1 (split-bf16 form; PBSO_DENSE_LAUNCHES)
    bool forced_block_ = true;                           // f32 block kernel: dense-profile buffers in block form (PBSO_FORCED_BLOCK=0: per sample)
    DevBuf<double> d_shapes_;
    DevBuf<long long> d_shape_off_;
    std::vector<long long> geom_off_h_;                  // host copy of d_geom_off_ (first FfatGeom of every object)
    // (float)(c3[m] * shape[dof][m]), [dof][m_pad] per object (rows as d_shapes_: d_g32_off_ = d_shape_off_ / m_pad): the
    // oscillator bank takes the spatial vector of a plain PointForce vertex hit from three of its rows (DESC_DIRECT)
    DevBuf<float> d_g32_;
    DevBuf<long long> d_g32_off_;
    bool direct_hits_ = true;                            // PBSO_DIRECT_HITS=0: such hits go through the combine kernel
    DevBuf<int> d_n_modes_;
    DevBuf<FfatGeom> d_geom_;
    DevBuf<long long> d_geom_off_;
    DevBuf<double> d_psi_;
    // objects whose modes share one FFAT map geometry (finalize): the maps once more, transposed (kernels.h, FfatShared)
    DevBuf<double> d_psi_t_, d_ffat_k_;
    DevBuf<int> d_ffat_valid_;
    DevBuf<FfatShared> d_ffat_shared_;
    std::vector<unsigned char> ffat_shared_h_;           // per object: its listener events go to ffat_lookup_shared_kernel
    int n_ffat_shared_ = 0;
    int64_t tot_ffat_shared_events_ = 0, tot_ffat_general_events_ = 0;
    DevBuf<double> d_slots_;                             // [n_slots][m_pad] ForceMessage::data rows
    DevBuf<double> d_xfer_;                              // [n_obj + scratch][m_pad]
    DevBuf<float> d_audio_, d_qnorm_;
    DevBuf<float> d_mix_parts_;                          // pbso_mix_objects: partial rows of the object groups
    // pbso_step_to_host: two device audio buffers used in turn, a copy stream, events both ways
    DevBuf<float> d_audio_host_[2];
    hipStream_t copy_stream_ = nullptr;
    hipEvent_t ev_host_bank_[2] = {nullptr, nullptr}, ev_host_copy_[2] = {nullptr, nullptr};
    int host_slot_ = 0, host_last_ = -1;
    DevBuf<unsigned long long> d_census_;                // PBSO_CENSUS=1: per-workgroup placement/timing
    bool census_ = false;
    int rotate_prio_ = 2;                                // PBSO_ROTATE_PRIO: 0 off, 1 rotation, 2 rotation + per-CU progress feedback
    DevBuf<unsigned> d_board_;                           // per-CU progress words of the feedback
    unsigned launch_seq_ = 0;                            // PBSO_ROTATE_PRIO: see kernels_iir.hip
    float *last_audio_ = nullptr;
    int last_nb_ = 0;
    std::atomic<size_t> n_slots_{0};

    // per-launch plan, double-buffered (host pinned + device copies)
    // per-launch plan, double-buffered: one pinned arena and its device copy ([BufDesc table | xfer_init | lists])
    struct PlanSet {
        PinBuf<unsigned char> h_arena;
        DevBuf<unsigned char> d_arena;
        DevBuf<float> d_tprof;                           // device-generated force profile rows (K2)
        size_t off_xfer_init = 0, front_bytes = 0, last_bytes = 0;
        void release();
    } set_[N_SETS];
    DevBuf<float> d_grows_[N_SETS];                      // g rows, one arena per plan set
    // plan scratch (host)
    std::vector<int> row_ptr_, slot_idx_, row_obj_, stage_slot_, busy_, prow_obj_;
    std::vector<float> tprof_;
    int n_frows_ = 0, n_prows_ = 0;
    // K2: device-side time profiles
    bool device_profiles_ = true;                        // PBSO_DEVICE_PROFILES=0: host fp64 profiles, uploaded
    // Wave priority of the force-profile kernel K2 (s_setprio 0..3).  K2 is a handful of latency-bound workgroups that run beside
    // the oscillator bank of the previous step; whichever of the two is longer bounds the step, and they share SIMDs.  Auto
    // (PBSO_K2_PRIO unset): 3 while the timed launches say K2 is the longer one, 0 otherwise -- a scheduling hint only, results
    // do not depend on it (8 x 4096 scraping without qnorm rows: 1400 -> 1460 x; with them the bank is longer and 0 is right).
    int k2_prio_ = 0;
    bool k2_prio_auto_ = true;
    double k2_ms_avg_ = 0, k2_bank_ms_avg_ = 0;
    int k2_ms_n_ = 0;
    bool ar_serial_ = false;                             // PBSO_AR_SERIAL=1: K2 runs the AR(2) recurrence as the reference's serial loop (16 us per row)
    std::vector<ProfEntry> prof_entries_;
    std::vector<ProfRow> prof_rows_;
    std::vector<int> chain_ptr_;
    std::atomic<size_t> n_ar_states_{0};
    DevBuf<ArState> d_arstate_;
    // K2, row-parallel form (the default; PBSO_K2_ROWS=0 or PBSO_AR_SERIAL=1: one workgroup walks an object's rows in order).
    // build_ar_tables() lists the launch's AR forces (streams), their uses and the candidate segments of their engines.
    bool k2_rows_ = true;
    bool timeline_ = false, timeline_have_base_ = false, timeline_keep_ = false;   // PBSO_TIMELINE=1 (diagnostics)
    bool host_profile_ = false;                          // PBSO_HOST_PROFILE=1 (diagnostics): host milliseconds by stage at destruction
    hipEvent_t timeline_ref_ = nullptr;
    EvQuad timeline_quad_ = {};                          // the launch the timeline's device times are relative to (its events live until the engine dies)
    double timeline_h0_ = 0;
    int k2_margin_pct_ = 100;                            // PBSO_K2_MARGIN_PCT: scales the candidate range (tests: < 100 forces the shortfall path)
    bool k2_rows_launch_ = false;                        // this launch takes the row-parallel form
    bool fuse_short_ = true;                             // pbso_engine_desc::fuse_short_launches
    // the second submitting thread (submit_queue.h; pbso_engine_desc::submit_thread): created by finalize, nullptr = every call at once
    SubmitQueue *submit_ = nullptr;
    unsigned long long set_batch_[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // per plan set: the batch of the launch that last used it
    std::vector<ArStream> ar_streams_;
    std::vector<ArUse> ar_uses_;
    std::vector<int> seg_stream_, ar_stream_of_state_, ar_last_use_, ar_epoch_, ar_param_;
    int ar_max_segs_ = 0;
    DevBuf<ArState> d_ar_snaps_;
    DevBuf<double> d_ar_vnorm_, d_ar_cbuf_;
    DevBuf<uint32_t> d_ar_vstate_;
    DevBuf<int> d_ar_segcount_;
    DevBuf<ArRec> d_ar_recs_;
    DevBuf<ArFin> d_ar_fins_;
    void build_ar_tables();
    int warm_copy_engines();
    std::vector<double> stage_;
    std::vector<ProjectEvent> proj_, proj_direct_;         // projections into pool rows / evaluated on the fly by the combine kernel
    BufDesc *plan_desc_ = nullptr;                       // the descriptor table being planned (front of the set's arena)
    std::vector<FfatEvent> ffat_;
    std::vector<FfatRun> ffat_runs_;
    std::vector<unsigned char> emitted_;
    // planner threads (PBSO_PLAN_THREADS, default 1): ctx_[t] plans a contiguous share of the busy objects.
    // More than one only pays when the threads share a last-level cache with the caller: on the 2-socket
    // EPYC hosts of the MI355X boxes unpinned helpers made planning AND the caller's enqueue slower (the
    // queues' cache lines migrate between cores every step), so the default is the caller's thread alone;
    // PBSO_PLAN_PIN=1 pins the helpers into the caller's 8-core complex (4 threads: 0.69 -> 0.35 ms).
    std::vector<PlanCtx> ctx_;
    PlanPool *pool_ = nullptr;
    int plan_threads_ = 1, plan_grain_ = 64;
    double last_plan_ms_ = 0;
    bool failed_ = false;                                // a step failed after it had started to consume messages
    std::string failed_why_;
    int chunk_buffers_ = 128;                            // longer steps are cut into launches of this many buffers
    int plan_b0_ = 0, plan_nb_total_ = 0;                // where the chunk being planned sits in the step
    int64_t harvest_step_ = -1;
    double hprof_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};               // PBSO_HOST_PROFILE=1: host milliseconds by stage, printed at destruction
    int64_t last_frows_ = 0, last_trows_ = 0;
};

// loaders (loaders.cpp)
int load_modes_file(const char *path, int *n_dof, int *n_modes, std::vector<double> &omega2,
                    std::vector<double> &modes);
int num_modes_audible(const std::vector<double> &omega2, double density, double audible_freq);
int load_material_file(const char *path, double out[5]);
int parse_fatcube(const unsigned char *bytes, size_t n, pbso_ffat_map *out);
int list_dir_files(const char *dir, const char *contains, std::vector<std::string> &names);
int read_file_bytes(const char *path, std::vector<unsigned char> &out);
int load_obj_file(const char *path, std::vector<double> &V, std::vector<int> &F, std::vector<double> &VN);

}  // namespace pbso
