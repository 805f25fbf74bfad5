// Host engine: see engine.h.  Compiled by hipcc for the HIP runtime API only;
// there is no CPU compute path here -- every sample is produced by the HIP
// kernels in kernels_iir.hip / kernels_exact.hip.
#include "engine.h"
#include "plan_pool.h"
#include "submit_queue.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>

#include <pthread.h>
#include <sched.h>

namespace pbso {

void PlanCtx::begin() {
    row_ptr.clear(); slot_idx.clear(); row_obj.clear(); stage_slot.clear(); chain_ptr.clear(); prow_obj.clear();
    tprof.clear(); prof_entries.clear(); prof_rows.clear(); stage.clear(); proj.clear(); proj_direct.clear(); ffat.clear();
    forced.clear(); freed_this_plan.clear(); freed_ar.clear();
    n_frows = n_prows = n_xfer = 0;
    chain_obj = -1;
    rc = 0;
    err.clear();
}

// ---------------------------------------------------------------------------
// Blocks replaced by a larger one are not freed on the spot: launches already queued on either engine
// stream may still use them, and waiting for those would drain the whole device in the middle of a step --
// the caller's own work (torch, an RCCL gather) included.  They are parked here and freed the next time
// the engine is idle anyway (Engine::sync, destruction).
namespace {
std::mutex g_retired_mutex;
std::vector<void *> g_retired;
}  // namespace
void free_retired_blocks() {
    std::vector<void *> v;
    {
        std::lock_guard<std::mutex> lk(g_retired_mutex);
        v.swap(g_retired);
    }
    if (v.empty()) return;
    // the list is shared by the engines of a process (one per ModalSolver in the facade): another engine's
    // launches may still use a block parked here, so this rare path waits for the whole device
    (void)hipDeviceSynchronize();
    for (void *q : v) (void)hipFree(q);
}
// Does something in this process's environment let only ONE kernel run at a time?  Returns its name, or nullptr.
//   rocprofv3 --pmc (rocprofiler-sdk counter collection serialises dispatches): ROCPROF_COUNTER_COLLECTION / ROCPROF_COUNTERS
//   are set for the profiled process; the older rocprof sets ROCP_METRICS / ROCPROFILER_METRICS.  A kernel trace alone
//   does not serialise (the default command runs under --kernel-trace --stats with the gate).
//   AMD_SERIALIZE_KERNEL (HIP: wait before / after every kernel launch), HIP_LAUNCH_BLOCKING / CUDA_LAUNCH_BLOCKING.
//   PBSO_START_GATE=0 says so by hand (a tool this list does not know); PBSO_START_GATE=1 overrides the list.
static const char *serialising_environment() {
    if (const char *v = std::getenv("PBSO_START_GATE")) return std::atoi(v) ? nullptr : "PBSO_START_GATE=0";
    auto on = [](const char *name) {
        const char *v = std::getenv(name);
        return v && *v && std::strcmp(v, "0") != 0;
    };
    static const char *const names[] = {"ROCPROF_COUNTER_COLLECTION", "ROCPROF_COUNTERS", "ROCPROF_PMC", "ROCP_METRICS", "ROCPROFILER_METRICS",
                                        "ROCPROFILER_PC_SAMPLING_BETA_ENABLED", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING", "CUDA_LAUNCH_BLOCKING"};
    for (const char *n : names)
        if (on(n)) return n;
    return nullptr;
}
std::atomic<long long> g_alloc_events{0};       // buffer (re)allocations so far (PBSO_TIMELINE diagnostics)
template <class T>
hipError_t DevBuf<T>::ensure(size_t n, bool keep, hipStream_t s) {
    if (n <= cap) return hipSuccess;
    g_alloc_events += 1;
    // 25 % headroom: per-step demand (forced rows, slots) fluctuates by a few percent
    size_t ncap = std::max(n + n / 4, cap + cap / 2);
    T *np = nullptr;
    hipError_t e = hipMalloc((void **)&np, ncap * sizeof(T));
    if (e != hipSuccess) return e;
    if (p) {
        // `keep`: the contents move in stream order on s -- every writer and reader of a kept buffer is ordered
        // after s (the preparation stream, or an event chain from it), so nobody sees the new block before the copy
        if (keep && cap) {
            e = hipMemcpyAsync(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) { (void)hipFree(np); return e; }
        }
        std::lock_guard<std::mutex> lk(g_retired_mutex);
        g_retired.push_back(p);
    }
    p = np;
    cap = ncap;
    return hipSuccess;
}
template <class T>
void DevBuf<T>::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
template <class T>
hipError_t PinBuf<T>::ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    size_t ncap = std::max(n + n / 4, cap + cap / 2);
    T *np = nullptr;
    hipError_t e = hipHostMalloc((void **)&np, ncap * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (p) (void)hipHostFree(p);
    p = np;
    cap = ncap;
    return hipSuccess;
}
// grows and keeps the first `keep` elements (the plan's fixed front part is written in place before the
// variable part is known)
template <class T>
hipError_t PinBuf<T>::ensure_keep(size_t n, size_t keep) {
    if (n <= cap) return hipSuccess;
    g_alloc_events += 1000;
    size_t ncap = std::max(n + n / 4, cap + cap / 2);
    T *np = nullptr;
    hipError_t e = hipHostMalloc((void **)&np, ncap * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (p) {
        if (keep) std::memcpy(np, p, std::min(keep, cap) * sizeof(T));
        (void)hipHostFree(p);
    }
    p = np;
    cap = ncap;
    return hipSuccess;
}
template <class T>
void PinBuf<T>::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}

void Engine::PlanSet::release() {
    h_arena.release(); d_arena.release(); d_tprof.release();
}

// ---------------------------------------------------------------------------
// forces.h
ForceProfile ForceProfile::make(int type, double gaussian_width_us, int sample_rate) {
    ForceProfile f;
    f.type = type;
    if (type == PBSO_GAUSSIAN_FORCE) {                       // forces.h:42-46
        f.width = gaussian_width_us;
        f.width_samples = std::max(1, (int)(f.width / 1000000. * sample_rate));
        f.center = (int)((f.cutoff - 0.5) * f.width_samples);
    }
    return f;
}

bool ForceProfile::add(double *t, int frames, int *extent) {
    switch (type) {
    case PBSO_POINT_FORCE:                                   // forces.h:81-90
        if (used) return false;
        t[0] += 1.;
        used = true;
        *extent = std::max(*extent, 1);
        return true;
    case PBSO_GAUSSIAN_FORCE:                                // forces.h:92-105
        if (width == 0 || count >= cutoff * 2 * width_samples) return false;
        *extent = frames;
        for (int ii = 0; ii < frames; ++ii) {
            const double p = -0.5 * std::pow((double)(count + ii - center) / (double)width_samples, 2);
            t[ii] += std::exp(p);
        }
        count += frames;
        return true;
    case PBSO_AUTOREGRESSIVE_FORCE:                          // forces.h:107-128
        *extent = frames;
        for (int ii = 0; ii < frames; ++ii) {
            double mu_tilde = 0.0;
            for (int jj = 0; jj < 2; ++jj) mu_tilde += a[jj] * buf[(buf_idx + 3 - jj - 1) % 3];
            mu_tilde += sigma * distribution(generator);
            buf[buf_idx] = mu_tilde;
            buf_idx = (buf_idx + 1) % 3;
            t[ii] += mu + mu_tilde;
        }
        return true;
    }
    return false;
}

void ForceProfile::set_param(const double a_[2], double sigma_, double mu_) {   // forces.h:130-137
    buf[0] = buf[1] = buf[2] = 0;
    a[0] = a_[0];
    a[1] = a_[1];
    sigma = sigma_;
    mu = mu_;
}

// ---------------------------------------------------------------------------
Engine::Engine(const pbso_engine_desc &d) : desc_(d) {}

Engine::~Engine() {
    for (Object &o : objs_)
        while (!o.force_q.empty()) { std::free(o.force_q.front().ext); o.force_q.pop_front(); }
    if (host_profile_ && tot_steps_ > 0)
        std::fprintf(stderr, "pbso host profile, ms per step over %lld steps: wait-for-set %.3f | plan: fill %.3f objects %.3f merge %.3f | "
                             "submit (uploads + launches) %.3f = pack + upload %.3f, preparation launches %.3f, bank launches %.3f | step total %.3f\n",
                     (long long)tot_steps_, hprof_[0] / tot_steps_, hprof_[1] / tot_steps_, hprof_[2] / tot_steps_, hprof_[3] / tot_steps_,
                     hprof_[4] / tot_steps_, hprof_[6] / tot_steps_, hprof_[7] / tot_steps_, hprof_[8] / tot_steps_, hprof_[5] / tot_steps_);
    delete pool_;
    delete submit_;                                    // (the worker makes what it still holds, then ends)
    submit_ = nullptr;
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (aux_stream_) (void)hipStreamSynchronize(aux_stream_);
    if (timeline_have_base_) {                         // (the reference launch's quad was kept out of the free list)
        for (hipEvent_t ev : {timeline_quad_.k0, timeline_quad_.k1, timeline_quad_.p0, timeline_quad_.p1, timeline_quad_.f0, timeline_quad_.f1})
            if (ev) (void)hipEventDestroy(ev);
    }
    if (prep_stream_) (void)hipStreamSynchronize(prep_stream_);
    free_retired_blocks();
    d_ca_.release(); d_cb_.release(); d_sq_.release(); d_sd_.release(); d_ss_.release(); d_c3_.release(); d_gq_.release();
    d_shapes_.release(); d_shape_off_.release(); d_g32_.release(); d_g32_off_.release(); d_n_modes_.release(); d_geom_.release();
    d_geom_off_.release(); d_psi_.release(); d_psi_t_.release(); d_ffat_k_.release(); d_ffat_valid_.release(); d_ffat_shared_.release(); d_slots_.release(); d_xfer_.release();
    d_pc_.release(); d_wtab_.release(); d_ftab_.release(); d_dump_row_.release(); d_xdump_.release(); d_xscale_.release(); d_wtab32_.release();
    d_ar_snaps_.release(); d_ar_vnorm_.release(); d_ar_cbuf_.release(); d_ar_vstate_.release(); d_ar_segcount_.release();
    d_ar_recs_.release(); d_ar_fins_.release();
    d_arstate_.release(); d_board_.release(); d_teams_.release(); d_split_.release(); d_ts_teams_.release(); d_ts_split_.release();
    d_audio_parts_.release(); d_audio_.release(); d_qnorm_.release(); d_mix_parts_.release();
    d_audio_host_[0].release(); d_audio_host_[1].release();
    if (copy_stream_) {
        (void)hipStreamSynchronize(copy_stream_);
        (void)hipStreamDestroy(copy_stream_);
        for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(ev_host_bank_[i]); (void)hipEventDestroy(ev_host_copy_[i]); }
    }
    d_census_.release();
    d_scan_.release();
    for (int i = 0; i < N_SETS; ++i) { d_xs_[i].release(); d_xtrow_[i].release(); d_vinc_[i].release(); }
    for (TcSet &ts : tc_) { ts.d_teams.release(); ts.d_split.release(); }
    for (DevBuf<float> &g : d_grows_) g.release();
    for (int i = 0; i < N_SETS; ++i)
        for (hipEvent_t ev : {ev_prep_done_[i], ev_k1_done_[i], ev_set_[i], ev_aux_fork_[i], ev_aux_join_[i]})
            if (ev) (void)hipEventDestroy(ev);
    if (prep_stream_) (void)hipStreamDestroy(prep_stream_);
    if (aux_stream_) (void)hipStreamDestroy(aux_stream_);
    if (sig_prep_) (void)hipFree(sig_prep_);
    if (sig_start_) (void)hipFree(sig_start_);
    if (host_start_) (void)hipHostFree(host_start_);
    for (hipStream_t cs : class_stream_)
        if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    for (hipEvent_t ev : ev_join_)
        if (ev) (void)hipEventDestroy(ev);
    for (PlanSet &ps : set_) ps.release();
    for (auto *v : {&ev_free_, &ev_pending_})
        for (EvQuad &q : *v)
            for (hipEvent_t ev : {q.k0, q.k1, q.p0, q.p1, q.f0, q.f1}) (void)hipEventDestroy(ev);
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
}

int Engine::fail(int code, const std::string &msg) {
    err_ = msg;
    return code;
}
int Engine::hip_fail(hipError_t e, const char *what) {
    err_ = std::string(what) + ": " + hipGetErrorString(e);
    return PBSO_ERR_HIP;
}
#define HIPTRY(expr)                                                   \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) return hip_fail(_e, #expr);              \
    } while (0)
#define LAUNCHTRY(expr)                                                \
    do {                                                               \
        int _e = (expr);                                               \
        if (_e != 0) return hip_fail((hipError_t)_e, #expr);           \
    } while (0)
// Engine::step_chunk's stream calls: made at once, or -- with the second submitting thread (submit_queue.h) -- recorded with their
// arguments evaluated HERE and made by the worker.  `defer` and `ops` are step_chunk's locals.
#define QHIP(fn, ...)                                                                      \
    do {                                                                                   \
        if (defer) ops.push_back(make_submit_op(#fn, fn, __VA_ARGS__));                    \
        else HIPTRY(fn(__VA_ARGS__));                                                      \
    } while (0)
#define QLAUNCH(fn, ...)                                                                   \
    do {                                                                                   \
        if (defer) ops.push_back(make_submit_op(#fn, fn, __VA_ARGS__));                    \
        else LAUNCHTRY(fn(__VA_ARGS__));                                                   \
    } while (0)
// a device buffer that has to GROW while recorded calls are still waiting: its keep-copy and the retirement of the old block are
// immediate calls and must come behind them in their streams -- wait for the worker first (rare: the first steps of an engine)
#define GROWTRY(buf, n, keep, s)                                                           \
    do {                                                                                   \
        if (defer && (size_t)(n) > (buf).cap) {                                            \
            int _d = drain_submit();                                                       \
            if (_d != PBSO_OK) return _d;                                                  \
        }                                                                                  \
        HIPTRY((buf).ensure((n), (keep), (s)));                                            \
    } while (0)

int Engine::init() {
    if (desc_.abi_version != PBSO_ABI_VERSION) return fail(PBSO_ERR_INVALID, "abi_version mismatch");
    if (desc_.frames_per_buffer > 0) B_ = desc_.frames_per_buffer;
    if (desc_.sample_rate > 0) rate_ = desc_.sample_rate;
    if (B_ % TILE != 0 || B_ / TILE > MAX_TILES)
        return fail(PBSO_ERR_INVALID, "frames_per_buffer must be a multiple of the 27-sample tile (at most 32 tiles); the reference uses 513");
    n_tiles_ = B_ / TILE;
    b_pad_ = (B_ + 15) / 16 * 16;
    if (desc_.recurrence_form != PBSO_FORM_BLOCK && desc_.recurrence_form != PBSO_FORM_BLOCK_BF16 && desc_.recurrence_form != PBSO_FORM_VELOCITY &&
        desc_.recurrence_form != PBSO_FORM_DIRECT)
        return fail(PBSO_ERR_INVALID, "recurrence_form");
    form_ = desc_.recurrence_form;
    // the block form tiles a buffer as 1 + 2 * 16 * 16 samples (the reference's 513); other lengths step per sample
    if (is_block() && B_ != 1 + 2 * BLOCK_J * BLOCK_N) form_ = PBSO_FORM_VELOCITY;
    // Launches that are mostly dense-profile buffers (sustained scraping): the f32 block kernel runs them in block form
    // itself (forced block path); the split-bf16 build has no such path and hands them to the per-sample kernel K1.
    // PBSO_DENSE_LAUNCHES=block|sample pins either; PBSO_FORCED_BLOCK=0 makes the block kernel step dense buffers per sample.
    dense_to_k1_ = form_ == PBSO_FORM_BLOCK_BF16;
    forced_block_ = desc_.forced_block >= 0;
    if (!forced_block_) dense_to_k1_ = true;
    if (desc_.dense_launches < 0 || desc_.dense_launches > 2) return fail(PBSO_ERR_INVALID, "dense_launches");
    if (desc_.dense_launches) dense_to_k1_ = desc_.dense_launches == 2;
    if (desc_.bank_kernel < PBSO_BANK_AUTO || desc_.bank_kernel > PBSO_BANK_PIPE) return fail(PBSO_ERR_INVALID, "bank_kernel");
    if (desc_.profile_kernel < 0 || desc_.profile_kernel > 2) return fail(PBSO_ERR_INVALID, "profile_kernel");
    if (desc_.pipe_consumers < 0 || desc_.pipe_consumers > 4) return fail(PBSO_ERR_INVALID, "pipe_consumers");
    if (desc_.profile_priority < 0 || desc_.profile_priority > 4) return fail(PBSO_ERR_INVALID, "profile_priority");
    if (desc_.stream_sync < 0 || desc_.stream_sync > 4) return fail(PBSO_ERR_INVALID, "stream_sync");
    latency_path_ = desc_.latency_path >= 0;
    fuse_short_ = desc_.fuse_short_launches >= 0;
    if (desc_.submit_thread > 0 && desc_.stream_sync == 4)
        return fail(PBSO_ERR_INVALID, "submit_thread with stream_sync = 4: the host-side gate would make the caller wait for the worker's launches");
    if (desc_.qnorm_mode < PBSO_QNORM_OFF || desc_.qnorm_mode > PBSO_QNORM_CLOSED)
        return fail(PBSO_ERR_INVALID, "qnorm_mode");
    {
        const int r = desc_.modes_per_lane;
        if (r != 0 && r != 1 && r != 2 && r != 3 && r != 4 && r != 8) return fail(PBSO_ERR_INVALID, "modes_per_lane must be 0,1,2,3,4,8");
    }
    int ndev = 0;
    HIPTRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(PBSO_ERR_HIP, "no HIP device: this engine has no CPU fallback");
    if (desc_.device < 0 || desc_.device >= ndev) return fail(PBSO_ERR_INVALID, "device ordinal out of range");
    HIPTRY(hipSetDevice(desc_.device));
    {
        // sizes of the chip (or of the partition this ordinal is: a CPX / NPS slice has fewer CUs)
        hipDeviceProp_t prop;
        HIPTRY(hipGetDeviceProperties(&prop, desc_.device));
        n_cus_ = std::max(1, prop.multiProcessorCount);
    }
    if (desc_.stream) {
        stream_ = (hipStream_t)desc_.stream;
    } else {
        HIPTRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        own_stream_ = true;
    }
    for (int i = 0; i < N_SETS; ++i) HIPTRY(hipEventCreateWithFlags(&ev_set_[i], hipEventDisableTiming));
    // preparation of step k+1 (plan upload, projection, FFAT lookup, force combination)
    // runs on its own stream beside the oscillator bank of step k
    // High priority: its kernels are small, the oscillator bank waits for them, and the runtime maps
    // streams of different priority to different hardware queues -- with equal priority the two
    // engine streams can land on ONE queue when the process owns many streams (torch + RCCL under
    // torch.distributed.run did exactly that: no overlap, +12 % per step).
    {
        int least = 0, greatest = 0;
        HIPTRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPTRY(hipStreamCreateWithPriority(&prep_stream_, hipStreamNonBlocking, greatest));
        prep_priority_ = greatest;
    }
    // A start gate that waits ON THE DEVICE is a kernel spinning on a memory word another kernel stores when it starts.  Anything that
    // lets only one kernel run at a time can dispatch the waiting kernel first -- it then never ends (round 6: every rocprofv3 --pmc
    // pass of the headline launch hung until its time limit with the gate and finished in 11 s without; profiles/NOTES.md).  Under the
    // policy (stream_sync = 0) such an environment gets no gate; 2 / 3 stay the caller's explicit choice.
    const char *serial_env = serialising_environment();
    if (desc_.stream_sync == 0 && serial_env) {
        gate_choice_ = -1;
        if (std::getenv("PBSO_TIMELINE")) std::fprintf(stderr, "openpbso_amd: no start gate: %s serialises kernel dispatches\n", serial_env);
    }
    if (desc_.stream_sync != 1 && gate_choice_ != -1) {
        // the HOST form of the start gate (stream_sync = 4, an option: see step_chunk): a word of pinned host memory the bank's first
        // workgroup writes and the submitting thread reads -- no waiting kernel on the device
        void *dptr = nullptr;
        if (hipHostMalloc((void **)&host_start_, sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&dptr, host_start_, 0) == hipSuccess) {
            *host_start_ = 0;
            host_start_dev_ = static_cast<unsigned long long *>(dptr);
        } else {
            if (host_start_) (void)hipHostFree(host_start_);
            host_start_ = nullptr;
            (void)hipGetLastError();
        }
        int can = 0;
        auto signal_word = [&](unsigned long long **p) {
            if (hipExtMallocWithFlags((void **)p, sizeof(unsigned long long), hipMallocSignalMemory) != hipSuccess) { *p = nullptr; return false; }
            return hipMemset(*p, 0, sizeof(unsigned long long)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        };
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, desc_.device) == hipSuccess && can) {
            start_gate_ = signal_word(&sig_start_);
            if (desc_.stream_sync == 2 || desc_.stream_sync == 0) sync_values_ = signal_word(&sig_prep_);      // (0: for short launches, step_chunk)
        }
        (void)hipGetLastError();
        if ((desc_.stream_sync == 2 || desc_.stream_sync == 3) && !(desc_.stream_sync == 2 ? sync_values_ : start_gate_))
            return fail(PBSO_ERR_HIP, "stream_sync = 2 / 3: the device has no hipStreamWaitValue64");
        if (desc_.stream_sync == 4 && !host_start_)
            return fail(PBSO_ERR_HIP, "stream_sync = 4: no pinned host memory for the gate's word");
        gate_choice_ = desc_.stream_sync == 4 ? 2 : start_gate_ ? 1 : 0;
    }
    for (int i = 0; i < N_SETS; ++i) {
        // (waited for by the engine's own streams only, never by the host or another device: without the system-scope fence
        //  a record costs the stream nothing -- 4 us with it, scripts/microbench/wait_value.hip; +0.4 % per step at 1024 x 512)
#ifndef PBSO_DEVICE_EVENT_FLAGS
#define PBSO_DEVICE_EVENT_FLAGS (hipEventDisableTiming | hipEventDisableSystemFence)
#endif
        HIPTRY(hipEventCreateWithFlags(&ev_prep_done_[i], PBSO_DEVICE_EVENT_FLAGS));
        HIPTRY(hipEventCreateWithFlags(&ev_k1_done_[i], PBSO_DEVICE_EVENT_FLAGS));
        HIPTRY(hipEventCreateWithFlags(&ev_aux_fork_[i], PBSO_DEVICE_EVENT_FLAGS));
        HIPTRY(hipEventCreateWithFlags(&ev_aux_join_[i], PBSO_DEVICE_EVENT_FLAGS));
    }
    plan_threads_ = std::min(16, std::max(1, desc_.plan_threads));
    ctx_.resize(plan_threads_);
    for (PlanCtx &c : ctx_) c.tbuf.assign(B_, 0.0);
    // which build of the oscillator bank to launch (see kernels_iir.hip)
    if (const char *v = std::getenv("PBSO_CENSUS")) census_ = std::atoi(v) != 0;
    if (const char *v = std::getenv("PBSO_ROTATE_PRIO")) rotate_prio_ = std::min(2, std::max(0, std::atoi(v)));
    if (const char *v = std::getenv("PBSO_TIMELINE")) timeline_ = std::atoi(v) != 0;
    if (const char *v = std::getenv("PBSO_PREP_SPLIT")) prep_split_ = std::min(2, std::max(0, std::atoi(v)));
    host_profile_ = std::getenv("PBSO_HOST_PROFILE") != nullptr;
    // which kernels run and how: per engine, from the descriptor (ABI 4); the environment only switches diagnostics on
    device_profiles_ = desc_.device_profiles >= 0;
    k2_rows_ = desc_.profile_kernel == 0;
    ar_serial_ = desc_.profile_kernel == 2;
    if (desc_.profile_margin_pct > 0) k2_margin_pct_ = std::min(400, desc_.profile_margin_pct);
    if (desc_.profile_priority > 0) { k2_prio_ = desc_.profile_priority - 1; k2_prio_auto_ = false; }
    direct_hits_ = desc_.direct_hits >= 0;
    timing_every_ = desc_.timing_every < 0 ? 0 : std::max(1, desc_.timing_every);
    if (desc_.chunk_buffers > 0) chunk_buffers_ = desc_.chunk_buffers;
    tc_mode_ = desc_.time_chunks;
    tc_shape_ = desc_.time_chunk_shape;
    if (desc_.scan_kernel < 0 || desc_.scan_kernel > 2) return fail(PBSO_ERR_INVALID, "scan_kernel");
    return PBSO_OK;
}

// BuildSolver -> ModalIntegrator<double>::Build + ctor, modal_integrator.h:47-101
int Engine::add_object(const pbso_object_desc &d, int *id) {
    if (finalized_) return fail(PBSO_ERR_STATE, "add_object after finalize");
    if (d.n_modes < 0 || d.n_omega < d.n_modes || (d.n_modes > 0 && !d.omega_squared))
        return fail(PBSO_ERR_INVALID, "N for modal integrator invalid");         // assert :56
    Object o;
    o.n_modes = d.n_modes;
    o.c1.resize(d.n_modes);
    o.c2.resize(d.n_modes);
    o.c3.resize(d.n_modes);
    const double h = 1.0 / (double)rate_;
    for (int ii = 0; ii < d.n_modes; ++ii) {
        const double omega0 = std::sqrt(d.omega_squared[ii] / d.density);         // :63
        const double xi = 0.5 * (d.alpha / omega0 + d.beta * omega0);             // :64
        const double a = 2.0 * xi * omega0;                                       // :65
        const double b = std::pow(omega0, 2);                                     // :66
        const double epsilon = std::exp(-a / 2 * h);                              // :89
        const double theta = h * std::sqrt(b - a * a / 4.0);                      // :90
        const double gamma = std::asin(a / (2.0 * std::sqrt(b)));                 // :91
        const double omega = std::sqrt(b);                                        // :92
        const double omega_d = std::sqrt(b - std::pow(a, 2) / 4.0);               // :93
        o.c1[ii] = 2.0 * epsilon * std::cos(theta);                               // :95
        o.c2[ii] = -std::pow(epsilon, 2);                                         // :96
        double c3 = 2.0 * (epsilon * std::cos(theta + gamma)
                           - std::pow(epsilon, 2) * std::cos(2.0 * theta + gamma));  // :97
        c3 /= (3.0 * omega * omega_d);                                            // :98
        c3 *= 1E9;                                                                // :99
        o.c3[ii] = c3;
    }
    if (d.mode_shapes && d.n_dof > 0) {
        o.n_dof = d.n_dof;
        o.shapes.assign(d.mode_shapes, d.mode_shapes + (size_t)d.n_modes * d.n_dof);
    }
    objs_.push_back(std::move(o));
    if (id) *id = (int)objs_.size() - 1;
    return PBSO_OK;
}

// ModalSolver::readFFATMaps, modal_solver.h:278-284 (std::map keyed by modeId)
int Engine::set_ffat_maps(int obj, const pbso_ffat_map *maps, int n) {
    if (finalized_) return fail(PBSO_ERR_STATE, "set_ffat_maps after finalize");
    if (!valid_obj(obj) || n < 0 || (n > 0 && !maps)) return fail(PBSO_ERR_INVALID, "set_ffat_maps arguments");
    Object &o = objs_[obj];
    o.have_maps = true;               // LoadAll returns a non-null (possibly empty) map, SURVEY Q12
    o.geom.clear();
    o.psi.clear();
    o.n_maps = 0;
    for (int i = 0; i < n; ++i) {
        const pbso_ffat_map &m = maps[i];
        if (m.mode_id < 0) return fail(PBSO_ERR_INVALID, "negative modeId");
        if ((int)o.geom.size() <= m.mode_id) {
            FfatGeom z;
            std::memset(&z, 0, sizeof(z));
            o.geom.resize(m.mode_id + 1, z);
        }
        FfatGeom &g = o.geom[m.mode_id];
        if (!g.valid) o.n_maps++;
        // every index GetMapVal can form must be inside psi
        for (int f = 0; f < 6; ++f) {
            if (m.n_elements[f][0] < 1 || m.n_elements[f][1] < 1 || m.strides[f] < 0 ||
                (long long)m.strides[f] + (long long)m.n_elements[f][0] * m.n_elements[f][1] > m.n_psi)
                return fail(PBSO_ERR_INVALID, "FFAT map strides/n_elements exceed psi");
        }
        g.k = m.k;
        g.cell_size = m.cell_size;
        for (int j = 0; j < 3; ++j) {
            g.center3[j] = m.center3[j];
            g.center[j] = m.center[j];
            g.bbox_low[j] = m.bbox_low[j];
            g.bbox_top[j] = m.bbox_top[j];
        }
        for (int f = 0; f < 6; ++f) {
            for (int j = 0; j < 3; ++j) g.low_corners[f][j] = m.low_corners[f][j];
            g.n_elements[f][0] = m.n_elements[f][0];
            g.n_elements[f][1] = m.n_elements[f][1];
            g.strides[f] = m.strides[f];
        }
        g.n_psi = m.n_psi;
        g.valid = 1;
        g.psi_off = (long long)o.psi.size();       // a replaced map leaves its old psi unused
        o.psi.insert(o.psi.end(), m.psi, m.psi + m.n_psi);
    }
    return PBSO_OK;
}

// Closed-form qnorm matrices (getQBufferNorm, modal_solver.h:270-273): G = sum_{k=0}^{B-1} (A^k)' e1 e1' A^k per mode,
// in fp64, in the basis x = (q_k, q_k - q_{k-1}) (well conditioned at low frequency; the direct-form kernel forms the
// difference itself): A = [[1-e, eps^2], [-e, eps^2]], e = 1 - c1 - c2, eps^2 = -c2.  Row r_k = e1' A^k: r_{k+1} = r_k A.
// Built by finalize() for the engines that report qnorm rows in closed form, and by listeners_enable() for a block
// engine created with PBSO_QNORM_OFF (the build of the bank that keeps block-start states always evaluates the rows).
int Engine::build_gq() {
    if (d_gq_.p) return PBSO_OK;
    const int N = (int)objs_.size();
    const size_t nm = (size_t)N * m_pad_;
    std::vector<float> gq(3 * nm, 0.f);
    for (int i = 0; i < N; ++i) {
        const Object &o = objs_[i];
        for (int m = 0; m < o.n_modes; ++m) {
            const double eps2 = -o.c2[m], e = (1.0 - o.c1[m]) - o.c2[m];
            const double a00 = 1.0 - e, a01 = eps2, a10 = -e, a11 = eps2;
            double r0 = 1.0, r1 = 0.0, s11 = 0.0, s12 = 0.0, s22 = 0.0;
            for (int k = 0; k < B_; ++k) {
                s11 += r0 * r0; s12 += r0 * r1; s22 += r1 * r1;
                const double n0 = r0 * a00 + r1 * a10, n1 = r0 * a01 + r1 * a11;
                r0 = n0; r1 = n1;
            }
            const size_t k = (size_t)i * m_pad_ + m;
            gq[k] = (float)s11;
            gq[nm + k] = (float)(2.0 * s12);
            gq[2 * nm + k] = (float)s22;
        }
    }
    HIPTRY(d_gq_.ensure(3 * nm));
    HIPTRY(hipMemcpy(d_gq_.p, gq.data(), 3 * nm * sizeof(float), hipMemcpyHostToDevice));
    return PBSO_OK;
}

int Engine::finalize() {
    HIPTRY(hipSetDevice(desc_.device));      // the caller's thread may have another device current
    if (finalized_) return fail(PBSO_ERR_STATE, "finalize called twice");
    if (objs_.empty()) return fail(PBSO_ERR_STATE, "no objects");
    const int N = (int)objs_.size();
    {
        // The second preparation stream (round 5, step_chunk's fork), for engines large enough for the fork to matter.  Created
        // HERE, not with the engine and not at the first fork: every stream a process owns shifts which hardware queue the next
        // one lands on -- created with every engine it doubled the cross-stream hand-over of one-buffer steps of small engines
        // (58 -> 112 us, scripts/latency.py, latency_path = -1: any third stream does, whatever its priority class), created at
        // the first fork it came to share a queue with the bank (8 x 4096 x 86 scraping: 0.30 -> 0.40 ms per step;
        // scripts/debug/r05_aux_queue.sh).  Large OBJECTS: that is where projection and combine are long enough to be worth a
        // stream (64 x 256 with a moving listener, host-bound, 0.13 - 0.15 ms per step without the third stream and 0.15 - 0.19
        // with it).  PBSO_PREP_SPLIT=2 creates it for any engine (tests), 0 never.
        long long modes = 0;
        for (const Object &o : objs_) modes += o.n_modes;
        if (prep_split_ == 2 || (prep_split_ == 1 && modes >= 32768 && modes >= 1024LL * N))
            HIPTRY(hipStreamCreateWithPriority(&aux_stream_, hipStreamNonBlocking, prep_priority_));
    }
    // Team shape: R oscillators per lane; an object of n modes needs ceil(n / 64R) waves, cut into
    // teams (workgroups) of at most MAX_WAVES_PER_TEAM waves.  The VALU issue rate needs ~4 waves per
    // SIMD (4096 on the chip, profiles/r01_microbench.txt).
    int R = desc_.modes_per_lane;
    if (R != 0 && R != 1 && R != 2 && R != 3 && R != 4 && R != 8) return fail(PBSO_ERR_INVALID, "modes_per_lane must be 0,1,2,3,4,8");
    const bool block = is_block();
    // (round 6: the block form's eight-modes-per-lane builds -- one wave per SIMD, 300 - 460 bytes of scratch per lane -- were never a
    //  policy's choice and are gone)
    if (block && (R == 3 || R == 8)) return fail(PBSO_ERR_INVALID, "modes_per_lane must be 0,1,2,4 in the block form");

    auto waves_of = [&](const Object &o, int r) { return std::max(1, (o.n_modes + 64 * r - 1) / (64 * r)); };
    auto total_waves = [&](int r) {
        long long w = 0;
        for (const Object &o : objs_) w += waves_of(o, r);
        return w;
    };
    if (R == 0 && block) {
        // The block form is paced by the matrix pipe: two waves per SIMD keep it busy (one issues MFMAs while
        // the other steps its coarse recurrence), and a wave holds 32 R operand registers for its W table.
        // The fewest modes per lane that fit the chip at two waves per SIMD; more objects run in rounds.
        R = 4;
        for (int r : {1, 2, 4})
            if (total_waves(r) <= 8LL * n_cus_) { R = r; break; }
    }
    if (R == 0) {
        // the fewest modes per lane whose waves are all resident at once (16 waves per CU is what the
        // LDS tiles allow: 4096 on the chip) -- a second round of workgroups costs more than the
        // deeper per-lane work (768 x 512: 2.83 ms with R = 1 in two rounds, 1.93 ms with R = 2).
        // Engines that cannot be resident at once with R <= 4 minimise rounds x instructions per
        // wave-sample (5 R + 2.2); R = 8 needs more registers than 4 waves per SIMD leave.
        R = 0;
        for (int r : {1, 2, 3, 4})
            if (total_waves(r) <= 16LL * n_cus_) { R = r; break; }
        if (R == 0) {
            double best = 0;
            for (int r : {1, 2, 3, 4}) {
                const double cost = (double)((total_waves(r) + 16LL * n_cus_ - 1) / (16LL * n_cus_)) * (5.0 * r + 2.2);
                if (R == 0 || cost < best) { R = r; best = cost; }
            }
        }
    }
    int wmax = 1;
    for (const Object &o : objs_) wmax = std::max(wmax, waves_of(o, R));
    R_ = R;
    m_pad_ = 64 * R * wmax;
    // K5 (kernels_scan.hip): launches cut along the time axis pick their team shape per launch, up to four modes per lane: rows
    // padded to whole 256-column waves of that shape (padding columns are dead: zero coefficients, zero state)
    tc_ok_ = block && tc_mode_ >= 0;
    if (tc_ok_) m_pad_ = (m_pad_ + 255) / 256 * 256;
    // teams, grouped into size classes (one launch of the oscillator bank per team size, largest
    // first); the SoA rows stay m_pad wide and a team touches its own columns only
    {
        // Team size.  A full chip (>= 4096 waves) runs whole objects as teams of up to 16 waves.  Below
        // that the launch is bound by the per-sample latency of a wave, which is shortest when the
        // wave shares its SIMD / CU with as few others as possible (measured: 8 x 4096 modes 425 -> 570 x
        // real time with 2-wave teams, 1 x 512 modes 778 -> 822 x with 1-wave teams): spread the waves
        // evenly over the 256 CUs.
        const int team_max = block ? MAX_WAVES_PER_BLOCK_TEAM : MAX_WAVES_PER_TEAM;
        int team_cap = (int)std::min<long long>(team_max, std::max<long long>(1, (total_waves(R) + n_cus_ - 1) / n_cus_));
        if (desc_.team_waves > 0) team_cap = std::min(team_max, desc_.team_waves);
        std::vector<std::vector<TeamDesc>> by_w(MAX_WAVES_PER_TEAM + 1);
        std::vector<SplitObj> split;
        n_part_rows_ = 0;
        W_ = 1;
        for (int i = 0; i < N; ++i) {
            const int w = waves_of(objs_[i], R);
            const int parts = (w + team_cap - 1) / team_cap;
            const int base = w / parts, rem = w % parts;
            int w0 = 0;
            if (parts > 1) {
                SplitObj so = {i, n_part_rows_, parts, 0};
                split.push_back(so);
            }
            for (int pi = 0; pi < parts; ++pi) {
                const int wp = base + (pi < rem ? 1 : 0);
                TeamDesc td = {i, 64 * R * w0, parts > 1 ? n_part_rows_++ : -1, 0};
                by_w[wp].push_back(td);
                W_ = std::max(W_, wp);
                w0 += wp;
            }
        }
        classes_.clear();
        std::vector<TeamDesc> flat;
        for (int w = MAX_WAVES_PER_TEAM; w >= 1; --w) {
            if (by_w[w].empty()) continue;
            SizeClass c;
            c.W = w;
            c.first = (int)flat.size();
            c.count = (int)by_w[w].size();
            flat.insert(flat.end(), by_w[w].begin(), by_w[w].end());
            classes_.push_back(c);
        }
        for (size_t k = 0; k < flat.size(); ++k) flat[k].id = (int)k;
        n_teams_ = (int)flat.size();
        total_team_waves_ = total_waves(R);
        if (classes_.size() > 1 && !ev_fork_) {
            HIPTRY(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
            for (int i = 0; i < N_CLASS_STREAMS; ++i) {
                HIPTRY(hipStreamCreateWithFlags(&class_stream_[i], hipStreamNonBlocking));
                HIPTRY(hipEventCreateWithFlags(&ev_join_[i], hipEventDisableTiming));
            }
        }
        n_split_ = (int)split.size();
        HIPTRY(d_teams_.ensure(flat.size()));
        HIPTRY(hipMemcpy(d_teams_.p, flat.data(), flat.size() * sizeof(TeamDesc), hipMemcpyHostToDevice));
        if (n_split_) {
            HIPTRY(d_split_.ensure(split.size()));
            HIPTRY(hipMemcpy(d_split_.p, split.data(), split.size() * sizeof(SplitObj), hipMemcpyHostToDevice));
        }
    }

    if (tc_ok_) {
        // team tables of the time-chunked launches, one per shape: whole objects as teams of up to 8 waves (an object that
        // needs more is cut into several teams whose partial sums sum_parts adds, as above)
        const int shapes[3] = {1, 2, 4};
        for (int k = 0; k < 3; ++k) {
            TcSet &ts = tc_[k];
            ts.R = shapes[k];
            std::vector<std::vector<TeamDesc>> by_w(MAX_WAVES_PER_BLOCK_TEAM + 1);
            std::vector<SplitObj> split;
            ts.n_part_rows = 0;
            ts.waves = 0;
            ts.cover = 0;
            const int tcap_desc = desc_.team_waves > 0 ? std::min(MAX_WAVES_PER_BLOCK_TEAM, desc_.team_waves) : MAX_WAVES_PER_BLOCK_TEAM;
            int wmax_set = 1;
            for (int i = 0; i < N; ++i) {
                const int w = waves_of(objs_[i], ts.R);
                // One mode per lane: an object that needs several teams anyway gets teams of FOUR waves -- three such workgroups
                // fit a CU's LDS (twelve waves, three per SIMD; one team of eight leaves it at two per SIMD), and these builds are
                // bound by what a wave does between its matrix bursts, not by the matrix pipe (8 x 4096 sustained scraping).
                const int tcap = (ts.R == 1 && w > MAX_WAVES_PER_BLOCK_TEAM && desc_.team_waves <= 0) ? 4 : tcap_desc;
                wmax_set = std::max(wmax_set, std::min(w, tcap));
                const int parts = (w + tcap - 1) / tcap;
                const int base = w / parts, rem = w % parts;
                int w0 = 0;
                if (parts > 1) {
                    SplitObj so = {i, ts.n_part_rows, parts, 0};
                    split.push_back(so);
                }
                for (int pi = 0; pi < parts; ++pi) {
                    const int wp = base + (pi < rem ? 1 : 0);
                    TeamDesc td = {i, 64 * ts.R * w0, parts > 1 ? ts.n_part_rows++ : -1, 0};
                    by_w[wp].push_back(td);
                    w0 += wp;
                }
                ts.waves += w;
                ts.cover += (long long)w * 64 * ts.R;
            }
            ts.classes.clear();
            std::vector<TeamDesc> flat;
            for (int w = MAX_WAVES_PER_BLOCK_TEAM; w >= 1; --w) {
                if (by_w[w].empty()) continue;
                SizeClass c;
                c.W = w;
                c.first = (int)flat.size();
                c.count = (int)by_w[w].size();
                flat.insert(flat.end(), by_w[w].begin(), by_w[w].end());
                ts.classes.push_back(c);
            }
            for (size_t j = 0; j < flat.size(); ++j) flat[j].id = (int)j;
            ts.n_teams = (int)flat.size();
            ts.n_split = (int)split.size();
            {
                // waves of this shape a CU holds at once: two 256-register waves per SIMD for two and four modes per lane; one mode
                // per lane: three per SIMD by its registers, as many whole workgroups as the LDS takes (allocated in 1280-byte granules)
                ts.waves_per_cu = 8;
                if (ts.R == 1) {
                    const size_t lds = (block_lds_bytes(wmax_set, 1) + 1279) / 1280 * 1280;
                    ts.waves_per_cu = (int)std::max<size_t>(wmax_set, std::min<size_t>(12, (160 * 1024 / lds) * wmax_set));
                }
            }
            HIPTRY(ts.d_teams.ensure(flat.size()));
            HIPTRY(hipMemcpy(ts.d_teams.p, flat.data(), flat.size() * sizeof(TeamDesc), hipMemcpyHostToDevice));
            if (ts.n_split) {
                HIPTRY(ts.d_split.ensure(split.size()));
                HIPTRY(hipMemcpy(ts.d_split.p, split.data(), split.size() * sizeof(SplitObj), hipMemcpyHostToDevice));
            }
        }
    }

    // K1p: fewer than one wave of oscillators per SIMD even with one mode per lane -- a team of several waves per 64
    // modes instead (kernels_pipe.hip: a producer and two consumers).  f32 block form only; desc.bank_kernel = BLOCK keeps
    // the one-wave-per-64-modes kernel (and its time chunks, K5) for every launch.
    {
        split_ok_ = false;
        long long chunks = 0;
        for (const Object &o : objs_) chunks += std::max(1, (o.n_modes + 63) / 64);
        const long long max_chunks = desc_.pipe_max_teams > 0 ? desc_.pipe_max_teams : 2LL * n_cus_;
        if (block && form_ == PBSO_FORM_BLOCK && R_ == 1 && chunks <= max_chunks && desc_.bank_kernel != PBSO_BANK_BLOCK) {
            std::vector<TeamDesc> ts;
            std::vector<SplitObj> tsplit;
            n_ts_part_rows_ = 0;
            for (int i = 0; i < N; ++i) {
                const int parts = std::max(1, (objs_[i].n_modes + 63) / 64);
                if (parts > 1) {
                    SplitObj so = {i, n_ts_part_rows_, parts, 0};
                    tsplit.push_back(so);
                }
                for (int c = 0; c < parts; ++c) {
                    TeamDesc td = {i, 64 * c, parts > 1 ? n_ts_part_rows_++ : -1, (int)ts.size()};
                    ts.push_back(td);
                }
            }
            n_ts_teams_ = (int)ts.size();
            n_ts_split_ = (int)tsplit.size();
            HIPTRY(d_ts_teams_.ensure(ts.size()));
            HIPTRY(hipMemcpy(d_ts_teams_.p, ts.data(), ts.size() * sizeof(TeamDesc), hipMemcpyHostToDevice));
            if (n_ts_split_) {
                HIPTRY(d_ts_split_.ensure(tsplit.size()));
                HIPTRY(hipMemcpy(d_ts_split_.p, tsplit.data(), tsplit.size() * sizeof(SplitObj), hipMemcpyHostToDevice));
            }
            split_ok_ = true;
            split_always_ = desc_.bank_kernel == PBSO_BANK_PIPE;
        }
    }

    const size_t nm = (size_t)N * m_pad_;
    std::vector<float> ca(nm, 0.f), cb(nm, 0.f);
    std::vector<double> c3(nm, 0.0);
    std::vector<int> nmodes(N);
    for (int i = 0; i < N; ++i) {
        const Object &o = objs_[i];
        nmodes[i] = o.n_modes;
        for (int m = 0; m < o.n_modes; ++m) {
            const size_t k = (size_t)i * m_pad_ + m;
            if (form_ != PBSO_FORM_DIRECT) {
                ca[k] = (float)(-o.c2[m]);                     // eps^2
                cb[k] = -(float)((1.0 - o.c1[m]) - o.c2[m]);   // -e, e = |1 - z|^2 = 1 - c1 - c2 (stored negated:
                                                               //  d = eps^2 d + (-e) q is then a plain v_fmac)
            } else {
                ca[k] = (float)o.c1[m];
                cb[k] = (float)o.c2[m];
            }
            c3[k] = o.c3[m];
        }
    }
    HIPTRY(d_ca_.ensure(nm));
    HIPTRY(d_cb_.ensure(nm));
    HIPTRY(d_sq_.ensure(nm));
    HIPTRY(d_sd_.ensure(nm));
    HIPTRY(d_ss_.ensure(nm));
    HIPTRY(d_c3_.ensure(nm));
    HIPTRY(d_n_modes_.ensure(N));
    HIPTRY(hipMemcpy(d_ca_.p, ca.data(), nm * sizeof(float), hipMemcpyHostToDevice));
    HIPTRY(hipMemcpy(d_cb_.p, cb.data(), nm * sizeof(float), hipMemcpyHostToDevice));
    HIPTRY(hipMemset(d_sq_.p, 0, nm * sizeof(float)));
    HIPTRY(hipMemset(d_sd_.p, 0, nm * sizeof(float)));
    HIPTRY(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(d_ss_.p), 0x3F800000, nm));      // scale 1.0f: state stored unscaled
    HIPTRY(hipMemcpy(d_c3_.p, c3.data(), nm * sizeof(double), hipMemcpyHostToDevice));
    if (desc_.qnorm_mode == PBSO_QNORM_CLOSED || (block && desc_.qnorm_mode != PBSO_QNORM_OFF)) {
        int rc = build_gq();
        if (rc != PBSO_OK) return rc;
    }

    if (block) {
        // Block form: per mode the one-sample matrix in the basis x = (q, q - q_prev) is
        // A = [[1 - e, eps^2], [-e, eps^2]] (e = 1 - c1 - c2, eps^2 = -c2).  Row 0 of A^j, j = 1..16, gives the
        // output weights (a_j, b_j); P = A^16 advances the state by one block.  All in fp64, rounded once.
        // P11 is stored as P11 - 1 (q' = q + (P11 - 1) q + P12 d keeps the small angle of low modes).
        std::vector<float> pc(4 * nm, 0.f), wt((size_t)nm / 2 * 64, 0.f);
        for (int i = 0; i < N; ++i) {
            const Object &o = objs_[i];
            for (int m = 0; m < o.n_modes; ++m) {
                const double eps2 = -o.c2[m], e = (1.0 - o.c1[m]) - o.c2[m];
                const double a00 = 1.0 - e, a01 = eps2, a10 = -e, a11 = eps2;
                double p00 = 1, p01 = 0, p10 = 0, p11 = 1;                 // A^j, starting from I
                const size_t k = (size_t)i * m_pad_ + m;
                float *w = wt.data() + (k / 2) * 64 + 16 * (2 * (k & 1));  // lane = 16 * (2 * mode-of-pair + comp) + (j - 1)
                for (int j = 1; j <= BLOCK_J; ++j) {
                    const double n00 = a00 * p00 + a01 * p10, n01 = a00 * p01 + a01 * p11;
                    const double n10 = a10 * p00 + a11 * p10, n11 = a10 * p01 + a11 * p11;
                    p00 = n00; p01 = n01; p10 = n10; p11 = n11;
                    w[j - 1] = (float)p00;
                    w[16 + j - 1] = (float)p01;
                }
                pc[k] = (float)(p00 - 1.0);
                pc[nm + k] = (float)p01;
                pc[2 * nm + k] = (float)p10;
                pc[3 * nm + k] = (float)p11;
            }
        }
        if (form_ == PBSO_FORM_BLOCK_BF16) {
            // Split-bf16 operand table: per group of 16 columns 8 rows of 64 dwords, rows 0..3 the hi parts, 4..7 the
            // lo parts; lane l = 16 kq + (j - 1), dword dd <-> mode 16 G + 4 kq + dd: low half a_j, high half b_j (the k
            // order of v_mfma_f32_16x16x32_bf16: k = 8 kq + 2 dd + comp).  hi = bf16(x) round-to-nearest-even,
            // lo = bf16(x - hi).
            auto bf16_rne = [](float x) -> uint32_t {
                uint32_t u;
                std::memcpy(&u, &x, 4);
                if ((u & 0x7F800000u) == 0x7F800000u) return u >> 16;                 // inf / nan
                return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
            };
            auto split = [&](double x, uint32_t &hi, uint32_t &lo) {
                const float xf = (float)x;
                hi = bf16_rne(xf);
                const uint32_t hb = hi << 16;
                float hf;
                std::memcpy(&hf, &hb, 4);
                lo = bf16_rne(xf - hf);
            };
            std::vector<float> wf(wt);                                               // the f32 table just built
            std::vector<uint32_t> w16(wt.size(), 0u);
            for (size_t G = 0; G < nm / 16; ++G)
                for (int l = 0; l < 64; ++l) {
                    const int j = l & 15, kq = l >> 4;
                    for (int dd = 0; dd < 4; ++dd) {
                        const size_t col = 16 * G + 4 * kq + dd;                      // flat column index (object * m_pad + column)
                        // f32 table: row col / 2, lane 16 * (2 * (col & 1) + comp) + j
                        const float a = wf[(col / 2) * 64 + 16 * (2 * (col & 1) + 0) + j];
                        const float b = wf[(col / 2) * 64 + 16 * (2 * (col & 1) + 1) + j];
                        uint32_t ah, al, bh, bl;
                        split((double)a * TRUNC_SPLIT_GAIN, ah, al);
                        split((double)b * TRUNC_SPLIT_GAIN, bh, bl);
                        w16[(8 * G + dd) * 64 + l] = ah | (bh << 16);
                        w16[(8 * G + 4 + dd) * 64 + l] = al | (bl << 16);
                    }
                }
            std::memcpy(wt.data(), w16.data(), wt.size() * sizeof(float));
        }
        HIPTRY(d_pc_.ensure(4 * nm));
        HIPTRY(hipMemcpy(d_pc_.p, pc.data(), 4 * nm * sizeof(float), hipMemcpyHostToDevice));
        if (tc_ok_) {
            // K5: a whole buffer as one step of the state -- A^B (first entry minus one, as P) and A^(B-1) u, u = (1, 1)': what a
            // force sample at the buffer's first sample leaves at its end.  fp64, rounded once.
            std::vector<float> sc(6 * nm, 0.f);
            for (int i = 0; i < N; ++i) {
                const Object &o = objs_[i];
                for (int m = 0; m < o.n_modes; ++m) {
                    const double eps2 = -o.c2[m], e = (1.0 - o.c1[m]) - o.c2[m];
                    const double a00 = 1.0 - e, a01 = eps2, a10 = -e, a11 = eps2;
                    double p00 = 1, p01 = 0, p10 = 0, p11 = 1;
                    const size_t k = (size_t)i * m_pad_ + m;
                    for (int j = 1; j <= B_; ++j) {
                        if (j == B_) {                                   // A^(B-1) u
                            sc[4 * nm + k] = (float)(p00 + p01);
                            sc[5 * nm + k] = (float)(p10 + p11);
                        }
                        const double n00 = a00 * p00 + a01 * p10, n01 = a00 * p01 + a01 * p11;
                        const double n10 = a10 * p00 + a11 * p10, n11 = a10 * p01 + a11 * p11;
                        p00 = n00; p01 = n01; p10 = n10; p11 = n11;
                    }
                    sc[k] = (float)(p00 - 1.0);
                    sc[nm + k] = (float)p01;
                    sc[2 * nm + k] = (float)p10;
                    sc[3 * nm + k] = (float)p11;
                }
            }
            HIPTRY(d_scan_.ensure(6 * nm));
            HIPTRY(hipMemcpy(d_scan_.p, sc.data(), 6 * nm * sizeof(float), hipMemcpyHostToDevice));
        }
        // (a time-chunked launch picks its own team shape: one or two modes per lane use the table; K5's dense_increment_kernel
        //  uses it whatever the projection and the bank's own path for dense buffers are)
        ftab_forced_ = form_ == PBSO_FORM_BLOCK && (R_ <= 2 || tc_ok_) && forced_block_;
        if (ftab_forced_ || tc_ok_) {
            // Forced block path without qnorm rows (kernels_block.hip, FT): a force sample f enters the state as f u, u = (1, 1)'
            // (d += f, q += d), so the samples 16 n + i, i = 1..16, of a dense profile move the next block-start state by
            // sum_i A^(16 - i) u f_i.  Plane 2 i' + c holds component c of A^(15 - i') u, i' = 0..15, per mode (fp64, rounded once).
            // Engines with more than two modes per lane keep the per-sample path (no registers left for 32 constants per mode).
            std::vector<float> ft((size_t)32 * nm, 0.f);
            for (int i = 0; i < N; ++i) {
                const Object &o = objs_[i];
                for (int m = 0; m < o.n_modes; ++m) {
                    const double eps2 = -o.c2[m], e = (1.0 - o.c1[m]) - o.c2[m];
                    double v0 = 1.0, v1 = 1.0;                              // A^delta u, delta = 0, 1, ...
                    const size_t k = (size_t)i * m_pad_ + m;
                    for (int delta = 0; delta < BLOCK_J; ++delta) {
                        const int ip = BLOCK_J - 1 - delta;                // in-block sample index (0-based) this power belongs to
                        ft[(size_t)(2 * ip) * nm + k] = (float)v0;
                        ft[(size_t)(2 * ip + 1) * nm + k] = (float)v1;
                        const double n0 = (1.0 - e) * v0 + eps2 * v1, n1 = -e * v0 + eps2 * v1;
                        v0 = n0; v1 = n1;
                    }
                }
            }
            HIPTRY(d_ftab_.ensure(ft.size()));
            HIPTRY(hipMemcpy(d_ftab_.p, ft.data(), ft.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        HIPTRY(d_wtab_.ensure(wt.size()));
        HIPTRY(hipMemcpy(d_wtab_.p, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
    }

    // mode shapes: mode-major (ModeData.h:24) -> vertex-major [dof][m_pad]
    {
        std::vector<long long> off(N, 0);
        size_t total = 0;
        for (int i = 0; i < N; ++i) {
            off[i] = (long long)total;
            total += (size_t)objs_[i].n_dof * m_pad_;
        }
        HIPTRY(d_shape_off_.ensure(N));
        HIPTRY(hipMemcpy(d_shape_off_.p, off.data(), N * sizeof(long long), hipMemcpyHostToDevice));
        std::vector<long long> row_off(N, 0);
        for (int i = 0; i < N; ++i) row_off[i] = off[i] / m_pad_;
        HIPTRY(d_g32_off_.ensure(N));
        HIPTRY(hipMemcpy(d_g32_off_.p, row_off.data(), N * sizeof(long long), hipMemcpyHostToDevice));
        HIPTRY(d_g32_.ensure(std::max<size_t>(total, 1)));
        if (total) {
            HIPTRY(d_shapes_.ensure(total));
            std::vector<double> vm;
            std::vector<float> vg;
            for (int i = 0; i < N; ++i) {
                Object &o = objs_[i];
                if (!o.n_dof) continue;
                vm.assign((size_t)o.n_dof * m_pad_, 0.0);
                vg.assign((size_t)o.n_dof * m_pad_, 0.f);
                for (int m = 0; m < o.n_modes; ++m)
                    for (int dof = 0; dof < o.n_dof; ++dof) {
                        const double u = o.shapes[(size_t)m * o.n_dof + dof];
                        vm[(size_t)dof * m_pad_ + m] = u;
                        vg[(size_t)dof * m_pad_ + m] = (float)(o.c3[m] * u);
                    }
                HIPTRY(hipMemcpy(d_shapes_.p + off[i], vm.data(), vm.size() * sizeof(double), hipMemcpyHostToDevice));
                HIPTRY(hipMemcpy(d_g32_.p + off[i], vg.data(), vg.size() * sizeof(float), hipMemcpyHostToDevice));
                std::vector<double>().swap(o.shapes);
            }
        }
    }
    // FFAT maps
    {
        std::vector<long long> goff(N, 0);
        std::vector<FfatGeom> geom;
        std::vector<double> psi;
        for (int i = 0; i < N; ++i) {
            Object &o = objs_[i];
            o.maps_cover_modes = true;
            for (int m = 0; m < o.n_modes; ++m)
                if (m >= (int)o.geom.size() || !o.geom[m].valid) { o.maps_cover_modes = false; break; }
            goff[i] = (long long)geom.size();
            const int ng = std::min((int)o.geom.size(), m_pad_);
            nmodes[i] = o.n_modes;
            for (int m = 0; m < ng; ++m) {
                FfatGeom g = o.geom[m];
                g.psi_off += (long long)psi.size();
                geom.push_back(g);
            }
            // pad so that every mode < n_modes has an entry
            for (int m = ng; m < o.n_modes; ++m) {
                FfatGeom z;
                std::memset(&z, 0, sizeof(z));
                geom.push_back(z);
            }
            psi.insert(psi.end(), o.psi.begin(), o.psi.end());
            std::vector<double>().swap(o.psi);
        }
        HIPTRY(d_geom_off_.ensure(N));
        HIPTRY(hipMemcpy(d_geom_off_.p, goff.data(), N * sizeof(long long), hipMemcpyHostToDevice));
        geom_off_h_ = goff;
        if (!geom.empty()) {
            HIPTRY(d_geom_.ensure(geom.size()));
            HIPTRY(hipMemcpy(d_geom_.p, geom.data(), geom.size() * sizeof(FfatGeom), hipMemcpyHostToDevice));
        }
        if (!psi.empty()) {
            HIPTRY(d_psi_.ensure(psi.size()));
            HIPTRY(hipMemcpy(d_psi_.p, psi.data(), psi.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        // Objects whose valid modes all carry ONE geometry (a map file's header: box, centre, cell size, face layout) -- every scene
        // we know of: the maps of an object are made over the same box -- get their maps a second time, transposed, and their listener
        // events the kernel that locates a position once per event (kernels_exact.hip, ffat_lookup_shared_kernel).  Compared field by
        // field, bit for bit; a single differing mode keeps the object on the general kernels.  PBSO_FFAT_SHARED=0: never.
        {
            const char *env = std::getenv("PBSO_FFAT_SHARED");
            const bool want = !(env && std::atoi(env) == 0);
            std::vector<FfatShared> shared(N);
            std::memset(shared.data(), 0, shared.size() * sizeof(FfatShared));
            std::vector<double> mode_k(geom.size(), 0.0), psi_t;
            std::vector<int> mode_valid(geom.size(), 0);
            ffat_shared_h_.assign(N, 0);
            n_ffat_shared_ = 0;
            auto same_geometry = [](const FfatGeom &a, const FfatGeom &b) {
                return a.n_psi == b.n_psi && !std::memcmp(&a.cell_size, &b.cell_size, sizeof(double)) &&
                       !std::memcmp(a.center3, b.center3, sizeof(a.center3)) && !std::memcmp(a.low_corners, b.low_corners, sizeof(a.low_corners)) &&
                       !std::memcmp(a.center, b.center, sizeof(a.center)) && !std::memcmp(a.bbox_low, b.bbox_low, sizeof(a.bbox_low)) &&
                       !std::memcmp(a.bbox_top, b.bbox_top, sizeof(a.bbox_top)) && !std::memcmp(a.n_elements, b.n_elements, sizeof(a.n_elements)) &&
                       !std::memcmp(a.strides, b.strides, sizeof(a.strides));
            };
            for (int i = 0; i < N && want; ++i) {
                const Object &o = objs_[i];
                const FfatGeom *g = geom.data() + goff[i];
                const int ng = (int)((i + 1 < N ? goff[i + 1] : (long long)geom.size()) - goff[i]);
                const int nm = std::min(o.n_modes, ng);
                int first = -1, n_valid = 0;
                bool same = true;
                for (int m = 0; m < nm && same; ++m) {
                    if (!g[m].valid) continue;
                    if (first < 0) first = m;
                    else same = same_geometry(g[first], g[m]);
                    n_valid += 1;
                }
                if (!same || first < 0 || n_valid < 2) continue;          // (one map: nothing to share)
                const int pitch = (nm + 15) / 16 * 16;
                const size_t n_psi = (size_t)g[first].n_psi;
                shared[i].psit_off = (long long)psi_t.size();
                shared[i].pitch = pitch;
                shared[i].first_valid = first;
                psi_t.resize(psi_t.size() + n_psi * pitch, 0.0);
                double *dst = psi_t.data() + shared[i].psit_off;
                for (size_t c0 = 0; c0 < n_psi; c0 += 512)          // (blocks of cells: the strided writes of one block stay in cache)
                    for (int m = 0; m < nm; ++m) {
                        if (!g[m].valid) continue;
                        const double *src = psi.data() + g[m].psi_off;
                        for (size_t c = c0; c < std::min(n_psi, c0 + 512); ++c) dst[c * pitch + m] = src[c];
                    }
                ffat_shared_h_[i] = 1;
                n_ffat_shared_ += 1;
            }
            for (size_t j = 0; j < geom.size(); ++j) { mode_k[j] = geom[j].k; mode_valid[j] = geom[j].valid; }
            if (n_ffat_shared_ > 0) {
                HIPTRY(d_ffat_shared_.ensure(N));
                HIPTRY(hipMemcpy(d_ffat_shared_.p, shared.data(), N * sizeof(FfatShared), hipMemcpyHostToDevice));
                HIPTRY(d_ffat_k_.ensure(mode_k.size()));
                HIPTRY(hipMemcpy(d_ffat_k_.p, mode_k.data(), mode_k.size() * sizeof(double), hipMemcpyHostToDevice));
                HIPTRY(d_ffat_valid_.ensure(mode_valid.size()));
                HIPTRY(hipMemcpy(d_ffat_valid_.p, mode_valid.data(), mode_valid.size() * sizeof(int), hipMemcpyHostToDevice));
                HIPTRY(d_psi_t_.ensure(psi_t.size()));
                HIPTRY(hipMemcpy(d_psi_t_.p, psi_t.data(), psi_t.size() * sizeof(double), hipMemcpyHostToDevice));
            }
        }
    }
    HIPTRY(hipMemcpy(d_n_modes_.p, nmodes.data(), N * sizeof(int), hipMemcpyHostToDevice));
    HIPTRY(d_board_.ensure(4096));
    HIPTRY(hipMemset(d_board_.p, 0, 4096 * sizeof(unsigned)));
    // transfer rows: [0,N) _latest_transfer, [N,2N) the 1-slot transfer queue, then per-launch scratch
    HIPTRY(d_xfer_.ensure((size_t)2 * N * m_pad_));
    HIPTRY(hipMemset(d_xfer_.p, 0, (size_t)2 * N * m_pad_ * sizeof(double)));
    (void)warm_copy_engines();
    if (desc_.submit_thread > 0) submit_ = new SubmitQueue(desc_.device);
    finalized_ = true;
    return PBSO_OK;
}

// One-time work of the runtime that would otherwise block a step's submission, done here, untimed (PBSO_WARM_COPIES=0: not).
// Found with PBSO_TIMELINE=1 / scripts/debug/r03_stalls.py: an engine's first ~30 launches contained two hipMemcpyAsync calls
// (the step's ONE upload) that took 6 - 7 ms each whatever their size, with no buffer growing -- in the middle of a real-time run,
// or of a 20-step timed region (64 x 256 with a listener move per buffer measured 2800 x or 4900 x run to run, depending on
// whether one fell into the 40 timed steps).  (1) The runtime opens its copy queues lazily: a few overlapping uploads and
// read-backs on both streams remove the first.  (2) The other comes with the first upload whose stream is still WAITING for an
// event of the other stream when the call is made -- which is what a step's upload looks like as soon as the host runs a whole
// plan set ahead of the oscillator bank (the wait for ev_k1_done_ in front of it); it happened at a random launch, whenever the
// host first got that far ahead.  Issuing exactly that pattern here removes it.
int Engine::warm_copy_engines() {
    if (desc_.warm_copies < 0) return PBSO_OK;
    // Best effort: a failure here costs a slow first launch, never the engine (errors are swallowed, everything is released).
    // A caller's stream is not touched: the second pattern only needs SOME stream the preparation stream waits for.
#ifndef PBSO_WARM_CHUNK_MB
#define PBSO_WARM_CHUNK_MB 1
#endif
    const size_t chunk = (size_t)PBSO_WARM_CHUNK_MB << 20;
    const int n = 6;
    PinBuf<unsigned char> h;
    DevBuf<unsigned char> d;
    hipStream_t other = own_stream_ ? stream_ : nullptr;
    hipEvent_t ev = nullptr, ev2 = nullptr;
    auto body = [&]() -> hipError_t {
        hipError_t e;
        if (!other && (e = hipStreamCreateWithFlags(&other, hipStreamNonBlocking)) != hipSuccess) return e;
        if ((e = h.ensure(chunk * n)) != hipSuccess) return e;
        if ((e = d.ensure(chunk * n)) != hipSuccess) return e;
        std::memset(h.p, 0, chunk * n);
        for (int rep = 0; rep < 2; ++rep) {           // (1) the copy queues, both directions, both streams
            for (int i = 0; i < n; ++i)
                if ((e = hipMemcpyAsync(d.p + chunk * i, h.p + chunk * i, chunk, hipMemcpyHostToDevice, (i & 1) ? other : prep_stream_)) != hipSuccess) return e;
            for (int i = 0; i < n; ++i)
                if ((e = hipMemcpyAsync(h.p + chunk * i, d.p + chunk * i, chunk, hipMemcpyDeviceToHost, (i & 1) ? prep_stream_ : other)) != hipSuccess) return e;
        }
        if ((e = hipStreamSynchronize(prep_stream_)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(other)) != hipSuccess) return e;
        // (2) uploads behind a wait for an event that has not happened yet
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&ev2, hipEventDisableTiming)) != hipSuccess) return e;
        for (int rep = 0; rep < 4; ++rep) {
            for (int i = 0; i < n; ++i)
                if ((e = hipMemcpyAsync(d.p + chunk * i, h.p + chunk * i, chunk, hipMemcpyHostToDevice, other)) != hipSuccess) return e;
            if ((e = hipEventRecord(ev, other)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(prep_stream_, ev, 0)) != hipSuccess) return e;
            if ((e = hipMemcpyAsync(d.p, h.p, (size_t)384 << 10, hipMemcpyHostToDevice, prep_stream_)) != hipSuccess) return e;
            if ((e = hipEventRecord(ev2, prep_stream_)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(other, ev2, 0)) != hipSuccess) return e;
        }
        if ((e = hipStreamSynchronize(prep_stream_)) != hipSuccess) return e;
        return hipStreamSynchronize(other);
    };
    (void)body();
    (void)hipGetLastError();
    if (prep_stream_) (void)hipStreamSynchronize(prep_stream_);
    if (other) (void)hipStreamSynchronize(other);
    if (ev) (void)hipEventDestroy(ev);
    if (ev2) (void)hipEventDestroy(ev2);
    if (other && other != stream_) (void)hipStreamDestroy(other);
    h.release();
    d.release();
    return PBSO_OK;
}

// ---------------------------------------------------------------------------
// ModalSolver::enqueueForceMessage, modal_solver.h:329-333
int Engine::enqueue_force_impl(int obj, const pbso_force_msg &m, int64_t not_before, const char **why) {
    if (!finalized_) { *why = "enqueue_force before finalize"; return PBSO_ERR_STATE; }
    if (!valid_obj(obj)) { *why = "object id"; return PBSO_ERR_INVALID; }
    Object &o = objs_[obj];
    if (m.force_type < PBSO_POINT_FORCE || m.force_type > PBSO_AUTOREGRESSIVE_FORCE)
        { *why = "unrecognized force type"; return PBSO_ERR_INVALID; }             // assert modal_solver.h:73
    HostForceMsg h;
    h.force_type = m.force_type;
    h.sustained_start = m.sustained_force_start != 0;
    h.sustained_end = m.sustained_force_end != 0;
    h.clear_all = m.clear_all_forces != 0;
    h.data_kind = m.data_kind;
    h.not_before = not_before;
    // (the Force object itself is built when the message is dequeued)
    const bool need_ext = m.data_kind == PBSO_DATA_FACE || m.gaussian_width_us != 0.0 ||
                          (m.data_kind == PBSO_DATA_EXPLICIT && !h.clear_all);
    auto make_ext = [&](int n_data) {
        MsgExt *x = (MsgExt *)std::malloc(sizeof(MsgExt) + sizeof(double) * (size_t)(n_data > 0 ? n_data - 1 : 0));
        x->coords[0] = m.coords[0]; x->coords[1] = m.coords[1]; x->coords[2] = m.coords[2];
        x->gaussian_width_us = m.gaussian_width_us;
        x->n_data = n_data;
        return x;
    };
    switch (m.data_kind) {
    case PBSO_DATA_EXPLICIT:
        // a clearAllForces message returns from step() before any dimension check (modal_solver.h:186-189):
        // the GUI sends it with an empty data vector (tools/real_time_modal_sound.cpp:745-747)
        if (h.clear_all) { h.data_kind = PBSO_DATA_ZERO; break; }
        if (!m.data || m.n_data != o.n_modes)
            { *why = "dimension of force message incorrect"; return PBSO_ERR_INVALID; }   // assert :258
        h.ext = make_ext(m.n_data);
        if (m.n_data) std::memcpy(h.ext->data, m.data, sizeof(double) * (size_t)m.n_data);
        break;
    case PBSO_DATA_VERTEX:
    case PBSO_DATA_FACE: {
        if (!o.n_dof) { *why = "object has no mode shapes for on-device projection"; return PBSO_ERR_INVALID; }
        const int nv = o.n_dof / 3;
        const int cnt = m.data_kind == PBSO_DATA_VERTEX ? 1 : 3;
        for (int j = 0; j < cnt; ++j)
            if (m.vids[j] < 0 || m.vids[j] >= nv) { *why = "vertex id out of range"; return PBSO_ERR_INVALID; }
        for (int j = 0; j < 3; ++j) {
            h.vids[j] = m.vids[j];
            h.vn[j] = m.vn[j];
        }
        break;
    }
    case PBSO_DATA_ZERO:
        break;
    default:
        { *why = "data_kind"; return PBSO_ERR_INVALID; }
    }
    if (o.force_q.size() >= 1023) { std::free(h.ext); return 0; }      // ReaderWriterQueue(512): ceilToPow2(513)-1 usable slots
    if (need_ext && !h.ext) h.ext = make_ext(0);
    // keep arrival order monotone: a message cannot overtake an earlier one (FIFO)
    if (!o.force_q.empty()) h.not_before = std::max(h.not_before, o.force_q.back().not_before);
    o.force_q.push_back(std::move(h));
    return 1;
}

// pbso_enqueue_vertex_hits: a step's plain vertex hits as borrowed parallel arrays, object by object.
int Engine::enqueue_vertex_hits(int n, const int *objs, const int *vids, const double *vn, const int64_t *stamps) {
    if (!finalized_) return fail(PBSO_ERR_STATE, "enqueue_vertex_hits before finalize");
    if (script_.n > 0) return fail(PBSO_ERR_STATE, "a hit script is already pending (one per step)");
    if (n == 0) return 0;
    const int N = (int)objs_.size();
    hit_off_.assign((size_t)N + 1, 0);
    int prev = 0;
    for (int i = 0; i < n; ++i) {
        const int o = objs[i];
        if (o < prev || o >= N) return fail(PBSO_ERR_INVALID, "enqueue_vertex_hits: object ids must be valid and ascending");
        const int n_dof = objs_[o].n_dof;
        if (vids[i] < 0 || 3 * vids[i] + 2 >= n_dof)
            return fail(PBSO_ERR_INVALID, n_dof ? "vertex id out of range" : "object has no mode shapes for on-device projection");
        prev = o;
        hit_off_[(size_t)o + 1] = i + 1;
    }
    for (int o = 0; o < N; ++o)                       // objects without hits: an empty range at the running offset
        if (hit_off_[(size_t)o + 1] < hit_off_[o]) hit_off_[(size_t)o + 1] = hit_off_[o];
    script_.n = n; script_.objs = objs; script_.vids = vids; script_.vn = vn; script_.stamps = stamps;
    return n;
}

// hits h0 .. h1 - 1 of the script (all of object oi) enter the object's queue exactly as pbso_enqueue_force would put them: a
// full queue (1023 slots, modal_solver.h:105) REJECTS a message -- try_enqueue returns false, modal_solver.h:329-333, and the tool
// ignores it (tools/real_time_modal_sound.cpp:610) -- so the hits that do not fit are dropped and counted, nothing fails
int Engine::script_to_queue(int oi, int h0, int h1, const char **why) {
    (void)why;
    Object &o = objs_[oi];
    for (int h = h0; h < h1; ++h) {
        if (o.force_q.size() >= 1023) { dropped_hits_.fetch_add(h1 - h); break; }
        HostForceMsg m;
        m.force_type = PBSO_POINT_FORCE;
        m.data_kind = PBSO_DATA_VERTEX;
        m.not_before = script_.stamps[h];
        m.vids[0] = script_.vids[h];
        for (int j = 0; j < 3; ++j) m.vn[j] = script_.vn[3 * (size_t)h + j];
        if (!o.force_q.empty()) m.not_before = std::max(m.not_before, o.force_q.back().not_before);
        o.force_q.push_back(std::move(m));
    }
    return PBSO_OK;
}

int Engine::flush_script() {
    if (script_.n <= 0) return PBSO_OK;
    const int N = (int)objs_.size();
    const char *why = "";
    for (int o = 0; o < N; ++o) {
        int rc = script_to_queue(o, hit_off_[o], hit_off_[(size_t)o + 1], &why);
        if (rc != PBSO_OK) { script_.n = 0; return fail(rc, why); }
    }
    script_.n = 0;
    return PBSO_OK;
}

// Planner, before an object's buffers are planned: its share of the pending hit script.  An object that is idle -- no live
// force, nothing queued, no pending listener / parameter call -- takes its hits straight into the descriptors of this launch:
// hit after hit at buffer max(stamp, previous hit's buffer + 1), which is where ModalSolver::step would have dequeued it
// (at most one message per buffer, modal_solver.h:184), each a DESC_DIRECT impulse.  Everything else goes through the queue.
int Engine::consume_script(PlanCtx &c, int oi, int nb) {
    if (script_.n <= 0) return PBSO_OK;
    const int h0 = hit_off_[oi], h1 = hit_off_[(size_t)oi + 1];
    if (h0 >= h1) return PBSO_OK;
    Object &o = objs_[oi];
    const bool idle = o.force_q.empty() && o.active.empty() && !o.sustained && o.pending.empty() && !o.trans_full &&
                      !(!o.use_transfer && o.latest_row != XFER_UNIT) && !o.arprm_full;
    int h = h0;
    if (idle && device_profiles_ && direct_hits_) {
        int next_b = 0;
        for (; h < h1; ++h) {
            const int64_t rel = script_.stamps[h] - buffers_done_;
            const int b = (int)std::max<int64_t>(rel, next_b);
            if (b >= nb) break;
            BufDesc &d = plan_desc_[(size_t)oi * nb + b];
            const double *vnd = script_.vn + 3 * (size_t)h;
            const float vn[3] = {(float)vnd[0], (float)vnd[1], (float)vnd[2]};
            d.frow = 3 * script_.vids[h];
            std::memcpy(&d.prow, &vn[0], 4);
            std::memcpy(&d.tile_mask, &vn[1], 4);
            std::memcpy(&d.pad[0], &vn[2], 4);
            d.flags |= DESC_IMPULSE | DESC_DIRECT;
            d.amp = 1.f;
            next_b = b + 1;
        }
    }
    const char *why = "";
    int rc = script_to_queue(oi, h, h1, &why);       // what is left: beyond this launch, or the object is busy
    if (rc != PBSO_OK) return cfail(c, rc, why);
    return PBSO_OK;
}

int Engine::enqueue_force(int obj, const pbso_force_msg &m, int64_t not_before) {
    if (script_.n > 0) { int frc = flush_script(); if (frc != PBSO_OK) return frc; }
    const char *why = "";
    const int rc = enqueue_force_impl(obj, m, not_before, &why);
    return rc < 0 ? fail(rc, why) : rc;
}

// A whole step of a force script.  Messages of one object keep their order.  With planner threads
// (PBSO_PLAN_THREADS > 1) every thread enqueues the messages of its own range of objects.
int Engine::enqueue_force_batch(int n, const int *objs, const pbso_force_msg *msgs, const int64_t *stamps,
                                unsigned char *accepted) {
    if (script_.n > 0) { int frc = flush_script(); if (frc != PBSO_OK) return frc; }
    const int N = (int)objs_.size();
    const int T = (plan_threads_ > 1 && n >= 4096 && N >= 64) ? plan_threads_ : 1;
    std::vector<int> taken(T, 0), rcs(T, 0);
    std::vector<const char *> whys(T, "");
    // A script given object by object (ids ascending; each object's messages in their own order) is cut at the threads'
    // object ranges by binary search: no thread walks messages that are not its own, and an object's queue takes its
    // messages back to back (one ring's tail stays in the core's cache).  Any other order: every thread walks the batch.
    bool by_object = n > 0 && objs[0] >= 0 && objs[n - 1] < N;
    for (int i = 1; i < n && by_object; ++i) by_object = objs[i - 1] <= objs[i];
    auto job = [&](int t) {
        const int lo = (int)((long long)N * t / T), hi = (int)((long long)N * (t + 1) / T);
        int i0 = 0, i1 = n;
        if (by_object) {
            i0 = (int)(std::lower_bound(objs, objs + n, lo) - objs);
            i1 = (int)(std::lower_bound(objs, objs + n, hi) - objs);
        }
        for (int i = i0; i < i1; ++i) {
            const int o = objs[i];
            const bool mine = by_object || (o >= lo && o < hi) || (t == 0 && (o < 0 || o >= N));   // bad ids: reported by thread 0
            if (!mine) continue;
            const int rc = enqueue_force_impl(o, msgs[i], stamps[i], &whys[t]);
            if (rc < 0) { rcs[t] = rc; return; }
            if (accepted) accepted[i] = rc ? 1 : 0;
            taken[t] += rc ? 1 : 0;
        }
    };
    if (T > 1) {
        if (!pool_) {
            pool_ = new PlanPool(plan_threads_ - 1, desc_.plan_pin ? sched_getcpu() : -1);
        }
        pool_->run(T, job);
    } else {
        job(0);
    }
    int total = 0;
    for (int t = 0; t < T; ++t) {
        if (rcs[t] < 0) return fail(rcs[t], whys[t]);
        total += taken[t];
    }
    return total;
}

static void push_timed(std::deque<TimedEvent> &q, const TimedEvent &ev) {
    auto it = q.end();
    while (it != q.begin() && (it - 1)->not_before > ev.not_before) --it;
    q.insert(it, ev);
}

// ModalSolver::enqueueArprmMessageNoFail, modal_solver.h:382-393
int Engine::enqueue_arprm(int obj, const double a[2], double sigma, double mu, int64_t not_before) {
    if (!finalized_) return fail(PBSO_ERR_STATE, "enqueue_arprm before finalize");
    if (!valid_obj(obj)) return fail(PBSO_ERR_INVALID, "object id");
    path_to_pending(objs_[obj], INT64_MAX);              // (a parameter message that waits for its slot holds the events behind it back)
    TimedEvent ev;
    ev.kind = TimedEvent::ARPRM;
    ev.not_before = not_before;
    ev.v[0] = a[0]; ev.v[1] = a[1]; ev.v[2] = sigma; ev.v[3] = mu;
    ev.flag = 0;
    push_timed(objs_[obj].pending, ev);
    return 1;
}

// what ModalSolver::enqueueArprmMessage's try_enqueue would find (modal_solver.h:378-381): the slot taken, or a message waiting for it
int Engine::arprm_pending(int obj) {
    if (!finalized_) return fail(PBSO_ERR_STATE, "arprm_pending before finalize");
    if (!valid_obj(obj)) return fail(PBSO_ERR_INVALID, "object id");
    const Object &o = objs_[obj];
    if (o.arprm_full) return 1;
    for (const TimedEvent &ev : o.pending)
        if (ev.kind == TimedEvent::ARPRM) return 1;
    return 0;
}

// ModalSolver::computeTransfer(pos), modal_solver.h:286-300
int Engine::compute_transfer(int obj, const double pos[3], int64_t not_before) {
    if (!finalized_) return fail(PBSO_ERR_STATE, "compute_transfer before finalize");
    if (!valid_obj(obj)) return fail(PBSO_ERR_INVALID, "object id");
    Object &o = objs_[obj];
    path_to_pending(o, INT64_MAX);                                // (a path given earlier: its positions are in front of this one)
    if (!o.have_maps) return 0;                                   // :290-291
    if (!o.maps_cover_modes)                                      // (the maps are fixed at finalize: checked once, there)
        return fail(PBSO_ERR_MISSING_MAP, "FFAT map for a modeId in 0..N_modes-1 is missing (std::map::at throws)");
    if (not_before <= buffers_done_ && o.pending.empty() && o.trans_full) return 0;   // try_enqueue on a full 1-slot queue
    TimedEvent ev;
    ev.kind = TimedEvent::TRANSFER;
    ev.not_before = not_before;
    ev.v[0] = pos[0]; ev.v[1] = pos[1]; ev.v[2] = pos[2]; ev.v[3] = 0;
    ev.flag = 0;
    push_timed(o.pending, ev);
    return 1;
}

// a listener PATH: n computeTransfer(pos) calls (modal_solver.h:286-300) in one -- what the tool's camera callback issues frame
// after frame (tools/real_time_modal_sound.cpp:844, 1172), pre-scheduled as the throughput harness has it
int Engine::compute_transfer_path(int n, const int *objs, const double *pos, const int64_t *stamps, unsigned char *accepted) {
    if (n < 0 || (n > 0 && (!objs || !pos || !stamps))) return fail(PBSO_ERR_INVALID, "compute_transfer_path arguments");
    if (!finalized_) return fail(PBSO_ERR_STATE, "compute_transfer before finalize");
    // Runs of ONE object with ascending stamps -- the shape a camera path and the throughput harness have -- are kept as an array
    // with a cursor (Object::path) for the planner; anything else goes call by call.
    int taken = 0;
    const int N = (int)objs_.size();
    for (int i = 0; i < n;) {
        const int o = objs[i];
        int j = i + 1;
        while (j < n && objs[j] == o && stamps[j] > stamps[j - 1]) ++j;
        bool run = o >= 0 && o < N && j - i >= 4;
        if (run) {
            Object &ob = objs_[o];
            // every call of the run must be one pbso_compute_transfer accepts and appends: maps there, behind what the object's path
            // already holds, and the first not an immediate call that finds the 1-slot queue taken
            run = ob.have_maps && ob.maps_cover_modes && (!ob.path_left() || stamps[i] > ob.path.back().stamp) &&
                  !(stamps[i] <= buffers_done_ && ob.pending.empty() && !ob.path_left() && ob.trans_full);
            if (run) {
                if (!ob.path_left()) { ob.path.clear(); ob.path_head = 0; }
                for (int e = i; e < j; ++e) ob.path.push_back(Object::PathEv{stamps[e], {pos[3 * (size_t)e], pos[3 * (size_t)e + 1], pos[3 * (size_t)e + 2]}});
                if (accepted) std::memset(accepted + i, 1, (size_t)(j - i));
                taken += j - i;
            }
        }
        if (!run) {
            for (int e = i; e < j; ++e) {
                const int rc = compute_transfer(objs[e], pos + 3 * (size_t)e, stamps[e]);
                if (rc < 0) return rc;
                if (accepted) accepted[e] = rc ? 1 : 0;
                taken += rc ? 1 : 0;
            }
        }
        i = j;
    }
    return taken;
}

// the positions of the object's path stamped below `before` enter its pending list as pbso_compute_transfer puts them
void Engine::path_to_pending(Object &o, int64_t before) {
    while (o.path_left() && o.path[o.path_head].stamp < before) {
        const Object::PathEv &pe = o.path[o.path_head++];
        TimedEvent ev;
        ev.kind = TimedEvent::TRANSFER;
        ev.not_before = pe.stamp;
        ev.v[0] = pe.pos[0]; ev.v[1] = pe.pos[1]; ev.v[2] = pe.pos[2]; ev.v[3] = 0;
        ev.flag = 0;
        push_timed(o.pending, ev);
    }
    if (!o.path_left()) { o.path.clear(); o.path_head = 0; }
}

// Planner, before an object's buffers are planned: the positions of its path that fall into this launch.  For an object with
// nothing else going on -- no live force, no message or stamped call due, the transfer in use, the 1-slot queue free -- a position
// stamped s is what plan_object makes of it: try_enqueue into _queue_trans at the first buffer t >= s (modal_solver.h:286-300),
// dequeued by the same step (:242-256) -- one lookup event and that buffer's transfer row; a second position that falls into the
// same buffer finds the queue full and is dropped.  Otherwise they go through the pending list.
int Engine::consume_path(PlanCtx &c, int oi, int nb) {
    Object &o = objs_[oi];
    if (!o.path_left()) return PBSO_OK;
    const int64_t horizon = buffers_done_ + nb;
    if (o.path[o.path_head].stamp >= horizon) return PBSO_OK;
    const bool quiet = o.pending.empty() && (o.force_q.empty() || o.force_q.front().not_before >= horizon) && o.active.empty() &&
                       !o.sustained && o.use_transfer && !o.trans_full && !o.arprm_full;
    if (!quiet) {
        path_to_pending(o, horizon);
        return PBSO_OK;
    }
    int last_b = -1;
    for (; o.path_left(); ++o.path_head) {
        const Object::PathEv &pe = o.path[o.path_head];
        const int64_t rel = pe.stamp - buffers_done_;
        if (rel >= nb) break;
        const int b = rel > 0 ? (int)rel : 0;
        if (b == last_b) continue;                       // the queue still holds the position before it: try_enqueue fails, the move is lost
        FfatEvent fe;
        fe.obj = oi;
        fe.row = c.xfer_base + c.n_xfer++;
        fe.pos[0] = pe.pos[0]; fe.pos[1] = pe.pos[1]; fe.pos[2] = pe.pos[2];
        c.ffat.push_back(fe);
        o.trans_row = fe.row;
        o.latest_row = fe.row;
        plan_desc_[(size_t)oi * nb + b].trow = fe.row;
        last_b = b;
    }
    if (!o.path_left()) { o.path.clear(); o.path_head = 0; }
    return PBSO_OK;
}

// ModalSolver::setUseTransfer, modal_solver.h:148-152
int Engine::set_use_transfer(int obj, int use, int64_t not_before) {
    if (!finalized_) return fail(PBSO_ERR_STATE, "set_use_transfer before finalize");
    if (!valid_obj(obj)) return fail(PBSO_ERR_INVALID, "object id");
    path_to_pending(objs_[obj], INT64_MAX);
    TimedEvent ev;
    ev.kind = TimedEvent::USE_TRANSFER;
    ev.not_before = not_before;
    ev.v[0] = ev.v[1] = ev.v[2] = ev.v[3] = 0;
    ev.flag = use ? 1 : 0;
    push_timed(objs_[obj].pending, ev);
    return PBSO_OK;
}

void Engine::release(PlanCtx &c, ActiveForce &af) {
    if (af.slot >= 0) c.freed_this_plan.push_back(af.slot);      // (negative: an on-the-fly projection, no pool row)
    if (af.ar_state >= 0) c.freed_ar.push_back(af.ar_state);
    af.ar_state = -1;
}

int Engine::alloc_slot(PlanCtx &c) {
    if (!c.free_slots.empty()) {
        int s = c.free_slots.back();
        c.free_slots.pop_back();
        return s;
    }
    return (int)n_slots_.fetch_add(1);
}

// ---------------------------------------------------------------------------
// One object, one buffer: ModalSolver::step lines 184-256.
int Engine::plan_object(PlanCtx &c, int oi, int b, int nb, int64_t t) {
    Object &o = objs_[oi];
    BufDesc &d = plan_desc_[(size_t)oi * nb + b];

    // GUI-thread calls stamped for this buffer or earlier
    while (!o.pending.empty() && o.pending.front().not_before <= t) {
        const TimedEvent ev = o.pending.front();
        if (ev.kind == TimedEvent::ARPRM) {
            if (o.arprm_full) break;              // NoFail: the caller spins until the 1-slot queue drains
            o.arprm_full = true;
            std::memcpy(o.arprm, ev.v, sizeof(o.arprm));
        } else if (ev.kind == TimedEvent::TRANSFER) {
            if (!o.trans_full) {                  // try_enqueue; a full queue drops the update
                FfatEvent fe;
                fe.obj = oi;
                fe.row = c.xfer_base + c.n_xfer++;
                fe.pos[0] = ev.v[0]; fe.pos[1] = ev.v[1]; fe.pos[2] = ev.v[2];
                c.ffat.push_back(fe);
                o.trans_full = true;
                o.trans_row = fe.row;
            }
        } else {
            o.use_transfer = ev.flag != 0;
        }
        o.pending.pop_front();
    }

    // :184 dequeue at most one force message
    const bool due_msg = !o.force_q.empty() && o.force_q.front().not_before <= t;
    bool plain_hit = false;
    if (due_msg && device_profiles_ && o.active.empty() && !o.sustained) {
        const HostForceMsg &m0 = o.force_q.front();
        plain_hit = m0.force_type == PBSO_POINT_FORCE && !m0.clear_all && !m0.sustained_start && !m0.sustained_end &&
                    (m0.data_kind == PBSO_DATA_VERTEX || m0.data_kind == PBSO_DATA_FACE);
    }
    if (plain_hit && direct_hits_ && o.force_q.front().data_kind == PBSO_DATA_VERTEX && 3 * o.force_q.front().vids[0] + 2 < o.n_dof) {
        // ... and when the hit is at a vertex, not even a row: the oscillator bank dots the hit's normal with three rows
        // of the object's (float)(c3 * shape) table itself (DESC_DIRECT).  No work for any preparation kernel.
        const HostForceMsg &m0 = o.force_q.front();
        const float vn[3] = {(float)m0.vn[0], (float)m0.vn[1], (float)m0.vn[2]};
        d.frow = 3 * m0.vids[0];
        std::memcpy(&d.prow, &vn[0], 4);
        std::memcpy(&d.tile_mask, &vn[1], 4);
        std::memcpy(&d.pad[0], &vn[2], 4);
        d.flags |= DESC_IMPULSE | DESC_DIRECT;
        d.amp = 1.f;
        std::free(m0.ext);
        o.force_q.pop_front();
    } else if (plain_hit) {
        const HostForceMsg &m0 = o.force_q.front();
        {
            // The common case in one go -- the hit of a plain PointForce on an object with no live force: what the
            // general path below does for it (active list of one, Force::Add -> 1 at sample 0, erased next step,
            // modal_solver.h:195-221, forces.h:81-90) comes to one impulse row whose spatial vector is the
            // projection of this hit, evaluated by the combine kernel.
            ProjectEvent pe;
            pe.obj = oi;
            pe.kind = m0.data_kind;
            pe.slot = -1;
            for (int j = 0; j < 3; ++j) {
                pe.vids[j] = m0.vids[j];
                pe.coords[j] = m0.coord(j);
                pe.vn[j] = m0.vn[j];
            }
            c.slot_idx.push_back(-((int)c.proj_direct.size() + 1));
            c.proj_direct.push_back(pe);
            std::free(m0.ext);
            o.force_q.pop_front();
            d.frow = c.n_frows++;
            c.forced.push_back(&d);
            c.row_obj.push_back(oi);
            c.row_ptr.push_back((int)c.slot_idx.size());
            d.flags |= DESC_IMPULSE;
            d.tile_mask = 1u;
            d.amp = 1.f;
        }
    } else if (due_msg) {
        const HostForceMsg mess = o.force_q.front();
        o.force_q.pop_front();
        struct FreeExt { MsgExt *p; ~FreeExt() { std::free(p); } } free_ext{mess.ext};
        if (mess.clear_all) {                                           // :186-189
            for (ActiveForce &af : o.active) release(c, af);
            o.active.clear();
            d.flags |= DESC_SKIP;
            emitted_[(size_t)oi * plan_nb_total_ + plan_b0_ + b] = 0;
            return PBSO_OK;
        }
        // the message's modal data becomes one immutable row of the slot pool -- except for the hit of a plain
        // PointForce (forces.h:81-90: alive for exactly one buffer), whose projection the combine kernel
        // evaluates on the fly: slot -(event + 1) refers to the event, no pool row is written or read back
        const bool direct = device_profiles_ && mess.force_type == PBSO_POINT_FORCE && !mess.sustained_start && !mess.sustained_end &&
                            !o.sustained && (mess.data_kind == PBSO_DATA_VERTEX || mess.data_kind == PBSO_DATA_FACE);
        const int slot = direct ? -((int)c.proj_direct.size() + 1) : alloc_slot(c);
        if (mess.data_kind == PBSO_DATA_EXPLICIT || mess.data_kind == PBSO_DATA_ZERO) {
            const size_t off = c.stage.size();
            c.stage.resize(off + m_pad_, 0.0);
            if (mess.data_kind == PBSO_DATA_EXPLICIT)
                std::copy(mess.ext->data, mess.ext->data + mess.ext->n_data, c.stage.begin() + off);
            c.stage_slot.push_back(slot);
        } else {
            ProjectEvent pe;
            pe.obj = oi;
            pe.kind = mess.data_kind;
            pe.slot = slot;
            for (int j = 0; j < 3; ++j) {
                pe.vids[j] = mess.vids[j];
                pe.coords[j] = mess.coord(j);
                pe.vn[j] = mess.vn[j];
            }
            (direct ? c.proj_direct : c.proj).push_back(pe);
        }
        ActiveForce af;
        af.slot = slot;
        af.force_type = mess.force_type;
        af.force = ForceProfile::make(mess.force_type, mess.gaussian_width_us(), rate_);   // fresh Force, tools/...:281-294
        bool slot_used = false;
        if (mess.sustained_start) {                                     // :190-194
            for (ActiveForce &x : o.active) release(c, x);
            o.active.clear();
            o.sustained = true;
            o.active.push_back(af);
            slot_used = true;
        }
        if (!o.sustained) {                                             // :195-196
            o.active.push_back(af);
            slot_used = true;
        } else {                                                        // :197-200 data only
            if (o.active.empty())
                return cfail(c, PBSO_ERR_ASSERT, "sustained force list is empty (reference dereferences begin() of an empty list)");
            if (o.active.front().slot != slot) {
                c.freed_this_plan.push_back(o.active.front().slot);
                o.active.front().slot = slot;
                slot_used = true;
            }
        }
        if (mess.sustained_end) {                                       // :201-204
            for (ActiveForce &x : o.active) release(c, x);
            o.active.clear();
            o.sustained = false;
            slot_used = true;   // freed through the list (or below)
        }
        if (!slot_used && slot >= 0) c.freed_this_plan.push_back(slot);
    }

    // :206-240 time profile and spatial sum
    if ((!o.active.empty() || o.sustained) && device_profiles_) {
        // Force::Add bookkeeping only (who is alive, _count, PointForce::used); the samples
        // of Gaussian / AR profiles are generated on the device (K2, kernels_exact.hip)
        const int row_begin = (int)c.slot_idx.size();
        const int entry_begin = (int)c.prof_entries.size();
        int n_point = 0;
        bool dense = false;
        auto emit = [&](ActiveForce &af, bool set_param) -> bool {
            ForceProfile &f = af.force;
            ProfEntry e;
            std::memset(&e, 0, sizeof(e));
            e.kind = f.type;
            e.state = -1;
            switch (f.type) {
            case PBSO_POINT_FORCE:                                   // forces.h:81-90
                if (f.used) return false;
                f.used = true;
                ++n_point;
                break;
            case PBSO_GAUSSIAN_FORCE:                                // forces.h:92-105
                if (f.width == 0 || f.count >= f.cutoff * 2 * f.width_samples) return false;
                e.count = f.count;
                e.center = f.center;
                e.width_samples = f.width_samples;
                f.count += B_;
                dense = true;
                break;
            default:                                                 // forces.h:107-137
                if (af.ar_state < 0) {
                    if (!c.free_ar.empty()) { af.ar_state = c.free_ar.back(); c.free_ar.pop_back(); }
                    else af.ar_state = (int)n_ar_states_.fetch_add(1);
                    e.flags |= 1;
                }
                if (set_param) {
                    e.flags |= 2;
                    e.a0 = o.arprm[0]; e.a1 = o.arprm[1]; e.sigma = o.arprm[2]; e.mu = o.arprm[3];
                }
                e.state = af.ar_state;
                dense = true;
                break;
            }
            c.prof_entries.push_back(e);
            return true;
        };
        if (!o.sustained) {
            size_t w = 0;
            for (size_t r = 0; r < o.active.size(); ++r) {
                ActiveForce &af = o.active[r];
                if (!emit(af, false)) {
                    release(c, af);                                            // erase
                } else {
                    c.slot_idx.push_back(af.slot);
                    if (af.force.type == PBSO_POINT_FORCE) {
                        // a PointForce adds its one sample and is erased by the NEXT step()'s Add (forces.h:84-85,
                        // modal_solver.h:212-215); nothing can observe it in between, so it leaves the list now
                        // and an object with no other live force needs no bookkeeping for the next buffer
                        release(c, af);
                    } else {
                        if (w != r) o.active[w] = std::move(af);
                        ++w;
                    }
                }
            }
            o.active.resize(w);
        } else {
            if (o.active.size() != 1)
                return cfail(c, PBSO_ERR_ASSERT, "Should only have 1 concurrent sustained force");   // assert :223
            ActiveForce &af = o.active.front();
            bool sp = false;
            if (af.force_type == PBSO_AUTOREGRESSIVE_FORCE && o.arprm_full) {   // :226-236
                o.arprm_full = false;
                sp = true;
            }
            emit(af, sp);                        // the return value is ignored, modal_solver.h:238
            c.slot_idx.push_back(af.slot);
        }
        if ((int)c.slot_idx.size() > row_begin && (dense || n_point)) {
            d.frow = c.n_frows++;
            c.forced.push_back(&d);
            c.row_obj.push_back(oi);
            c.row_ptr.push_back((int)c.slot_idx.size());
            if (!dense) {
                d.flags |= DESC_IMPULSE;          // PointForce(s) only: n * delta[0], no profile row
                d.tile_mask = 1u;
                d.amp = (float)n_point;
                c.prof_entries.resize(entry_begin);
            } else {
                d.prow = c.n_prows++;
                c.prow_obj.push_back(oi);
                d.tile_mask = n_tiles_ >= 32 ? 0xFFFFFFFFu : ((1u << n_tiles_) - 1u);
                ProfRow pr = {d.prow, entry_begin, (int)c.prof_entries.size()};
                c.prof_rows.push_back(pr);
                if (c.chain_obj != oi) {           // rows of one object are contiguous (object-major plan)
                    c.chain_ptr.push_back((int)c.prof_rows.size() - 1);
                    c.chain_obj = oi;
                }
            }
        } else {
            c.slot_idx.resize(row_begin);          // nothing active produced samples: a force-free buffer
            c.prof_entries.resize(entry_begin);
        }
    } else if (!o.active.empty() || o.sustained) {
        double *T = c.tbuf.data();
        std::fill(T, T + c.t_extent, 0.0);
        c.t_extent = 0;
        const int row_begin = (int)c.slot_idx.size();
        if (!o.sustained) {
            size_t w = 0;
            for (size_t r = 0; r < o.active.size(); ++r) {
                ActiveForce &af = o.active[r];
                const bool added = af.force.add(T, B_, &c.t_extent);
                if (!added) {
                    release(c, af);                                            // erase
                } else {
                    c.slot_idx.push_back(af.slot);
                    if (w != r) o.active[w] = std::move(af);
                    ++w;
                }
            }
            o.active.resize(w);
        } else {
            if (o.active.size() != 1)
                return cfail(c, PBSO_ERR_ASSERT, "Should only have 1 concurrent sustained force");   // assert :223
            ActiveForce &af = o.active.front();
            if (af.force_type == PBSO_AUTOREGRESSIVE_FORCE && o.arprm_full) {   // :226-236
                o.arprm_full = false;
                af.force.set_param(o.arprm, o.arprm[2], o.arprm[3]);
            }
            af.force.add(T, B_, &c.t_extent);
            c.slot_idx.push_back(af.slot);
        }
        uint32_t mask = 0;
        int last_nz = -1;
        for (int i = 0; i < c.t_extent; ++i)
            if (T[i] != 0.0) { mask |= 1u << (i / TILE); last_nz = i; }
        if ((int)c.slot_idx.size() > row_begin && mask) {
            d.frow = c.n_frows++;
            c.forced.push_back(&d);
            d.tile_mask = mask;
            c.row_obj.push_back(oi);
            c.row_ptr.push_back((int)c.slot_idx.size());
            if (last_nz == 0) {
                d.flags |= DESC_IMPULSE;                  // PointForce(s): amp * delta[0], no profile row
                d.amp = (float)T[0];
            } else {
                d.prow = c.n_prows++;
                c.prow_obj.push_back(oi);
                const size_t off = c.tprof.size();
                c.tprof.resize(off + b_pad_, 0.f);
                for (int i = 0; i <= last_nz; ++i) c.tprof[off + i] = (float)T[i];
            }
        } else {
            c.slot_idx.resize(row_begin);          // S * 0 == 0: a force-free buffer
        }
    }

    // :242-256 transfer selection (single caller thread: try_lock always succeeds)
    if (o.use_transfer) {
        if (o.trans_full) {
            o.latest_row = o.trans_row;
            d.trow = o.trans_row;
            o.trans_full = false;
        }
    } else if (o.latest_row != XFER_UNIT) {
        o.latest_row = XFER_UNIT;
        d.trow = XFER_UNIT;
    }
    return PBSO_OK;
}

// All buffers of one object.  Objects are independent, and between stamped
// messages an idle object needs no bookkeeping at all: jump to the next stamp.
int Engine::plan_object_span(PlanCtx &c, int oi, int nb) {
    Object &o = objs_[oi];
    {
        int rc = consume_script(c, oi, nb);
        if (rc == PBSO_OK) rc = consume_path(c, oi, nb);
        if (rc != PBSO_OK) return rc;
    }
    int b = 0;
    while (b < nb) {
        const int64_t t = buffers_done_ + b;
        // The buffer of a listener PATH in one step: nothing is due but one computeTransfer (a camera move per frame,
        // tools/real_time_modal_sound.cpp:844, 1172), no force is alive, the 1-slot queue is free -- what plan_object does for it
        // (try_enqueue into _queue_trans, modal_solver.h:286-300; dequeued at :242-256 in the same step) comes to one lookup event
        // and this buffer's transfer row.
        if (!o.pending.empty() && o.pending.front().kind == TimedEvent::TRANSFER && o.pending.front().not_before <= t &&
            o.active.empty() && !o.sustained && o.use_transfer && !o.trans_full &&
            (o.force_q.empty() || o.force_q.front().not_before > t) && (o.pending.size() == 1 || o.pending[1].not_before > t)) {
            const TimedEvent &ev = o.pending.front();
            FfatEvent fe;
            fe.obj = oi;
            fe.row = c.xfer_base + c.n_xfer++;
            fe.pos[0] = ev.v[0]; fe.pos[1] = ev.v[1]; fe.pos[2] = ev.v[2];
            c.ffat.push_back(fe);
            o.pending.pop_front();
            o.trans_row = fe.row;
            o.latest_row = fe.row;
            plan_desc_[(size_t)oi * nb + b].trow = fe.row;
            ++b;
            continue;
        }
        const bool due = (!o.pending.empty() && o.pending.front().not_before <= t) ||
                         (!o.force_q.empty() && o.force_q.front().not_before <= t);
        const bool live = !o.active.empty() || o.sustained || (o.trans_full && o.use_transfer) ||
                          (!o.use_transfer && o.latest_row != XFER_UNIT);
        if (!due && !live) {
            int64_t next = INT64_MAX;
            if (!o.pending.empty()) next = std::min(next, o.pending.front().not_before);
            if (!o.force_q.empty()) next = std::min(next, o.force_q.front().not_before);
            if (next >= buffers_done_ + nb) break;
            b = (int)(next - buffers_done_);
            continue;
        }
        int rc = plan_object(c, oi, b, nb, t);
        if (rc != PBSO_OK) return rc;
        ++b;
    }
    return PBSO_OK;
}

int Engine::plan(int nb) {
    const int N = (int)objs_.size();
    PlanSet &ps = set_[cur_set_];
    // one pinned arena per plan set, uploaded with ONE copy: [descriptors | xfer_init | everything the planner emits]
    ps.off_xfer_init = arena_align((size_t)N * nb * sizeof(BufDesc));
    ps.front_bytes = ps.off_xfer_init + arena_align((size_t)N * sizeof(int));
    HIPTRY(ps.h_arena.ensure_keep(std::max(ps.front_bytes, ps.last_bytes), 0));
    plan_desc_ = reinterpret_cast<BufDesc *>(ps.h_arena.p);
    int *h_xfer_init = reinterpret_cast<int *>(ps.h_arena.p + ps.off_xfer_init);
    const auto tp0 = std::chrono::steady_clock::now();
    const BufDesc dflt = {-1, -1, 0u, 0.f, XFER_KEEP, 0u, {0, 0}};
    std::fill(plan_desc_, plan_desc_ + (size_t)N * nb, dflt);
    busy_.clear();
    for (int i = 0; i < N; ++i) {
        const Object &o = objs_[i];
        h_xfer_init[i] = o.latest_row;
        if (!o.force_q.empty() || !o.active.empty() || !o.pending.empty() || o.trans_full ||
            o.sustained || (!o.use_transfer && o.latest_row != XFER_UNIT) ||
            (script_.n > 0 && hit_off_[(size_t)i + 1] > hit_off_[i]) || (o.path_left() && o.path[o.path_head].stamp < buffers_done_ + nb))
            busy_.push_back(i);
    }
    // contiguous shares of the busy objects, one planning context (host thread) each
    const int nbusy = (int)busy_.size();
    const int T = std::max(1, std::min(plan_threads_, nbusy / plan_grain_));      // at least plan_grain_ objects per thread
    std::vector<int> lo(T + 1);
    for (int t = 0; t <= T; ++t) lo[t] = (int)((long long)nbusy * t / T);
    // every stamped computeTransfer that can fire in this batch may need one scratch row
    std::vector<size_t> need(T, 0);
    size_t need_all = 0;
    for (int t = 0; t < T; ++t) {
        for (int k = lo[t]; k < lo[t + 1]; ++k) {
            for (const TimedEvent &ev : objs_[busy_[k]].pending) {
                if (ev.not_before >= buffers_done_ + nb) break;      // sorted by stamp (push_timed): the rest is later
                if (ev.kind == TimedEvent::TRANSFER) ++need[t];
            }
            // (the object's path: at most one position fires per buffer)
            need[t] += std::min<size_t>(objs_[busy_[k]].path.size() - objs_[busy_[k]].path_head, (size_t)nb);
        }
        need_all += need[t];
    }
    if ((int)need_all > xfer_cap_) {
        const int ncap = std::max<int>((int)need_all, 2 * xfer_cap_ + 16);
        // rows [2N + s*cap, ...) change meaning with cap: nothing may be in flight (ensure() drains) -- nor still waiting in the
        // submitting thread's queue: its recorded launches hold the old block's address, and the keep-copy must come behind them
        {
            int drc = drain_submit();
            if (drc != PBSO_OK) return drc;
        }
        HIPTRY(d_xfer_.ensure((size_t)(2 * N + N_SETS * ncap) * m_pad_, true, stream_));
        HIPTRY(hipDeviceSynchronize());
        xfer_cap_ = ncap;
    }
    {
        int xb = 2 * N + cur_set_ * xfer_cap_;
        for (int t = 0; t < T; ++t) {
            ctx_[t].begin();
            ctx_[t].xfer_base = xb;
            xb += (int)need[t];
        }
    }
    auto job = [&](int t) {
        PlanCtx &c = ctx_[t];
        for (int k = lo[t]; k < lo[t + 1]; ++k) {
            c.rc = plan_object_span(c, busy_[k], nb);
            if (c.rc != PBSO_OK) return;
        }
    };
    const auto tp1 = std::chrono::steady_clock::now();
    if (T > 1) {
        if (!pool_) {
            pool_ = new PlanPool(plan_threads_ - 1, desc_.plan_pin ? sched_getcpu() : -1);
        }
        pool_->run(T, job);
    } else {
        job(0);
    }
    const auto tp2 = std::chrono::steady_clock::now();
    script_.n = 0;                                   // (every object with hits was busy: all of the script is consumed)
    for (int t = 0; t < T; ++t)
        if (ctx_[t].rc != PBSO_OK) return fail(ctx_[t].rc, ctx_[t].err);

    // merge in object order: the numbering is the one a single context would have produced
    row_ptr_.assign(1, 0);
    slot_idx_.clear(); row_obj_.clear(); tprof_.clear(); stage_.clear(); stage_slot_.clear();
    proj_.clear(); proj_direct_.clear(); ffat_.clear(); prof_entries_.clear(); prof_rows_.clear(); chain_ptr_.clear(); prow_obj_.clear();
    n_frows_ = 0;
    n_prows_ = 0;
    for (int t = 0; t < T; ++t) {
        PlanCtx &c = ctx_[t];
        const int base_f = n_frows_, base_p = n_prows_, base_s = (int)slot_idx_.size();
        const int base_e = (int)prof_entries_.size(), base_r = (int)prof_rows_.size();
        if (t > 0) {
            for (BufDesc *d : c.forced) {
                d->frow += base_f;
                if (d->prow >= 0) d->prow += base_p;
            }
        }
        for (int e : c.row_ptr) row_ptr_.push_back(base_s + e);
        const int base_d = (int)proj_direct_.size();
        if (base_d == 0) slot_idx_.insert(slot_idx_.end(), c.slot_idx.begin(), c.slot_idx.end());
        else for (int e : c.slot_idx) slot_idx_.push_back(e >= 0 ? e : e - base_d);     // on-the-fly projections: global event index
        proj_direct_.insert(proj_direct_.end(), c.proj_direct.begin(), c.proj_direct.end());
        row_obj_.insert(row_obj_.end(), c.row_obj.begin(), c.row_obj.end());
        prow_obj_.insert(prow_obj_.end(), c.prow_obj.begin(), c.prow_obj.end());
        tprof_.insert(tprof_.end(), c.tprof.begin(), c.tprof.end());
        prof_entries_.insert(prof_entries_.end(), c.prof_entries.begin(), c.prof_entries.end());
        for (ProfRow r : c.prof_rows) {
            r.prow += base_p;
            r.entry_begin += base_e;
            r.entry_end += base_e;
            prof_rows_.push_back(r);
        }
        for (int ci : c.chain_ptr) chain_ptr_.push_back(base_r + ci);
        stage_.insert(stage_.end(), c.stage.begin(), c.stage.end());
        stage_slot_.insert(stage_slot_.end(), c.stage_slot.begin(), c.stage_slot.end());
        proj_.insert(proj_.end(), c.proj.begin(), c.proj.end());
        ffat_.insert(ffat_.end(), c.ffat.begin(), c.ffat.end());
        n_frows_ += c.n_frows;
        n_prows_ += c.n_prows;
        // what this pass released becomes reusable from the next plan on (this launch still reads it)
        c.free_slots.insert(c.free_slots.end(), c.freed_this_plan.begin(), c.freed_this_plan.end());
        c.free_ar.insert(c.free_ar.end(), c.freed_ar.begin(), c.freed_ar.end());
    }
    build_ar_tables();
    const auto tp3 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    hprof_[1] += ms(tp0, tp1);
    hprof_[2] += ms(tp1, tp2);
    hprof_[3] += ms(tp2, tp3);
    return PBSO_OK;
}

// ---------------------------------------------------------------------------
// The row-parallel form of K2 (kernels_exact.hip): which AR forces add samples in this launch, in which rows, and how many
// candidate pairs of each force's engine to evaluate.  A force needs ceil(n_uses * frames / 2) accepted pairs (one more when a
// cached variate is not there to start with: counted); a candidate is accepted with probability pi / 4, so
// pairs / 0.7854 + 8 sigma of the binomial + one batch is asked for -- ar_zero_state_kernel continues the sequence by itself
// should that ever fall short.
void Engine::build_ar_tables() {
    ar_streams_.clear(); ar_uses_.clear(); seg_stream_.clear();
    ar_max_segs_ = 0;
    k2_rows_launch_ = device_profiles_ && k2_rows_ && !ar_serial_ && B_ >= 3 && !prof_rows_.empty();
    if (!k2_rows_launch_) return;
    const size_t n_states = n_ar_states_.load();
    if (ar_stream_of_state_.size() < n_states) ar_stream_of_state_.resize(n_states, -1);
    ar_last_use_.clear(); ar_epoch_.clear(); ar_param_.clear();
    for (size_t ei = 0; ei < prof_entries_.size(); ++ei) {
        ProfEntry &e = prof_entries_[ei];
        if (e.kind != PBSO_AUTOREGRESSIVE_FORCE) continue;
        int &si = ar_stream_of_state_[(size_t)e.state];
        if (si < 0) {
            si = (int)ar_streams_.size();
            ArStream S;
            std::memset(&S, 0, sizeof(S));
            S.state = e.state;
            S.reset = (e.flags & 1) ? 1 : 0;
            ar_streams_.push_back(S);
            ar_last_use_.push_back(-1); ar_epoch_.push_back(-1); ar_param_.push_back(-1);
        } else if (e.flags & 1) {
            k2_rows_launch_ = false;                 // a slot constructed twice within one launch: the planner never does this
            break;
        }
        ArUse U;
        std::memset(&U, 0, sizeof(U));
        U.entry = (int)ei;
        U.stream = si;
        U.u = ar_streams_[(size_t)si].n_uses++;
        if (e.flags & 3) { ar_epoch_[(size_t)si] = U.u; ar_param_[(size_t)si] = (int)ei; }
        U.epoch_u = ar_epoch_[(size_t)si];
        U.param_entry = ar_param_[(size_t)si];
        e.count = (int)ar_uses_.size();
        ar_last_use_[(size_t)si] = (int)ar_uses_.size();
        ar_uses_.push_back(U);
    }
    int use0 = 0, seg0 = 0;
    for (size_t si = 0; si < ar_streams_.size(); ++si) {
        ArStream &S = ar_streams_[si];
        ar_stream_of_state_[(size_t)S.state] = -1;
        if (!k2_rows_launch_) continue;
        ar_uses_[(size_t)ar_last_use_[si]].last = 1;
        S.use0 = use0;
        use0 += S.n_uses;
        const double pairs = 0.5 * ((double)S.n_uses * B_ + 1.0);
        const double want = (pairs / 0.78539 + 8.0 * std::sqrt(pairs) + 256.0) * (k2_margin_pct_ / 100.0);
        S.n_seg = std::max(1, (int)std::ceil(want / K2_SEG));
        S.seg_base = seg0;
        seg0 += S.n_seg;
        ar_max_segs_ = std::max(ar_max_segs_, S.n_seg);
        seg_stream_.insert(seg_stream_.end(), (size_t)S.n_seg, (int)si);
    }
    if (ar_max_segs_ > 8192) k2_rows_launch_ = false;     // (the prefix table of a stream's segments lives in LDS: > 8 M candidates per force and launch)
    if (!k2_rows_launch_) { ar_streams_.clear(); ar_uses_.clear(); seg_stream_.clear(); }
}

// ---------------------------------------------------------------------------
// One step = nb buffers for every object.  Long steps are cut into launches of at most
// chunk_buffers_ buffers so that the host plans chunk c+1 while the device runs chunk c (the same
// overlap consecutive steps have); results do not depend on the cut (state, force lists and queues
// carry over exactly as between steps).
// K5: does this launch run time-chunked, with which team shape and how many buffers per chunk?  Auto: only while the scene
// leaves SIMDs idle (fewer than two waves per SIMD in its largest shape) and few of the launch's (object, buffer) pairs
// carry a dense force profile (the scan steps those per sample, serially per object).  The shape: the most modes per lane
// that still fills the chip when every buffer is its own chunk (a wave of four modes per lane amortises its operand table
// over more work), unless the padding to whole waves wastes columns; the chunk length minimises rounds x (length + start-up).
bool Engine::choose_time_chunks(int nb, int n_dense_rows, int *set, int *cb) const {
    if (!tc_ok_ || !d_scan_.p || (nb < 2 && tc_mode_ <= 0)) return false;
    if (n_dump_ > 0) return false;                       // (objects that keep their block-start states for a listener mix: the walk in buffer order, as K1p)
    const long long N = (long long)objs_.size();
    // Dense-profile buffers (Gaussian / AR, sustained contact: forces.h:92-128, modal_solver.h:222-240) take their place in the scan
    // through their increments (dense_increment_kernel); the bank steps them in its forced block path.
    if (n_dense_rows > 0 && !d_ftab_.p) return false;
    // (engines whose bank steps dense buffers per sample -- the split-bf16 projection, forced_block < 0 -- keep their launches with
    //  many of them on the kernels that walk the buffers in order, as before round 5)
    if (tc_mode_ == 0 && !ftab_forced_ && (long long)n_dense_rows * 8 > N * nb) return false;
    // (every dense row costs m_pad x 8 bytes of increments per plan set: a launch whose rows would take more than 1 GiB walks its
    //  buffers in order instead -- 1024 x 512 sustained scraping in ten-second steps is 880 640 rows = 3.6 GB)
    if ((long long)n_dense_rows * m_pad_ * 8 > (1LL << 30)) return false;
    const bool dense_majority = (long long)n_dense_rows * 2 > N * nb;
    // (a scene small enough for the pipeline kernel keeps its mostly-dense launches WITHOUT qnorm rows there: teams of five waves per
    //  64 modes walk the buffers in order and evaluate every increment once -- cut in time they are evaluated twice, for the scan and
    //  in the bank: 8 x 4096 x 86 scraping 0.25 ms per step against 0.27.  With qnorm rows the re-stepped samples are the long stage
    //  and want the whole chip's vector ALUs at three waves per SIMD: cut in time 0.33 ms, five-wave teams 0.49, three-wave teams 0.45)
    if (dense_majority && tc_mode_ == 0 && use_split() && ftab_forced_ && k2_rows_launch_ && desc_.qnorm_mode == PBSO_QNORM_OFF) return false;
    int k = 0;
    for (int c = 2; c >= 1; --c)
        if (tc_[c].cover * 100 <= tc_[0].cover * 115 && tc_[c].waves * nb >= (long long)tc_[c].waves_per_cu * n_cus_) { k = c; break; }
    // a launch of mostly dense buffers: ONE mode per lane -- the only shape whose registers hold the increment table F next to the
    // operand table W, so that the state moves a block at a time (F . T_n on the matrix pipe) instead of a sample at a time, and
    // whose per-sample chain for the qnorm rows has three waves per SIMD to hide behind
    if (dense_majority && ftab_forced_) k = 0;
    if (tc_shape_ == 1 || tc_shape_ == 2 || tc_shape_ == 4) k = tc_shape_ == 4 ? 2 : tc_shape_ - 1;
    const long long capacity = (long long)tc_[k].waves_per_cu * n_cus_;
    const long long waves = tc_[k].waves;
    int best = nb;
    if (tc_mode_ > 0) {
        best = std::min(tc_mode_, nb);
    } else {
        // cost of a launch in buffer times: rounds of workgroups x (chunk length + 0.3 for a workgroup's start-up: operand table,
        // coefficients, state).  A scene LARGER than the chip, whose walk is a launch of several rounds of full-length
        // workgroups, does not pack perfectly -- a CU may be handed a ninth workgroup while another gets seven
        // (scripts/census.py 2048: a third, nearly empty round; 1100 / 1536 / 2048 x 512 walk 1.68 / 2.04 / 2.72 ms: half a round
        // more than their 1.07 / 1.5 / 2 rounds) -- and the shorter the workgroups, the cheaper that tail: such scenes are cut in
        // time too (1.22 / 1.66 / 2.21 ms).  A scene that fills the chip exactly (1024 x 512: one round) is not.
        const bool over_full = waves > capacity;
        // (round 6, ADVICE r05: a mostly-dense scene that fills the chip anyway gains nothing from the cut -- its increments are
        //  evaluated twice, for the scan and in the bank: 1024 x 512 sustained scraping 7.90 ms per second of audio cut in time against
        //  7.37 for the walk, profiles/r06_bench_c4scr_*.json -- so the cut is for dense scenes that leave wave slots free)
        if (dense_majority && waves >= capacity) return false;
        if (over_full && !dense_majority) {
            // measured, in rounds of the exactly-full chip (scripts/debug/r04_tcrounds.py): the walk of 1100 / 1536 / 2048 / 3000
            // objects x 512 modes takes 1.57 / 1.90 / 2.54 / 3.07 -- whole rounds + a last, partly filled one that runs faster the
            // emptier it is (0.57 + 0.54 x its fill), or half a round of stragglers when the rounds are exactly full -- and the
            // same scenes cut in time 1.07 x their share of the chip (1.14 / 1.57 / 2.12 / 3.17): at 3000 the walk wins
            const double frac = (double)waves / (double)capacity, whole = std::floor(frac), fill = frac - whole;
            const double walk = whole + (fill > 1e-9 ? 0.57 + 0.54 * fill : 0.5);
            if (frac * 1.07 >= walk) return false;
        }
        double best_cost = 0;
        for (int c = nb; c >= 1; --c) {
            const long long chunks = (nb + c - 1) / c;
            const long long rounds = (waves * chunks + capacity - 1) / capacity;
            const double cost = ((double)rounds + (over_full ? 0.5 : 0.0)) * (c + 0.3);
            if (c == nb || cost < best_cost - 1e-9) { best = c; best_cost = cost; }
        }
    }
    if ((nb + best - 1) / best < 2 && tc_mode_ <= 0) return false;       // (forced: every launch, so that any cut of a step runs the same arithmetic)
    *set = k;
    *cb = best;
    return true;
}

int Engine::step(int nb, void *d_audio_user) {
    HIPTRY(hipSetDevice(desc_.device));      // the caller's thread may have another device current
    if (!finalized_) return fail(PBSO_ERR_STATE, "step before finalize");
    if (nb <= 0) return fail(PBSO_ERR_INVALID, "n_buffers must be > 0");
    if (failed_) return fail(PBSO_ERR_STATE, "an earlier step failed half-way (" + failed_why_ + "): queues and force lists are no "
                                             "longer consistent, create a new engine");
    const int N = (int)objs_.size();
    const bool defer = submit_ != nullptr;             // (GROWTRY: a buffer that grows waits for the recorded calls first)
    float *audio = (float *)d_audio_user;
    // outputs of the whole step (growth drains the device first, see DevBuf::ensure)
    if (!audio) {
        GROWTRY(d_audio_, (size_t)N * nb * B_, false, stream_);
        audio = d_audio_.p;
    }
    if (desc_.qnorm_mode != PBSO_QNORM_OFF || n_dump_ > 0) GROWTRY(d_qnorm_, (size_t)N * nb * m_pad_, false, stream_);
    if (n_dump_ > 0) {
        // block-start states of the objects a multi-listener mix was asked for (pbso_listeners_enable)
        GROWTRY(d_xdump_, (size_t)n_dump_ * nb * 32 * m_pad_ * 2, false, stream_);
        GROWTRY(d_xscale_, (size_t)n_dump_ * nb * m_pad_, false, stream_);
        if (dump_rows_dirty_) {
            if (defer) { int drc = drain_submit(); if (drc != PBSO_OK) return drc; }
            GROWTRY(d_dump_row_, N, false, stream_);
            HIPTRY(hipMemcpyAsync(d_dump_row_.p, dump_row_.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, stream_));
            HIPTRY(hipStreamSynchronize(stream_));           // (dump_row_ is pageable host memory)
            dump_rows_dirty_ = false;
        }
        dump_nb_ = nb;
        std::fill(dump_valid_.begin(), dump_valid_.end(), 1);
    }
    {
        // (a step's launches may run on different kernels: room for either kind's partial rows)
        int rows = std::max(use_split() ? n_ts_part_rows_ : 0, n_part_rows_);
        if (tc_ok_) for (const TcSet &ts : tc_) rows = std::max(rows, ts.n_part_rows);
        if (rows) GROWTRY(d_audio_parts_, (size_t)rows * nb * B_, false, stream_);
    }
    emitted_.assign((size_t)N * nb, 1);
    const int64_t step_id = tot_steps_;
    double plan_ms = 0;
    for (int b0 = 0; b0 < nb; b0 += chunk_buffers_) {
        int rc = step_chunk(std::min(chunk_buffers_, nb - b0), b0, nb, audio, step_id);
        if (rc != PBSO_OK) {
            // the reference would have aborted here (assert) or be equally stuck (device error)
            failed_ = true;
            failed_why_ = err_;
            return rc;
        }
        plan_ms += last_plan_ms_;
    }
    last_plan_ms_ = plan_ms;
    tot_plan_ms_ += plan_ms;
    tot_steps_ += 1;
    last_audio_ = audio;
    last_nb_ = nb;
    return PBSO_OK;
}

int Engine::step_chunk(int nb, int b0, int nb_total, float *audio, int64_t step_id) {
    const int N = (int)objs_.size();
    PlanSet &ps = set_[cur_set_];
    plan_b0_ = b0;
    plan_nb_total_ = nb_total;
    const auto tw0 = std::chrono::steady_clock::now();
    // the second submitting thread: this launch's stream calls are recorded (QHIP / QLAUNCH) and handed over at the end
    const bool defer = submit_ != nullptr;
    std::vector<SubmitOp> ops;
    if (defer) {
        ops.reserve(48);
        submit_->wait(set_batch_[cur_set_]);             // the launch that last used this plan set has been MADE (its events recorded) ...
        std::string why;
        if (submit_->error(&why)) return fail(PBSO_ERR_HIP, "a launch of an earlier step failed on the submitting thread: " + why);
    }
    HIPTRY(hipEventSynchronize(ev_set_[cur_set_]));      // ... and this set's previous uploads are done

    const auto t0 = std::chrono::steady_clock::now();
    hprof_[0] += std::chrono::duration<double, std::milli>(t0 - tw0).count();
    int rc = plan(nb);
    if (rc != PBSO_OK) return rc;
    // keep _latest_transfer / the transfer queue in their persistent rows after the launch
    std::vector<int> copy_latest, copy_queued;
    for (int i : busy_) {
        Object &o = objs_[i];
        if (o.latest_row >= 0 && o.latest_row != i) {
            copy_latest.push_back(o.latest_row);
            copy_latest.push_back(i);
            o.latest_row = i;
        }
        if (o.trans_full && o.trans_row != N + i) {
            copy_queued.push_back(o.trans_row);
            copy_queued.push_back(N + i);
            o.trans_row = N + i;
        }
    }
    const int n_chains = (int)chain_ptr_.size();
    chain_ptr_.push_back((int)prof_rows_.size());
    last_plan_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const auto tsub0 = std::chrono::steady_clock::now();

    const int n_frows = n_frows_;
    last_frows_ = n_frows;
    last_trows_ = (int64_t)ffat_.size();
    // Latency path: a short single-launch step submitted while the device is idle (the real-time facade: step, wait, step) has
    // nothing to run beside -- its preparation goes on the bank's own stream, and the 11 us an event takes to hand a launch
    // from one stream to the other (profiles/r04_stream_sync.txt) are saved.  Ordering only: same kernels, same arguments.
    bool one_stream = false;
    if (latency_path_ && nb == nb_total && nb <= 4 && !defer) {      // (the query below needs the previous launch's events recorded)
        one_stream = last_set_ < 0 || hipEventQuery(ev_k1_done_[last_set_]) == hipSuccess;      // (the previous bank, hence its preparation)
        (void)hipGetLastError();
    }
    // K5: a launch without (many) dense-profile buffers on a chip the scene cannot fill runs the block kernel as (team, chunk of
    // buffers) workgroups behind a scan of the buffer-start states (kernels_scan.hip)
    int tc_set = -1, tc_cb = 0;
    const bool tc_launch = is_block() && !split_always_ && choose_time_chunks(nb, n_prows_, &tc_set, &tc_cb);
    hipStream_t sk = stream_, sp = one_stream ? sk : prep_stream_;
    // (the preparation stream takes over again behind a launch that went without it: behind that launch's bank)
    if (!one_stream && last_one_stream_ && last_set_ >= 0) QHIP(hipStreamWaitEvent, sp, ev_k1_done_[last_set_], 0);
    DevBuf<float> &grows = d_grows_[cur_set_];
    // device arenas (growth drains the device first, see DevBuf::ensure)
    GROWTRY(d_slots_, std::max<size_t>(1, n_slots_.load()) * m_pad_, true, sp);
    GROWTRY(grows, std::max<size_t>(1, (size_t)n_frows) * m_pad_, false, sp);
    const bool qn = desc_.qnorm_mode != PBSO_QNORM_OFF;

    {
        int hrc = harvest_timing(ev_pending_.size() >= 256);      // (what has finished; everything when too many events are live)
        if (hrc != PBSO_OK) return hrc;
    }
    EvQuad evq;
    if (!ev_free_.empty()) {
        evq = ev_free_.back();
        ev_free_.pop_back();
    } else {
        HIPTRY(hipEventCreate(&evq.k0));
        HIPTRY(hipEventCreate(&evq.k1));
        HIPTRY(hipEventCreate(&evq.p0));
        HIPTRY(hipEventCreate(&evq.p1));
        HIPTRY(hipEventCreate(&evq.f0));
        HIPTRY(hipEventCreate(&evq.f1));
    }
    evq.step_id = step_id;
    evq.has_k2 = false;
    auto host_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    evq.h_enter = std::chrono::duration<double, std::milli>(t0.time_since_epoch()).count();
    // (two timing events before and three after the bank cost the stream ~15 us per launch -- 2 % of a 0.7 ms step)
    const bool timed = timing_every_ > 0 && (launch_seq_ % (unsigned)timing_every_) == 0;
    // ---- preparation stream: this set's device buffers are free once the oscillator bank that last read them (N_SETS
    //      launches ago) has finished.  (The upload stays on this stream: on one of its own the next launch's scan starts as soon
    //      as this launch's has finished -- beside the START of a bank, whose workgroups then wait for the slots it holds:
    //      512 x 512 x 86 0.51 -> 0.59 ms per step, scripts/debug/r04_sync.sh.)
    QHIP(hipStreamWaitEvent, sp, ev_k1_done_[cur_set_], 0);
    if (timed) QHIP(hipEventRecord, evq.p0, sp);
    evq.h_prep = host_ms();
    // ---- ONE upload: everything the planner produced sits behind the descriptors in the set's pinned arena.
    // (Twelve separate copies cost the host 0.1 ms of API calls per step, and the small ones went through
    // blit kernels that cannot start while the oscillator bank fills the register file.)
    std::vector<int> cp;
    const int n_cl = (int)copy_latest.size() / 2, n_cq = (int)copy_queued.size() / 2;
    {
        // layout: [src...] then [dst...]
        cp.resize((size_t)2 * (n_cl + n_cq));
        for (int i = 0; i < n_cl; ++i) { cp[i] = copy_latest[2 * i]; cp[n_cl + n_cq + i] = copy_latest[2 * i + 1]; }
        for (int i = 0; i < n_cq; ++i) { cp[n_cl + i] = copy_queued[2 * i]; cp[n_cl + n_cq + n_cl + i] = copy_queued[2 * i + 1]; }
    }
    // A launch of ONE buffer (the real-time step): the rows of explicit data and the projections whose vectors outlive the buffer are
    // taken by the combine kernel itself -- every slot filled now is read by exactly one row of this launch -- instead of a scatter and
    // a projection launch in front of it (kernels_exact.hip: force_combine_kernel).  Any slot that is NOT read exactly once (a force
    // whose profile is all zero in this buffer leaves its row out) keeps the separate launches.
    bool fuse_combine = fuse_short_ && nb == 1 && n_frows > 0 && (!stage_slot_.empty() || !proj_.empty());
    if (fuse_combine) {
        const int n_ev = (int)(proj_direct_.size() + proj_.size());
        std::unordered_map<int, std::pair<int, int>> code;      // slot -> (entry, references)
        for (size_t e = 0; e < proj_.size(); ++e) code[proj_[e].slot] = {-((int)(proj_direct_.size() + e) + 1), 0};
        for (size_t r = 0; r < stage_slot_.size(); ++r) code[stage_slot_[r]] = {-(n_ev + (int)r + 1), 0};
        fuse_combine = code.size() == proj_.size() + stage_slot_.size();      // (a slot filled twice in one launch: never, but not fused)
        for (int si : slot_idx_) {
            if (si < 0) continue;
            auto it = code.find(si);
            if (it != code.end()) ++it->second.second;
        }
        for (const auto &kv : code) fuse_combine = fuse_combine && kv.second.second == 1;
        if (fuse_combine) {
            for (int &si : slot_idx_) {
                if (si < 0) continue;
                auto it = code.find(si);
                if (it != code.end()) si = it->second.first;
            }
            proj_direct_.insert(proj_direct_.end(), proj_.begin(), proj_.end());
            proj_.clear();
        }
    }
    size_t off = ps.front_bytes;
    auto place = [&](size_t bytes) { const size_t o = off; off += arena_align(bytes); return o; };
    const size_t o_row_ptr = place(row_ptr_.size() * sizeof(int)), o_slot_idx = place(slot_idx_.size() * sizeof(int));
    const size_t o_row_obj = place(row_obj_.size() * sizeof(int));
    const size_t o_prow_obj = place(prow_obj_.size() * sizeof(int));
    const size_t o_pent = place(device_profiles_ ? prof_entries_.size() * sizeof(ProfEntry) : 0);
    const size_t o_prow = place(device_profiles_ ? prof_rows_.size() * sizeof(ProfRow) : 0);
    const size_t o_chain = place(device_profiles_ ? chain_ptr_.size() * sizeof(int) : 0);
    const size_t o_aruse = place(ar_uses_.size() * sizeof(ArUse)), o_arstream = place(ar_streams_.size() * sizeof(ArStream));
    const size_t o_arseg = place(seg_stream_.size() * sizeof(int));
    const size_t o_tprof = place(device_profiles_ ? 0 : tprof_.size() * sizeof(float));
    const size_t o_stage = place(stage_.size() * sizeof(double)), o_stage_slot = place(stage_slot_.size() * sizeof(int));
    const size_t o_proj = place(proj_.size() * sizeof(ProjectEvent)), o_projd = place(proj_direct_.size() * sizeof(ProjectEvent));
    // the listener events as runs of one object each (the planner lists them object by object): one geometry read per (run, mode)
    // (events of objects whose modes share one map geometry first: they go to the kernel that locates a position once per event)
    int n_ffat_sh = 0;
    if (n_ffat_shared_ > 0 && !ffat_.empty()) {
        auto mid = std::stable_partition(ffat_.begin(), ffat_.end(), [&](const FfatEvent &e) { return ffat_shared_h_[e.obj] != 0; });
        n_ffat_sh = (int)(mid - ffat_.begin());
    }
    const int n_ffat_gen = (int)ffat_.size() - n_ffat_sh;
    tot_ffat_shared_events_ += n_ffat_sh;
    tot_ffat_general_events_ += n_ffat_gen;
    ffat_runs_.clear();
    for (size_t i = (size_t)n_ffat_sh; i < ffat_.size(); ++i) {
        if (ffat_runs_.empty() || ffat_runs_.back().obj != ffat_[i].obj) ffat_runs_.push_back(FfatRun{ffat_[i].obj, (int)i, 0, 0});
        ffat_runs_.back().count += 1;
    }
    const size_t o_ffat = place(ffat_.size() * sizeof(FfatEvent)), o_copy = place(cp.size() * sizeof(int));
    const size_t o_ffat_runs = place(ffat_runs_.size() * sizeof(FfatRun));
    HIPTRY(ps.h_arena.ensure_keep(off, ps.front_bytes));
    plan_desc_ = reinterpret_cast<BufDesc *>(ps.h_arena.p);      // (the arena may have moved)
    ps.last_bytes = off;
    unsigned char *ha = ps.h_arena.p;
    auto put = [&](size_t o, const void *src, size_t bytes) { if (bytes) std::memcpy(ha + o, src, bytes); };
    put(o_row_ptr, row_ptr_.data(), row_ptr_.size() * sizeof(int));
    put(o_slot_idx, slot_idx_.data(), slot_idx_.size() * sizeof(int));
    put(o_row_obj, row_obj_.data(), row_obj_.size() * sizeof(int));
    put(o_prow_obj, prow_obj_.data(), prow_obj_.size() * sizeof(int));
    if (device_profiles_) {
        put(o_pent, prof_entries_.data(), prof_entries_.size() * sizeof(ProfEntry));
        put(o_prow, prof_rows_.data(), prof_rows_.size() * sizeof(ProfRow));
        put(o_chain, chain_ptr_.data(), chain_ptr_.size() * sizeof(int));
        put(o_aruse, ar_uses_.data(), ar_uses_.size() * sizeof(ArUse));
        put(o_arstream, ar_streams_.data(), ar_streams_.size() * sizeof(ArStream));
        put(o_arseg, seg_stream_.data(), seg_stream_.size() * sizeof(int));
        if (k2_rows_launch_) {
            const size_t ns = std::max<size_t>(1, ar_streams_.size()), nu = std::max<size_t>(1, ar_uses_.size());
            const size_t ng = std::max<size_t>(1, seg_stream_.size());
            GROWTRY(d_ar_snaps_, ns, false, sp);
            GROWTRY(d_ar_fins_, ns, false, sp);
            GROWTRY(d_ar_recs_, nu, false, sp);
            GROWTRY(d_ar_cbuf_, nu * (size_t)b_pad_, false, sp);
            GROWTRY(d_ar_vnorm_, ng * 2 * K2_SEG, false, sp);
            GROWTRY(d_ar_vstate_, ng * K2_SEG, false, sp);
            GROWTRY(d_ar_segcount_, ng, false, sp);
        }
        GROWTRY(ps.d_tprof, std::max<size_t>(1, (size_t)n_prows_) * b_pad_, false, sp);
        GROWTRY(d_arstate_, std::max<size_t>(1, n_ar_states_.load()), true, sp);
    } else {
        put(o_tprof, tprof_.data(), tprof_.size() * sizeof(float));
    }
    put(o_stage, stage_.data(), stage_.size() * sizeof(double));
    put(o_stage_slot, stage_slot_.data(), stage_slot_.size() * sizeof(int));
    put(o_proj, proj_.data(), proj_.size() * sizeof(ProjectEvent));
    put(o_projd, proj_direct_.data(), proj_direct_.size() * sizeof(ProjectEvent));
    put(o_ffat, ffat_.data(), ffat_.size() * sizeof(FfatEvent));
    put(o_ffat_runs, ffat_runs_.data(), ffat_runs_.size() * sizeof(FfatRun));
    put(o_copy, cp.data(), cp.size() * sizeof(int));
    GROWTRY(ps.d_arena, off, false, sp);
    QHIP(hipMemcpyAsync, ps.d_arena.p, ha, off, hipMemcpyHostToDevice, sp);
    QHIP(hipEventRecord, ev_set_[cur_set_], sp);          // this set's pinned arena is reusable
    // the start gate: this launch's preparation kernels behind the START of the previous launch's bank (engine.h).  By policy
    // for LONG launches only: there the gate's few microseconds are nothing, and what it prevents is expensive -- a scan of 860
    // buffers that the preparation stream runs TWO launches ahead trails the bank it was squeezed beside by 0.2 ms, holds LDS
    // while the next bank starts, and that bank's left-over workgroups wait a whole workgroup's length (512 x 512 x 860: every
    // other launch 9 instead of 4.6 ms in one run of four).  Gated, a scan is never more than one launch ahead, and the bank
    // that needs it waits for it.
    // (stream_sync = 4, round 5: the gate as a wait of the SUBMITTING thread on a word of pinned host memory the previous bank's
    //  first workgroup writes.  Built because hipStreamWaitValue64 runs on this stack as a kernel that waits -- `__amd_rocclr_streamOpsWait`
    //  sits 1.1 of a 1.28 ms bank until workgroups retire, scripts/debug/r05_timeline_share.sh -- and was suspected behind the share's
    //  outlier runs; those came from the planner's helper threads (PlanPool::run), both forms show them alike, and the host form
    //  costs the host its second launch of run-ahead: an option, not the policy.)
    const bool gate_now = !one_stream && last_bank_seq_ > 0 && (desc_.stream_sync == 3 || desc_.stream_sync == 4 || nb >= 256);
    host_gate_used_ = false;
    if (gate_now && host_start_ && desc_.stream_sync == 4) {
        const auto tg0 = std::chrono::steady_clock::now();
        volatile unsigned long long *w = host_start_;
        while (*w < last_bank_seq_) {
            std::this_thread::yield();
            // (fail-safe: a bank that never starts -- a device fault -- must not hang the caller; the launch then goes ungated)
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count() > 2.0) {
                ++gate_timeouts_;                         // (pbso_engine_info::total_gate_timeouts)
                break;
            }
        }
        host_gate_used_ = true;
    } else if (gate_now && start_gate_) {
        QHIP(hipStreamWaitValue64, sp, sig_start_, last_bank_seq_, hipStreamWaitValueGte, ~0ull);
    }
    evq.h_copy = host_ms();
    const auto tsub1 = std::chrono::steady_clock::now();
    unsigned char *da = ps.d_arena.p;
    const BufDesc *d_desc = reinterpret_cast<const BufDesc *>(da);
    const int *d_xfer_init = reinterpret_cast<const int *>(da + ps.off_xfer_init);
    const int *d_row_ptr = reinterpret_cast<const int *>(da + o_row_ptr), *d_slot_idx = reinterpret_cast<const int *>(da + o_slot_idx);
    const int *d_row_obj = reinterpret_cast<const int *>(da + o_row_obj), *d_chain = reinterpret_cast<const int *>(da + o_chain);
    const ProfEntry *d_pent = reinterpret_cast<const ProfEntry *>(da + o_pent);
    const ProfRow *d_prow = reinterpret_cast<const ProfRow *>(da + o_prow);
    const float *d_tprof = device_profiles_ ? ps.d_tprof.p : reinterpret_cast<const float *>(da + o_tprof);
    const double *d_stage = reinterpret_cast<const double *>(da + o_stage);
    const int *d_stage_slot = reinterpret_cast<const int *>(da + o_stage_slot), *d_copy = reinterpret_cast<const int *>(da + o_copy);
    const ProjectEvent *d_proj = reinterpret_cast<const ProjectEvent *>(da + o_proj), *d_projd = reinterpret_cast<const ProjectEvent *>(da + o_projd);
    const FfatEvent *d_ffat = reinterpret_cast<const FfatEvent *>(da + o_ffat);

    const bool dense_heavy = is_block() && dense_to_k1_ && (long long)n_prows_ * 2 > (long long)N * nb;
    IirParams kp;
    kp.ca = d_ca_.p; kp.cb = d_cb_.p; kp.sq = d_sq_.p; kp.sd = d_sd_.p; kp.ss = d_ss_.p;
    kp.desc = d_desc;
    kp.grows = grows.p;
    kp.g32 = d_g32_.p;
    kp.g32_off = d_g32_off_.p;
    kp.tprof = d_tprof;
    kp.xfer_rows = d_xfer_.p;
    kp.xfer_init = d_xfer_init;
    kp.audio = audio + (size_t)b0 * B_;
    kp.qnorm = (qn || n_dump_ > 0) ? d_qnorm_.p : nullptr;
    kp.xdump = n_dump_ > 0 ? d_xdump_.p : nullptr;
    kp.xscale = d_xscale_.p;
    kp.dump_row = d_dump_row_.p;
    kp.qn_nb = nb_total;
    kp.qn_b0 = b0;
    kp.gq = d_gq_.p;
    kp.gq_plane = (long long)N * m_pad_;
    kp.census = nullptr;
    if (census_) {
        size_t rows = (size_t)std::max(n_teams_, use_split() ? n_ts_teams_ : 0);
        if (tc_ok_) for (const TcSet &ts : tc_) rows = std::max(rows, (size_t)ts.n_teams * nb);
        GROWTRY(d_census_, rows * CENSUS_WORDS, false, sk);
        kp.census = d_census_.p;
    }
    kp.nb = nb; kp.n_tiles = n_tiles_; kp.m_pad = m_pad_; kp.b_pad = b_pad_;
    kp.audio_stride = (long long)nb_total * B_;
    // the per-CU progress feedback only pays when several teams compete for every SIMD
    kp.rotate_prio = (rotate_prio_ == 2 && total_team_waves_ < 12LL * n_cus_) ? 1 : rotate_prio_;
    kp.board = d_board_.p;
    kp.launch_seq = ++launch_seq_;
    kp.start_flag = (host_start_ && desc_.stream_sync == 4) ? host_start_dev_ : (start_gate_ ? sig_start_ : nullptr);
    kp.start_seq = ++bank_seq_;
    kp.pc = d_pc_.p;
    kp.wtab = d_wtab_.p;
    kp.frames = B_;
    kp.ftab = ftab_forced_ ? d_ftab_.p : nullptr;
    kp.forced_block = (forced_block_ && n_prows_ > 0) ? 1 : 0;      // (the build with the forced block path only when a buffer needs it)
    // K2 -> time-profile rows ; K3 / scatter -> data slots ; K4 -> transfer rows ; combine -> g rows
    // Round 5: a launch with many dense-profile rows FORKS its preparation.  The force profiles (variates -> zero-state chains ->
    // rows) and the dense increments that need them are one dependent chain, projection -> combine another, and nothing connects
    // the two before the scan; in one stream they ran one after the other BESIDE the previous bank -- 8 x 4096 x 86 sustained
    // scraping with qnorm rows: 31 + 54 + 73 (profiles) + 29 + 15 (project, combine) + 78 (increments) = 280 us beside a bank of
    // 283, then the scan: the preparation, not the bank, set the step (scripts/debug/r05_timeline_c5.sh).  Forked, projection +
    // FFAT + combine run on aux_stream_ behind the upload and join the preparation stream in front of the scan (or of the
    // hand-over to the bank): every later reader is ordered behind prep_stream_ as before.
    const bool split_prep = !one_stream && aux_stream_ && prep_split_ > 0 && device_profiles_ && n_prows_ > 0 &&
                            (prep_split_ == 2 || n_prows_ >= 64) && (n_frows > 0 || !proj_.empty() || !ffat_.empty() || !stage_slot_.empty());
    hipStream_t sa = split_prep ? aux_stream_ : sp;
    if (split_prep) {
        tot_prep_splits_ += 1;
        QHIP(hipEventRecord, ev_aux_fork_[cur_set_], sp);         // (behind the upload and, for gated launches, the start gate)
        QHIP(hipStreamWaitEvent, sa, ev_aux_fork_[cur_set_], 0);
    }
    if (device_profiles_ && timed && n_chains > 0) { QHIP(hipEventRecord, evq.f0, sp); evq.has_k2 = true; }
    // (round 6, second half) a one-buffer launch whose profile rows go through the fused kernel and whose combine takes its rows itself:
    // both in ONE launch -- they touch different arrays (kernels_exact.hip, force_rows_combine_kernel)
    const bool rows_and_combine = device_profiles_ && k2_rows_launch_ && fuse_combine && sa == sp && !prof_rows_.empty() && n_frows > 0 &&
                                  fuse_short_ && !ar_uses_.empty() && ar_uses_.size() == ar_streams_.size();
    if (rows_and_combine)
        QLAUNCH(launch_force_rows_combine, d_prow, (int)prof_rows_.size(), d_pent, reinterpret_cast<const ArUse *>(da + o_aruse),
                                            reinterpret_cast<const ArStream *>(da + o_arstream), ar_max_segs_, d_arstate_.p, d_ar_snaps_.p,
                                            d_ar_vnorm_.p, d_ar_vstate_.p, d_ar_segcount_.p, d_ar_cbuf_.p, d_ar_recs_.p, d_ar_fins_.p,
                                            ps.d_tprof.p, B_, b_pad_, b_pad_, d_row_ptr, d_slot_idx, d_row_obj, n_frows, d_slots_.p, d_c3_.p,
                                            grows.p, d_projd, d_shapes_.p, d_shape_off_.p, d_n_modes_.p, m_pad_, (int)proj_direct_.size(),
                                            d_stage, d_stage_slot, sp);
    else if (device_profiles_ && k2_rows_launch_)
        QLAUNCH(launch_force_rows, d_prow, (int)prof_rows_.size(), d_pent, reinterpret_cast<const ArUse *>(da + o_aruse), (int)ar_uses_.size(),
                                    reinterpret_cast<const ArStream *>(da + o_arstream), reinterpret_cast<const int *>(da + o_arseg),
                                    (int)seg_stream_.size(), ar_max_segs_, d_arstate_.p, d_ar_snaps_.p, d_ar_vnorm_.p, d_ar_vstate_.p,
                                    d_ar_segcount_.p, d_ar_cbuf_.p, d_ar_recs_.p, d_ar_fins_.p, ps.d_tprof.p, B_, b_pad_, b_pad_,
                                    /* every AR force adds its samples once (a launch of one buffer): one launch instead of three */
                                    fuse_short_ && !ar_uses_.empty() && ar_uses_.size() == ar_streams_.size(), sp);
    else if (device_profiles_)
        QLAUNCH(launch_force_profiles, d_chain, n_chains, d_prow, d_pent, d_arstate_.p, ps.d_tprof.p, B_, b_pad_, ar_serial_ ? 1 : 0, k2_prio_, sp);
    if (evq.has_k2) QHIP(hipEventRecord, evq.f1, sp);
    if (!fuse_combine) QLAUNCH(launch_scatter_rows, d_stage, d_stage_slot, (int)stage_slot_.size(), d_slots_.p, m_pad_, sa);
    QLAUNCH(launch_modal_project, d_proj, (int)proj_.size(), d_shapes_.p, d_shape_off_.p, d_n_modes_.p, d_slots_.p, m_pad_, sa);
    // (a few events: one thread per (event, mode); listener paths -- many events per object -- by runs)
    if (n_ffat_sh > 0)
        QLAUNCH(launch_ffat_lookup_shared, d_ffat, n_ffat_sh, d_ffat_shared_.p, d_geom_.p, d_geom_off_.p, d_n_modes_.p, d_ffat_k_.p,
                                            d_ffat_valid_.p, d_psi_t_.p, d_xfer_.p, m_pad_, sa);
    if ((size_t)n_ffat_gen >= 4 * ffat_runs_.size() && !ffat_runs_.empty())
        QLAUNCH(launch_ffat_lookup_runs, d_ffat, reinterpret_cast<const FfatRun *>(da + o_ffat_runs), (int)ffat_runs_.size(), d_geom_.p,
                                          d_geom_off_.p, d_n_modes_.p, d_psi_.p, d_xfer_.p, m_pad_, sa);
    else
        QLAUNCH(launch_ffat_lookup, d_ffat + n_ffat_sh, n_ffat_gen, d_geom_.p, d_geom_off_.p, d_n_modes_.p, d_psi_.p, d_xfer_.p, m_pad_, sa);
    // (the combine stays on the preparation stream even when the bank fills the register file: its workgroups
    //  move in as the bank's retire, so most of it is done when the last one leaves -- queued behind the bank in
    //  the bank's own stream it would start only then: 0.785 instead of 0.735 ms per step at 1024 x 512)
    if (!rows_and_combine)
        QLAUNCH(launch_force_combine, d_row_ptr, d_slot_idx, d_row_obj, n_frows, d_slots_.p, d_c3_.p, grows.p, d_projd, d_shapes_.p,
                                       d_shape_off_.p, d_n_modes_.p, m_pad_, (int)proj_direct_.size(), d_stage, d_stage_slot, sa);
    if (split_prep) QHIP(hipEventRecord, ev_aux_join_[cur_set_], sa);
    if (split_prep && !tc_launch) QHIP(hipStreamWaitEvent, sp, ev_aux_join_[cur_set_], 0);
    if (tc_launch) {
        // The scan hands the state from launch to launch by itself (the chunked bank launches never write it), so it runs HERE, on
        // the preparation stream, beside the previous launch's oscillator bank -- behind a launch of another kind it waits for
        // that launch's bank, which wrote the state it starts from.
        const int n_chunks = (nb + tc_cb - 1) / tc_cb;
        GROWTRY(d_xs_[cur_set_], (size_t)N * n_chunks * m_pad_ * 2, false, sp);
        GROWTRY(d_xtrow_[cur_set_], (size_t)N * n_chunks, false, sp);
        // dense-profile buffers (Gaussian / AR, forces.h:92-128): what each leaves in the state per unit gain, all of them at once
        const float *vinc = nullptr;
        if (n_prows_ > 0) {
            GROWTRY(d_vinc_[cur_set_], (size_t)n_prows_ * m_pad_ * 2, false, sp);
            QLAUNCH(launch_dense_increments, d_pc_.p, d_ftab_.p, (long long)N * m_pad_, d_tprof, reinterpret_cast<const int *>(da + o_prow_obj),
                                              d_n_modes_.p, n_prows_, m_pad_, b_pad_, B_, d_vinc_[cur_set_].p, sp);
            vinc = d_vinc_[cur_set_].p;
            tot_tc_dense_launches_ += 1;
        }
        if (!last_launch_tc_ && last_set_ >= 0) QHIP(hipStreamWaitEvent, sp, ev_k1_done_[last_set_], 0);
        if (split_prep) QHIP(hipStreamWaitEvent, sp, ev_aux_join_[cur_set_], 0);      // (the increments above did not need the gains; the scan does)
        // Cut along the time axis itself -- one wave per chunk (kernels_scan.hip, SEG) -- in two cases.  (a) The whole scan is a
        // handful of waves: it halves the latency of a scan that has the device to itself (1 x 512 x 86: 11.3 -> 5.7 us).  (b) LONG
        // chunks of a scene whose serial scan is at most two waves per SIMD: that scan can only start as the previous bank's
        // workgroups retire (the bank holds every register), so the waves placed last start when the bank ends and what the next
        // bank waits for is ONE WAVE'S walk over all buffers of the launch -- 82 us for 860 buffers; cut into the bank's own
        // chunks a wave walks 108 and the whole scan is 45 us of a throughput-bound kernel, most of it beside the bank's tail:
        // 128 x 512 x 860 1.31 -> 1.27 ms per step, 256 x 512 x 860 2.45 -> 2.42 (scripts/debug/r05_scan_seg2.sh).  NOT for
        // four serial waves per SIMD (512 x 512: the serial scan is throughput-bound already and does less work: 4.79 against
        // 4.87 ms) and not for short chunks (86-buffer steps, 11 buffers per chunk: 0.156 against 0.160 ms -- eight times the
        // waves for one batch each).  (c) A DENSE scan (every buffer a hit with a row of increments: 64 hits per serial batch,
        // sixteen in flight) of at most two waves per CU: 8 x 4096 x 86 sustained scraping with qnorm rows 0.320 -> 0.312 ms per
        // step (scripts/debug/r05_seg_short.sh).  One buffer per chunk keeps the serial scan, whose arithmetic does not depend
        // on where a step is cut.
        const bool seg_fits = n_chunks >= 2 && n_chunks <= SCAN_SEG_MAX;
        const long long scan_waves = (long long)N * (m_pad_ / 64);
        const bool seg = seg_fits && desc_.scan_kernel != 1 &&
                         (desc_.scan_kernel == 2 || (tc_cb > 1 && scan_waves * n_chunks <= 2LL * n_cus_) ||
                          (tc_cb >= 32 && scan_waves <= 8LL * n_cus_) || (tc_cb > 1 && vinc && scan_waves <= 2LL * n_cus_));
        if (seg) tot_seg_scans_ += 1;
        QLAUNCH(launch_iir_scan, kp, N, d_scan_.p, tc_cb, n_chunks, d_xs_[cur_set_].p, d_xtrow_[cur_set_].p, direct_hits_, vinc, seg, sp);
    }
    // ---- compute stream: the bank after its preparation (and, stream order, after the previous bank)
    if (one_stream) {
        tot_one_stream_launches_ += 1;
    } else if (sync_values_ && (desc_.stream_sync == 2 || nb < 256)) {
        // (round 6: by policy for launches of fewer than 256 buffers -- a step of 86 buffers: 64 x 256 with a listener move per buffer
        //  0.0765 -> 0.0717 ms, 16 / 128 / 256 / 512 x 512 impulses 4 / 2.7 / 1.7 / 1.5 % per step, 64 x 512 and 1 x 512 unchanged,
        //  profiles/r06_value_handover.txt; long launches keep the event: nothing to gain there, and their numbers stand as measured)
        QLAUNCH(launch_signal_value, sig_prep_, ++prep_seq_, sp);
        QHIP(hipStreamWaitValue64, sk, sig_prep_, prep_seq_, hipStreamWaitValueGte, ~0ull);
    } else {
        QHIP(hipEventRecord, ev_prep_done_[cur_set_], sp);
        QHIP(hipStreamWaitEvent, sk, ev_prep_done_[cur_set_], 0);
    }
    const auto tsub2 = std::chrono::steady_clock::now();
    if (timed) QHIP(hipEventRecord, evq.k0, sk);
    evq.h_bank = host_ms();
    kp.audio_parts = d_audio_parts_.p ? d_audio_parts_.p + (size_t)b0 * B_ : nullptr;
    // Side by side only while everything is resident at once (largest teams first, on the engine's
    // stream); an engine that needs several rounds of workgroups runs its classes one after the other
    // (512 x 512 + 4096 x 64: 3.1 ms in sequence, 3.7 ms side by side -- teams of different size fragment
    // the CUs' LDS and wave slots).
    // Block form: a launch in which most (object, buffer) pairs carry a dense force profile -- sustained
    // scraping, forces.h:107-128 -- has nothing for the matrix pipe to do (every sample is forced) and runs on the
    // per-sample kernel K1, whose inner loop is built for exactly that; state layout, teams and scaled-state
    // rules are shared, so the two kernels hand over at any launch boundary.  (Audio then is bit-identical
    // across different cuts of a step only while both cuts pick the same kernel; always within tolerance.)
    // PBSO_DENSE_LAUNCHES=block keeps every launch on the block kernel (bit-identical audio for any cut of a step).
    // The pipeline kernel of under-filled engines (K1p) takes the launch -- also one whose (object, buffer)
    // pairs mostly carry a dense force profile (sustained scraping) when the profiles come from K2's row-parallel form
    // (8 x 4096 x 86: 0.30 ms against K1b's 0.43 without qnorm rows, 0.48 against 0.74 with).  Beside K2's CHAIN form -- then
    // the step's critical chain, which a wave on every SIMD slows 2.7 x (1.28 ms against 0.48 for 8 chains of 86 rows beside
    // K1b's 512 waves) -- K1b's forced block path keeps those launches (state and teams hand over at any launch boundary).
    // Without the F table of the forced block path (PBSO_FORCED_BLOCK=0) and without qnorm rows they go to K1b as well.
    const bool dense_majority = (long long)n_prows_ * 2 > (long long)N * nb;
    const bool split_dense_ok = k2_rows_launch_ && (desc_.qnorm_mode != PBSO_QNORM_OFF || ftab_forced_);
    const bool split_launch = !tc_launch && use_split() && (split_always_ || !dense_majority || split_dense_ok);      // (PBSO_SPLIT=2: always)
    (tc_launch || split_launch || !(dense_heavy || !is_block()) ? tot_block_launches_ : tot_sample_launches_) += 1;
    if (split_launch) tot_split_launches_ += 1;
    if (tc_launch) {
        tot_tc_launches_ += 1;
        last_tc_shape_ = tc_[tc_set].R;
        last_tc_cb_ = tc_cb;
        last_tc_teams_ = tc_[tc_set].n_teams;
    }
    if (n_dump_ > 0) {
        // the mix needs block states: a launch on the per-sample kernel leaves none, a dense-profile buffer neither
        for (int i = 0; i < N; ++i) {
            if (dump_row_[i] < 0) continue;
            if (!tc_launch && (dense_heavy || !is_block())) { dump_valid_[i] = 0; continue; }
            for (int b = 0; b < nb; ++b)
                if (!(plan_desc_[(size_t)i * nb + b].flags & DESC_DIRECT) && plan_desc_[(size_t)i * nb + b].prow >= 0) { dump_valid_[i] = 0; break; }
        }
    }
    bool used[N_CLASS_STREAMS] = {false, false, false};
    if (tc_launch) {
        TcSet &ts = tc_[tc_set];
        kp.tc_cb = tc_cb;
        if (const char *v = std::getenv("PBSO_TC_LDS_PAD")) kp.lds_pad = std::atoi(v);      // (diagnostics: scripts/debug/r06_c5_occupancy.sh)
        kp.tc_xs = d_xs_[cur_set_].p;
        kp.tc_xtrow = d_xtrow_[cur_set_].p;
        kp.census_stride = ts.n_teams;
        // The priority rotation (kernels_block.hip) assumes the teams of a CU walk their buffers in step.  A launch of several
        // ROUNDS of workgroups does not: a team that arrives when another retires starts at its own chunk's first buffer, two
        // teams of a CU then hold the same priority and the oldest-first arbitration is back -- 1024 x 512 forced into two
        // chunks: 1.33 ms against 1.07 without the rotation; 700 x 512: 0.751 against 0.742 (scripts/debug/r04_tcrounds.py).
        // (A phase taken from the wall clock instead of the team's buffer count does not help: 0.752 ms at 700 x 512.)
        kp.rotate_prio = (rotate_prio_ && ts.waves * (long long)((nb + tc_cb - 1) / tc_cb) <= 8LL * n_cus_) ? 1 : 0;
        for (const SizeClass &c : ts.classes) {
            kp.teams = ts.d_teams.p + c.first;
            QLAUNCH(iir_block::launch_iir_block, kp, c.count, ts.R, c.W, desc_.qnorm_mode, form_ == PBSO_FORM_BLOCK_BF16 ? 1 : 0, sk);
        }
    }
    if (split_launch) {
        kp.teams = d_ts_teams_.p;
        {
            // two consumers (one group each) also where that puts a second wave on some SIMDs: 8 x 4096 scraping -- 512 teams, 1536
            // waves -- runs 3120 x against 2800 with one consumer (no qnorm rows) and 1700 x against 1590 (with them)
            int nc = 2;
            // qnorm rows of a mostly-dense launch: the consumers re-step every sample for the sums and are the long stage -- a third
            // consumer wave that only steps (half of the chains) when that still leaves at most two waves per SIMD
            if (dense_majority && desc_.qnorm_mode != PBSO_QNORM_OFF && ftab_forced_ && 4LL * n_ts_teams_ <= 8LL * n_cus_) nc = 3;
            // round 5: a mostly-dense launch as teams of FIVE -- the block increments F . T_n come from two waves of their own, a buffer
            // ahead of the producer, which is left with the 32 coarse steps (kernels_pipe.hip, iir_pipe5_kernel)
            if (dense_majority && ftab_forced_ && desc_.qnorm_mode == PBSO_QNORM_OFF) nc = 4;
            if (desc_.pipe_consumers > 0) nc = desc_.pipe_consumers;
            QLAUNCH(iir_pipe::launch_iir_pipe, kp, n_ts_teams_, nc, desc_.qnorm_mode, sk);
        }
    }
    const bool fork = !tc_launch && !split_launch && classes_.size() > 1 && ev_fork_ && total_team_waves_ <= 16LL * n_cus_;
    if (fork) QHIP(hipEventRecord, ev_fork_, sk);
    for (size_t ci = 0; ci < (tc_launch || split_launch ? 0 : classes_.size()); ++ci) {
        const SizeClass &c = classes_[ci];
        hipStream_t s = sk;
        if (fork && ci > 0) {
            const int j = (int)((ci - 1) % N_CLASS_STREAMS);
            s = class_stream_[j];
            if (!used[j]) {
                QHIP(hipStreamWaitEvent, s, ev_fork_, 0);
                used[j] = true;
            }
        }
        kp.teams = d_teams_.p + c.first;
        if (is_block() && !dense_heavy)
            QLAUNCH(iir_block::launch_iir_block, kp, c.count, R_, c.W, desc_.qnorm_mode, form_ == PBSO_FORM_BLOCK_BF16 ? 1 : 0, s);
        else
            QLAUNCH(iir_scalar::launch_iir_bank, kp, c.count, R_, c.W, form_ == PBSO_FORM_DIRECT ? 1 : 0, desc_.qnorm_mode, s);
    }
    for (int j = 0; j < N_CLASS_STREAMS; ++j) {
        if (!used[j]) continue;
        QHIP(hipEventRecord, ev_join_[j], class_stream_[j]);
        QHIP(hipStreamWaitEvent, sk, ev_join_[j], 0);
    }
    // objects stepped by several teams: the teams' partial sums of this launch's buffers, added in team order
    // ... and _latest_transfer = trans (modal_solver.h:251) in the same launch; then re-park a still-queued transfer
    QLAUNCH(launch_sum_parts_copy_rows, tc_launch ? tc_[tc_set].d_split.p : split_launch ? d_ts_split_.p : d_split_.p,
                                         tc_launch ? tc_[tc_set].n_split : split_launch ? n_ts_split_ : n_split_, kp.audio_parts,
                                         audio + (size_t)b0 * B_, (long long)nb_total * B_, (long long)nb * B_, d_copy, d_copy + (n_cl + n_cq), n_cl,
                                         d_xfer_.p, m_pad_, sk);
    if (timed) QHIP(hipEventRecord, evq.k1, sk);
    QLAUNCH(launch_copy_rows, d_copy + n_cl, d_copy + (n_cl + n_cq) + n_cl, n_cq, d_xfer_.p, m_pad_, sk);
    if (timed) QHIP(hipEventRecord, evq.p1, sk);
    evq.h_done = host_ms();
    QHIP(hipEventRecord, ev_k1_done_[cur_set_], sk);
    if (defer) {
        evq.batch = submit_->pushed() + 1;
        set_batch_[cur_set_] = submit_->push(std::move(ops));     // (from here on the worker makes the calls; the caller goes on to plan)
    }
    if (timed) ev_pending_.push_back(evq);
    else ev_free_.push_back(evq);
    buffers_done_ += nb;
    last_launch_tc_ = tc_launch;
    last_one_stream_ = one_stream;
    last_bank_seq_ = kp.start_seq;
    last_set_ = cur_set_;
    cur_set_ = (cur_set_ + 1) % N_SETS;
    hprof_[4] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - last_plan_ms_;
    hprof_[6] += std::chrono::duration<double, std::milli>(tsub1 - tsub0).count();
    hprof_[7] += std::chrono::duration<double, std::milli>(tsub2 - tsub1).count();
    hprof_[8] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tsub2).count();
    hprof_[5] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
    return PBSO_OK;
}

// the second submitting thread has made every recorded call (NOT a device synchronisation); a call that failed there surfaces here
int Engine::drain_submit() {
    if (!submit_) return PBSO_OK;
    submit_->drain();
    std::string why;
    if (submit_->error(&why)) {
        failed_ = true;
        failed_why_ = why;
        return fail(PBSO_ERR_HIP, "a launch failed on the submitting thread: " + why);
    }
    return PBSO_OK;
}

int Engine::sync() {
    {
        int drc = drain_submit();
        if (drc != PBSO_OK) return drc;
    }
    HIPTRY(hipSetDevice(desc_.device));      // the caller's thread may have another device current
    if (aux_stream_) HIPTRY(hipStreamSynchronize(aux_stream_));
    if (prep_stream_) HIPTRY(hipStreamSynchronize(prep_stream_));
    if (stream_) HIPTRY(hipStreamSynchronize(stream_));
    if (copy_stream_) HIPTRY(hipStreamSynchronize(copy_stream_));
    for (hipStream_t cs : class_stream_)
        if (cs) HIPTRY(hipStreamSynchronize(cs));
    free_retired_blocks();                   // nothing of this engine is in flight any more
    return PBSO_OK;
}

int Engine::read_audio(float *out, size_t n) {
    { int src = sync(); if (src != PBSO_OK) return src; }      // (results come from two streams: the bank's, and the scan's state on the preparation stream)
    if (!last_audio_) return fail(PBSO_ERR_STATE, "no step yet");
    const size_t total = (size_t)objs_.size() * last_nb_ * B_;
    if (n != total) return fail(PBSO_ERR_INVALID, "read_audio size mismatch");
    HIPTRY(hipMemcpyAsync(out, last_audio_, total * sizeof(float), hipMemcpyDeviceToHost, stream_));
    return sync();
}

// some objects' rows of the last step's audio: out[n_rows][n_buffers * B]
int Engine::read_audio_rows(const int *rows, int n_rows, float *out) {
    if (!last_audio_) return fail(PBSO_ERR_STATE, "no step yet");
    if (n_rows < 0 || (n_rows > 0 && (!rows || !out))) return fail(PBSO_ERR_INVALID, "read_audio_rows arguments");
    { int src = sync(); if (src != PBSO_OK) return src; }
    const size_t row = (size_t)last_nb_ * B_;
    for (int i = 0; i < n_rows; ++i) {
        if (!valid_obj(rows[i])) return fail(PBSO_ERR_INVALID, "read_audio_rows: object id out of range");
        HIPTRY(hipMemcpy(out + (size_t)i * row, last_audio_ + (size_t)rows[i] * row, row * sizeof(float), hipMemcpyDeviceToHost));
    }
    return PBSO_OK;
}

int Engine::read_census(unsigned long long *out, size_t n) {
    { int src = sync(); if (src != PBSO_OK) return src; }      // (results come from two streams: the bank's, and the scan's state on the preparation stream)
    if (!census_ || !d_census_.p) return fail(PBSO_ERR_STATE, "census not enabled (PBSO_CENSUS=1) or no step yet");
    // (n_teams rows; for the pipeline kernel its own table's rows; for a time-chunked launch (teams of its shape) x (chunks) rows,
    //  chunk-major: any whole number of rows the launch can have written)
    if (n == 0 || n % CENSUS_WORDS || n > d_census_.cap)
        return fail(PBSO_ERR_INVALID, "read_census size mismatch (12 words per team, at most the rows of the last launch)");
    HIPTRY(hipMemcpyAsync(out, d_census_.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream_));
    return sync();
}

int Engine::read_emitted(unsigned char *out, size_t n) {
    if (n != emitted_.size()) return fail(PBSO_ERR_INVALID, "read_emitted size mismatch");
    std::memcpy(out, emitted_.data(), n);
    return PBSO_OK;
}

int Engine::read_qnorm(int obj, int buffer, float *out, int n) {
    { int src = sync(); if (src != PBSO_OK) return src; }      // (results come from two streams: the bank's, and the scan's state on the preparation stream)
    if (desc_.qnorm_mode == PBSO_QNORM_OFF) return fail(PBSO_ERR_STATE, "qnorm_mode is OFF");
    if (!valid_obj(obj) || buffer < 0 || buffer >= last_nb_ || n < 0 || n > m_pad_)
        return fail(PBSO_ERR_INVALID, "read_qnorm arguments");
    HIPTRY(hipMemcpyAsync(out, d_qnorm_.p + ((size_t)obj * last_nb_ + buffer) * m_pad_, (size_t)n * sizeof(float),
                          hipMemcpyDeviceToHost, stream_));
    return sync();
}

int Engine::read_state(int obj, double *q1, double *q2, int n) {
    { int src = sync(); if (src != PBSO_OK) return src; }      // (results come from two streams: the bank's, and the scan's state on the preparation stream)
    if (!finalized_) return fail(PBSO_ERR_STATE, "read_state before finalize");
    if (!valid_obj(obj) || n < 0 || n > m_pad_) return fail(PBSO_ERR_INVALID, "read_state arguments");
    // the arrays hold (scale x state) and the scale (kernels_iir.hip, "scaled state")
    std::vector<float> a(n), b(n), sc(n);
    HIPTRY(hipMemcpyAsync(a.data(), d_sq_.p + (size_t)obj * m_pad_, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
    HIPTRY(hipMemcpyAsync(b.data(), d_sd_.p + (size_t)obj * m_pad_, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
    HIPTRY(hipMemcpyAsync(sc.data(), d_ss_.p + (size_t)obj * m_pad_, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
    int rc = sync();
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        const double qa = (double)a[i] / (double)sc[i], qb = (double)b[i] / (double)sc[i];
        q1[i] = qa;
        q2[i] = form_ != PBSO_FORM_DIRECT ? qa - qb : qb;
    }
    return PBSO_OK;
}

// Restore of the integrator state (the pair pbso_read_state returns): stored unscaled (scale 1), in the form's
// own variables -- (q_{k-1}, q_{k-1} - q_{k-2}) for the velocity and block forms, (q_{k-1}, q_{k-2}) for the direct form.
int Engine::write_state(int obj, const double *q1, const double *q2, int n) {
    if (!finalized_) return fail(PBSO_ERR_STATE, "write_state before finalize");
    if (!valid_obj(obj) || n < 0 || n > objs_[obj].n_modes) return fail(PBSO_ERR_INVALID, "write_state arguments");
    if (failed_) return fail(PBSO_ERR_STATE, "an earlier step failed half-way: create a new engine");
    HIPTRY(hipSetDevice(desc_.device));
    int rc = sync();
    if (rc) return rc;
    std::vector<float> a(n), b(n), sc(n, 1.f);
    for (int i = 0; i < n; ++i) {
        a[i] = (float)q1[i];
        b[i] = form_ != PBSO_FORM_DIRECT ? (float)(q1[i] - q2[i]) : (float)q2[i];
    }
    const size_t off = (size_t)obj * m_pad_;
    HIPTRY(hipMemcpyAsync(d_sq_.p + off, a.data(), n * sizeof(float), hipMemcpyHostToDevice, stream_));
    HIPTRY(hipMemcpyAsync(d_sd_.p + off, b.data(), n * sizeof(float), hipMemcpyHostToDevice, stream_));
    HIPTRY(hipMemcpyAsync(d_ss_.p + off, sc.data(), n * sizeof(float), hipMemcpyHostToDevice, stream_));
    return sync();
}

// ModalSolver::getLatestTransfer, modal_solver.h:145-147
int Engine::get_latest_transfer(int obj, double *out) {
    { int src = sync(); if (src != PBSO_OK) return src; }      // (results come from two streams: the bank's, and the scan's state on the preparation stream)
    if (!finalized_) return fail(PBSO_ERR_STATE, "get_latest_transfer before finalize");
    if (!valid_obj(obj)) return fail(PBSO_ERR_INVALID, "object id");
    const Object &o = objs_[obj];
    if (o.latest_row == XFER_UNIT) {
        for (int i = 0; i < o.n_modes; ++i) { out[i] = 1.0; out[i] *= 1E7; }     // setToUnit :89-92
        return PBSO_OK;
    }
    int src = sync();
    if (src) return src;
    HIPTRY(hipMemcpyAsync(out, d_xfer_.p + (size_t)o.latest_row * m_pad_, (size_t)o.n_modes * sizeof(double),
                          hipMemcpyDeviceToHost, stream_));
    return sync();
}

// ModalSolver::computeTransfer(pos, T *trans), modal_solver.h:302-315, batched
// Multi-listener output: from the next step on the block kernel keeps this object's block-start states.
int Engine::listeners_enable(int obj) {
    if (!finalized_) return fail(PBSO_ERR_STATE, "listeners_enable before finalize");
    if (!valid_obj(obj)) return fail(PBSO_ERR_INVALID, "object id");
    if (!is_block()) return fail(PBSO_ERR_STATE, "the multi-listener mix needs one of the block forms (frames_per_buffer = 513)");
    if (dump_row_.empty()) { dump_row_.assign(objs_.size(), -1); dump_valid_.assign(objs_.size(), 0); }
    if (dump_row_[obj] >= 0) return PBSO_OK;
    const Object &o = objs_[obj];
    // this object's f32 operand table (the engine's own table may hold the split-bf16 form)
    std::vector<float> wt((size_t)m_pad_ / 2 * 64, 0.f);
    for (int m = 0; m < o.n_modes; ++m) {
        const double eps2 = -o.c2[m], e = (1.0 - o.c1[m]) - o.c2[m];
        const double a00 = 1.0 - e, a01 = eps2, a10 = -e, a11 = eps2;
        double p00 = 1, p01 = 0, p10 = 0, p11 = 1;
        float *w = wt.data() + (size_t)(m / 2) * 64 + 16 * (2 * (m & 1));
        for (int j = 1; j <= BLOCK_J; ++j) {
            const double n00 = a00 * p00 + a01 * p10, n01 = a00 * p01 + a01 * p11;
            const double n10 = a10 * p00 + a11 * p10, n11 = a10 * p01 + a11 * p11;
            p00 = n00; p01 = n01; p10 = n10; p11 = n11;
            w[j - 1] = (float)p00;
            w[16 + j - 1] = (float)p01;
        }
    }
    HIPTRY(hipSetDevice(desc_.device));
    int rc = sync();
    if (rc) return rc;
    // the build of the bank that keeps block-start states evaluates the closed-form qnorm rows whatever qnorm_mode says
    // (a PBSO_QNORM_OFF engine has no G planes yet: the kernel would read through a null pointer)
    rc = build_gq();
    if (rc) return rc;
    HIPTRY(d_wtab32_.ensure((size_t)(n_dump_ + 1) * wt.size(), true, stream_));
    HIPTRY(hipMemcpyAsync(d_wtab32_.p + (size_t)n_dump_ * wt.size(), wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice, stream_));
    HIPTRY(hipStreamSynchronize(stream_));
    dump_row_[obj] = n_dump_++;
    dump_rows_dirty_ = true;
    return PBSO_OK;
}

// out[n_listeners][last_nb * frames]: the last step's audio of `obj` as heard at each position
int Engine::mix_listeners(int obj, const double *pos, int n_listeners, float *out, size_t n_out) {
    { int src = sync(); if (src != PBSO_OK) return src; }      // (results come from two streams: the bank's, and the scan's state on the preparation stream)
    HIPTRY(hipSetDevice(desc_.device));
    if (!finalized_) return fail(PBSO_ERR_STATE, "mix_listeners before finalize");
    if (!valid_obj(obj) || n_listeners <= 0 || !pos || !out) return fail(PBSO_ERR_INVALID, "mix_listeners arguments");
    if (dump_row_.empty() || dump_row_[obj] < 0) return fail(PBSO_ERR_STATE, "pbso_listeners_enable was not called for this object");
    if (dump_nb_ <= 0 || dump_nb_ != last_nb_) return fail(PBSO_ERR_STATE, "no step since pbso_listeners_enable");
    if (!dump_valid_[obj])
        return fail(PBSO_ERR_STATE, "the last step has buffers of this object without block states (dense force profile or per-sample launch)");
    if (n_out != (size_t)n_listeners * last_nb_ * B_) return fail(PBSO_ERR_INVALID, "mix_listeners output size");
    Object &o = objs_[obj];
    DevBuf<double> rows;
    DevBuf<float> dout;
    int rc = PBSO_OK;
    hipError_t e = rows.ensure((size_t)n_listeners * m_pad_, false, stream_);
    if (e == hipSuccess) e = dout.ensure(n_out, false, stream_);
    if (e == hipSuccess && o.have_maps && o.n_maps > 0) {
        // ModalSolver::computeTransfer(pos, T*) per listener (modal_solver.h:302-315)
        if (o.n_maps < o.n_modes) { rows.release(); dout.release(); return fail(PBSO_ERR_MISSING_MAP, "FFAT maps do not cover the audible modes"); }
        std::vector<FfatEvent> evs(n_listeners);
        for (int i = 0; i < n_listeners; ++i) {
            evs[i].obj = obj;
            evs[i].row = i;
            for (int j = 0; j < 3; ++j) evs[i].pos[j] = pos[3 * i + j];
        }
        DevBuf<FfatEvent> dev;
        e = dev.ensure(n_listeners, false, stream_);
        if (e == hipSuccess) e = hipMemcpyAsync(dev.p, evs.data(), n_listeners * sizeof(FfatEvent), hipMemcpyHostToDevice, stream_);
        if (e == hipSuccess) {
            // (the object's modes share one map geometry: every listener located once, lane = mode -- kernels_exact.hip, round 6)
            int le = (!ffat_shared_h_.empty() && ffat_shared_h_[obj])
                         ? launch_ffat_lookup_shared(dev.p, n_listeners, d_ffat_shared_.p, d_geom_.p, d_geom_off_.p, d_n_modes_.p, d_ffat_k_.p,
                                                     d_ffat_valid_.p, d_psi_t_.p, rows.p, m_pad_, stream_)
                         : launch_ffat_lookup(dev.p, n_listeners, d_geom_.p, d_geom_off_.p, d_n_modes_.p, d_psi_.p, rows.p, m_pad_, stream_);
            if (le) e = (hipError_t)le;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        dev.release();
    } else if (e == hipSuccess) {
        std::vector<double> unit((size_t)n_listeners * m_pad_, 1E7);          // no maps: TransMessage::setToUnit (modal_solver.h:89-92)
        e = hipMemcpyAsync(rows.p, unit.data(), unit.size() * sizeof(double), hipMemcpyHostToDevice, stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
    }
    if (e == hipSuccess) {
        // a buffer whose waves left the scaled-state representation (a transfer weight outside [2^-20, 2^40]) was stepped
        // per sample inside the block kernel and kept no block states: its scale row says 0.  The planner cannot know.
        std::vector<float> sc((size_t)last_nb_ * m_pad_);
        e = hipMemcpyAsync(sc.data(), d_xscale_.p + (size_t)dump_row_[obj] * last_nb_ * m_pad_, sc.size() * sizeof(float), hipMemcpyDeviceToHost, stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(stream_);
        if (e == hipSuccess)
            for (int b = 0; b < last_nb_ && rc == PBSO_OK; ++b)
                for (int m = 0; m < o.n_modes; ++m)
                    if (sc[(size_t)b * m_pad_ + m] == 0.f) {
                        rc = fail(PBSO_ERR_STATE, "the last step has a buffer of this object that was stepped per sample (a transfer weight outside "
                                                  "the scaled-state range): no block states to mix");
                        break;
                    }
    }
    if (rc != PBSO_OK) { rows.release(); dout.release(); return rc; }
    if (e == hipSuccess) {
        const size_t row = (size_t)dump_row_[obj];
        int le = iir_block::launch_listener_mix(d_xdump_.p + row * last_nb_ * 32 * m_pad_ * 2, d_xscale_.p + row * last_nb_ * m_pad_,
                                                d_wtab32_.p + row * ((size_t)m_pad_ / 2 * 64), rows.p, dout.p, last_nb_, m_pad_, o.n_modes,
                                                n_listeners, (long long)last_nb_ * B_, stream_);
        if (le) e = (hipError_t)le;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, n_out * sizeof(float), hipMemcpyDeviceToHost, stream_);
    if (e == hipSuccess) e = hipStreamSynchronize(stream_);
    if (e != hipSuccess) rc = hip_fail(e, "mix_listeners");
    rows.release();
    dout.release();
    return rc;
}

// Host delivery (the reference's consumer is a host queue of SoundMessages, modal_solver.h:79-82, 346-363): a step whose audio
// of ALL objects lands in caller memory.  Pinned host memory (pbso_host_alloc, hipHostMalloc, hipHostRegister) is mapped into
// the device's address space: the oscillator bank writes its samples STRAIGHT into it over PCIe -- every sample is stored
// exactly once, coalesced -- so delivery needs no second pass and the step runs at the link's rate (1024 x 512 x 86: 3.5 ms per
// step against 3.2 ms for the bare copy of the same 181 MB; scripts/debug/r04_d2h.py).  What was tried first -- the bank into
// device memory, then hipMemcpyAsync on a copy stream beside the next step's bank -- does NOT overlap on this stack: the runtime
// runs the device-to-host copy as a blit KERNEL, which finds no free registers while the bank's two 256-register waves per SIMD
// are resident (4.35 ms per step waited one by one, 4.65 ms "pipelined").  That staged path remains for pageable memory.
int Engine::step_to_host(int nb, float *host_out, size_t n) {
    HIPTRY(hipSetDevice(desc_.device));
    if (!finalized_) return fail(PBSO_ERR_STATE, "step before finalize");
    const size_t total = (size_t)objs_.size() * (size_t)std::max(nb, 0) * B_;
    if (!host_out || nb <= 0 || n != total) return fail(PBSO_ERR_INVALID, "step_to_host arguments (n_floats = n_objects * n_buffers * frames_per_buffer)");
    if (!copy_stream_) {
        {
            // (highest priority: should the runtime run the copy as a blit kernel, its workgroups take the CUs the retiring
            //  bank frees before the next bank's do)
            int least = 0, greatest = 0;
            HIPTRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIPTRY(hipStreamCreateWithPriority(&copy_stream_, hipStreamNonBlocking, greatest));
        }
        for (int i = 0; i < 2; ++i) {
            HIPTRY(hipEventCreateWithFlags(&ev_host_bank_[i], hipEventDisableTiming));
            HIPTRY(hipEventCreateWithFlags(&ev_host_copy_[i], hipEventDisableTiming));
        }
    }
    const int slot = host_slot_;
    {
        hipPointerAttribute_t attr;
        std::memset(&attr, 0, sizeof(attr));
        if (hipPointerGetAttributes(&attr, host_out) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
            int rc = step(nb, attr.devicePointer);
            if (rc != PBSO_OK) return rc;
            if ((rc = drain_submit()) != PBSO_OK) return rc;          // (the event below goes behind the step's launches)
            HIPTRY(hipEventRecord(ev_host_copy_[slot], stream_));      // "delivered" = the bank (and what follows it) has finished
            host_last_ = slot;
            host_slot_ ^= 1;
            return PBSO_OK;
        }
        (void)hipGetLastError();                         // (a pageable pointer is not an error here)
    }
    HIPTRY(d_audio_host_[slot].ensure(total, false, stream_));
    HIPTRY(hipStreamWaitEvent(stream_, ev_host_copy_[slot], 0));          // the copy that last read this buffer is done
    int rc = step(nb, d_audio_host_[slot].p);
    if (rc != PBSO_OK) return rc;
    if ((rc = drain_submit()) != PBSO_OK) return rc;
    HIPTRY(hipEventRecord(ev_host_bank_[slot], stream_));
    HIPTRY(hipStreamWaitEvent(copy_stream_, ev_host_bank_[slot], 0));
    HIPTRY(hipMemcpyAsync(host_out, d_audio_host_[slot].p, total * sizeof(float), hipMemcpyDeviceToHost, copy_stream_));
    HIPTRY(hipEventRecord(ev_host_copy_[slot], copy_stream_));
    host_last_ = slot;
    host_slot_ ^= 1;
    return PBSO_OK;
}

int Engine::host_wait() {
    if (host_last_ < 0) return PBSO_OK;
    HIPTRY(hipSetDevice(desc_.device));
    HIPTRY(hipEventSynchronize(ev_host_copy_[host_last_]));
    return PBSO_OK;
}

// The step's audio summed over the objects, on the device, in a fixed order (what one output stream plays when the scene's
// objects sound together; the per-rank half of the device group's PBSO_GATHER_MIX).  Asynchronous on the engine's stream.
int Engine::mix_objects(void *d_out) {
    if (!last_audio_ || !d_out) return fail(PBSO_ERR_STATE, "mix_objects: no step yet (or a null output)");
    HIPTRY(hipSetDevice(desc_.device));
    const int N = (int)objs_.size();
    const long long n = (long long)last_nb_ * B_;
    HIPTRY(d_mix_parts_.ensure((size_t)mix_objects_groups(N) * n, false, stream_));
    LAUNCHTRY(launch_mix_objects(last_audio_, N, n, n, d_mix_parts_.p, (float *)d_out, stream_));
    return PBSO_OK;
}

int Engine::object_n_maps(int obj) {
    if (!valid_obj(obj)) return fail(PBSO_ERR_INVALID, "object id");
    return objs_[obj].have_maps ? objs_[obj].n_maps : 0;
}

int Engine::compute_transfer_batch(int obj, const double *pos, int n_pos, double *out, int out_cols) {
    HIPTRY(hipSetDevice(desc_.device));      // the caller's thread may have another device current
    if (!finalized_) return fail(PBSO_ERR_STATE, "compute_transfer_batch before finalize");
    if (!valid_obj(obj) || n_pos < 0 || (n_pos && (!pos || !out))) return fail(PBSO_ERR_INVALID, "arguments");
    Object &o = objs_[obj];
    if (!o.have_maps) return 0;
    const int nmap = o.n_maps;
    if (nmap > o.n_modes) return fail(PBSO_ERR_INVALID, "more FFAT maps than audible modes is not supported");
    if (out_cols < nmap) return fail(PBSO_ERR_INVALID, "compute_transfer_batch: out_cols is smaller than the object's map count (pbso_object_n_maps)");
    for (int m = 0; m < nmap; ++m)
        if (m >= (int)o.geom.size() || !o.geom[m].valid)
            return fail(PBSO_ERR_MISSING_MAP, "FFAT map ids are not 0..size-1 (std::map::at throws)");
    if (!n_pos) return 1;
    // Positions go up once; the lookups run in chunks (one workgroup per mode and 1024 positions, the mode's map
    // staged in LDS) into two row buffers, and the copy of chunk c to the caller's memory (the copy stream) runs beside
    // the kernel of chunk c + 1.
    const int CH = 16384;
    const int n_chunks = (n_pos + CH - 1) / CH;
    DevBuf<double> rows, dpos;
    hipEvent_t ev_k[2] = {nullptr, nullptr};
    int rc = PBSO_OK;
    hipError_t e = rows.ensure((size_t)2 * std::min(n_pos, CH) * m_pad_, false, stream_);
    if (e == hipSuccess) e = dpos.ensure((size_t)3 * n_pos, false, stream_);
    if (e == hipSuccess) e = hipMemcpyAsync(dpos.p, pos, (size_t)3 * n_pos * sizeof(double), hipMemcpyHostToDevice, stream_);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ev_k[i], hipEventDisableTiming);
    const FfatGeom *g0 = d_geom_.p + geom_off_h_[obj];
    // (round 6) an object whose modes share one map geometry: the positions as lookup events of the shared-geometry kernel -- located
    // once per position, lane = mode, the maps read cell-major (kernels_exact.hip) -- instead of one workgroup per (mode, 1024 positions)
    const bool shared = !ffat_shared_h_.empty() && ffat_shared_h_[obj] && e == hipSuccess;
    DevBuf<FfatEvent> dev_ev;
    if (shared) {
        std::vector<FfatEvent> evs((size_t)n_pos);
        for (int p = 0; p < n_pos; ++p) {
            evs[p].obj = obj;
            evs[p].row = (p % CH) + ((p / CH) & 1) * CH;          // (its row in the two-chunk row buffer)
            for (int j = 0; j < 3; ++j) evs[p].pos[j] = pos[3 * (size_t)p + j];
        }
        e = dev_ev.ensure((size_t)n_pos, false, stream_);
        if (e == hipSuccess) e = hipMemcpy(dev_ev.p, evs.data(), (size_t)n_pos * sizeof(FfatEvent), hipMemcpyHostToDevice);
    }
    auto launch_chunk = [&](int c) -> hipError_t {
        const int p0 = c * CH, n = std::min(CH, n_pos - p0);
        int le = shared ? launch_ffat_lookup_shared(dev_ev.p + p0, n, d_ffat_shared_.p, d_geom_.p, d_geom_off_.p, d_n_modes_.p, d_ffat_k_.p,
                                                    d_ffat_valid_.p, d_psi_t_.p, rows.p, m_pad_, stream_)
                        : launch_ffat_batch(dpos.p + 3 * (size_t)p0, n, g0, nmap, d_psi_.p, rows.p + (size_t)(c & 1) * CH * m_pad_, m_pad_, stream_);
        if (le) return (hipError_t)le;
        return hipEventRecord(ev_k[c & 1], stream_);
    };
    if (e == hipSuccess) e = launch_chunk(0);
    for (int c = 0; c < n_chunks && e == hipSuccess; ++c) {
        if (c + 1 < n_chunks) e = launch_chunk(c + 1);          // (its row buffer was copied out two chunks ago: the copy below is synchronous)
        const int p0 = c * CH, n = std::min(CH, n_pos - p0);
        if (e == hipSuccess) e = hipStreamWaitEvent(prep_stream_, ev_k[c & 1], 0);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(out + (size_t)p0 * out_cols, (size_t)out_cols * sizeof(double), rows.p + (size_t)(c & 1) * CH * m_pad_,
                                 (size_t)m_pad_ * sizeof(double), (size_t)nmap * sizeof(double), n, hipMemcpyDeviceToHost, prep_stream_);
        if (e == hipSuccess) e = hipStreamSynchronize(prep_stream_);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream_);
    if (e != hipSuccess) rc = hip_fail(e, "compute_transfer_batch");
    for (hipEvent_t ev : ev_k)
        if (ev) (void)hipEventDestroy(ev);
    rows.release();
    dpos.release();
    dev_ev.release();
    return rc == PBSO_OK ? 1 : rc;
}

int Engine::info(pbso_engine_info *out) {
    std::memset(out, 0, sizeof(*out));
    out->n_objects = (int)objs_.size();
    out->frames_per_buffer = B_;
    out->modes_padded = m_pad_;
    out->modes_per_lane = R_;
    out->waves_per_object = W_;
    out->n_teams = n_teams_;
    out->lds_bytes_per_workgroup = !finalized_ ? 0 : is_block() ? (int)block_lds_bytes(W_, R_) : (int)iir_lds_bytes(W_, n_tiles_);
    out->recurrence_form = form_;
    out->total_block_launches = tot_block_launches_;
    out->total_sample_launches = tot_sample_launches_;
    out->total_split_launches = tot_split_launches_;
    out->total_time_chunk_launches = tot_tc_launches_;
    out->total_dense_increment_launches = tot_tc_dense_launches_;
    out->total_segmented_scans = tot_seg_scans_;
    out->last_time_chunk_shape = last_tc_shape_;
    out->last_time_chunk_buffers = last_tc_cb_;
    out->last_time_chunk_teams = last_tc_teams_;
    out->total_dropped_hits = dropped_hits_.load();
    out->total_one_stream_launches = tot_one_stream_launches_;
    out->start_gate = gate_choice_;
    out->total_gate_timeouts = gate_timeouts_;
    out->total_host_submit_ms = hprof_[4];
    out->total_ffat_shared_events = tot_ffat_shared_events_;
    out->total_ffat_general_events = tot_ffat_general_events_;
    out->buffers_done = buffers_done_;
    out->last_step_host_plan_ms = last_plan_ms_;
    out->last_step_forced_rows = last_frows_;
    out->last_step_transfer_rows = last_trows_;
    int rc = harvest_timing(true);
    if (rc) return rc;
    out->last_step_kernel_ms = last_kernel_ms_;
    out->last_step_device_ms = last_device_ms_;
    out->total_kernel_ms = tot_kernel_ms_;
    out->total_device_ms = tot_device_ms_;
    out->total_timed_launches = tot_timed_launches_;
    out->total_host_plan_ms = tot_plan_ms_;
    out->total_steps = tot_steps_;
    return PBSO_OK;
}

// Reads the HIP-event pairs of finished launches into the totals.  blocking: waits for everything in flight (pbso_get_info);
// otherwise only the launches whose last event has already completed are read (called once per launch: keeps the number of live
// events small and lets the K2 priority follow the workload without ever stalling the host).
int Engine::harvest_timing(bool blocking) {
    if (ev_pending_.empty()) return PBSO_OK;
    if (blocking) {
        int rc = sync();
        if (rc) return rc;
    }
    size_t done = 0;
    for (; done < ev_pending_.size(); ++done) {
        EvQuad &q = ev_pending_[done];
        if (submit_ && q.batch > submit_->done()) break;      // (its events are not recorded yet: an unrecorded event reads as complete)
        if (!blocking && hipEventQuery(q.p1) != hipSuccess) break;
        float ms = 0, ms2 = 0;
        HIPTRY(hipEventElapsedTime(&ms, q.k0, q.k1));
        HIPTRY(hipEventElapsedTime(&ms2, q.p0, q.p1));
        if (timeline_) {                            // PBSO_TIMELINE=1: where each timed launch's events fall (ms since the first one)
            float a = 0, b = 0, c = 0, d = 0;
            if (!timeline_have_base_) { timeline_ref_ = q.p0; timeline_quad_ = q; timeline_have_base_ = true; timeline_keep_ = true; timeline_h0_ = q.h_prep; }
            HIPTRY(hipEventElapsedTime(&a, timeline_ref_, q.p0));
            HIPTRY(hipEventElapsedTime(&b, timeline_ref_, q.k0));
            HIPTRY(hipEventElapsedTime(&c, timeline_ref_, q.k1));
            HIPTRY(hipEventElapsedTime(&d, timeline_ref_, q.p1));
            std::fprintf(stderr, "pbso timeline: step %lld device: prep starts %.3f | bank starts %.3f ends %.3f | launch done %.3f || host: step entered %.3f, "
                                 "prep submitted from %.3f (upload call returned %.3f), bank submitted at %.3f, all submitted %.3f\n", (long long)q.step_id, a, b, c, d,
                         q.h_enter - timeline_h0_, q.h_prep - timeline_h0_, q.h_copy - timeline_h0_, q.h_bank - timeline_h0_, q.h_done - timeline_h0_);
        }
        if (q.step_id != harvest_step_) {          // "last step" sums the launches of one step
            harvest_step_ = q.step_id;
            last_kernel_ms_ = 0;
            last_device_ms_ = 0;
        }
        if (q.has_k2) {
            float msf = 0;
            HIPTRY(hipEventElapsedTime(&msf, q.f0, q.f1));
            // exponential averages over the last few timed launches
            k2_ms_avg_ = k2_ms_n_ ? 0.75 * k2_ms_avg_ + 0.25 * msf : msf;
            k2_bank_ms_avg_ = k2_ms_n_ ? 0.75 * k2_bank_ms_avg_ + 0.25 * ms : ms;
            k2_ms_n_ += 1;
        }
        last_kernel_ms_ += ms;
        tot_kernel_ms_ += ms;
        tot_timed_launches_ += 1;
        last_device_ms_ += ms2;
        tot_device_ms_ += ms2;
        if (timeline_keep_) timeline_keep_ = false;      // (the reference launch's events stay out of the free list)
        else ev_free_.push_back(q);
    }
    ev_pending_.erase(ev_pending_.begin(), ev_pending_.begin() + (long)done);
    if (k2_prio_auto_ && k2_ms_n_ >= 2) {
        // the longer of the two kernels that run side by side gets the SIMDs they share (with hysteresis)
        const int want = k2_ms_avg_ > (k2_prio_ ? 0.85 : 0.95) * k2_bank_ms_avg_ ? 3 : 0;
        if (want != k2_prio_ && host_profile_)
            std::fprintf(stderr, "pbso: force-profile kernel %.3f ms, oscillator bank %.3f ms per timed launch: K2 wave priority %d -> %d\n",
                         k2_ms_avg_, k2_bank_ms_avg_, k2_prio_, want);
        k2_prio_ = want;
    }
    return PBSO_OK;
}

}  // namespace pbso
