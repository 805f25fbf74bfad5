/*
 * openpbso_amd.h -- C ABI of the MI355X-native modal sound engine.
 *
 * Drop-in boundary for the per-audio-sample hot path of jhwang7628/openpbso.
 * The reference has no FFI of its own: the boundary it sits behind is the
 * public interface of the header-only template ModalSolver<double,513>
 * (modal_solver.h:100-179) plus its message structs.  Every entry point below
 * names the reference interface it replaces (file:line, relative to the
 * reference root).  One engine batches MANY independent ModalSolver instances
 * ("objects") on one GPU; object i behaves like its own ModalSolver<double>.
 *
 * Conventions
 *  - every function returns an int status (PBSO_OK == 0, errors < 0); the
 *    bool-returning queue operations of the reference return 1 (true) /
 *    0 (false) / <0 (error).  No exceptions cross this boundary.
 *  - plain pointers and sizes only; `void *` device pointers are HIP device
 *    addresses, `void *stream` is a hipStream_t.
 *  - one caller thread per engine (the facade keeps the reference's
 *    single-producer/single-consumer discipline).
 *  - time is counted in audio buffers of `frames_per_buffer` samples; buffer 0
 *    is the first buffer the first pbso_step() produces.  `not_before` stamps
 *    let a batch caller say at which buffer a message becomes visible to the
 *    simulation thread's try_dequeue (the GUI thread's wall-clock in the
 *    reference).  not_before = 0 means "already there".  Stamped calls other than force
 *    messages (AR parameters, listener position, use-transfer flag) of one object form a FIFO in
 *    stamp order: an AR-parameter message that finds the 1-slot queue full waits there -- the
 *    NoFail spin of modal_solver.h:382-393 -- and so do the calls stamped behind it.
 *  - the HIP path is the only path: there is no CPU fallback.
 */
#ifndef OPENPBSO_AMD_H
#define OPENPBSO_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBSO_ABI_VERSION 6
#define PBSO_SAMPLE_RATE 44100          /* config.h:13 */
#define PBSO_FRAMES_PER_BUFFER 513      /* config.h:14 */

enum pbso_status {
    PBSO_OK = 0,
    PBSO_ERR_INVALID = -1,      /* bad argument (the reference would assert) */
    PBSO_ERR_HIP = -2,          /* a HIP runtime call failed; see pbso_last_error */
    PBSO_ERR_STATE = -3,        /* call not valid in the engine's current state */
    PBSO_ERR_IO = -4,           /* file missing / unreadable / malformed */
    PBSO_ERR_MISSING_MAP = -5,  /* std::map::at would throw (modal_solver.h:296,312) */
    PBSO_ERR_ASSERT = -6,       /* a live reference assert would fire (e.g. modal_solver.h:223) */
    PBSO_ERR_NOMEM = -7
};

/* forces.h:12-16 */
enum pbso_force_type {
    PBSO_POINT_FORCE = 0,
    PBSO_GAUSSIAN_FORCE = 1,
    PBSO_AUTOREGRESSIVE_FORCE = 2
};

/* how the spatial (modal) part of a force message is given */
enum pbso_force_data_kind {
    PBSO_DATA_EXPLICIT = 0,   /* data[n_data] doubles: ForceMessage::data as the GUI built it */
    PBSO_DATA_VERTEX = 1,     /* GetModalForceVertex on the device (tools/real_time_modal_sound.cpp:268-280) */
    PBSO_DATA_FACE = 2,       /* GetModalForceFace on the device   (tools/real_time_modal_sound.cpp:236-252) */
    PBSO_DATA_ZERO = 3        /* setZero(N): the dummy start/stop messages (tools/...:754-776) */
};

enum pbso_recurrence_form {
    PBSO_FORM_BLOCK = 0,      /* default: block state-space form.  Between forces a mode is autonomous, so the 16
                                 samples after a state x = (q, q-q_prev) are e1' A^j x and the state 16 samples on
                                 is A^16 x: the sum over modes becomes a [16 x 2M].[2M x 16 blocks] product on the
                                 f32 matrix pipe (exact f32 FMA chains), the state advances 33 times per buffer
                                 instead of 513.  Buffers with a dense force profile (Gaussian / AR) step every
                                 sample as PBSO_FORM_VELOCITY does.  Needs frames_per_buffer = 513 (else the
                                 engine runs PBSO_FORM_VELOCITY); qnorm is evaluated in closed form.            */
    PBSO_FORM_VELOCITY = 1,   /* per-sample recurrence, state (q, q-q_prev), coefficients (eps^2, 1-c1-c2): fp32-safe */
    PBSO_FORM_DIRECT = 2,     /* per-sample, the reference's literal q = c1 q1 + c2 q2 + c3 Q in fp32 */
    PBSO_FORM_BLOCK_BF16 = 3  /* the block form with the OUTPUT projection (which has no feedback) evaluated as a
                                 split-bf16 product: every f32 operand x = hi + lo (two bf16, 16 significant bits),
                                 W.X ~ Whi.Xhi + Whi.Xlo + Wlo.Xhi on v_mfma_f32_16x16x32_bf16 (16x the f32 MFMA rate,
                                 f32 accumulation).  The state recurrence stays exact f32.  Error 1.5e-5 ... 3e-5 of
                                 the peak (PBSO_FORM_BLOCK: 7e-6; stated tolerance 5e-4).                               */
};

enum pbso_qnorm_mode {
    PBSO_QNORM_OFF = 0,       /* getQBufferNorm never consumed: skip the 5th FMA */
    PBSO_QNORM_ALL = 1,       /* per-sample accumulation, one row per (object, buffer): modal_solver.h:262-273 */
    PBSO_QNORM_CLOSED = 2     /* same rows; in buffers that are force-free after their first sample the sum
                                 of q^2 is the quadratic form x0' G x0 of the state after sample 0 (G per mode,
                                 fp64 on the host) -- exact in exact arithmetic; other buffers as mode 1 */
};

typedef struct pbso_engine pbso_engine;

typedef struct pbso_engine_desc {
    int abi_version;          /* PBSO_ABI_VERSION */
    int device;               /* HIP device ordinal */
    int frames_per_buffer;    /* 0 -> 513 (ModalSolver's BUF_SIZE template argument) */
    int sample_rate;          /* 0 -> 44100 */
    int recurrence_form;      /* enum pbso_recurrence_form */
    int qnorm_mode;           /* enum pbso_qnorm_mode */
    int modes_per_lane;       /* 0 = auto; 1, 2, 3, 4 or 8 oscillators per lane (block form: 1, 2 or 4) */
    void *stream;             /* hipStream_t to launch on; NULL -> engine-owned non-blocking stream (the
                               * legacy null stream cannot be selected: order other work on it with
                               * pbso_sync).  A second, engine-owned high-priority stream prepares the
                               * next launch (projection, FFAT lookup, force profiles).                */
    /* ---- ABI 4: which kernels run and how.  Per ENGINE (two engines of one process may differ); 0 is the engine's
     * own policy in every field, so a zeroed descriptor is the product's default.  None of them changes what is
     * computed beyond the stated tolerance; tests and A/B runs use them to pin one path.                           */
    int bank_kernel;          /* enum pbso_bank_kernel */
    int time_chunks;          /* K5, launches cut along the time axis (a scan of the buffer-start states makes the buffers of a
                               * launch independent): 0 = when the scene leaves SIMDs idle, < 0 = never, n > 0 = always, n buffers
                               * per chunk */
    int direct_hits;          /* < 0: plain vertex hits go through the fp64 projection / combine kernels instead of the bank's f32 table */
    int forced_block;         /* < 0: dense-profile buffers are stepped per sample (no block form for them) */
    int dense_launches;       /* launches that are mostly dense-profile buffers: 0 policy, 1 block kernel, 2 per-sample kernel */
    int device_profiles;      /* < 0: force time profiles (forces.h:81-137) on the host in fp64, uploaded */
    int profile_kernel;       /* K2: 0 row-parallel, 1 chain kernel (parallel AR scan), 2 chain kernel with the reference's serial AR loop */
    int profile_margin_pct;   /* K2 row-parallel: candidate range in percent of the expected need (0 -> 100; tests: < 100 forces the shortfall path) */
    int profile_priority;     /* K2 chain kernel's wave priority: 0 auto, 1..4 -> s_setprio 0..3 */
    int team_waves;           /* > 0: waves per team (workgroup) of the oscillator bank; 0 = policy */
    int pipe_consumers;       /* pipeline kernel: 0 policy, 1..3 consumer waves per team, 4 = the five-role team of mostly-dense launches */
    long long pipe_max_teams; /* pipeline kernel eligibility: at most this many 64-mode teams (0 -> 2 per CU) */
    int chunk_buffers;        /* a step longer than this is cut into several launches (0 -> 128).  A launch's fixed costs (ramp, write drain,
                               * hand-over between the streams: ~35 us) are per launch: throughput callers that step many seconds per call
                               * set it to their step */
    int plan_threads;         /* host planner threads (0 -> 1: the caller's thread) */
    int plan_pin;             /* != 0: pin the helper threads into the caller's core complex */
    int timing_every;         /* HIP-event pairs around every n-th launch (0 -> 1; < 0: none) */
    int warm_copies;          /* < 0: pbso_finalize does not warm the runtime's copy queues */
    int stream_sync;          /* how the engine's two streams order their launches.  0 policy: events, and -- where the device supports
                               * hipStreamWaitValue64 -- a START GATE in front of the preparation kernels of launches of >= 256 buffers:
                               * they wait for a value the previous launch's bank kernel stores when it starts, so the preparation
                               * never runs two launches ahead (a long scan that does trails its bank and collides with the next
                               * bank's start); since round 6 also the hand-over of launches of fewer than 256 buffers to the bank's stream
                               * through a value in signal memory instead of an event (half the latency: 1.5 - 6 % per step of 86 buffers).
                               * 1 events only.  2 the hand-over through a value for every launch.  3 the gate for every launch (costs 0.5 - 1.5 %
                               * per step of 86 buffers).  2 and 3 fail at creation where the device lacks the interface.  4 (round 5): the gate
                               * for every launch as a wait of the SUBMITTING thread on a word of pinned host memory the previous bank's first
                               * workgroup writes (hipStreamWaitValue64 runs as a waiting kernel on this stack); costs the host one launch of
                               * run-ahead; fails at creation without pinned host memory.
                               * Round 6: the DEVICE form of the gate is a kernel that spins on a memory word until another kernel
                               * has started; where something serialises kernel dispatches -- a profiler collecting counters
                               * (rocprofv3 --pmc: five of five passes ended at their time limit with the gate, none without),
                               * AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING -- the waiting kernel can be dispatched first and never ends.  Under 0
                               * (policy) the engine looks for such an environment at creation and then orders its launches by events only
                               * (pbso_engine_info::start_gate says what it chose); 2 and 3 are the caller's explicit choice */
    int latency_path;         /* < 0: never prepare a launch on the bank's own stream (0: a step of at most four buffers submitted
                               * while the device is idle does -- nothing to overlap with, one stream hand-over less) */
    /* ---- ABI 5 */
    int time_chunk_shape;     /* modes per lane of the teams of a time-chunked launch: 0 policy, 1 / 2 / 4 (A/B runs) */
    int scan_kernel;          /* K5's scan of chunk-start states: 0 policy (cut along the time axis itself -- one wave per chunk, the chunks' affine
                               * maps composed in LDS -- for launches of 2 .. 8 chunks of more than one buffer whose whole scan is a few hundred
                               * waves: it shortens a lone scan, not a throughput-bound one), 1 always the serial scan, 2 the segmented one
                               * wherever the launch has 2 .. 8 chunks */
    /* ---- ABI 6 */
    int fuse_short_launches;  /* < 0: never.  0 (policy): a launch in which every AutoregressiveForce adds its samples once -- any launch of
                               * one buffer: the real-time step -- evaluates variates, zero-state uses and profile rows in ONE kernel instead
                               * of three, and in a launch of one buffer the combine kernel takes the rows of explicit data and the
                               * projections that outlive the buffer itself instead of a scatter and a projection launch in front of it:
                               * a sustained-contact buffer is three kernels (profiles, combine, oscillator bank) instead of seven.  The same
                               * values in the same order: bit-identical to the separate launches */
    int submit_thread;        /* > 0: a second host thread makes the HIP calls of a launch (uploads, kernel launches, event records) while
                               * pbso_step returns and the caller's thread plans the next one -- for scenes whose step is set by the HOST
                               * (64 x 256 with a listener move per buffer: 0.045 ms of planning + 0.043 ms of calls against a 0.057 ms
                               * bank).  What changes for the caller: when pbso_step returns, its launches are not necessarily in their
                               * streams yet.  Every entry point that reads results or uses the engine's streams waits for that thread first;
                               * a caller that puts work of its OWN on the engine's stream behind a step calls pbso_flush in between; an
                               * error of a recorded call surfaces at the next entry point that waits.  0 (default): every call is made by
                               * pbso_step itself.  Not with stream_sync = 4; engines of a device group never use it */
} pbso_engine_desc;

enum pbso_bank_kernel {
    PBSO_BANK_AUTO = 0,       /* per launch: the block kernel K1b (chunked along time when the scene is small, K5), the pipeline
                                 kernel K1p for small scenes whose launches are mostly dense-profile buffers */
    PBSO_BANK_BLOCK = 1,      /* K1b always (never the pipeline kernel) */
    PBSO_BANK_PIPE = 2        /* K1p whenever the engine is eligible for it (f32 block form, one mode per lane, few teams) */
};

/* --- lifetime ------------------------------------------------------------- */
int pbso_engine_create(const pbso_engine_desc *desc, pbso_engine **out);
void pbso_engine_destroy(pbso_engine *e);
const char *pbso_last_error(const pbso_engine *e);
const char *pbso_status_string(int status);
int pbso_abi_version(void);

/* --- BuildSolver (tools/real_time_modal_sound.cpp:309-345) ------------------
 * new ModalSolver<double>(n_modes) + ModalIntegrator<double>::Build(density,
 * omega_squared, alpha, beta, 1/44100, n_modes) (modal_integrator.h:47-101).
 * mode_shapes (optional) is ModeData::_modes, mode-major [>=n_modes][n_dof]
 * doubles (ModeData.h:24), needed only for PBSO_DATA_VERTEX/FACE messages.   */
typedef struct pbso_object_desc {
    int n_modes;                    /* N_modesAudible */
    int n_omega;                    /* entries in omega_squared (>= n_modes; the reference asserts) */
    const double *omega_squared;    /* ModeData::_omegaSquared */
    double density, alpha, beta;    /* ModalMaterial */
    int n_dof;                      /* 3 * |V|, 0 if no mode shapes */
    const double *mode_shapes;      /* [n_modes][n_dof] or NULL */
} pbso_object_desc;
int pbso_add_object(pbso_engine *e, const pbso_object_desc *d, int *object_id);

/* the reference's file conventions (tools/...:480-501, 316-329): reads
 * <modes_path> (ModeData.h:61-83), <material_path> (ModalMaterial.h:35-55),
 * optional <ffat_dir>/freq_threshold.txt and <ffat_dir>/ *.fatcube
 * (ffat_map_serialize.h:166-279).  ffat_dir may be NULL.                      */
int pbso_add_object_from_files(pbso_engine *e, const char *modes_path,
                               const char *material_path, const char *ffat_dir,
                               int *object_id, int *n_modes_audible);

/* FFAT_Map<double,3> runtime fields (ffat_solver.h:171-180, 281-293) as the
 * .fatcube loader fills them (ffat_map_serialize.h:166-254).                  */
typedef struct pbso_ffat_map {
    int mode_id;
    double k;
    double center3[3];
    double cell_size;
    double low_corners[6][3];
    int n_elements[6][2];
    int strides[6];
    double center[3];
    double bbox_low[3];
    double bbox_top[3];
    int n_psi;
    const double *psi;
} pbso_ffat_map;
/* ModalSolver::readFFATMaps (modal_solver.h:278-284), maps already parsed */
int pbso_object_set_ffat_maps(pbso_engine *e, int object_id, const pbso_ffat_map *maps, int n_maps);
/* ModalSolver::readFFATMaps from a directory of .fatcube files */
int pbso_object_read_ffat_maps(pbso_engine *e, int object_id, const char *dir);
/* parse one .fatcube byte string (ffat_map.proto:8-51). psi is malloc'ed:
 * release with pbso_ffat_map_free.                                            */
int pbso_fatcube_parse(const unsigned char *bytes, size_t n, pbso_ffat_map *out);
void pbso_ffat_map_free(pbso_ffat_map *m);

/* --- loaders, usable without an engine -------------- */
/* ModeData<double>::read (ModeData.h:61-83).  *omega_squared [n_modes] and
 * *modes [n_modes][n_dof] are malloc'ed: release with pbso_free.              */
int pbso_modes_read(const char *path, int *n_dof, int *n_modes, double **omega_squared, double **modes);
/* ModeData<double>::numModesAudible (ModeData.h:120-148) */
int pbso_num_modes_audible(const double *omega_squared, int n_modes, double density, double audible_freq);
/* ModalMaterial<double>::Read (ModalMaterial.h:35-55): out = density,
 * youngsModulus, poissonRatio, alpha, beta                                    */
int pbso_material_read(const char *path, double out[5]);
/* <name>.tet.obj as the tool reads it (tools/real_time_modal_sound.cpp:508-509): igl::read_triangle_mesh
 * (only `v` / `f` records; 1-based, negative and a/b/c indices; polygons become triangle fans) and
 * igl::per_vertex_normals with libigl's default (area) weighting, each row normalised -- the tool normalises
 * VN.row(vid) again at the hit (:607).  libigl is an un-vendored submodule of the reference: the weighting is
 * its documented default, not pinned by code in the reference tree.  vertices [n_vertices][3], faces
 * [n_faces][3] (0-based), vertex_normals [n_vertices][3] are malloc'ed: release with pbso_free.             */
int pbso_obj_read(const char *path, int *n_vertices, int *n_faces, double **vertices, int **faces,
                  double **vertex_normals);
void pbso_free(void *p);

/* uploads all objects to HBM; no pbso_add_object afterwards */
int pbso_finalize(pbso_engine *e);

/* --- messages (modal_solver.h:27-98, forces.h:50-55) ---------------------- */
typedef struct pbso_force_msg {
    int force_type;                 /* ForceMessage::forceType; selects the Force subclass */
    double gaussian_width_us;       /* GaussianForce(width), forces.h:42-46 */
    int sustained_force_start;      /* ForceMessage::sustainedForceStart */
    int sustained_force_end;        /* ForceMessage::sustainedForceEnd */
    int clear_all_forces;           /* ForceMessage::clearAllForces */
    int data_kind;                  /* enum pbso_force_data_kind */
    const double *data;             /* PBSO_DATA_EXPLICIT: ForceMessage::data */
    int n_data;
    int vids[3];                    /* VERTEX: vids[0]; FACE: the three vertex ids */
    double coords[3];               /* FACE: barycentric coordinates */
    double vn[3];                   /* hit normal (ONE normal for all three vertices, tools/...:241) */
} pbso_force_msg;

/* ModalSolver::enqueueForceMessage (modal_solver.h:329-333): 1 = enqueued,
 * 0 = the 1023-slot queue is full.                                            */
int pbso_enqueue_force(pbso_engine *e, int object_id, const pbso_force_msg *m, int64_t not_before);
/* The same for a pre-scheduled force script (the throughput harness drives the
 * engine from a script instead of a GUI thread): n messages, message i for object_ids[i] with
 * stamp not_before[i], enqueued in order.  accepted[i] (may be NULL) receives each call's
 * result; returns the number enqueued, or a negative pbso_status on the first hard error.     */
int pbso_enqueue_force_batch(pbso_engine *e, int n, const int *object_ids, const pbso_force_msg *msgs,
                             const int64_t *not_before, unsigned char *accepted);
/* A step's worth of PLAIN VERTEX HITS -- what the tool does on a mouse click: GetModalForceVertex with a fresh
 * PointForce, then enqueueForceMessage (tools/real_time_modal_sound.cpp:268-295, 594-622) -- as parallel arrays,
 * object by object (object_ids ascending, not_before ascending within an object): hit i strikes vertex vids[i] of
 * object object_ids[i] along vn[3 i .. 3 i + 2] at buffer stamp not_before[i].  Semantically n calls of
 * pbso_enqueue_force in that order, except that nothing is copied: the arrays are BORROWED until the next pbso_step
 * returns.  That step consumes them -- the hits of an idle object with an empty queue go straight into the step's
 * descriptors (no queue round trip), the others, and the hits stamped beyond the step, enter the object's queue as if
 * enqueued one by one (a full queue, 1023 slots, rejects a hit exactly as enqueueForceMessage returns false for it,
 * modal_solver.h:329-333: the hit is dropped and counted in pbso_engine_info::total_dropped_hits).  One script may be
 * pending at a time; any other enqueue call for the engine first moves a pending script into the queues (order is kept).
 * Returns n, or PBSO_ERR_INVALID (ids / vertex ids out of range, object order) / PBSO_ERR_STATE (a script is pending). */
int pbso_enqueue_vertex_hits(pbso_engine *e, int n, const int *object_ids, const int *vids, const double *vn,
                             const int64_t *not_before);
/* ModalSolver::enqueueArprmMessageNoFail (modal_solver.h:382-393); 1-slot queue: a message that finds the slot taken waits in
 * the engine and enters when step() has taken the one before it (the reference's caller spins on try_enqueue meanwhile) */
int pbso_enqueue_arprm(pbso_engine *e, int object_id, const double a[2], double sigma,
                       double mu, int64_t not_before);
/* ABI 6.  1 while the object's AR-parameter slot holds a message that step() has not taken yet (or one is waiting for it): what
 * ModalSolver::enqueueArprmMessage's try_enqueue would find full (modal_solver.h:378-381); 0 when a try_enqueue would succeed.  A
 * caller that wants the reference's bounded spin (enqueueArprmMessageNoFail with maxIte >= 0) polls this between attempts. */
int pbso_arprm_pending(pbso_engine *e, int object_id);
/* ModalSolver::computeTransfer(pos) (modal_solver.h:286-300): FFAT lookup for
 * every mode on the device, result offered to the 1-slot transfer queue.
 * 1 = enqueued, 0 = no maps / queue still full.                               */
int pbso_compute_transfer(pbso_engine *e, int object_id, const double pos[3], int64_t not_before);
/* A listener PATH, the twin of pbso_enqueue_force_batch for ModalSolver::computeTransfer(pos) (modal_solver.h:286-300; the
 * tool calls it from its camera callback, tools/real_time_modal_sound.cpp:844, 1172): n positions, position i for object
 * object_ids[i] at stamp not_before[i], issued in order -- semantically n calls of pbso_compute_transfer, one entry into the
 * library.  accepted[i] (may be NULL) receives each call's result; returns the number accepted, or a negative pbso_status on
 * the first hard error (bad id, PBSO_ERR_MISSING_MAP).  The arrays are read before the call returns.                        */
int pbso_compute_transfer_path(pbso_engine *e, int n, const int *object_ids, const double *pos, const int64_t *not_before,
                               unsigned char *accepted);
/* ModalSolver::computeTransfer(pos, T *trans) (modal_solver.h:302-315), batched over n_pos listener
 * positions.  Like the reference it writes _ffat_maps->size() (= pbso_object_n_maps) entries per position:
 * out[n_pos][out_cols] doubles, columns [0, n_maps) of every row are written, the others left alone;
 * out_cols < n_maps is PBSO_ERR_INVALID.  1 = done, 0 = the object has no maps.                         */
int pbso_compute_transfer_batch(pbso_engine *e, int object_id, const double *pos, int n_pos, double *out, int out_cols);
/* Multi-listener output (the reference evaluates the transfer for many positions, modal_solver.h:302-315,
 * tools/...:916-927, but mixes one listener): after pbso_listeners_enable(obj) every pbso_step keeps the block-start
 * states of that object (block forms only), and pbso_mix_listeners returns the LAST step's audio of the object as heard
 * at each of n_listeners positions -- out[n_listeners][n_buffers * 513] floats, host memory -- i.e. what n_listeners
 * ModalSolvers fed the same forces and computeTransfer(pos_l) would emit.  The per-sample dot q . transfer_l becomes an
 * [L x 16 samples x 2M] . [2M x 16 blocks] contraction on the f32 matrix pipe.  PBSO_ERR_STATE if the last step holds a
 * buffer of the object with a dense force profile (Gaussian / AR: stepped per sample, no block states).            */
int pbso_listeners_enable(pbso_engine *e, int object_id);
int pbso_mix_listeners(pbso_engine *e, int object_id, const double *pos, int n_listeners, float *out, size_t n_out);
/* _ffat_maps->size() of an object (modal_solver.h:312); 0 while readFFATMaps has not been called */
int pbso_object_n_maps(pbso_engine *e, int object_id);
/* ModalSolver::setUseTransfer (modal_solver.h:148-152) */
int pbso_set_use_transfer(pbso_engine *e, int object_id, int use, int64_t not_before);
/* ModalSolver::getLatestTransfer (modal_solver.h:145-147): n_modes doubles */
int pbso_get_latest_transfer(pbso_engine *e, int object_id, double *out);

/* --- ModalSolver::step (modal_solver.h:181-276) for every object, n_buffers
 * times, one launch.  Asynchronous on the engine's stream.                    */
int pbso_step(pbso_engine *e, int n_buffers);
/* same, audio written to a caller-owned device buffer [n_objects][n_buffers*B] fp32 */
int pbso_step_into(pbso_engine *e, int n_buffers, void *d_audio);
/* ABI 6, engines with submit_thread > 0: returns when every launch of the pbso_step calls so far is in its stream (NOT when the device is
 * done: that is pbso_sync) -- from then on work of the caller on the engine's stream is ordered behind them.  A no-op otherwise. */
int pbso_flush(pbso_engine *e);
/* ... and delivered to the HOST, where the reference's consumer lives (the PortAudio callback takes SoundMessages from a host
 * queue, modal_solver.h:79-82, 346-363; tools/real_time_modal_sound.cpp:192-212): host_out[n_objects][n_buffers * B] fp32.
 * host_out from pbso_host_alloc (or any pinned, device-mapped host memory): the oscillator bank writes its samples straight
 * into it over PCIe -- no copy pass; the step then runs at the link's rate.  Pageable memory: the bank writes device memory
 * and a copy follows on its own stream (two device buffers in turn).  Asynchronous either way: pbso_host_wait blocks until the
 * LAST pbso_step_to_host's samples are in host_out; a caller that alternates two host buffers consumes step k while step
 * k + 1 runs.                                                                                                              */
int pbso_step_to_host(pbso_engine *e, int n_buffers, float *host_out, size_t n_floats);
int pbso_host_wait(pbso_engine *e);
int pbso_host_alloc(size_t bytes, void **out);      /* pinned host memory (hipHostMalloc) for callers that do not link HIP */
void pbso_host_free(void *p);
int pbso_sync(pbso_engine *e);

/* results of the LAST step.  audio is the reference's SoundMessage::data
 * (pressure units; PaModalCallback divides by 1e10, tools/...:207-210), fp32,
 * [n_objects][n_buffers * frames_per_buffer].  emitted[obj][buf] is 0 where
 * the reference's step() returned early without a buffer (clearAllForces,
 * modal_solver.h:186-189); those samples are 0.                               */
int pbso_read_audio(pbso_engine *e, float *host_out, size_t n_floats);
/* the rows of some objects only: host_out[n_rows][n_buffers * frames_per_buffer] */
int pbso_read_audio_rows(pbso_engine *e, const int *object_ids, int n_rows, float *host_out);
int pbso_read_emitted(pbso_engine *e, unsigned char *host_out, size_t n);
/* getQBufferNorm (modal_solver.h:153-159) for (object, buffer) of the last step */
int pbso_read_qnorm(pbso_engine *e, int object_id, int buffer, float *host_out, int n);
/* integrator state after the last step: q_{k-1}, q_{k-2} (modal_integrator.h:24) */
int pbso_read_state(pbso_engine *e, int object_id, double *q1, double *q2, int n);
/* the counterpart (checkpoint / resume: the reference has none; its integrator state is the ring
 * _q[3] of modal_integrator.h:24,29): sets q_{k-1}, q_{k-2} of the first n modes of an object (rounded once to
 * fp32; modes beyond n keep their state).  Force lists, queues and the transfer in effect are not part of it:
 * a resumed run re-sends its pending messages.  Waits for the launches in flight.                              */
int pbso_write_state(pbso_engine *e, int object_id, const double *q1, const double *q2, int n);
void *pbso_audio_device_ptr(pbso_engine *e);

/* Sum over the engine's objects of the LAST step's audio, on the device, in object order (deterministic):
 * d_out[n_buffers * frames_per_buffer] fp32 (a device pointer).  What a consumer of ONE mixed stream plays when the
 * scene's objects sound together (the reference mixes nothing: one ModalSolver, one PortAudio stream,
 * tools/real_time_modal_sound.cpp:192-212); also the per-rank half of PBSO_GATHER_MIX below.  Asynchronous on the engine's stream. */
int pbso_mix_objects(pbso_engine *e, void *d_out);

/* --- device group ("create/destroy engine (sample rate, buffer size 513, device list)", 8(e)) -------------
 * Objects are independent -- every ModalSolver owns its integrator state, force list and maps, modal_solver.h:100-126 -- so
 * a job of many objects shards over the GPUs of a node with no exchange while stepping: each RANK (one GPU, one engine) owns a
 * contiguous block of objects, balanced by the sum of modes.  RCCL over xGMI is used only to gather finished audio buffers,
 * called from here (C++), not from a Python launcher: one process may drive several GPUs (devices[]), or one process per GPU
 * joins a job of world_size ranks through a shared unique id (ncclCommInitRank).  librccl is loaded when the first group
 * with more than one rank is created; an engine alone never needs it.                                                    */
typedef struct pbso_group pbso_group;
enum pbso_gather_mode {
    PBSO_GATHER_ALL = 1,      /* every rank receives every object's buffers: ncclAllGather, in place (each engine writes its buffers
                                 straight into its slice of the gather target) */
    PBSO_GATHER_ROOT = 2,     /* only rank 0 receives them: ncclSend / ncclRecv */
    PBSO_GATHER_MIX = 3       /* the consumer wants ONE mixed stream: every rank sums its objects' buffers on the device
                                 (pbso_mix_objects) and the ranks all-reduce n_buffers * 513 floats */
};
#define PBSO_GROUP_ID_BYTES 128
enum pbso_group_transport {
    PBSO_GROUP_RCCL = 0,        /* the product: RCCL whenever the job has more than one rank (a one-rank group has nothing to exchange) */
    PBSO_GROUP_RCCL_ALWAYS = 1, /* tests: a one-rank group builds its communicator too (ncclCommInitRank) and issues every collective --
                                   the in-place ncclAllGather, ncclAllReduce, and for GATHER_ROOT an ncclSend / ncclRecv pair to itself
                                   whose received rows are the result -- so the RCCL code runs on a one-GPU box */
    PBSO_GROUP_LOOPBACK = 2     /* tests: ALL ranks of the job in this process, several of them on one device (devices[] may repeat); the
                                   collectives are plain device copies.  Everything the group does around them -- shards, padding of
                                   ragged shards, slice offsets, the two gather targets and their events, ROOT / MIX -- is the product's */
};
typedef struct pbso_group_desc {
    int abi_version;          /* PBSO_ABI_VERSION */
    const int *devices;       /* HIP ordinals THIS process drives: one rank (engine) each */
    int n_devices;
    int world_size;           /* ranks of the whole job; 0 -> n_devices (a single process) */
    int first_rank;           /* rank of devices[0]; this process owns ranks first_rank .. first_rank + n_devices - 1 */
    const void *unique_id;    /* world_size > n_devices: the PBSO_GROUP_ID_BYTES bytes pbso_group_unique_id() gave ONE process,
                                 handed to all of them by the launcher (a file, an environment variable, a store) */
    pbso_engine_desc engine;  /* settings of every engine; `device` and `stream` are the group's */
    /* ---- ABI 5 */
    int transport;            /* enum pbso_group_transport; 0 = the product's */
} pbso_group_desc;
int pbso_group_unique_id(void *out_bytes);
int pbso_group_create(const pbso_group_desc *d, pbso_group **out);
void pbso_group_destroy(pbso_group *g);
const char *pbso_group_last_error(const pbso_group *g);
/* the job: mode counts of ALL its objects, the same list in every process.  Computes the shards -- contiguous blocks whose cut
 * points are the object boundaries nearest to the ideal prefix sums of the mode counts.                                    */
int pbso_group_plan(pbso_group *g, const int *modes_per_object, int n_objects);
/* the rule by itself (no group, no GPU): cuts[world_size + 1], rank r owns ids [cuts[r], cuts[r + 1]) */
int pbso_shard_by_modes(const int *modes_per_object, int n_objects, int world_size, int *cuts);
int pbso_group_rank_span(pbso_group *g, int rank, int *lo, int *hi);          /* global ids [lo, hi) of a rank */
int pbso_group_owner(pbso_group *g, int global_id, int *rank, int *local_id);
/* BuildSolver for object global_id (pbso_add_object on its owner), in ascending id order; ids of ranks in OTHER processes are
 * accepted and ignored, so every process may run the same loop over the job's objects.                                    */
int pbso_group_add_object(pbso_group *g, int global_id, const pbso_object_desc *d);
int pbso_group_finalize(pbso_group *g);
pbso_engine *pbso_group_engine(pbso_group *g, int rank);                      /* NULL for a rank of another process */
/* pbso_enqueue_force on the object's owner (1 / 0 / < 0); an object of another process: 1, nothing done (its owner does it) */
int pbso_group_enqueue_force(pbso_group *g, int global_id, const pbso_force_msg *m, int64_t not_before);
/* ModalSolver::step n_buffers times on every local rank; each engine writes into its slice of the group's gather target
 * (two targets used in turn: the gather of step k runs beside the oscillator bank of step k + 1)                          */
int pbso_group_step(pbso_group *g, int n_buffers);
/* the collective for the LAST step, asynchronous (its own stream per rank, ordered behind the step); enum pbso_gather_mode  */
int pbso_group_gather(pbso_group *g, int mode);
int pbso_group_sync(pbso_group *g);
/* the last gather's result on a local rank, a device pointer: ALL -> [world_size * rows_per_rank][n_buffers * 513] (rank r's
 * objects from row r * rows_per_rank; shards smaller than the largest are padded with silent rows), ROOT -> the same on rank 0
 * and the rank's own rows elsewhere, MIX -> [n_buffers * 513].  rows / row_floats (may be NULL) receive the shape.          */
void *pbso_group_result_device_ptr(pbso_group *g, int rank, size_t *rows, size_t *row_floats);
int pbso_group_read_result(pbso_group *g, int rank, float *host_out, size_t n_floats);   /* that buffer, synchronously */

/* PaModalCallback body (tools/real_time_modal_sound.cpp:207-210): mono sound
 * -> interleaved stereo float32 scaled by 1e-10.                              */
void pbso_pa_convert(const float *sound, unsigned long frames, float *out_stereo);

/* --- introspection -------------------------------------------------------- */
typedef struct pbso_engine_info {
    int n_objects;
    int frames_per_buffer;
    int modes_padded;         /* oscillators per object after padding */
    int modes_per_lane;
    int waves_per_object;     /* largest team (workgroup) in waves */
    int lds_bytes_per_workgroup;
    int64_t buffers_done;
    double last_step_kernel_ms;       /* HIP-event time of the oscillator-bank kernel, last step */
    double last_step_device_ms;       /* HIP-event time of the whole device pipeline, last step */
    double last_step_host_plan_ms;    /* host bookkeeping (modal_solver.h:184-256) */
    int64_t last_step_forced_rows;
    int64_t last_step_transfer_rows;
    /* running totals since engine creation (HIP events on the engine's stream,
     * one pair per pbso_step around the oscillator-bank kernel / the pipeline) */
    double total_kernel_ms;
    double total_device_ms;
    double total_host_plan_ms;
    int64_t total_steps;
    int n_teams;              /* workgroups of the oscillator bank per launch (objects with more than
                               * 16 waves of modes are stepped by several) */
    int recurrence_form;      /* the form that runs (PBSO_FORM_BLOCK falls back to VELOCITY when
                               * frames_per_buffer != 513) */
    int64_t total_block_launches;     /* oscillator-bank launches on the block kernel (K1b) ...            */
    int64_t total_sample_launches;    /* ... and on the per-sample kernel (K1): every launch of the per-sample
                                       * forms, and launches of the block form in which more than half of the
                                       * (object, buffer) pairs carry a dense force profile (sustained contact) */
    int64_t total_timed_launches;     /* launches whose HIP-event times are in total_kernel_ms / total_device_ms: all of
                                       * them, or every n-th with pbso_engine_desc::timing_every = n (an event pair costs
                                       * the stream ~8 us per launch)                                                 */
    int64_t total_split_launches;     /* of the block launches, those on the pipeline kernel of small scenes (K1p, kernels_pipe.hip:
                                       * a producer wave and two consumer waves per 64 modes)                               */
    int64_t total_time_chunk_launches;/* of the block launches, those cut along the time axis (K5, kernels_scan.hip: a scan of the
                                       * buffer-start states, then the block kernel over (team, chunk of buffers) workgroups) */
    int64_t total_dropped_hits;       /* hits of pbso_enqueue_vertex_hits scripts that found their object's 1023-slot queue full:
                                       * rejected, as enqueueForceMessage would have been (modal_solver.h:329-333)              */
    int64_t total_one_stream_launches;/* launches whose preparation ran on the bank's own stream (the latency path: a short step
                                       * submitted while the device was idle -- the real-time facade's pattern)                   */
    /* ---- ABI 5 */
    int64_t total_dense_increment_launches; /* of the time-chunked launches, those with dense-profile buffers (Gaussian / AR: forces.h:92-128):
                                       * dense_increment_kernel evaluated what each leaves in the state, all of them at once      */
    int64_t total_segmented_scans;    /* ... whose scan of chunk-start states ran cut along the time axis (one wave per chunk)             */
    int last_time_chunk_shape;        /* the last time-chunked launch: modes per lane of its teams (0: none yet), ...              */
    int last_time_chunk_buffers;      /* ... buffers per chunk, ...                                                               */
    int last_time_chunk_teams;        /* ... and teams per chunk (its census has teams x chunks rows, chunk-major)                */
    /* ---- ABI 6 */
    int start_gate;                   /* the start gate of long launches (pbso_engine_desc::stream_sync): 0 none (asked for, or no interface),
                                       * 1 a device-side wait (hipStreamWaitValue64), 2 the submitting thread waits on pinned host memory,
                                       * -1 none because the environment serialises kernel dispatches (a waiting kernel would never end)   */
    int64_t total_gate_timeouts;      /* host-side gate only: waits that gave up after 2 s (the launch then went ungated)                 */
    double total_host_submit_ms;      /* the caller's thread in pbso_step behind the planner: packing the plan, the upload and launch calls -- or, with
                                       * submit_thread, recording them (the wait for a free plan set is not in it)                        */
    int64_t total_ffat_shared_events; /* listener events located once per EVENT (objects whose modes share one map geometry, round 6) ...   */
    int64_t total_ffat_general_events;/* ... and those evaluated per (event, mode)                                                          */
} pbso_engine_info;
int pbso_get_info(pbso_engine *e, pbso_engine_info *out);
/* diagnostics (engine created with env PBSO_CENSUS=1): for every object's
 * workgroup of the last oscillator-bank launch: start, end (100 MHz ticks),
 * HW_REG_HW_ID, HW_REG_XCC_ID, shader-clock count at start and end.
 * out[n_teams][12]; words 6..9 (block form): shader cycles of wave 0 spent in the buffer head, the MFMA pipeline, the barrier, the combine.
 * A launch on the kernel of under-filled engines (total_split_launches) has one row per team of 64 modes -- n may then be that
 * count x 12 --: words 0..2 the producer's cycles (head | stepping | wait at the barrier), 3 its HW_ID | XCC_ID << 32, 6..8
 * consumer 0's (head + taps + qnorm chains | projection | wait), 9, 10 the consumers' HW_ID words, 11 qnorm chain groups stepped
 * in unit-force form.                                                                                                            */
int pbso_read_census(pbso_engine *e, unsigned long long *out, size_t n);

#ifdef __cplusplus
}
#endif
#endif
