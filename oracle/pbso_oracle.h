/*
 * pbso_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * fp64 CPU restatement of the openpbso modal-sound hot path (the reference's
 * ModalSolver<double>::step and everything it calls).  It is the checker for
 * the HIP engine: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (openpbso_amd/) never links,
 * imports or calls anything in this directory.
 *
 * PARITY PIN STATUS.  The reference ships no tests, golden vectors or
 * fixtures for this path, and the path cannot be built
 * here (Eigen, libigl, protobuf C++ and PortAudio are absent).
 * What pins this oracle:
 *   - the known-answer vectors of SURVEY.md Appendix A (produced at survey
 *     time from the reference's unmodified modal_integrator.h / forces.h),
 *     checked in tests/test_oracle_kat.py;
 *   - oracle/_ref: the reference's own ModeData.h and ModalMaterial.h
 *     (std-only headers) compiled where they lie, for the loader formats;
 *   - independent formulations (scipy.signal.lfilter, closed-form impulse
 *     response, brute-force numpy FFAT lookup, Python protobuf encoder).
 * A3/A4/A6/A7 (solver step, projection, FFAT lookup) are line-by-line
 * restatements with NO reference-produced vector behind them: for those rows
 * parity is unpinned beyond the cross-checks above.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).
 */
#ifndef PBSO_ORACLE_H
#define PBSO_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OR_SAMPLE_RATE 44100        /* config.h:13 */
#define OR_FRAMES_PER_BUFFER 513    /* config.h:14 */

/* ---- A1: ModalIntegrator<double>::Build + ctor (modal_integrator.h:47-101) */
void or_build_ab(double density, const double *omega_squared, int n,
                 double alpha, double beta, double *a, double *b);
void or_iir_coeffs(const double *a, const double *b, int n, double h,
                   double *c1, double *c2, double *c3);

/* ---- A2: ModalIntegrator<double>::Step (modal_integrator.h:103-123) */
typedef struct or_integrator {
    int n;
    double h;
    double *c1, *c2, *c3;
    double *q[3];   /* 3-slot ring, modal_integrator.h:24 */
    int cur;        /* _q_curr_ptr, modal_integrator.h:29 */
} or_integrator;
or_integrator *or_integrator_build(double density, const double *omega_squared,
                                   int n_omega, double alpha, double beta,
                                   double h, int n);
void or_integrator_free(or_integrator *it);
const double *or_integrator_step(or_integrator *it, const double *Q);
const double *or_integrator_step_free(or_integrator *it);

/* ---- libstdc++ std::default_random_engine + std::normal_distribution<double>
 *      (the third-party dependency forces.h:71-72 leans on; GCC 11 libstdc++,
 *      bits/random.h minstd_rand0, bits/random.tcc generate_canonical and
 *      normal_distribution::operator()).  Restated; checked against the real
 *      <random> in tests. */
typedef struct or_rng {
    uint32_t x;         /* minstd_rand0 state, default seed 1 */
    double saved;
    int saved_available;
} or_rng;
void or_rng_init(or_rng *r);
double or_rng_normal(or_rng *r);

/* ---- A5: forces.h */
enum { OR_POINT_FORCE = 0, OR_GAUSSIAN_FORCE = 1, OR_AR_FORCE = 2 };  /* forces.h:12-16 */
typedef struct or_force {
    int type;
    /* PointForce (forces.h:25-31) */
    int used;
    /* GaussianForce (forces.h:33-48) */
    double width;
    int width_samples, count, center, cutoff;
    /* AutoregressiveForce (forces.h:60-79) */
    double buf[3];
    int buf_idx;
    double a[2], sigma, mu;
    or_rng rng;
} or_force;
void or_force_init_point(or_force *f);
void or_force_init_gaussian(or_force *f, double width_us);
void or_force_init_ar(or_force *f);
void or_force_ar_set_param(or_force *f, const double a[2], double sigma, double mu);
/* returns 1 if the force contributed (Force::Add), forces.h:81-128 */
int or_force_add(or_force *f, double *buf /* [OR_FRAMES_PER_BUFFER] */);

/* ---- A6: GetModalForceVertex / GetModalForceFace
 *      (tools/real_time_modal_sound.cpp:268-295, 236-266).  modes is
 *      mode-major: modes[m*ndof + 3*vid + c] (ModeData.h:24). */
void or_modal_force_vertex(int n, const double *modes, int ndof, int vid,
                           const double vn[3], double *data);
void or_modal_force_face(int n, const double *modes, int ndof,
                         const int vids[3], const double coords[3],
                         const double vn[3], double *data);

/* ---- A7: FFAT_Map<double,3> runtime subset (ffat_solver.h) */
typedef struct or_ffat_map {
    int mode_id;
    double k;               /* FFAT_Map<T,3>::_k */
    double center3[3];      /* FFAT_Map<T,3>::_center */
    int is_compressed;
    /* _shells[2] : FFAT_Map<T,1> */
    double cell_size;
    double low_corners[6][3];
    int n_elements[6][2];
    int strides[6];
    double center[3];
    double bbox_low[3];
    double bbox_top[3];
    int n_psi;
    double *psi;            /* _Psi(:,0) */
} or_ffat_map;
void or_ffat_intersect(const or_ffat_map *m, const double p[3],
                       double surf[3], int map_ind[3]);
void or_ffat_interpolate(const or_ffat_map *m, const double surf[3],
                         const int nn[3], int idx[4][3], double coeffs[4]);
int or_ffat_quad_stride(const or_ffat_map *m, const int mi[3]);
double or_ffat_get_map_val(const or_ffat_map *m, const double p[3]);
/* synthetic uniform cube, geometry as ResampleToUniformCube (ffat_solver.h:538-558) */
void or_ffat_make_uniform_cube(or_ffat_map *m, int mode_id, double k,
                               const double center[3], double cell_size, int dim,
                               const double *psi /* [6*dim*dim] */);
void or_ffat_free(or_ffat_map *m);

/* ---- A8: loaders */
/* ModeData<double>::read (ModeData.h:61-83). Caller frees *omega2, *modes. */
int or_modes_read(const char *path, int *ndof, int *nmodes, double **omega2,
                  double **modes);
int or_modes_write(const char *path, int ndof, int nmodes, const double *omega2,
                   const double *modes);
/* ModeData<double>::numModesAudible (ModeData.h:120-148) */
int or_num_modes_audible(const double *omega2, int nmodes, double density,
                         double audible_freq);
/* ModalMaterial<double>::Read (ModalMaterial.h:35-55); out = density, youngs,
 * poisson, alpha, beta.  returns 0 on success. */
int or_material_read(const char *path, double out[5]);
/* FFAT_Map_Serialize_Double::Load (ffat_map_serialize.h:166-254) from bytes. */
int or_fatcube_parse(const unsigned char *bytes, size_t n, or_ffat_map *out);
int or_fatcube_load(const char *path, or_ffat_map *out);

/* ---- A3 + A4: ModalSolver<double>::step (modal_solver.h:181-276) */
typedef struct or_force_msg {
    double *data;           /* length n (owned) */
    int n;
    int force_type;         /* ForceMessage::forceType */
    or_force force;         /* deep-copied with the message, modal_solver.h:39-76 */
    int sustained_start, sustained_end, clear_all;
} or_force_msg;

typedef struct or_solver or_solver;
or_solver *or_solver_new(int n_modes);
void or_solver_free(or_solver *s);
/* takes ownership of the integrator (setIntegrator, modal_solver.h:142-144) */
void or_solver_set_integrator(or_solver *s, or_integrator *it);
/* copies maps (readFFATMaps); maps[i].mode_id is the std::map key */
void or_solver_set_ffat_maps(or_solver *s, const or_ffat_map *maps, int n);
/* returns 1 on success, 0 if the 1023-slot queue is full */
int or_solver_enqueue_force(or_solver *s, const double *data, int n,
                            const or_force *force, int sustained_start,
                            int sustained_end, int clear_all);
int or_solver_enqueue_arprm(or_solver *s, const double a[2], double sigma, double mu);
/* computeTransfer(pos): returns 1 if enqueued, 0 if dropped/no maps, -1 if a
 * modeId is missing (std::map::at would throw). modal_solver.h:286-300 */
int or_solver_compute_transfer(or_solver *s, const double pos[3]);
/* computeTransfer(pos, T*): modal_solver.h:302-315 */
int or_solver_compute_transfer_out(or_solver *s, const double pos[3], double *out);
void or_solver_set_use_transfer(or_solver *s, int use);
const double *or_solver_latest_transfer(const or_solver *s);
/* one buffer. returns 1 and fills sound[513], qnorm[n_modes] (either may be
 * NULL); returns 0 if the step returned early (clearAllForces). */
int or_solver_step(or_solver *s, double *sound, double *qnorm);
int or_solver_n_active(const or_solver *s);
const double *or_solver_state(const or_solver *s, int which /*0: q_{k-1}, 1: q_{k-2}*/);

/* ---- A9: PaModalCallback scaling (tools/real_time_modal_sound.cpp:207-210) */
void or_pa_callback_convert(const double *sound, int frames, float *out_stereo);

/* ---- throughput leg for bench.py's cpu_baseline: steps n_obj independent
 *      solvers (one PointForce at buffer 0 each, unit transfer) for n_buffers,
 *      objects spread over OpenMP threads; returns seconds. */
double or_bench_run(int n_obj, int n_modes, int n_buffers, int n_threads,
                    const double *omega2 /* [n_obj*n_modes] */, double density,
                    double alpha, double beta, const double *hit_data /* [n_obj*n_modes] */,
                    const unsigned char *hit_mask /* [n_obj*n_buffers] */,
                    double *sound_out /* [n_obj*n_buffers*513] or NULL */,
                    int flush_denormals);

#ifdef __cplusplus
}
#endif
#endif
