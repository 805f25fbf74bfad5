/*
 * pbso_oracle.c -- TEST INFRASTRUCTURE ONLY (see pbso_oracle.h for the
 * parity-pin status).  fp64 restatement of openpbso's ModalSolver hot path.
 * Build with -ffp-contract=off: the reference's CMakeLists.txt sets no
 * optimisation/arch flags, so its arithmetic is un-fused IEEE double.
 */
#include "pbso_oracle.h"

#include <dirent.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__x86_64__)
#include <xmmintrin.h>
#include <pmmintrin.h>
#endif

#define B OR_FRAMES_PER_BUFFER

/* ========================================================================== */
/* A1  modal_integrator.h:47-70  (Build: Rayleigh damping -> a, b)             */
/* ========================================================================== */
void or_build_ab(double density, const double *omega_squared, int n,
                 double alpha, double beta, double *a, double *b) {
    for (int ii = 0; ii < n; ++ii) {
        double omega = sqrt(omega_squared[ii] / density);      /* :63 */
        double xi = 0.5 * (alpha / omega + beta * omega);      /* :64 */
        a[ii] = 2.0 * xi * omega;                              /* :65 */
        b[ii] = pow(omega, 2);                                 /* :66 */
    }
}

/* modal_integrator.h:86-100 (ctor: IIR coefficients, DyRT convention) */
void or_iir_coeffs(const double *a, const double *b, int n, double h,
                   double *c1, double *c2, double *c3) {
    for (int ii = 0; ii < n; ++ii) {
        double epsilon = exp(-a[ii] / 2 * h);                          /* :89 */
        double theta = h * sqrt(b[ii] - a[ii] * a[ii] / 4.0);          /* :90 */
        double gamma = asin(a[ii] / (2.0 * sqrt(b[ii])));              /* :91 */
        double omega = sqrt(b[ii]);                                    /* :92 */
        double omega_d = sqrt(b[ii] - pow(a[ii], 2) / 4.0);            /* :93 */
        c1[ii] = 2.0 * epsilon * cos(theta);                           /* :95 */
        c2[ii] = -pow(epsilon, 2);                                     /* :96 */
        c3[ii] = 2.0 * (epsilon * cos(theta + gamma)
                        - pow(epsilon, 2) * cos(2.0 * theta + gamma)); /* :97 */
        c3[ii] /= (3.0 * omega * omega_d);                             /* :98 */
        c3[ii] *= 1E9;                                                 /* :99 */
    }
}

or_integrator *or_integrator_build(double density, const double *omega_squared,
                                   int n_omega, double alpha, double beta,
                                   double h, int n) {
    if (n < 0) n = n_omega;                       /* :53-55 */
    if (n > n_omega) return NULL;                 /* assert :56 */
    or_integrator *it = (or_integrator *)calloc(1, sizeof(*it));
    it->n = n;
    it->h = h;
    size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(double);
    double *a = (double *)malloc(nb), *b = (double *)malloc(nb);
    it->c1 = (double *)malloc(nb);
    it->c2 = (double *)malloc(nb);
    it->c3 = (double *)malloc(nb);
    for (int i = 0; i < 3; ++i) it->q[i] = (double *)calloc(n > 0 ? n : 1, sizeof(double));
    or_build_ab(density, omega_squared, n, alpha, beta, a, b);
    or_iir_coeffs(a, b, n, h, it->c1, it->c2, it->c3);
    free(a);
    free(b);
    it->cur = 0;
    return it;
}

void or_integrator_free(or_integrator *it) {
    if (!it) return;
    free(it->c1); free(it->c2); free(it->c3);
    for (int i = 0; i < 3; ++i) free(it->q[i]);
    free(it);
}

/* ========================================================================== */
/* A2  modal_integrator.h:103-113 (forced) and :115-123 (free)                 */
/* ========================================================================== */
const double *or_integrator_step(or_integrator *it, const double *Q) {
    double *qk = it->q[(it->cur + 1) % 3];
    const double *q1 = it->q[(it->cur) % 3];
    const double *q2 = it->q[(it->cur + 2) % 3];
    const double *c1 = it->c1, *c2 = it->c2, *c3 = it->c3;
    const int n = it->n;
    for (int i = 0; i < n; ++i)                       /* :109-110 */
        qk[i] = (c1[i] * q1[i] + c2[i] * q2[i]) + c3[i] * Q[i];
    it->cur = (it->cur + 1) % 3;
    return qk;
}

const double *or_integrator_step_free(or_integrator *it) {
    double *qk = it->q[(it->cur + 1) % 3];
    const double *q1 = it->q[(it->cur) % 3];
    const double *q2 = it->q[(it->cur + 2) % 3];
    for (int i = 0; i < it->n; ++i)                   /* :120 */
        qk[i] = it->c1[i] * q1[i] + it->c2[i] * q2[i];
    it->cur = (it->cur + 1) % 3;
    return qk;
}

/* ========================================================================== */
/* libstdc++ <random>: minstd_rand0, generate_canonical<double,53>,            */
/* normal_distribution<double> (Marsaglia polar), GCC 11 bits/random.tcc       */
/* ========================================================================== */
void or_rng_init(or_rng *r) {
    r->x = 1u;                  /* default_seed */
    r->saved = 0.0;
    r->saved_available = 0;
}

static uint32_t minstd_next(or_rng *r) {
    r->x = (uint32_t)(((uint64_t)r->x * 16807ull) % 2147483647ull);
    return r->x;
}

static double canonical53(or_rng *r) {
    /* range R = max-min+1 = 2147483646; log2(R) floors to 30; m = (53+29)/30 = 2 */
    const long double R = 2147483646.0L;
    double sum = 0.0, tmp = 1.0;
    for (int k = 2; k != 0; --k) {
        sum += (double)(minstd_next(r) - 1u) * tmp;
        tmp = (double)((long double)tmp * R);
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

double or_rng_normal(or_rng *r) {
    double ret;
    if (r->saved_available) {
        r->saved_available = 0;
        ret = r->saved;
    } else {
        double x, y, r2;
        do {
            x = 2.0 * canonical53(r) - 1.0;
            y = 2.0 * canonical53(r) - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        const double mult = sqrt(-2 * log(r2) / r2);
        r->saved = x * mult;
        r->saved_available = 1;
        ret = y * mult;
    }
    return ret * 1.0 + 0.0;     /* stddev 1, mean 0 */
}

/* ========================================================================== */
/* A5  forces.h                                                                */
/* ========================================================================== */
void or_force_init_point(or_force *f) {
    memset(f, 0, sizeof(*f));
    f->type = OR_POINT_FORCE;
    f->used = 0;                                       /* forces.h:28 */
}

void or_force_init_gaussian(or_force *f, double width_us) {
    memset(f, 0, sizeof(*f));
    f->type = OR_GAUSSIAN_FORCE;
    f->width = width_us;
    f->cutoff = 5;                                     /* :40 */
    f->count = 0;                                      /* :38 */
    int ws = (int)(width_us / 1000000. * OR_SAMPLE_RATE);  /* :44 */
    f->width_samples = ws > 1 ? ws : 1;
    f->center = (int)((f->cutoff - 0.5) * f->width_samples); /* :45 */
}

void or_force_init_ar(or_force *f) {
    memset(f, 0, sizeof(*f));
    f->type = OR_AR_FORCE;
    f->buf[0] = f->buf[1] = f->buf[2] = 0.0;           /* :74 */
    f->buf_idx = 0;
    f->a[0] = 0.783; f->a[1] = 0.116;                  /* :75 */
    f->sigma = 0.00148; f->mu = 0.142;
    or_rng_init(&f->rng);                              /* default-constructed, :71 */
}

void or_force_ar_set_param(or_force *f, const double a[2], double sigma, double mu) {
    f->buf[0] = f->buf[1] = f->buf[2] = 0.0;           /* :133; _bufIdx and RNG untouched */
    f->a[0] = a[0]; f->a[1] = a[1];
    f->sigma = sigma;
    f->mu = mu;
}

static double ar_mu_effective(or_force *f) {            /* forces.h:107-117 */
    const int len = 3;
    double mu_tilde = 0.0;
    for (int ii = 0; ii < 2; ++ii)
        mu_tilde += f->a[ii] * f->buf[(f->buf_idx + len - ii - 1) % len];
    mu_tilde += f->sigma * or_rng_normal(&f->rng);
    f->buf[f->buf_idx] = mu_tilde;
    f->buf_idx = (f->buf_idx + 1) % len;
    return f->mu + mu_tilde;
}

int or_force_add(or_force *f, double *buf) {
    switch (f->type) {
    case OR_POINT_FORCE:                               /* forces.h:81-90 */
        if (f->used) return 0;
        buf[0] += 1.;
        f->used = 1;
        return 1;
    case OR_GAUSSIAN_FORCE:                            /* forces.h:92-105 */
        if (f->width == 0 || f->count >= f->cutoff * 2 * f->width_samples) return 0;
        for (int ii = 0; ii < B; ++ii) {
            const double p = -0.5 * pow((double)(f->count + ii - f->center)
                                        / (double)f->width_samples, 2);
            buf[ii] += exp(p);
        }
        f->count += B;
        return 1;
    case OR_AR_FORCE:                                  /* forces.h:119-128 */
        for (int ii = 0; ii < B; ++ii) buf[ii] += ar_mu_effective(f);
        return 1;
    }
    return 0;
}

/* ========================================================================== */
/* A6  tools/real_time_modal_sound.cpp:268-280 (vertex), :236-252 (face)       */
/* ========================================================================== */
void or_modal_force_vertex(int n, const double *modes, int ndof, int vid,
                           const double vn[3], double *data) {
    for (int mm = 0; mm < n; ++mm) {
        const double *u = modes + (size_t)mm * ndof;
        data[mm] = vn[0] * u[vid * 3 + 0]
                 + vn[1] * u[vid * 3 + 1]
                 + vn[2] * u[vid * 3 + 2];
    }
}

void or_modal_force_face(int n, const double *modes, int ndof,
                         const int vids[3], const double coords[3],
                         const double vn[3], double *data) {
    for (int mm = 0; mm < n; ++mm) {
        const double *u = modes + (size_t)mm * ndof;
        double acc = 0.0;                              /* setZero :243 */
        for (int jj = 0; jj < 3; ++jj) {
            acc += vn[0] * u[vids[jj] * 3 + 0] * coords[jj]
                 + vn[1] * u[vids[jj] * 3 + 1] * coords[jj]
                 + vn[2] * u[vids[jj] * 3 + 2] * coords[jj];
        }
        data[mm] = acc;
    }
}

/* ========================================================================== */
/* A7  ffat_solver.h runtime subset                                            */
/* ========================================================================== */
static double dmin(double x, double y) { return (y < x) ? y : x; }   /* std::min */
static double dmax(double x, double y) { return (x < y) ? y : x; }   /* std::max */
static int iclamp(int x, int l, int h) {                             /* :699-701 */
    int m = x > l ? x : l;
    return m < h ? m : h;
}

/* FFAT_Map<T,1>::Intersect, ffat_solver.h:676-712 */
void or_ffat_intersect(const or_ffat_map *m, const double p[3],
                       double surf[3], int map_ind[3]) {
    double d[3], t_enter[3];
    for (int i = 0; i < 3; ++i) {
        d[i] = m->center[i] - p[i];                          /* :681 */
        double tmin = (m->bbox_low[i] - p[i]) / d[i];        /* :682 */
        double tmax = (m->bbox_top[i] - p[i]) / d[i];        /* :683 */
        t_enter[i] = dmin(tmin, tmax);                       /* :684 */
    }
    double t_en = t_enter[0];                                /* maxCoeff :685 */
    for (int i = 1; i < 3; ++i) if (t_enter[i] > t_en) t_en = t_enter[i];
    for (int i = 0; i < 3; ++i) surf[i] = p[i] + t_en * d[i];/* :686 */
    double min_dist = 1.7976931348623157e308;                /* :688 */
    map_ind[0] = 0;  /* reference leaves it uninitialised if all compares fail */
    for (int dd = 0; dd < 3; ++dd) {
        if (fabs(m->bbox_low[dd] - surf[dd]) < min_dist) {   /* :690-693 */
            min_dist = fabs(m->bbox_low[dd] - surf[dd]);
            map_ind[0] = dd * 2 + 1;
        }
        if (fabs(m->bbox_top[dd] - surf[dd]) < min_dist) {   /* :694-697 */
            min_dist = fabs(m->bbox_top[dd] - surf[dd]);
            map_ind[0] = dd * 2;
        }
    }
    int dk = map_ind[0] / 2, di = (dk + 1) % 3, dj = (dk + 2) % 3;
    const double *low = m->low_corners[map_ind[0]];
    map_ind[1] = (int)floor((surf[di] - low[di]) / m->cell_size);   /* :705-706 */
    map_ind[2] = (int)floor((surf[dj] - low[dj]) / m->cell_size);   /* :707-708 */
    map_ind[1] = iclamp(map_ind[1], 0, m->n_elements[map_ind[0]][0] - 1);
    map_ind[2] = iclamp(map_ind[2], 0, m->n_elements[map_ind[0]][1] - 1);
}

/* FFAT_Map<T,1>::Interpolate, ffat_solver.h:736-803 */
void or_ffat_interpolate(const or_ffat_map *m, const double surf[3],
                         const int nn[3], int idx[4][3], double coeffs[4]) {
    int dk = nn[0] / 2, di = (dk + 1) % 3, dj = (dk + 2) % 3;
    int x, y, xp, yp;
    double tx, ty;
    const int Nx = m->n_elements[nn[0]][0];
    const int Ny = m->n_elements[nn[0]][1];
    const double *low = m->low_corners[nn[0]];
    const double h = m->cell_size;
    double x_float = (surf[di] - (low[di] + 0.5 * h)) / h;      /* :756 */
    double y_float = (surf[dj] - (low[dj] + 0.5 * h)) / h;      /* :757 */
    x = (int)floor(x_float);
    y = (int)floor(y_float);
    if (x < 0) { x = 0; xp = 0; tx = 0; }                       /* :762-766 */
    else if (x >= 0 && x < Nx - 1) { xp = x + 1; tx = x_float - (double)x; }
    else { x = Nx - 1; xp = Nx - 1; tx = 0; }
    if (y < 0) { y = 0; yp = 0; ty = 0; }                       /* :776-780 */
    else if (y >= 0 && y < Ny - 1) { yp = y + 1; ty = y_float - (double)y; }
    else { y = Ny - 1; yp = Ny - 1; ty = 0; }
    tx = dmin(dmax(tx, 0.0), 1.0);                              /* :791 */
    ty = dmin(dmax(ty, 0.0), 1.0);                              /* :792 */
    idx[0][0] = nn[0]; idx[0][1] = x;  idx[0][2] = y;           /* c00 */
    idx[1][0] = nn[0]; idx[1][1] = xp; idx[1][2] = y;           /* c10 */
    idx[2][0] = nn[0]; idx[2][1] = x;  idx[2][2] = yp;          /* c01 */
    idx[3][0] = nn[0]; idx[3][1] = xp; idx[3][2] = yp;          /* c11 */
    coeffs[0] = (1.0 - tx) * (1.0 - ty);                        /* :799-802 */
    coeffs[1] = tx * (1.0 - ty);
    coeffs[2] = (1.0 - tx) * ty;
    coeffs[3] = tx * ty;
}

/* FFAT_Map<T,1>::GetDataQuadStride, ffat_solver.h:141-144 */
int or_ffat_quad_stride(const or_ffat_map *m, const int mi[3]) {
    return m->strides[mi[0]] + mi[1] * m->n_elements[mi[0]][1] + mi[2];
}

/* FFAT_Map<T,3>::GetMapVal (:1180-1206) + FFAT_Solver<T,3>::Reconstruct (:899-906) */
double or_ffat_get_map_val(const or_ffat_map *m, const double p[3]) {
    double surf[3], coeffs[4];
    int nn[3], idx[4][3];
    or_ffat_intersect(m, p, surf, nn);
    or_ffat_interpolate(m, surf, nn, idx, coeffs);
    double psi0 = 0.0;
    for (int kk = 0; kk < 4; ++kk) {
        const int id = or_ffat_quad_stride(m, idx[kk]);
        psi0 += coeffs[kk] * m->psi[id];                        /* :1202 */
    }
    /* (p - _center).norm(): Eigen's unrolled redux over a fixed 3-vector
     * evaluates x^2 + (y^2 + z^2). */
    const double dx = p[0] - m->center3[0], dy = p[1] - m->center3[1],
                 dz = p[2] - m->center3[2];
    const double r = sqrt(dx * dx + (dy * dy + dz * dz));
    const double kr = m->k * r;                                 /* :904 */
    return fabs(psi0 / kr);                                     /* :905 */
}

/* uniform cube geometry: ffat_solver.h:538-558 */
void or_ffat_make_uniform_cube(or_ffat_map *m, int mode_id, double k,
                               const double center[3], double cell_size, int dim,
                               const double *psi) {
    memset(m, 0, sizeof(*m));
    m->mode_id = mode_id;
    m->k = k;
    m->cell_size = cell_size;
    for (int i = 0; i < 3; ++i) {
        m->center3[i] = center[i];
        m->center[i] = center[i];
        m->bbox_low[i] = 1.7976931348623157e308;
        m->bbox_top[i] = -1.7976931348623157e308;
    }
    for (int dd = 0; dd < 6; ++dd) {
        const int dk = dd / 2, di = (dk + 1) % 3, dj = (dk + 2) % 3;
        const int nml = dd % 2 == 0 ? +1 : -1;
        double *corner = m->low_corners[dd];
        if (nml == -1) corner[dk] = center[dk] - dim / 2 * cell_size;
        else           corner[dk] = center[dk] + dim / 2 * cell_size;
        corner[di] = center[di] - dim / 2 * cell_size;
        corner[dj] = center[dj] - dim / 2 * cell_size;
        m->n_elements[dd][0] = dim;
        m->n_elements[dd][1] = dim;
        m->strides[dd] = dd * dim * dim;
        for (int jj = 0; jj < 3; ++jj) {
            m->bbox_low[jj] = dmin(m->bbox_low[jj], corner[jj]);
            m->bbox_top[jj] = dmax(m->bbox_top[jj], corner[jj]);
        }
    }
    m->n_psi = 6 * dim * dim;
    m->psi = (double *)malloc(sizeof(double) * (size_t)m->n_psi);
    memcpy(m->psi, psi, sizeof(double) * (size_t)m->n_psi);
}

void or_ffat_free(or_ffat_map *m) {
    if (m && m->psi) { free(m->psi); m->psi = NULL; }
}

static void ffat_copy(or_ffat_map *dst, const or_ffat_map *src) {
    *dst = *src;
    dst->psi = (double *)malloc(sizeof(double) * (size_t)(src->n_psi > 0 ? src->n_psi : 1));
    if (src->n_psi > 0) memcpy(dst->psi, src->psi, sizeof(double) * (size_t)src->n_psi);
}

/* ========================================================================== */
/* A8  loaders                                                                 */
/* ========================================================================== */
int or_modes_read(const char *path, int *ndof, int *nmodes, double **omega2,
                  double **modes) {                    /* ModeData.h:61-83 */
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    int32_t nd = 0, nm = 0;
    if (fread(&nd, sizeof(int32_t), 1, f) != 1 || fread(&nm, sizeof(int32_t), 1, f) != 1) {
        fclose(f); return -2;
    }
    if (nd < 0 || nm < 0) { fclose(f); return -3; }
    double *os = (double *)malloc(sizeof(double) * (size_t)(nm > 0 ? nm : 1));
    double *md = (double *)malloc(sizeof(double) * (size_t)(nm > 0 ? nm : 1) * (size_t)(nd > 0 ? nd : 1));
    size_t got = fread(os, sizeof(double), (size_t)nm, f);
    got += fread(md, sizeof(double), (size_t)nm * (size_t)nd, f);
    fclose(f);
    if (got != (size_t)nm + (size_t)nm * (size_t)nd) { free(os); free(md); return -4; }
    *ndof = nd; *nmodes = nm; *omega2 = os; *modes = md;
    return 0;
}

int or_modes_write(const char *path, int ndof, int nmodes, const double *omega2,
                   const double *modes) {              /* ModeData.h:87-107 */
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int32_t nd = ndof, nm = nmodes;
    fwrite(&nd, sizeof(int32_t), 1, f);
    fwrite(&nm, sizeof(int32_t), 1, f);
    fwrite(omega2, sizeof(double), (size_t)nmodes, f);
    fwrite(modes, sizeof(double), (size_t)nmodes * (size_t)ndof, f);
    fclose(f);
    return 0;
}

int or_num_modes_audible(const double *omega2, int nmodes, double density,
                         double audible_freq) {        /* ModeData.h:120-148 */
#define OR_FREQ(os) (sqrt((os) / density) / (2. * M_PI))
    if (nmodes == 0 || OR_FREQ(omega2[0]) > audible_freq) return 0;      /* :132 */
    if (OR_FREQ(omega2[nmodes - 1]) <= audible_freq) return nmodes;      /* :135 */
    int ii;
    for (ii = 0; ii < nmodes; ++ii)
        if (OR_FREQ(omega2[ii]) > audible_freq) break;                   /* :139-143 */
#undef OR_FREQ
    return ii;
}

int or_material_read(const char *path, double out[5]) { /* ModalMaterial.h:35-55 */
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char *line = NULL;
    size_t cap = 0;
    ssize_t len;
    char *keep = NULL;
    while ((len = getline(&line, &cap, f)) >= 0) {
        if (line[0] != '#') { keep = line; break; }   /* :45-49 */
    }
    for (int i = 0; i < 5; ++i) out[i] = 0.0;
    if (keep) {
        /* iss >> density >> youngsModulus >> poissonRatio >> alpha >> beta */
        sscanf(keep, "%lf %lf %lf %lf %lf", &out[0], &out[1], &out[2], &out[3], &out[4]);
    }
    free(line);
    fclose(f);
    return 0;
}

/* ---- proto3 wire decoding for ffat_map.proto:8-51 ------------------------- */
typedef struct { const unsigned char *p, *end; int err; } pb_rd;

static uint64_t pb_varint(pb_rd *r) {
    uint64_t v = 0;
    int shift = 0;
    while (r->p < r->end && shift < 64) {
        unsigned char c = *r->p++;
        v |= (uint64_t)(c & 0x7f) << shift;
        if (!(c & 0x80)) return v;
        shift += 7;
    }
    r->err = 1;
    return 0;
}
static double pb_f64(pb_rd *r) {
    double v = 0;
    if (r->end - r->p < 8) { r->err = 1; return 0; }
    memcpy(&v, r->p, 8);
    r->p += 8;
    return v;
}
static pb_rd pb_sub(pb_rd *r) {
    pb_rd s = {r->p, r->p, 0};
    uint64_t n = pb_varint(r);
    if (r->err || (uint64_t)(r->end - r->p) < n) { r->err = 1; return s; }
    s.p = r->p; s.end = r->p + n;
    r->p += n;
    return s;
}
static void pb_skip(pb_rd *r, int wt) {
    switch (wt) {
    case 0: (void)pb_varint(r); break;
    case 1: if (r->end - r->p < 8) r->err = 1; else r->p += 8; break;
    case 2: (void)pb_sub(r); break;
    case 5: if (r->end - r->p < 4) r->err = 1; else r->p += 4; break;
    default: r->err = 1;
    }
}

typedef struct { double *v; int n, cap; } dvec;
typedef struct { int *v; int n, cap; } ivec;
static void dpush(dvec *d, double x) {
    if (d->n == d->cap) { d->cap = d->cap ? d->cap * 2 : 16; d->v = (double *)realloc(d->v, sizeof(double) * (size_t)d->cap); }
    d->v[d->n++] = x;
}
static void ipush(ivec *d, int x) {
    if (d->n == d->cap) { d->cap = d->cap ? d->cap * 2 : 16; d->v = (int *)realloc(d->v, sizeof(int) * (size_t)d->cap); }
    d->v[d->n++] = x;
}
/* message vec { repeated double item = 1; } (packed or not) */
static void pb_vec(pb_rd s, dvec *out) {
    while (s.p < s.end && !s.err) {
        uint64_t key = pb_varint(&s);
        int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) { pb_rd q = pb_sub(&s); while (q.p < q.end && !q.err) dpush(out, pb_f64(&q)); }
        else if (fn == 1 && wt == 1) dpush(out, pb_f64(&s));
        else pb_skip(&s, wt);
    }
}
/* message vec_i { repeated int32 item = 1; } */
static void pb_vec_i(pb_rd s, ivec *out) {
    while (s.p < s.end && !s.err) {
        uint64_t key = pb_varint(&s);
        int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) { pb_rd q = pb_sub(&s); while (q.p < q.end && !q.err) ipush(out, (int)(int64_t)pb_varint(&q)); }
        else if (fn == 1 && wt == 0) ipush(out, (int)(int64_t)pb_varint(&s));
        else pb_skip(&s, wt);
    }
}
/* message mat { repeated vec item = 1; }: rows appended, row lengths recorded */
static void pb_mat(pb_rd s, dvec *out, ivec *rowlen) {
    while (s.p < s.end && !s.err) {
        uint64_t key = pb_varint(&s);
        int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) { int n0 = out->n; pb_vec(pb_sub(&s), out); ipush(rowlen, out->n - n0); }
        else pb_skip(&s, wt);
    }
}
static void pb_mat_i(pb_rd s, ivec *out, ivec *rowlen) {
    while (s.p < s.end && !s.err) {
        uint64_t key = pb_varint(&s);
        int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) { int n0 = out->n; pb_vec_i(pb_sub(&s), out); ipush(rowlen, out->n - n0); }
        else pb_skip(&s, wt);
    }
}
static int copy3(const dvec *d, double out[3]) {
    if (d->n != 3) return -1;          /* DESERIALIZE_VEC(..., false) asserts size */
    out[0] = d->v[0]; out[1] = d->v[1]; out[2] = d->v[2];
    return 0;
}

/* ffat_map_t_1, ffat_map_serialize.h:176-222 */
static int pb_shell(pb_rd s, or_ffat_map *m) {
    int rc = 0;
    while (s.p < s.end && !s.err) {
        uint64_t key = pb_varint(&s);
        int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 1) m->cell_size = pb_f64(&s);
        else if (fn == 2 && wt == 2) {            /* lowcorners: N_i x 3 */
            dvec d = {0}; ivec rl = {0};
            pb_mat(pb_sub(&s), &d, &rl);
            if (rl.n != 6) rc = -10;
            for (int i = 0; i < rl.n && i < 6 && !rc; ++i) {
                if (rl.v[i] < 3) { rc = -11; break; }
                int off = 0; for (int j = 0; j < i; ++j) off += rl.v[j];
                for (int j = 0; j < 3; ++j) m->low_corners[i][j] = d.v[off + j];
            }
            free(d.v); free(rl.v);
        } else if (fn == 3 && wt == 2) {          /* n_elements: N_i x 2 */
            ivec d = {0}, rl = {0};
            pb_mat_i(pb_sub(&s), &d, &rl);
            if (rl.n != 6) rc = -12;
            for (int i = 0; i < rl.n && i < 6 && !rc; ++i) {
                if (rl.v[i] < 2) { rc = -13; break; }
                int off = 0; for (int j = 0; j < i; ++j) off += rl.v[j];
                m->n_elements[i][0] = d.v[off];
                m->n_elements[i][1] = d.v[off + 1];
            }
            free(d.v); free(rl.v);
        } else if (fn == 4 && wt == 2) {          /* strides (resized to item_size) */
            ivec d = {0};
            pb_vec_i(pb_sub(&s), &d);
            if (d.n != 6) rc = -14;
            for (int i = 0; i < d.n && i < 6; ++i) m->strides[i] = d.v[i];
            free(d.v);
        } else if ((fn == 5 || fn == 6 || fn == 7) && wt == 2) {
            dvec d = {0};
            pb_vec(pb_sub(&s), &d);
            double *dst = fn == 5 ? m->center : fn == 6 ? m->bbox_low : m->bbox_top;
            if (copy3(&d, dst)) rc = -15;
            free(d.v);
        } else pb_skip(&s, wt);
    }
    return s.err ? -1 : rc;
}

/* ffat_map_t_3, ffat_map_serialize.h:223-253 */
static int pb_map3(pb_rd s, or_ffat_map *m) {
    int rc = 0;
    while (s.p < s.end && !s.err) {
        uint64_t key = pb_varint(&s);
        int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 1) m->k = pb_f64(&s);
        else if (fn == 2 && wt == 2) {
            dvec d = {0};
            pb_vec(pb_sub(&s), &d);
            if (copy3(&d, m->center3)) rc = -20;
            free(d.v);
        } else if (fn == 3 && wt == 2) { int r2 = pb_shell(pb_sub(&s), m); if (r2) rc = r2; }
        else if (fn == 4 && wt == 0) m->is_compressed = pb_varint(&s) != 0;
        else if (fn == 5 && wt == 2) {            /* psi: one vec per column; column 0 is used */
            dvec d = {0}; ivec rl = {0};
            pb_mat(pb_sub(&s), &d, &rl);
            int rows = rl.n > 0 ? rl.v[0] : 0;
            free(m->psi);
            m->n_psi = rows;
            m->psi = (double *)malloc(sizeof(double) * (size_t)(rows > 0 ? rows : 1));
            if (rows > 0) memcpy(m->psi, d.v, sizeof(double) * (size_t)rows);
            free(d.v); free(rl.v);
        } else if (fn == 6 && wt == 0) m->mode_id = (int)(int64_t)pb_varint(&s);
        else pb_skip(&s, wt);
    }
    return s.err ? -1 : rc;
}

int or_fatcube_parse(const unsigned char *bytes, size_t n, or_ffat_map *out) {
    memset(out, 0, sizeof(*out));   /* proto3 defaults: modeid 0, is_compressed false, k 0 */
    pb_rd r = {bytes, bytes + n, 0};
    int rc = 0;
    while (r.p < r.end && !r.err) {
        uint64_t key = pb_varint(&r);
        int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) { int r2 = pb_map3(pb_sub(&r), out); if (r2) rc = r2; }
        else pb_skip(&r, wt);
    }
    if (r.err) rc = -1;
    if (rc) {                        /* a rejected file leaves nothing behind (found by `make -C oracle asan`) */
        free(out->psi);
        out->psi = NULL;
        out->n_psi = 0;
    }
    return rc;
}

int or_fatcube_load(const char *path, or_ffat_map *out) {
    FILE *f = fopen(path, "rb");
    if (!f) return -100;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *buf = (unsigned char *)malloc((size_t)(n > 0 ? n : 1));
    size_t got = fread(buf, 1, (size_t)n, f);
    fclose(f);
    int rc = got == (size_t)n ? or_fatcube_parse(buf, (size_t)n, out) : -101;
    free(buf);
    return rc;
}

/* ========================================================================== */
/* A3 + A4  ModalSolver<double>::step, modal_solver.h:181-276                  */
/* ========================================================================== */
#define FORCE_Q_CAP 1023   /* ReaderWriterQueue(512): ceilToPow2(513)-1, readerwriterqueue.h:101 */
#define TRANS_Q_CAP 1      /* ReaderWriterQueue(1) */
#define ARPRM_Q_CAP 1

typedef struct arprm { double a[2], sigma, mu; } arprm;

struct or_solver {
    int n_modes;
    or_integrator *it;
    /* force queue (FIFO ring) */
    or_force_msg *fq;
    int fq_head, fq_count;
    /* active forces: std::list<ForceMessage> */
    or_force_msg *active;
    int n_active, cap_active;
    /* trans queue (cap 1), arprm queue (cap 1) */
    double *trans_q; int trans_q_full;
    arprm arprm_q;   int arprm_q_full;
    double *latest_transfer;    /* _latest_transfer.data, unit 1e7 at ctor :89-97,137 */
    double *mess_trans;         /* _mess_trans.data */
    int use_transfer;           /* _useTransfer (:139) */
    int use_transfer_cache;     /* _useTransferCache (:140) */
    int sustained;              /* _sustainedForces (:126) */
    or_ffat_map *maps; int n_maps; int have_maps;
    double *space;              /* _forceSpreadBufferSpace */
    double time_buf[B];         /* _forceSpreadBufferTime */
    double *Q;                  /* temporary S*T(ii), modal_solver.h:266 */
};

static void msg_free(or_force_msg *m) { free(m->data); m->data = NULL; }
static void msg_copy(or_force_msg *dst, const or_force_msg *src) {
    *dst = *src;                               /* includes the Force state (deep copy) */
    dst->data = (double *)malloc(sizeof(double) * (size_t)(src->n > 0 ? src->n : 1));
    if (src->n > 0) memcpy(dst->data, src->data, sizeof(double) * (size_t)src->n);
}

or_solver *or_solver_new(int n_modes) {        /* modal_solver.h:128-141 */
    or_solver *s = (or_solver *)calloc(1, sizeof(*s));
    size_t nb = (size_t)(n_modes > 0 ? n_modes : 1);
    s->n_modes = n_modes;
    s->fq = (or_force_msg *)calloc(FORCE_Q_CAP, sizeof(or_force_msg));
    s->trans_q = (double *)calloc(nb, sizeof(double));
    s->latest_transfer = (double *)malloc(nb * sizeof(double));
    s->mess_trans = (double *)malloc(nb * sizeof(double));
    for (int i = 0; i < n_modes; ++i) { s->latest_transfer[i] = 1.0; s->latest_transfer[i] *= 1E7; }
    for (int i = 0; i < n_modes; ++i) { s->mess_trans[i] = 1.0; s->mess_trans[i] *= 1E7; }
    s->use_transfer = 1;
    s->use_transfer_cache = 1;
    s->space = (double *)calloc(nb, sizeof(double));
    s->Q = (double *)calloc(nb, sizeof(double));
    return s;
}

void or_solver_free(or_solver *s) {
    if (!s) return;
    for (int i = 0; i < s->fq_count; ++i) msg_free(&s->fq[(s->fq_head + i) % FORCE_Q_CAP]);
    free(s->fq);
    for (int i = 0; i < s->n_active; ++i) msg_free(&s->active[i]);
    free(s->active);
    free(s->trans_q); free(s->latest_transfer); free(s->mess_trans);
    for (int i = 0; i < s->n_maps; ++i) or_ffat_free(&s->maps[i]);
    free(s->maps);
    free(s->space); free(s->Q);
    or_integrator_free(s->it);
    free(s);
}

void or_solver_set_integrator(or_solver *s, or_integrator *it) { s->it = it; }

void or_solver_set_ffat_maps(or_solver *s, const or_ffat_map *maps, int n) {
    for (int i = 0; i < s->n_maps; ++i) or_ffat_free(&s->maps[i]);
    free(s->maps);
    s->maps = (or_ffat_map *)calloc((size_t)(n > 0 ? n : 1), sizeof(or_ffat_map));
    s->n_maps = 0;
    /* std::map<int,FFAT_Map>: a later file with the same modeId replaces the earlier */
    for (int i = 0; i < n; ++i) {
        int slot = -1;
        for (int j = 0; j < s->n_maps; ++j) if (s->maps[j].mode_id == maps[i].mode_id) slot = j;
        if (slot < 0) slot = s->n_maps++;
        else or_ffat_free(&s->maps[slot]);
        ffat_copy(&s->maps[slot], &maps[i]);
    }
    s->have_maps = 1;                      /* LoadAll always returns a non-null map, Q12 */
}

int or_solver_enqueue_force(or_solver *s, const double *data, int n,
                            const or_force *force, int sustained_start,
                            int sustained_end, int clear_all) {
    if (s->fq_count >= FORCE_Q_CAP) return 0;
    or_force_msg m;
    memset(&m, 0, sizeof(m));
    m.n = n;
    m.data = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (n > 0) memcpy(m.data, data, sizeof(double) * (size_t)n);
    if (force) { m.force = *force; m.force_type = force->type; }
    else { or_force_init_point(&m.force); m.force_type = OR_POINT_FORCE; }  /* ForceMessage() :35-37 */
    m.sustained_start = sustained_start;
    m.sustained_end = sustained_end;
    m.clear_all = clear_all;
    s->fq[(s->fq_head + s->fq_count) % FORCE_Q_CAP] = m;
    s->fq_count++;
    return 1;
}

int or_solver_enqueue_arprm(or_solver *s, const double a[2], double sigma, double mu) {
    if (s->arprm_q_full) return 0;
    s->arprm_q.a[0] = a[0]; s->arprm_q.a[1] = a[1];
    s->arprm_q.sigma = sigma; s->arprm_q.mu = mu;
    s->arprm_q_full = 1;
    return 1;
}

static const or_ffat_map *map_at(const or_solver *s, int id) {
    for (int i = 0; i < s->n_maps; ++i) if (s->maps[i].mode_id == id) return &s->maps[i];
    return NULL;
}

int or_solver_compute_transfer(or_solver *s, const double pos[3]) { /* :286-300 */
    if (!s->have_maps) return 0;
    for (int ii = 0; ii < s->n_modes; ++ii) {
        const or_ffat_map *m = map_at(s, ii);
        if (!m) return -1;                         /* std::map::at throws */
        s->mess_trans[ii] = fabs(or_ffat_get_map_val(m, pos));
    }
    if (s->trans_q_full) return 0;                 /* try_enqueue on the 1-slot queue */
    memcpy(s->trans_q, s->mess_trans, sizeof(double) * (size_t)s->n_modes);
    s->trans_q_full = 1;
    return 1;
}

int or_solver_compute_transfer_out(or_solver *s, const double pos[3], double *out) { /* :302-315 */
    if (!s->have_maps) return 0;
    const int N = s->n_maps;
    for (int ii = 0; ii < N; ++ii) {
        const or_ffat_map *m = map_at(s, ii);
        if (!m) return -1;
        out[ii] = fabs(or_ffat_get_map_val(m, pos));
    }
    return 1;
}

void or_solver_set_use_transfer(or_solver *s, int use) { s->use_transfer = use; }
const double *or_solver_latest_transfer(const or_solver *s) { return s->latest_transfer; }
int or_solver_n_active(const or_solver *s) { return s->n_active; }
const double *or_solver_state(const or_solver *s, int which) {
    return which == 0 ? s->it->q[s->it->cur % 3] : s->it->q[(s->it->cur + 2) % 3];
}

static void active_clear(or_solver *s) {
    for (int i = 0; i < s->n_active; ++i) msg_free(&s->active[i]);
    s->n_active = 0;
}
static void active_push(or_solver *s, const or_force_msg *m) {
    if (s->n_active == s->cap_active) {
        s->cap_active = s->cap_active ? s->cap_active * 2 : 8;
        s->active = (or_force_msg *)realloc(s->active, sizeof(or_force_msg) * (size_t)s->cap_active);
    }
    msg_copy(&s->active[s->n_active++], m);
}

int or_solver_step(or_solver *s, double *sound, double *qnorm) {
    const int N = s->n_modes;
    /* :184 dequeue at most one force message */
    if (s->fq_count > 0) {
        or_force_msg mess = s->fq[s->fq_head];
        s->fq_head = (s->fq_head + 1) % FORCE_Q_CAP;
        s->fq_count--;
        if (mess.clear_all) {                      /* :186-189 */
            active_clear(s);
            msg_free(&mess);
            return 0;
        }
        if (mess.sustained_start) {                /* :190-194 */
            active_clear(s);
            s->sustained = 1;
            active_push(s, &mess);
        }
        if (!s->sustained) {                       /* :195-196 */
            active_push(s, &mess);
        } else {                                   /* :197-200 copy data only */
            if (s->n_active < 1) { msg_free(&mess); return -1; }   /* begin() of an empty list: UB in the reference */
            or_force_msg *front = &s->active[0];
            if (front->n != mess.n) {
                front->data = (double *)realloc(front->data, sizeof(double) * (size_t)(mess.n > 0 ? mess.n : 1));
                front->n = mess.n;
            }
            if (mess.n > 0) memcpy(front->data, mess.data, sizeof(double) * (size_t)mess.n);
        }
        if (mess.sustained_end) {                  /* :201-204 */
            active_clear(s);
            s->sustained = 0;
        }
        msg_free(&mess);
    }
    for (int i = 0; i < B; ++i) s->time_buf[i] = 0.0;      /* :206 */
    if (!s->sustained) {                                   /* :207-221 */
        for (int i = 0; i < N; ++i) s->space[i] = 0.0;
        int w = 0;
        for (int r = 0; r < s->n_active; ++r) {
            or_force_msg *it = &s->active[r];
            int added = or_force_add(&it->force, s->time_buf);
            if (!added) {
                msg_free(it);                              /* erase */
            } else {
                for (int i = 0; i < N; ++i) s->space[i] += it->data[i];
                if (w != r) s->active[w] = *it;
                ++w;
            }
        }
        s->n_active = w;
    } else {                                               /* :222-240 */
        if (s->n_active != 1) return -1;                   /* assert :223-224 (live: no NDEBUG) */
        or_force_msg *it = &s->active[0];
        if (it->force_type == OR_AR_FORCE) {
            if (s->arprm_q_full) {
                s->arprm_q_full = 0;
                or_force_ar_set_param(&it->force, s->arprm_q.a, s->arprm_q.sigma, s->arprm_q.mu);
            }
        }
        or_force_add(&it->force, s->time_buf);
        for (int i = 0; i < N; ++i) s->space[i] = it->data[i];
    }
    /* :242-256 transfer selection (single-threaded: try_lock always succeeds) */
    int use_transfer = s->use_transfer;
    if (use_transfer) {
        if (s->trans_q_full) {
            memcpy(s->latest_transfer, s->trans_q, sizeof(double) * (size_t)N);
            s->trans_q_full = 0;
        }
    } else {
        for (int i = 0; i < N; ++i) { s->latest_transfer[i] = 1.0; s->latest_transfer[i] *= 1E7; }
    }
    s->use_transfer_cache = use_transfer;

    /* :262-272 hot loop */
    if (qnorm) for (int i = 0; i < N; ++i) qnorm[i] = 0.0;
    const double *tr = s->latest_transfer;
    for (int ii = 0; ii < B; ++ii) {
        const double t = s->time_buf[ii];
        for (int i = 0; i < N; ++i) s->Q[i] = s->space[i] * t;   /* temporary, :266 */
        const double *q = or_integrator_step(s->it, s->Q);
        double dot = 0.0;
        for (int i = 0; i < N; ++i) dot += q[i] * tr[i];         /* :267-269 */
        if (sound) sound[ii] = dot;
        if (qnorm) for (int i = 0; i < N; ++i) qnorm[i] += q[i] * q[i];   /* :270 */
    }
    if (qnorm) for (int i = 0; i < N; ++i) qnorm[i] = sqrt(qnorm[i]);     /* :272 */
    return 1;
}

/* A9: tools/real_time_modal_sound.cpp:207-210 */
void or_pa_callback_convert(const double *sound, int frames, float *out) {
    for (int i = 0; i < frames; ++i) {
        *out++ = (float)(sound[i] / 1E10);
        *out++ = (float)(sound[i] / 1E10);
    }
}

/* ========================================================================== */
/* cpu_baseline leg                                                            */
/* ========================================================================== */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double or_bench_run(int n_obj, int n_modes, int n_buffers, int n_threads,
                    const double *omega2, double density, double alpha, double beta,
                    const double *hit_data, const unsigned char *hit_mask,
                    double *sound_out, int flush_denormals) {
    /* Objects are dealt to the threads in contiguous static shares; every thread BUILDS its own
     * solvers (first touch on its own NUMA node) and then steps them.  Only the stepping is timed:
     * from the moment every thread has built its share until the last thread is done.           */
    or_solver **sv = (or_solver **)calloc((size_t)n_obj, sizeof(*sv));
    double t0 = 0, t1 = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
#if defined(__x86_64__)
        if (flush_denormals) {
            _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
            _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
        }
#endif
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int o = 0; o < n_obj; ++o) {
            sv[o] = or_solver_new(n_modes);
            or_solver_set_integrator(sv[o], or_integrator_build(
                density, omega2 + (size_t)o * n_modes, n_modes, alpha, beta,
                1. / (double)OR_SAMPLE_RATE, n_modes));
            or_solver_set_use_transfer(sv[o], 0);
        }   /* implicit barrier: everything is built */
#ifdef _OPENMP
#pragma omp single
#endif
        t0 = now_s();   /* implicit barrier after single */
        double sound[B];
        double *qn = (double *)malloc(sizeof(double) * (size_t)(n_modes > 0 ? n_modes : 1));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int o = 0; o < n_obj; ++o) {
            for (int b = 0; b < n_buffers; ++b) {
                if (hit_mask[(size_t)o * n_buffers + b])
                    or_solver_enqueue_force(sv[o], hit_data + (size_t)o * n_modes, n_modes, NULL, 0, 0, 0);
                or_solver_step(sv[o], sound, qn);        /* qnorm is part of step(), :262-273 */
                if (sound_out)
                    memcpy(sound_out + ((size_t)o * n_buffers + b) * B, sound, sizeof(sound));
            }
        }   /* implicit barrier: the slowest thread is done */
#ifdef _OPENMP
#pragma omp single
#endif
        t1 = now_s();
        free(qn);
#if defined(__x86_64__)
        if (flush_denormals) {
            _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_OFF);
            _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_OFF);
        }
#endif
    }
    for (int o = 0; o < n_obj; ++o) or_solver_free(sv[o]);
    free(sv);
    return t1 - t0;
}
