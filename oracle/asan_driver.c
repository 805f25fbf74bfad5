/* TEST INFRASTRUCTURE ONLY: exercises oracle/pbso_oracle.c under AddressSanitizer + UBSan (CPU build only).
 * Usage: oracle_asan_check <file>...   -- golden fixtures; every file is also fed in truncated and
 * byte-flipped form to the matching parser.  A sanitizer report aborts with a non-zero status.        */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pbso_oracle.h"

static unsigned char *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *b = (unsigned char *)malloc(len > 0 ? (size_t)len : 1);
    *n = fread(b, 1, (size_t)len, f);
    fclose(f);
    return b;
}

static int ends_with(const char *s, const char *suf) {
    size_t a = strlen(s), b = strlen(suf);
    return a >= b && strcmp(s + a - b, suf) == 0;
}

static void solver_run(void) {
    /* a small ModalSolver life: point + Gaussian + sustained AR forces, AR parameter update, FFAT transfer,
     * clearAllForces, queue overflow */
    enum { M = 37, NB = 12 };
    double om[M], data[M], sound[OR_FRAMES_PER_BUFFER], qn[M], psi[6 * 4 * 4], pos[3] = {0.3, -0.2, 0.45};
    for (int i = 0; i < M; ++i) {
        const double f = 150.0 + 400.0 * i;
        om[i] = 2500.0 * (2 * 3.141592653589793 * f) * (2 * 3.141592653589793 * f);
        data[i] = 1e-3 * (i % 5 - 2);
    }
    or_solver *s = or_solver_new(M);
    or_solver_set_integrator(s, or_integrator_build(2500.0, om, M, 6.0, 1e-7, 1.0 / OR_SAMPLE_RATE, M));
    or_ffat_map maps[M];
    const double c[3] = {0, 0, 0};
    for (int i = 0; i < M; ++i) {
        for (int j = 0; j < 96; ++j) psi[j] = 1e6 * (1 + (i * 7 + j) % 13);
        or_ffat_make_uniform_cube(&maps[i], i, 0.1 * (i + 1), c, 0.02, 4, psi);
    }
    or_solver_set_ffat_maps(s, maps, M);
    or_force g, ar;
    or_force_init_gaussian(&g, 900.0);
    or_force_init_ar(&ar);
    const double a[2] = {0.6, 0.2};
    for (int b = 0; b < NB; ++b) {
        pos[0] += 0.01;
        or_solver_compute_transfer(s, pos);
        if (b == 0) or_solver_enqueue_force(s, data, M, NULL, 0, 0, 0);
        if (b == 1) or_solver_enqueue_force(s, data, M, &g, 0, 0, 0);
        if (b == 4) or_solver_enqueue_force(s, data, M, &ar, 1, 0, 0);
        if (b == 6) or_solver_enqueue_arprm(s, a, 0.002, 0.1);
        if (b == 8) or_solver_enqueue_force(s, data, M, &ar, 0, 1, 0);
        if (b == 10) or_solver_enqueue_force(s, NULL, 0, NULL, 0, 0, 1);
        or_solver_step(s, sound, qn);
    }
    int taken = 0;
    for (int i = 0; i < 1100; ++i) taken += or_solver_enqueue_force(s, data, M, NULL, 0, 0, 0);
    if (taken != 1023) { fprintf(stderr, "queue capacity %d\n", taken); exit(2); }
    float stereo[2 * OR_FRAMES_PER_BUFFER];
    or_pa_callback_convert(sound, OR_FRAMES_PER_BUFFER, stereo);
    or_solver_free(s);
    for (int i = 0; i < M; ++i) or_ffat_free(&maps[i]);
}

int main(int argc, char **argv) {
    solver_run();
    int n_files = 0;
    for (int i = 1; i < argc; ++i) {
        size_t n = 0;
        unsigned char *b = slurp(argv[i], &n);
        if (!b) { fprintf(stderr, "cannot read %s\n", argv[i]); return 3; }
        ++n_files;
        if (ends_with(argv[i], ".fatcube")) {
            or_ffat_map m;
            memset(&m, 0, sizeof(m));
            if (or_fatcube_parse(b, n, &m) != 0) printf("rejected: %s\n", argv[i]);
            else {
                /* (an empty file is a valid, all-default message with no Psi: the reference would index an
                 *  empty matrix there, so only maps that carry data are evaluated) */
                int ok = m.n_psi > 0;
                for (int f = 0; f < 6; ++f)
                    ok = ok && m.n_elements[f][0] > 0 && m.n_elements[f][1] > 0 && m.strides[f] >= 0 &&
                         (long long)m.strides[f] + (long long)m.n_elements[f][0] * m.n_elements[f][1] <= m.n_psi;
                if (ok) {
                    const double p[3] = {0.4, 0.1, -0.3};
                    volatile double v = or_ffat_get_map_val(&m, p);
                    (void)v;
                }
                or_ffat_free(&m);
            }
            for (size_t cut = 0; cut < n; ++cut) {                      /* every truncation */
                memset(&m, 0, sizeof(m));
                unsigned char *t = (unsigned char *)malloc(cut ? cut : 1);     /* exact-size copy: overreads trip ASan */
                memcpy(t, b, cut);
                if (or_fatcube_parse(t, cut, &m) == 0) or_ffat_free(&m);
                free(t);
            }
            for (size_t k = 0; k < n; ++k) {                            /* every byte, three corruptions */
                static const unsigned char x[3] = {0xFF, 0x80, 0x01};
                for (int j = 0; j < 3; ++j) {
                    unsigned char *t = (unsigned char *)malloc(n);
                    memcpy(t, b, n);
                    t[k] ^= x[j];
                    memset(&m, 0, sizeof(m));
                    if (or_fatcube_parse(t, n, &m) == 0) or_ffat_free(&m);
                    free(t);
                }
            }
        } else if (ends_with(argv[i], ".modes")) {
            int nd = 0, nm = 0;
            double *o = NULL, *md = NULL;
            if (or_modes_read(argv[i], &nd, &nm, &o, &md) != 0) printf("rejected: %s\n", argv[i]);
            else {
                volatile int na = or_num_modes_audible(o, nm, 2500.0, 20000.0);
                (void)na;
                free(o);
                free(md);
            }
            char tmp[] = "/tmp/pbso_asan_XXXXXX";
            int fd = mkstemp(tmp);
            if (fd >= 0) {
                FILE *f = fdopen(fd, "wb");
                fwrite(b, 1, n / 2, f);                                 /* truncated file */
                fclose(f);
                if (or_modes_read(tmp, &nd, &nm, &o, &md) == 0) { free(o); free(md); }
                remove(tmp);
            }
        } else if (ends_with(argv[i], ".txt")) {
            double mat[5];
            (void)or_material_read(argv[i], mat);
        }
        free(b);
    }
    printf("oracle asan/ubsan check ok (%d files)\n", n_files);
    return 0;
}
