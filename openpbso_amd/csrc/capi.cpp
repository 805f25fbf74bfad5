// extern "C" surface declared in include/openpbso_amd.h.  No exception leaves
// this file.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>

#include "engine.h"

using pbso::Engine;

struct pbso_engine {
    Engine *impl;
    std::string err;
};

#define GUARD_BEGIN try {
#define GUARD_END(e)                                            \
    } catch (const std::bad_alloc &) {                          \
        if (e) (e)->err = "out of host memory";                 \
        return PBSO_ERR_NOMEM;                                  \
    } catch (const std::exception &ex) {                        \
        if (e) (e)->err = ex.what();                            \
        return PBSO_ERR_INVALID;                                \
    } catch (...) {                                             \
        if (e) (e)->err = "unknown exception";                  \
        return PBSO_ERR_INVALID;                                \
    }

extern "C" {

int pbso_abi_version(void) { return PBSO_ABI_VERSION; }

const char *pbso_status_string(int s) {
    switch (s) {
    case PBSO_OK: return "ok";
    case PBSO_ERR_INVALID: return "invalid argument";
    case PBSO_ERR_HIP: return "HIP runtime error";
    case PBSO_ERR_STATE: return "invalid engine state";
    case PBSO_ERR_IO: return "I/O error";
    case PBSO_ERR_MISSING_MAP: return "FFAT map missing";
    case PBSO_ERR_ASSERT: return "reference assertion would fail";
    case PBSO_ERR_NOMEM: return "out of memory";
    }
    return "unknown status";
}

int pbso_engine_create(const pbso_engine_desc *desc, pbso_engine **out) {
    if (!desc || !out) return PBSO_ERR_INVALID;
    *out = nullptr;
    pbso_engine *e = nullptr;
    GUARD_BEGIN
    e = new pbso_engine{nullptr, {}};
    e->impl = new Engine(*desc);
    int rc = e->impl->init();
    if (rc != PBSO_OK) {
        // keep the object alive so that the caller can read the error text
        e->err = e->impl->last_error();
        delete e->impl;
        e->impl = nullptr;
        *out = e;
        return rc;
    }
    *out = e;
    return PBSO_OK;
    GUARD_END(e)
}

void pbso_engine_destroy(pbso_engine *e) {
    if (!e) return;
    try {
        delete e->impl;
    } catch (...) {
    }
    delete e;
}

const char *pbso_last_error(const pbso_engine *e) {
    if (!e) return "null engine";
    if (e->impl && e->impl->last_error()[0]) return e->impl->last_error();
    return e->err.c_str();
}

#define NEED(e)                                       \
    if (!(e) || !(e)->impl) return PBSO_ERR_STATE

int pbso_add_object(pbso_engine *e, const pbso_object_desc *d, int *id) {
    NEED(e);
    if (!d) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    return e->impl->add_object(*d, id);
    GUARD_END(e)
}

int pbso_fatcube_parse(const unsigned char *bytes, size_t n, pbso_ffat_map *out) {
    if (!bytes || !out) return PBSO_ERR_INVALID;
    try {
        return pbso::parse_fatcube(bytes, n, out);
    } catch (...) {
        return PBSO_ERR_NOMEM;
    }
}

void pbso_ffat_map_free(pbso_ffat_map *m) {
    if (m && m->psi) {
        std::free((void *)m->psi);
        m->psi = nullptr;
        m->n_psi = 0;
    }
}

int pbso_object_set_ffat_maps(pbso_engine *e, int obj, const pbso_ffat_map *maps, int n) {
    NEED(e);
    GUARD_BEGIN
    return e->impl->set_ffat_maps(obj, maps, n);
    GUARD_END(e)
}

// FFAT_Map_Serialize_Double::LoadAll, ffat_map_serialize.h:267-279
int pbso_object_read_ffat_maps(pbso_engine *e, int obj, const char *dir) {
    NEED(e);
    if (!dir) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    std::vector<std::string> names;
    // a directory that cannot be opened yields an empty, non-null map (io.cpp:31-34)
    (void)pbso::list_dir_files(dir, ".fatcube", names);
    std::vector<pbso_ffat_map> maps(names.size());
    int rc = PBSO_OK;
    size_t loaded = 0;
    for (; loaded < names.size(); ++loaded) {
        std::vector<unsigned char> bytes;
        rc = pbso::read_file_bytes(names[loaded].c_str(), bytes);
        if (rc == PBSO_OK) rc = pbso::parse_fatcube(bytes.data(), bytes.size(), &maps[loaded]);
        if (rc != PBSO_OK) {
            e->err = "cannot parse " + names[loaded];
            break;
        }
    }
    if (rc == PBSO_OK) rc = e->impl->set_ffat_maps(obj, maps.data(), (int)maps.size());
    for (size_t i = 0; i < loaded && i < maps.size(); ++i) pbso_ffat_map_free(&maps[i]);
    return rc;
    GUARD_END(e)
}

// main() file conventions + BuildSolver, tools/real_time_modal_sound.cpp:480-525, 309-345
int pbso_add_object_from_files(pbso_engine *e, const char *modes_path, const char *material_path,
                               const char *ffat_dir, int *object_id, int *n_modes_audible) {
    NEED(e);
    if (!modes_path || !material_path) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    double mat[5];
    if (pbso::load_material_file(material_path, mat) != PBSO_OK) {
        e->err = std::string("cannot read material file ") + material_path;
        return PBSO_ERR_IO;
    }
    int n_dof = 0, n_modes = 0;
    std::vector<double> omega2, modes;
    if (pbso::load_modes_file(modes_path, &n_dof, &n_modes, omega2, modes) != PBSO_OK) {
        e->err = std::string("cannot open file for reading modes: ") + modes_path;
        return PBSO_ERR_IO;
    }
    double max_freq = 20000.;                                         // tools/...:326-328
    if (ffat_dir) {
        std::ifstream stream((std::string(ffat_dir) + "/freq_threshold.txt").c_str());
        if (stream) {                                                 // tools/...:319-325
            std::string line;
            std::getline(stream, line);
            std::istringstream iss(line);
            iss >> max_freq;
        }
    }
    const int n_aud = pbso::num_modes_audible(omega2, mat[0], max_freq);
    pbso_object_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_modes = n_aud;
    d.n_omega = n_modes;
    d.omega_squared = omega2.data();
    d.density = mat[0];
    d.alpha = mat[3];
    d.beta = mat[4];
    d.n_dof = n_dof;
    d.mode_shapes = modes.data();        // first n_aud modes are the leading rows
    int id = -1;
    int rc = e->impl->add_object(d, &id);
    if (rc != PBSO_OK) return rc;
    if (object_id) *object_id = id;
    if (n_modes_audible) *n_modes_audible = n_aud;
    if (ffat_dir) rc = pbso_object_read_ffat_maps(e, id, ffat_dir);
    return rc;
    GUARD_END(e)
}

int pbso_modes_read(const char *path, int *n_dof, int *n_modes, double **omega_squared, double **modes) {
    if (!path || !n_dof || !n_modes || !omega_squared || !modes) return PBSO_ERR_INVALID;
    try {
        std::vector<double> om, md;
        int rc = pbso::load_modes_file(path, n_dof, n_modes, om, md);
        if (rc != PBSO_OK) return rc;
        double *a = (double *)std::malloc(sizeof(double) * std::max<size_t>(om.size(), 1));
        double *b = (double *)std::malloc(sizeof(double) * std::max<size_t>(md.size(), 1));
        if (!a || !b) { std::free(a); std::free(b); return PBSO_ERR_NOMEM; }
        if (!om.empty()) std::memcpy(a, om.data(), sizeof(double) * om.size());
        if (!md.empty()) std::memcpy(b, md.data(), sizeof(double) * md.size());
        *omega_squared = a;
        *modes = b;
        return PBSO_OK;
    } catch (...) {
        return PBSO_ERR_NOMEM;
    }
}

int pbso_obj_read(const char *path, int *n_vertices, int *n_faces, double **vertices, int **faces, double **vertex_normals) {
    if (!path || !n_vertices || !n_faces || !vertices || !faces || !vertex_normals) return PBSO_ERR_INVALID;
    try {
        std::vector<double> V, VN;
        std::vector<int> F;
        int rc = pbso::load_obj_file(path, V, F, VN);
        if (rc != PBSO_OK) return rc;
        double *a = (double *)std::malloc(sizeof(double) * std::max<size_t>(V.size(), 1));
        int *b = (int *)std::malloc(sizeof(int) * std::max<size_t>(F.size(), 1));
        double *c = (double *)std::malloc(sizeof(double) * std::max<size_t>(VN.size(), 1));
        if (!a || !b || !c) { std::free(a); std::free(b); std::free(c); return PBSO_ERR_NOMEM; }
        if (!V.empty()) std::memcpy(a, V.data(), sizeof(double) * V.size());
        if (!F.empty()) std::memcpy(b, F.data(), sizeof(int) * F.size());
        if (!VN.empty()) std::memcpy(c, VN.data(), sizeof(double) * VN.size());
        *n_vertices = (int)(V.size() / 3);
        *n_faces = (int)(F.size() / 3);
        *vertices = a;
        *faces = b;
        *vertex_normals = c;
        return PBSO_OK;
    } catch (...) {
        return PBSO_ERR_NOMEM;
    }
}

int pbso_num_modes_audible(const double *omega_squared, int n_modes, double density, double audible_freq) {
    if (n_modes < 0 || (n_modes > 0 && !omega_squared)) return PBSO_ERR_INVALID;
    try {
        std::vector<double> om(omega_squared, omega_squared + n_modes);
        return pbso::num_modes_audible(om, density, audible_freq);
    } catch (...) {
        return PBSO_ERR_NOMEM;
    }
}

int pbso_material_read(const char *path, double out[5]) {
    if (!path || !out) return PBSO_ERR_INVALID;
    try {
        return pbso::load_material_file(path, out);
    } catch (...) {
        return PBSO_ERR_IO;
    }
}

void pbso_free(void *p) { std::free(p); }

int pbso_finalize(pbso_engine *e) {
    NEED(e);
    GUARD_BEGIN
    return e->impl->finalize();
    GUARD_END(e)
}

int pbso_enqueue_force(pbso_engine *e, int obj, const pbso_force_msg *m, int64_t nb) {
    NEED(e);
    if (!m) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    return e->impl->enqueue_force(obj, *m, nb);
    GUARD_END(e)
}

int pbso_enqueue_force_batch(pbso_engine *e, int n, const int *objs, const pbso_force_msg *msgs,
                             const int64_t *nb, unsigned char *accepted) {
    NEED(e);
    if (n < 0 || (n && (!objs || !msgs || !nb))) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    return e->impl->enqueue_force_batch(n, objs, msgs, nb, accepted);
    GUARD_END(e)
}

int pbso_enqueue_vertex_hits(pbso_engine *e, int n, const int *objs, const int *vids, const double *vn, const int64_t *nb) {
    NEED(e);
    if (n < 0 || (n && (!objs || !vids || !vn || !nb))) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    return e->impl->enqueue_vertex_hits(n, objs, vids, vn, nb);
    GUARD_END(e)
}

int pbso_enqueue_arprm(pbso_engine *e, int obj, const double a[2], double sigma, double mu, int64_t nb) {
    NEED(e);
    if (!a) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    return e->impl->enqueue_arprm(obj, a, sigma, mu, nb);
    GUARD_END(e)
}

int pbso_arprm_pending(pbso_engine *e, int obj) {
    NEED(e);
    GUARD_BEGIN
    return e->impl->arprm_pending(obj);
    GUARD_END(e)
}

int pbso_compute_transfer(pbso_engine *e, int obj, const double pos[3], int64_t nb) {
    NEED(e);
    if (!pos) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    return e->impl->compute_transfer(obj, pos, nb);
    GUARD_END(e)
}

int pbso_listeners_enable(pbso_engine *e, int obj) {
    NEED(e);
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->listeners_enable(obj);
    GUARD_END(e)
}

int pbso_mix_listeners(pbso_engine *e, int obj, const double *pos, int n_listeners, float *out, size_t n_out) {
    NEED(e);
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->mix_listeners(obj, pos, n_listeners, out, n_out);
    GUARD_END(e)
}

int pbso_object_n_maps(pbso_engine *e, int obj) {
    NEED(e);
    GUARD_BEGIN
    return e->impl->object_n_maps(obj);
    GUARD_END(e)
}

int pbso_compute_transfer_batch(pbso_engine *e, int obj, const double *pos, int n_pos, double *out, int out_cols) {
    NEED(e);
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->compute_transfer_batch(obj, pos, n_pos, out, out_cols);
    GUARD_END(e)
}

int pbso_set_use_transfer(pbso_engine *e, int obj, int use, int64_t nb) {
    NEED(e);
    GUARD_BEGIN
    return e->impl->set_use_transfer(obj, use, nb);
    GUARD_END(e)
}

int pbso_get_latest_transfer(pbso_engine *e, int obj, double *out) {
    NEED(e);
    if (!out) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->get_latest_transfer(obj, out);
    GUARD_END(e)
}

int pbso_step(pbso_engine *e, int n_buffers) {
    NEED(e);
    GUARD_BEGIN
    return e->impl->step(n_buffers, nullptr);
    GUARD_END(e)
}

int pbso_flush(pbso_engine *e) {
    NEED(e);
    GUARD_BEGIN
    return e->impl->drain_submit();
    GUARD_END(e)
}

int pbso_step_into(pbso_engine *e, int n_buffers, void *d_audio) {
    NEED(e);
    if (!d_audio) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    return e->impl->step(n_buffers, d_audio);
    GUARD_END(e)
}

int pbso_read_audio_rows(pbso_engine *e, const int *object_ids, int n_rows, float *host_out) {
    NEED(e);
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->read_audio_rows(object_ids, n_rows, host_out);
    GUARD_END(e)
}

int pbso_compute_transfer_path(pbso_engine *e, int n, const int *object_ids, const double *pos, const int64_t *not_before,
                               unsigned char *accepted) {
    NEED(e);
    GUARD_BEGIN
    return e->impl->compute_transfer_path(n, object_ids, pos, not_before, accepted);
    GUARD_END(e)
}

int pbso_mix_objects(pbso_engine *e, void *d_out) {
    NEED(e);
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->mix_objects(d_out);
    GUARD_END(e)
}

int pbso_step_to_host(pbso_engine *e, int n_buffers, float *host_out, size_t n_floats) {
    NEED(e);
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->step_to_host(n_buffers, host_out, n_floats);
    GUARD_END(e)
}

int pbso_host_wait(pbso_engine *e) {
    NEED(e);
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->host_wait();
    GUARD_END(e)
}

int pbso_host_alloc(size_t bytes, void **out) {
    if (!out) return PBSO_ERR_INVALID;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? PBSO_OK : PBSO_ERR_NOMEM;
}

void pbso_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int pbso_sync(pbso_engine *e) {
    NEED(e);
    GUARD_BEGIN
    return e->impl->sync();
    GUARD_END(e)
}

int pbso_read_audio(pbso_engine *e, float *out, size_t n) {
    NEED(e);
    if (!out) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->read_audio(out, n);
    GUARD_END(e)
}

int pbso_read_emitted(pbso_engine *e, unsigned char *out, size_t n) {
    NEED(e);
    if (!out) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    return e->impl->read_emitted(out, n);
    GUARD_END(e)
}

int pbso_read_qnorm(pbso_engine *e, int obj, int buffer, float *out, int n) {
    NEED(e);
    if (!out) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->read_qnorm(obj, buffer, out, n);
    GUARD_END(e)
}

int pbso_read_state(pbso_engine *e, int obj, double *q1, double *q2, int n) {
    NEED(e);
    if (!q1 || !q2) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->read_state(obj, q1, q2, n);
    GUARD_END(e)
}

int pbso_write_state(pbso_engine *e, int obj, const double *q1, const double *q2, int n) {
    NEED(e);
    if (!q1 || !q2) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->write_state(obj, q1, q2, n);
    GUARD_END(e)
}

void *pbso_audio_device_ptr(pbso_engine *e) {
    if (!e || !e->impl) return nullptr;
    return e->impl->audio_ptr();
}

// PaModalCallback, tools/real_time_modal_sound.cpp:207-210
void pbso_pa_convert(const float *sound, unsigned long frames, float *out) {
    for (unsigned long i = 0; i < frames; ++i) {
        const float v = (float)((double)sound[i] / 1E10);
        *out++ = v;
        *out++ = v;
    }
}

int pbso_read_census(pbso_engine *e, unsigned long long *out, size_t n) {
    NEED(e);
    if (!out) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->read_census(out, n);
    GUARD_END(e)
}

int pbso_get_info(pbso_engine *e, pbso_engine_info *out) {
    NEED(e);
    if (!out) return PBSO_ERR_INVALID;
    GUARD_BEGIN
    { int drc_ = e->impl->drain_submit(); if (drc_ != PBSO_OK) return drc_; }
    return e->impl->info(out);
    GUARD_END(e)
}

}  // extern "C"
