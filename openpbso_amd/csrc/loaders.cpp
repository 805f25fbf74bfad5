// File formats the path consumes, read without
// protobuf / libigl:
//   <name>_surf.modes       ModeData<double>::read            (ModeData.h:61-83)
//   <name>_material.txt     ModalMaterial<double>::Read       (ModalMaterial.h:35-55)
//   *.fatcube               FFAT_Map_Serialize_Double::Load   (ffat_map_serialize.h:166-254,
//                           wire format of ffat_map.proto:8-51)
//   directory scan          ListDirFiles                      (io.cpp:18-35)
#include <dirent.h>
#include <sys/stat.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "engine.h"

namespace pbso {

int read_file_bytes(const char *path, std::vector<unsigned char> &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return PBSO_ERR_IO;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0, std::ios::beg);
    out.resize((size_t)std::max<std::streamoff>(n, 0));
    if (n > 0) f.read((char *)out.data(), n);
    return f.good() || f.eof() ? PBSO_OK : PBSO_ERR_IO;
}

// native-endian: int32 nDOF, int32 nModes, double omegaSquared[nModes], double modes[nModes][nDOF]
int load_modes_file(const char *path, int *n_dof, int *n_modes, std::vector<double> &omega2,
                    std::vector<double> &modes) {
    std::vector<unsigned char> bytes;
    if (read_file_bytes(path, bytes) != PBSO_OK) return PBSO_ERR_IO;
    if (bytes.size() < 8) return PBSO_ERR_IO;
    int32_t nd, nm;
    std::memcpy(&nd, bytes.data(), 4);
    std::memcpy(&nm, bytes.data() + 4, 4);
    if (nd < 0 || nm < 0) return PBSO_ERR_IO;
    const size_t need = 8 + sizeof(double) * ((size_t)nm + (size_t)nm * (size_t)nd);
    if (bytes.size() < need) return PBSO_ERR_IO;
    omega2.resize(nm);
    modes.resize((size_t)nm * nd);
    std::memcpy(omega2.data(), bytes.data() + 8, sizeof(double) * (size_t)nm);
    std::memcpy(modes.data(), bytes.data() + 8 + sizeof(double) * (size_t)nm, sizeof(double) * (size_t)nm * nd);
    *n_dof = nd;
    *n_modes = nm;
    return PBSO_OK;
}

// ModeData::numModesAudible, ModeData.h:120-148 (assumes ascending eigenvalues)
int num_modes_audible(const std::vector<double> &omega2, double density, double audible_freq) {
    auto freq = [&](double os) { return std::sqrt(os / density) / (2. * M_PI); };
    if (omega2.empty() || freq(omega2.front()) > audible_freq) return 0;
    if (freq(omega2.back()) <= audible_freq) return (int)omega2.size();
    size_t ii;
    for (ii = 0; ii < omega2.size(); ++ii)
        if (freq(omega2[ii]) > audible_freq) break;
    return (int)ii;
}

// leading '#' lines skipped; first other line: density youngsModulus poissonRatio alpha beta
int load_material_file(const char *path, double out[5]) {
    std::ifstream stream(path);
    if (!stream) return PBSO_ERR_IO;
    std::string line;
    while (std::getline(stream, line)) {
        if (line[0] != '#') break;
    }
    std::istringstream iss(line);
    for (int i = 0; i < 5; ++i) out[i] = 0.0;
    iss >> out[0];
    iss >> out[1];
    iss >> out[2];
    iss >> out[3];
    iss >> out[4];
    return PBSO_OK;
}

// io.cpp:18-35: entries not starting with '.', that stat() finds, whose FULL
// path (directory included) contains `contains`; readdir order.
int list_dir_files(const char *dirname, const char *contains, std::vector<std::string> &names) {
    DIR *dir = opendir(dirname);
    if (!dir) return PBSO_ERR_IO;
    struct dirent *ent;
    while ((ent = readdir(dir)) != nullptr) {
        const std::string f = std::string(dirname) + "/" + ent->d_name;
        struct stat st;
        if (stat(f.c_str(), &st) == 0 && ent->d_name[0] != '.' && contains &&
            f.find(contains) != std::string::npos)
            names.push_back(f);
    }
    closedir(dir);
    return PBSO_OK;
}

// ---- <name>.tet.obj: igl::read_triangle_mesh + igl::per_vertex_normals, tools/real_time_modal_sound.cpp:508-509 ----
// Only `v` and `f` records matter to the path.  Face indices are 1-based, may be negative (relative to the
// vertices read so far) and may carry /vt/vn suffixes; polygons are cut into a triangle fan (what
// read_triangle_mesh does through polygon_mesh_to_triangle_mesh).  Normals: libigl's default weighting is by
// face area -- n_v = normalize(sum over incident faces of double-area x unit face normal) = normalize(sum of
// the faces' edge cross products).  libigl is an un-vendored submodule of the reference (README pins a
// modified 2.1.0), so this weighting is its documented default, not pinned by code under /root/reference.
int load_obj_file(const char *path, std::vector<double> &V, std::vector<int> &F, std::vector<double> &VN) {
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return PBSO_ERR_IO;
    V.clear();
    F.clear();
    std::vector<char> buf(1 << 16);
    std::vector<int> poly;
    int rc = PBSO_OK;
    while (std::fgets(buf.data(), (int)buf.size(), fp)) {
        const char *p = buf.data();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char *end = nullptr;
            const char *q = p + 1;
            double x[3];
            for (int j = 0; j < 3; ++j) {
                x[j] = std::strtod(q, &end);
                if (end == q) { rc = PBSO_ERR_IO; break; }
                q = end;
            }
            if (rc != PBSO_OK) break;
            V.insert(V.end(), x, x + 3);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            poly.clear();
            const char *q = p + 1;
            const int nv = (int)(V.size() / 3);
            for (;;) {
                while (*q == ' ' || *q == '\t') ++q;
                if (*q == 0 || *q == '\n' || *q == '\r' || *q == '#') break;
                char *end = nullptr;
                const long idx = std::strtol(q, &end, 10);
                if (end == q) { rc = PBSO_ERR_IO; break; }
                const long v = idx > 0 ? idx - 1 : nv + idx;             // negative: relative to the last vertex read
                if (idx == 0 || v < 0 || v >= nv) { rc = PBSO_ERR_IO; break; }
                poly.push_back((int)v);
                q = end;
                while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;   // skip /vt/vn
            }
            if (rc != PBSO_OK) break;
            if (poly.size() < 3) { rc = PBSO_ERR_IO; break; }
            for (size_t k = 1; k + 1 < poly.size(); ++k) {
                F.push_back(poly[0]);
                F.push_back(poly[k]);
                F.push_back(poly[k + 1]);
            }
        }
    }
    std::fclose(fp);
    if (rc != PBSO_OK) return rc;
    VN.assign(V.size(), 0.0);
    for (size_t f = 0; f + 2 < F.size(); f += 3) {
        const double *a = &V[3 * (size_t)F[f]], *b = &V[3 * (size_t)F[f + 1]], *c = &V[3 * (size_t)F[f + 2]];
        const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        const double n[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 3; ++j) VN[3 * (size_t)F[f + k] + j] += n[j];
    }
    for (size_t v = 0; v + 2 < VN.size(); v += 3) {
        const double len = std::sqrt(VN[v] * VN[v] + VN[v + 1] * VN[v + 1] + VN[v + 2] * VN[v + 2]);
        if (len > 0) { VN[v] /= len; VN[v + 1] /= len; VN[v + 2] /= len; }      // a vertex of no face keeps (0, 0, 0)
    }
    return PBSO_OK;
}

// ---- proto3 wire reader -----------------------------------------------------
namespace {
struct Rd {
    const unsigned char *p, *end;
    bool err = false;
    uint64_t varint() {
        uint64_t v = 0;
        for (int shift = 0; p < end && shift < 64; shift += 7) {
            const unsigned char c = *p++;
            v |= (uint64_t)(c & 0x7f) << shift;
            if (!(c & 0x80)) return v;
        }
        err = true;
        return 0;
    }
    double f64() {
        double v = 0;
        if (end - p < 8) { err = true; return 0; }
        std::memcpy(&v, p, 8);
        p += 8;
        return v;
    }
    Rd sub() {
        const uint64_t n = varint();
        Rd s{p, p};
        if (err || (uint64_t)(end - p) < n) { err = true; s.err = true; return s; }
        s.end = p + n;
        p += n;
        return s;
    }
    void skip(int wt) {
        switch (wt) {
        case 0: (void)varint(); break;
        case 1: if (end - p < 8) err = true; else p += 8; break;
        case 2: { Rd s = sub(); if (s.err) err = true; break; }
        case 5: if (end - p < 4) err = true; else p += 4; break;
        default: err = true;
        }
    }
};

// message vec { repeated double item = 1; } -- packed or one-by-one
bool read_vec(Rd s, std::vector<double> &out) {
    while (s.p < s.end && !s.err) {
        const uint64_t key = s.varint();
        const int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) { Rd q = s.sub(); while (q.p < q.end && !q.err) out.push_back(q.f64()); if (q.err) s.err = true; }
        else if (fn == 1 && wt == 1) out.push_back(s.f64());
        else s.skip(wt);
    }
    return !s.err;
}
bool read_vec_i(Rd s, std::vector<int> &out) {
    while (s.p < s.end && !s.err) {
        const uint64_t key = s.varint();
        const int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) { Rd q = s.sub(); while (q.p < q.end && !q.err) out.push_back((int)(int64_t)q.varint()); if (q.err) s.err = true; }
        else if (fn == 1 && wt == 0) out.push_back((int)(int64_t)s.varint());
        else s.skip(wt);
    }
    return !s.err;
}
// message mat { repeated vec item = 1; }
bool read_mat(Rd s, std::vector<std::vector<double>> &out) {
    while (s.p < s.end && !s.err) {
        const uint64_t key = s.varint();
        const int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) { out.emplace_back(); if (!read_vec(s.sub(), out.back())) s.err = true; }
        else s.skip(wt);
    }
    return !s.err;
}
bool read_mat_i(Rd s, std::vector<std::vector<int>> &out) {
    while (s.p < s.end && !s.err) {
        const uint64_t key = s.varint();
        const int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) { out.emplace_back(); if (!read_vec_i(s.sub(), out.back())) s.err = true; }
        else s.skip(wt);
    }
    return !s.err;
}
bool take3(Rd s, double out[3]) {
    std::vector<double> v;
    if (!read_vec(s, v) || v.size() != 3) return false;       // DESERIALIZE_VEC(.., false) asserts the size
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
    return true;
}

// ffat_map_t_1 -> FFAT_Map<double,1>, ffat_map_serialize.h:176-222
bool read_shell(Rd s, pbso_ffat_map *m) {
    bool ok = true;
    unsigned seen = 0;
    while (s.p < s.end && !s.err && ok) {
        const uint64_t key = s.varint();
        const int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn >= 1 && fn <= 7) seen |= 1u << fn;
        if (fn == 1 && wt == 1) m->cell_size = s.f64();
        else if (fn == 2 && wt == 2) {
            std::vector<std::vector<double>> rows;
            ok = read_mat(s.sub(), rows) && rows.size() == 6;
            for (size_t i = 0; ok && i < 6; ++i) {
                ok = rows[i].size() >= 3;
                for (int j = 0; ok && j < 3; ++j) m->low_corners[i][j] = rows[i][j];
            }
        } else if (fn == 3 && wt == 2) {
            std::vector<std::vector<int>> rows;
            ok = read_mat_i(s.sub(), rows) && rows.size() == 6;
            for (size_t i = 0; ok && i < 6; ++i) {
                ok = rows[i].size() >= 2;
                if (ok) { m->n_elements[i][0] = rows[i][0]; m->n_elements[i][1] = rows[i][1]; }
            }
        } else if (fn == 4 && wt == 2) {
            std::vector<int> v;
            ok = read_vec_i(s.sub(), v) && v.size() == 6;
            for (size_t i = 0; ok && i < 6; ++i) m->strides[i] = v[i];
        } else if (fn == 5 && wt == 2) ok = take3(s.sub(), m->center);
        else if (fn == 6 && wt == 2) ok = take3(s.sub(), m->bbox_low);
        else if (fn == 7 && wt == 2) ok = take3(s.sub(), m->bbox_top);
        else s.skip(wt);
    }
    // the reference's loader asserts the fixed-size vectors (DESERIALIZE_VEC(.., false))
    // and GetMapVal indexes all six faces: a shell without them is unusable
    return ok && !s.err && (seen & 0xFC) == 0xFC;
}

// ffat_map_t_3 -> FFAT_Map<double,3>, ffat_map_serialize.h:223-253
bool read_map3(Rd s, pbso_ffat_map *m, bool *compressed) {
    bool ok = true;
    unsigned seen = 0;
    while (s.p < s.end && !s.err && ok) {
        const uint64_t key = s.varint();
        const int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn >= 1 && fn <= 6) seen |= 1u << fn;
        if (fn == 1 && wt == 1) m->k = s.f64();
        else if (fn == 2 && wt == 2) ok = take3(s.sub(), m->center3);
        else if (fn == 3 && wt == 2) ok = read_shell(s.sub(), m);
        else if (fn == 4 && wt == 0) *compressed = s.varint() != 0;
        else if (fn == 5 && wt == 2) {
            // psi: one vec per column of _Psi; the runtime reads column 0 only
            std::vector<std::vector<double>> cols;
            ok = read_mat(s.sub(), cols);
            if (ok) {
                const size_t rows = cols.empty() ? 0 : cols[0].size();
                std::free((void *)m->psi);
                double *psi = (double *)std::malloc(sizeof(double) * std::max<size_t>(rows, 1));
                if (rows) std::memcpy(psi, cols[0].data(), sizeof(double) * rows);
                m->psi = psi;
                m->n_psi = (int)rows;
            }
        } else if (fn == 6 && wt == 0) m->mode_id = (int)(int64_t)s.varint();
        else s.skip(wt);
    }
    return ok && !s.err && (seen & 0x2C) == 0x2C;      // center, shells and psi must be there
}
}  // namespace

int parse_fatcube(const unsigned char *bytes, size_t n, pbso_ffat_map *out) {
    std::memset(out, 0, sizeof(*out));          // proto3 defaults: modeid 0, is_compressed false
    Rd r{bytes, bytes + n};
    bool ok = true, compressed = false;
    while (r.p < r.end && !r.err && ok) {
        const uint64_t key = r.varint();
        const int fn = (int)(key >> 3), wt = (int)(key & 7);
        if (fn == 1 && wt == 2) ok = read_map3(r.sub(), out, &compressed);
        else r.skip(wt);
    }
    if (!ok || r.err || compressed) {
        // compressed Psi is dead at run time in the reference: refuse it loudly
        std::free((void *)out->psi);
        out->psi = nullptr;
        out->n_psi = 0;
        return PBSO_ERR_IO;
    }
    return PBSO_OK;
}

}  // namespace pbso
