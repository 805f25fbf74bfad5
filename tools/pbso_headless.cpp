// pbso_headless -- headless counterpart of the reference's GUI tool for the hot path.
//
// Takes the reference's command-line flags (tools/real_time_modal_sound.cpp:42-64:
// -d/--data_dir, -name/--obj_name, -m/--mesh, -s/--surf_mode, -t/--material,
// -p/--ffat_map) and its directory convention (:480-501), builds the solver the way
// BuildSolver does (:309-345) through the C ABI, replaces the mouse by a hit script and
// the camera by a listener script, steps N buffers on the MI355X and writes what
// PaModalCallback would have played (:207-210, sound / 1e10) as a mono float32 WAV.
//
//   --hits FILE      lines: <buffer> <vertex_id> <nx> <ny> <nz> [point|gauss <width_us>|ar]
//                    or:    <buffer> <vertex_id> - [point|gauss <width_us>|ar]   (normal = VN.row(vid) of the mesh:
//                    igl::per_vertex_normals of the .obj, tools/...:509,607 -- needs -m / -d)
//   --listener FILE  lines: <buffer> <x> <y> <z>          (computeTransfer at that buffer)
//   --buffers N      number of 513-sample buffers (default 86 ~ 1 s)
//   --out FILE       output WAV (default out.wav);  --raw FILE also dumps the fp32 sound values
//   --devices 0,1,.. several GPUs through the C ABI's device group (one engine per GPU, objects sharded by the sum of their
//                    modes, RCCL only for the final collective): the scene is --copies K instances of the object (default: one
//                    per device), copy c hears the hit and listener scripts c * --copy-shift buffers later (default 1), and
//                    the WAV is their MIX (PBSO_GATHER_MIX: every GPU sums its objects, the GPUs all-reduce one row)
#include <dirent.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "openpbso_amd.h"

static void die(const std::string &msg) {
    std::fprintf(stderr, "pbso_headless: %s\n", msg.c_str());
    std::exit(1);
}
static void check(pbso_engine *e, int rc, const char *what) {
    if (rc < 0) die(std::string(what) + ": " + pbso_status_string(rc) + ": " + (e ? pbso_last_error(e) : ""));
}

// ListDirFiles(d, names, ".tet.obj") + Basename + prefix up to the first '.', tools/...:483-487
static std::string guess_name(const std::string &dir) {
    DIR *d = opendir(dir.c_str());
    if (!d) die("cannot open data dir " + dir);
    std::string found;
    while (dirent *ent = readdir(d)) {
        const std::string f = dir + "/" + ent->d_name;
        if (ent->d_name[0] != '.' && f.find(".tet.obj") != std::string::npos) { found = ent->d_name; break; }
    }
    closedir(d);
    if (found.empty()) die("no *.tet.obj in " + dir);
    return found.substr(0, found.find_first_of("."));
}

static void write_wav_f32(const std::string &path, const std::vector<float> &mono, int rate) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) die("cannot write " + path);
    const uint32_t data_bytes = (uint32_t)(mono.size() * 4), riff = 36 + data_bytes, fmt_len = 16, byte_rate = rate * 4;
    const uint16_t fmt = 3 /* IEEE float */, ch = 1, align = 4, bits = 32;
    const uint32_t r = rate;
    std::fwrite("RIFF", 1, 4, f); std::fwrite(&riff, 4, 1, f); std::fwrite("WAVEfmt ", 1, 8, f);
    std::fwrite(&fmt_len, 4, 1, f); std::fwrite(&fmt, 2, 1, f); std::fwrite(&ch, 2, 1, f);
    std::fwrite(&r, 4, 1, f); std::fwrite(&byte_rate, 4, 1, f); std::fwrite(&align, 2, 1, f); std::fwrite(&bits, 2, 1, f);
    std::fwrite("data", 1, 4, f); std::fwrite(&data_bytes, 4, 1, f);
    std::fwrite(mono.data(), 4, mono.size(), f);
    std::fclose(f);
}

struct Hit { long b; pbso_force_msg m; };
struct Pos { long b; double p[3]; };

// the scene on several GPUs (include/openpbso_amd.h "device group")
static int run_group(const std::vector<int> &devices, int copies, int shift, const std::string &modes, const std::string &material,
                     const std::string &ffat, const std::vector<Hit> &hits, const std::vector<Pos> &path, int n_buffers,
                     std::vector<float> &sound) {
    auto gcheck = [](pbso_group *g, int rc, const char *what) {
        if (rc < 0) die(std::string(what) + ": " + pbso_status_string(rc) + ": " + (g ? pbso_group_last_error(g) : ""));
    };
    // BuildSolver's inputs (tools/...:309-345), read once: the audible mode count is what the shards are balanced by
    double mat[5];
    if (pbso_material_read(material.c_str(), mat) != PBSO_OK) die("cannot read material file " + material);
    int n_dof = 0, n_modes = 0;
    double *om = nullptr, *md = nullptr;
    if (pbso_modes_read(modes.c_str(), &n_dof, &n_modes, &om, &md) != PBSO_OK) die("cannot read modes " + modes);
    double max_freq = 20000.;                            // tools/...:316-329
    if (!ffat.empty()) {
        std::ifstream f((ffat + "/freq_threshold.txt").c_str());
        if (f) f >> max_freq;
    }
    const int n_aud = pbso_num_modes_audible(om, n_modes, mat[0], max_freq);
    pbso_group_desc gd;
    std::memset(&gd, 0, sizeof(gd));
    gd.abi_version = PBSO_ABI_VERSION;
    gd.devices = devices.data();
    gd.n_devices = (int)devices.size();
    gd.engine.abi_version = PBSO_ABI_VERSION;
    gd.engine.qnorm_mode = PBSO_QNORM_OFF;
    pbso_group *g = nullptr;
    gcheck(g, pbso_group_create(&gd, &g), "group_create");
    std::vector<int> mc(copies, n_aud);
    gcheck(g, pbso_group_plan(g, mc.data(), copies), "group_plan");
    pbso_object_desc od;
    std::memset(&od, 0, sizeof(od));
    od.n_modes = n_aud; od.n_omega = n_modes; od.omega_squared = om;
    od.density = mat[0]; od.alpha = mat[3]; od.beta = mat[4];
    od.n_dof = n_dof; od.mode_shapes = md;
    for (int c = 0; c < copies; ++c) {
        gcheck(g, pbso_group_add_object(g, c, &od), "group_add_object");
        int rank = 0, local = 0;
        gcheck(g, pbso_group_owner(g, c, &rank, &local), "group_owner");
        pbso_engine *e = pbso_group_engine(g, rank);
        if (e && !ffat.empty()) check(e, pbso_object_read_ffat_maps(e, local, ffat.c_str()), "read_ffat_maps");
    }
    pbso_free(om);
    pbso_free(md);
    gcheck(g, pbso_group_finalize(g), "group_finalize");
    for (int c = 0; c < copies; ++c) {
        int rank = 0, local = 0;
        gcheck(g, pbso_group_owner(g, c, &rank, &local), "group_owner");
        pbso_engine *e = pbso_group_engine(g, rank);
        if (path.empty()) check(e, pbso_set_use_transfer(e, local, 0, 0), "set_use_transfer");
        for (const Pos &p : path) check(e, pbso_compute_transfer(e, local, p.p, p.b + (long)c * shift), "compute_transfer");
        for (const Hit &h : hits) {
            const int rc = pbso_group_enqueue_force(g, c, &h.m, h.b + (long)c * shift);
            gcheck(g, rc, "group_enqueue_force");
            if (rc == 0) die("force queue full");
        }
    }
    gcheck(g, pbso_group_step(g, n_buffers), "group_step");
    gcheck(g, pbso_group_gather(g, PBSO_GATHER_MIX), "group_gather");
    sound.resize((size_t)n_buffers * PBSO_FRAMES_PER_BUFFER);
    gcheck(g, pbso_group_read_result(g, 0, sound.data(), sound.size()), "group_read_result");
    std::printf("%d copies x %d audible modes on %d device(s): ranks own", copies, n_aud, (int)devices.size());
    for (int r = 0; r < (int)devices.size(); ++r) {
        int lo = 0, hi = 0;
        pbso_group_rank_span(g, r, &lo, &hi);
        std::printf(" [%d, %d)", lo, hi);
    }
    std::printf("\n");
    pbso_group_destroy(g);
    return 0;
}

int main(int argc, char **argv) {
    std::string d, name, mesh, modes, material, ffat, hits, listener, out = "out.wav", raw, devices_arg;
    int n_buffers = 86, copies = 0, copy_shift = 1;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
        if (a == "-d" || a == "--data_dir") d = val();
        else if (a == "-name" || a == "--obj_name") name = val();
        else if (a == "-m" || a == "--mesh") mesh = val();
        else if (a == "-s" || a == "--surf_mode") modes = val();
        else if (a == "-t" || a == "--material") material = val();
        else if (a == "-p" || a == "--ffat_map") ffat = val();
        else if (a == "--hits") hits = val();
        else if (a == "--listener") listener = val();
        else if (a == "--buffers") n_buffers = std::atoi(val().c_str());
        else if (a == "--out") out = val();
        else if (a == "--raw") raw = val();
        else if (a == "--devices") devices_arg = val();
        else if (a == "--copies") copies = std::atoi(val().c_str());
        else if (a == "--copy-shift") copy_shift = std::atoi(val().c_str());
        else die("unknown flag " + a);
    }
    if (!d.empty()) {                                   // fixed directory structure, tools/...:480-495
        if (name.empty()) name = guess_name(d);
        std::printf("object name: %s\n", name.c_str());
        mesh = d + "/" + name + ".tet.obj";
        modes = d + "/" + name + "_surf.modes";
        material = d + "/" + name + "_material.txt";
        ffat = d + "/" + name + "_ffat_maps";
    }
    if (modes.empty() || material.empty()) die("need -d <dir> or -s <modes> -t <material> [-m <obj>] [-p <ffat dir>]");

    // assert(modes->numDOF() == V.rows()*3), tools/...:515
    int n_dof = 0, n_modes = 0;
    {
        double *om = nullptr, *md = nullptr;
        const int mrc = pbso_modes_read(modes.c_str(), &n_dof, &n_modes, &om, &md);
        if (mrc != PBSO_OK) die(std::string("modes_read: ") + pbso_status_string(mrc) + ": " + modes);
        pbso_free(om);
        pbso_free(md);
    }
    // igl::read_triangle_mesh(obj_file, V, F); igl::per_vertex_normals(V, F, VN);   tools/...:508-509
    std::vector<double> VN;
    if (!mesh.empty()) {
        int nv = 0, nf = 0;
        double *V = nullptr, *vn = nullptr;
        int *F = nullptr;
        const int orc = pbso_obj_read(mesh.c_str(), &nv, &nf, &V, &F, &vn);
        if (orc == PBSO_OK) {
            if (nv * 3 != n_dof) die("DOFs mismatch: .obj has " + std::to_string(nv) + " vertices, modes have nDOF " + std::to_string(n_dof));
            VN.assign(vn, vn + 3 * (size_t)nv);
            std::printf("mesh: %d vertices, %d triangles\n", nv, nf);
            pbso_free(V);
            pbso_free(F);
            pbso_free(vn);
        } else if (d.empty()) {
            die("cannot read mesh " + mesh);
        }
    }
    // the scripts that stand in for the camera and the mouse
    std::vector<Pos> path;
    if (!listener.empty()) {
        std::ifstream f(listener);
        if (!f) die("cannot read " + listener);
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream iss(line);
            Pos p;
            if (!(iss >> p.b >> p.p[0] >> p.p[1] >> p.p[2])) die("bad listener line: " + line);
            path.push_back(p);
        }
    }
    std::vector<Hit> hit_list;
    if (!hits.empty()) {
        std::ifstream f(hits);
        if (!f) die("cannot read " + hits);
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream iss(line);
            long b; int vid; double n[3]; std::string type = "point", tok;
            if (!(iss >> b >> vid >> tok)) die("bad hit line: " + line);
            if (tok == "-") {                                // the tool's own path: vn = VN.row(vid), tools/...:607
                if (VN.empty()) die("hit without a normal needs the mesh (-m / -d)");
                if (vid < 0 || 3 * (size_t)vid + 2 >= VN.size()) die("vertex id out of range: " + line);
                for (int j = 0; j < 3; ++j) n[j] = VN[3 * (size_t)vid + j];
            } else {
                n[0] = std::atof(tok.c_str());
                if (!(iss >> n[1] >> n[2])) die("bad hit line: " + line);
            }
            iss >> type;
            Hit h;
            h.b = b;
            std::memset(&h.m, 0, sizeof(h.m));
            h.m.data_kind = PBSO_DATA_VERTEX;
            h.m.vids[0] = vid;
            const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);   // VN.row(vid).normalized(), tools/...:607
            for (int j = 0; j < 3; ++j) h.m.vn[j] = n[j] / len;
            if (type == "gauss") { h.m.force_type = PBSO_GAUSSIAN_FORCE; iss >> h.m.gaussian_width_us; }
            else if (type == "ar") h.m.force_type = PBSO_AUTOREGRESSIVE_FORCE;
            else h.m.force_type = PBSO_POINT_FORCE;
            hit_list.push_back(h);
        }
    }

    std::vector<float> sound((size_t)n_buffers * PBSO_FRAMES_PER_BUFFER);
    double device_ms = 0;
    if (!devices_arg.empty()) {
        std::vector<int> devices;
        std::istringstream ds(devices_arg);
        std::string tok;
        while (std::getline(ds, tok, ',')) devices.push_back(std::atoi(tok.c_str()));
        if (devices.empty()) die("--devices needs a list of HIP ordinals");
        run_group(devices, copies > 0 ? copies : (int)devices.size(), copy_shift, modes, material, ffat, hit_list, path, n_buffers, sound);
    } else {
        pbso_engine_desc desc;
        std::memset(&desc, 0, sizeof(desc));
        desc.abi_version = PBSO_ABI_VERSION;
        desc.qnorm_mode = PBSO_QNORM_OFF;
        pbso_engine *e = nullptr;
        int rc = pbso_engine_create(&desc, &e);
        check(e, rc, "engine_create");
        int obj = -1, n_aud = 0;
        check(e, pbso_add_object_from_files(e, modes.c_str(), material.c_str(), ffat.empty() ? nullptr : ffat.c_str(), &obj, &n_aud),
              "add_object_from_files");
        std::printf("modes: %d of %d audible, nDOF %d\n", n_aud, n_modes, n_dof);
        check(e, pbso_finalize(e), "finalize");
        for (const Pos &p : path) check(e, pbso_compute_transfer(e, obj, p.p, p.b), "compute_transfer");
        if (path.empty()) check(e, pbso_set_use_transfer(e, obj, 0, 0), "set_use_transfer");   // unit transfer
        for (const Hit &h : hit_list) {
            rc = pbso_enqueue_force(e, obj, &h.m, h.b);
            check(e, rc, "enqueue_force");
            if (rc == 0) die("force queue full");
        }
        check(e, pbso_step(e, n_buffers), "step");
        check(e, pbso_read_audio(e, sound.data(), sound.size()), "read_audio");
        pbso_engine_info info;
        check(e, pbso_get_info(e, &info), "get_info");
        device_ms = info.last_step_device_ms;
        pbso_engine_destroy(e);
    }
    std::vector<float> mono(sound.size());
    for (size_t i = 0; i < sound.size(); ++i) mono[i] = (float)((double)sound[i] / 1E10);   // tools/...:208
    write_wav_f32(out, mono, PBSO_SAMPLE_RATE);
    if (!raw.empty()) {
        FILE *f = std::fopen(raw.c_str(), "wb");
        if (!f) die("cannot write " + raw);
        std::fwrite(sound.data(), 4, sound.size(), f);
        std::fclose(f);
    }
    std::printf("%d buffers (%.3f s of audio) in %.3f ms on the device -> %s\n", n_buffers,
                n_buffers * (double)PBSO_FRAMES_PER_BUFFER / PBSO_SAMPLE_RATE, device_ms, out.c_str());
    return 0;
}

