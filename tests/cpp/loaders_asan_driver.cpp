// Exercises openpbso_amd/csrc/loaders.cpp (the product's GPU-free file readers) under AddressSanitizer +
// UBSan on the CPU build: golden fixtures, and every truncation / byte corruption of them.
// Usage: loaders_asan_check <file>...      A sanitizer report aborts with a non-zero status.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

// (capi.cpp, which owns pbso_ffat_map_free, pulls in the whole engine: the two lines are repeated here)
extern "C" void pbso_ffat_map_free(pbso_ffat_map *m) {
    if (m && m->psi) { std::free((void *)m->psi); m->psi = nullptr; m->n_psi = 0; }
}

static bool ends_with(const std::string &s, const char *suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

static void parse(const unsigned char *p, size_t n) {
    std::vector<unsigned char> exact(p, p + n);          // exact-size copy: an overread trips ASan
    pbso_ffat_map m;
    std::memset(&m, 0, sizeof(m));
    if (pbso::parse_fatcube(exact.data(), exact.size(), &m) == PBSO_OK) pbso_ffat_map_free(&m);
}

int main(int argc, char **argv) {
    int n_files = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string path = argv[i];
        std::vector<unsigned char> b;
        if (pbso::read_file_bytes(path.c_str(), b) != PBSO_OK) { std::fprintf(stderr, "cannot read %s\n", argv[i]); return 3; }
        ++n_files;
        if (ends_with(path, ".fatcube")) {
            pbso_ffat_map m;
            std::memset(&m, 0, sizeof(m));
            if (pbso::parse_fatcube(b.data(), b.size(), &m) != PBSO_OK) std::printf("rejected: %s\n", argv[i]);
            else pbso_ffat_map_free(&m);
            for (size_t cut = 0; cut < b.size(); ++cut) parse(b.data(), cut);
            for (size_t k = 0; k < b.size(); ++k)
                for (unsigned char x : {0xFF, 0x80, 0x01}) {
                    std::vector<unsigned char> t(b);
                    t[k] ^= x;
                    parse(t.data(), t.size());
                }
        } else if (ends_with(path, ".modes")) {
            int nd = 0, nm = 0;
            std::vector<double> om, md;
            if (pbso::load_modes_file(path.c_str(), &nd, &nm, om, md) != PBSO_OK) std::printf("rejected: %s\n", argv[i]);
            else (void)pbso::num_modes_audible(om, 2500.0, 20000.0);
            char tmp[] = "/tmp/pbso_lasan_XXXXXX";
            const int fd = mkstemp(tmp);
            if (fd >= 0) {
                FILE *f = fdopen(fd, "wb");
                for (size_t cut : {(size_t)0, (size_t)3, (size_t)8, b.size() / 2, b.size() ? b.size() - 1 : 0}) {
                    if (cut > b.size()) continue;
                    f = std::freopen(tmp, "wb", f);
                    std::fwrite(b.data(), 1, cut, f);
                    std::fflush(f);
                    (void)pbso::load_modes_file(tmp, &nd, &nm, om, md);
                }
                std::fclose(f);
                std::remove(tmp);
            }
        } else if (ends_with(path, ".obj")) {
            std::vector<double> V, VN;
            std::vector<int> F;
            if (pbso::load_obj_file(path.c_str(), V, F, VN) != PBSO_OK) std::printf("rejected: %s\n", argv[i]);
        } else if (ends_with(path, ".txt")) {
            double mat[5];
            (void)pbso::load_material_file(path.c_str(), mat);
        }
    }
    std::vector<std::string> names;
    (void)pbso::list_dir_files("/nonexistent_dir_for_the_asan_check", ".fatcube", names);
    std::printf("loaders asan/ubsan check ok (%d files)\n", n_files);
    return 0;
}
