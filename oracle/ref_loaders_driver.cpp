// TEST INFRASTRUCTURE ONLY.  Driver (our code) around the reference's own
// std-only loader headers, included from where they lie under /root/reference
// (never copied): ModeData.h (read/write/numModesAudible) and ModalMaterial.h
// (Read).  Output is hex-float text that tests compare with the oracle and the
// product loaders; fixtures made with it live in tests/golden/.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "ModeData.h"
#include "ModalMaterial.h"

static int usage() {
    std::fprintf(stderr,
        "usage: ref_loaders modes_dump <file>\n"
        "       ref_loaders modes_roundtrip <in> <out>\n"
        "       ref_loaders audible <file> <density> <freq>\n"
        "       ref_loaders material <file>\n");
    return 2;
}

int main(int argc, char **argv) {
    if (argc < 3) return usage();
    const std::string cmd = argv[1];
    if (cmd == "modes_dump") {
        ModeData<double> md;
        md.read(argv[2]);
        std::printf("%d %d\n", md.numDOF(), md.numModes());
        for (int i = 0; i < md.numModes(); ++i) std::printf("%a\n", md.omegaSquared(i));
        for (int i = 0; i < md.numModes(); ++i)
            for (int j = 0; j < md.numDOF(); ++j) std::printf("%a\n", md.mode(i).at(j));
        return 0;
    }
    if (cmd == "modes_roundtrip" && argc >= 4) {
        ModeData<double> md;
        md.read(argv[2]);
        md.write(argv[3]);
        return 0;
    }
    if (cmd == "audible" && argc >= 5) {
        ModeData<double> md;
        md.read(argv[2]);
        std::printf("%d\n", md.numModesAudible(std::atof(argv[3]), std::atof(argv[4])));
        return 0;
    }
    if (cmd == "material") {
        ModalMaterial<double> *m = ModalMaterial<double>::Read(argv[2]);
        if (!m) { std::printf("null\n"); return 0; }
        std::printf("%a %a %a %a %a\n", m->density, m->youngsModulus, m->poissonRatio,
                    m->alpha, m->beta);
        delete m;
        return 0;
    }
    return usage();
}
