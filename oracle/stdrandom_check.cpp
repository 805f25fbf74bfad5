// TEST INFRASTRUCTURE ONLY.  Emits n draws of the exact libstdc++ objects that
// AutoregressiveForce default-constructs (reference forces.h:71-72), as
// hex-float text, so tests can pin oracle/pbso_oracle.c:or_rng_normal().
#include <cstdio>
#include <cstdlib>
#include <random>
int main(int argc, char **argv) {
    int n = argc > 1 ? std::atoi(argv[1]) : 16;
    std::default_random_engine generator;
    std::normal_distribution<double> distribution;
    for (int i = 0; i < n; ++i) std::printf("%a\n", distribution(generator));
    return 0;
}
