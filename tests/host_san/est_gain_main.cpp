// Test driver (CPU only): the library's host builder of the estimator gain, G = pinv(A_s'A_s) A_s' (fmpc_host.cpp,
// reference README.md:478), on a model read from a raw file:  est_gain_main <A_s.bin> <p> <nx> <G.bin>
// A_s: p x nx column-major doubles; G: nx x p row-major doubles; prints the numerical rank.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fmpc_host.h"

int main(int argc, char** argv) {
    if (argc != 5) return 2;
    const int p = atoi(argv[2]), nx = atoi(argv[3]);
    std::vector<double> A((size_t)p * nx), G;
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(A.data(), sizeof(double), A.size(), f) != A.size()) return 3;
    fclose(f);
    const int rank = fmpc_host_estimator_gain(A.data(), p, nx, G);
    f = fopen(argv[4], "wb");
    if (!f || fwrite(G.data(), sizeof(double), G.size(), f) != G.size()) return 4;
    fclose(f);
    printf("%d\n", rank);
    return 0;
}
