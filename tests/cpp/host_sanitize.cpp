// AddressSanitizer / UBSan run of the host-only parts of the C++ class surface (rapidnet_amd/csrc/host/): the null-space
// routine (Engine::calculateMatLandMatLhat, Engine.cu:466-669) on random full-rank and rank-deficient E, checked for
// E L = 0, L'L = I and E Lhat = -Ed.  (The JSON loaders run under the same sanitizers through `test_host loaders`.)
// Built and run by tests/test_host_sanitized.py; exit code 0 = clean.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../rapidnet_amd/csrc/host/NullSpace.hpp"

#define CHECK(c)                                                                                                        \
    do {                                                                                                                \
        if (!(c)) { std::fprintf(stderr, "check failed at line %d: %s\n", __LINE__, #c); std::exit(2); }                \
    } while (0)

int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> u(-1, 1);
    int runs = 0;
    for (int trial = 0; trial < 200; trial++) {
        const int nu = 2 + rng() % 12, ne = 1 + rng() % (nu - 1), nd = 1 + rng() % 5;
        std::vector<double> E((size_t)ne * nu), Ed((size_t)ne * nd), L, Lhat;
        for (auto &v : E) v = u(rng);
        for (auto &v : Ed) v = u(rng);
        const bool deficient = trial % 5 == 4 && ne >= 2;
        if (deficient) for (int j = 0; j < nu; j++) E[(size_t)j * ne + ne - 1] = 2 * E[(size_t)j * ne];   // last row = 2 x first row
        if (deficient) for (int j = 0; j < nd; j++) Ed[(size_t)j * ne + ne - 1] = 2 * Ed[(size_t)j * ne];
        const int rank = computeNullSpaceAndParticular(E.data(), Ed.data(), ne, nu, nd, L, Lhat);
        CHECK(rank == (deficient ? ne - 1 : ne));
        const int nv = nu - ne;
        CHECK((int)L.size() >= nu * nv && (int)Lhat.size() == nu * nd);
        for (int c = 0; c < nv; c++) {
            for (int r = 0; r < ne; r++) {      // E L = 0
                double s = 0;
                for (int j = 0; j < nu; j++) s += E[(size_t)j * ne + r] * L[(size_t)c * nu + j];
                CHECK(std::fabs(s) < 1e-9);
            }
            for (int c2 = 0; c2 <= c; c2++) {  // L'L = I
                double s = 0;
                for (int j = 0; j < nu; j++) s += L[(size_t)c * nu + j] * L[(size_t)c2 * nu + j];
                CHECK(std::fabs(s - (c == c2 ? 1.0 : 0.0)) < 1e-9);
            }
        }
        for (int c = 0; c < nd; c++)            // E Lhat = -Ed (consistent systems: full rank, or the duplicated row)
            for (int r = 0; r < ne; r++) {
                double s = 0;
                for (int j = 0; j < nu; j++) s += E[(size_t)j * ne + r] * Lhat[(size_t)c * nu + j];
                CHECK(std::fabs(s + Ed[(size_t)c * ne + r]) < 1e-8);
            }
        runs++;
    }
    std::printf("null-space runs %d\n", runs);
    return 0;
}
