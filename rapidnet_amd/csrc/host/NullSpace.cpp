// NullSpace.cpp -- see NullSpace.hpp.
#include "NullSpace.hpp"

#include <algorithm>
#include <cmath>

int computeNullSpaceAndParticular(const double *E, const double *Ed, int ne, int nu, int nd, std::vector<double> &L,
                                  std::vector<double> &Lhat) {
    // A = E' (nu x ne).  One-sided Jacobi on the columns of A: A V = U S, V orthogonal (ne x ne).
    std::vector<double> A((size_t)nu * ne), V((size_t)ne * ne, 0.0);
    for (int i = 0; i < ne; i++) for (int j = 0; j < nu; j++) A[j + (size_t)i * nu] = E[i + (size_t)j * ne];
    for (int i = 0; i < ne; i++) V[i + (size_t)i * ne] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < ne - 1; p++)
            for (int q = p + 1; q < ne; q++) {
                double app = 0, aqq = 0, apq = 0;
                for (int k = 0; k < nu; k++) {
                    const double a = A[k + (size_t)p * nu], b = A[k + (size_t)q * nu];
                    app += a * a; aqq += b * b; apq += a * b;
                }
                if (std::fabs(apq) <= 1e-300 || std::fabs(apq) <= 1e-15 * std::sqrt(app * aqq)) continue;
                off = std::max(off, std::fabs(apq) / std::sqrt(app * aqq));
                const double tau = (aqq - app) / (2 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
                const double c = 1 / std::sqrt(1 + t * t), s = c * t;
                for (int k = 0; k < nu; k++) {
                    const double a = A[k + (size_t)p * nu], b = A[k + (size_t)q * nu];
                    A[k + (size_t)p * nu] = c * a - s * b; A[k + (size_t)q * nu] = s * a + c * b;
                }
                for (int k = 0; k < ne; k++) {
                    const double a = V[k + (size_t)p * ne], b = V[k + (size_t)q * ne];
                    V[k + (size_t)p * ne] = c * a - s * b; V[k + (size_t)q * ne] = s * a + c * b;
                }
            }
        if (off < 1e-15) break;
    }
    // singular values and U1 = A V / S  (nu x ne)
    std::vector<double> S(ne), U1((size_t)nu * ne, 0.0);
    double smax = 0;
    for (int i = 0; i < ne; i++) {
        double n2 = 0;
        for (int k = 0; k < nu; k++) n2 += A[k + (size_t)i * nu] * A[k + (size_t)i * nu];
        S[i] = std::sqrt(n2);
        smax = std::max(smax, S[i]);
    }
    int rank = 0;
    for (int i = 0; i < ne; i++)
        if (S[i] > 1e-12 * smax) { rank++; for (int k = 0; k < nu; k++) U1[k + (size_t)i * nu] = A[k + (size_t)i * nu] / S[i]; }
    // Lhat = -pinv(E) Ed with pinv(E) = U1 S^-1 V'   (E = V S U1')
    Lhat.assign((size_t)nu * nd, 0.0);
    std::vector<double> W((size_t)ne * nd, 0.0);   // W = S^-1 V' Ed
    for (int j = 0; j < nd; j++)
        for (int i = 0; i < ne; i++) {
            if (S[i] <= 1e-12 * smax) continue;
            double s = 0;
            for (int k = 0; k < ne; k++) s += V[k + (size_t)i * ne] * Ed[k + (size_t)j * ne];
            W[i + (size_t)j * ne] = s / S[i];
        }
    for (int j = 0; j < nd; j++)
        for (int k = 0; k < nu; k++) {
            double s = 0;
            for (int i = 0; i < ne; i++) s += U1[k + (size_t)i * nu] * W[i + (size_t)j * ne];
            Lhat[k + (size_t)j * nu] = -s;
        }
    // L: orthonormal completion of range(U1) -- Gram-Schmidt (twice) of the unit vectors against U1 and the basis so far
    const int nvv = nu - rank;
    L.assign((size_t)nu * nvv, 0.0);
    std::vector<double> cand(nu);
    int found = 0;
    for (int e = 0; e < nu && found < nvv; e++) {
        std::fill(cand.begin(), cand.end(), 0.0);
        cand[e] = 1.0;
        for (int pass = 0; pass < 2; pass++) {
            for (int i = 0; i < ne; i++) {
                if (S[i] <= 1e-12 * smax) continue;
                double d = 0;
                for (int k = 0; k < nu; k++) d += U1[k + (size_t)i * nu] * cand[k];
                for (int k = 0; k < nu; k++) cand[k] -= d * U1[k + (size_t)i * nu];
            }
            for (int i = 0; i < found; i++) {
                double d = 0;
                for (int k = 0; k < nu; k++) d += L[k + (size_t)i * nu] * cand[k];
                for (int k = 0; k < nu; k++) cand[k] -= d * L[k + (size_t)i * nu];
            }
        }
        double n2 = 0;
        for (int k = 0; k < nu; k++) n2 += cand[k] * cand[k];
        if (n2 < 1e-10) continue;   // e was (numerically) in the span already
        const double inv = 1 / std::sqrt(n2);
        for (int k = 0; k < nu; k++) L[k + (size_t)found * nu] = cand[k] * inv;
        found++;
    }
    return rank;
}
