// NullSpace.hpp -- Engine::calculateMatLandMatLhat (reference src/Engine.cu:466-669) on the host, in fp64:
//   L    = basis of null(E)  = the trailing nu-ne left singular vectors of E'      (Engine.cu:541-553, 611-615)
//   Lhat = -pinv(E) Ed       = -U(:,1:ne) S^-1 V' Ed                                (Engine.cu:566-586)
// The reference calls cusolverDnDgesvd; here the SVD of the (small, nu x ne) matrix E' is a one-sided Jacobi sweep.
// The null-space basis is unique only up to an orthogonal transform: x, u, Hx, z, the residual and the duals do not
// depend on the choice, v does.
#ifndef RAPIDNET_NULLSPACE_HPP_
#define RAPIDNET_NULLSPACE_HPP_

#include <vector>

// E: ne x nu, Ed: ne x nd (column-major).  Outputs (column-major): L nu x (nu-ne), Lhat nu x nd.
// Returns the numerical rank of E (ne for a well-posed network).
int computeNullSpaceAndParticular(const double *E, const double *Ed, int ne, int nu, int nd, std::vector<double> &L,
                                  std::vector<double> &Lhat);

#endif
