// Host-side builder of the spectral deflation space of the forward solve (irregular garments).
//
// Jacobi-PCG on P = M + h^2 A^T A needs 15 ... 25 iterations per PD iteration on the cloth and demo meshes, but 260 ... 340 on the
// reference's fine dress (dress-v7k-f14k.obj): a few smooth, almost mass-only modes of the hanging garment have eigenvalues 1e-4 ... 1e-2
// of the scaled matrix's largest (cond 5e4) and the right-hand sides of a PD iteration are dominated by exactly those modes. The reference
// does not see this (it applies a prefactored Cholesky, Simulation.cpp:1267). The matrix is constant and shared by all rollouts, so its
// k lowest eigenvectors U are computed ONCE here (Chebyshev-filtered subspace iteration on Ahat = D^-1/2 P D^-1/2, the matrix of the packet
// kernel), and every solve starts with a Galerkin projection onto them,
//     c = (U^T Ahat U)^-1 U^T r,   x += U c,   r -= (Ahat U) c,
// after which the Krylov space CG builds stays orthogonal to U (an invariant subspace, up to the 1e-4 the vectors are computed to — the
// projection uses Ahat U and the Gram matrix, not the eigenvalues, so approximate vectors cost nothing but a little of the effect).
// Measured offline on that mesh (scipy, smooth right-hand sides): 336 iterations -> 69 with k = 16, 63 with k = 32; strip aggregates,
// block-Jacobi and polynomial coarse spaces do not get below 130.
#pragma once
#include <vector>
#include "dc_system.h"

namespace dc {

struct HostDeflation {
  bool ok = false;
  int k = 0;                      // vectors
  int rows = 0;                   // padded row count of the tables (the packet kernel's)
  int probe_iterations = 0;       // Jacobi-CG iterations of the probe solve that decided (smooth right-hand side, 1e-4)
  std::vector<float> U;           // [rows][k] row-major, scaled system (u_hat = D^1/2 u), orthonormal columns; zero rows for padding
  std::vector<float> AU;          // [rows][k] Ahat U
  std::vector<float> G;           // [k][k]   (U^T Ahat U)^-1
  std::vector<double> ritz;       // the k Ritz values (diagnostics)

  // want < 0: decide by the probe solve (more than `auto_threshold` iterations -> 16 vectors); 0: never; k > 0: k vectors (<= 32).
  // rows_padded: row count of the device tables (>= N). Returns true when a space was built.
  bool build(const HostSystem &H, int want, int rows_padded, int auto_threshold = 80);
};

}  // namespace dc
