// Host-side builder of the explicit inverse used by the small-mesh kernels (dc_denselib.h).
//
// For meshes of up to ~1000 vertices (the hat and sock scenes of the reference) the scalar system matrix P is small enough
// that the inverse of its scaled form  Ahat = D^-1/2 P D^-1/2  (N x N fp32, 1.3 MB at N = 579) stays in the L2 of every XCD,
// shared by all rollouts of a batch. One product with it replaces the ~30-90 Jacobi-PCG iterations of a global step; what the
// fp32 rounding of the inverse leaves (relative residual ~1e-5 at condition number 1e3) is removed by iterative refinement with
// the packet SpMV, so the stopping rule of the solve (relative residual <= cg_rel_tol) is unchanged. The reference itself
// applies a prefactored Cholesky of P at this point (Simulation.cpp:1267, factorizeDirectSolverLLT :4514-4534).
#pragma once
#include <vector>
#include "dc_system.h"

namespace dc {

struct HostDense {
  bool ok = false;
  int n = 0;
  int ld = 0;                 // leading dimension: n rounded up to 64 (one wave reads 64 consecutive entries of a row)
  int rows = 0;               // n + kDensePadRows zero rows: the kernels read whole groups of rows without bounds checks
  double defect = 0;          // max over probe vectors of |v - Ahat (inv32 v)| / |v|: the contraction of one refinement step
  std::vector<float> inv;     // [rows][ld] row-major (symmetric: row j doubles as column j)

  // false when n > max_n, the Cholesky factorisation breaks down or the rounded inverse is not a contraction (defect >= 1e-2)
  bool build(const HostSystem &H, int max_n);
};

constexpr int kDensePadRows = 128;

}  // namespace dc
