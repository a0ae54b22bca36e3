// CDNA4 (gfx950) forward step, split variant with spectral deflation: the instances of k_pd_step_cl (dc_forward_cl_kernel.h) whose PCG solves
// start with the Galerkin projection onto the 16 lowest eigenvectors of the scaled system matrix (dc_deflate.h; built by the engine for meshes on
// which plain Jacobi-PCG needs hundreds of iterations). Reference: the global solve of Simulation::step, Simulation.cpp:1267.
#define DC_KERNEL_TU
#include "dc_forward_cl_kernel.h"

namespace dc {

hipError_t launch_pd_step_cluster_deflated(const DevSystem &S, const DevCluster &CL, const DevWork &W, const FwdArgs &A, int b0, int nb, hipStream_t st) {
  if (!S.defl_u) return hipErrorInvalidValue;
#define DC_CL_CASE(V) case V: return A.inline_detect ? launch_cl_inst<V, true, false, true>(S, CL, W, A, b0, nb, st) : launch_cl_inst<V, false, false, true>(S, CL, W, A, b0, nb, st);
  switch (CL.pk_vpt) {
    DC_CL_CASE(1) DC_CL_CASE(2) DC_CL_CASE(3) DC_CL_CASE(4) DC_CL_CASE(6) DC_CL_CASE(8) DC_CL_CASE(12)
    default: return hipErrorInvalidValue;
  }
#undef DC_CL_CASE
}

}  // namespace dc
