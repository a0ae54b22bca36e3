// CDNA4 (gfx950) forward step, packet-ELL resident variant with spectral deflation: the instances of k_pd_step_pk (dc_forward_pk_kernel.h)
// whose PCG solves start with the Galerkin projection onto the 16 lowest eigenvectors of the scaled system matrix (dc_deflate.h) — the
// engine builds that space only for meshes on which plain Jacobi-PCG needs hundreds of iterations (the reference's 7 742-vertex dress: 262 per
// PD iteration). Reference: the global solve of Simulation::step, Simulation.cpp:1267.
#define DC_KERNEL_TU
#include "dc_forward_pk_kernel.h"

namespace dc {

template <int VPT, int XL, bool H16 = false>
static void launch_deflated(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  if (A.inline_detect) launch_pk_inst<512, VPT, XL, true, false, H16, true>(S, W, A, B, st);
  else launch_pk_inst<512, VPT, XL, false, false, H16, true>(S, W, A, B, st);
}

bool launch_pd_step_packet_deflated(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  if (!S.pk_ok || !S.defl_u || S.pk_threads != 512) return false;
  static const int h16 = getenv("DC_PK_H16") ? atoi(getenv("DC_PK_H16")) : 1;
  switch (S.pk_vpt) {
    case 4: launch_deflated<4, 0>(S, W, A, B, st); break;
    case 6: launch_deflated<6, 0>(S, W, A, B, st); break;
    case 8: launch_deflated<8, 0>(S, W, A, B, st); break;
    case 10: launch_deflated<10, 0>(S, W, A, B, st); break;
    case 12: launch_deflated<12, 0>(S, W, A, B, st); break;
    case 16: launch_deflated<16, 2>(S, W, A, B, st); break;
    case 20:
      if (h16 && S.win_ok) launch_deflated<20, 12, true>(S, W, A, B, st);
      else launch_deflated<20, 6>(S, W, A, B, st);
      break;
    default: return false;
  }
  return true;
}

}  // namespace dc
