// CDNA4 (gfx950) forward step, split variant: one rollout is run by K workgroups (dc_cluster.h), part p owning the vertex rows
// [p R, (p + 1) R). Same algorithm as dc_forward_pk.hip (Simulation::step, Simulation.cpp:1043-1428; global solve :1267 replaced
// by Jacobi-PCG on the symmetrically scaled packet matrix), same per-part data placement (search direction in LDS, residual /
// A p / iterate in registers, element windows in LDS); what is new is what crosses the parts:
//   * PD level: the velocity iterate v is written and read write-through (sc1), its hand-over flag is the exchange that carries
//     the convergence norm; per time step the tape state changes hands under an agent-scope release / acquire pair;
//   * PCG level, ONE exchange per iteration (round 6, dc_forward_cl_kernel.h): [p.Ap, p.r, r.Ap, Ap.Ap, r.r partials + the HB boundary rows of A p];
//     the next residual's norm is r.r - 2 alpha r.Ap + alpha^2 Ap.Ap, and every part keeps the neighbours' boundary rows of the residual AND of
//     the search direction (stored as scaled halves) in its LDS and updates them itself (r_halo -= alpha (A p)_halo, p_halo = r_halo + beta p_halo),
//     so neither travels. (DC_SXCG=0: the two-exchange fp32 loop of rounds 2-5 — [p.Ap] and [r.r + boundary rows of the new residual].)
//   * self contacts couple arbitrary vertices: detection + layering run on part 0; the layered friction pass of an iteration is evaluated by every
//     part in its own LDS when the parts share an XCD (the contact lists cross through L1-bypassing loads), else on part 0 between two barriers.
// All parts take identical control-flow decisions: every scalar that steers a loop is a sum over the parts in part order.
#define DC_KERNEL_TU
#include "dc_forward_cl_kernel.h"

namespace dc {

hipError_t launch_pd_step_cluster_deflated(const DevSystem &S, const DevCluster &CL, const DevWork &W, const FwdArgs &A, int b0, int nb, hipStream_t st);      // dc_forward_cl_defl.hip

// nb rollouts starting at b0, K workgroups each; the caller has zeroed the exchange area and made sure K nb <= CUs. The grid is
// rounded up to a multiple of 8 rollouts: with the observed round-robin placement (block b on XCD b % 8) the K parts of a rollout
// then land on ONE XCD whatever nb is (cluster_map), which lets their exchanges stay in that XCD's L2; the padding workgroups
// exit at once. Correctness does not depend on the placement (xch_hello checks it at run time).
hipError_t launch_pd_step_cluster(const DevSystem &S, const DevCluster &CL, const DevWork &W, const FwdArgs &A, int b0, int nb, hipStream_t st) {
  // Single-exchange CG (dc_forward_cl_kernel.h, PIPE = true): ONE exchange per CG iteration instead of two — standard CG whose r.r of the next
  // residual comes from r.r - 2 alpha r.Ap + alpha^2 Ap.Ap (all four sums and the boundary rows of A p in the one exchange). Default since
  // round 6; DC_SXCG=0 selects the two-exchange loop (A/B runs). History: round 4's pipelined CG (Ghysels & Vanroose) also had one exchange
  // per iteration but carried A r, A p, A s by vector recurrences that drift in fp32 (7e-5 on positions at N = 16 384, docs/HISTORY.md);
  // here no vector is recurred.
  if (S.defl_u && S.fwd_defl) return launch_pd_step_cluster_deflated(S, CL, W, A, b0, nb, st);
  static const bool sx = !(getenv("DC_SXCG") && getenv("DC_SXCG")[0] == '0');
#define DC_CL_CASE(V) case V: if (sx) return A.inline_detect ? launch_cl_inst<V, true, true>(S, CL, W, A, b0, nb, st) : launch_cl_inst<V, false, true>(S, CL, W, A, b0, nb, st); \
                              return A.inline_detect ? launch_cl_inst<V, true, false>(S, CL, W, A, b0, nb, st) : launch_cl_inst<V, false, false>(S, CL, W, A, b0, nb, st);
  switch (CL.pk_vpt) {
    DC_CL_CASE(1) DC_CL_CASE(2) DC_CL_CASE(3) DC_CL_CASE(4) DC_CL_CASE(6) DC_CL_CASE(8) DC_CL_CASE(12)
    default: return hipErrorInvalidValue;
  }
#undef DC_CL_CASE
}

}  // namespace dc
