// CDNA4 (gfx950) forward step, packet-ELL resident variant (N <= 10240, matrix bandwidth <= 511).
//
// Same algorithm and per-rollout workgroup ownership as dc_forward.hip / dc_forward_res.hip; the inner Jacobi PCG
// (>95 % of the sweeps of a step) is restructured around what bounds it on a CU, the 64 B/clk vector-memory pipe
// that streams the shared matrix, and the register file:
//   * the PCG runs on the symmetrically scaled system  (D^-1/2 P D^-1/2) xs = D^-1/2 rhs  (unit diagonal): plain CG
//     on it IS the Jacobi-preconditioned CG on P (same iterates, same r^T D^-1 r stopping rule) but needs no
//     diagonal / preconditioner traffic inside the loop;
//   * off-diagonals travel as 16-byte packets {v0, v1, v2, d0 | d1 << 10 | d2 << 20}: three fp32 values and three
//     10-bit column deltas relative to the row (biased by 512) = 5.33 B per non-zero instead of 8, one
//     global_load_dwordx4 per three non-zeros, wave-sliced so a wave reads 1 KiB contiguous per instruction;
//     the next row's packets are in flight while the current row is consumed;
//   * the search direction p lives in LDS as a float2 (x, y) plane + a float z plane (a neighbour costs one
//     ds_read_b64 + one ds_read_b32), the residual r and A p in registers, the iterate x in registers for the
//     first VPT - XL rows of a thread and in the LDS left over by p for the rest: nothing of a CG iteration
//     touches global memory except the (L2-resident, batch-shared) packet stream;
//   * the per-constraint work of a PD iteration (local projections, friction, right-hand side) runs through the LDS
//     element windows of dc_winlib.h, and a launch can carry all time steps of a rollout (FwdArgs::nsteps) with the
//     self-collision detection of each step inlined (dc_selflib.h).
// Reference: Simulation::step (Simulation.cpp:1043-1428), global solve :1267 (SimplicialLLT::solve) replaced by
// this PCG on the correction system (see dc_forward.hip header).
#define DC_KERNEL_TU
#include "dc_forward_pk_kernel.h"

namespace dc {

bool launch_pd_step_packet_deflated(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st);      // dc_forward_pk_defl.hip

// 512 (or 768) threads own VPT = pk_vpt rows each (the packet tables are built for exactly that padding, dc_engine.hip).
bool launch_pd_step_packet(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  if (!S.pk_ok) return false;
  if (S.defl_u && S.fwd_defl) return launch_pd_step_packet_deflated(S, W, A, B, st);
  static const int h16 = getenv("DC_PK_H16") ? atoi(getenv("DC_PK_H16")) : 1;      // (development switch: 0 = the fp32 direction planes)
  if (S.pk_threads == 768) {
    if (S.pk_vpt != 14) return false;
    if (h16 && S.win_ok) launch_pk_h16<768, 14, 7>(S, W, A, B, st);
    else launch_pk<768, 14, 3>(S, W, A, B, st);
    return true;
  }
#ifdef DC_PK_ONLY20      // development builds: only the 10 000-vertex variant (compile time)
  if (S.pk_vpt != 20) return false;
  if (h16 && S.win_ok) launch_pk_h16<512, 20, 12>(S, W, A, B, st);
  else launch_pk<512, 20, 6>(S, W, A, B, st);
  return true;
#else
  switch (S.pk_vpt) {
    case 1: launch_pk<512, 1, 0>(S, W, A, B, st); break;
    case 2: launch_pk<512, 2, 0>(S, W, A, B, st); break;
    case 3: launch_pk<512, 3, 0>(S, W, A, B, st); break;
    case 4: launch_pk<512, 4, 0>(S, W, A, B, st); break;
    case 6: launch_pk<512, 6, 0>(S, W, A, B, st); break;
    case 8: launch_pk<512, 8, 0>(S, W, A, B, st); break;
    case 10: launch_pk<512, 10, 0>(S, W, A, B, st); break;
    case 12: launch_pk<512, 12, 0>(S, W, A, B, st); break;
    case 16: launch_pk<512, 16, 2>(S, W, A, B, st); break;
    case 20:
      if (h16 && S.win_ok) launch_pk_h16<512, 20, 12>(S, W, A, B, st);
      else launch_pk<512, 20, 6>(S, W, A, B, st);
      break;
    default: return false;
  }
  return true;
#endif
}

}  // namespace dc
