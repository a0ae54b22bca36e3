// CDNA4 (gfx950) self-collision detection + contact layering, one workgroup per rollout, once per time step.
//
// Reference: Simulation::collisionDetection (Simulation.cpp:225-373, self part :281-352), isSelfCollision (:194-220),
// contactSorting (:422-624). Runs BEFORE the step kernel of the same slot and leaves, per rollout, the self
// contacts grouped layer by layer (no vertex twice in a layer) in the record's SelfRec.
//
//   1. bounding box (the reference seeds it with particles[0].pos == s_n[0], kept), longest axis, <= 512 cells
//   2. counting sort of the vertices by cell (LDS histogram), positions copied in sorted order
//   3. every vertex scans the following sorted vertices up to cell + sweepCellRadius + 2 (the reference's window),
//      |x_i - x_j| <= 1, swept distance test with the reference's tMid (factor 2 kept), connected pairs excluded
//   4. pairs sorted by (id1, id2) -> deterministic, independent of atomic order (the reference's order depends on
//      OpenMP timing; only the order inside a layer could differ, which does not change r)
//   5. greedy layering, executed by ONE thread exactly as the reference's std::map / std::set code does
#define DC_KERNEL_TU
#include "dc_devlib.h"
#include "dc_selflib.h"

namespace dc {

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_self_detect(const DevSystem *__restrict__ Sp, DevWork W, FwdArgs A) {
  __shared__ int lds[kSelfDetectLdsInts];
  self_detect_rollout<THREADS>(*Sp, W, blockIdx.x, A.x_in, A.v_in, A.rec_prim, A.self, A.fu, A.fv, A.fv_scale, lds, A.fv2);
}

static int pick_threads_sd(int N) { return N <= 1536 ? 256 : (N <= 6144 ? 512 : 1024); }

void launch_self_detect(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  switch (pick_threads_sd(S.N)) {
    case 256: hipLaunchKernelGGL(k_self_detect<256>, dim3(B), dim3(256), 0, st, S.self_dev, W, A); break;
    case 512: hipLaunchKernelGGL(k_self_detect<512>, dim3(B), dim3(512), 0, st, S.self_dev, W, A); break;
    default: hipLaunchKernelGGL(k_self_detect<1024>, dim3(B), dim3(1024), 0, st, S.self_dev, W, A); break;
  }
}

}  // namespace dc
