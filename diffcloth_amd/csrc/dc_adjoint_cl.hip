// CDNA4 (gfx950) adjoint step, split variant: one rollout is run by K workgroups (dc_cluster.h), part p owning the vertex rows
// [p R, (p + 1) R). Same algorithm as dc_adjoint.hip, direct solve (adjoint_mode 1: block-Jacobi preconditioned BiCGSTAB on
// K = M + h^2 (A - dp/dx)^T A (I + dr_df)^T, the semantics of Simulation::solveDirect, Simulation.cpp:1431-1440, inside
// Simulation::stepBackward, :1455-1780). The Krylov vectors stay in global memory and every part touches only its own rows of
// them; what crosses the parts:
//   * the input of an operator application over the reach of the element windows: its HB boundary rows travel as granules when
//     the vector is written and land in a small LDS cache the window staging reads;
//   * the partial sums of every dot product (five exchanges per BiCGSTAB iteration);
//   * with self contacts (which couple arbitrary vertices) y = (I + dr_df)^T z is formed in global memory instead: own rows by
//     every part, the layered transposed pass by part 0, fence barriers in between.
#define DC_KERNEL_TU
#include <cstdlib>
#include "dc_devlib.h"
#include "dc_winlib.h"
#include "dc_cluster.h"
#include "dc_adjprecond.h"
#include <algorithm>

namespace dc {

constexpr int kCycles = 1;      // BiCGSTAB cycles of the direct adjoint solve (see the comment at the loop)

namespace {

struct AdjCl {
  const float *xnew, *rec_f, *rec_n, *mu;
  const int *rec_prim;
  BufVec yb;                    // y = (I + dr_df)^T z of the rollout, read across parts: write-through stores, L1-bypassing loads
  float *lds, *hc;              // element-window LDS; halo cache: 3 planes of 2 HB floats
  int lds_floats;
  SelfRec self;
  int nself, b, part, r0, r1, R, HB, w0, w1;
};

// w = dr_df^T z for the (block-diagonal) primitive contacts: Simulation::calculatedr_df (Simulation.cpp:700-711)
__device__ __forceinline__ f3 contact_JT_cl(const DevSystem &S, const AdjCl &C, int i, f3 z) {
  const int prim = C.rec_prim[i];
  if (prim < 0) return mk(0, 0, 0);
  const int N = S.N;
  f3 n = ld3(C.rec_n, i, N);
  f3 d = ld3(C.rec_f, i, N) - prim_vout(S.prims[prim], n) * S.mass[i];
  return dri_dfi_T(n, d, C.mu[S.prims[prim].group], z);
}

// store the halo rows an exchange delivered into the LDS cache (all threads; followed by a barrier at the caller)
template <int THREADS, int HPT>
__device__ __forceinline__ void halo_to_cache(const AdjCl &C, const f3 (&hv)[HPT]) {
#pragma unroll
  for (int q = 0; q < HPT; q++) {
    const int j = threadIdx.x + q * THREADS;
    if (j < 2 * C.HB) { C.hc[j] = hv[q].x; C.hc[2 * C.HB + j] = hv[q].y; C.hc[4 * C.HB + j] = hv[q].z; }
  }
}

// out = K z on the own rows, z = zin (optionally scaled by D^-1: right preconditioning); the boundary rows of zin are in the
// halo cache. Also returns this thread's partial sums of out.d1 and out.out (d1 may be null). Ends WITHOUT a barrier.
// Returns false when an exchange timed out.
template <int THREADS>
__device__ __forceinline__ bool adjoint_operator_cl(const DevSystem &S, const DevCluster &CL, const AdjCl &C, Xch &X, const float *zin,
                                                    bool precond, float *out, const float *d1, float &dot1, float &dot2) {
  const int N = S.N;
  const float h2 = S.h * S.h;
  float a1 = 0.f, a2 = 0.f;
  auto vert = [&](int i, f3 sum, f3 yi) {
    f3 z = ld3(zin, i, N);
    if (precond) z = z * S.dinv[i];
    f3 o = z * S.mass[i] + sum;
    if (S.att_of_vertex[i] >= 0) o = o + yi * (h2 * S.k_att);   // attachment: dp/dx = 0 (AttachmentSpring.cpp:35-37)
    st3(out, i, N, o);
    if (d1) a1 += dot(o, ld3(d1, i, N));
    a2 += dot(o, o);
  };
  if (C.nself > 0) {
    // layered self contacts couple vertices of different parts: y = (I + dr_df)^T z goes through global memory
    for (int i = C.r0 + threadIdx.x; i < C.r1; i += THREADS) {
      f3 z = ld3(zin, i, N);
      if (precond) z = z * S.dinv[i];
      st3c(C.yb, i, z);
    }
    if (!xch_barrier<THREADS>(X)) return false;
    if (C.part == 0) {
      if (!self_JT_layers_lds_v<THREADS>(S, C.self, C.b, C.yb, C.lds, C.lds_floats)) self_JT_layers_v<THREADS>(S, C.self, C.b, C.yb);
    }
    if (!xch_barrier<THREADS>(X)) return false;
    element_windows_t<THREADS>(CL, C.w0, C.w1, C.lds, [&](int i) {
      f3 z = ld3c(C.yb, i);
      return z + contact_JT_cl(S, C, i, z);
    }, In2Plain{C.xnew, N}, AdjTriOp{h2}, AdjBendOp{h2}, vert);
  } else {
    // primitive contacts only: dr_df is block diagonal, y_i is formed per vertex while the window is staged
    const int lo_own = C.r0, hi_own = C.r0 + C.R, HB = C.HB;
    element_windows_t<THREADS>(CL, C.w0, C.w1, C.lds, [&](int i) {
      f3 z;
      if (i >= lo_own && i < hi_own) z = ld3(zin, i, N);
      else { const int j = i < lo_own ? i - (lo_own - HB) : HB + (i - hi_own); z = mk(C.hc[j], C.hc[2 * HB + j], C.hc[4 * HB + j]); }
      if (precond) z = z * S.dinv[i];
      return z + contact_JT_cl(S, C, i, z);
    }, In2Plain{C.xnew, N}, AdjTriOp{h2}, AdjBendOp{h2}, vert);
  }
  dot1 = a1; dot2 = a2;
  return true;
}

}  // namespace

template <int THREADS, bool BLK>
__global__ __launch_bounds__(THREADS) void k_adjoint_step_cl(const DevSystem *__restrict__ Sp, const DevCluster *__restrict__ Cp, DevWork W,
                                                             BwdArgs A, int b0, int nb_real, int hc_off, int tail_off) {
  const DevSystem &S = *Sp;
  const DevCluster &CL = *Cp;
  constexpr int HPT = (1024 + THREADS - 1) / THREADS;
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  const int tid = threadIdx.x;
  const int N = S.N, K = CL.K, R = CL.R, HB = CL.HB;
  int lb, part;
  cluster_map(K, lb, part);
  if (lb >= nb_real) return;       // padding workgroups: the launch is rounded up to a multiple of 8 rollouts (see the launcher)
  const int b = b0 + lb;
  Xch X = xch_init(CL, lb, part, dyn_lds + tail_off);
  if (!xch_hello<THREADS>(X)) return;
  const int r0 = part * R, r1 = min(N, r0 + R);
  const size_t off = (size_t) b * 3 * N;
  double sums[3];
  f3 hv[HPT];
  f3 none[1];

  for (int step = 0; step < A.nsteps; step++) {
  if (step > 0) {
    __syncthreads();
    A.x_new -= A.slot_state; A.rec_f -= A.slot_state; A.rec_n -= A.slot_state; A.rec_prim -= A.slot_prim;
    A.x_prev -= A.slot_state; A.v_prev -= A.slot_state;
    A.self.pair -= A.slot_self; A.self.nrm -= A.slot_self; A.self.dvec -= A.slot_self; A.self.meta -= A.slot_meta; A.self.verts -= 2 * A.slot_self;
    if (A.d_param) A.d_param -= A.slot_param;
    A.x_fixed -= A.slot_xf; A.stats -= A.slot_stats;
    if (A.d_xfixed) A.d_xfixed -= A.slot_xf;
    if (A.ix) A.ix -= A.slot_ix;
    if (A.iv) A.iv -= A.slot_ix;
    A.is_start = (A.slot - step == 1) ? 1 : 0;            // isStart: Simulation.cpp:3947
  }
  AdjCl C;
  C.lds = dyn_lds; C.lds_floats = hc_off; C.hc = dyn_lds + hc_off;
  C.xnew = A.x_new + off; C.rec_f = A.rec_f + off; C.rec_n = A.rec_n + off;
  C.rec_prim = A.rec_prim + (size_t) b * N;
  C.mu = A.mu + (size_t) b * S.ngroups;
  C.yb = buf_vec(W.vbest + off, N, X.same_xcd);
  C.self = A.self; C.b = b; C.part = part; C.r0 = r0; C.r1 = r1; C.R = R; C.HB = HB;
  C.w0 = part * CL.wpp; C.w1 = min(CL.nwin, C.w0 + CL.wpp);
  C.nself = (S.contact_enabled && S.self_enabled) ? A.self.meta[(size_t) b * kMetaStride] : 0;
  float *gx = A.gx + off, *gv = A.gv + off;
  float *gin = W.g + off, *u = W.vnow + off;
  float *r = W.cg_r + off, *p = W.cg_p + off, *v = W.cg_ap + off, *t = W.cg_x + off, *rhat = W.sd_sx + off;   // (detection scratch, idle here)
  float *ph = W.pre_p + off, *sh = W.pre_s + off, *minv = W.minv + (size_t) b * 9 * N;   // M^-1 p, M^-1 s, the block inverses
  if constexpr (BLK) {   // K's own 3 x 3 diagonal blocks at this step's x_new, inverted (dc_adjprecond.h); own rows
    for (int i = r0 + tid; i < r1; i += THREADS)
      store_block_inverse(elastic_diag_block(S, C.xnew, i), S.mass[i], [&](f3 e) { return contact_JT_cl(S, C, i, e); }, minv, i, N);
  }
  // BLK: M^-1 = those block inverses, the vectors M^-1 p / M^-1 s are stored and are what the operator is applied to;
  // otherwise M^-1 = diag(P)^-1, applied inside the operator
  auto pre = [&](int i, f3 z) { return BLK ? block_pre(minv, i, N, z) : z; };
  const float h = S.h, h2 = S.h * S.h;

  // ---- gradient clipping (Simulation.cpp:1460-1466), u = 0, r = rhat = p = g ----
  float part_s = 0.f;
  for (int i = r0 + tid; i < r1; i += THREADS) { f3 q = ld3(gx, i, N); part_s += dot(q, q); }
  if (!xch_allsum<THREADS>(X, part_s, 0.f, 0.f, sums)) return;
  double gnorm = sqrt(sums[0]);
  float gscale = 1.f;
  int clipped = 0;
  if (A.clip && gnorm > (double) A.clip_thr * N) { gscale = (float) ((double) A.clip_thr * N / gnorm); clipped = 1; gnorm = (double) A.clip_thr * N; }
  int status = 0;          // 1 converged, 2 stalled at the fp32 floor / breakdown, 0 cap hit
  int iters = 0;
  double udiff = 0;
  for (int i = r0 + tid; i < r1; i += THREADS) { st3(gin, i, N, ld3(gx, i, N) * gscale); st3(u, i, N, mk(0, 0, 0)); }
  double rr = 0;
  if (gnorm > 0) {
    const double stop = (double) A.rel_tol * (double) A.rel_tol * gnorm * gnorm;
    const int kcap = A.it_cap > 0 ? 4 * A.it_cap : 1600;
    constexpr int VB = 4;
    // kCycles > 1: when the recurrence residual says "converged", recompute g - K u and restart from u if that is not below the
    // tolerance. Measured on the C4 workload (r02m): +4 iterations of 41, gradient error against the fp64 oracle unchanged to three
    // digits (7.31e-5 -> 7.31e-5) — the fp32 floor of this solve is eps * cond(K) in the operator's coefficients, not residual
    // drift — so one cycle is the default.
    for (int cycle = 0, kdone = 0; cycle < kCycles; cycle++) {
    // r = rhat = p = g - K u (u = 0 in the first cycle: K u = 0 without applying the operator)
    if (cycle > 0) {
      // u travels as the operator's input: its boundary rows first
      xch_begin(X);
      for (int l = tid; l < R; l += THREADS) {
        const int i = r0 + l;
        f3 q = i < N ? ld3(u, i, N) : mk(0, 0, 0);
        xch_publish_boundary(X, l, R, q.x, q.y, q.z);
      }
      xch_publish_sums(X, 0.f, 0.f, 0.f);
      if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
      halo_to_cache<THREADS, HPT>(C, hv);
      __syncthreads();
      float e1, e2;
      if (!adjoint_operator_cl<THREADS>(S, CL, C, X, u, false, v, nullptr, e1, e2)) return;
      __syncthreads();
    }
    xch_begin(X);
    part_s = 0.f;
    for (int l = tid; l < R; l += THREADS) {
      const int i = r0 + l;
      f3 q = mk(0, 0, 0);
      if (i < N) {
        q = ld3(gin, i, N);
        if (cycle > 0) q = q - ld3(v, i, N);
        st3(r, i, N, q); st3(rhat, i, N, q); st3(p, i, N, q);
        part_s += dot(q, q);
        if constexpr (BLK) { q = pre(i, q); st3(ph, i, N, q); }
      }
      xch_publish_boundary(X, l, R, q.x, q.y, q.z);      // the operator's input: M^-1 p (BLK), or p itself, scaled by diag(P)^-1 inside the operator
    }
    xch_publish_sums(X, part_s, 0.f, 0.f);
    if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
    halo_to_cache<THREADS, HPT>(C, hv);
    __syncthreads();
    double rho = sums[0];
    rr = rho;
    double best_rr = rr;
    int since_progress = 0;
    status = (rr <= stop) ? 1 : 0;
    if (status == 0 && cycle > 0 && cycle == kCycles - 1) status = 2;      // still above the tolerance after two restarts: the fp32 floor of this system
    if (status != 0) break;
    for (int k = kdone; k < kcap && status == 0; k++, kdone++) {
      float d1, d2;
      // v = K M^-1 p ;  alpha = rho / (rhat . v)
      if (!adjoint_operator_cl<THREADS>(S, CL, C, X, BLK ? ph : p, !BLK, v, rhat, d1, d2)) return;
      if (!xch_allsum<THREADS>(X, d1, 0.f, 0.f, sums)) return;
      const double rv = sums[0];
      if (!(fabs(rv) > 1e-300)) { status = 2; break; }
      const float alpha = (float) (rho / rv);
      // s = r - alpha v  (in place), its boundary rows to the neighbours
      xch_begin(X);
      part_s = 0.f;
      for (int l0 = tid; l0 < R; l0 += VB * THREADS) {
        f3 rq[VB], vq[VB];
#pragma unroll
        for (int j = 0; j < VB; j++) { const int ic = min(r0 + l0 + j * THREADS, r1 - 1); rq[j] = ld3(r, ic, N); vq[j] = ld3(v, ic, N); }
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int l = l0 + j * THREADS, i = r0 + l;
          f3 s = rq[j] - vq[j] * alpha;
          if (i >= r1) s = mk(0, 0, 0);
          if (i < r1) { st3(r, i, N, s); part_s += dot(s, s); if constexpr (BLK) { s = pre(i, s); st3(sh, i, N, s); } }
          if (l < R) xch_publish_boundary(X, l, R, s.x, s.y, s.z);      // boundary rows of M^-1 s
        }
      }
      {
        xch_publish_sums(X, part_s, 0.f, 0.f);
        if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
        halo_to_cache<THREADS, HPT>(C, hv);
        __syncthreads();
      }
      const double ss = sums[0];
      iters++;
      if (ss <= stop) {
        for (int i = r0 + tid; i < r1; i += THREADS) st3(u, i, N, ld3(u, i, N) + (BLK ? ld3(ph, i, N) * alpha : ld3(p, i, N) * (alpha * S.dinv[i])));
        rr = ss; status = 1; break;
      }
      // t = K M^-1 s ;  omega = (t . s) / (t . t)
      if (!adjoint_operator_cl<THREADS>(S, CL, C, X, BLK ? sh : r, !BLK, t, r, d1, d2)) return;
      if (!xch_allsum<THREADS>(X, d1, d2, 0.f, sums)) return;
      const double ts = sums[0], tt = sums[1];
      if (!(tt > 1e-300)) { status = 2; break; }
      const float omega = (float) (ts / tt);
      // u += alpha D^-1 p + omega D^-1 s ;  r = s - omega t ;  rho_new = rhat . r
      float pa = 0.f, pb = 0.f;
      for (int l0 = tid; l0 < R; l0 += VB * THREADS) {
        f3 sq[VB], uq[VB], pq[VB], tq[VB], hq[VB], zq[VB];
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int ic = min(r0 + l0 + j * THREADS, r1 - 1);
          sq[j] = ld3(r, ic, N); uq[j] = ld3(u, ic, N); tq[j] = ld3(t, ic, N); hq[j] = ld3(rhat, ic, N);
          if constexpr (BLK) { zq[j] = ld3(sh, ic, N); pq[j] = ld3(ph, ic, N); }
          else { const float di = S.dinv[ic]; zq[j] = sq[j] * di; pq[j] = ld3(p, ic, N) * di; }
        }
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int i = r0 + l0 + j * THREADS;
          f3 rn = sq[j] - tq[j] * omega;
          if (i < r1) {
            st3(u, i, N, uq[j] + pq[j] * alpha + zq[j] * omega);
            st3(r, i, N, rn);
            pa += dot(rn, hq[j]);
            pb += dot(rn, rn);
          }
        }
      }
      if (!xch_allsum<THREADS>(X, pa, pb, 0.f, sums)) return;
      const double rho_new = sums[0];
      rr = sums[1];
      if (rr <= stop) { status = 1; break; }
      if (rr < best_rr) { best_rr = rr; since_progress = 0; }
      else if (++since_progress >= A.stall_window) { status = 2; break; }
      if (!(fabs(rho_new) > 1e-300) || !(fabs(omega) > 0.f)) { status = 2; break; }
      const float beta = (float) ((rho_new / rho) * ((double) alpha / (double) omega));
      rho = rho_new;
      // p = r + beta (p - omega v), its boundary rows to the neighbours
      xch_begin(X);
      for (int l0 = tid; l0 < R; l0 += VB * THREADS) {
        f3 rq[VB], pq[VB], vq[VB];
#pragma unroll
        for (int j = 0; j < VB; j++) { const int ic = min(r0 + l0 + j * THREADS, r1 - 1); rq[j] = ld3(r, ic, N); pq[j] = ld3(p, ic, N); vq[j] = ld3(v, ic, N); }
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int l = l0 + j * THREADS, i = r0 + l;
          f3 pn = rq[j] + (pq[j] - vq[j] * omega) * beta;
          if (i >= r1) pn = mk(0, 0, 0);
          if (i < r1) { st3(p, i, N, pn); if constexpr (BLK) { pn = pre(i, pn); st3(ph, i, N, pn); } }
          if (l < R) xch_publish_boundary(X, l, R, pn.x, pn.y, pn.z);      // boundary rows of M^-1 p
        }
      }
      xch_publish_sums(X, 0.f, 0.f, 0.f);
      if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
      halo_to_cache<THREADS, HPT>(C, hv);
      __syncthreads();
    }
    if (status != 1) break;       // cap, breakdown or stall: no further cycle
    __syncthreads();
    }   // cycle
  }
  // "stalled at the fp32 floor" (2) is only claimed near the tolerance: a breakdown or stall with the residual still more than 100 x
  // above it (an adjoint system beyond an fp32 Krylov solve, e.g. a strongly compressed fine garment) is reported as NOT converged
  if (status == 2 && rr > 1e4 * (double) A.rel_tol * (double) A.rel_tol * gnorm * gnorm) status = 0;
  udiff = sqrt(rr) / (gnorm > 0 ? gnorm : 1.0);     // relative residual (of the last recomputed or recurrence residual)
  __syncthreads();
  // ---- gradients w.r.t. the previous state and parameters (Simulation.cpp:1534, 1608-1650) ----
  // y = (I + dr_df)^T u* in global memory (own rows; with self contacts the layered pass on part 0)
  if (C.nself > 0) {
    for (int i = r0 + tid; i < r1; i += THREADS) st3c(C.yb, i, ld3(u, i, N));
    if (!xch_barrier<THREADS>(X)) return;
    if (part == 0) {
      if (!self_JT_layers_lds_v<THREADS>(S, C.self, b, C.yb, C.lds, C.lds_floats)) self_JT_layers_v<THREADS>(S, C.self, b, C.yb);
    }
    if (!xch_barrier<THREADS>(X)) return;
    for (int i = r0 + tid; i < r1; i += THREADS) { f3 z = ld3c(C.yb, i); st3c(C.yb, i, z + contact_JT_cl(S, C, i, z)); }
  } else {
    for (int i = r0 + tid; i < r1; i += THREADS) { f3 z = ld3(u, i, N); st3c(C.yb, i, z + contact_JT_cl(S, C, i, z)); }
  }
  float pacc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (A.d_param) {
    // the element sums read y at arbitrary vertices: y changes hands; elements are dealt to the parts in contiguous ranges
    if (!xch_barrier<THREADS>(X)) return;
    const int T = S.T, E = S.E;
    const float *xnew = C.xnew;
    const BufVec &yv = C.yb;
    const int t0 = (int) ((long long) T * part / K), t1 = (int) ((long long) T * (part + 1) / K);
    for (int tt = t0 + tid; tt < t1; tt += THREADS) {
      const int i0 = S.tri_v[tt], i1 = S.tri_v[T + tt], i2 = S.tri_v[2 * T + tt];
      const float4 D = S.tri_D[tt];
      f3 x0 = ld3(xnew, i0, N);
      f3 e0 = ld3(xnew, i1, N) - x0, e1 = ld3(xnew, i2, N) - x0;
      f3 f0 = e0 * D.x + e1 * D.z, f1 = e0 * D.y + e1 * D.w;
      Polar P = polar3x2(f0, f1);
      f3 g0 = (P.t0 - f0) * S.tri_w2[tt], g1 = (P.t1 - f1) * S.tri_w2[tt];
      f3 c1 = g0 * D.x + g1 * D.y, c2 = g0 * D.z + g1 * D.w;
      f3 q0 = ld3c(yv, i0);
      pacc[0] += dot(c1, ld3c(yv, i1) - q0) + dot(c2, ld3c(yv, i2) - q0);
    }
    const int e0i = (int) ((long long) E * part / K), e1i = (int) ((long long) E * (part + 1) / K);
    for (int e = e0i + tid; e < e1i; e += THREADS) {
      const int i0 = S.bend_v[e], i1 = S.bend_v[E + e], i2 = S.bend_v[2 * E + e], i3 = S.bend_v[3 * E + e];
      const float4 w = S.bend_w[e];
      const float2 nw = S.bend_nw[e];
      f3 x0 = ld3(xnew, i0, N);
      f3 ev = (ld3(xnew, i1, N) - x0) * w.y + (ld3(xnew, i2, N) - x0) * w.z + (ld3(xnew, i3, N) - x0) * w.w;
      f3 pp = mk(0, 0, 0);
      if (nw.x > 1e-6f) pp = normalized(ev) * nw.x;
      f3 q0 = ld3c(yv, i0);
      f3 ey = (ld3c(yv, i1) - q0) * w.y + (ld3c(yv, i2) - q0) * w.z + (ld3c(yv, i3) - q0) * w.w;
      pacc[1] += dot((pp - ev) * nw.y, ey);
    }
  }
  float dmu_part[kMaxPrims];
#pragma unroll
  for (int k = 0; k < kMaxPrims; k++) dmu_part[k] = 0.f;
  float *dxf = A.d_xfixed ? A.d_xfixed + (size_t) b * 3 * S.Af : nullptr;
  const f3 grav = mk(S.gx, S.gy, S.gz);
  for (int i = r0 + tid; i < r1; i += THREADS) {
    f3 ui = ld3(u, i, N);
    const float m = S.mass[i];
    f3 w = ld3c(C.yb, i) - ui;
    if (A.d_param) {
      f3 yi = ui + w;
      const int a = S.att_of_vertex[i];
      if (a >= 0) pacc[2] += S.k_att * dot(ld3(A.x_fixed + (size_t) b * 3 * S.Af, a, S.Af) - ld3(C.xnew, i, N), yi);
      const float ar = m / S.density;
      f3 xp = ld3(A.x_prev + off, i, N), vp = ld3(A.v_prev + off, i, N);
      pacc[3] += ar * (dot(ui, xp + vp * h + grav * h2 - ld3(C.xnew, i, N)) + h * dot(w, vp + grav * h));
      pacc[4] += h2 * yi.x; pacc[5] += h2 * yi.y; pacc[6] += h2 * yi.z;
    }
    const int prim = C.rec_prim[i];
    if (prim >= 0) {
      f3 n = ld3(C.rec_n, i, N);
      f3 d = ld3(C.rec_f, i, N) - prim_vout(S.prims[prim], n) * m;
      const int grp = S.prims[prim].group;
      const float contrib = dot(dri_dmu(n, d, C.mu[grp]), ui) * h;
#pragma unroll
      for (int k = 0; k < kMaxPrims; k++) dmu_part[k] += (k == grp) ? contrib : 0.f;
    }
    f3 dx = ui * m - ld3(gv, i, N) * (1.0f / h);
    f3 dv = (ui + w) * (h * m);
    if (A.ix) dx = dx + ld3(A.ix + off, i, N);
    if (A.iv) dv = dv + ld3(A.iv + off, i, N);
    if (!A.is_start) dx = dx + dv * (1.0f / h);
    st3(gx, i, N, dx);
    st3(gv, i, N, dv);
    const int a = S.att_of_vertex[i];
    if (a >= 0 && dxf) st3(dxf, a, S.Af, (ui + w) * (h2 * S.k_att));   // A_t_dp_dxfixed (Simulation.cpp:3035-3048)
  }
  // sums over the parts, three values per exchange
  if (A.d_mu) {
    for (int k0 = 0; k0 < S.ngroups; k0 += 3) {
      float val[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        val[c] = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxPrims; k++) val[c] += (k == k0 + c) ? dmu_part[k] : 0.f;
      }
      if (!xch_allsum<THREADS>(X, val[0], val[1], val[2], sums)) return;
      if (tid == 0 && part == 0)
        for (int c = 0; c < 3 && k0 + c < S.ngroups; c++) A.d_mu[(size_t) b * S.ngroups + k0 + c] += (float) sums[c];
    }
  }
  if (A.d_param) {
    float *dp = A.d_param + (size_t) b * 8;
    const float scale[9] = {S.k_stretch > 0.f ? h2 / S.k_stretch : 0.f, S.k_bend > 0.f ? h2 / S.k_bend : 0.f,
                            S.k_att > 0.f ? h2 / S.k_att : 0.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f};
    for (int k0 = 0; k0 < 7; k0 += 3) {
      if (!xch_allsum<THREADS>(X, pacc[k0], k0 + 1 < 7 ? pacc[min(k0 + 1, 6)] : 0.f, k0 + 2 < 7 ? pacc[min(k0 + 2, 6)] : 0.f, sums)) return;
      if (tid == 0 && part == 0)
        for (int c = 0; c < 3 && k0 + c < 7; c++) dp[k0 + c] = (float) (sums[c] * scale[k0 + c]);
    }
  }
  if (tid == 0 && part == 0) {
    dc_bwd_stats s;
    s.converged = status; s.adjoint_iters = iters; s.cg_iters = 0; s.clipped = clipped;
    s.used_direct = 1; s.last_udiff = (float) udiff;
    s.refine_cycles = 0; s.fp64_iters = 0;
    A.stats[b] = s;
  }
  (void) none;
  }   // step
}

// nb rollouts starting at b0, K workgroups each (adjoint_mode 1 only); exchange area zeroed by the caller, K nb <= CUs.
hipError_t launch_adjoint_step_cluster(const DevSystem &S, const DevCluster &CL, const DevWork &W, const BwdArgs &A, int b0, int nb, hipStream_t st) {
  constexpr int THREADS = 1024;
  const int hc_off = (CL.win_lds_bytes / 4 + 3) / 4 * 4;
  const int tail_off = hc_off + 6 * CL.HB;
  const size_t lds = sizeof(float) * (size_t) (tail_off + kXchLdsFloats);
  if (lds > 160 * 1024 - 256 || A.mode != 1) return hipErrorInvalidValue;
  if (A.block_pre) {
    hipError_t e = hipFuncSetAttribute((const void *) k_adjoint_step_cl<THREADS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_adjoint_step_cl<THREADS, true>), dim3((nb + 7) / 8 * 8 * CL.K), dim3(THREADS), lds, st, S.self_dev, CL.self_dev, W, A, b0, nb, hc_off, tail_off);
  } else {
    hipError_t e = hipFuncSetAttribute((const void *) k_adjoint_step_cl<THREADS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_adjoint_step_cl<THREADS, false>), dim3((nb + 7) / 8 * 8 * CL.K), dim3(THREADS), lds, st, S.self_dev, CL.self_dev, W, A, b0, nb, hc_off, tail_off);
  }
  return hipGetLastError();
}

}  // namespace dc
