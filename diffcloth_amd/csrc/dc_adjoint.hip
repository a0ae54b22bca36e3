// CDNA4 (gfx950) adjoint kernel: Simulation::stepBackward() (reference Simulation.cpp:1455-1780), one
// workgroup per rollout, matrix-free.
//
//   reference fixed point   P u = g + dP^T u,  dP^T u = h^2 (dp/dx)^T A y - C^T w,  w = dr_df^T u,  y = u + w
//   adjoint operator        K u := (P - dP^T) u = M u + h^2 (A - dp/dx)^T A y          (C = h^2 A^T A, P = M + C)
//
// The sparse Jacobian dproj_dxnew of the reference (66 % of its backward time, serial triplet assembly) is
// never formed: the per-element blocks are re-derived from x_new in registers and applied on the fly, and the
// contact Jacobian dr_df is applied per contact from (n, d, mu).
//
// Two solvers for K u = g:
//   mode 0  the reference's iteration (Simulation.cpp:1561-1600): u <- u + P^-1 (g - K u) with the block-Jacobi
//           PCG of the forward pass for P^-1, stop on |u_new - u|_2 / N < backwardConvergenceThreshold; when the
//           cap is reached it falls back to the direct solve, as the reference does with SparseLU (:1589-1594);
//   mode 1  direct solve (semantics of backwardGradientForceDirectSolver / solveDirect, Simulation.cpp:1431-1440):
//           block-Jacobi preconditioned BiCGSTAB on K itself, relative residual <= adjoint_rel_tol.
#define DC_KERNEL_TU
#include <cstdlib>
#include "dc_devlib.h"
#include "dc_winlib.h"
#include "dc_denselib.h"
#include "dc_adjprecond.h"

namespace dc {

constexpr int kCycles = 1;      // BiCGSTAB cycles of the direct adjoint solve (see the comment at the loop)

#ifdef DC_PROFILE_PHASES
#define PH_DECL long long ph_t = clock64(); long long ph_acc[4] = {0, 0, 0, 0};
#define PH(k) { long long n_ = clock64(); ph_acc[k] += n_ - ph_t; ph_t = n_; }
#define PH_PRINT if (blockIdx.x == 0 && threadIdx.x == 0) printf("[phases adj] iters %d | per iter: operator(x2) %lld vector-ops %lld | setup %lld final %lld cycles\n", iters, ph_acc[1] / max(iters, 1), ph_acc[2] / max(iters, 1), ph_acc[0], ph_acc[3]);
#else
#define PH_DECL
#define PH(k)
#define PH_PRINT
#endif

namespace {

// two simultaneous workgroup sums
template <int THREADS>
__device__ __forceinline__ void block_sum2(double &a, double &b, double *red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) { red[w] = a; red[THREADS / 64 + w] = b; }
  __syncthreads();
  double sa = 0, sb = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; k++) { sa += red[k]; sb += red[THREADS / 64 + k]; }
  a = sa; b = sb;
}

struct AdjCtx {
  const float *xnew, *rec_f, *rec_n, *mu;
  const int *rec_prim;
  float *y, *corner, *lds;
  int lds_floats;               // size of the dynamic LDS region (0 without element windows)
  SelfRec self;
  int nself, b;
};

// w = dr_df^T z for the (block-diagonal) primitive contacts: Simulation::calculatedr_df (Simulation.cpp:700-711)
__device__ __forceinline__ f3 contact_JT(const DevSystem &S, const AdjCtx &C, int i, f3 z) {
  const int prim = C.rec_prim[i];
  if (prim < 0) return mk(0, 0, 0);
  const int N = S.N;
  f3 n = ld3(C.rec_n, i, N);
  f3 d = ld3(C.rec_f, i, N) - prim_vout(S.prims[prim], n) * S.mass[i];
  return dri_dfi_T(n, d, C.mu[S.prims[prim].group], z);
}

// y = (I + dr_df)^T z with (I + dr_df) = (I + J_L) ... (I + J_0)(I + J_prim) (calculatedr_df, Simulation.cpp:686-768):
// self layers L..0 first (Gauss-Seidel order reversed), the block-diagonal primitive part last. Ends with a barrier.
template <int THREADS>
__device__ __forceinline__ void contact_transpose(const DevSystem &S, const AdjCtx &C, const float *zin, bool precond, float *y) {
  const int N = S.N, tid = threadIdx.x;
  if (C.nself > 0) {
    for (int i = tid; i < N; i += THREADS) {
      f3 z = ld3(zin, i, N);
      if (precond) z = z * S.dinv[i];
      st3(y, i, N, z);
    }
    __syncthreads();
    if (!self_JT_layers_lds<THREADS>(S, C.self, C.b, y, C.lds, C.lds_floats)) self_JT_layers<THREADS>(S, C.self, C.b, y);
    for (int i = tid; i < N; i += THREADS) {
      f3 z = ld3(y, i, N);
      st3(y, i, N, z + contact_JT(S, C, i, z));
    }
  } else {
    constexpr int VB = 4;
    for (int i0 = tid; i0 < N; i0 += VB * THREADS) {
      f3 zq[VB], nq[VB], fq[VB];
      int pr[VB];
      float mq[VB];
#pragma unroll
      for (int j = 0; j < VB; j++) {
        const int ic = min(i0 + j * THREADS, N - 1);
        zq[j] = ld3(zin, ic, N);
        if (precond) zq[j] = zq[j] * S.dinv[ic];
        pr[j] = C.rec_prim[ic]; nq[j] = ld3(C.rec_n, ic, N); fq[j] = ld3(C.rec_f, ic, N); mq[j] = S.mass[ic];
      }
#pragma unroll
      for (int j = 0; j < VB; j++) {
        const int i = i0 + j * THREADS;
        f3 w = mk(0, 0, 0);
        if (pr[j] >= 0) {
          f3 d = fq[j] - prim_vout(S.prims[pr[j]], nq[j]) * mq[j];
          w = dri_dfi_T(nq[j], d, C.mu[S.prims[pr[j]].group], zq[j]);
        }
        if (i < N) st3(y, i, N, zq[j] + w);
      }
    }
  }
  __syncthreads();
}

// out = K z with z = zin (optionally scaled by D^-1: right preconditioning). Also returns the partial sums
// of out.d1 and out.out of this thread (d1 may be null). Ends WITHOUT a barrier: callers reduce next.
template <int THREADS>
__device__ __forceinline__ void adjoint_operator_global(const DevSystem &S, const AdjCtx &C, const float *zin, bool precond,
                                                        float *out, const float *d1, float &dot1, float &dot2) {
  const int N = S.N, T = S.T, E = S.E, NC = S.NC, tid = threadIdx.x;
  const float h2 = S.h * S.h;
  // __restrict__ + unroll: lets the scheduler overlap the index -> gather -> store chains of neighbouring
  // iterations (each wave otherwise exposes two dependent memory latencies per element)
  float *__restrict__ y = C.y;
  float *__restrict__ corner = C.corner;
  const float *__restrict__ xnew = C.xnew;
  // ---- y = (I + dr_df)^T z ----
  contact_transpose<THREADS>(S, C, zin, precond, y);
  // ---- per element: h^2 (A - dp/dx)^T A y ----
  // triangles: Triangle::projectToManifoldBackward (Triangle.cpp:354-451) in closed form:
  //   dT(Y) = TJ <TJ,Y> / tr(S) + (I - T T^T) Y S^-1,   TJ = [t1, -t0]
#pragma unroll 2
  for (int t = tid; t < T; t += THREADS) {
    const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
    const float4 D = S.tri_D[t];
    f3 x0 = ld3(xnew, i0, N);
    f3 e0 = ld3(xnew, i1, N) - x0, e1 = ld3(xnew, i2, N) - x0;
    Polar P = polar3x2(e0 * D.x + e1 * D.z, e0 * D.y + e1 * D.w);
    f3 q0 = ld3(y, i0, N);
    f3 d0 = ld3(y, i1, N) - q0, d1v = ld3(y, i2, N) - q0;
    f3 y0 = d0 * D.x + d1v * D.z, y1 = d0 * D.y + d1v * D.w;
    const float c = (dot(P.t1, y0) - dot(P.t0, y1)) / P.trS;
    f3 z0 = y0 * P.i00 + y1 * P.i01, z1 = y0 * P.i01 + y1 * P.i11;
    z0 = z0 - P.t0 * dot(P.t0, z0) - P.t1 * dot(P.t1, z0);
    z1 = z1 - P.t0 * dot(P.t0, z1) - P.t1 * dot(P.t1, z1);
    const float s = h2 * S.tri_w2[t];
    f3 r0 = (y0 - (P.t1 * c + z0)) * s, r1 = (y1 - (z1 - P.t0 * c)) * s;
    f3 c1 = r0 * D.x + r1 * D.y, c2 = r0 * D.z + r1 * D.w;
    st3(corner, t, NC, mk(0, 0, 0) - c1 - c2); st3(corner, T + t, NC, c1); st3(corner, 2 * T + t, NC, c2);
  }
  // bending: TriangleBending::backwardGradient (TriangleBending.cpp:154-172)
#pragma unroll 2
  for (int e = tid; e < E; e += THREADS) {
    const int i0 = S.bend_v[e], i1 = S.bend_v[E + e], i2 = S.bend_v[2 * E + e], i3 = S.bend_v[3 * E + e];
    const float4 w = S.bend_w[e];
    const float2 nw = S.bend_nw[e];
    f3 q0 = ld3(y, i0, N);
    f3 ey = (ld3(y, i1, N) - q0) * w.y + (ld3(y, i2, N) - q0) * w.z + (ld3(y, i3, N) - q0) * w.w;
    f3 res = ey;
    if (nw.x > 1e-6f) {
      f3 x0 = ld3(xnew, i0, N);
      f3 ev = (ld3(xnew, i1, N) - x0) * w.y + (ld3(xnew, i2, N) - x0) * w.z + (ld3(xnew, i3, N) - x0) * w.w;
      float en = sqrtf(dot(ev, ev));
      f3 eh = ev * (1.0f / en);
      res = ey - (ey - eh * dot(eh, ey)) * (nw.x / en);
    }
    res = res * (h2 * nw.y);
    const int base = 3 * T;
    st3(corner, base + e, NC, res * w.x); st3(corner, base + E + e, NC, res * w.y);
    st3(corner, base + 2 * E + e, NC, res * w.z); st3(corner, base + 3 * E + e, NC, res * w.w);
  }
  __syncthreads();
  // ---- vertex gather: out = M z + sum(corners) + attachment term ----
  dot1 = 0.f; dot2 = 0.f;
#pragma unroll 2
  for (int i = tid; i < N; i += THREADS) {
    f3 z = ld3(zin, i, N);
    if (precond) z = z * S.dinv[i];
    f3 o = z * S.mass[i];
    const int k1 = S.inc_ptr[i + 1];
    for (int k = S.inc_ptr[i]; k < k1; k++) o = o + ld3(corner, S.inc_idx[k], NC);
    if (S.att_of_vertex[i] >= 0) o = o + ld3(y, i, N) * (h2 * S.k_att);   // attachment: dp/dx = 0 (AttachmentSpring.cpp:35-37)
    st3(out, i, N, o);
    if (d1) dot1 += dot(o, ld3(d1, i, N));
    dot2 += dot(o, o);
  }
}

// Same operator with the element pass inside LDS (element windows, dc_winlib.h): no corner array, no atomics.
template <int THREADS, bool WIN>
__device__ __forceinline__ void adjoint_operator(const DevSystem &S, const AdjCtx &C, const float *zin, bool precond,
                                                 float *out, const float *d1, float &dot1, float &dot2) {
  if constexpr (!WIN) { adjoint_operator_global<THREADS>(S, C, zin, precond, out, d1, dot1, dot2); return; }
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
    // layered self contacts couple vertices: y = (I + dr_df)^T z is formed in global memory first
    contact_transpose<THREADS>(S, C, zin, precond, C.y);     // ends with a barrier
    element_windows<THREADS>(S, C.lds, StagePlanar{C.y, N}, C.xnew, AdjTriOp{h2}, AdjBendOp{h2}, vert);
  } else {
    // primitive contacts only: dr_df is block diagonal, y_i is formed per vertex while the window is staged
    element_windows<THREADS>(S, C.lds, [&](int i) {
      f3 z = ld3(zin, i, N);
      if (precond) z = z * S.dinv[i];
      return z + contact_JT(S, C, i, z);
    }, C.xnew, AdjTriOp{h2}, AdjBendOp{h2}, vert);
  }
  dot1 = a1; dot2 = a2;
}

}  // namespace

// BLK: direct solve preconditioned with K's own 3 x 3 diagonal blocks (dc_adjprecond.h) instead of diag(P)^-1
template <int THREADS, bool WIN, bool DENSE, bool BLK>
__global__ __launch_bounds__(THREADS) void k_adjoint_step(const DevSystem *__restrict__ Sp, DevWork W, BwdArgs A) {
  const DevSystem &S = *Sp;
  extern __shared__ float dyn_lds[];      // element windows (S.win_lds_bytes)
  __shared__ double red[2 * (THREADS / 64)];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = S.N;
  const size_t off = (size_t) b * 3 * N;
  // A.nsteps consecutive steps of the backward sweep of this rollout in one launch (dc_rollout_backward): step s
  // differentiates tape slot k - s; the carried gradient stays in gx / gv. Rollouts are independent, so none of them has
  // to wait for the slowest one after every step.
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
  AdjCtx C;
  C.lds = dyn_lds; C.lds_floats = WIN ? S.win_lds_bytes / 4 : 0;
  C.xnew = A.x_new + off; C.rec_f = A.rec_f + off; C.rec_n = A.rec_n + off;
  C.rec_prim = A.rec_prim + (size_t) b * N;
  C.mu = A.mu + (size_t) b * S.ngroups;
  C.y = W.vbest + off; C.corner = W.corner + (size_t) b * 3 * S.NC;
  C.self = A.self; C.b = b;
  C.nself = (S.contact_enabled && S.self_enabled) ? A.self.meta[(size_t) b * kMetaStride] : 0;
  float *gx = A.gx + off, *gv = A.gv + off;
  float *gin = W.g + off, *u = W.vnow + off;
  float *cg_r = W.cg_r + off, *cg_p = W.cg_p + off, *cg_ap = W.cg_ap + off, *cg_x = W.cg_x + off;
  const float h = S.h, h2 = S.h * S.h;

  // ---- gradient clipping (Simulation.cpp:1460-1466) and u = 0 ----
  float part = 0.f;
  for (int i = tid; i < N; i += THREADS) { f3 q = ld3(gx, i, N); part += dot(q, q); }
  double gnorm = sqrt(block_sum<THREADS>((double) part, red));
  float gscale = 1.f;
  int clipped = 0;
  if (A.clip && gnorm > (double) A.clip_thr * N) { gscale = (float) ((double) A.clip_thr * N / gnorm); clipped = 1; gnorm = (double) A.clip_thr * N; }
  for (int i = tid; i < N; i += THREADS) {
    st3(gin, i, N, ld3(gx, i, N) * gscale);
    st3(u, i, N, mk(0, 0, 0));
  }
  PH_DECL
  int status = 0;          // 1 converged, 2 stalled at the fp32 floor, 0 cap hit
  int iters = 0, cg_total = 0, used_direct = 0;
  double udiff = 0;
  __syncthreads();

  bool need_direct = (A.mode == 1);
  if (A.mode == 0 && gnorm > 0) {
    // ---- the reference's iteration: u <- u + P^-1 (g - K u) ----
    double min_udiff = 1e300;
    int since_progress = 0;
    for (int it = 0; it < A.it_cap; it++) {
      float d1, d2;
      adjoint_operator<THREADS, WIN>(S, C, u, false, cg_ap, nullptr, d1, d2);
      __syncthreads();          // the windows hand out vertices in their own order: K u is complete only after a barrier
      part = 0.f;
      for (int i = tid; i < N; i += THREADS) {
        f3 r = ld3(gin, i, N) - ld3(cg_ap, i, N);
        const float di = S.dinv[i];
        st3(cg_r, i, N, r);
        st3(cg_p, i, N, r * di);
        st3(cg_x, i, N, mk(0, 0, 0));
        part += dot(r, r) * di;
      }
      const double rz = block_sum<THREADS>((double) part, red);
      if constexpr (DENSE) {
        // small meshes: P^-1 r = D^-1/2 Ahat^-1 D^-1/2 r with the explicit fp32 inverse (dc_denselib.h) — its ~1e-5 relative
        // error is at the level of the inner PCG tolerance, and the outer iteration works on the true residual g - K u
        const int ld = S.dense_ld;
        float2 *dxy = (float2 *) dyn_lds;
        float *dz = dyn_lds + 2 * ld, *dpart = dyn_lds + 3 * ld;
        __syncthreads();                                   // cg_r complete, LDS free
        for (int i = tid; i < ld; i += THREADS) {
          f3 q = mk(0, 0, 0);
          if (i < N) q = ld3(cg_r, i, N) * S.sq_dinv[i];
          dxy[i] = make_float2(q.x, q.y); dz[i] = q.z;
        }
        __syncthreads();
        const int Cn = dense_partials<THREADS>(S, dxy, dz, dpart);
        __syncthreads();
        for (int i = tid; i < N; i += THREADS) st3(cg_x, i, N, dense_row_sum(dpart, ld, Cn, i) * S.sq_dinv[i]);
        cg_total += 1;
        (void) rz;
      } else {
        cg_total += block_pcg<THREADS>(S, cg_r, cg_p, cg_ap, cg_x, rz, A.cg_tol, A.cg_max, red);
      }
      part = 0.f;
      for (int i = tid; i < N; i += THREADS) {
        f3 d = ld3(cg_x, i, N);
        st3(u, i, N, ld3(u, i, N) + d);
        part += dot(d, d);
      }
      udiff = sqrt(block_sum<THREADS>((double) part, red)) / (double) N;
      iters = it + 1;
      if (udiff < (double) A.bwd_tol) { status = 1; break; }
      if (udiff < min_udiff) since_progress = 0;
      if (udiff < min_udiff) min_udiff = udiff;
      if (++since_progress >= A.stall_window) { status = 2; break; }
    }
    need_direct = (status == 0);     // cap reached without convergence -> direct solve (Simulation.cpp:1589-1594)
    __syncthreads();
  }

  if (need_direct && gnorm > 0) {
    // ---- block-Jacobi preconditioned BiCGSTAB on K u = g, starting from the current u ----
    used_direct = 1;
    constexpr int VB = 4;
    PH(0)
    float *r = cg_r, *p = cg_p, *v = cg_ap, *t = cg_x, *rhat = W.sd_sx + off;    // (detection scratch, idle in the backward pass)
    float *ph = W.pre_p + off, *sh = W.pre_s + off, *minv = W.minv + (size_t) b * 9 * N;   // M^-1 p, M^-1 s, the block inverses
    if constexpr (BLK) {   // K's own 3 x 3 diagonal blocks at this step's x_new, inverted (dc_adjprecond.h)
      for (int i = tid; i < N; i += THREADS)
        store_block_inverse(elastic_diag_block(S, C.xnew, i), S.mass[i], [&](f3 e) { return contact_JT(S, C, i, e); }, minv, i, N);
    }
    auto pre = [&](int i, f3 z) { return block_pre(minv, i, N, z); };
    float d1, d2;
    const double stop = (double) A.rel_tol * (double) A.rel_tol * gnorm * gnorm;
    const int kcap = A.it_cap > 0 ? 4 * A.it_cap : 1600;
    double rr = 0;
    // kCycles > 1: when the recurrence residual says "converged", recompute g - K u and restart from u if that is not below the
    // tolerance. Measured on the C4 workload (r02m): +4 iterations of 41, gradient error against the fp64 oracle unchanged to three
    // digits (7.31e-5 -> 7.31e-5) — the fp32 floor of this solve is eps * cond(K) in the operator's coefficients, not residual
    // drift — so one cycle is the default.
    for (int cycle = 0, kdone = 0; cycle < kCycles; cycle++) {
    // r = rhat = p = g - K u; the direct mode starts from u = 0: K u = 0 without applying the operator (one of ~75 applications)
    const bool u_is_zero = (A.mode == 1 && cycle == 0);
    if (!u_is_zero) {
      adjoint_operator<THREADS, WIN>(S, C, u, false, v, nullptr, d1, d2);
      __syncthreads();            // (as above; inside the loop the block reductions that follow every application do this)
    }
    part = 0.f;
    for (int i = tid; i < N; i += THREADS) {
      f3 q = ld3(gin, i, N);
      if (!u_is_zero) q = q - ld3(v, i, N);
      st3(r, i, N, q); st3(rhat, i, N, q); st3(p, i, N, q);
      if constexpr (BLK) st3(ph, i, N, pre(i, q));
      part += dot(q, q);
    }
    double rho = block_sum<THREADS>((double) part, red);   // rhat.r = r.r
    rr = rho;
    double best_rr = rr;
    int since_progress = 0;
    status = (rr <= stop) ? 1 : 0;
    if (status == 0 && cycle > 0 && cycle == kCycles - 1) status = 2;      // still above the tolerance after two restarts: the fp32 floor of this system
    if (status != 0) break;
    for (int k = kdone; k < kcap && status == 0; k++, kdone++) {
      // v = K M^-1 p ;  alpha = rho / (rhat . v)
      PH(2)
      if constexpr (BLK) adjoint_operator<THREADS, WIN>(S, C, ph, false, v, rhat, d1, d2);
      else adjoint_operator<THREADS, WIN>(S, C, p, true, v, rhat, d1, d2);
      double rv = block_sum<THREADS>((double) d1, red);
      PH(1)
      if (!(fabs(rv) > 1e-300)) { status = 2; break; }
      const float alpha = (float) (rho / rv);
      // s = r - alpha v  (in place)
      // (vector updates: VB vertices of a thread per round, all loads issued before the first store)
      part = 0.f;
      for (int i0 = tid; i0 < N; i0 += VB * THREADS) {
        f3 rq[VB], vq[VB];
#pragma unroll
        for (int j = 0; j < VB; j++) { const int ic = min(i0 + j * THREADS, N - 1); rq[j] = ld3(r, ic, N); vq[j] = ld3(v, ic, N); }
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int i = i0 + j * THREADS;
          f3 s = rq[j] - vq[j] * alpha;
          if (i < N) { st3(r, i, N, s); if constexpr (BLK) st3(sh, i, N, pre(i, s)); part += dot(s, s); }
        }
      }
      double ss = block_sum<THREADS>((double) part, red);
      iters++;
      if (ss <= stop) {
        for (int i = tid; i < N; i += THREADS) st3(u, i, N, ld3(u, i, N) + (BLK ? ld3(ph, i, N) * alpha : ld3(p, i, N) * (alpha * S.dinv[i])));
        rr = ss; status = 1; break;
      }
      // t = K M^-1 s ;  omega = (t . s) / (t . t)
      PH(2)
      if constexpr (BLK) adjoint_operator<THREADS, WIN>(S, C, sh, false, t, r, d1, d2);
      else adjoint_operator<THREADS, WIN>(S, C, r, true, t, r, d1, d2);
      double ts = (double) d1, tt = (double) d2;
      block_sum2<THREADS>(ts, tt, red);
      PH(1)
      if (!(tt > 1e-300)) { status = 2; break; }
      const float omega = (float) (ts / tt);
      // u += alpha M^-1 p + omega M^-1 s ;  r = s - omega t ;  rho_new = rhat . r
      float pa = 0.f, pb = 0.f;
      for (int i0 = tid; i0 < N; i0 += VB * THREADS) {
        f3 sq[VB], uq[VB], pq[VB], tq[VB], hq[VB], zq[VB];
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int ic = min(i0 + j * THREADS, N - 1);
          sq[j] = ld3(r, ic, N); uq[j] = ld3(u, ic, N); tq[j] = ld3(t, ic, N); hq[j] = ld3(rhat, ic, N);
          if constexpr (BLK) { zq[j] = ld3(sh, ic, N); pq[j] = ld3(ph, ic, N); }
          else { const float di = S.dinv[ic]; zq[j] = sq[j] * di; pq[j] = ld3(p, ic, N) * di; }
        }
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int i = i0 + j * THREADS;
          f3 rn = sq[j] - tq[j] * omega;
          if (i < N) {
            st3(u, i, N, uq[j] + pq[j] * alpha + zq[j] * omega);
            st3(r, i, N, rn);
            pa += dot(rn, hq[j]);
            pb += dot(rn, rn);
          }
        }
      }
      double rho_new = (double) pa;
      rr = (double) pb;
      block_sum2<THREADS>(rho_new, rr, red);
      if (rr <= stop) { status = 1; break; }
      if (rr < best_rr) { best_rr = rr; since_progress = 0; }
      else if (++since_progress >= A.stall_window) { status = 2; break; }
      if (!(fabs(rho_new) > 1e-300) || !(fabs(omega) > 0.f)) { status = 2; break; }
      const float beta = (float) ((rho_new / rho) * ((double) alpha / (double) omega));
      rho = rho_new;
      // p = r + beta (p - omega v)
      for (int i0 = tid; i0 < N; i0 += VB * THREADS) {
        f3 rq[VB], pq[VB], vq[VB];
#pragma unroll
        for (int j = 0; j < VB; j++) { const int ic = min(i0 + j * THREADS, N - 1); rq[j] = ld3(r, ic, N); pq[j] = ld3(p, ic, N); vq[j] = ld3(v, ic, N); }
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int i = i0 + j * THREADS;
          if (i < N) { const f3 pn = rq[j] + (pq[j] - vq[j] * omega) * beta; st3(p, i, N, pn); if constexpr (BLK) st3(ph, i, N, pre(i, pn)); }
        }
      }
      __syncthreads();
    }
    if (status != 1) break;       // cap, breakdown or stall: no further cycle
    __syncthreads();
    }   // cycle
    // "stalled at the fp32 floor" (2) is only claimed near the tolerance: a breakdown or stall with the residual still more than 100 x
    // above it (an adjoint system beyond an fp32 Krylov solve, e.g. a strongly compressed fine garment) is reported as NOT converged
    if (status == 2 && rr > 1e4 * stop) status = 0;
    udiff = sqrt(rr) / (gnorm > 0 ? gnorm : 1.0);     // relative residual (of the last recomputed or recurrence residual)
  }
  __syncthreads();
  PH(2)
  // ---- gradients w.r.t. the previous state and parameters (Simulation.cpp:1534, 1608-1650) ----
  float dmu_part[kMaxPrims];
#pragma unroll
  for (int k = 0; k < kMaxPrims; k++) dmu_part[k] = 0.f;
  float *dxf = A.d_xfixed ? A.d_xfixed + (size_t) b * 3 * S.Af : nullptr;
  contact_transpose<THREADS>(S, C, u, false, C.y);       // y = (I + dr_df)^T u*
  // parameter gradients of this step (Simulation.cpp:1672-1764), all of the form  <y, d(rhs)/d(theta)>:
  //   pacc[0..2]  sum over the elements of one type of  y . A^T (p(x_new) - A x_new)   -> dL/dk_type = h^2 / k * pacc
  //   pacc[3]     density term (:1672-1679, adddr_dd = false)
  //   pacc[4..6]  h^2 * sum_i y_i  (dL_dfext_vec summed, :1702-1764; the host applies the wind chain rule)
  float pacc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (A.d_param) {
    const int T = S.T, E = S.E;
    const float *xnew = C.xnew;
    const float *yv = C.y;
    for (int t = tid; t < T; t += THREADS) {
      const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
      const float4 D = S.tri_D[t];
      f3 x0 = ld3(xnew, i0, N);
      f3 e0 = ld3(xnew, i1, N) - x0, e1 = ld3(xnew, i2, N) - x0;
      f3 f0 = e0 * D.x + e1 * D.z, f1 = e0 * D.y + e1 * D.w;
      Polar P = polar3x2(f0, f1);
      f3 g0 = (P.t0 - f0) * S.tri_w2[t], g1 = (P.t1 - f1) * S.tri_w2[t];
      f3 c1 = g0 * D.x + g1 * D.y, c2 = g0 * D.z + g1 * D.w;
      f3 q0 = ld3(yv, i0, N);
      pacc[0] += dot(c1, ld3(yv, i1, N) - q0) + dot(c2, ld3(yv, i2, N) - q0);
    }
    for (int e = tid; e < E; e += THREADS) {
      const int i0 = S.bend_v[e], i1 = S.bend_v[E + e], i2 = S.bend_v[2 * E + e], i3 = S.bend_v[3 * E + e];
      const float4 w = S.bend_w[e];
      const float2 nw = S.bend_nw[e];
      f3 x0 = ld3(xnew, i0, N);
      f3 ev = (ld3(xnew, i1, N) - x0) * w.y + (ld3(xnew, i2, N) - x0) * w.z + (ld3(xnew, i3, N) - x0) * w.w;
      f3 p = mk(0, 0, 0);
      if (nw.x > 1e-6f) p = normalized(ev) * nw.x;
      f3 q0 = ld3(yv, i0, N);
      f3 ey = (ld3(yv, i1, N) - q0) * w.y + (ld3(yv, i2, N) - q0) * w.z + (ld3(yv, i3, N) - q0) * w.w;
      pacc[1] += dot((p - ev) * nw.y, ey);
    }
  }
  const f3 grav = mk(S.gx, S.gy, S.gz);
  for (int i = tid; i < N; i += THREADS) {
    f3 ui = ld3(u, i, N);
    const float m = S.mass[i];
    f3 w = ld3(C.y, i, N) - ui;
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
  if (A.d_mu) {
    for (int k = 0; k < S.ngroups; k++) {
      const double s = block_sum<THREADS>((double) dmu_part[k], red);
      if (tid == 0) A.d_mu[(size_t) b * S.ngroups + k] += (float) s;
    }
  }
  if (A.d_param) {
    float *dp = A.d_param + (size_t) b * 8;
    const float scale[7] = {S.k_stretch > 0.f ? h2 / S.k_stretch : 0.f, S.k_bend > 0.f ? h2 / S.k_bend : 0.f,
                            S.k_att > 0.f ? h2 / S.k_att : 0.f, 1.f, 1.f, 1.f, 1.f};
    for (int k = 0; k < 7; k++) {
      const double s = block_sum<THREADS>((double) pacc[k], red);
      if (tid == 0) dp[k] = (float) (s * scale[k]);
    }
  }
  if (tid == 0) {
    dc_bwd_stats s;
    s.converged = status; s.adjoint_iters = iters; s.cg_iters = cg_total; s.clipped = clipped;
    s.used_direct = used_direct; s.last_udiff = (float) udiff;
    A.stats[b] = s;
  }
  PH(3)
  PH_PRINT
  }   // step
}

static int pick_threads_bwd(int N) {
  static const int forced = getenv("DC_BWD_THREADS") ? atoi(getenv("DC_BWD_THREADS")) : 0;     // development switch
  if (forced == 256 || forced == 512 || forced == 1024) return forced;
  // 16 waves per rollout at every mesh size: the Krylov iteration is a chain of barrier-separated phases with global-memory
  // round trips, and with one workgroup per CU (256 rollouts) only the waves of that workgroup can hide them — measured
  // 1.3 - 1.7 x over 256 / 512 threads from N = 579 to N = 3634 (tools/bench_configs.py), even with idle lanes at N < 1024
  (void) N;
  return 1024;
}

template <int THREADS, bool DENSE, bool BLK>
static void launch_adj_b(const DevSystem &S, const DevWork &W, const BwdArgs &A, int B, hipStream_t st) {
  if (!S.win_ok) { hipLaunchKernelGGL((k_adjoint_step<THREADS, false, false, false>), dim3(B), dim3(THREADS), 0, st, S.self_dev, W, A); return; }
  size_t lds = (size_t) S.win_lds_bytes;
  if (DENSE) lds = std::max(lds, sizeof(float) * (size_t) (3 * S.dense_ld + dense_lds_floats(S.dense_ld, THREADS / 64)));
  static size_t configured[kMaxDevices] = {};        // the attribute is per device: one entry per device this process has used
  int dev = 0;
  (void) hipGetDevice(&dev);
  size_t &done = configured[dev >= 0 && dev < kMaxDevices ? dev : 0];
  if (lds > done || dev >= kMaxDevices) {
    (void) hipFuncSetAttribute((const void *) k_adjoint_step<THREADS, true, DENSE, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    done = lds;
  }
  hipLaunchKernelGGL((k_adjoint_step<THREADS, true, DENSE, BLK>), dim3(B), dim3(THREADS), lds, st, S.self_dev, W, A);
}
template <int THREADS, bool DENSE>
static void launch_adj(const DevSystem &S, const DevWork &W, const BwdArgs &A, int B, hipStream_t st) {
  // the block preconditioner belongs to the direct solve (mode 1); the reference's iteration (mode 0) uses P^-1 as the reference does
  if (A.block_pre && A.mode == 1 && !DENSE) launch_adj_b<THREADS, DENSE, true>(S, W, A, B, st);
  else launch_adj_b<THREADS, DENSE, false>(S, W, A, B, st);
}

void launch_adjoint_step(const DevSystem &S, const DevWork &W, const BwdArgs &A, int B, hipStream_t st) {
  // small meshes, reference iteration (mode 0): the inner solve with P is one product with the explicit inverse (dc_dense.h)
  if (S.dense_inv && S.win_ok && A.mode == 0 && pick_threads_bwd(S.N) == 1024) { launch_adj<1024, true>(S, W, A, B, st); return; }
  switch (pick_threads_bwd(S.N)) {
    case 256: launch_adj<256, false>(S, W, A, B, st); break;
    case 512: launch_adj<512, false>(S, W, A, B, st); break;
    default: launch_adj<1024, false>(S, W, A, B, st); break;
  }
}

}  // namespace dc
