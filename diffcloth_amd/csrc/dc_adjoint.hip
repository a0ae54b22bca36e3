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
//   mode 1  direct solve (semantics of backwardGradientForceDirectSolver / solveDirect, Simulation.cpp:1431-1440): mixed-precision refinement on K
//           itself — fp32 correction solves (CG first with the diag(P) preconditioner, block-Jacobi preconditioned BiCGSTAB otherwise or once CG
//           stalls) of a residual evaluated in fp64, fp64 BiCGSTAB fall-back — to a relative residual <= adjoint_rel_tol.
#define DC_KERNEL_TU
#include <cstdlib>
#include "dc_devlib.h"
#include "dc_winlib.h"
#include "dc_denselib.h"
#include "dc_adjprecond.h"
#include "dc_adjoint64.h"

namespace dc {

constexpr int kMaxRefine = 6;          // fp32 correction solves of the mixed-precision direct adjoint solve before the fp64 fall-back
constexpr double kFallbackGain = 1e-4; // the fp64 fall-back aims this far below the caller's relative tolerance (direct-solve semantics)
constexpr double kInnerFloor = 1e-3;   // an fp32 correction solve never aims below this fraction of its own right-hand side

#ifdef DC_PROFILE_PHASES
#define PH_DECL long long ph_t = clock64(); long long ph_acc[4] = {0, 0, 0, 0};
#define PH(k) { long long n_ = clock64(); ph_acc[k] += n_ - ph_t; ph_t = n_; }
#define PH_PRINT if (blockIdx.x == 0 && threadIdx.x == 0) { printf("[phases adj] iters %d | per step: fp64 residuals %lld fp32 solves %lld setup %lld final %lld cycles | per iter: contact^T x2 %lld windows x2 (stage %lld tri %lld bend %lld vertex %lld) s-update %lld d,r-update %lld p-update %lld\n", iters, ph_acc[1], ph_acc[2], ph_acc[0], ph_acc[3], g_adj_ph[0] / max(iters, 1), g_win_ph[0] / max(iters, 1), g_win_ph[1] / max(iters, 1), g_win_ph[2] / max(iters, 1), g_win_ph[3] / max(iters, 1), g_adj_ph[1] / max(iters, 1), g_adj_ph[2] / max(iters, 1), g_adj_ph[3] / max(iters, 1)); for (int q_ = 0; q_ < 4; q_++) g_adj_ph[q_] = g_win_ph[q_] = 0; }
static __device__ long long g_adj_ph[4];
#define APH_DECL long long aph_t = clock64();
#define APH(k) { if (blockIdx.x == 0 && threadIdx.x == 0) { long long n_ = clock64(); g_adj_ph[k] += n_ - aph_t; aph_t = n_; } }
#define APH_SKIP { aph_t = clock64(); }
#else
#define PH_DECL
#define PH(k)
#define PH_PRINT
#define APH_DECL
#define APH(k)
#define APH_SKIP
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

// The four reductions of a BiCGSTAB iteration: fp64 as before, but inside a wave through DPP row operations (wave_sum_d, dc_adjoint64.h; a
// wave sum of doubles through __shfl_down is twelve ds_bpermute round trips through the LDS crossbar) and with ONE barrier each — every
// reduction of the iteration has a slot of its own in `red4` ([4][2][waves]), and between a slot's read and its next write lies the rest of
// the iteration with all the barriers of two operator applications. On a one-window mesh (hat, sock, T-shirt) an iteration is ~20
// barrier-separated phases of a microsecond: four reductions of two barriers and twelve crossbar trips each were a sixth of it.
// NOT fp32 inside the wave (measured, tools/r05_ab/r05_run21.sh): <rhat, r> cancels, and with wave sums rounded to fp32 one T-shirt rollout
// of 256 needed 214 instead of 78 iterations per step — the launch waits for the slowest rollout (5.5 -> 9.2 ms per batch step).
template <int THREADS>
__device__ __forceinline__ double krylov_sum(float v, double *slot) {
#ifdef DC_ADJ_OLDSUM
  return block_sum<THREADS>((double) v, slot);
#endif
  const double vs = wave_sum_d((double) v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) slot[w] = vs;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; k++) s += slot[k];
  return s;
}
template <int THREADS>
__device__ __forceinline__ void krylov_sum2(float a, float b, double *slot, double &sa, double &sb) {
#ifdef DC_ADJ_OLDSUM
  sa = (double) a; sb = (double) b; block_sum2<THREADS>(sa, sb, slot); return;
#endif
  const double as = wave_sum_d((double) a), bs = wave_sum_d((double) b);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  constexpr int NW = THREADS / 64;
  if (l == 0) { slot[w] = as; slot[NW + w] = bs; }
  __syncthreads();
  sa = 0; sb = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) { sa += slot[k]; sb += slot[NW + k]; }
}


struct AdjCtx {
  const float *xnew, *rec_f, *rec_n, *mu;
  const int *rec_prim;
  float *y, *corner, *lds;
  int lds_floats;               // size of the dynamic LDS region (0 without element windows)
  SelfRec self;
  int nself, b;
  // contact vertices of this step (k_adjoint_step builds them once per step): y differs from z only there
  int *mark;                    // [N] bit 0 = in the working set of the self contacts, bit 1 = in contact with a primitive, bits 2.. = slot in `ylist`
  const int *plist;             // [nplist] the vertices in contact with a primitive
  int nplist;
  // y at the contact vertices, in LDS behind the element windows' region (slot = position in plist, or nplist + position in the self contacts'
  // working set): the windows' staging takes y from here instead of loading the y plane for EVERY span vertex next to z (12 of 44 bytes
  // per span vertex and application; ~1 600 of 10 000 vertices of the bench cloth are in contact). Null when the list does not fit.
  float *ylist;
  int nself_verts;
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

// z (optionally scaled by D^-1) in, y out — at the working-set vertices of the self contacts only (self_JT_layers_lds_v)
struct SelfInOut {
  const float *zin, *dinv;
  float *y;
  int N;
  __device__ __forceinline__ float ld(int idx) const { const int v = idx >= 2 * N ? idx - 2 * N : (idx >= N ? idx - N : idx); return dinv ? zin[idx] * dinv[v] : zin[idx]; }
  __device__ __forceinline__ void st(int idx, float val) const { y[idx] = val; }
};

// Same operator with the element pass inside LDS (element windows, dc_winlib.h): no corner array, no atomics.
// PRE: the per-vertex phase issues a vertex's global reads ahead of the gather (vert_with_pre, dc_winlib.h) — pays on the 10 000-vertex
// cloth (12.95 -> 12.73 ms per batch step), costs the block-preconditioned garment kernels 36 B more scratch and 3 ... 8 % (they pass false)
template <int THREADS, bool WIN, bool PRE>
__device__ __forceinline__ void adjoint_operator(const DevSystem &S, const AdjCtx &C, const float *zin, bool precond,
                                                 float *out, const float *d1, float &dot1, float &dot2) {
  if constexpr (!WIN) { adjoint_operator_global<THREADS>(S, C, zin, precond, out, d1, dot1, dot2); return; }
  const int N = S.N;
  const float h2 = S.h * S.h;
  float a1 = 0.f, a2 = 0.f;
  struct VertIn { f3 z, d; float m; int a; };      // a vertex's global reads, issued ahead of the gather (dc_winlib.h)
  auto vert_plain = [&](int i, f3 sum, f3 yi) {
    f3 z = ld3(zin, i, N);
    if (precond) z = z * S.dinv[i];
    f3 o = z * S.mass[i] + sum;
    if (S.att_of_vertex[i] >= 0) o = o + yi * (h2 * S.k_att);
    st3(out, i, N, o);
    if (d1) a1 += dot(o, ld3(d1, i, N));
    a2 += dot(o, o);
  };
  auto vert_pre = vert_with_pre([&](int i) {
    VertIn q;
    q.z = ld3(zin, i, N);
    if (precond) q.z = q.z * S.dinv[i];
    q.d = d1 ? ld3(d1, i, N) : mk(0, 0, 0);
    q.m = S.mass[i]; q.a = S.att_of_vertex[i];
    return q;
  }, [&](int i, f3 sum, f3 yi, const VertIn &q) {
    f3 o = q.z * q.m + sum;
    if (q.a >= 0) o = o + yi * (h2 * S.k_att);   // attachment: dp/dx = 0 (AttachmentSpring.cpp:35-37)
    st3(out, i, N, o);
    if (d1) a1 += dot(o, q.d);
    a2 += dot(o, o);
  });
  auto vert = [&]() { if constexpr (PRE) return vert_pre; else return vert_plain; }();
  // y = (I + dr_df)^T z differs from z only at the contact vertices: the layered self contacts couple the ~2 x nself vertices of
  // their working set (the layers run on those alone, in LDS, before the windows use it), the primitive contacts are block diagonal
  // (one pass over the list of their vertices); both leave y in global memory, every other vertex is staged as z itself. (Before:
  // two passes over all N vertices per operator application, 49 k of its 290 k cycles on the 10 000-vertex cloth with 500 self and
  // 400 primitive contacts; forming y_i inside the staging instead costs three dependent loads per span vertex, halo included.)
  APH_DECL
  if (C.nself == 0 && (C.mark == nullptr || S.nwin < 4)) {
    // primitive contacts only on a mesh of a few windows (the garment scenes): an operator application there is a chain of latencies, and
    // the list pass with its barrier is one more link — y_i is formed per vertex while the window is staged (the halo is small)
    element_windows<THREADS>(S, C.lds, [&](int i) {
      f3 z = ld3(zin, i, N);
      if (precond) z = z * S.dinv[i];
      return z + contact_JT(S, C, i, z);
    }, C.xnew, AdjTriOp{h2}, AdjBendOp{h2}, vert);
    dot1 = a1; dot2 = a2;
    return;
  }
  bool sparse = C.mark != nullptr;
  // the 1024-thread kernels (always alone on their CU: the LDS behind the windows is theirs) take y of the contact vertices from the LDS list
  // and have no other sparse form: a step whose contact vertices do not fit the list (a sheet lying on a plane) forms y over all vertices below
  constexpr bool YL = THREADS == 1024;
  if constexpr (YL) sparse = sparse && C.ylist != nullptr;
  if (sparse && C.nself > 0) sparse = self_JT_layers_lds_v<THREADS>(S, C.self, C.b, SelfInOut{zin, precond ? S.dinv : nullptr, C.y, N}, C.lds, C.lds_floats);   // ends with a barrier
  if (sparse) {
    const int *mark = C.mark;
    float *ylist = YL ? C.ylist : nullptr;
    if (YL && C.nself > 0) {      // the self pass's working set is still in the windows' LDS ([3][M] planar at its start): its slots follow the primitives'
      const int M = C.nself_verts;
      float *yl = ylist + 3 * C.nplist;
      for (int s = threadIdx.x; s < M; s += THREADS) { yl[3 * s] = C.lds[s]; yl[3 * s + 1] = C.lds[M + s]; yl[3 * s + 2] = C.lds[2 * M + s]; }
    }                                // (a vertex with both kinds of contact: the primitive's slot is the one its mark names, written below)
    for (int q = threadIdx.x; q < C.nplist; q += THREADS) {
      const int i = C.plist[q];
      f3 z;
      if (mark[i] & 1) z = ld3(C.y, i, N);
      else { z = ld3(zin, i, N); if (precond) z = z * S.dinv[i]; }
      const f3 yv = z + contact_JT(S, C, i, z);
      st3(C.y, i, N, yv);
      if constexpr (YL) { ylist[3 * q] = yv.x; ylist[3 * q + 1] = yv.y; ylist[3 * q + 2] = yv.z; }
    }
    __syncthreads();
    APH(0)
    auto stage_y = [&](int i) {
      const int m = mark[i];
      f3 z = ld3(zin, i, N);
      if (precond) z = z * S.dinv[i];
      if constexpr (YL) {            // y of a contact vertex from the LDS list: no load of the y plane at all
        if (m) { const float *yl = ylist + 3 * (m >> 2); z = mk(yl[0], yl[1], yl[2]); }
        return z;
      } else {
      // (both candidates loaded, then selected: one memory round trip, not two — loading y only behind the mark: 12.7 -> 13.2 ms per batch step)
      const f3 yi = ld3(C.y, i, N);
      return mk(m ? yi.x : z.x, m ? yi.y : z.y, m ? yi.z : z.z);
      }
    };
    if constexpr (PRE) {
      // the per-vertex operator takes y_i (attachment vertices only) from its own reads instead of the window's input plane: the per-vertex phase
      // then reads the result planes only and the next window stages without a barrier in between (dc_winlib.h, NOA)
      struct VertInY { f3 z, d, y; float m; int a; };
      auto vert_sp = vert_with_pre_noa([&](int i) {
        VertInY q;
        q.z = ld3(zin, i, N);
        if (precond) q.z = q.z * S.dinv[i];
        q.d = d1 ? ld3(d1, i, N) : mk(0, 0, 0);
        q.m = S.mass[i]; q.a = S.att_of_vertex[i];
        q.y = mk(0, 0, 0);
        if (q.a >= 0) q.y = mark[i] ? ld3(C.y, i, N) : q.z;
        return q;
      }, [&](int i, f3 sum, f3, const VertInY &q) {
        f3 o = q.z * q.m + sum;
        if (q.a >= 0) o = o + q.y * (h2 * S.k_att);   // attachment: dp/dx = 0 (AttachmentSpring.cpp:35-37)
        st3(out, i, N, o);
        if (d1) a1 += dot(o, q.d);
        a2 += dot(o, o);
      });
      element_windows<THREADS>(S, C.lds, stage_y, C.xnew, AdjTriOp{h2}, AdjBendOp{h2}, vert_sp);
    } else
    element_windows<THREADS>(S, C.lds, stage_y, C.xnew, AdjTriOp{h2}, AdjBendOp{h2}, vert);
  } else {
    // (self-contact working set beyond the LDS: y = (I + dr_df)^T z is formed in global memory first)
    contact_transpose<THREADS>(S, C, zin, precond, C.y);     // ends with a barrier
    element_windows<THREADS>(S, C.lds, StagePlanar{C.y, N}, C.xnew, AdjTriOp{h2}, AdjBendOp{h2}, vert);
  }
  dot1 = a1; dot2 = a2;
}

}  // namespace

// vectors of the fp32 correction solve (planar [3][N] of one rollout)
struct Krylov32 {
  float *rhs, *d, *r, *p, *v, *t, *rhat, *ph, *sh, *minv;
};

// Coarse level of the fp32 solve's preconditioner, dst += Z (Z^T P Z)^-1 Z^T src per coordinate, Z = D^-1/2 U over the deflation space the engine
// built for this mesh (dc_deflate.h; see precondition64 in dc_adjoint64.h for the why). The 48 sums Z^T src are formed in fp64 from the fp32
// vectors — they are differences of large terms and G amplifies them by 1 / lambda — in the element windows' LDS (idle between operator
// applications). All threads call; ends with a barrier. Inlined into the COARSE instance of the solve only: a call inside the loop of the
// plain instance made it save its registers around it (hat x 64: 30 -> 67 ms per batch step with the call never taken, measured).
template <int THREADS>
__device__ __forceinline__ void coarse_add32(const DevSystem &S, const float *src, float *dst, float *lds) {
  const int N = S.N, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int DK = kCoarseVectors, NV = 3 * DK, NW = THREADS / 64;
  double *red = (double *) lds;              // [NW][NV]
  double *vals = red + 16 * NV;              // [NV]
  const float4 DC_G *U4 = (const float4 DC_G *) S.defl_u;
  __syncthreads();
  // (four passes of four vectors: one pass with 48 fp64 accumulators per thread was slower, 26.9 -> 30.1 ms per hat batch step)
#pragma unroll 1
  for (int j4 = 0; j4 < DK / 4; j4++) {
    double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < N; i += THREADS) {
      const float4 u = U4[(size_t) i * (DK / 4) + j4];
      const double sq = (double) S.sq_dinv[i];
      const f3 q = ld3(src, i, N);
      const double qx = sq * (double) q.x, qy = sq * (double) q.y, qz = sq * (double) q.z;
      acc[0] += u.x * qx; acc[1] += u.x * qy; acc[2] += u.x * qz;
      acc[3] += u.y * qx; acc[4] += u.y * qy; acc[5] += u.y * qz;
      acc[6] += u.z * qx; acc[7] += u.z * qy; acc[8] += u.z * qz;
      acc[9] += u.w * qx; acc[10] += u.w * qy; acc[11] += u.w * qz;
    }
#pragma unroll
    for (int m = 0; m < 12; m++) {
      const double v = wave_sum_d(acc[m]);
      if (lane == 0) red[wv * NV + j4 * 12 + m] = v;
    }
  }
  __syncthreads();
  double tj = 0;
  if (tid < NV) for (int w = 0; w < NW; w++) tj += red[w * NV + tid];
  __syncthreads();
  if (tid < NV) vals[tid] = tj;
  __syncthreads();
  double cj = 0;
  if (tid < NV) {
    const int j = tid / 3, c = tid - 3 * j;
    for (int l = 0; l < DK; l++) cj += (double) S.defl_g[j * DK + l] * vals[l * 3 + c];
  }
  __syncthreads();
  if (tid < NV) ((float *) red)[tid] = (float) cj;      // c as fp32 for the axpy
  __syncthreads();
  const float *cf = (const float *) red;
  for (int i = tid; i < N; i += THREADS) {
    float zx = 0.f, zy = 0.f, zz = 0.f;
#pragma unroll 1
    for (int j4 = 0; j4 < DK / 4; j4++) {
      const float4 u = U4[(size_t) i * (DK / 4) + j4];
      const float *c = cf + j4 * 12;
      zx += u.x * c[0] + u.y * c[3] + u.z * c[6] + u.w * c[9];
      zy += u.x * c[1] + u.y * c[4] + u.z * c[7] + u.w * c[10];
      zz += u.x * c[2] + u.y * c[5] + u.z * c[8] + u.w * c[11];
    }
    const float sq = S.sq_dinv[i];
    st3(dst, i, N, ld3(dst, i, N) + mk(zx * sq, zy * sq, zz * sq));
  }
  __syncthreads();
}

// One fp32 correction solve of the direct adjoint solve: right-preconditioned BiCGSTAB on K d = rhs from d = 0 until the recurrence
// residual satisfies |r|^2 <= in_stop (in_status 1), a breakdown / stall (2) or the iteration cap (0). BLK: preconditioner = K's own
// inverted 3 x 3 diagonal blocks (V.minv), else diag(P)^-1 inside the operator. A function of its own (not inlined) so that its
// register allocation is that of this loop alone — inlined into the kernel next to the fp64 code, the loop reloaded 400 spilled
// values per iteration.
struct Ret32 {
  int status, kdone, iters;
  double rr;
};
template <int THREADS, bool WIN, bool BLK, bool COARSE = false>
__device__ DC_OUTLINED Ret32 bicgstab32_solve(const DevSystem &S, AdjCtx C, Krylov32 V, double in_stop, int kcap, int stall_window,
                                                            double *red, int kdone, int iters) {
  const int N = S.N, tid = threadIdx.x;
  constexpr int VB = 4;
  __shared__ double red4[8 * (THREADS / 64)];      // (krylov_sum: a slot per reduction of the iteration)
  float *gin = V.rhs, *u = V.d, *r = V.r, *p = V.p, *v = V.v, *t = V.t, *rhat = V.rhat, *ph = V.ph, *sh = V.sh;
  const float *minv = V.minv;
  auto pre = [&](int i, f3 z) { return block_pre(minv, i, N, z); };
  float d1, d2, part;
  double rr;
    // r = rhat = p = rhs, d = 0 (K d = 0 without applying the operator)
    part = 0.f;
    for (int i = tid; i < N; i += THREADS) {
      f3 q = ld3(gin, i, N);
      st3(r, i, N, q); st3(rhat, i, N, q); st3(p, i, N, q); st3(u, i, N, mk(0, 0, 0));
      if constexpr (BLK) st3(ph, i, N, pre(i, q));
      part += dot(q, q);
    }
    double rho = block_sum<THREADS>((double) part, red);   // rhat.r = r.r
    if constexpr (COARSE) coarse_add32<THREADS>(S, p, ph, C.lds);
    rr = rho;
    double best_rr = rr;
    int since_progress = 0;
    int in_status = (rr <= in_stop) ? 1 : 0;
    for (int k = kdone; k < kcap && in_status == 0; k++, kdone++) {
      // v = K M^-1 p ;  alpha = rho / (rhat . v)
      if constexpr (BLK) adjoint_operator<THREADS, WIN, false>(S, C, ph, false, v, rhat, d1, d2);
      else adjoint_operator<THREADS, WIN, true>(S, C, p, true, v, rhat, d1, d2);
      double rv = krylov_sum<THREADS>(d1, red4);
      if (!(fabs(rv) > 1e-300)) { in_status = 2; break; }
      const float alpha = (float) (rho / rv);
      APH_DECL
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
      double ss = krylov_sum<THREADS>(part, red4 + 2 * (THREADS / 64));
      if constexpr (COARSE) { if (ss > in_stop) coarse_add32<THREADS>(S, r, sh, C.lds); }
      APH(1)
      iters++;
      if (ss <= in_stop) {
        for (int i = tid; i < N; i += THREADS) st3(u, i, N, ld3(u, i, N) + (BLK ? ld3(ph, i, N) * alpha : ld3(p, i, N) * (alpha * S.dinv[i])));
        rr = ss; in_status = 1; break;
      }
      // t = K M^-1 s ;  omega = (t . s) / (t . t)
      if constexpr (BLK) adjoint_operator<THREADS, WIN, false>(S, C, sh, false, t, r, d1, d2);
      else adjoint_operator<THREADS, WIN, true>(S, C, r, true, t, r, d1, d2);
      double ts, tt;
      krylov_sum2<THREADS>(d1, d2, red4 + 4 * (THREADS / 64), ts, tt);
      if (!(tt > 1e-300)) { in_status = 2; break; }
      const float omega = (float) (ts / tt);
      APH_SKIP
      // d += alpha M^-1 p + omega M^-1 s ;  r = s - omega t ;  rho_new = rhat . r
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
      double rho_new;
      krylov_sum2<THREADS>(pa, pb, red4 + 6 * (THREADS / 64), rho_new, rr);
      APH(2)
      if (rr <= in_stop) { in_status = 1; break; }
      if (rr < best_rr) { best_rr = rr; since_progress = 0; }
      else if (++since_progress >= stall_window) { in_status = 2; break; }
      if (!(fabs(rho_new) > 1e-300) || !(fabs(omega) > 0.f)) { in_status = 2; break; }
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
      if constexpr (COARSE) coarse_add32<THREADS>(S, p, ph, C.lds);
      __syncthreads();
      APH(3)
    }
  __syncthreads();
  return Ret32{in_status, kdone, iters, rr};
}

// The same correction solve by preconditioned CG (round 6; diag(P)^-1 preconditioner only). K = P - dP^T is symmetric without contact and
// non-symmetric by a few per cent with it (|K - K^T|_F / |K|_F = 4 % on the bench step; tools/offline_adjoint_methods.py), and CG — ONE operator
// application and two vector passes per iteration where BiCGSTAB takes two and three — reaches a correction solve's 1e-3 in ~10 % fewer
// applications there (58 ... 60 against 64 ... 68 over a step's two solves). CG is not a method for a non-symmetric matrix; what makes it
// admissible is the refinement around it: every correction solve is followed by the fp64 residual of the true operator, so the solve only has to
// be a contraction. The caller switches to BiCGSTAB for the remaining cycles of a step as soon as a CG cycle fails to contract that residual
// 4 x, breaks down (p.Kp <= 0) or stalls — the direct-solve semantics (Simulation.cpp:1431-1440) are the refinement's, not the inner method's.
// kdone / kcap count operator applications in pairs like BiCGSTAB iterations (two CG iterations = one); Ret32::iters returns the CG iterations of the solve (dc_bwd_stats::cg_iters).
template <int THREADS, bool WIN>
__device__ DC_OUTLINED Ret32 cg32_solve(const DevSystem &S, AdjCtx C, Krylov32 V, double in_stop, int kcap, int stall_window,
                                        double *red, int kdone) {
  const int N = S.N, tid = threadIdx.x;
  constexpr int VB = 4;
  __shared__ double redc[6 * (THREADS / 64)];      // (krylov_sum: a slot per reduction of the iteration)
  float *gin = V.rhs, *u = V.d, *r = V.r, *p = V.p, *v = V.v;
  float part = 0.f, part2 = 0.f, d1, d2;
  for (int i = tid; i < N; i += THREADS) {
    const f3 q = ld3(gin, i, N);
    const f3 z = q * S.dinv[i];
    st3(r, i, N, q); st3(p, i, N, z); st3(u, i, N, mk(0, 0, 0));
    part += dot(q, z); part2 += dot(q, q);
  }
  double rz = (double) part, rr = (double) part2;
  block_sum2<THREADS>(rz, rr, red);
  double best_rr = rr;
  int since_progress = 0, its = 0;
  int in_status = (rr <= in_stop) ? 1 : 0;
  // CG on a slightly non-symmetric K either converges like on its symmetric part or stagnates (measured r06 on the bench workload: median 65
  // applications per step against BiCGSTAB's 76, but 1 % of the steps beyond 478 and up to the cap of 3 200 — a launch waits for its slowest
  // rollout: 80 ms per batch step instead of 12). Hence a short leash: no new minimum of |r| for kCgStall iterations, or kCgCycleCap iterations in one
  // solve, ends the solve unconverged (status 0) and the caller hands the rest of the step to BiCGSTAB; the correction reached so far is kept
  // if the fp64 residual says it helped.
  for (int k = 2 * kdone; k < 2 * kcap && in_status == 0 && its < kCgCycleCap; k++) {
    // v = K p ;  alpha = (r . z) / (p . v)
    adjoint_operator<THREADS, WIN, true>(S, C, p, false, v, p, d1, d2);
    const double pv = krylov_sum<THREADS>(d1, redc);
    if (!(pv > 1e-300)) { in_status = 2; break; }      // not a descent direction in K's "energy": this system is not for CG
    const float alpha = (float) (rz / pv);
    // d += alpha p ;  r -= alpha v ;  (r . D^-1 r), (r . r)
    float pa = 0.f, pb = 0.f;
    for (int i0 = tid; i0 < N; i0 += VB * THREADS) {
      f3 rq[VB], vq[VB], pq[VB], uq[VB];
      float dq[VB];
#pragma unroll
      for (int j = 0; j < VB; j++) {
        const int ic = min(i0 + j * THREADS, N - 1);
        rq[j] = ld3(r, ic, N); vq[j] = ld3(v, ic, N); pq[j] = ld3(p, ic, N); uq[j] = ld3(u, ic, N); dq[j] = S.dinv[ic];
      }
#pragma unroll
      for (int j = 0; j < VB; j++) {
        const int i = i0 + j * THREADS;
        const f3 rn = rq[j] - vq[j] * alpha;
        if (i < N) { st3(r, i, N, rn); st3(u, i, N, uq[j] + pq[j] * alpha); pa += dot(rn, rn) * dq[j]; pb += dot(rn, rn); }
      }
    }
    double rz_new;
    krylov_sum2<THREADS>(pa, pb, redc + 2 * (THREADS / 64), rz_new, rr);
    its++;
    if (rr <= in_stop) { in_status = 1; break; }
    if (rr < best_rr) { best_rr = rr; since_progress = 0; }
    else if (++since_progress >= min(stall_window, kCgStall)) { in_status = 2; break; }      // (|r| of CG is not monotone: a few iterations without a new minimum are normal)
    if (!(rr < 1e8 * best_rr)) { in_status = 2; break; }      // diverging (NaN-safe)
    const float beta = (float) (rz_new / rz);
    rz = rz_new;
    // p = D^-1 r + beta p
    for (int i0 = tid; i0 < N; i0 += VB * THREADS) {
      f3 rq[VB], pq[VB];
      float dq[VB];
#pragma unroll
      for (int j = 0; j < VB; j++) { const int ic = min(i0 + j * THREADS, N - 1); rq[j] = ld3(r, ic, N); pq[j] = ld3(p, ic, N); dq[j] = S.dinv[ic]; }
#pragma unroll
      for (int j = 0; j < VB; j++) {
        const int i = i0 + j * THREADS;
        if (i < N) st3(p, i, N, rq[j] * dq[j] + pq[j] * beta);
      }
    }
    __syncthreads();
  }
  __syncthreads();
  return Ret32{in_status, kdone + (its + 1) / 2, its, rr};      // (iters = the CG iterations of this solve)
}

// The contact vertices' y list (AdjCtx::ylist) in what the windows leave of the CU's LDS — for the 1024-thread kernels, which have their CU to
// themselves at 128 registers per lane whatever their LDS (4 waves per SIMD): the larger request costs no mesh a second workgroup per CU.
// Returns the bytes to add to the launch's dynamic LDS and sets A.ycap.
// `lds_limit` = what the device grants a workgroup (hipDeviceAttributeMaxSharedMemoryPerBlock), `static_lds` = the instance's own static LDS
// (hipFuncGetAttributes::sharedSizeBytes): both queried, not assumed — see adj_lds_budget below.
template <int THREADS>
static size_t ylist_room(const DevSystem &S, size_t lds, BwdArgs &A, size_t lds_limit, size_t static_lds) {
  A.ycap = 0; A.ybase = (int) (lds / 4);
  const size_t reserve = static_lds + 256;
  if (THREADS != 1024 || lds + reserve + 12 > lds_limit) return 0;
  const size_t room = lds_limit - reserve - lds;
  A.ycap = (int) std::min(room / 12, (size_t) S.N + 2 * (size_t) S.self_cap);
  return (size_t) A.ycap * 12;
}
// LDS limit of the current device and static LDS of a kernel instance, each queried once per (device, instance)
struct AdjLdsBudget { size_t limit, static_lds; bool ok; };
static AdjLdsBudget adj_lds_budget(const void *func, int dev) {
  AdjLdsBudget b{(size_t) 64 * 1024, (size_t) 4096, false};
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0) b.limit = (size_t) v;
  // (a workgroup may be granted the whole CU's LDS through hipFuncAttributeMaxDynamicSharedMemorySize: 160 KB on gfx950)
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && (size_t) v > b.limit) b.limit = (size_t) v;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, func) == hipSuccess) { b.static_lds = fa.sharedSizeBytes; b.ok = true; }
  return b;
}
// Configures the instance's dynamic LDS for this launch: windows (`lds`) + the y list in what is left. When the device refuses the larger
// request the launch goes without the list (ycap = 0) instead of failing without a diagnosis.
template <int THREADS>
static size_t adj_configure_lds(const void *func, const DevSystem &S, size_t lds, BwdArgs &Ay, size_t (&configured)[kMaxDevices], AdjLdsBudget (&budget)[kMaxDevices]) {
  int dev = 0;
  (void) hipGetDevice(&dev);
  const int slot = dev >= 0 && dev < kMaxDevices ? dev : 0;
  if (!budget[slot].ok || dev >= kMaxDevices) budget[slot] = adj_lds_budget(func, dev);
  const size_t base = lds;
  lds += ylist_room<THREADS>(S, lds, Ay, budget[slot].limit, budget[slot].static_lds);
  size_t &done = configured[slot];
  if (lds > done || dev >= kMaxDevices) {
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) == hipSuccess) done = lds;
    else {
      (void) hipGetLastError();
      Ay.ycap = 0; lds = base;
      if (lds > done) { if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) == hipSuccess) done = lds; }
    }
  }
  return lds;
}

// BLK: direct solve preconditioned with K's own 3 x 3 diagonal blocks (dc_adjprecond.h) instead of diag(P)^-1
template <int THREADS, bool WIN, bool DENSE, bool BLK, bool COARSE = false>
__global__ __launch_bounds__(THREADS) void k_adjoint_step(const DevSystem *__restrict__ Sp, DevWork W, BwdArgs A) {
  const DevSystem &S = *Sp;
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];      // element windows (S.win_lds_bytes)
  __shared__ double red[3 * (THREADS / 64)];
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
    A.x_prev -= A.slot_state; A.v_prev -= A.slot_state; A.v_new -= A.slot_state;
    A.self.pair -= A.slot_self; A.self.nrm -= A.slot_self; A.self.dvec -= A.slot_self; A.self.meta -= A.slot_meta; A.self.verts -= 2 * A.slot_self;
    if (A.d_param) A.d_param -= A.slot_param;
    A.x_fixed -= A.slot_xf; A.stats -= A.slot_stats;
    if (A.d_xfixed) A.d_xfixed -= A.slot_xf;
    if (A.ix) A.ix -= A.slot_ix;
    if (A.iv) A.iv -= A.slot_ix;
    A.is_start = (A.slot - step == A.start_at) ? 1 : 0;            // isStart: Simulation.cpp:3947
    A.inj_x = A.inj_f = A.inj_n = A.inj_sn = A.inj_sd = nullptr;      // (a record from outside is differentiated by a launch of its own)
    if (A.ys) A.ys -= A.slot_state;
  }
  AdjCtx C;
  C.lds = dyn_lds; C.lds_floats = WIN ? S.win_lds_bytes / 4 : 0;
  C.xnew = A.x_new + off; C.rec_f = A.rec_f + off; C.rec_n = A.rec_n + off;
  C.rec_prim = A.rec_prim + (size_t) b * N;
  C.mu = A.mu + (size_t) b * S.ngroups;
  C.y = W.vbest + off; C.corner = W.corner + (size_t) b * 3 * S.NC;
  C.self = A.self; C.b = b;
  C.nself = (S.contact_enabled && S.self_enabled) ? A.self.meta[(size_t) b * kMetaStride] : 0;
  // the contact vertices of this step (the detection's cell / order arrays are free during the backward sweep)
  C.mark = A.dense_y ? nullptr : W.sd_cell + (size_t) b * N;
  C.plist = nullptr; C.nplist = 0; C.ylist = nullptr; C.nself_verts = 0;
  if (C.mark) {
    int *plist = W.sd_order + (size_t) b * N;
    __shared__ int nplist;
    if (tid == 0) nplist = 0;
    __syncthreads();
    for (int i = tid; i < N; i += THREADS) {
      const bool pc = C.rec_prim[i] >= 0;
      int m = 0;
      if (pc) { const int q = atomicAdd(&nplist, 1); plist[q] = i; m = 2 | (q << 2); }        // (order free: the vertices are independent)
      C.mark[i] = m;
    }
    __syncthreads();
    int M = 0;
    if (C.nself > 0) {
      M = A.self.meta[(size_t) b * kMetaStride + kMetaStride - 1];
      const int *verts = A.self.verts + (size_t) b * 2 * S.self_cap;
      // (plain read-modify-write: verts holds a vertex at most once — the contract stated in dc_selflib.h section 6 and next to self_JT_layers_lds_v)
      for (int q = tid; q < M; q += THREADS) { const int v = verts[q], m = C.mark[v]; C.mark[v] = (m & 2) ? (m | 1) : (1 | ((nplist + q) << 2)); }
    }
    C.plist = plist; C.nplist = nplist; C.nself_verts = M;
    C.ylist = (WIN && nplist + M <= A.ycap) ? dyn_lds + A.ybase : nullptr;
    __syncthreads();
  }
  float *gx = A.gx + off;
  float *gin = W.g + off, *u = W.vnow + off;
  float *cg_r = W.cg_r + off, *cg_p = W.cg_p + off, *cg_ap = W.cg_ap + off, *cg_x = W.cg_x + off;
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
  int status = gnorm > 0 ? 0 : 1;          // 1 converged (a zero gradient has the solution u = 0), 2 stalled at the fp32 floor, 0 cap hit
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
      adjoint_operator<THREADS, WIN, false>(S, C, u, false, cg_ap, nullptr, d1, d2);
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

  // fp64 side (dc_adjoint64.h): true residual of the mixed-precision refinement, fall-back solve, gradient assembly
  TeamOne<THREADS> tm{N, red};
  Adj64 C64;
  C64.xprev = A.x_prev + off; C64.vnew = A.v_new + off;
  C64.xnew = C.xnew; C64.rec_f = C.rec_f; C64.rec_n = C.rec_n; C64.mu = C.mu; C64.rec_prim = C.rec_prim;
  C64.self = C.self; C64.nself = C.nself; C64.b = b; C64.lds = dyn_lds; C64.lds_floats = C.lds_floats;
  adj64_inject(C64, A, b, N, S.self_cap);
  Work64 W64;
  W64.u = W.u64 + off; W64.r = W.r64 + off; W64.y = W.y64 + off; W64.x = W.x64 + off; W64.corner = W.c64 + (size_t) b * 3 * S.NC;
  W64.rhat = W.k64[0] + off; W64.p = W.k64[1] + off; W64.v = W.k64[2] + off; W64.t = W.k64[3] + off; W64.ph = W.k64[4] + off; W64.sh = W.k64[5] + off;
  int cycles = 0, iters64 = 0, verified = 1;
  // u64 = the result of the reference iteration (mode 0), or 0 — or, inside a fused sweep (A.warm), the previous step's solution: the
  // carried gradient and the step's record change smoothly from one time step to the next, so u*(k + 1) scaled by the factor that
  // minimises the residual along it is a far better start than 0 for the adjoint of step k (one more fp64 operator application)
  const bool warm = A.warm && A.mode == 1 && step > 0 && gnorm > 0;
  if (!warm) for (int i = tid; i < N; i += THREADS) st3d(W64.u, i, N, (A.mode == 0) ? tod(ld3(u, i, N)) : mkd(0, 0, 0));
  prepare_x64<THREADS>(S, C64, tm, W64.x);
  double rr_warm = -1;
  if (warm) {
    double a1 = 0, a2 = 0;      // w = K u_prev into W64.r; <w, g>, <w, w>
    apply_K64<THREADS>(S, C64, tm, W64.u, W64, [&](int i, d3 o) {
      st3d(W64.r, i, N, o);
      const d3 gi = tod(ld3(gx, i, N) * gscale);
      a1 += dot(o, gi); a2 += dot(o, o);
    });
    double s3[3];
    tm.sum3(a1, a2, 0, s3);
    const double gam = s3[1] > 0 ? s3[0] / s3[1] : 0.0;
    double a3 = 0;
    for (int i = tid; i < N; i += THREADS) {
      const d3 q = tod(ld3(gx, i, N) * gscale) - ld3d(W64.r, i, N) * gam;
      st3d(W64.r, i, N, q); st3d(W64.u, i, N, ld3d(W64.u, i, N) * gam);
      a3 += dot(q, q);
    }
    tm.sum3(a3, 0, 0, s3);
    rr_warm = s3[0];
    for (int i = tid; i < N; i += THREADS) st3(gin, i, N, tof(ld3d(W64.r, i, N)));
    __syncthreads();
  }

  if (need_direct && gnorm > 0) {
    // ---- direct solve of K u = g in mixed precision: block-Jacobi preconditioned BiCGSTAB in fp32 for corrections d of the
    //      residual r = g - K u evaluated in fp64 (u += d), until |r| <= rel_tol |g| in fp64; fp64 BiCGSTAB when that stalls ----
    used_direct = 1;
    PH(0)
    float *r = cg_r, *p = cg_p, *v = cg_ap, *t = cg_x, *rhat = W.sd_sx + off;    // (detection scratch, idle in the backward pass)
    float *ph = W.pre_p + off, *sh = W.pre_s + off, *minv = W.minv + (size_t) b * 9 * N;   // M^-1 p, M^-1 s, the block inverses
    auto build_blocks = [&]() {   // K's own 3 x 3 diagonal blocks at this step's x_new, inverted (dc_adjprecond.h)
      for (int i = tid; i < N; i += THREADS)
        store_block_inverse(elastic_diag_block(S, C.xnew, i), S.mass[i], [&](f3 e) { return contact_JT(S, C, i, e); }, minv, i, N);
    };
    if constexpr (BLK) build_blocks();
    const double stop = (double) A.rel_tol * (double) A.rel_tol * gnorm * gnorm;
    // (meshes the engine found ill-conditioned: where the fp64 fall-back of THIS instance has the coarse level — the condition of
    // bicgstab64, dc_adjoint64.h — the fp32 solve hands over after 400 iterations; everywhere else it keeps its budget, the fall-back
    // there is block-Jacobi only and much slower per digit: ADVICE r04)
    const bool fb_coarse = COARSE && S.defl_u != nullptr && S.adj_coarse && C.lds_floats >= kCoarseLdsFloats;
    const int kcap = std::min(A.it_cap > 0 ? 4 * A.it_cap : 1600, fb_coarse ? 400 : 1 << 30);
    // true residual of the start: g itself for u = 0 (mode 1), g - K u after the reference iteration hit its cap (mode 0);
    // `gin` holds the right-hand side of the next fp32 solve: g, later the fp64 residual rounded to fp32
    double rr_true = warm ? rr_warm : gnorm * gnorm;
    if (A.mode == 0) {
      rr_true = residual64<THREADS>(S, C64, tm, W64, gx, gscale).rr;
      for (int i = tid; i < N; i += THREADS) st3(gin, i, N, tof(ld3d(W64.r, i, N)));
      __syncthreads();
    }
    double rr = rr_true;
    double op_err = 0;         // measured error of the fp32 operator (see below)
    bool fallback = false;
    bool use_cg = !BLK && !COARSE && A.cg_first != 0;      // CG first; BiCGSTAB once a CG cycle has not contracted the fp64 residual (cg32_solve)
    status = (rr_true <= stop) ? 1 : 0;
    for (int kdone = 0; status == 0 && !fallback; cycles++) {
    // inner tolerance of this cycle: what is left to the target, but never below kInnerFloor of the cycle's own right-hand side
    // (an fp32 solve does not resolve more than that of an fp64 residual; measured on the prototype tests/proto_adjoint.py:
    // 1e-3 -> two cycles of 20 + 16 iterations where one solve to 1e-6 takes 41)
    const double rel_now = sqrt(rr_true) / gnorm;
    const double in_tol = A.fp32_only ? (double) A.rel_tol : fmax(0.3 * (double) A.rel_tol / rel_now, kInnerFloor);
    const double in_stop = in_tol * in_tol * rr_true;
    Krylov32 KV{gin, u, r, p, v, t, rhat, ph, sh, minv};
    const bool cg_cycle = use_cg;
    Ret32 r32;
    if (cg_cycle) { r32 = cg32_solve<THREADS, WIN>(S, C, KV, in_stop, kcap, A.stall_window, red, kdone); cg_total += r32.iters; r32.iters = iters; }
    // where the fp64 fall-back has the coarse level (ill-conditioned meshes: hat, dress-7742) an fp32 BiCGSTAB solve that has not found a new minimum of
    // |r| for kCoarseStall iterations hands over AT ONCE instead of running to its 400-iteration hand-over: on the pressed-on hat the fp32 stage then
    // ran 30 ... 185 instead of 400 iterations and the fp64 solve needed what it needed anyway (230 ... 430) — backward 59.7 -> 51.5 ms per batch step of
    // 64 rollouts, gradients unchanged to three digits (round 6, gpurun_out/r06_17)
    else r32 = bicgstab32_solve<THREADS, WIN, BLK, COARSE>(S, C, KV, in_stop, kcap, fb_coarse ? min(A.stall_window, kCoarseStall) : A.stall_window, red, kdone, iters);
    const int in_status = r32.status;
    kdone = r32.kdone; iters = r32.iters; rr = r32.rr;
    __syncthreads();
    PH(2)
    // u += d (fp64)
    for (int i = tid; i < N; i += THREADS) st3d(W64.u, i, N, ld3d(W64.u, i, N) + tod(ld3(u, i, N)));
    __syncthreads();
    if (A.fp32_only) {      // round-2 behaviour: the recurrence residual is all there is
      status = in_status;
      if (status == 2 && rr > 1e4 * stop) status = 0;
      rr_true = rr;
      cycles++;
      break;
    }
    // A correction solve works on a right-hand side that IS an fp64 residual: its recurrence residual estimates the residual of the
    // updated u up to the fp32 operator's error relative to that (small) right-hand side. That error is MEASURED whenever a solve is
    // followed by an fp64 evaluation (always after the first solve, whose right-hand side is g itself): op_err = | |r_true| - |r_rec| |
    // / |rhs|. A later solve that converged cleanly is accepted without another fp64 evaluation (0.74 ms of 14 per step on the
    // 10k-vertex workload) only when its recurrence residual PLUS twice that error applied to its own right-hand side is inside the
    // tolerance; the bound is what is then reported as last_udiff (dc_bwd_stats::residual_verified = 0).
    if (!A.verify_all && cycles >= 1 && in_status == 1) {
      const double bound = sqrt(rr) + 2.0 * op_err * sqrt(rr_true);
      if (bound * bound <= stop) { rr_true = bound * bound; status = 1; verified = 0; cycles++; break; }
    }
    // the true residual, in fp64
    double rr_new = residual64<THREADS>(S, C64, tm, W64, gx, gscale).rr;
    op_err = fmax(op_err, fabs(sqrt(rr_new) - sqrt(rr)) / sqrt(rr_true));
    PH(1)
    if (rr_new <= stop) { rr_true = rr_new; status = 1; cycles++; break; }
    // progress of this cycle: at least a factor 4 in the norm, else the fp32 solve is of no further use
    // (NaN-safe: a diverged correction fails the comparison and is taken back)
    const bool better = rr_new < rr_true;
    if (!better) {
      for (int i = tid; i < N; i += THREADS) st3d(W64.u, i, N, ld3d(W64.u, i, N) - tod(ld3(u, i, N)));
      __syncthreads();
      rr_new = residual64<THREADS>(S, C64, tm, W64, gx, gscale).rr;
    }
    if (!(rr_new < 0.0625 * rr_true) || in_status != 1 || cycles + 1 >= kMaxRefine || kdone >= kcap) {
      // a CG cycle that did not deliver: the remaining cycles of this step are BiCGSTAB's (the correction was kept only if it helped)
      if (cg_cycle && cycles + 1 < kMaxRefine && kdone < kcap) use_cg = false;
      else fallback = true;
    }
    rr_true = rr_new;
    if (!fallback) {
      for (int i = tid; i < N; i += THREADS) st3(gin, i, N, tof(ld3d(W64.r, i, N)));
      __syncthreads();
    }
    }   // cycle
    if (fallback) {
      // ---- fp64 BiCGSTAB on the same operator from (u, r): the reference's SparseLU always returns a solution ----
      if constexpr (!BLK) build_blocks();
      __syncthreads();
      // A system the fp32 solve cannot handle is ill-conditioned (cond(K) ~ 3e7 on the squashed 7 742-vertex dress: a relative
      // residual of 2e-8 still left the solution 2e-2 off the fp64 LU's): like a direct solve, go for the residual fp64 allows
      // (kFallbackGain below the caller's tolerance), and call it converged when the caller's tolerance holds.
      const double stop_fb = fmax(stop * kFallbackGain * kFallbackGain, 1e-26 * gnorm * gnorm);
      double rr64 = rr_true;
      for (int pass = 0; pass < 3; pass++) {
        const auto r64 = bicgstab64<THREADS, COARSE>(S, C64, tm, W64, minv, stop_fb, 20000, rr64, iters64);
        iters64 = r64.iters;
        rr64 = residual64<THREADS>(S, C64, tm, W64, gx, gscale).rr;     // the recurrence drifts over thousands of iterations: check, go again
        if (rr64 <= stop) status = 1;
        if (rr64 <= stop_fb || r64.res == 0) break;
      }
      rr_true = rr64;
    }
    udiff = sqrt(rr_true) / (gnorm > 0 ? gnorm : 1.0);     // relative residual: fp64-evaluated (mixed precision) or the fp32 recurrence's
  }
  __syncthreads();
  PH(2)
  // ---- gradients w.r.t. the previous state and parameters (Simulation.cpp:1534, 1608-1650), in fp64 from u ----
  finish_gradients64<THREADS>(S, C64, tm, W64, A, C.y);
  if (tid == 0) {
    dc_bwd_stats s;
    s.converged = status; s.adjoint_iters = iters; s.cg_iters = cg_total; s.clipped = clipped;
    s.used_direct = used_direct; s.last_udiff = (float) udiff;
    s.refine_cycles = cycles; s.fp64_iters = iters64; s.residual_verified = (used_direct && !A.fp32_only) ? verified : 0;
    s.workgroups = 1;
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
  BwdArgs Ay = A;
  static size_t configured[kMaxDevices] = {};        // the attribute is per device: one entry per device this process has used
  static AdjLdsBudget budget[kMaxDevices] = {};
  lds = adj_configure_lds<THREADS>((const void *) k_adjoint_step<THREADS, true, DENSE, BLK>, S, lds, Ay, configured, budget);
  hipLaunchKernelGGL((k_adjoint_step<THREADS, true, DENSE, BLK>), dim3(B), dim3(THREADS), lds, st, S.self_dev, W, Ay);
}
// the instances with the coarse level of the preconditioner (meshes the engine built a deflation space for, direct solve, block preconditioner):
// kernels of their own — inlined next to the plain solve the coarse code cost the headline's adjoint 4 % without ever running
template <int THREADS>
static void launch_adj_coarse(const DevSystem &S, const DevWork &W, const BwdArgs &A, int B, hipStream_t st) {
  size_t lds = std::max((size_t) S.win_lds_bytes, sizeof(float) * (size_t) kCoarseLdsFloats);
  BwdArgs Ay = A;
  static size_t configured[kMaxDevices] = {};
  static AdjLdsBudget budget[kMaxDevices] = {};
  lds = adj_configure_lds<THREADS>((const void *) k_adjoint_step<THREADS, true, false, true, true>, S, lds, Ay, configured, budget);
  hipLaunchKernelGGL((k_adjoint_step<THREADS, true, false, true, true>), dim3(B), dim3(THREADS), lds, st, S.self_dev, W, Ay);
}
template <int THREADS, bool DENSE>
static void launch_adj(const DevSystem &S, const DevWork &W, const BwdArgs &A, int B, hipStream_t st) {
  // the block preconditioner belongs to the direct solve (mode 1); the reference's iteration (mode 0) uses P^-1 as the reference does
  if (A.block_pre && A.mode == 1 && !DENSE) launch_adj_b<THREADS, DENSE, true>(S, W, A, B, st);
  else launch_adj_b<THREADS, DENSE, false>(S, W, A, B, st);
}

void launch_adjoint_step(const DevSystem &S, const DevWork &W, const BwdArgs &A, int B, hipStream_t st) {
#ifdef DC_ADJ_ONLY_BENCH      // development builds: only the instances of the 10 000-vertex headline (compile time)
  launch_adj_b<1024, false, false>(S, W, A, B, st);
#else
  // small meshes, reference iteration (mode 0): the inner solve with P is one product with the explicit inverse (dc_dense.h)
  if (S.dense_inv && S.win_ok && A.mode == 0 && pick_threads_bwd(S.N) == 1024) { launch_adj<1024, true>(S, W, A, B, st); return; }
  if (S.adj_coarse && S.defl_u && S.win_ok && S.win_lds_bytes / 4 >= kCoarseLdsFloats && A.block_pre && A.mode == 1 && pick_threads_bwd(S.N) == 1024) { launch_adj_coarse<1024>(S, W, A, B, st); return; }
  switch (pick_threads_bwd(S.N)) {
    case 256: launch_adj<256, false>(S, W, A, B, st); break;
    case 512: launch_adj<512, false>(S, W, A, B, st); break;
    default: launch_adj<1024, false>(S, W, A, B, st); break;
  }
#endif
}

}  // namespace dc
