// fp64 side of the adjoint step (Simulation::stepBackward, reference Simulation.cpp:1455-1780): the reference solves
// (P - dP^T) u = g in fp64 (SimplicialLLT / SparseLU, :1431-1440, :1561-1600); the fp32 Krylov solve of dc_adjoint.hip alone sits
// at eps_fp32 * cond(K) in the operator's coefficients (1-3e-4 relative on the 10k-vertex cloth and the hat, no answer at all on a
// compressed fine garment). This header holds what lifts it to the reference's accuracy and robustness:
//   * K u evaluated matrix-free in fp64 from the fp64 rest-shape tables (DevSystem::*64) — the true residual g - K u of the
//     mixed-precision refinement (fp32 BiCGSTAB solves for corrections, dc_adjoint.hip / dc_adjoint_cl.hip);
//   * a block-Jacobi preconditioned BiCGSTAB in fp64 on the same operator — the fall-back when the fp32 solve makes no progress
//     (adjoint systems beyond fp32: cond(K) ~ 3e7 on the squashed 7 742-vertex dress), standing in for SparseLU's "always
//     returns a solution";
//   * the state / parameter gradients of the step (Simulation.cpp:1534, 1608-1650, 1672-1764) accumulated in fp64.
// Everything is written once over a Team: TeamOne (one workgroup owns the rollout, dc_adjoint.hip) or TeamParts (K workgroups,
// part p owns rows [r0, r1), dc_adjoint_cl.hip); a Team supplies the row range, barriers, sums and the access path to the one
// vector that crosses parts (y = (I + dr_df)^T z).
// The element pass is element-centred with an fp64 corner array (W.corner; see element_pass64).
#pragma once
#include "dc_devlib.h"
#include "dc_adjprecond.h"

// Inlined by default: out of line (-DDC_ADJ_OUTLINE, A/B builds) the fp32 correction solve pays the call ABI — measured r03a on the
// 10k-vertex workload: 45.8 ms per fwd+bwd batch step out of line against 35.6 ms inlined.
#ifdef DC_ADJ_OUTLINE
#define DC_OUTLINED __attribute__((noinline))
#else
#define DC_OUTLINED __forceinline__
#endif

namespace dc {

// The leash of the CG correction solves (dc_adjoint.hip: cg32_solve; the split kernel's CG branch): a solve ends unconverged after this many iterations
// without a new minimum of |r|, or this many iterations in all — the rest of the step is then BiCGSTAB's.
constexpr int kCgStall = 10, kCgCycleCap = 64;
// fp32 BiCGSTAB solves of the instances whose fp64 fall-back has the coarse level: iterations without a new minimum of |r| before the hand-over
constexpr int kCoarseStall = 30;

struct d3 {
  double x, y, z;
};
__device__ __forceinline__ d3 mkd(double x, double y, double z) { return {x, y, z}; }
__device__ __forceinline__ d3 tod(f3 a) { return {(double) a.x, (double) a.y, (double) a.z}; }
__device__ __forceinline__ f3 tof(d3 a) { return {(float) a.x, (float) a.y, (float) a.z}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator*(d3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ d3 ld3d(const double *p, int i, int n) { return {p[i], p[n + i], p[2 * n + i]}; }
__device__ __forceinline__ void st3d(double *p, int i, int n, d3 v) { p[i] = v.x; p[n + i] = v.y; p[2 * n + i] = v.z; }

// fp64 work vectors of one rollout (planar [3][N] each)
struct Work64 {
  double *u, *r, *y;                        // solution, true residual g - K u, y = (I + dr_df)^T z (crosses parts)
  double *x;                                // x_new in fp64 (xnew64), filled once per backward step by prepare_x64 (crosses parts)
  double *corner;                           // [3][NC] per-constraint-corner results of the element pass (crosses parts)
  double *rhat, *p, *v, *t, *ph, *sh;       // fall-back BiCGSTAB
};

// what the fp64 operator reads of the step being differentiated
struct Adj64 {
  const float *xnew, *rec_f, *rec_n, *mu;
  const float *xprev, *vnew;                // state the step started from, velocity it ended with (tape slots k - 1 and k)
  const int *rec_prim;
  SelfRec self;
  int nself, b;
  float *lds;                               // LDS scratch of the layered self-contact pass
  int lds_floats;
  // record handed in from outside (dc_set_record), this rollout's share: x_new, f, primitive-contact normals as planar [3][N] doubles, the
  // self contacts' normals / d as [cap][3] doubles in the order of the contact list; null for a record the forward kernels made. The fp64
  // operator then works on these values (cases decided in fp64, as the reference does) instead of on the fp32 tape.
  const double *inj_x, *inj_f, *inj_n, *inj_sn, *inj_sd;
};
__device__ __forceinline__ void adj64_inject(Adj64 &C, const BwdArgs &A, int b, int N, int cap) {
  const size_t off = (size_t) b * 3 * N, so = (size_t) b * 3 * cap;
  C.inj_x = A.inj_x ? A.inj_x + off : nullptr; C.inj_f = A.inj_f ? A.inj_f + off : nullptr; C.inj_n = A.inj_n ? A.inj_n + off : nullptr;
  C.inj_sn = A.inj_sn ? A.inj_sn + so : nullptr; C.inj_sd = A.inj_sd ? A.inj_sd + so : nullptr;
}

// ---- one workgroup owns the rollout ----
template <int THREADS>
struct TeamOne {
  int N;
  double *red;                              // LDS [3 * THREADS / 64]
  __device__ __forceinline__ int r0() const { return 0; }
  __device__ __forceinline__ int r1() const { return N; }
  __device__ __forceinline__ bool leader() const { return true; }
  __device__ __forceinline__ int part() const { return 0; }
  __device__ __forceinline__ int parts() const { return 1; }
  __device__ __forceinline__ bool barrier() { __syncthreads(); return true; }
  __device__ __forceinline__ bool sum3(double a, double b, double c, double (&s)[3]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); c += __shfl_down(c, o, 64); }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    constexpr int NW = THREADS / 64;
    __syncthreads();
    if (l == 0) { red[w] = a; red[NW + w] = b; red[2 * NW + w] = c; }
    __syncthreads();
    double sa = 0, sb = 0, sc = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { sa += red[k]; sb += red[NW + k]; sc += red[2 * NW + k]; }
    s[0] = sa; s[1] = sb; s[2] = sc;
    return true;
  }
  // all-parts sums of n values every part holds in LDS (in place; `gather`: LDS for parts() x n values); ends with a barrier
  __device__ __forceinline__ bool allsum_lds(double *, int, double *) { __syncthreads(); return true; }
  // access to y (plain: the workgroup's own L1 / L2 path)
  struct YV {
    double *p;
    __device__ __forceinline__ double ld(int idx) const { return p[idx]; }
    __device__ __forceinline__ void st(int idx, double v) const { p[idx] = v; }
  };
  __device__ __forceinline__ YV yv(double *y) const { return YV{y}; }
  __device__ __forceinline__ YV yv(double *y, int) const { return YV{y}; }      // (n = entries per plane; the split Team sizes its buffer resource with it)
};

template <class YV> __device__ __forceinline__ d3 ld3y(const YV &a, int i, int n) { return mkd(a.ld(i), a.ld(n + i), a.ld(2 * n + i)); }
template <class YV> __device__ __forceinline__ void st3y(const YV &a, int i, int n, d3 v) { a.st(i, v.x); a.st(n + i, v.y); a.st(2 * n + i, v.z); }

// 1 / sqrt(x) and 1 / x in fp64 from the hardware estimates (v_rsq_f64, v_rcp_f64) + two Newton steps (relative error < 1e-15 for
// normal x): the IEEE expansions of sqrt() and '/' cost ~40 instructions each, and the fp64 operator needs three per triangle.
__device__ __forceinline__ double rsqrt_d(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0);
  y = fma(y * fma(0.375, e, 0.5), e, y);
  e = fma(-x * y, y, 1.0);
  return fma(0.5 * y, e, y);
}
__device__ __forceinline__ double rcp_d(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  return fma(r, fma(-x, r, 1.0), r);
}

// ---- closed forms in fp64 (same formulas as dc_devlib.h / dc_winlib.h) ----
struct PolarD {
  d3 t0, t1;
  double i00, i01, i11, trS;
};
__device__ __forceinline__ PolarD polar3x2d(d3 f0, d3 f1) {
  const double a = dot(f0, f0), b = dot(f0, f1), c = dot(f1, f1);
  const double det = fmax(a * c - b * b, 1e-300);
  const double s = det * rsqrt_d(det), tt = a + c + 2.0 * s, t = tt * rsqrt_d(tt), inv = rcp_d(t * s);
  PolarD P;
  P.i00 = (c + s) * inv; P.i01 = -b * inv; P.i11 = (a + s) * inv; P.trS = t;
  P.t0 = f0 * P.i00 + f1 * P.i01;
  P.t1 = f0 * P.i01 + f1 * P.i11;
  return P;
}
// J^T u of one contact (Simulation::calculatedri_dfi, Simulation.cpp:881-919). The CASE (take-off / stick / slide) is decided by the
// fp32 arithmetic of dri_dfi_T on the fp32 record, so that the fp64 operator and the fp32 operator it refines are the same matrix
// up to rounding; the formula is then evaluated in fp64.
__device__ __forceinline__ d3 dri_dfi_T_d(f3 nf, f3 df, float mu, d3 u) {
  const float sdf = dot(df, nf);
  if (sdf >= 0.f) return mkd(0, 0, 0);
  const f3 dTf = df - nf * sdf;
  const float nTf = sqrtf(dot(dTf, dTf));
  if (nTf <= mu * fabsf(sdf)) return mkd(0, 0, 0) - u;
  const d3 n = tod(nf), d = tod(df);
  const double sd = dot(d, n);
  const d3 dT = d - n * sd;
  const double nT = sqrt(dot(dT, dT));
  const d3 a = dT * (1.0 / nT);
  d3 q = u - a * dot(a, u);
  q = q - n * dot(n, q);
  d3 w = n * (-dot(n, u));
  w = w + (q * (sd / nT) + n * dot(a, u)) * (double) mu;
  return w;
}
// the same with the contact's n and d given in fp64 (a record handed in from outside, dc_set_record): case decided in fp64
__device__ __forceinline__ d3 dri_dfi_T_dd(d3 n, d3 d, double mu, d3 u) {
  const double sd = dot(d, n);
  if (sd >= 0.0) return mkd(0, 0, 0);
  const d3 dT = d - n * sd;
  const double nT = sqrt(dot(dT, dT));
  if (nT <= mu * fabs(sd)) return mkd(0, 0, 0) - u;
  const d3 a = dT * (1.0 / nT);
  d3 q = u - a * dot(a, u);
  q = q - n * dot(n, q);
  d3 w = n * (-dot(n, u));
  w = w + (q * (sd / nT) + n * dot(a, u)) * mu;
  return w;
}
__device__ __forceinline__ d3 dri_dmu_dd(d3 n, d3 d, double mu) {
  const double sd = dot(d, n);
  if (sd >= 0.0) return mkd(0, 0, 0);
  const d3 dT = d - n * sd;
  const double nT = sqrt(dot(dT, dT));
  if (!(nT > mu * fabs(sd))) return mkd(0, 0, 0);
  return dT * (-fabs(sd) / nT);
}
__device__ __forceinline__ d3 prim_vout_d(const DevPrim &p, d3 n) {
  return p.rotates ? cross(mkd(0, 1, 0), n) * 8.0 : mkd(0, 0, 0);   // Primitive.cpp:254-257
}
// dr/dmu (Simulation::calculatedri_dmu, Simulation.cpp:865-879), case decided in fp32 like above
__device__ __forceinline__ d3 dri_dmu_d(f3 nf, f3 df, float mu) {
  const float sdf = dot(df, nf);
  if (sdf >= 0.f) return mkd(0, 0, 0);
  const f3 dTf = df - nf * sdf;
  const float nTf = sqrtf(dot(dTf, dTf));
  if (!(nTf > mu * fabsf(sdf))) return mkd(0, 0, 0);
  const d3 n = tod(nf), d = tod(df);
  const double sd = dot(d, n);
  const d3 dT = d - n * sd;
  return dT * (-fabs(sd) / sqrt(dot(dT, dT)));
}
// w = dr_df^T z of the vertex's primitive contact (Simulation::calculatedr_df, Simulation.cpp:700-711)
__device__ __forceinline__ d3 contact_JT_d(const DevSystem &S, const Adj64 &C, int i, d3 z) {
  const int prim = C.rec_prim[i];
  if (prim < 0) return mkd(0, 0, 0);
  const int N = S.N;
  if (C.inj_f) {
    const d3 n = ld3d(C.inj_n, i, N);
    return dri_dfi_T_dd(n, ld3d(C.inj_f, i, N) - prim_vout_d(S.prims[prim], n) * S.mass64[i], (double) C.mu[S.prims[prim].group], z);
  }
  const f3 n = ld3(C.rec_n, i, N);
  const f3 d = ld3(C.rec_f, i, N) - prim_vout(S.prims[prim], n) * S.mass[i];     // the fp32 record of the forward pass
  return dri_dfi_T_d(n, d, C.mu[S.prims[prim].group], z);
}

// x_new of vertex i in fp64. The tape holds fl32(x_prev + h v_new) (the forward kernels' final update): positions of magnitude 2..8
// carry 1.2e-7..4.8e-7 of storage rounding, which alone moves the gradient of a stiff scene by 2e-5..6e-5 (measured on the hat with
// the fp64 oracle's own record rounded to fp32, tests/analyze_dump.py) — the unrounded sum is rebuilt from its fp32 terms. A state
// that is not that sum (a step that hit the iteration cap and reverted, Simulation.cpp:1357-1367) is taken as stored.
__device__ __forceinline__ d3 xnew64(const DevSystem &S, const Adj64 &C, int i) {
  const int N = S.N;
  if (C.inj_x) return ld3d(C.inj_x, i, N);
  const d3 xs = tod(ld3(C.xnew, i, N));
  const d3 xh = tod(ld3(C.xprev, i, N)) + tod(ld3(C.vnew, i, N)) * S.h64;
  auto pick = [](double s, double h) { return fabs(h - s) <= 2.4e-7 * fmax(fabs(s), 1e-3) ? h : s; };
  return mkd(pick(xs.x, xh.x), pick(xs.y, xh.y), pick(xs.z, xh.z));
}

// ---- layered self contacts, transposed (calculatedr_df, Simulation.cpp:713-760): z <- (I + J_0)^T ... (I + J_L)^T z in fp64 ----
// One workgroup (the Team's leader) runs it; `z` is the rollout's y through the Team's access path. In LDS when the working set
// fits (one wave walks the layers, see self_JT_layers_lds_v in dc_devlib.h), else through global memory with a barrier per layer.
template <int THREADS, class YV>
__device__ __forceinline__ void self_JT_layers_d(const DevSystem &S, const Adj64 &C, const YV &z) {
  const int cap = S.self_cap, N = S.N, tid = threadIdx.x, b = C.b;
  const SelfRec &R = C.self;
  const int *meta = R.meta + (size_t) b * kMetaStride;
  const int Cn = min(meta[0], cap), nl = meta[1], M = meta[kMetaStride - 1];
  const int2 *pair = R.pair + (size_t) b * cap;
  const float4 *nrm = R.nrm + (size_t) b * cap;
  const float4 *dvec = R.dvec + (size_t) b * cap;
  const int need = 8 * M + 8 * Cn + nl + 8;      // floats: 3 M doubles z, M doubles 1/m, 2 C float4, nl + 1 offsets
  if (C.inj_sn) {      // record from outside: the contacts' n and d in fp64, through global memory
    __syncthreads();
    for (int l = nl - 1; l >= 0; l--) {
      const int k1 = meta[2 + l + 1];
      for (int k = meta[2 + l] + tid; k < k1; k += THREADS) {
        const int2 ab = pair[k];
        const d3 n = mkd(C.inj_sn[3 * k], C.inj_sn[3 * k + 1], C.inj_sn[3 * k + 2]), d = mkd(C.inj_sd[3 * k], C.inj_sd[3 * k + 1], C.inj_sd[3 * k + 2]);
        const double mA = S.mass64[ab.x], mB = S.mass64[ab.y];
        d3 zA = ld3y(z, ab.x, N), zB = ld3y(z, ab.y, N);
        d3 g = dri_dfi_T_dd(n, d, (double) kClothMu, zA - zB) * ((mA * mB) / (mA + mB));
        st3y(z, ab.x, N, zA + g * (1.0 / mA));
        st3y(z, ab.y, N, zB - g * (1.0 / mB));
      }
      __syncthreads();
    }
    return;
  }
  if (S.self_lds && need <= C.lds_floats) {
    const int *verts = R.verts + (size_t) b * 2 * cap;
    double *lz = (double *) C.lds, *lim = lz + 3 * M;
    float4 *ln = (float4 *) (lim + M + (M & 1));       // 16-byte aligned
    float4 *ld = ln + Cn;
    int *loff = (int *) (ld + Cn);
    __syncthreads();
    for (int s = tid; s < M; s += THREADS) {
      const int v = verts[s];
      lz[s] = z.ld(v); lz[M + s] = z.ld(N + v); lz[2 * M + s] = z.ld(2 * N + v);
      lim[s] = 1.0 / S.mass64[v];
    }
    for (int k = tid; k < Cn; k += THREADS) { ln[k] = nrm[k]; ld[k] = dvec[k]; }
    for (int l = tid; l <= nl; l += THREADS) loff[l] = meta[2 + l];
    __syncthreads();
    auto contact = [&](int k) {
      const float4 n4 = ln[k], d4 = ld[k];
      const int sl = __float_as_int(n4.w), sa = sl & 0xffff, sb = sl >> 16;
      const double iA = lim[sa], iB = lim[sb];
      d3 zA = mkd(lz[sa], lz[M + sa], lz[2 * M + sa]), zB = mkd(lz[sb], lz[M + sb], lz[2 * M + sb]);
      d3 g = dri_dfi_T_d(mk(n4.x, n4.y, n4.z), mk(d4.x, d4.y, d4.z), kClothMu, zA - zB) * (1.0 / (iA + iB));
      zA = zA + g * iA; zB = zB - g * iB;
      lz[sa] = zA.x; lz[M + sa] = zA.y; lz[2 * M + sa] = zA.z;
      lz[sb] = zB.x; lz[M + sb] = zB.y; lz[2 * M + sb] = zB.z;
    };
    if (nl <= kWideLayers) {
      for (int l = nl - 1; l >= 0; l--) {
        const int k1 = loff[l + 1];
        for (int k = loff[l] + tid; k < k1; k += THREADS) contact(k);
        __syncthreads();
      }
    } else {
      if (tid < 64) {
        for (int l = nl - 1; l >= 0; l--) {
          const int k1 = loff[l + 1];
          for (int k = loff[l] + tid; k < k1; k += 64) contact(k);
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
      }
      __syncthreads();
    }
    for (int s = tid; s < M; s += THREADS) { const int v = verts[s]; z.st(v, lz[s]); z.st(N + v, lz[M + s]); z.st(2 * N + v, lz[2 * M + s]); }
    __syncthreads();
    return;
  }
  __syncthreads();
  for (int l = nl - 1; l >= 0; l--) {
    const int k1 = meta[2 + l + 1];
    for (int k = meta[2 + l] + tid; k < k1; k += THREADS) {
      const int2 ab = pair[k];
      const float4 n4 = nrm[k], d4 = dvec[k];
      const double mA = S.mass64[ab.x], mB = S.mass64[ab.y];
      d3 zA = ld3y(z, ab.x, N), zB = ld3y(z, ab.y, N);
      d3 g = dri_dfi_T_d(mk(n4.x, n4.y, n4.z), mk(d4.x, d4.y, d4.z), kClothMu, zA - zB) * ((mA * mB) / (mA + mB));
      st3y(z, ab.x, N, zA + g * (1.0 / mA));
      st3y(z, ab.y, N, zB - g * (1.0 / mB));
    }
    __syncthreads();
  }
}

// y = (I + dr_df)^T z on the Team's rows (z: own rows, plain), left in W.y through the Team's access path: self layers L..0 first,
// the block-diagonal primitive part last (calculatedr_df, Simulation.cpp:686-768). Ends with a Team barrier: y is complete.
template <int THREADS, class Team>
__device__ __forceinline__ bool form_y64(const DevSystem &S, const Adj64 &C, Team &tm, const double *z, double *y) {
  const int N = S.N;
  const auto Y = tm.yv(y);
  if (C.nself > 0) {
    for (int i = tm.r0() + threadIdx.x; i < tm.r1(); i += THREADS) st3y(Y, i, N, ld3d(z, i, N));
    if (!tm.barrier()) return false;
    if (tm.leader()) self_JT_layers_d<THREADS>(S, C, Y);
    if (!tm.barrier()) return false;
    for (int i = tm.r0() + threadIdx.x; i < tm.r1(); i += THREADS) { const d3 q = ld3y(Y, i, N); st3y(Y, i, N, q + contact_JT_d(S, C, i, q)); }
  } else {
    for (int i = tm.r0() + threadIdx.x; i < tm.r1(); i += THREADS) { const d3 q = ld3d(z, i, N); st3y(Y, i, N, q + contact_JT_d(S, C, i, q)); }
  }
  return tm.barrier();
}

// x_new in fp64 on the Team's rows (once per backward step); ends with a Team barrier
template <int THREADS, class Team>
__device__ __forceinline__ bool prepare_x64(const DevSystem &S, const Adj64 &C, Team &tm, double *x) {
  const auto X = tm.yv(x);
  for (int i = tm.r0() + threadIdx.x; i < tm.r1(); i += THREADS) st3y(X, i, S.N, xnew64(S, C, i));
  return tm.barrier();
}

// Element pass of the fp64 operator: h^2 w^2 [(A - dp/dx)^T A y] of every element of [t0, t1) x [e0, e1), one thread per element, written
// per constraint corner (Triangle::projectToManifoldBackward Triangle.cpp:354-451 in closed form, TriangleBending::backwardGradient
// TriangleBending.cpp:154-172); y and x_new (fp64, prepare_x64) are read, the corners written through the Team's access path.
// (Element-centred with a corner array: the vertex-centred gather that re-evaluated every element at each of its 3 - 4 vertices
// was bound by the fp64 rate — 2.1 M cycles per application at N = 10 000, measured with the in-kernel phase timers, r03h.)
template <int THREADS, class YV>
__device__ __forceinline__ void element_pass64(const DevSystem &S, const YV &X, const YV &Y, const YV &CV, int t0, int t1, int e0, int e1) {
  const int N = S.N, T = S.T, E = S.E, NC = S.NC;
  const double h2 = S.h64 * S.h64;
  for (int t = t0 + threadIdx.x; t < t1; t += THREADS) {
    const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
    const double Dx = S.tri_D64[t], Dy = S.tri_D64[T + t], Dz = S.tri_D64[2 * T + t], Dw = S.tri_D64[3 * T + t];
    const d3 x0 = ld3y(X, i0, N);
    const d3 a0 = ld3y(X, i1, N) - x0, a1 = ld3y(X, i2, N) - x0;
    const PolarD P = polar3x2d(a0 * Dx + a1 * Dz, a0 * Dy + a1 * Dw);
    const d3 q0 = ld3y(Y, i0, N);
    const d3 d0 = ld3y(Y, i1, N) - q0, d1 = ld3y(Y, i2, N) - q0;
    const d3 y0 = d0 * Dx + d1 * Dz, y1 = d0 * Dy + d1 * Dw;
    const double c = (dot(P.t1, y0) - dot(P.t0, y1)) * rcp_d(P.trS);
    d3 z0 = y0 * P.i00 + y1 * P.i01, z1 = y0 * P.i01 + y1 * P.i11;
    z0 = z0 - P.t0 * dot(P.t0, z0) - P.t1 * dot(P.t1, z0);
    z1 = z1 - P.t0 * dot(P.t0, z1) - P.t1 * dot(P.t1, z1);
    const double s = h2 * S.tri_w2_64[t];
    const d3 r0 = (y0 - (P.t1 * c + z0)) * s, r1 = (y1 - (z1 - P.t0 * c)) * s;
    const d3 c1 = r0 * Dx + r1 * Dy, c2 = r0 * Dz + r1 * Dw;
    st3y(CV, t, NC, mkd(0, 0, 0) - c1 - c2); st3y(CV, T + t, NC, c1); st3y(CV, 2 * T + t, NC, c2);
  }
  for (int e = e0 + threadIdx.x; e < e1; e += THREADS) {
    const int i0 = S.bend_v[e], i1 = S.bend_v[E + e], i2 = S.bend_v[2 * E + e], i3 = S.bend_v[3 * E + e];
    const double w0 = S.bend_w64[e], w1 = S.bend_w64[E + e], w2 = S.bend_w64[2 * E + e], w3 = S.bend_w64[3 * E + e];
    const double nrest = S.bend_nw64[e], wsq = S.bend_nw64[E + e];
    const d3 q0 = ld3y(Y, i0, N);
    const d3 ey = (ld3y(Y, i1, N) - q0) * w1 + (ld3y(Y, i2, N) - q0) * w2 + (ld3y(Y, i3, N) - q0) * w3;
    d3 res = ey;
    if (nrest > 1e-6) {
      const d3 x0 = ld3y(X, i0, N);
      const d3 ev = (ld3y(X, i1, N) - x0) * w1 + (ld3y(X, i2, N) - x0) * w2 + (ld3y(X, i3, N) - x0) * w3;
      const double ien = rsqrt_d(dot(ev, ev));
      const d3 eh = ev * ien;
      res = ey - (ey - eh * dot(eh, ey)) * (nrest * ien);
    }
    res = res * (h2 * wsq);
    const int base = 3 * T;
    st3y(CV, base + e, NC, res * w0); st3y(CV, base + E + e, NC, res * w1); st3y(CV, base + 2 * E + e, NC, res * w2); st3y(CV, base + 3 * E + e, NC, res * w3);
  }
}

// out = K z on the Team's rows (fp64): K = M + h^2 (A - dp/dx)^T A (I + dr_df)^T. The elements are dealt to the Team's parts in contiguous
// ranges, then every own row sums its corners in the fixed order of the incidence list (deterministic, no atomics). vert(i, K z at
// vertex i) is called for every own row exactly once (store, accumulate dot products ...). Ends WITHOUT a barrier: the caller reduces
// next (which is also what keeps a part from overwriting y / the corners while another part still reads them).
template <int THREADS, class Team, class VertOp>
__device__ __forceinline__ bool apply_K64(const DevSystem &S, const Adj64 &C, Team &tm, const double *z, const Work64 &W, VertOp vert) {
  if (!form_y64<THREADS>(S, C, tm, z, W.y)) return false;
  const int N = S.N, NC = S.NC, K = tm.parts(), part = tm.part();
  const auto Y = tm.yv(W.y), X = tm.yv(W.x), CV = tm.yv(W.corner, NC);
  element_pass64<THREADS>(S, X, Y, CV, (int) ((long long) S.T * part / K), (int) ((long long) S.T * (part + 1) / K),
                          (int) ((long long) S.E * part / K), (int) ((long long) S.E * (part + 1) / K));
  if (!tm.barrier()) return false;
  const double hk = S.h64 * S.h64 * S.k_att64;
  for (int i = tm.r0() + threadIdx.x; i < tm.r1(); i += THREADS) {
    d3 o = ld3d(z, i, N) * S.mass64[i];
    const int k1 = S.inc_ptr[i + 1];
    for (int k = S.inc_ptr[i]; k < k1; k++) o = o + ld3y(CV, S.inc_idx[k], NC);
    if (S.att_of_vertex[i] >= 0) o = o + ld3y(Y, i, N) * hk;       // attachment: dp/dx = 0 (AttachmentSpring.cpp:35-37)
    vert(i, o);
  }
  return true;
}

__device__ __forceinline__ d3 block_pre_d(const float *__restrict__ minv, int i, int N, d3 r) {
  return mkd((double) minv[i] * r.x + (double) minv[N + i] * r.y + (double) minv[2 * N + i] * r.z,
             (double) minv[3 * N + i] * r.x + (double) minv[4 * N + i] * r.y + (double) minv[5 * N + i] * r.z,
             (double) minv[6 * N + i] * r.x + (double) minv[7 * N + i] * r.y + (double) minv[8 * N + i] * r.z);
}

// Preconditioner of the fall-back, dst = M^-1 src on the Team's rows: the inverted 3 x 3 diagonal blocks of K and — where the engine built a
// deflation space for the forward solve (irregular garments, dc_deflate.h: the 16 lowest eigenvectors U of D^-1/2 P D^-1/2) — an additive
// coarse correction over Z = D^-1/2 U per coordinate,  M^-1 = B^-1 + Z (Z^T P Z)^-1 Z^T : the smooth, mass-dominated modes that make P
// ill-conditioned on those meshes are nearly the same for K = P - dP^T (dP acts on strain, those modes carry almost none). Offline on the
// squashed dress-7742 step (scipy, fp64, 1e-7): 3 795 -> 624 BiCGSTAB iterations; the Galerkin operator of K itself instead of P's: 618.
// Z^T src (48 values) is a sum over the Team: block sums in LDS, then ONE exchange of 48 fp64 granules per part (Team::allsum_lds). It has to
// be fp64: the sums are differences of large terms and G multiplies them by 1 / lambda (1.6e4) — with fp32 partial sums the preconditioner
// changed from application to application by ~1e-3 and BiCGSTAB (not a flexible method) went astray (NaN after 3 557 iterations, measured).
// `lds`: kCoarseLdsFloats of scratch (the element windows' LDS, idle between operator applications).
// Wave-wide sum of a double through DPP row operations on its two words (VALU only; __shfl_down on a double is two ds_bpermute through the
// LDS crossbar per step — 576 of them per wave for the 48 coarse sums, 30 us per application with 16 waves on the CU, measured on the hat).
// Same lane pattern as xch_wave_sum (dc_cluster.h); the total is returned in every lane.
__device__ __forceinline__ double wave_sum_d(double v) {
#define DC_DPP_D(x, ctrl, rmask) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, rmask, 0xF, true), \
                                                  __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, rmask, 0xF, true))
  v += DC_DPP_D(v, 0xB1, 0xF);     // quad_perm [1,0,3,2]
  v += DC_DPP_D(v, 0x4E, 0xF);     // quad_perm [2,3,0,1]
  v += DC_DPP_D(v, 0x141, 0xF);    // row_half_mirror
  v += DC_DPP_D(v, 0x140, 0xF);    // row_mirror
  v += DC_DPP_D(v, 0x142, 0xA);    // row_bcast:15
  v += DC_DPP_D(v, 0x143, 0xC);    // row_bcast:31
#undef DC_DPP_D
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

constexpr int kCoarseVectors = 16;
constexpr int kCoarseLdsFloats = 2 * (16 * 3 * kCoarseVectors + 3 * kCoarseVectors);      // scratch of precondition64 in floats (16 = waves or parts, at most)
template <int THREADS, class Team>
__device__ __forceinline__ bool precondition64(const DevSystem &S, Team &tm, const float *__restrict__ minv, const double *src, double *dst, float *lds) {
  const int N = S.N, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int DK = kCoarseVectors, NV = 3 * DK, NW = THREADS / 64;
  double *red = (double *) lds;              // [NW][NV] wave sums; afterwards the Team's gather area [parts][NV]
  double *vals = red + 16 * NV;              // [NV] Z^T src, then c = G Z^T src
  const float4 DC_G *U4 = (const float4 DC_G *) S.defl_u;
  __syncthreads();                           // (the scratch is the element windows' LDS: whoever used it last is done)
  for (int j4 = 0; j4 < DK / 4; j4++) {
    double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tm.r0() + tid; i < tm.r1(); i += THREADS) {
      const float4 u = U4[(size_t) i * (DK / 4) + j4];
      const d3 q = ld3d(src, i, N) * (double) S.sq_dinv[i];
      acc[0] += u.x * q.x; acc[1] += u.x * q.y; acc[2] += u.x * q.z;
      acc[3] += u.y * q.x; acc[4] += u.y * q.y; acc[5] += u.y * q.z;
      acc[6] += u.z * q.x; acc[7] += u.z * q.y; acc[8] += u.z * q.z;
      acc[9] += u.w * q.x; acc[10] += u.w * q.y; acc[11] += u.w * q.z;
    }
#pragma unroll
    for (int m = 0; m < 12; m++) {
      const double v = wave_sum_d(acc[m]);
      if (lane == 0) red[wv * NV + j4 * 12 + m] = v;
    }
  }
  __syncthreads();
  if (tid < NV) {
    double v = 0;
    for (int w = 0; w < NW; w++) v += red[w * NV + tid];
    vals[tid] = v;                           // entry (vector j, coordinate c) at 3 j + c
  }
  if (!tm.allsum_lds(vals, NV, red)) return false;      // fp64 across the parts: Z^T src is a difference of large terms and G amplifies it by 1 / lambda
  double cj = 0;
  if (tid < NV) {
    const int j = tid / 3, c = tid - 3 * j;
    for (int l = 0; l < DK; l++) cj += (double) S.defl_g[j * DK + l] * vals[l * 3 + c];
  }
  __syncthreads();
  if (tid < NV) vals[tid] = cj;
  __syncthreads();
  for (int i = tm.r0() + tid; i < tm.r1(); i += THREADS) {
    d3 z = mkd(0, 0, 0);
    for (int j4 = 0; j4 < DK / 4; j4++) {
      const float4 u = U4[(size_t) i * (DK / 4) + j4];
      const double *c = vals + j4 * 12;
      z.x += u.x * c[0] + u.y * c[3] + u.z * c[6] + u.w * c[9];
      z.y += u.x * c[1] + u.y * c[4] + u.z * c[7] + u.w * c[10];
      z.z += u.x * c[2] + u.y * c[5] + u.z * c[8] + u.w * c[11];
    }
    st3d(dst, i, N, block_pre_d(minv, i, N, ld3d(src, i, N)) + z * (double) S.sq_dinv[i]);
  }
  __syncthreads();       // (the scratch goes back to the operator)
  return true;
}

// Fall-back: right-preconditioned BiCGSTAB in fp64 on K u = g, continuing from (W.u, W.r = g - K u, rr = |r|^2). Preconditioner:
// the inverted 3 x 3 diagonal blocks of K in `minv` (dc_adjprecond.h; fp32 storage, applied in fp64). A breakdown (rho, omega or
// rhat.v vanishing) restarts the recurrence from the current residual. Returns 1 when |r|^2 <= stop2 by the recurrence (the caller
// re-evaluates the true residual), 0 at the cap, -1 when an exchange of the Team failed. `rr` and `iters` are updated.
// (Structs travel BY VALUE into these non-inlined functions: an object whose address is passed to a call lives in scratch memory
// for the whole kernel, and every later read of one of its fields becomes a scratch load.)
template <class Team>
struct Ret64 {
  int res;            // 1 converged by the recurrence, 0 cap / breakdown, -1 an exchange of the Team failed
  int iters;
  double rr;
  Team tm;            // the Team's exchange state moves on inside the call
};
template <int THREADS, bool COARSE = false, class Team = void>
__device__ DC_OUTLINED Ret64<Team> bicgstab64(const DevSystem &S, Adj64 C, Team tm, Work64 W, const float *__restrict__ minv,
                                                            double stop2, int kcap, double rr, int iters) {
  const int N = S.N, tid = threadIdx.x;
  auto ret = [&](int res) { return Ret64<Team>{res, iters, rr, tm}; };
  double s3[3];
  double rho = rr;
  int restarts = 0;
  // two-level preconditioner (precondition64) in the COARSE instances of the kernels only: the code inside the plain kernels cost the
  // headline's adjoint 4 % whether it ran or not (scratch 352 -> 672 B per lane: these functions are inlined, see DC_OUTLINED)
  const bool coarse = COARSE && S.defl_u != nullptr && S.adj_coarse && C.lds_floats >= kCoarseLdsFloats;
  for (int i = tm.r0() + tid; i < tm.r1(); i += THREADS) {
    const d3 q = ld3d(W.r, i, N);
    st3d(W.rhat, i, N, q); st3d(W.p, i, N, q);
    if (!coarse) st3d(W.ph, i, N, block_pre_d(minv, i, N, q));
  }
  if constexpr (COARSE) { if (coarse && !precondition64<THREADS>(S, tm, minv, W.p, W.ph, C.lds)) return ret(-1); }
  if (!tm.barrier()) return ret(-1);
  for (int k = 0; k < kcap; k++) {
    if (rr <= stop2) return ret(1);
    bool restart = false;
    // v = K M^-1 p ; alpha = rho / (rhat . v)
    double a1 = 0;
    if (!apply_K64<THREADS>(S, C, tm, W.ph, W, [&](int i, d3 o) { st3d(W.v, i, N, o); a1 += dot(o, ld3d(W.rhat, i, N)); })) return ret(-1);
    if (!tm.sum3(a1, 0, 0, s3)) return ret(-1);
    const double rv = s3[0];
    double alpha = 0, omega = 0;
    if (!(fabs(rv) > 1e-300 * fmax(1.0, fabs(rho)))) restart = true;
    if (!restart) {
      alpha = rho / rv;
      // s = r - alpha v (in place), sh = M^-1 s
      double ss = 0;
      for (int i = tm.r0() + tid; i < tm.r1(); i += THREADS) {
        const d3 s = ld3d(W.r, i, N) - ld3d(W.v, i, N) * alpha;
        st3d(W.r, i, N, s);
        if (!coarse) st3d(W.sh, i, N, block_pre_d(minv, i, N, s));
        ss += dot(s, s);
      }
      if (!tm.sum3(ss, 0, 0, s3)) return ret(-1);
      if constexpr (COARSE) {
        if (coarse && s3[0] > stop2) {      // (the exchange of the sum above fenced sh in the plain case; here it is formed after it)
          if (!precondition64<THREADS>(S, tm, minv, W.r, W.sh, C.lds)) return ret(-1);
          if (!tm.barrier()) return ret(-1);
        }
      }
      iters++;
      if (s3[0] <= stop2) {
        for (int i = tm.r0() + tid; i < tm.r1(); i += THREADS) st3d(W.u, i, N, ld3d(W.u, i, N) + ld3d(W.ph, i, N) * alpha);
        rr = s3[0];
        return ret(1);
      }
      // t = K M^-1 s ; omega = (t . s) / (t . t)
      double b1 = 0, b2 = 0;
      if (!apply_K64<THREADS>(S, C, tm, W.sh, W, [&](int i, d3 o) { st3d(W.t, i, N, o); b1 += dot(o, ld3d(W.r, i, N)); b2 += dot(o, o); })) return ret(-1);
      if (!tm.sum3(b1, b2, 0, s3)) return ret(-1);
      omega = s3[1] > 1e-300 ? s3[0] / s3[1] : 0.0;
      // u += alpha M^-1 p + omega M^-1 s ; r = s - omega t ; rho_new = rhat . r
      double pa = 0, pb = 0;
      for (int i = tm.r0() + tid; i < tm.r1(); i += THREADS) {
        const d3 rn = ld3d(W.r, i, N) - ld3d(W.t, i, N) * omega;
        st3d(W.u, i, N, ld3d(W.u, i, N) + ld3d(W.ph, i, N) * alpha + ld3d(W.sh, i, N) * omega);
        st3d(W.r, i, N, rn);
        pa += dot(rn, ld3d(W.rhat, i, N)); pb += dot(rn, rn);
      }
      if (!tm.sum3(pa, pb, 0, s3)) return ret(-1);
      const double rho_new = s3[0];
      rr = s3[1];
      if (rr <= stop2) return ret(1);
      if (!(fabs(rho_new) > 1e-30 * rr) || !(fabs(omega) > 0.0) || !isfinite(rr)) restart = true;
      else {
        const double beta = (rho_new / rho) * (alpha / omega);
        rho = rho_new;
        for (int i = tm.r0() + tid; i < tm.r1(); i += THREADS) {
          const d3 pn = ld3d(W.r, i, N) + (ld3d(W.p, i, N) - ld3d(W.v, i, N) * omega) * beta;
          st3d(W.p, i, N, pn);
          if (!coarse) st3d(W.ph, i, N, block_pre_d(minv, i, N, pn));
        }
        if constexpr (COARSE) { if (coarse && !precondition64<THREADS>(S, tm, minv, W.p, W.ph, C.lds)) return ret(-1); }
      }
    }
    if (restart) {
      if (++restarts > 50 || !isfinite(rr)) return ret(0);
      for (int i = tm.r0() + tid; i < tm.r1(); i += THREADS) {
        const d3 q = ld3d(W.r, i, N);
        st3d(W.rhat, i, N, q); st3d(W.p, i, N, q);
        if (!coarse) st3d(W.ph, i, N, block_pre_d(minv, i, N, q));
      }
      if constexpr (COARSE) { if (coarse && !precondition64<THREADS>(S, tm, minv, W.p, W.ph, C.lds)) return ret(-1); }
      rho = rr;
    }
    if (!tm.barrier()) return ret(-1);
  }
  return ret(rr <= stop2 ? 1 : 0);
}

// r = g - K u on the Team's rows, g = gscale * gx (the clipped carried gradient, fp32); returns |r|^2 in Ret64::rr.
template <int THREADS, class Team>
__device__ DC_OUTLINED Ret64<Team> residual64(const DevSystem &S, Adj64 C, Team tm, Work64 W, const float *__restrict__ gx, float gscale) {
  const int N = S.N;
  double a = 0;
  if (!apply_K64<THREADS>(S, C, tm, W.u, W, [&](int i, d3 o) {
        const d3 q = tod(ld3(gx, i, N) * gscale) - o;
        st3d(W.r, i, N, q);
        a += dot(q, q);
      })) return Ret64<Team>{-1, 0, 0.0, tm};
  double s3[3];
  if (!tm.sum3(a, 0, 0, s3)) return Ret64<Team>{-1, 0, 0.0, tm};
  return Ret64<Team>{1, 0, s3[0], tm};
}

// Gradients of the step from the converged u (W.u), all in fp64 (Simulation.cpp:1534, 1608-1650, 1672-1764): dL_dx / dL_dv into the
// carried gx / gv (fp32 storage), dL_dxfixed, dL_dmu (accumulated), the per-step parameter terms d_param[0..6]
//   [0..2]  sum over the elements of one type of  y . A^T (p(x_new) - A x_new)   -> dL/dk_type = h^2 / k * sum
//   [3]     density term (:1672-1679, adddr_dd = false)
//   [4..6]  h^2 * sum_i y_i  (dL_dfext_vec summed, :1702-1764; the host applies the wind chain rule)
// and y = (I + dr_df)^T u as fp32 in y32 (dc_get_force_gradient reads it). Elements are dealt to the Team's parts in contiguous ranges.
template <int THREADS, class Team>
__device__ DC_OUTLINED Ret64<Team> finish_gradients64(const DevSystem &S, Adj64 C, Team tm, Work64 W, BwdArgs A, float *__restrict__ y32) {
  auto ret = [&](int res) { return Ret64<Team>{res, 0, 0.0, tm}; };
  const int N = S.N, tid = threadIdx.x, b = C.b;
  const size_t off = (size_t) b * 3 * N;
  float *gx = A.gx + off, *gv = A.gv + off;
  const double h = S.h64, h2 = h * h;
  if (!form_y64<THREADS>(S, C, tm, W.u, W.y)) return ret(-1);
  const auto Y = tm.yv(W.y), X = tm.yv(W.x);
  double pacc[7] = {0, 0, 0, 0, 0, 0, 0};
  if (A.d_param) {
    const int T = S.T, E = S.E, K = tm.parts(), part = tm.part();
    const int t0 = (int) ((long long) T * part / K), t1 = (int) ((long long) T * (part + 1) / K);
    for (int t = t0 + tid; t < t1; t += THREADS) {
      const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
      const double Dx = S.tri_D64[t], Dy = S.tri_D64[T + t], Dz = S.tri_D64[2 * T + t], Dw = S.tri_D64[3 * T + t];
      const d3 x0 = ld3y(X, i0, N);
      const d3 e0 = ld3y(X, i1, N) - x0, e1 = ld3y(X, i2, N) - x0;
      const d3 f0 = e0 * Dx + e1 * Dz, f1 = e0 * Dy + e1 * Dw;
      const PolarD P = polar3x2d(f0, f1);
      const double w2 = S.tri_w2_64[t];
      const d3 g0 = (P.t0 - f0) * w2, g1 = (P.t1 - f1) * w2;
      const d3 c1 = g0 * Dx + g1 * Dy, c2 = g0 * Dz + g1 * Dw;
      const d3 q0 = ld3y(Y, i0, N);
      pacc[0] += dot(c1, ld3y(Y, i1, N) - q0) + dot(c2, ld3y(Y, i2, N) - q0);
    }
    const int e0i = (int) ((long long) E * part / K), e1i = (int) ((long long) E * (part + 1) / K);
    for (int e = e0i + tid; e < e1i; e += THREADS) {
      const int i0 = S.bend_v[e], i1 = S.bend_v[E + e], i2 = S.bend_v[2 * E + e], i3 = S.bend_v[3 * E + e];
      const double w1 = S.bend_w64[E + e], w2 = S.bend_w64[2 * E + e], w3 = S.bend_w64[3 * E + e];
      const double nrest = S.bend_nw64[e], wsq = S.bend_nw64[E + e];
      const d3 x0 = ld3y(X, i0, N);
      const d3 ev = (ld3y(X, i1, N) - x0) * w1 + (ld3y(X, i2, N) - x0) * w2 + (ld3y(X, i3, N) - x0) * w3;
      d3 p = mkd(0, 0, 0);
      if (nrest > 1e-6) { const double n2 = dot(ev, ev); p = n2 > 0 ? ev * (nrest * rsqrt_d(n2)) : ev * nrest; }
      const d3 q0 = ld3y(Y, i0, N);
      const d3 ey = (ld3y(Y, i1, N) - q0) * w1 + (ld3y(Y, i2, N) - q0) * w2 + (ld3y(Y, i3, N) - q0) * w3;
      pacc[1] += dot((p - ev) * wsq, ey);
    }
  }
  double dmu_part[kMaxPrims];
#pragma unroll
  for (int k = 0; k < kMaxPrims; k++) dmu_part[k] = 0.0;
  float *dxf = A.d_xfixed ? A.d_xfixed + (size_t) b * 3 * S.Af : nullptr;
  const d3 grav = mkd(S.g64[0], S.g64[1], S.g64[2]);
  for (int i = tm.r0() + tid; i < tm.r1(); i += THREADS) {
    const d3 ui = ld3d(W.u, i, N), yi = ld3y(Y, i, N), w = yi - ui;
    const double m = S.mass64[i];
    if (A.d_param) {
      const int a = S.att_of_vertex[i];
      if (a >= 0) pacc[2] += S.k_att64 * dot(tod(ld3(A.x_fixed + (size_t) b * 3 * S.Af, a, S.Af)) - ld3y(X, i, N), yi);
      const double ar = m / S.density64;
      const d3 xp = tod(ld3(A.x_prev + off, i, N)), vp = tod(ld3(A.v_prev + off, i, N));
      pacc[3] += ar * (dot(ui, xp + vp * h + grav * h2 - ld3y(X, i, N)) + h * dot(w, vp + grav * h));
      pacc[4] += h2 * yi.x; pacc[5] += h2 * yi.y; pacc[6] += h2 * yi.z;
    }
    const int prim = C.rec_prim[i];
    if (prim >= 0) {
      const int grp = S.prims[prim].group;
      double contrib;
      if (C.inj_f) {
        const d3 n = ld3d(C.inj_n, i, N);
        contrib = dot(dri_dmu_dd(n, ld3d(C.inj_f, i, N) - prim_vout_d(S.prims[prim], n) * m, (double) C.mu[grp]), ui) * h;
      } else {
        const f3 n = ld3(C.rec_n, i, N);
        const f3 d = ld3(C.rec_f, i, N) - prim_vout(S.prims[prim], n) * S.mass[i];
        contrib = dot(dri_dmu_d(n, d, C.mu[grp]), ui) * h;
      }
#pragma unroll
      for (int k = 0; k < kMaxPrims; k++) dmu_part[k] += (k == grp) ? contrib : 0.0;
    }
    d3 dx = ui * m - tod(ld3(gv, i, N)) * (1.0 / h);
    d3 dv = yi * (h * m);
    if (A.ix) dx = dx + tod(ld3(A.ix + off, i, N));
    if (A.iv) dv = dv + tod(ld3(A.iv + off, i, N));
    if (!A.is_start) dx = dx + dv * (1.0 / h);
    st3(gx, i, N, tof(dx));
    st3(gv, i, N, tof(dv));
    st3(y32, i, N, tof(yi));
    if (A.ys) st3(A.ys + off, i, N, tof(yi));
    const int a = S.att_of_vertex[i];
    if (a >= 0 && dxf) st3(dxf, a, S.Af, tof(yi * (h2 * S.k_att64)));   // A_t_dp_dxfixed (Simulation.cpp:3035-3048)
  }
  double s3[3];
  const bool writer = tid == 0 && tm.leader();
  if (A.d_mu) {
    for (int k0 = 0; k0 < S.ngroups; k0 += 3) {
      double val[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        val[c] = 0.0;
#pragma unroll
        for (int k = 0; k < kMaxPrims; k++) val[c] += (k == k0 + c) ? dmu_part[k] : 0.0;
      }
      if (!tm.sum3(val[0], val[1], val[2], s3)) return ret(-1);
      if (writer)
        for (int c = 0; c < 3 && k0 + c < S.ngroups; c++) A.d_mu[(size_t) b * S.ngroups + k0 + c] += (float) s3[c];
    }
  }
  if (A.d_param) {
    float *dp = A.d_param + (size_t) b * 8;
    const double scale[9] = {S.k_stretch64 > 0 ? h2 / S.k_stretch64 : 0.0, S.k_bend64 > 0 ? h2 / S.k_bend64 : 0.0,
                             S.k_att64 > 0 ? h2 / S.k_att64 : 0.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0};
    for (int k0 = 0; k0 < 7; k0 += 3) {
      if (!tm.sum3(pacc[k0], k0 + 1 < 7 ? pacc[k0 + 1 < 7 ? k0 + 1 : 6] : 0.0, k0 + 2 < 7 ? pacc[k0 + 2 < 7 ? k0 + 2 : 6] : 0.0, s3)) return ret(-1);
      if (writer)
        for (int c = 0; c < 3 && k0 + c < 7; c++) dp[k0 + c] = (float) (s3[c] * scale[k0 + c]);
    }
  }
  return ret(1);
}

}  // namespace dc
