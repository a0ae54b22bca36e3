// CDNA4 (gfx950) kernels of the DiffCloth stepper.
//
// Execution model: ONE workgroup owns ONE rollout for a whole time step. All Projective-Dynamics
// iterations, the inner block-Jacobi PCG solves, the convergence test and the best-iterate tracking of
// Simulation::step() (reference Simulation.cpp:1043-1428) run inside a single launch, synchronised by
// workgroup barriers only — no host round trip, no grid-wide sync, per-rollout early exit for free.
// 256 CUs x (1..8 workgroups) rollouts are in flight at once; rollouts are the data-parallel axis.
//
// Reformulation used (same fixed point as the reference):
//   f      = [h^2 A^T (p(x) - A x) + M (s_n - x_n)] / h          (== b~ - C v_now of Simulation.cpp:1248-1249)
//   P dv   = f + r(f) - M v_now ,  v_new = v_now + dv             (== v_new = P^-1 (b~ + r), :1267)
// so the constraint residual p - A x is evaluated per element in fp32 without cancellation against P x_n,
// and the global solve is a PCG for the *correction*, warm-started for free.
#include "dc_device.h"

namespace dc {

// ---------------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------------
struct f3 {
  float x, y, z;
};
__device__ __forceinline__ f3 mk(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ f3 fma3(f3 a, float s, f3 acc) { return {fmaf(a.x, s, acc.x), fmaf(a.y, s, acc.y), fmaf(a.z, s, acc.z)}; }
__device__ __forceinline__ f3 ld3(const float *p, int i, int n) { return {p[i], p[n + i], p[2 * n + i]}; }
__device__ __forceinline__ void st3(float *p, int i, int n, f3 v) { p[i] = v.x; p[n + i] = v.y; p[2 * n + i] = v.z; }
__device__ __forceinline__ f3 normalized(f3 a) {
  float n2 = dot(a, a);
  return n2 > 0.f ? a * (1.0f / sqrtf(n2)) : a;    // Eigen::normalized(): unchanged when the norm is 0
}

// Sum over the whole workgroup; every thread receives the result. Partials are combined in fp64.
template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double *red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; k++) s += red[k];
  return s;
}

// Closest 3x2 isometry T = F S^-1, S = (F^T F)^(1/2) in closed form.  Equals Q * (U V^T) of
// Triangle::projectToManifold (Triangle.cpp:329-351): the Gram-Schmidt frame Q spans F's column space, so
// Q Q^T F S^-1 = F S^-1.
struct Polar {
  f3 t0, t1;
  float i00, i01, i11, trS;   // S^-1 (symmetric) and trace(S)
};
__device__ __forceinline__ Polar polar3x2(f3 f0, f3 f1) {
  float a = dot(f0, f0), b = dot(f0, f1), c = dot(f1, f1);
  float det = fmaxf(a * c - b * b, 1e-30f);
  float s = sqrtf(det);
  float t = sqrtf(a + c + 2.f * s);
  float inv = 1.0f / (t * s);
  Polar P;
  P.i00 = (c + s) * inv; P.i01 = -b * inv; P.i11 = (a + s) * inv; P.trS = t;
  P.t0 = f0 * P.i00 + f1 * P.i01;
  P.t1 = f0 * P.i01 + f1 * P.i11;
  return P;
}

// Signorini–Coulomb response r(d) of one contact: Simulation::calcualteDryFrictionForce (Simulation.cpp:829-862).
__device__ __forceinline__ f3 dry_friction(f3 n, f3 d, float mu) {
  float sd = dot(d, n);
  if (sd >= 0.f) return mk(0, 0, 0);                       // take-off
  f3 dN = n * sd, dT = d - dN;
  float nT = sqrtf(dot(dT, dT));
  f3 r = mk(0, 0, 0) - dN;
  if (nT <= mu * fabsf(sd)) return r - dT;                 // stick
  return r - dT * (mu * fabsf(sd) / nT);                   // slide
}
// w = J^T u with J = dr/dd of the same contact: Simulation::calculatedri_dfi (Simulation.cpp:881-919), applied
// matrix-free. take-off: 0; stick: -u; slide: J = -n n^T + mu (b (I - a a^T)/|dT| (I - n n^T) + a n^T).
__device__ __forceinline__ f3 dri_dfi_T(f3 n, f3 d, float mu, f3 u) {
  float sd = dot(d, n);
  if (sd >= 0.f) return mk(0, 0, 0);
  f3 dN = n * sd, dT = d - dN;
  float nT = sqrtf(dot(dT, dT));
  if (nT <= mu * fabsf(sd)) return mk(0, 0, 0) - u;
  f3 a = dT * (1.0f / nT);
  // J^T u = -n (n.u) + mu [ (I - n n^T) (I - a a^T) u * b/|dT| + n (a.u) ]
  f3 q = u - a * dot(a, u);
  q = q - n * dot(n, q);
  f3 w = n * (-dot(n, u));
  w = w + (q * (sd / nT) + n * dot(a, u)) * mu;
  return w;
}
// dr/dmu of the same contact: Simulation::calculatedri_dmu (Simulation.cpp:865-879).
__device__ __forceinline__ f3 dri_dmu(f3 n, f3 d, float mu) {
  float sd = dot(d, n);
  if (sd >= 0.f) return mk(0, 0, 0);
  f3 dT = d - n * sd;
  float nT = sqrtf(dot(dT, dT));
  if (nT > mu * fabsf(sd)) return dT * (-fabsf(sd) / nT);
  return mk(0, 0, 0);
}

// Primitive::isInContact family (Sphere Primitive.cpp:221-261, Capsule :570-604) for one flattened primitive.
__device__ __forceinline__ bool prim_in_contact(const DevPrim &p, f3 pos, f3 &normal) {
  f3 c = mk(p.cx, p.cy, p.cz);
  f3 q = pos - c;
  if (p.kind == DC_PRIM_SPHERE) {
    float dist = sqrtf(dot(q, q)) - p.radius;
    normal = normalized(q);
    return dist < 0.1f;
  }
  // capsule: Primitive.h:198-211 projectionOnLine + Primitive.cpp:583-603
  f3 top = mk(p.tx, p.ty, p.tz);
  float ab2 = dot(top, top);
  f3 pr = top * (dot(q, top) / ab2);
  float AB = sqrtf(ab2), AP = sqrtf(dot(pr, pr));
  f3 pb = pr - top;
  float PB = sqrtf(dot(pb, pb));
  float t = AP / AB;
  if (PB > AB) t = -t;
  float rl = p.radius / p.length;
  if ((t < 0.f - rl) || (t > 1.f + rl)) return false;
  float dist;
  if (t < 0.f) { dist = sqrtf(dot(q, q)) - p.radius; normal = normalized(q); }
  else if (t > 1.f) { f3 e = q - top; dist = sqrtf(dot(e, e)) - (p.radius + 0.1f); normal = normalized(e); }
  else { f3 e = q - pr; dist = sqrtf(dot(e, e)) - (p.radius + 0.1f); normal = normalized(e); }
  return dist < 0.1f;
}
// Simulation::isInContactWithObstacle (Simulation.cpp:153-191): t = 0, h/2, h; first primitive / first sample wins.
// Children of one LowerLeg share a group and are tested, per sample, in child order (Primitive.cpp:410-418).
__device__ __forceinline__ int detect_primitive(const DevSystem &S, f3 pos, f3 vel, f3 &normal) {
  int p0 = 0;
  while (p0 < S.nprim) {
    int p1 = p0;
    while (p1 < S.nprim && S.prims[p1].group == S.prims[p0].group) p1++;
    for (int k = 0; k < 3; k++) {
      f3 q = pos + vel * (S.h * 0.5f * (float) k);
      for (int p = p0; p < p1; p++)
        if (prim_in_contact(S.prims[p], q, normal)) return p;
    }
    p0 = p1;
  }
  return -1;
}
__device__ __forceinline__ f3 prim_vout(const DevPrim &p, f3 n) {
  return p.rotates ? cross(mk(0, 1, 0), n) * 8.0f : mk(0, 0, 0);   // Primitive.cpp:254-257, static primitives
}

// ---------------------------------------------------------------------------------------------------
// block-Jacobi PCG for P d = rhs (P = P_s (x) I3, so the 3x3 diagonal blocks are P_ii * I3).
// On entry: cg_r = rhs, cg_p = D^-1 rhs, cg_x = 0 and rz = rhs . D^-1 rhs (already reduced).
// Replaces SimplicialLLT::solve of Simulation.cpp:1267 / :1577.
// ---------------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ int block_pcg(const DevSystem &S, float *cg_r, float *cg_p, float *cg_ap, float *cg_x,
                                         double rz, float rel_tol, int max_iter, double *red) {
  const int N = S.N, tid = threadIdx.x;
  const double stop = (double) rel_tol * (double) rel_tol * rz;
  if (!(rz > 1e-300)) return 0;
  int it = 0;
  __syncthreads();
  for (; it < max_iter;) {
    float part = 0.f;
    for (int i = tid; i < N; i += THREADS) {
      float ax = 0.f, ay = 0.f, az = 0.f;
      const int k1 = S.P_ptr[i + 1];
      for (int k = S.P_ptr[i]; k < k1; k++) {
        const float a = S.P_val[k];
        const int j = S.P_col[k];
        ax = fmaf(a, cg_p[j], ax); ay = fmaf(a, cg_p[N + j], ay); az = fmaf(a, cg_p[2 * N + j], az);
      }
      cg_ap[i] = ax; cg_ap[N + i] = ay; cg_ap[2 * N + i] = az;
      part += cg_p[i] * ax + cg_p[N + i] * ay + cg_p[2 * N + i] * az;
    }
    const double pAp = block_sum<THREADS>((double) part, red);
    const float alpha = (float) (rz / pAp);
    part = 0.f;
    for (int i = tid; i < N; i += THREADS) {
      const float di = S.dinv[i];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int o = c * N + i;
        cg_x[o] = fmaf(alpha, cg_p[o], cg_x[o]);
        const float r = fmaf(-alpha, cg_ap[o], cg_r[o]);
        cg_r[o] = r;
        part = fmaf(r * di, r, part);
      }
    }
    const double rz_new = block_sum<THREADS>((double) part, red);
    it++;
    if (!(rz_new > stop)) break;
    const float beta = (float) (rz_new / rz);
    rz = rz_new;
    for (int i = tid; i < N; i += THREADS) {
      const float di = S.dinv[i];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int o = c * N + i;
        cg_p[o] = fmaf(beta, cg_p[o], cg_r[o] * di);
      }
    }
    __syncthreads();
  }
  return it;
}

// ---------------------------------------------------------------------------------------------------
// Forward: Simulation::step()
// ---------------------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_pd_step(DevSystem S, DevWork W, FwdArgs A) {
  __shared__ double red[THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = S.N, T = S.T, E = S.E, NC = S.NC;
  const size_t off = (size_t) b * 3 * N;
  const float *xn = A.x_in + off, *vn = A.v_in + off;
  float *g = W.g + off, *vnow = W.vnow + off, *vbest = W.vbest + off;
  float *cg_r = W.cg_r + off, *cg_p = W.cg_p + off, *cg_ap = W.cg_ap + off, *cg_x = W.cg_x + off;
  float *corner = W.corner + (size_t) b * 3 * NC;
  float *rec_f = A.rec_f + off, *rec_r = A.rec_r + off, *rec_n = A.rec_n + off;
  int *rec_prim = A.rec_prim + (size_t) b * N;
  const float *xfix = A.x_fixed + (size_t) b * 3 * S.Af;
  const float *mu = A.mu + (size_t) b * S.ngroups;
  const float h = S.h;
  const f3 grav = mk(S.gx, S.gy, S.gz);
  const f3 fu = A.fu ? mk(A.fu[3 * b], A.fu[3 * b + 1], A.fu[3 * b + 2]) : mk(0, 0, 0);

  // ---- step set-up: s_n, initial guess, contact detection (Simulation.cpp:1097-1160, :1254-1256) ----
  float part = 0.f;
  int ncontact = 0;
  for (int i = tid; i < N; i += THREADS) {
    const float m = S.mass[i];
    f3 v = ld3(vn, i, N);
    f3 fext = grav * m + fu;                      // fillForces (Simulation.cpp:55-116)
    f3 v0 = v + fext * (h / m);                   // (s_n - x_n) / h
    st3(vnow, i, N, v0);
    st3(g, i, N, v0 * m);                         // M (s_n - x_n) / h
    part += dot(v0, v0);
    int prim = -1;
    f3 nrm = mk(0, 0, 0);
    if (S.contact_enabled) prim = detect_primitive(S, ld3(xn, i, N), v0, nrm);
    rec_prim[i] = prim;
    st3(rec_n, i, N, nrm);
    ncontact += (prim >= 0);
  }
  double min_xdiff = (double) h * sqrt(block_sum<THREADS>((double) part, red)) / (double) N;
  const int total_contacts = (int) block_sum<THREADS>((double) ncontact, red);
  bool improved = false, converged = false, stalled = false;
  int iters = 0, cg_total = 0, since_progress = 0;
  double xdiff = 0;

  for (int iter = 0; iter < A.pd_cap; iter++) {
    // ---- local step: per-element projection residual, written per constraint corner ----
    // triangles: Triangle::project (Triangle.cpp:310-351); contribution h * w^2 * (T - F) D^T
    for (int t = tid; t < T; t += THREADS) {
      const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
      const float4 D = S.tri_D[t];
      f3 x0 = ld3(xn, i0, N), v0 = ld3(vnow, i0, N);
      // edges as (x_n differences) + h (v differences): exact fp32 differences, no cancellation error
      f3 e0 = (ld3(xn, i1, N) - x0) + (ld3(vnow, i1, N) - v0) * h;
      f3 e1 = (ld3(xn, i2, N) - x0) + (ld3(vnow, i2, N) - v0) * h;
      f3 f0 = e0 * D.x + e1 * D.z, f1 = e0 * D.y + e1 * D.w;
      Polar P = polar3x2(f0, f1);
      const float s = h * S.tri_w2[t];
      f3 g0 = (P.t0 - f0) * s, g1 = (P.t1 - f1) * s;
      f3 c1 = g0 * D.x + g1 * D.y, c2 = g0 * D.z + g1 * D.w;
      f3 c0 = mk(0, 0, 0) - c1 - c2;
      st3(corner, t, NC, c0); st3(corner, T + t, NC, c1); st3(corner, 2 * T + t, NC, c2);
    }
    // bending: TriangleBending::project (TriangleBending.cpp:138-151); contribution h * w^2 * w_i * (p - e)
    for (int e = tid; e < E; e += THREADS) {
      const int i0 = S.bend_v[e], i1 = S.bend_v[E + e], i2 = S.bend_v[2 * E + e], i3 = S.bend_v[3 * E + e];
      const float4 w = S.bend_w[e];
      const float2 nw = S.bend_nw[e];
      f3 x0 = ld3(xn, i0, N), v0 = ld3(vnow, i0, N);
      // sum_i w_i x_i with sum_i w_i = 0  ->  sum_{i>0} w_i (x_i - x_0)
      f3 ev = ((ld3(xn, i1, N) - x0) + (ld3(vnow, i1, N) - v0) * h) * w.y;
      ev = ev + ((ld3(xn, i2, N) - x0) + (ld3(vnow, i2, N) - v0) * h) * w.z;
      ev = ev + ((ld3(xn, i3, N) - x0) + (ld3(vnow, i3, N) - v0) * h) * w.w;
      f3 p = mk(0, 0, 0);
      if (nw.x > 1e-6f) p = normalized(ev) * nw.x;
      f3 d = (p - ev) * (h * nw.y);
      const int base = 3 * T;
      st3(corner, base + e, NC, d * w.x); st3(corner, base + E + e, NC, d * w.y);
      st3(corner, base + 2 * E + e, NC, d * w.z); st3(corner, base + 3 * E + e, NC, d * w.w);
    }
    __syncthreads();
    // ---- vertex pass: f, friction r, right-hand side of the correction solve ----
    part = 0.f;
    for (int i = tid; i < N; i += THREADS) {
      f3 f = ld3(g, i, N);
      const int k1 = S.inc_ptr[i + 1];
      for (int k = S.inc_ptr[i]; k < k1; k++) f = f + ld3(corner, S.inc_idx[k], NC);
      f3 v = ld3(vnow, i, N);
      const int a = S.att_of_vertex[i];
      if (a >= 0) {    // AttachmentSpring::project (AttachmentSpring.cpp:25-29): h * k_att * (x_fixed - x_i)
        // (x_fixed - x_n) is an exact fp32 difference; only then subtract the small h v term (k_att = 1e4 amplifies error)
        f = f + ((ld3(xfix, a, S.Af) - ld3(xn, i, N)) - v * h) * (h * S.k_att);
      }
      const float m = S.mass[i];
      f3 r = mk(0, 0, 0);
      const int prim = rec_prim[i];
      if (prim >= 0) {  // calculateDryFrictionVector, primitive part (Simulation.cpp:640-652)
        f3 n = ld3(rec_n, i, N);
        f3 d = f - prim_vout(S.prims[prim], n) * m;
        r = dry_friction(n, d, mu[S.prims[prim].group]);
      }
      st3(rec_f, i, N, f);
      st3(rec_r, i, N, r);
      f3 rhs = f + r - v * m;
      const float di = S.dinv[i];
      st3(cg_r, i, N, rhs);
      st3(cg_p, i, N, rhs * di);
      st3(cg_x, i, N, mk(0, 0, 0));
      part += dot(rhs, rhs) * di;
    }
    const double rz = block_sum<THREADS>((double) part, red);
    // ---- global step: P dv = rhs (Simulation.cpp:1267) ----
    cg_total += block_pcg<THREADS>(S, cg_r, cg_p, cg_ap, cg_x, rz, A.cg_tol, A.cg_max, red);
    // ---- update + convergence (Simulation.cpp:1268, 1310-1373) ----
    part = 0.f;
    for (int i = tid; i < N; i += THREADS) {
      f3 d = ld3(cg_x, i, N);
      st3(vnow, i, N, ld3(vnow, i, N) + d);
      part += dot(d, d);
    }
    xdiff = (double) h * sqrt(block_sum<THREADS>((double) part, red)) / (double) N;
    iters = iter + 1;
    converged = xdiff < (double) A.fwd_tol;
    if (xdiff < min_xdiff) {
      if (xdiff < 0.99 * min_xdiff) since_progress = 0;
      min_xdiff = xdiff;
      improved = true;
      if (!converged)
        for (int i = tid; i < N; i += THREADS) st3(vbest, i, N, ld3(vnow, i, N));
    }
    if (converged) break;
    // fp32 floor: |x_new - x_now| stopped decreasing although the tolerance (often 1e-9..1e-10 in the reference's
    // scene tables, below fp32 resolution) is not met -> return the best iterate instead of burning the whole cap
    if (++since_progress >= A.stall_window) { stalled = true; break; }
  }
  // ---- write the new state (revert to the best iterate when the cap was hit, Simulation.cpp:1357-1367) ----
  float *xo = A.x_out + off, *vo = A.v_out + off;
  for (int i = tid; i < N; i += THREADS) {
    f3 x = ld3(xn, i, N);
    if (converged) { f3 v = ld3(vnow, i, N); st3(vo, i, N, v); st3(xo, i, N, x + v * h); }
    else if (improved) { f3 v = ld3(vbest, i, N); st3(vo, i, N, v); st3(xo, i, N, x + v * h); }
    else { st3(vo, i, N, ld3(vn, i, N)); st3(xo, i, N, x); }
  }
  if (tid == 0) {
    dc_step_stats s;
    s.converged = converged ? 1 : (stalled ? 2 : 0); s.pd_iters = iters; s.cg_iters = cg_total; s.prim_contacts = total_contacts;
    s.self_contacts = 0; s.last_xdiff = (float) xdiff;
    A.stats[b] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// Backward: Simulation::stepBackward() (Simulation.cpp:1455-1780), matrix-free.
//   fixed point     P u = g + dP^T u,  dP^T u = h^2 (dp/dx)^T A y - C w,  w = dr_df^T u,  y = u + w
//   correction form residual = g - M u - h^2 (A - dp/dx)^T A y      (C = h^2 A^T A, P = M + C)
// The sparse Jacobian dproj_dxnew of the reference (66 % of its backward time) is never assembled: the
// per-element blocks are re-derived from x_new in registers and applied on the fly.
// ---------------------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_adjoint_step(DevSystem S, DevWork W, BwdArgs A) {
  __shared__ double red[THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = S.N, T = S.T, E = S.E, NC = S.NC;
  const size_t off = (size_t) b * 3 * N;
  const float *xnew = A.x_new + off, *rec_f = A.rec_f + off, *rec_n = A.rec_n + off;
  const int *rec_prim = A.rec_prim + (size_t) b * N;
  const float *mu = A.mu + (size_t) b * S.ngroups;
  float *gx = A.gx + off, *gv = A.gv + off;
  float *gin = W.g + off, *u = W.vnow + off, *y = W.vbest + off;
  float *cg_r = W.cg_r + off, *cg_p = W.cg_p + off, *cg_ap = W.cg_ap + off, *cg_x = W.cg_x + off;
  float *corner = W.corner + (size_t) b * 3 * NC;
  const float h = S.h, h2 = S.h * S.h;

  // ---- gradient clipping (Simulation.cpp:1460-1466) and u = 0 ----
  float part = 0.f;
  for (int i = tid; i < N; i += THREADS) { f3 q = ld3(gx, i, N); part += dot(q, q); }
  const double gnorm = sqrt(block_sum<THREADS>((double) part, red));
  float gscale = 1.f;
  int clipped = 0;
  if (A.clip && gnorm > (double) A.clip_thr * N) { gscale = (float) ((double) A.clip_thr * N / gnorm); clipped = 1; }
  for (int i = tid; i < N; i += THREADS) {
    st3(gin, i, N, ld3(gx, i, N) * gscale);
    st3(u, i, N, mk(0, 0, 0));
  }
  bool converged = false, stalled = false;
  int iters = 0, cg_total = 0, since_progress = 0;
  double udiff = 0, min_udiff = 1e300;
  const int cap = A.it_cap;
  __syncthreads();
  for (int it = 0; it < cap; it++) {
    // ---- y = (I + dr_df)^T u, primitive contacts are block diagonal (Simulation.cpp:700-711) ----
    for (int i = tid; i < N; i += THREADS) {
      f3 ui = ld3(u, i, N);
      f3 w = mk(0, 0, 0);
      const int prim = rec_prim[i];
      if (prim >= 0) {
        f3 n = ld3(rec_n, i, N);
        f3 d = ld3(rec_f, i, N) - prim_vout(S.prims[prim], n) * S.mass[i];
        w = dri_dfi_T(n, d, mu[S.prims[prim].group], ui);
      }
      st3(y, i, N, ui + w);
    }
    __syncthreads();
    // ---- per element: h^2 (A - dp/dx)^T A y ----
    // triangles: Triangle::projectToManifoldBackward (Triangle.cpp:354-451) in closed form:
    //   dT(Y) = TJ <TJ,Y> / tr(S) + (I - T T^T) Y S^-1,   TJ = [t1, -t0]
    for (int t = tid; t < T; t += THREADS) {
      const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
      const float4 D = S.tri_D[t];
      f3 x0 = ld3(xnew, i0, N);
      f3 e0 = ld3(xnew, i1, N) - x0, e1 = ld3(xnew, i2, N) - x0;
      Polar P = polar3x2(e0 * D.x + e1 * D.z, e0 * D.y + e1 * D.w);
      f3 q0 = ld3(y, i0, N);
      f3 d0 = ld3(y, i1, N) - q0, d1 = ld3(y, i2, N) - q0;
      f3 y0 = d0 * D.x + d1 * D.z, y1 = d0 * D.y + d1 * D.w;
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
    for (int e = tid; e < E; e += THREADS) {
      const int i0 = S.bend_v[e], i1 = S.bend_v[E + e], i2 = S.bend_v[2 * E + e], i3 = S.bend_v[3 * E + e];
      const float4 w = S.bend_w[e];
      const float2 nw = S.bend_nw[e];
      f3 x0 = ld3(xnew, i0, N);
      f3 ev = (ld3(xnew, i1, N) - x0) * w.y + (ld3(xnew, i2, N) - x0) * w.z + (ld3(xnew, i3, N) - x0) * w.w;
      f3 q0 = ld3(y, i0, N);
      f3 ey = (ld3(y, i1, N) - q0) * w.y + (ld3(y, i2, N) - q0) * w.z + (ld3(y, i3, N) - q0) * w.w;
      f3 res = ey;
      if (nw.x > 1e-6f) {
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
    // ---- residual of (P - dP^T) u = g, right-hand side of the correction solve ----
    part = 0.f;
    for (int i = tid; i < N; i += THREADS) {
      f3 r = ld3(gin, i, N) - ld3(u, i, N) * S.mass[i];
      const int k1 = S.inc_ptr[i + 1];
      for (int k = S.inc_ptr[i]; k < k1; k++) r = r - ld3(corner, S.inc_idx[k], NC);
      if (S.att_of_vertex[i] >= 0) r = r - ld3(y, i, N) * (h2 * S.k_att);   // attachment: dp/dx = 0
      const float di = S.dinv[i];
      st3(cg_r, i, N, r);
      st3(cg_p, i, N, r * di);
      st3(cg_x, i, N, mk(0, 0, 0));
      part += dot(r, r) * di;
    }
    const double rz = block_sum<THREADS>((double) part, red);
    cg_total += block_pcg<THREADS>(S, cg_r, cg_p, cg_ap, cg_x, rz, A.cg_tol, A.cg_max, red);
    part = 0.f;
    for (int i = tid; i < N; i += THREADS) {
      f3 d = ld3(cg_x, i, N);
      st3(u, i, N, ld3(u, i, N) + d);
      part += dot(d, d);
    }
    udiff = sqrt(block_sum<THREADS>((double) part, red)) / (double) N;
    iters = it + 1;
    if (udiff < (double) A.bwd_tol) { converged = true; break; }
    // fp32 floor guard, as in the forward loop (u is updated in place, so the last iterate is the best one)
    if (udiff < 0.99 * min_udiff) since_progress = 0;
    if (udiff < min_udiff) min_udiff = udiff;
    if (++since_progress >= A.stall_window) { stalled = true; break; }
  }
  __syncthreads();
  // ---- gradients w.r.t. the previous state and parameters (Simulation.cpp:1534, 1608-1650) ----
  float dmu_part[kMaxPrims];
#pragma unroll
  for (int k = 0; k < kMaxPrims; k++) dmu_part[k] = 0.f;
  float *dxf = A.d_xfixed ? A.d_xfixed + (size_t) b * 3 * S.Af : nullptr;
  for (int i = tid; i < N; i += THREADS) {
    f3 ui = ld3(u, i, N);
    const float m = S.mass[i];
    f3 w = mk(0, 0, 0);
    const int prim = rec_prim[i];
    if (prim >= 0) {
      f3 n = ld3(rec_n, i, N);
      f3 d = ld3(rec_f, i, N) - prim_vout(S.prims[prim], n) * m;
      const int grp = S.prims[prim].group;
      w = dri_dfi_T(n, d, mu[grp], ui);
      const float contrib = dot(dri_dmu(n, d, mu[grp]), ui) * h;
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
  if (tid == 0) {
    dc_bwd_stats s;
    s.converged = converged ? 1 : (stalled ? 2 : 0); s.adjoint_iters = iters; s.cg_iters = cg_total; s.clipped = clipped; s.last_udiff = (float) udiff;
    A.stats[b] = s;
  }
}

// ---------------------------------------------------------------------------------------------------
// layout conversion at the boundary: host float64 xyz-interleaved  <->  device float32 planar
// ---------------------------------------------------------------------------------------------------
__global__ void k_f64i_to_f32p(const double *__restrict__ src, float *__restrict__ dst, int n, long total) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long b = t / n;
  int i = (int) (t - b * n);
  const double *s = src + (b * n + i) * 3;
  float *d = dst + b * 3 * n;
  d[i] = (float) s[0]; d[n + i] = (float) s[1]; d[2 * n + i] = (float) s[2];
}
__global__ void k_f32p_to_f64i(const float *__restrict__ src, double *__restrict__ dst, int n, long total) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long b = t / n;
  int i = (int) (t - b * n);
  const float *s = src + b * 3 * n;
  double *d = dst + (b * n + i) * 3;
  d[0] = s[i]; d[1] = s[n + i]; d[2] = s[2 * n + i];
}
__global__ void k_seed_gradient(const float *__restrict__ x, const float *__restrict__ target, float *__restrict__ gx,
                                float *__restrict__ gv, int n3, long total, float scale) {
  long t = (long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int k = (int) (t % n3);
  gx[t] = scale * (x[t] - target[k]);
  gv[t] = 0.f;
}

void launch_f64i_to_f32p(const double *src, float *dst, int B, int n, hipStream_t st) {
  long total = (long) B * n;
  if (total == 0) return;
  hipLaunchKernelGGL(k_f64i_to_f32p, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, src, dst, n, total);
}
void launch_f32p_to_f64i(const float *src, double *dst, int B, int n, hipStream_t st) {
  long total = (long) B * n;
  if (total == 0) return;
  hipLaunchKernelGGL(k_f32p_to_f64i, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, src, dst, n, total);
}
void launch_seed_gradient(const float *x, const float *target, float *gx, float *gv, int B, int N, float scale, hipStream_t st) {
  long total = (long) B * 3 * N;
  hipLaunchKernelGGL(k_seed_gradient, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, st, x, target, gx, gv, 3 * N, total, scale);
}

static int pick_threads(int N) { return N <= 1536 ? 256 : (N <= 6144 ? 512 : 1024); }

void launch_pd_step(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  switch (pick_threads(S.N)) {
    case 256: hipLaunchKernelGGL(k_pd_step<256>, dim3(B), dim3(256), 0, st, S, W, A); break;
    case 512: hipLaunchKernelGGL(k_pd_step<512>, dim3(B), dim3(512), 0, st, S, W, A); break;
    default: hipLaunchKernelGGL(k_pd_step<1024>, dim3(B), dim3(1024), 0, st, S, W, A); break;
  }
}
void launch_adjoint_step(const DevSystem &S, const DevWork &W, const BwdArgs &A, int B, hipStream_t st) {
  switch (pick_threads(S.N)) {
    case 256: hipLaunchKernelGGL(k_adjoint_step<256>, dim3(B), dim3(256), 0, st, S, W, A); break;
    case 512: hipLaunchKernelGGL(k_adjoint_step<512>, dim3(B), dim3(512), 0, st, S, W, A); break;
    default: hipLaunchKernelGGL(k_adjoint_step<1024>, dim3(B), dim3(1024), 0, st, S, W, A); break;
  }
}

}  // namespace dc
