// Device-side building blocks shared by the forward and adjoint kernels (included by dc_forward.hip / dc_adjoint.hip).
#pragma once
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
// One plane of a planar vector as a buffer resource of `count` floats: a store (load) at an index >= count is dropped (returns 0) by the
// hardware's range check — rows past the end of a vector cost no branch, and count = 0 switches a whole stream of stores off.
struct PlaneBuf {
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ void st(int i, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rs, i * 4, 0, 0); }
  __device__ __forceinline__ float ld(int i) const { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, i * 4, 0, 0)); }
};
__device__ __forceinline__ PlaneBuf plane_buf(const float *p, int count) {
  PlaneBuf b;
  b.rs = __builtin_amdgcn_make_buffer_rsrc((void *) p, 0, count * 4, 0x00020000);
  return b;
}
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
// The per-element passes are bound by the fp32 VALU rate (not by memory): the 1-ulp hardware square root /
// reciprocal (v_sqrt_f32, v_rcp_f32, v_rsq_f32) replace the ~12-instruction IEEE expansions of sqrtf and '/'.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ f3 normalized_fast(f3 a) {
  float n2 = dot(a, a);
  return n2 > 0.f ? a * fast_rsqrt(n2) : a;
}

__device__ __forceinline__ Polar polar3x2(f3 f0, f3 f1) {
  float a = dot(f0, f0), b = dot(f0, f1), c = dot(f1, f1);
  float det = fmaxf(a * c - b * b, 1e-30f);
  float s = fast_sqrt(det);
  float t = fast_sqrt(a + c + 2.f * s);
  float inv = fast_rcp(t * s);
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

// Primitive::isInContact family (Sphere Primitive.cpp:221-261, Capsule :570-604, Plane :66-130, Bowl :362-381) for one
// flattened primitive.
__device__ __forceinline__ bool prim_in_contact(const DevPrim &p, f3 pos, f3 &normal) {
  f3 c = mk(p.cx, p.cy, p.cz);
  f3 q = pos - c;
  if (p.kind == DC_PRIM_SPHERE || p.kind == DC_PRIM_SPHERE_DISCRETIZED) {      // (discretised: the normal is replaced by detect_primitive)
    float dist = sqrtf(dot(q, q)) - p.radius;
    normal = normalized(q);
    return dist < 0.1f;
  }
  if (p.kind == DC_PRIM_BOWL) {     // Bowl::isInContact (Primitive.cpp:362-381): inside of the lower half of a shell, eps 0.005
    const float len = sqrtf(dot(q, q));
    normal = q * (-1.0f / len);
    return (len - p.radius <= 0.005f) && !(pos.y > p.cy) && (len > p.radius - 0.005f);
  }
  if (p.kind == DC_PRIM_PLANE) {    // Plane::isInContact (Primitive.cpp:66-130): finite rectangle, eps 0.4, thickness 5, edge tol 5e-4
    const f3 ul = mk(p.tx, p.ty, p.tz), ur = mk(p.ux, p.uy, p.uz), lr = ul * -1.0f, ll = ur * -1.0f;
    const float eps = 0.4f, edge_tol = 0.0005f;
    const float br = sqrtf(fmaxf(dot(ul, ul), dot(ur, ur)));
    if (sqrtf(dot(q, q)) > br + eps) return false;
    const f3 n = normalized(cross(ur, ul));
    const float dp = dot(n, q);
    if (fabsf(dp) > eps) return false;         // (with it the thickness test of :84-85 can never trigger)
    const f3 pp = q - n * dp;
    auto inside = [&](f3 a, f3 b, f3 c) {      // Primitive.h:176-190
      const f3 AB = b - a, AC = c - a, nn = cross(AB, AC), AP = pp - a;
      const float n2 = dot(nn, nn);
      const float alpha = dot(cross(AB, AP), nn) / n2, beta = dot(cross(AP, AC), nn) / n2, gamma = 1.f - alpha - beta;
      return alpha >= 0.f && beta >= 0.f && gamma >= 0.f && gamma <= 1.f && alpha <= 1.f && beta <= 1.f;
    };
    if (inside(ul, ur, ll) || inside(ll, ur, lr)) { normal = n; return true; }   // dp >= -eps here: the sign factor of :93 is +1
    const f3 ea[4] = {ul, ur, ll, ul}, eb[4] = {ur, lr, lr, ll};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const f3 AB = eb[k] - ea[k], AP = q - ea[k];
      const f3 pr = AB * (dot(AP, AB) / dot(AB, AB));
      const f3 Pp = ea[k] + pr;
      const float ABl = sqrtf(dot(AB, AB)), APl = sqrtf(dot(pr, pr));
      const f3 pb = Pp - eb[k];
      float t = APl / ABl;
      if (sqrtf(dot(pb, pb)) > ABl) t = -t;
      const f3 off = q - Pp;
      if (sqrtf(dot(off, off)) < edge_tol && t > -edge_tol && t < 1.f + edge_tol) {
        normal = normalized(t < 0.f ? q - ea[k] : (t > 1.f ? q - eb[k] : off));
        return true;
      }
    }
    return false;
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
// Sphere::isInContact, discretized branch (Primitive.cpp:230-253): the contact normal becomes the face normal of the LAST triangle (creation
// order) of the sphere's own mesh whose prism holds the point (Primitive::pointInsideTriangle, Primitive.h:176-190: all barycentric weights
// in [0, 1]) and whose "projection" — alpha p1 + beta p2 + gamma p0, the reference's weights on the wrong corners, kept — is closer than the
// radius (that rules out the antipodal face). Inside / outside is a comparison against 0: done in fp64 like the reference, on the sample point
// re-formed in fp64 from its fp32 terms. Runs once per contacting vertex and step (3 120 faces at the reference's resolution).
__device__ __attribute__((noinline)) void sphere_mesh_normal(const DevSystem &S, const DevPrim &p, double qx, double qy, double qz, f3 &normal) {
  const double DC_G *T = S.dsph_tri;
  const double r = (double) p.radius;
  for (int t = 0; t < S.dsph_ntri; t++, T += 12) {
    const double ax = T[0], ay = T[1], az = T[2];
    const double abx = T[3] - ax, aby = T[4] - ay, abz = T[5] - az, acx = T[6] - ax, acy = T[7] - ay, acz = T[8] - az;
    const double nx = aby * acz - abz * acy, ny = abz * acx - abx * acz, nz = abx * acy - aby * acx;
    const double n2 = nx * nx + ny * ny + nz * nz;
    const double apx = qx - ax, apy = qy - ay, apz = qz - az;
    const double alpha = ((aby * apz - abz * apy) * nx + (abz * apx - abx * apz) * ny + (abx * apy - aby * apx) * nz) / n2;
    const double beta = ((apy * acz - apz * acy) * nx + (apz * acx - apx * acz) * ny + (apx * acy - apy * acx) * nz) / n2;
    const double gamma = 1.0 - alpha - beta;
    if (!(alpha >= 0 && beta >= 0 && gamma >= 0 && gamma <= 1 && alpha <= 1 && beta <= 1)) continue;
    const double px = alpha * T[3] + beta * T[6] + gamma * ax, py = alpha * T[4] + beta * T[7] + gamma * ay, pz = alpha * T[5] + beta * T[8] + gamma * az;
    const double dx = qx - px, dy = qy - py, dz = qz - pz;
    if (sqrt(dx * dx + dy * dy + dz * dz) < r) normal = mk((float) T[9], (float) T[10], (float) T[11]);
  }
}
// Simulation::isInContactWithObstacle (Simulation.cpp:153-191): t = 0, h/2, h; first primitive / first sample wins.
// Children of one LowerLeg share a group and are tested, per sample, in child order (Primitive.cpp:410-418).
// (not inlined: it runs once per vertex and step, and its plane / capsule branches must not weigh on the register allocation
// of the PD / PCG loops of the kernels that call it)
__device__ __attribute__((noinline)) int detect_primitive(const DevSystem &S, f3 pos, f3 vel, f3 &normal) {
  int p0 = 0;
  while (p0 < S.nprim) {
    int p1 = p0;
    while (p1 < S.nprim && S.prims[p1].group == S.prims[p0].group) p1++;
    for (int k = 0; k < 3; k++) {
      f3 q = pos + vel * (S.h * 0.5f * (float) k);
      for (int p = p0; p < p1; p++)
        if (prim_in_contact(S.prims[p], q, normal)) {
          if (S.prims[p].kind == DC_PRIM_SPHERE_DISCRETIZED && S.dsph_ntri > 0) {
            const double ts = S.h64 * (0.5 * (double) k);
            sphere_mesh_normal(S, S.prims[p], (double) pos.x + (double) vel.x * ts - (double) S.prims[p].cx, (double) pos.y + (double) vel.y * ts - (double) S.prims[p].cy,
                               (double) pos.z + (double) vel.z * ts - (double) S.prims[p].cz, normal);
          }
          return p;
        }
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
// rz_stop: the r . D^-1 r the relative tolerance refers to (the right-hand side's; differs from rz after a deflation projection).
template <int THREADS>
__device__ __forceinline__ int block_pcg(const DevSystem &S, float *cg_r, float *cg_p, float *cg_ap, float *cg_x,
                                         double rz, float rel_tol, int max_iter, double *red, double rz_stop = -1.0) {
  const int N = S.N, tid = threadIdx.x;
  const double stop = (double) rel_tol * (double) rel_tol * (rz_stop >= 0 ? rz_stop : rz);
  if (!(rz > 1e-300) || !(rz > stop)) return 0;
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

// Spectral deflation for the global-memory solve (round 6; dc_deflate.h — the resident kernels have their own versions): Galerkin projection of
// the residual onto the 16 lowest eigenvectors U of the SCALED matrix Ahat = D^-1/2 P D^-1/2, in the unscaled variables block_pcg works in:
//   rhat = D^-1/2 r,  c = (U^T Ahat U)^-1 U^T rhat per coordinate,  x += D^-1/2 U c,  r -= D^1/2 (Ahat U) c,  p = D^-1 r.
// The meshes that end up on this kernel are the ones too wide for the packet tables — the reference's 17 562-vertex dress — and exactly the
// irregular garments whose Jacobi-PCG needs hundreds of iterations (355 per PD iteration there). Returns the new r . D^-1 r. `scr` = 64 * 48 + 96
// floats of LDS. All threads call; ends with a barrier.
template <int THREADS>
__device__ __forceinline__ double deflate_global(const DevSystem &S, float *cg_r, float *cg_p, float *cg_x, float *scr, double *red) {
  constexpr int DK = 16, NV = 3 * DK, NW = THREADS / 64;
  const int N = S.N, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float4 DC_G *U4 = (const float4 DC_G *) S.defl_u;
  const float4 DC_G *AU4 = (const float4 DC_G *) S.defl_au;
  float *wsum = scr, *tv = scr + NW * NV, *cv = tv + NV;      // [NW][48] wave sums, [48] U^T rhat, [48] c
  __syncthreads();
#pragma unroll 1
  for (int j4 = 0; j4 < DK / 4; j4++) {
    float acc[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < N; i += THREADS) {
      const float4 u = U4[(size_t) i * (DK / 4) + j4];
      const float sq = S.sq_dinv[i];
      const f3 q = ld3(cg_r, i, N) * sq;
      acc[0] = fmaf(u.x, q.x, acc[0]); acc[1] = fmaf(u.x, q.y, acc[1]); acc[2] = fmaf(u.x, q.z, acc[2]);
      acc[3] = fmaf(u.y, q.x, acc[3]); acc[4] = fmaf(u.y, q.y, acc[4]); acc[5] = fmaf(u.y, q.z, acc[5]);
      acc[6] = fmaf(u.z, q.x, acc[6]); acc[7] = fmaf(u.z, q.y, acc[7]); acc[8] = fmaf(u.z, q.z, acc[8]);
      acc[9] = fmaf(u.w, q.x, acc[9]); acc[10] = fmaf(u.w, q.y, acc[10]); acc[11] = fmaf(u.w, q.z, acc[11]);
    }
#pragma unroll
    for (int m = 0; m < 12; m++) {
      float v = acc[m];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) wsum[wv * NV + j4 * 12 + m] = v;      // entry (vector 4 j4 + m / 3, coordinate m % 3)
    }
  }
  __syncthreads();
  if (tid < NV) {
    float t = 0.f;
    for (int w = 0; w < NW; w++) t += wsum[w * NV + tid];
    tv[tid] = t;
  }
  __syncthreads();
  if (tid < NV) {
    const int j = tid / 3, c = tid - 3 * j;
    float sacc = 0.f;
    for (int l = 0; l < DK; l++) sacc = fmaf(S.defl_g[j * DK + l], tv[l * 3 + c], sacc);
    cv[tid] = sacc;
  }
  __syncthreads();
  float part = 0.f;
  for (int i = tid; i < N; i += THREADS) {
    float dx[3] = {0.f, 0.f, 0.f}, dr[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int j4 = 0; j4 < DK / 4; j4++) {
      const float4 u = U4[(size_t) i * (DK / 4) + j4], a = AU4[(size_t) i * (DK / 4) + j4];
      const float *c = cv + j4 * 12;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        dx[k] += u.x * c[k] + u.y * c[3 + k] + u.z * c[6 + k] + u.w * c[9 + k];
        dr[k] += a.x * c[k] + a.y * c[3 + k] + a.z * c[6 + k] + a.w * c[9 + k];
      }
    }
    const float sq = S.sq_dinv[i], isq = 1.0f / sq, di = S.dinv[i];
    const f3 x = ld3(cg_x, i, N) + mk(dx[0], dx[1], dx[2]) * sq;
    const f3 r = ld3(cg_r, i, N) - mk(dr[0], dr[1], dr[2]) * isq;
    st3(cg_x, i, N, x); st3(cg_r, i, N, r); st3(cg_p, i, N, r * di);
    part += dot(r, r) * di;
  }
  return block_sum<THREADS>((double) part, red);
}

// ---------------------------------------------------------------------------------------------------
// Self contacts: the layers of Simulation::contactSorting applied in sequence (Gauss-Seidel over layers, contacts
// of one layer are vertex-disjoint and run in parallel).
// ---------------------------------------------------------------------------------------------------
// Accessors of a rollout's planar [3][N] vector: PlainVec = ordinary global loads / stores; the split kernels pass BufVec
// (dc_cluster.h: write-through / L1-bypassing accesses) for vectors that other workgroups read or write.
struct PlainVec {
  float *p;
  __device__ __forceinline__ float ld(int idx) const { return p[idx]; }
  __device__ __forceinline__ void st(int idx, float v) const { p[idx] = v; }
};
template <class V> __device__ __forceinline__ f3 ld3v(const V &a, int i, int n) { return mk(a.ld(i), a.ld(n + i), a.ld(2 * n + i)); }
template <class V> __device__ __forceinline__ void st3v(const V &a, int i, int n, f3 v) { a.st(i, v.x); a.st(n + i, v.y); a.st(2 * n + i, v.z); }

constexpr float kClothMu = 0.1f;    // clothFrictionalCoeff, hard-coded in the reference (Simulation.cpp:666, :729)

// calculateDryFrictionVector, self part (Simulation.cpp:655-678): r += k * friction(d), d = (f+r)_A/m_A - (f+r)_B/m_B.
// f, r are the rollout's planar [3][N] arrays in global memory; stores d per contact. Call with all threads; the
// caller must have synchronised after writing f / r, this function ends with a barrier.
template <int THREADS, class FV, class RV>
__device__ __forceinline__ void self_friction_layers_v(const DevSystem &S, const SelfRec &R, int b, const FV &f, const RV &r) {
  const int cap = S.self_cap, N = S.N;
  const int *meta = R.meta + (size_t) b * kMetaStride;
  const int nl = meta[1];
  const int2 *pair = R.pair + (size_t) b * cap;
  const float4 *nrm = R.nrm + (size_t) b * cap;
  float4 *dvec = R.dvec + (size_t) b * cap;
  for (int l = 0; l < nl; l++) {
    const int k1 = meta[2 + l + 1];
    for (int k = meta[2 + l] + threadIdx.x; k < k1; k += THREADS) {
      const int2 ab = pair[k];
      const float4 n4 = nrm[k];
      const f3 n = mk(n4.x, n4.y, n4.z);
      const float mA = S.mass[ab.x], mB = S.mass[ab.y];
      f3 rA = ld3v(r, ab.x, N), rB = ld3v(r, ab.y, N);
      f3 d = (ld3v(f, ab.x, N) + rA) * (1.0f / mA) - (ld3v(f, ab.y, N) + rB) * (1.0f / mB);
      dvec[k] = make_float4(d.x, d.y, d.z, 0.f);
      f3 ri = dry_friction(n, d, kClothMu) * ((mA * mB) / (mA + mB));
      st3v(r, ab.x, N, rA + ri);
      st3v(r, ab.y, N, rB - ri);
    }
    __syncthreads();
  }
}
template <int THREADS>
__device__ __forceinline__ void self_friction_layers(const DevSystem &S, const SelfRec &R, int b, const float *f, float *r) {
  self_friction_layers_v<THREADS>(S, R, b, PlainVec{(float *) f}, PlainVec{r});
}

// z <- (I + J_0)^T ... (I + J_L)^T z for the self layers (calculatedr_df, Simulation.cpp:713-760, transposed):
// per contact (A,B): g = k D^T (z_A - z_B), z_A += g / m_A, z_B -= g / m_B, D = dri_dfi(n, d, 0.1).
// z is the rollout's planar [3][N] array in global memory. Ends with a barrier.
template <int THREADS, class ZV>
__device__ __forceinline__ void self_JT_layers_v(const DevSystem &S, const SelfRec &R, int b, const ZV &z) {
  const int cap = S.self_cap, N = S.N;
  const int *meta = R.meta + (size_t) b * kMetaStride;
  const int nl = meta[1];
  const int2 *pair = R.pair + (size_t) b * cap;
  const float4 *nrm = R.nrm + (size_t) b * cap;
  const float4 *dvec = R.dvec + (size_t) b * cap;
  for (int l = nl - 1; l >= 0; l--) {
    const int k1 = meta[2 + l + 1];
    for (int k = meta[2 + l] + threadIdx.x; k < k1; k += THREADS) {
      const int2 ab = pair[k];
      const float4 n4 = nrm[k], d4 = dvec[k];
      const float mA = S.mass[ab.x], mB = S.mass[ab.y];
      f3 zA = ld3v(z, ab.x, N), zB = ld3v(z, ab.y, N);
      f3 g = dri_dfi_T(mk(n4.x, n4.y, n4.z), mk(d4.x, d4.y, d4.z), kClothMu, zA - zB) * ((mA * mB) / (mA + mB));
      st3v(z, ab.x, N, zA + g * (1.0f / mA));
      st3v(z, ab.y, N, zB - g * (1.0f / mB));
    }
    __syncthreads();
  }
}
template <int THREADS>
__device__ __forceinline__ void self_JT_layers(const DevSystem &S, const SelfRec &R, int b, float *z) {
  self_JT_layers_v<THREADS>(S, R, b, PlainVec{z});
}

// ---- the same two passes inside LDS ----------------------------------------------------------------------------
// A chain of L contacts takes L layers (contactSorting), each with a handful of contacts: the versions above pay one
// workgroup barrier and one global-memory round trip per layer. Here the working set (the values of the M distinct
// vertices of the contacts, the contact records, the layer offsets) is staged into LDS once, ONE wave walks the layers in
// order (LDS operations of a wave execute in program order, so a layer sees the previous layer's writes without any
// workgroup barrier), and the results are scattered back. `lds` offers `lds_floats` floats; returns false (nothing done)
// when the working set does not fit, and the caller takes the global-memory version.
__device__ __forceinline__ int self_lds_need(int M, int C, int nl) { return 7 * M + 8 * C + nl + 2; }
constexpr int kWideLayers = 8;      // up to this many layers the LDS passes use the whole workgroup with a barrier per layer

template <int THREADS, class FV, class RV>
__device__ __forceinline__ bool self_friction_layers_lds_v(const DevSystem &S, const SelfRec &R, int b, const FV &f, const RV &r,
                                                           float *lds, int lds_floats) {
  const int cap = S.self_cap, N = S.N, tid = threadIdx.x;
  const int *meta = R.meta + (size_t) b * kMetaStride;
  const int C = min(meta[0], cap), nl = meta[1], M = meta[kMetaStride - 1];
  if (!S.self_lds || self_lds_need(M, C, nl) > lds_floats) return false;
  const int2 *pair = R.pair + (size_t) b * cap;
  const float4 *nrm = R.nrm + (size_t) b * cap;
  float4 *dvec = R.dvec + (size_t) b * cap;
  const int *verts = R.verts + (size_t) b * 2 * cap;
  float *lf = lds, *lr = lds + 3 * M, *lim = lds + 6 * M;
  float4 *ln = (float4 *) (lds + 7 * M + ((4 - (7 * M) % 4) % 4));       // 16-byte aligned
  float4 *ld = ln + C;
  int *loff = (int *) (ld + C);
  (void) pair;
  for (int s = tid; s < M; s += THREADS) {
    const int v = verts[s];
    lf[s] = f.ld(v); lf[M + s] = f.ld(N + v); lf[2 * M + s] = f.ld(2 * N + v);
    lr[s] = r.ld(v); lr[M + s] = r.ld(N + v); lr[2 * M + s] = r.ld(2 * N + v);
    lim[s] = 1.0f / S.mass[v];
  }
  for (int k = tid; k < C; k += THREADS) ln[k] = nrm[k];
  for (int l = tid; l <= nl; l += THREADS) loff[l] = meta[2 + l];
  __syncthreads();
  auto contact = [&](int k) {
    const float4 n4 = ln[k];
    const int sl = __float_as_int(n4.w), sa = sl & 0xffff, sb = sl >> 16;
    const f3 n = mk(n4.x, n4.y, n4.z);
    const float iA = lim[sa], iB = lim[sb];
    f3 rA = mk(lr[sa], lr[M + sa], lr[2 * M + sa]), rB = mk(lr[sb], lr[M + sb], lr[2 * M + sb]);
    f3 d = (mk(lf[sa], lf[M + sa], lf[2 * M + sa]) + rA) * iA - (mk(lf[sb], lf[M + sb], lf[2 * M + sb]) + rB) * iB;
    ld[k] = make_float4(d.x, d.y, d.z, 0.f);
    f3 ri = dry_friction(n, d, kClothMu) * (1.0f / (iA + iB));           // k = mA mB / (mA + mB)
    rA = rA + ri; rB = rB - ri;
    lr[sa] = rA.x; lr[M + sa] = rA.y; lr[2 * M + sa] = rA.z;
    lr[sb] = rB.x; lr[M + sb] = rB.y; lr[2 * M + sb] = rB.z;
  };
  if (nl <= kWideLayers) {
    // few, wide layers (a fold: hundreds of vertex-disjoint pairs in layer 0): every thread takes contacts, a barrier per layer
    for (int l = 0; l < nl; l++) {
      const int k1 = loff[l + 1];
      for (int k = loff[l] + tid; k < k1; k += THREADS) contact(k);
      __syncthreads();
    }
  } else {
    // many, thin layers (chains): ONE wave walks them, its LDS operations execute in program order — no barrier per layer
    if (tid < 64) {
      for (int l = 0; l < nl; l++) {
        const int k1 = loff[l + 1];
        for (int k = loff[l] + tid; k < k1; k += 64) contact(k);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");             // compiler: keep the layers' LDS accesses in order
      }
    }
    __syncthreads();
  }
  for (int s = tid; s < M; s += THREADS) { const int v = verts[s]; r.st(v, lr[s]); r.st(N + v, lr[M + s]); r.st(2 * N + v, lr[2 * M + s]); }
  for (int k = tid; k < C; k += THREADS) dvec[k] = ld[k];
  __syncthreads();
  return true;
}
template <int THREADS>
__device__ __forceinline__ bool self_friction_layers_lds(const DevSystem &S, const SelfRec &R, int b, const float *f, float *r,
                                                         float *lds, int lds_floats) {
  return self_friction_layers_lds_v<THREADS>(S, R, b, PlainVec{(float *) f}, PlainVec{r}, lds, lds_floats);
}

// CONTRACT used by the adjoint's y list (dc_adjoint.hip, adjoint_operator): on a `true` return the working set's results are still in LDS as
// three planes [3][M] of floats at `lds` (lz), slot s = position of the vertex in R.verts — the caller copies them from there before the element
// windows reuse that LDS. R.verts holds every vertex at most once (dc_selflib.h, section 6), so a slot names one vertex.
template <int THREADS, class ZV>
__device__ __forceinline__ bool self_JT_layers_lds_v(const DevSystem &S, const SelfRec &R, int b, const ZV &z, float *lds, int lds_floats) {
  const int cap = S.self_cap, N = S.N, tid = threadIdx.x;
  const int *meta = R.meta + (size_t) b * kMetaStride;
  const int C = min(meta[0], cap), nl = meta[1], M = meta[kMetaStride - 1];
  if (!S.self_lds || self_lds_need(M, C, nl) > lds_floats) return false;
  const float4 *nrm = R.nrm + (size_t) b * cap;
  const float4 *dvec = R.dvec + (size_t) b * cap;
  const int *verts = R.verts + (size_t) b * 2 * cap;
  float *lz = lds, *lim = lds + 3 * M;
  float4 *ln = (float4 *) (lds + 7 * M + ((4 - (7 * M) % 4) % 4));
  float4 *ld = ln + C;
  int *loff = (int *) (ld + C);
  for (int s = tid; s < M; s += THREADS) {
    const int v = verts[s];
    lz[s] = z.ld(v); lz[M + s] = z.ld(N + v); lz[2 * M + s] = z.ld(2 * N + v);
    lim[s] = 1.0f / S.mass[v];
  }
  for (int k = tid; k < C; k += THREADS) { ln[k] = nrm[k]; ld[k] = dvec[k]; }
  for (int l = tid; l <= nl; l += THREADS) loff[l] = meta[2 + l];
  __syncthreads();
  auto contact = [&](int k) {
    const float4 n4 = ln[k], d4 = ld[k];
    const int sl = __float_as_int(n4.w), sa = sl & 0xffff, sb = sl >> 16;
    const float iA = lim[sa], iB = lim[sb];
    f3 zA = mk(lz[sa], lz[M + sa], lz[2 * M + sa]), zB = mk(lz[sb], lz[M + sb], lz[2 * M + sb]);
    f3 g = dri_dfi_T(mk(n4.x, n4.y, n4.z), mk(d4.x, d4.y, d4.z), kClothMu, zA - zB) * (1.0f / (iA + iB));
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
  return true;
}
template <int THREADS>
__device__ __forceinline__ bool self_JT_layers_lds(const DevSystem &S, const SelfRec &R, int b, float *z, float *lds, int lds_floats) {
  return self_JT_layers_lds_v<THREADS>(S, R, b, PlainVec{z}, lds, lds_floats);
}

}  // namespace dc
