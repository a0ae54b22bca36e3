// Block-Jacobi preconditioner of the adjoint operator K = M + h^2 (A - dp/dx)^T A (I + dr_df)^T (dc_adjoint.hip): the inverse of
// K's own 3 x 3 diagonal blocks, rebuilt from x_new once per backward step.
//
// Why not D = diag(P) (the forward solve's Jacobi): P = M + h^2 A^T A is isotropic, K is not — an isometric element resists
// in-plane stretch but (I - dp/dx) removes the out-of-plane and rotational part of A^T A, and a sticking contact removes the
// elastic part of its vertex altogether (I + dr_df^T = 0 there). Per vertex, in closed form (Y = y g^T is the deformation-gradient
// perturbation a displacement y of corner c produces, g = the corner's row of inv_deltaUV; T, S the polar factors of F):
//   triangle   h^2 w^2 [ |g|^2 I - a a^T / tr S - (g^T S^-1 g) n n^T ],   a = t1 g_x - t0 g_y,  n = t0 x t1
//              (from dT(Y) = TJ <TJ,Y> / tr S + (I - T T^T) Y S^-1, Triangle.cpp:354-451 in closed form)
//   flap       h^2 w^2 c^2 [ I - (n_rest / |e|) (I - e e^T / |e|^2) ]     (TriangleBending.cpp:154-172), c = the corner's cotan weight
//   clip       h^2 k_att I                                                  (AttachmentSpring.cpp:35-37: dp/dx = 0)
//   K_ii = m I + E_ii (I + J_i^T),  J_i = dr/df of the vertex's primitive contact (Simulation.cpp:881-919); self contacts are
//   left out of the preconditioner (they couple vertex pairs).
#pragma once
#include "dc_devlib.h"

namespace dc {

struct Sym3 {
  float xx, xy, xz, yy, yz, zz;
};
__device__ __forceinline__ void sym_add_iso(Sym3 &m, float s) { m.xx += s; m.yy += s; m.zz += s; }
__device__ __forceinline__ void sym_add_outer(Sym3 &m, f3 a, float s) {
  m.xx = fmaf(s * a.x, a.x, m.xx); m.xy = fmaf(s * a.x, a.y, m.xy); m.xz = fmaf(s * a.x, a.z, m.xz);
  m.yy = fmaf(s * a.y, a.y, m.yy); m.yz = fmaf(s * a.y, a.z, m.yz); m.zz = fmaf(s * a.z, a.z, m.zz);
}
__device__ __forceinline__ f3 sym_mul(const Sym3 &m, f3 v) {
  return mk(m.xx * v.x + m.xy * v.y + m.xz * v.z, m.xy * v.x + m.yy * v.y + m.yz * v.z, m.xz * v.x + m.yz * v.y + m.zz * v.z);
}

// E_ii + h^2 k_att I of vertex i at the linearisation point xnew (planar [3][N])
__device__ __forceinline__ Sym3 elastic_diag_block(const DevSystem &S, const float *__restrict__ xnew, int i) {
  const int N = S.N, T = S.T, E = S.E;
  const float h2 = S.h * S.h;
  Sym3 B = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int k1 = S.inc_ptr[i + 1];
  for (int k = S.inc_ptr[i]; k < k1; k++) {
    const int idx = S.inc_idx[k];
    if (idx < 3 * T) {
      const int corner = idx / T, t = idx - corner * T;
      const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
      const float4 D = S.tri_D[t];
      const f3 x0 = ld3(xnew, i0, N);
      const f3 e0 = ld3(xnew, i1, N) - x0, e1 = ld3(xnew, i2, N) - x0;
      const Polar P = polar3x2(e0 * D.x + e1 * D.z, e0 * D.y + e1 * D.w);
      float gx, gy;
      if (corner == 1) { gx = D.x; gy = D.y; }
      else if (corner == 2) { gx = D.z; gy = D.w; }
      else { gx = -(D.x + D.z); gy = -(D.y + D.w); }
      const float s = h2 * S.tri_w2[t];
      const f3 a = P.t1 * gx - P.t0 * gy;
      const f3 n = cross(P.t0, P.t1);                       // unit: T has orthonormal columns
      sym_add_iso(B, s * (gx * gx + gy * gy));
      sym_add_outer(B, a, -s * fast_rcp(P.trS));
      sym_add_outer(B, n, -s * (gx * gx * P.i00 + 2.f * gx * gy * P.i01 + gy * gy * P.i11));
    } else {
      const int q = idx - 3 * T, corner = q / E, e = q - corner * E;
      const int i0 = S.bend_v[e], i1 = S.bend_v[E + e], i2 = S.bend_v[2 * E + e], i3 = S.bend_v[3 * E + e];
      const float4 w = S.bend_w[e];
      const float2 nw = S.bend_nw[e];
      const float c = corner == 0 ? w.x : (corner == 1 ? w.y : (corner == 2 ? w.z : w.w));
      const float s = h2 * nw.y * c * c;
      sym_add_iso(B, s);
      if (nw.x > 1e-6f) {
        const f3 x0 = ld3(xnew, i0, N);
        const f3 ev = (ld3(xnew, i1, N) - x0) * w.y + (ld3(xnew, i2, N) - x0) * w.z + (ld3(xnew, i3, N) - x0) * w.w;
        const float ien = fast_rsqrt(fmaxf(dot(ev, ev), 1e-30f));
        const f3 eh = ev * ien;
        sym_add_iso(B, -s * nw.x * ien);
        sym_add_outer(B, eh, s * nw.x * ien);
      }
    }
  }
  if (S.att_of_vertex[i] >= 0) sym_add_iso(B, h2 * S.k_att);
  return B;
}

// inverse of K_ii = m I + B (I + J^T), J^T given by its action jt(e) on a vector; written to minv[k * N + i], k = 0..8 (row major)
template <class JT>
__device__ __forceinline__ void store_block_inverse(const Sym3 &B, float m, JT jt, float *__restrict__ minv, int i, int N) {
  const f3 ex = mk(1, 0, 0), ey = mk(0, 1, 0), ez = mk(0, 0, 1);
  const f3 cx = sym_mul(B, ex + jt(ex)) + ex * m, cy = sym_mul(B, ey + jt(ey)) + ey * m, cz = sym_mul(B, ez + jt(ez)) + ez * m;   // columns of K_ii
  // inverse by the adjugate: rows of the inverse are the cross products of the columns over the determinant
  const f3 r0 = cross(cy, cz), r1 = cross(cz, cx), r2 = cross(cx, cy);
  const float det = dot(cx, r0);
  float inv = 1.0f / det;
  f3 q0 = r0 * inv, q1 = r1 * inv, q2 = r2 * inv;
  // K_ii = m I + (elastic block) has no eigenvalue below m while the elastic block is positive semi-definite; under compression
  // I - dp/dx has negative directions and the block can come close to singular or turn indefinite — its inverse would then blow a
  // residual component up instead of damping it (BiCGSTAB breaks down, seen on a squashed 7 742-vertex garment). A block whose inverse
  // has an entry above 2 / m (an eigenvalue below m / 2), a non-positive determinant or a non-finite inverse is replaced by the scalar
  // 1 / (m + tr(B) / 3), the diag(P)-like value of that vertex.
  const float lim = 2.0f / m;
  const float big = fmaxf(fmaxf(fmaxf(fabsf(q0.x), fabsf(q0.y)), fmaxf(fabsf(q0.z), fabsf(q1.x))),
                          fmaxf(fmaxf(fabsf(q1.y), fabsf(q1.z)), fmaxf(fmaxf(fabsf(q2.x), fabsf(q2.y)), fabsf(q2.z))));
  if (!(det > 1e-30f) || !isfinite(inv) || !(big <= lim)) {
    const float d = 1.0f / (m + fmaxf(0.f, (B.xx + B.yy + B.zz) * (1.0f / 3.0f)));
    q0 = mk(d, 0, 0); q1 = mk(0, d, 0); q2 = mk(0, 0, d);
  }
  minv[i] = q0.x; minv[N + i] = q0.y; minv[2 * N + i] = q0.z;
  minv[3 * N + i] = q1.x; minv[4 * N + i] = q1.y; minv[5 * N + i] = q1.z;
  minv[6 * N + i] = q2.x; minv[7 * N + i] = q2.y; minv[8 * N + i] = q2.z;
}

// z = M_i^-1 r
__device__ __forceinline__ f3 block_pre(const float *__restrict__ minv, int i, int N, f3 r) {
  return mk(minv[i] * r.x + minv[N + i] * r.y + minv[2 * N + i] * r.z,
            minv[3 * N + i] * r.x + minv[4 * N + i] * r.y + minv[5 * N + i] * r.z,
            minv[6 * N + i] * r.x + minv[7 * N + i] * r.y + minv[8 * N + i] * r.z);
}

}  // namespace dc
