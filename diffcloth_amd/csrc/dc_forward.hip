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
#include <cstdlib>
#define DC_KERNEL_TU
#include "dc_devlib.h"
#include "dc_winlib.h"

namespace dc {

// ---------------------------------------------------------------------------------------------------
// Forward: Simulation::step()
// ---------------------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_pd_step(const DevSystem *__restrict__ Sp, DevWork W, FwdArgs A) {
  const DevSystem &S = *Sp;
  __shared__ double red[THREADS / 64];
  __shared__ float defl_scr[(THREADS / 64) * 48 + 96];      // deflate_global (irregular garments beyond the packet tables)
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
    if (A.fv) fext = fext + ld3(A.fv + off, i, N) * (A.fv_scale ? A.fv_scale[b] : 1.f);
    if (A.fv2) fext = fext + ld3(A.fv2 + off, i, N);
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
  const int nself = (S.contact_enabled && S.self_enabled) ? A.self.meta[(size_t) b * kMetaStride] : 0;   // from k_self_detect
  bool improved = false, converged = false, stalled = false;
  int iters = 0, cg_total = 0, since_progress = 0;
  double xdiff = 0;

  for (int iter = 0; iter < A.pd_cap; iter++) {
    // ---- local step: per-element projection residual, written per constraint corner ----
    // triangles: Triangle::project (Triangle.cpp:310-351); contribution h * w^2 * (T - F) D^T
    for (int t = tid; t < T; t += THREADS) {
      const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
      const float4 D = S.tri_D[t];
      // fp64-strain element operator (dc_winlib.h: the fp32 evaluation of T - F carries the 6e-8 roundings of F into a difference of
      // nearly equal quantities)
      f3 g0, g1;
      HybridTriOp{S.h64}(ld3(xn, i0, N), ld3(xn, i1, N), ld3(xn, i2, N), ld3(vnow, i0, N), ld3(vnow, i1, N), ld3(vnow, i2, N), D, S.tri_Dlo[t], S.tri_w2[t], g0, g1);
      f3 c1 = g0 * D.x + g1 * D.y, c2 = g0 * D.z + g1 * D.w;
      f3 c0 = mk(0, 0, 0) - c1 - c2;
      st3(corner, t, NC, c0); st3(corner, T + t, NC, c1); st3(corner, 2 * T + t, NC, c2);
    }
    // bending: TriangleBending::project (TriangleBending.cpp:138-151); contribution h * w^2 * w_i * (p - e)
    for (int e = tid; e < E; e += THREADS) {
      const int i0 = S.bend_v[e], i1 = S.bend_v[E + e], i2 = S.bend_v[2 * E + e], i3 = S.bend_v[3 * E + e];
      const float4 w = S.bend_w[e];
      const float2 nw = S.bend_nw[e];
      f3 d;
      HybridBendOp{S.h64}(ld3(xn, i0, N), ld3(xn, i1, N), ld3(xn, i2, N), ld3(xn, i3, N), ld3(vnow, i0, N), ld3(vnow, i1, N), ld3(vnow, i2, N),
                          ld3(vnow, i3, N), w, S.bend_lo[e], nw.x, nw.y, d);
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
    if (nself > 0) {   // self contacts: layered Gauss-Seidel on r (Simulation.cpp:655-678), then rebuild the right-hand side
      __syncthreads();
      self_friction_layers<THREADS>(S, A.self, b, rec_f, rec_r);
      part = 0.f;
      for (int i = tid; i < N; i += THREADS) {
        f3 rhs = ld3(rec_f, i, N) + ld3(rec_r, i, N) - ld3(vnow, i, N) * S.mass[i];
        const float di = S.dinv[i];
        st3(cg_r, i, N, rhs);
        st3(cg_p, i, N, rhs * di);
        part += dot(rhs, rhs) * di;
      }
    }
    double rz = block_sum<THREADS>((double) part, red);
    // ---- global step: P dv = rhs (Simulation.cpp:1267) ----
    const double rz_rhs = rz;
    if (S.defl_u && S.fwd_defl && rz > 1e-300) rz = deflate_global<THREADS>(S, cg_r, cg_p, cg_x, defl_scr, red);      // irregular garments (dc_devlib.h)
    cg_total += block_pcg<THREADS>(S, cg_r, cg_p, cg_ap, cg_x, rz, A.cg_tol, A.cg_max, red, rz_rhs);
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
      since_progress = 0;     // any new minimum counts: slow monotone convergence must never look like a stall
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
    s.self_contacts = nself; s.last_xdiff = (float) xdiff;
    s.self_overflow = (S.contact_enabled && S.self_enabled) ? A.self.meta[(size_t) b * kMetaStride + kMetaStride - 2] : 0;
    A.stats[b] = s;
  }
}

static int pick_threads_fwd(int N) { return N <= 1536 ? 256 : (N <= 6144 ? 512 : 1024); }

// DC_FWD_VARIANT (read once, development switch): "global" forces this file's global-memory kernel (any N);
// "0" / "1" force the ELL resident kernel (dc_forward_res.hip) in one of its two thread shapes. Default: the
// packet resident kernel (dc_forward_pk.hip) when its tables exist, else ELL resident, else global.
static int fwd_variant() {
  static int v = -3;
  if (v == -3) {
    const char *e = getenv("DC_FWD_VARIANT");
    v = !e ? -2 : (e[0] == 'g' ? -1 : atoi(e));
  }
  return v;
}

bool pd_step_fusable(const DevSystem &S) { return fwd_variant() == -2 && S.pk_ok && S.pk_vpt > 0; }

void launch_pd_step(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  const int variant = fwd_variant();
  if (variant == -2 && launch_pd_step_packet(S, W, A, B, st)) return;
  if (variant != -1 && launch_pd_step_resident(S, W, A, B, st, variant == 1 ? 1 : 0)) return;
  switch (pick_threads_fwd(S.N)) {
    case 256: hipLaunchKernelGGL(k_pd_step<256>, dim3(B), dim3(256), 0, st, S.self_dev, W, A); break;
    case 512: hipLaunchKernelGGL(k_pd_step<512>, dim3(B), dim3(512), 0, st, S.self_dev, W, A); break;
    default: hipLaunchKernelGGL(k_pd_step<1024>, dim3(B), dim3(1024), 0, st, S.self_dev, W, A); break;
  }
}

}  // namespace dc
