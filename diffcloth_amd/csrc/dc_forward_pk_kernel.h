// The one-workgroup-per-rollout PD step kernel (template) and its launcher; instantiated by dc_forward_pk.hip (plain) and
// dc_forward_pk_defl.hip (DEFL: the solves start with the spectral-deflation projection, dc_deflate.h). Two translation units because the
// projection code inside the kernel costs the PCG loop registers whether it runs or not (VGPR spills 136 -> 608 on the 10 000-vertex variant).
#pragma once
#include "dc_devlib.h"
#include "dc_winlib.h"
#include "dc_denselib.h"
#include "dc_selflib.h"
#include "dc_pklib.h"
#include <algorithm>


namespace dc {

// (Round 6, measured and not kept here: the single-reduction CG iteration the split kernels use — r.Ad, Ad.Ad and the true r.r formed in the product
// pass, |r'|^2 = r.r - 2 alpha r.Ad + alpha^2 Ad.Ad, one barrier less per iteration. This kernel is VALU-bound (0.54 busy): the three extra dot
// products per row and two more wave reductions cost more issue slots than the barrier gives back — headline forward 15.5 -> 16.1 ms, sock
// 24.9 -> 25.8 ms, dress 35.0 -> 35.8 ms per batch step. docs/HISTORY.md "Round 6".)
#ifdef DC_PROFILE_PHASES
#define PH_DECL long long ph_t = clock64(); long long ph_acc[6] = {0, 0, 0, 0, 0, 0}; if (blockIdx.x == 0 && threadIdx.x == 0) { g_win_ph[0] = g_win_ph[1] = g_win_ph[2] = g_win_ph[3] = 0; }
#define PH(k) { long long n_ = clock64(); ph_acc[k] += n_ - ph_t; ph_t = n_; }
#define PH_PRINT if (blockIdx.x == 0 && threadIdx.x == 0) printf("[phases pk] pd %d cg %d | per PD iter: local %lld vertex %lld pd-update %lld | per CG iter: spmv %lld pAp-red %lld upd+red %lld cycles | windows per PD iter: stage %lld tri %lld bend %lld vertex %lld\n", iters, cg_total, ph_acc[0] / iters, ph_acc[1] / iters, ph_acc[5] / iters, ph_acc[2] / max(cg_total, 1), ph_acc[3] / max(cg_total, 1), ph_acc[4] / max(cg_total, 1), g_win_ph[0] / iters, g_win_ph[1] / iters, g_win_ph[2] / iters, g_win_ph[3] / iters);
#else
#define PH_DECL
#define PH(k)
#define PH_PRINT
#endif

// DETECT: the self-collision detection + layering of every step is inlined (fused sweeps with self-collision). It is a
// template parameter, not a run-time branch: the mere presence of that code in the kernel changes the register allocation
// of the PCG loop (SpMV 21 k -> 29 k cycles), which runs without it must not pay for.
// H16: the search direction lives in LDS as four halves per row (x, y, z, unused) scaled by a power of two per iteration: one ds_read_b64
// per non-zero instead of a b64 + a b32, 8 instead of 12 bytes per row — the LDS this frees holds more rows of the iterate (XL), i.e. fewer
// registers. CG does not need an exact direction, only consistency: the step length is the exact line search along the direction actually
// used, alpha = <d, r> / <d, A d>, and r, x are updated with that same d, so r stays the residual of x (fp32, as before) and the stopping
// rule is unchanged; the rounding of d (2^-11 relative) costs a little conjugacy, nothing else. Needs the element windows (S.win_ok).
template <int THREADS, int VPT, int XL, bool DETECT, bool DENSE, bool H16, bool DEFL>
__global__ __launch_bounds__(THREADS) void k_pd_step_pk(const DevSystem *__restrict__ Sp, DevWork W, FwdArgs A) {
  static_assert(!(H16 && DENSE), "the explicit-inverse solve keeps the fp32 planes");
  const DevSystem &S = *Sp;
  constexpr int NP = THREADS * VPT;
  constexpr int WAVES = THREADS / 64;
  constexpr int XR = VPT - XL;
  constexpr int PF = H16 ? 2 : 3;        // floats per row of the search direction in LDS
  extern __shared__ float lp[];          // search direction: float2 (x, y) [NP] then float z [NP] (H16: h4 [NP]); then x rows [XL][3][THREADS]
  float *lx = lp + PF * NP;
  h4 *lh = (h4 *) lp;
  const unsigned lh_addr = lds_byte_address(lp);
  float *ldense = lp + 3 * THREADS * (VPT + XL);      // DENSE: partial sums of the product with the explicit inverse
  __shared__ double red[THREADS / 64];
  __shared__ double red2[H16 ? 2 * (THREADS / 64) : 1];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = S.N, T = S.T, E = S.E, NC = S.NC;
  const size_t off = (size_t) b * 3 * N;
  float *g = W.g + off, *vnow = W.vnow + off, *vbest = W.vbest + off;
  float *dprev = W.cg_x + off;                // scaled correction of the previous PD iteration (first search direction of the next solve)
  float *corner = W.corner + (size_t) b * 3 * NC;
  // A.nsteps consecutive time steps of this rollout in one launch (dc_rollout_forward without self-collision): rollouts
  // are independent, so nothing forces them to wait for the slowest one after every step. Step s reads tape slot k + s
  // and writes slot k + s + 1 (slot strides: A.slot_state floats, A.slot_prim ints, A.slot_stats entries).
  // H16 product: packet-row address and packet count of the wave's k-th chunk, held in lane k for the whole launch (see spmv)
  unsigned tbl_lo = 0, tbl_hi = 0;
  int tbl_n = 0;
  if constexpr (H16) {
    const int ch = wv + min(lane, VPT - 1) * WAVES;
    const unsigned long long a = (unsigned long long) (S.pk + S.pk_ptr[ch]);
    tbl_lo = (unsigned) a; tbl_hi = (unsigned) (a >> 32); tbl_n = S.pk_n[ch];
  }
  for (int step = 0; step < A.nsteps; step++) {
  if (step > 0) __syncthreads();              // the previous step's state written by the whole workgroup
  const size_t so = (size_t) step * A.slot_state;
  const float *xn = A.x_in + off + so, *vn = A.v_in + off + so;
  float *rec_f = A.rec_f + off + so, *rec_r = A.rec_r + off + so, *rec_n = A.rec_n + off + so;
  int *rec_prim = A.rec_prim + (size_t) b * N + (size_t) step * A.slot_prim;
  // this step's fixed-point targets and external forces (constant over the launch, or one set per step: dc_set_*_schedule)
  const float *xfix = A.x_fixed + (size_t) step * A.slot_xfix + (size_t) b * 3 * S.Af;
  const float *fu_s = A.fu ? A.fu + (size_t) step * A.slot_fu : nullptr;
  const float *fvs_s = A.fv_scale ? A.fv_scale + (size_t) step * A.slot_fvs : nullptr;
  SelfRec srec = A.self;                      // self contacts of this step's record
  srec.pair += (size_t) step * A.slot_self; srec.nrm += (size_t) step * A.slot_self; srec.dvec += (size_t) step * A.slot_self;
  srec.meta += (size_t) step * A.slot_meta; srec.verts += (size_t) step * 2 * A.slot_self;
  if constexpr (DETECT) {                     // fused sweeps: detection + layering of this step run here (dc_selflib.h)
    self_detect_rollout<THREADS>(S, W, b, A.x_in + so, A.v_in + so, A.rec_prim + (size_t) step * A.slot_prim, srec, fu_s, A.fv, fvs_s, (int *) lp, A.fv2);
    __syncthreads();
  }
  const float *mu = A.mu + (size_t) b * S.ngroups;
  const float h = S.h;
  const f3 grav = mk(S.gx, S.gy, S.gz);
  const f3 fu = fu_s ? mk(fu_s[3 * b], fu_s[3 * b + 1], fu_s[3 * b + 2]) : mk(0, 0, 0);
  const float fvs = fvs_s ? fvs_s[b] : 1.f;

  // ---- step set-up: s_n, initial guess, contact detection (Simulation.cpp:1097-1160, :1254-1256) ----
  float part = 0.f;
  int ncontact = 0;
  for (int i = tid; i < N; i += THREADS) {
    const float m = S.mass[i];
    f3 v = ld3(vn, i, N);
    f3 fext = grav * m + fu;                      // fillForces (Simulation.cpp:55-116)
    if (A.fv) fext = fext + ld3(A.fv + off, i, N) * fvs;
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
  const int nself = (S.contact_enabled && S.self_enabled) ? srec.meta[(size_t) b * kMetaStride] : 0;   // from the detection pass
  bool improved = false, converged = false, stalled = false, best_is_current = false;
  int iters = 0, cg_total = 0, since_progress = 0;
  double xdiff = 0;
  float dnorm = 0.f;                        // H16: |d_prev|_2, the scaled correction of the previous PD iteration

  PH_DECL
  for (int iter = 0; iter < A.pd_cap; iter++) {
    // opaque zero, refreshed per PD iteration: the per-row indices of the unrolled loops below are loop invariant, and
    // LICM would hoist ~20 rows x several arrays of them out of the PD loop into registers that do not exist (357
    // dwords spilled at the loop head and reloaded in every phase)
    int zp;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zp));
    const int tq = tid + zp;
    // per-vertex part of the step given the summed element forces of the vertex: f, friction r, scaled right-hand
    // side of the correction solve
    auto vertex_body = [&](int i, f3 fint) -> f3 {
      f3 f = ld3(g, i, N) + fint;
      f3 v = ld3(vnow, i, N);
      const int a = S.att_of_vertex[i];
      if (a >= 0) f = f + ((ld3(xfix, a, S.Af) - ld3(xn, i, N)) - v * h) * (h * S.k_att);   // AttachmentSpring.cpp:25-29
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
      return (f + r - v * m) * S.sq_dinv[i];       // scaled residual D^-1/2 rhs
    };
    part = 0.f;
    bool self_done = false;
    if (S.win_ok) {
      // ---- local step + vertex pass, window by window inside LDS (dc_winlib.h) ----
      float *scr = W.cg_r + off;
#ifndef DC_FWD_NO_VPRE
      // the vertex's unconditional global reads, issued by the per-vertex phase ahead of the gather (dc_winlib.h, vert_with_pre)
      struct VIn { f3 g, v; int a; float m; int prim; float sq; };
      auto vert = vert_with_pre_noa([&](int i) {
        VIn q;
        q.g = ld3(g, i, N); q.v = ld3(vnow, i, N); q.a = S.att_of_vertex[i]; q.m = S.mass[i]; q.prim = rec_prim[i]; q.sq = S.sq_dinv[i];
        return q;
      }, [&](int i, f3 fint, f3, const VIn &q) {
        f3 f = q.g + fint;
        if (q.a >= 0) f = f + ((ld3(xfix, q.a, S.Af) - ld3(xn, i, N)) - q.v * h) * (h * S.k_att);   // AttachmentSpring.cpp:25-29
        f3 r = mk(0, 0, 0);
        if (q.prim >= 0) {  // calculateDryFrictionVector, primitive part (Simulation.cpp:640-652)
          f3 n = ld3(rec_n, i, N);
          f3 d = f - prim_vout(S.prims[q.prim], n) * q.m;
          r = dry_friction(n, d, mu[S.prims[q.prim].group]);
        }
        st3(rec_f, i, N, f);
        st3(rec_r, i, N, r);
        f3 rhs = (f + r - q.v * q.m) * q.sq;       // scaled residual D^-1/2 rhs
        st3(scr, i, N, rhs);
        part += dot(rhs, rhs);
      });
#else
      auto vert = [&](int i, f3 sum, f3) {
        f3 rhs = vertex_body(i, sum);
        st3(scr, i, N, rhs);
        part += dot(rhs, rhs);
      };
#endif
      element_windows<THREADS, kFwdOpsPrecise>(S, lp, StagePlanar{xn, N}, vnow, fwd_tri_op(h, S.h64), fwd_bend_op(h, S.h64), vert);   // fp64-strain operators (dc_winlib.h)
      __syncthreads();
      PH(0)
      if (nself > 0 && !A.self_full) {
        // self contacts: layered Gauss-Seidel on r (Simulation.cpp:655-678) over the working set of the contacts, in the LDS the windows
        // have just left; then the right-hand side of those ~2 x nself vertices alone is formed again (all N before: a quarter
        // of the per-vertex phase on the 10 000-vertex cloth with 500 contacts)
        if (self_friction_layers_lds<THREADS>(S, srec, b, rec_f, rec_r, lp, 3 * NP)) {     // ends with a barrier
          const int M = srec.meta[(size_t) b * kMetaStride + kMetaStride - 1];
          const int *verts = srec.verts + (size_t) b * 2 * S.self_cap;
          for (int q = tid; q < M; q += THREADS) {
            const int i = verts[q];
            st3(scr, i, N, (ld3(rec_f, i, N) + ld3(rec_r, i, N) - ld3(vnow, i, N) * S.mass[i]) * S.sq_dinv[i]);
          }
          __syncthreads();
          self_done = true;
          part = 0.f;
        }
      }
      if constexpr (!H16) {
      const PlaneBuf tb0 = plane_buf(scr, N), tb1 = plane_buf(scr + N, N), tb2 = plane_buf(scr + 2 * N, N);      // (rows past N read 0)
      for (int k0 = 0; k0 < VPT; k0 += 4) {     // 4 rows = 12 loads in flight per thread, then the LDS stores
        float t[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int i = tq + min(k0 + j, VPT - 1) * THREADS;
          t[j][0] = tb0.ld(i); t[j][1] = tb1.ld(i); t[j][2] = tb2.ld(i);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (k0 + j < VPT) {
            const int i = tq + (k0 + j) * THREADS;
            ((float2 *) lp)[i] = make_float2(t[j][0], t[j][1]); lp[2 * NP + i] = t[j][2];
            if (self_done) part += dot(mk(t[j][0], t[j][1], t[j][2]), mk(t[j][0], t[j][1], t[j][2]));
          }
        }
      }
      }
    } else {
      // ---- local step: per-element projection residual, written per constraint corner (global memory) ----
      for (int t = tid; t < T; t += THREADS) {      // Triangle::project (Triangle.cpp:310-351)
        const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
        f3 r0, r1;
        const float4 D = S.tri_D[t];
        HybridTriOp{S.h64}(ld3(xn, i0, N), ld3(xn, i1, N), ld3(xn, i2, N), ld3(vnow, i0, N), ld3(vnow, i1, N), ld3(vnow, i2, N), D, S.tri_Dlo[t], S.tri_w2[t], r0, r1);
        f3 c1 = r0 * D.x + r1 * D.y, c2 = r0 * D.z + r1 * D.w;
        st3(corner, t, NC, mk(0, 0, 0) - c1 - c2); st3(corner, T + t, NC, c1); st3(corner, 2 * T + t, NC, c2);
      }
      for (int e = tid; e < E; e += THREADS) {      // TriangleBending::project (TriangleBending.cpp:138-151)
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
      PH(0)
      for (int i = tid; i < NP; i += THREADS) {
        f3 rhs = mk(0, 0, 0);
        if (i < N) {
          f3 fint = mk(0, 0, 0);
          const int k1 = S.inc_ptr[i + 1];
          for (int q = S.inc_ptr[i]; q < k1; q++) fint = fint + ld3(corner, S.inc_idx[q], NC);
          rhs = vertex_body(i, fint);
          part += dot(rhs, rhs);
        }
        ((float2 *) lp)[i] = make_float2(rhs.x, rhs.y); lp[2 * NP + i] = rhs.z;
      }
    }
    if (nself > 0 && !self_done) {   // (working set beyond the LDS, or no element windows) the same through global memory, then rebuild the right-hand side
      __syncthreads();
      // (the LDS version uses the search-direction planes as scratch: they are rebuilt, padding rows included, below)
      if (!self_friction_layers_lds<THREADS>(S, srec, b, rec_f, rec_r, lp, 3 * NP)) self_friction_layers<THREADS>(S, srec, b, rec_f, rec_r);
      part = 0.f;
      for (int i = tid; i < NP; i += THREADS) {
        f3 rhs = mk(0, 0, 0);
        if (i < N) rhs = (ld3(rec_f, i, N) + ld3(rec_r, i, N) - ld3(vnow, i, N) * S.mass[i]) * S.sq_dinv[i];
        if constexpr (H16) { if (i < N) st3(W.cg_r + off, i, N, rhs); }
        else { ((float2 *) lp)[i] = make_float2(rhs.x, rhs.y); lp[2 * NP + i] = rhs.z; }
        part += dot(rhs, rhs);
      }
      if constexpr (H16) __syncthreads();
    }
    // residual, A p and (most of) the iterate of the scaled CG live in registers from here to the update
    float rr[VPT][3], ap[VPT][3], xx[XR > 0 ? XR : 1][3];
    if constexpr (H16) {      // the right-hand side goes from the work array straight into the residual registers
      // (range-checked buffer loads: rows past N read 0, a row costs one shift for its three addresses — as plain loads the clamped
      // 64-bit indices of the 20 rows were common subexpressions of every row loop of the PD iteration, lived across the CG loop in
      // scratch, and each reload was an `s_waitcnt vmcnt(0)` in the middle of the loads it fed)
      const float *scr = W.cg_r + off;
      const PlaneBuf sb0 = plane_buf(scr, N), sb1 = plane_buf(scr + N, N), sb2 = plane_buf(scr + 2 * N, N);
      part = 0.f;
#pragma unroll
      for (int k = 0; k < VPT; k++) {
        const int i = tq + k * THREADS;
        rr[k][0] = sb0.ld(i); rr[k][1] = sb1.ld(i); rr[k][2] = sb2.ld(i);
      }
#pragma unroll
      for (int k = 0; k < VPT; k++) part = fmaf(rr[k][0], rr[k][0], fmaf(rr[k][1], rr[k][1], fmaf(rr[k][2], rr[k][2], part)));
    }
    double rz = block_sum<THREADS>((double) part, red);
    float hs = 1.f, pn = 0.f;               // H16: scale of the direction in LDS (a power of two) and the bound on its entries it comes from
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int i = tq + k * THREADS;
      if constexpr (!H16) {
        const float2 q = ((const float2 *) lp)[i];
        rr[k][0] = q.x; rr[k][1] = q.y; rr[k][2] = lp[2 * NP + i];
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        if (k < XR) xx[k < XR ? k : 0][c] = 0.f;
        else lx[((k - XR) * 3 + c) * THREADS + tq] = 0.f;
      }
    }
    // Spectral deflation (dc_deflate.h; irregular garments only: S.defl_u is null otherwise): Galerkin projection of the current residual
    // onto the 16 lowest eigenvectors U of the scaled matrix, c = (U^T A U)^-1 U^T r, x += U c, r -= (A U) c. Done once per solve — after
    // the recycled first direction when there is one — it leaves a residual orthogonal to an (almost) invariant subspace, and the Krylov
    // space CG then builds stays orthogonal to it: the smooth modes that make Jacobi-PCG need 260 ... 340 iterations on the reference's fine
    // dress are solved exactly up front. Uses the start of the direction's LDS as scratch (the caller rewrites the direction afterwards).
    const double rz_b = rz;                 // |b|^2: the stopping rule stays relative to the right-hand side
    constexpr bool defl = DEFL;
    auto deflate = [&](double &rzv) {
      constexpr int DK = 16;
      float *scr = lp;                      // [WAVES][48] wave sums, then [48] U^T r, then [48] c
      const float4 DC_G *U4 = (const float4 DC_G *) S.defl_u;
      const float4 DC_G *AU4 = (const float4 DC_G *) S.defl_au;
      __syncthreads();
#pragma unroll 1
      for (int c = 0; c < 3; c++) {
        float acc[DK];
#pragma unroll
        for (int j = 0; j < DK; j++) acc[j] = 0.f;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int i = tq + k * THREADS;
          const float rv = c == 0 ? rr[k][0] : (c == 1 ? rr[k][1] : rr[k][2]);
#pragma unroll
          for (int q = 0; q < DK / 4; q++) {
            const float4 u = U4[(size_t) i * (DK / 4) + q];
            acc[4 * q] = fmaf(u.x, rv, acc[4 * q]); acc[4 * q + 1] = fmaf(u.y, rv, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(u.z, rv, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(u.w, rv, acc[4 * q + 3]);
          }
        }
#pragma unroll
        for (int j = 0; j < DK; j++) {
          const float v = wave_sum_f(acc[j]);
          if (lane == 0) scr[wv * 48 + c * DK + j] = v;
        }
      }
      __syncthreads();
      if (tid < 48) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; w++) sacc += scr[w * 48 + tid];
        scr[WAVES * 48 + tid] = sacc;
      }
      __syncthreads();
      if (tid < 48) {
        const int c = tid / DK, j = tid % DK;
        float sacc = 0.f;
#pragma unroll
        for (int l = 0; l < DK; l++) sacc = fmaf(S.defl_g[j * DK + l], scr[WAVES * 48 + c * DK + l], sacc);
        scr[WAVES * 48 + 48 + tid] = sacc;
      }
      __syncthreads();
      float partd = 0.f;
#pragma unroll 1
      for (int c = 0; c < 3; c++) {
        float cc[DK];
#pragma unroll
        for (int j = 0; j < DK; j++) cc[j] = scr[WAVES * 48 + 48 + c * DK + j];
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int i = tq + k * THREADS;
          float dx = 0.f, dr = 0.f;
#pragma unroll
          for (int q = 0; q < DK / 4; q++) {
            const float4 u = U4[(size_t) i * (DK / 4) + q], a = AU4[(size_t) i * (DK / 4) + q];
            dx = fmaf(u.x, cc[4 * q], fmaf(u.y, cc[4 * q + 1], fmaf(u.z, cc[4 * q + 2], fmaf(u.w, cc[4 * q + 3], dx))));
            dr = fmaf(a.x, cc[4 * q], fmaf(a.y, cc[4 * q + 1], fmaf(a.z, cc[4 * q + 2], fmaf(a.w, cc[4 * q + 3], dr))));
          }
          if (c == 0) { rr[k][0] -= dr; partd = fmaf(rr[k][0], rr[k][0], partd); }
          else if (c == 1) { rr[k][1] -= dr; partd = fmaf(rr[k][1], rr[k][1], partd); }
          else { rr[k][2] -= dr; partd = fmaf(rr[k][2], rr[k][2], partd); }
          if (k < XR) { if (c == 0) xx[k < XR ? k : 0][0] += dx; else if (c == 1) xx[k < XR ? k : 0][1] += dx; else xx[k < XR ? k : 0][2] += dx; }
          else lx[((k - XR) * 3 + c) * THREADS + tq] += dx;
        }
      }
      rzv = block_sum<THREADS>((double) partd, red);      // (its barriers also end the use of the scratch)
    };
    if (defl && !(A.cg_seed && iter > 0) && rz > 1e-300) {      // no recycled direction in this solve: project first
      deflate(rz);
      if constexpr (!H16) {                 // the direction planes held r before the scratch use: p0 = the projected r
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int i = tq + k * THREADS;
          ((float2 *) lp)[i] = make_float2(rr[k][0], rr[k][1]); lp[2 * NP + i] = rr[k][2];
        }
      }
    }
    if constexpr (H16) {                    // first direction d = r (|r|_inf <= |r|_2 = sqrt(rz)), rounded to halves
      pn = sqrtf((float) rz); hs = half_scale(pn);
#pragma unroll
      for (int k = 0; k < VPT; k++) lh[tq + k * THREADS] = pack_h4(rr[k][0] * hs, rr[k][1] * hs, rr[k][2] * hs);
    }
    PH(1)
    // ap = Ahat * (the vector in lp), part2 += <lp, ap>; rows of a thread tid + k * THREADS
    auto spmv = [&](int wz, float &part2, bool with_pr, float &part3) {
      if constexpr (H16) {
        // Row table from the lanes (tbl_*: lane k = the wave's k-th chunk): one v_readlane per value instead of scalar loads of pk_ptr / pk_n
        // and of the table pointers themselves in every row — each of those was an `s_waitcnt lgkmcnt(0)`, i.e. an exposed scalar-cache
        // round trip that also drains the LDS gathers in flight (94 scalar loads per product in the round-4 code object). The lane
        // index carries the opaque zero so that the 3 x VPT values are not hoisted out of the CG loop into SGPRs that do not exist.
        const int zl = wz - wv;
        auto row_of = [&](int k) -> const int4 DC_G * {
          const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) tbl_lo, k + zl), hi = (unsigned) __builtin_amdgcn_readlane((int) tbl_hi, k + zl);
          return (const int4 DC_G *) (((unsigned long long) hi << 32) | lo) + lane;
        };
        int4 nxt[PB];
        const int4 DC_G *row = row_of(0);
#pragma unroll
        for (int j = 0; j < PB; j++) nxt[j] = row[j * 64];
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int i = (wz + k * WAVES) * 64 + lane;
          const int np = __builtin_amdgcn_readlane(tbl_n, k + zl);
          int4 cur[PB];
#pragma unroll
          for (int j = 0; j < PB; j++) cur[j] = nxt[j];
          const int4 DC_G *row_next = row;
          if (k + 1 < VPT) {
            row_next = row_of(k + 1);
#pragma unroll
            for (int j = 0; j < PB; j++) nxt[j] = row_next[j * 64];
          }
          unsigned rowbase = lh_addr + 8u * (unsigned) (i - 512);
          asm volatile("" : "+v"(rowbase));      // opaque: one register per row, not (row + delta) * 8 + LDS base per non-zero
          float ax, ay, az;
          pk_v2i own;
          consume_h_row(cur, rowbase, ax, ay, az, own);
          for (int s0 = PB; s0 < np; s0 += PB) {        // rows wider than one batch
#pragma unroll
            for (int j = 0; j < PB; j++) cur[j] = row[(s0 + j) * 64];
            consume_h(cur, rowbase, ax, ay, az);
          }
          row = row_next;
          ap[k][0] = ax; ap[k][1] = ay; ap[k][2] = az;
          dot3_h(ax, ay, az, own, part2);
          dot3_h(rr[k][0], rr[k][1], rr[k][2], own, part3);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      int4 nxt[PB];
      load_batch(nxt, S.pk + S.pk_ptr[wz] + lane, 0);
#pragma unroll
      for (int k = 0; k < VPT; k++) {
        const int chunk = wz + k * WAVES;   // wave-uniform: pk_ptr / pk_n are scalar loads
        const int i = chunk * 64 + lane;
        const int np = S.pk_n[chunk];
        const int4 *row = S.pk + S.pk_ptr[chunk] + lane;
        int4 cur[PB];
#pragma unroll
        for (int j = 0; j < PB; j++) cur[j] = nxt[j];
        if (k + 1 < VPT) load_batch(nxt, S.pk + S.pk_ptr[chunk + WAVES] + lane, 0);
        const float2 pxy = ((const float2 *) lp)[i];
        const float pz = lp[2 * NP + i];
        float ax = pxy.x, ay = pxy.y, az = pz;        // unit diagonal
        const int base = i - 512;
        consume<NP>(cur, lp, base, ax, ay, az);
        for (int s0 = PB; s0 < np; s0 += PB) {        // rows wider than one batch
          load_batch(cur, row, s0);
          consume<NP>(cur, lp, base, ax, ay, az);
        }
        ap[k][0] = ax; ap[k][1] = ay; ap[k][2] = az;
        part2 += pxy.x * ax + pxy.y * ay + pz * az;
        if (with_pr) part3 += pxy.x * rr[k][0] + pxy.y * rr[k][1] + pz * rr[k][2];      // seeded pass only (uniform branch)
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    };
    // ---- global step: Jacobi PCG on P dv = rhs as plain CG on the scaled system, resident in LDS + registers ----
    if constexpr (DENSE) {
      // ---- small meshes: dv = Ahat^-1 rhs by the explicit fp32 inverse (dc_denselib.h) + iterative refinement with the
      // packet SpMV until the same stopping rule holds (relative residual <= cg_tol); lp holds the current residual ----
      if (rz > 1e-300) {
        const double stop = (double) A.cg_tol * (double) A.cg_tol * rz;
        for (int it = 0; it < A.cg_max;) {
          __syncthreads();
          int zs;
          asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
          const int wz = wv + zs, tz = tid + zs;
          const int C = dense_partials<THREADS>(S, (const float2 *) lp, lp + 2 * NP, ldense);
          __syncthreads();
#pragma unroll
          for (int k = 0; k < VPT; k++) {            // z = Ahat^-1 r : accumulate into the iterate, hand to the SpMV
            const int i = tz + k * THREADS;
            const f3 z = dense_row_sum(ldense, S.dense_ld, C, i);
            xx[k][0] += z.x; xx[k][1] += z.y; xx[k][2] += z.z;
            ((float2 *) lp)[i] = make_float2(z.x, z.y); lp[2 * NP + i] = z.z;
          }
          __syncthreads();
          float part2 = 0.f;
          float nopr = 0.f;
          spmv(wz, part2, false, nopr);
          part2 = 0.f;
#pragma unroll
          for (int k = 0; k < VPT; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) { rr[k][c] -= ap[k][c]; part2 = fmaf(rr[k][c], rr[k][c], part2); }
          const double rz_new = block_sum_f<THREADS>(part2, red);      // (its barriers order the lp reads above and the writes below)
          it++; cg_total++;
          if (!(rz_new > stop)) break;
#pragma unroll
          for (int k = 0; k < VPT; k++) {
            const int i = tz + k * THREADS;
            ((float2 *) lp)[i] = make_float2(rr[k][0], rr[k][1]); lp[2 * NP + i] = rr[k][2];
          }
        }
      }
    } else
    if (rz_b > 1e-300 && rz > (double) A.cg_tol * (double) A.cg_tol * rz_b) {
      const double stop = (double) A.cg_tol * (double) A.cg_tol * rz_b;
      // Recycled first direction (A.cg_seed): successive PD iterations of a step produce strongly correlated corrections, so the
      // previous solution d is a far better first search direction than the residual: x = gamma d with gamma = <d, r> / <d, A d>
      // (the energy-norm minimiser along d), then ordinary CG restarted from the new residual (beta = 0). One extra product,
      // and the relative stopping rule (against the right-hand side) is met several iterations earlier.
      bool seed = A.cg_seed && iter > 0;
      if (seed) {
        if constexpr (H16) hs = half_scale(dnorm);      // |d_prev|_2 from the update loop of the previous PD iteration
        const PlaneBuf pb0 = plane_buf(dprev, N), pb1 = plane_buf(dprev + N, N), pb2 = plane_buf(dprev + 2 * N, N);
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int i = tq + k * THREADS;
          const float d0 = pb0.ld(i), d1 = pb1.ld(i), d2 = pb2.ld(i);      // (0 past N)
          if constexpr (H16) lh[i] = pack_h4(d0 * hs, d1 * hs, d2 * hs);
          else { ((float2 *) lp)[i] = make_float2(d0, d1); lp[2 * NP + i] = d2; }
        }
      }
      for (int it = 0; it < A.cg_max;) {
        __syncthreads();
        float part2 = 0.f, part3 = 0.f;
        int zs;                               // opaque zero: keeps the per-row addresses out of LICM's reach (they
        asm volatile("s_mov_b32 %0, 0" : "=s"(zs));   // would be hoisted into ~60 live registers otherwise)
        const int wz = wv + zs, tz = tid + zs;
        spmv(wz, part2, seed, part3);
        PH(2)
        double pAp, pr = rz;
        if constexpr (H16) block_sum2_f_nb<THREADS>(part2, part3, red2, pAp, pr);      // exact line search along the rounded direction
        else {
          pAp = block_sum_f<THREADS>(part2, red);
          if (seed) pr = block_sum_f<THREADS>(part3, red);
        }
        PH(3)
        const float alpha = pAp > 1e-300 ? (float) (pr / pAp) : 0.f;
        part2 = 0.f;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int i = tz + k * THREADS;
          float pv[3];
          if constexpr (H16) { const h4 q = lh[i]; pv[0] = (float) q.x; pv[1] = (float) q.y; pv[2] = (float) q.z; }
          else { const float2 pxy = ((const float2 *) lp)[i]; pv[0] = pxy.x; pv[1] = pxy.y; pv[2] = lp[2 * NP + i]; }
#pragma unroll
          for (int c = 0; c < 3; c++) {
            if (k < XR) xx[k < XR ? k : 0][c] = fmaf(alpha, pv[c], xx[k < XR ? k : 0][c]);
            else lx[((k - XR) * 3 + c) * THREADS + tz] = fmaf(alpha, pv[c], lx[((k - XR) * 3 + c) * THREADS + tz]);
            rr[k][c] = fmaf(-alpha, ap[k][c], rr[k][c]);
            part2 = fmaf(rr[k][c], rr[k][c], part2);
          }
        }
        double rz_new = H16 ? block_sum_f_nb<THREADS>(part2, red) : block_sum_f<THREADS>(part2, red);
        it++; cg_total++;
        if (!(rz_new > stop)) break;
        if (defl && seed) {                 // after the recycled direction: project the residual, then plain CG from it (beta = 0)
          deflate(rz_new);
          if (!(rz_new > stop)) break;
        }
        const float beta = seed ? 0.f : (float) (rz_new / rz);
        seed = false;
        rz = rz_new;
        if constexpr (H16) {
          // d_new = r + beta d_old in true units; in LDS units: hs_new r + (beta hs_new / hs_old) d~_old, entries bounded by |r|_2 + beta * bound_old
          pn = sqrtf((float) rz_new) + beta * pn;
          const float hs_new = half_scale(pn), c2 = beta * hs_new / hs;
          hs = hs_new;
          // DEFL: deflate() has used the first rows of the direction's LDS as float scratch; after it beta = 0 and the old direction
          // must not be READ as halves at all (a float's bits can be an Inf / NaN half, and 0 * Inf is NaN): its bits are masked off
          // (tests/test_gpu_configs.py::test_forced_deflation_on_the_bench_cloth)
          [[maybe_unused]] const int keep = (beta != 0.f) ? -1 : 0;
#pragma unroll
          for (int k = 0; k < VPT; k++) {
            const int i = tz + k * THREADS;
            h4 q = lh[i];
            if constexpr (DEFL) {
              int2 b;
              __builtin_memcpy(&b, &q, 8);
              b.x &= keep; b.y &= keep;
              __builtin_memcpy(&q, &b, 8);
            }
            lh[i] = pack_h4(fmaf(c2, (float) q.x, rr[k][0] * hs), fmaf(c2, (float) q.y, rr[k][1] * hs), fmaf(c2, (float) q.z, rr[k][2] * hs));
          }
        } else {
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int i = tz + k * THREADS;
          const float2 pxy = ((const float2 *) lp)[i];
          ((float2 *) lp)[i] = make_float2(fmaf(beta, pxy.x, rr[k][0]), fmaf(beta, pxy.y, rr[k][1]));
          lp[2 * NP + i] = fmaf(beta, lp[2 * NP + i], rr[k][2]);
        }
        }
        PH(4)
      }
    }
    // ---- update + convergence (Simulation.cpp:1268, 1310-1373) ----
    // rows in groups of 4: loads (clamped index, no divergence), arithmetic, stores; delta v replaces A p in its
    // registers and stays there for the best-iterate bookkeeping below
    part = 0.f;
    float partd = 0.f;
    // (stores through range-checked buffer resources: the rows past N of the last chunks are dropped by the hardware, and the recycled
    // direction's array has no records at all when the seed is off — guarded with `if (i < N)` every one of the 3 x VPT stores was an
    // exec-masked block of its own with its address arithmetic inside, 46 ... 60 k cycles per PD iteration)
    const PlaneBuf vb0 = plane_buf(vnow, N), vb1 = plane_buf(vnow + N, N), vb2 = plane_buf(vnow + 2 * N, N);
    const PlaneBuf qb = plane_buf((const float *) S.sq_dinv, N);
    const int nd = A.cg_seed ? N : 0;
    const PlaneBuf db0 = plane_buf(dprev, nd), db1 = plane_buf(dprev + N, nd), db2 = plane_buf(dprev + 2 * N, nd);
#pragma unroll
    for (int k0 = 0; k0 < VPT; k0 += 4) {
      float vq[4][3], sq[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (k0 + j < VPT) {
          const int i = tq + (k0 + j) * THREADS;
          sq[j] = qb.ld(i);
          vq[j][0] = vb0.ld(i); vq[j][1] = vb1.ld(i); vq[j][2] = vb2.ld(i);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (k0 + j < VPT) {
          const int k = k0 + j;
          const int i = tq + k * THREADS;
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float xs = (k < XR) ? xx[k < XR ? k : 0][c] : lx[((k - XR) * 3 + c) * THREADS + tq];
            ap[k][c] = xs * sq[j];             // delta v (A p is dead here)
            (c == 0 ? vb0 : (c == 1 ? vb1 : vb2)).st(i, vq[j][c] + ap[k][c]);
            (c == 0 ? db0 : (c == 1 ? db1 : db2)).st(i, xs);
            if (i < N) { part = fmaf(ap[k][c], ap[k][c], part); if constexpr (H16) partd = fmaf(xs, xs, partd); }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (H16) {
      double sp, sd;
      block_sum2_f<THREADS>(part, partd, red2, sp, sd);
      xdiff = (double) h * sqrt(sp) / (double) N;
      dnorm = sqrtf((float) sd);
    } else
    xdiff = (double) h * sqrt(block_sum<THREADS>((double) part, red)) / (double) N;
    PH(5)
    iters = iter + 1;
    converged = xdiff < (double) A.fwd_tol;
    if (xdiff < min_xdiff) {
      since_progress = 0;     // any new minimum counts: slow monotone convergence must never look like a stall
      min_xdiff = xdiff;
      improved = true;
      best_is_current = true;            // the best iterate is v itself: nothing to copy while the run keeps improving
    } else if (best_is_current) {
      // first non-improving iteration after a minimum: the best iterate is the previous one = v - delta (delta is still in registers)
      best_is_current = false;
#pragma unroll
      for (int k = 0; k < VPT; k++) {
        const int i = tq + k * THREADS;
        if (i < N) { vbest[i] = vnow[i] - ap[k][0]; vbest[N + i] = vnow[N + i] - ap[k][1]; vbest[2 * N + i] = vnow[2 * N + i] - ap[k][2]; }
      }
    }
    if (converged) break;
    if (++since_progress >= A.stall_window) { stalled = true; break; }   // fp32 floor, see dc_forward.hip
  }
  // ---- write the new state (revert to the best iterate when the cap was hit, Simulation.cpp:1357-1367) ----
  float *xo = A.x_out + off + so, *vo = A.v_out + off + so;
  for (int i = tid; i < N; i += THREADS) {
    f3 x = ld3(xn, i, N);
    if (converged) { f3 v = ld3(vnow, i, N); st3(vo, i, N, v); st3(xo, i, N, x + v * h); }
    else if (improved) { f3 v = ld3(best_is_current ? vnow : vbest, i, N); st3(vo, i, N, v); st3(xo, i, N, x + v * h); }
    else { st3(vo, i, N, ld3(vn, i, N)); st3(xo, i, N, x); }
  }
  if (tid == 0) {
    dc_step_stats s;
    s.converged = converged ? 1 : (stalled ? 2 : 0); s.pd_iters = iters; s.cg_iters = cg_total; s.prim_contacts = total_contacts;
    s.self_contacts = nself; s.last_xdiff = (float) xdiff;
    s.self_overflow = (S.contact_enabled && S.self_enabled) ? srec.meta[(size_t) b * kMetaStride + kMetaStride - 2] : 0;
    A.stats[b + (size_t) step * A.slot_stats] = s;
  }
  PH_PRINT
  }   // step
}

template <int THREADS, int VPT, int XL, bool DETECT, bool DENSE, bool H16 = false, bool DEFL = false>
static void launch_pk_inst(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  size_t lds = (size_t) THREADS * ((H16 ? 2 : 3) * VPT + 3 * XL) * sizeof(float);
  if (DENSE) lds += sizeof(float) * (size_t) dense_lds_floats(S.dense_ld, THREADS / 64);
  if (S.win_ok) lds = std::max(lds, (size_t) S.win_lds_bytes);
  if (A.inline_detect) lds = std::max(lds, sizeof(int) * (size_t) kSelfDetectLdsInts);
  static size_t configured[kMaxDevices] = {};        // the attribute is per device: one entry per device this process has used
  int dev = 0;
  (void) hipGetDevice(&dev);
  size_t &done = configured[dev >= 0 && dev < kMaxDevices ? dev : 0];
  if (lds > done || dev >= kMaxDevices) {
    (void) hipFuncSetAttribute((const void *) k_pd_step_pk<THREADS, VPT, XL, DETECT, DENSE, H16, DEFL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    done = lds;
  }
  hipLaunchKernelGGL((k_pd_step_pk<THREADS, VPT, XL, DETECT, DENSE, H16, DEFL>), dim3(B), dim3(THREADS), lds, st, S.self_dev, W, A);
}

template <int THREADS, int VPT, int XL>
static void launch_pk(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  if constexpr (VPT <= 3) {     // small meshes: the explicit-inverse solve when the engine built it (dc_dense.h)
    if (S.dense_inv) {
      if (A.inline_detect) launch_pk_inst<THREADS, VPT, XL, true, true>(S, W, A, B, st);
      else launch_pk_inst<THREADS, VPT, XL, false, true>(S, W, A, B, st);
      return;
    }
  }
  if (A.inline_detect) launch_pk_inst<THREADS, VPT, XL, true, false>(S, W, A, B, st);
  else launch_pk_inst<THREADS, VPT, XL, false, false>(S, W, A, B, st);
}
// the half-precision-direction variant (needs the element windows): XL = rows of the iterate in the LDS the 8-byte direction rows free
template <int THREADS, int VPT, int XL>
static void launch_pk_h16(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  if (A.inline_detect) launch_pk_inst<THREADS, VPT, XL, true, false, true>(S, W, A, B, st);
  else launch_pk_inst<THREADS, VPT, XL, false, false, true>(S, W, A, B, st);
}

}  // namespace dc
