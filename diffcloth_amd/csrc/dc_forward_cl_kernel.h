// The split forward-step kernel (template) and its launcher; instantiated by dc_forward_cl.hip (plain) and dc_forward_cl_defl.hip (DEFL: the
// solves start with the spectral-deflation projection, dc_deflate.h) — two translation units, see dc_forward_pk_kernel.h.
#pragma once
#include "dc_devlib.h"
#include "dc_winlib.h"
#include "dc_selflib.h"
#include "dc_pklib.h"
#include "dc_cluster.h"
#include <algorithm>
#include <cstdlib>

namespace dc {

#ifdef DC_PROFILE_PHASES
#define CPH_DECL long long cph_t = clock64(); long long cph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define CPH(k) { long long n_ = clock64(); cph[k] += n_ - cph_t; cph_t = n_; }
#define CPH_PRINT if (b == b0 && tid == 0) printf("[phases cl part %d] pd %d cg %d | per PD iter: windows %lld self %lld rhs-exch %lld update+exch %lld | per CG iter: spmv %lld exch-pAp %lld upd+publish %lld exch-rr+halo %lld p-update %lld cycles\n", part, iters, cg_total, cph[0] / iters, cph[1] / iters, cph[2] / iters, cph[7] / iters, cph[3] / max(cg_total, 1), cph[4] / max(cg_total, 1), cph[5] / max(cg_total, 1), cph[6] / max(cg_total, 1), 0ll);
#else
#define CPH_DECL
#define CPH(k)
#define CPH_PRINT
#endif

// Layered self friction (Simulation.cpp:655-678) evaluated REDUNDANTLY by every part of a split rollout (round 6): each part stages the working set
// of the contacts (f, r of its ~2 x nself vertices through the sc1 path, the contact lists of part 0's detection through Sc1Table), walks the
// layers in its own LDS — identical inputs, identical arithmetic, identical results in every part — and then writes r and re-forms the right-hand
// side for ITS OWN rows of the working set only. Before: part 0 alone ran the layers between two cross-part barriers and every part re-formed the
// right-hand side of all its rows from global memory afterwards; the second barrier and that pass are gone. Returns false (nothing done) when the
// working set does not fit the LDS offered — a function of values every part reads identically, so all parts take the same branch.
// Only called when Xch::same_xcd holds (dc_cluster.h: Sc1Table). Same LDS layout and contact code as self_friction_layers_lds_v (dc_devlib.h).
template <int THREADS, class FV, class RV, class RHS>
__device__ __forceinline__ bool self_friction_layers_parts(const DevSystem &S, const SelfRec &R, int b, const FV &f, const RV &r, float *lds, int lds_floats,
                                                           int r0, int r1, bool write_d, RHS rhs_of) {
  const int cap = S.self_cap, N = S.N, tid = threadIdx.x;
  const Sc1Table meta = sc1_table(R.meta + (size_t) b * kMetaStride, sizeof(int) * (size_t) kMetaStride);
  const Sc1Table nrm = sc1_table(R.nrm + (size_t) b * cap, sizeof(float4) * (size_t) cap);
  const Sc1Table verts = sc1_table(R.verts + (size_t) b * 2 * cap, sizeof(int) * 2 * (size_t) cap);
  const int C = min(meta.ldi(0), cap), nl = meta.ldi(1), M = meta.ldi(kMetaStride - 1);
  if (!S.self_lds || self_lds_need(M, C, nl) > lds_floats || nl + 2 >= kMetaStride - 4) return false;
  float4 *dvec = R.dvec + (size_t) b * cap;
  float *lf = lds, *lr = lds + 3 * M, *lim = lds + 6 * M;
  float4 *ln = (float4 *) (lds + 7 * M + ((4 - (7 * M) % 4) % 4));       // 16-byte aligned
  float4 *ld = ln + C;
  int *loff = (int *) (ld + C);
  for (int s = tid; s < M; s += THREADS) {
    const int v = verts.ldi(s);
    lf[s] = f.ld(v); lf[M + s] = f.ld(N + v); lf[2 * M + s] = f.ld(2 * N + v);
    lr[s] = r.ld(v); lr[M + s] = r.ld(N + v); lr[2 * M + s] = r.ld(2 * N + v);
    lim[s] = 1.0f / S.mass[v];
  }
  for (int k = tid; k < C; k += THREADS) ln[k] = nrm.ld4(k);
  for (int l = tid; l <= nl; l += THREADS) loff[l] = meta.ldi(2 + l);
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
    for (int l = 0; l < nl; l++) {
      const int k1 = loff[l + 1];
      for (int k = loff[l] + tid; k < k1; k += THREADS) contact(k);
      __syncthreads();
    }
  } else {
    if (tid < 64) {
      for (int l = 0; l < nl; l++) {
        const int k1 = loff[l + 1];
        for (int k = loff[l] + tid; k < k1; k += 64) contact(k);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");             // compiler: keep the layers' LDS accesses in order
      }
    }
    __syncthreads();
  }
  for (int s = tid; s < M; s += THREADS) {
    const int v = verts.ldi(s);
    if (v >= r0 && v < r1) {
      const f3 rv = mk(lr[s], lr[M + s], lr[2 * M + s]);
      r.st(v, rv.x); r.st(N + v, rv.y); r.st(2 * N + v, rv.z);
      rhs_of(v, mk(lf[s], lf[M + s], lf[2 * M + s]), rv);
    }
  }
  if (write_d) for (int k = tid; k < C; k += THREADS) dvec[k] = ld[k];
  __syncthreads();
  return true;
}

// PIPE: the inner solve is the single-exchange CG (one exchange per iteration instead of two), see the loop
template <int THREADS, int VPT, bool DETECT, bool PIPE, bool DEFL>
__global__ __launch_bounds__(THREADS) void k_pd_step_cl(const DevSystem *__restrict__ Sp, const DevCluster *__restrict__ Cp, DevWork W,
                                                        FwdArgs A, int b0, int nb_real, int tail_off, int fric_floats) {
  const DevSystem &S = *Sp;
  const DevCluster &CL = *Cp;
  constexpr int WAVES = THREADS / 64;
  constexpr int HPT = (1024 + THREADS - 1) / THREADS;      // halo rows per thread: 2 HB <= 1024
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = CL.K, R = CL.R, HB = CL.HB, GL = R + 2 * HB;
  int lb, part;
  cluster_map(K, lb, part);
  if (lb >= nb_real) return;       // padding workgroups: the launch is rounded up to a multiple of 8 rollouts (see the launcher)
  if (CL.test_drop && lb == 0 && part == K - 1) return;      // test hook: a part that never arrives (tests/test_gpu_cluster.py)
  const int b = b0 + lb;
  const bool redundant_layers = CL.redundant_self != 0;
  Xch X = xch_init(CL, lb, part, lds + tail_off);
  if (!xch_hello<THREADS>(X)) return;
  float2 *gxy = (float2 *) lds;            // search direction over rows [r0 - HB, r0 + R + HB): (x, y) plane, then the z plane
  float *gz = lds + 2 * GL;
  // PIPE instances keep the direction as four halves per row (H16, dc_pklib.h: one ds_read_b64 and three v_fma_mix per non-zero, as the
  // 10 000-vertex one-workgroup kernel does) scaled by a power of two per iteration: 2 GL floats; behind it the neighbours' residual rows
  // [2 HB] in fp32 (float2 plane, then the z plane)
  constexpr bool H16 = PIPE;
  h4 *lh = (h4 *) lds;
  const unsigned lh_addr = lds_byte_address(lds);
  float2 *gxy1 = (float2 *) (lds + (H16 ? 2 : 3) * GL);
  float *gz1 = lds + (H16 ? 2 : 3) * GL + 4 * HB;
  const int N = S.N;
  const int r0 = part * R, r1 = min(N, r0 + R);
  const int nch = R >> 6, cbase = r0 >> 6;
  const size_t off = (size_t) b * 3 * N;
  float *g = W.g + off, *vnow = W.vnow + off, *vbest = W.vbest + off, *scr = W.cg_r + off;
  const BufVec vnb = buf_vec(vnow, N, X.same_xcd);
  const BufVec dpb = buf_vec(W.cg_x + off, N, X.same_xcd);    // scaled correction of the previous PD iteration (seed of the next solve)
  const int w0 = part * CL.wpp, w1 = min(CL.nwin, w0 + CL.wpp);

  for (int step = 0; step < A.nsteps; step++) {
  // the previous step's state (written by every part with plain stores) changes hands
  X.site = 1;
  if (step > 0) {   // (with the inlined detection part 0 then reads the whole state through plain loads: acquire)
    if constexpr (DETECT) { if (!xch_fence_barrier<THREADS>(X)) return; }
    else { if (!xch_barrier<THREADS>(X)) return; }
  }
  const size_t so = (size_t) step * A.slot_state;
  // this step's fixed-point targets and external forces (constant over the launch, or one set per step: dc_set_*_schedule)
  const float *xfix = A.x_fixed + (size_t) step * A.slot_xfix + (size_t) b * 3 * S.Af;
  const float *fu_s = A.fu ? A.fu + (size_t) step * A.slot_fu : nullptr;
  const float *fvs_s = A.fv_scale ? A.fv_scale + (size_t) step * A.slot_fvs : nullptr;
  // the tape state and f / r are read across parts: write-through stores, L1-bypassing loads (xnb, vinb, rfb, rrb, xob, vob)
  const BufVec xnb = buf_vec(A.x_in + off + so, N, X.same_xcd), vinb = buf_vec(A.v_in + off + so, N, X.same_xcd);
  const BufVec rfb = buf_vec(A.rec_f + off + so, N, X.same_xcd), rrb = buf_vec(A.rec_r + off + so, N, X.same_xcd);
  float *rec_n = A.rec_n + off + so;
  int *rec_prim = A.rec_prim + (size_t) b * N + (size_t) step * A.slot_prim;
  SelfRec srec = A.self;
  srec.pair += (size_t) step * A.slot_self; srec.nrm += (size_t) step * A.slot_self; srec.dvec += (size_t) step * A.slot_self;
  srec.meta += (size_t) step * A.slot_meta; srec.verts += (size_t) step * 2 * A.slot_self;
  if constexpr (DETECT) {                     // detection + layering of this step: part 0, then the lists change hands
    X.site = 2;
    if (part == 0) {
      self_detect_rollout<THREADS>(S, W, b, A.x_in + so, A.v_in + so, A.rec_prim + (size_t) step * A.slot_prim, srec, fu_s, A.fv, fvs_s, (int *) lds, A.fv2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the lists have left the CU before the exchange below tells the other parts they exist
      __syncthreads();
    }
  }
  // number of self contacts of this step: part 0 knows it (its own detection, or the stand-alone detection kernel's record) and
  // hands it to the others inside an exchange, so that no part ever has to read it from memory another part wrote
  int nself = 0;
  {
    double ns[3];
    float mine = 0.f;
    if (part == 0 && tid == 0 && S.contact_enabled && S.self_enabled) mine = (float) srec.meta[(size_t) b * kMetaStride];
    if (!xch_allsum<THREADS>(X, mine, 0.f, 0.f, ns)) return;     // (the others wait here for part 0's detection)
    nself = (int) (ns[0] + 0.5);
  }
  const float *mu = A.mu + (size_t) b * S.ngroups;
  const float h = S.h;
  const f3 grav = mk(S.gx, S.gy, S.gz);
  const f3 fu = fu_s ? mk(fu_s[3 * b], fu_s[3 * b + 1], fu_s[3 * b + 2]) : mk(0, 0, 0);
  const float fvs = fvs_s ? fvs_s[b] : 1.f;

  // ---- step set-up on the own rows: s_n, initial guess, contact detection (Simulation.cpp:1097-1160, :1254-1256) ----
  float part_s = 0.f;
  int ncontact = 0;
  for (int i = r0 + tid; i < r1; i += THREADS) {
    const float m = S.mass[i];
    f3 v = ld3c(vinb, i);
    f3 fext = grav * m + fu;                      // fillForces (Simulation.cpp:55-116)
    if (A.fv) fext = fext + ld3(A.fv + off, i, N) * fvs;
    if (A.fv2) fext = fext + ld3(A.fv2 + off, i, N);
    f3 v0 = v + fext * (h / m);                   // (s_n - x_n) / h
    st3c(vnb, i, v0);
    st3(g, i, N, v0 * m);                         // M (s_n - x_n) / h
    part_s += dot(v0, v0);
    int prim = -1;
    f3 nrm = mk(0, 0, 0);
    if (S.contact_enabled) prim = detect_primitive(S, ld3c(xnb, i), v0, nrm);
    rec_prim[i] = prim;
    st3(rec_n, i, N, nrm);
    ncontact += (prim >= 0);
  }
  xch_drain();                                    // v is read by the neighbours' windows: its stores must have left the CU
  double sums[3];
  X.site = 3;
  if (!xch_allsum<THREADS>(X, part_s, (float) ncontact, 0.f, sums)) return;
  double min_xdiff = (double) h * sqrt(sums[0]) / (double) N;
  const int total_contacts = (int) (sums[1] + 0.5);
  bool improved = false, converged = false, stalled = false, best_is_current = false;
  int iters = 0, cg_total = 0, since_progress = 0;
  double xdiff = 0;
  float dnorm = 0.f;                        // H16: |d_prev|_2 over the rollout, the scaled correction of the previous PD iteration (scale of the recycled direction)

  CPH_DECL
  for (int iter = 0; iter < A.pd_cap; iter++) {
    int zp;                                       // opaque zero against LICM of the unrolled row indices (dc_forward_pk.hip)
    asm volatile("s_mov_b32 %0, 0" : "=s"(zp));
    const int tq = tid + zp;
    auto vertex_body = [&](int i, f3 fint) -> f3 {
      f3 f = ld3(g, i, N) + fint;
      f3 v = ld3c(vnb, i);
      const int a = S.att_of_vertex[i];
      if (a >= 0) f = f + ((ld3(xfix, a, S.Af) - ld3c(xnb, i)) - v * h) * (h * S.k_att);   // AttachmentSpring.cpp:25-29
      const float m = S.mass[i];
      f3 r = mk(0, 0, 0);
      const int prim = rec_prim[i];
      if (prim >= 0) {  // calculateDryFrictionVector, primitive part (Simulation.cpp:640-652)
        f3 n = ld3(rec_n, i, N);
        f3 d = f - prim_vout(S.prims[prim], n) * m;
        r = dry_friction(n, d, mu[S.prims[prim].group]);
      }
      st3c(rfb, i, f);
      st3c(rrb, i, r);
      return (f + r - v * m) * CL.sq_dinv[i];       // scaled residual D^-1/2 rhs
    };
    // ---- local step + vertex pass through this part's element windows ----
    float psum = 0.f;
    auto vert = vert_noa([&](int i, f3 sum, f3) {      // (does not use input 1 at the vertex: dc_winlib.h, NOA)
      f3 rhs = vertex_body(i, sum);
      st3(scr, i, N, rhs);
    });
    element_windows_t<THREADS, kFwdOpsPrecise>(CL, w0, w1, lds, In2Sc1{xnb}, In2Sc1{vnb}, fwd_tri_op(h, S.h64), fwd_bend_op(h, S.h64), vert);   // fp64-strain operators (dc_winlib.h)
    __syncthreads();
    CPH(0)
    X.site = 4;
    if (nself > 0) {   // layered self friction (Simulation.cpp:655-678) over the rollout's f / r, then the right-hand side of the vertices it touched
      if (!xch_barrier<THREADS>(X)) return;
      bool redundant = false;
      if (X.same_xcd && redundant_layers)      // every part for itself (self_friction_layers_parts above): no second barrier, no pass over all rows
        redundant = self_friction_layers_parts<THREADS>(S, srec, b, rfb, rrb, lds, fric_floats, r0, r1, part == 0, [&](int i, f3 fv, f3 rv) {
          st3(scr, i, N, (fv + rv - ld3c(vnb, i) * S.mass[i]) * CL.sq_dinv[i]);
        });
      if (!redundant) {
        if (part == 0) {
          if (!self_friction_layers_lds_v<THREADS>(S, srec, b, rfb, rrb, lds, fric_floats)) self_friction_layers_v<THREADS>(S, srec, b, rfb, rrb);
        }
        if (!xch_barrier<THREADS>(X)) return;
        for (int i = r0 + tid; i < r1; i += THREADS) {
          f3 rhs = (ld3c(rfb, i) + ld3c(rrb, i) - ld3c(vnb, i) * S.mass[i]) * CL.sq_dinv[i];
          st3(scr, i, N, rhs);
        }
        __syncthreads();
      }
    }
    CPH(1)
    // ---- residual into registers, search direction p0 = r0 into the gather array, boundary rows to the neighbours ----
    float rr[VPT][3], ap[VPT][3], xx[VPT][3];
    X.site = 5;
    xch_begin(X);
    psum = 0.f;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int l = tq + k * THREADS, i = r0 + l;
      const bool on = l < R && i < N;
      const int ic = on ? i : r0;
      const float okf = on ? 1.f : 0.f;
      rr[k][0] = scr[ic] * okf; rr[k][1] = scr[N + ic] * okf; rr[k][2] = scr[2 * N + ic] * okf;
#pragma unroll
      for (int c = 0; c < 3; c++) { xx[k][c] = 0.f; psum = fmaf(rr[k][c], rr[k][c], psum); }      // |rhs|^2 of the own rows (after the self-friction pass, if any)
      if (l < R) {
        if constexpr (!H16) { gxy[HB + l] = make_float2(rr[k][0], rr[k][1]); gz[HB + l] = rr[k][2]; }
        xch_publish_boundary(X, l, R, rr[k][0], rr[k][1], rr[k][2]);
      }
    }
    double rz;
    float hs = 1.f, pn = 0.f;               // H16: scale of the direction in LDS (a power of two) and the bound on its entries it comes from
    {
      xch_publish_sums(X, psum, 0.f, 0.f);
      f3 hv[HPT];
      if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
      rz = sums[0];
      if constexpr (H16) {                  // first direction d = r (|r|_inf <= |r|_2 = sqrt(rz)), rounded to halves; the neighbours' rows of r kept in fp32
        pn = sqrtf((float) rz); hs = half_scale(pn);
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int l = tq + k * THREADS;
          if (l < R) lh[HB + l] = pack_h4(rr[k][0] * hs, rr[k][1] * hs, rr[k][2] * hs);
        }
#pragma unroll
        for (int q = 0; q < HPT; q++) {
          const int j = tid + q * THREADS;
          if (j < 2 * HB) {
            const int li = j < HB ? j : R + j;
            lh[li] = pack_h4(hv[q].x * hs, hv[q].y * hs, hv[q].z * hs);
            gxy1[j] = make_float2(hv[q].x, hv[q].y); gz1[j] = hv[q].z;
          }
        }
      } else {
#pragma unroll
      for (int q = 0; q < HPT; q++) {
        const int j = tid + q * THREADS;
        if (j < 2 * HB) { const int li = j < HB ? j : R + j; gxy[li] = make_float2(hv[q].x, hv[q].y); gz[li] = hv[q].z; }
      }
      }
    }
    // With few rows per thread the first packet batch of every row (16 registers per row) stays in registers for the whole solve:
    // the matrix is the same in all ~25 iterations, and a part that is only a few rows deep cannot hide the L2 latency of
    // re-reading it behind its own arithmetic (measured r02w: 4.9 k cycles per product of 3 rows, 1.6 x the per-row cost of the
    // 20-row kernel).
    constexpr bool MATREG = VPT <= 4;
    int4 mat[MATREG ? VPT : 1][PB];
    if constexpr (MATREG) {
#pragma unroll
      for (int k = 0; k < VPT; k++) load_batch(mat[k], CL.pk + CL.pk_ptr[cbase + min(wv + k * WAVES, nch - 1)] + lane, 0);
    }
    // ap = Ahat p on the own rows (p incl. halo in the gather array), part2 += <p, ap>
    auto spmv = [&](int wz, const float2 *vxy, const float *vz, float &part2, bool with_pr, float &part3) {
      int4 nxt[PB];
      if constexpr (!MATREG) load_batch(nxt, CL.pk + CL.pk_ptr[cbase + min(wz, nch - 1)] + lane, 0);
#pragma unroll
      for (int k = 0; k < VPT; k++) {
        const int lc = wz + k * WAVES;          // wave-uniform local chunk
        // a chunk past the part's rows (R / 64 is not a multiple of the wave count: 20 chunks on 8 waves leave waves 4 ... 7 without a third one)
        // is SKIPPED, not computed and masked: the SIMD's other wave gets the issue slots (round 6; before, every wave ran all VPT rows)
        if (lc >= nch) { ap[k][0] = 0.f; ap[k][1] = 0.f; ap[k][2] = 0.f; continue; }
        const int chunk = cbase + lc;
        const int np = CL.pk_n[chunk];
        const int4 *row = CL.pk + CL.pk_ptr[chunk] + lane;
        int4 cur[PB];
        if constexpr (MATREG) {
#pragma unroll
          for (int j = 0; j < PB; j++) cur[j] = mat[k][j];
        } else {
#pragma unroll
          for (int j = 0; j < PB; j++) cur[j] = nxt[j];
          if (k + 1 < VPT) load_batch(nxt, CL.pk + CL.pk_ptr[cbase + min(lc + WAVES, nch - 1)] + lane, 0);
        }
        const int li = HB + lc * 64 + lane;
        if constexpr (H16) {
          unsigned rowbase = lh_addr + 8u * (unsigned) (li - 512);
          asm volatile("" : "+v"(rowbase));      // opaque: one register per row (dc_forward_pk_kernel.h)
          float ax, ay, az;
          pk_v2i own;
          consume_h_row(cur, rowbase, ax, ay, az, own);      // gathers + own entry in one group, own entry enters last (dc_pklib.h)
          for (int s0 = PB; s0 < np; s0 += PB) {        // rows wider than one batch
            load_batch(cur, row, s0);
            consume_h(cur, rowbase, ax, ay, az);
          }
          ap[k][0] = ax; ap[k][1] = ay; ap[k][2] = az;
          dot3_h(ax, ay, az, own, part2);
          dot3_h(rr[k][0], rr[k][1], rr[k][2], own, part3);      // d.r along the ROUNDED direction: the exact line search needs it in every iteration
        } else {
        const float2 pxy = vxy[li];
        const float pz = vz[li];
        float ax = pxy.x, ay = pxy.y, az = pz;        // unit diagonal
        const int base = li - 512;
        consume_pf(cur, vxy, vz, base, ax, ay, az);      // all gathers of the batch in flight before the first product (dc_pklib.h)
        for (int s0 = PB; s0 < np; s0 += PB) {        // rows wider than one batch
          load_batch(cur, row, s0);
          consume_pf(cur, vxy, vz, base, ax, ay, az);
        }
        ap[k][0] = ax; ap[k][1] = ay; ap[k][2] = az;
        part2 += pxy.x * ax + pxy.y * ay + pz * az;
        if (with_pr) part3 += pxy.x * rr[k][0] + pxy.y * rr[k][1] + pz * rr[k][2];     // seeded pass only (uniform branch)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    __syncthreads();
    CPH(2)
    // ---- global step: CG on the scaled system = Jacobi PCG on P dv = rhs ----
    if constexpr (PIPE) {
      // Single-exchange CG (round 6). Standard CG — same vectors, same updates, no vector recurrences — whose TWO exchanges per iteration
      // ([p.Ap] and [r.r + boundary rows of the new residual]) are folded into ONE: the product pass also forms r.Ap, Ap.Ap and the true
      // r.r of the CURRENT residual, and they travel with the boundary rows of A p. Then
      //     alpha   = r.r / p.Ap                          (seeded pass: d.r / d.Ad — the line search along the recycled direction)
      //     |r'|^2  = r.r - 2 alpha r.Ap + alpha^2 Ap.Ap   (one step from a TRUE r.r: the rounding of the three sums, ~1e-6 relative, never accumulates)
      //     beta    = |r'|^2 / r.r
      // and every part updates, besides its own rows, its copy of the neighbours' boundary rows of r and p itself: r_halo -= alpha (A p)_halo
      // with the received rows (bitwise the neighbour's own update), p_halo = r_halo + beta p_halo. The neighbours' residual rows live in a
      // second small LDS array (rh). What the pipelined CG of round 4 lost (its recurrences for A r, A p, A s drift in fp32: 7e-5 at
      // N = 16 384) cannot happen here: the only recurred quantity is the scalar |r'|^2, re-based on the true value every iteration.
      float2 *rhxy = gxy1;                       // [2 HB] residual rows of the neighbours (same indexing as the halo rows of the direction: j < HB lower, HB + j upper)
      float *rhz = gz1;
      float *lsum6 = X.lsum + 16;
      if (rz > 1e-300) {
        const double stop = (double) A.cg_tol * (double) A.cg_tol * rz;
        // (the neighbours' rows of r0 = rhs arrived with the exchange above: they are in rh already — H16 — or copied from the direction's halo rows now)
        if constexpr (!H16) {
#pragma unroll
          for (int q = 0; q < HPT; q++) {
            const int j = tid + q * THREADS;
            if (j < 2 * HB) { const int li = j < HB ? j : R + j; rhxy[j] = gxy[li]; rhz[j] = gz[li]; }
          }
        }
        bool seed = A.cg_seed && iter > 0;
        if (seed) {
          __syncthreads();                                  // (the direction's rows are rewritten by other threads than the ones that wrote / copied them)
          if constexpr (H16) hs = half_scale(dnorm);        // |d_prev|_2 of the rollout, from the exchange that ended the previous PD iteration
          for (int j = tid; j < R + 2 * HB; j += THREADS) {
            const int i = r0 - HB + j;                      // gather index j <-> global row i
            const bool on = i >= 0 && i < N && (j < HB || j >= HB + R || i < r1);
            f3 d = mk(0, 0, 0);
            if (on) d = ld3c(dpb, i);
            if constexpr (H16) lh[j] = pack_h4(d.x * hs, d.y * hs, d.z * hs);
            else { gxy[j] = make_float2(d.x, d.y); gz[j] = d.z; }
          }
          __syncthreads();
        }
        for (int it = 0; it < A.cg_max;) {
          int zs;
          asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
          const int wz = wv + zs, tz = tid + zs;
          float s_pap = 0.f, s_pr = 0.f;
          spmv(wz, gxy, gz, s_pap, seed, s_pr);
          float s_rap = 0.f, s_apap = 0.f, s_rr = 0.f;
          X.site = 6;
          xch_begin(X);
#pragma unroll
          for (int k = 0; k < VPT; k++) {
            const int l = tz + k * THREADS;
#pragma unroll
            for (int c = 0; c < 3; c++) {
              s_rap = fmaf(rr[k][c], ap[k][c], s_rap); s_apap = fmaf(ap[k][c], ap[k][c], s_apap); s_rr = fmaf(rr[k][c], rr[k][c], s_rr);
            }
            if (l < R) xch_publish_boundary(X, l, R, ap[k][0], ap[k][1], ap[k][2]);
          }
          xch_publish_sums6(X, WAVES, s_pap, s_pr, s_rap, s_apap, s_rr, 0.f);
          CPH(3)
          double s6[6];
          f3 hv[HPT];
          if (!xch_finish6<THREADS, HPT, true>(X, lsum6, s6, hv)) return;
          CPH(4)
          const double rrt = s6[4];
          const double pr = (H16 || seed) ? s6[1] : rrt;      // H16: exact line search along the rounded direction
          const double alpha_d = s6[0] > 1e-300 ? pr / s6[0] : 0.0;
          const float alpha = (float) alpha_d;
          const double rz_new = rrt - 2.0 * alpha_d * s6[2] + alpha_d * alpha_d * s6[3];
          it++; cg_total++;
          const bool done = !(rz_new > stop);
          const float beta = (seed || done) ? 0.f : (float) (rz_new / rrt);
          seed = false;
          if constexpr (H16) {
            // d_new = r + beta d_old in true units; in LDS units: hs_new r + (beta hs_new / hs_old) d~_old, entries bounded by |r|_2 + beta * bound_old
            pn = sqrtf((float) fmax(rz_new, 0.0)) + beta * pn;
            const float hs_new = half_scale(pn), c2 = beta * hs_new / hs;
            hs = hs_new;
#pragma unroll
            for (int k = 0; k < VPT; k++) {
              const int l = tz + k * THREADS;
              const int lc = min(l, R - 1);
              const h4 q = lh[HB + lc];
              const float pv[3] = {(float) q.x, (float) q.y, (float) q.z};
#pragma unroll
              for (int c = 0; c < 3; c++) {
                xx[k][c] = fmaf(alpha, pv[c], xx[k][c]);
                rr[k][c] = fmaf(-alpha, ap[k][c], rr[k][c]);
              }
              if (l < R) lh[HB + l] = pack_h4(fmaf(c2, pv[0], rr[k][0] * hs), fmaf(c2, pv[1], rr[k][1] * hs), fmaf(c2, pv[2], rr[k][2] * hs));
            }
            if (done) break;
#pragma unroll
            for (int q = 0; q < HPT; q++) {         // the neighbours' boundary rows: their residual from the received rows of A d, then their direction
              const int j = tid + q * THREADS;
              if (j < 2 * HB) {
                const int li = j < HB ? j : R + j;
                const float2 rxy = rhxy[j];
                const float rx = fmaf(-alpha, hv[q].x, rxy.x), ry = fmaf(-alpha, hv[q].y, rxy.y), rzh = fmaf(-alpha, hv[q].z, rhz[j]);
                rhxy[j] = make_float2(rx, ry); rhz[j] = rzh;
                const h4 p4 = lh[li];
                lh[li] = pack_h4(fmaf(c2, (float) p4.x, rx * hs), fmaf(c2, (float) p4.y, ry * hs), fmaf(c2, (float) p4.z, rzh * hs));
              }
            }
          } else {
#pragma unroll
          for (int k = 0; k < VPT; k++) {
            const int l = tz + k * THREADS;
            const int lc = min(l, R - 1);
            const float2 pxy = gxy[HB + lc];
            const float pv[3] = {pxy.x, pxy.y, gz[HB + lc]};
#pragma unroll
            for (int c = 0; c < 3; c++) {
              xx[k][c] = fmaf(alpha, pv[c], xx[k][c]);
              rr[k][c] = fmaf(-alpha, ap[k][c], rr[k][c]);
            }
            if (l < R) {
              gxy[HB + l] = make_float2(fmaf(beta, pv[0], rr[k][0]), fmaf(beta, pv[1], rr[k][1]));
              gz[HB + l] = fmaf(beta, pv[2], rr[k][2]);
            }
          }
          if (done) break;
#pragma unroll
          for (int q = 0; q < HPT; q++) {         // the neighbours' boundary rows: their residual from the received rows of A p, then their direction
            const int j = tid + q * THREADS;
            if (j < 2 * HB) {
              const int li = j < HB ? j : R + j;
              const float2 rxy = rhxy[j];
              const float rx = fmaf(-alpha, hv[q].x, rxy.x), ry = fmaf(-alpha, hv[q].y, rxy.y), rzh = fmaf(-alpha, hv[q].z, rhz[j]);
              rhxy[j] = make_float2(rx, ry); rhz[j] = rzh;
              const float2 pxy = gxy[li];
              gxy[li] = make_float2(fmaf(beta, pxy.x, rx), fmaf(beta, pxy.y, ry));
              gz[li] = fmaf(beta, gz[li], rzh);
            }
          }
          }
          CPH(5)
          __syncthreads();
          CPH(6)
        }
      }
    } else
    if (rz > 1e-300) {
      const double stop = (double) A.cg_tol * (double) A.cg_tol * rz;
      // Spectral deflation (DEFL; dc_deflate.h, the one-workgroup version is in dc_forward_pk_kernel.h): Galerkin projection of the current
      // residual onto the 16 lowest eigenvectors U of the scaled matrix, x += U c, r -= (A U) c with c = (U^T A U)^-1 U^T r per coordinate.
      // U^T r is a sum over the parts: sixteen three-value exchanges per solve (the price of ~8 CG iterations, for ~200 saved); the last
      // exchange carries the new r.r and the boundary rows of the new residual, like the one that ends a CG iteration.
      [[maybe_unused]] auto deflate = [&](double &rz_out, f3 (&hvo)[HPT]) -> bool {
        constexpr int DK = 16;
        __shared__ float dfl_t[3 * DK], dfl_c[3 * DK];
        const float4 DC_G *U4 = (const float4 DC_G *) S.defl_u;
        const float4 DC_G *AU4 = (const float4 DC_G *) S.defl_au;
        for (int j4 = 0; j4 < DK / 4; j4++) {
          float acc[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
          for (int k = 0; k < VPT; k++) {
            const int l = tid + k * THREADS, i = r0 + l;
            if (l < R && i < N) {
              const float4 u = U4[(size_t) i * (DK / 4) + j4];
#pragma unroll
              for (int c = 0; c < 3; c++) {
                acc[0][c] = fmaf(u.x, rr[k][c], acc[0][c]); acc[1][c] = fmaf(u.y, rr[k][c], acc[1][c]);
                acc[2][c] = fmaf(u.z, rr[k][c], acc[2][c]); acc[3][c] = fmaf(u.w, rr[k][c], acc[3][c]);
              }
            }
          }
          for (int q = 0; q < 4; q++) {
            X.site = 8;
            if (!xch_allsum<THREADS>(X, acc[q][0], acc[q][1], acc[q][2], sums)) return false;
            if (tid < 3) dfl_t[(j4 * 4 + q) * 3 + tid] = (float) sums[tid];
          }
        }
        __syncthreads();
        if (tid < 3 * DK) {
          const int j = tid / 3, c = tid - 3 * j;
          float sacc = 0.f;
          for (int l = 0; l < DK; l++) sacc = fmaf(S.defl_g[j * DK + l], dfl_t[l * 3 + c], sacc);
          dfl_c[tid] = sacc;
        }
        __syncthreads();
        float part = 0.f;
        X.site = 9;
        xch_begin(X);
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int l = tid + k * THREADS, i = r0 + l;
          if (l < R && i < N) {
            for (int j4 = 0; j4 < DK / 4; j4++) {
              const float4 u = U4[(size_t) i * (DK / 4) + j4], au = AU4[(size_t) i * (DK / 4) + j4];
              const float *cj = dfl_c + j4 * 12;
#pragma unroll
              for (int c = 0; c < 3; c++) {
                xx[k][c] += u.x * cj[c] + u.y * cj[3 + c] + u.z * cj[6 + c] + u.w * cj[9 + c];
                rr[k][c] -= au.x * cj[c] + au.y * cj[3 + c] + au.z * cj[6 + c] + au.w * cj[9 + c];
              }
            }
          }
          part = fmaf(rr[k][0], rr[k][0], part); part = fmaf(rr[k][1], rr[k][1], part); part = fmaf(rr[k][2], rr[k][2], part);
          if (l < R) xch_publish_boundary(X, l, R, rr[k][0], rr[k][1], rr[k][2]);
        }
        xch_publish_sums(X, part, 0.f, 0.f);
        if (!xch_finish<THREADS, HPT, true>(X, sums, hvo)) return false;
        rz_out = sums[0];
        return true;
      };
      // the new search direction after a projection: the residual (own rows from registers, the neighbours' boundary rows as received)
      [[maybe_unused]] auto direction_from_residual = [&](const f3 (&hvi)[HPT]) {
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int l = tid + k * THREADS;
          if (l < R) { gxy[HB + l] = make_float2(rr[k][0], rr[k][1]); gz[HB + l] = rr[k][2]; }
        }
#pragma unroll
        for (int q = 0; q < HPT; q++) {
          const int j = tid + q * THREADS;
          if (j < 2 * HB) { const int li = j < HB ? j : R + j; gxy[li] = make_float2(hvi[q].x, hvi[q].y); gz[li] = hvi[q].z; }
        }
        __syncthreads();
      };
      if constexpr (DEFL) {
        if (!(A.cg_seed && iter > 0)) {        // no recycled direction in this solve: project first
          f3 hv0[HPT];
          if (!deflate(rz, hv0)) return;
          direction_from_residual(hv0);
        }
      }
      // Recycled first direction (A.cg_seed, see dc_forward_pk.hip): the previous PD iteration's correction d, read back with its
      // boundary rows from the array every part stored its rows to before the exchange of the update; x = <d, r> / <d, A d> d,
      // then CG restarted from the new residual. Saves exchanges as well as products.
      bool seed = A.cg_seed && iter > 0;
      if (seed) {
        for (int j = tid; j < R + 2 * HB; j += THREADS) {
          const int i = r0 - HB + j;                      // gather index j <-> global row i
          const bool on = i >= 0 && i < N && (j < HB || j >= HB + R || i < r1);
          f3 d = mk(0, 0, 0);
          if (on) d = ld3c(dpb, i);
          gxy[j] = make_float2(d.x, d.y); gz[j] = d.z;
        }
        __syncthreads();
      }
      for (int it = 0; it < A.cg_max;) {
        float part2 = 0.f, part3 = 0.f;
        int zs;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
        const int wz = wv + zs, tz = tid + zs;
        spmv(wz, gxy, gz, part2, seed, part3);
        CPH(3)
        X.site = 6;
        if (!xch_allsum<THREADS>(X, part2, part3, 0.f, sums)) return;
        CPH(4)
        const double pr = seed ? sums[1] : rz;
        const float alpha = sums[0] > 1e-300 ? (float) (pr / sums[0]) : 0.f;
        part2 = 0.f;
        X.site = 7;
        xch_begin(X);
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int l = tz + k * THREADS;
          const int lc = min(l, R - 1);
          const float2 pxy = gxy[HB + lc];
          const float pv[3] = {pxy.x, pxy.y, gz[HB + lc]};
#pragma unroll
          for (int c = 0; c < 3; c++) {
            xx[k][c] = fmaf(alpha, pv[c], xx[k][c]);
            rr[k][c] = fmaf(-alpha, ap[k][c], rr[k][c]);
            part2 = fmaf(rr[k][c], rr[k][c], part2);
          }
          if (l < R) xch_publish_boundary(X, l, R, rr[k][0], rr[k][1], rr[k][2]);
        }
        xch_publish_sums(X, part2, 0.f, 0.f);
        CPH(5)
        f3 hv[HPT];
        if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
        CPH(6)
        double rz_new = sums[0];
        it++; cg_total++;
        if (!(rz_new > stop)) break;
        if constexpr (DEFL) {
          if (seed) {                 // after the recycled direction: project the residual, then plain CG from it (beta = 0)
            if (!deflate(rz_new, hv)) return;
            if (!(rz_new > stop)) break;
          }
        }
        const float beta = seed ? 0.f : (float) (rz_new / rz);
        seed = false;
        rz = rz_new;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int l = tz + k * THREADS;
          if (l < R) {
            const float2 pxy = gxy[HB + l];
            gxy[HB + l] = make_float2(fmaf(beta, pxy.x, rr[k][0]), fmaf(beta, pxy.y, rr[k][1]));
            gz[HB + l] = fmaf(beta, gz[HB + l], rr[k][2]);
          }
        }
#pragma unroll
        for (int q = 0; q < HPT; q++) {         // the neighbours' boundary rows of p, updated here from their residual rows
          const int j = tid + q * THREADS;
          if (j < 2 * HB) {
            const int li = j < HB ? j : R + j;
            const float2 pxy = gxy[li];
            gxy[li] = make_float2(fmaf(beta, pxy.x, hv[q].x), fmaf(beta, pxy.y, hv[q].y));
            gz[li] = fmaf(beta, gz[li], hv[q].z);
          }
        }
        __syncthreads();
      }
    }
    // ---- update + convergence (Simulation.cpp:1268, 1310-1373); delta v replaces A p in its registers ----
    psum = 0.f;
    float partd = 0.f;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int l = tq + k * THREADS, i = r0 + l;
      const bool on = l < R && i < N;
      const int ic = on ? i : r0;
      const float sq = CL.sq_dinv[ic];
      const f3 vq = ld3c(vnb, ic);
      ap[k][0] = xx[k][0] * sq; ap[k][1] = xx[k][1] * sq; ap[k][2] = xx[k][2] * sq;
      if (on) {
        if (A.cg_seed) st3c(dpb, i, mk(xx[k][0], xx[k][1], xx[k][2]));
        st3c(vnb, i, mk(vq.x + ap[k][0], vq.y + ap[k][1], vq.z + ap[k][2]));
        psum = fmaf(ap[k][0], ap[k][0], psum); psum = fmaf(ap[k][1], ap[k][1], psum); psum = fmaf(ap[k][2], ap[k][2], psum);
        partd = fmaf(xx[k][0], xx[k][0], partd); partd = fmaf(xx[k][1], xx[k][1], partd); partd = fmaf(xx[k][2], xx[k][2], partd);
      }
    }
    X.site = 8;
    xch_drain();                                  // the new v must have left the CU before the norm (= its hand-over flag) is published
    if (!xch_allsum<THREADS>(X, psum, partd, 0.f, sums)) return;
    dnorm = sqrtf((float) sums[1]);
    xdiff = (double) h * sqrt(sums[0]) / (double) N;
    CPH(7)
    iters = iter + 1;
    converged = xdiff < (double) A.fwd_tol;
    if (xdiff < min_xdiff) {
      since_progress = 0;
      min_xdiff = xdiff;
      improved = true;
      best_is_current = true;
    } else if (best_is_current) {
      // first non-improving iteration after a minimum: the best iterate is the previous one = v - delta (delta is still in registers)
      best_is_current = false;
#pragma unroll
      for (int k = 0; k < VPT; k++) {
        const int l = tq + k * THREADS, i = r0 + l;
        if (l < R && i < N) {
          const f3 v = ld3c(vnb, i);
          vbest[i] = v.x - ap[k][0]; vbest[N + i] = v.y - ap[k][1]; vbest[2 * N + i] = v.z - ap[k][2];
        }
      }
    }
    if (converged) break;
    if (++since_progress >= A.stall_window) { stalled = true; break; }
  }
  // ---- write the new state of the own rows (revert to the best iterate when the cap was hit, Simulation.cpp:1357-1367) ----
  const BufVec xob = buf_vec(A.x_out + off + so, N, X.same_xcd), vob = buf_vec(A.v_out + off + so, N, X.same_xcd);
  for (int i = r0 + tid; i < r1; i += THREADS) {
    f3 x = ld3c(xnb, i);
    if (converged) { f3 v = ld3c(vnb, i); st3c(vob, i, v); st3c(xob, i, x + v * h); }
    else if (improved) { f3 v = best_is_current ? ld3c(vnb, i) : ld3(vbest, i, N); st3c(vob, i, v); st3c(xob, i, x + v * h); }
    else { st3c(vob, i, ld3c(vinb, i)); st3c(xob, i, x); }
  }
  if (tid == 0 && part == 0) {
    dc_step_stats s;
    s.converged = converged ? 1 : (stalled ? 2 : 0); s.pd_iters = iters; s.cg_iters = cg_total; s.prim_contacts = total_contacts;
    s.self_contacts = nself; s.last_xdiff = (float) xdiff;
    s.self_overflow = (S.contact_enabled && S.self_enabled) ? srec.meta[(size_t) b * kMetaStride + kMetaStride - 2] : 0;
    A.stats[b + (size_t) step * A.slot_stats] = s;
  }
  CPH_PRINT
  }   // step
}

template <int VPT, bool DETECT, bool PIPE, bool DEFL = false>
static hipError_t launch_cl_inst(const DevSystem &S, const DevCluster &CL, const DevWork &W, const FwdArgs &A, int b0, int nb, hipStream_t st) {
  constexpr int THREADS = 512;
  const int GL = CL.R + 2 * CL.HB;
  int floats = std::max((PIPE ? 2 : 3) * GL + (PIPE ? 6 * CL.HB : 0), CL.win_lds_bytes / 4);      // (PIPE: direction as 8-byte rows + the neighbours' residual rows)
  const int fric_floats = floats;      // LDS offered to the layered friction pass: the same with and without the inlined detection
  if (DETECT) floats = std::max(floats, kSelfDetectLdsInts);
  const int tail_off = (floats + 3) / 4 * 4;
  const size_t lds = sizeof(float) * (size_t) (tail_off + kXchLdsFloats);
  if (lds > 160 * 1024 - 256 - (DEFL ? 512 : 0)) return hipErrorInvalidValue;      // (DEFL: 384 bytes of static LDS)
  hipError_t e = hipFuncSetAttribute((const void *) k_pd_step_cl<THREADS, VPT, DETECT, PIPE, DEFL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_pd_step_cl<THREADS, VPT, DETECT, PIPE, DEFL>), dim3((nb + 7) / 8 * 8 * CL.K), dim3(THREADS), lds, st, S.self_dev, CL.self_dev, W, A, b0, nb, tail_off, fric_floats);
  return hipGetLastError();
}

}  // namespace dc
