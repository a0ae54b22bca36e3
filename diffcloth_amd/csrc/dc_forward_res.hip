// CDNA4 (gfx950) forward step, LDS/register-resident variant (N <= 12288 vertices).
//
// Same algorithm and same per-rollout workgroup ownership as dc_forward.hip, but the inner block-Jacobi PCG —
// >95 % of the step's sweeps — never leaves the CU:
//   * the search direction p lives in LDS (3 planes x THREADS*VPT floats, up to 144 KB of the 160 KB),
//   * the iterate x, the residual r and A p live in registers (each thread owns VPT vertices, strided),
//   * the shared matrix P is streamed from L2 in wave-sliced ELL (64 rows x width, 8 B per non-zero, coalesced),
//     neighbours' p are gathered from LDS,
// so a CG iteration touches no HBM at all. Reference: Simulation::step (Simulation.cpp:1043-1428), global solve
// :1267 (SimplicialLLT::solve) replaced by this PCG on the correction system (see dc_forward.hip header).
#define DC_KERNEL_TU
#include "dc_devlib.h"
#include "dc_winlib.h"

namespace dc {

typedef float v4f __attribute__((ext_vector_type(4)));

// Optional phase timing (build with -DDC_PROFILE_PHASES): thread 0 of workgroup 0 prints shader-clock totals.
#ifdef DC_PROFILE_PHASES
#define PH_DECL long long ph_t = clock64(); long long ph_acc[6] = {0, 0, 0, 0, 0, 0};
#define PH(k) { long long n_ = clock64(); ph_acc[k] += n_ - ph_t; ph_t = n_; }
#define PH_PRINT if (blockIdx.x == 0 && threadIdx.x == 0) printf("[phases] pd %d cg %d | per PD iter: local %lld vertex %lld pd-update %lld | per CG iter: spmv %lld pAp-red %lld upd+red %lld cycles\n", iters, cg_total, ph_acc[0] / iters, ph_acc[1] / iters, ph_acc[5] / iters, ph_acc[2] / cg_total, ph_acc[3] / cg_total, ph_acc[4] / cg_total);
#else
#define PH_DECL
#define PH(k)
#define PH_PRINT
#endif

template <int THREADS, int VPT>
__global__ __launch_bounds__(THREADS) void k_pd_step_res(const DevSystem *__restrict__ Sp, DevWork W, FwdArgs A) {
  const DevSystem &S = *Sp;
  constexpr int NP = THREADS * VPT;
  extern __shared__ float lp[];          // [3][NP] search direction
  __shared__ double red[THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int N = S.N, T = S.T, E = S.E, NC = S.NC;
  const size_t off = (size_t) b * 3 * N;
  const float *xn = A.x_in + off, *vn = A.v_in + off;
  float *g = W.g + off, *vnow = W.vnow + off, *vbest = W.vbest + off;
  float *corner = W.corner + (size_t) b * 3 * NC;
  float4 *ap4 = W.ap4 + (size_t) b * N;      // per-vertex float4 scratch: one 16-byte coalesced access per vertex
  float *rec_f = A.rec_f + off, *rec_r = A.rec_r + off, *rec_n = A.rec_n + off;
  int *rec_prim = A.rec_prim + (size_t) b * N;
  const float *xfix = A.x_fixed + (size_t) b * 3 * S.Af;
  const float *mu = A.mu + (size_t) b * S.ngroups;
  const float h = S.h;
  const f3 grav = mk(S.gx, S.gy, S.gz);
  const f3 fu = A.fu ? mk(A.fu[3 * b], A.fu[3 * b + 1], A.fu[3 * b + 2]) : mk(0, 0, 0);
  const int nchunks = (N + 63) >> 6;

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

  PH_DECL
  for (int iter = 0; iter < A.pd_cap; iter++) {
    // ---- local step: per-element projection residual, written per constraint corner ----
    for (int t = tid; t < T; t += THREADS) {      // Triangle::project (Triangle.cpp:310-351)
      const int i0 = S.tri_v[t], i1 = S.tri_v[T + t], i2 = S.tri_v[2 * T + t];
      const float4 D = S.tri_D[t];
      // fp64-strain element operator (dc_winlib.h: the fp32 evaluation of T - F carries the 6e-8 roundings of F into a difference of
      // nearly equal quantities)
      f3 g0, g1;
      HybridTriOp{S.h64}(ld3(xn, i0, N), ld3(xn, i1, N), ld3(xn, i2, N), ld3(vnow, i0, N), ld3(vnow, i1, N), ld3(vnow, i2, N), D, S.tri_Dlo[t], S.tri_w2[t], g0, g1);
      f3 c1 = g0 * D.x + g1 * D.y, c2 = g0 * D.z + g1 * D.w;
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
    // ---- vertex pass: f, friction r, right-hand side of the correction solve -> registers / LDS ----
    part = 0.f;
    for (int i = tid; i < NP; i += THREADS) {
      f3 rhs = mk(0, 0, 0);
      float di = 0.f;
      if (i < N) {
        f3 f = ld3(g, i, N);
        const int k1 = S.inc_ptr[i + 1];
        for (int q = S.inc_ptr[i]; q < k1; q++) f = f + ld3(corner, S.inc_idx[q], NC);
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
        rhs = f + r - v * m;
        di = S.dinv[i];
        part += dot(rhs, rhs) * di;
        ap4[i] = make_float4(rhs.x, rhs.y, rhs.z, 0.f);
      }
      lp[i] = rhs.x * di; lp[NP + i] = rhs.y * di; lp[2 * NP + i] = rhs.z * di;
    }
    if (nself > 0) {   // self contacts: layered Gauss-Seidel on r (Simulation.cpp:655-678), then rebuild the right-hand side
      __syncthreads();
      self_friction_layers<THREADS>(S, A.self, b, rec_f, rec_r);
      part = 0.f;
      for (int i = tid; i < N; i += THREADS) {
        f3 rhs = ld3(rec_f, i, N) + ld3(rec_r, i, N) - ld3(vnow, i, N) * S.mass[i];
        const float di = S.dinv[i];
        ap4[i] = make_float4(rhs.x, rhs.y, rhs.z, 0.f);
        lp[i] = rhs.x * di; lp[NP + i] = rhs.y * di; lp[2 * NP + i] = rhs.z * di;
        part += dot(rhs, rhs) * di;
      }
    }
    double rz = block_sum<THREADS>((double) part, red);
    // residual and iterate of the PCG live in registers from here to the update (A p goes through a coalesced
    // per-thread scratch in global memory: keeping it in registers too would spill at VPT >= 8)
    float rr[VPT][3], xx[VPT][3];
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int i = tid + k * THREADS;
      const float4 q = (i < N) ? ap4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      rr[k][0] = q.x; rr[k][1] = q.y; rr[k][2] = q.z;
      xx[k][0] = xx[k][1] = xx[k][2] = 0.f;
    }
    PH(1)
    // ---- global step: block-Jacobi PCG on P dv = rhs, resident in LDS + registers ----
    if (rz > 1e-300) {
      const double stop = (double) A.cg_tol * (double) A.cg_tol * rz;
      for (int it = 0; it < A.cg_max;) {
        __syncthreads();
        part = 0.f;
#pragma unroll 1
        for (int k = 0; k < VPT; k++) {     // SpMV rows of this thread; touches no register array -> not unrolled
          const int i = tid + k * THREADS;
          const int chunk = i >> 6;
          float ax = 0.f, ay = 0.f, az = 0.f;
          if (chunk < nchunks) {
            const int2 *row = S.ell + S.ell_ptr[chunk] + lane;
            const int w = S.ell_w[chunk];
            // batches of 8 non-zeros: 8 independent 8-byte loads in flight, then 24 independent LDS gathers
            // (a plain s-loop serialises L2 latency -> LDS latency -> FMA per non-zero)
            for (int s0 = 0; s0 < w; s0 += 8) {
              int2 e[8];
#pragma unroll
              // unconditional (clamped) loads + select on the value: a conditional load forces a wait at its merge
              // point and serialises the batch
              for (int j = 0; j < 8; j++) {
                e[j] = row[min(s0 + j, w - 1) * 64];
                e[j].y = (s0 + j < w) ? e[j].y : 0;
              }
#pragma unroll
              for (int j = 0; j < 8; j++) {
                const float a = __int_as_float(e[j].y);
                ax = fmaf(a, lp[e[j].x], ax); ay = fmaf(a, lp[NP + e[j].x], ay); az = fmaf(a, lp[2 * NP + e[j].x], az);
              }
            }
          }
          if (i < N) __builtin_nontemporal_store(v4f{ax, ay, az, 0.f}, (v4f *) &ap4[i]);   // keep P's ELL stream in L2
          part += lp[i] * ax + lp[NP + i] * ay + lp[2 * NP + i] * az;
        }
        PH(2)
        const double pAp = block_sum<THREADS>((double) part, red);
        PH(3)
        const float alpha = (float) (rz / pAp);
        part = 0.f;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int i = tid + k * THREADS;
          const float di = (i < N) ? S.dinv[i] : 0.f;
          const v4f q = (i < N) ? __builtin_nontemporal_load((const v4f *) &ap4[i]) : v4f{0.f, 0.f, 0.f, 0.f};
          const float apk[3] = {q.x, q.y, q.z};
#pragma unroll
          for (int c = 0; c < 3; c++) {
            xx[k][c] = fmaf(alpha, lp[c * NP + i], xx[k][c]);
            rr[k][c] = fmaf(-alpha, apk[c], rr[k][c]);
            part = fmaf(rr[k][c] * di, rr[k][c], part);
          }
        }
        const double rz_new = block_sum<THREADS>((double) part, red);
        it++; cg_total++;
        if (!(rz_new > stop)) break;
        const float beta = (float) (rz_new / rz);
        rz = rz_new;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
          const int i = tid + k * THREADS;
          const float di = (i < N) ? S.dinv[i] : 0.f;
#pragma unroll
          for (int c = 0; c < 3; c++) lp[c * NP + i] = fmaf(beta, lp[c * NP + i], rr[k][c] * di);
        }
        PH(4)
      }
    }
    // ---- update + convergence (Simulation.cpp:1268, 1310-1373) ----
    part = 0.f;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int i = tid + k * THREADS;
      if (i < N) {
        f3 d = mk(xx[k][0], xx[k][1], xx[k][2]);
        st3(vnow, i, N, ld3(vnow, i, N) + d);
        part += dot(d, d);
      }
    }
    xdiff = (double) h * sqrt(block_sum<THREADS>((double) part, red)) / (double) N;
    PH(5)
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
    if (++since_progress >= A.stall_window) { stalled = true; break; }   // fp32 floor, see dc_forward.hip
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
  PH_PRINT
}

template <int THREADS, int VPT>
static void launch_res(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st) {
  const size_t lds = (size_t) 3 * THREADS * VPT * sizeof(float);
  static bool configured[kMaxDevices] = {};            // the attribute is per device
  int dev = 0;
  (void) hipGetDevice(&dev);
  if (dev < 0 || dev >= kMaxDevices || !configured[dev]) {
    (void) hipFuncSetAttribute((const void *) k_pd_step_res<THREADS, VPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (dev >= 0 && dev < kMaxDevices) configured[dev] = true;
  }
  hipLaunchKernelGGL((k_pd_step_res<THREADS, VPT>), dim3(B), dim3(THREADS), lds, st, S.self_dev, W, A);
}

// Picks (threads, vertices per thread) so that THREADS * VPT >= N with as many waves as the register budget allows.
bool launch_pd_step_resident(const DevSystem &S, const DevWork &W, const FwdArgs &A, int B, hipStream_t st, int variant) {
  const int N = S.N;
  if (N <= 256) launch_res<256, 1>(S, W, A, B, st);
  else if (N <= 512) launch_res<256, 2>(S, W, A, B, st);
  else if (N <= 1024) launch_res<256, 4>(S, W, A, B, st);
  else if (N <= 1536) launch_res<256, 6>(S, W, A, B, st);
  else if (N <= 2048) launch_res<512, 4>(S, W, A, B, st);
  else if (N <= 4096) launch_res<512, 8>(S, W, A, B, st);
  else if (N <= 6144) launch_res<512, 12>(S, W, A, B, st);
  else if (N <= 8192) launch_res<1024, 8>(S, W, A, B, st);
  else if (N <= 10240) { if (variant == 1) launch_res<512, 20>(S, W, A, B, st); else launch_res<1024, 10>(S, W, A, B, st); }
  else if (N <= 12288) { if (variant == 1) launch_res<512, 24>(S, W, A, B, st); else launch_res<1024, 12>(S, W, A, B, st); }
  else return false;
  return true;
}

}  // namespace dc
