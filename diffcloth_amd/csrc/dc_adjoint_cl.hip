// CDNA4 (gfx950) adjoint step, split variant: one rollout is run by K workgroups (dc_cluster.h), part p owning the vertex rows
// [p R, (p + 1) R). Same algorithm as dc_adjoint.hip, direct solve (adjoint_mode 1: mixed-precision refinement whose fp32 correction solves are
// preconditioned CG first — diag(P) instances — and block-Jacobi preconditioned BiCGSTAB otherwise, on
// K = M + h^2 (A - dp/dx)^T A (I + dr_df)^T, the semantics of Simulation::solveDirect, Simulation.cpp:1431-1440, inside
// Simulation::stepBackward, :1455-1780). The Krylov vectors stay in global memory and every part touches only its own rows of
// them; what crosses the parts:
//   * the input of an operator application over the reach of the element windows: its HB boundary rows travel as granules when
//     the vector is written and land in a small LDS cache the window staging reads;
//   * the partial sums of every dot product (three exchanges per CG iteration = per operator application, five per BiCGSTAB iteration = per two);
//   * with self contacts (which couple arbitrary vertices) y = (I + dr_df)^T z is formed in global memory instead: own rows by
//     every part, the layered transposed pass by part 0, fence barriers in between.
#define DC_KERNEL_TU
#include <cstdlib>
#include "dc_devlib.h"
#include "dc_winlib.h"
#include "dc_cluster.h"
#include "dc_adjprecond.h"
#include "dc_adjoint64.h"
#include <algorithm>

#ifdef DC_PROFILE_PHASES
#define CAPH_DECL long long caph_t = clock64(); long long caph[4] = {0, 0, 0, 0};
#define CAPH(k) { long long n_ = clock64(); caph[k] += n_ - caph_t; caph_t = n_; }
#define CAPH_PRINT if (blockIdx.x == 0 && threadIdx.x == 0) printf("[phases adj-cl] iters %d cycles %d | per step: setup %lld fp32 solves %lld fp64 residuals %lld final %lld cycles\n", iters, cycles, caph[0], caph[1], caph[2], caph[3]);
#else
#define CAPH_DECL
#define CAPH(k)
#define CAPH_PRINT
#endif

namespace dc {

constexpr int kMaxRefine = 6;          // fp32 correction solves of the mixed-precision direct adjoint solve before the fp64 fall-back
constexpr double kInnerFloor = 1e-3;   // (dc_adjoint.hip)
constexpr double kFallbackGain = 1e-4;

typedef int v2i __attribute__((ext_vector_type(2)));

// The parts of one rollout as a Team of dc_adjoint64.h: row range of this part, barriers and sums through the granule exchange
// (partial sums travel as fp32: 1e-7 relative on a Krylov scalar or a residual norm), y through sc1 accesses.
template <int THREADS>
struct TeamParts {
  Xch X;
  int N, part_, K_, r0_, r1_;
  __device__ __forceinline__ int r0() const { return r0_; }
  __device__ __forceinline__ int r1() const { return r1_; }
  __device__ __forceinline__ bool leader() const { return part_ == 0; }
  __device__ __forceinline__ int part() const { return part_; }
  __device__ __forceinline__ int parts() const { return K_; }
  __device__ __forceinline__ bool barrier() { return xch_barrier<THREADS>(X); }
  // a and b travel as full doubles (two words of a granule), one exchange each (fp32 partial sums stall the fp64 BiCGSTAB of an
  // ill-conditioned system: 1e-4 on the 7 742-vertex dress, measured r03c); c as fp32 with a
  __device__ __forceinline__ bool sum3(double a, double b, double c, double (&s)[3]) {
    double sa, sc, sb = 0, sd;
    if (!xch_allsum_d<THREADS>(X, a, (float) c, sa, sc)) return false;
    if (!xch_allsum_d<THREADS>(X, b, 0.f, sb, sd)) return false;
    s[0] = sa; s[1] = sb; s[2] = sc;
    return true;
  }
  // all-parts sums of n <= 2 HB fp64 values every part holds in LDS (in place), ONE exchange: part p publishes value j as the 64 bits of a
  // granule in its boundary-row slot j (no halo is in flight during the preconditioner), thread (p, j) of every part fetches it into
  // `gather` [K][n] and the first n threads add the parts up in part order — identical totals everywhere. Needs n K <= THREADS.
  __device__ __forceinline__ bool allsum_lds(double *vals, int n, double *gather) {
    const int tid = threadIdx.x;
    __syncthreads();
    xch_begin(X);
    if (tid < n) {
      const double v = vals[tid];
      v4i g = {__double2loint(v), __double2hiint(v), 0, (int) X.seq};
      xch_store(X, g, xch_off(X, X.part, kXchWaves + tid));
    }
    bool ok = true;
    if (tid < n * K_) {
      v4i g;
      ok = xch_poll(X, xch_off(X, tid / n, kXchWaves + tid % n), g);
      gather[tid] = __hiloint2double(g.y, g.x);
    }
    if (!ok) *X.ldead = 1;
    __syncthreads();
    if (tid < n) {
      double t = 0;
      for (int p = 0; p < K_; p++) t += gather[p * n + tid];
      vals[tid] = t;
    }
    __syncthreads();
    return *X.ldead == 0;
  }
  struct YV {
    __amdgpu_buffer_rsrc_t rs;
    bool same;
    __device__ __forceinline__ double ld(int idx) const {
      const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rs, idx * 8, 0, 16);
      return __hiloint2double(q.y, q.x);
    }
    __device__ __forceinline__ void st(int idx, double v) const {
      const v2i q = {__double2loint(v), __double2hiint(v)};
      if (same) __builtin_amdgcn_raw_buffer_store_b64(q, rs, idx * 8, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b64(q, rs, idx * 8, 0, 16);
    }
  };
  __device__ __forceinline__ YV yv(double *y) const { return yv(y, N); }
  __device__ __forceinline__ YV yv(double *y, int n) const { return YV{__builtin_amdgcn_make_buffer_rsrc((void *) y, 0, 3 * n * 8, 0x00020000), X.same_xcd}; }
};

namespace {

struct AdjCl {
  const float *xnew, *rec_f, *rec_n, *mu;
  const int *rec_prim;
  BufVec yb;                    // y = (I + dr_df)^T z of the rollout, read across parts: write-through stores, L1-bypassing loads
  float *lds, *hc;              // element-window LDS; halo cache: 3 planes of 2 HB floats
  int lds_floats;
  SelfRec self;
  int nself, b, part, r0, r1, R, HB, w0, w1;
};

// w = dr_df^T z for the (block-diagonal) primitive contacts: Simulation::calculatedr_df (Simulation.cpp:700-711)
__device__ __forceinline__ f3 contact_JT_cl(const DevSystem &S, const AdjCl &C, int i, f3 z) {
  const int prim = C.rec_prim[i];
  if (prim < 0) return mk(0, 0, 0);
  const int N = S.N;
  f3 n = ld3(C.rec_n, i, N);
  f3 d = ld3(C.rec_f, i, N) - prim_vout(S.prims[prim], n) * S.mass[i];
  return dri_dfi_T(n, d, C.mu[S.prims[prim].group], z);
}

// store the halo rows an exchange delivered into the LDS cache (all threads; followed by a barrier at the caller)
template <int THREADS, int HPT>
__device__ __forceinline__ void halo_to_cache(const AdjCl &C, const f3 (&hv)[HPT]) {
#pragma unroll
  for (int q = 0; q < HPT; q++) {
    const int j = threadIdx.x + q * THREADS;
    if (j < 2 * C.HB) { C.hc[j] = hv[q].x; C.hc[2 * C.HB + j] = hv[q].y; C.hc[4 * C.HB + j] = hv[q].z; }
  }
}

// out = K z on the own rows, z = zin (optionally scaled by D^-1: right preconditioning); the boundary rows of zin are in the
// halo cache. Also returns this thread's partial sums of out.d1 and out.out (d1 may be null). Ends WITHOUT a barrier.
// Returns false when an exchange timed out.
template <int THREADS>
__device__ __forceinline__ bool adjoint_operator_cl(const DevSystem &S, const DevCluster &CL, const AdjCl &C, Xch &X, const float *zin,
                                                    bool precond, float *out, const float *d1, float &dot1, float &dot2) {
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
    // layered self contacts couple vertices of different parts: y = (I + dr_df)^T z goes through global memory
    for (int i = C.r0 + threadIdx.x; i < C.r1; i += THREADS) {
      f3 z = ld3(zin, i, N);
      if (precond) z = z * S.dinv[i];
      st3c(C.yb, i, z);
    }
    if (!xch_barrier<THREADS>(X)) return false;
    if (C.part == 0) {
      if (!self_JT_layers_lds_v<THREADS>(S, C.self, C.b, C.yb, C.lds, C.lds_floats)) self_JT_layers_v<THREADS>(S, C.self, C.b, C.yb);
    }
    if (!xch_barrier<THREADS>(X)) return false;
    element_windows_t<THREADS>(CL, C.w0, C.w1, C.lds, [&](int i) {
      f3 z = ld3c(C.yb, i);
      return z + contact_JT_cl(S, C, i, z);
    }, In2Plain{C.xnew, N}, AdjTriOp{h2}, AdjBendOp{h2}, vert);
  } else {
    // primitive contacts only: dr_df is block diagonal, y_i is formed per vertex while the window is staged
    const int lo_own = C.r0, hi_own = C.r0 + C.R, HB = C.HB;
    element_windows_t<THREADS>(CL, C.w0, C.w1, C.lds, [&](int i) {
      f3 z;
      if (i >= lo_own && i < hi_own) z = ld3(zin, i, N);
      else { const int j = i < lo_own ? i - (lo_own - HB) : HB + (i - hi_own); z = mk(C.hc[j], C.hc[2 * HB + j], C.hc[4 * HB + j]); }
      if (precond) z = z * S.dinv[i];
      return z + contact_JT_cl(S, C, i, z);
    }, In2Plain{C.xnew, N}, AdjTriOp{h2}, AdjBendOp{h2}, vert);
  }
  dot1 = a1; dot2 = a2;
  return true;
}

}  // namespace

template <int THREADS, bool BLK, bool COARSE = false>
__global__ __launch_bounds__(THREADS) void k_adjoint_step_cl(const DevSystem *__restrict__ Sp, const DevCluster *__restrict__ Cp, DevWork W,
                                                             BwdArgs A, int b0, int nb_real, int hc_off, int tail_off) {
  const DevSystem &S = *Sp;
  const DevCluster &CL = *Cp;
  constexpr int HPT = (1024 + THREADS - 1) / THREADS;
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  const int tid = threadIdx.x;
  const int N = S.N, K = CL.K, R = CL.R, HB = CL.HB;
  int lb, part;
  cluster_map(K, lb, part);
  if (lb >= nb_real) return;       // padding workgroups: the launch is rounded up to a multiple of 8 rollouts (see the launcher)
  if (CL.test_drop && lb == 0 && part == K - 1) return;      // test hook: a part that never arrives (tests/test_gpu_cluster.py)
  const int b = b0 + lb;
  Xch X = xch_init(CL, lb, part, dyn_lds + tail_off);
  if (!xch_hello<THREADS>(X)) return;
  const int r0 = part * R, r1 = min(N, r0 + R);
  const size_t off = (size_t) b * 3 * N;
  double sums[3];
  f3 hv[HPT];
  f3 none[1];

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
  AdjCl C;
  C.lds = dyn_lds; C.lds_floats = hc_off; C.hc = dyn_lds + hc_off;
  C.xnew = A.x_new + off; C.rec_f = A.rec_f + off; C.rec_n = A.rec_n + off;
  C.rec_prim = A.rec_prim + (size_t) b * N;
  C.mu = A.mu + (size_t) b * S.ngroups;
  C.yb = buf_vec(W.vbest + off, N, X.same_xcd);
  C.self = A.self; C.b = b; C.part = part; C.r0 = r0; C.r1 = r1; C.R = R; C.HB = HB;
  C.w0 = part * CL.wpp; C.w1 = min(CL.nwin, C.w0 + CL.wpp);
  C.nself = (S.contact_enabled && S.self_enabled) ? A.self.meta[(size_t) b * kMetaStride] : 0;
  float *gx = A.gx + off;
  float *gin = W.g + off, *u = W.vnow + off;
  float *r = W.cg_r + off, *p = W.cg_p + off, *v = W.cg_ap + off, *t = W.cg_x + off, *rhat = W.sd_sx + off;   // (detection scratch, idle here)
  float *ph = W.pre_p + off, *sh = W.pre_s + off, *minv = W.minv + (size_t) b * 9 * N;   // M^-1 p, M^-1 s, the block inverses
  if constexpr (BLK) {   // K's own 3 x 3 diagonal blocks at this step's x_new, inverted (dc_adjprecond.h); own rows
    for (int i = r0 + tid; i < r1; i += THREADS)
      store_block_inverse(elastic_diag_block(S, C.xnew, i), S.mass[i], [&](f3 e) { return contact_JT_cl(S, C, i, e); }, minv, i, N);
  }
  // BLK: M^-1 = those block inverses, the vectors M^-1 p / M^-1 s are stored and are what the operator is applied to;
  // otherwise M^-1 = diag(P)^-1, applied inside the operator
  auto pre = [&](int i, f3 z) { return BLK ? block_pre(minv, i, N, z) : z; };

  // ---- gradient clipping (Simulation.cpp:1460-1466), u = 0, r = rhat = p = g ----
  float part_s = 0.f;
  for (int i = r0 + tid; i < r1; i += THREADS) { f3 q = ld3(gx, i, N); part_s += dot(q, q); }
  if (!xch_allsum<THREADS>(X, part_s, 0.f, 0.f, sums)) return;
  double gnorm = sqrt(sums[0]);
  float gscale = 1.f;
  int clipped = 0;
  if (A.clip && gnorm > (double) A.clip_thr * N) { gscale = (float) ((double) A.clip_thr * N / gnorm); clipped = 1; gnorm = (double) A.clip_thr * N; }
  int status = gnorm > 0 ? 0 : 1;          // 1 converged (a zero gradient has the solution u = 0), 2 stalled at the fp32 floor / breakdown, 0 cap hit
  int iters = 0, cg_total = 0;
  double udiff = 0;
  for (int i = r0 + tid; i < r1; i += THREADS) { st3(gin, i, N, ld3(gx, i, N) * gscale); st3(u, i, N, mk(0, 0, 0)); }
  // fp64 side (dc_adjoint64.h): true residual of the mixed-precision refinement, fall-back solve, gradient assembly — the same
  // code as the one-workgroup kernel, over the parts of this rollout
  TeamParts<THREADS> tm{X, N, part, K, r0, r1};
  Adj64 C64;
  C64.xprev = A.x_prev + off; C64.vnew = A.v_new + off;
  C64.xnew = C.xnew; C64.rec_f = C.rec_f; C64.rec_n = C.rec_n; C64.mu = C.mu; C64.rec_prim = C.rec_prim;
  C64.self = C.self; C64.nself = C.nself; C64.b = b; C64.lds = dyn_lds; C64.lds_floats = hc_off;
  adj64_inject(C64, A, b, N, S.self_cap);
  Work64 W64;
  W64.u = W.u64 + off; W64.r = W.r64 + off; W64.y = W.y64 + off; W64.x = W.x64 + off; W64.corner = W.c64 + (size_t) b * 3 * S.NC;
  W64.rhat = W.k64[0] + off; W64.p = W.k64[1] + off; W64.v = W.k64[2] + off; W64.t = W.k64[3] + off; W64.ph = W.k64[4] + off; W64.sh = W.k64[5] + off;
  int cycles = 0, iters64 = 0, verified = 1;
  CAPH_DECL
  for (int i = r0 + tid; i < r1; i += THREADS) st3d(W64.u, i, N, mkd(0, 0, 0));
  tm.X = X;
  if (!prepare_x64<THREADS>(S, C64, tm, W64.x)) return;
  X = tm.X;
  CAPH(0)
  double rr_true = gnorm * gnorm;
  if (gnorm > 0) {
    // ---- direct solve of K u = g in mixed precision (see dc_adjoint.hip): fp32 BiCGSTAB for corrections of the fp64 residual ----
    const double stop = (double) A.rel_tol * (double) A.rel_tol * gnorm * gnorm;
    // (meshes the engine found ill-conditioned: where the fp64 fall-back of THIS instance has the coarse level — the condition of
    // bicgstab64, dc_adjoint64.h — the fp32 solve hands over after 400 iterations; everywhere else it keeps its budget, the fall-back
    // there is block-Jacobi only and much slower per digit: ADVICE r04)
    const bool fb_coarse = COARSE && S.defl_u != nullptr && S.adj_coarse && hc_off >= kCoarseLdsFloats;
    const int kcap = std::min(A.it_cap > 0 ? 4 * A.it_cap : 1600, fb_coarse ? 400 : 1 << 30);
    constexpr int VB = 4;
    bool fallback = false;
    double rr = rr_true;
    double op_err = 0;         // measured error of the fp32 operator (see below)
    bool use_cg = !BLK && !COARSE && A.cg_first != 0;      // correction solves by CG first, BiCGSTAB once a CG cycle has not delivered (dc_adjoint.hip: cg32_solve)
    status = (rr_true <= stop) ? 1 : 0;
    for (int kdone = 0; status == 0 && !fallback; cycles++) {
    const double rel_now = sqrt(rr_true) / gnorm;
    const double in_tol = A.fp32_only ? (double) A.rel_tol : fmax(0.3 * (double) A.rel_tol / rel_now, kInnerFloor);
    const bool cg_cycle = use_cg;
    int in_status = 0;
    if (cg_cycle) {
      // ---- preconditioned CG (diag(P)^-1) on K d = rhs, on a short leash (dc_adjoint.hip: cg32_solve): one operator application and three
      //      exchanges per iteration (p.Kp; r.D^-1 r and r.r; the boundary rows of the new p) where BiCGSTAB takes two and five ----
      xch_begin(X);
      part_s = 0.f;
      float part_z = 0.f;
      for (int l = tid; l < R; l += THREADS) {
        const int i = r0 + l;
        f3 z = mk(0, 0, 0);
        if (i < N) {
          const f3 q = ld3(gin, i, N);
          z = q * S.dinv[i];
          st3(r, i, N, q); st3(p, i, N, z); st3(u, i, N, mk(0, 0, 0));
          part_s += dot(q, q); part_z += dot(q, z);
        }
        xch_publish_boundary(X, l, R, z.x, z.y, z.z);      // the operator's input: p = D^-1 r
      }
      xch_publish_sums(X, part_s, part_z, 0.f);
      if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
      halo_to_cache<THREADS, HPT>(C, hv);
      __syncthreads();
      rr = sums[0];
      double rz = sums[1];
      const double in_stop = in_tol * in_tol * rr;
      double best_rr = rr;
      int since_progress = 0, its = 0;
      in_status = (rr <= in_stop || !(rr > 0)) ? 1 : 0;
      for (int k = 2 * kdone; k < 2 * kcap && in_status == 0 && its < kCgCycleCap; k++) {
        float d1, d2;
        if (!adjoint_operator_cl<THREADS>(S, CL, C, X, p, false, v, p, d1, d2)) return;
        if (!xch_allsum<THREADS>(X, d1, 0.f, 0.f, sums)) return;
        const double pv = sums[0];
        if (!(pv > 1e-300)) { in_status = 2; break; }
        const float alpha = (float) (rz / pv);
        float pa = 0.f, pb = 0.f;
        for (int l0 = tid; l0 < R; l0 += VB * THREADS) {
          f3 rq[VB], vq[VB], pq[VB], uq[VB];
          float dq[VB];
#pragma unroll
          for (int j = 0; j < VB; j++) {
            const int ic = min(r0 + l0 + j * THREADS, r1 - 1);
            rq[j] = ld3(r, ic, N); vq[j] = ld3(v, ic, N); pq[j] = ld3(p, ic, N); uq[j] = ld3(u, ic, N); dq[j] = S.dinv[ic];
          }
#pragma unroll
          for (int j = 0; j < VB; j++) {
            const int i = r0 + l0 + j * THREADS;
            const f3 rn = rq[j] - vq[j] * alpha;
            if (i < r1) { st3(r, i, N, rn); st3(u, i, N, uq[j] + pq[j] * alpha); pa += dot(rn, rn) * dq[j]; pb += dot(rn, rn); }
          }
        }
        if (!xch_allsum<THREADS>(X, pa, pb, 0.f, sums)) return;
        const double rz_new = sums[0];
        rr = sums[1];
        its++;
        if (rr <= in_stop) { in_status = 1; break; }
        if (rr < best_rr) { best_rr = rr; since_progress = 0; }
        else if (++since_progress >= min(A.stall_window, kCgStall)) { in_status = 2; break; }
        if (!(rr < 1e8 * best_rr)) { in_status = 2; break; }
        const float beta = (float) (rz_new / rz);
        rz = rz_new;
        xch_begin(X);
        for (int l0 = tid; l0 < R; l0 += VB * THREADS) {
          f3 rq[VB], pq[VB];
          float dq[VB];
#pragma unroll
          for (int j = 0; j < VB; j++) { const int ic = min(r0 + l0 + j * THREADS, r1 - 1); rq[j] = ld3(r, ic, N); pq[j] = ld3(p, ic, N); dq[j] = S.dinv[ic]; }
#pragma unroll
          for (int j = 0; j < VB; j++) {
            const int l = l0 + j * THREADS, i = r0 + l;
            f3 pn = rq[j] * dq[j] + pq[j] * beta;
            if (i >= r1) pn = mk(0, 0, 0);
            if (i < r1) st3(p, i, N, pn);
            if (l < R) xch_publish_boundary(X, l, R, pn.x, pn.y, pn.z);
          }
        }
        xch_publish_sums(X, 0.f, 0.f, 0.f);
        if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
        halo_to_cache<THREADS, HPT>(C, hv);
        __syncthreads();
      }
      cg_total += its;
      kdone += (its + 1) / 2;
    } else {
    // r = rhat = p = rhs (gin: g, later the fp64 residual rounded to fp32), d = 0; the operator's input travels to the neighbours
    xch_begin(X);
    part_s = 0.f;
    for (int l = tid; l < R; l += THREADS) {
      const int i = r0 + l;
      f3 q = mk(0, 0, 0);
      if (i < N) {
        q = ld3(gin, i, N);
        st3(r, i, N, q); st3(rhat, i, N, q); st3(p, i, N, q); st3(u, i, N, mk(0, 0, 0));
        part_s += dot(q, q);
        if constexpr (BLK) { q = pre(i, q); st3(ph, i, N, q); }
      }
      xch_publish_boundary(X, l, R, q.x, q.y, q.z);      // the operator's input: M^-1 p (BLK), or p itself, scaled by diag(P)^-1 inside the operator
    }
    xch_publish_sums(X, part_s, 0.f, 0.f);
    if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
    halo_to_cache<THREADS, HPT>(C, hv);
    __syncthreads();
    double rho = sums[0];
    rr = rho;
    const double in_stop = in_tol * in_tol * rho;
    double best_rr = rr;
    int since_progress = 0;
    in_status = (rr <= in_stop || !(rr > 0)) ? 1 : 0;
    for (int k = kdone; k < kcap && in_status == 0; k++, kdone++) {
      float d1, d2;
      // v = K M^-1 p ;  alpha = rho / (rhat . v)
      if (!adjoint_operator_cl<THREADS>(S, CL, C, X, BLK ? ph : p, !BLK, v, rhat, d1, d2)) return;
      if (!xch_allsum<THREADS>(X, d1, 0.f, 0.f, sums)) return;
      const double rv = sums[0];
      if (!(fabs(rv) > 1e-300)) { in_status = 2; break; }
      const float alpha = (float) (rho / rv);
      // s = r - alpha v  (in place), its boundary rows to the neighbours
      xch_begin(X);
      part_s = 0.f;
      for (int l0 = tid; l0 < R; l0 += VB * THREADS) {
        f3 rq[VB], vq[VB];
#pragma unroll
        for (int j = 0; j < VB; j++) { const int ic = min(r0 + l0 + j * THREADS, r1 - 1); rq[j] = ld3(r, ic, N); vq[j] = ld3(v, ic, N); }
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int l = l0 + j * THREADS, i = r0 + l;
          f3 s = rq[j] - vq[j] * alpha;
          if (i >= r1) s = mk(0, 0, 0);
          if (i < r1) { st3(r, i, N, s); part_s += dot(s, s); if constexpr (BLK) { s = pre(i, s); st3(sh, i, N, s); } }
          if (l < R) xch_publish_boundary(X, l, R, s.x, s.y, s.z);      // boundary rows of M^-1 s
        }
      }
      {
        xch_publish_sums(X, part_s, 0.f, 0.f);
        if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
        halo_to_cache<THREADS, HPT>(C, hv);
        __syncthreads();
      }
      const double ss = sums[0];
      iters++;
      if (ss <= in_stop) {
        for (int i = r0 + tid; i < r1; i += THREADS) st3(u, i, N, ld3(u, i, N) + (BLK ? ld3(ph, i, N) * alpha : ld3(p, i, N) * (alpha * S.dinv[i])));
        rr = ss; in_status = 1; break;
      }
      // t = K M^-1 s ;  omega = (t . s) / (t . t)
      if (!adjoint_operator_cl<THREADS>(S, CL, C, X, BLK ? sh : r, !BLK, t, r, d1, d2)) return;
      if (!xch_allsum<THREADS>(X, d1, d2, 0.f, sums)) return;
      const double ts = sums[0], tt = sums[1];
      if (!(tt > 1e-300)) { in_status = 2; break; }
      const float omega = (float) (ts / tt);
      // d += alpha D^-1 p + omega D^-1 s ;  r = s - omega t ;  rho_new = rhat . r
      float pa = 0.f, pb = 0.f;
      for (int l0 = tid; l0 < R; l0 += VB * THREADS) {
        f3 sq[VB], uq[VB], pq[VB], tq[VB], hq[VB], zq[VB];
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int ic = min(r0 + l0 + j * THREADS, r1 - 1);
          sq[j] = ld3(r, ic, N); uq[j] = ld3(u, ic, N); tq[j] = ld3(t, ic, N); hq[j] = ld3(rhat, ic, N);
          if constexpr (BLK) { zq[j] = ld3(sh, ic, N); pq[j] = ld3(ph, ic, N); }
          else { const float di = S.dinv[ic]; zq[j] = sq[j] * di; pq[j] = ld3(p, ic, N) * di; }
        }
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int i = r0 + l0 + j * THREADS;
          f3 rn = sq[j] - tq[j] * omega;
          if (i < r1) {
            st3(u, i, N, uq[j] + pq[j] * alpha + zq[j] * omega);
            st3(r, i, N, rn);
            pa += dot(rn, hq[j]);
            pb += dot(rn, rn);
          }
        }
      }
      if (!xch_allsum<THREADS>(X, pa, pb, 0.f, sums)) return;
      const double rho_new = sums[0];
      rr = sums[1];
      if (rr <= in_stop) { in_status = 1; break; }
      if (rr < best_rr) { best_rr = rr; since_progress = 0; }
      else if (++since_progress >= (fb_coarse ? min(A.stall_window, kCoarseStall) : A.stall_window)) { in_status = 2; break; }      // (dc_adjoint.hip: early hand-over)
      if (!(fabs(rho_new) > 1e-300) || !(fabs(omega) > 0.f)) { in_status = 2; break; }
      const float beta = (float) ((rho_new / rho) * ((double) alpha / (double) omega));
      rho = rho_new;
      // p = r + beta (p - omega v), its boundary rows to the neighbours
      xch_begin(X);
      for (int l0 = tid; l0 < R; l0 += VB * THREADS) {
        f3 rq[VB], pq[VB], vq[VB];
#pragma unroll
        for (int j = 0; j < VB; j++) { const int ic = min(r0 + l0 + j * THREADS, r1 - 1); rq[j] = ld3(r, ic, N); pq[j] = ld3(p, ic, N); vq[j] = ld3(v, ic, N); }
#pragma unroll
        for (int j = 0; j < VB; j++) {
          const int l = l0 + j * THREADS, i = r0 + l;
          f3 pn = rq[j] + (pq[j] - vq[j] * omega) * beta;
          if (i >= r1) pn = mk(0, 0, 0);
          if (i < r1) { st3(p, i, N, pn); if constexpr (BLK) { pn = pre(i, pn); st3(ph, i, N, pn); } }
          if (l < R) xch_publish_boundary(X, l, R, pn.x, pn.y, pn.z);      // boundary rows of M^-1 p
        }
      }
      xch_publish_sums(X, 0.f, 0.f, 0.f);
      if (!xch_finish<THREADS, HPT, true>(X, sums, hv)) return;
      halo_to_cache<THREADS, HPT>(C, hv);
      __syncthreads();
    }
    }      // (BiCGSTAB cycle)
    __syncthreads();
    // u += d (fp64)
    for (int i = r0 + tid; i < r1; i += THREADS) st3d(W64.u, i, N, ld3d(W64.u, i, N) + tod(ld3(u, i, N)));
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
    CAPH(1)
    tm.X = X;
    auto rs = residual64<THREADS>(S, C64, tm, W64, gx, gscale);
    tm = rs.tm; X = tm.X;
    CAPH(2)
    if (rs.res < 0) return;
    double rr_new = rs.rr;
    op_err = fmax(op_err, fabs(sqrt(rr_new) - sqrt(rr)) / sqrt(rr_true));
    if (rr_new <= stop) { rr_true = rr_new; status = 1; cycles++; break; }
    if (!(rr_new < rr_true)) {      // a diverged correction (NaN-safe): take it back
      for (int i = r0 + tid; i < r1; i += THREADS) st3d(W64.u, i, N, ld3d(W64.u, i, N) - tod(ld3(u, i, N)));
      __syncthreads();
      rs = residual64<THREADS>(S, C64, tm, W64, gx, gscale);
      tm = rs.tm; X = tm.X;
      if (rs.res < 0) return;
      rr_new = rs.rr;
    }
    if (!(rr_new < 0.0625 * rr_true) || in_status != 1 || cycles + 1 >= kMaxRefine || kdone >= kcap) {
      if (cg_cycle && cycles + 1 < kMaxRefine && kdone < kcap) use_cg = false;      // a CG cycle that did not deliver: BiCGSTAB for the rest of the step
      else fallback = true;
    }
    rr_true = rr_new;
    if (!fallback) {
      for (int i = r0 + tid; i < r1; i += THREADS) st3(gin, i, N, tof(ld3d(W64.r, i, N)));
      __syncthreads();
    }
    }   // cycle
    if (fallback) {
      // ---- fp64 BiCGSTAB on the same operator from (u, r): the reference's SparseLU always returns a solution ----
      if constexpr (!BLK) {
        for (int i = r0 + tid; i < r1; i += THREADS)
          store_block_inverse(elastic_diag_block(S, C.xnew, i), S.mass[i], [&](f3 e) { return contact_JT_cl(S, C, i, e); }, minv, i, N);
      }
      __syncthreads();
      const double stop_fb = fmax(stop * kFallbackGain * kFallbackGain, 1e-26 * gnorm * gnorm);     // (dc_adjoint.hip)
      double rr64 = rr_true;
      for (int pass = 0; pass < 3; pass++) {
        tm.X = X;
        auto r64 = bicgstab64<THREADS, COARSE>(S, C64, tm, W64, minv, stop_fb, 20000, rr64, iters64);
        tm = r64.tm; X = tm.X;
        if (r64.res < 0) return;
        iters64 = r64.iters;
        auto rc = residual64<THREADS>(S, C64, tm, W64, gx, gscale);     // the recurrence drifts over thousands of iterations: check, go again
        tm = rc.tm; X = tm.X;
        if (rc.res < 0) return;
        rr64 = rc.rr;
        if (rr64 <= stop) status = 1;
        if (rr64 <= stop_fb || r64.res == 0) break;
      }
      rr_true = rr64;
    }
  }
  udiff = sqrt(rr_true) / (gnorm > 0 ? gnorm : 1.0);     // relative residual: fp64-evaluated (mixed precision) or the fp32 recurrence's
  __syncthreads();
  CAPH(1)
  // ---- gradients w.r.t. the previous state and parameters (Simulation.cpp:1534, 1608-1650), in fp64 from u ----
  {
    tm.X = X;
    auto rf = finish_gradients64<THREADS>(S, C64, tm, W64, A, W.vbest + off);
    tm = rf.tm; X = tm.X;
    if (rf.res < 0) return;
  }
  CAPH(3)
  CAPH_PRINT
  if (tid == 0 && part == 0) {
    dc_bwd_stats s;
    s.converged = status; s.adjoint_iters = iters; s.cg_iters = cg_total; s.clipped = clipped;
    s.used_direct = 1; s.last_udiff = (float) udiff;
    s.refine_cycles = cycles; s.fp64_iters = iters64; s.residual_verified = A.fp32_only ? 0 : verified;
    s.workgroups = K;
    A.stats[b] = s;
  }
  (void) none;
  }   // step
}

// nb rollouts starting at b0, K workgroups each (adjoint_mode 1 only); exchange area zeroed by the caller, K nb <= CUs.
#ifndef DC_ADJ_CL_THREADS
#define DC_ADJ_CL_THREADS 512      // round 6: 512 threads x 256 registers (241 spilled VGPRs, 556 B scratch) instead of 1024 x 128 (855 / 1 144 B): -8 % per step
#endif
hipError_t launch_adjoint_step_cluster(const DevSystem &S, const DevCluster &CL, const DevWork &W, const BwdArgs &A, int b0, int nb, hipStream_t st) {
  constexpr int THREADS = DC_ADJ_CL_THREADS;
  const int hc_off = (CL.win_lds_bytes / 4 + 3) / 4 * 4;
  const int tail_off = hc_off + 6 * CL.HB;
  const size_t lds = sizeof(float) * (size_t) (tail_off + kXchLdsFloats);
  if (lds > 160 * 1024 - 256 || A.mode != 1) return hipErrorInvalidValue;
  if (A.block_pre && S.adj_coarse && S.defl_u) {      // the fall-back with the coarse level: an instance of its own (dc_adjoint.hip launch_adj_coarse)
    hipError_t e = hipFuncSetAttribute((const void *) k_adjoint_step_cl<THREADS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_adjoint_step_cl<THREADS, true, true>), dim3((nb + 7) / 8 * 8 * CL.K), dim3(THREADS), lds, st, S.self_dev, CL.self_dev, W, A, b0, nb, hc_off, tail_off);
  } else if (A.block_pre) {
    hipError_t e = hipFuncSetAttribute((const void *) k_adjoint_step_cl<THREADS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_adjoint_step_cl<THREADS, true>), dim3((nb + 7) / 8 * 8 * CL.K), dim3(THREADS), lds, st, S.self_dev, CL.self_dev, W, A, b0, nb, hc_off, tail_off);
  } else {
    hipError_t e = hipFuncSetAttribute((const void *) k_adjoint_step_cl<THREADS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_adjoint_step_cl<THREADS, false>), dim3((nb + 7) / 8 * 8 * CL.K), dim3(THREADS), lds, st, S.self_dev, CL.self_dev, W, A, b0, nb, hc_off, tail_off);
  }
  return hipGetLastError();
}

}  // namespace dc
