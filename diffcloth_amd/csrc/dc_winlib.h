// Device side of the element windows (tables: dc_windows.h): stage -> per-element phase -> per-vertex phase,
// window by window, entirely in LDS; plus the per-element operators of the forward local step and of the adjoint.
#pragma once
#include <type_traits>
#include "dc_devlib.h"

namespace dc {

struct WinLds {
  float2 *a1xy, *a2xy, *erxy;   // (x, y) planes: input 1, input 2 over the vertex span; element result vectors
  float *a1z, *a2z, *erz;       // z planes
};

template <class TB>
__device__ __forceinline__ WinLds win_lds(const TB &S, float *lds) {
  const int vc = S.win_vcap, nr = S.win_nrcap;
  WinLds L;
  L.a1xy = (float2 *) lds; L.a2xy = (float2 *) (lds + 2 * vc); L.erxy = (float2 *) (lds + 4 * vc);
  L.a1z = lds + 4 * vc + 2 * nr; L.a2z = L.a1z + vc; L.erz = L.a2z + vc;
  return L;
}

__device__ __forceinline__ f3 ldw(const float2 *xy, const float *z, int j) { const float2 q = xy[j]; return mk(q.x, q.y, z[j]); }
__device__ __forceinline__ void stw(float2 *xy, float *z, int j, f3 v) { xy[j] = make_float2(v.x, v.y); z[j] = v.z; }

// Input 1 comes from stage1(i) (a per-vertex function of global data, evaluated while the window is staged: the adjoint
// forms y = (I + dr_df)^T z there, so y never exists in global memory), input 2 is the planar [3][N] vector in2.
// tri_op(a0,a1,a2, b0,b1,b2, D, w2, r0, r1) and bend_op(a[4], b[4], w, n, w2, res) see the values of input 1 (a) and
// input 2 (b) at the element's vertices; vert_op(i, sum, a_i) receives sum_corners coef * result and input 1 at the
// vertex, for every vertex exactly once. Call with all threads; starts with a barrier (LDS may still be in use by the
// caller) and ends WITHOUT one.
// TB = the table set (DevSystem's own windows, or the DevCluster tables of the split kernels: same member names); the windows
// [w0, w1) are processed; in2(i) returns input 2 at vertex i (In2Plain: a planar [3][N] vector in global memory; the split
// kernels read vectors other workgroups write through In2Sc1, dc_cluster.h).
struct In2Plain {
  const float *__restrict__ v;
  int N;
  __device__ __forceinline__ f3 operator()(int i) const { return mk(v[i], v[N + i], v[2 * N + i]); }
};
// Elements a thread keeps in flight in the per-element phase: the largest batch (-DDC_WIN_EB, A/B builds) and the dispatch of a batch
// size to its unrolled variant.
#ifndef DC_WIN_EB
#define DC_WIN_EB 4
#endif
constexpr int kWinMaxBatch = DC_WIN_EB;
#ifndef DC_WIN_SUB1024
#define DC_WIN_SUB1024 1
#endif
constexpr int kWinSubBatch1024 = DC_WIN_SUB1024;
#ifndef DC_WIN_LEGACY1024
#define DC_WIN_LEGACY1024 1
#endif
constexpr bool kWinLegacy1024 = DC_WIN_LEGACY1024 != 0;
// scheduling fence behind the gather pass of several elements (all their LDS reads issued before the arithmetic starts); -DDC_WIN_NOFENCE: A/B
// builds. Not with one element at a time — the 1024-thread kernels: there the fence (like any batch of more than one element) sent the
// register allocation of the 128-register kernels from 133 to 4 500 spilled registers (measured on the code objects, round 5).
#ifdef DC_WIN_NOFENCE
#define DC_WIN_GATHER_FENCE
#else
#define DC_WIN_GATHER_FENCE if constexpr (SUB > 1) __builtin_amdgcn_sched_barrier(0);
#endif
template <int B, class F>
__device__ __forceinline__ void batch_exact(int take, F f) {
  if constexpr (B <= 1) f(std::integral_constant<int, 1>());
  else {
    if (take >= B) f(std::integral_constant<int, B>());
    else batch_exact<B - 1>(take, f);
  }
}
// `left` rounds remain for this WAVE: the batch takes all of them when they fit, half of them (rounded up) when two batches do — 5 rounds run
// as 3 + 2, not 4 + 1, so that no batch is left without a partner element to overlap with — else MAXB
template <int MAXB, class F>
__device__ __forceinline__ int batch_dispatch(int left, F f) {
  const int take = left >= 2 * MAXB ? MAXB : (left > MAXB ? (left + 1) / 2 : left);
  batch_exact<MAXB>(take, f);
  return take;
}

// Sub-phase timing of the window passes (-DDC_PROFILE_PHASES): thread 0 of workgroup 0 accumulates shader-clock totals of the
// staging, triangle, bending and per-vertex parts; the step kernels print them with their own phase totals.
#ifdef DC_PROFILE_PHASES
static __device__ long long g_win_ph[4];
#define WPH_DECL long long wph_t = clock64();
#define WPH(k) { if (blockIdx.x == 0 && threadIdx.x == 0) { long long n_ = clock64(); g_win_ph[k] += n_ - wph_t; wph_t = n_; } }
#else
#define WPH_DECL
#define WPH(k)
#endif

// PRECISE: the element operators also receive the low-order parts of the element's rest data (wtri_Dlo / wbend_lo) — the
// fp64-strain operators of the forward step (PreciseTriOp / PreciseBendOp, HybridTriOp / HybridBendOp below).
// A per-vertex operator may come with a `pre(i)` member: the global loads its vertex needs (returned as a value, handed back as the
// fourth argument). The per-vertex phase calls it BEFORE the coefficient packets are consumed, for both vertices of a pair, so that
// those loads travel with the packets instead of starting after the gather — and, for the second vertex of a pair, after the first
// vertex's stores, which the compiler may not move them across (one exposed L2 round trip per vertex and window otherwise).
template <class V, class = void> struct vert_has_pre : std::false_type {};
template <class V> struct vert_has_pre<V, std::void_t<decltype(std::declval<V &>().pre(0))>> : std::true_type {};
// NOA ("no a"): the operator does not use input 1 at the vertex (its third argument). The per-vertex phase of a window then reads the result
// planes only, and the NEXT window may stage its span — which overwrites the input planes — without a barrier in between: a window costs two barriers
// instead of three, and the waves that finish their vertices first have their staging loads in flight while the others finish.
template <class P, class O, bool NOA = false>
struct VertWithPre {
  static constexpr bool kIgnoresInput1 = NOA;
  P p; O o;
  __device__ __forceinline__ auto pre(int i) { return p(i); }
  template <class Q> __device__ __forceinline__ void operator()(int i, f3 sum, f3 a, const Q &q) { o(i, sum, a, q); }
};
template <class P, class O> __device__ __forceinline__ VertWithPre<P, O> vert_with_pre(P p, O o) { return VertWithPre<P, O>{p, o}; }
template <class P, class O> __device__ __forceinline__ VertWithPre<P, O, true> vert_with_pre_noa(P p, O o) { return VertWithPre<P, O, true>{p, o}; }
template <class O>
struct VertNoA {
  static constexpr bool kIgnoresInput1 = true;
  O o;
  __device__ __forceinline__ void operator()(int i, f3 sum, f3 a) { o(i, sum, a); }
};
template <class O> __device__ __forceinline__ VertNoA<O> vert_noa(O o) { return VertNoA<O>{o}; }
template <class V, class = void> struct vert_ignores_a : std::false_type {};
template <class V> struct vert_ignores_a<V, std::enable_if_t<V::kIgnoresInput1>> : std::true_type {};

template <int EB, bool PRECISE>
struct WinTriRecs { int4 r[EB]; float4 D[EB]; float4 Dl[PRECISE ? EB : 1]; };
template <int EB, bool PRECISE>
struct WinBendRecs { int4 r[EB]; float4 w[EB]; float4 wl[PRECISE ? EB : 1]; };

template <int THREADS, bool PRECISE = false, class TB, class Stage1, class In2, class TriOp, class BendOp, class VertOp>
__device__ __forceinline__ void element_windows_t(const TB &S, int w0, int w1, float *lds, Stage1 stage1,
                                                  In2 in2, TriOp tri_op, BendOp bend_op, VertOp vert_op) {
  const int tid = threadIdx.x, lane = tid & 63;
  const WinLds L = win_lds(S, lds);
  constexpr int MB = kWinMaxBatch;
  constexpr int VPB = 6;
  constexpr bool PAIR = THREADS < 1024;
  for (int w = w0; w < w1; w++) {
    const int4 d0 = S.win[2 * w], d1 = S.win[2 * w + 1];
    const int v0 = d0.x, v1 = d0.y, lo = d0.z, vs = d0.w, toff = d1.x, nt = d1.y, boff = d1.z, nb = d1.w;
    WPH_DECL
    // record loads of a batch (clamped index, no divergence)
    auto tri_load = [&](auto nc, WinTriRecs<MB, PRECISE> &R, int t0) {
#pragma unroll
      for (int j = 0; j < decltype(nc)::value; j++) {
        const int t = min(t0 + j * THREADS, nt - 1);
        R.r[j] = S.wtri_rec[toff + t]; R.D[j] = S.wtri_D[toff + t];
        if constexpr (PRECISE) R.Dl[j] = S.wtri_Dlo[toff + t];
      }
    };
    auto bend_load = [&](auto nc, WinBendRecs<MB, PRECISE> &R, int e0) {
#pragma unroll
      for (int j = 0; j < decltype(nc)::value; j++) {
        const int e = min(e0 + j * THREADS, nb - 1);
        R.r[j] = S.wbend_rec[boff + e]; R.w[j] = S.wbend_w[boff + e];
        if constexpr (PRECISE) R.wl[j] = S.wbend_lo[boff + e];
      }
    };
    WinTriRecs<MB, PRECISE> tcur;
    WinBendRecs<MB, PRECISE> bcur;
    const int dump = 2 * nt + nb + 1 + lane;      // kWinDumpSlots result slots behind the zero vector (dc_windows.cpp: nrcap)
    constexpr bool NOA = vert_ignores_a<VertOp>::value;
    if (!NOA || w == w0) __syncthreads();      // (NOA: the previous window's per-vertex phase reads the result planes only; the staging below writes the input planes)
    // SVR span vertices per thread and round (clamped index, no divergence): their global loads overlap. A span is ~1.3 windows wide:
    // three vertices per thread stage it in ONE round of the 512-thread kernels (two rounds = two exposed memory round trips per window
    // before: 5.2 k of a window's 29 k cycles in the forward step), two in one round of the 1024-thread kernels.
    constexpr int SVR = THREADS < 1024 ? 3 : 2;
    for (int j0 = tid; j0 < vs; j0 += SVR * THREADS) {
      f3 u[SVR], sv[SVR];
#pragma unroll
      for (int q = 0; q < SVR; q++) {
        const int j = j0 + q * THREADS, i = lo + (j < vs ? j : j0);
        u[q] = in2(i); sv[q] = stage1(i);
      }
#pragma unroll
      for (int q = 0; q < SVR; q++) {
        const int j = j0 + q * THREADS;
        if (q == 0 || j < vs) { stw(L.a2xy, L.a2z, j, u[q]); stw(L.a1xy, L.a1z, j, sv[q]); }
      }
    }
    __syncthreads();
    if (tid == 0) stw(L.erxy, L.erz, 2 * nt + nb, mk(0, 0, 0));      // the zero vector the padding entries of the per-vertex rows point at (read behind the next barrier;
                                                                     // written behind this one: the result planes may be in use by the previous window until here)
    WPH(0)
    // per-element phase: up to 4 elements of a thread at a time, their records loaded up front (clamped index, no
    // divergence) so that the L2 round trips and the gather -> math chains of a batch overlap instead of queueing up.
    // The batch size follows the number of rounds left (4, 3 or 2 elements per thread: three unrolled variants) so that
    // e.g. 2250 triangles on 1024 threads cost 3 rounds, not 4. (Exactly ceil(n / THREADS) rounds behind wave-uniform
    // branches inside ONE batch was 2x slower: the branches end the overlap.)
    // A batch runs in three passes over its elements — gather the vertices from LDS and form the edge vectors (every operator works
    // on differences to the element's first vertex), compute, store the results — so that the batch is ONE basic block whose LDS reads
    // all precede its LDS writes: written element by element (read, compute, write, next element) the compiler has to keep element
    // j + 1's reads behind element j's writes (it cannot prove the planes disjoint), and the elements of a batch ran strictly one
    // after the other, each with its LDS round trips and its dependent chain of ~60 fp64-rate operations exposed at 2 waves per SIMD.
    // Elements past the end of the last round (clamped duplicates) store into the lane's dump slot behind the zero vector: a guarded
    // store would make each element an exec-masked block of its own again.
    // (SUB elements at a time: the whole batch where the registers allow it — the 512-thread kernels, 256 registers per lane — and
    // kWinSubBatch1024 = 1 in the 1024-thread kernels, whose 128 registers do not hold the edge vectors of several elements: their four
    // waves per SIMD hide what a wave's own elements cannot)
    constexpr int SUBMAX = THREADS >= 1024 ? kWinSubBatch1024 : MB;
    auto tri_compute = [&](auto ebc, const WinTriRecs<MB, PRECISE> &R, int t0) {
      constexpr int EB = decltype(ebc)::value;
      constexpr int SUB = EB < SUBMAX ? EB : SUBMAX;
      if constexpr (SUBMAX == 1 && kWinLegacy1024) {      // one element at a time, its store guarded (the 1024-thread kernels: measured 6 % faster there than the passes below)
#pragma unroll
        for (int j = 0; j < EB; j++) {
          const int t = t0 + j * THREADS;
          const int j0 = R.r[j].x & 0xffff, j1 = (int) ((unsigned) R.r[j].x >> 16), j2 = R.r[j].y;
          f3 r0, r1;
          if constexpr (PRECISE)
            tri_op(ldw(L.a1xy, L.a1z, j0), ldw(L.a1xy, L.a1z, j1), ldw(L.a1xy, L.a1z, j2), ldw(L.a2xy, L.a2z, j0),
                   ldw(L.a2xy, L.a2z, j1), ldw(L.a2xy, L.a2z, j2), R.D[j], R.Dl[j], __int_as_float(R.r[j].z), r0, r1);
          else
            tri_op(ldw(L.a1xy, L.a1z, j0), ldw(L.a1xy, L.a1z, j1), ldw(L.a1xy, L.a1z, j2), ldw(L.a2xy, L.a2z, j0),
                   ldw(L.a2xy, L.a2z, j1), ldw(L.a2xy, L.a2z, j2), R.D[j], __int_as_float(R.r[j].z), r0, r1);
          if (t < nt) { stw(L.erxy, L.erz, t, r0 * R.D[j].x + r1 * R.D[j].y); stw(L.erxy, L.erz, nt + t, r0 * R.D[j].z + r1 * R.D[j].w); }
        }
      } else
#pragma unroll
      for (int s0 = 0; s0 < EB; s0 += SUB) {
        f3 ea0[SUB], ea1[SUB], eb0[SUB], eb1[SUB];
#pragma unroll
        for (int u = 0; u < SUB; u++) {
          const int j = min(s0 + u, EB - 1);
          const int j0 = R.r[j].x & 0xffff, j1 = (int) ((unsigned) R.r[j].x >> 16), j2 = R.r[j].y;
          const f3 p0 = ldw(L.a1xy, L.a1z, j0), p1 = ldw(L.a1xy, L.a1z, j1), p2 = ldw(L.a1xy, L.a1z, j2);
          const f3 q0 = ldw(L.a2xy, L.a2z, j0), q1 = ldw(L.a2xy, L.a2z, j1), q2 = ldw(L.a2xy, L.a2z, j2);
          ea0[u] = p1 - p0; ea1[u] = p2 - p0; eb0[u] = q1 - q0; eb1[u] = q2 - q0;
        }
        DC_WIN_GATHER_FENCE
        f3 c1[SUB], c2[SUB];
#pragma unroll
        for (int u = 0; u < SUB; u++) {
          const int j = min(s0 + u, EB - 1);
          f3 r0, r1;
          if constexpr (PRECISE) tri_op.edges(ea0[u], ea1[u], eb0[u], eb1[u], R.D[j], R.Dl[j], __int_as_float(R.r[j].z), r0, r1);
          else tri_op.edges(ea0[u], ea1[u], eb0[u], eb1[u], R.D[j], __int_as_float(R.r[j].z), r0, r1);
          // the triangle's contributions to its corners 1 and 2 (Triangle.cpp: A^T applied to the residual columns; corner 0 = -(c1 + c2))
          c1[u] = r0 * R.D[j].x + r1 * R.D[j].y; c2[u] = r0 * R.D[j].z + r1 * R.D[j].w;
        }
#pragma unroll
        for (int u = 0; u < SUB; u++) {
          if (s0 + u < EB) {
            const int t = t0 + (s0 + u) * THREADS;
            const bool live = t < nt;
            stw(L.erxy, L.erz, live ? t : dump, c1[u]);
            stw(L.erxy, L.erz, live ? nt + t : dump, c2[u]);
          }
        }
      }
    };
    auto bend_compute = [&](auto ebc, const WinBendRecs<MB, PRECISE> &R, int e0) {
      constexpr int EB = decltype(ebc)::value;
      constexpr int SUB = EB < SUBMAX ? EB : SUBMAX;
      if constexpr (SUBMAX == 1 && kWinLegacy1024) {
#pragma unroll
        for (int j = 0; j < EB; j++) {
          const int e = e0 + j * THREADS;
          const int j0 = R.r[j].x & 0xffff, j1 = (int) ((unsigned) R.r[j].x >> 16), j2 = R.r[j].y & 0xffff, j3 = (int) ((unsigned) R.r[j].y >> 16);
          f3 res;
          if constexpr (PRECISE)
            bend_op(ldw(L.a1xy, L.a1z, j0), ldw(L.a1xy, L.a1z, j1), ldw(L.a1xy, L.a1z, j2), ldw(L.a1xy, L.a1z, j3),
                    ldw(L.a2xy, L.a2z, j0), ldw(L.a2xy, L.a2z, j1), ldw(L.a2xy, L.a2z, j2), ldw(L.a2xy, L.a2z, j3), R.w[j], R.wl[j],
                    __int_as_float(R.r[j].z), __int_as_float(R.r[j].w), res);
          else
            bend_op(ldw(L.a1xy, L.a1z, j0), ldw(L.a1xy, L.a1z, j1), ldw(L.a1xy, L.a1z, j2), ldw(L.a1xy, L.a1z, j3),
                    ldw(L.a2xy, L.a2z, j0), ldw(L.a2xy, L.a2z, j1), ldw(L.a2xy, L.a2z, j2), ldw(L.a2xy, L.a2z, j3), R.w[j],
                    __int_as_float(R.r[j].z), __int_as_float(R.r[j].w), res);
          if (e < nb) stw(L.erxy, L.erz, 2 * nt + e, res);
        }
      } else
#pragma unroll
      for (int s0 = 0; s0 < EB; s0 += SUB) {
        f3 ea1[SUB], ea2[SUB], ea3[SUB], eb1[SUB], eb2[SUB], eb3[SUB];
#pragma unroll
        for (int u = 0; u < SUB; u++) {
          const int j = min(s0 + u, EB - 1);
          const int j0 = R.r[j].x & 0xffff, j1 = (int) ((unsigned) R.r[j].x >> 16), j2 = R.r[j].y & 0xffff, j3 = (int) ((unsigned) R.r[j].y >> 16);
          const f3 p0 = ldw(L.a1xy, L.a1z, j0), q0 = ldw(L.a2xy, L.a2z, j0);
          ea1[u] = ldw(L.a1xy, L.a1z, j1) - p0; ea2[u] = ldw(L.a1xy, L.a1z, j2) - p0; ea3[u] = ldw(L.a1xy, L.a1z, j3) - p0;
          eb1[u] = ldw(L.a2xy, L.a2z, j1) - q0; eb2[u] = ldw(L.a2xy, L.a2z, j2) - q0; eb3[u] = ldw(L.a2xy, L.a2z, j3) - q0;
        }
        DC_WIN_GATHER_FENCE
        f3 res[SUB];
#pragma unroll
        for (int u = 0; u < SUB; u++) {
          const int j = min(s0 + u, EB - 1);
          if constexpr (PRECISE)
            bend_op.edges(ea1[u], ea2[u], ea3[u], eb1[u], eb2[u], eb3[u], R.w[j], R.wl[j], __int_as_float(R.r[j].z), __int_as_float(R.r[j].w), res[u]);
          else
            bend_op.edges(ea1[u], ea2[u], ea3[u], eb1[u], eb2[u], eb3[u], R.w[j], __int_as_float(R.r[j].z), __int_as_float(R.r[j].w), res[u]);
        }
#pragma unroll
        for (int u = 0; u < SUB; u++) {
          if (s0 + u < EB) {
            const int e = e0 + (s0 + u) * THREADS;
            stw(L.erxy, L.erz, e < nb ? 2 * nt + e : dump, res[u]);
          }
        }
      }
    };
    // Rounds per WAVE (wave-uniform control flow, no barrier inside): a wave whose lanes are all past the end of the list in the last
    // round skips it — the redirected stores above would otherwise make every wave compute a full round of duplicates there (2 100
    // triangles on 1024 threads: a third round for the sake of 52 lanes of wave 0)
    // (the 1024-thread kernels keep the rounds of the workgroup and their guarded stores: a masked element is an exec-masked block a wave
    // without live lanes branches over. Measured round 5, rounds per wave there: one-workgroup adjoint 11.90 -> 11.79 ms, split adjoint
    // 3.60 -> 4.08 ms per batch step)
    const int wbase = (SUBMAX == 1 && kWinLegacy1024) ? 0 : __builtin_amdgcn_readfirstlane(tid & ~63);
    for (int q = 0, rounds = max(nt - wbase + THREADS - 1, 0) / THREADS; q < rounds;) {
      const int left = rounds - q, t0 = q * THREADS + tid;
      q += batch_dispatch<MB>(left, [&](auto ebc) { tri_load(ebc, tcur, t0); tri_compute(ebc, tcur, t0); });
    }
    WPH(1)
    for (int q = 0, rounds = max(nb - wbase + THREADS - 1, 0) / THREADS; q < rounds;) {
      const int left = rounds - q, e0 = q * THREADS + tid;
      q += batch_dispatch<MB>(left, [&](auto ebc) { bend_load(ebc, bcur, e0); bend_compute(ebc, bcur, e0); });
    }
    __syncthreads();
    WPH(2)
    // per-vertex phase. A triangle's two result vectors are its contributions to its corners 1 and 2 (corner 0: minus their sum), so a
    // triangle entry of a vertex is a position with a sign (16 bits, 8 per packet); a flap entry is a position with the corner's weight
    // (2 per packet). All packets of a vertex travel in one batch of loads (1 + 6 = a valence-6 vertex: 8 triangle entries + 12
    // flaps); clamped index + masked coefficient instead of divergence, wider rows take another round. With fewer threads than owned
    // vertices (512-thread kernels) a thread handles two vertices at once so that the table loads of both rounds overlap.
    auto gather_tri = [&](const int4 &w, bool live, float &sx, float &sy, float &sz) {
      const int wd[4] = {w.x, w.y, w.z, w.w};
      const int one = live ? 0x3f800000 : 0;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int code = (e & 1) ? (int) ((unsigned) wd[e >> 1] >> 16) : (wd[e >> 1] & 0xffff);
        const float c = __int_as_float(one | ((code & 1) << 31));      // +1 / -1 (0 for a masked packet)
        const float2 q = L.erxy[code >> 1];
        const float z = L.erz[code >> 1];
        sx = fmaf(c, q.x, sx); sy = fmaf(c, q.y, sy); sz = fmaf(c, z, sz);
      }
    };
    auto gather = [&](const int4 (&e)[VPB], int s0, int np, float &sx, float &sy, float &sz) {
#pragma unroll
      for (int j = 0; j < VPB; j++) {
        const float on = (s0 + j < np) ? 1.f : 0.f;
        const float2 qa = L.erxy[e[j].x], qb = L.erxy[e[j].z];
        const float za = L.erz[e[j].x], zb = L.erz[e[j].z];
        const float ca = __int_as_float(e[j].y) * on, cb = __int_as_float(e[j].w) * on;
        sx = fmaf(ca, qa.x, sx); sy = fmaf(ca, qa.y, sy); sz = fmaf(ca, za, sz);
        sx = fmaf(cb, qb.x, sx); sy = fmaf(cb, qb.y, sy); sz = fmaf(cb, zb, sz);
      }
    };
    for (int i = v0 + tid; i < v1; i += (PAIR ? 2 : 1) * THREADS) {
      const int cha = __builtin_amdgcn_readfirstlane(i >> 6);     // v0 and THREADS are multiples of 64
      const int na = S.winc_n[cha], nta = na >> 16, nba = na & 0xffff;
      const int4 DC_G *rowa = S.winc + S.winc_ptr[cha] + lane;
      const int4 DC_G *frowa = rowa + nta * 64;
      float ax = 0.f, ay = 0.f, az = 0.f;
      if constexpr (PAIR) {
        const int ib = i + THREADS;
        const bool vb = ib < v1;
        const int chb = __builtin_amdgcn_readfirstlane((vb ? ib : i) >> 6);
        const int nb_ = S.winc_n[chb], ntb = nb_ >> 16, nbb = nb_ & 0xffff;
        const int4 DC_G *rowb = S.winc + S.winc_ptr[chb] + lane;
        const int4 DC_G *frowb = rowb + ntb * 64;
        float bx = 0.f, by = 0.f, bz = 0.f;
        auto prea = [&]() { if constexpr (vert_has_pre<VertOp>::value) return vert_op.pre(i); else return 0; }();
        auto preb = [&]() { if constexpr (vert_has_pre<VertOp>::value) return vert_op.pre(vb ? ib : i); else return 0; }();
        {   // first batch: the triangle packet and the first flap packets of both vertices in flight together
          const int4 ta = rowa[0], tb = rowb[0];
          const int4 ta1 = rowa[min(1, nta - 1) * 64], tb1 = rowb[min(1, ntb - 1) * 64];      // (irregular meshes: up to 16 entries in the first batch)
          int4 ea[VPB], eb[VPB];
#pragma unroll
          for (int j = 0; j < VPB; j++) { ea[j] = frowa[min(j, nba - 1) * 64]; eb[j] = frowb[min(j, nbb - 1) * 64]; }
          gather_tri(ta, true, ax, ay, az);
          gather_tri(tb, true, bx, by, bz);
          if (max(nta, ntb) > 1) {      // wave-uniform
            gather_tri(ta1, nta > 1, ax, ay, az);
            gather_tri(tb1, ntb > 1, bx, by, bz);
          }
          gather(ea, 0, nba, ax, ay, az);
          gather(eb, 0, nbb, bx, by, bz);
        }
        for (int s0 = 2; s0 < max(nta, ntb); s0++) {      // vertices of more than 16 triangle entries
          const int4 ta = rowa[min(s0, nta - 1) * 64], tb = rowb[min(s0, ntb - 1) * 64];
          gather_tri(ta, s0 < nta, ax, ay, az);
          gather_tri(tb, s0 < ntb, bx, by, bz);
        }
        for (int s0 = VPB; s0 < max(nba, nbb); s0 += VPB) {
          int4 ea[VPB], eb[VPB];
#pragma unroll
          for (int j = 0; j < VPB; j++) { ea[j] = frowa[min(s0 + j, nba - 1) * 64]; eb[j] = frowb[min(s0 + j, nbb - 1) * 64]; }
          gather(ea, s0, nba, ax, ay, az);
          gather(eb, s0, nbb, bx, by, bz);
        }
        if constexpr (vert_has_pre<VertOp>::value) {
          vert_op(i, mk(ax, ay, az), ldw(L.a1xy, L.a1z, i - lo), prea);
          if (vb) vert_op(ib, mk(bx, by, bz), ldw(L.a1xy, L.a1z, ib - lo), preb);
        } else {
          vert_op(i, mk(ax, ay, az), ldw(L.a1xy, L.a1z, i - lo));
          if (vb) vert_op(ib, mk(bx, by, bz), ldw(L.a1xy, L.a1z, ib - lo));
        }
      } else {
        auto prea = [&]() { if constexpr (vert_has_pre<VertOp>::value) return vert_op.pre(i); else return 0; }();
        {
          const int4 ta = rowa[0], ta1 = rowa[min(1, nta - 1) * 64];
          int4 ea[VPB];
#pragma unroll
          for (int j = 0; j < VPB; j++) ea[j] = frowa[min(j, nba - 1) * 64];
          gather_tri(ta, true, ax, ay, az);
          if (nta > 1) gather_tri(ta1, true, ax, ay, az);
          gather(ea, 0, nba, ax, ay, az);
        }
        for (int s0 = 2; s0 < nta; s0++) gather_tri(rowa[s0 * 64], true, ax, ay, az);
        for (int s0 = VPB; s0 < nba; s0 += VPB) {
          int4 ea[VPB];
#pragma unroll
          for (int j = 0; j < VPB; j++) ea[j] = frowa[min(s0 + j, nba - 1) * 64];
          gather(ea, s0, nba, ax, ay, az);
        }
        if constexpr (vert_has_pre<VertOp>::value) vert_op(i, mk(ax, ay, az), ldw(L.a1xy, L.a1z, i - lo), prea);
        else vert_op(i, mk(ax, ay, az), ldw(L.a1xy, L.a1z, i - lo));
      }
    }
    WPH(3)
  }
}

template <int THREADS, bool PRECISE = false, class Stage1, class TriOp, class BendOp, class VertOp>
__device__ __forceinline__ void element_windows(const DevSystem &S, float *lds, Stage1 stage1,
                                                const float *__restrict__ in2, TriOp tri_op, BendOp bend_op, VertOp vert_op) {
  element_windows_t<THREADS, PRECISE>(S, 0, S.nwin, lds, stage1, In2Plain{in2, S.N}, tri_op, bend_op, vert_op);
}

// ---- forward local step: a = x_n, b = v (current iterate); x = x_n + h v, edges formed as differences first ----
// stage1 for a plain planar vector
struct StagePlanar {
  const float *v;
  int N;
  __device__ __forceinline__ f3 operator()(int i) const { return mk(v[i], v[N + i], v[2 * N + i]); }
};

struct FwdTriOp {   // Triangle::project (Triangle.cpp:310-351): columns of h w^2 (T - F)
  float h;
  __device__ __forceinline__ void operator()(f3 x0, f3 x1, f3 x2, f3 v0, f3 v1, f3 v2, float4 D, float w2, f3 &r0, f3 &r1) const {
    edges(x1 - x0, x2 - x0, v1 - v0, v2 - v0, D, w2, r0, r1);
  }
  // (a0, a1) = edges of x_n from the first vertex, (b0, b1) = the same of v
  __device__ __forceinline__ void edges(f3 a0, f3 a1, f3 b0, f3 b1, float4 D, float w2, f3 &r0, f3 &r1) const {
    f3 e0 = a0 + b0 * h, e1 = a1 + b1 * h;
    f3 f0 = e0 * D.x + e1 * D.z, f1 = e0 * D.y + e1 * D.w;
    Polar P = polar3x2(f0, f1);
    const float s = h * w2;
    r0 = (P.t0 - f0) * s; r1 = (P.t1 - f1) * s;
  }
};
struct FwdBendOp {  // TriangleBending::project (TriangleBending.cpp:138-151)
  float h;
  __device__ __forceinline__ void operator()(f3 x0, f3 x1, f3 x2, f3 x3, f3 v0, f3 v1, f3 v2, f3 v3, float4 w, float n, float w2, f3 &res) const {
    edges(x1 - x0, x2 - x0, x3 - x0, v1 - v0, v2 - v0, v3 - v0, w, n, w2, res);
  }
  __device__ __forceinline__ void edges(f3 a1, f3 a2, f3 a3, f3 b1, f3 b2, f3 b3, float4 w, float n, float w2, f3 &res) const {
    f3 ev = (a1 + b1 * h) * w.y;
    ev = ev + (a2 + b2 * h) * w.z;
    ev = ev + (a3 + b3 * h) * w.w;
    f3 p = mk(0, 0, 0);
    if (n > 1e-6f) p = normalized_fast(ev) * n;
    res = (p - ev) * (h * w2);
  }
};

// ---- the same two operators with fp64 element math, for the RECORD of a converged step (f, and with it the contact vectors d the
// adjoint differentiates): T - F and p - e are differences of nearly equal quantities (strain 1e-3..1e-2), and the fp32 evaluation
// above carries the 6e-8 roundings of the edges, of inv_deltaUV and of F into them — 2e-5 relative in f on the 10k-vertex cloth, the
// dominant term of the GPU-vs-oracle gradient difference there (measured by substitution, tests/analyze_dump.py). Inputs: the exact
// fp32 edge parts (x_n differences, velocity differences), rest data as fl32 value + low-order part. Runs once per time step.
struct PreciseTriOp {
  double h;
  __device__ __forceinline__ void operator()(f3 x0, f3 x1, f3 x2, f3 v0, f3 v1, f3 v2, float4 D, float4 Dl, float w2, f3 &r0, f3 &r1) const {
    edges(x1 - x0, x2 - x0, v1 - v0, v2 - v0, D, Dl, w2, r0, r1);
  }
  __device__ __forceinline__ void edges(f3 a0, f3 a1, f3 b0, f3 b1, float4 D, float4 Dl, float w2, f3 &r0, f3 &r1) const {
    const double e0x = (double) a0.x + h * (double) b0.x, e0y = (double) a0.y + h * (double) b0.y, e0z = (double) a0.z + h * (double) b0.z;
    const double e1x = (double) a1.x + h * (double) b1.x, e1y = (double) a1.y + h * (double) b1.y, e1z = (double) a1.z + h * (double) b1.z;
    const double Dx = (double) D.x + (double) Dl.x, Dy = (double) D.y + (double) Dl.y, Dz = (double) D.z + (double) Dl.z, Dw = (double) D.w + (double) Dl.w;
    const double f0x = e0x * Dx + e1x * Dz, f0y = e0y * Dx + e1y * Dz, f0z = e0z * Dx + e1z * Dz;
    const double f1x = e0x * Dy + e1x * Dw, f1y = e0y * Dy + e1y * Dw, f1z = e0z * Dy + e1z * Dw;
    // closest isometry T = F S^-1 in closed form (polar3x2, dc_devlib.h), fp64
    const double a = f0x * f0x + f0y * f0y + f0z * f0z, b = f0x * f1x + f0y * f1y + f0z * f1z, c = f1x * f1x + f1y * f1y + f1z * f1z;
    const double s = sqrt(fmax(a * c - b * b, 1e-300)), t = sqrt(a + c + 2.0 * s), inv = 1.0 / (t * s);
    const double i00 = (c + s) * inv, i01 = -b * inv, i11 = (a + s) * inv;
    const double sc = h * (double) w2;
    r0 = mk((float) ((f0x * i00 + f1x * i01 - f0x) * sc), (float) ((f0y * i00 + f1y * i01 - f0y) * sc), (float) ((f0z * i00 + f1z * i01 - f0z) * sc));
    r1 = mk((float) ((f0x * i01 + f1x * i11 - f1x) * sc), (float) ((f0y * i01 + f1y * i11 - f1y) * sc), (float) ((f0z * i01 + f1z * i11 - f1z) * sc));
  }
};
struct PreciseBendOp {   // wl = low-order parts of the cotan weights 1..3 and (.w) of the rest norm
  double h;
  __device__ __forceinline__ void operator()(f3 x0, f3 x1, f3 x2, f3 x3, f3 v0, f3 v1, f3 v2, f3 v3, float4 w, float4 wl, float n, float w2, f3 &res) const {
    edges(x1 - x0, x2 - x0, x3 - x0, v1 - v0, v2 - v0, v3 - v0, w, wl, n, w2, res);
  }
  __device__ __forceinline__ void edges(f3 a1, f3 a2, f3 a3, f3 b1, f3 b2, f3 b3, float4 w, float4 wl, float n, float w2, f3 &res) const {
    const double w1 = (double) w.y + (double) wl.x, w2d = (double) w.z + (double) wl.y, w3 = (double) w.w + (double) wl.z;
    const double ex = ((double) a1.x + h * (double) b1.x) * w1 + ((double) a2.x + h * (double) b2.x) * w2d + ((double) a3.x + h * (double) b3.x) * w3;
    const double ey = ((double) a1.y + h * (double) b1.y) * w1 + ((double) a2.y + h * (double) b2.y) * w2d + ((double) a3.y + h * (double) b3.y) * w3;
    const double ez = ((double) a1.z + h * (double) b1.z) * w1 + ((double) a2.z + h * (double) b2.z) * w2d + ((double) a3.z + h * (double) b3.z) * w3;
    double fac = -1.0;                                    // p = 0: res = -e
    if (n > 1e-6f) {
      const double nn = (double) n + (double) wl.w, len = sqrt(ex * ex + ey * ey + ez * ez);
      fac = len > 0.0 ? nn / len - 1.0 : 0.0;             // p - e = e (n / |e| - 1)
    }
    const double sc = fac * h * (double) w2;
    res = mk((float) (ex * sc), (float) (ey * sc), (float) (ez * sc));
  }
};

// The same accuracy where it matters at a fraction of the fp64 work — for use in EVERY PD iteration: only the deformation gradient F
// and the strain E = F^T F - I (bending: e and |e|^2 - n^2) are formed in fp64; everything after them is a function of the small
// quantity E evaluated in fp32 without cancellation:
//   C = I + E,  s = sqrt(det C),  t = sqrt(tr C + 2 s),  S = C^(1/2) = (C + s I) / t        (2 x 2 closed form)
//   s - 1 = q / (sqrt(1 + q) + 1),  q = tr E + det E;   t - 2 = (tr E + 2 (s - 1)) / (t + 2);   S - I = (E + (s - 1 - (t - 2)) I) / t
//   T - F = F (S^-1 - I) = -F (S - I) S^-1
struct HybridTriOp {
  double h;
  __device__ __forceinline__ void operator()(f3 x0, f3 x1, f3 x2, f3 v0, f3 v1, f3 v2, float4 D, float4 Dl, float w2, f3 &r0, f3 &r1) const {
    edges(x1 - x0, x2 - x0, v1 - v0, v2 - v0, D, Dl, w2, r0, r1);
  }
  __device__ __forceinline__ void edges(f3 a0, f3 a1, f3 b0, f3 b1, float4 D, float4 Dl, float w2, f3 &r0, f3 &r1) const {
    const double e0x = (double) a0.x + h * (double) b0.x, e0y = (double) a0.y + h * (double) b0.y, e0z = (double) a0.z + h * (double) b0.z;
    const double e1x = (double) a1.x + h * (double) b1.x, e1y = (double) a1.y + h * (double) b1.y, e1z = (double) a1.z + h * (double) b1.z;
    const double Dx = (double) D.x + (double) Dl.x, Dy = (double) D.y + (double) Dl.y, Dz = (double) D.z + (double) Dl.z, Dw = (double) D.w + (double) Dl.w;
    const double f0x = e0x * Dx + e1x * Dz, f0y = e0y * Dx + e1y * Dz, f0z = e0z * Dx + e1z * Dz;
    const double f1x = e0x * Dy + e1x * Dw, f1y = e0y * Dy + e1y * Dw, f1z = e0z * Dy + e1z * Dw;
    const float e00 = (float) (f0x * f0x + f0y * f0y + f0z * f0z - 1.0), e01 = (float) (f0x * f1x + f0y * f1y + f0z * f1z),
                e11 = (float) (f1x * f1x + f1y * f1y + f1z * f1z - 1.0);
    const f3 f0 = mk((float) f0x, (float) f0y, (float) f0z), f1 = mk((float) f1x, (float) f1y, (float) f1z);
    const float trE = e00 + e11, q = trE + (e00 * e11 - e01 * e01);
    const float sm1 = q * fast_rcp(fast_sqrt(fmaxf(1.f + q, 1e-30f)) + 1.f);
    const float t = fast_sqrt(fmaxf(4.f + trE + 2.f * sm1, 1e-30f));
    const float tm2 = (trE + 2.f * sm1) * fast_rcp(t + 2.f);
    const float c = sm1 - tm2, it = fast_rcp(t);
    const float a = (e00 + c) * it, b = e01 * it, d = (e11 + c) * it;               // S - I
    const float idet = fast_rcp((1.f + a) * (1.f + d) - b * b);
    const float m00 = (a + (a * d - b * b)) * idet, m01 = b * idet, m11 = (d + (a * d - b * b)) * idet;   // (S - I) S^-1
    const float sc = -(float) h * w2;
    r0 = (f0 * m00 + f1 * m01) * sc; r1 = (f0 * m01 + f1 * m11) * sc;
  }
};
struct HybridBendOp {
  double h;
  __device__ __forceinline__ void operator()(f3 x0, f3 x1, f3 x2, f3 x3, f3 v0, f3 v1, f3 v2, f3 v3, float4 w, float4 wl, float n, float w2, f3 &res) const {
    edges(x1 - x0, x2 - x0, x3 - x0, v1 - v0, v2 - v0, v3 - v0, w, wl, n, w2, res);
  }
  __device__ __forceinline__ void edges(f3 a1, f3 a2, f3 a3, f3 b1, f3 b2, f3 b3, float4 w, float4 wl, float n, float w2, f3 &res) const {
    const double w1 = (double) w.y + (double) wl.x, w2d = (double) w.z + (double) wl.y, w3 = (double) w.w + (double) wl.z;
    const double ex = ((double) a1.x + h * (double) b1.x) * w1 + ((double) a2.x + h * (double) b2.x) * w2d + ((double) a3.x + h * (double) b3.x) * w3;
    const double ey = ((double) a1.y + h * (double) b1.y) * w1 + ((double) a2.y + h * (double) b2.y) * w2d + ((double) a3.y + h * (double) b3.y) * w3;
    const double ez = ((double) a1.z + h * (double) b1.z) * w1 + ((double) a2.z + h * (double) b2.z) * w2d + ((double) a3.z + h * (double) b3.z) * w3;
    float fac = -1.f;                                     // p = 0: res = -e
    if (n > 1e-6f) {
      const double nn = (double) n + (double) wl.w, l2 = ex * ex + ey * ey + ez * ez;
      const float diff = (float) (nn * nn - l2), len = fast_sqrt((float) l2);
      fac = len > 0.f ? diff * fast_rcp(len * ((float) nn + len)) : 0.f;      // n / |e| - 1 = (n^2 - |e|^2) / (|e| (n + |e|))
    }
    const float sc = fac * (float) h * w2;
    res = mk((float) ex * sc, (float) ey * sc, (float) ez * sc);
  }
};

// The element operators of the forward local step, chosen at COMPILE time (a run-time choice inside the PD loop cost 1.5 ms of 23 on
// the 10k-vertex workload through the register allocation of the loop, measured r03d / r03e): fp64 strain by default, -DDC_ELEMENT_OPS=0
// the all-fp32 operators of rounds 1-2, =2 the all-fp64 ones (A/B builds).
#ifndef DC_ELEMENT_OPS
#define DC_ELEMENT_OPS 1
#endif
#if DC_ELEMENT_OPS == 0
constexpr bool kFwdOpsPrecise = false;
__device__ __forceinline__ FwdTriOp fwd_tri_op(float h, double) { return FwdTriOp{h}; }
__device__ __forceinline__ FwdBendOp fwd_bend_op(float h, double) { return FwdBendOp{h}; }
#elif DC_ELEMENT_OPS == 2
constexpr bool kFwdOpsPrecise = true;
__device__ __forceinline__ PreciseTriOp fwd_tri_op(float, double h) { return PreciseTriOp{h}; }
__device__ __forceinline__ PreciseBendOp fwd_bend_op(float, double h) { return PreciseBendOp{h}; }
#else
constexpr bool kFwdOpsPrecise = true;
__device__ __forceinline__ HybridTriOp fwd_tri_op(float, double h) { return HybridTriOp{h}; }
__device__ __forceinline__ HybridBendOp fwd_bend_op(float, double h) { return HybridBendOp{h}; }
#endif

// ---- adjoint: a = y, b = x_new; h^2 w^2 (A - dp/dx)^T A y per element ----
struct AdjTriOp {   // Triangle::projectToManifoldBackward (Triangle.cpp:354-451) in closed form
  float h2;
  __device__ __forceinline__ void operator()(f3 q0, f3 q1, f3 q2, f3 x0, f3 x1, f3 x2, float4 D, float w2, f3 &r0, f3 &r1) const {
    edges(q1 - q0, q2 - q0, x1 - x0, x2 - x0, D, w2, r0, r1);
  }
  // (d0, d1) = edges of y from the first vertex, (e0, e1) = the same of x_new
  __device__ __forceinline__ void edges(f3 d0, f3 d1, f3 e0, f3 e1, float4 D, float w2, f3 &r0, f3 &r1) const {
    Polar P = polar3x2(e0 * D.x + e1 * D.z, e0 * D.y + e1 * D.w);
    f3 y0 = d0 * D.x + d1 * D.z, y1 = d0 * D.y + d1 * D.w;
    const float c = (dot(P.t1, y0) - dot(P.t0, y1)) * fast_rcp(P.trS);
    f3 z0 = y0 * P.i00 + y1 * P.i01, z1 = y0 * P.i01 + y1 * P.i11;
    z0 = z0 - P.t0 * dot(P.t0, z0) - P.t1 * dot(P.t1, z0);
    z1 = z1 - P.t0 * dot(P.t0, z1) - P.t1 * dot(P.t1, z1);
    const float s = h2 * w2;
    r0 = (y0 - (P.t1 * c + z0)) * s; r1 = (y1 - (z1 - P.t0 * c)) * s;
  }
};
struct AdjBendOp {  // TriangleBending::backwardGradient (TriangleBending.cpp:154-172)
  float h2;
  __device__ __forceinline__ void operator()(f3 q0, f3 q1, f3 q2, f3 q3, f3 x0, f3 x1, f3 x2, f3 x3, float4 w, float n, float w2, f3 &res) const {
    edges(q1 - q0, q2 - q0, q3 - q0, x1 - x0, x2 - x0, x3 - x0, w, n, w2, res);
  }
  __device__ __forceinline__ void edges(f3 d1, f3 d2, f3 d3, f3 e1, f3 e2, f3 e3, float4 w, float n, float w2, f3 &res) const {
    f3 ey = d1 * w.y + d2 * w.z + d3 * w.w;
    res = ey;
    if (n > 1e-6f) {
      f3 ev = e1 * w.y + e2 * w.z + e3 * w.w;
      const float ien = fast_rsqrt(dot(ev, ev));
      f3 eh = ev * ien;
      res = ey - (ey - eh * dot(eh, ey)) * (n * ien);
    }
    res = res * (h2 * w2);
  }
};

}  // namespace dc
