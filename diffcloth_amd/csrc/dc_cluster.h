// Split ("cluster") execution of one rollout by K workgroups: device-side descriptor and the inter-workgroup exchange.
//
// With fewer rollouts than CUs (BASELINE C4 on 8 GPUs: 32 per GPU; hatController.py: 20) one workgroup per rollout leaves most
// of the chip idle, and a 160 KB LDS bounds the mesh one workgroup can hold. Here part p of a rollout owns the vertex rows
// [p R, (p + 1) R) and runs the same algorithm on them; what crosses parts is
//   * the boundary rows of the vector an operator is applied to (the matrix / element reach is HB rows), and
//   * the partial sums of the dot products / norms,
// both carried by ONE kind of message: 16-byte granules {x, y, z, tag} written with write-through (sc1) stores and polled with
// sc1 loads until the tag equals the exchange's sequence number (MI355X_MICROARCH.md "inter-workgroup visibility", form R2: the
// data is the flag — no fence, no separate flag, placement independent). Every exchange waits for the partial-sum granule of
// EVERY part, so a part can never be more than one exchange ahead of another: two buffers (sequence parity) suffice.
// Whole arrays that cross parts (tape state, records, the adjoint's y) go through the same sc1 path (BufVec): plain stores with
// agent-scope release / acquire fences around an exchange were measured NOT to be sufficient on this part (a part read a stale
// self-contact count), so a count travels inside an exchange and xch_fence_barrier only remains in front of the inlined
// self-collision detection, whose plain loads read positions another part wrote a whole time step earlier.
// Every spin is bounded (kSpinLimit of the 100 MHz wall clock): a part that gives up raises DevCluster::err and all parts of
// the rollout leave the kernel; the host reports DC_ERR_HIP. All parts of a launch are resident by construction (the launcher
// never starts more workgroups than the device has CUs).
#pragma once
#include "dc_devlib.h"

namespace dc {

typedef int v4i __attribute__((ext_vector_type(4)));

struct DevCluster {
  int K;              // workgroups (parts) per rollout
  int R;              // vertex rows per part (multiple of 64); part p owns [p R, min(N, (p + 1) R)), rows >= N are padding
  int HB;             // boundary rows exchanged each side (multiple of 64, >= bandwidth of P, <= R)
  int wpp;            // element windows per part
  int nb;             // rollouts per launch (K nb <= CUs)
  int xch_stride;     // granules per (part, parity): kXchWaves + 2 HB
  // element windows of size R / wpp (same member names as DevSystem's set: dc_winlib.h is generic over both)
  const int4 DC_C *win;
  const int4 DC_G *wtri_rec;
  const float4 DC_G *wtri_D;
  const int4 DC_G *wbend_rec;
  const float4 DC_G *wbend_w;
  const float4 DC_G *wtri_Dlo;
  const float4 DC_G *wbend_lo;
  const int4 DC_G *winc;
  const int DC_C *winc_ptr;
  const int DC_C *winc_n;
  int nwin, win_vcap, win_nrcap, win_lds_bytes;
  // packet matrix padded to K R rows (layout: dc_packets.h)
  const int4 DC_G *pk;
  const int DC_C *pk_ptr;
  const int DC_C *pk_n;
  const float DC_G *sq_dinv;   // [K R]
  int pk_vpt, pad0;            // rows per thread of the 512-thread forward kernel = ceil(R / 512)
  v4i *xch;                    // [nb][K][2][xch_stride] granules, zeroed before every launch
  unsigned *err;               // [4] sticky: [0] != 0 -> an exchange timed out
  long long spin_limit;        // bound of every spin in ticks of the 100 MHz wall clock (kSpinLimit; tests shorten it: DC_TEST_SPIN_MS)
  int redundant_self;          // forward: every part evaluates the layered self friction itself when the rollout's parts share an XCD (DC_SELF_REDUNDANT=0: part 0 alone, rounds 2-5)
  int test_drop;               // test hook (DC_TEST_DROP_PART=1): the last part of the launch's first rollout leaves at once — its peers must time out cleanly
  const DevCluster *self_dev;
};

constexpr long long kSpinLimit = 200000000ll;     // 2 s of the 100 MHz wall clock

// ---- sc1 (write-through / L1-bypassing) access to a planar [3][N] vector of one rollout through a buffer resource ----
// Loads always bypass the L1 (sc1: served by the XCD's L2, or by memory when the line is not there). Stores are write-through
// (sc1: the line goes to memory and leaves the L2) unless `same_xcd`: when every part of the rollout was found on ONE XCD
// (xch_hello), that XCD's L2 is the coherence point and an ordinary store, which keeps the line in the L2, is both sufficient and
// faster (the readers' loads then hit the L2 instead of going to memory).
struct BufVec {
  __amdgpu_buffer_rsrc_t rs;
  int N;
  bool same_xcd;
  __device__ __forceinline__ float ld(int idx) const { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4, 0, 16)); }
  __device__ __forceinline__ void st(int idx, float v) const {
    if (same_xcd) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rs, idx * 4, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rs, idx * 4, 0, 16);
  }
};
__device__ __forceinline__ BufVec buf_vec(const float *p, int N, bool same_xcd) {
  BufVec b;
  b.rs = __builtin_amdgcn_make_buffer_rsrc((void *) p, 0, 3 * N * 4, 0x00020000);
  b.N = N; b.same_xcd = same_xcd;
  return b;
}
__device__ __forceinline__ float ldc1(const BufVec &b, int idx) { return b.ld(idx); }
__device__ __forceinline__ void stc1(const BufVec &b, int idx, float v) { b.st(idx, v); }
__device__ __forceinline__ f3 ld3c(const BufVec &b, int i) { return mk(ldc1(b, i), ldc1(b, b.N + i), ldc1(b, 2 * b.N + i)); }
__device__ __forceinline__ void st3c(const BufVec &b, int i, f3 v) { stc1(b, i, v.x); stc1(b, b.N + i, v.y); stc1(b, 2 * b.N + i, v.z); }
struct In2Sc1 {       // input loader of element_windows_t (stage1 / in2) for a vector other workgroups write (dc_winlib.h)
  BufVec b;
  __device__ __forceinline__ f3 operator()(int i) const { return ld3c(b, i); }
};

// ---- exchange state of one workgroup ----
constexpr int kXchWaves = 16;          // sum granules per (part, parity): one per wave of the publishing workgroup (<= 1024 threads)
struct Xch {
  __amdgpu_buffer_rsrc_t rs;    // the rollout's exchange area
  unsigned seq;                 // sequence number of the current exchange (tag); starts at 0 = "nothing yet"
  int part, K, HB, stride;
  int site;                     // diagnostic: which exchange of the kernel is running (recorded when a poll gives up)
  bool same_xcd;                // every part of this rollout runs on one XCD (xch_hello): granules may stay in its L2
  long long limit;              // spin bound (DevCluster::spin_limit)
  float *lsum;                  // LDS [2][4]: the totals of the current exchange, double-buffered by sequence parity
  int *ldead;                   // LDS flag: an exchange of this workgroup timed out
  unsigned *err;
};

// block -> (rollout of this launch, part). Observed placement: block b runs on XCD b % 8; the parts of one rollout are put on
// one XCD when the launch has a multiple of 8 rollouts (speed only: the protocol does not depend on placement).
__device__ __forceinline__ void cluster_map(int K, int &lb, int &part) {
  const int blk = blockIdx.x, nb = gridDim.x / K;
  if ((nb & 7) == 0) { const int x = blk & 7, j = blk >> 3; lb = (j / K) * 8 + x; part = j % K; }
  else { lb = blk / K; part = blk % K; }
}

__device__ __forceinline__ Xch xch_init(const DevCluster &CL, int lb, int part, float *lds_tail) {
  Xch X;
  const size_t per = (size_t) CL.K * 2 * CL.xch_stride;
  X.rs = __builtin_amdgcn_make_buffer_rsrc((void *) (CL.xch + (size_t) lb * per), 0, (int) (per * 16), 0x00020000);
  X.seq = 0; X.site = 0; X.same_xcd = false; X.part = part; X.K = CL.K; X.HB = CL.HB; X.stride = CL.xch_stride;
  X.lsum = lds_tail; X.ldead = (int *) (lds_tail + 8); X.err = CL.err; X.limit = CL.spin_limit;
  if (threadIdx.x == 0) *X.ldead = 0;
  return X;
}
constexpr int kXchLdsFloats = 32;      // tail of the dynamic LDS the exchange uses (lsum[2][4], ldead, padding; [16, 32): the six-sum totals, two parities of 8)

__device__ __forceinline__ int xch_off(const Xch &X, int part, int g) { return ((part * 2 + (int) (X.seq & 1u)) * X.stride + g) * 16; }

// Wave-wide sum through DPP row operations (VALU only); the total is returned in every lane.
__device__ __forceinline__ float xch_wave_sum(float v) {
#define DC_DPP(x, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xF, true))
  v += DC_DPP(v, 0xB1, 0xF);     // quad_perm [1,0,3,2]
  v += DC_DPP(v, 0x4E, 0xF);     // quad_perm [2,3,0,1]
  v += DC_DPP(v, 0x141, 0xF);    // row_half_mirror
  v += DC_DPP(v, 0x140, 0xF);    // row_mirror
  v += DC_DPP(v, 0x142, 0xA);    // row_bcast:15
  v += DC_DPP(v, 0x143, 0xC);    // row_bcast:31
#undef DC_DPP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// start the next exchange (all threads, uniformly)
__device__ __forceinline__ void xch_begin(Xch &X) { X.seq++; }
__device__ __forceinline__ void xch_store(const Xch &X, v4i g, int off) {
  if (X.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(g, X.rs, off, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(g, X.rs, off, 0, 16);
}
// Every thread hands in its partial sums; each wave reduces them (DPP) and its first lane publishes ONE granule — no workgroup
// barrier, no LDS: a wave's contribution is on its way as soon as that wave is done.
__device__ __forceinline__ void xch_publish_sums(const Xch &X, float a, float b, float c) {
  a = xch_wave_sum(a); b = xch_wave_sum(b); c = xch_wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    v4i g = {__float_as_int(a), __float_as_int(b), __float_as_int(c), (int) X.seq};
    xch_store(X, g, xch_off(X, X.part, (int) (threadIdx.x >> 6)));
  }
}
// one boundary row: slot in [0, HB) = this part's first HB rows (read by part - 1), [HB, 2 HB) = its last HB rows (part + 1)
__device__ __forceinline__ void xch_publish_row(const Xch &X, int slot, float x, float y, float z) {
  v4i g = {__float_as_int(x), __float_as_int(y), __float_as_int(z), (int) X.seq};
  xch_store(X, g, xch_off(X, X.part, kXchWaves + slot));
}
// publish the boundary rows held by this thread: local row l = tid + k THREADS of a part of R rows, value v
__device__ __forceinline__ void xch_publish_boundary(const Xch &X, int l, int R, float x, float y, float z) {
  if (l < X.HB) xch_publish_row(X, l, x, y, z);
  if (l >= R - X.HB && l < R) xch_publish_row(X, X.HB + l - (R - X.HB), x, y, z);
}

__device__ __forceinline__ bool xch_poll(const Xch &X, int off, v4i &g) {
  long long t0 = 0;
  for (unsigned spins = 0;; spins++) {
    g = __builtin_amdgcn_raw_buffer_load_b128(X.rs, off, 0, 16);
    if ((unsigned) g.w == X.seq) return true;
    if ((spins & 255u) == 255u) {
      const long long now = (long long) __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      else if (__hip_atomic_load(X.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      else if (now - t0 > X.limit) {
        // the first poll to give up leaves a record: [1] sequence number waited for, [2] tag seen, [3] site | part << 8 | granule offset << 12
        if (atomicCAS(X.err, 0u, 1u) == 0u) { X.err[1] = X.seq; X.err[2] = (unsigned) g.w; X.err[3] = (unsigned) X.site | ((unsigned) X.part << 8) | ((unsigned) (off / 16) << 12); }
        return false;
      }
    }
    if (spins > 16u) __builtin_amdgcn_s_sleep(1);
  }
}

// Wait for the wave sums of all parts (wave 0 polls them, one or two granules per lane, adds them up in a fixed order and leaves
// the totals in LDS) and (HALO) for the neighbours' boundary rows. hv[q] receives halo row j = tid + q THREADS (j in [0, HB): rows
// r0 - HB + j from part - 1; j in [HB, 2 HB): rows r0 + R + (j - HB) from part + 1; zero where there is no neighbour). sums[c] are
// identical in every part (same granules, same order). Call with all threads; contains ONE workgroup barrier (the LDS totals are
// double-buffered by sequence parity). Returns false when the exchange timed out (the caller leaves the kernel).
template <int THREADS, int HPT, bool HALO>
__device__ __forceinline__ bool xch_finish(const Xch &X, double (&sums)[3], f3 (&hv)[HPT]) {
  constexpr int NW = THREADS / 64;
  const int tid = threadIdx.x;
  bool ok = true;
  float *slot = X.lsum + 4 * (int) (X.seq & 1u);
  if (tid < 64) {
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int j = tid + 64 * q;                // granule j = wave (j % NW) of part (j / NW)
      if (j < X.K * NW) {
        v4i g;
        ok = xch_poll(X, xch_off(X, j / NW, j % NW), g) && ok;
        a += __int_as_float(g.x); b += __int_as_float(g.y); c += __int_as_float(g.z);
      }
    }
    a = xch_wave_sum(a); b = xch_wave_sum(b); c = xch_wave_sum(c);
    if (tid == 0) { slot[0] = a; slot[1] = b; slot[2] = c; }
  }
  if constexpr (HALO) {
#pragma unroll
    for (int q = 0; q < HPT; q++) {
      const int j = tid + q * THREADS;
      hv[q] = mk(0, 0, 0);
      if (j < 2 * X.HB) {
        const bool lower = j < X.HB;
        const int src = lower ? X.part - 1 : X.part + 1;
        if (src >= 0 && src < X.K) {
          v4i g;
          ok = xch_poll(X, xch_off(X, src, kXchWaves + (lower ? X.HB + j : j - X.HB)), g) && ok;
          hv[q] = mk(__int_as_float(g.x), __int_as_float(g.y), __int_as_float(g.z));
        }
      }
    }
  }
  if (!ok) *X.ldead = 1;
  __syncthreads();
  sums[0] = (double) slot[0]; sums[1] = (double) slot[1]; sums[2] = (double) slot[2];
  return *X.ldead == 0;
}

// ---- six sums in one exchange: every wave publishes TWO granules (its slot w and slot NW + w; kXchWaves >= 2 NW) ----
// Used by the single-exchange CG of the split forward kernel (dc_forward_cl_kernel.h): p.Ap, p.r, r.Ap, Ap.Ap and r.r of an iteration
// travel together with the boundary rows of A p.
__device__ __forceinline__ void xch_publish_sums6(const Xch &X, int nw, float a, float b, float c, float d, float e, float f) {
  a = xch_wave_sum(a); b = xch_wave_sum(b); c = xch_wave_sum(c); d = xch_wave_sum(d); e = xch_wave_sum(e); f = xch_wave_sum(f);
  if ((threadIdx.x & 63) == 0) {
    const int w = (int) (threadIdx.x >> 6);
    v4i g0 = {__float_as_int(a), __float_as_int(b), __float_as_int(c), (int) X.seq};
    v4i g1 = {__float_as_int(d), __float_as_int(e), __float_as_int(f), (int) X.seq};
    xch_store(X, g0, xch_off(X, X.part, w));
    xch_store(X, g1, xch_off(X, X.part, nw + w));
  }
}
// the matching wait: waves 0 and 1 poll the two granules of every (part, wave) pair, add them up in a fixed order (bitwise the same
// totals on every part) and leave six totals in LDS; all waves then poll the halo rows. ONE workgroup barrier. lsum6 = 12 floats of LDS (two parities) behind the regular totals.
template <int THREADS, int HPT, bool HALO>
__device__ __forceinline__ bool xch_finish6(const Xch &X, float *lsum6, double (&sums)[6], f3 (&hv)[HPT]) {
  constexpr int NW = THREADS / 64;
  static_assert(2 * NW <= kXchWaves, "two sum granules per wave");
  const int tid = threadIdx.x;
  bool ok = true;
  float *slot = lsum6 + 8 * (int) (X.seq & 1u);
  if (tid < 128) {
    // wave 0 polls the first triple of every (part, wave) pair, wave 1 the second: one poll per lane in flight (K NW <= 64), as in xch_finish
    const int half = tid >> 6, ln = tid & 63;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int j = ln + 64 * q;                // (part, wave) pair j: part j / NW, wave j % NW
      if (j < X.K * NW) {
        v4i g;
        ok = xch_poll(X, xch_off(X, j / NW, half * NW + j % NW), g) && ok;
        a += __int_as_float(g.x); b += __int_as_float(g.y); c += __int_as_float(g.z);
      }
    }
    a = xch_wave_sum(a); b = xch_wave_sum(b); c = xch_wave_sum(c);
    if (ln == 0) { slot[3 * half] = a; slot[3 * half + 1] = b; slot[3 * half + 2] = c; }
  }
  if constexpr (HALO) {
#pragma unroll
    for (int q = 0; q < HPT; q++) {
      const int j = tid + q * THREADS;
      hv[q] = mk(0, 0, 0);
      if (j < 2 * X.HB) {
        const bool lower = j < X.HB;
        const int src = lower ? X.part - 1 : X.part + 1;
        if (src >= 0 && src < X.K) {
          v4i g;
          ok = xch_poll(X, xch_off(X, src, kXchWaves + (lower ? X.HB + j : j - X.HB)), g) && ok;
          hv[q] = mk(__int_as_float(g.x), __int_as_float(g.y), __int_as_float(g.z));
        }
      }
    }
  }
  if (!ok) *X.ldead = 1;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 6; k++) sums[k] = (double) slot[k];
  return *X.ldead == 0;
}

// all-parts sum of three per-thread partial values (a complete exchange)
template <int THREADS>
__device__ __forceinline__ bool xch_allsum(Xch &X, float a, float b, float c, double (&s)[3]) {
  xch_begin(X);
  xch_publish_sums(X, a, b, c);
  f3 none[1];
  return xch_finish<THREADS, 1, false>(X, s, none);
}

// all-parts sum of one fp64 value (its 64 bits in two words of the granule) and one fp32 value per thread: every wave reduces in fp64 and
// publishes {lo word, hi word, c, tag}; wave 0 of the consumer adds the granules up in fp64. Same protocol and barrier count as xch_allsum.
template <int THREADS>
__device__ __forceinline__ bool xch_allsum_d(Xch &X, double a, float c, double &sa, double &sc) {
  constexpr int NW = THREADS / 64;
  xch_begin(X);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  c = xch_wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    v4i g = {__double2loint(a), __double2hiint(a), __float_as_int(c), (int) X.seq};      // the 64 bits of the double: full range and precision
    xch_store(X, g, xch_off(X, X.part, (int) (threadIdx.x >> 6)));
  }
  bool ok = true;
  double *slot = (double *) (X.lsum + 4 * (int) (X.seq & 1u));      // 16 bytes per parity: the fp64 total; the fp32 one goes after both
  float *cslot = X.lsum + 10 + (int) (X.seq & 1u);
  if (threadIdx.x < 64) {
    double da = 0;
    float fc = 0.f;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int j = threadIdx.x + 64 * q;
      if (j < X.K * NW) {
        v4i g;
        ok = xch_poll(X, xch_off(X, j / NW, j % NW), g) && ok;
        da += __hiloint2double(g.y, g.x);
        fc += __int_as_float(g.z);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) da += __shfl_down(da, o, 64);
    fc = xch_wave_sum(fc);
    if (threadIdx.x == 0) { slot[0] = da; *cslot = fc; }
  }
  if (!ok) *X.ldead = 1;
  __syncthreads();
  sa = slot[0]; sc = (double) *cslot;
  return *X.ldead == 0;
}

// First exchange of a kernel (write-through stores): the parts tell each other which XCD they run on (HW_REG_XCC_ID). When all
// K are on the same one, its L2 is their coherence point and the stores of every later exchange may stay there.
template <int THREADS>
__device__ __forceinline__ bool xch_hello(Xch &X) {
  const float id = (float) (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15);
  const bool one = threadIdx.x == 0;
  double s[3];
  if (!xch_allsum<THREADS>(X, one ? id : 0.f, one ? id * id : 0.f, 0.f, s)) return false;
  X.same_xcd = (s[0] * s[0] == (double) X.K * s[1]);       // sum^2 == K * sum of squares  <=>  all equal
  return true;
}

// ---- L1-bypassing (sc1) loads of small shared tables another part wrote (the self-contact lists of part 0's detection) ----
// Valid as a visibility path only when all parts of the rollout run on ONE XCD (Xch::same_xcd): the writer's plain stores have reached that
// XCD's L2 once its vmcnt has drained, and an sc1 load is served by that L2, never by this CU's (possibly stale) L1.
struct Sc1Table {
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ int ldi(int idx) const { return __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4, 0, 16); }
  __device__ __forceinline__ float4 ld4(int idx) const {
    const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 16, 0, 16);
    return make_float4(__int_as_float(q.x), __int_as_float(q.y), __int_as_float(q.z), __int_as_float(q.w));
  }
};
__device__ __forceinline__ Sc1Table sc1_table(const void *p, size_t bytes) {
  Sc1Table t;
  t.rs = __builtin_amdgcn_make_buffer_rsrc((void *) p, 0, (int) bytes, 0x00020000);
  return t;
}

// every store of the workgroup so far has left the CU (needed before granules that tell others "my sc1 stores are done")
__device__ __forceinline__ void xch_drain() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Barrier over the parts of the rollout for data written with write-through (sc1) stores and read with sc1 loads: drain, exchange.
template <int THREADS>
__device__ __forceinline__ bool xch_barrier(Xch &X) {
  xch_drain();
  double z[3];
  return xch_allsum<THREADS>(X, 0.f, 0.f, 0.f, z);
}

// The same for arrays written with PLAIN stores (the self-contact lists of the inlined detection): agent-scope release before,
// agent-scope acquire after (one lane each; the acquire drops this CU's L1 and scalar cache).
template <int THREADS>
__device__ __forceinline__ bool xch_fence_barrier(Xch &X) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // restated where the compiler cannot drop it (visibility guide, hazard 12)
  }
  __syncthreads();                                        // the release precedes every wave's granule
  double z[3];
  const bool ok = xch_allsum<THREADS>(X, 0.f, 0.f, 0.f, z);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __builtin_amdgcn_s_dcache_inv();
  }
  __syncthreads();
  return ok;
}

}  // namespace dc
