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
// Per-step hand-overs of whole arrays (tape state, self-contact lists) use plain stores bracketed by agent-scope
// release / acquire fences around an exchange (xch_fence_barrier).
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
  int xch_stride;     // granules per (part, parity): 1 + 2 HB
  // element windows of size R / wpp (same member names as DevSystem's set: dc_winlib.h is generic over both)
  const int4 DC_C *win;
  const int4 DC_G *wtri_rec;
  const float4 DC_G *wtri_D;
  const int4 DC_G *wbend_rec;
  const float4 DC_G *wbend_w;
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
  const DevCluster *self_dev;
};

constexpr long long kSpinLimit = 200000000ll;     // 2 s of the 100 MHz wall clock

// ---- sc1 (write-through / L1-bypassing) access to a planar [3][N] vector of one rollout through a buffer resource ----
struct BufVec {
  __amdgpu_buffer_rsrc_t rs;
  int N;
  __device__ __forceinline__ float ld(int idx) const { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4, 0, 16)); }
  __device__ __forceinline__ void st(int idx, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rs, idx * 4, 0, 16); }
};
__device__ __forceinline__ BufVec buf_vec(const float *p, int N) {
  BufVec b;
  b.rs = __builtin_amdgcn_make_buffer_rsrc((void *) p, 0, 3 * N * 4, 0x00020000);
  b.N = N;
  return b;
}
__device__ __forceinline__ float ldc1(const BufVec &b, int idx) {
  return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(b.rs, idx * 4, 0, 16));
}
__device__ __forceinline__ void stc1(const BufVec &b, int idx, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), b.rs, idx * 4, 0, 16);
}
__device__ __forceinline__ f3 ld3c(const BufVec &b, int i) { return mk(ldc1(b, i), ldc1(b, b.N + i), ldc1(b, 2 * b.N + i)); }
__device__ __forceinline__ void st3c(const BufVec &b, int i, f3 v) { stc1(b, i, v.x); stc1(b, b.N + i, v.y); stc1(b, 2 * b.N + i, v.z); }
struct In2Sc1 {       // input loader of element_windows_t (stage1 / in2) for a vector other workgroups write (dc_winlib.h)
  BufVec b;
  __device__ __forceinline__ f3 operator()(int i) const { return ld3c(b, i); }
};

// ---- exchange state of one workgroup ----
struct Xch {
  __amdgpu_buffer_rsrc_t rs;    // the rollout's exchange area
  unsigned seq;                 // sequence number of the current exchange (tag); starts at 0 = "nothing yet"
  int part, K, HB, stride;
  int site;                     // diagnostic: which exchange of the kernel is running (recorded when a poll gives up)
  float *lsum;                  // LDS [4 * 8]: partial sums of all parts of the current exchange
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
  X.seq = 0; X.site = 0; X.part = part; X.K = CL.K; X.HB = CL.HB; X.stride = CL.xch_stride;
  X.lsum = lds_tail; X.ldead = (int *) (lds_tail + 32); X.err = CL.err;
  if (threadIdx.x == 0) *X.ldead = 0;
  return X;
}
constexpr int kXchLdsFloats = 48;      // tail of the dynamic LDS the exchange uses (lsum[32], ldead, padding)

__device__ __forceinline__ int xch_off(const Xch &X, int part, int g) { return ((part * 2 + (int) (X.seq & 1u)) * X.stride + g) * 16; }

// start the next exchange (all threads, uniformly)
__device__ __forceinline__ void xch_begin(Xch &X) { X.seq++; }
// this part's partial sums (ONE thread)
__device__ __forceinline__ void xch_publish_sums(const Xch &X, float a, float b, float c) {
  v4i g = {__float_as_int(a), __float_as_int(b), __float_as_int(c), (int) X.seq};
  __builtin_amdgcn_raw_buffer_store_b128(g, X.rs, xch_off(X, X.part, 0), 0, 16);
}
// one boundary row: slot in [0, HB) = this part's first HB rows (read by part - 1), [HB, 2 HB) = its last HB rows (part + 1)
__device__ __forceinline__ void xch_publish_row(const Xch &X, int slot, float x, float y, float z) {
  v4i g = {__float_as_int(x), __float_as_int(y), __float_as_int(z), (int) X.seq};
  __builtin_amdgcn_raw_buffer_store_b128(g, X.rs, xch_off(X, X.part, 1 + slot), 0, 16);
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
      else if (now - t0 > kSpinLimit) {
        // the first poll to give up leaves a record: [1] sequence number waited for, [2] tag seen, [3] site | part << 8 | granule offset << 12
        if (atomicCAS(X.err, 0u, 1u) == 0u) { X.err[1] = X.seq; X.err[2] = (unsigned) g.w; X.err[3] = (unsigned) X.site | ((unsigned) X.part << 8) | ((unsigned) (off / 16) << 12); }
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// Wait for the partial sums of all parts and (HALO) for the neighbours' boundary rows. hv[q] receives halo row j = tid + q THREADS
// (j in [0, HB): rows r0 - HB + j from part - 1; j in [HB, 2 HB): rows r0 + R + (j - HB) from part + 1; zero where there is no
// neighbour). sums[c] = sum over the parts in part order (identical in every part). Call with all threads; ends with a barrier.
// Returns false when the exchange timed out (the caller leaves the kernel).
template <int THREADS, int HPT, bool HALO>
__device__ __forceinline__ bool xch_consume(const Xch &X, double (&sums)[3], f3 (&hv)[HPT]) {
  const int tid = threadIdx.x;
  bool ok = true;
  if (tid < X.K) {
    v4i g;
    ok = xch_poll(X, xch_off(X, tid, 0), g);
    X.lsum[4 * tid] = __int_as_float(g.x); X.lsum[4 * tid + 1] = __int_as_float(g.y); X.lsum[4 * tid + 2] = __int_as_float(g.z);
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
          ok = xch_poll(X, xch_off(X, src, 1 + (lower ? X.HB + j : j - X.HB)), g) && ok;
          hv[q] = mk(__int_as_float(g.x), __int_as_float(g.y), __int_as_float(g.z));
        }
      }
    }
  }
  if (!ok) *X.ldead = 1;
  __syncthreads();
  sums[0] = sums[1] = sums[2] = 0;
  for (int p = 0; p < X.K; p++) { sums[0] += (double) X.lsum[4 * p]; sums[1] += (double) X.lsum[4 * p + 1]; sums[2] += (double) X.lsum[4 * p + 2]; }
  const bool alive = *X.ldead == 0;
  __syncthreads();            // lsum is rewritten by the next exchange
  return alive;
}

// sums only: every thread passes the workgroup's partial sums (already reduced over the workgroup)
template <int THREADS>
__device__ __forceinline__ bool xch_sums(Xch &X, double (&s)[3]) {
  xch_begin(X);
  if (threadIdx.x == 0) xch_publish_sums(X, (float) s[0], (float) s[1], (float) s[2]);
  f3 none[1];
  return xch_consume<THREADS, 1, false>(X, s, none);
}

// every store of the workgroup so far has left the CU (needed before a granule that tells others "my sc1 stores are done")
__device__ __forceinline__ void xch_drain() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Hand-over of arrays written with PLAIN stores: agent-scope release, an all-parts exchange, agent-scope acquire (one lane each;
// the acquire drops this CU's L1 and scalar cache so that plain / scalar loads see the other parts' data).
template <int THREADS>
__device__ __forceinline__ bool xch_fence_barrier(Xch &X) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // restated where the compiler cannot drop it (visibility guide, hazard 12)
  }
  double z[3] = {0, 0, 0};
  const bool ok = xch_sums<THREADS>(X, z);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __builtin_amdgcn_s_dcache_inv();
  }
  __syncthreads();
  return ok;
}

}  // namespace dc
