// Building blocks of the packet-ELL resident PCG shared by dc_forward_pk.hip (one workgroup per rollout) and
// dc_forward_cl.hip (a rollout split over several workgroups): DPP wave reduction, packet batches, the row consumer.
#pragma once
#include <type_traits>
#include "dc_devlib.h"

namespace dc {
namespace {

constexpr int PB = 4;   // packets per batch; rows are stored padded to a multiple of PB packets

// Wave-wide sum through DPP row operations (VALU only) instead of six ds_bpermute round trips: quad swaps, row mirrors,
// then the row_bcast:15 / row_bcast:31 steps that carry the row totals across the wave; the total lands in lane 63.
__device__ __forceinline__ float wave_sum_f(float v) {
  auto dpp = [](float x, auto ctrl, auto row_mask) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(row_mask)::value, 0xF, true));
  };
  using std::integral_constant;
  v += dpp(v, integral_constant<int, 0xB1>(), integral_constant<int, 0xF>());    // quad_perm [1,0,3,2]
  v += dpp(v, integral_constant<int, 0x4E>(), integral_constant<int, 0xF>());    // quad_perm [2,3,0,1]
  v += dpp(v, integral_constant<int, 0x141>(), integral_constant<int, 0xF>());   // row_half_mirror
  v += dpp(v, integral_constant<int, 0x140>(), integral_constant<int, 0xF>());   // row_mirror: every lane holds its row's sum
  v += dpp(v, integral_constant<int, 0x142>(), integral_constant<int, 0xA>());   // row_bcast:15 -> rows 1 and 3 add rows 0 and 2
  v += dpp(v, integral_constant<int, 0x143>(), integral_constant<int, 0xC>());   // row_bcast:31 -> rows 2, 3 add the lower half
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// fp32 inside a wave, fp64 across the waves (the CG scalars only steer the iteration; the fixed point of the PD
// loop does not depend on them)
template <int THREADS>
__device__ __forceinline__ double block_sum_f(float v, double *red) {
  v = wave_sum_f(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = (double) v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; k++) s += red[k];
  return s;
}

__device__ __forceinline__ void load_batch(int4 (&e)[PB], const int4 *__restrict__ row, int s0) {
#pragma unroll
  for (int j = 0; j < PB; j++) e[j] = row[(s0 + j) * 64];
}

__device__ __forceinline__ void consume_p(const int4 (&e)[PB], const float2 *lxy, const float *lz, int base, float &ax, float &ay, float &az) {
#pragma unroll
  for (int j = 0; j < PB; j++) {
    const int c0 = base + (e[j].w & 1023), c1 = base + ((e[j].w >> 10) & 1023), c2 = base + ((e[j].w >> 20) & 1023);
    const float a0 = __int_as_float(e[j].x), a1 = __int_as_float(e[j].y), a2 = __int_as_float(e[j].z);
    const float2 q0 = lxy[c0], q1 = lxy[c1], q2 = lxy[c2];
    const float z0 = lz[c0], z1 = lz[c1], z2 = lz[c2];
    ax = fmaf(a0, q0.x, ax); ay = fmaf(a0, q0.y, ay); az = fmaf(a0, z0, az);
    ax = fmaf(a1, q1.x, ax); ay = fmaf(a1, q1.y, ay); az = fmaf(a1, z1, az);
    ax = fmaf(a2, q2.x, ax); ay = fmaf(a2, q2.y, ay); az = fmaf(a2, z2, az);
  }
}
// the same with all 24 gathers of the batch issued before the first product (see consume_h): the kernels that have the registers for it
__device__ __forceinline__ void consume_pf(const int4 (&e)[PB], const float2 *lxy, const float *lz, int base, float &ax, float &ay, float &az) {
  float2 q[PB][3];
  float z[PB][3];
#pragma unroll
  for (int j = 0; j < PB; j++) {
    const int c0 = base + (e[j].w & 1023), c1 = base + ((e[j].w >> 10) & 1023), c2 = base + ((e[j].w >> 20) & 1023);
    q[j][0] = lxy[c0]; q[j][1] = lxy[c1]; q[j][2] = lxy[c2];
    z[j][0] = lz[c0]; z[j][1] = lz[c1]; z[j][2] = lz[c2];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < PB; j++) {
    const float a0 = __int_as_float(e[j].x), a1 = __int_as_float(e[j].y), a2 = __int_as_float(e[j].z);
    ax = fmaf(a0, q[j][0].x, ax); ay = fmaf(a0, q[j][0].y, ay); az = fmaf(a0, z[j][0], az);
    ax = fmaf(a1, q[j][1].x, ax); ay = fmaf(a1, q[j][1].y, ay); az = fmaf(a1, z[j][1], az);
    ax = fmaf(a2, q[j][2].x, ax); ay = fmaf(a2, q[j][2].y, ay); az = fmaf(a2, z[j][2], az);
  }
}
// The same with the search direction held as four halves per row (x, y, z, unused: 8 bytes, ONE ds_read_b64 per non-zero instead of a
// b64 + a b32, and a third less LDS): the products are v_fma_mix_f32 (half operand converted inside the FMA).
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// (v_fma_mix_f32 written out: left to itself the compiler converts x and y with two v_cvt_f32_f16 and pairs them in a v_pk_fma_f32 —
// 7.5 instructions per non-zero where this is 6: bit-field extract, shift-add onto the row's LDS byte address, ds_read_b64, three FMAs
// that take the half operand as it is)
__device__ __forceinline__ void fma3_h(float a, int lo, int hi, float &ax, float &ay, float &az) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(ax) : "v"(a), "v"(lo));
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(ay) : "v"(a), "v"(lo));
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(az) : "v"(a), "v"(hi));
}
typedef int pk_v2i __attribute__((ext_vector_type(2)));
typedef const pk_v2i __attribute__((address_space(3))) *lds_int2p;
// rowbase = LDS byte address of the direction's row (i - 512): column c of the row's packets sits at rowbase + 8 (c - i + 512)
__device__ __forceinline__ void consume_h(const int4 (&e)[PB], unsigned rowbase, float &ax, float &ay, float &az) {
  // all gathers of the batch are issued before the first product: 12 ds_read_b64 in flight (24 registers) instead of 3
  pk_v2i q[PB][3];
#pragma unroll
  for (int j = 0; j < PB; j++) {
    const unsigned w = (unsigned) e[j].w;
    unsigned d0, d1, d2;      // (opaque extracts: the compiler otherwise rewrites extract-and-scale as shift + mask and needs a third instruction to add)
    asm("v_bfe_u32 %0, %1, 0, 10" : "=v"(d0) : "v"(w));
    asm("v_bfe_u32 %0, %1, 10, 10" : "=v"(d1) : "v"(w));
    asm("v_bfe_u32 %0, %1, 20, 10" : "=v"(d2) : "v"(w));
    q[j][0] = *(lds_int2p) (size_t) (rowbase + (d0 << 3));
    q[j][1] = *(lds_int2p) (size_t) (rowbase + (d1 << 3));
    q[j][2] = *(lds_int2p) (size_t) (rowbase + (d2 << 3));
  }
  // Nothing crosses: the compiler's scheduler otherwise interleaves read / wait / product to shorten the live ranges of the gathered
  // values — two or three reads in flight and an exposed LDS round trip per non-zero on 13 of the 20 rows of the 10 000-vertex kernel
  // (code object of round 4: `ds_read_b64; s_waitcnt lgkmcnt(0); 3 x v_fma_mix` chains). Behind the fence the waits count down
  // lgkmcnt(11) ... (0) while the products issue.
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < PB; j++) {
    fma3_h(__int_as_float(e[j].x), q[j][0].x, q[j][0].y, ax, ay, az);
    fma3_h(__int_as_float(e[j].y), q[j][1].x, q[j][1].y, ax, ay, az);
    fma3_h(__int_as_float(e[j].z), q[j][2].x, q[j][2].y, ax, ay, az);
  }
}
// First batch of a row: the row's OWN direction entry (the unit diagonal's operand, LDS address rowbase + 8 * 512) is read in the same
// group as the 12 gathers and enters the sums last, as one more v_fma_mix with coefficient 1 — read on its own ahead of the gathers
// (to initialise the sums) it cost an exposed LDS round trip per row before the first gather was even issued. `own` returns its bits for
// the dot products; they are made to depend on the finished sums so that their conversions are scheduled behind the products.
__device__ __forceinline__ void consume_h_row(const int4 (&e)[PB], unsigned rowbase, float &ax, float &ay, float &az, pk_v2i &own) {
  pk_v2i q[PB][3];
#pragma unroll
  for (int j = 0; j < PB; j++) {
    const unsigned w = (unsigned) e[j].w;
    unsigned d0, d1, d2;
    asm("v_bfe_u32 %0, %1, 0, 10" : "=v"(d0) : "v"(w));
    asm("v_bfe_u32 %0, %1, 10, 10" : "=v"(d1) : "v"(w));
    asm("v_bfe_u32 %0, %1, 20, 10" : "=v"(d2) : "v"(w));
    q[j][0] = *(lds_int2p) (size_t) (rowbase + (d0 << 3));
    q[j][1] = *(lds_int2p) (size_t) (rowbase + (d1 << 3));
    q[j][2] = *(lds_int2p) (size_t) (rowbase + (d2 << 3));
  }
  pk_v2i o = *(lds_int2p) (size_t) (rowbase + 4096u);
  __builtin_amdgcn_sched_barrier(0);
  ax = 0.f; ay = 0.f; az = 0.f;
#pragma unroll
  for (int j = 0; j < PB; j++) {
    fma3_h(__int_as_float(e[j].x), q[j][0].x, q[j][0].y, ax, ay, az);
    fma3_h(__int_as_float(e[j].y), q[j][1].x, q[j][1].y, ax, ay, az);
    fma3_h(__int_as_float(e[j].z), q[j][2].x, q[j][2].y, ax, ay, az);
  }
  asm("v_fma_mix_f32 %0, 1.0, %1, %0 op_sel_hi:[0,1,0]" : "+v"(ax) : "v"(o.x));
  asm("v_fma_mix_f32 %0, 1.0, %1, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(ay) : "v"(o.x));
  asm("v_fma_mix_f32 %0, 1.0, %1, %0 op_sel_hi:[0,1,0]" : "+v"(az) : "v"(o.y));
  own = o;
}
// acc += <(x, y, z), the three halves of q>: the dot products of the product loop take the direction's halves as they are, behind the sums
__device__ __forceinline__ void dot3_h(float x, float y, float z, pk_v2i q, float &acc) {
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(x), "v"(q.x));
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(y), "v"(q.x));
  asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(z), "v"(q.y));
}
__device__ __forceinline__ unsigned lds_byte_address(const void *p) {
  return (unsigned) (size_t) (const __attribute__((address_space(3))) char *) p;
}
__device__ __forceinline__ h4 pack_h4(float x, float y, float z) {
  typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
  const hp2 a = __builtin_amdgcn_cvt_pkrtz(x, y), b = __builtin_amdgcn_cvt_pkrtz(z, 0.f);
  int2 bits;
  __builtin_memcpy(&bits.x, &a, 4); __builtin_memcpy(&bits.y, &b, 4);
  h4 q;
  __builtin_memcpy(&q, &bits, 8);
  return q;
}
// power of two s with bound * s in [4096, 8192): scale of a vector whose entries are bounded by `bound` before it is rounded to halves
__device__ __forceinline__ float half_scale(float bound) {
  if (!(bound > 1e-30f) || !(bound < 1e30f)) return 1.f;
  int e;
  (void) frexpf(bound, &e);            // bound = m 2^e, m in [0.5, 1)
  return ldexpf(1.f, 13 - e);
}
// two block sums in one pass (fp32 inside a wave, fp64 across the waves)
template <int THREADS>
__device__ __forceinline__ void block_sum2_f(float &a, float &b, double *red, double &sa, double &sb) {
  a = wave_sum_f(a); b = wave_sum_f(b);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  constexpr int NW = THREADS / 64;
  __syncthreads();
  if (l == 0) { red[w] = (double) a; red[NW + w] = (double) b; }
  __syncthreads();
  sa = 0; sb = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) { sa += red[k]; sb += red[NW + k]; }
}

// The two reductions of a CG iteration without their leading barrier: each has a buffer of its own (`red2`: p.Ap and p.r, `red`: r.r),
// written once per iteration, and between a buffer's read and its next write lie at least two barriers of the loop (the other reduction's
// and the one that ends the direction update) — 3 instead of 5 barriers per iteration.
template <int THREADS>
__device__ __forceinline__ double block_sum_f_nb(float v, double *red) {
  v = wave_sum_f(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) red[w] = (double) v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; k++) s += red[k];
  return s;
}
template <int THREADS>
__device__ __forceinline__ void block_sum2_f_nb(float &a, float &b, double *red, double &sa, double &sb) {
  a = wave_sum_f(a); b = wave_sum_f(b);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  constexpr int NW = THREADS / 64;
  if (l == 0) { red[w] = (double) a; red[NW + w] = (double) b; }
  __syncthreads();
  sa = 0; sb = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) { sa += red[k]; sb += red[NW + k]; }
}

template <int NP>
__device__ __forceinline__ void consume(const int4 (&e)[PB], const float *lp, int base, float &ax, float &ay, float &az) {
  consume_p(e, (const float2 *) lp, lp + 2 * NP, base, ax, ay, az);
}

}  // namespace
}  // namespace dc
