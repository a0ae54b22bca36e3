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
template <int NP>
__device__ __forceinline__ void consume(const int4 (&e)[PB], const float *lp, int base, float &ax, float &ay, float &az) {
  consume_p(e, (const float2 *) lp, lp + 2 * NP, base, ax, ay, az);
}

}  // namespace
}  // namespace dc
