// Device side of the explicit inverse of small systems (dc_dense.h): y = Ahat^-1 x for the three coordinate planes of one
// rollout, x in LDS, the matrix (batch-shared, L2 resident) streamed once per product.
#pragma once
#include "dc_devlib.h"

namespace dc {

// number of column chunks the product is split into so that every wave of the workgroup has ~4 (row group, chunk) units
__host__ __device__ __forceinline__ int dense_chunks(int ld, int waves) {
  const int R = ld >> 6;
  int C = (4 * waves + R - 1) / R;
  return C < 1 ? 1 : (C > 8 ? 8 : C);
}
// floats of LDS the partial sums need
__host__ __device__ __forceinline__ int dense_lds_floats(int ld, int waves) { return 3 * ld * dense_chunks(ld, waves); }

// Partial products: unit (row group rw, chunk c) = rows 64 rw .. 64 rw + 63 against columns [c JC, (c + 1) JC), one wave
// per unit, lane = row. The matrix is symmetric, so "row i, column j" is read as inv[j * ld + i]: the 64 lanes of a wave
// read 256 consecutive bytes, and x_j is one LDS broadcast. part[(c * 3 + comp) * ld + i]; the caller sums over c after a
// barrier. Needs a barrier before (x complete) — returns the number of chunks.
template <int THREADS>
__device__ __forceinline__ int dense_partials(const DevSystem &S, const float2 *__restrict__ xy, const float *__restrict__ xz, float *__restrict__ part) {
  constexpr int WAVES = THREADS / 64, UN = 16;
  const int n = S.N, ld = S.dense_ld, R = ld >> 6;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = dense_chunks(ld, WAVES);
  const int JC = (((n + C - 1) / C) + UN - 1) / UN * UN;           // C * JC <= n + kDensePadRows (zero rows)
  for (int u = wv; u < R * C; u += WAVES) {
    const int rw = u % R, c = u / R;
    const int i = rw * 64 + lane;
    const float DC_G *col = S.dense_inv + (size_t) (c * JC) * ld + i;
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int j0 = c * JC; j0 < (c + 1) * JC; j0 += UN) {
      float bv[UN];
#pragma unroll
      for (int jj = 0; jj < UN; jj++) bv[jj] = col[(size_t) jj * ld];
      col += (size_t) UN * ld;
#pragma unroll
      for (int jj = 0; jj < UN; jj++) {
        const int j = min(j0 + jj, n - 1);                          // rows >= n of the matrix are zero
        const float2 q = xy[j];
        const float zz = xz[j];
        ax = fmaf(bv[jj], q.x, ax); ay = fmaf(bv[jj], q.y, ay); az = fmaf(bv[jj], zz, az);
      }
    }
    part[(c * 3 + 0) * ld + i] = ax; part[(c * 3 + 1) * ld + i] = ay; part[(c * 3 + 2) * ld + i] = az;
  }
  return C;
}

__device__ __forceinline__ f3 dense_row_sum(const float *part, int ld, int C, int i) {
  f3 s = mk(0, 0, 0);
  if (i < ld)
    for (int c = 0; c < C; c++) { s.x += part[(c * 3 + 0) * ld + i]; s.y += part[(c * 3 + 1) * ld + i]; s.z += part[(c * 3 + 2) * ld + i]; }
  return s;
}

}  // namespace dc
