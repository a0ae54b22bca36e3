#include "dc_dense.h"
#include <algorithm>
#include <cmath>
#include <cstdint>

namespace dc {

bool HostDense::build(const HostSystem &H, int max_n) {
  *this = HostDense();
  n = H.N;
  if (n <= 0 || n > max_n) return false;
  ld = (n + 63) / 64 * 64;
  rows = n + kDensePadRows;
  const size_t nn = (size_t) n;
  // Ahat with the fp32 scaling the kernels use (dc_packets.cpp)
  std::vector<double> s(nn), A(nn * nn, 0.0);
  for (int i = 0; i < n; i++) {
    double d = 0;
    for (int k = H.P_ptr[i]; k < H.P_ptr[i + 1]; k++) if (H.P_col[k] == i) d = H.P_val[k];
    if (!(d > 0)) return false;
    s[i] = (double) (float) std::sqrt((double) (float) (1.0 / d));
  }
  for (int i = 0; i < n; i++)
    for (int k = H.P_ptr[i]; k < H.P_ptr[i + 1]; k++) A[i * nn + H.P_col[k]] = H.P_val[k] * s[i] * s[H.P_col[k]];
  // Cholesky A = L L^T, row by row (both operands of the inner product are contiguous)
  std::vector<double> L(nn * nn, 0.0);
  for (size_t i = 0; i < nn; i++)
    for (size_t j = 0; j <= i; j++) {
      double acc = A[i * nn + j];
      const double *li = &L[i * nn], *lj = &L[j * nn];
      for (size_t k = 0; k < j; k++) acc -= li[k] * lj[k];
      if (i == j) {
        if (!(acc > 0)) return false;
        L[i * nn + i] = std::sqrt(acc);
      } else {
        L[i * nn + j] = acc / L[j * nn + j];
      }
    }
  // Y = (L^-1)^T stored by rows: Y[j][i] = (L^-1)[i][j], i >= j
  std::vector<double> &Y = A;        // A is no longer needed
  std::fill(Y.begin(), Y.end(), 0.0);
  for (size_t j = 0; j < nn; j++) {
    double *yj = &Y[j * nn];
    for (size_t i = j; i < nn; i++) {
      const double *li = &L[i * nn];
      double acc = (i == j) ? 1.0 : 0.0;
      for (size_t k = j; k < i; k++) acc -= li[k] * yj[k];
      yj[i] = acc / li[i];
    }
  }
  // Ahat^-1 = L^-T L^-1 :  inv[a][b] = sum_{i >= max(a,b)} Y[a][i] Y[b][i]
  inv.assign((size_t) rows * ld, 0.f);
  for (size_t a = 0; a < nn; a++)
    for (size_t b = 0; b <= a; b++) {
      const double *ya = &Y[a * nn], *yb = &Y[b * nn];
      double acc = 0;
      for (size_t i = a; i < nn; i++) acc += ya[i] * yb[i];
      inv[a * ld + b] = inv[b * ld + a] = (float) acc;
    }
  // probe: contraction of  v -> v - Ahat (inv32 v)  (Ahat rebuilt from the sparse matrix)
  uint64_t seed = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (double) (int64_t) (seed >> 11) / 9007199254740992.0 - 0.5; };
  std::vector<double> v(nn), w(nn), r(nn);
  for (int probe = 0; probe < 3; probe++) {
    double vn = 0, rn = 0;
    for (size_t i = 0; i < nn; i++) { v[i] = rnd(); vn += v[i] * v[i]; }
    for (size_t i = 0; i < nn; i++) {
      double acc = 0;
      const float *row = &inv[i * ld];
      for (size_t j = 0; j < nn; j++) acc += (double) row[j] * v[j];
      w[i] = acc;
    }
    for (int i = 0; i < n; i++) {
      double acc = 0;
      for (int k = H.P_ptr[i]; k < H.P_ptr[i + 1]; k++) acc += H.P_val[k] * s[i] * s[H.P_col[k]] * w[H.P_col[k]];
      r[i] = v[i] - acc;
      rn += r[i] * r[i];
    }
    defect = std::max(defect, std::sqrt(rn / vn));
  }
  ok = defect < 1e-2;
  if (!ok) inv.clear();
  return ok;
}

}  // namespace dc
