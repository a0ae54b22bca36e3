#include "dc_deflate.h"
#include <algorithm>
#include <cmath>
#include <cstdint>

namespace dc {
namespace {

struct Scaled {                   // Ahat = D^-1/2 P D^-1/2 in CSR (unit diagonal)
  int N;
  const std::vector<int> &ptr, &col;
  std::vector<double> val, sq;
  explicit Scaled(const HostSystem &H) : N(H.N), ptr(H.P_ptr), col(H.P_col), val(H.P_val.size()), sq(H.N) {
    for (int i = 0; i < N; i++) {
      double d = 1.0;
      for (int q = ptr[i]; q < ptr[i + 1]; q++) if (col[q] == i) d = H.P_val[q];
      sq[i] = 1.0 / std::sqrt(d);
    }
    for (int i = 0; i < N; i++)
      for (int q = ptr[i]; q < ptr[i + 1]; q++) val[q] = H.P_val[q] * sq[i] * sq[col[q]];
  }
  void mul(const double *x, double *y) const {
    for (int i = 0; i < N; i++) {
      double s = 0;
      for (int q = ptr[i]; q < ptr[i + 1]; q++) s += val[q] * x[col[q]];
      y[i] = s;
    }
  }
  double gershgorin() const {
    double m = 0;
    for (int i = 0; i < N; i++) {
      double s = 0;
      for (int q = ptr[i]; q < ptr[i + 1]; q++) s += std::fabs(val[q]);
      m = std::max(m, s);
    }
    return m;
  }
};

// plain CG on the scaled system (= Jacobi-PCG on P), relative residual tol; returns the iteration count
int probe_cg(const Scaled &A, const std::vector<double> &b, double tol, int cap) {
  const int N = A.N;
  std::vector<double> x(N, 0.0), r = b, p = b, Ap(N);
  double rz = 0;
  for (double v : r) rz += v * v;
  const double stop = tol * tol * rz;
  if (!(rz > 0)) return 0;
  for (int it = 1; it <= cap; it++) {
    A.mul(p.data(), Ap.data());
    double pAp = 0;
    for (int i = 0; i < N; i++) pAp += p[i] * Ap[i];
    const double al = rz / pAp;
    double rn = 0;
    for (int i = 0; i < N; i++) { x[i] += al * p[i]; r[i] -= al * Ap[i]; rn += r[i] * r[i]; }
    if (rn <= stop) return it;
    const double be = rn / rz;
    rz = rn;
    for (int i = 0; i < N; i++) p[i] = r[i] + be * p[i];
  }
  return cap;
}

// columns of V (N x m, column-major) orthonormalised by modified Gram-Schmidt, twice
void orthonormalise(std::vector<double> &V, int N, int m) {
  for (int pass = 0; pass < 2; pass++)
    for (int j = 0; j < m; j++) {
      double *vj = &V[(size_t) j * N];
      for (int l = 0; l < j; l++) {
        const double *vl = &V[(size_t) l * N];
        double d = 0;
        for (int i = 0; i < N; i++) d += vj[i] * vl[i];
        for (int i = 0; i < N; i++) vj[i] -= d * vl[i];
      }
      double n = 0;
      for (int i = 0; i < N; i++) n += vj[i] * vj[i];
      n = std::sqrt(std::max(n, 1e-300));
      for (int i = 0; i < N; i++) vj[i] /= n;
    }
}

// eigen-decomposition of a small symmetric matrix (cyclic Jacobi): S (m x m, row-major) -> eigenvalues ascending in w, vectors in the
// columns of Q (row-major m x m)
void jacobi_eig(std::vector<double> S, int m, std::vector<double> &w, std::vector<double> &Q) {
  Q.assign((size_t) m * m, 0.0);
  for (int i = 0; i < m; i++) Q[(size_t) i * m + i] = 1.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int i = 0; i < m; i++) for (int j = i + 1; j < m; j++) off += S[(size_t) i * m + j] * S[(size_t) i * m + j];
    if (off < 1e-30) break;
    for (int p = 0; p < m; p++)
      for (int q = p + 1; q < m; q++) {
        const double apq = S[(size_t) p * m + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double th = (S[(size_t) q * m + q] - S[(size_t) p * m + p]) / (2 * apq);
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1)), c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < m; k++) {      // columns p, q of S and Q
          const double skp = S[(size_t) k * m + p], skq = S[(size_t) k * m + q];
          S[(size_t) k * m + p] = c * skp - s * skq; S[(size_t) k * m + q] = s * skp + c * skq;
          const double qkp = Q[(size_t) k * m + p], qkq = Q[(size_t) k * m + q];
          Q[(size_t) k * m + p] = c * qkp - s * qkq; Q[(size_t) k * m + q] = s * qkp + c * qkq;
        }
        for (int k = 0; k < m; k++) {      // rows p, q of S
          const double spk = S[(size_t) p * m + k], sqk = S[(size_t) q * m + k];
          S[(size_t) p * m + k] = c * spk - s * sqk; S[(size_t) q * m + k] = s * spk + c * sqk;
        }
      }
  }
  std::vector<int> idx(m);
  for (int i = 0; i < m; i++) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return S[(size_t) a * m + a] < S[(size_t) b * m + b]; });
  w.resize(m);
  std::vector<double> Qs((size_t) m * m);
  for (int j = 0; j < m; j++) {
    w[j] = S[(size_t) idx[j] * m + idx[j]];
    for (int k = 0; k < m; k++) Qs[(size_t) k * m + j] = Q[(size_t) k * m + idx[j]];
  }
  Q.swap(Qs);
}

// inverse of a small SPD matrix (Gauss-Jordan with partial pivoting), row-major
bool invert(std::vector<double> A, int m, std::vector<double> &inv) {
  inv.assign((size_t) m * m, 0.0);
  for (int i = 0; i < m; i++) inv[(size_t) i * m + i] = 1.0;
  for (int c = 0; c < m; c++) {
    int piv = c;
    for (int r = c + 1; r < m; r++) if (std::fabs(A[(size_t) r * m + c]) > std::fabs(A[(size_t) piv * m + c])) piv = r;
    if (!(std::fabs(A[(size_t) piv * m + c]) > 1e-300)) return false;
    if (piv != c) for (int k = 0; k < m; k++) { std::swap(A[(size_t) piv * m + k], A[(size_t) c * m + k]); std::swap(inv[(size_t) piv * m + k], inv[(size_t) c * m + k]); }
    const double d = 1.0 / A[(size_t) c * m + c];
    for (int k = 0; k < m; k++) { A[(size_t) c * m + k] *= d; inv[(size_t) c * m + k] *= d; }
    for (int r = 0; r < m; r++) {
      if (r == c) continue;
      const double f = A[(size_t) r * m + c];
      if (f == 0) continue;
      for (int k = 0; k < m; k++) { A[(size_t) r * m + k] -= f * A[(size_t) c * m + k]; inv[(size_t) r * m + k] -= f * inv[(size_t) c * m + k]; }
    }
  }
  return true;
}

}  // namespace

bool HostDeflation::build(const HostSystem &H, int want, int rows_padded, int auto_threshold) {
  *this = HostDeflation();
  const int N = H.N;
  if (want == 0 || N < 256 || rows_padded < N) return false;
  Scaled A(H);
  {  // probe: how many Jacobi-PCG iterations a smooth right-hand side takes (the momentum of a uniform velocity, scaled: D^-1/2 m)
    std::vector<double> b(N);
    for (int i = 0; i < N; i++) b[i] = A.sq[i] * H.mass[i];
    probe_iterations = probe_cg(A, b, 1e-4, 2000);
  }
  if (want < 0 && probe_iterations <= auto_threshold) return false;
  k = want > 0 ? std::min(want, 32) : 16;
  const int m = k + 8;                      // block size of the subspace iteration (guard vectors)
  // Chebyshev-filtered subspace iteration for the m lowest eigenpairs: V <- T_deg((b + a - 2 Ahat) / (b - a)) V damps [a, b] and amplifies
  // [0, a); a follows the largest Ritz value of the block, b is a Gershgorin bound of the spectrum
  const double bnd = A.gershgorin() * 1.001;
  double a = bnd / 20.0;
  const int deg = 40, outer = 10;
  std::vector<double> V((size_t) N * m), W((size_t) N * m), T0(N), T1(N), T2(N);
  uint64_t seed = 0x9E3779B97F4A7C15ull;
  for (double &v : V) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; v = (double) (int64_t) (seed >> 11) / 9007199254740992.0 - 0.5; }
  orthonormalise(V, N, m);
  std::vector<double> w, Q;
  for (int it = 0; it < outer; it++) {
    const double e = (bnd - a) / 2, c = (bnd + a) / 2;
    for (int j = 0; j < m; j++) {           // three-term recurrence per column: T_{n+1} = 2 L T_n - T_{n-1}, L = (c - Ahat) / e
      double *v = &V[(size_t) j * N];
      std::copy(v, v + N, T0.begin());
      A.mul(T0.data(), T2.data());
      for (int i = 0; i < N; i++) T1[i] = (c * T0[i] - T2[i]) / e;
      for (int n = 2; n <= deg; n++) {
        A.mul(T1.data(), T2.data());
        for (int i = 0; i < N; i++) { const double t = 2.0 * (c * T1[i] - T2[i]) / e - T0[i]; T0[i] = T1[i]; T1[i] = t; }
        if ((n & 7) == 0) {                 // keep the magnitudes in range (the filter amplifies by cosh(deg * acosh(x)))
          double mx = 0;
          for (int i = 0; i < N; i++) mx = std::max(mx, std::fabs(T1[i]));
          if (mx > 1e100) for (int i = 0; i < N; i++) { T0[i] /= mx; T1[i] /= mx; }
        }
      }
      std::copy(T1.begin(), T1.end(), v);
    }
    orthonormalise(V, N, m);
    for (int j = 0; j < m; j++) A.mul(&V[(size_t) j * N], &W[(size_t) j * N]);      // W = Ahat V
    std::vector<double> S((size_t) m * m);
    for (int p = 0; p < m; p++)
      for (int q = p; q < m; q++) {
        double d = 0;
        for (int i = 0; i < N; i++) d += V[(size_t) p * N + i] * W[(size_t) q * N + i];
        S[(size_t) p * m + q] = S[(size_t) q * m + p] = d;
      }
    jacobi_eig(S, m, w, Q);
    std::vector<double> Vn((size_t) N * m, 0.0);                                     // V <- V Q (Ritz vectors, ascending)
    for (int j = 0; j < m; j++)
      for (int l = 0; l < m; l++) {
        const double q = Q[(size_t) l * m + j];
        if (q == 0) continue;
        for (int i = 0; i < N; i++) Vn[(size_t) j * N + i] += q * V[(size_t) l * N + i];
      }
    V.swap(Vn);
    a = std::max(w[m - 1], 1e-6 * bnd);
  }
  // tables: U (the k lowest Ritz vectors), Ahat U, (U^T Ahat U)^-1
  std::vector<double> AUd((size_t) N * k), Sg((size_t) k * k), Gd;
  for (int j = 0; j < k; j++) A.mul(&V[(size_t) j * N], &AUd[(size_t) j * N]);
  for (int p = 0; p < k; p++)
    for (int q = p; q < k; q++) {
      double d = 0;
      for (int i = 0; i < N; i++) d += V[(size_t) p * N + i] * AUd[(size_t) q * N + i];
      Sg[(size_t) p * k + q] = Sg[(size_t) q * k + p] = d;
    }
  if (!invert(Sg, k, Gd)) { k = 0; return false; }
  rows = rows_padded;
  U.assign((size_t) rows * k, 0.f); AU.assign((size_t) rows * k, 0.f); G.assign(Gd.begin(), Gd.end());
  for (int i = 0; i < N; i++)
    for (int j = 0; j < k; j++) { U[(size_t) i * k + j] = (float) V[(size_t) j * N + i]; AU[(size_t) i * k + j] = (float) AUd[(size_t) j * N + i]; }
  ritz.assign(w.begin(), w.begin() + k);
  ok = true;
  return true;
}

}  // namespace dc
