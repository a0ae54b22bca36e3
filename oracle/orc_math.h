// TEST INFRASTRUCTURE ONLY — fp64 CPU oracle for the DiffCloth hot path.
// Nothing under diffcloth_amd/ may include, link or call this code.
//
// Tiny dependency-free dense helpers standing in for the Eigen fixed-size types the
// reference uses (src/code/engine/Macros.h:30-80: Vec3d, Mat3x3d, Mat6x9d, ...).
#pragma once
#include <cmath>
#include <cstdio>
#include <vector>
#include <cassert>

namespace orc {

struct V3 {
  double x = 0, y = 0, z = 0;
  V3() {}
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  double &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  V3 operator+(const V3 &o) const { return V3(x + o.x, y + o.y, z + o.z); }
  V3 operator-(const V3 &o) const { return V3(x - o.x, y - o.y, z - o.z); }
  V3 operator-() const { return V3(-x, -y, -z); }
  V3 operator*(double s) const { return V3(x * s, y * s, z * s); }
  V3 operator/(double s) const { return V3(x / s, y / s, z / s); }
  V3 &operator+=(const V3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
  V3 &operator-=(const V3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
  double dot(const V3 &o) const { return x * o.x + y * o.y + z * o.z; }
  V3 cross(const V3 &o) const { return V3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
  double sqnorm() const { return dot(*this); }
  double norm() const { return std::sqrt(sqnorm()); }
  // Eigen's normalized(): divides by the norm when it is > 0, returns the vector unchanged otherwise.
  V3 normalized() const { double n = norm(); return n > 0 ? (*this) / n : *this; }
};
inline V3 operator*(double s, const V3 &v) { return v * s; }

inline V3 seg3(const std::vector<double> &v, int i) { return V3(v[3 * i], v[3 * i + 1], v[3 * i + 2]); }
inline V3 seg3(const double *v, int i) { return V3(v[3 * i], v[3 * i + 1], v[3 * i + 2]); }
inline void addseg3(std::vector<double> &v, int i, const V3 &a) { v[3 * i] += a.x; v[3 * i + 1] += a.y; v[3 * i + 2] += a.z; }
inline void setseg3(std::vector<double> &v, int i, const V3 &a) { v[3 * i] = a.x; v[3 * i + 1] = a.y; v[3 * i + 2] = a.z; }

// Dynamic-size dense row-major matrix; only used for the per-constraint Jacobian chain rule.
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t) r_ * c_, 0.0) {}
  double &operator()(int i, int j) { return a[(size_t) i * c + j]; }
  double operator()(int i, int j) const { return a[(size_t) i * c + j]; }
  static Mat identity(int n) { Mat m(n, n); for (int i = 0; i < n; i++) m(i, i) = 1; return m; }
  Mat operator*(const Mat &o) const {
    assert(c == o.r);
    Mat m(r, o.c);
    for (int i = 0; i < r; i++)
      for (int k = 0; k < c; k++) {
        double v = (*this)(i, k);
        if (v == 0) continue;
        for (int j = 0; j < o.c; j++) m(i, j) += v * o(k, j);
      }
    return m;
  }
  Mat operator+(const Mat &o) const { Mat m = *this; for (size_t i = 0; i < a.size(); i++) m.a[i] += o.a[i]; return m; }
  Mat operator-(const Mat &o) const { Mat m = *this; for (size_t i = 0; i < a.size(); i++) m.a[i] -= o.a[i]; return m; }
  Mat operator*(double s) const { Mat m = *this; for (auto &v : m.a) v *= s; return m; }
  Mat T() const { Mat m(c, r); for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) m(j, i) = (*this)(i, j); return m; }
  Mat block(int i0, int j0, int nr, int nc) const {
    Mat m(nr, nc);
    for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) m(i, j) = (*this)(i0 + i, j0 + j);
    return m;
  }
  void setBlock(int i0, int j0, const Mat &b) {
    for (int i = 0; i < b.r; i++) for (int j = 0; j < b.c; j++) (*this)(i0 + i, j0 + j) = b(i, j);
  }
};

inline Mat outer(const V3 &a, const V3 &b) {
  Mat m(3, 3);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = a[i] * b[j];
  return m;
}
inline V3 mul3(const Mat &m, const V3 &v) {
  return V3(m(0, 0) * v.x + m(0, 1) * v.y + m(0, 2) * v.z, m(1, 0) * v.x + m(1, 1) * v.y + m(1, 2) * v.z,
            m(2, 0) * v.x + m(2, 1) * v.y + m(2, 2) * v.z);
}
inline V3 mul3T(const Mat &m, const V3 &v) {
  return V3(m(0, 0) * v.x + m(1, 0) * v.y + m(2, 0) * v.z, m(0, 1) * v.x + m(1, 1) * v.y + m(2, 1) * v.z,
            m(0, 2) * v.x + m(1, 2) * v.y + m(2, 2) * v.z);
}
// Kronecker product, as the reference's kronecker<> helper (engine/UtilityFunctions.h:52-65).
inline Mat kron(const Mat &A, const Mat &B) {
  Mat m(A.r * B.r, A.c * B.c);
  for (int i = 0; i < A.r; i++) for (int j = 0; j < A.c; j++)
    for (int k = 0; k < B.r; k++) for (int l = 0; l < B.c; l++) m(i * B.r + k, j * B.c + l) = A(i, j) * B(k, l);
  return m;
}

// SVD of a 2x2 matrix by one-sided Jacobi, returning U, singular values, V (A = U diag(s) V^T).
// Stands in for Eigen::JacobiSVD<Mat2x2d> (Triangle.cpp:345, :424). Only U V^T and V diag(s) V^T are
// consumed downstream, and both are unique for a non-singular A.
inline void svd2x2(const double A[4], double U[4], double s[2], double V[4]) {
  // A = [a b; c d] row-major. Symmetric eigen-decomposition of A^T A gives V and s; U = A V / s.
  double a = A[0], b = A[1], c = A[2], d = A[3];
  double m00 = a * a + c * c, m01 = a * b + c * d, m11 = b * b + d * d;
  double theta = 0.5 * std::atan2(2 * m01, m00 - m11);
  double ct = std::cos(theta), st = std::sin(theta);
  // V columns: (ct, st), (-st, ct)
  V[0] = ct; V[1] = -st; V[2] = st; V[3] = ct;
  double l0 = m00 * ct * ct + 2 * m01 * ct * st + m11 * st * st;
  double l1 = m00 * st * st - 2 * m01 * ct * st + m11 * ct * ct;
  s[0] = std::sqrt(std::max(l0, 0.0));
  s[1] = std::sqrt(std::max(l1, 0.0));
  // U = A V diag(1/s)
  double av00 = a * V[0] + b * V[2], av01 = a * V[1] + b * V[3];
  double av10 = c * V[0] + d * V[2], av11 = c * V[1] + d * V[3];
  if (s[0] > 1e-300) { U[0] = av00 / s[0]; U[2] = av10 / s[0]; } else { U[0] = 1; U[2] = 0; }
  if (s[1] > 1e-300 * (1 + s[0])) { U[1] = av01 / s[1]; U[3] = av11 / s[1]; } else { U[1] = -U[2]; U[3] = U[0]; }
}

}  // namespace orc
