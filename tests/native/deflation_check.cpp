// CPU check of the deflation-space builder (diffcloth_amd/csrc/dc_deflate.cpp): a badly graded sheet, the 16 lowest Ritz pairs.
#include "dc_deflate.h"
#include <chrono>
#include <cmath>
#include <cstdio>
using namespace dc;
int main() {
  const int nx = 60, ny = 40;
  std::vector<double> pos;
  std::vector<int> tri;
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      const double x = 4.0 * std::pow((double) i / (nx - 1), 2.2), y = 3.0 * std::pow((double) j / (ny - 1), 1.7);
      pos.push_back(x); pos.push_back(0.02 * std::sin(3 * x) * std::cos(2 * y)); pos.push_back(y);
    }
  for (int j = 0; j + 1 < ny; j++)
    for (int i = 0; i + 1 < nx; i++) {
      const int a = j * nx + i, b = a + 1, c = a + nx, d = c + 1;
      tri.insert(tri.end(), {a, b, d}); tri.insert(tri.end(), {a, d, c});
    }
  HostSystem H;
  if (!H.set_mesh(nx * ny, pos.data(), (int) tri.size() / 3, tri.data())) { printf("mesh: %s\n", H.error.c_str()); return 1; }
  H.att_vertex = {0, nx - 1};
  if (!H.build_numerics(1.0 / 120, 0.2, 800.0, 0.05, 10000.0)) { printf("numerics: %s\n", H.error.c_str()); return 1; }
  HostDeflation D;
  const auto t0 = std::chrono::steady_clock::now();
  const bool built = D.build(H, -1, (H.N + 63) / 64 * 64);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("N %d: probe %d iterations, built %d, k %d, %.2f s\n", H.N, D.probe_iterations, (int) built, D.k, dt);
  if (!built || D.k != 16 || D.probe_iterations <= 80) { printf("expected a deflation space\n"); return 2; }
  double worst = 0, orth = 0, gram = 0;
  for (int j = 0; j < D.k; j++) {
    double r = 0, n = 0, uau = 0;
    for (int i = 0; i < H.N; i++) {
      const double u = D.U[(size_t) i * D.k + j], au = D.AU[(size_t) i * D.k + j];
      r += (au - D.ritz[j] * u) * (au - D.ritz[j] * u); n += u * u; uau += u * au;
    }
    worst = std::max(worst, std::sqrt(r)); orth = std::max(orth, std::fabs(n - 1));
    gram = std::max(gram, std::fabs(uau * D.G[(size_t) j * D.k + j] - 1.0));       // the vectors are Ritz vectors: U^T A U is diagonal
  }
  for (int j = 1; j < D.k; j++) if (!(D.ritz[j] >= D.ritz[j - 1])) { printf("Ritz values not ascending\n"); return 3; }
  printf("lowest Ritz values %.3e ... %.3e, worst eigen-residual %.2e, | |u|^2 - 1 | %.1e, | u^T A u G_jj - 1 | %.1e\n", D.ritz[0], D.ritz[D.k - 1], worst, orth, gram);
  if (!(worst < 1e-3 && orth < 1e-4 && gram < 1e-2 && dt < 20)) return 4;
  HostDeflation off;
  if (off.build(H, 0, (H.N + 63) / 64 * 64)) { printf("forward_deflation = 0 must not build\n"); return 5; }
  printf("ALL OK\n");
  return 0;
}
