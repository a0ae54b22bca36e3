// Test harness (CPU): the host-side table builders of the HIP kernels, checked against the plain constraint system.
//   g++ -O1 -std=c++17 -I diffcloth_amd/csrc tests/native/host_tables_check.cpp diffcloth_amd/csrc/dc_system.cpp \
//       diffcloth_amd/csrc/dc_windows.cpp diffcloth_amd/csrc/dc_packets.cpp -o host_tables_check
// Prints one line per check and exits non-zero on the first failure (driven by tests/test_host_native.py).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>
#include "dc_packets.h"
#include "dc_dense.h"
#include "dc_system.h"
#include "dc_windows.h"

using namespace dc;

static void fail(const char *what) { std::printf("FAIL %s\n", what); std::exit(1); }
static float asfloat(int b) { float f; std::memcpy(&f, &b, 4); return f; }

// triangulated nx x ny grid, optionally with shuffled vertex numbering
static void grid(int nx, int ny, bool shuffle, std::vector<double> &pos, std::vector<int> &tri) {
  std::mt19937 rng(7);
  std::vector<int> perm(nx * ny);
  for (size_t i = 0; i < perm.size(); i++) perm[i] = (int) i;
  if (shuffle) std::shuffle(perm.begin(), perm.end(), rng);
  std::uniform_real_distribution<double> jit(-0.01, 0.01);
  pos.assign(3 * (size_t) nx * ny, 0.0);
  for (int a = 0; a < ny; a++)
    for (int b = 0; b < nx; b++) {
      const int v = perm[a * nx + b];
      pos[3 * v] = 0.05 * b + jit(rng); pos[3 * v + 1] = 0.05 * a + jit(rng); pos[3 * v + 2] = jit(rng);
    }
  tri.clear();
  for (int a = 0; a + 1 < ny; a++)
    for (int b = 0; b + 1 < nx; b++) {
      const int v00 = perm[a * nx + b], v01 = perm[a * nx + b + 1], v10 = perm[(a + 1) * nx + b], v11 = perm[(a + 1) * nx + b + 1];
      tri.insert(tri.end(), {v00, v01, v11});
      tri.insert(tri.end(), {v00, v11, v10});
    }
}

static void check_packets(const HostSystem &H) {
  HostPackets P;
  if (!P.build(H)) fail("packets: build refused a banded system");
  if (512 * P.vpt < H.N || (int) P.sq_dinv.size() != 512 * P.vpt) fail("packets: padding");
  // decode every packet back into (row, col, value) and compare with D^-1/2 P D^-1/2
  for (int r = 0; r < 512 * P.vpt; r++) {
    const int ch = r / 64, l = r % 64;
    std::map<int, double> got;
    if (P.pk_n[ch] % 4 != 0 || P.pk_n[ch] < 4) fail("packets: rows must be padded to a multiple of 4 packets");
    for (int s = 0; s < P.pk_n[ch]; s++) {
      const int *q = &P.pk[4 * ((size_t) P.pk_ptr[ch] + (size_t) s * 64 + l)];
      for (int k = 0; k < 3; k++) {
        const int d = (q[3] >> (10 * k)) & 1023, col = r + d - 512;
        const float v = asfloat(q[k]);
        if (v == 0.f) { if (d != 512) fail("packets: padding entries must point at the row itself"); continue; }
        if (col < 0 || col >= H.N || r >= H.N) fail("packets: column out of range");
        if (got.count(col)) fail("packets: duplicate column");
        got[col] = v;
      }
    }
    if (r >= H.N) { if (!got.empty() || P.sq_dinv[r] != 0.f) fail("packets: padding rows must be empty"); continue; }
    double diag = 0;
    for (int k = H.P_ptr[r]; k < H.P_ptr[r + 1]; k++) if (H.P_col[k] == r) diag = H.P_val[k];
    if (std::fabs(P.sq_dinv[r] - std::sqrt(1.0 / diag)) > 1e-6 * std::sqrt(1.0 / diag)) fail("packets: sq_dinv");
    int off = 0;
    for (int k = H.P_ptr[r]; k < H.P_ptr[r + 1]; k++) {
      const int col = H.P_col[k];
      if (col == r) continue;
      off++;
      double dc = 0;
      for (int k2 = H.P_ptr[col]; k2 < H.P_ptr[col + 1]; k2++) if (H.P_col[k2] == col) dc = H.P_val[k2];
      const double want = H.P_val[k] / std::sqrt(diag * dc);
      if (!got.count(col) || std::fabs(got[col] - want) > 2e-6 * std::fabs(want) + 1e-12) fail("packets: scaled off-diagonal value");
    }
    if (off != (int) got.size()) fail("packets: number of off-diagonals");
  }
  std::printf("ok packets N=%d vpt=%d bandwidth=%d\n", H.N, P.vpt, P.bandwidth);
}

static void check_windows(const HostSystem &H, size_t budget, int expect_min_windows) {
  HostWindows W;
  if (!W.build(H, budget)) fail("windows: build refused");
  if (W.lds_bytes > budget || W.nwin < expect_min_windows) fail("windows: LDS budget / window count");
  const int N = H.N, T = H.T, E = H.E;
  // random per-element result vectors (two per triangle, one per flap), one scalar each is enough for the bookkeeping
  std::mt19937 rng(3);
  std::uniform_real_distribution<double> u(-1, 1);
  std::vector<double> r0(T), r1(T), rb(E);
  for (double &x : r0) x = u(rng);
  for (double &x : r1) x = u(rng);
  for (double &x : rb) x = u(rng);
  // reference: corner sums in HostSystem::inc_idx order
  std::vector<double> want(N, 0.0);
  for (int v = 0; v < N; v++)
    for (int k = H.inc_ptr[v]; k < H.inc_ptr[v + 1]; k++) {
      const int idx = H.inc_idx[k];
      if (idx < 3 * T) {
        const int c = idx / T, t = idx % T;
        const float Dx = (float) H.tri_D[4 * t], Dy = (float) H.tri_D[4 * t + 1], Dz = (float) H.tri_D[4 * t + 2], Dw = (float) H.tri_D[4 * t + 3];
        const double c1 = r0[t] * Dx + r1[t] * Dy, c2 = r0[t] * Dz + r1[t] * Dw;
        want[v] += c == 1 ? c1 : (c == 2 ? c2 : -c1 - c2);
      } else {
        const int c = (idx - 3 * T) / E, e = (idx - 3 * T) % E;
        want[v] += rb[e] * (float) H.bend_w[4 * e + c];
      }
    }
  std::vector<int> owner(N, 0);
  std::vector<double> got(N, 0.0);
  for (int w = 0; w < W.nwin; w++) {
    const int *d = &W.win[8 * w];
    const int v0 = d[0], v1 = d[1], lo = d[2], vs = d[3], toff = d[4], nt = d[5], boff = d[6], nb = d[7];
    if (v0 % 64 != 0 || vs > W.vcap || 2 * nt + nb + 1 > W.nrcap || lo > v0 || lo + vs < v1) fail("windows: descriptor");
    std::vector<double> er(2 * nt + nb + 1, 0.0);      // + the zero vector of the padding entries
    for (int k = 0; k < nt; k++) {
      const int *r = &W.tri_rec[4 * (size_t) (toff + k)];
      const int t = r[3], j[3] = {r[0] & 0xffff, (int) ((unsigned) r[0] >> 16), r[1]};
      for (int q = 0; q < 3; q++) if (lo + j[q] != H.tri[3 * t + q]) fail("windows: triangle record vertices");
      if (asfloat(r[2]) != (float) H.tri_w2[t]) fail("windows: triangle weight");
      for (int q = 0; q < 4; q++) if (W.tri_D[4 * (size_t) (toff + k) + q] != (float) H.tri_D[4 * t + q]) fail("windows: triangle D");
      {   // two result planes: the triangle's contributions to its corners 1 and 2 (what the device stores)
        const float Dx = (float) H.tri_D[4 * t], Dy = (float) H.tri_D[4 * t + 1], Dz = (float) H.tri_D[4 * t + 2], Dw = (float) H.tri_D[4 * t + 3];
        er[k] = r0[t] * Dx + r1[t] * Dy; er[nt + k] = r0[t] * Dz + r1[t] * Dw;
      }
    }
    // flaps are identified by their vertices
    std::map<std::vector<int>, int> flap;
    for (int e = 0; e < E; e++) flap[{H.bend_v[4 * e], H.bend_v[4 * e + 1], H.bend_v[4 * e + 2], H.bend_v[4 * e + 3]}] = e;
    for (int k = 0; k < nb; k++) {
      const int *r = &W.bend_rec[4 * (size_t) (boff + k)];
      std::vector<int> vv = {lo + (r[0] & 0xffff), lo + (int) ((unsigned) r[0] >> 16), lo + (r[1] & 0xffff), lo + (int) ((unsigned) r[1] >> 16)};
      if (!flap.count(vv)) fail("windows: flap record vertices");
      const int e = flap[vv];
      if (asfloat(r[2]) != (float) H.bend_n[e] || asfloat(r[3]) != (float) H.bend_w2[e]) fail("windows: flap rest data");
      er[2 * nt + k] = rb[e];
    }
    for (int v = v0; v < v1; v++) {
      owner[v]++;
      const int ch = v / 64, l = v % 64;
      double s = 0;
      const int nt4 = W.inc_n[ch] >> 16, nb4 = W.inc_n[ch] & 0xffff;
      if (nt4 < 1 || nb4 < 1) fail("windows: packet counts of a chunk");
      for (int pk = 0; pk < nt4; pk++) {          // triangle entries: position << 1 | minus, 16 bits each
        const int *q = &W.inc[4 * ((size_t) W.inc_ptr[ch] + (size_t) pk * 64 + l)];
        for (int e8 = 0; e8 < 8; e8++) {
          const int code = (e8 & 1) ? (int) ((unsigned) q[e8 >> 1] >> 16) : (q[e8 >> 1] & 0xffff);
          const int pos = code >> 1;
          if (pos < 0 || pos > 2 * nt + nb) fail("windows: triangle entry outside the window's result vectors");
          if (pos >= 2 * nt && pos != 2 * nt + nb) fail("windows: triangle entry points at a flap vector");
          s += (code & 1) ? -er[pos] : er[pos];
        }
      }
      for (int pk = 0; pk < nb4; pk++) {          // flap pairs
        const int *q = &W.inc[4 * ((size_t) W.inc_ptr[ch] + (size_t) (nt4 + pk) * 64 + l)];
        for (int h = 0; h < 2; h++) {
          const float coef = asfloat(q[2 * h + 1]);
          if (q[2 * h] < 2 * nt || q[2 * h] > 2 * nt + nb) fail("windows: flap entry outside the window's flap vectors");
          s += coef * er[q[2 * h]];
        }
      }
      got[v] = s;
    }
  }
  for (int v = 0; v < N; v++) {
    if (owner[v] != 1) fail("windows: every vertex must be owned by exactly one window");
    if (std::fabs(got[v] - want[v]) > 1e-5 * (1 + std::fabs(want[v]))) fail("windows: incidence sums differ from the corner sums");   // fp32 coefficients
  }
  std::printf("ok windows N=%d nwin=%d own=%d vcap=%d nrcap=%d lds=%zu\n", N, W.nwin, W.own, W.vcap, W.nrcap, W.lds_bytes);
}

// On a regular mesh the first-use element order makes slot j of consecutive owned vertices consecutive LDS positions (the per-vertex
// phase then reads its result vectors without bank conflicts): fraction of neighbouring-lane pairs whose slot positions differ by 1.
static double regular_slot_fraction(const HostWindows &W) {
  long good = 0, all = 0;
  for (int w = 0; w < W.nwin; w++) {
    const int *d = &W.win[8 * w];
    const int v0 = d[0], v1 = d[1], nt = d[5], nb = d[7];
    for (int v = v0; v + 1 < v1; v++) {
      if (v / 64 != (v + 1) / 64) continue;
      const int ch = v / 64, l = v % 64, nt4 = W.inc_n[ch] >> 16, nb4 = W.inc_n[ch] & 0xffff;
      for (int pk = 0; pk < nt4; pk++)
        for (int e8 = 0; e8 < 8; e8++) {
          auto code = [&](int lane) { const int *q = &W.inc[4 * ((size_t) W.inc_ptr[ch] + (size_t) pk * 64 + lane)]; return (e8 & 1) ? (int) ((unsigned) q[e8 >> 1] >> 16) : (q[e8 >> 1] & 0xffff); };
          const int a = code(l) >> 1, b = code(l + 1) >> 1;
          if (a == 2 * nt + nb || b == 2 * nt + nb) continue;
          all++; good += (b - a == 1);
        }
      for (int pk = 0; pk < nb4; pk++)
        for (int h = 0; h < 2; h++) {
          const int a = W.inc[4 * ((size_t) W.inc_ptr[ch] + (size_t) (nt4 + pk) * 64 + l) + 2 * h], b = W.inc[4 * ((size_t) W.inc_ptr[ch] + (size_t) (nt4 + pk) * 64 + l + 1) + 2 * h];
          if (a == 2 * nt + nb || b == 2 * nt + nb) continue;
          all++; good += (b - a == 1);
        }
    }
  }
  return all ? (double) good / (double) all : 0.0;
}

int main() {
  std::vector<double> pos;
  std::vector<int> tri;
  // 1. a 60 x 40 grid in natural numbering: packets + several windows under a small LDS budget
  grid(60, 40, false, pos, tri);
  HostSystem H;
  if (!H.set_mesh(60 * 40, pos.data(), (int) tri.size() / 3, tri.data())) fail("set_mesh");
  if (!H.build_numerics(1.0 / 120, 0.3, 200.0, 0.02, 1e4)) fail("build_numerics");
  check_packets(H);
  check_windows(H, 150 * 1024, 1);
  check_windows(H, 40 * 1024, 3);
  {
    HostWindows W;
    if (!W.build(H, 150 * 1024)) fail("windows: build");
    const double frac = regular_slot_fraction(W);
    std::printf("first-use order on the 60 x 40 grid: %.2f of the neighbouring-lane slot pairs read consecutive positions\n", frac);
    if (frac < 0.7) fail("windows: the first-use element order does not line the slots of consecutive vertices up");
  }
  // 2. the same grid with shuffled numbering: bandwidth ~N; RCM brings it back under the packet limit
  grid(60, 40, true, pos, tri);
  const int bw0 = mesh_bandwidth((int) tri.size() / 3, tri.data());
  std::vector<int> order = rcm_order(60 * 40, (int) tri.size() / 3, tri.data());
  std::vector<int> inv(order.size(), -1);
  for (size_t k = 0; k < order.size(); k++) { if (order[k] < 0 || order[k] >= (int) order.size() || inv[order[k]] != -1) fail("rcm: not a permutation"); inv[order[k]] = (int) k; }
  std::vector<int> tri2(tri.size());
  for (size_t k = 0; k < tri.size(); k++) tri2[k] = inv[tri[k]];
  const int bw1 = mesh_bandwidth((int) tri2.size() / 3, tri2.data());
  if (!(bw0 > 1500 && bw1 < 120)) fail("rcm: bandwidth not reduced");
  std::vector<double> pos2(pos.size());
  for (size_t k = 0; k < order.size(); k++) for (int d = 0; d < 3; d++) pos2[3 * k + d] = pos[3 * (size_t) order[k] + d];
  HostSystem H2;
  if (!H2.set_mesh(60 * 40, pos2.data(), (int) tri2.size() / 3, tri2.data())) fail("set_mesh (renumbered)");
  if (!H2.build_numerics(1.0 / 120, 0.3, 200.0, 0.02, 1e4)) fail("build_numerics (renumbered)");
  check_packets(H2);
  check_windows(H2, 60 * 1024, 2);
  std::printf("ok rcm bandwidth %d -> %d\n", bw0, bw1);
  // 3. the shuffled numbering itself is refused by the packet builder (bandwidth > 511) - the engine falls back
  HostSystem H3;
  if (!H3.set_mesh(60 * 40, pos.data(), (int) tri.size() / 3, tri.data()) || !H3.build_numerics(1.0 / 120, 0.3, 200.0, 0.02, 1e4)) fail("shuffled system");
  HostPackets P3;
  if (P3.build(H3)) fail("packets: a bandwidth > 511 must be refused");
  std::printf("ok refusal bandwidth=%d\n", P3.bandwidth);
  // 4. explicit inverse of the scaled matrix of a small mesh (dc_dense.h): symmetric, zero padded, Ahat * inv = I to fp32
  {
    grid(24, 20, false, pos, tri);
    HostSystem H4;
    if (!H4.set_mesh(24 * 20, pos.data(), (int) tri.size() / 3, tri.data()) || !H4.build_numerics(1.0 / 100, 0.224, 1200.0, 120.0, 1e4)) fail("small system");
    HostDense D;
    if (!D.build(H4, 768)) fail("dense: build refused");
    const int n = H4.N;
    if (D.n != n || D.ld != 512 || D.rows != n + kDensePadRows || D.inv.size() != (size_t) D.rows * D.ld) fail("dense: layout");
    for (int i = 0; i < D.rows; i++)
      for (int j = 0; j < D.ld; j++) {
        const float v = D.inv[(size_t) i * D.ld + j];
        if ((i >= n || j >= n) && v != 0.f) fail("dense: padding must be zero");
        if (i < n && j < n && v != D.inv[(size_t) j * D.ld + i]) fail("dense: not symmetric");
      }
    std::vector<double> s(n);
    for (int i = 0; i < n; i++)
      for (int k = H4.P_ptr[i]; k < H4.P_ptr[i + 1]; k++) if (H4.P_col[k] == i) s[i] = std::sqrt(1.0 / H4.P_val[k]);
    double worst = 0;
    for (int c = 0; c < n; c += 37) {           // columns of Ahat * inv against the identity
      for (int i = 0; i < n; i++) {
        double acc = 0;
        for (int k = H4.P_ptr[i]; k < H4.P_ptr[i + 1]; k++) acc += H4.P_val[k] * s[i] * s[H4.P_col[k]] * (double) D.inv[(size_t) H4.P_col[k] * D.ld + c];
        worst = std::max(worst, std::fabs(acc - (i == c ? 1.0 : 0.0)));
      }
    }
    std::printf("dense defect=%.2e worst=%.2e\n", D.defect, worst);
    if (!(worst < 2e-3) || !(D.defect < 2e-3)) fail("dense: not an inverse");
    HostDense D2;
    if (D2.build(H4, 100) || D2.ok) fail("dense: size limit ignored");
    std::printf("ok dense n=%d ld=%d defect=%.2e max|Ahat inv - I|=%.2e\n", n, D.ld, D.defect, worst);
  }
  std::printf("ALL OK\n");
  return 0;
}
