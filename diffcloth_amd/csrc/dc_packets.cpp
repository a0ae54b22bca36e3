#include "dc_packets.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace dc {

static const bool kDefault768 = false;      // (measured default)

bool HostPackets::build(const HostSystem &H) {
  static const int allowed[] = {1, 2, 3, 4, 6, 8, 10, 12, 16, 20};      // instantiated rows-per-thread of k_pd_step_pk
  const int need = (H.N + 511) / 512;
  int v = 0;
  for (int a : allowed) if (a >= need) { v = a; break; }
  if (v == 0) {          // too large for the one-workgroup kernel: report the bandwidth, no tables
    *this = HostPackets();
    for (int r = 0; r < H.N; r++)
      for (int k = H.P_ptr[r]; k < H.P_ptr[r + 1]; k++) bandwidth = std::max(bandwidth, std::abs(H.P_col[k] - r));
    return false;
  }
  // meshes of 9 217 ... 10 752 rows: 768 threads x 14 rows (12 waves = 3 per SIMD at 168 registers) instead of 512 x 20 (2 per SIMD):
  // the resident PCG and the element windows are latency-bound, a third wave per SIMD hides more of it (DC_PK_THREADS=512 / 768 forces)
  static const char *envt = getenv("DC_PK_THREADS");
  const bool want768 = envt ? atoi(envt) == 768 : kDefault768;
  if (want768 && H.N > 768 * 12 && H.N <= 768 * 14) {
    if (!build_rows(H, 768 * 14)) return false;
    vpt = 14; threads = 768;
    return true;
  }
  if (!build_rows(H, 512 * v)) return false;
  vpt = v;
  return true;
}

bool HostPackets::build_rows(const HostSystem &H, int rows_padded) {
  *this = HostPackets();
  const int N = H.N;
  for (int r = 0; r < N; r++)
    for (int k = H.P_ptr[r]; k < H.P_ptr[r + 1]; k++) bandwidth = std::max(bandwidth, std::abs(H.P_col[k] - r));
  if (bandwidth > 511 || rows_padded < N || rows_padded % 64 != 0) return false;
  const int NPk = rows_padded, nch = NPk / 64, PBk = 4;
  std::vector<double> sq(N);
  sq_dinv.assign(NPk, 0.f);
  for (int i = 0; i < N; i++) {
    double d = 0;
    for (int k = H.P_ptr[i]; k < H.P_ptr[i + 1]; k++) if (H.P_col[k] == i) d = H.P_val[k];
    const float dinv = (float) (1.0 / d);              // the fp32 preconditioner entry the other kernels use
    sq[i] = std::sqrt((double) dinv);
    sq_dinv[i] = (float) sq[i];
  }
  pk_ptr.assign(nch, 0); pk_n.assign(nch, 0);
  for (int ch = 0; ch < nch; ch++) {
    int w = 0;
    for (int r = 64 * ch; r < std::min(N, 64 * ch + 64); r++) w = std::max(w, H.P_ptr[r + 1] - H.P_ptr[r] - 1);
    const int np = std::max(PBk, ((w + 2) / 3 + PBk - 1) / PBk * PBk);
    pk_ptr[ch] = (int) (pk.size() / 4); pk_n[ch] = np;
    pk.resize(pk.size() + (size_t) 4 * 64 * np, 0);
    for (int l = 0; l < 64; l++) {
      const int r = 64 * ch + l;
      int kk = r < N ? H.P_ptr[r] : 0;
      const int kend = r < N ? H.P_ptr[r + 1] : 0;
      for (int s = 0; s < np; s++) {
        int bits[3] = {0, 0, 0}, wd = 0;
        for (int q = 0; q < 3; q++) {
          int d = 512;
          while (kk < kend && H.P_col[kk] == r) kk++;          // the diagonal is implicit (= 1 after scaling)
          if (kk < kend) {
            const int col = H.P_col[kk];
            const float v = (float) (H.P_val[kk] * sq[r] * sq[col]);
            std::memcpy(&bits[q], &v, sizeof(int));
            d = col - r + 512;
            kk++;
          }
          wd |= d << (10 * q);
        }
        const size_t o = 4 * ((size_t) pk_ptr[ch] + (size_t) s * 64 + l);
        pk[o] = bits[0]; pk[o + 1] = bits[1]; pk[o + 2] = bits[2]; pk[o + 3] = wd;
      }
    }
  }
  ok = true;
  return true;
}

}  // namespace dc
