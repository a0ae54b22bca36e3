#include "dc_windows.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace dc {

namespace {

inline int fbits(float v) { int b; std::memcpy(&b, &v, sizeof(int)); return b; }

struct Plan { int nwin = 0, vcap = 0, nrcap = 0; };

// window w of `own` vertices: which elements touch it and which vertex span they cover
struct WinScan {
  std::vector<int> tris, bends;
  int lo = 0, hi = 0;
};

void scan_window(const HostSystem &H, int v0, int v1, WinScan &s) {
  s.tris.clear(); s.bends.clear();
  s.lo = v0; s.hi = v1;
  auto in = [&](int v) { return v >= v0 && v < v1; };
  for (int t = 0; t < H.T; t++) {
    const int *q = &H.tri[3 * t];
    if (in(q[0]) || in(q[1]) || in(q[2])) {
      s.tris.push_back(t);
      for (int k = 0; k < 3; k++) { s.lo = std::min(s.lo, q[k]); s.hi = std::max(s.hi, q[k] + 1); }
    }
  }
  for (int e = 0; e < H.E; e++) {
    const int *q = &H.bend_v[4 * e];
    if (in(q[0]) || in(q[1]) || in(q[2]) || in(q[3])) {
      s.bends.push_back(e);
      for (int k = 0; k < 4; k++) { s.lo = std::min(s.lo, q[k]); s.hi = std::max(s.hi, q[k] + 1); }
    }
  }
}

// Order of the elements inside a window. Their result vectors sit in LDS at the element's position, and in the per-vertex phase lane
// l (vertex v0 + l) reads, in its slot j, the result of its j-th incident element: with the elements in mesh order a regular mesh
// turns that into a strided access (the two triangles of a quad cell alternate: slot j of consecutive vertices is 2 elements
// = 4 result vectors apart, a 4-way LDS bank conflict on every read of the phase). Placing the elements in the order in which a
// slot-major sweep over the owned vertices first meets them makes slot j of consecutive vertices consecutive positions wherever
// the mesh is regular, and costs nothing where it is not. (Only positions change: the sums keep their order, results are bit-identical.)
void order_by_first_use(const HostSystem &H, int v0, int v1, bool bends, std::vector<int> &list) {
  const int T = H.T, E = H.E;
  std::vector<char> placed(bends ? E : T, 0), member(bends ? E : T, 0);
  for (int id : list) member[id] = 1;
  std::vector<int> out;
  out.reserve(list.size());
  std::vector<std::vector<int>> inc(v1 - v0);
  size_t depth = 0;
  for (int v = v0; v < v1; v++) {
    for (int k = H.inc_ptr[v]; k < H.inc_ptr[v + 1]; k++) {
      const int idx = H.inc_idx[k];
      const bool is_bend = idx >= 3 * T;
      if (is_bend != bends) continue;
      const int id = is_bend ? (idx - 3 * T) % E : idx % T;
      if (member[id]) inc[v - v0].push_back(id);
    }
    depth = std::max(depth, inc[v - v0].size());
  }
  for (size_t j = 0; j < depth; j++)
    for (int v = v0; v < v1; v++)
      if (j < inc[v - v0].size()) {
        const int id = inc[v - v0][j];
        if (!placed[id]) { placed[id] = 1; out.push_back(id); }
      }
  for (int id : list)
    if (!placed[id]) out.push_back(id);
  list.swap(out);
}

// sizing pass for one window size
Plan plan_for(const HostSystem &H, int own, const std::vector<int> &emin, const std::vector<int> &emax) {
  Plan p;
  p.nwin = (H.N + own - 1) / own;
  for (int w = 0; w < p.nwin; w++) {
    const int v0 = w * own, v1 = std::min(H.N, v0 + own);
    int lo = v0, hi = v1, nt = 0, nb = 0;
    auto in = [&](int v) { return v >= v0 && v < v1; };
    for (int t = 0; t < H.T; t++) {
      if (emax[t] < v0 || emin[t] >= v1) continue;
      const int *q = &H.tri[3 * t];
      if (in(q[0]) || in(q[1]) || in(q[2])) { nt++; lo = std::min(lo, emin[t]); hi = std::max(hi, emax[t] + 1); }
    }
    for (int e = 0; e < H.E; e++) {
      const int id = H.T + e;
      if (emax[id] < v0 || emin[id] >= v1) continue;
      const int *q = &H.bend_v[4 * e];
      if (in(q[0]) || in(q[1]) || in(q[2]) || in(q[3])) { nb++; lo = std::min(lo, emin[id]); hi = std::max(hi, emax[id] + 1); }
    }
    p.vcap = std::max(p.vcap, hi - lo);
    // + the zero vector the padding entries of the incidence rows point at + one dump slot per lane for the masked elements of the
    // last round (dc_winlib.h: their stores are redirected, not skipped)
    p.nrcap = std::max(p.nrcap, 2 * nt + nb + 1 + kWinDumpSlots);
  }
  return p;
}

}  // namespace

bool HostWindows::build(const HostSystem &H, size_t lds_budget) {
  *this = HostWindows();
  const int N = H.N, T = H.T, E = H.E;
  if (N <= 0 || T <= 0) return false;
  std::vector<int> emin(T + E), emax(T + E);
  for (int t = 0; t < T; t++) {
    emin[t] = std::min({H.tri[3 * t], H.tri[3 * t + 1], H.tri[3 * t + 2]});
    emax[t] = std::max({H.tri[3 * t], H.tri[3 * t + 1], H.tri[3 * t + 2]});
  }
  for (int e = 0; e < E; e++) {
    const int *q = &H.bend_v[4 * e];
    emin[T + e] = std::min({q[0], q[1], q[2], q[3]});
    emax[T + e] = std::max({q[0], q[1], q[2], q[3]});
  }
  // the fewest windows (fewest barriers) whose LDS footprint fits, balanced; local ids are 16-bit
  const int kmax = (N + 63) / 64;
  Plan best;
  int best_own = 0;
  for (int nw = 1; nw <= std::min(kmax, 64) && best_own == 0; nw++) {
    const int k = (kmax + nw - 1) / nw;
    if (nw > 1 && (kmax + nw - 2) / (nw - 1) == k) continue;      // same window size as the previous candidate
    Plan p = plan_for(H, 64 * k, emin, emax);
    const size_t bytes = sizeof(float) * ((size_t) 6 * p.vcap + (size_t) 3 * p.nrcap);
    if (bytes <= lds_budget && p.vcap <= 65535 && p.nrcap <= 32767) { best = p; best_own = 64 * k; }
  }
  if (best_own == 0) return false;
  // the per-vertex phase runs one owned vertex per thread and round: a window of a whole number of 1024-vertex rounds
  // leaves no mostly-idle last round (1152 owned vertices = 2 rounds for 1024 threads, 3 for 512)
  if (best.nwin > 1 && best_own > 1024 && best_own % 1024 != 0) {
    const int rounded = best_own / 1024 * 1024;
    Plan p = plan_for(H, rounded, emin, emax);
    best = p; best_own = rounded;
  }
  return build_own(H, best_own);
}

bool HostWindows::build_own(const HostSystem &H, int own_size) {
  *this = HostWindows();
  const int N = H.N, T = H.T, E = H.E;
  if (N <= 0 || T <= 0 || own_size <= 0 || own_size % 64 != 0) return false;
  std::vector<int> emin(T + E), emax(T + E);
  for (int t = 0; t < T; t++) {
    emin[t] = std::min({H.tri[3 * t], H.tri[3 * t + 1], H.tri[3 * t + 2]});
    emax[t] = std::max({H.tri[3 * t], H.tri[3 * t + 1], H.tri[3 * t + 2]});
  }
  for (int e = 0; e < E; e++) {
    const int *q = &H.bend_v[4 * e];
    emin[T + e] = std::min({q[0], q[1], q[2], q[3]});
    emax[T + e] = std::max({q[0], q[1], q[2], q[3]});
  }
  const Plan best = plan_for(H, own_size, emin, emax);
  if (best.vcap > 65535 || best.nrcap > 32767) return false;      // positions travel as 15 bits + sign
  const int best_own = own_size;
  own = best_own; nwin = best.nwin; vcap = best.vcap; nrcap = best.nrcap;
  lds_bytes = sizeof(float) * ((size_t) 6 * vcap + (size_t) 3 * nrcap);

  const int nchunks = (N + 63) / 64;
  inc_ptr.assign(nchunks, 0); inc_n.assign(nchunks, 0);
  std::vector<int> tri_local(T), bend_local(E);
  WinScan s;
  for (int w = 0; w < nwin; w++) {
    const int v0 = w * own, v1 = std::min(N, v0 + own);
    scan_window(H, v0, v1, s);
    static const bool reorder = !(getenv("DC_WIN_ORDER") && getenv("DC_WIN_ORDER")[0] == '0');     // development switch
    if (reorder) {
      order_by_first_use(H, v0, v1, false, s.tris);
      order_by_first_use(H, v0, v1, true, s.bends);
    }
    const int ntri = (int) s.tris.size(), nbend = (int) s.bends.size();
    const int tri_off = (int) (tri_rec.size() / 4), bend_off = (int) (bend_rec.size() / 4);
    const int d[8] = {v0, v1, s.lo, s.hi - s.lo, tri_off, ntri, bend_off, nbend};
    win.insert(win.end(), d, d + 8);
    for (int k = 0; k < ntri; k++) {
      const int t = s.tris[k];
      tri_local[t] = k;
      const int j0 = H.tri[3 * t] - s.lo, j1 = H.tri[3 * t + 1] - s.lo, j2 = H.tri[3 * t + 2] - s.lo;
      const int r[4] = {j0 | (j1 << 16), j2, fbits((float) H.tri_w2[t]), t};
      tri_rec.insert(tri_rec.end(), r, r + 4);
      for (int q = 0; q < 4; q++) tri_D.push_back((float) H.tri_D[4 * t + q]);
      for (int q = 0; q < 4; q++) tri_Dlo.push_back((float) (H.tri_D[4 * t + q] - (double) (float) H.tri_D[4 * t + q]));
    }
    for (int k = 0; k < nbend; k++) {
      const int e = s.bends[k];
      bend_local[e] = k;
      const int *q = &H.bend_v[4 * e];
      const int r[4] = {(q[0] - s.lo) | ((q[1] - s.lo) << 16), (q[2] - s.lo) | ((q[3] - s.lo) << 16), fbits((float) H.bend_n[e]),
                        fbits((float) H.bend_w2[e])};
      bend_rec.insert(bend_rec.end(), r, r + 4);
      for (int c = 0; c < 4; c++) bend_w.push_back((float) H.bend_w[4 * e + c]);
      for (int c = 1; c < 4; c++) bend_lo.push_back((float) (H.bend_w[4 * e + c] - (double) (float) H.bend_w[4 * e + c]));
      bend_lo.push_back((float) (H.bend_n[e] - (double) (float) H.bend_n[e]));
    }
    // incidence rows of the owned vertices, in the corner order of HostSystem::inc_idx. A triangle's two result vectors are its
    // contributions to its corners 1 and 2 (the device multiplies the residual columns by inv_deltaUV in the per-element phase), corner 0
    // gets minus their sum: triangle entries are positions with a sign, 16 bits each; a flap's single vector needs its corner weight.
    const int zero_slot = 2 * ntri + nbend;
    for (int ch = v0 / 64; ch < (v1 + 63) / 64; ch++) {
      std::vector<std::vector<int>> codes(64);
      std::vector<std::vector<std::pair<int, float>>> rows(64);
      int wt = 0, wb = 0;
      for (int l = 0; l < 64; l++) {
        const int v = 64 * ch + l;
        if (v >= v1) break;
        for (int k = H.inc_ptr[v]; k < H.inc_ptr[v + 1]; k++) {
          const int idx = H.inc_idx[k];
          if (idx < 3 * T) {
            const int corner = idx / T, t = idx % T;
            const int p0 = tri_local[t], p1 = ntri + tri_local[t];
            if (corner == 1) codes[l].push_back(p0 << 1);
            else if (corner == 2) codes[l].push_back(p1 << 1);
            else { codes[l].push_back((p0 << 1) | 1); codes[l].push_back((p1 << 1) | 1); }
          } else {
            const int corner = (idx - 3 * T) / E, e = (idx - 3 * T) % E;
            rows[l].push_back({2 * ntri + bend_local[e], (float) H.bend_w[4 * e + corner]});
          }
        }
        wt = std::max(wt, (int) codes[l].size());
        wb = std::max(wb, (int) rows[l].size());
      }
      const int nt4 = std::max(1, (wt + 7) / 8);             // packets of 8 triangle entries
      const int nb4 = std::max(2, ((wb + 1) / 2 + 1) / 2 * 2);   // packets of 2 flap pairs, an even number of them
      inc_ptr[ch] = (int) (inc.size() / 4); inc_n[ch] = nb4 | (nt4 << 16);
      const size_t base = inc.size();
      inc.resize(base + (size_t) 4 * 64 * (nt4 + nb4), 0);
      for (int l = 0; l < 64; l++) {
        for (int k = 0; k < 8 * nt4; k++) {
          const int code = k < (int) codes[l].size() ? codes[l][k] : (zero_slot << 1);
          const size_t o = base + 4 * ((size_t) (k / 8) * 64 + l) + (k % 8) / 2;
          inc[o] |= (k % 2) ? (code << 16) : code;
        }
        for (size_t k = 0; k < rows[l].size(); k++) {
          const size_t o = base + 4 * ((size_t) (nt4 + k / 2) * 64 + l) + 2 * (k % 2);
          inc[o] = rows[l][k].first; inc[o + 1] = fbits(rows[l][k].second);
        }
        for (size_t k = rows[l].size(); k < (size_t) 2 * nb4; k++) {      // padding pairs: the zero vector, weight 0
          const size_t o = base + 4 * ((size_t) (nt4 + k / 2) * 64 + l) + 2 * (k % 2);
          inc[o] = zero_slot; inc[o + 1] = 0;
        }
      }
    }
  }
  ok = true;
  return true;
}

}  // namespace dc
