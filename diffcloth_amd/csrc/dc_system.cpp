// Host-side constraint-system builder. See dc_system.h for the reference lines each block reproduces.
#include "dc_system.h"
#include <algorithm>
#include <cmath>
#include <tuple>

namespace dc {

namespace {
struct P3 {
  double x, y, z;
};
inline P3 sub(const P3 &a, const P3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(const P3 &a, const P3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double len(const P3 &a) { return std::sqrt(dot(a, a)); }
inline P3 axpy(const P3 &a, double s, const P3 &b) { return {a.x + s * b.x, a.y + s * b.y, a.z + s * b.z}; }
inline P3 scale(const P3 &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline P3 at(const std::vector<double> &v, int i) { return {v[3 * i], v[3 * i + 1], v[3 * i + 2]}; }
inline double heron(double a, double b, double c) {
  double s = 0.5 * (a + b + c);
  return std::sqrt(s * (s - a) * (s - b) * (s - c));
}
}  // namespace

bool HostSystem::set_mesh(int n, const double *pos, int t, const int *tris) {
  if (n <= 0 || t <= 0 || !pos || !tris) { error = "set_mesh: empty mesh"; return false; }
  for (int k = 0; k < 3 * t; k++)
    if (tris[k] < 0 || tris[k] >= n) { error = "set_mesh: triangle index out of range"; return false; }
  N = n; T = t;
  rest.assign(pos, pos + 3 * (size_t) n);
  tri.assign(tris, tris + 3 * (size_t) t);

  // --- per-triangle material frame: inverse of the 2x2 rest edge matrix expressed in an orthonormal frame ---
  tri_D.assign(4 * (size_t) T, 0.0);
  tri_area.assign(T, 0.0);
  area.assign(N, 0.0);
  for (int f = 0; f < T; f++) {
    P3 p0 = at(rest, tri[3 * f]), p1 = at(rest, tri[3 * f + 1]), p2 = at(rest, tri[3 * f + 2]);
    P3 e0 = sub(p1, p0), e1 = sub(p2, p0);
    double l0 = len(e0);
    if (!(l0 > 0)) { error = "set_mesh: degenerate triangle"; return false; }
    P3 u = scale(e0, 1.0 / l0);
    P3 w = axpy(e1, -dot(e1, u), u);
    double lw = len(w);
    if (!(lw > 0)) { error = "set_mesh: degenerate triangle"; return false; }
    w = scale(w, 1.0 / lw);
    // rest edge matrix in the (u,w) frame: [[u.e0, u.e1],[w.e0, w.e1]]
    double m00 = dot(u, e0), m01 = dot(u, e1), m10 = dot(w, e0), m11 = dot(w, e1);
    double det = m00 * m11 - m01 * m10;
    tri_D[4 * f + 0] = m11 / det;
    tri_D[4 * f + 1] = -m01 / det;
    tri_D[4 * f + 2] = -m10 / det;
    tri_D[4 * f + 3] = m00 / det;
    tri_area[f] = std::fabs(0.5 * det);
    for (int k = 0; k < 3; k++) area[tri[3 * f + k]] += tri_area[f] / 3.0;
  }

  // --- bending flaps: one per interior edge, ordered by (min,max) vertex key; the two opposite vertices keep
  //     the order in which their triangles appear in the mesh ---
  std::vector<std::tuple<int, int, int, int>> half;   // (min, max, triangle, opposite)
  half.reserve(3 * (size_t) T);
  for (int f = 0; f < T; f++) {
    const int *v = &tri[3 * f];
    const int pairs[3][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}};   // (v1Idx, v2Idx, other) in the reference's loop order
    for (auto &pr : pairs) half.emplace_back(std::min(v[pr[0]], v[pr[1]]), std::max(v[pr[0]], v[pr[1]]), f, v[pr[2]]);
  }
  std::stable_sort(half.begin(), half.end(), [](const auto &a, const auto &b) {
    return std::make_pair(std::get<0>(a), std::get<1>(a)) < std::make_pair(std::get<0>(b), std::get<1>(b));
  });
  bend_v.clear(); bend_w.clear(); bend_n.clear(); bend_A.clear();
  for (size_t k = 0; k < half.size();) {
    size_t j = k + 1;
    while (j < half.size() && std::get<0>(half[j]) == std::get<0>(half[k]) && std::get<1>(half[j]) == std::get<1>(half[k])) j++;
    if (j - k > 2) { error = "set_mesh: non-manifold edge (shared by more than two triangles)"; return false; }
    if (j - k == 2) {
      int q[4] = {std::get<0>(half[k]), std::get<1>(half[k]), std::get<3>(half[k]), std::get<3>(half[k + 1])};
      P3 p[4];
      for (int i = 0; i < 4; i++) p[i] = at(rest, q[i]);
      double l01 = len(sub(p[1], p[0])), l02 = len(sub(p[2], p[0])), l03 = len(sub(p[3], p[0]));
      double l12 = len(sub(p[1], p[2])), l13 = len(sub(p[1], p[3]));
      double a0 = heron(l01, l02, l12), a1 = heron(l01, l13, l03);
      double c02 = (l01 * l01 - l02 * l02 + l12 * l12) / (4.0 * a0);
      double c12 = (l01 * l01 + l02 * l02 - l12 * l12) / (4.0 * a0);
      double c03 = (l01 * l01 - l03 * l03 + l13 * l13) / (4.0 * a1);
      double c13 = (l01 * l01 + l03 * l03 - l13 * l13) / (4.0 * a1);
      double wv[4] = {c02 + c03, c12 + c13, -(c02 + c12), -(c03 + c13)};
      P3 e = {0, 0, 0};
      for (int i = 0; i < 4; i++) e = axpy(e, wv[i], p[i]);
      for (int i = 0; i < 4; i++) { bend_v.push_back(q[i]); bend_w.push_back(wv[i]); }
      bend_n.push_back(len(e));
      bend_A.push_back(a0 + a1);
    }
    k = j;
  }
  E = (int) bend_n.size();

  // --- vertex -> constraint-corner incidence (gather lists of the RHS assembly) ---
  std::vector<int> cnt(N + 1, 0);
  for (int f = 0; f < T; f++) for (int k = 0; k < 3; k++) cnt[tri[3 * f + k] + 1]++;
  for (int e = 0; e < E; e++) for (int k = 0; k < 4; k++) cnt[bend_v[4 * e + k] + 1]++;
  inc_ptr.assign(N + 1, 0);
  for (int i = 0; i < N; i++) inc_ptr[i + 1] = inc_ptr[i] + cnt[i + 1];
  inc_idx.assign(inc_ptr[N], 0);
  std::vector<int> fill(inc_ptr.begin(), inc_ptr.end() - 1);
  for (int f = 0; f < T; f++) for (int k = 0; k < 3; k++) inc_idx[fill[tri[3 * f + k]]++] = k * T + f;
  for (int e = 0; e < E; e++) for (int k = 0; k < 4; k++) inc_idx[fill[bend_v[4 * e + k]]++] = 3 * T + k * E + e;

  // --- collision radii: half the shortest incident rest edge minus 0.01; connected-pair table ---
  radii.assign(N, 100.0 / 2.0 - 0.01);
  std::vector<double> min_edge(N, 100.0);
  std::vector<std::pair<int, int>> pairs;
  pairs.reserve(9 * (size_t) T);
  for (int f = 0; f < T; f++) {
    const int *v = &tri[3 * f];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        pairs.emplace_back(v[a], v[b]);
        if (a != b) min_edge[v[a]] = std::min(min_edge[v[a]], len(sub(at(rest, v[a]), at(rest, v[b]))));
      }
  }
  for (int i = 0; i < N; i++) radii[i] = min_edge[i] / 2.0 - 0.01;
  std::sort(pairs.begin(), pairs.end());
  pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
  conn_ptr.assign(N + 1, 0);
  for (auto &pr : pairs) conn_ptr[pr.first + 1]++;
  for (int i = 0; i < N; i++) conn_ptr[i + 1] += conn_ptr[i];
  conn_idx.resize(pairs.size());
  for (size_t k = 0; k < pairs.size(); k++) conn_idx[k] = pairs[k].second;
  return true;
}

bool HostSystem::build_numerics(double h, double density, double k_stretch, double k_bend, double k_att) {
  if (N == 0) { error = "build: no mesh"; return false; }
  for (int a : att_vertex)
    if (a < 0 || a >= N) { error = "build: attachment vertex out of range"; return false; }
  mass.resize(N);
  for (int i = 0; i < N; i++) mass[i] = density * area[i];
  tri_w2.resize(T);
  for (int f = 0; f < T; f++) tri_w2[f] = tri_area[f] * k_stretch;
  bend_w2.resize(E);
  for (int e = 0; e < E; e++) bend_w2[e] = k_bend * 3.0 / bend_A[e];

  // scalar P_s = diag(mass) + h^2 * sum_rows w^2 c c^T, accumulated as (row, col, value) triples
  struct Ent { int r, c; double v; };
  std::vector<Ent> ent;
  ent.reserve(18 * (size_t) T + 16 * (size_t) E + N + att_vertex.size());
  const double h2 = h * h;
  for (int i = 0; i < N; i++) ent.push_back({i, i, mass[i]});
  for (int f = 0; f < T; f++) {
    const int *v = &tri[3 * f];
    const double *D = &tri_D[4 * f];
    for (int col = 0; col < 2; col++) {   // the two columns of the deformation gradient = two scalar rows
      double c[3] = {-(D[0 * 2 + col] + D[1 * 2 + col]), D[0 * 2 + col], D[1 * 2 + col]};
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) ent.push_back({v[a], v[b], h2 * tri_w2[f] * c[a] * c[b]});
    }
  }
  for (int e = 0; e < E; e++)
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++)
      ent.push_back({bend_v[4 * e + a], bend_v[4 * e + b], h2 * bend_w2[e] * bend_w[4 * e + a] * bend_w[4 * e + b]});
  for (int a : att_vertex) ent.push_back({a, a, h2 * k_att});
  std::sort(ent.begin(), ent.end(), [](const Ent &a, const Ent &b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
  P_ptr.assign(N + 1, 0); P_col.clear(); P_val.clear();
  for (size_t k = 0; k < ent.size();) {
    size_t j = k;
    double s = 0;
    while (j < ent.size() && ent[j].r == ent[k].r && ent[j].c == ent[k].c) s += ent[j++].v;
    P_col.push_back(ent[k].c);
    P_val.push_back(s);
    P_ptr[ent[k].r + 1]++;
    k = j;
  }
  for (int i = 0; i < N; i++) P_ptr[i + 1] += P_ptr[i];
  return true;
}

int mesh_bandwidth(int t, const int *tris) {
  int bw = 0;
  for (int k = 0; k < t; k++)
    for (int a = 0; a < 3; a++) bw = std::max(bw, std::abs(tris[3 * k + a] - tris[3 * k + (a + 1) % 3]));
  return bw;
}

std::vector<int> rcm_order(int n, int t, const int *tris) {
  std::vector<std::vector<int>> adj(n);
  for (int k = 0; k < t; k++)
    for (int a = 0; a < 3; a++) {
      const int u = tris[3 * k + a], v = tris[3 * k + (a + 1) % 3];
      adj[u].push_back(v); adj[v].push_back(u);
    }
  for (auto &a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
  auto by_degree = [&](int a, int b) { return adj[a].size() != adj[b].size() ? adj[a].size() < adj[b].size() : a < b; };
  for (auto &a : adj) std::sort(a.begin(), a.end(), by_degree);
  std::vector<int> order, level(n, -1);
  std::vector<char> done(n, 0);
  order.reserve(n);
  // breadth-first levels from `root` over the not-yet-ordered vertices; returns the last vertex reached
  auto bfs_far = [&](int root, int &depth) {
    std::vector<int> q{root};
    std::vector<int> seen{root};
    level[root] = 0;
    size_t head = 0;
    int last = root;
    while (head < q.size()) {
      const int u = q[head++];
      last = u;
      for (int v : adj[u]) if (!done[v] && level[v] < 0) { level[v] = level[u] + 1; q.push_back(v); seen.push_back(v); }
    }
    depth = level[last];
    // among the deepest level prefer the smallest degree (George-Liu pseudo-peripheral heuristic)
    for (int v : seen) if (level[v] == depth && by_degree(v, last)) last = v;
    for (int v : seen) level[v] = -1;
    return last;
  };
  for (int seed = 0; seed < n; seed++) {
    if (done[seed]) continue;
    int root = seed, depth = -1;
    for (int it = 0; it < 8; it++) {           // walk to a pseudo-peripheral vertex of this component
      int d;
      const int far = bfs_far(root, d);
      if (d <= depth) break;
      depth = d; root = far;
    }
    std::vector<int> q{root};
    done[root] = 1;
    size_t head = 0;
    while (head < q.size()) {
      const int u = q[head++];
      order.push_back(u);
      for (int v : adj[u]) if (!done[v]) { done[v] = 1; q.push_back(v); }
    }
  }
  std::reverse(order.begin(), order.end());
  return order;
}

}  // namespace dc
