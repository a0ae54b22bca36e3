// Self-collision detection + contact layering of one rollout as a device function (used by the stand-alone k_self_detect
// launch and, inlined, by the fused multi-step forward kernel). Reference: Simulation::collisionDetection
// (Simulation.cpp:225-373, self part :281-352), isSelfCollision (:194-220), contactSorting (:422-624).
#pragma once
#include "dc_devlib.h"

namespace dc {


template <int THREADS>
__device__ __forceinline__ float block_max(float v, float *redf) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) redf[w] = v;
  __syncthreads();
  float s = redf[0];
#pragma unroll
  for (int k = 1; k < THREADS / 64; k++) s = fmaxf(s, redf[k]);
  return s;
}

// Simulation::isSelfCollision (Simulation.cpp:194-220); a.idx = ia, b.idx = ib. Returns the normal for id1 = min.
__device__ __forceinline__ bool self_collision(float thresh, f3 xa, f3 xb, f3 va, f3 vb, float h, int ia, int ib, f3 &normal) {
  f3 v0 = xa - xb, v = va - vb;
  f3 p1 = v0 + v * h;
  float minDist = fminf(sqrtf(dot(v0, v0)), sqrtf(dot(p1, p1)));
  float tMid = -2.f * dot(v, v0) / dot(v, v);
  if ((tMid >= 0.f) && (tMid <= h)) { f3 pm = v0 + v * tMid; minDist = fminf(minDist, sqrtf(dot(pm, pm))); }
  if (minDist < thresh) {
    normal = normalized(v0) * ((ia < ib) ? 1.f : -1.f);
    return true;
  }
  return false;
}

__device__ __forceinline__ bool connected(const DevSystem &S, int i, int j) {   // pointpointConnectionTable (Simulation.cpp:2236-2239)
  int lo = S.conn_ptr[i], hi = S.conn_ptr[i + 1];
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    int c = S.conn_idx[mid];
    if (c == j) return true;
    if (c < j) lo = mid + 1; else hi = mid;
  }
  return false;
}

// Simulation::contactSorting (Simulation.cpp:422-624), serial. Contacts are given sorted by (p1, p2).
// tmp layout (ints): ids[2C] | adj_ptr[2C+1] | adj_other[2C] | adj_contact[2C] | deg[2C] | frontier[2C] | newf[2C] |
//                    involved[2C] | layer[C] | alive[C]
__device__ void contact_sorting_serial(int C, const int2 *pair, const int *rec_prim, const int *dev_of, int *tmp, int cap, int *meta, int *layer_out) {
  int *ids = tmp, *adj_ptr = ids + 2 * cap, *adj_other = adj_ptr + 2 * cap + 1, *adj_contact = adj_other + 2 * cap;
  int *deg = adj_contact + 2 * cap, *frontier = deg + 2 * cap, *newf = frontier + 2 * cap, *involved = newf + 2 * cap;
  int *layer = involved + 2 * cap, *alive = layer + cap;
  // unique sorted particle ids (std::map iteration order)
  int M = 0;
  {
    // merge of the p1 sequence (sorted) and all p2 values: insertion into a sorted array (C is small)
    for (int k = 0; k < C; k++) {
      const int vals[2] = {pair[k].x, pair[k].y};
      for (int q = 0; q < 2; q++) {
        int v = vals[q], lo = 0, hi = M;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (ids[mid] < v) lo = mid + 1; else hi = mid; }
        if (lo < M && ids[lo] == v) continue;
        for (int t = M; t > lo; t--) ids[t] = ids[t - 1];
        ids[lo] = v; M++;
      }
    }
  }
  auto local = [&](int v) { int lo = 0, hi = M; while (lo < hi) { int mid = (lo + hi) >> 1; if (ids[mid] < v) lo = mid + 1; else hi = mid; } return lo; };
  for (int m = 0; m <= M; m++) adj_ptr[m] = 0;
  for (int k = 0; k < C; k++) { adj_ptr[local(pair[k].x) + 1]++; adj_ptr[local(pair[k].y) + 1]++; }
  for (int m = 0; m < M; m++) adj_ptr[m + 1] += adj_ptr[m];
  for (int m = 0; m < M; m++) deg[m] = 0;
  // neighbours ascending: for vertex q first the contacts where q is id2 (others < q, ascending in k), then id1
  for (int k = 0; k < C; k++) { int b = local(pair[k].y); int o = adj_ptr[b] + deg[b]++; adj_other[o] = local(pair[k].x); adj_contact[o] = k; }
  for (int k = 0; k < C; k++) { int a = local(pair[k].x); int o = adj_ptr[a] + deg[a]++; adj_other[o] = local(pair[k].y); adj_contact[o] = k; }
  for (int k = 0; k < C; k++) { alive[k] = 1; layer[k] = -1; }
  for (int m = 0; m < M; m++) { frontier[m] = 0; newf[m] = 0; involved[m] = 0; }
  int processed = 0, maxLayer = 0, nfront = 0;
  // primitive contacts go into the first layer: they seed the frontier when they also have self contacts
  for (int m = 0; m < M; m++) if (rec_prim[dev_of ? dev_of[ids[m]] : ids[m]] >= 0 && deg[m] > 0) { frontier[m] = 1; nfront++; }
  auto first_alive = [&](int m) { for (int o = adj_ptr[m]; o < adj_ptr[m + 1]; o++) if (alive[adj_contact[o]]) return o; return -1; };
  // lonely pairs (O-O) -> layer 0
  for (int m = 0; m < M; m++) {
    if (deg[m] != 1) continue;
    int o = first_alive(m);
    int other = adj_other[o];
    if (deg[other] != 1) continue;
    if (frontier[m] || frontier[other]) continue;
    int k = adj_contact[o];
    alive[k] = 0; deg[m]--; deg[other]--; processed++; layer[k] = 0;
  }
  int currentLayer = 1;
  while (processed != C) {
    while (nfront > 0) {
      for (int m = 0; m < M; m++) { newf[m] = 0; involved[m] = 0; }
      int nnew = 0;
      if (currentLayer > maxLayer) maxLayer = currentLayer;
      for (int m = 0; m < M; m++) {
        if (!frontier[m]) continue;
        if (deg[m] == 0) continue;
        if (involved[m]) continue;
        for (int o = adj_ptr[m]; o < adj_ptr[m + 1]; o++) {
          int k = adj_contact[o];
          if (!alive[k]) continue;
          int other = adj_other[o];
          if (involved[other]) continue;
          alive[k] = 0; deg[m]--; deg[other]--; processed++;
          involved[m] = 1; involved[other] = 1;
          layer[k] = currentLayer;
          if (deg[other] > 0 && !newf[other]) { newf[other] = 1; nnew++; }
          break;
        }
      }
      currentLayer++;
      for (int m = 0; m < M; m++) frontier[m] = newf[m];
      nfront = nnew;
    }
    if (processed != C) {
      int pick = -1;
      for (int m = 0; m < M; m++) if (deg[m] == 1) { pick = m; break; }      // prioritise a chain head
      if (pick < 0) for (int m = 0; m < M; m++) if (deg[m] > 0) { pick = m; break; }   // a loop
      if (pick < 0) break;
      frontier[pick] = 1; nfront = 1;
    }
  }
  int nlayers = maxLayer + 1;
  if (nlayers > kMaxLayers) nlayers = kMaxLayers;        // overflow layers are merged into the last one
  meta[0] = C; meta[1] = nlayers;
  for (int l = 0; l <= nlayers; l++) meta[2 + l] = 0;
  for (int k = 0; k < C; k++) { int l = min(layer[k] < 0 ? 0 : layer[k], nlayers - 1); layer[k] = l; meta[2 + l + 1]++; }
  for (int l = 0; l < nlayers; l++) meta[2 + l + 1] += meta[2 + l];
  for (int k = 0; k < C; k++) layer_out[k] = layer[k];
}


// Detection + layering of one rollout for the step that reads (xn, vn); `lds` = kSelfDetectLdsInts ints of LDS scratch.
// rec_prim / self / fu are the step's record pointers of the whole batch (indexed by b inside). Call with all threads.
constexpr int kSelfCells = 4096;               // bins of the 2-D broad-phase grid
constexpr int kSelfDetectLdsInts = 16 + (kSelfCells + 1) + kSelfCells + 1 + 2048;
template <int THREADS>
__device__ __forceinline__ void self_detect_rollout(const DevSystem &S, const DevWork &W, int b, const float *x_in, const float *v_in,
                                                    int *rec_prim_all, const SelfRec &selfrec, const float *fu_all, const float *fv_all, int *lds) {
  float *redf = (float *) lds;              // [16]
  int *hist = lds + 16;                     // [kSelfCells + 1]
  int *cursor = hist + kSelfCells + 1;      // [kSelfCells]
  int &s_count = *(cursor + kSelfCells);
  unsigned *s_keys = (unsigned *) (cursor + kSelfCells + 1);   // [2048]
  const int tid = threadIdx.x;
  const int N = S.N, cap = S.self_cap;
  const size_t off = (size_t) b * 3 * N;
  const float *xn = x_in + off, *vn = v_in + off;
  int *cell = W.sd_cell + (size_t) b * N, *order = W.sd_order + (size_t) b * N;
  float *sx = W.sd_sx + off;
  int2 *raw = W.sd_rawpair + (size_t) b * cap;
  float4 *rawn = W.sd_rawn + (size_t) b * cap;
  int *tmp = W.sd_tmp + (size_t) b * 24 * cap;
  int2 *opair = selfrec.pair + (size_t) b * cap;
  float4 *onrm = selfrec.nrm + (size_t) b * cap;
  int *meta = selfrec.meta + (size_t) b * kMetaStride;
  const float h = S.h;
  const f3 grav = mk(S.gx, S.gy, S.gz);
  const f3 fu = fu_all ? mk(fu_all[3 * b], fu_all[3 * b + 1], fu_all[3 * b + 2]) : mk(0, 0, 0);
  const float *fv = fv_all ? fv_all + (size_t) b * 3 * N : nullptr;
  auto v_guess = [&](int i) {                                        // (s_n - x_n) / h
    const float m = S.mass[i];
    f3 fext = grav * m + fu;
    if (fv) fext = fext + ld3(fv, i, N);
    return ld3(vn, i, N) + fext * (h / m);
  };

  if (!S.contact_enabled || !S.self_enabled) {
    if (tid == 0) { meta[0] = 0; meta[1] = 0; meta[2] = 0; meta[kMetaStride - 1] = 0; }
    return;
  }
  // ---- 1. bounding box / longest axis / cells (Simulation.cpp:283-300) ----
  // ids: the contact list, its order and the layering follow the CALLER's vertex numbering (the reference's
  // std::map / id1 < id2 conventions), whatever the device numbering is
  const int *user_of = S.user_of, *dev_of = S.dev_of;
  const int i_first = dev_of ? dev_of[0] : 0;
  f3 p0 = ld3(xn, i_first, N) + v_guess(i_first) * h;        // particles[0].pos at this point of the reference == s_n[0]
  float mx[3] = {p0.x, p0.y, p0.z}, mn[3] = {p0.x, p0.y, p0.z};
  float vmax2 = 0.f;
  for (int i = tid; i < N; i += THREADS) {
    f3 x = ld3(xn, i, N);
    mx[0] = fmaxf(mx[0], x.x); mx[1] = fmaxf(mx[1], x.y); mx[2] = fmaxf(mx[2], x.z);
    mn[0] = fminf(mn[0], x.x); mn[1] = fminf(mn[1], x.y); mn[2] = fminf(mn[2], x.z);
    f3 v = v_guess(i);
    vmax2 = fmaxf(vmax2, dot(v, v));
    // primitive contacts seed the layering frontier (Simulation.cpp:455-460); the step kernel recomputes the same
    // flags (same inputs, same code) and also stores the normals
    f3 nrm;
    rec_prim_all[(size_t) b * N + i] = detect_primitive(S, x, v, nrm);
  }
  float maxd[3], mind[3];
  for (int d = 0; d < 3; d++) { maxd[d] = block_max<THREADS>(mx[d], redf); mind[d] = -block_max<THREADS>(-mn[d], redf); }
  const float vmax = sqrtf(block_max<THREADS>(vmax2, redf));
  int axis = 0;
  for (int d = 1; d < 3; d++) if (maxd[d] - mind[d] > maxd[axis] - mind[axis]) axis = d;
  const float dimA = maxd[axis] - mind[axis], minA = mind[axis];
  const float maxR = S.max_radii;
  const int cellNum = max(min(512, (int) (dimA / (maxR * 2.f))), 1);
  const float cellDim = fmaxf(dimA / (float) cellNum, 1e-30f);
  const int window = (int) ceilf(maxR * 2.f / cellDim) + 2 + 2;   // cellId2 < cellId1 + sweepCellRadius + 2
  // ---- 2. broad phase: counting sort into a 2-D grid over the two longest axes ----
  // The reference sweeps 1-D cells along the longest axis and tests p2 in cells [c1, c1 + sweepCellRadius + 2). That
  // criterion (on the 1-D cell ids, kept below) never rejects a pair closer than r1 + r2, so any exact broad phase that
  // also applies it yields the same pair set; the 2-D grid visits ~25 candidates per vertex instead of ~600.
  int axisB = (axis + 1) % 3;
  for (int d = 0; d < 3; d++) if (d != axis && maxd[d] - mind[d] > maxd[axisB] - mind[axisB]) axisB = d;
  const float dimB = maxd[axisB] - mind[axisB], minB = mind[axisB];
  const float reach_max = fminf(2.f * maxR + 2.f * vmax * h, 1.0f);      // no accepted pair is farther apart than this
  const float cs = fmaxf(fmaxf(reach_max, fmaxf(dimA, dimB) / 64.f), 1e-30f);
  const int nA = min(64, (int) (dimA / cs) + 1), nB = min(64, (int) (dimB / cs) + 1), ncell = nA * nB;
  for (int c = tid; c <= ncell; c += THREADS) hist[c] = 0;
  if (tid == 0) s_count = 0;
  __syncthreads();
  const float *xa = xn + (size_t) axis * N, *xb = xn + (size_t) axisB * N;
  for (int i = tid; i < N; i += THREADS) {
    const int ca = max(min((int) ((xa[i] - minA) / cs), nA - 1), 0), cb = max(min((int) ((xb[i] - minB) / cs), nB - 1), 0);
    const int c = ca * nB + cb;
    cell[i] = c;
    atomicAdd(&hist[c + 1], 1);
  }
  __syncthreads();
  {  // exclusive prefix sum of the bin counts: per-thread chunks, serial scan of the chunk totals, add back
    const int chunk = (ncell + THREADS - 1) / THREADS, c0 = tid * chunk, c1 = min(ncell, c0 + chunk);
    int sum = 0;
    for (int c = c0; c < c1; c++) sum += hist[c + 1];
    cursor[tid] = sum;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int t = 0; t < THREADS; t++) { const int v = cursor[t]; cursor[t] = run; run += v; } }
    __syncthreads();
    int run = cursor[tid];
    for (int c = c0; c < c1; c++) { run += hist[c + 1]; hist[c + 1] = run; }      // hist[c] = first sorted slot of cell c
  }
  __syncthreads();
  for (int c = tid; c < ncell; c += THREADS) cursor[c] = hist[c];
  __syncthreads();
  for (int i = tid; i < N; i += THREADS) {
    int s = atomicAdd(&cursor[cell[i]], 1);
    order[s] = i;
  }
  __syncthreads();
  for (int s = tid; s < N; s += THREADS) st3(sx, s, N, ld3(xn, order[s], N));
  __syncthreads();
  // ---- 3. candidate pairs: every unordered pair of the 3 x 3 (or wider) cell neighbourhood once (s2 > s) ----
  const int K = (int) ceilf(reach_max / cs);
  for (int s = tid; s < N; s += THREADS) {
    const int i = order[s];
    const f3 xi = ld3(sx, s, N), vi = v_guess(i);
    const float ri = S.radii[i];
    const int ci = cell[i], cia = ci / nB, cib = ci - cia * nB;
    const int c1d_i = max(min((int) ((xa[i] - minA) / cellDim), cellNum - 1), 0);      // the reference's 1-D sweep cell
    const float reach = ri + maxR + (sqrtf(dot(vi, vi)) + vmax) * h;     // conservative: swept distance >= dist - |v_rel| h
    const float reach2 = fminf(reach * reach, 1.0f);
    for (int a2 = max(cia - K, 0); a2 <= min(cia + K, nA - 1); a2++) {
      const int s_lo = hist[a2 * nB + max(cib - K, 0)], s_hi = hist[a2 * nB + min(cib + K, nB - 1) + 1];   // cells of one row are contiguous
      for (int s2 = max(s_lo, s + 1); s2 < s_hi; s2++) {
        f3 dx = xi - ld3(sx, s2, N);
        float d2 = dot(dx, dx);
        if (d2 > 1.0f || d2 >= reach2) continue;           // dist > 1.0 early-out (Simulation.cpp:323) + cheap reject
        const int j = order[s2];
        const int c1d_j = max(min((int) ((xa[j] - minA) / cellDim), cellNum - 1), 0);
        if (abs(c1d_i - c1d_j) >= window) continue;        // outside the reference's sweep window: never tested there
        if (connected(S, i, j)) continue;
        // the reference calls isSelfCollision(particles[p1], particles[p2]) with p1 from the lower cell; the test is
        // symmetric and the normal is oriented from id2 to id1, so the call order does not matter
        f3 nrm;
        const int ui = user_of ? user_of[i] : i, uj = user_of ? user_of[j] : j;
        if (!self_collision(ri + S.radii[j], xi, ld3(sx, s2, N), vi, v_guess(j), h, ui, uj, nrm)) continue;
        int k = atomicAdd(&s_count, 1);
        if (k < cap) { raw[k] = make_int2(min(ui, uj), max(ui, uj)); rawn[k] = make_float4(nrm.x, nrm.y, nrm.z, 0.f); }
      }
    }
  }
  __syncthreads();
  const int C = min(s_count, cap);
  // ---- 4. deterministic order: rank sort by key id1 * N + id2 ----
  // tmp (24*cap ints): [0, 18cap+1) serial layering structures | [19cap, 20cap) layer of each sorted contact |
  //                     [20cap, 22cap) pairs in sorted order
  int2 *spair = (int2 *) (tmp + 20 * cap);
  if (C <= 2048) {
    for (int k = tid; k < C; k += THREADS) s_keys[k] = (unsigned) raw[k].x * (unsigned) N + (unsigned) raw[k].y;
    __syncthreads();
    for (int k = tid; k < C; k += THREADS) {
      const unsigned key = s_keys[k];
      int rank = 0;
      for (int q = 0; q < C; q++) rank += (s_keys[q] < key) ? 1 : 0;
      spair[rank] = raw[k];
      onrm[rank] = rawn[k];                      // temporarily in sorted (not yet layered) order
    }
  } else {
    for (int k = tid; k < C; k += THREADS) {
      const unsigned key = (unsigned) raw[k].x * (unsigned) N + (unsigned) raw[k].y;
      int rank = 0;
      for (int q = 0; q < C; q++) rank += (((unsigned) raw[q].x * (unsigned) N + (unsigned) raw[q].y) < key) ? 1 : 0;
      spair[rank] = raw[k];
      onrm[rank] = rawn[k];
    }
  }
  __syncthreads();
  // ---- 5. layering (serial, thread 0) then a stable partition by layer ----
  int *layer_of = tmp + 19 * cap;
  if (tid == 0) contact_sorting_serial(C, spair, rec_prim_all + (size_t) b * N, dev_of, tmp, cap, meta, layer_of);
  __syncthreads();
  // position inside the layer = number of earlier (sorted) contacts of the same layer
  for (int k = tid; k < C; k += THREADS) {
    const int l = layer_of[k];
    int pos = 0;
    for (int q = 0; q < k; q++) pos += (layer_of[q] == l) ? 1 : 0;
    const int dst = meta[2 + l] + pos;
    opair[dst] = dev_of ? make_int2(dev_of[spair[k].x], dev_of[spair[k].y]) : spair[k];   // stored in device numbering
    rawn[dst] = onrm[k];                         // raw buffer reused as the destination of the normals
  }
  __syncthreads();
  for (int k = tid; k < C; k += THREADS) onrm[k] = rawn[k];
  __syncthreads();
  // ---- 6. working set of the layered friction passes: the distinct vertices of the contacts get slots 0..M-1 (in order
  // of first appearance) so that those passes can run inside LDS; slots ride in nrm.w, the vertex list in selfrec.verts
  if (tid == 0) {
    int *slot_of = cell;                       // [N] scratch of the detection, free again
    int *verts = selfrec.verts + (size_t) b * 2 * cap;
    for (int k = 0; k < C; k++) { slot_of[opair[k].x] = -1; slot_of[opair[k].y] = -1; }
    int M = 0;
    for (int k = 0; k < C; k++) {
      const int2 ab = opair[k];
      if (slot_of[ab.x] < 0) { slot_of[ab.x] = M; verts[M++] = ab.x; }
      if (slot_of[ab.y] < 0) { slot_of[ab.y] = M; verts[M++] = ab.y; }
      onrm[k].w = __int_as_float(slot_of[ab.x] | (slot_of[ab.y] << 16));
    }
    meta[kMetaStride - 1] = M;
  }
}


}  // namespace dc
