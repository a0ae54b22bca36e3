// Self-collision detection + contact layering of one rollout as a device function (used by the stand-alone k_self_detect
// launch and, inlined, by the fused multi-step forward kernel). Reference: Simulation::collisionDetection
// (Simulation.cpp:225-373, self part :281-352), isSelfCollision (:194-220), contactSorting (:422-624).
#pragma once
#include "dc_devlib.h"
#include "dc_selftmp.h"

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

// Exclusive prefix sum of a[0, n) in place (a in global memory or LDS); returns the total to every thread. `scratch` =
// THREADS / 64 ints of LDS. Call with all threads; starts and ends with a barrier.
template <int THREADS>
__device__ __forceinline__ int block_scan_excl(int *a, int n, int *scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = (n + THREADS - 1) / THREADS, c0 = min(n, tid * chunk), c1 = min(n, c0 + chunk);
  __syncthreads();                               // a[] was written by other threads
  int sum = 0;
  for (int c = c0; c < c1; c++) sum += a[c];
  int v = sum;                                   // inclusive scan of the chunk totals inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
  __syncthreads();
  if (lane == 63) scratch[wave] = v;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; w++) { const int t = scratch[w]; base += (w < wave) ? t : 0; total += t; }
  int run = base + v - sum;
  for (int c = c0; c < c1; c++) { const int t = a[c]; a[c] = run; run += t; }
  __syncthreads();
  return total;
}

// Layout of the per-rollout layering scratch (DevWork::sd_tmp), in ints; U = 2 cap + 2 entries per per-vertex array
// (a contact list of C pairs has at most 2 C distinct vertices).
struct SelfTmp {
  int *ids, *adj_ptr, *adj_other, *adj_contact, *deg, *frontier, *newf, *involved, *c2, *act, *layer, *alive;
  int2 *spair;
  __device__ SelfTmp(int *tmp, int cap) {
    const int U = 2 * cap + 2;
    ids = tmp; adj_ptr = tmp + U; adj_other = tmp + 2 * U; adj_contact = tmp + 3 * U; deg = tmp + 4 * U; frontier = tmp + 5 * U;
    newf = tmp + 6 * U; involved = tmp + 7 * U; c2 = tmp + 8 * U; act = tmp + 9 * U; layer = tmp + 10 * U; alive = layer + cap;
    spair = (int2 *) (alive + cap);              // 10 U + 2 cap is even: 8-byte aligned
  }
};

// Simulation::contactSorting (Simulation.cpp:422-624), the frontier propagation: everything that is order independent
// (vertex table, adjacency, the lonely pairs of layer 0) has been built by the whole workgroup (self_detect_rollout);
// what is left here, on ONE thread exactly as the reference's std::map / std::set code does it, is the greedy walk over
// the `nact` vertices that still have contacts (act[] ascending = std::map iteration order). Returns maxLayer.
__device__ int contact_layering_rest(const SelfTmp &t, int nact, int remaining) {
  int maxLayer = 0, nfront = 0, currentLayer = 1;
  for (int q = 0; q < nact; q++) nfront += t.frontier[t.act[q]];
  while (remaining > 0) {
    while (nfront > 0) {
      for (int q = 0; q < nact; q++) { const int m = t.act[q]; t.newf[m] = 0; t.involved[m] = 0; }
      int nnew = 0;
      if (currentLayer > maxLayer) maxLayer = currentLayer;
      for (int q = 0; q < nact; q++) {
        const int m = t.act[q];
        if (!t.frontier[m]) continue;
        if (t.deg[m] == 0) continue;
        if (t.involved[m]) continue;
        for (int o = t.adj_ptr[m]; o < t.adj_ptr[m + 1]; o++) {
          const int k = t.adj_contact[o];
          if (!t.alive[k]) continue;
          const int other = t.adj_other[o];
          if (t.involved[other]) continue;
          t.alive[k] = 0; t.deg[m]--; t.deg[other]--; remaining--;
          t.involved[m] = 1; t.involved[other] = 1;
          t.layer[k] = currentLayer;
          if (t.deg[other] > 0 && !t.newf[other]) { t.newf[other] = 1; nnew++; }
          break;
        }
      }
      currentLayer++;
      for (int q = 0; q < nact; q++) { const int m = t.act[q]; t.frontier[m] = t.newf[m]; }
      nfront = nnew;
    }
    if (remaining > 0) {
      int pick = -1;
      for (int q = 0; q < nact; q++) if (t.deg[t.act[q]] == 1) { pick = t.act[q]; break; }      // prioritise a chain head
      if (pick < 0) for (int q = 0; q < nact; q++) if (t.deg[t.act[q]] > 0) { pick = t.act[q]; break; }   // a loop
      if (pick < 0) break;
      t.frontier[pick] = 1; nfront = 1;
    }
  }
  return maxLayer;
}


// Detection + layering of one rollout for the step that reads (xn, vn); `lds` = kSelfDetectLdsInts ints of LDS scratch.
// rec_prim / self / fu are the step's record pointers of the whole batch (indexed by b inside). Call with all threads.
template <int THREADS>
__device__ __forceinline__ void self_detect_rollout(const DevSystem &S, const DevWork &W, int b, const float *x_in, const float *v_in,
                                                    int *rec_prim_all, const SelfRec &selfrec, const float *fu_all, const float *fv_all, const float *fvs_all, int *lds, const float *fv2_all = nullptr) {
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
  int *tmp = W.sd_tmp + (size_t) b * self_tmp_ints(cap);
  int2 *opair = selfrec.pair + (size_t) b * cap;
  float4 *onrm = selfrec.nrm + (size_t) b * cap;
  int *meta = selfrec.meta + (size_t) b * kMetaStride;
  const float h = S.h;
  const f3 grav = mk(S.gx, S.gy, S.gz);
  const f3 fu = fu_all ? mk(fu_all[3 * b], fu_all[3 * b + 1], fu_all[3 * b + 2]) : mk(0, 0, 0);
  const float *fv = fv_all ? fv_all + (size_t) b * 3 * N : nullptr;
  const float fvs = fvs_all ? fvs_all[b] : 1.f;
  auto v_guess = [&](int i) {                                        // (s_n - x_n) / h
    const float m = S.mass[i];
    f3 fext = grav * m + fu;
    if (fv) fext = fext + ld3(fv, i, N) * fvs;
    if (fv2_all) fext = fext + ld3(fv2_all + (size_t) b * 3 * N, i, N);
    return ld3(vn, i, N) + fext * (h / m);
  };

  if (!S.contact_enabled || !S.self_enabled) {
    if (tid == 0) { meta[0] = 0; meta[1] = 0; meta[2] = 0; meta[kMetaStride - 1] = 0; meta[kMetaStride - 2] = 0; meta[kMetaStride - 3] = 0; }
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
  const int raw_count = s_count;
  const int C = min(raw_count, cap);
  // ---- 4. deterministic order: rank sort by key id1 * N + id2 ----
  const SelfTmp t(tmp, cap);
  int2 *spair = t.spair;
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
  // ---- 5. layering (Simulation::contactSorting, Simulation.cpp:422-624) ----
  // 5a. vertex table: local id of a vertex = its rank among the distinct vertices of the contacts (std::map order);
  //     lid[] lives in the cell-sort scratch `order` (free since step 3), indexed by the caller's vertex id
  int *lid = order;
  for (int i = tid; i < N; i += THREADS) lid[i] = 0;
  __syncthreads();
  int bad = 0;                                   // internal consistency (never expected): reported through the overflow flags, bit 2
  for (int k = tid; k < C; k += THREADS) {
    const int2 ab = spair[k];
    if ((unsigned) ab.x >= (unsigned) N || (unsigned) ab.y >= (unsigned) N) { bad = 1; spair[k] = make_int2(0, 0); }
  }
  __syncthreads();
  for (int k = tid; k < C; k += THREADS) { lid[spair[k].x] = 1; lid[spair[k].y] = 1; }
  const int M = block_scan_excl<THREADS>(lid, N, hist);
  for (int k = tid; k < C; k += THREADS) { t.ids[lid[spair[k].x]] = spair[k].x; t.ids[lid[spair[k].y]] = spair[k].y; }
  for (int m = tid; m <= M; m += THREADS) { t.deg[m] = 0; t.c2[m] = 0; }
  __syncthreads();
  // 5b. degrees, adjacency in the reference's order: the neighbours of q ascending = first the contacts where q is id2
  //     (their id1 < q, ascending in the sorted list), then those where q is id1
  for (int k = tid; k < C; k += THREADS) {
    const int a = lid[spair[k].x], bb = lid[spair[k].y];
    atomicAdd(&t.deg[a], 1); atomicAdd(&t.deg[bb], 1); atomicAdd(&t.c2[bb], 1);
  }
  __syncthreads();
  for (int m = tid; m <= M; m += THREADS) t.adj_ptr[m] = (m < M) ? t.deg[m] : 0;
  block_scan_excl<THREADS>(t.adj_ptr, M + 1, hist);
  for (int k = tid; k < C; k += THREADS) {
    const int2 ab = spair[k];
    const int a = lid[ab.x], bb = lid[ab.y];
    int r2 = 0;                                  // earlier contacts with the same id2
    for (int q = 0; q < k; q++) r2 += (spair[q].y == ab.y) ? 1 : 0;
    int r1 = 0;                                  // earlier contacts with the same id1 (contiguous in the sorted list)
    while (k - r1 - 1 >= 0 && spair[k - r1 - 1].x == ab.x) r1++;
    const int ob = t.adj_ptr[bb] + r2, oa = t.adj_ptr[a] + t.c2[a] + r1;
    if ((unsigned) ob < (unsigned) (2 * C) && (unsigned) oa < (unsigned) (2 * C)) {
      t.adj_other[ob] = a; t.adj_contact[ob] = k;
      t.adj_other[oa] = bb; t.adj_contact[oa] = k;
    } else bad = 1;
    t.alive[k] = 1; t.layer[k] = -1;
  }
  // primitive contacts go into the first layer: they seed the frontier when they also have self contacts (:455-460)
  for (int m = tid; m < M; m += THREADS) {
    const int v = t.ids[m];
    t.frontier[m] = (rec_prim_all[(size_t) b * N + (dev_of ? dev_of[v] : v)] >= 0 && t.deg[m] > 0) ? 1 : 0;
    t.newf[m] = 0; t.involved[m] = 0;
  }
  __syncthreads();
  // 5c. lonely pairs (O-O) -> layer 0 (:462-480): both ends have no other contact, so no two threads touch one vertex
  int lonely = 0;
  for (int k = tid; k < C; k += THREADS) {
    const int a = lid[spair[k].x], bb = lid[spair[k].y];
    if (t.deg[a] == 1 && t.deg[bb] == 1 && !t.frontier[a] && !t.frontier[bb]) {
      t.alive[k] = 0; t.layer[k] = 0; t.deg[a] = 0; t.deg[bb] = 0; lonely++;
    }
  }
  __syncthreads();
  for (int m = tid; m <= M; m += THREADS) t.act[m] = (m < M && t.deg[m] > 0) ? 1 : 0;
  const int nact = block_scan_excl<THREADS>(t.act, M + 1, hist);       // act[m] = position of m among the active vertices
  {
    // compaction in place is a race (act[] is read as positions and written as the list): go through newf-free scratch c2
    for (int m = tid; m < M; m += THREADS) if (t.deg[m] > 0) t.c2[t.act[m]] = m;
    __syncthreads();
    for (int q = tid; q < nact; q += THREADS) t.act[q] = t.c2[q];
  }
  int nl_total = 0;
  for (int k = tid; k < C; k += THREADS) nl_total += (t.alive[k]) ? 1 : 0;
  (void) lonely;
  __syncthreads();
  if (tid == 0) { hist[16] = 0; hist[17] = 0; }  // remaining contacts: block sum through an LDS atomic
  __syncthreads();
  if (nl_total) atomicAdd(&hist[16], nl_total);
  if (bad) atomicOr(&hist[17], 4);
  __syncthreads();
  const int remaining = hist[16];
  // 5d. frontier propagation on one thread (the contacts that are not lonely pairs; typically a small fraction)
  if (tid == 0) {
    int maxLayer = 0;
    if (remaining > 0) maxLayer = contact_layering_rest(t, nact, remaining);
    int nlayers = maxLayer + 1, flags = ((raw_count > cap) ? 1 : 0) | hist[17];
    if (nlayers > kMaxLayers) { nlayers = kMaxLayers; flags |= 2; }       // reported: dc_step_stats::self_overflow
    meta[0] = C; meta[1] = nlayers;
    meta[kMetaStride - 1] = M; meta[kMetaStride - 2] = flags; meta[kMetaStride - 3] = raw_count;
  }
  __syncthreads();
  const int nlayers = meta[1];
  for (int l = tid; l <= nlayers; l += THREADS) meta[2 + l] = 0;
  __syncthreads();
  for (int k = tid; k < C; k += THREADS) {
    const int l = min(t.layer[k] < 0 ? 0 : t.layer[k], nlayers - 1);
    t.layer[k] = l;
    atomicAdd(&meta[2 + l + 1], 1);
  }
  __syncthreads();
  if (tid == 0) for (int l = 0; l < nlayers; l++) meta[2 + l + 1] += meta[2 + l];
  __syncthreads();
  // stable partition by layer: position inside the layer = number of earlier (sorted) contacts of the same layer
  for (int k = tid; k < C; k += THREADS) {
    const int l = t.layer[k];
    int pos = 0;
    for (int q = 0; q < k; q++) pos += (t.layer[q] == l) ? 1 : 0;
    const int dst = min(meta[2 + l] + pos, C - 1);
    const int2 ab = spair[k];
    opair[dst] = dev_of ? make_int2(dev_of[ab.x], dev_of[ab.y]) : ab;   // stored in device numbering
    // ---- 6. working set of the layered friction passes: the distinct vertices of the contacts are slots 0..M-1 (their
    // rank, 5a) so that those passes can run inside LDS; slots ride in nrm.w, the vertex list in selfrec.verts
    float4 nq = onrm[k];
    nq.w = __int_as_float(lid[ab.x] | (lid[ab.y] << 16));
    rawn[dst] = nq;                              // raw buffer reused as the destination of the normals
  }
  {
    // CONTRACT (the adjoint relies on it, dc_adjoint.hip: the per-step mark of a working-set vertex stores its slot with a plain read-modify-write): the
    // list holds every vertex AT MOST ONCE — t.ids[0 .. M) are the distinct vertices of the contacts by rank (5a), and dev_of is a permutation.
    // dc_set_record deduplicates a list handed in from outside the same way (dc_engine.hip).
    int *verts = selfrec.verts + (size_t) b * 2 * cap;
    for (int m = tid; m < M; m += THREADS) { const int v = t.ids[m]; verts[m] = dev_of ? dev_of[v] : v; }
  }
  __syncthreads();
  for (int k = tid; k < C; k += THREADS) onrm[k] = rawn[k];
  __syncthreads();
}


}  // namespace dc
