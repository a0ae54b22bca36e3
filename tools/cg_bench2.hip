// Developer micro-benchmark (not part of the product): packet-ELL variant of the resident block-Jacobi PCG.
// Off-diagonals of a row travel as 16-byte packets {v0, v1, v2, d0 | d1 << 10 | d2 << 20} (three fp32 values and
// three 10-bit column deltas relative to the row, biased by 512); the diagonal is a separate fp32 per row.
//   hipcc --offload-arch=gfx950 -O3 tools/cg_bench2.hip -o cg_bench2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double *red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; k++) s += red[k];
  return s;
}

// MODE 0: full CG.  MODE 1: SpMV only (no reductions / updates).
// XM 0: iterate x in registers.  XM 1: x read-modify-written in a float4 global scratch (fire and forget).
// PB: packets per software-pipelined batch (the next row's first batch is in flight while this row is consumed).
template <int PB>
__device__ __forceinline__ void load_batch(int4 (&e)[PB], const int4 *__restrict__ row, int s0, int np) {
#pragma unroll
  for (int j = 0; j < PB; j++) e[j] = row[(s0 + j) * 64];   // rows are stored padded to a multiple of PB packets
}

template <int PB, int NP>
__device__ __forceinline__ void consume(const int4 (&e)[PB], const float *lp, int base, float &ax, float &ay, float &az) {
#pragma unroll
  for (int j = 0; j < PB; j++) {
    const int c0 = base + (e[j].w & 1023), c1 = base + ((e[j].w >> 10) & 1023), c2 = base + ((e[j].w >> 20) & 1023);
    const float a0 = __int_as_float(e[j].x), a1 = __int_as_float(e[j].y), a2 = __int_as_float(e[j].z);
    const float2 *lxy = (const float2 *) lp;
    const float *lz = lp + 2 * NP;
    const float2 q0 = lxy[c0], q1 = lxy[c1], q2 = lxy[c2];
    const float z0 = lz[c0], z1 = lz[c1], z2 = lz[c2];
    ax = fmaf(a0, q0.x, ax); ay = fmaf(a0, q0.y, ay); az = fmaf(a0, z0, az);
    ax = fmaf(a1, q1.x, ax); ay = fmaf(a1, q1.y, ay); az = fmaf(a1, z1, az);
    ax = fmaf(a2, q2.x, ax); ay = fmaf(a2, q2.y, ay); az = fmaf(a2, z2, az);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int THREADS>
__device__ __forceinline__ double block_sum_f(float v, double *red) {   // fp32 inside a wave, fp64 across waves
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = (double) v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; k++) s += red[k];
  return s;
}

// Symmetrically scaled system  (D^-1/2 P D^-1/2) xs = D^-1/2 rhs  (unit diagonal): plain CG on it IS Jacobi-PCG on P,
// with no dinv / diagonal traffic in the loop.  XL rows of the iterate live in LDS, VPT - XL in registers.
template <int THREADS, int VPT, int MODE, int XL, int PB>
__global__ __launch_bounds__(THREADS) void k_cg(const int4 *__restrict__ pk, const int *__restrict__ pk_ptr,
                                                const int *__restrict__ pk_n, const float *__restrict__ diag,
                                                const float *__restrict__ dinv, const float *__restrict__ rhs,
                                                float *__restrict__ xout, int ldr, int iters, long long *cycles) {
  constexpr int NP = THREADS * VPT;
  constexpr int WAVES = THREADS / 64;
  constexpr int XR = VPT - XL;
  extern __shared__ float lp[];
  float *lx = lp + 3 * NP;                 // [XL][3][THREADS]
  __shared__ double red[WAVES];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float rr[VPT][3], xx[XR > 0 ? XR : 1][3], ap[VPT][3];
  float part = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; k++) {
    const int i = tid + k * THREADS;
    const float sq = sqrtf(dinv[i]);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      rr[k][c] = rhs[((size_t) b * 3 + c) * ldr + i] * sq;
      if (k >= XL) xx[k >= XL ? k - XL : 0][c] = 0.f; else lx[(k * 3 + c) * THREADS + tid] = 0.f;
      ap[k][c] = 0.f;
      lp[(c < 2 ? 2 * i + c : 2 * NP + i)] = rr[k][c];
      part = fmaf(rr[k][c], rr[k][c], part);
    }
  }
  double rz = block_sum_f<THREADS>(part, red);
  long long t0 = clock64();
  long long ph[5] = {0, 0, 0, 0, 0}, tp = t0;
#define PH(k) { long long n_ = clock64(); ph[k] += n_ - tp; tp = n_; }
  for (int it = 0; it < iters; it++) {
    __syncthreads();
    part = 0.f;
    int zs;                                          // opaque zero: keeps per-row addresses out of LICM's reach
    asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
    const int wz = wv + zs, tz = tid + zs;
    int4 nxt[PB];
    load_batch<PB>(nxt, pk + pk_ptr[wz] + lane, 0, pk_n[wz]);
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int chunk = wz + k * WAVES;             // wave-uniform: scalar loads below
      const int i = chunk * 64 + lane;
      const int np = pk_n[chunk];
      const int4 *row = pk + pk_ptr[chunk] + lane;
      int4 cur[PB];
#pragma unroll
      for (int j = 0; j < PB; j++) cur[j] = nxt[j];
      if (k + 1 < VPT) load_batch<PB>(nxt, pk + pk_ptr[chunk + WAVES] + lane, 0, pk_n[chunk + WAVES]);
      const float2 pxy = ((const float2 *) lp)[i];
      const float px = pxy.x, py = pxy.y, pz = lp[2 * NP + i];
      float ax = px, ay = py, az = pz;
      const int base = i - 512;
      consume<PB, NP>(cur, lp, base, ax, ay, az);
      for (int s0 = PB; s0 < np; s0 += PB) {        // rows wider than one batch (rare)
        load_batch<PB>(cur, row, s0, np);
        consume<PB, NP>(cur, lp, base, ax, ay, az);
      }
      ap[k][0] = ax; ap[k][1] = ay; ap[k][2] = az;
      part += px * ax + py * ay + pz * az;
#ifndef RPR
#define RPR 1
#endif
      if ((k + 1) % RPR == 0) __builtin_amdgcn_sched_barrier(0);      // rows per scheduling region (experiment: -DRPR=2, 4)
    }
    if (MODE != 0) {
#pragma unroll
      for (int k = 0; k < VPT; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) rr[k][c] += ap[k][c];
      continue;
    }
    PH(0)
    const double pAp = block_sum_f<THREADS>(part, red);
    PH(1)
    const float alpha = (float) (rz / pAp);
    part = 0.f;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int i = tz + k * THREADS;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        if (k >= XL) xx[k >= XL ? k - XL : 0][c] = fmaf(alpha, lp[(c < 2 ? 2 * i + c : 2 * NP + i)], xx[k >= XL ? k - XL : 0][c]);
        else lx[(k * 3 + c) * THREADS + tid] = fmaf(alpha, lp[(c < 2 ? 2 * i + c : 2 * NP + i)], lx[(k * 3 + c) * THREADS + tid]);
        rr[k][c] = fmaf(-alpha, ap[k][c], rr[k][c]);
        part = fmaf(rr[k][c], rr[k][c], part);
      }
    }
    PH(2)
    const double rz_new = block_sum_f<THREADS>(part, red);
    PH(3)
    const float beta = (float) (rz_new / rz);
    rz = rz_new;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int i = tz + k * THREADS;
#pragma unroll
      for (int c = 0; c < 3; c++) lp[(c < 2 ? 2 * i + c : 2 * NP + i)] = fmaf(beta, lp[(c < 2 ? 2 * i + c : 2 * NP + i)], rr[k][c]);
    }
    PH(4)
  }
  long long t1 = clock64();
#pragma unroll
  for (int k = 0; k < VPT; k++) {
    const int i = tid + k * THREADS;
    const float sq = sqrtf(dinv[i]);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float xv = (k >= XL) ? xx[k >= XL ? k - XL : 0][c] : lx[(k * 3 + c) * THREADS + tid];
      xout[((size_t) b * 3 + c) * ldr + i] = xv * sq + rr[k][c] * (MODE != 0);
    }
  }
  if (tid == 0) { cycles[b * 8] = t1 - t0; for (int k = 0; k < 5; k++) cycles[b * 8 + 1 + k] = ph[k]; }
}

template <int THREADS, int VPT, int MODE, int XL, int PB>
void run(const char *name, int B, int iters, const int4 *pk, const int *pptr, const int *pn, const float *diag, const float *dinv,
         const float *rhs, float *xout, int ldr, long long *cyc) {
  const size_t lds = (size_t) 3 * THREADS * (VPT + XL) * sizeof(float);
  CK(hipFuncSetAttribute((const void *) k_cg<THREADS, VPT, MODE, XL, PB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_cg<THREADS, VPT, MODE, XL, PB>), dim3(B), dim3(THREADS), lds, 0, pk, pptr, pn, diag, dinv, rhs, xout, ldr, iters, cyc);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    long long c6[6];
    CK(hipMemcpy(c6, cyc, 6 * sizeof(long long), hipMemcpyDeviceToHost));
    long long c0 = c6[0];
    std::vector<float> x(8);
    CK(hipMemcpy(x.data(), xout, 8 * sizeof(float), hipMemcpyDeviceToHost));
    if (rep == 1) printf("%-34s B=%d iters=%d: %.3f ms  -> %.2f us/iter, block0 %.0f cycles/iter, x[0..2]= %g %g %g\n", name, B, iters, ms,
                         ms * 1e3 / iters, (double) c0 / iters, x[0], x[1], x[2]);
    if (rep == 1 && MODE == 0) printf("      phases/iter: spmv %lld  red1 %lld  upd %lld  red2 %lld  pupd %lld\n", c6[1] / iters, c6[2] / iters, c6[3] / iters, c6[4] / iters, c6[5] / iters);
  }
}

static int PBS = 4;
int main(int argc, char **argv) {
  if (argc > 3) PBS = atoi(argv[3]);
  const int G = 100, N = G * G, NP = 10752, B = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 200;
  const int off[13][2] = {{0, 0}, {0, 1}, {0, -1}, {1, 0}, {-1, 0}, {1, -1}, {-1, 1}, {1, 1}, {-1, -1}, {2, -1}, {-2, 1}, {1, -2}, {-1, 2}};
  const int nchunks = NP / 64;
  std::vector<std::vector<std::pair<int, float>>> rows(NP);
  std::vector<float> diag(NP, 1.f), dinv(NP, 0.f);
  for (int r = 0; r < N; r++) {
    int gi = r / G, gj = r % G;
    float d = 0.0006f;
    for (int k = 1; k < 13; k++) {
      int a = gi + off[k][0], c = gj + off[k][1];
      if (a < 0 || c < 0 || a >= G || c >= G) continue;
      float v = -0.002f - 0.0001f * (k % 3);
      rows[r].push_back({a * G + c, v});
      d += -v;
    }
    std::sort(rows[r].begin(), rows[r].end());
    diag[r] = d; dinv[r] = 1.0f / d;
  }
  for (int r = 0; r < N; r++) for (auto &e : rows[r]) e.second *= std::sqrt(dinv[r]) * std::sqrt(dinv[e.first]);
  std::vector<int> pptr(nchunks), pn(nchunks), flat;
  for (int ch = 0; ch < nchunks; ch++) {
    int w = 0;
    for (int l = 0; l < 64; l++) w = std::max(w, (int) rows[64 * ch + l].size());
    const int np = std::max(PBS, ((w + 2) / 3 + PBS - 1) / PBS * PBS);
    pptr[ch] = (int) flat.size() / 4; pn[ch] = np;
    flat.resize(flat.size() + (size_t) 4 * 64 * np);
    for (int s = 0; s < np; s++) for (int l = 0; l < 64; l++) {
      int r = 64 * ch + l;
      int wd = 0, bits[3] = {0, 0, 0};
      for (int q = 0; q < 3; q++) {
        int e = 3 * s + q, d = 512;
        if (e < (int) rows[r].size()) { d = rows[r][e].first - r + 512; memcpy(&bits[q], &rows[r][e].second, 4); }
        if (d < 0 || d > 1023) { printf("delta out of range\n"); return 1; }
        wd |= d << (10 * q);
      }
      size_t o = 4 * ((size_t) pptr[ch] + (size_t) s * 64 + l);
      flat[o] = bits[0]; flat[o + 1] = bits[1]; flat[o + 2] = bits[2]; flat[o + 3] = wd;
    }
  }
  std::vector<float> rhs((size_t) B * 3 * NP, 0.f);
  for (int b = 0; b < B; b++) for (int c = 0; c < 3; c++) for (int i = 0; i < N; i++) rhs[((size_t) b * 3 + c) * NP + i] = 1e-3f * sinf(0.01f * i + c + b);
  int *d_flat, *d_pptr, *d_pn; float *d_diag, *d_dinv, *d_rhs, *d_x; long long *d_cyc;
  CK(hipMalloc(&d_flat, flat.size() * 4)); CK(hipMemcpy(d_flat, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_pptr, nchunks * 4)); CK(hipMemcpy(d_pptr, pptr.data(), nchunks * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_pn, nchunks * 4)); CK(hipMemcpy(d_pn, pn.data(), nchunks * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_diag, NP * 4)); CK(hipMemcpy(d_diag, diag.data(), NP * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_dinv, NP * 4)); CK(hipMemcpy(d_dinv, dinv.data(), NP * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_rhs, rhs.size() * 4)); CK(hipMemcpy(d_rhs, rhs.data(), rhs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_x, rhs.size() * 4));
  CK(hipMalloc(&d_cyc, B * 8 * sizeof(long long)));
  printf("matrix: %zu bytes packet-ELL, N=%d\n", flat.size() * 4, N);
  float4 *d_xs;
  CK(hipMalloc(&d_xs, (size_t) B * NP * sizeof(float4)));
#define RUN(T, V, M, X, P, name) run<T, V, M, X, P>(name, B, iters, (const int4 *) d_flat, d_pptr, d_pn, d_diag, d_dinv, d_rhs, d_x, NP, d_cyc)
  if (PBS == 4) {
    RUN(1024, 10, 0, 3, 4, "1024x10 XL3 PB4 full");
    RUN(512, 20, 0, 6, 4, "512x20 XL6 PB4 full");
    RUN(512, 20, 1, 6, 4, "512x20 XL6 PB4 spmv");
  } else {
    RUN(1024, 10, 0, 3, 2, "1024x10 XL3 PB2 full");
    RUN(512, 20, 0, 6, 2, "512x20 XL6 PB2 full");
    RUN(512, 20, 0, 0, 2, "512x20 XL0 PB2 full");
  }
  return 0;
}
