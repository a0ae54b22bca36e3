// Developer micro-benchmark (not part of the product): one workgroup per rollout runs K iterations of the
// LDS/register-resident block-Jacobi PCG on a synthetic 13-point grid matrix in wave-sliced ELL, to tune the
// core loop of dc_forward_res.hip in isolation.   hipcc --offload-arch=gfx950 -O3 tools/cg_bench.hip -o cg_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

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

// VARIANT 0: r, x, Ap all in registers.  VARIANT 1: Ap through a float4 global scratch.
template <int THREADS, int VPT, int VARIANT>
__global__ __launch_bounds__(THREADS) void k_cg(const int2 *__restrict__ ell, const int *__restrict__ ell_ptr,
                                                const int *__restrict__ ell_w, const float *__restrict__ dinv,
                                                const float *__restrict__ rhs, float *__restrict__ xout, float4 *__restrict__ scratch,
                                                int iters, long long *cycles) {
  constexpr int NP = THREADS * VPT;
  extern __shared__ float lp[];
  __shared__ double red[THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  float rr[VPT][3], xx[VPT][3], ap[VARIANT == 0 ? VPT : 1][3];
  float part = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; k++) {
    const int i = tid + k * THREADS;
    const float di = dinv[i];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      rr[k][c] = rhs[((size_t) b * 3 + c) * NP + i];
      xx[k][c] = 0.f;
      lp[c * NP + i] = rr[k][c] * di;
      part = fmaf(rr[k][c] * di, rr[k][c], part);
    }
  }
  double rz = block_sum<THREADS>((double) part, red);
  float4 *sc = scratch + (size_t) b * NP;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    __syncthreads();
    part = 0.f;
#pragma unroll
    for (int kk = 0; kk < VPT; kk++) {
      const int k = kk;
      const int i = tid + k * THREADS;
      const int chunk = i >> 6;
      const int2 *row = ell + ell_ptr[chunk] + lane;
      const int w = ell_w[chunk];
      float ax = 0.f, ay = 0.f, az = 0.f;
      for (int s0 = 0; s0 < w; s0 += 8) {
        int2 e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { e[j] = row[min(s0 + j, w - 1) * 64]; e[j].y = (s0 + j < w) ? e[j].y : 0; }
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float a = __int_as_float(e[j].y);
          ax = fmaf(a, lp[e[j].x], ax); ay = fmaf(a, lp[NP + e[j].x], ay); az = fmaf(a, lp[2 * NP + e[j].x], az);
        }
      }
      if (VARIANT == 0) { ap[VARIANT == 0 ? k : 0][0] = ax; ap[VARIANT == 0 ? k : 0][1] = ay; ap[VARIANT == 0 ? k : 0][2] = az; }
      else sc[i] = make_float4(ax, ay, az, 0.f);
      part += lp[i] * ax + lp[NP + i] * ay + lp[2 * NP + i] * az;
      __builtin_amdgcn_sched_barrier(0);
    }
    const double pAp = block_sum<THREADS>((double) part, red);
    const float alpha = (float) (rz / pAp);
    part = 0.f;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int i = tid + k * THREADS;
      const float di = dinv[i];
      float a3[3];
      if (VARIANT == 0) { a3[0] = ap[VARIANT == 0 ? k : 0][0]; a3[1] = ap[VARIANT == 0 ? k : 0][1]; a3[2] = ap[VARIANT == 0 ? k : 0][2]; }
      else { float4 q = sc[i]; a3[0] = q.x; a3[1] = q.y; a3[2] = q.z; }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        xx[k][c] = fmaf(alpha, lp[c * NP + i], xx[k][c]);
        rr[k][c] = fmaf(-alpha, a3[c], rr[k][c]);
        part = fmaf(rr[k][c] * di, rr[k][c], part);
      }
    }
    const double rz_new = block_sum<THREADS>((double) part, red);
    const float beta = (float) (rz_new / rz);
    rz = rz_new;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int i = tid + k * THREADS;
      const float di = dinv[i];
#pragma unroll
      for (int c = 0; c < 3; c++) lp[c * NP + i] = fmaf(beta, lp[c * NP + i], rr[k][c] * di);
    }
  }
  long long t1 = clock64();
#pragma unroll
  for (int k = 0; k < VPT; k++) {
    const int i = tid + k * THREADS;
#pragma unroll
    for (int c = 0; c < 3; c++) xout[((size_t) b * 3 + c) * NP + i] = xx[k][c];
  }
  if (tid == 0) cycles[b] = t1 - t0;
}

template <int THREADS, int VPT, int VARIANT>
void run(const char *name, int B, int N, int iters, const int2 *ell, const int *eptr, const int *ew, const float *dinv,
         const float *rhs, float *xout, float4 *scratch, long long *cyc) {
  const size_t lds = (size_t) 3 * THREADS * VPT * sizeof(float);
  CK(hipFuncSetAttribute((const void *) k_cg<THREADS, VPT, VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_cg<THREADS, VPT, VARIANT>), dim3(B), dim3(THREADS), lds, 0, ell, eptr, ew, dinv, rhs, xout, scratch, iters, cyc);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    long long c0;
    CK(hipMemcpy(&c0, cyc, sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<float> x(8);
    CK(hipMemcpy(x.data(), xout, 8 * sizeof(float), hipMemcpyDeviceToHost));
    if (rep == 1) printf("%-28s B=%d iters=%d: %.3f ms  -> %.2f us/iter, block0 %.0f cycles/iter, x[0..2]= %g %g %g\n", name, B, iters, ms,
                         ms * 1e3 / iters, (double) c0 / iters, x[0], x[1], x[2]);
  }
}

int main(int argc, char **argv) {
  const int G = 100, N = G * G, NP = 10240, B = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 200;
  // 13-point stencil: self, 6 edge neighbours, 6 flap opposites of the triangulated grid (SPD, diagonally dominant)
  const int off[13][2] = {{0, 0}, {0, 1}, {0, -1}, {1, 0}, {-1, 0}, {1, -1}, {-1, 1}, {1, 1}, {-1, -1}, {2, -1}, {-2, 1}, {1, -2}, {-1, 2}};
  const int nchunks = NP / 64;
  std::vector<int> eptr(nchunks), ew(nchunks), flat;
  std::vector<std::vector<std::pair<int, float>>> rows(NP);
  for (int r = 0; r < N; r++) {
    int gi = r / G, gj = r % G;
    float diag = 0.0006f;
    for (int k = 1; k < 13; k++) {
      int a = gi + off[k][0], c = gj + off[k][1];
      if (a < 0 || c < 0 || a >= G || c >= G) continue;
      float v = -0.002f - 0.0001f * (k % 3);
      rows[r].push_back({a * G + c, v});
      diag += -v;
    }
    rows[r].push_back({r, diag});
    std::sort(rows[r].begin(), rows[r].end());
  }
  std::vector<float> dinv(NP, 0.f);
  for (int r = 0; r < N; r++) for (auto &p : rows[r]) if (p.first == r) dinv[r] = 1.0f / p.second;
  for (int ch = 0; ch < nchunks; ch++) {
    int w = 0;
    for (int l = 0; l < 64; l++) w = std::max(w, (int) rows[64 * ch + l].size());
    eptr[ch] = (int) flat.size() / 2; ew[ch] = w;
    flat.resize(flat.size() + (size_t) 2 * 64 * w);
    for (int s = 0; s < w; s++) for (int l = 0; l < 64; l++) {
      int r = 64 * ch + l, col = std::min(r, N - 1); float val = 0.f;
      if (s < (int) rows[r].size()) { col = rows[r][s].first; val = rows[r][s].second; }
      int bits; memcpy(&bits, &val, 4);
      size_t o = 2 * ((size_t) eptr[ch] + (size_t) s * 64 + l);
      flat[o] = col; flat[o + 1] = bits;
    }
  }
  std::vector<float> rhs((size_t) B * 3 * NP, 0.f);
  for (int b = 0; b < B; b++) for (int c = 0; c < 3; c++) for (int i = 0; i < N; i++) rhs[((size_t) b * 3 + c) * NP + i] = 1e-3f * sinf(0.01f * i + c + b);
  int *d_flat, *d_eptr, *d_ew; float *d_dinv, *d_rhs, *d_x; float4 *d_sc; long long *d_cyc;
  CK(hipMalloc(&d_flat, flat.size() * 4)); CK(hipMemcpy(d_flat, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_eptr, nchunks * 4)); CK(hipMemcpy(d_eptr, eptr.data(), nchunks * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_ew, nchunks * 4)); CK(hipMemcpy(d_ew, ew.data(), nchunks * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_dinv, NP * 4)); CK(hipMemcpy(d_dinv, dinv.data(), NP * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_rhs, rhs.size() * 4)); CK(hipMemcpy(d_rhs, rhs.data(), rhs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_x, rhs.size() * 4));
  CK(hipMalloc(&d_sc, (size_t) B * NP * sizeof(float4)));
  CK(hipMalloc(&d_cyc, B * sizeof(long long)));
  printf("matrix: %zu bytes ELL, N=%d\n", flat.size() * 4, N);
  run<1024, 10, 0>("1024x10 Ap in regs", B, N, iters, (int2 *) d_flat, d_eptr, d_ew, d_dinv, d_rhs, d_x, d_sc, d_cyc);
  run<1024, 10, 1>("1024x10 Ap via scratch", B, N, iters, (int2 *) d_flat, d_eptr, d_ew, d_dinv, d_rhs, d_x, d_sc, d_cyc);
  run<512, 20, 0>("512x20 Ap in regs", B, N, iters, (int2 *) d_flat, d_eptr, d_ew, d_dinv, d_rhs, d_x, d_sc, d_cyc);
  run<512, 20, 1>("512x20 Ap via scratch", B, N, iters, (int2 *) d_flat, d_eptr, d_ew, d_dinv, d_rhs, d_x, d_sc, d_cyc);
  run<256, 40, 0>("256x40 Ap in regs", B, N, iters, (int2 *) d_flat, d_eptr, d_ew, d_dinv, d_rhs, d_x, d_sc, d_cyc);
  return 0;
}
