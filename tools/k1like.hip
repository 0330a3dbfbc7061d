// What bounds the stage kernels' memory rate?  A copy with K1's exact access pattern (tools/membench.hip has the flat / sweep / march orders):
// the k-major XCD-partitioned tile order of csrc/ins_flux64.hip, workgroups of 8 wavefronts with one barrier per plane, every wavefront loading R+2 rows
// of 3 components per plane (the two halo rows are shared with the y-neighbour wavefront) and storing R rows, marching a z-chunk — with
//   VEC = 1:  8 B per lane, 64 columns per wavefront, R = 4      (what k_flux64 does today: buffer_load_b64)
//   VEC = 2: 16 B per lane, 128 columns per wavefront, R = 2     (two x-columns per lane: buffer_load_b128), pairs starting at padded column 1
//                                                                (8-B-misaligned pairs: the interior starts at column 1) or at column 0 (aligned)
// and, independently, the base address of the arrays shifted by a few hundred bytes (allocation alignment / channel interleave of the 2064 / 4112-B rows).
// Build: hipcc -O3 --offload-arch=gfx950 tools/k1like.hip -o tools/k1like ; run: tools/k1like [n] [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                      \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

template <int VEC>
struct Val;
template <>
struct Val<1> {
  double x;
  __device__ void load(const double* p) { x = *p; }
  __device__ void store(double* p) const { *p = x; }
  __device__ void add(const Val& o) { x += o.x; }
};
template <>
struct Val<2> {
  v2d x;
  __device__ void load(const double* p) { x = *reinterpret_cast<const v2d*>(p); }  // 8-B aligned at least: global_load_dwordx4 takes it
  __device__ void store(double* p) const { *reinterpret_cast<v2d*>(p) = x; }
  __device__ void add(const Val& o) { x += o.x; }
};

// c0: padded column of lane 0 of the first wavefront (1: interior start, pairs misaligned by 8 B; 0: aligned pairs, lane 0 starts on the ghost column)
template <int VEC, int R, int XW, int NW, bool BAR>
__global__ __launch_bounds__(64 * NW) void k_k1like(const double* __restrict__ u, double* __restrict__ F, int N, long long sc, int zc, int ntx, int nty, int ntz,
                                                    int c0) {
  const int nty_local = (nty + 7) >> 3;
  int seq = (int)(blockIdx.x >> 3);
  if (seq >= ntx * nty_local * ntz) return;
  const int txi = seq % ntx;
  seq /= ntx;
  const int tyi = (int)(blockIdx.x & 7) * nty_local + seq % nty_local;
  const int tzi = seq / nty_local;
  if (tyi >= nty) return;
  const int lane = threadIdx.x, wave = threadIdx.y;
  const int wx = wave % XW, wy = wave / XW;
  const int n = N - 2;
  const int col = c0 + ((txi * XW + wx) * 64 + lane) * VEC;  // padded column of this lane's first element
  const int jb0 = (tyi * (NW / XW) + wy) * R;                 // interior row of the first output row
  const int k0 = 1 + tzi * zc, k1 = min(k0 + zc, N - 1);
  const bool active = col + VEC - 1 <= N - 1 && jb0 < n;
  const long long sz = (long long)N * N;
  const int colc = min(col, N - VEC);
  long long rowoff[R + 2];
#pragma unroll
  for (int rr = 0; rr < R + 2; ++rr) rowoff[rr] = (long long)min(jb0 + rr, N - 1) * N + colc;
  Val<VEC> P[2][3][R + 2];
  auto load_plane = [&](Val<VEC> (&Q)[3][R + 2], int k) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int rr = 0; rr < R + 2; ++rr) Q[c][rr].load(u + c * sc + (long long)k * sz + rowoff[rr]);
  };
  if (active) {
    load_plane(P[0], k0 - 1);
    load_plane(P[1], k0);
  }
  int k = k0;
  while (k < k1) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (BAR) __builtin_amdgcn_s_barrier();
      if (active) {
        // "plane k" = P[b^1]... consume P[b] (plane k-1 role) with P[b^1] (plane k), reload P[b] with plane k+1
        Val<VEC> out[3][R];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) {
            out[c][rr] = P[b ^ 1][c][rr + 1];
            out[c][rr].add(P[b ^ 1][c][rr]);
            out[c][rr].add(P[b ^ 1][c][rr + 2]);
            out[c][rr].add(P[b][c][rr + 1]);
          }
        load_plane(P[b], min(k + 1, N - 1));
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int rr = 0; rr < R; ++rr)
            if (jb0 + rr < n) out[c][rr].store(F + c * sc + (long long)k * sz + rowoff[rr + 1]);
      }
      if (++k >= k1) break;
    }
  }
}

template <int VEC, int R, int XW, int NW, bool BAR>
float run(const double* u, double* F, int N, int zc, int c0, int iters) {
  const int n = N - 2;
  const int ntx = (n + 64 * VEC * XW - 1) / (64 * VEC * XW);  // (c0 = 0: the last interior column pair is left out — 0.2 % of the bytes)
  const int nty = (n + (NW / XW) * R - 1) / ((NW / XW) * R);
  const int ntz = (n + zc - 1) / zc;
  const unsigned nb = 8u * ntx * ((nty + 7) / 8) * ntz;
  const long long sc = (long long)N * N * N;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_k1like<VEC, R, XW, NW, BAR>), dim3(nb), dim3(64, NW), 0, 0, u, F, N, sc, zc, ntx, nty, ntz, c0);
  CHECK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k_k1like<VEC, R, XW, NW, BAR>), dim3(nb), dim3(64, NW), 0, 0, u, F, N, sc, zc, ntx, nty, ntz, c0);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipGetLastError());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

__global__ void k_flat16(const v2d* __restrict__ a, v2d* __restrict__ b, long long n2) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n2; t += (long long)gridDim.x * 256) b[t] = a[t];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int iters = argc > 2 ? atoi(argv[2]) : 10;
  const int N = n + 2;
  const long long sc = (long long)N * N * N;
  const size_t bytes = (size_t)sc * 3 * sizeof(double);
  const size_t slack = 1 << 16;
  char *ub, *fb;
  CHECK(hipMalloc(&ub, bytes + slack));
  CHECK(hipMalloc(&fb, bytes + slack));
  CHECK(hipMemset(ub, 0, bytes + slack));
  CHECK(hipMemset(fb, 0, bytes + slack));
  const double alg = 2.0 * 3 * 8 * (double)n * n * n;  // algorithmic bytes (48 B per interior cell)
  printf("n=%d  u=%p F=%p  (hipMalloc bases; mod 4096 = %zu, %zu)\n", n, (void*)ub, (void*)fb, (size_t)ub % 4096, (size_t)fb % 4096);
  {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_flat16, dim3(16384), dim3(256), 0, 0, (const v2d*)ub, (v2d*)fb, (long long)(bytes / 16));
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_flat16, dim3(16384), dim3(256), 0, 0, (const v2d*)ub, (v2d*)fb, (long long)(bytes / 16));
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("flat 16 B/lane copy of the same arrays            : %.4f ms  %.0f GB/s (actual bytes)\n", ms, 2.0 * bytes / ms / 1e6);
  }
  const int shifts[] = {0, 16, 64, 128, 256, 1024, 2048};
  for (int zc : {64, 32}) {
    for (int sh : shifts) {
      const double* u = reinterpret_cast<const double*>(ub + sh);
      double* F = reinterpret_cast<double*>(fb + sh);
      if (sh != 0 && zc != 64) continue;
      float t;
      t = run<1, 4, 2, 8, true>(u, F, N, zc, 1, iters);
      printf("zc=%2d shift=%4d   8 B/lane R=4 XW=2 NW=8 bar             : %.4f ms  %.0f GB/s algorithmic\n", zc, sh, t, alg / t / 1e6);
      t = run<2, 2, 2, 8, true>(u, F, N, zc, 1, iters);
      printf("zc=%2d shift=%4d  16 B/lane R=2 XW=2 NW=8 bar (misaligned) : %.4f ms  %.0f GB/s algorithmic\n", zc, sh, t, alg / t / 1e6);
      t = run<2, 2, 2, 8, true>(u, F, N, zc, 0, iters);
      printf("zc=%2d shift=%4d  16 B/lane R=2 XW=2 NW=8 bar (aligned c0=0): %.4f ms  %.0f GB/s algorithmic\n", zc, sh, t, alg / t / 1e6);
      if (sh == 0) {
        t = run<2, 2, 1, 8, true>(u, F, N, zc, 0, iters);
        printf("zc=%2d shift=%4d  16 B/lane R=2 XW=1 NW=8 bar (aligned)      : %.4f ms  %.0f GB/s algorithmic\n", zc, sh, t, alg / t / 1e6);
        t = run<2, 4, 2, 8, true>(u, F, N, zc, 0, iters);
        printf("zc=%2d shift=%4d  16 B/lane R=4 XW=2 NW=8 bar (aligned)      : %.4f ms  %.0f GB/s algorithmic\n", zc, sh, t, alg / t / 1e6);
        t = run<2, 2, 2, 4, true>(u, F, N, zc, 0, iters);
        printf("zc=%2d shift=%4d  16 B/lane R=2 XW=2 NW=4 bar (aligned)      : %.4f ms  %.0f GB/s algorithmic\n", zc, sh, t, alg / t / 1e6);
        t = run<2, 2, 2, 8, false>(u, F, N, zc, 0, iters);
        printf("zc=%2d shift=%4d  16 B/lane R=2 XW=2 NW=8 nobar (aligned)    : %.4f ms  %.0f GB/s algorithmic\n", zc, sh, t, alg / t / 1e6);
        t = run<1, 4, 2, 8, false>(u, F, N, zc, 1, iters);
        printf("zc=%2d shift=%4d   8 B/lane R=4 XW=2 NW=8 nobar              : %.4f ms  %.0f GB/s algorithmic\n", zc, sh, t, alg / t / 1e6);
        t = run<1, 2, 2, 8, true>(u, F, N, zc, 1, iters);
        printf("zc=%2d shift=%4d   8 B/lane R=2 XW=2 NW=8 bar                : %.4f ms  %.0f GB/s algorithmic\n", zc, sh, t, alg / t / 1e6);
      }
    }
  }
  return 0;
}
