// Access-pattern ceiling probe for the K1 stencil on MI355X: pure copies (u -> F, 3 components of a
// padded N^3 field) with the candidate traversal orders.  Build: hipcc -O3 --offload-arch=gfx950
// tools/membench.hip -o tools/membench ;  run: tools/membench [n]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));             \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// A: flat 16 B / lane
__global__ __launch_bounds__(256) void k_flat16(const double2* __restrict__ a, double2* __restrict__ b, long long n2) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n2; t += (long long)gridDim.x * 256) b[t] = a[t];
}
// B: flat 8 B / lane
__global__ __launch_bounds__(256) void k_flat8(const double* __restrict__ a, double* __restrict__ b, long long n) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) b[t] = a[t];
}

// flat, 4 x 16 B loads in flight per lane before the stores; NT = non-temporal stores
template <bool NT>
__global__ __launch_bounds__(256) void k_flat16x4(const double2* __restrict__ a, double2* __restrict__ b, long long n2) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n2; t += 4 * stride) {
    double2 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (t + q * stride < n2) ? a[t + q * stride] : make_double2(0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (t + q * stride < n2) {
        if (NT) {
          __builtin_nontemporal_store(v[q].x, &b[t + q * stride].x);
          __builtin_nontemporal_store(v[q].y, &b[t + q * stride].y);
        } else
          b[t + q * stride] = v[q];
      }
  }
}
__global__ __launch_bounds__(256) void k_read16(const double2* __restrict__ a, double* __restrict__ out, long long n2) {
  double acc = 0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n2; t += (long long)gridDim.x * 256) {
    double2 v = a[t];
    acc += v.x + v.y;
  }
  if (acc == 1.2345) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write16(double2* __restrict__ b, long long n2) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n2; t += (long long)gridDim.x * 256) b[t] = make_double2(1.0, 2.0);
}

// Row-block march: a workgroup of 4 waves covers 256 consecutive x-columns (aligned) of R rows and marches z.
template <int R, int DEPTH, bool NT>
__global__ __launch_bounds__(256) void k_rowmarch(const double* __restrict__ u, double* __restrict__ F, int N, long long sc, int chunk,
                                                  int ntx, int nty, int ntz) {
  const int nb = gridDim.x, per = nb >> 3;
  int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (t >= ntx * nty * ntz) return;
  const int txi = t % ntx;
  t /= ntx;
  const int tyi = t % nty;
  const int tzi = t / nty;
  const int i = txi * 256 + threadIdx.x;
  const int ic = min(i, N - 1);
  const bool xout = i <= N - 1;
  const int jb = tyi * R;
  const int k0 = tzi * chunk, k1 = min(k0 + chunk, N);
  long long off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) off[r] = (long long)min(jb + r, N - 1) * N + ic;
  const long long sz = (long long)N * N;
  double buf[DEPTH][3][R];
  auto load = [&](double (&B)[3][R], int k) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) B[c][r] = u[c * sc + off[r] + (long long)min(k, N - 1) * sz];
  };
  auto store = [&](double (&B)[3][R], int k) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (xout && jb + r <= N - 1) {
          double* p = &F[c * sc + (long long)(jb + r) * N + i + (long long)k * sz];
          if (NT)
            __builtin_nontemporal_store(B[c][r] + 1.0, p);
          else
            *p = B[c][r] + 1.0;
        }
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) load(buf[d], k0 + d);
  int k = k0;
  while (true) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      load(buf[(d + DEPTH - 1) % DEPTH], k + DEPTH - 1);
      store(buf[d], k);
      if (++k >= k1) return;
    }
  }
}

template <int R, int DEPTH, bool NT>
float run_rowmarch(const double* u, double* F, int N, int chunk, int iters) {
  const int ntx = (N + 255) / 256, nty = (N + R - 1) / R, ntz = (N + chunk - 1) / chunk;
  const long long ntiles = (long long)ntx * nty * ntz;
  const unsigned nb = (unsigned)((ntiles + 7) / 8 * 8);
  const long long sc = (long long)N * N * N;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_rowmarch<R, DEPTH, NT>), dim3(nb), dim3(256), 0, 0, u, F, N, sc, chunk, ntx, nty, ntz);
  CHECK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k_rowmarch<R, DEPTH, NT>), dim3(nb), dim3(256), 0, 0, u, F, N, sc, chunk, ntx, nty, ntz);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

// Plane sweep: k-major traversal.  XCD e (= blockIdx & 7) owns y-range e; within an XCD tiles go x fastest,
// then y, then k (slowest), so all CUs sit inside a ~1-plane window and each XCD's L2 sees its slab of the
// neighbouring planes again one plane later.  NP = planes read per output plane (3 emulates the stencil's
// k-1, k, k+1 reads; the extra ones should be L2/MALL hits).  ZC = planes marched per block.
template <int R, int NP, int ZC, bool NT>
__global__ __launch_bounds__(256) void k_sweep(const double* __restrict__ u, double* __restrict__ F, int N, long long sc, int ntx,
                                               int nty_local, int nchunk) {
  const int xcd = blockIdx.x & 7;
  int seq = blockIdx.x >> 3;
  if (seq >= ntx * nty_local * nchunk) return;
  const int txi = seq % ntx;
  seq /= ntx;
  const int tyl = seq % nty_local;
  const int kc = seq / nty_local;
  const int i = txi * 256 + threadIdx.x;
  const int ic = min(i, N - 1);
  const bool xout = i <= N - 1;
  const int jb = (xcd * nty_local + tyl) * R;
  if (jb > N - 1) return;
  const long long sz = (long long)N * N;
  long long off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) off[r] = (long long)min(jb + r, N - 1) * N + ic;
#pragma unroll
  for (int q = 0; q < ZC; ++q) {
    const int k = kc * ZC + q;
    if (k > N - 1) return;
    double v[3][R];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double a = u[c * sc + off[r] + (long long)k * sz];
        if (NP == 3) a += u[c * sc + off[r] + (long long)max(k - 1, 0) * sz] + u[c * sc + off[r] + (long long)min(k + 1, N - 1) * sz];
        v[c][r] = a;
      }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (xout && jb + r <= N - 1) {
          double* p = &F[c * sc + (long long)(jb + r) * N + i + (long long)k * sz];
          if (NT)
            __builtin_nontemporal_store(v[c][r] + 1.0, p);
          else
            *p = v[c][r] + 1.0;
        }
  }
}

template <int R, int NP, int ZC, bool NT>
float run_sweep(const double* u, double* F, int N, int iters) {
  const int ntx = (N + 255) / 256;
  const int nty = (N + R - 1) / R;
  const int nty_local = (nty + 7) / 8;
  const int nchunk = (N + ZC - 1) / ZC;
  const unsigned nb = (unsigned)(8LL * ntx * nty_local * nchunk);
  const long long sc = (long long)N * N * N;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_sweep<R, NP, ZC, NT>), dim3(nb), dim3(256), 0, 0, u, F, N, sc, ntx, nty_local, nchunk);
  CHECK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k_sweep<R, NP, ZC, NT>), dim3(nb), dim3(256), 0, 0, u, F, N, sc, ntx, nty_local, nchunk);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

// C/D/E: tile march.  Wave = 64 lanes along x; each wave owns R rows; marches MARCH_Z ? z : y.
//   HALO: lanes 1..62 store (62-column tiles, like the DPP flux kernel) else all 64 lanes (aligned tiles).
//   DEPTH: planes of load-ahead (register buffers).
template <int R, bool MARCH_Z, bool HALO, int DEPTH>
__global__ __launch_bounds__(256) void k_march(const double* __restrict__ u, double* __restrict__ F, int N, long long sc, int chunk,
                                               int ntx, int nt1, int nt2) {
  const int nb = gridDim.x, per = nb >> 3;
  int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (t >= ntx * nt1 * nt2) return;
  const int t1 = t % nt1;
  t /= nt1;
  const int txi = t % ntx;
  const int t2 = t / ntx;
  const int lane = threadIdx.x, wave = threadIdx.y;
  const int XO = HALO ? 62 : 64;
  const int i = HALO ? txi * XO + lane : txi * XO + lane;
  const int ic = min(i, N - 1);
  const bool xout = HALO ? (lane >= 1 && lane <= 62 && i <= N - 2) : (i <= N - 1);
  // fixed direction (rows) and marching direction
  const int fb = (HALO ? 1 : 0) + (t1 * 4 + wave) * R;  // first fixed-dir index
  if (fb > N - 1) return;
  const int m0 = (HALO ? 1 : 0) + t2 * chunk, m1 = min(m0 + chunk, N - (HALO ? 1 : 0));
  const long long sf = MARCH_Z ? (long long)N : (long long)N * N;  // stride of the fixed (row) direction
  const long long sm = MARCH_Z ? (long long)N * N : (long long)N;  // stride of the marching direction
  long long off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) off[r] = (long long)min(fb + r, N - 1) * sf + ic;
  double buf[DEPTH][3][R];
  auto load = [&](double (&B)[3][R], int m) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) B[c][r] = u[c * sc + off[r] + (long long)min(m, N - 1) * sm];
  };
  auto store = [&](double (&B)[3][R], int m) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (xout && fb + r <= N - 1) F[c * sc + (long long)(fb + r) * sf + i + (long long)m * sm] = B[c][r] + 1.0;
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) load(buf[d], m0 + d);
  int m = m0;
  while (true) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      load(buf[(d + DEPTH - 1) % DEPTH], m + DEPTH - 1);
      store(buf[d], m);
      if (++m >= m1) return;
    }
  }
}

template <int R, bool MARCH_Z, bool HALO, int DEPTH>
float run_march(const double* u, double* F, int N, int chunk, int iters) {
  const int XO = HALO ? 62 : 64;
  const int span = HALO ? N - 2 : N;
  const int ntx = (span + XO - 1) / XO, nt1 = (span + 4 * R - 1) / (4 * R), nt2 = (span + chunk - 1) / chunk;
  const long long ntiles = (long long)ntx * nt1 * nt2;
  const unsigned nb = (unsigned)((ntiles + 7) / 8 * 8);
  const long long sc = (long long)N * N * N;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_march<R, MARCH_Z, HALO, DEPTH>), dim3(nb), dim3(64, 4), 0, 0, u, F, N, sc, chunk, ntx, nt1, nt2);
  CHECK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it)
    hipLaunchKernelGGL((k_march<R, MARCH_Z, HALO, DEPTH>), dim3(nb), dim3(64, 4), 0, 0, u, F, N, sc, chunk, ntx, nt1, nt2);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int N = n + 2;
  const long long ncell = (long long)N * N * N;
  const size_t bytes = (size_t)ncell * 3 * sizeof(double);
  double *u, *F;
  CHECK(hipMalloc(&u, bytes));
  CHECK(hipMalloc(&F, bytes));
  CHECK(hipMemset(u, 0, bytes));
  CHECK(hipMemset(F, 0, bytes));
  const double gb = 48.0 * (double)n * n * n / 1e9;  // same "algorithmic" bytes as K1
  const int iters = 10;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float ms;
  {
    const long long n2 = ncell * 3 / 2;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_flat16, dim3(8192), dim3(256), 0, 0, (const double2*)u, (double2*)F, n2);
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_flat16, dim3(8192), dim3(256), 0, 0, (const double2*)u, (double2*)F, n2);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("n=%d flat 16B/lane             : %.4f ms  %.0f GB/s (actual bytes %.0f GB/s)\n", n, ms / iters, gb / (ms / iters) * 1e3,
           2.0 * bytes / 1e9 / (ms / iters) * 1e3);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_flat8, dim3(8192), dim3(256), 0, 0, u, F, ncell * 3);
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_flat8, dim3(8192), dim3(256), 0, 0, u, F, ncell * 3);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("n=%d flat  8B/lane             : %.4f ms  %.0f GB/s\n", n, ms / iters, gb / (ms / iters) * 1e3);
  }
  {
    const long long n2 = ncell * 3 / 2;
#define FLAT(NAME, LAUNCH, BYTES)                                                                        \
  do {                                                                                                   \
    for (int w = 0; w < 2; ++w) LAUNCH;                                                                  \
    CHECK(hipEventRecord(e0));                                                                           \
    for (int it = 0; it < iters; ++it) LAUNCH;                                                           \
    CHECK(hipEventRecord(e1));                                                                           \
    CHECK(hipEventSynchronize(e1));                                                                      \
    CHECK(hipEventElapsedTime(&ms, e0, e1));                                                             \
    printf("n=%d %-28s: %.4f ms  actual %.0f GB/s\n", n, NAME, ms / iters, (BYTES) / 1e9 / (ms / iters) * 1e3); \
  } while (0)
    FLAT("flat16 x4 unroll", hipLaunchKernelGGL(k_flat16x4<false>, dim3(4096), dim3(256), 0, 0, (const double2*)u, (double2*)F, n2), 2.0 * bytes);
    FLAT("flat16 x4 unroll nt-store", hipLaunchKernelGGL(k_flat16x4<true>, dim3(4096), dim3(256), 0, 0, (const double2*)u, (double2*)F, n2), 2.0 * bytes);
    FLAT("flat16 x4 unroll 16k blocks", hipLaunchKernelGGL(k_flat16x4<false>, dim3(16384), dim3(256), 0, 0, (const double2*)u, (double2*)F, n2), 2.0 * bytes);
    FLAT("read-only 16B", hipLaunchKernelGGL(k_read16, dim3(8192), dim3(256), 0, 0, (const double2*)u, F, n2), 1.0 * bytes);
    FLAT("write-only 16B", hipLaunchKernelGGL(k_write16, dim3(8192), dim3(256), 0, 0, (double2*)F, n2), 1.0 * bytes);
  }
#define RUNSW(R, NP, ZC, NT)                                                                                      \
  do {                                                                                                            \
    float t = run_sweep<R, NP, ZC, NT>(u, F, N, iters);                                                           \
    printf("n=%d sweep R=%d planes-read=%d zc=%d %s: %.4f ms  %.0f GB/s\n", n, R, NP, ZC, NT ? "nt" : "  ", t, gb / t * 1e3); \
  } while (0)
  RUNSW(1, 1, 1, false);
  RUNSW(2, 1, 1, false);
  RUNSW(4, 1, 1, false);
  RUNSW(4, 1, 1, true);
  RUNSW(1, 3, 1, false);
  RUNSW(2, 3, 1, false);
  RUNSW(4, 3, 1, false);
  RUNSW(4, 3, 1, true);
  RUNSW(2, 3, 2, false);
  RUNSW(2, 3, 4, false);
  RUNSW(4, 3, 4, false);
  RUNSW(4, 3, 4, true);
#define RUNROW(R, DEPTH, NT, CH)                                                                                  \
  do {                                                                                                            \
    float t = run_rowmarch<R, DEPTH, NT>(u, F, N, CH, iters);                                                     \
    printf("n=%d rowmarch z R=%d depth=%d %s chunk=%3d: %.4f ms  %.0f GB/s\n", n, R, DEPTH, NT ? "nt" : "  ", CH, t, gb / t * 1e3); \
  } while (0)
  RUNROW(1, 3, false, 64);
  RUNROW(2, 3, false, 64);
  RUNROW(4, 2, false, 64);
  RUNROW(4, 3, false, 64);
  RUNROW(4, 3, true, 64);
  RUNROW(4, 3, false, 16);
  RUNROW(8, 2, false, 64);
  RUNROW(8, 2, true, 64);
  RUNROW(8, 2, false, 16);
#define RUN(R, MZ, HALO, DEPTH, CH)                                                                               \
  do {                                                                                                            \
    float t = run_march<R, MZ, HALO, DEPTH>(u, F, N, CH, iters);                                                  \
    printf("n=%d march %s R=%d %s depth=%d chunk=%3d: %.4f ms  %.0f GB/s\n", n, MZ ? "z" : "y", R, HALO ? "halo62 " : "align64", DEPTH, \
           CH, t, gb / t * 1e3);                                                                                  \
  } while (0)
  RUN(2, true, true, 2, 64);
  RUN(2, true, true, 3, 64);
  RUN(2, true, true, 4, 64);
  RUN(4, true, true, 2, 64);
  RUN(4, true, true, 3, 64);
  RUN(4, true, true, 4, 64);
  RUN(8, true, true, 2, 64);
  RUN(2, true, false, 3, 64);
  RUN(4, true, false, 3, 64);
  RUN(4, true, false, 4, 128);
  RUN(2, false, true, 3, 64);
  RUN(4, false, true, 3, 64);
  RUN(4, false, true, 4, 64);
  RUN(4, false, false, 3, 64);
  RUN(8, false, false, 2, 64);
  return 0;
}
