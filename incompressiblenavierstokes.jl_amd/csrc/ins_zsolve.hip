// Fused z-pass of the spectral Poisson solve (pressure.jl:323-344 restricted to the z direction):
//     forward FFT along z  ->  p̂ *= -1/(âx+ây+âz)/prod(Np), mean mode zeroed  ->  inverse FFT along z
// in ONE kernel and one trip through HBM (rocFFT needs three: z-FFT, the symbol kernel, inverse z-FFT).
//
// Data: complex doubles laid out [kz][line], `line` = (ky, kx) flattened and contiguous (the output of a
// batched 2-D R2C over (x, y)).  A workgroup owns TK consecutive lines (a TK*16-byte contiguous segment of
// every z-row), stages the TK x nz tile in LDS and runs the transform there:
//   * forward = in-place decimation-in-frequency (radix 4, plus one radix-2 stage when log2(nz) is odd):
//     natural order in, digit-reversed order out;
//   * the symbol is applied to the digit-reversed spectrum (index map computed per element);
//   * inverse = in-place decimation-in-time with the mirrored stage order and conjugated twiddles:
//     digit-reversed in, natural order out.  No reordering pass is ever needed.
// nz must be a power of two in [8, 1024]; other sizes keep the rocFFT path.
#include <cmath>
#include <cstdlib>

#include "ins_internal.h"

namespace {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {  // a * conj(b)
  return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ double2 mul_pi(double2 a) { return make_double2(-a.y, a.x); }  // a * (+i)

// Frequency index held at storage position p after the DIF stages (radix-2 first when ODD, then radix 4).
template <int LOGN>
__device__ __forceinline__ int freq_of_pos(int p) {
  constexpr bool ODD = LOGN & 1;
  int k = 0, mult = 1, L = 1 << LOGN;
  if (ODD) {
    const int q = p / (L / 2);
    p -= q * (L / 2);
    k += q * mult;
    mult *= 2;
    L /= 2;
  }
#pragma unroll
  for (int s = 0; s < LOGN / 2; ++s) {
    const int q = p / (L / 4);
    p -= q * (L / 4);
    k += q * mult;
    mult *= 4;
    L /= 4;
  }
  return k;
}

template <int LOGN, int TK>
__global__ __launch_bounds__(256) void k_zsolve(double2* __restrict__ data, long long nl, const double* __restrict__ ax, int kxn,
                                                const double* __restrict__ ay, const double* __restrict__ az,
                                                const double2* __restrict__ tw_g, double inv_n, int zero_mean, int kxs, int skel) {
  constexpr int N = 1 << LOGN;
  constexpr bool ODD = LOGN & 1;
  extern __shared__ double2 lds_dyn[];  // dynamic: tiles above 64 KB need the opt-in limit
  double2* buf = lds_dyn;               // [N][TK]
  double2* tw = lds_dyn + N * TK;       // [N]
  const int t = threadIdx.x;
  const long long l0 = (long long)blockIdx.x * TK;
  const int col = t % TK;
  const long long line = l0 + col;
  // lines are (ky, kx) pairs stored with row stride kxs >= kxn; the padding columns hold nothing
  const int lkx = (int)(line % kxs);
  const bool live = line < nl && lkx < kxn;
  for (int m = t; m < N; m += 256) tw[m] = tw_g[m];
  constexpr int RPT = 256 / TK;  // z-rows covered by one sweep of the workgroup
  constexpr int NIT = N / RPT;  // all loads of a work-item in flight before the first LDS write
  {
    double2 v[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) v[q] = live ? data[(long long)(t / TK + q * RPT) * nl + line] : make_double2(0.0, 0.0);
#pragma unroll
    for (int q = 0; q < NIT; ++q) buf[(t / TK + q * RPT) * TK + col] = v[q];
  }
  // 1/((âx + ây) + âz): the (x,y) part of the symbol is fixed per line
  double axy = 1.0;
  if (live) axy = ax[lkx] + ay[(int)(line / kxs)];
  const bool mean_line = zero_mean && line == 0;
  __syncthreads();

  // ---------------- forward: DIF ----------------
  int L = N;
  if (!skel) {  // skel: timing experiment (INS_ZSOLVE_SKEL): loads and stores only
  if (ODD) {
    for (int w = t; w < (N / 2) * TK; w += 256) {
      const int c = w % TK, j = w / TK;  // one group of length N
      double2 a0 = buf[j * TK + c], a1 = buf[(j + N / 2) * TK + c];
      buf[j * TK + c] = cadd(a0, a1);
      buf[(j + N / 2) * TK + c] = cmul(csub(a0, a1), tw[j]);
    }
    L = N / 2;
    __syncthreads();
  }
#pragma unroll 1
  for (; L >= 4; L >>= 2) {
    const int Q = L / 4, step = N / L;
    const bool last = L == 4;
    for (int w = t; w < (N / 4) * TK; w += 256) {
      const int c = w % TK, b = w / TK;
      const int g = b / Q, j = b - g * Q;
      const int base = (g * L + j) * TK + c;
      const double2 a0 = buf[base], a1 = buf[base + Q * TK], a2 = buf[base + 2 * Q * TK], a3 = buf[base + 3 * Q * TK];
      const double2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi(csub(a1, a3));
      double2 y0 = cadd(t0, t2), y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
      if (!last) {
        const double2 w1 = tw[j * step], w2 = cmul(w1, w1), w3 = cmul(w2, w1);  // one table read instead of three (LDS-bound stages)
        y1 = cmul(y1, w1);
        y2 = cmul(y2, w2);
        y3 = cmul(y3, w3);
      } else {
        // last forward stage (j == 0, unit twiddles): apply the symbol right here.  Position p holds frequency
        // freq_of_pos(p); c is this thread's own column, so `axy` / `mean_line` are the right ones.
        const int p0 = g * 4;
        double2* y[4] = {&y0, &y1, &y2, &y3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = freq_of_pos<LOGN>(p0 + q);
          const double den = axy + az[k];
          const double s = (mean_line && k == 0) ? 0.0 : -inv_n / den;
          y[q]->x *= s;
          y[q]->y *= s;
        }
        // ... and the first inverse stage (L = 4, unit twiddles) works on the same four values: do it now, one LDS round trip
        // and one barrier fewer (the inverse loop below starts at L = 16)
        const double2 u0 = cadd(y0, y2), u1 = csub(y0, y2), u2 = cadd(y1, y3), u3 = mul_pi(csub(y1, y3));
        y0 = cadd(u0, u2);
        y1 = cadd(u1, u3);
        y2 = csub(u0, u2);
        y3 = csub(u1, u3);
      }
      buf[base] = y0;
      buf[base + Q * TK] = y1;
      buf[base + 2 * Q * TK] = y2;
      buf[base + 3 * Q * TK] = y3;
    }
    __syncthreads();
  }

  // ---------------- inverse: DIT, mirrored stage order, conjugated twiddles ----------------
#pragma unroll 1
  for (L = 16; L <= (ODD ? N / 2 : N); L <<= 2) {
    const int Q = L / 4, step = N / L;
    for (int w = t; w < (N / 4) * TK; w += 256) {
      const int c = w % TK, b = w / TK;
      const int g = b / Q, j = b - g * Q;
      const int base = (g * L + j) * TK + c;
      double2 x0 = buf[base], x1 = buf[base + Q * TK], x2 = buf[base + 2 * Q * TK], x3 = buf[base + 3 * Q * TK];
      if (L > 4) {
        const double2 w1 = tw[j * step], w2 = cmul(w1, w1), w3 = cmul(w2, w1);
        x1 = cmulc(x1, w1);
        x2 = cmulc(x2, w2);
        x3 = cmulc(x3, w3);
      }
      const double2 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = mul_pi(csub(x1, x3));
      buf[base] = cadd(t0, t2);
      buf[base + Q * TK] = cadd(t1, t3);
      buf[base + 2 * Q * TK] = csub(t0, t2);
      buf[base + 3 * Q * TK] = csub(t1, t3);
    }
    __syncthreads();
  }
  if (ODD) {
    for (int w = t; w < (N / 2) * TK; w += 256) {
      const int c = w % TK, j = w / TK;
      const double2 x0 = buf[j * TK + c], x1 = cmulc(buf[(j + N / 2) * TK + c], tw[j]);
      buf[j * TK + c] = cadd(x0, x1);
      buf[(j + N / 2) * TK + c] = csub(x0, x1);
    }
    __syncthreads();
  }
  }
  if (live)
#pragma unroll
    for (int q = 0; q < NIT; ++q) data[(long long)(t / TK + q * RPT) * nl + line] = buf[(t / TK + q * RPT) * TK + col];
}

template <int LOGN, int TK>
int launch_zsolve(double2* data, long long nl, const double* ax, int kxn, const double* ay, const double* az, const double2* tw,
                  double inv_n, bool zero_mean, hipStream_t s, int kxs) {
  const unsigned nb = (unsigned)((nl + TK - 1) / TK);
  constexpr size_t lds = ((size_t)(1 << LOGN) * TK + (1 << LOGN)) * sizeof(double2);
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    INS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_zsolve<LOGN, TK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_zsolve<LOGN, TK>), dim3(nb), dim3(256), lds, s, data, nl, ax, kxn, ay, az, tw, inv_n, zero_mean ? 1 : 0, kxs, ins_opt(OPT_INS_ZSOLVE_SKEL) ? 1 : 0);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

}  // namespace

bool ins_zsolve_supported(int nz) {
  if (ins_opt(OPT_INS_DISABLE_ZSOLVE)) return false;
  return nz >= 16 && nz <= 1024 && (nz & (nz - 1)) == 0;
}

// Twiddles W_nz^m = exp(-2πi m / nz), m = 0..nz-1, on the device (caller frees).
int ins_zsolve_twiddles(int nz, double** out) {
  std::vector<double> h(2 * (size_t)nz);
  for (int m = 0; m < nz; ++m) {
    // octant symmetry keeps the table accurate to the last bit where it matters
    const double a = -2.0 * M_PI * (double)m / (double)nz;
    h[2 * m] = std::cos(a);
    h[2 * m + 1] = std::sin(a);
  }
  for (int m = 0; m < nz; ++m) {  // exact values on the axes
    if ((4 * m) % nz == 0) {
      const int q = (4 * m) / nz;  // multiples of a quarter turn
      const double c[4] = {1, 0, -1, 0}, sn[4] = {0, -1, 0, 1};
      h[2 * m] = c[q];
      h[2 * m + 1] = sn[q];
    }
  }
  double* d = nullptr;
  INS_HIP_TRY(hipMalloc(&d, h.size() * sizeof(double)));
  hipError_t e = hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d);
    ins_set_error("twiddle upload: %s", hipGetErrorString(e));
    return INS_ERR_HIP;
  }
  *out = d;
  return INS_OK;
}

// data[kz][line] (line = ky*kxs + kx, nl = kxs * nky lines of which kx < kxn are live), in place.  kxs <= 0: kxs = kxn.
int ins_k_zsolve(double* data, int nz, long long nl, const double* ax, int kxn, const double* ay, const double* az, const double* tw,
                 double inv_n, bool zero_mean, hipStream_t s, int kxs) {
  if (kxs <= 0) kxs = kxn;
  double2* d = reinterpret_cast<double2*>(data);
  const double2* w = reinterpret_cast<const double2*>(tw);
  switch (nz) {
    case 16: return launch_zsolve<4, 16>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 32: return launch_zsolve<5, 16>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 64: return launch_zsolve<6, 16>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 128: return launch_zsolve<7, 16>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 256: return launch_zsolve<8, 8>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 512: return launch_zsolve<9, 8>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);    // 72 KB tile
    case 1024: return launch_zsolve<10, 4>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);  // 80 KB tile
  }
  ins_set_error("ins_k_zsolve: unsupported nz = %d", nz);
  return INS_ERR_UNSUPPORTED;
}
