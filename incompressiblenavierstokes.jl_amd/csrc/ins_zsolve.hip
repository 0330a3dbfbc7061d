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
#include <vector>

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

// NT threads per workgroup.  The workgroup is PERSISTENT: it walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and loads the next tile
// into registers while the transform of the current one runs in LDS, so HBM stays busy during the LDS stages (the one-tile-per-workgroup
// form left HBM idle while a CU's workgroups computed: 0.83 ms at 512^3 against 0.64 ms for its loads and stores alone).
template <int LOGN, int TK, int NT>
__global__ __launch_bounds__(NT) void k_zsolve(double2* __restrict__ data, long long nl, const double* __restrict__ ax, int kxn,
                                               const double* __restrict__ ay, const double* __restrict__ az,
                                               const double2* __restrict__ tw_g, double inv_n, int zero_mean, int kxs, int skel, int ntiles) {
  constexpr int N = 1 << LOGN;
  constexpr bool ODD = LOGN & 1;
  extern __shared__ double2 lds_dyn[];  // dynamic: tiles above 64 KB need the opt-in limit
  double2* buf = lds_dyn;               // [N][TK]
  double2* tw = lds_dyn + N * TK;       // [N]
  const int t = threadIdx.x;
  const int col = t % TK;
  for (int m = t; m < N; m += NT) tw[m] = tw_g[m];
  constexpr int RPT = NT / TK;  // z-rows covered by one sweep of the workgroup
  constexpr int NIT = N / RPT;  // all loads of a work-item in flight before the first LDS write
  static_assert(N % RPT == 0 && NT % TK == 0, "tile shape");
  double2 v[NIT];
  auto tile_line = [&](int tile) { return (long long)tile * TK + col; };
  // lines are (ky, kx) pairs stored with row stride kxs >= kxn; the padding columns hold nothing
  auto is_live = [&](long long line) { return line < nl && (int)(line % kxs) < kxn; };
  auto prefetch = [&](int tile) {
    const long long line = tile_line(tile);
    const bool live = is_live(line);
#pragma unroll
    for (int q = 0; q < NIT; ++q) v[q] = live ? data[(long long)(t / TK + q * RPT) * nl + line] : make_double2(0.0, 0.0);
  };
  int tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
  const long long line = tile_line(tile);
  const int lkx = (int)(line % kxs);
  const bool live = is_live(line);
#pragma unroll
  for (int q = 0; q < NIT; ++q) buf[(t / TK + q * RPT) * TK + col] = v[q];
  // 1/((âx + ây) + âz): the (x,y) part of the symbol is fixed per line
  double axy = 1.0;
  if (live) axy = ax[lkx] + ay[(int)(line / kxs)];
  const bool mean_line = zero_mean && line == 0;
  __syncthreads();
  if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);  // in flight during the LDS stages below

  // ---------------- forward: DIF ----------------
  int L = N;
  if (!skel) {  // skel: timing experiment (INS_ZSOLVE_SKEL): loads and stores only
  if (ODD) {
    for (int w = t; w < (N / 2) * TK; w += NT) {
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
    for (int w = t; w < (N / 4) * TK; w += NT) {
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
    for (int w = t; w < (N / 4) * TK; w += NT) {
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
    for (int w = t; w < (N / 2) * TK; w += NT) {
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
  __syncthreads();  // the tile in LDS is overwritten by the next one
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Three-pass form for nz = 256 (4 x 8 x 8) and 512 (8 x 8 x 8): every pass is one radix-R1 / radix-8 transform held in REGISTERS.
//   forward (DIF):  pass 1 straight from global memory (a work-item loads the R1 planes j + q nz/R1 of its line — no LDS fill),
//                   pass 2 and pass 3 through LDS;  the symbol and the first inverse pass act on the registers of pass 3;
//   inverse (DIT, mirrored order, conjugated twiddles):  pass 2 through LDS, pass 3 reads LDS and stores straight to global memory.
// Four LDS round trips of the tile instead of the eleven of the radix-4 kernel above (its LDS traffic and twiddle arithmetic cost as
// much as its HBM traffic: 0.83 ms at 512^3 with 0.58 ms for loads and stores alone), and the tile never sits in LDS without work.
// LDS image: position n of line c at ((n ^ ((n >> 3) & 1)) * TK + c): the XOR keeps every access of the three passes conflict-free
// for ds_read/write_b128 (pass 3 reads positions 8 g + q: without it the sixteen lanes of a b128 group would share eight slots).
// ------------------------------------------------------------------------------------------------------------------------------
// (the three-pass kernel is a template over the complex element type C: double2, or float2 for the `_f32` family)
template <typename C>
struct zreal;
template <>
struct zreal<double2> {
  using t = double;
};
template <>
struct zreal<float2> {
  using t = float;
};
template <typename C>
__device__ __forceinline__ C zmk(typename zreal<C>::t x, typename zreal<C>::t y) {
  C c;
  c.x = x;
  c.y = y;
  return c;
}
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
template <bool INV, typename C>
__device__ __forceinline__ C rot90(C a) {  // a * (-i) forward, a * (+i) inverse
  return INV ? zmk<C>(-a.y, a.x) : zmk<C>(a.y, -a.x);
}
template <bool INV, typename C>
__device__ __forceinline__ void dft4(C& x0, C& x1, C& x2, C& x3) {
  const C t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = rot90<INV>(csub(x1, x3));
  x0 = cadd(t0, t2);
  x1 = cadd(t1, t3);
  x2 = csub(t0, t2);
  x3 = csub(t1, t3);
}
// y[k] = sum_n x[n] w^(nk), w = exp(-+2 pi i / R), in place, natural order in and out
template <bool INV, typename C>
__device__ __forceinline__ void dft3(C& x0, C& x1, C& x2) {
  using T = typename zreal<C>::t;
  constexpr T h = (T)0.86602540378443864676;  // sqrt(3)/2
  const C t1 = cadd(x1, x2), d = csub(x1, x2);
  const C t2 = zmk<C>(x0.x - (T)0.5 * t1.x, x0.y - (T)0.5 * t1.y);
  const C t3 = INV ? zmk<C>(-h * d.y, h * d.x) : zmk<C>(h * d.y, -h * d.x);  // (-+i) sqrt(3)/2 (x1 - x2)
  x0 = cadd(x0, t1);
  x1 = cadd(t2, t3);
  x2 = csub(t2, t3);
}
template <bool INV, typename C>
__device__ __forceinline__ void dft5(C& x0, C& x1, C& x2, C& x3, C& x4) {
  using T = typename zreal<C>::t;
  constexpr T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;  // cos(2 pi / 5), cos(4 pi / 5)
  constexpr T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;   // sin(2 pi / 5), sin(4 pi / 5)
  const C a = cadd(x1, x4), b = cadd(x2, x3), d = csub(x1, x4), e = csub(x2, x3);
  const C m1 = zmk<C>(x0.x + c1 * a.x + c2 * b.x, x0.y + c1 * a.y + c2 * b.y);
  const C m2 = zmk<C>(x0.x + c2 * a.x + c1 * b.x, x0.y + c2 * a.y + c1 * b.y);
  const C u = zmk<C>(s1 * d.x + s2 * e.x, s1 * d.y + s2 * e.y), v = zmk<C>(s2 * d.x - s1 * e.x, s2 * d.y - s1 * e.y);
  const C iu = rot90<INV>(u), iv = rot90<INV>(v);  // (-+i) u, (-+i) v
  x0 = cadd(x0, cadd(a, b));
  x1 = cadd(m1, iu);
  x4 = csub(m1, iu);
  x2 = cadd(m2, iv);
  x3 = csub(m2, iv);
}
template <int R, bool INV, typename C>
__device__ __forceinline__ void dft(C (&x)[R]) {
  using T = typename zreal<C>::t;
  if constexpr (R == 2) {
    const C a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
  } else if constexpr (R == 4) {
    dft4<INV>(x[0], x[1], x[2], x[3]);
  } else if constexpr (R == 3) {
    dft3<INV>(x[0], x[1], x[2]);
  } else if constexpr (R == 5) {
    dft5<INV>(x[0], x[1], x[2], x[3], x[4]);
  } else if constexpr (R == 10) {
    // even / odd halves (radix 5 each), then the radix-2 combination with W10^k = (cos, -+sin)(36 k degrees)
    dft5<INV>(x[0], x[2], x[4], x[6], x[8]);
    dft5<INV>(x[1], x[3], x[5], x[7], x[9]);
    constexpr T wc[5] = {(T)1, (T)0.80901699437494742410, (T)0.30901699437494742410, (T)-0.30901699437494742410, (T)-0.80901699437494742410};
    constexpr T ws[5] = {(T)0, (T)0.58778525229247312917, (T)0.95105651629515357212, (T)0.95105651629515357212, (T)0.58778525229247312917};
    C e[5], o[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      e[k] = x[2 * k];
      const C z = x[2 * k + 1];
      const T sn = INV ? ws[k] : -ws[k];
      o[k] = k == 0 ? z : zmk<C>(z.x * wc[k] - z.y * sn, z.x * sn + z.y * wc[k]);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      x[k] = cadd(e[k], o[k]);
      x[k + 5] = csub(e[k], o[k]);
    }
  } else if constexpr (R == 6) {
    // even / odd thirds (radix 3 each), then the radix-2 combination with W6^k
    dft3<INV>(x[0], x[2], x[4]);
    dft3<INV>(x[1], x[3], x[5]);
    constexpr T h = (T)0.86602540378443864676;
    const C o1 = x[3], o2 = x[5];
    // W6^1 = (1 -+ i sqrt3)/2, W6^2 = (-1 -+ i sqrt3)/2
    const C w1 = INV ? zmk<C>((T)0.5 * o1.x - h * o1.y, (T)0.5 * o1.y + h * o1.x) : zmk<C>((T)0.5 * o1.x + h * o1.y, (T)0.5 * o1.y - h * o1.x);
    const C w2 = INV ? zmk<C>((T)-0.5 * o2.x - h * o2.y, (T)-0.5 * o2.y + h * o2.x) : zmk<C>((T)-0.5 * o2.x + h * o2.y, (T)-0.5 * o2.y - h * o2.x);
    const C e0 = x[0], e1 = x[2], e2 = x[4], w0 = x[1];
    x[0] = cadd(e0, w0);
    x[3] = csub(e0, w0);
    x[1] = cadd(e1, w1);
    x[4] = csub(e1, w1);
    x[2] = cadd(e2, w2);
    x[5] = csub(e2, w2);
  } else {
    static_assert(R == 8, "radix");
    // even / odd halves (radix 4 each), then the radix-2 combination with W8^k
    dft4<INV>(x[0], x[2], x[4], x[6]);
    dft4<INV>(x[1], x[3], x[5], x[7]);
    constexpr T h = (T)0.70710678118654752440;
    const C o1 = x[3], o2 = x[5], o3 = x[7];
    // W8^1 = (1 -+ i)/sqrt2, W8^2 = -+i, W8^3 = (-1 -+ i)/sqrt2
    const C w1 = INV ? zmk<C>((o1.x - o1.y) * h, (o1.x + o1.y) * h) : zmk<C>((o1.x + o1.y) * h, (o1.y - o1.x) * h);
    const C w2 = rot90<INV>(o2);
    const C w3 = INV ? zmk<C>((-o3.x - o3.y) * h, (o3.x - o3.y) * h) : zmk<C>((o3.y - o3.x) * h, (-o3.x - o3.y) * h);
    const C e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], w0 = x[1];
    x[0] = cadd(e0, w0);
    x[4] = csub(e0, w0);
    x[1] = cadd(e1, w1);
    x[5] = csub(e1, w1);
    x[2] = cadd(e2, w2);
    x[6] = csub(e2, w2);
    x[3] = cadd(e3, w3);
    x[7] = csub(e3, w3);
  }
}
// x[q] *= w^q (q = 1..R-1), powers by multiplication (one table read per transform); CONJ: conjugated twiddle
template <int R, bool CONJ, typename C>
__device__ __forceinline__ void twiddle(C (&x)[R], C w1) {
  if (CONJ) w1.y = -w1.y;
  x[1] = cmul(x[1], w1);
  if constexpr (R == 2) return;
  const C w2 = cmul(w1, w1), w3 = cmul(w2, w1);
  x[2] = cmul(x[2], w2);
  if constexpr (R >= 4) x[3] = cmul(x[3], w3);
  if constexpr (R == 5 || R == 10) {
    const C w4 = cmul(w2, w2);
    x[4] = cmul(x[4], w4);
    if constexpr (R == 10) {
      const C w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3), w8 = cmul(w4, w4), w9 = cmul(w8, w1);
      x[5] = cmul(x[5], w5);
      x[6] = cmul(x[6], w6);
      x[7] = cmul(x[7], w7);
      x[8] = cmul(x[8], w8);
      x[9] = cmul(x[9], w9);
    }
  }
  if constexpr (R == 6) {
    const C w4 = cmul(w2, w2), w5 = cmul(w4, w1);
    x[4] = cmul(x[4], w4);
    x[5] = cmul(x[5], w5);
  }
  if constexpr (R == 8) {
    const C w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
    x[4] = cmul(x[4], w4);
    x[5] = cmul(x[5], w5);
    x[6] = cmul(x[6], w6);
    x[7] = cmul(x[7], w7);
  }
}

template <int LOGN, int TK, int NT, typename C = double2>
__global__ __launch_bounds__(NT) void k_zsolve3(C* __restrict__ data, long long nl, const double* __restrict__ ax, int kxn,
                                                const double* __restrict__ ay, const double* __restrict__ az,
                                                const C* __restrict__ tw_g, double inv_n, int zero_mean, int kxs, int skel, int ntiles) {
  using T = typename zreal<C>::t;
  // size code as in ins_fft.hip: 32 + m stands for 3 * 2^m, 64 + m for 5 * 2^m
  constexpr int N = LOGN >= 64 ? 5 << (LOGN & 31) : (LOGN >= 32 ? 3 << (LOGN & 31) : 1 << LOGN);
  // 256 = 4 x 8 x 8, 512 = 8 x 8 x 8, 192 = 3 x 8 x 8, 384 = 6 x 8 x 8, 320 = 5 x 8 x 8, 640 = 10 x 8 x 8; 96 = 3 x 4 x 8, 160 = 5 x 4 x 8
  constexpr int R3 = 8, R2 = N % 64 == 0 ? 8 : 4, R1 = N / (R2 * R3);
  static_assert(R1 * R2 * R3 == N && (R1 == 4 || R1 == 8 || R1 == 3 || R1 == 6 || R1 == 5 || R1 == 10), "nz = 96, 160, 192, 256, 320, 384, 512 or 640");
  constexpr int L2 = N / R1;                  // block length of pass 2 (64)
  extern __shared__ __align__(16) unsigned char lds_raw3[];
  C* buf = reinterpret_cast<C*>(lds_raw3);  // [N][TK], swizzled
  C* tw = buf + N * TK;                      // [N]
  const int t = threadIdx.x;
  auto at = [&](int n, int c) -> C& { return buf[(n ^ ((n >> 3) & 1)) * TK + c]; };
  for (int m = t; m < N; m += NT) tw[m] = tw_g[m];
  constexpr int RPI = NT / TK;  // transforms of one line column started per sweep of the workgroup
  constexpr int B1 = (N / R1 + RPI - 1) / RPI, B2 = (N / R2 + RPI - 1) / RPI, B3 = (N / R3 + RPI - 1) / RPI;  // sweeps per pass
  constexpr bool G2 = (N / R2) % RPI != 0 || (N / R3) % RPI != 0;  // 3 * 2^m, 5 * 2^m: the last sweep of passes 2 and 3 is partial
  static_assert((N / R1) % RPI == 0 && NT % TK == 0, "tile shape");
  const int c = t % TK;  // NT is a multiple of TK: a work-item keeps its line column in every pass
  auto is_live = [&](long long line) { return line < nl && (int)(line % kxs) < kxn; };  // padding columns of a row (kx >= kxn) hold nothing
  // The workgroup is persistent: tiles blockIdx.x, blockIdx.x + gridDim.x, ...; the pass-1 loads of the next tile are issued as soon as the
  // registers are free (right after pass 1 of the current tile), so they fly during the LDS passes and the stores of the current tile.
  C x1[B1][R1];
  auto prefetch = [&](int tile) {
    const long long line = (long long)tile * TK + c;
    const bool live = is_live(line);
#pragma unroll
    for (int i = 0; i < B1; ++i) {
      const int j = t / TK + i * (NT / TK);
#pragma unroll
      for (int q = 0; q < R1; ++q) x1[i][q] = live ? data[(long long)(j + q * (N / R1)) * nl + line] : zmk<C>(0, 0);
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  __syncthreads();  // twiddle table
  for (; tile < ntiles; tile += gridDim.x) {
    const long long line = (long long)tile * TK + c;
    const int lkx = (int)(line % kxs);
    const bool live = is_live(line);
    double axy = 1.0;
    if (live) axy = ax[lkx] + ay[(int)(line / kxs)];
    const bool mean_line = zero_mean && line == 0;

    // ---- forward pass 1: (global ->) registers -> LDS
#pragma unroll
    for (int i = 0; i < B1; ++i) {
      const int j = t / TK + i * (NT / TK);
      if (!skel) {
        dft<R1, false>(x1[i]);
        twiddle<R1, false>(x1[i], tw[j]);
      }
#pragma unroll
      for (int q = 0; q < R1; ++q) at(j + q * (N / R1), c) = x1[i][q];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
    if (!skel) {
      // ---- forward pass 2 (blocks of L2 = 64)
#pragma unroll
      for (int i = 0; i < B2; ++i) {
        const int b = t / TK + i * (NT / TK), g1 = b / (L2 / R2), j = b % (L2 / R2);
        if (G2 && b >= N / R2) break;
        C x[R2];
#pragma unroll
        for (int q = 0; q < R2; ++q) x[q] = at(g1 * L2 + j + q * (L2 / R2), c);
        dft<R2, false>(x);
        twiddle<R2, false>(x, tw[j * R1]);  // W_64^j = W_N^(R1 j)
#pragma unroll
        for (int q = 0; q < R2; ++q) at(g1 * L2 + j + q * (L2 / R2), c) = x[q];
      }
      __syncthreads();
      // ---- forward pass 3 (blocks of 8, unit twiddles) + symbol + inverse pass 1
#pragma unroll
      for (int i = 0; i < B3; ++i) {
        const int g = t / TK + i * (NT / TK);
        if (G2 && g >= N / R3) break;
        C x[R3];
#pragma unroll
        for (int q = 0; q < R3; ++q) x[q] = at(g * R3 + q, c);
        dft<R3, false>(x);
        // position p = q1 N/R1 + q2 N/(R1 R2) + q3 holds frequency k = q1 + R1 q2 + R1 R2 q3; here g = q1 R2 + q2, q3 = q
        const int kbase = g / R2 + R1 * (g % R2);
#pragma unroll
        for (int q = 0; q < R3; ++q) {
          const int k = kbase + R1 * R2 * q;
          const T sc = (T)((mean_line && k == 0) ? 0.0 : -inv_n / (axy + az[k]));
          x[q].x *= sc;
          x[q].y *= sc;
        }
        dft<R3, true>(x);
#pragma unroll
        for (int q = 0; q < R3; ++q) at(g * R3 + q, c) = x[q];
      }
      __syncthreads();
      // ---- inverse pass 2
#pragma unroll
      for (int i = 0; i < B2; ++i) {
        const int b = t / TK + i * (NT / TK), g1 = b / (L2 / R2), j = b % (L2 / R2);
        if (G2 && b >= N / R2) break;
        C x[R2];
#pragma unroll
        for (int q = 0; q < R2; ++q) x[q] = at(g1 * L2 + j + q * (L2 / R2), c);
        twiddle<R2, true>(x, tw[j * R1]);
        dft<R2, true>(x);
#pragma unroll
        for (int q = 0; q < R2; ++q) at(g1 * L2 + j + q * (L2 / R2), c) = x[q];
      }
      __syncthreads();
    }
    // ---- inverse pass 3: LDS -> registers -> global
#pragma unroll
    for (int i = 0; i < B1; ++i) {
      const int j = t / TK + i * (NT / TK);
      C x[R1];
#pragma unroll
      for (int q = 0; q < R1; ++q) x[q] = at(j + q * (N / R1), c);
      if (!skel) {
        twiddle<R1, true>(x, tw[j]);
        dft<R1, true>(x);
      }
      if (live)
#pragma unroll
        for (int q = 0; q < R1; ++q) data[(long long)(j + q * (N / R1)) * nl + line] = x[q];
    }
    __syncthreads();  // the LDS tile is rewritten by the next tile's pass 1
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// One direction of the transform ALONE on the register passes of k_zsolve3 (the y passes of the spectral solve: data[plane][ky][kx], a tile = TK
// consecutive kx of one plane x all N ky, elements of a line `es` = kxs complex values apart).  The LDS radix-4 kernel of ins_fft.hip (k_yfft) makes
// ten LDS accesses per element for its 32 B of memory traffic — on spectra that fit the Infinity Cache (256^3: 135 MB) and at 512^3 alike it runs at the
// rate of its LDS stages, 4.8 TB/s, not of the memory behind it; here an element crosses LDS twice (R1 x 8 x 8 in registers: 4 accesses).
//   forward: natural ky in; storage position p = q1 R2 R3 + q2 R3 + q3 holds frequency k = q1 + R1 q2 + R1 R2 q3 (ins_line3_permute_symbol)
//   inverse: that order in, natural ky out; unscaled both ways (the z pass carries the whole 1/N).
template <int LOGN, int TK, int NT, bool INV, typename C>
__global__ __launch_bounds__(NT) void k_line3(C* __restrict__ data, long long es, long long pstride, int kxn, int tiles_x, int ntiles,
                                              const C* __restrict__ tw_g) {
  constexpr int N = LOGN >= 64 ? 5 << (LOGN & 31) : (LOGN >= 32 ? 3 << (LOGN & 31) : 1 << LOGN);
  constexpr int R3 = 8, R2 = N % 64 == 0 ? 8 : 4, R1 = N / (R2 * R3);
  static_assert(R1 * R2 * R3 == N && (R1 == 2 || R1 == 4 || R1 == 8 || R1 == 3 || R1 == 6 || R1 == 5 || R1 == 10), "length");
  constexpr int L2 = N / R1;
  extern __shared__ __align__(16) unsigned char lds_raw_l3[];
  C* buf = reinterpret_cast<C*>(lds_raw_l3);  // [N][TK], swizzled
  C* tw = buf + N * TK;                        // [N]
  const int t = threadIdx.x;
  auto at = [&](int n, int c) -> C& { return buf[(n ^ ((n >> 3) & 1)) * TK + c]; };
  for (int m = t; m < N; m += NT) tw[m] = tw_g[m];
  constexpr int RPI = NT / TK;
  constexpr int B1 = (N / R1 + RPI - 1) / RPI, B2 = (N / R2 + RPI - 1) / RPI, B3 = (N / R3 + RPI - 1) / RPI;
  constexpr bool G2 = (N / R2) % RPI != 0 || (N / R3) % RPI != 0;
  static_assert((N / R1) % RPI == 0 && NT % TK == 0, "tile shape");
  const int c = t % TK, r0 = t / TK;
  constexpr int PF = INV ? B3 * R3 : B1 * R1;  // values of the next tile held in registers while this one is in LDS
  C xp[PF];
  auto tile_base = [&](int tile, bool& live) -> C* {
    const int plane = tile / tiles_x, kx = (tile - plane * tiles_x) * TK + c;
    live = kx < kxn;
    return data + (long long)plane * pstride + kx;
  };
  auto prefetch = [&](int tile) {
    bool live;
    const C* b = tile_base(tile, live);
    if constexpr (!INV) {
#pragma unroll
      for (int i = 0; i < B1; ++i)
#pragma unroll
        for (int q = 0; q < R1; ++q) xp[i * R1 + q] = live ? b[(long long)(r0 + i * RPI + q * (N / R1)) * es] : zmk<C>(0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < B3; ++i) {
        const int g = r0 + i * RPI;
#pragma unroll
        for (int q = 0; q < R3; ++q) xp[i * R3 + q] = (live && (!G2 || g < N / R3)) ? b[(long long)(g * R3 + q) * es] : zmk<C>(0, 0);
      }
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  __syncthreads();  // twiddle table
  for (; tile < ntiles; tile += gridDim.x) {
    bool live;
    C* b = tile_base(tile, live);
    if constexpr (!INV) {
      // ---- pass 1: registers -> LDS
#pragma unroll
      for (int i = 0; i < B1; ++i) {
        const int j = r0 + i * RPI;
        C x[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) x[q] = xp[i * R1 + q];
        dft<R1, false>(x);
        twiddle<R1, false>(x, tw[j]);
#pragma unroll
        for (int q = 0; q < R1; ++q) at(j + q * (N / R1), c) = x[q];
      }
      __syncthreads();
      if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
      // ---- pass 2 (blocks of L2)
#pragma unroll
      for (int i = 0; i < B2; ++i) {
        const int bb = r0 + i * RPI, g1 = bb / (L2 / R2), j = bb % (L2 / R2);
        if (G2 && bb >= N / R2) break;
        C x[R2];
#pragma unroll
        for (int q = 0; q < R2; ++q) x[q] = at(g1 * L2 + j + q * (L2 / R2), c);
        dft<R2, false>(x);
        twiddle<R2, false>(x, tw[j * R1]);
#pragma unroll
        for (int q = 0; q < R2; ++q) at(g1 * L2 + j + q * (L2 / R2), c) = x[q];
      }
      __syncthreads();
      // ---- pass 3 (blocks of 8, unit twiddles): LDS -> registers -> global, in storage order
#pragma unroll
      for (int i = 0; i < B3; ++i) {
        const int g = r0 + i * RPI;
        if (G2 && g >= N / R3) break;
        C x[R3];
#pragma unroll
        for (int q = 0; q < R3; ++q) x[q] = at(g * R3 + q, c);
        dft<R3, false>(x);
        if (live)
#pragma unroll
          for (int q = 0; q < R3; ++q) b[(long long)(g * R3 + q) * es] = x[q];
      }
    } else {
      // ---- inverse pass 1: registers -> LDS
#pragma unroll
      for (int i = 0; i < B3; ++i) {
        const int g = r0 + i * RPI;
        if (G2 && g >= N / R3) break;
        C x[R3];
#pragma unroll
        for (int q = 0; q < R3; ++q) x[q] = xp[i * R3 + q];
        dft<R3, true>(x);
#pragma unroll
        for (int q = 0; q < R3; ++q) at(g * R3 + q, c) = x[q];
      }
      __syncthreads();
      if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
      // ---- inverse pass 2
#pragma unroll
      for (int i = 0; i < B2; ++i) {
        const int bb = r0 + i * RPI, g1 = bb / (L2 / R2), j = bb % (L2 / R2);
        if (G2 && bb >= N / R2) break;
        C x[R2];
#pragma unroll
        for (int q = 0; q < R2; ++q) x[q] = at(g1 * L2 + j + q * (L2 / R2), c);
        twiddle<R2, true>(x, tw[j * R1]);
        dft<R2, true>(x);
#pragma unroll
        for (int q = 0; q < R2; ++q) at(g1 * L2 + j + q * (L2 / R2), c) = x[q];
      }
      __syncthreads();
      // ---- inverse pass 3: LDS -> registers -> global, natural order
#pragma unroll
      for (int i = 0; i < B1; ++i) {
        const int j = r0 + i * RPI;
        C x[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) x[q] = at(j + q * (N / R1), c);
        twiddle<R1, true>(x, tw[j]);
        dft<R1, true>(x);
        if (live)
#pragma unroll
          for (int q = 0; q < R1; ++q) b[(long long)(j + q * (N / R1)) * es] = x[q];
      }
    }
    __syncthreads();  // the LDS tile is rewritten by the next tile's first pass
  }
}

template <int LOGN, int TK, int NT, typename C>
int launch_line3(C* data, int kxn, int kxs, int nplanes, const C* tw, bool inverse, hipStream_t s, long long es = 0, long long pstride = 0) {
  constexpr int N = LOGN >= 64 ? 5 << (LOGN & 31) : (LOGN >= 32 ? 3 << (LOGN & 31) : 1 << LOGN);
  if (es == 0) {  // y direction: lines of one plane
    es = kxs;
    pstride = (long long)N * kxs;
  }
  constexpr size_t lds = ((size_t)N * TK + N) * sizeof(C);
  const int tiles_x = (kxn + TK - 1) / TK;
  const long long ntiles = (long long)tiles_x * nplanes;
  static bool attr_set[2] = {false, false};
  if (lds > 64 * 1024 && !attr_set[inverse]) {
    if (inverse)
      INS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_line3<LOGN, TK, NT, true, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    else
      INS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_line3<LOGN, TK, NT, false, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[inverse] = true;
  }
  // one tile per workgroup (measured equal to or better than persistent workgroups at 256^3 and 512^3); INS_LINE3_WGS = workgroups per CU makes them persistent
  const long long per_cu = ins_opt(OPT_INS_LINE3_WGS) > 0 ? ins_opt(OPT_INS_LINE3_WGS) : (1LL << 20);
  const unsigned nb = (unsigned)std::min<long long>(ntiles, 256LL * per_cu);
  if (inverse)
    hipLaunchKernelGGL((k_line3<LOGN, TK, NT, true, C>), dim3(nb), dim3(NT), lds, s, data, es, pstride, kxn, tiles_x, (int)ntiles, tw);
  else
    hipLaunchKernelGGL((k_line3<LOGN, TK, NT, false, C>), dim3(nb), dim3(NT), lds, s, data, es, pstride, kxn, tiles_x, (int)ntiles, tw);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

template <int LOGN, int TK, int NT, typename C = double2>
int launch_zsolve3(C* data, long long nl, const double* ax, int kxn, const double* ay, const double* az, const C* tw,
                   double inv_n, bool zero_mean, hipStream_t s, int kxs) {
  const int ntiles = (int)((nl + TK - 1) / TK);
  constexpr int N = LOGN >= 64 ? 5 << (LOGN & 31) : (LOGN >= 32 ? 3 << (LOGN & 31) : 1 << LOGN);
  constexpr size_t lds = ((size_t)N * TK + N) * sizeof(C);
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    INS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_zsolve3<LOGN, TK, NT, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  // 16-line tiles (136 KB of LDS: one workgroup per CU) run persistent, one workgroup per CU walking its share of the tiles with the next
  // tile's loads in flight (512^3: 0.587 -> 0.538 ms); 8-line tiles fit several workgroups per CU and run one tile per workgroup (persistent
  // they are slower: 0.599 vs 0.573 ms).  INS_ZSOLVE_WGS = workgroups per CU overrides.
  const int fit = std::max(1, std::min((int)(160 * 1024 / lds), 2048 / NT));
  const long long per_cu = ins_opt(OPT_INS_ZSOLVE_WGS) > 0 ? ins_opt(OPT_INS_ZSOLVE_WGS) : (TK >= 16 ? fit : (1LL << 20));
  const unsigned nb = (unsigned)std::min<long long>(ntiles, 256LL * per_cu);
  hipLaunchKernelGGL((k_zsolve3<LOGN, TK, NT, C>), dim3(nb), dim3(NT), lds, s, data, nl, ax, kxn, ay, az, tw, inv_n, zero_mean ? 1 : 0, kxs,
                     ins_opt(OPT_INS_ZSOLVE_SKEL) ? 1 : 0, ntiles);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

template <int LOGN, int TK, int NT = 256>
int launch_zsolve(double2* data, long long nl, const double* ax, int kxn, const double* ay, const double* az, const double2* tw,
                  double inv_n, bool zero_mean, hipStream_t s, int kxs) {
  const int ntiles = (int)((nl + TK - 1) / TK);
  constexpr size_t lds = ((size_t)(1 << LOGN) * TK + (1 << LOGN)) * sizeof(double2);
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    INS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_zsolve<LOGN, TK, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  // persistent workgroups: as many as are resident at once (LDS per workgroup decides), each walks its share of the tiles.
  // INS_ZSOLVE_WGS: workgroups per CU (0 = what fits; a large value = one tile per workgroup, the non-persistent form)
  const int fit = std::max(1, std::min((int)(160 * 1024 / lds), 2048 / NT));
  const long long per_cu = ins_opt(OPT_INS_ZSOLVE_WGS) > 0 ? ins_opt(OPT_INS_ZSOLVE_WGS) : fit;
  const unsigned nb = (unsigned)std::min<long long>(ntiles, 256LL * per_cu);
  hipLaunchKernelGGL((k_zsolve<LOGN, TK, NT>), dim3(nb), dim3(NT), lds, s, data, nl, ax, kxn, ay, az, tw, inv_n, zero_mean ? 1 : 0, kxs,
                     ins_opt(OPT_INS_ZSOLVE_SKEL) ? 1 : 0, ntiles);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

}  // namespace

bool ins_zsolve_supported(int nz) {
  if (ins_opt(OPT_INS_DISABLE_ZSOLVE)) return false;
  // 3 x 8 x 8, 6 x 8 x 8, 5 x 8 x 8 and 10 x 8 x 8 in the three-pass kernel
  if ((nz == 96 || nz == 192 || nz == 384 || nz == 160 || nz == 320 || nz == 640) && !ins_opt(OPT_INS_OWNFFT_POW2_ONLY)) return true;
  return nz >= 16 && nz <= 1024 && (nz & (nz - 1)) == 0;
}

// The y passes of the single-GPU spectral solve on the register passes (k_line3): lengths the three-pass kernel has, plus 128 = 2 x 8 x 8.
bool ins_line3_supported(int n) {
  if (ins_opt(OPT_INS_DISABLE_LINE3)) return false;
  if (n == 128 || n == 256 || n == 512) return true;
  return (n == 96 || n == 192 || n == 384 || n == 160 || n == 320 || n == 640) && !ins_opt(OPT_INS_OWNFFT_POW2_ONLY);
}
// out[p] = ay[k(p)]: the frequency held at storage position p after the forward pass of k_line3
void ins_line3_permute_symbol(int n, const double* ay, double* out) {
  const int R3 = 8, R2 = n % 64 == 0 ? 8 : 4, R1 = n / (R2 * R3);
  for (int p = 0; p < n; ++p) {
    const int q1 = p / (R2 * R3), q2 = (p / R3) % R2, q3 = p % R3;
    out[p] = ay[q1 + R1 * q2 + R1 * R2 * q3];
  }
}
template <typename C>
static int line3_y(C* d, int kxn, int kxs, int n1, int n2, const C* w, bool inverse, hipStream_t s, long long es = 0, long long ps = 0) {
  const int tk = (int)ins_opt(OPT_INS_LINE3_TK);
  switch (n1) {
    case 96: return launch_line3<32 + 5, 16, 256, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
    case 160: return launch_line3<64 + 5, 16, 256, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
    case 128: return launch_line3<7, 16, 256, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
    case 192: return launch_line3<32 + 6, 8, 256, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
    case 256:
      // 256^3, one solve: LDS kernel 352-364 us; 8 lines 339, 16 lines 338 (256 work-items) / 341 (512), 32 lines 377 (profiles/r03_line3_lab.txt)
      if (tk == 8) return launch_line3<8, 8, 256, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
      return launch_line3<8, 16, 256, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
    case 320: return launch_line3<64 + 6, 8, 256, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
    case 384: return launch_line3<32 + 7, 8, 256, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
    case 512:
      // 512^3, one solve: LDS kernel 3184 us; 8 lines x 512 work-items, one tile per workgroup 3100; 256 work-items 3128; persistent 3194; 16 lines 3168
      if (tk == 16) return launch_line3<9, 16, 512, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
      return launch_line3<9, 8, 512, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
    case 640: return launch_line3<64 + 7, 8, 512, C>(d, kxn, kxs, n2, w, inverse, s, es, ps);
  }
  ins_set_error("k_line3: unsupported length %d", n1);
  return INS_ERR_UNSUPPORTED;
}
// storage position of frequency k after the forward pass (the inverse of the map in ins_line3_permute_symbol)
int ins_line3_pos_of_freq(int n, int k) {
  const int R3 = 8, R2 = n % 64 == 0 ? 8 : 4, R1 = n / (R2 * R3);
  const int q1 = k % R1, q2 = (k / R1) % R2, q3 = k / (R1 * R2);
  return q1 * R2 * R3 + q2 * R3 + q3;
}
// forward transform along z of data[kz][ky][kx] (rows of kxs complex values, n1 rows per plane): the "planes" of the kernel are the ky rows, elements of a
// line are kxs * n1 apart (observespectrum on the library's own passes, ins_spectrum.hip)
int ins_k_line3_z(double* phat, int kxn, int n1, int n2, const double* tw, hipStream_t s, int kxs);
int ins_k_line3_y(double* phat, int kxn, int n1, int n2, const double* tw, bool inverse, hipStream_t s, int kxs) {
  return line3_y<double2>(reinterpret_cast<double2*>(phat), kxn, kxs, n1, n2, reinterpret_cast<const double2*>(tw), inverse, s);
}
int ins_k_line3_z(double* phat, int kxn, int n1, int n2, const double* tw, hipStream_t s, int kxs) {
  // length n2 along z; n1 "planes" (the ky rows) kxs apart; elements of a line kxs * n1 apart
  return line3_y<double2>(reinterpret_cast<double2*>(phat), kxn, kxs, n2, n1, reinterpret_cast<const double2*>(tw), false, s, (long long)kxs * n1, (long long)kxs);
}
int ins_k_line3_y_f32(float* phat, int kxn, int n1, int n2, const float* tw, bool inverse, hipStream_t s, int kxs) {
  return line3_y<float2>(reinterpret_cast<float2*>(phat), kxn, kxs, n1, n2, reinterpret_cast<const float2*>(tw), inverse, s);
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

// float2 spectra (`_f32` family): the three-pass kernel with twice the lines per tile (same bytes per row segment); symbols stay double
bool ins_zsolve_f32_supported(int nz) { return !ins_opt(OPT_INS_DISABLE_ZSOLVE) && (nz == 192 || nz == 256 || nz == 384 || nz == 512); }
int ins_zsolve_twiddles_f32(int nz, float** out) {
  std::vector<float> h(2 * (size_t)nz);
  for (int m = 0; m < nz; ++m) {
    const double a = -2.0 * M_PI * (double)m / (double)nz;
    h[2 * m] = (float)std::cos(a);
    h[2 * m + 1] = (float)std::sin(a);
    if ((4 * m) % nz == 0) {
      const int q = (4 * m) / nz;
      const float c[4] = {1, 0, -1, 0}, sn[4] = {0, -1, 0, 1};
      h[2 * m] = c[q];
      h[2 * m + 1] = sn[q];
    }
  }
  float* d = nullptr;
  INS_HIP_TRY(hipMalloc(&d, h.size() * sizeof(float)));
  if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    ins_set_error("twiddle upload failed");
    return INS_ERR_HIP;
  }
  *out = d;
  return INS_OK;
}
int ins_k_zsolve_f32(float* data, int nz, long long nl, const double* ax, int kxn, const double* ay, const double* az, const float* tw, double inv_n,
                     bool zero_mean, hipStream_t s, int kxs) {
  if (kxs <= 0) kxs = kxn;
  float2* d = reinterpret_cast<float2*>(data);
  const float2* w = reinterpret_cast<const float2*>(tw);
  switch (nz) {
    case 192: return launch_zsolve3<32 + 6, 16, 256, float2>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 256: return launch_zsolve3<8, 16, 256, float2>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 384: return launch_zsolve3<32 + 7, 16, 256, float2>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 512: {
      const int tk = (int)ins_opt(OPT_INS_ZSOLVE_TK), nt = (int)ins_opt(OPT_INS_ZSOLVE_NT);
      // 16 lines (70 KB of LDS: two workgroups per CU) against 32 (135 KB, one): 512^3 fp32 step 15.7 -> 14.65 ms, the pass 0.63 -> 0.37 ms
      if (tk == 32) return launch_zsolve3<9, 32, 512, float2>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
      if (tk == 16 && nt == 256) return launch_zsolve3<9, 16, 256, float2>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
      return launch_zsolve3<9, 16, 512, float2>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    }
  }
  ins_set_error("ins_k_zsolve_f32: unsupported nz = %d", nz);
  return INS_ERR_UNSUPPORTED;
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
    case 192: return launch_zsolve3<32 + 6, 8, 256>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 384: return launch_zsolve3<32 + 7, 8, 256>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 96: return launch_zsolve3<32 + 5, 16, 256>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 160: return launch_zsolve3<64 + 5, 16, 256>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 320: return launch_zsolve3<64 + 6, 8, 256>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 640: return launch_zsolve3<64 + 7, 8, 512>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);  // 92 KB tile
    case 256:
      if (!ins_opt(OPT_INS_ZSOLVE_RADIX4)) {
        return launch_zsolve3<8, 8, 256>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
      }
      return launch_zsolve<8, 8>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
    case 512: {
      const int tk = (int)ins_opt(OPT_INS_ZSOLVE_TK), nt = (int)ins_opt(OPT_INS_ZSOLVE_NT);
      if (!ins_opt(OPT_INS_ZSOLVE_RADIX4)) {
        if (tk == 16 && nt == 1024) return launch_zsolve3<9, 16, 1024>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
        if (tk == 8 && nt == 512) return launch_zsolve3<9, 8, 512>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
        if (tk == 8) return launch_zsolve3<9, 8, 256>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
        return launch_zsolve3<9, 16, 512>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);  // 512 B per plane and tile: 4.1 TB/s (8 lines: 3.8)
      }
      if (tk == 16 && nt == 1024) return launch_zsolve<9, 16, 1024>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
      if (tk == 16 && nt == 512) return launch_zsolve<9, 16, 512>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
      if (tk == 16) return launch_zsolve<9, 16>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
      if (tk == 4) return launch_zsolve<9, 4>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
      if (nt == 512) return launch_zsolve<9, 8, 512>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
      if (nt == 1024) return launch_zsolve<9, 8, 1024>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);
      return launch_zsolve<9, 8>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);    // 72 KB tile
    }
    case 1024: return launch_zsolve<10, 4>(d, nl, ax, kxn, ay, az, w, inv_n, zero_mean, s, kxs);  // 80 KB tile
  }
  ins_set_error("ins_k_zsolve: unsupported nz = %d", nz);
  return INS_ERR_UNSUPPORTED;
}

// Experiment hook (not part of the public ABI, like ins_tune_*): time `reps` launches of the fused z pass on `data` = nbox independent
// [nz][nl] blocks (nl = nky * kxs lines, kxn live columns), average ms per sweep over all boxes.  tools/zpass_lab.py varies the plane
// stride (nl * 16 B) at a fixed byte count with it.
extern "C" int ins_dbg_zsolve(double* data, int nz, long long nl, int kxn, int kxs, int nbox, int reps, float* ms) {
  double *ax = nullptr, *ay = nullptr, *az = nullptr, *tw = nullptr;
  const int nky = (int)(nl / kxs);
  std::vector<double> h(std::max(std::max(kxs, nky), nz), 1.0);
  INS_HIP_TRY(hipMalloc(&ax, kxs * 8));
  INS_HIP_TRY(hipMalloc(&ay, nky * 8));
  INS_HIP_TRY(hipMalloc(&az, nz * 8));
  INS_HIP_TRY(hipMemcpy(ax, h.data(), kxs * 8, hipMemcpyHostToDevice));
  INS_HIP_TRY(hipMemcpy(ay, h.data(), nky * 8, hipMemcpyHostToDevice));
  INS_HIP_TRY(hipMemcpy(az, h.data(), nz * 8, hipMemcpyHostToDevice));
  int rc = ins_zsolve_twiddles(nz, &tw);
  if (rc) return rc;
  hipEvent_t e0, e1;
  INS_HIP_TRY(hipEventCreate(&e0));
  INS_HIP_TRY(hipEventCreate(&e1));
  for (int r = -1; r < reps; ++r) {
    if (r == 0) INS_HIP_TRY(hipEventRecord(e0, nullptr));
    for (int b = 0; b < nbox; ++b)
      if ((rc = ins_k_zsolve(data + 2LL * b * nz * nl, nz, nl, ax, kxn, ay, az, tw, 1.0 / nz, false, nullptr, kxs))) return rc;
  }
  INS_HIP_TRY(hipEventRecord(e1, nullptr));
  INS_HIP_TRY(hipEventSynchronize(e1));
  INS_HIP_TRY(hipEventElapsedTime(ms, e0, e1));
  *ms /= (float)std::max(reps, 1);
  (void)hipFree(ax);
  (void)hipFree(ay);
  (void)hipFree(az);
  (void)hipFree(tw);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return INS_OK;
}
