// Own FFT passes of the spectral Poisson solve for power-of-two boxes (pressure.jl:316-347 without rocFFT).
//
// Five passes over HBM, every transform done in LDS with the same in-place radix-4/2 engine as ins_zsolve.hip:
//   1. k_xfwd : real -> half-complex along x (contiguous).  Two neighbouring y-rows are packed as one complex row
//               (re = row j, im = row j+1), transformed with ONE complex FFT and separated by Hermitian symmetry.
//               SRC = 1 computes the right-hand side Ω·div(u*) on the fly (divergence! + scalewithvolume! +
//               copyto!(pI, ...): operators.jl:117-125, 81-95, pressure.jl:320) — K2 is folded into this pass.
//   2. k_yfft<FWD>: along y for every (z-plane, kx) line; output stays in digit-reversed ky order (the symbol
//               vector ây is permuted to match, nothing is ever reordered).
//   3. k_zsolve (ins_zsolve.hip): z-FFT · symbol · inverse z-FFT.
//   4. k_yfft<INV>: decimation-in-time from the digit-reversed order back to natural ky.
//   5. k_xinv : half-complex -> real along x for row pairs, writes the unpadded pI.
// Normalisation 1/(nx ny nz) is folded into the symbol (pass 3).  rocFFT is not involved at all, which also keeps
// this path clear of the ROCm 7.2 real-plan cache bug (ins_fftcheck.hip).
#include <cmath>

#include "ins_internal.h"

namespace {

// The passes are templates over the complex element type C (double2, or float2 for the `_f32` family: half the bytes per pass).
template <typename C>
struct real_of;
template <>
struct real_of<double2> {
  using t = double;
};
template <>
struct real_of<float2> {
  using t = float;
};
template <typename C>
using real_t = typename real_of<C>::t;
template <typename C>
__device__ __forceinline__ C mkc(real_t<C> x, real_t<C> y) {
  C c;
  c.x = x;
  c.y = y;
  return c;
}
template <typename C>
__device__ __forceinline__ C cmul(C a, C b) { return mkc<C>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename C>
__device__ __forceinline__ C cmulc(C a, C b) { return mkc<C>(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
template <typename C>
__device__ __forceinline__ C cadd(C a, C b) { return mkc<C>(a.x + b.x, a.y + b.y); }
template <typename C>
__device__ __forceinline__ C csub(C a, C b) { return mkc<C>(a.x - b.x, a.y - b.y); }
template <typename C>
__device__ __forceinline__ C mul_mi(C a) { return mkc<C>(a.y, -a.x); }
template <typename C>
__device__ __forceinline__ C mul_pi(C a) { return mkc<C>(-a.y, a.x); }

// Transform lengths.  The template parameter LOGN of everything below is a SIZE CODE: codes < 32 are log2 N of a power of two; code
// 32 + m stands for N = 3 * 2^m (a radix-3 stage in front of the power-of-two stages: 192 = 3 * 64, 384 = 3 * 128), so that boxes with
// such sides run on these passes too instead of rocFFT (pressure.jl:316 plans any even n).
// (round 3: code 64 + m stands for N = 5 * 2^m — 320 = 5 * 64, 640 = 5 * 128 — with a radix-5 stage in front.)
constexpr int fft_r3(int code) { return code >= 64 ? 5 : (code >= 32 ? 3 : 1); }  // the odd leading radix (1: none)
constexpr int fft_lg(int code) { return code & 31; }                           // log2 of the power-of-two part
constexpr int fft_len(int code) { return fft_r3(code) << fft_lg(code); }

// Storage position of frequency k after the DIF stages (radix 3 first for the 3 * 2^m lengths, then radix 2 when the remaining log2 is odd,
// then radix 4): inverse of freq_of_pos() in ins_zsolve.hip.
template <int LOGN>
__host__ __device__ __forceinline__ int pos_of_freq(int k) {
  constexpr int LG = fft_lg(LOGN), R3 = fft_r3(LOGN);
  constexpr bool ODD = LG & 1;
  int p = 0, L = 1 << LG;
  if (R3 != 1) {
    p = (k % R3) * L;
    k /= R3;
  }
  if (ODD) {
    p += (k & 1) * (L / 2);
    k >>= 1;
    L /= 2;
  }
#pragma unroll
  for (int s = 0; s < LG / 2; ++s) {
    p += (k & 3) * (L / 4);
    k >>= 2;
    L /= 4;
  }
  return p;
}

// x layout (RFAST, SR == 1): in the radix-4 stages with Q = 4 and Q = 1 the 16 lanes of a quarter wave read 16-B elements 64 B or 256 B apart, and the twiddle
// reads of the stages with Q <= 16 are 64 B .. 256 B apart as well: four-way bank conflicts — 61-63 % of the LDS cycles of k_xfwd / k_xinv at 256^3
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, tools/run_lds_pmc.sh).  So a row (and the twiddle table) is stored with every element ROTATED inside its group
// of 16 by an amount that depends on the group: bits 4-5 of the index rotate by one element, bits 6-7 by four.  Unit-stride accesses stay conflict-free
// (a rotation permutes a group); the Q = 1 stage becomes conflict-free as it is; the Q = 4 stage needs its butterflies dealt to the lanes so that a quarter
// wave spans bits 6-7 instead of bits 4-5 (fft_swz_g).  Lengths whose power-of-two part is a multiple of 256 (256, 512, 1024).
template <int LOGN>
constexpr bool fft_swz_on() { return fft_r3(LOGN) == 1 && fft_len(LOGN) % 256 == 0; }
template <bool ON>
__host__ __device__ __forceinline__ int fft_swz(int i) {
  return ON ? ((i & ~15) | ((i + ((i >> 4) & 3) + 4 * ((i >> 6) & 3)) & 15)) : i;
}
__device__ __forceinline__ int fft_swz_g(int gg) { return (gg & ~15) | ((gg & 3) << 2) | ((gg >> 2) & 3); }  // 16-element block index: bit pairs 0-1 and 2-3 swapped

// In-place transforms of NC independent lines held in LDS; element (r, c) lives at buf[r*SR + c*SC].
// RFAST: consecutive work-items take consecutive butterflies of one line (x layout, SR == 1) instead of the
// same butterfly of consecutive lines (y/z layout, SC == 1) — keeps LDS accesses unit-stride in both layouts.
// radix-3 butterfly, forward (w = e^{-2πi/3}) / inverse (conjugate)
template <bool INV, typename C>
__device__ __forceinline__ void bfly3(C& x0, C& x1, C& x2) {
  using T = real_t<C>;
  constexpr T h = (T)0.86602540378443864676;  // sqrt(3)/2
  const C t1 = cadd(x1, x2), d = csub(x1, x2);
  const C t2 = mkc<C>(x0.x - (T)0.5 * t1.x, x0.y - (T)0.5 * t1.y);
  const C t3 = INV ? mkc<C>(-h * d.y, h * d.x) : mkc<C>(h * d.y, -h * d.x);  // (+-i) sqrt(3)/2 (x1 - x2)
  x0 = cadd(x0, t1);
  x1 = cadd(t2, t3);
  x2 = csub(t2, t3);
}

// radix-5 butterfly, forward (w = e^{-2πi/5}) / inverse (conjugate)
template <bool INV, typename C>
__device__ __forceinline__ void bfly5(C& x0, C& x1, C& x2, C& x3, C& x4) {
  using T = real_t<C>;
  constexpr T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;  // cos(2π/5), cos(4π/5)
  constexpr T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;   // sin(2π/5), sin(4π/5)
  const C a = cadd(x1, x4), b = cadd(x2, x3), d = csub(x1, x4), e = csub(x2, x3);
  const C m1 = mkc<C>(x0.x + c1 * a.x + c2 * b.x, x0.y + c1 * a.y + c2 * b.y);
  const C m2 = mkc<C>(x0.x + c2 * a.x + c1 * b.x, x0.y + c2 * a.y + c1 * b.y);
  // forward: X1 = m1 - i (s1 d + s2 e), X4 = m1 + i (...), X2 = m2 - i (s2 d - s1 e), X3 = m2 + i (...)
  const C u = mkc<C>(s1 * d.x + s2 * e.x, s1 * d.y + s2 * e.y), v = mkc<C>(s2 * d.x - s1 * e.x, s2 * d.y - s1 * e.y);
  const C iu = INV ? mkc<C>(-u.y, u.x) : mkc<C>(u.y, -u.x), iv = INV ? mkc<C>(-v.y, v.x) : mkc<C>(v.y, -v.x);  // (-+ i) u, (-+ i) v
  x0 = mkc<C>(x0.x + a.x + b.x, x0.y + a.y + b.y);
  x1 = cadd(m1, iu);
  x4 = csub(m1, iu);
  x2 = cadd(m2, iv);
  x3 = csub(m2, iv);
}

template <int LOGN, int NC, int SR, int SC, bool RFAST, int NT = 256, typename C>
__device__ __forceinline__ void fft_dif(C* __restrict__ buf, const C* __restrict__ tw, int t) {
  constexpr int N = fft_len(LOGN), R3 = fft_r3(LOGN), M = N / R3;  // M: the power-of-two sub-length
  constexpr bool ODD = fft_lg(LOGN) & 1;
  if (R3 == 3) {  // N = 3 M: one radix-3 stage over the whole line, then the three sub-blocks of length M run the stages below side by side
    for (int w = t; w < M * NC; w += NT) {
      const int c = RFAST ? w / M : w % NC, j = RFAST ? w % M : w / NC;
      C* x = buf + c * SC;
      C a0 = x[j * SR], a1 = x[(j + M) * SR], a2 = x[(j + 2 * M) * SR];
      bfly3<false>(a0, a1, a2);
      x[j * SR] = a0;
      x[(j + M) * SR] = cmul(a1, tw[j]);
      x[(j + 2 * M) * SR] = cmul(a2, tw[2 * j]);
    }
    __syncthreads();
  }
  if (R3 == 5) {  // N = 5 M: one radix-5 stage over the whole line, twiddles W_N^(q j)
    for (int w = t; w < M * NC; w += NT) {
      const int c = RFAST ? w / M : w % NC, j = RFAST ? w % M : w / NC;
      C* x = buf + c * SC;
      C a0 = x[j * SR], a1 = x[(j + M) * SR], a2 = x[(j + 2 * M) * SR], a3 = x[(j + 3 * M) * SR], a4 = x[(j + 4 * M) * SR];
      bfly5<false>(a0, a1, a2, a3, a4);
      x[j * SR] = a0;
      x[(j + M) * SR] = cmul(a1, tw[j]);
      x[(j + 2 * M) * SR] = cmul(a2, tw[2 * j]);
      x[(j + 3 * M) * SR] = cmul(a3, tw[3 * j]);
      x[(j + 4 * M) * SR] = cmul(a4, tw[4 * j]);
    }
    __syncthreads();
  }
  constexpr bool SWZ = RFAST && fft_swz_on<LOGN>();  // rows and twiddle table stored rotated (see fft_swz); SR == 1 then
  int L = M;
  if (ODD) {
    for (int w = t; w < (N / 2) * NC; w += NT) {
      const int c = RFAST ? w / (N / 2) : w % NC, jj = RFAST ? w % (N / 2) : w / NC;
      const int sub = jj / (M / 2), j = jj - sub * (M / 2);
      C* x = buf + c * SC + sub * M * SR;
      const int i0 = fft_swz<SWZ>(j) * SR, i1 = fft_swz<SWZ>(j + M / 2) * SR;
      const C a0 = x[i0], a1 = x[i1];
      x[i0] = cadd(a0, a1);
      x[i1] = cmul(csub(a0, a1), tw[fft_swz<SWZ>(j * R3)]);  // W_M^j = W_N^(R3 j)
    }
    L = M / 2;
    __syncthreads();
  }
#pragma unroll 1
  for (; L >= 4; L >>= 2) {
    const int Q = L / 4, step = N / L;
    for (int w = t; w < (N / 4) * NC; w += NT) {
      const int c = RFAST ? w / (N / 4) : w % NC, bb = RFAST ? w % (N / 4) : w / NC;
      const int sub = bb / (M / 4), b = bb - sub * (M / 4);
      int g = b / Q;
      const int j = b - g * Q;
      if (SWZ && Q == 4) g = fft_swz_g(g);
      C* x = buf + c * SC + sub * M * SR;
      const int e = g * L + j;
      const int i0 = fft_swz<SWZ>(e) * SR, i1 = fft_swz<SWZ>(e + Q) * SR, i2 = fft_swz<SWZ>(e + 2 * Q) * SR, i3 = fft_swz<SWZ>(e + 3 * Q) * SR;
      const C a0 = x[i0], a1 = x[i1], a2 = x[i2], a3 = x[i3];
      const C t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi(csub(a1, a3));
      C y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
      if (L > 4) {
        const C w1 = tw[fft_swz<SWZ>(j * step)], w2 = cmul(w1, w1), w3 = cmul(w2, w1);  // one table read instead of three (LDS-bound stages)
        y1 = cmul(y1, w1);
        y2 = cmul(y2, w2);
        y3 = cmul(y3, w3);
      }
      x[i0] = cadd(t0, t2);
      x[i1] = y1;
      x[i2] = y2;
      x[i3] = y3;
    }
    __syncthreads();
  }
}

template <int LOGN, int NC, int SR, int SC, bool RFAST, int NT = 256, typename C>
__device__ __forceinline__ void fft_dit(C* __restrict__ buf, const C* __restrict__ tw, int t) {
  constexpr int N = fft_len(LOGN), R3 = fft_r3(LOGN), M = N / R3;
  constexpr bool ODD = fft_lg(LOGN) & 1;
  constexpr bool SWZ = RFAST && fft_swz_on<LOGN>();
#pragma unroll 1
  for (int L = 4; L <= (ODD ? M / 2 : M); L <<= 2) {
    const int Q = L / 4, step = N / L;
    for (int w = t; w < (N / 4) * NC; w += NT) {
      const int c = RFAST ? w / (N / 4) : w % NC, bb = RFAST ? w % (N / 4) : w / NC;
      const int sub = bb / (M / 4), b = bb - sub * (M / 4);
      int g = b / Q;
      const int j = b - g * Q;
      if (SWZ && Q == 4) g = fft_swz_g(g);
      C* x = buf + c * SC + sub * M * SR;
      const int e = g * L + j;
      const int i0 = fft_swz<SWZ>(e) * SR, i1 = fft_swz<SWZ>(e + Q) * SR, i2 = fft_swz<SWZ>(e + 2 * Q) * SR, i3 = fft_swz<SWZ>(e + 3 * Q) * SR;
      C x0 = x[i0], x1 = x[i1], x2 = x[i2], x3 = x[i3];
      if (L > 4) {
        const C w1 = tw[fft_swz<SWZ>(j * step)], w2 = cmul(w1, w1), w3 = cmul(w2, w1);
        x1 = cmulc(x1, w1);
        x2 = cmulc(x2, w2);
        x3 = cmulc(x3, w3);
      }
      const C t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = mul_pi(csub(x1, x3));
      x[i0] = cadd(t0, t2);
      x[i1] = cadd(t1, t3);
      x[i2] = csub(t0, t2);
      x[i3] = csub(t1, t3);
    }
    __syncthreads();
  }
  if (ODD) {
    for (int w = t; w < (N / 2) * NC; w += NT) {
      const int c = RFAST ? w / (N / 2) : w % NC, jj = RFAST ? w % (N / 2) : w / NC;
      const int sub = jj / (M / 2), j = jj - sub * (M / 2);
      C* x = buf + c * SC + sub * M * SR;
      const int i0 = fft_swz<SWZ>(j) * SR, i1 = fft_swz<SWZ>(j + M / 2) * SR;
      const C x0 = x[i0], x1 = cmulc(x[i1], tw[fft_swz<SWZ>(j * R3)]);
      x[i0] = cadd(x0, x1);
      x[i1] = csub(x0, x1);
    }
    __syncthreads();
  }
  if (R3 == 3) {
    for (int w = t; w < M * NC; w += NT) {
      const int c = RFAST ? w / M : w % NC, j = RFAST ? w % M : w / NC;
      C* x = buf + c * SC;
      C a0 = x[j * SR], a1 = cmulc(x[(j + M) * SR], tw[j]), a2 = cmulc(x[(j + 2 * M) * SR], tw[2 * j]);
      bfly3<true>(a0, a1, a2);
      x[j * SR] = a0;
      x[(j + M) * SR] = a1;
      x[(j + 2 * M) * SR] = a2;
    }
    __syncthreads();
  }
  if (R3 == 5) {
    for (int w = t; w < M * NC; w += NT) {
      const int c = RFAST ? w / M : w % NC, j = RFAST ? w % M : w / NC;
      C* x = buf + c * SC;
      C a0 = x[j * SR], a1 = cmulc(x[(j + M) * SR], tw[j]), a2 = cmulc(x[(j + 2 * M) * SR], tw[2 * j]), a3 = cmulc(x[(j + 3 * M) * SR], tw[3 * j]),
        a4 = cmulc(x[(j + 4 * M) * SR], tw[4 * j]);
      bfly5<true>(a0, a1, a2, a3, a4);
      x[j * SR] = a0;
      x[(j + M) * SR] = a1;
      x[(j + 2 * M) * SR] = a2;
      x[(j + 3 * M) * SR] = a3;
      x[(j + 4 * M) * SR] = a4;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------
// Pass 2 / 4: FFT along y.  data[plane][ky][kx]; a workgroup owns TK consecutive kx of one plane.
// ------------------------------------------------------------------------------------------------------------
// PACKED (slab path): the other side of the pass is the transposed, kx-chunked exchange buffer
//   packed[chunk c][dest/src rank q][kz_local][ky_local][kx_local],  ky_pos = q*nyl + ky_local,  kx = c*cw + kx_local,
// so the forward pass writes what the all-to-all sends and the inverse pass reads what it received — no separate
// pack / unpack passes (ins_slab.hip k_transpose_pack is only used with rocFFT plans).
struct PackMap {
  double2* packed;
  int nyl, nzl, cw;
};

template <int LOGN, int TK, bool INVERSE, bool PACKED, typename C = double2>
__global__ __launch_bounds__(256) void k_yfft(C* __restrict__ data, int kxn, int kxs, const C* __restrict__ tw_g, PackMap pm) {
  constexpr int N = fft_len(LOGN);
  extern __shared__ __align__(16) unsigned char lds_raw[];
  C* lds_dyn = reinterpret_cast<C*>(lds_raw);
  C* buf = lds_dyn;          // [N][TK]
  C* tw = lds_dyn + N * TK;  // [N]
  const int t = threadIdx.x;
  const int col = t % TK;
  const int kx = blockIdx.x * TK + col;
  const bool live = kx < kxn;
  C* base = data + (long long)blockIdx.y * N * kxs + kx;  // kxs = row stride (>= kxn; a multiple of 8 keeps tiles line-aligned)
  // packed address of (row r, this kx, this plane)
  long long pbase = 0, prow = 0, pq = 0;
  if (PACKED) {
    const int c = kx / pm.cw, kxl = kx - c * pm.cw;
    const int kxc = min(pm.cw, kxn - c * pm.cw);
    const long long nranks = N / pm.nyl;
    pbase = nranks * pm.nzl * pm.nyl * (long long)(c * pm.cw) + kxl + (long long)kxc * pm.nyl * blockIdx.y;
    prow = kxc;                                   // stride of ky_local
    pq = (long long)kxc * pm.nyl * pm.nzl;        // stride of the rank index q
  }
  auto paddr = [&](int r) {
    const int q = r / pm.nyl, kyl = r - q * pm.nyl;
    return pbase + prow * kyl + pq * q;
  };
  for (int m = t; m < N; m += 256) tw[m] = tw_g[m];
  constexpr int RPT = 256 / TK;
  constexpr int NIT = N / RPT;  // rows per work-item: all its loads are issued before the first LDS write
  {
    C v[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int r = t / TK + q * RPT;
      v[q] = mkc<C>(0.0, 0.0);
      if (live) v[q] = (PACKED && INVERSE) ? static_cast<C*>((void*)pm.packed)[paddr(r)] : base[(long long)r * kxs];
    }
#pragma unroll
    for (int q = 0; q < NIT; ++q) buf[(t / TK + q * RPT) * TK + col] = v[q];
  }
  __syncthreads();
  if (INVERSE)
    fft_dit<LOGN, TK, TK, 1, false>(buf, tw, t);
  else
    fft_dif<LOGN, TK, TK, 1, false>(buf, tw, t);
  if (live)
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int r = t / TK + q * RPT;
      if (PACKED && !INVERSE)
        static_cast<C*>((void*)pm.packed)[paddr(r)] = buf[r * TK + col];
      else
        base[(long long)r * kxs] = buf[r * TK + col];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Pass 1: x forward, NP row pairs per workgroup.  SRC 0: rows come from pI; SRC 1: rows = Ω·div(u) (K2 fused, periodic
// wrap in all directions); SRC 2: the same on a z-slab (z neighbour from the ghost plane); SRC 3: the 2-D divergence; SRC 4: the divergence on a grid with walls (ghost volumes of u valid).
// ------------------------------------------------------------------------------------------------------------
template <int LOGN, int NP, int SRC, typename C = double2>
__global__ __launch_bounds__(256) void k_xfwd(GridDev g, const double* __restrict__ src, C* __restrict__ out, int n1,
                                              const C* __restrict__ tw_g, int kxs, int kz0, int skel) {
  using T = real_t<C>;  // (C = float2: `src` points to float data — SRC 0: the float pI; SRC 5: the float velocity field)
  constexpr int N = fft_len(LOGN);
  constexpr int KXN = N / 2 + 1;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  C* lds_dyn = reinterpret_cast<C*>(lds_raw);
  C* buf = lds_dyn;          // [NP][N]
  C* tw = lds_dyn + NP * N;  // [N]
  const int t = threadIdx.x;
  const int kz = kz0 + blockIdx.y;  // interior plane index (kz0: first plane of this launch)
  const int j0 = blockIdx.x * 2 * NP;
  constexpr bool SWZ = fft_swz_on<LOGN>();  // rows and twiddles stored rotated inside groups of 16 (bank conflicts of the late stages: fft_swz)
  for (int m = t; m < N; m += 256) tw[fft_swz<SWZ>(m)] = tw_g[m];
  T* bufd = reinterpret_cast<T*>(buf);
  // every work-item owns NIT (row, column) points; all their loads are issued (clamped rows, no branches) before the first
  // LDS write, so one round trip to HBM covers the whole tile instead of one per row
  constexpr int NIT = (2 * NP * N) / 256;
  {
    T v[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int idx = t + 256 * q;
      const int row = idx / N, i = idx - row * N;
      const int j = min(j0 + row, n1 - 1);
      if (SRC == 0) {
        v[q] = reinterpret_cast<const T*>(src)[i + (long long)N * (j + (long long)n1 * kz)];
      } else if (SRC == 6) {
        // one padded scalar array (a velocity component): its interior volumes, ghosts stripped on the fly (observespectrum, ins_spectrum.hip)
        v[q] = (T)src[(g.ip_lo[0] + i) + (g.ip_lo[1] + j) * g.sx[1] + (g.ip_lo[2] + kz) * g.sx[2]];
      } else if (SRC == 5) {
        // as SRC 1, from a FLOAT velocity field (the `_f32` family solves its pressure equation with these fp64 passes, ins_f32.hip):
        // differences and metrics in double from the float values
        const float* sf = reinterpret_cast<const float*>(src);
        const int I0 = i + 1, I1 = j + 1, I2 = kz + 1;
        const long long c = I0 + I1 * g.sx[1] + I2 * g.sx[2];
        const long long cx = I0 == 1 ? c + (long long)(g.N[0] - 3) : c - 1;
        const long long cy = I1 == 1 ? c + (long long)(g.N[1] - 3) * g.sx[1] : c - g.sx[1];
        const long long cz = I2 == 1 ? c + (long long)(g.N[2] - 3) * g.sx[2] : c - g.sx[2];
        double d = ((double)sf[c] - (double)sf[cx]) * g.rdx[0][I0];
        d += ((double)sf[g.sc + c] - (double)sf[g.sc + cy]) * g.rdx[1][I1];
        d += ((double)sf[2 * g.sc + c] - (double)sf[2 * g.sc + cz]) * g.rdx[2][I2];
        v[q] = (T)(d * (g.dx[0][I0] * g.dx[1][I1] * g.dx[2][I2]));
      } else if (SRC == 4) {
        // Ω · div(u) at the pressure point (i, j, kz) of a grid with walls: the ghost volumes of u are valid (k_div_to_pI<3, false>)
        const int I0 = g.ip_lo[0] + i, I1 = g.ip_lo[1] + j, I2 = g.ip_lo[2] + kz;
        const long long c = I0 + I1 * g.sx[1] + I2 * g.sx[2];
        double d = (src[c] - src[c - 1]) * g.rdx[0][I0];
        d += (src[g.sc + c] - src[g.sc + c - g.sx[1]]) * g.rdx[1][I1];
        d += (src[2 * g.sc + c] - src[2 * g.sc + c - g.sx[2]]) * g.rdx[2][I2];
        v[q] = (T)(d * (g.dx[0][I0] * g.dx[1][I1] * g.dx[2][I2]));
      } else if (SRC == 3) {
        // 2-D: Ω · div(u*) at interior cell (i, j) with periodic wrap (k_div_to_pI<2, true>)
        const int I0 = i + 1, I1 = j + 1;
        const long long c = I0 + I1 * g.sx[1];
        const long long cx = I0 == 1 ? c + (long long)(g.N[0] - 3) : c - 1;
        const long long cy = I1 == 1 ? c + (long long)(g.N[1] - 3) * g.sx[1] : c - g.sx[1];
        const double d = (src[c] - src[cx]) * g.rdx[0][I0] + (src[g.sc + c] - src[g.sc + cy]) * g.rdx[1][I1];
        v[q] = (T)(d * (g.dx[0][I0] * g.dx[1][I1]));
      } else {
        // Ω · div(u*) at interior cell (i, j, kz): periodic wrap instead of ghost reads (k_div_to_pI<3, true>)
        const int I0 = i + 1, I1 = j + 1, I2 = kz + 1;
        const long long c = I0 + I1 * g.sx[1] + I2 * g.sx[2];
        const long long cx = I0 == 1 ? c + (long long)(g.N[0] - 3) : c - 1;
        const long long cy = I1 == 1 ? c + (long long)(g.N[1] - 3) * g.sx[1] : c - g.sx[1];
        const long long cz = (SRC == 1 && I2 == 1) ? c + (long long)(g.N[2] - 3) * g.sx[2] : c - g.sx[2];  // SRC 2: slab ghost plane
        double d = 0.0;
        d += (src[c] - src[cx]) * g.rdx[0][I0];
        d += (src[g.sc + c] - src[g.sc + cy]) * g.rdx[1][I1];
        d += (src[2 * g.sc + c] - src[2 * g.sc + cz]) * g.rdx[2][I2];
        const double om = g.dx[0][I0] * g.dx[1][I1] * g.dx[2][I2];
        v[q] = (T)(d * om);
      }
    }
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int idx = t + 256 * q;
      const int row = idx / N, i = idx - row * N;
      bufd[2 * ((row >> 1) * N + fft_swz<SWZ>(i)) + (row & 1)] = (j0 + row < n1) ? v[q] : (T)0;
    }
  }
  __syncthreads();
  if (!skel) fft_dif<LOGN, NP, 1, N, true>(buf, tw, t);  // skel (INS_X_SKEL): timing experiment, loads / LDS scatter / stores only
  // separate the two real rows:  A[k] = (Z[k] + conj Z[N-k]) / 2,   B[k] = (Z[k] - conj Z[N-k]) / (2i)
  for (int idx = t; idx < NP * KXN; idx += 256) {
    const int p = idx / KXN, s = idx - p * KXN;
    const int j = j0 + 2 * p;
    if (j >= n1) continue;
    const C zk = buf[p * N + fft_swz<SWZ>(pos_of_freq<LOGN>(s))];
    const C zm = buf[p * N + fft_swz<SWZ>(pos_of_freq<LOGN>((N - s) % N))];
    const C a = mkc<C>((T)0.5 * (zk.x + zm.x), (T)0.5 * (zk.y - zm.y));
    const C b = mkc<C>((T)0.5 * (zk.y + zm.y), (T)-0.5 * (zk.x - zm.x));
    const long long o = s + (long long)kxs * (j + (long long)n1 * kz);
    out[o] = a;
    out[o + kxs] = b;  // row j + 1 (n1 is even)
  }
}

// ------------------------------------------------------------------------------------------------------------
// Pass 5: x inverse for NP row pairs: Z[k] = A[k] + i B[k], Z[N-k] = conj A[k] + i conj B[k]  ->  DIT  ->  pI rows.
// ------------------------------------------------------------------------------------------------------------
template <int LOGN, int NP, typename C = double2>
__global__ __launch_bounds__(256) void k_xinv(const C* __restrict__ in, real_t<C>* __restrict__ pI, int n1,
                                              const C* __restrict__ tw_g, int kxs, int skel) {
  using T = real_t<C>;
  constexpr int N = fft_len(LOGN);
  constexpr int KXN = N / 2 + 1;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  C* lds_dyn = reinterpret_cast<C*>(lds_raw);
  C* buf = lds_dyn;
  C* tw = lds_dyn + NP * N;
  const int t = threadIdx.x;
  const int kz = blockIdx.y;
  const int j0 = blockIdx.x * 2 * NP;
  constexpr bool SWZ = fft_swz_on<LOGN>();
  for (int m = t; m < N; m += 256) tw[fft_swz<SWZ>(m)] = tw_g[m];
  constexpr int NIT = (NP * KXN + 255) / 256;  // all loads first (clamped, branch-free), then the LDS scatter
  {
    C av[NIT], bv[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int idx = min(t + 256 * q, NP * KXN - 1);
      const int p = idx / KXN, s = idx - p * KXN;
      const int j = min(j0 + 2 * p, n1 - 2);
      const long long o = s + (long long)kxs * (j + (long long)n1 * kz);
      av[q] = in[o];
      bv[q] = in[o + kxs];
    }
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int idx = t + 256 * q;
      if (idx >= NP * KXN) break;
      const int p = idx / KXN, s = idx - p * KXN;
      C a = av[q], b = bv[q];
      if (j0 + 2 * p >= n1) a = b = mkc<C>(0, 0);
      if (s == 0 || s == N / 2) {  // C2R semantics: DC and Nyquist bins are real
        a.y = 0;
        b.y = 0;
      }
      buf[p * N + fft_swz<SWZ>(pos_of_freq<LOGN>(s))] = mkc<C>(a.x - b.y, a.y + b.x);
      if (s != 0 && s != N / 2) buf[p * N + fft_swz<SWZ>(pos_of_freq<LOGN>(N - s))] = mkc<C>(a.x + b.y, b.x - a.y);
    }
  }
  __syncthreads();
  if (!skel) fft_dit<LOGN, NP, 1, N, true>(buf, tw, t);
  const T* bufd = reinterpret_cast<const T*>(buf);
  for (int idx = t; idx < 2 * NP * N; idx += 256) {
    const int row = idx / N, i = idx - row * N;
    const int j = j0 + row;
    if (j < n1) pI[i + (long long)N * (j + (long long)n1 * kz)] = bufd[2 * ((row >> 1) * N + fft_swz<SWZ>(i)) + (row & 1)];
  }
}

// ------------------------------------------------------------------------------------------------------------
// psolver_direct with a periodic uniform z direction (ins_fdm.hip): the z eigenvectors are Fourier modes, so the two z GEMMs and the
// scaling pass become one pass.  The work array is REAL, r[z][y][x]; two adjacent x-columns (a = 2m, b = 2m+1) are taken as the real and
// imaginary part of one complex z-line: forward FFT in LDS, the two spectra separated through the k <-> N-k pairing inside the line
// (A = (Z_k + conj Z_{N-k})/2, B = (Z_k - conj Z_{N-k})/(2i)), each divided by its own h (λx + λy + λz(k)) — the columns differ in λx —,
// recombined, inverse FFT.  Singular systems (ins_fdm.hip): mean(f)·(Vᵀ1) is subtracted at k = 0, the null mode is dropped, and the
// k = 0 coefficients give the block partials of Σ (Vᵀ1) q' = n·mean(p).
// ------------------------------------------------------------------------------------------------------------
struct FdmZArgs {
  double2* data;          // [N][nl] complex view of the real array: nl = (n0/2) n1 lines
  long long nl;
  int n0h;                // n0 / 2
  const double *lx, *ly;  // generalised eigenvalues of the x and y factors
  const double* lz;       // λz(k), k = 0..N/2: -(4/h²) sin²(πk/N);  DCT: k = 0..N-1: -(4/h²) sin²(πk/2N)
  const double2* wq;      // DCT: e^{-iπk/2N}, k = 0..N/2
  const double *ox, *oy;  // Vxᵀ1, Vyᵀ1
  double h, tol;
  int singular;
  const double* meanf;    // device scalar mean(f)
  double* partial;        // [gridDim.x] block partials of Σ ox oy R'[k=0] / h
};

// DCT (uniform z between two walls: Neumann pressure on both sides): the z eigenvectors are cos(π(2n+1)k/2N), so the same pass serves with three
// changes (Makhoul's N-point form of the DCT-II): the line enters the FFT reordered (v[m] = x[2m], v[N-1-m] = x[2m+1]); the spectrum of a real
// line gives its cosine coefficients as C[k] = Re y, C[N-k] = -Im y with y = e^{-iπk/2N} V[k]; after the scaling the spectrum is rebuilt as
// V'[k] = e^{+iπk/2N} (C'[k] - i C'[N-k]) and the inverse FFT returns the reordered solution.  k = 0 is the plain sum as in the Fourier case,
// so the singular-system bookkeeping is unchanged.
template <int LOGN, int TK, bool DCT = false>
__global__ __launch_bounds__(256) void k_fdm_z(FdmZArgs a, const double2* __restrict__ tw_g) {
  constexpr int N = fft_len(LOGN);
  extern __shared__ double2 lds_dyn[];
  double2* buf = lds_dyn;          // [N][TK]
  double2* tw = lds_dyn + N * TK;  // [N]
  __shared__ double red[4];
  const int t = threadIdx.x;
  const int col = t % TK;
  const long long line = (long long)blockIdx.x * TK + col;
  const bool live = line < a.nl;
  for (int m = t; m < N; m += 256) tw[m] = tw_g[m];
  constexpr int RPT = 256 / TK;
  constexpr int NIT = N / RPT;
  {
    double2 v[NIT];
#pragma unroll
    for (int q = 0; q < NIT; ++q) v[q] = live ? a.data[(long long)(t / TK + q * RPT) * a.nl + line] : make_double2(0.0, 0.0);
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int z = t / TK + q * RPT;
      const int r = DCT ? ((z & 1) ? N - 1 - (z >> 1) : (z >> 1)) : z;  // plane z -> row of the transform
      buf[r * TK + col] = v[q];
    }
  }
  __syncthreads();
  fft_dif<LOGN, TK, TK, 1, false>(buf, tw, t);
  __syncthreads();
  const double mf = a.singular ? *a.meanf : 0.0;
  const double inv = 1.0 / (a.h * (double)N);
  double acc = 0.0;
  for (int idx = t; idx < (N / 2 + 1) * TK; idx += 256) {
    const int c = idx % TK, k = idx / TK;
    const long long ln = (long long)blockIdx.x * TK + c;
    if (ln >= a.nl) continue;
    const int m = (int)(ln % a.n0h), j = (int)(ln / a.n0h);
    const int ia = 2 * m, ib = ia + 1;
    const int pk = pos_of_freq<LOGN>(k), pm = pos_of_freq<LOGN>((N - k) % N);
    const double2 zk = buf[pk * TK + c], zm = buf[pm * TK + c];
    const bool self = k == 0 || k == N / 2;
    double2 A = self ? make_double2(zk.x, 0.0) : make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
    double2 B = self ? make_double2(zk.y, 0.0) : make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
    if constexpr (DCT) {
      // cosine coefficients of the two columns at k and N-k, each with its own eigenvalue
      const double2 w = a.wq[k];
      const double2 ya = cmul(A, w), yb = cmul(B, w);
      double ca[2] = {ya.x, -ya.y}, cb[2] = {yb.x, -yb.y};  // [0]: k, [1]: N - k
      const int kk[2] = {k, N - k};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q == 1 && self) {  // k = 0: C[N] does not exist; k = N/2: the same coefficient
          ca[1] = k == 0 ? 0.0 : ca[0];
          cb[1] = k == 0 ? 0.0 : cb[0];
          break;
        }
        const double lyz = a.ly[j] + a.lz[kk[q]];
        const double la = a.lx[ia] + lyz, lb = a.lx[ib] + lyz;
        if (a.singular) {
          const double oa = a.ox[ia] * a.oy[j], ob = a.ox[ib] * a.oy[j];
          if (kk[q] == 0) {
            ca[q] -= mf * oa * (double)N;
            cb[q] -= mf * ob * (double)N;
          }
          ca[q] *= fabs(la) <= a.tol ? 0.0 : 1.0 / la;
          cb[q] *= fabs(lb) <= a.tol ? 0.0 : 1.0 / lb;
          if (kk[q] == 0) acc += (oa * ca[q] + ob * cb[q]) / a.h;
        } else {
          ca[q] /= la;
          cb[q] /= lb;
        }
      }
      const double2 wc = make_double2(w.x, -w.y);
      A = cmul(make_double2(ca[0], -ca[1]), wc);  // V'[k] = e^{+iπk/2N} (C'[k] - i C'[N-k])
      B = cmul(make_double2(cb[0], -cb[1]), wc);
      buf[pk * TK + c] = make_double2(inv * (A.x - B.y), inv * (A.y + B.x));
      if (!self) buf[pm * TK + c] = make_double2(inv * (A.x + B.y), inv * (B.x - A.y));
      continue;
    }
    const double lyz = a.ly[j] + a.lz[k];
    const double la = a.lx[ia] + lyz, lb = a.lx[ib] + lyz;
    if (a.singular) {
      const double oa = a.ox[ia] * a.oy[j], ob = a.ox[ib] * a.oy[j];
      if (k == 0) {
        A.x -= mf * oa * (double)N;
        B.x -= mf * ob * (double)N;
      }
      const double sa = fabs(la) <= a.tol ? 0.0 : 1.0 / la, sb = fabs(lb) <= a.tol ? 0.0 : 1.0 / lb;
      A.x *= sa;
      A.y *= sa;
      B.x *= sb;
      B.y *= sb;
      if (k == 0) acc += (oa * A.x + ob * B.x) / a.h;
    } else {
      A.x /= la;
      A.y /= la;
      B.x /= lb;
      B.y /= lb;
    }
    buf[pk * TK + c] = make_double2(inv * (A.x - B.y), inv * (A.y + B.x));
    if (!self) buf[pm * TK + c] = make_double2(inv * (A.x + B.y), inv * (B.x - A.y));
  }
  __syncthreads();
  fft_dit<LOGN, TK, TK, 1, false>(buf, tw, t);
  __syncthreads();
  if (live)
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int z = t / TK + q * RPT;
      const int r = DCT ? ((z & 1) ? N - 1 - (z >> 1) : (z >> 1)) : z;
      a.data[(long long)z * a.nl + line] = buf[r * TK + col];
    }
  if (a.singular) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((t & 63) == 0) red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) a.partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

template <typename K>
int set_lds(K kernel, size_t lds) {
  if (lds > 64 * 1024) INS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return INS_OK;
}

template <int LOGN, typename C = double2>
int launch_y(C* data, int kxn, int kxs, int nplanes, const C* tw, bool inverse, const PackMap* pm, hipStream_t s) {
  constexpr int N = fft_len(LOGN);
  // 256-B segments per row where LDS allows (<= 80 KB tiles): 16 / 8 / 4 double2 columns, twice as many float2 ones
  constexpr int TK = (N <= 256 ? 16 : (N <= 512 ? 8 : 4)) * (int)(sizeof(double2) / sizeof(C));
  constexpr size_t lds = ((size_t)N * TK + N) * sizeof(C);
  dim3 grid((kxn + TK - 1) / TK, nplanes);
  PackMap none{nullptr, 1, 1, 1};
#define INS_Y_LAUNCH(INV, PK)                                                                                   \
  do {                                                                                                          \
    int rc = set_lds(&k_yfft<LOGN, TK, INV, PK, C>, lds);                                                       \
    if (rc) return rc;                                                                                          \
    hipLaunchKernelGGL((k_yfft<LOGN, TK, INV, PK, C>), grid, dim3(256), lds, s, data, kxn, kxs, tw, pm ? *pm : none);   \
  } while (0)
  if constexpr (sizeof(C) == sizeof(double2)) {
    if (pm) {
      if (inverse)
        INS_Y_LAUNCH(true, true);
      else
        INS_Y_LAUNCH(false, true);
      INS_LAUNCH_CHECK();
      return INS_OK;
    }
  }
  if (inverse)
    INS_Y_LAUNCH(true, false);
  else
    INS_Y_LAUNCH(false, false);
#undef INS_Y_LAUNCH
  INS_LAUNCH_CHECK();
  return INS_OK;
}

template <int LOGN>
int launch_xfwd(const GridDev& g, const double* src, int from_u, double2* out, int n1, int n2, const double2* tw, int kxs, hipStream_t s, int kz0) {
  constexpr int N = fft_len(LOGN);
  constexpr int NP = fft_r3(LOGN) == 5 ? 640 / N : (fft_r3(LOGN) == 3 ? 768 / N : (N >= 1024 ? 2 : (1024 / N > 16 ? 16 : 1024 / N)));  // 2 NP N a multiple of 256
  constexpr size_t lds = ((size_t)NP * N + N) * sizeof(double2);
  dim3 grid((n1 + 2 * NP - 1) / (2 * NP), n2);
  if (from_u == 6)
    hipLaunchKernelGGL((k_xfwd<LOGN, NP, 6>), grid, dim3(256), lds, s, g, src, out, n1, tw, kxs, kz0, (int)ins_opt(OPT_INS_X_SKEL));
  else if (from_u == 5)
    hipLaunchKernelGGL((k_xfwd<LOGN, NP, 5>), grid, dim3(256), lds, s, g, src, out, n1, tw, kxs, kz0, (int)ins_opt(OPT_INS_X_SKEL));
  else if (from_u == 4)
    hipLaunchKernelGGL((k_xfwd<LOGN, NP, 4>), grid, dim3(256), lds, s, g, src, out, n1, tw, kxs, kz0, (int)ins_opt(OPT_INS_X_SKEL));
  else if (from_u == 3)
    hipLaunchKernelGGL((k_xfwd<LOGN, NP, 3>), grid, dim3(256), lds, s, g, src, out, n1, tw, kxs, kz0, (int)ins_opt(OPT_INS_X_SKEL));
  else if (from_u == 2)
    hipLaunchKernelGGL((k_xfwd<LOGN, NP, 2>), grid, dim3(256), lds, s, g, src, out, n1, tw, kxs, kz0, (int)ins_opt(OPT_INS_X_SKEL));
  else if (from_u)
    hipLaunchKernelGGL((k_xfwd<LOGN, NP, 1>), grid, dim3(256), lds, s, g, src, out, n1, tw, kxs, kz0, (int)ins_opt(OPT_INS_X_SKEL));
  else
    hipLaunchKernelGGL((k_xfwd<LOGN, NP, 0>), grid, dim3(256), lds, s, g, src, out, n1, tw, kxs, kz0, (int)ins_opt(OPT_INS_X_SKEL));
  INS_LAUNCH_CHECK();
  return INS_OK;
}
// float2 spectrum: from the float pI (from_u = 0) or with Ω·div(u) of the float velocity field formed on the fly (from_u != 0)
template <int LOGN>
int launch_xfwd32(const GridDev& g, const float* src, int from_u, float2* out, int n1, int n2, const float2* tw, int kxs, hipStream_t s) {
  constexpr int N = fft_len(LOGN);
  constexpr int NP = fft_r3(LOGN) == 5 ? 640 / N : (fft_r3(LOGN) == 3 ? 768 / N : (N >= 1024 ? 2 : (1024 / N > 16 ? 16 : 1024 / N)));
  constexpr size_t lds = ((size_t)NP * N + N) * sizeof(float2);
  dim3 grid((n1 + 2 * NP - 1) / (2 * NP), n2);
  const double* srcd = reinterpret_cast<const double*>(src);  // the kernel reads float data behind this pointer (C = float2)
  if (from_u)
    hipLaunchKernelGGL((k_xfwd<LOGN, NP, 5, float2>), grid, dim3(256), lds, s, g, srcd, out, n1, tw, kxs, 0, (int)ins_opt(OPT_INS_X_SKEL));
  else
    hipLaunchKernelGGL((k_xfwd<LOGN, NP, 0, float2>), grid, dim3(256), lds, s, g, srcd, out, n1, tw, kxs, 0, (int)ins_opt(OPT_INS_X_SKEL));
  INS_LAUNCH_CHECK();
  return INS_OK;
}

template <int LOGN, typename C = double2>
int launch_xinv(const C* in, real_t<C>* pI, int n1, int n2, const C* tw, int kxs, hipStream_t s) {
  constexpr int N = fft_len(LOGN);
  constexpr int NP = fft_r3(LOGN) == 5 ? 640 / N : (fft_r3(LOGN) == 3 ? 768 / N : (N >= 1024 ? 2 : (1024 / N > 16 ? 16 : 1024 / N)));  // 2 NP N a multiple of 256
  constexpr size_t lds = ((size_t)NP * N + N) * sizeof(C);
  dim3 grid((n1 + 2 * NP - 1) / (2 * NP), n2);
  hipLaunchKernelGGL((k_xinv<LOGN, NP, C>), grid, dim3(256), lds, s, in, pI, n1, tw, kxs, (int)ins_opt(OPT_INS_X_SKEL));
  INS_LAUNCH_CHECK();
  return INS_OK;
}


// ------------------------------------------------------------------------------------------------------------
// Four passes instead of five: the z direction of the solve rides on the two y passes.
//
// After the x and y transforms the problem is one periodic tridiagonal system per (kx, ky) line,
//     (âx + ây) p_k + c (2 p_k - p_{k+1} - p_{k-1}) = -f̂_k / (nx ny),      c = Ω/Δz²,
// whose circulant matrix has exactly the eigenvalues âx + ây + âz the reference divides by (pressure.jl:326-341): the same linear system as the
// z-FFT · symbol · inverse z-FFT pass (ins_zsolve.hip), solved by the partition (SPIKE) method of ins_ztri.hip with the partitions INSIDE one GPU:
//   k_yz_fwd   : a workgroup owns TK consecutive kx and ONE z-partition of m planes; plane after plane it loads the tile, runs the y-FFT in LDS and
//                applies the forward elimination of its block to the spectrum it has just produced (the recurrence is elementwise in (kx, ky): every
//                work-item carries the state of its own elements from plane to plane), and leaves the two interface numbers of every line;
//   k_yz_iface : the block-circulant 2P x 2P interface system per line (a P-point DFT over partitions + 2x2 systems), and the singular line;
//   k_yz_bwd   : back substitution, plane after plane from the top of the partition, each plane followed by the inverse y-FFT in LDS.
// Closed forms of pivots and spikes in the decaying root r of r + 1/r = (âx + ây)/c + 2 as in ins_ztri.hip (nothing is stored but the field).
// Per solve this moves 16 B per cell less (one of the five read+write passes over the spectrum) plus 4 complex numbers per line and partition.
// ------------------------------------------------------------------------------------------------------------
struct YzArgs {
  double2* data;      // [n2][N][kxs]
  int kxn, kxs, n2, m, P;
  const double* ax;   // [kxn]
  const double* ay;   // [N], storage (digit-reversed) order of ky
  double c, scale;    // Ω/Δz², -1/(nx ny)
  double2* edges;     // [P][stride]: yF[lines] | yL[lines] | the singular line's m right-hand sides;  lines = N kxn, line = pos(ky) kxn + kx
  long long stride;
  double2* bc;        // [P][2][lines]: p_last of the partition below, p_first of the partition above
  double2* sol0;      // [n2]: the singular line's solution
  int skel;           // measurement only (INS_YZ_SKEL): 1 = no y-FFT, 2 = no recurrence arithmetic — wrong results by design
};

__device__ __forceinline__ double yz_rcp(double x) {  // hardware estimate + one Newton step (as ins_ztri.hip)
  const double y = __builtin_amdgcn_rcp(x);
  return y * (2.0 - x * y);
}

// B planes per round (their y-FFTs run side by side in LDS: a 256 x 8 tile alone gives a 1024-work-item workgroup half a butterfly per work-item and
// stage); the loads of rounds r + 1 and r + 2 are in flight while round r is transformed (one workgroup per CU: nobody else hides the HBM latency).
// PF: rounds of loads in flight behind the one being transformed (2 where the registers allow it; 1 for N = 512: a round there is long enough to cover the latency)
template <int LOGN, int TK, int NT, int B, int PF>
__global__ __launch_bounds__(NT) void k_yz_fwd(YzArgs a, const double2* __restrict__ tw_g) {
  constexpr int N = fft_len(LOGN);
  constexpr int RPT = NT / TK, NIT = N / RPT;  // rows per sweep of the workgroup, (row, kx) elements per work-item
  constexpr int NC = TK * B;                   // LDS columns: plane b of the round, kx column c -> b * TK + c
  static_assert(N % RPT == 0 && NIT >= 1, "tile shape");
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double2* buf = reinterpret_cast<double2*>(lds_raw);  // [N][NC]
  double2* tw = buf + N * NC;                          // [N]
  const int t = threadIdx.x, col = t % TK, r0 = t / TK;
  const int kx = blockIdx.x * TK + col;
  const bool live = kx < a.kxn;
  const int part = blockIdx.y, z0 = part * a.m;
  const long long ps = (long long)N * a.kxs;  // plane stride
  double2* base = a.data + (long long)z0 * ps + kx;
  for (int i = t; i < N; i += NT) tw[i] = tw_g[i];
  const long long lines = (long long)N * a.kxn;
  // per-element constants and state
  double r_[NIT], pa[NIT];
  double2 gp[NIT], SA[NIT];
  bool sing[NIT];
#pragma unroll
  for (int q = 0; q < NIT; ++q) {
    const int row = r0 + q * RPT;
    const double sxy = live ? a.ax[kx] + a.ay[row] : 1.0;
    sing[q] = sxy == 0.0;
    const double sc = sing[q] ? 1.0 : sxy / a.c;
    const double sq = sqrt(sc * (4.0 + sc));
    const double lnr = -log1p(0.5 * (sc + sq));  // r = 2 / (s + 2 + sq): the decaying root, without cancellation
    r_[q] = exp(lnr);
    pa[q] = r_[q];
    gp[q] = make_double2(0.0, 0.0);
    SA[q] = gp[q];
  }
  const int R = a.m / B;  // rounds (even: the host checks)
  auto load_round = [&](double2 (&v)[B][NIT], int r) {
    const int rc = min(r, R - 1);  // (past the partition: re-read the last round instead of branching)
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int q = 0; q < NIT; ++q)
        if (live) v[b][q] = base[(long long)(rc * B + b) * ps + (long long)(r0 + q * RPT) * a.kxs];
  };
  // gfx9 counts loads and stores in ONE counter and only loads return in order: once a store is outstanding, waiting for an older load means waiting for
  // everything (vmcnt(0)) — this round's stores AND the loads issued two rounds ahead.  So the next round's loads (issued a round ago) are waited for
  // here, BEFORE this round's stores are issued, while only loads are outstanding (`pin`: an empty asm that reads the registers).
  auto pin = [&](double2 (&v)[B][NIT]) {
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int q = 0; q < NIT; ++q) asm volatile("" ::"v"(v[b][q].x), "v"(v[b][q].y));
  };
  auto round = [&](double2 (&v)[B][NIT], double2 (&vnext)[B][NIT], int r) {
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int q = 0; q < NIT; ++q) buf[(r0 + q * RPT) * NC + b * TK + col] = v[b][q];
    __syncthreads();
    load_round(v, r + PF);  // PF rounds ahead
    if (a.skel != 1) fft_dif<LOGN, NC, NC, 1, false, NT>(buf, tw, t);
    pin(vnext);
    if (live) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int k = r * B + b;
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
          const int row = r0 + q * RPT;
          double2 g = buf[row * NC + b * TK + col];
          g.x *= a.scale;
          g.y *= a.scale;
          double2 out = g;
          if (sing[q]) {
            a.edges[(long long)part * a.stride + 2 * lines + k] = g;  // the singular line: its right-hand side goes to k_yz_iface
          } else if (a.skel != 2) {
            const double rr = r_[q], E = pa[q] * pa[q];
            const double inv = (rr / a.c) * (1.0 - E) * yz_rcp(1.0 - E * rr * rr);  // 1/den_k = (r/c)(1 - E_k)/(1 - E_k r²), E_k = r^(2k+2)
            gp[q].x = inv * (g.x + a.c * gp[q].x);
            gp[q].y = inv * (g.y + a.c * gp[q].y);
            SA[q].x += pa[q] * g.x;  // Σ r^(k+1) g_k
            SA[q].y += pa[q] * g.y;
            pa[q] *= rr;
            out = gp[q];
          }
          base[(long long)k * ps + (long long)row * a.kxs] = out;
        }
      }
    }
    __syncthreads();  // every read of this round's spectra is done before the next round overwrites the tiles
  };
  double2 va[B][NIT], vb[B][NIT];
#pragma unroll
  for (int b = 0; b < B; ++b)
#pragma unroll
    for (int q = 0; q < NIT; ++q) va[b][q] = vb[b][q] = make_double2(0.0, 0.0);
  load_round(va, 0);
  if constexpr (PF == 2) {
    load_round(vb, 1);
    for (int r = 0; r < R; r += 2) {
      round(va, vb, r);
      round(vb, va, r + 1);
    }
  } else {
    for (int r = 0; r < R; ++r) round(va, va, r);
  }
  if (live) {
    // (A⁻¹ g)_last = gp_{m-1};  (A⁻¹ g)_first = (SA - r^(m+1) SB)/(c D) with SB = Σ r^(m-k) g_k = c D gp_{m-1} + r^(m+1) SA
    double2* e = a.edges + (long long)part * a.stride;
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const long long l = (long long)(r0 + q * RPT) * a.kxn + kx;
      if (sing[q]) {
        e[l] = e[lines + l] = make_double2(0.0, 0.0);
      } else {
        const double lnr = log(r_[q]);
        const double rm1 = exp((a.m + 1.0) * lnr);                      // r^(m+1)
        const double idc = 1.0 / (-expm1((2.0 * a.m + 2.0) * lnr) * a.c);  // 1/(c D), D = 1 - r^(2m+2)
        const double f = idc * (1.0 - rm1 * rm1);
        e[l] = make_double2(f * SA[q].x - rm1 * gp[q].x, f * SA[q].y - rm1 * gp[q].y);
        e[lines + l] = gp[q];
      }
    }
  }
}

// interface system of every line for all P partitions (ins_ztri.hip k_ztri_iface solves it for one rank):
//   F_q - α L_{q-1} - β F_{q+1} = yF_q,   L_q - β L_{q-1} - α F_{q+1} = yL_q,   α = v_0, β = v_{m-1};   bc[q] = (L_{q-1}, F_{q+1})
template <int PT>
__global__ __launch_bounds__(64) void k_yz_iface(YzArgs a) {
  __shared__ double2 tw[PT];
  constexpr int P = PT;
  const long long lines = (long long)(a.stride - a.m) / 2;
  if (blockIdx.x == gridDim.x - 1) {
    // extra workgroup: the singular line.  -c (p_{k+1} - 2 p_k + p_{k-1}) = g_k - mean(g), periodic, mean(p) = 0 (pressure.jl:336-341 gauge):
    // q_k = p_{k+1} - p_k = q_0 - S_k, S_k = Σ_{j=1..k} h_j, h = (g - ḡ)/c; p_k = p_0 + k q_0 - T_k, T_k = Σ_{j<k} S_j; q_0 = T_N/N.  One wavefront, serial chunks.
    const int lane = threadIdx.x, Nz = a.n2;
    const int C = (Nz + 63) / 64, s0 = min(lane * C, Nz), e0 = min(s0 + C, Nz);
    auto g_at = [&](int k) { return a.edges[(long long)(k / a.m) * a.stride + 2 * lines + (k % a.m)]; };
    auto wsum = [&](double2 v) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        v.x += __shfl_xor(v.x, off, 64);
        v.y += __shfl_xor(v.y, off, 64);
      }
      return v;
    };
    auto wscan = [&](double2 v) {  // exclusive prefix sum over lanes
      double2 inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double tx = __shfl_up(inc.x, off, 64), ty = __shfl_up(inc.y, off, 64);
        if (lane >= off) {
          inc.x += tx;
          inc.y += ty;
        }
      }
      return make_double2(inc.x - v.x, inc.y - v.y);
    };
    const double ic = 1.0 / a.c, iN = 1.0 / Nz;
    double2 sum = make_double2(0.0, 0.0);
    for (int k = s0; k < e0; ++k) {
      const double2 g = g_at(k);
      sum.x += g.x;
      sum.y += g.y;
    }
    sum = wsum(sum);
    const double2 gbar = make_double2(iN * sum.x, iN * sum.y);
    auto h_at = [&](int k) {
      const double2 g = g_at(k);
      return make_double2((g.x - gbar.x) * ic, (g.y - gbar.y) * ic);
    };
    const double2 h0 = h_at(0);
    double2 A = make_double2(0.0, 0.0);
    for (int k = s0; k < e0; ++k) {
      const double2 h = h_at(k);
      A.x += h.x;
      A.y += h.y;
    }
    const double2 Hbase = wscan(A);
    double2 B = make_double2(0.0, 0.0), H = Hbase;
    for (int k = s0; k < e0; ++k) {
      const double2 h = h_at(k);
      H.x += h.x;
      H.y += h.y;
      B.x += H.x - h0.x;
      B.y += H.y - h0.y;
    }
    const double2 Tbase = wscan(B);
    const double2 Bs = wsum(B);
    const double2 q0 = make_double2(iN * Bs.x, iN * Bs.y);
    double2 Cs = make_double2(0.0, 0.0), T = Tbase;
    H = Hbase;
    for (int k = s0; k < e0; ++k) {
      Cs.x += k * q0.x - T.x;
      Cs.y += k * q0.y - T.y;
      const double2 h = h_at(k);
      H.x += h.x;
      H.y += h.y;
      T.x += H.x - h0.x;
      T.y += H.y - h0.y;
    }
    Cs = wsum(Cs);
    const double2 p0 = make_double2(-iN * Cs.x, -iN * Cs.y);
    T = Tbase;
    H = Hbase;
    for (int k = s0; k < e0; ++k) {
      a.sol0[k] = make_double2(p0.x + k * q0.x - T.x, p0.y + k * q0.y - T.y);
      const double2 h = h_at(k);
      H.x += h.x;
      H.y += h.y;
      T.x += H.x - h0.x;
      T.y += H.y - h0.y;
    }
    return;
  }
  if ((int)threadIdx.x < P) {
    double sn, cs;
    sincospi(-2.0 * (double)threadIdx.x / P, &sn, &cs);
    tw[threadIdx.x] = make_double2(cs, sn);
  }
  __syncthreads();
  const long long l = (long long)blockIdx.x * 64 + threadIdx.x;
  if (l >= lines) return;
  const int ky = (int)(l / a.kxn), kx = (int)(l - (long long)ky * a.kxn);
  const double sxy = a.ax[kx] + a.ay[ky];
  if (sxy == 0.0) {
#pragma unroll
    for (int q = 0; q < P; ++q) a.bc[((long long)q * 2) * lines + l] = a.bc[((long long)q * 2 + 1) * lines + l] = make_double2(0.0, 0.0);
    return;
  }
  const double sc = sxy / a.c, sq = sqrt(sc * (4.0 + sc)), lnr = -log1p(0.5 * (sc + sq));
  const double r = exp(lnr), D = -expm1((2.0 * a.m + 2.0) * lnr);
  const double alpha = r * (-expm1(2.0 * a.m * lnr)) / D;
  const double beta = exp(a.m * lnr) * (1.0 - r * r) / D;
  auto cm = [](double2 x, double2 y) { return make_double2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x); };
  double2 hF[PT], hL[PT];
#pragma unroll
  for (int j = 0; j < P; ++j) hF[j] = hL[j] = make_double2(0.0, 0.0);
#pragma unroll
  for (int q = 0; q < P; ++q) {  // forward DFT over partitions: e^{-iθ_j q}
    const double2 yF = a.edges[(long long)q * a.stride + l], yL = a.edges[(long long)q * a.stride + lines + l];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const double2 w = tw[(j * q) % P];
      const double2 x = cm(w, yF), y = cm(w, yL);
      hF[j].x += x.x;
      hF[j].y += x.y;
      hL[j].x += y.x;
      hL[j].y += y.y;
    }
  }
#pragma unroll
  for (int j = 0; j < P; ++j) {
    const double2 em = tw[j], ep = make_double2(em.x, -em.y);  // e^{-iθ}, e^{+iθ}
    // [ 1 - β e^{+iθ}    -α e^{-iθ} ] [F̂]   [ŷF]
    // [   -α e^{+iθ}   1 - β e^{-iθ} ] [L̂] = [ŷL]
    const double2 a11 = make_double2(1.0 - beta * ep.x, -beta * ep.y), a22 = make_double2(1.0 - beta * em.x, -beta * em.y);
    const double2 a12 = make_double2(-alpha * em.x, -alpha * em.y), a21 = make_double2(-alpha * ep.x, -alpha * ep.y);
    const double idet = 1.0 / (1.0 - 2.0 * beta * em.x + beta * beta - alpha * alpha);
    const double2 x1 = cm(a22, hF[j]), x2 = cm(a12, hL[j]), y1 = cm(a11, hL[j]), y2 = cm(a21, hF[j]);
    hF[j] = make_double2(idet * (x1.x - x2.x), idet * (x1.y - x2.y));
    hL[j] = make_double2(idet * (y1.x - y2.x), idet * (y1.y - y2.y));
  }
  const double iP = 1.0 / P;
#pragma unroll
  for (int q = 0; q < P; ++q) {  // inverse DFT at partitions q-1 (its last value) and q+1 (its first value): e^{+iθ_j q'}
    const int qp = (q + P - 1) % P, qn = (q + 1) % P;
    double2 Lp = make_double2(0.0, 0.0), Fn = Lp;
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const double2 wp = tw[(j * qp) % P], wn = tw[(j * qn) % P];
      const double2 x = cm(make_double2(wp.x, -wp.y), hL[j]), y = cm(make_double2(wn.x, -wn.y), hF[j]);
      Lp.x += x.x;
      Lp.y += x.y;
      Fn.x += y.x;
      Fn.y += y.y;
    }
    a.bc[((long long)q * 2) * lines + l] = make_double2(iP * Lp.x, iP * Lp.y);
    a.bc[((long long)q * 2 + 1) * lines + l] = make_double2(iP * Fn.x, iP * Fn.y);
  }
}

// back substitution with the interface values folded in (ins_ztri.hip k_ztri_bwd), plane after plane from the top of the partition, each plane
// followed by the inverse y-FFT:   g̃ = g + c L e_0 + c F e_{m-1}  =>  gp̃_k = gp_k + c L φ_k (+ c F/den_{m-1} at k = m-1),  φ_k = r^(k+1)(1-r²)/(c(1-E_k r²));
//   p_{m-1} = gp̃_{m-1},  p_k = gp̃_k + (c/den_k) p_{k+1}
template <int LOGN, int TK, int NT, int B, int PF>
__global__ __launch_bounds__(NT) void k_yz_bwd(YzArgs a, const double2* __restrict__ tw_g) {
  constexpr int N = fft_len(LOGN);
  constexpr int RPT = NT / TK, NIT = N / RPT;
  constexpr int NC = TK * B;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double2* buf = reinterpret_cast<double2*>(lds_raw);
  double2* tw = buf + N * NC;
  const int t = threadIdx.x, col = t % TK, r0 = t / TK;
  const int kx = blockIdx.x * TK + col;
  const bool live = kx < a.kxn;
  const int part = blockIdx.y, z0 = part * a.m;
  const long long ps = (long long)N * a.kxs;
  double2* base = a.data + (long long)z0 * ps + kx;
  for (int i = t; i < N; i += NT) tw[i] = tw_g[i];
  // the singular line's solution on this partition's planes (selecting it per element inside the recurrence — `sing ? sol : p` — crashes ROCm 7.2's
  // backend in the 1024-work-item instantiation, MachineCopyPropagation; it is patched into the tile instead)
  double2* sol = tw + N;  // [m]
  for (int i = t; i < a.m; i += NT) sol[i] = a.sol0[z0 + i];
  const long long lines = (long long)N * a.kxn;
  const int R = a.m / B;  // rounds; round r holds planes m-1-rB ... m-rB-B (b = 0 is the upper one)
  auto load_round = [&](double2 (&v)[B][NIT], int r) {
    const int rc = min(r, R - 1);
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int q = 0; q < NIT; ++q)
        if (live) v[b][q] = base[(long long)(a.m - 1 - rc * B - b) * ps + (long long)(r0 + q * RPT) * a.kxs];
  };
  double2 va[B][NIT], vb[B][NIT];
#pragma unroll
  for (int b = 0; b < B; ++b)
#pragma unroll
    for (int q = 0; q < NIT; ++q) va[b][q] = vb[b][q] = make_double2(0.0, 0.0);
  load_round(va, 0);
  if constexpr (PF == 2) load_round(vb, 1);
  double r_[NIT], pa[NIT];
  double2 p[NIT], Lp[NIT];
#pragma unroll
  for (int q = 0; q < NIT; ++q) {
    const int row = r0 + q * RPT;
    const double sxy = live ? a.ax[kx] + a.ay[row] : 1.0;
    const double sc = sxy == 0.0 ? 1.0 : sxy / a.c;  // (the singular line runs on a placeholder and is patched below)
    const double sq = sqrt(sc * (4.0 + sc));
    const double lnr = -log1p(0.5 * (sc + sq));
    const double r = exp(lnr);
    r_[q] = r;
    pa[q] = exp((double)a.m * lnr);  // r^(k+1) at k = m - 1
    const long long l = (long long)row * a.kxn + kx;
    Lp[q] = live ? a.bc[((long long)part * 2) * lines + l] : make_double2(0.0, 0.0);
    const double2 Fn = live ? a.bc[((long long)part * 2 + 1) * lines + l] : make_double2(0.0, 0.0);
    // the upper interface value enters at the last plane of the partition only: fold c/den_{m-1} F into that plane's right-hand side here
    const double E = pa[q] * pa[q];
    const double cinv = r * (1.0 - E) * yz_rcp(1.0 - E * r * r);
    va[0][q].x += cinv * Fn.x;
    va[0][q].y += cinv * Fn.y;
    p[q] = make_double2(0.0, 0.0);
  }
  auto pin = [&](double2 (&v)[B][NIT]) {  // see k_yz_fwd
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int q = 0; q < NIT; ++q) asm volatile("" ::"v"(v[b][q].x), "v"(v[b][q].y));
  };
  auto round = [&](double2 (&v)[B][NIT], double2 (&vnext)[B][NIT], int r) {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int k = a.m - 1 - r * B - b;
      // r^(k+1): a running product re-seeded from exp() every 16 planes (a seed that underflows cannot become significant within 16 steps: r >= ~0.1
      // on these grids; ins_ztri.hip)
      if ((k & 15) == 15 && k != a.m - 1) {
#pragma unroll
        for (int q = 0; q < NIT; ++q) pa[q] = exp((k + 1.0) * log(r_[q]));
      }
#pragma unroll
      for (int q = 0; q < NIT; ++q) {
        const double rr = r_[q];
        const double E = pa[q] * pa[q], qq = yz_rcp(1.0 - E * rr * rr);
        const double cinv = rr * (1.0 - E) * qq;          // c / den_k
        const double phi = pa[q] * (1.0 - rr * rr) * qq;  // c φ_k
        p[q].x = (v[b][q].x + phi * Lp[q].x) + cinv * p[q].x;
        p[q].y = (v[b][q].y + phi * Lp[q].y) + cinv * p[q].y;
        pa[q] *= yz_rcp(rr);
        buf[(r0 + q * RPT) * NC + b * TK + col] = p[q];
      }
      // the singular line (kx = 0, ky = 0: storage row 0 of the first tile, owned by work-item 0, whose own recurrence ran on a placeholder)
      if (t == 0 && blockIdx.x == 0) buf[b * TK] = sol[k];
    }
    __syncthreads();
    load_round(v, r + PF);
    if (a.skel != 1) fft_dit<LOGN, NC, NC, 1, false, NT>(buf, tw, t);
    pin(vnext);
    if (live) {
#pragma unroll
      for (int b = 0; b < B; ++b)
#pragma unroll
        for (int q = 0; q < NIT; ++q)
          base[(long long)(a.m - 1 - r * B - b) * ps + (long long)(r0 + q * RPT) * a.kxs] = buf[(r0 + q * RPT) * NC + b * TK + col];
    }
    __syncthreads();
  };
  if constexpr (PF == 2) {
    for (int r = 0; r < R; r += 2) {
      round(va, vb, r);
      round(vb, va, r + 1);
    }
  } else {
    for (int r = 0; r < R; ++r) round(va, va, r);
  }
}

template <int LOGN>
int launch_yz(YzArgs& a, const double2* tw, hipStream_t s) {
  constexpr int N = fft_len(LOGN);
  constexpr int TK = 8, NT = 1024;
  constexpr int B = N <= 256 ? 2 : 1;  // planes per round: 64 KB of tiles either way
  constexpr int PF = N <= 256 ? 2 : 1;
  if (a.m % (PF * B)) {
    ins_set_error("fused y/z passes: %d planes per partition, need a multiple of %d", a.m, PF * B);
    return INS_ERR_UNSUPPORTED;
  }
  const size_t lds = ((size_t)N * TK * B + N + a.m) * sizeof(double2);
  const dim3 grid((a.kxn + TK - 1) / TK, a.P);
  int rc = set_lds(&k_yz_fwd<LOGN, TK, NT, B, PF>, lds);
  if (rc) return rc;
  if ((rc = set_lds(&k_yz_bwd<LOGN, TK, NT, B, PF>, lds))) return rc;
  hipLaunchKernelGGL((k_yz_fwd<LOGN, TK, NT, B, PF>), grid, dim3(NT), lds, s, a, tw);
  const long long lines = (long long)N * a.kxn;
  const dim3 gi((unsigned)((lines + 63) / 64) + 1);
  switch (a.P) {
    case 2: hipLaunchKernelGGL(k_yz_iface<2>, gi, dim3(64), 0, s, a); break;
    case 4: hipLaunchKernelGGL(k_yz_iface<4>, gi, dim3(64), 0, s, a); break;
    case 8: hipLaunchKernelGGL(k_yz_iface<8>, gi, dim3(64), 0, s, a); break;
    case 16: hipLaunchKernelGGL(k_yz_iface<16>, gi, dim3(64), 0, s, a); break;
    default: ins_set_error("fused y/z passes: %d partitions not built", a.P); return INS_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL((k_yz_bwd<LOGN, TK, NT, B, PF>), grid, dim3(NT), lds, s, a, tw);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Small planes (n0, n1 <= 64): the x and the y pass of one kz-plane in ONE kernel.  A 64 x 64 plane is 32 KB as row pairs and 33 KB as half
// spectrum, so a workgroup holds the whole plane in LDS: x forward (paired-row real transform) -> split of the pairs into spec[ky][kx] -> y forward
// (same storage order of ky as k_yfft) -> global; and back.  Boxes of 16^3 .. 64^3 are bound by the latency of their dependent launches (~5.7 us
// each): this makes a stage four launches (stage kernel, xy forward, z solve, xy inverse) instead of six.
//   SRC 0: rows from pI; SRC 1: rows = Ω·div(u) with periodic wrap (as k_xfwd).
// ------------------------------------------------------------------------------------------------------------
template <int LX, int LY, int SRC, int NT>
__global__ __launch_bounds__(NT) void k_xyfwd(GridDev g, const double* __restrict__ src, double2* __restrict__ out, const double2* __restrict__ twx_g,
                                               const double2* __restrict__ twy_g, int kxs) {
  using C = double2;
  constexpr int N0 = 1 << LX, N1 = 1 << LY, KXN = N0 / 2 + 1, NP = N1 / 2, KP = KXN;
  extern __shared__ __align__(16) unsigned char lds_raw_xy[];
  C* bufx = reinterpret_cast<C*>(lds_raw_xy);  // [NP][N0]: row pairs (real part = row 2p, imaginary part = row 2p + 1)
  C* spec = bufx + NP * N0;                      // [N1][KP]
  C* twx = spec + N1 * KP;                       // [N0]
  C* twy = twx + N0;                             // [N1]
  const int t = threadIdx.x, kz = blockIdx.x;
  for (int m = t; m < N0; m += NT) twx[m] = twx_g[m];
  for (int m = t; m < N1; m += NT) twy[m] = twy_g[m];
  double* bufd = reinterpret_cast<double*>(bufx);
  for (int idx = t; idx < N1 * N0; idx += NT) {
    const int j = idx / N0, i = idx - j * N0;
    double v;
    if (SRC == 0) {
      v = src[i + (long long)N0 * (j + (long long)N1 * kz)];
    } else {
      const int I0 = i + 1, I1 = j + 1, I2 = kz + 1;
      const long long c = I0 + I1 * g.sx[1] + I2 * g.sx[2];
      const long long cx = I0 == 1 ? c + (long long)(g.N[0] - 3) : c - 1;
      const long long cy = I1 == 1 ? c + (long long)(g.N[1] - 3) * g.sx[1] : c - g.sx[1];
      const long long cz = I2 == 1 ? c + (long long)(g.N[2] - 3) * g.sx[2] : c - g.sx[2];
      double d = (src[c] - src[cx]) * g.rdx[0][I0];
      d += (src[g.sc + c] - src[g.sc + cy]) * g.rdx[1][I1];
      d += (src[2 * g.sc + c] - src[2 * g.sc + cz]) * g.rdx[2][I2];
      v = d * (g.dx[0][I0] * g.dx[1][I1] * g.dx[2][I2]);
    }
    bufd[2 * ((j >> 1) * N0 + i) + (j & 1)] = v;
  }
  __syncthreads();
  fft_dif<LX, NP, 1, N0, true, NT>(bufx, twx, t);
  // separate the two real rows of every pair: A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / (2i)
  for (int idx = t; idx < NP * KXN; idx += NT) {
    const int p = idx / KXN, s = idx - p * KXN;
    const C zk = bufx[p * N0 + pos_of_freq<LX>(s)];
    const C zm = bufx[p * N0 + pos_of_freq<LX>((N0 - s) % N0)];
    spec[(2 * p) * KP + s] = mkc<C>(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
    spec[(2 * p + 1) * KP + s] = mkc<C>(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
  }
  __syncthreads();
  fft_dif<LY, KXN, KP, 1, false, NT>(spec, twy, t);  // lines = kx (unit stride), elements = ky (KP apart); ky stays in digit-reversed storage order
  C* o = out + (long long)kz * N1 * kxs;
  for (int idx = t; idx < N1 * KXN; idx += NT) {
    const int j = idx / KXN, s = idx - j * KXN;
    o[(long long)j * kxs + s] = spec[j * KP + s];
  }
}

template <int LX, int LY, int NT>
__global__ __launch_bounds__(NT) void k_xyinv(const double2* __restrict__ in, double* __restrict__ pI, const double2* __restrict__ twx_g,
                                               const double2* __restrict__ twy_g, int kxs) {
  using C = double2;
  constexpr int N0 = 1 << LX, N1 = 1 << LY, KXN = N0 / 2 + 1, NP = N1 / 2, KP = KXN;
  extern __shared__ __align__(16) unsigned char lds_raw_xy[];
  C* bufx = reinterpret_cast<C*>(lds_raw_xy);
  C* spec = bufx + NP * N0;
  C* twx = spec + N1 * KP;
  C* twy = twx + N0;
  const int t = threadIdx.x, kz = blockIdx.x;
  for (int m = t; m < N0; m += NT) twx[m] = twx_g[m];
  for (int m = t; m < N1; m += NT) twy[m] = twy_g[m];
  const C* ip = in + (long long)kz * N1 * kxs;
  for (int idx = t; idx < N1 * KXN; idx += NT) {
    const int j = idx / KXN, s = idx - j * KXN;
    spec[j * KP + s] = ip[(long long)j * kxs + s];
  }
  __syncthreads();
  fft_dit<LY, KXN, KP, 1, false, NT>(spec, twy, t);  // back to natural ky
  // Z[k] = A[k] + i B[k], Z[N-k] = conj A[k] + i conj B[k] into the storage order the x DIT expects
  for (int idx = t; idx < NP * KXN; idx += NT) {
    const int p = idx / KXN, s = idx - p * KXN;
    C a = spec[(2 * p) * KP + s], b = spec[(2 * p + 1) * KP + s];
    if (s == 0 || s == N0 / 2) {  // C2R semantics: DC and Nyquist bins are real
      a.y = 0;
      b.y = 0;
    }
    bufx[p * N0 + pos_of_freq<LX>(s)] = mkc<C>(a.x - b.y, a.y + b.x);
    if (s != 0 && s != N0 / 2) bufx[p * N0 + pos_of_freq<LX>(N0 - s)] = mkc<C>(a.x + b.y, b.x - a.y);
  }
  __syncthreads();
  fft_dit<LX, NP, 1, N0, true, NT>(bufx, twx, t);
  const double* bufd = reinterpret_cast<const double*>(bufx);
  for (int idx = t; idx < N1 * N0; idx += NT) {
    const int j = idx / N0, i = idx - j * N0;
    pI[i + (long long)N0 * (j + (long long)N1 * kz)] = bufd[2 * ((j >> 1) * N0 + i) + (j & 1)];
  }
}

// 2-D grids of up to 64 x 64 volumes: the WHOLE spectral solve (x forward with the divergence formed inside, y forward, symbol, y inverse, x inverse) as one
// workgroup — one launch instead of three; such a grid is 4096 volumes, and its step is nothing but launch latency.  SRC 0: rows from pI; SRC 3: Ω·div(u), 2-D, periodic wrap.
template <int LX, int LY, int SRC, int NT>
__global__ __launch_bounds__(NT) void k_xysolve2d(GridDev g, const double* __restrict__ src, double* __restrict__ pI, const double2* __restrict__ twx_g,
                                                  const double2* __restrict__ twy_g, const double* __restrict__ ax, const double* __restrict__ ay, double inv_n) {
  using C = double2;
  constexpr int N0 = 1 << LX, N1 = 1 << LY, KXN = N0 / 2 + 1, NP = N1 / 2, KP = KXN;
  extern __shared__ __align__(16) unsigned char lds_raw_xy[];
  C* bufx = reinterpret_cast<C*>(lds_raw_xy);
  C* spec = bufx + NP * N0;
  C* twx = spec + N1 * KP;
  C* twy = twx + N0;
  const int t = threadIdx.x;
  for (int m = t; m < N0; m += NT) twx[m] = twx_g[m];
  for (int m = t; m < N1; m += NT) twy[m] = twy_g[m];
  double* bufd = reinterpret_cast<double*>(bufx);
  for (int idx = t; idx < N1 * N0; idx += NT) {
    const int j = idx / N0, i = idx - j * N0;
    double v;
    if (SRC == 0) {
      v = src[i + (long long)N0 * j];
    } else {
      const int I0 = i + 1, I1 = j + 1;
      const long long c = I0 + I1 * g.sx[1];
      const long long cx = I0 == 1 ? c + (long long)(g.N[0] - 3) : c - 1;
      const long long cy = I1 == 1 ? c + (long long)(g.N[1] - 3) * g.sx[1] : c - g.sx[1];
      const double d = (src[c] - src[cx]) * g.rdx[0][I0] + (src[g.sc + c] - src[g.sc + cy]) * g.rdx[1][I1];
      v = d * (g.dx[0][I0] * g.dx[1][I1]);
    }
    bufd[2 * ((j >> 1) * N0 + i) + (j & 1)] = v;
  }
  __syncthreads();
  fft_dif<LX, NP, 1, N0, true, NT>(bufx, twx, t);
  for (int idx = t; idx < NP * KXN; idx += NT) {
    const int p = idx / KXN, s = idx - p * KXN;
    const C zk = bufx[p * N0 + pos_of_freq<LX>(s)];
    const C zm = bufx[p * N0 + pos_of_freq<LX>((N0 - s) % N0)];
    spec[(2 * p) * KP + s] = mkc<C>(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
    spec[(2 * p + 1) * KP + s] = mkc<C>(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
  }
  __syncthreads();
  fft_dif<LY, KXN, KP, 1, false, NT>(spec, twy, t);
  // phat = -phat / (âx + ây) / prod(Np), phat[0, 0] = 0 (pressure.jl:326-341); frequency ky sits at storage position pos_of_freq(ky)
  for (int idx = t; idx < N1 * KXN; idx += NT) {
    const int k = idx / KXN, s = idx - k * KXN;
    const double sc = (k == 0 && s == 0) ? 0.0 : -inv_n / (ax[s] + ay[k]);
    C& z = spec[pos_of_freq<LY>(k) * KP + s];
    z.x *= sc;
    z.y *= sc;
  }
  __syncthreads();
  fft_dit<LY, KXN, KP, 1, false, NT>(spec, twy, t);
  for (int idx = t; idx < NP * KXN; idx += NT) {
    const int p = idx / KXN, s = idx - p * KXN;
    C a = spec[(2 * p) * KP + s], b = spec[(2 * p + 1) * KP + s];
    if (s == 0 || s == N0 / 2) {
      a.y = 0;
      b.y = 0;
    }
    bufx[p * N0 + pos_of_freq<LX>(s)] = mkc<C>(a.x - b.y, a.y + b.x);
    if (s != 0 && s != N0 / 2) bufx[p * N0 + pos_of_freq<LX>(N0 - s)] = mkc<C>(a.x + b.y, b.x - a.y);
  }
  __syncthreads();
  fft_dit<LX, NP, 1, N0, true, NT>(bufx, twx, t);
  for (int idx = t; idx < N1 * N0; idx += NT) {
    const int j = idx / N0, i = idx - j * N0;
    pI[i + (long long)N0 * j] = bufd[2 * ((j >> 1) * N0 + i) + (j & 1)];
  }
}

template <int LX, int LY>
int launch_xysolve2d(const GridDev& g, const double* src, int from_u, double* pI, const double2* twx, const double2* twy, const double* ax, const double* ay,
                     hipStream_t s) {
  constexpr int N0 = 1 << LX, N1 = 1 << LY, KXN = N0 / 2 + 1;
  constexpr int NT = N0 * N1 >= 2048 ? 1024 : 256;
  constexpr size_t lds = ((size_t)(N1 / 2) * N0 + (size_t)N1 * KXN + N0 + N1) * sizeof(double2);
  const double inv_n = 1.0 / ((double)N0 * N1);
  int rc;
  if (from_u) {
    if ((rc = set_lds(&k_xysolve2d<LX, LY, 3, NT>, lds))) return rc;
    hipLaunchKernelGGL((k_xysolve2d<LX, LY, 3, NT>), dim3(1), dim3(NT), lds, s, g, src, pI, twx, twy, ax, ay, inv_n);
  } else {
    if ((rc = set_lds(&k_xysolve2d<LX, LY, 0, NT>, lds))) return rc;
    hipLaunchKernelGGL((k_xysolve2d<LX, LY, 0, NT>), dim3(1), dim3(NT), lds, s, g, src, pI, twx, twy, ax, ay, inv_n);
  }
  INS_LAUNCH_CHECK();
  return INS_OK;
}

template <int LX, int LY>
int launch_xy(const GridDev& g, const double* src, int from_u, double2* spec, double* pI, int n2, const double2* twx, const double2* twy, int kxs,
              bool inverse, hipStream_t s) {
  constexpr int N0 = 1 << LX, N1 = 1 << LY, KXN = N0 / 2 + 1;
  constexpr int NT = N0 * N1 >= 2048 ? 1024 : 256;  // a plane is one workgroup's serial work: 1024 work-items from 64 x 32 volumes on
  constexpr size_t lds = ((size_t)(N1 / 2) * N0 + (size_t)N1 * KXN + N0 + N1) * sizeof(double2);
  int rc;
  if (inverse) {
    if ((rc = set_lds(&k_xyinv<LX, LY, NT>, lds))) return rc;
    hipLaunchKernelGGL((k_xyinv<LX, LY, NT>), dim3(n2), dim3(NT), lds, s, spec, pI, twx, twy, kxs);
  } else if (from_u) {
    if ((rc = set_lds(&k_xyfwd<LX, LY, 1, NT>, lds))) return rc;
    hipLaunchKernelGGL((k_xyfwd<LX, LY, 1, NT>), dim3(n2), dim3(NT), lds, s, g, src, spec, twx, twy, kxs);
  } else {
    if ((rc = set_lds(&k_xyfwd<LX, LY, 0, NT>, lds))) return rc;
    hipLaunchKernelGGL((k_xyfwd<LX, LY, 0, NT>), dim3(n2), dim3(NT), lds, s, g, src, spec, twx, twy, kxs);
  }
  INS_LAUNCH_CHECK();
  return INS_OK;
}

}  // namespace

#define INS_POW2_SWITCH(n, CALL)            \
  switch (n) {                              \
    case 16: return CALL(4);                \
    case 32: return CALL(5);                \
    case 64: return CALL(6);                \
    case 128: return CALL(7);               \
    case 256: return CALL(8);               \
    case 512: return CALL(9);               \
    case 1024: return CALL(10);             \
    case 96: return CALL(32 + 5);           \
    case 192: return CALL(32 + 6);          \
    case 384: return CALL(32 + 7);          \
    case 160: return CALL(64 + 5);          \
    case 320: return CALL(64 + 6);          \
    case 640: return CALL(64 + 7);          \
  }                                         \
  ins_set_error("own FFT: unsupported length %d", n); \
  return INS_ERR_UNSUPPORTED;

bool ins_ownfft_supported(const int np[3]) {  // power-of-two boxes (slab and 2-D paths)
  if (ins_opt(OPT_INS_DISABLE_OWNFFT)) return false;
  for (int a = 0; a < 3; ++a)
    if (np[a] < 16 || np[a] > 1024 || (np[a] & (np[a] - 1))) return false;
  return ins_zsolve_supported(np[2]);
}
// z-slab path: own x / y passes for power-of-two and 3 * 2^m sides (the z direction is either the distributed tridiagonal solve — any plane count — or the
// transposes around the fused z kernel / a rocFFT z plan)
bool ins_ownfft_supported_slab(const int np[3]) {
  if (ins_opt(OPT_INS_DISABLE_OWNFFT)) return false;
  for (int a = 0; a < 2; ++a) {
    const bool pow2 = np[a] >= 16 && np[a] <= 1024 && !(np[a] & (np[a] - 1));
    const bool r3 = (np[a] == 96 || np[a] == 192 || np[a] == 384 || np[a] == 160 || np[a] == 320 || np[a] == 640) && !ins_opt(OPT_INS_OWNFFT_POW2_ONLY);
    if (!pow2 && !r3) return false;
  }
  return np[2] >= 2;
}
// single-GPU 3-D solver: sides of 3 * 2^m (192, 384) run on the own passes too (a radix-3 stage in front; INS_OWNFFT_POW2_ONLY keeps rocFFT for them)
bool ins_ownfft_supported_mixed(const int np[3]) {
  if (ins_opt(OPT_INS_DISABLE_OWNFFT)) return false;
  for (int a = 0; a < 3; ++a) {
    const bool pow2 = np[a] >= 16 && np[a] <= 1024 && !(np[a] & (np[a] - 1));
    const bool r3 = (np[a] == 96 || np[a] == 192 || np[a] == 384 || np[a] == 160 || np[a] == 320 || np[a] == 640) && !ins_opt(OPT_INS_OWNFFT_POW2_ONLY);  // 3 * 2^m, 5 * 2^m
    if (!pow2 && !r3) return false;
  }
  return ins_zsolve_supported(np[2]);
}

// ây permuted to the digit-reversed storage order that pass 2 leaves behind: out[pos] = ay[freq(pos)].
void ins_ownfft_permute_symbol(int n, const double* ay, double* out) {
  const int r3 = (n % 5 == 0) ? 5 : ((n % 3 == 0) ? 3 : 1), m = n / r3;
  int logm = 0;
  while ((1 << logm) < m) ++logm;
  for (int k = 0; k < n; ++k) {
    int p = 0, kk = k, L = m;
    if (r3 != 1) {
      p = (kk % r3) * m;
      kk /= r3;
    }
    if (logm & 1) {
      p += (kk & 1) * (L / 2);
      kk >>= 1;
      L /= 2;
    }
    for (int s = 0; s < logm / 2; ++s) {
      p += (kk & 3) * (L / 4);
      kk >>= 2;
      L /= 4;
    }
    out[p] = ay[k];
  }
}

bool ins_ownfft_xy_supported(int n0, int n1) {
  auto small = [](int n) { return n == 16 || n == 32 || n == 64; };
  return !ins_opt(OPT_INS_DISABLE_XYFUSED) && small(n0) && small(n1);
}
// from_u: 0 = rows from pI, 1 = Ω·div(u) formed on the fly (periodic wrap); inverse: spectrum -> pI
int ins_k_ownfft_xy(const ins_grid* G, const double* src, int from_u, double* phat, double* pI, int n0, int n1, int n2, const double* twx, const double* twy,
                    bool inverse, hipStream_t s, int kxs) {
  static const GridDev no_grid{};
  const GridDev& g = G ? G->g : no_grid;
  double2* sp = reinterpret_cast<double2*>(phat);
  const double2 *wx = reinterpret_cast<const double2*>(twx), *wy = reinterpret_cast<const double2*>(twy);
  auto lg = [](int n) { return n == 16 ? 4 : (n == 32 ? 5 : 6); };
#define INS_XY(A, B) \
  if (lg(n0) == A && lg(n1) == B) return launch_xy<A, B>(g, src, from_u, sp, pI, n2, wx, wy, kxs, inverse, s);
  INS_XY(4, 4) INS_XY(4, 5) INS_XY(4, 6) INS_XY(5, 4) INS_XY(5, 5) INS_XY(5, 6) INS_XY(6, 4) INS_XY(6, 5) INS_XY(6, 6)
#undef INS_XY
  ins_set_error("ins_k_ownfft_xy: unsupported plane %d x %d", n0, n1);
  return INS_ERR_UNSUPPORTED;
}

// the whole 2-D solve of a grid of up to 64 x 64 volumes as one launch: pI <- solution of L p = (from_u ? Ω·div(u) : pI)
int ins_k_ownfft_xysolve2d(const ins_grid* G, const double* src, int from_u, double* pI, int n0, int n1, const double* twx, const double* twy, const double* ax,
                           const double* ay, hipStream_t s) {
  static const GridDev no_grid{};
  const GridDev& g = G ? G->g : no_grid;
  const double2 *wx = reinterpret_cast<const double2*>(twx), *wy = reinterpret_cast<const double2*>(twy);
  auto lg = [](int n) { return n == 16 ? 4 : (n == 32 ? 5 : 6); };
#define INS_XY(A, B) \
  if (lg(n0) == A && lg(n1) == B) return launch_xysolve2d<A, B>(g, src, from_u, pI, wx, wy, ax, ay, s);
  INS_XY(4, 4) INS_XY(4, 5) INS_XY(4, 6) INS_XY(5, 4) INS_XY(5, 5) INS_XY(5, 6) INS_XY(6, 4) INS_XY(6, 5) INS_XY(6, 6)
#undef INS_XY
  ins_set_error("ins_k_ownfft_xysolve2d: unsupported grid %d x %d", n0, n1);
  return INS_ERR_UNSUPPORTED;
}

int ins_k_ownfft_xfwd(const ins_grid* G, const double* src, int from_u, double* phat, int n0, int n1, int n2, const double* tw,
                      hipStream_t s, int kxs, int kz0) {
  if (kxs <= 0) kxs = n0 / 2 + 1;
  static const GridDev no_grid{};  // SRC = 0 never touches the grid
  const GridDev& g = G ? G->g : no_grid;
  double2* out = reinterpret_cast<double2*>(phat);
  const double2* w = reinterpret_cast<const double2*>(tw);
#define CALL(LG) launch_xfwd<LG>(g, src, from_u, out, n1, n2, w, kxs, s, kz0)
  INS_POW2_SWITCH(n0, CALL)
#undef CALL
}

int ins_k_ownfft_xinv(const double* phat, double* pI, int n0, int n1, int n2, const double* tw, hipStream_t s, int kxs) {
  if (kxs <= 0) kxs = n0 / 2 + 1;
  const double2* in = reinterpret_cast<const double2*>(phat);
  const double2* w = reinterpret_cast<const double2*>(tw);
#define CALL(LG) launch_xinv<LG>(in, pI, n1, n2, w, kxs, s)
  INS_POW2_SWITCH(n0, CALL)
#undef CALL
}

// The fused y + z passes (k_yz_*): partition count for a box, scratch size (complex numbers), and the solve itself on the spectrum after the x pass.
// P: the smallest power of two that gives every CU a workgroup (tiles of 8 kx) with partitions of at least 8 planes; 0 = not for this box.
// NOT the default (INS_YZ_FUSED=1 or a forced partition count select it): measured same-box the four-pass solve does not beat the five-pass one
// (profiles/r03_yz_fused_lab.txt; one solve through psolver(p): 256^3 428 against 363 us, 512^3 3391 against 3187 us).  What was found on the way:
//   * gfx9 counts loads and stores in one counter and only loads return in order, so a wait for a prefetched load behind an outstanding store is a wait
//     for everything: the next round's loads are now waited for before this round's stores are issued (512^3: 3982 -> 3391 us per solve);
//   * a spill reload inside the plane loop is a scratch load and drains the prefetch the same way (N = 512: one round of loads in flight instead of two);
//   * with the y-FFT left out (INS_YZ_SKEL=1) the marching passes run at 71 us (256^3) / 670 us (512^3) against 57 / 432 us of HBM time, and the
//     interface kernel costs 30 us at 256^3 (16 partitions): at 256^3 even a free FFT would lose (384 against 363 us).  The FFT phases (1.1-1.4 us per
//     radix-4 stage, VALU-bound on index arithmetic) do not overlap the memory phases: ONE 1024-work-item workgroup per CU marches in lock step, and a
//     box of this size has only tiles x partitions = 272 / 264 independent marches — more partitions cost 4 complex numbers per line each.
int ins_ownfft_yz_partitions(int kxn, int n1, int n2) {
  if (ins_opt(OPT_INS_DISABLE_YZ_FUSED)) return 0;
  if (!ins_opt(OPT_INS_YZ_FUSED) && !ins_opt(OPT_INS_YZ_PARTITIONS)) return 0;
  if (!(n1 == 128 || n1 == 256 || n1 == 512) || n2 < 64 || (n2 & (n2 - 1))) return 0;
  const int forced = (int)ins_opt(OPT_INS_YZ_PARTITIONS);
  if (forced == 2 || forced == 4 || forced == 8 || forced == 16) return (n2 % forced == 0 && (n2 / forced) % 4 == 0) ? forced : 0;
  const int tiles = (kxn + 7) / 8;
  int P = 2;
  while (P < 16 && tiles * P < 240 && n2 / (2 * P) >= 8) P *= 2;
  return tiles * P >= 128 ? P : 0;
}
long long ins_ownfft_yz_scratch(int kxn, int n1, int n2, int P) {
  const long long lines = (long long)n1 * kxn;
  return (long long)P * (2 * lines + n2 / P) + (long long)P * 2 * lines + n2;
}
int ins_k_ownfft_yz_solve(double* phat, int kxn, int n1, int n2, int kxs, int P, const double* ax, const double* ay, double c, double scale, const double* tw_y,
                          double* scratch, hipStream_t s) {
  YzArgs a;
  a.data = reinterpret_cast<double2*>(phat);
  a.kxn = kxn;
  a.kxs = kxs;
  a.n2 = n2;
  a.m = n2 / P;
  a.P = P;
  a.ax = ax;
  a.ay = ay;
  a.c = c;
  a.scale = scale;
  const long long lines = (long long)n1 * kxn;
  a.stride = 2 * lines + a.m;
  a.edges = reinterpret_cast<double2*>(scratch);
  a.bc = a.edges + (long long)P * a.stride;
  a.sol0 = a.bc + (long long)P * 2 * lines;
  a.skel = (int)ins_opt(OPT_INS_YZ_SKEL);
  const double2* w = reinterpret_cast<const double2*>(tw_y);
  switch (n1) {
    case 128: return launch_yz<7>(a, w, s);
    case 256: return launch_yz<8>(a, w, s);
    case 512: return launch_yz<9>(a, w, s);
  }
  ins_set_error("fused y/z passes: unsupported length %d", n1);
  return INS_ERR_UNSUPPORTED;
}

int ins_k_ownfft_y(double* phat, int kxn, int n1, int n2, const double* tw, bool inverse, hipStream_t s, int kxs) {
  if (kxs <= 0) kxs = kxn;
  double2* d = reinterpret_cast<double2*>(phat);
  const double2* w = reinterpret_cast<const double2*>(tw);
#define CALL(LG) launch_y<LG>(d, kxn, kxs, n2, w, inverse, nullptr, s)
  INS_POW2_SWITCH(n1, CALL)
#undef CALL
}

// ---- float2 spectra (the `_f32` family, ins_f32.hip): same passes, half the bytes; tw = float2 twiddles (ins_zsolve_twiddles_f32)
int ins_k_ownfft_xfwd_f32(const ins_grid* G, const float* src, int from_u, float* phat, int n0, int n1, int n2, const float* tw, hipStream_t s, int kxs) {
  static const GridDev no_grid{};
  const GridDev& g = G ? G->g : no_grid;
  float2* out = reinterpret_cast<float2*>(phat);
  const float2* w = reinterpret_cast<const float2*>(tw);
#define CALL(LG) launch_xfwd32<LG>(g, src, from_u, out, n1, n2, w, kxs, s)
  INS_POW2_SWITCH(n0, CALL)
#undef CALL
}
int ins_k_ownfft_xinv_f32(const float* phat, float* pI, int n0, int n1, int n2, const float* tw, hipStream_t s, int kxs) {
  const float2* in = reinterpret_cast<const float2*>(phat);
  const float2* w = reinterpret_cast<const float2*>(tw);
#define CALL(LG) launch_xinv<LG, float2>(in, pI, n1, n2, w, kxs, s)
  INS_POW2_SWITCH(n0, CALL)
#undef CALL
}
int ins_k_ownfft_y_f32(float* phat, int kxn, int n1, int n2, const float* tw, bool inverse, hipStream_t s, int kxs) {
  float2* d = reinterpret_cast<float2*>(phat);
  const float2* w = reinterpret_cast<const float2*>(tw);
#define CALL(LG) launch_y<LG, float2>(d, kxn, kxs, n2, w, inverse, nullptr, s)
  INS_POW2_SWITCH(n1, CALL)
#undef CALL
}

// y pass whose far side is the packed, kx-chunked transpose buffer (see PackMap): forward writes it, inverse reads it.
int ins_k_ownfft_y_packed(double* phat, double* packed, int kxn, int n1, int nzl, int nyl, int cw, const double* tw, bool inverse,
                          hipStream_t s) {
  double2* d = reinterpret_cast<double2*>(phat);
  const double2* w = reinterpret_cast<const double2*>(tw);
  PackMap pm{reinterpret_cast<double2*>(packed), nyl, nzl, cw};
#define CALL(LG) launch_y<LG>(d, kxn, kxn, nzl, w, inverse, &pm, s)
  INS_POW2_SWITCH(n1, CALL)
#undef CALL
}

// In-place on the real work array of the direct solver (n0 even, nz a power of two in 32..512); see k_fdm_z.  Returns the block count.
int ins_k_fdm_z(double* data, int n0, int n1, int nz, const double* lx, const double* ly, const double* lz, const double* ox, const double* oy,
                double h, double tol, int singular, const double* meanf, double* partial, const double* tw, int* nblk, hipStream_t s, const double* dct_w) {
  FdmZArgs a;
  a.wq = reinterpret_cast<const double2*>(dct_w);
  a.data = reinterpret_cast<double2*>(data);
  a.n0h = n0 / 2;
  a.nl = (long long)a.n0h * n1;
  a.lx = lx; a.ly = ly; a.lz = lz; a.ox = ox; a.oy = oy;
  a.h = h; a.tol = tol; a.singular = singular; a.meanf = meanf; a.partial = partial;
  constexpr int TK = 8;
  const unsigned nb = (unsigned)((a.nl + TK - 1) / TK);
  if (nblk) *nblk = (int)nb;
  const double2* w = reinterpret_cast<const double2*>(tw);
#define INS_FDMZ(LOG)                                                                                          \
  {                                                                                                             \
    const size_t lds = ((size_t)(1 << LOG) * TK + (1 << LOG)) * sizeof(double2);                                \
    if (dct_w) {                                                                                                \
      int rc = set_lds(&k_fdm_z<LOG, TK, true>, lds);                                                           \
      if (rc) return rc;                                                                                        \
      hipLaunchKernelGGL((k_fdm_z<LOG, TK, true>), dim3(nb), dim3(256), lds, s, a, w);                          \
    } else {                                                                                                    \
      int rc = set_lds(&k_fdm_z<LOG, TK>, lds);                                                                 \
      if (rc) return rc;                                                                                        \
      hipLaunchKernelGGL((k_fdm_z<LOG, TK>), dim3(nb), dim3(256), lds, s, a, w);                                \
    }                                                                                                           \
    break;                                                                                                      \
  }
  switch (nz) {
    case 32: INS_FDMZ(5)
    case 64: INS_FDMZ(6)
    case 128: INS_FDMZ(7)
    case 256: INS_FDMZ(8)
    case 512: INS_FDMZ(9)
    default: ins_set_error("ins_k_fdm_z: unsupported nz = %d", nz); return INS_ERR_UNSUPPORTED;
  }
#undef INS_FDMZ
  INS_LAUNCH_CHECK();
  return INS_OK;
}
