// Distributed z solve of the spectral Poisson solver WITHOUT transposes (pressure.jl:289-351 restated).
//
// After the (x, y) transforms of a z-slab the Poisson problem decouples into one periodic tridiagonal system per (kx, ky) line,
//     (âx + ây) p_k + c (2 p_k - p_{k+1} - p_{k-1}) = g_k,      c = Ω/Δz²,   g = -f̂ / (nx ny),   k = 0..nz-1 (periodic),
// whose circulant matrix is exactly what the reference's z-FFT diagonalises (eigenvalues âx + ây + 4c sin²(π kz/nz) = âx+ây+âz).
// The FFT route needs the whole line on one rank: two all-to-all transposes of the full half spectrum per solve (135 MB per rank
// and transpose at 256³ per rank — more than a stage's compute over one xGMI link).  Here each rank keeps its nzl planes and the
// lines are solved by the partition (SPIKE) method; ranks exchange TWO complex numbers per line (one small all-gather):
//
//   1. forward elimination of the local block A_r = tridiag(-c, d, -c) (d = âx+ây+2c) on g, in place, and in the same sweep the two
//      numbers yF = (A_r⁻¹ g)_first, yL = (A_r⁻¹ g)_last as dot products with the spikes A_r⁻¹ e_first, A_r⁻¹ e_last;
//   2. all-gather of (yF, yL) [host: torch.distributed]; every rank solves the 2P x 2P interface system per line — it is
//      block-circulant (all local blocks are equal), so a P-point DFT over ranks reduces it to P independent 2x2 systems;
//   3. back substitution with the two interface values (p_last of the rank below, p_first of the rank above) folded in.
// The blocks are symmetric Toeplitz, so pivots, spikes and the eliminated boundary columns have closed forms in the decaying
// root r of r + 1/r = d/c; nothing but the field itself is stored.  Two streaming passes (read + write each) replace the fused
// z-FFT pass of the single-GPU path.  The (0,0) line is singular (d = 2c): its nz values are gathered with the interface data and
// solved by two prefix sums with the reference's gauge (zero mean, pressure.jl:336-341).
#include <cmath>

#include "ins_internal.h"

namespace {

struct ZtriArgs {
  double2* data;       // [m][n1][kxs]  (work array after the x / y transforms)
  int kxn, kxs, n1, m; // live kx, row stride, rows, local planes
  int nranks, rank;
  const double* ax;    // [kxn]
  const double* ay;    // [n1]  (storage order of ky)
  double c, scale;     // Ω/Δz², -1/(nx ny)
};

struct LineConst {
  double lnr, r, r2, D, rm1;  // ln r, r, r², 1 - r^(2m+2), r^(m+1)
};

__device__ __forceinline__ LineConst line_const(double s, int m) {  // s = (âx + ây)/c = d/c - 2 > 0
  LineConst L;
  const double sq = sqrt(s * (4.0 + s));
  L.lnr = -log1p(0.5 * (s + sq));  // r = 2 / (s + 2 + sq): the decaying root, without cancellation
  L.r = exp(L.lnr);
  L.r2 = L.r * L.r;
  L.D = -expm1((2.0 * m + 2.0) * L.lnr);
  L.rm1 = exp((m + 1.0) * L.lnr);
  return L;
}

// reciprocal: hardware estimate + one Newton step (the quotient feeds a recurrence that is itself only accurate to rounding)
__device__ __forceinline__ double frcp(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  return y * (2.0 - x * y);
}

__device__ __forceinline__ double2 operator*(double a, double2 v) { return make_double2(a * v.x, a * v.y); }
__device__ __forceinline__ double2 operator+(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// ---- pass 1: forward elimination + spike dot products -------------------------------------------------------------------
// pivots 1/den_k = (r/c)(1 - E_k)/(1 - E_k r²), E_k = r^(2k+2);  gp_k = (g_k + c gp_{k-1}) / den_k
// spikes v_k = (r^(k+1) - r^(m+1) r^(m-k)) / (1 - r^(2m+2)) = c (A_r⁻¹ e_0)_k,  w_k = v_{m-1-k}
// One work-item per REAL component of a line (re and im obey the same real recurrence): twice the parallelism of one per line,
// which matters because a slab has only kxn·ny lines and every one is a serial march over the local planes.
template <bool SKEL>  // SKEL (INS_ZTRI_SKEL=1, measurement only): the same loads and stores without the recurrence
__global__ __launch_bounds__(256) void k_ztri_fwd(ZtriArgs a, double* __restrict__ edge, double* __restrict__ line0, long long l_lo, long long l_cnt) {
  // flat index over (ky, kx, re/im) of the line range of this launch: a 256-wide workgroup reads 2 KB of consecutive memory per plane
  const long long ps = 2LL * a.kxs * a.n1;
  if (l_lo == 0 && blockIdx.x == gridDim.x - 1) {
    // the extra workgroup of a range that starts at line 0 (kx = ky = 0, the singular line): hand its scaled right-hand side to the
    // gather with one work-item per plane.  (Left to the line's own two work-items this is a chain of m dependent HBM round trips
    // that the whole launch then waits for: 100 us instead of 60 at m = 256.)  The solve itself is ztri_line0 (in the k_ztri_bwd launch).
    const double* x0 = reinterpret_cast<const double*>(a.data);
    for (int k = threadIdx.x; k < a.m; k += 256) {
      line0[2 * k] = a.scale * x0[k * ps];
      line0[2 * k + 1] = a.scale * x0[k * ps + 1];
    }
    return;
  }
  const long long r2 = (long long)blockIdx.x * 256 + threadIdx.x;  // component index inside the range
  if (r2 >= 2 * l_cnt) return;
  const long long lines2 = 2 * l_cnt, l2 = 2 * l_lo + r2;
  const int ky = (int)(l2 / (2 * a.kxn)), t = (int)(l2 - (long long)ky * 2 * a.kxn);
  const int kx = t >> 1;
  double* x = reinterpret_cast<double*>(a.data) + t + 2LL * a.kxs * ky;
  const double sxy = a.ax[kx] + a.ay[ky];
  if (sxy == 0.0) {  // the singular line (only kx = ky = 0 has a zero eigenvalue: see the extra workgroup above)
    edge[r2] = edge[lines2 + r2] = 0.0;
    return;
  }
  const LineConst L = line_const(sxy / a.c, a.m);
  const double rc = L.r / a.c, idc = 1.0 / (L.D * a.c);
  double E = L.r2, pa = L.r;
  double gp = 0.0, SA = 0.0, SB = 0.0;  // Σ r^(k+1) g_k, Σ r^(m-k) g_k
  const double rinv = 1.0 / L.r;
  // 16-plane chunks, double-buffered in registers: the loads of chunk c+1 are issued before the stores of chunk c (the compiler
  // may not move a load above a store to the same array, and a load-compute-store chain per plane costs one HBM round trip each)
  constexpr int CH = 16;  // planes per chunk (32 measured no faster: the sweep is not limited by bytes in flight)
  double buf[CH], nxt[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) buf[j] = x[(long long)min(j, a.m - 1) * ps];
  for (int k0 = 0; k0 < a.m; k0 += CH) {
#pragma unroll
    for (int j = 0; j < CH; ++j) nxt[j] = x[(long long)min(k0 + CH + j, a.m - 1) * ps];
    // r^(m-k) by a running product re-seeded from exp() every chunk: r >= ~0.1 on these grids, so a seed that underflows
    // cannot become significant within a chunk (16 steps), and the product never has to climb out of an underflow
    double pb = exp((a.m - k0) * L.lnr);
    auto one = [&](int j) {
      const int k = k0 + j;
      const double g = a.scale * buf[j];
      if (SKEL) {
        x[k * ps] = g;
        return;
      }
      const double inv = rc * (1.0 - E) * frcp(1.0 - E * L.r2);
      gp = inv * (g + a.c * gp);
      x[k * ps] = gp;
      SA += pa * g;
      SB += pb * g;
      E *= L.r2;
      pa *= L.r;
      pb *= rinv;
    };
    if (k0 + CH <= a.m) {  // whole chunk: straight-line code (one scalar test instead of one per plane)
#pragma unroll
      for (int j = 0; j < CH; ++j) one(j);
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j)
        if (j < a.m - k0) one(j);  // predicated, not a rolled loop: buf[] must stay in registers
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) buf[j] = nxt[j];
  }
  edge[r2] = idc * (SA - L.rm1 * SB);           // Σ v_k g_k / c
  edge[lines2 + r2] = idc * (SB - L.rm1 * SA);  // Σ w_k g_k / c
}

// ---- interface: block-circulant 2P x 2P system per line, by a DFT over ranks ---------------------------------------------
//   F_r - α L_{r-1} - β F_{r+1} = yF_r,   L_r - β L_{r-1} - α F_{r+1} = yL_r,   α = v_0, β = v_{m-1}
// edges_all: [rank][ per-rank block of `stride` complex: yF[lines], yL[lines], line0[m] ];  out bc: [2][lines] = (L_{r-1}, F_{r+1})
constexpr int ZTRI_MAX_RANKS = 16;
// PT: the rank count as a compile-time constant (0 = any count up to ZTRI_MAX_RANKS).  With a run-time count the per-rank values live
// in scratch memory and every term of the O(P²) sums waits for a scratch load: 61 us instead of ~10 at P = 8, 131 k lines.
template <int PT>
__global__ __launch_bounds__(256) void k_ztri_iface(ZtriArgs a, const double2* __restrict__ edges_all, long long stride, double2* __restrict__ bc,
                                                    long long l_lo, long long l_cnt) {
  // e^{-2πi k/P}, k < P, once per workgroup (evaluated per (j, q) pair the P² sincospi calls were 14x the rest of this kernel at P = 8)
  __shared__ double2 tw[ZTRI_MAX_RANKS];
  const int P = PT ? PT : a.nranks;
  if ((int)threadIdx.x < P) {
    double sn, cs;
    sincospi(-2.0 * (double)threadIdx.x / P, &sn, &cs);
    tw[threadIdx.x] = make_double2(cs, sn);
  }
  __syncthreads();
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= l_cnt) return;
  const long long lines = l_cnt, l = r;  // edges_all / bc are indexed inside the range
  const long long lg = l_lo + r;
  const int ky = (int)(lg / a.kxn), kx = (int)(lg - (long long)ky * a.kxn);
  const double sxy = a.ax[kx] + a.ay[ky];
  if (sxy == 0.0) {
    bc[l] = bc[lines + l] = make_double2(0.0, 0.0);
    return;
  }
  const LineConst L = line_const(sxy / a.c, a.m);
  const double alpha = L.r * (-expm1(2.0 * a.m * L.lnr)) / L.D;
  const double beta = exp(a.m * L.lnr) * (1.0 - L.r2) / L.D;
  double2 yF[PT ? PT : ZTRI_MAX_RANKS], yL[PT ? PT : ZTRI_MAX_RANKS];
#pragma unroll PT ? PT : 1
  for (int q = 0; q < P; ++q) {
    yF[q] = edges_all[q * stride + l];
    yL[q] = edges_all[q * stride + lines + l];
  }
  double2 Lprev = make_double2(0.0, 0.0), Fnext = Lprev;
  const int rp = (a.rank + P - 1) % P, rn = (a.rank + 1) % P;
  auto conj = [](double2 z) { return make_double2(z.x, -z.y); };
#pragma unroll PT ? PT : 1
  for (int j = 0; j < P; ++j) {
    double2 hF = make_double2(0.0, 0.0), hL = hF;
#pragma unroll PT ? PT : 1
    for (int q = 0, jq = 0; q < P; ++q) {  // forward DFT over ranks: e^{-iθ_j q}
      const double2 w = tw[jq];
      hF = hF + cmul(w, yF[q]);
      hL = hL + cmul(w, yL[q]);
      jq += j;
      if (jq >= P) jq -= P;
    }
    const double2 em = tw[j], ep = conj(em);  // e^{-iθ}, e^{+iθ}
    const double cs = em.x;
    // [ 1 - β e^{+iθ}    -α e^{-iθ} ] [F̂]   [ŷF]
    // [   -α e^{+iθ}   1 - β e^{-iθ} ] [L̂] = [ŷL]
    const double2 a11 = make_double2(1.0 - beta * ep.x, -beta * ep.y), a22 = make_double2(1.0 - beta * em.x, -beta * em.y);
    const double2 a12 = (-alpha) * em, a21 = (-alpha) * ep;
    const double det = 1.0 - 2.0 * beta * cs + beta * beta - alpha * alpha;  // a11 a22 - a12 a21 (real)
    const double2 Fh = (1.0 / det) * (cmul(a22, hF) + (-1.0) * cmul(a12, hL));
    const double2 Lh = (1.0 / det) * (cmul(a11, hL) + (-1.0) * cmul(a21, hF));
    Lprev = Lprev + cmul(conj(tw[(j * rp) % P]), Lh);  // inverse DFT at ranks r-1 and r+1: e^{+iθ_j r'}
    Fnext = Fnext + cmul(conj(tw[(j * rn) % P]), Fh);
  }
  bc[l] = (1.0 / P) * Lprev;
  bc[lines + l] = (1.0 / P) * Fnext;
}

__device__ void ztri_line0(const ZtriArgs& a, const double2* __restrict__ edges_all, long long stride, long long line0_off, int lane);

// ---- pass 2: back substitution with the interface values folded in -------------------------------------------------------
// g̃ = g + c L_{r-1} e_0 + c F_{r+1} e_{m-1}  =>  gp̃_k = gp_k + c L φ_k (+ c F /den_{m-1} at k = m-1),  φ_k = r^(k+1)(1-r²)/(c(1-E_k r²))
// p_{m-1} = gp̃_{m-1},  p_k = gp̃_k + (c/den_k) p_{k+1}
__global__ __launch_bounds__(256) void k_ztri_bwd(ZtriArgs a, const double* __restrict__ bc, const double2* __restrict__ edges_all, long long stride,
                                                  long long l_lo, long long l_cnt) {
  if (l_lo == 0 && blockIdx.x == gridDim.x - 1) {  // extra workgroup of the range that holds line 0: the singular line's solve
    if (threadIdx.x < 64) ztri_line0(a, edges_all, stride, 2 * l_cnt, threadIdx.x);
    return;
  }
  // flat index over (ky, kx, re/im) of the line range of this launch: a 256-wide workgroup reads 2 KB of consecutive memory per plane
  const long long r2 = (long long)blockIdx.x * 256 + threadIdx.x;  // component index inside the range
  if (r2 >= 2 * l_cnt) return;
  const long long lines2 = 2 * l_cnt, l2 = 2 * l_lo + r2;
  const int ky = (int)(l2 / (2 * a.kxn)), t = (int)(l2 - (long long)ky * 2 * a.kxn);
  const int kx = t >> 1;
  const long long ps = 2LL * a.kxs * a.n1;
  double* x = reinterpret_cast<double*>(a.data) + t + 2LL * a.kxs * ky;
  const double sxy = a.ax[kx] + a.ay[ky];
  if (sxy == 0.0) return;  // ztri_line0
  const LineConst L = line_const(sxy / a.c, a.m);
  const double Lp = bc[r2], Fn = bc[lines2 + r2];  // bc of this range: [L (cnt)][F (cnt)]
  const double omr2 = 1.0 - L.r2;
  double p = 0.0;
  const double rinv = 1.0 / L.r;
  constexpr int CH = 16;  // double-buffered chunks, see k_ztri_fwd
  double buf[CH], nxt[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) buf[j] = x[(long long)max(a.m - 1 - j, 0) * ps];
  for (int k0 = a.m - 1; k0 >= 0; k0 -= CH) {
#pragma unroll
    for (int j = 0; j < CH; ++j) nxt[j] = x[(long long)max(k0 - CH - j, 0) * ps];
    double pa = exp((k0 + 1.0) * L.lnr);  // r^(k+1), running product re-seeded every chunk
    auto one = [&](int j) {
      const int k = k0 - j;
      const double q = frcp(1.0 - pa * pa * L.r2);
      const double cinv = L.r * (1.0 - pa * pa) * q;  // c / den_k
      double v = buf[j] + (pa * omr2 * q) * Lp;        // + c L φ_k
      if (k == a.m - 1) v += cinv * Fn;
      p = v + cinv * p;
      x[k * ps] = p;
      pa *= rinv;
    };
    if (k0 - CH >= -1) {
#pragma unroll
      for (int j = 0; j < CH; ++j) one(j);
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j)
        if (j <= k0) one(j);
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) buf[j] = nxt[j];
  }
}

// ---- the singular line: -c (p_{k+1} - 2 p_k + p_{k-1}) = g_k - mean(g), periodic over N = P m, mean(p) = 0 -----------------
// q_k = p_{k+1} - p_k = q_0 - S_k,  S_k = Σ_{j=1..k} h_j,  h = (g - ḡ)/c;   p_k = p_0 + k q_0 - T_k,  T_k = Σ_{j<k} S_j;
// periodicity: q_0 = T_N / N;  gauge: p_0 = -(1/N) Σ_k (k q_0 - T_k).   One wavefront, chunks per lane, shuffle scans.
__device__ __forceinline__ double2 wave_sum(double2 v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v.x += __shfl_xor(v.x, off, 64);
    v.y += __shfl_xor(v.y, off, 64);
  }
  return v;
}
__device__ __forceinline__ double2 wave_excl_scan(double2 v, int lane) {  // exclusive prefix sum over lanes
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
}

// (one wavefront: run by the extra workgroup of k_ztri_bwd, beside the other lines' back substitution)
__device__ void ztri_line0(const ZtriArgs& a, const double2* __restrict__ edges_all, long long stride, long long line0_off, int lane) {
  const int P = a.nranks, m = a.m, N = P * m;
  const int C = (N + 63) / 64, s = min(lane * C, N), e = min(s + C, N);
  auto g_at = [&](int k) { return edges_all[(long long)(k / m) * stride + line0_off + (k % m)]; };
  const double ic = 1.0 / a.c, iN = 1.0 / N;
  double2 sum = make_double2(0.0, 0.0);
  for (int k = s; k < e; ++k) sum = sum + g_at(k);
  const double2 gbar = iN * wave_sum(sum);
  auto h_at = [&](int k) {
    const double2 g = g_at(k);
    return make_double2((g.x - gbar.x) * ic, (g.y - gbar.y) * ic);
  };
  const double2 h0 = h_at(0);
  // chunk sums of h -> H base;  S_j = H_j - h_0 (inclusive prefix minus the j = 0 term)
  double2 A = make_double2(0.0, 0.0);
  for (int k = s; k < e; ++k) A = A + h_at(k);
  const double2 Hbase = wave_excl_scan(A, lane);
  double2 B = make_double2(0.0, 0.0), H = Hbase;
  for (int k = s; k < e; ++k) {
    H = H + h_at(k);
    B = B + make_double2(H.x - h0.x, H.y - h0.y);
  }
  const double2 Tbase = wave_excl_scan(B, lane);  // T_s = Σ_{j<s} S_j
  const double2 q0 = iN * wave_sum(B);            // T_N / N
  double2 Cs = make_double2(0.0, 0.0), T = Tbase;
  H = Hbase;
  for (int k = s; k < e; ++k) {
    Cs = Cs + make_double2(k * q0.x - T.x, k * q0.y - T.y);
    H = H + h_at(k);
    T = T + make_double2(H.x - h0.x, H.y - h0.y);
  }
  const double2 p0 = (-iN) * wave_sum(Cs);
  const long long ps = (long long)a.kxs * a.n1;
  const int k0 = a.rank * m;
  T = Tbase;
  H = Hbase;
  for (int k = s; k < e; ++k) {
    if (k >= k0 && k < k0 + m) a.data[(long long)(k - k0) * ps] = make_double2(p0.x + k * q0.x - T.x, p0.y + k * q0.y - T.y);
    H = H + h_at(k);
    T = T + make_double2(H.x - h0.x, H.y - h0.y);
  }
}

}  // namespace

// launched by ins_slab.hip (which owns the slab handle): all on stream s
// Line range [l_lo, l_lo + l_cnt) of the kxn·n1 lines.  edge: [yF (l_cnt) | yL (l_cnt) | the singular line's m values when the range holds line 0]
int ins_k_ztri_forward(double* work, int kxn, int kxs, int n1, int m, int nranks, int rank, const double* ax, const double* ay, double c,
                       double scale, double* edge, long long l_lo, long long l_cnt, hipStream_t s) {
  ZtriArgs a{reinterpret_cast<double2*>(work), kxn, kxs, n1, m, nranks, rank, ax, ay, c, scale};
  if (l_cnt <= 0) return INS_OK;
  const int extra = l_lo == 0 ? 1 : 0;  // one more workgroup for the singular line's right-hand side
  const bool skel = ins_opt(OPT_INS_ZTRI_SKEL) != 0;
  if (skel)
    hipLaunchKernelGGL(k_ztri_fwd<true>, dim3(cdiv(2 * l_cnt, 256) + extra), dim3(256), 0, s, a, edge, edge + 4 * l_cnt, l_lo, l_cnt);
  else
    hipLaunchKernelGGL(k_ztri_fwd<false>, dim3(cdiv(2 * l_cnt, 256) + extra), dim3(256), 0, s, a, edge, edge + 4 * l_cnt, l_lo, l_cnt);  // one work-item per real component
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// edges_all: the gathered edge buffers of this range, `stride` complex per rank; bc: scratch of 2·l_cnt complex
int ins_k_ztri_finish(double* work, int kxn, int kxs, int n1, int m, int nranks, int rank, const double* ax, const double* ay, double c,
                      const double* edges_all, long long stride, double* bc, long long l_lo, long long l_cnt, hipStream_t s) {
  if (nranks > ZTRI_MAX_RANKS) {
    ins_set_error("tridiagonal z solve: at most %d ranks", ZTRI_MAX_RANKS);
    return INS_ERR_UNSUPPORTED;
  }
  if (l_cnt <= 0) return INS_OK;
  ZtriArgs a{reinterpret_cast<double2*>(work), kxn, kxs, n1, m, nranks, rank, ax, ay, c, 0.0};
  const double2* ea = reinterpret_cast<const double2*>(edges_all);
  double2* b = reinterpret_cast<double2*>(bc);
#define INS_IFACE(PT) hipLaunchKernelGGL(k_ztri_iface<PT>, dim3(cdiv(l_cnt, 256)), dim3(256), 0, s, a, ea, stride, b, l_lo, l_cnt)
  switch (nranks) {
    case 1: INS_IFACE(1); break;
    case 2: INS_IFACE(2); break;
    case 4: INS_IFACE(4); break;
    case 8: INS_IFACE(8); break;
    default: INS_IFACE(0); break;
  }
#undef INS_IFACE
  hipLaunchKernelGGL(k_ztri_bwd, dim3(cdiv(2 * l_cnt, 256) + (l_lo == 0 ? 1 : 0)), dim3(256), 0, s, a, (const double*)bc, ea, stride, l_lo, l_cnt);
  INS_LAUNCH_CHECK();
  return INS_OK;
}
