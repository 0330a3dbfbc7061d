// psolver_direct for tensor-product grids: fast diagonalisation (pressure.jl:101-154 restated).
//
// The reference factorises `laplacian_mat = P'ΩMBuGBpP` (matrices.jl:484-492) with SuiteSparse (CPU) or cuDSS.  On this
// package's grids (Cartesian product of 1-D grids, any mix of Periodic / Dirichlet / Symmetric / Pressure sides) the same matrix
// separates,   L = Tx⊗Dy⊗Dz + Dx⊗Ty⊗Dz + Dx⊗Dy⊗Tz,   Dα = diag(Δα[Ip]),  Tα = the 1-D second-difference matrix that
// `laplacian!` applies along α including its boundary branches (operators.jl:328-350).  With the generalised eigenpairs
// Tα Vα = Dα Vα Λα (VαᵀDαVα = I, computed once on the host),  L⁻¹ f = (Vx⊗Vy⊗Vz) · (Vx⊗Vy⊗Vz)ᵀ f / (λx+λy+λz):
// six fp64 GEMMs (rocBLAS, MFMA f64) and one scaling pass per solve — a DIRECT solve of the same linear system.  Singular case
// (no Pressure side): the reference factorises the bordered system [L e; e' 0][p; λ] = [f; 0] (pressure.jl:133-140), whose solution
// is L p = f - mean(f) e with e'p = 0.  Here: subtract mean(f), drop the (now empty) null mode, shift p to zero mean over Ip.
#include <rocblas/rocblas.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "ins_internal.h"

struct ins_fdm {
  rocblas_handle h = nullptr;
  int D = 3;
  int n[3] = {1, 1, 1};
  double* V[3] = {nullptr, nullptr, nullptr};    // n[a] x n[a], column-major, Dα-orthonormal eigenvectors
  double* lam[3] = {nullptr, nullptr, nullptr};  // eigenvalues (<= 0)
  double *a = nullptr, *b = nullptr;             // two n0*n1*n2 work arrays
  double* sums = nullptr;                        // [0..4095] block partials of Σ p, [4096] mean(f), [4097] mean(p)
  double* ones[3] = {nullptr, nullptr, nullptr};  // Vαᵀ 1: (Vx⊗Vy⊗Vz)ᵀ 1 = ones_x ⊗ ones_y ⊗ ones_z
  long long null_index = 0;                      // storage index of the null mode (λx = λy = λz = 0)
  double null_scale = 1.0;                       // V₀ = null_scale · 1  =>  Σ f = q[null] / null_scale
  bool singular = true;
  double lam_tol = 0.0;
  int null_i[3] = {0, 0, 0};            // null mode of each direction (storage index) ...
  double null_s[3] = {1.0, 1.0, 1.0};   // ... and the constant its eigenvector equals
  // periodic uniform z: Fourier modes instead of the dense Vz (ins_fdm_enable_zfft)
  bool zfft = false;
  bool zdct = false;        // uniform z between two walls: cosine modes (ins_fdm_enable_zdct), same fused pass in its DCT form
  double* zwq = nullptr;    // e^{-iπk/2n2}, k = 0..n2/2
  double hz = 0.0;
  double* lzk = nullptr;    // λz(k), k = 0..n2/2
  double* ztw = nullptr;    // twiddles
  double* zpart = nullptr;  // block partials of the fused z kernel
  // periodic uniform x as well (channel flows): real-to-complex x passes (ins_fft.hip) instead of the dense Vx
  bool xfft = false;
  int kxs = 0;              // complex row stride of the half spectrum (n0/2+1 rounded up to 8)
  double hx = 0.0;
  double* lxd = nullptr;    // λx per REAL slot of a spectrum row (re and im of a mode share it), 2 kxs
  double* oxd = nullptr;    // Vxᵀ1 in that layout: sqrt(n0/hx) at the real part of kx = 0
  double* xtw = nullptr;
  // periodic uniform x AND y with any z (walls in z: the orientation of TurbulentChannel.jl): x and y through the own FFT passes, z through Vz
  bool xyfft = false;
  double hy = 0.0;
  double* lyp = nullptr;    // λy in the digit-reversed order the own y pass leaves behind
  double* oyp = nullptr;    // Vyᵀ1 in that layout: sqrt(n1/hy) at position 0
  double* ytw = nullptr;
  // Symmetric directions (cosine / tanh / uniform walls: Tα and Dα commute with the reflection i -> n-1-i): every eigenvector is even or odd, so
  // with the modes ordered [even | odd] and the data folded into (f_i + f_{n-1-i} | f_i - f_{n-1-i}) a transform along that direction is two
  // GEMMs of half the size — half the flops of the dense transform.  Vh[a]: the top halves of the reordered eigenvectors, (n/2) x n, column-major.
  bool fold[3] = {false, false, false};
  double* Vh[3] = {nullptr, nullptr, nullptr};
};

namespace {

// out of place (in place the work-item of i would overwrite position n/2 + i, which the work-item of n/2 - 1 - i still has to read):
// forward  (f_i, f_{n-1-i}) -> (f_i + f_{n-1-i} at i, f_i - f_{n-1-i} at n/2 + i), i < n/2, along every direction flagged in `mask`;
// inverse  (e_i, o_i) -> (e_i + o_i at i, e_i - o_i at n-1-i).  One work-item per group of 2^k mirror images.
__global__ __launch_bounds__(256) void k_fdm_fold(const double* __restrict__ x, double* __restrict__ y, int n0, int n1, int n2, int mask, int inverse) {
  const int h0 = (mask & 1) ? n0 / 2 : n0, h1 = (mask & 2) ? n1 / 2 : n1, h2 = (mask & 4) ? n2 / 2 : n2;
  const long long total = (long long)h0 * h1 * h2;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int i = (int)(t % h0);
    const long long r = t / h0;
    const int j = (int)(r % h1), k = (int)(r / h1);
    // the two storage positions along each folded direction: natural layout (i, n-1-i), folded layout (i, n/2 + i)
    const int ia[2] = {i, (mask & 1) ? (inverse ? n0 / 2 + i : n0 - 1 - i) : i}, ib[2] = {i, (mask & 1) ? (inverse ? n0 - 1 - i : n0 / 2 + i) : i};
    const int ja[2] = {j, (mask & 2) ? (inverse ? n1 / 2 + j : n1 - 1 - j) : j}, jb[2] = {j, (mask & 2) ? (inverse ? n1 - 1 - j : n1 / 2 + j) : j};
    const int ka[2] = {k, (mask & 4) ? (inverse ? n2 / 2 + k : n2 - 1 - k) : k}, kb[2] = {k, (mask & 4) ? (inverse ? n2 - 1 - k : n2 / 2 + k) : k};
    const int c0 = (mask & 1) ? 2 : 1, c1 = (mask & 2) ? 2 : 1, c2 = (mask & 4) ? 2 : 1;
    double v[2][2][2];
    for (int c = 0; c < c2; ++c)
      for (int b = 0; b < c1; ++b)
        for (int a = 0; a < c0; ++a) v[c][b][a] = x[ia[a] + (long long)n0 * (ja[b] + (long long)n1 * ka[c])];
    if (mask & 1)
      for (int c = 0; c < c2; ++c)
        for (int b = 0; b < c1; ++b) {
          const double p = v[c][b][0], q = v[c][b][1];
          v[c][b][0] = p + q;
          v[c][b][1] = p - q;
        }
    if (mask & 2)
      for (int c = 0; c < c2; ++c)
        for (int a = 0; a < c0; ++a) {
          const double p = v[c][0][a], q = v[c][1][a];
          v[c][0][a] = p + q;
          v[c][1][a] = p - q;
        }
    if (mask & 4)
      for (int b = 0; b < c1; ++b)
        for (int a = 0; a < c0; ++a) {
          const double p = v[0][b][a], q = v[1][b][a];
          v[0][b][a] = p + q;
          v[1][b][a] = p - q;
        }
    for (int c = 0; c < c2; ++c)
      for (int b = 0; b < c1; ++b)
        for (int a = 0; a < c0; ++a) y[ib[a] + (long long)n0 * (jb[b] + (long long)n1 * kb[c])] = v[c][b][a];
  }
}


// Singular systems (no PressureBC side) are handled in eigen-space, where both gauges are rank-one:
//   mean(f) removal:  f - m·1  <=>  q - m·(Vᵀ1),  Vᵀ1 = ox ⊗ oy ⊗ oz,  and  Σ f = q[null] / null_scale  (V₀ is the constant vector);
//   mean(p) = (Vᵀ1)ᵀ q' / n, accumulated while q' is written — so no pass over f or p is spent on either.
__global__ void k_fdm_null(const double* __restrict__ q, long long null_index, double inv, double* __restrict__ sums) {
  sums[4096] = q[null_index] * inv;  // mean(f)
}

// q' = (q - mean(f)·ox oy oz) / (λx + λy + λz), null mode -> 0; block partials of Σ ox oy oz q'
__global__ __launch_bounds__(256) void k_fdm_scale(double* __restrict__ q, const double* __restrict__ lx, const double* __restrict__ ly,
                                                   const double* __restrict__ lz, const double* __restrict__ ox, const double* __restrict__ oy,
                                                   const double* __restrict__ oz, int n0, int n1, int n2, double tol, int singular,
                                                   double* __restrict__ sums) {
  __shared__ double lds[4];
  const long long total = (long long)n0 * n1 * n2;
  const double mf = singular ? sums[4096] : 0.0;
  double acc = 0.0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int i = (int)(t % n0);
    const long long r = t / n0;
    const int j = (int)(r % n1), k = (int)(r / n1);
    double lam = lx[i] + ly[j];
    if (lz) lam += lz[k];
    double v = q[t];
    if (singular) {
      double o = ox[i] * oy[j];
      if (oz) o *= oz[k];
      v = (fabs(lam) <= tol) ? 0.0 : (v - mf * o) / lam;
      acc += o * v;
    } else {
      v /= lam;
    }
    q[t] = v;
  }
  if (singular) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
  }
}

// mean(f) from the (x, y) null line after the x and y transforms: Σ f = Σ_z r[ix0, iy0, z] / (sx sy)
__global__ __launch_bounds__(64) void k_fdm_null_z(const double* __restrict__ r, long long idx_xy, long long n01, int n2, double inv,
                                                   double* __restrict__ sums) {
  double acc = 0.0;
  for (int z = threadIdx.x; z < n2; z += 64) acc += r[idx_xy + (long long)z * n01];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) sums[4096] = acc * inv;
}

// mean(p) from the block partials of the fused z kernel
__global__ __launch_bounds__(256) void k_fdm_mean_z(const double* __restrict__ partial, int nblk, double inv_n, double* __restrict__ sums) {
  __shared__ double lds[4];
  double acc = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) acc += partial[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sums[4097] = (lds[0] + lds[1] + lds[2] + lds[3]) * inv_n;
}

__global__ __launch_bounds__(256) void k_fdm_mean(double* __restrict__ sums, int nblk, double inv_n) {
  __shared__ double lds[4];
  double acc = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) acc += sums[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sums[4097] = (lds[0] + lds[1] + lds[2] + lds[3]) * inv_n;  // mean(p), subtracted by the consumer
}

#define INS_BLAS_TRY(expr)                                                                     \
  do {                                                                                         \
    rocblas_status _s = (expr);                                                                \
    if (_s != rocblas_status_success) {                                                        \
      ins_set_error("%s:%d: %s -> rocblas_status %d", __FILE__, __LINE__, #expr, (int)_s);     \
      return INS_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

}  // namespace

int ins_fdm_destroy(ins_fdm* F) {
  if (!F) return INS_OK;
  if (F->h) (void)rocblas_destroy_handle(F->h);
  for (int a = 0; a < 3; ++a) {
    if (F->V[a]) (void)hipFree(F->V[a]);
    if (F->lam[a]) (void)hipFree(F->lam[a]);
    if (F->ones[a]) (void)hipFree(F->ones[a]);
    if (F->Vh[a]) (void)hipFree(F->Vh[a]);
  }
  if (F->a) (void)hipFree(F->a);
  if (F->b) (void)hipFree(F->b);
  if (F->sums) (void)hipFree(F->sums);
  if (F->lzk) (void)hipFree(F->lzk);
  if (F->ztw) (void)hipFree(F->ztw);
  if (F->zwq) (void)hipFree(F->zwq);
  if (F->zpart) (void)hipFree(F->zpart);
  if (F->lxd) (void)hipFree(F->lxd);
  if (F->oxd) (void)hipFree(F->oxd);
  if (F->xtw) (void)hipFree(F->xtw);
  if (F->lyp) (void)hipFree(F->lyp);
  if (F->oyp) (void)hipFree(F->oyp);
  if (F->ytw) (void)hipFree(F->ytw);
  delete F;
  return INS_OK;
}

// V[a]: host n[a] x n[a] column-major; lam[a]: host n[a]
int ins_fdm_create(int D, const int n[3], const double* const V[3], const double* const lam[3], int singular, ins_fdm** out) {
  ins_fdm* F = new ins_fdm();
  F->D = D;
  F->singular = singular != 0;
  long long total = 1;
  double lmax = 0.0;
  bool ok = rocblas_create_handle(&F->h) == rocblas_status_success;
  for (int a = 0; ok && a < D; ++a) {
    F->n[a] = n[a];
    total *= n[a];
    // Symmetric direction?  Every eigenvector even or odd under i -> n-1-i (to 1e-9 of its norm), as many even as odd ones (n even).
    // Then the modes are reordered [even | odd] — eigenvalues, Vαᵀ1 and the null mode follow — and the top halves are kept for the half GEMMs.
    const int na = n[a], hh = na / 2;
    std::vector<int> perm(na);
    for (int j = 0; j < na; ++j) perm[j] = j;
    bool sym = na % 2 == 0 && na >= 8 && !ins_opt(OPT_INS_DISABLE_FDM_FOLD);
    if (sym) {
      std::vector<int> ev, od;
      for (int j = 0; sym && j < na; ++j) {
        const double* v = V[a] + (size_t)na * j;
        double se = 0.0, so = 0.0, nn = 0.0;
        for (int i = 0; i < na; ++i) {
          se += (v[i] - v[na - 1 - i]) * (v[i] - v[na - 1 - i]);
          so += (v[i] + v[na - 1 - i]) * (v[i] + v[na - 1 - i]);
          nn += v[i] * v[i];
        }
        if (se <= 1e-18 * nn)
          ev.push_back(j);
        else if (so <= 1e-18 * nn)
          od.push_back(j);
        else
          sym = false;
      }
      sym = sym && (int)ev.size() == hh && (int)od.size() == hh;
      if (sym) {
        for (int j = 0; j < hh; ++j) {
          perm[j] = ev[j];
          perm[hh + j] = od[j];
        }
      }
    }
    F->fold[a] = sym;
    std::vector<double> Vs((size_t)na * na), lam_s(na);
    for (int j = 0; j < na; ++j) {
      lam_s[j] = lam[a][perm[j]];
      for (int i = 0; i < na; ++i) Vs[i + (size_t)na * j] = V[a][i + (size_t)na * perm[j]];
    }
    const double* Va = Vs.data();
    const double* la = lam_s.data();
    // Vαᵀ 1 and the null mode of this direction (largest eigenvalue, ~0 when the system is singular)
    std::vector<double> o(n[a], 0.0);
    int inull = 0;
    for (int j = 0; j < n[a]; ++j) {
      for (int i = 0; i < n[a]; ++i) o[j] += Va[i + (size_t)n[a] * j];
      if (std::fabs(la[j]) < std::fabs(la[inull])) inull = j;
    }
    // Singular system: every direction's factor has the constant vector in its null space, so the eigenvalue of that mode IS zero; the
    // eigensolver returns it to ~eps·λmax.  It is stored as an exact 0, so that the scaling kernels recognise the ONE null mode of the
    // Kronecker sum by λx + λy + λz == 0 and nothing else: a magnitude threshold would have to sit between that noise (eps·λmax, λmax ~
    // 4/h_min²) and the lowest physical mode (~π²/L²), and on strongly stretched grids (cosine grid, N >= 1024) there is no such gap.
    std::vector<double> lam_up(la, la + n[a]);
    if (F->singular) lam_up[inull] = 0.0;
    ok = hipMalloc(&F->V[a], (size_t)n[a] * n[a] * 8) == hipSuccess && hipMalloc(&F->lam[a], (size_t)n[a] * 8) == hipSuccess &&
         hipMemcpy(F->V[a], Va, (size_t)n[a] * n[a] * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(F->lam[a], lam_up.data(), (size_t)n[a] * 8, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && sym) {
      std::vector<double> top((size_t)hh * na);
      for (int j = 0; j < na; ++j)
        for (int i = 0; i < hh; ++i) top[i + (size_t)hh * j] = Va[i + (size_t)na * j];
      ok = hipMalloc(&F->Vh[a], top.size() * 8) == hipSuccess && hipMemcpy(F->Vh[a], top.data(), top.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    }
    for (int i = 0; i < n[a]; ++i) lmax = std::fmax(lmax, std::fabs(lam[a][i]));
    F->null_index += (long long)inull * (a == 0 ? 1 : (a == 1 ? n[0] : (long long)n[0] * n[1]));
    F->null_scale *= Va[(size_t)n[a] * inull];
    F->null_i[a] = inull;
    F->null_s[a] = Va[(size_t)n[a] * inull];
    ok = ok && hipMalloc(&F->ones[a], (size_t)n[a] * 8) == hipSuccess &&
         hipMemcpy(F->ones[a], o.data(), (size_t)n[a] * 8, hipMemcpyHostToDevice) == hipSuccess;
  }
  ok = ok && hipMalloc(&F->a, total * 8) == hipSuccess && hipMalloc(&F->b, total * 8) == hipSuccess && hipMalloc(&F->sums, 4098 * 8) == hipSuccess &&
       hipMemset(F->sums, 0, 4098 * 8) == hipSuccess;
  if (!ok) {
    ins_set_error("ins_fdm_create: allocation / upload failed");
    ins_fdm_destroy(F);
    return INS_ERR_HIP;
  }
  (void)lmax;
  F->lam_tol = 0.0;  // only the exact null mode (λx = λy = λz = 0 as stored above / in the analytic Fourier tables) is dropped
  *out = F;
  return INS_OK;
}

// Periodic uniform z direction with spacing hz: use Fourier modes for it (one fused pass, ins_fft.hip k_fdm_z) instead of the dense Vz
// (two GEMMs + the scaling pass).  Taken only when the host's eigenvalues of the z factor are the analytic set -(4/hz²) sin²(πk/n) —
// i.e. when the factor really is the periodic second difference on a uniform grid; otherwise the GEMM route stays.
int ins_fdm_enable_zfft(ins_fdm* F, double hz, const double* lam_z_host) {
  const bool off = ins_opt(OPT_INS_DISABLE_FDM_ZFFT) != 0;  // A/B switch
  const int n2 = F->n[2];
  if (off || F->D != 3 || (F->n[0] & 1) || n2 < 32 || n2 > 512 || (n2 & (n2 - 1))) return INS_OK;
  std::vector<double> want(n2), have(lam_z_host, lam_z_host + n2), lzk(n2 / 2 + 1);
  for (int k = 0; k < n2; ++k) {
    const double sn = std::sin(M_PI * (double)k / n2);
    want[k] = -4.0 * sn * sn / (hz * hz);
    if (k <= n2 / 2) lzk[k] = want[k];
  }
  std::sort(want.begin(), want.end());
  std::sort(have.begin(), have.end());
  const double scale = 4.0 / (hz * hz);
  for (int k = 0; k < n2; ++k)
    if (std::fabs(want[k] - have[k]) > 1e-9 * scale) return INS_OK;
  const int nblk = (int)(((long long)(F->n[0] / 2) * F->n[1] + 7) / 8);
  if (hipMalloc(&F->lzk, lzk.size() * 8) != hipSuccess || hipMemcpy(F->lzk, lzk.data(), lzk.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(&F->zpart, (size_t)nblk * 8) != hipSuccess || hipMemset(F->zpart, 0, (size_t)nblk * 8) != hipSuccess) {
    ins_set_error("ins_fdm_enable_zfft: allocation failed");
    return INS_ERR_HIP;
  }
  int rc = ins_zsolve_twiddles(n2, &F->ztw);
  if (rc) return rc;
  F->hz = hz;
  F->zfft = true;
  F->fold[2] = false;
  return INS_OK;
}

// Uniform z between two walls (the pressure problem has Neumann ends there: Dirichlet or Symmetric velocity sides): the eigenvectors of the
// factor are cos(π(2i+1)k/2n), eigenvalues -(4/hz²) sin²(πk/2n) — taken, as above, only when the host's eigenvalues are that set.  The fused
// pass (k_fdm_z<.., DCT>) replaces the two z GEMMs and the scaling pass.
int ins_fdm_enable_zdct(ins_fdm* F, double hz, const double* lam_z_host) {
  const int n2 = F->n[2];
  if (ins_opt(OPT_INS_DISABLE_FDM_ZDCT) || F->zfft || F->xfft || F->xyfft || F->D != 3 || (F->n[0] & 1) || n2 < 32 || n2 > 512 || (n2 & (n2 - 1))) return INS_OK;
  std::vector<double> want(n2), have(lam_z_host, lam_z_host + n2), lzk(n2), wq(2 * (n2 / 2 + 1));
  for (int k = 0; k < n2; ++k) {
    const double sn = std::sin(M_PI * (double)k / (2.0 * n2));
    want[k] = lzk[k] = -4.0 * sn * sn / (hz * hz);
    if (k <= n2 / 2) {
      wq[2 * k] = std::cos(M_PI * (double)k / (2.0 * n2));
      wq[2 * k + 1] = -std::sin(M_PI * (double)k / (2.0 * n2));
    }
  }
  std::sort(want.begin(), want.end());
  std::sort(have.begin(), have.end());
  const double scale = 4.0 / (hz * hz);
  for (int k = 0; k < n2; ++k)
    if (std::fabs(want[k] - have[k]) > 1e-9 * scale) return INS_OK;
  const int nblk = (int)(((long long)(F->n[0] / 2) * F->n[1] + 7) / 8);
  if (hipMalloc(&F->lzk, lzk.size() * 8) != hipSuccess || hipMemcpy(F->lzk, lzk.data(), lzk.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(&F->zwq, wq.size() * 8) != hipSuccess || hipMemcpy(F->zwq, wq.data(), wq.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc(&F->zpart, (size_t)nblk * 8) != hipSuccess || hipMemset(F->zpart, 0, (size_t)nblk * 8) != hipSuccess) {
    ins_set_error("ins_fdm_enable_zdct: allocation failed");
    return INS_ERR_HIP;
  }
  int rc = ins_zsolve_twiddles(n2, &F->ztw);
  if (rc) return rc;
  F->hz = hz;
  F->zdct = true;
  F->fold[2] = false;
  return INS_OK;
}

// Periodic uniform x on top of a Fourier z (channel flows): the x eigenvectors are Fourier modes too.  The real-to-complex x pass turns a
// row into n0/2+1 complex modes (padded to kxs); the y GEMMs and the fused z pass are real-linear and act on the interleaved (re, im) slots as
// on any other real column — both slots of a mode carry the same λx, so the z pass's per-column scaling is simply the complex scaling.
// The 1/sqrt(hx n0) of the D-orthonormal convention rides on the two y GEMMs (alpha).
int ins_fdm_enable_xfft(ins_fdm* F, double hx, const double* lam_x_host) {
  const bool off = ins_opt(OPT_INS_DISABLE_FDM_XFFT) != 0;  // A/B switch
  const int n0 = F->n[0], n1 = F->n[1], n2 = F->n[2];
  if (off || !F->zfft || n0 < 16 || n0 > 1024 || (n0 & (n0 - 1)) || (n1 & 1)) return INS_OK;
  std::vector<double> want(n0), have(lam_x_host, lam_x_host + n0);
  for (int k = 0; k < n0; ++k) {
    const double sn = std::sin(M_PI * (double)k / n0);
    want[k] = -4.0 * sn * sn / (hx * hx);
  }
  std::vector<double> ws(want);
  std::sort(ws.begin(), ws.end());
  std::sort(have.begin(), have.end());
  for (int k = 0; k < n0; ++k)
    if (std::fabs(ws[k] - have[k]) > 1e-9 * 4.0 / (hx * hx)) return INS_OK;
  const int kxn = n0 / 2 + 1, kxs = (kxn + 7) & ~7;
  std::vector<double> lxd(2 * kxs, -1e300), oxd(2 * kxs, 0.0);  // padding slots: data is zero, keep the division harmless
  for (int m = 0; m < kxn; ++m) lxd[2 * m] = lxd[2 * m + 1] = want[m];
  oxd[0] = std::sqrt((double)n0 / hx);
  const size_t elems = std::max<size_t>((size_t)n0 * n1 * n2, (size_t)2 * kxs * n1 * n2);
  const int nblk = (int)(((long long)kxs * n1 + 7) / 8);
  (void)hipFree(F->a);
  (void)hipFree(F->b);
  (void)hipFree(F->zpart);
  F->a = F->b = F->zpart = nullptr;
  bool ok = hipMalloc(&F->a, elems * 8) == hipSuccess && hipMalloc(&F->b, elems * 8) == hipSuccess && hipMemset(F->a, 0, elems * 8) == hipSuccess &&
            hipMemset(F->b, 0, elems * 8) == hipSuccess && hipMalloc(&F->zpart, (size_t)nblk * 8) == hipSuccess &&
            hipMemset(F->zpart, 0, (size_t)nblk * 8) == hipSuccess && hipMalloc(&F->lxd, lxd.size() * 8) == hipSuccess &&
            hipMalloc(&F->oxd, oxd.size() * 8) == hipSuccess && hipMemcpy(F->lxd, lxd.data(), lxd.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(F->oxd, oxd.data(), oxd.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    ins_set_error("ins_fdm_enable_xfft: allocation failed");
    return INS_ERR_HIP;
  }
  int rc = ins_zsolve_twiddles(n0, &F->xtw);
  if (rc) return rc;
  F->kxs = kxs;
  F->hx = hx;
  F->null_i[0] = 0;
  F->null_s[0] = 1.0 / std::sqrt(hx * (double)n0);
  F->xfft = true;
  F->fold[0] = F->fold[1] = false;  // the spectrum rows of this route are not folded (next: fold y there too)
  return INS_OK;
}

// Periodic uniform x and y, any z (walls in z): real-to-complex x pass, own y pass (digit-reversed ky), z through its dense eigenvectors on the
// interleaved (re, im) slots, the ordinary scaling pass with the eigenvalue vectors laid out like the data, and back.
static bool analytic_spectrum(const double* lam_host, int n, double h) {
  std::vector<double> want(n), have(lam_host, lam_host + n);
  for (int k = 0; k < n; ++k) {
    const double sn = std::sin(M_PI * (double)k / n);
    want[k] = -4.0 * sn * sn / (h * h);
  }
  std::sort(want.begin(), want.end());
  std::sort(have.begin(), have.end());
  for (int k = 0; k < n; ++k)
    if (std::fabs(want[k] - have[k]) > 1e-9 * 4.0 / (h * h)) return false;
  return true;
}

int ins_fdm_enable_xyfft(ins_fdm* F, double hx, double hy, const double* lam_x_host, const double* lam_y_host) {
  const bool off = ins_opt(OPT_INS_DISABLE_FDM_XYFFT) != 0;  // A/B switch
  const int n0 = F->n[0], n1 = F->n[1], n2 = F->D == 3 ? F->n[2] : 1;
  auto pow2 = [](int n) { return n >= 16 && n <= 1024 && !(n & (n - 1)); };
  if (off || F->xfft || F->zfft || F->D != 3 || !pow2(n0) || !pow2(n1)) return INS_OK;
  if (!analytic_spectrum(lam_x_host, n0, hx) || !analytic_spectrum(lam_y_host, n1, hy)) return INS_OK;
  const int kxn = n0 / 2 + 1, kxs = (kxn + 7) & ~7;
  std::vector<double> lxd(2 * kxs, -1e300), oxd(2 * kxs, 0.0), ly(n1), lyp(n1), oyp(n1, 0.0);
  for (int m = 0; m < kxn; ++m) {
    const double sn = std::sin(M_PI * (double)m / n0);
    lxd[2 * m] = lxd[2 * m + 1] = -4.0 * sn * sn / (hx * hx);
  }
  for (int k = 0; k < n1; ++k) {
    const double sn = std::sin(M_PI * (double)k / n1);
    ly[k] = -4.0 * sn * sn / (hy * hy);
  }
  ins_ownfft_permute_symbol(n1, ly.data(), lyp.data());
  oxd[0] = std::sqrt((double)n0 / hx);
  oyp[0] = std::sqrt((double)n1 / hy);  // frequency 0 sits at position 0
  const size_t elems = std::max<size_t>((size_t)n0 * n1 * n2, (size_t)2 * kxs * n1 * n2);
  (void)hipFree(F->a);
  (void)hipFree(F->b);
  F->a = F->b = nullptr;
  bool ok = hipMalloc(&F->a, elems * 8) == hipSuccess && hipMalloc(&F->b, elems * 8) == hipSuccess && hipMemset(F->a, 0, elems * 8) == hipSuccess &&
            hipMemset(F->b, 0, elems * 8) == hipSuccess && hipMalloc(&F->lxd, lxd.size() * 8) == hipSuccess && hipMalloc(&F->oxd, oxd.size() * 8) == hipSuccess &&
            hipMalloc(&F->lyp, lyp.size() * 8) == hipSuccess && hipMalloc(&F->oyp, oyp.size() * 8) == hipSuccess &&
            hipMemcpy(F->lxd, lxd.data(), lxd.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(F->oxd, oxd.data(), oxd.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(F->lyp, lyp.data(), lyp.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(F->oyp, oyp.data(), oyp.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    ins_set_error("ins_fdm_enable_xyfft: allocation failed");
    return INS_ERR_HIP;
  }
  int rc = ins_zsolve_twiddles(n0, &F->xtw);
  if (rc) return rc;
  if ((rc = ins_zsolve_twiddles(n1, &F->ytw))) return rc;
  F->kxs = kxs;
  F->hx = hx;
  F->hy = hy;
  // null mode: kx = 0 (real slot), ky = 0 (position 0), the z null mode as before; V₀ = scale · 1 with the Fourier directions' 1/sqrt(h n)
  const double sx = 1.0 / std::sqrt(hx * (double)n0), sy = 1.0 / std::sqrt(hy * (double)n1);
  F->null_index = (long long)F->null_i[2] * (2LL * kxs) * n1;
  F->null_scale = sx * sy * F->null_s[2];
  F->xyfft = true;
  F->fold[0] = F->fold[1] = F->fold[2] = false;
  return INS_OK;
}

// in: f on the unpadded block (n0,n1,n2) in F->a; out: p in F->a
// G, u given (only meaningful when ins_fdm_takes_u): the right-hand side Ω·div(u) is formed inside the x pass instead of being read from F->a
// folded_io: the caller wrote the right-hand side in folded form and reads the solution in folded form (project!: ins_poisson.hip)
int ins_fdm_solve(ins_fdm* F, hipStream_t s, const ins_grid* G, const double* u, bool folded_io) {
  const int n0 = F->n[0], n1 = F->n[1], n2 = F->D == 3 ? F->n[2] : 1;
  const long long n01 = (long long)n0 * n1, total = n01 * n2;
  const double one = 1.0, zero = 0.0;
  INS_BLAS_TRY(rocblas_set_stream(F->h, s));
  INS_BLAS_TRY(rocblas_set_pointer_mode(F->h, rocblas_pointer_mode_host));
  double *x = F->a, *y = F->b;
  if (F->xfft) {  // x and z in Fourier modes, y through its eigenvectors: x r2c, y GEMM, fused z pass, y GEMM, x c2r
    const int m2 = 2 * F->kxs;
    const long long m2n1 = (long long)m2 * n1;
    const double alpha = 1.0 / std::sqrt(F->hx * (double)n0);
    int rc = ins_k_ownfft_xfwd(u ? G : nullptr, u ? u : x, u ? 4 : 0, y, n0, n1, n2, F->xtw, s, F->kxs, 0);
    if (rc) return rc;
    INS_BLAS_TRY(rocblas_dgemm_strided_batched(F->h, rocblas_operation_none, rocblas_operation_none, m2, n1, n1, &alpha, y, m2, m2n1, F->V[1], n1, 0,
                                               &zero, x, m2, m2n1, n2));
    if (F->singular)
      hipLaunchKernelGGL(k_fdm_null_z, dim3(1), dim3(64), 0, s, x, (long long)m2 * F->null_i[1], m2n1, n2,
                         1.0 / (F->null_s[0] * F->null_s[1] * (double)total), F->sums);
    int nb = 0;
    rc = ins_k_fdm_z(x, m2, n1, n2, F->lxd, F->lam[1], F->lzk, F->oxd, F->ones[1], F->hz, F->lam_tol, F->singular ? 1 : 0, F->sums + 4096, F->zpart,
                     F->ztw, &nb, s);
    if (rc) return rc;
    if (F->singular) hipLaunchKernelGGL(k_fdm_mean_z, dim3(1), dim3(256), 0, s, F->zpart, nb, 1.0 / (double)total, F->sums);
    INS_LAUNCH_CHECK();
    INS_BLAS_TRY(rocblas_dgemm_strided_batched(F->h, rocblas_operation_none, rocblas_operation_transpose, m2, n1, n1, &alpha, x, m2, m2n1, F->V[1], n1,
                                               0, &zero, y, m2, m2n1, n2));
    return ins_k_ownfft_xinv(y, x, n0, n1, n2, F->xtw, s, F->kxs);
  }
  if (F->xyfft) {  // x r2c, y FFT (own passes), z GEMM on the interleaved slots, scaling, and back
    const int kxn = n0 / 2 + 1, m2 = 2 * F->kxs;
    const long long rows = (long long)m2 * n1, tot2 = rows * n2;
    const double alpha = 1.0 / (std::sqrt(F->hx * (double)n0) * std::sqrt(F->hy * (double)n1));
    const int nblk2 = (int)std::min<long long>((tot2 + 255) / 256, 4096);
    int rc = ins_k_ownfft_xfwd(u ? G : nullptr, u ? u : x, u ? 4 : 0, y, n0, n1, n2, F->xtw, s, F->kxs, 0);
    if (rc) return rc;
    if ((rc = ins_k_ownfft_y(y, kxn, n1, n2, F->ytw, false, s, F->kxs))) return rc;
    INS_BLAS_TRY(rocblas_dgemm(F->h, rocblas_operation_none, rocblas_operation_none, (int)rows, n2, n2, &alpha, y, (int)rows, F->V[2], n2, &zero, x, (int)rows));
    if (F->singular) hipLaunchKernelGGL(k_fdm_null, dim3(1), dim3(1), 0, s, x, F->null_index, 1.0 / (F->null_scale * (double)total), F->sums);
    hipLaunchKernelGGL(k_fdm_scale, dim3(nblk2), dim3(256), 0, s, x, F->lxd, F->lyp, F->lam[2], F->oxd, F->oyp, F->ones[2], m2, n1, n2, F->lam_tol,
                       F->singular ? 1 : 0, F->sums);
    if (F->singular) hipLaunchKernelGGL(k_fdm_mean, dim3(1), dim3(256), 0, s, F->sums, nblk2, 1.0 / (double)total);
    INS_LAUNCH_CHECK();
    INS_BLAS_TRY(rocblas_dgemm(F->h, rocblas_operation_none, rocblas_operation_transpose, (int)rows, n2, n2, &alpha, x, (int)rows, F->V[2], n2, &zero, y,
                               (int)rows));
    if ((rc = ins_k_ownfft_y(y, kxn, n1, n2, F->ytw, true, s, F->kxs))) return rc;
    return ins_k_ownfft_xinv(y, x, n0, n1, n2, F->xtw, s, F->kxs);
  }
  const int nblk = (int)std::min<long long>((total + 255) / 256, 4096);
  // Transforms along one direction; a symmetric direction (F->fold) runs as two GEMMs of half the size on the folded data.
  auto gemm_x = [&](bool fwd, const double* in, double* out) -> int {
    if (F->fold[0]) {
      const int h = n0 / 2;
      for (int part = 0; part < 2; ++part)
        INS_BLAS_TRY(rocblas_dgemm(F->h, fwd ? rocblas_operation_transpose : rocblas_operation_none, rocblas_operation_none, h, n1 * n2, h, &one,
                                   F->Vh[0] + (size_t)part * h * h, h, in + part * h, n0, &zero, out + part * h, n0));
      return INS_OK;
    }
    INS_BLAS_TRY(rocblas_dgemm(F->h, fwd ? rocblas_operation_transpose : rocblas_operation_none, rocblas_operation_none, n0, n1 * n2, n0, &one, F->V[0], n0,
                               in, n0, &zero, out, n0));
    return INS_OK;
  };
  auto gemm_y = [&](bool fwd, const double* in, double* out) -> int {
    const rocblas_operation opb = fwd ? rocblas_operation_none : rocblas_operation_transpose;
    if (F->fold[1]) {
      const int h = n1 / 2;
      for (int part = 0; part < 2; ++part)
        INS_BLAS_TRY(rocblas_dgemm_strided_batched(F->h, rocblas_operation_none, opb, n0, h, h, &one, in + (size_t)part * n0 * h, n0, n01,
                                                   F->Vh[1] + (size_t)part * h * h, h, 0, &zero, out + (size_t)part * n0 * h, n0, n01, n2));
      return INS_OK;
    }
    INS_BLAS_TRY(rocblas_dgemm_strided_batched(F->h, rocblas_operation_none, opb, n0, n1, n1, &one, in, n0, n01, F->V[1], n1, 0, &zero, out, n0, n01, n2));
    return INS_OK;
  };
  auto gemm_z = [&](bool fwd, const double* in, double* out) -> int {
    const rocblas_operation opb = fwd ? rocblas_operation_none : rocblas_operation_transpose;
    if (F->fold[2]) {
      const int h = n2 / 2;
      for (int part = 0; part < 2; ++part)
        INS_BLAS_TRY(rocblas_dgemm(F->h, rocblas_operation_none, opb, (int)n01, h, h, &one, in + (size_t)part * n01 * h, (int)n01,
                                   F->Vh[2] + (size_t)part * h * h, h, &zero, out + (size_t)part * n01 * h, (int)n01));
      return INS_OK;
    }
    INS_BLAS_TRY(rocblas_dgemm(F->h, rocblas_operation_none, opb, (int)n01, n2, n2, &one, in, (int)n01, F->V[2], n2, &zero, out, (int)n01));
    return INS_OK;
  };
  const int fmask = (F->fold[0] ? 1 : 0) | (F->fold[1] ? 2 : 0) | ((F->D == 3 && F->fold[2]) ? 4 : 0);
  const int nfold = (int)std::min<long long>((total / (fmask ? 2 : 1) + 255) / 256, 8192);
  int rc;
  // x: where the data is, y: the other buffer.  The fold and the unfold are out-of-place passes, so with them every branch below makes an even
  // number of buffer changes and the result lands in F->a as without them.
  if (fmask && !folded_io) {
    hipLaunchKernelGGL(k_fdm_fold, dim3(nfold), dim3(256), 0, s, (const double*)x, y, n0, n1, n2, fmask, 0);
    std::swap(x, y);
  }
  // forward: q = (Vxᵀ ⊗ Vyᵀ ⊗ Vzᵀ) f
  if ((rc = gemm_x(true, x, y))) return rc;
  if ((rc = gemm_y(true, y, x))) return rc;
  if (F->zfft || F->zdct) {  // x holds (Vxᵀ ⊗ Vyᵀ) f: the z transform, the scaling and the inverse z transform are one pass
    const long long idx_xy = F->null_i[0] + (long long)n0 * F->null_i[1];
    if (F->singular)
      hipLaunchKernelGGL(k_fdm_null_z, dim3(1), dim3(64), 0, s, x, idx_xy, n01, n2, 1.0 / (F->null_s[0] * F->null_s[1] * (double)total), F->sums);
    int nb = 0;
    rc = ins_k_fdm_z(x, n0, n1, n2, F->lam[0], F->lam[1], F->lzk, F->ones[0], F->ones[1], F->hz, F->lam_tol, F->singular ? 1 : 0, F->sums + 4096,
                     F->zpart, F->ztw, &nb, s, F->zdct ? F->zwq : nullptr);
    if (rc) return rc;
    if (F->singular) hipLaunchKernelGGL(k_fdm_mean_z, dim3(1), dim3(256), 0, s, F->zpart, nb, 1.0 / (double)total, F->sums);
    INS_LAUNCH_CHECK();
    if ((rc = gemm_y(false, x, y))) return rc;
    if ((rc = gemm_x(false, y, x))) return rc;
    if (fmask && !folded_io) {
      hipLaunchKernelGGL(k_fdm_fold, dim3(nfold), dim3(256), 0, s, (const double*)x, y, n0, n1, n2, fmask, 1);
      std::swap(x, y);
    }
    INS_LAUNCH_CHECK();
    if (x != F->a) INS_HIP_TRY(hipMemcpyAsync(F->a, x, total * 8, hipMemcpyDeviceToDevice, s));
    return INS_OK;
  }
  if (F->D == 3) {
    if ((rc = gemm_z(true, x, y))) return rc;
    std::swap(x, y);
  }
  // x holds Vᵀf
  if (F->singular) hipLaunchKernelGGL(k_fdm_null, dim3(1), dim3(1), 0, s, x, F->null_index, 1.0 / (F->null_scale * (double)total), F->sums);
  hipLaunchKernelGGL(k_fdm_scale, dim3(nblk), dim3(256), 0, s, x, F->lam[0], F->lam[1], F->D == 3 ? F->lam[2] : nullptr, F->ones[0], F->ones[1],
                     F->D == 3 ? F->ones[2] : nullptr, n0, n1, n2, F->lam_tol, F->singular ? 1 : 0, F->sums);
  if (F->singular) hipLaunchKernelGGL(k_fdm_mean, dim3(1), dim3(256), 0, s, F->sums, nblk, 1.0 / (double)total);
  INS_LAUNCH_CHECK();
  // backward: p = (Vx ⊗ Vy ⊗ Vz) q
  if (F->D == 3) {
    if ((rc = gemm_z(false, x, y))) return rc;
    std::swap(x, y);
  }
  if ((rc = gemm_y(false, x, y))) return rc;
  if ((rc = gemm_x(false, y, x))) return rc;
  if (fmask && !folded_io) {
    hipLaunchKernelGGL(k_fdm_fold, dim3(nfold), dim3(256), 0, s, (const double*)x, y, n0, n1, n2, fmask, 1);
    std::swap(x, y);
  }
  INS_LAUNCH_CHECK();
  // keep the result in F->a in every case
  if (x != F->a) INS_HIP_TRY(hipMemcpyAsync(F->a, x, total * 8, hipMemcpyDeviceToDevice, s));
  return INS_OK;
}

double* ins_fdm_buffer(ins_fdm* F) { return F->a; }
int ins_fdm_fold_mask(const ins_fdm* F) { return (F->fold[0] ? 1 : 0) | (F->fold[1] ? 2 : 0) | (F->fold[2] ? 4 : 0); }

// the solve starts with an x pass that can form Ω·div(u) itself (ins_fdm_solve(F, s, G, u))
bool ins_fdm_takes_u(const ins_fdm* F) { return F->xfft || F->xyfft; }
int ins_fdm_modes(const ins_fdm* F) { return (F->zfft ? 1 : 0) | (F->xfft ? 2 : 0) | (F->xyfft ? 4 : 0) | (F->zdct ? 8 : 0); }

// device scalar mean(p[Ip]) of the last solve: the consumer of the buffer subtracts it (e'p = 0, pressure.jl:133-140); nullptr when L is regular
const double* ins_fdm_mean(ins_fdm* F) { return F->singular ? F->sums + 4097 : nullptr; }
