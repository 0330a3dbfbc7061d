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

#include <cmath>

#include "ins_internal.h"

struct ins_fdm {
  rocblas_handle h = nullptr;
  int D = 3;
  int n[3] = {1, 1, 1};
  double* V[3] = {nullptr, nullptr, nullptr};    // n[a] x n[a], column-major, Dα-orthonormal eigenvectors
  double* lam[3] = {nullptr, nullptr, nullptr};  // eigenvalues (<= 0)
  double *a = nullptr, *b = nullptr;             // two n0*n1*n2 work arrays
  double* sums = nullptr;                        // partial sums for the mean shift
  bool singular = true;
  double lam_tol = 0.0;
};

namespace {

// q /= (λx + λy + λz), null mode -> 0
__global__ __launch_bounds__(256) void k_fdm_scale(double* __restrict__ q, const double* __restrict__ lx, const double* __restrict__ ly,
                                                   const double* __restrict__ lz, int n0, int n1, int n2, double tol) {
  const long long total = (long long)n0 * n1 * n2;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int i = (int)(t % n0);
    const long long r = t / n0;
    const int j = (int)(r % n1), k = (int)(r / n1);
    double lam = lx[i] + ly[j];
    if (lz) lam += lz[k];
    q[t] = (fabs(lam) <= tol) ? 0.0 : q[t] / lam;
  }
}

__global__ __launch_bounds__(256) void k_fdm_partial_sum(const double* __restrict__ p, long long n, double* __restrict__ partial) {
  __shared__ double lds[4];
  double acc = 0.0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) acc += p[t];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// p -= mean, mean from the per-block partial sums (no host round trip)
__global__ __launch_bounds__(256) void k_fdm_shift(double* __restrict__ p, long long n, const double* __restrict__ partial, int nblk) {
  __shared__ double mean;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[b];
    mean = s / (double)n;
  }
  __syncthreads();
  const double m = mean;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) p[t] -= m;
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
  }
  if (F->a) (void)hipFree(F->a);
  if (F->b) (void)hipFree(F->b);
  if (F->sums) (void)hipFree(F->sums);
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
    ok = hipMalloc(&F->V[a], (size_t)n[a] * n[a] * 8) == hipSuccess && hipMalloc(&F->lam[a], (size_t)n[a] * 8) == hipSuccess &&
         hipMemcpy(F->V[a], V[a], (size_t)n[a] * n[a] * 8, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(F->lam[a], lam[a], (size_t)n[a] * 8, hipMemcpyHostToDevice) == hipSuccess;
    for (int i = 0; i < n[a]; ++i) lmax = std::fmax(lmax, std::fabs(lam[a][i]));
  }
  ok = ok && hipMalloc(&F->a, total * 8) == hipSuccess && hipMalloc(&F->b, total * 8) == hipSuccess && hipMalloc(&F->sums, 1024 * 8) == hipSuccess;
  if (!ok) {
    ins_set_error("ins_fdm_create: allocation / upload failed");
    ins_fdm_destroy(F);
    return INS_ERR_HIP;
  }
  F->lam_tol = F->singular ? 1e-10 * lmax * D : 0.0;
  *out = F;
  return INS_OK;
}

// in: f on the unpadded block (n0,n1,n2) in F->a; out: p in F->a
int ins_fdm_solve(ins_fdm* F, hipStream_t s) {
  const int n0 = F->n[0], n1 = F->n[1], n2 = F->D == 3 ? F->n[2] : 1;
  const long long n01 = (long long)n0 * n1, total = n01 * n2;
  const double one = 1.0, zero = 0.0;
  INS_BLAS_TRY(rocblas_set_stream(F->h, s));
  INS_BLAS_TRY(rocblas_set_pointer_mode(F->h, rocblas_pointer_mode_host));
  double *x = F->a, *y = F->b;
  const int nblk = (int)std::min<long long>((total + 255) / 256, 1024);
  const unsigned nshift = (unsigned)std::min<long long>((total + 255) / 256, 4096);
  if (F->singular) {  // bordered system: L p = f - mean(f) e  (λ = e'f / e'e), which makes the right-hand side solvable
    hipLaunchKernelGGL(k_fdm_partial_sum, dim3(nblk), dim3(256), 0, s, F->a, total, F->sums);
    hipLaunchKernelGGL(k_fdm_shift, dim3(nshift), dim3(256), 0, s, F->a, total, F->sums, nblk);
    INS_LAUNCH_CHECK();
  }
  // forward: q = (Vxᵀ ⊗ Vyᵀ ⊗ Vzᵀ) f
  INS_BLAS_TRY(rocblas_dgemm(F->h, rocblas_operation_transpose, rocblas_operation_none, n0, n1 * n2, n0, &one, F->V[0], n0, x, n0, &zero, y, n0));
  INS_BLAS_TRY(rocblas_dgemm_strided_batched(F->h, rocblas_operation_none, rocblas_operation_none, n0, n1, n1, &one, y, n0, n01, F->V[1], n1, 0,
                                             &zero, x, n0, n01, n2));
  if (F->D == 3) {
    INS_BLAS_TRY(rocblas_dgemm(F->h, rocblas_operation_none, rocblas_operation_none, (int)n01, n2, n2, &one, x, (int)n01, F->V[2], n2, &zero, y,
                               (int)n01));
    std::swap(x, y);
  }
  // x holds Vᵀf
  hipLaunchKernelGGL(k_fdm_scale, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, s, x, F->lam[0], F->lam[1],
                     F->D == 3 ? F->lam[2] : nullptr, n0, n1, n2, F->lam_tol);
  INS_LAUNCH_CHECK();
  // backward: p = (Vx ⊗ Vy ⊗ Vz) q
  if (F->D == 3) {
    INS_BLAS_TRY(rocblas_dgemm(F->h, rocblas_operation_none, rocblas_operation_transpose, (int)n01, n2, n2, &one, x, (int)n01, F->V[2], n2, &zero,
                               y, (int)n01));
    std::swap(x, y);
  }
  INS_BLAS_TRY(rocblas_dgemm_strided_batched(F->h, rocblas_operation_none, rocblas_operation_transpose, n0, n1, n1, &one, x, n0, n01, F->V[1], n1,
                                             0, &zero, y, n0, n01, n2));
  INS_BLAS_TRY(rocblas_dgemm(F->h, rocblas_operation_none, rocblas_operation_none, n0, n1 * n2, n0, &one, F->V[0], n0, y, n0, &zero, x, n0));
  // x is F->a when D == 3 (two swaps) ... keep the result in F->a in every case
  if (x != F->a) INS_HIP_TRY(hipMemcpyAsync(F->a, x, total * 8, hipMemcpyDeviceToDevice, s));
  if (F->singular) {  // e'p = 0, the bordered system's constraint (pressure.jl:133-140)
    hipLaunchKernelGGL(k_fdm_partial_sum, dim3(nblk), dim3(256), 0, s, F->a, total, F->sums);
    hipLaunchKernelGGL(k_fdm_shift, dim3(nshift), dim3(256), 0, s, F->a, total, F->sums, nblk);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}

double* ins_fdm_buffer(ins_fdm* F) { return F->a; }
