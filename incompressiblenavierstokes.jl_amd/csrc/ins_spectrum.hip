// observespectrum (processors.jl:303-332) on the device: per velocity component strip the ghosts, one rocFFT real-to-complex
// transform over the Np interior points (hipFFT API), then shell sums  ehat[i] = Σ_{m ∈ inds[i]} Σ_α |û_α[m]|² / (2 prod(Np)²)
// over the host-built index sets of spectral_stuff (utils.jl:49-108).  The reference transforms complex-to-complex and keeps the
// non-negative quadrant k_α < K_α = Np_α / 2; the real-to-complex output holds exactly those modes.
#include "ins_internal.h"

struct ins_spectrum {
  const ins_grid* grid;
  hipfftHandle plan = 0;
  bool has_plan = false;
  double* real = nullptr;                // Np
  hipfftDoubleComplex* hat = nullptr;    // (Np0/2+1) Np1 [Np2]
  long long* offsets = nullptr;          // nbin + 1
  long long* inds = nullptr;             // positions inside `hat`
  double* weights = nullptr;             // optional weight per index (get_scale_numbers: 1/|k|)
  int nbin = 0;
  int np[3] = {1, 1, 1};
  double scale = 0.0;
};

namespace {

template <int D>
__global__ __launch_bounds__(256) void k_strip(GridDev g, const double* __restrict__ f, double* __restrict__ out, int n0, int n1) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= n0 || j >= n1) return;
  const long long c = (g.ip_lo[0] + i) + (g.ip_lo[1] + j) * g.sx[1] + (D == 3 ? (g.ip_lo[2] + k) * g.sx[2] : 0);
  out[i + (long long)n0 * (j + (long long)n1 * k)] = f[c];
}

// one wavefront per shell
__global__ __launch_bounds__(64) void k_shell_sums(const double2* __restrict__ hat, const long long* __restrict__ offsets,
                                                   const long long* __restrict__ inds, const double* __restrict__ weights, double scale,
                                                   double* __restrict__ ehat) {
  const int bin = blockIdx.x;
  double acc = 0.0;
  for (long long q = offsets[bin] + threadIdx.x; q < offsets[bin + 1]; q += 64) {
    const double2 v = hat[inds[q]];
    acc += (weights ? weights[q] : 1.0) * (v.x * v.x + v.y * v.y);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) ehat[bin] += scale * acc;
}

}  // namespace

extern "C" int ins_spectrum_destroy(ins_spectrum_t* S) {
  if (!S) return INS_OK;
  if (S->has_plan) (void)hipfftDestroy(S->plan);
  (void)hipFree(S->real);
  (void)hipFree(S->hat);
  (void)hipFree(S->offsets);
  (void)hipFree(S->inds);
  (void)hipFree(S->weights);
  delete S;
  return INS_OK;
}

extern "C" int ins_spectrum_create(const ins_grid_t* G, int nbin, const int64_t* offsets, const int64_t* inds, ins_spectrum_t** out) {
  return ins_spectrum_create_weighted(G, nbin, offsets, inds, nullptr, out);
}

extern "C" int ins_spectrum_create_weighted(const ins_grid_t* G, int nbin, const int64_t* offsets, const int64_t* inds, const double* weights,
                                            ins_spectrum_t** out) {
  INS_REQUIRE(G && offsets && inds && out && nbin >= 1, "bad argument");
  const GridDev& g = G->g;
  ins_spectrum* S = new ins_spectrum();
  S->grid = G;
  S->nbin = nbin;
  long long ntot = 1;
  int K[3] = {1, 1, 1};
  for (int a = 0; a < g.D; ++a) {
    S->np[a] = g.ip_hi[a] - g.ip_lo[a];
    K[a] = S->np[a] / 2;
    ntot *= S->np[a];
  }
  S->scale = 1.0 / (2.0 * (double)ntot * (double)ntot);
  const long long h0 = S->np[0] / 2 + 1;
  const long long nhat = h0 * S->np[1] * S->np[2];
  const long long nk = (long long)K[0] * K[1] * K[2];
  // index sets address the K-array of the reference (column-major over K): re-address them inside the real-to-complex output
  const long long nind = offsets[nbin];
  std::vector<long long> pos((size_t)nind);
  for (long long q = 0; q < nind; ++q) {
    const long long m = inds[q];
    if (m < 0 || m >= nk) {
      delete S;
      ins_set_error("spectrum index %lld outside the %lld retained modes", m, nk);
      return INS_ERR_INVALID;
    }
    const long long kx = m % K[0], ky = (m / K[0]) % K[1], kz = m / ((long long)K[0] * K[1]);
    pos[q] = kx + h0 * (ky + (long long)S->np[1] * kz);
  }
  std::vector<long long> off(offsets, offsets + nbin + 1);
  bool ok = hipMalloc(&S->real, ntot * sizeof(double)) == hipSuccess && hipMalloc(&S->hat, nhat * sizeof(hipfftDoubleComplex)) == hipSuccess &&
            hipMalloc(&S->offsets, (nbin + 1) * sizeof(long long)) == hipSuccess &&
            hipMalloc(&S->inds, std::max<long long>(nind, 1) * sizeof(long long)) == hipSuccess;
  if (ok) ok = hipMemcpy(S->offsets, off.data(), (nbin + 1) * sizeof(long long), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && nind) ok = hipMemcpy(S->inds, pos.data(), nind * sizeof(long long), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && weights && nind)
    ok = hipMalloc(&S->weights, nind * sizeof(double)) == hipSuccess && hipMemcpy(S->weights, weights, nind * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    ins_spectrum_destroy(S);
    ins_set_error("spectrum: device allocation failed");
    return INS_ERR_HIP;
  }
  int dims[3];
  for (int a = 0; a < g.D; ++a) dims[a] = S->np[g.D - 1 - a];  // slowest first
  // A plain hipFFT plan.  (The validating factory of the Poisson solvers, csrc/ins_fftcheck.hip, is not used here: when it suspects the
  // rocFFT plan-cache defect it resets rocFFT's process-wide state, which a diagnostic must not do to its host — PyTorch's torch.fft plans
  // die with it — and which leaks ~30 MB of code objects per reset.  The observer's output is checked against the oracle in the tests.)
  if (hipfftPlanMany(&S->plan, g.D, dims, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, 1) != HIPFFT_SUCCESS) {
    ins_spectrum_destroy(S);
    ins_set_error("spectrum: hipfftPlanMany failed");
    return INS_ERR_FFT;
  }
  S->has_plan = true;
  *out = S;
  return INS_OK;
}

extern "C" int ins_spectrum_f64(ins_spectrum_t* S, const double* u, double* ehat, void* stream) {
  INS_REQUIRE(S && u && ehat, "null argument");
  const GridDev& g = S->grid->g;
  hipStream_t s = as_stream(stream);
  INS_FFT_TRY(hipfftSetStream(S->plan, s));
  INS_HIP_TRY(hipMemsetAsync(ehat, 0, S->nbin * sizeof(double), s));
  dim3 block(64, 4, 1), grid(cdiv(S->np[0], 64), cdiv(S->np[1], 4), (unsigned)S->np[2]);
  for (int a = 0; a < g.D; ++a) {
    if (g.D == 2)
      hipLaunchKernelGGL(k_strip<2>, grid, block, 0, s, g, u + a * g.sc, S->real, S->np[0], S->np[1]);
    else
      hipLaunchKernelGGL(k_strip<3>, grid, block, 0, s, g, u + a * g.sc, S->real, S->np[0], S->np[1]);
    INS_LAUNCH_CHECK();
    INS_FFT_TRY(hipfftExecD2Z(S->plan, S->real, S->hat));
    hipLaunchKernelGGL(k_shell_sums, dim3(S->nbin), dim3(64), 0, s, reinterpret_cast<const double2*>(S->hat), S->offsets, S->inds, S->weights, S->scale, ehat);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}
