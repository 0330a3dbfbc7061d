// observespectrum (processors.jl:303-332) on the device: per velocity component strip the ghosts, one real-to-complex transform over the
// Np interior points, then shell sums  ehat[i] = Σ_{m ∈ inds[i]} Σ_α |û_α[m]|² / (2 prod(Np)²)
// over the host-built index sets of spectral_stuff (utils.jl:49-108).  The reference transforms complex-to-complex and keeps the
// non-negative quadrant k_α < K_α = Np_α / 2; the real-to-complex output holds exactly those modes.
// Transform: 3-D boxes whose sides the library's own passes take (x: paired-row real transform of ins_fft.hip; y, z: the register passes k_line3 of
// ins_zsolve.hip, which leave ky / kz in THEIR storage order — the index sets are re-addressed to it once, at creation) — three passes of 24 B per
// volume instead of a rocFFT 3-D plan (round 2: 1.71 ms for the three components at 256^3, 0.09 of 8 TB/s); other boxes: a hipFFT D2Z plan.
// Shell sums: the index list is cut into chunks of 4096 that do not cross shells, one workgroup per chunk, and the chunk sums of a shell are added in
// order by one work-item — deterministic, and 2 000 workgroups instead of one wavefront per shell.
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
  // own passes (3-D)
  bool own = false;
  int kxs = 0;
  double* tw[3] = {nullptr, nullptr, nullptr};
  // chunked shell sums
  int nchunk = 0;
  long long* chunk_lo = nullptr;   // nchunk + 1 starts inside the index list (chunk c = [chunk_lo[c], chunk_hi[c]))
  long long* chunk_hi = nullptr;
  int* bin_chunk0 = nullptr;       // nbin + 1: first chunk of every shell
  double* partial = nullptr;       // nchunk
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

// one workgroup per chunk of the index list (a chunk lies inside one shell)
__global__ __launch_bounds__(256) void k_shell_chunks(const double2* __restrict__ hat, const long long* __restrict__ lo, const long long* __restrict__ hi,
                                                      const long long* __restrict__ inds, const double* __restrict__ weights, double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  for (long long q = lo[blockIdx.x] + threadIdx.x; q < hi[blockIdx.x]; q += 256) {
    const double2 v = hat[inds[q]];
    acc += (weights ? weights[q] : 1.0) * (v.x * v.x + v.y * v.y);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// one work-item per shell: its chunk sums in order
__global__ __launch_bounds__(64) void k_shell_finish(const int* __restrict__ bin_chunk0, const double* __restrict__ partial, int nbin, double scale,
                                                     double* __restrict__ ehat) {
  const int bin = blockIdx.x * 64 + threadIdx.x;
  if (bin >= nbin) return;
  double acc = 0.0;
  for (int c = bin_chunk0[bin]; c < bin_chunk0[bin + 1]; ++c) acc += partial[c];
  ehat[bin] += scale * acc;
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
  for (double* t : S->tw)
    if (t) (void)hipFree(t);
  (void)hipFree(S->chunk_lo);
  (void)hipFree(S->chunk_hi);
  (void)hipFree(S->bin_chunk0);
  (void)hipFree(S->partial);
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
  // own passes: every side a length the x pass takes, y and z lengths the register passes take (INS_SPECTRUM_ROCFFT=1 keeps the hipFFT plan)
  S->own = g.D == 3 && !ins_opt(OPT_INS_SPECTRUM_ROCFFT) && ins_ownfft_supported_mixed(S->np) && ins_line3_supported(S->np[1]) && ins_line3_supported(S->np[2]);
  S->kxs = S->own ? (int)((h0 + 7) & ~7LL) : (int)h0;  // rows padded to whole 128-B lines, as in the Poisson solver
  const long long nhat = (long long)S->kxs * S->np[1] * S->np[2];
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
    const long long py = S->own ? ins_line3_pos_of_freq(S->np[1], (int)ky) : ky, pz = S->own ? ins_line3_pos_of_freq(S->np[2], (int)kz) : kz;
    pos[q] = kx + (long long)S->kxs * (py + (long long)S->np[1] * pz);
  }
  // chunks of the index list, none crossing a shell
  constexpr long long CH = 4096;
  std::vector<long long> clo, chi;
  std::vector<int> bc0(nbin + 1, 0);
  for (int b = 0; b < nbin; ++b) {
    bc0[b] = (int)clo.size();
    for (long long q = offsets[b]; q < offsets[b + 1]; q += CH) {
      clo.push_back(q);
      chi.push_back(std::min<long long>(q + CH, offsets[b + 1]));
    }
  }
  bc0[nbin] = (int)clo.size();
  S->nchunk = (int)clo.size();
  std::vector<long long> off(offsets, offsets + nbin + 1);
  bool ok = hipMalloc(&S->real, ntot * sizeof(double)) == hipSuccess && hipMalloc(&S->hat, nhat * sizeof(hipfftDoubleComplex)) == hipSuccess &&
            hipMalloc(&S->offsets, (nbin + 1) * sizeof(long long)) == hipSuccess &&
            hipMalloc(&S->inds, std::max<long long>(nind, 1) * sizeof(long long)) == hipSuccess;
  if (ok) ok = hipMemcpy(S->offsets, off.data(), (nbin + 1) * sizeof(long long), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && nind) ok = hipMemcpy(S->inds, pos.data(), nind * sizeof(long long), hipMemcpyHostToDevice) == hipSuccess;
  if (ok && weights && nind)
    ok = hipMalloc(&S->weights, nind * sizeof(double)) == hipSuccess && hipMemcpy(S->weights, weights, nind * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    const size_t nc = std::max<size_t>(clo.size(), 1);
    ok = hipMalloc(&S->chunk_lo, nc * sizeof(long long)) == hipSuccess && hipMalloc(&S->chunk_hi, nc * sizeof(long long)) == hipSuccess &&
         hipMalloc(&S->bin_chunk0, (nbin + 1) * sizeof(int)) == hipSuccess && hipMalloc(&S->partial, nc * sizeof(double)) == hipSuccess &&
         hipMemcpy(S->bin_chunk0, bc0.data(), (nbin + 1) * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && !clo.empty())
      ok = hipMemcpy(S->chunk_lo, clo.data(), clo.size() * sizeof(long long), hipMemcpyHostToDevice) == hipSuccess &&
           hipMemcpy(S->chunk_hi, chi.data(), chi.size() * sizeof(long long), hipMemcpyHostToDevice) == hipSuccess;
  }
  if (ok && S->own) {
    ok = hipMemset(S->hat, 0, nhat * sizeof(hipfftDoubleComplex)) == hipSuccess;
    for (int a = 0; ok && a < 3; ++a) ok = ins_zsolve_twiddles(S->np[a], &S->tw[a]) == INS_OK;
  }
  if (!ok) {
    ins_spectrum_destroy(S);
    ins_set_error("spectrum: device allocation failed");
    return INS_ERR_HIP;
  }
  if (S->own) {
    *out = S;
    return INS_OK;
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
  if (!S->own) INS_FFT_TRY(hipfftSetStream(S->plan, s));
  INS_HIP_TRY(hipMemsetAsync(ehat, 0, S->nbin * sizeof(double), s));
  dim3 block(64, 4, 1), grid(cdiv(S->np[0], 64), cdiv(S->np[1], 4), (unsigned)S->np[2]);
  for (int a = 0; a < g.D; ++a) {
    if (!S->own) {
      if (g.D == 2)
        hipLaunchKernelGGL(k_strip<2>, grid, block, 0, s, g, u + a * g.sc, S->real, S->np[0], S->np[1]);
      else
        hipLaunchKernelGGL(k_strip<3>, grid, block, 0, s, g, u + a * g.sc, S->real, S->np[0], S->np[1]);
      INS_LAUNCH_CHECK();
    }
    if (S->own) {
      double* ph = reinterpret_cast<double*>(S->hat);
      const int kxn = S->np[0] / 2 + 1;
      int rc;
      if ((rc = ins_k_ownfft_xfwd(S->grid, u + a * g.sc, 6, ph, S->np[0], S->np[1], S->np[2], S->tw[0], s, S->kxs))) return rc;  // ghosts stripped inside the x pass
      if ((rc = ins_k_line3_y(ph, kxn, S->np[1], S->np[2], S->tw[1], false, s, S->kxs))) return rc;
      if ((rc = ins_k_line3_z(ph, kxn, S->np[1], S->np[2], S->tw[2], s, S->kxs))) return rc;
    } else {
      INS_FFT_TRY(hipfftExecD2Z(S->plan, S->real, S->hat));
    }
    if (S->nchunk) hipLaunchKernelGGL(k_shell_chunks, dim3(S->nchunk), dim3(256), 0, s, reinterpret_cast<const double2*>(S->hat), S->chunk_lo, S->chunk_hi, S->inds, S->weights, S->partial);
    hipLaunchKernelGGL(k_shell_finish, dim3(cdiv(S->nbin, 64)), dim3(64), 0, s, S->bin_chunk0, S->partial, S->nbin, S->scale, ehat);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}
