// Plan self-check.  rocFFT in ROCm 7.2 (gfx950) can return WRONG real transforms depending on which plans
// were created earlier in the process (reproducer: tools/dbg_hipfft.cpp — with a 32x32 batched D2Z/Z2D pair alive,
// a new 16x64 D2Z plan is off by O(1)); a round trip does not reveal it because real pre/post-processing with a
// wrong twiddle table is still self-inverse.  Every spectral solver therefore validates its freshly created real
// plans once: a few output bins of two batches are compared with a direct DFT evaluated on the host, and the
// inverse plan is validated through a round trip of the (now trusted) forward plan.
#include <cmath>
#include <complex>

#include <rocfft/rocfft.h>

#include <atomic>

#include <mutex>

#include "ins_internal.h"

// fwd: D2Z plan over `rank` dims n[0..rank-1] (slowest first), batch `batch`, contiguous default layout.
int ins_validate_real_plans(hipfftHandle fwd, hipfftHandle inv, int rank, const int* n, int batch) {
  long long nreal = 1, ncplx = 1;
  for (int d = 0; d < rank; ++d) {
    nreal *= n[d];
    ncplx *= (d == rank - 1) ? n[d] / 2 + 1 : n[d];
  }
  const int kxn = n[rank - 1] / 2 + 1;
  std::vector<double> in((size_t)nreal * batch);
  unsigned s = 2463534242u;
  for (auto& v : in) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    v = (double)(s >> 8) / (1 << 24) - 0.5;
  }
  double *din = nullptr, *dback = nullptr;
  hipfftDoubleComplex* dout = nullptr;
  struct Guard {
    double*& a;
    double*& b;
    hipfftDoubleComplex*& c;
    ~Guard() {
      if (a) (void)hipFree(a);
      if (b) (void)hipFree(b);
      if (c) (void)hipFree(c);
    }
  } guard{din, dback, dout};
  INS_HIP_TRY(hipMalloc(&din, in.size() * sizeof(double)));
  INS_HIP_TRY(hipMalloc(&dback, in.size() * sizeof(double)));
  INS_HIP_TRY(hipMalloc(&dout, (size_t)ncplx * batch * sizeof(hipfftDoubleComplex)));
  INS_HIP_TRY(hipMemcpy(din, in.data(), in.size() * sizeof(double), hipMemcpyHostToDevice));
  INS_FFT_TRY(hipfftSetStream(fwd, nullptr));
  INS_FFT_TRY(hipfftSetStream(inv, nullptr));
  INS_FFT_TRY(hipfftExecD2Z(fwd, din, dout));
  INS_HIP_TRY(hipDeviceSynchronize());
  std::vector<std::complex<double>> out((size_t)ncplx * batch);
  INS_HIP_TRY(hipMemcpy(out.data(), dout, out.size() * sizeof(hipfftDoubleComplex), hipMemcpyDeviceToHost));
  // direct DFT of a handful of bins in the first and the last batch
  double worst = 0.0, scale = 0.0;
  const int nb = batch > 1 ? 2 : 1;
  for (int bi = 0; bi < nb; ++bi) {
    const int b = bi == 0 ? 0 : batch - 1;
    const double* x = in.data() + (size_t)b * nreal;
    for (int probe = 0; probe < 6; ++probe) {
      int k[3] = {0, 0, 0}, nn[3] = {1, 1, 1};
      for (int d = 0; d < rank; ++d) nn[3 - rank + d] = n[d];
      // probes spread over the spectrum, including the last kx bin
      k[2] = (probe * 7 + 1) % kxn;
      if (probe == 5) k[2] = kxn - 1;
      k[1] = (probe * 5 + 2) % nn[1];
      k[0] = (probe * 3 + 1) % nn[0];
      std::complex<double> acc = 0.0;
      for (int a = 0; a < nn[0]; ++a)
        for (int c = 0; c < nn[1]; ++c) {
          const double ph0 = (double)k[0] * a / nn[0] + (double)k[1] * c / nn[1];
          std::complex<double> row = 0.0;
          const double* xr = x + ((size_t)a * nn[1] + c) * nn[2];
          for (int e = 0; e < nn[2]; ++e) {
            const double ph = -2.0 * M_PI * ((double)k[2] * e / nn[2]);
            row += xr[e] * std::complex<double>(std::cos(ph), std::sin(ph));
          }
          const double p0 = -2.0 * M_PI * ph0;
          acc += row * std::complex<double>(std::cos(p0), std::sin(p0));
        }
      const size_t idx = (size_t)b * ncplx + ((size_t)k[0] * nn[1] + k[1]) * kxn + k[2];
      worst = std::fmax(worst, std::abs(acc - out[idx]));
      scale = std::fmax(scale, std::abs(acc));
    }
  }
  if (!(worst <= 1e-9 * std::fmax(scale, 1.0))) {
    ins_set_error("rocFFT returned a wrong D2Z transform for lengths [%d x %d x %d] batch %d (error %.3e of %.3e): known ROCm 7.2 "
                  "plan-cache bug when real plans of other sizes are alive in the process (tools/dbg_hipfft.cpp); destroy the other "
                  "solvers first",
                  rank > 2 ? n[rank - 3] : 1, rank > 1 ? n[rank - 2] : 1, n[rank - 1], batch, worst, scale);
    return INS_ERR_FFT;
  }
  // inverse through the round trip
  INS_FFT_TRY(hipfftExecZ2D(inv, dout, dback));
  INS_HIP_TRY(hipDeviceSynchronize());
  std::vector<double> back(in.size());
  INS_HIP_TRY(hipMemcpy(back.data(), dback, back.size() * sizeof(double), hipMemcpyDeviceToHost));
  double err = 0.0;
  for (size_t i = 0; i < in.size(); i += 97) err = std::fmax(err, std::fabs(back[i] / (double)nreal - in[i]));
  if (!(err <= 1e-10)) {
    ins_set_error("rocFFT Z2D plan failed its round-trip check (error %.3e)", err);
    return INS_ERR_FFT;
  }
  return INS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Plan factory used by every spectral solver that still needs rocFFT (box sizes the own passes of ins_fft.hip do not take): create the
// D2Z / Z2D pair and validate it.  A pair that fails validation is NEVER handed out: the error is returned (INS_ERR_FFT, message says what to
// do).  Only when the host has opted in (option / environment INS_FFT_ALLOW_RESET=1) and no other solver of this library holds rocFFT plans,
// rocFFT's process-wide caches are reset (rocfft_cleanup + rocfft_setup) and the pair is created once more: that reset invalidates every
// rocFFT / hipFFT plan the HOST owns (AMDGPU.jl's, PyTorch's torch.fft cache), so it is the host's decision, not the library's.
// Creation, validation and reset are serialised by one mutex (solver creation from several host threads).
// ------------------------------------------------------------------------------------------------------------
static std::mutex g_fft_mutex;
static std::atomic<int> g_live_fft_solvers{0};
static std::atomic<int> g_fft_resets{0};  // how often rocFFT's process-wide state was reset (other hipFFT users must drop their plans)

void ins_fft_solver_released() { g_live_fft_solvers.fetch_sub(1); }

static hipfftResult make_pair(hipfftHandle* fwd, hipfftHandle* inv, int rank, int* n, int batch) {
  hipfftResult r1, r2;
  if (batch > 1 || rank == 1) {
    r1 = hipfftPlanMany(fwd, rank, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, batch);
    if (r1 != HIPFFT_SUCCESS) return r1;
    r2 = hipfftPlanMany(inv, rank, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, batch);
  } else if (rank == 2) {
    r1 = hipfftPlan2d(fwd, n[0], n[1], HIPFFT_D2Z);
    if (r1 != HIPFFT_SUCCESS) return r1;
    r2 = hipfftPlan2d(inv, n[0], n[1], HIPFFT_Z2D);
  } else {
    r1 = hipfftPlan3d(fwd, n[0], n[1], n[2], HIPFFT_D2Z);
    if (r1 != HIPFFT_SUCCESS) return r1;
    r2 = hipfftPlan3d(inv, n[0], n[1], n[2], HIPFFT_Z2D);
  }
  if (r2 != HIPFFT_SUCCESS) (void)hipfftDestroy(*fwd);
  return r2;
}

int ins_fft_make_real_plans(hipfftHandle* fwd, hipfftHandle* inv, int rank, int* n, int batch) {
  std::lock_guard<std::mutex> lock(g_fft_mutex);
  hipfftResult r = make_pair(fwd, inv, rank, n, batch);
  if (r != HIPFFT_SUCCESS) {
    ins_set_error("hipfftPlan (rank %d, batch %d) failed: %d", rank, batch, (int)r);
    return INS_ERR_FFT;
  }
  int rc = ins_validate_real_plans(*fwd, *inv, rank, n, batch);
  if (rc == INS_ERR_FFT && g_live_fft_solvers.load() == 0 && ins_opt(OPT_INS_FFT_ALLOW_RESET)) {
    (void)hipfftDestroy(*fwd);
    (void)hipfftDestroy(*inv);
    (void)hipDeviceSynchronize();
    rocfft_cleanup();
    rocfft_setup();
    g_fft_resets.fetch_add(1);
    r = make_pair(fwd, inv, rank, n, batch);
    if (r != HIPFFT_SUCCESS) {
      ins_set_error("hipfftPlan (rank %d, batch %d) failed after rocFFT reset: %d", rank, batch, (int)r);
      return INS_ERR_FFT;
    }
    rc = ins_validate_real_plans(*fwd, *inv, rank, n, batch);
  }
  if (rc) {
    (void)hipfftDestroy(*fwd);
    (void)hipfftDestroy(*inv);
    if (rc == INS_ERR_FFT && !ins_opt(OPT_INS_FFT_ALLOW_RESET)) {
      std::string first = ins_last_error();
      ins_set_error("%s -- rocFFT returned a plan pair that fails validation (plan-cache defect of ROCm 7.2, DESIGN.md §3).  Destroy the other "
                    "real-transform plans of this process, or allow the library to reset rocFFT's process-wide state with "
                    "ins_set_option(\"INS_FFT_ALLOW_RESET\", 1) (the host must then re-create its own rocFFT/hipFFT plans; see ins_fft_reset_count)",
                    first.c_str());
    }
    return rc;
  }
  g_live_fft_solvers.fetch_add(1);
  return INS_OK;
}

// Number of rocFFT resets this library has performed in this process.  A reset invalidates every hipFFT / rocFFT plan that lives outside
// the library (PyTorch keeps a plan cache for torch.fft): a host that uses rocFFT itself compares this counter around solver creation and
// drops its plans when it moved (ins_amd does: torch.backends.cuda.cufft_plan_cache.clear()).
extern "C" int ins_fft_reset_count(void) { return g_fft_resets.load(); }
