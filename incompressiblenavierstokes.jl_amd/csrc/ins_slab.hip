// z-slab pieces of the multi-GPU path.  One process per GPU; rank r owns nzl = nz/P interior
// z-planes plus one ghost plane on each side, x and y stay whole.  This translation unit holds only the
// LOCAL work; the exchanges between ranks (u / p halo planes, the two FFT transposes) are done by the host
// with RCCL through torch.distributed (incompressiblenavierstokes.jl_amd/distributed.py).
//
// Distributed spectral solve (pressure.jl:289-351 restated for slabs):
//   forward_xy : 2-D R2C over (x, y) for each local plane, then pack to [dest q][kz_local][ky_local][kx]
//   (all-to-all)   -> every rank holds all nz planes of its ny/P rows:  [kz][ky_local][kx]
//   solve_z    : 1-D FFT along z, divide by the Laplacian symbol (mean mode zeroed, 1/prod(Np) folded in), inverse
//   (all-to-all back)
//   inverse_xy : unpack, 2-D C2R per plane -> pI
#include <algorithm>
#include <cmath>

#include "ins_internal.h"

bool ins_flux64_supported(const ins_grid* G);

struct ins_slab_fft {
  int np[3];         // global interior sizes
  int rank, nranks;
  int nzl, nyl, kxn; // local planes, local rows (transposed layout), nx/2+1
  hipfftHandle xy_fwd = 0, xy_inv = 0, z_fwd = 0;
  bool plans = false, has_z_plan = false;
  double *ax = nullptr, *ay = nullptr, *az = nullptr;  // symbol vectors (ay: this rank's ky range)
  double* tw = nullptr;                                 // twiddles of the fused z kernel (power-of-two nz)
  bool ownfft = false;                                  // power-of-two box: own x / y passes too (ins_fft.hip), no rocFFT
  double *tw_x = nullptr, *tw_y = nullptr;
  const ins_grid* dummy_grid = nullptr;
  // transpose-free z solve (ins_ztri.hip): every ky in storage order, Ω/Δz², interface values (L_{r-1}, F_{r+1}) per line
  double* ay_full = nullptr;
  // tridiagonal route: its y passes run on the register passes (k_line3, ins_zsolve.hip) where they exist; ay3 = the full symbol vector in THEIR storage order
  // (the transposes route keeps the LDS passes: its packed y kernels and the ky ownership of the ranks are tied to the digit-reversed order)
  bool y3 = false;
  double* ay3 = nullptr;
  double cz = 0.0;
  double* ztri_bc = nullptr;
  int kxs = 0;  // row stride of `work` on the transpose-free route: kxn rounded up to whole 128-B lines with the own passes
};

int ins_k_ztri_forward(double* work, int kxn, int kxs, int n1, int m, int nranks, int rank, const double* ax, const double* ay, double c,
                       double scale, double* edge, long long l_lo, long long l_cnt, hipStream_t s);
int ins_k_ztri_finish(double* work, int kxn, int kxs, int n1, int m, int nranks, int rank, const double* ax, const double* ay, double c,
                      const double* edges_all, long long stride, double* bc, long long l_lo, long long l_cnt, hipStream_t s);

namespace {

// work[kzl][ky][kx]  <->  buf[q][kzl][kyl][kxl],  ky = q*nyl + kyl,  kx = kx0 + kxl, kxl < kxc   (complex = double2)
// A chunk [kx0, kx0+kxc) of the spectrum is packed contiguously so that it can travel (and be solved) on its own.
template <bool PACK>
__global__ __launch_bounds__(256) void k_transpose_pack(double2* __restrict__ work, double2* __restrict__ buf, int kxn, int ny, int nyl,
                                                        int nzl, int kx0, int kxc) {
  const int kxl = blockIdx.x * 64 + threadIdx.x;
  const int ky = blockIdx.y * 4 + threadIdx.y;
  const int kz = blockIdx.z;
  if (kxl >= kxc || ky >= ny) return;
  const int q = ky / nyl, kyl = ky - q * nyl;
  const long long w = (kx0 + kxl) + (long long)kxn * (ky + (long long)ny * kz);
  const long long b = kxl + (long long)kxc * (kyl + (long long)nyl * (kz + (long long)nzl * q));
  if (PACK)
    buf[b] = work[w];
  else
    work[w] = buf[b];
}

// K3 on the transposed layout [kz][kyl][kx]
__global__ __launch_bounds__(256) void k_symbol_slab(double2* __restrict__ phat, const double* __restrict__ ax, const double* __restrict__ ay,
                                                     const double* __restrict__ az, int kxn, int nyl, int nz, double inv_n, int zero_mean) {
  const int kx = blockIdx.x * 64 + threadIdx.x;
  const int ky = blockIdx.y * 4 + threadIdx.y;
  const int kz = blockIdx.z;
  if (kx >= kxn || ky >= nyl) return;
  const long long q = kx + (long long)kxn * (ky + (long long)nyl * kz);
  const double den = ax[kx] + ay[ky] + az[kz];
  const bool mean = zero_mean && kx == 0 && ky == 0 && kz == 0;
  const double s = mean ? 0.0 : -inv_n / den;
  double2 v = phat[q];
  v.x *= s;
  v.y *= s;
  phat[q] = v;
}

// K2 on a slab: x, y neighbours through periodic wrap, z through the ghost plane (filled by the halo exchange)
__global__ __launch_bounds__(256) void k_div_slab(GridDev g, const double* __restrict__ u, double* __restrict__ pI, int n0, int n1) {
  const int ii = blockIdx.x * 64 + threadIdx.x;
  const int jj = blockIdx.y * 4 + threadIdx.y;
  const int kk = blockIdx.z;
  if (ii >= n0 || jj >= n1) return;
  const int I[3] = {1 + ii, 1 + jj, 1 + kk};
  const long long c = I[0] + I[1] * g.sx[1] + I[2] * g.sx[2];
  double d = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double* ua = u + a * g.sc;
    const long long cm = (a < 2 && I[a] == 1) ? c + (long long)(g.N[a] - 3) * g.sx[a] : c - g.sx[a];
    d += (ua[c] - ua[cm]) * g.rdx[a][I[a]];
  }
  const double om = g.dx[0][I[0]] * g.dx[1][I[1]] * g.dx[2][I[2]];
  pI[ii + (long long)n0 * (jj + (long long)n1 * kk)] = d * om;
}

// K4 on a slab: like k_grad_ghost3 but the z-neighbour of the top local plane comes from `p_top` (the next
// rank's first pI plane) and only x / y ghost images are written (z ghosts belong to the halo exchange).
__global__ __launch_bounds__(256) void k_grad_slab(GridDev g, double* __restrict__ u, const double* __restrict__ pI,
                                                   const double* __restrict__ p_top, int n0, int n1, int n2) {
  const int ii = blockIdx.x * 64 + threadIdx.x;
  const int jj = blockIdx.y * 4 + threadIdx.y;
  const int kk = blockIdx.z;
  if (ii >= n0 || jj >= n1) return;
  const int I[3] = {ii + 1, jj + 1, kk + 1};
  const long long q = ii + (long long)n0 * (jj + (long long)n1 * kk);
  const long long c = I[0] + I[1] * g.sx[1] + I[2] * g.sx[2];
  const double pc = pI[q];
  const double px = pI[(ii + 1 < n0) ? q + 1 : q - (n0 - 1)];
  const double py = pI[(jj + 1 < n1) ? q + n0 : q - (long long)(n1 - 1) * n0];
  const double pz = (kk + 1 < n2) ? pI[q + (long long)n0 * n1] : p_top[ii + (long long)n0 * jj];
  double un[3];
  un[0] = u[c] - (px - pc) * g.rdxu[0][I[0]];
  un[1] = u[g.sc + c] - (py - pc) * g.rdxu[1][I[1]];
  un[2] = u[2 * g.sc + c] - (pz - pc) * g.rdxu[2][I[2]];
  int img[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) img[a] = I[a] == 1 ? g.N[a] - 1 : (I[a] == g.N[a] - 2 ? 0 : -1);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    bool ok = true;
    long long cc = I[2] * g.sx[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const bool use = (m >> a) & 1;
      ok = ok && (!use || img[a] >= 0);
      cc += (long long)(use ? img[a] : I[a]) * g.sx[a];
    }
    if (ok) {
      u[cc] = un[0];
      u[cc + g.sc] = un[1];
      u[cc + 2 * g.sc] = un[2];
    }
  }
}

}  // namespace

extern "C" int ins_slab_fft_create(const int32_t np[3], const double h[3], int rank, int nranks, ins_slab_fft_t** out) {
  INS_REQUIRE(np && h && out, "null argument");
  INS_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank");
  INS_REQUIRE(np[2] % nranks == 0 && np[1] % nranks == 0, "nz and ny must be divisible by the number of ranks");
  for (int a = 0; a < 3; ++a) INS_REQUIRE(np[a] >= 2 && np[a] % 2 == 0, "Spectral psolver requires even number of volumes.");
  ins_slab_fft* S = new ins_slab_fft();
  for (int a = 0; a < 3; ++a) S->np[a] = np[a];
  S->rank = rank;
  S->nranks = nranks;
  S->nzl = np[2] / nranks;
  S->nyl = np[1] / nranks;
  S->kxn = np[0] / 2 + 1;
  const double om = h[0] * h[1] * h[2];
  auto symbol = [&](int a, int k) {  // pressure.jl:305-311
    const double sn = std::sin(M_PI * ((double)k / np[a]));
    return 4 * om * sn * sn / (h[a] * h[a]);
  };
  std::vector<double> ax(S->kxn), ay(S->nyl), az(np[2]), ayf(np[1]);
  for (int k = 0; k < S->kxn; ++k) ax[k] = symbol(0, k);
  S->ownfft = ins_ownfft_supported_slab(np);  // x and y: 2^m or 3 * 2^m; z: any even plane count (tridiagonal route), fused z kernel / rocFFT plan (transposes)
  S->cz = om / (h[2] * h[2]);
  S->kxs = S->ownfft ? ((S->kxn + 7) & ~7) : S->kxn;
  if (S->ownfft) {  // the own y pass leaves ky in digit-reversed storage order: slice the permuted symbol vector
    std::vector<double> full(np[1]);
    for (int k = 0; k < np[1]; ++k) full[k] = symbol(1, k);
    ins_ownfft_permute_symbol(np[1], full.data(), ayf.data());
    for (int k = 0; k < S->nyl; ++k) ay[k] = ayf[rank * S->nyl + k];
  } else {
    for (int k = 0; k < np[1]; ++k) ayf[k] = symbol(1, k);
    for (int k = 0; k < S->nyl; ++k) ay[k] = symbol(1, rank * S->nyl + k);
  }
  for (int k = 0; k < np[2]; ++k) az[k] = symbol(2, k);
  S->y3 = S->ownfft && ins_line3_supported(np[1]);
  if (S->y3) {
    std::vector<double> full(np[1]), p3(np[1]);
    for (int k = 0; k < np[1]; ++k) full[k] = symbol(1, k);
    ins_line3_permute_symbol(np[1], full.data(), p3.data());
    if (hipMalloc(&S->ay3, p3.size() * 8) != hipSuccess || hipMemcpy(S->ay3, p3.data(), p3.size() * 8, hipMemcpyHostToDevice) != hipSuccess) S->y3 = false;
  }
  bool ok = hipMalloc(&S->ax, ax.size() * 8) == hipSuccess && hipMalloc(&S->ay, ay.size() * 8) == hipSuccess &&
            hipMalloc(&S->az, az.size() * 8) == hipSuccess && hipMalloc(&S->ay_full, ayf.size() * 8) == hipSuccess &&
            hipMalloc(&S->ztri_bc, (size_t)4 * S->kxn * np[1] * 8) == hipSuccess;
  ok = ok && hipMemcpy(S->ay_full, ayf.data(), ayf.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(S->ax, ax.data(), ax.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(S->ay, ay.data(), ay.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(S->az, az.data(), az.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    ins_set_error("ins_slab_fft_create: symbol upload failed");
    ins_slab_fft_destroy(S);
    return INS_ERR_HIP;
  }
  // batched 2-D transforms over (y, x) for nzl planes; 1-D transforms along z (stride nyl*kxn) for nyl*kxn lines
  int n2[2] = {np[1], np[0]};
  int nz1[1] = {np[2]};
  const int lines = S->nyl * S->kxn;
  const bool zfused = ins_zsolve_supported(np[2]);  // then no rocFFT z plan is needed at all
  if (S->ownfft) {
    int rcv = ins_zsolve_twiddles(np[0], &S->tw_x);
    if (!rcv) rcv = ins_zsolve_twiddles(np[1], &S->tw_y);
    if (rcv) {
      ins_slab_fft_destroy(S);
      return rcv;
    }
  } else {
    int rcv = ins_fft_make_real_plans(&S->xy_fwd, &S->xy_inv, 2, n2, S->nzl);  // validated (ins_fftcheck.hip)
    if (rcv) {
      ins_slab_fft_destroy(S);
      return rcv;
    }
    S->plans = true;
  }
  if (!zfused) {
    hipfftResult r3 = hipfftPlanMany(&S->z_fwd, 1, nz1, nz1, lines, 1, nz1, lines, 1, HIPFFT_Z2Z, lines);
    if (r3 != HIPFFT_SUCCESS) {
      ins_set_error("ins_slab_fft_create: hipfftPlanMany(z) failed (%d)", (int)r3);
      ins_slab_fft_destroy(S);
      return INS_ERR_FFT;
    }
    S->has_z_plan = true;
  }
  if (zfused) {
    int rc = ins_zsolve_twiddles(np[2], &S->tw);
    if (rc) {
      ins_slab_fft_destroy(S);
      return rc;
    }
  }
  *out = S;
  return INS_OK;
}

extern "C" int ins_slab_fft_destroy(ins_slab_fft_t* S) {
  if (!S) return INS_OK;
  if (S->plans) {
    (void)hipfftDestroy(S->xy_fwd);
    (void)hipfftDestroy(S->xy_inv);
    if (S->has_z_plan) (void)hipfftDestroy(S->z_fwd);
    ins_fft_solver_released();
  }
  if (S->tw) (void)hipFree(S->tw);
  if (S->tw_x) (void)hipFree(S->tw_x);
  if (S->tw_y) (void)hipFree(S->tw_y);
  if (S->ax) (void)hipFree(S->ax);
  if (S->ay) (void)hipFree(S->ay);
  if (S->az) (void)hipFree(S->az);
  if (S->ay_full) (void)hipFree(S->ay_full);
  if (S->ay3) (void)hipFree(S->ay3);
  if (S->ztri_bc) (void)hipFree(S->ztri_bc);
  delete S;
  return INS_OK;
}

extern "C" int ins_slab_fft_sizes(const ins_slab_fft_t* S, int64_t* real_elems, int64_t* complex_elems) {
  INS_REQUIRE(S && real_elems && complex_elems, "null argument");
  *real_elems = (int64_t)S->np[0] * S->np[1] * S->nzl;
  *complex_elems = (int64_t)S->kxs * S->np[1] * S->nzl;  // >= kxn * nyl * nz, the transposed buffers of the FFT route
  return INS_OK;
}

static int slab_xy_forward(ins_slab_fft* S, double* pI, double* work, hipStream_t s) {
  if (S->ownfft) {
    int rc = ins_k_ownfft_xfwd(nullptr, pI, 0, work, S->np[0], S->np[1], S->nzl, S->tw_x, s);
    if (!rc) rc = ins_k_ownfft_y(work, S->kxn, S->np[1], S->nzl, S->tw_y, false, s);
    return rc;
  }
  INS_FFT_TRY(hipfftSetStream(S->xy_fwd, s));
  INS_FFT_TRY(hipfftExecD2Z(S->xy_fwd, pI, reinterpret_cast<hipfftDoubleComplex*>(work)));
  return INS_OK;
}

static int slab_xy_inverse(ins_slab_fft* S, double* work, double* pI, hipStream_t s) {
  if (S->ownfft) {
    int rc = ins_k_ownfft_y(work, S->kxn, S->np[1], S->nzl, S->tw_y, true, s);
    if (!rc) rc = ins_k_ownfft_xinv(work, pI, S->np[0], S->np[1], S->nzl, S->tw_x, s);
    return rc;
  }
  INS_FFT_TRY(hipfftSetStream(S->xy_inv, s));
  INS_FFT_TRY(hipfftExecZ2D(S->xy_inv, reinterpret_cast<hipfftDoubleComplex*>(work), pI));
  return INS_OK;
}

template <bool PACK>
static int slab_pack(ins_slab_fft* S, double* work, double* buf, int kx0, int kxc, hipStream_t s) {
  INS_REQUIRE(kx0 >= 0 && kxc >= 1 && kx0 + kxc <= S->kxn, "bad kx chunk");
  dim3 block(64, 4, 1), grid(cdiv(kxc, 64), cdiv(S->np[1], 4), S->nzl);
  hipLaunchKernelGGL(k_transpose_pack<PACK>, grid, block, 0, s, reinterpret_cast<double2*>(work), reinterpret_cast<double2*>(buf), S->kxn,
                     S->np[1], S->nyl, S->nzl, kx0, kxc);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

static int slab_solve_z(ins_slab_fft* S, double* buf, int kx0, int kxc, hipStream_t s) {
  INS_REQUIRE(kx0 >= 0 && kxc >= 1 && kx0 + kxc <= S->kxn, "bad kx chunk");
  const double inv_n = 1.0 / ((double)S->np[0] * S->np[1] * S->np[2]);
  if (S->tw)  // one fused pass instead of z-FFT + symbol + inverse z-FFT; a chunk is just a narrower [kz][kyl][kxc] array
    return ins_k_zsolve(buf, S->np[2], (long long)S->nyl * kxc, S->ax + kx0, kxc, S->ay, S->az, S->tw, inv_n, S->rank == 0 && kx0 == 0, s);
  INS_REQUIRE(kx0 == 0 && kxc == S->kxn, "kx chunks need a power-of-two nz (fused z kernel)");
  hipfftDoubleComplex* c = reinterpret_cast<hipfftDoubleComplex*>(buf);
  INS_FFT_TRY(hipfftSetStream(S->z_fwd, s));
  INS_FFT_TRY(hipfftExecZ2Z(S->z_fwd, c, c, HIPFFT_FORWARD));
  dim3 block(64, 4, 1), grid(cdiv(S->kxn, 64), cdiv(S->nyl, 4), S->np[2]);
  hipLaunchKernelGGL(k_symbol_slab, grid, block, 0, s, reinterpret_cast<double2*>(buf), S->ax, S->ay, S->az, S->kxn, S->nyl, S->np[2], inv_n,
                     S->rank == 0);
  INS_LAUNCH_CHECK();
  INS_FFT_TRY(hipfftExecZ2Z(S->z_fwd, c, c, HIPFFT_BACKWARD));
  return INS_OK;
}

extern "C" int ins_slab_fft_forward_xy(ins_slab_fft_t* S, double* pI, double* work, double* sendbuf, void* stream) {
  INS_REQUIRE(S && pI && work && sendbuf, "null argument");
  int rc = slab_xy_forward(S, pI, work, as_stream(stream));
  if (rc) return rc;
  return slab_pack<true>(S, work, sendbuf, 0, S->kxn, as_stream(stream));
}

extern "C" int ins_slab_fft_solve_z(ins_slab_fft_t* S, double* buf, void* stream) {
  INS_REQUIRE(S && buf, "null argument");
  return slab_solve_z(S, buf, 0, S->kxn, as_stream(stream));
}

extern "C" int ins_slab_fft_inverse_xy(ins_slab_fft_t* S, double* recvbuf, double* work, double* pI, void* stream) {
  INS_REQUIRE(S && recvbuf && work && pI, "null argument");
  int rc = slab_pack<false>(S, work, recvbuf, 0, S->kxn, as_stream(stream));
  if (rc) return rc;
  return slab_xy_inverse(S, work, pI, as_stream(stream));
}

// ---- kx-chunked variants: the transposes of different chunks can overlap each other and the z solve -------------
extern "C" int ins_slab_fft_xy_forward_only(ins_slab_fft_t* S, double* pI, double* work, void* stream) {
  INS_REQUIRE(S && pI && work, "null argument");
  return slab_xy_forward(S, pI, work, as_stream(stream));
}
extern "C" int ins_slab_fft_pack_chunk(ins_slab_fft_t* S, double* work, double* sendbuf, int kx0, int kxc, void* stream) {
  INS_REQUIRE(S && work && sendbuf, "null argument");
  return slab_pack<true>(S, work, sendbuf, kx0, kxc, as_stream(stream));
}
extern "C" int ins_slab_fft_solve_z_chunk(ins_slab_fft_t* S, double* buf, int kx0, int kxc, void* stream) {
  INS_REQUIRE(S && buf, "null argument");
  return slab_solve_z(S, buf, kx0, kxc, as_stream(stream));
}
extern "C" int ins_slab_fft_unpack_chunk(ins_slab_fft_t* S, double* recvbuf, double* work, int kx0, int kxc, void* stream) {
  INS_REQUIRE(S && recvbuf && work, "null argument");
  return slab_pack<false>(S, work, recvbuf, kx0, kxc, as_stream(stream));
}
extern "C" int ins_slab_fft_xy_inverse_only(ins_slab_fft_t* S, double* work, double* pI, void* stream) {
  INS_REQUIRE(S && work && pI, "null argument");
  return slab_xy_inverse(S, work, pI, as_stream(stream));
}
// ---- power-of-two boxes: own passes write / read the packed exchange buffer directly (no pack / unpack passes), and the
// right-hand side can be formed inside the x pass straight from the slab's u (K2 fused) -------------------------------
extern "C" int ins_slab_fft_forward_packed(ins_slab_fft_t* S, const ins_grid_t* G, const double* src, int from_u, double* work, double* sendbuf,
                                           int cw, void* stream) {
  INS_REQUIRE(S && src && work && sendbuf, "null argument");
  INS_REQUIRE(S->ownfft, "packed passes need x and y sides of 2^m or 3 * 2^m");
  INS_REQUIRE(cw >= 1 && cw <= S->kxn, "bad chunk width");
  if (from_u) {
    INS_REQUIRE(G && G->g.D == 3 && G->g.N[0] == S->np[0] + 2 && G->g.N[1] == S->np[1] + 2 && G->g.N[2] == S->nzl + 2, "grid does not match the slab");
  }
  hipStream_t s = as_stream(stream);
  int rc = ins_k_ownfft_xfwd(from_u ? G : nullptr, src, from_u ? 2 : 0, work, S->np[0], S->np[1], S->nzl, S->tw_x, s);
  if (rc) return rc;
  return ins_k_ownfft_y_packed(work, sendbuf, S->kxn, S->np[1], S->nzl, S->nyl, cw, S->tw_y, false, s);
}

extern "C" int ins_slab_fft_inverse_packed(ins_slab_fft_t* S, double* recvbuf, double* work, double* pI, int cw, void* stream) {
  INS_REQUIRE(S && recvbuf && work && pI, "null argument");
  INS_REQUIRE(S->ownfft, "packed passes need x and y sides of 2^m or 3 * 2^m");
  INS_REQUIRE(cw >= 1 && cw <= S->kxn, "bad chunk width");
  hipStream_t s = as_stream(stream);
  int rc = ins_k_ownfft_y_packed(work, recvbuf, S->kxn, S->np[1], S->nzl, S->nyl, cw, S->tw_y, true, s);
  if (rc) return rc;
  return ins_k_ownfft_xinv(work, pI, S->np[0], S->np[1], S->nzl, S->tw_x, s);
}

extern "C" int ins_slab_fft_is_own(const ins_slab_fft_t* S) { return S && S->ownfft; }

// ---- transpose-free solve: the z direction as distributed tridiagonal systems (ins_ztri.hip) ----------------------------------
// The kxn·ny lines can be cut into `nchunks` ranges so that the gather of one range travels while the next range is swept.
// Per rank the exchange buffer of a range holds [yF (cnt) | yL (cnt) | the singular line's nzl local values (range 0 only)], complex.
static void ztri_range(const ins_slab_fft* S, int c, int nchunks, long long* lo, long long* cnt) {
  const long long lines = (long long)S->kxn * S->np[1];
  long long per = (lines + nchunks - 1) / nchunks;
  per = (per + 127) / 128 * 128;  // whole 256-thread workgroups of re/im components
  *lo = std::min(lines, (long long)c * per);
  *cnt = std::min(per, lines - *lo);
}

extern "C" int ins_slab_ztri_chunk(const ins_slab_fft_t* S, int c, int nchunks, int64_t* line_lo, int64_t* line_cnt, int64_t* edge_doubles) {
  INS_REQUIRE(S && nchunks >= 1 && c >= 0 && c < nchunks, "bad chunk");
  long long lo, cnt;
  ztri_range(S, c, nchunks, &lo, &cnt);
  if (line_lo) *line_lo = lo;
  if (line_cnt) *line_cnt = cnt;
  if (edge_doubles) *edge_doubles = 2 * (2 * cnt + (c == 0 ? S->nzl : 0));
  return INS_OK;
}

extern "C" int ins_slab_ztri_edge_elems(const ins_slab_fft_t* S, int64_t* doubles) {
  INS_REQUIRE(S && doubles, "null argument");
  return ins_slab_ztri_chunk(S, 0, 1, nullptr, nullptr, doubles);
}

// y pass of the tridiagonal route (no transposes: every rank holds all ky of its planes, so only the symbol has to know the storage order)
static int slab_y(ins_slab_fft_t* S, double* work, bool inverse, hipStream_t s) {
  if (S->y3) return ins_k_line3_y(work, S->kxn, S->np[1], S->nzl, S->tw_y, inverse, s, S->kxs);
  return ins_k_ownfft_y(work, S->kxn, S->np[1], S->nzl, S->tw_y, inverse, s, S->kxs);
}

// (x, y) forward transforms of the local planes into `work` (from_u = 1: Ω·div(u) formed inside the x pass, own passes only; 2: the x
// pass was done by ins_slab_xfwd_planes; 0: src = pI).
extern "C" int ins_slab_ztri_transform(ins_slab_fft_t* S, const ins_grid_t* G, const double* src, int from_u, double* work, void* stream) {
  INS_REQUIRE(S && src && work, "null argument");
  INS_REQUIRE(S->nzl >= 2, "the tridiagonal z solve needs >= 2 local planes");
  hipStream_t s = as_stream(stream);
  int rc;
  if (from_u == 2) {  // x pass already done plane range by plane range (ins_slab_xfwd_planes)
    INS_REQUIRE(S->ownfft, "forming the right-hand side inside the x pass needs x and y sides of 2^m or 3 * 2^m");
    return slab_y(S, work, false, s);
  }
  if (from_u) {
    INS_REQUIRE(S->ownfft, "forming the right-hand side inside the x pass needs x and y sides of 2^m or 3 * 2^m");
    INS_REQUIRE(G && G->g.D == 3 && G->g.N[0] == S->np[0] + 2 && G->g.N[1] == S->np[1] + 2 && G->g.N[2] == S->nzl + 2, "grid does not match the slab");
    if ((rc = ins_k_ownfft_xfwd(G, src, 2, work, S->np[0], S->np[1], S->nzl, S->tw_x, s, S->kxs))) return rc;
    return slab_y(S, work, false, s);
  }
  if (S->ownfft) {
    if ((rc = ins_k_ownfft_xfwd(nullptr, src, 0, work, S->np[0], S->np[1], S->nzl, S->tw_x, s, S->kxs))) return rc;
    return slab_y(S, work, false, s);
  }
  return slab_xy_forward(S, const_cast<double*>(src), work, s);
}

// forward elimination along z of line range c (in place on `work`) and this rank's interface data of the range -> edge
extern "C" int ins_slab_ztri_sweep_forward(ins_slab_fft_t* S, double* work, double* edge, int c, int nchunks, void* stream) {
  INS_REQUIRE(S && work && edge && nchunks >= 1 && c >= 0 && c < nchunks, "bad argument");
  long long lo, cnt;
  ztri_range(S, c, nchunks, &lo, &cnt);
  const double scale = -1.0 / ((double)S->np[0] * S->np[1]);
  return ins_k_ztri_forward(work, S->kxn, S->kxs, S->np[1], S->nzl, S->nranks, S->rank, S->ax, S->y3 ? S->ay3 : S->ay_full, S->cz, scale, edge, lo, cnt, as_stream(stream));
}

// edges_all = the gathered edge buffers of range c (rank-major): interface solve + back substitution of the range
extern "C" int ins_slab_ztri_sweep_backward(ins_slab_fft_t* S, double* work, const double* edges_all, int c, int nchunks, void* stream) {
  INS_REQUIRE(S && work && edges_all && nchunks >= 1 && c >= 0 && c < nchunks, "bad argument");
  long long lo, cnt;
  ztri_range(S, c, nchunks, &lo, &cnt);
  const long long stride = 2 * cnt + (c == 0 ? S->nzl : 0);
  return ins_k_ztri_finish(work, S->kxn, S->kxs, S->np[1], S->nzl, S->nranks, S->rank, S->ax, S->y3 ? S->ay3 : S->ay_full, S->cz, edges_all, stride,
                           S->ztri_bc + 4 * lo, lo, cnt, as_stream(stream));
}

// inverse y / x transforms of `work` -> pI
extern "C" int ins_slab_ztri_inverse(ins_slab_fft_t* S, double* work, double* pI, void* stream) {
  INS_REQUIRE(S && work && pI, "null argument");
  hipStream_t s = as_stream(stream);
  if (!S->ownfft) return slab_xy_inverse(S, work, pI, s);
  int rc = slab_y(S, work, true, s);
  if (rc) return rc;
  return ins_k_ownfft_xinv(work, pI, S->np[0], S->np[1], S->nzl, S->tw_x, s, S->kxs);
}

// one-range forms
extern "C" int ins_slab_ztri_forward(ins_slab_fft_t* S, const ins_grid_t* G, const double* src, int from_u, double* work, double* edge, void* stream) {
  INS_REQUIRE(edge, "null argument");
  int rc = ins_slab_ztri_transform(S, G, src, from_u, work, stream);
  return rc ? rc : ins_slab_ztri_sweep_forward(S, work, edge, 0, 1, stream);
}

extern "C" int ins_slab_ztri_finish(ins_slab_fft_t* S, double* work, const double* edges_all, double* pI, void* stream) {
  INS_REQUIRE(pI, "null argument");
  int rc = ins_slab_ztri_sweep_backward(S, work, edges_all, 0, 1, stream);
  return rc ? rc : ins_slab_ztri_inverse(S, work, pI, stream);
}

// The x pass of ins_slab_ztri_forward (from_u = 1) for local planes [kz0, kz0 + nkz) only: planes >= 1 need no ghost plane of u, so they
// can be transformed while the w plane below the slab is still travelling; call ins_slab_ztri_forward with from_u = 2 afterwards.
extern "C" int ins_slab_xfwd_planes(ins_slab_fft_t* S, const ins_grid_t* G, const double* u, double* work, int kz0, int nkz, void* stream) {
  INS_REQUIRE(S && G && u && work, "null argument");
  INS_REQUIRE(S->ownfft, "forming the right-hand side inside the x pass needs x and y sides of 2^m or 3 * 2^m");
  INS_REQUIRE(G->g.D == 3 && G->g.N[0] == S->np[0] + 2 && G->g.N[1] == S->np[1] + 2 && G->g.N[2] == S->nzl + 2, "grid does not match the slab");
  INS_REQUIRE(kz0 >= 0 && nkz >= 0 && kz0 + nkz <= S->nzl, "bad plane range");
  if (nkz == 0) return INS_OK;
  return ins_k_ownfft_xfwd(G, u, 2, work, S->np[0], S->np[1], nkz, S->tw_x, as_stream(stream), S->kxs, kz0);
}

/* 1 when kx chunks are supported (power-of-two nz -> fused z kernel). */
extern "C" int ins_slab_fft_can_chunk(const ins_slab_fft_t* S) { return S && S->tw != nullptr; }

static int check_slab_grid(const ins_grid* G) {
  INS_REQUIRE(G, "null argument");
  const GridDev& g = G->g;
  INS_REQUIRE(g.D == 3, "slab kernels are 3-D");
  INS_REQUIRE(g.bc[0][0] == INS_BC_PERIODIC && g.bc[1][0] == INS_BC_PERIODIC, "slab kernels need periodic x and y");
  INS_REQUIRE(g.bc[2][0] == INS_BC_HALO && g.bc[2][1] == INS_BC_HALO, "slab grid must mark z as INS_BC_HALO");
  INS_REQUIRE(G->all_dof, "slab grid must be all-DOF");
  return INS_OK;
}

extern "C" int ins_slab_divergence_f64(const ins_grid_t* G, const double* u, double* pI, void* stream) {
  int rc = check_slab_grid(G);
  if (rc) return rc;
  INS_REQUIRE(u && pI, "null argument");
  const GridDev& g = G->g;
  const int n0 = g.N[0] - 2, n1 = g.N[1] - 2, n2 = g.N[2] - 2;
  hipLaunchKernelGGL(k_div_slab, dim3(cdiv(n0, 64), cdiv(n1, 4), n2), dim3(64, 4, 1), 0, as_stream(stream), g, u, pI, n0, n1);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

extern "C" int ins_slab_applypressure_f64(const ins_grid_t* G, double* u, const double* pI, const double* p_top, void* stream) {
  int rc = check_slab_grid(G);
  if (rc) return rc;
  INS_REQUIRE(u && pI && p_top, "null argument");
  const GridDev& g = G->g;
  const int n0 = g.N[0] - 2, n1 = g.N[1] - 2, n2 = g.N[2] - 2;
  hipLaunchKernelGGL(k_grad_slab, dim3(cdiv(n0, 64), cdiv(n1, 4), n2), dim3(64, 4, 1), 0, as_stream(stream), g, u, pI, p_top, n0, n1, n2);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// K1 + K6 on any all-DOF 3-D grid (a slab, or a whole periodic box): k_out = momentum(u_in) when k_out != NULL,
// ustar(interior) = ustart + Σ_q coefs[q] ks[q] + coef_self * momentum(u_in); ustart == NULL means ustart = u_in.
// Stage kernel with the PREVIOUS stage's projection applied in registers (k_momentum_flux<..., CORR = 2>): `ustar_prev` is that
// stage's uncorrected velocity with valid z-ghost planes, `p_ext` = [1 plane below | nzl local planes | 2 planes above] of its
// pressure (unpadded in x, y).  Exactly-uniform slabs only.
static int stage_momentum_corr(const ins_grid_t* G, double visc, const double* ustar_prev, const double* p_ext, double* k_out, const double* ustart,
                               double* ustar, int nterms, const double* coefs, const double* const* ks, double coef_self, int part, void* stream,
                               double c0m1 = 0.0, double self_in = 0.0, double* ustart_out = nullptr);

extern "C" int ins_stage_momentum_corr_f64(const ins_grid_t* G, double visc, const double* ustar_prev, const double* p_ext, double* k_out,
                                           const double* ustart, double* ustar, int nterms, const double* coefs, const double* const* ks,
                                           double coef_self, void* stream) {
  return stage_momentum_corr(G, visc, ustar_prev, p_ext, k_out, ustart, ustar, nterms, coefs, ks, coef_self, 0, stream);
}

extern "C" int ins_stage_momentum_corr_part_f64(const ins_grid_t* G, double visc, const double* ustar_prev, const double* p_ext, double* k_out,
                                                const double* ustart, double* ustar, int nterms, const double* coefs, const double* const* ks,
                                                double coef_self, double c0m1, double self_in, double* ustart_out, int part, void* stream) {
  INS_REQUIRE(part >= 0 && part <= 2, "part must be 0 (all), 1 (interior z-chunks) or 2 (boundary z-chunks)");
  INS_REQUIRE((self_in == 0.0 && !ustart_out && ustart) || ins_flux64_supported(G), "self_in / ustart_out need the 64-wide stage kernel (ins_flux64.hip)");
  INS_REQUIRE(ustart || (nterms == 0 && self_in == 0.0), "without ustart (chained first stage) there are no earlier terms");
  return stage_momentum_corr(G, visc, ustar_prev, p_ext, k_out, ustart, ustar, nterms, coefs, ks, coef_self, part, stream, c0m1, self_in, ustart_out);
}

static int stage_momentum_corr(const ins_grid_t* G, double visc, const double* ustar_prev, const double* p_ext, double* k_out, const double* ustart,
                               double* ustar, int nterms, const double* coefs, const double* const* ks, double coef_self, int part, void* stream,
                               double c0m1, double self_in, double* ustart_out) {
  int rc = check_slab_grid(G);
  if (rc) return rc;
  INS_REQUIRE(ustar_prev && p_ext && ustar, "null argument");
  INS_REQUIRE(G->uniform_exact, "in-kernel correction needs an exactly uniform grid");
  INS_REQUIRE(G->g.N[2] >= 4, "slab needs at least two local planes");
  INS_REQUIRE(nterms >= 0 && nterms <= INS_MAX_STAGES && (nterms == 0 || (coefs && ks)), "bad stage terms");
  INS_REQUIRE(ustar != ustar_prev, "the stage velocity cannot overwrite the stencil input");
  RkEpi epi;
  memset(&epi, 0, sizeof(epi));
  epi.n = nterms;
  for (int q = 0; q < nterms; ++q) {
    epi.coef[q] = coefs[q];
    epi.k[q] = ks[q];
  }
  epi.c0m1 = c0m1;
  epi.self_in = self_in;
  epi.ustart_out = ustart_out;
  epi.coef_self = coef_self;
  epi.ustart = ustart;
  epi.ustar = ustar;
  epi.write_k = k_out != nullptr;
  static double* dummy = nullptr;
  return ins_k_momentum_rk_fused_corr_slab(G, visc, ustar_prev, p_ext, k_out ? k_out : dummy, epi, as_stream(stream), part);
}

extern "C" int ins_stage_momentum_f64(const ins_grid_t* G, double visc, const double* u_in, double* k_out, const double* ustart,
                                      double* ustar, int nterms, const double* coefs, const double* const* ks, double coef_self,
                                      void* stream) {
  INS_REQUIRE(G && u_in && ustar, "null argument");
  INS_REQUIRE(G->g.D == 3 && G->all_dof, "stage kernel needs a 3-D all-DOF grid (periodic box or slab)");
  INS_REQUIRE(nterms >= 0 && nterms <= INS_MAX_STAGES, "too many stage terms");
  INS_REQUIRE(nterms == 0 || (coefs && ks), "null stage terms");
  INS_REQUIRE(ustar != u_in, "the stage velocity cannot overwrite the stencil input");
  RkEpi epi;
  memset(&epi, 0, sizeof(epi));
  epi.n = nterms;
  for (int q = 0; q < nterms; ++q) {
    epi.coef[q] = coefs[q];
    epi.k[q] = ks[q];
  }
  epi.coef_self = coef_self;
  epi.ustart = ustart;
  epi.ustar = ustar;
  epi.write_k = k_out != nullptr;
  static double* dummy = nullptr;
  return ins_k_momentum_rk_fused(G, visc, u_in, k_out ? k_out : dummy, epi, as_stream(stream));
}
