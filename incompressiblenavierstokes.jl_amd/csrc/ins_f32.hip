// The `_f32` entry-point family ("leaves `_f32` as a suffix-parallel family"; the reference is generic in the element
// type and recommends single precision on GPUs, docs/src/manual/precision.md:3-16; examples/DecayingTurbulence3D.jl:16 runs T = Float32).
//
// Scope: all-periodic uniform boxes, 2-D and 3-D — the configuration the reference's own Float32 example runs (spectral pressure solver).
//   K1 (momentum!)          : the 64-outputs-per-wavefront stage kernel of ins_flux64.hip instantiated for float (wide 3-D boxes: half
//                             the bytes of the fp64 kernel at the same memory rate), a plain one-cell-per-work-item kernel elsewhere;
//   spectral projection     : 3-D power-of-two boxes: the library's five fused fp64 passes, the right-hand side Ω·div(u) formed from the float field
//                             inside the x pass (ins_fft.hip, SRC = 5) — a mixed-precision projection whose pressure is more accurate than a
//                             float solve; other boxes: Ω·div(u) -> hipFFT R2C -> symbol -> C2R; then padded p with periodic ghosts,
//                             u -= ∇p, periodic ghosts of u;
//   explicit RK             : the stage loop of step_explicit_runge_kutta.jl:4-59 on those kernels, the stage combination as K1's epilogue
//                             on wide 3-D boxes.
// Arrays are the reference layout with Float32 elements.  The grid handle is the fp64 one (metrics are computed in double and rounded
// once per launch).  Non-periodic or stretched grids return INS_ERR_UNSUPPORTED: no silent fp64 fallback.
#include <cmath>
#include <cstdlib>

#include "ins_internal.h"

bool ins_flux64_supported(const ins_grid* G);
bool ins_k_spectral_own3d(const ins_poisson* ps);
int ins_k_spectral_solve_from_u32(ins_poisson* ps, const float* u32, hipStream_t s);
const double* ins_k_spectral_pI(const ins_poisson* ps);
bool ins_k_spectral_own3d_f32(const ins_poisson* ps);
int ins_k_spectral_solve_f32(ins_poisson* ps, const float* u32, float* pI32, float* phat32, int kxs32, const float* twx, const float* twy,
                             const float* twz, hipStream_t s);
int ins_zsolve_twiddles_f32(int nz, float** out);
int ins_k_flux64_f32(const ins_grid* G, double visc, const float* u, float* F, const RkEpi* epi, const float* pI, int corr_mode, hipStream_t s,
                     int part = 0);
// any other grid (walls, stretched spacings, 2-D or 3-D): csrc/ins_f32g.hip
int ins_k32g_apply_bc_u(const ins_grid* G, float* u, hipStream_t s);
int ins_k32g_apply_bc_p(const ins_grid* G, float* p, hipStream_t s);
int ins_k32g_momentum(const ins_grid* G, float visc, const float* u, float* F, hipStream_t s);
int ins_k32g_divergence(const ins_grid* G, const float* u, float* div, hipStream_t s);
int ins_k32g_solve(const ins_grid* G, ins_poisson* ps64, double* p64, float* p, hipStream_t s);
int ins_k32g_project(const ins_grid* G, ins_poisson* ps64, double* p64, float* u, float* p, hipStream_t s);

struct ins_poisson32 {
  const ins_grid* grid = nullptr;
  hipfftHandle fwd = 0, inv = 0;
  bool plans = false;
  float* pI = nullptr;
  hipfftComplex* phat = nullptr;
  float* ahat[3] = {nullptr, nullptr, nullptr};
  int np[3] = {1, 1, 1}, kmax[3] = {1, 1, 1};
  // 3-D power-of-two boxes: project! solves its pressure equation with the library's fused fp64 passes (right-hand side formed from the
  // float field inside the x pass, five passes instead of hipFFT's 3-D plans + separate kernels; the pressure is MORE accurate than a float
  // solve would give).  512^3 RK44 step: 26.9 ms with hipFFT's float plans -> see DESIGN.md §5.
  ins_poisson* ps64 = nullptr;
  // sides the three-pass z kernel takes (192, 256, 384, 512 planes): the same five passes on FLOAT2 spectra — half the bytes of every pass; ps64
  // then only supplies the symbol vectors (INS_F32_FP64_SPECTRA=1 keeps the fp64 passes)
  bool own32 = false;
  float* phat32 = nullptr;
  float* tw32[3] = {nullptr, nullptr, nullptr};
  int kxs32 = 0;
  // ins_poisson_wrap_f32: a Float32 solver around a caller-owned fp64 solver of any kind (walls / stretched grids: psolver_direct, psolver_cg)
  ins_poisson* wrap64 = nullptr;
  double* p64 = nullptr;  // padded fp64 right-hand side / solution
  float* div32 = nullptr; // padded scratch of the divergence diagnostic
};

struct ins_rk32 {
  const ins_grid* grid = nullptr;
  ins_poisson32* ps = nullptr;
  int nstage = 0;
  std::vector<double> A, c;
  std::vector<float*> ku;
  float* ub[2] = {nullptr, nullptr};
  float* p = nullptr;
  float* pu = nullptr;  // unpadded float copy of the stage pressure (in-register correction of the next stage's stencil kernel)
  float* ustart = nullptr;  // chained steps: the corrected start field of steps 2..K (the caller's array holds the uncorrected last stage velocity then)
};

namespace {

struct Box32 {
  int D, N[3], n[3];
  long long sx[3], sc;
  float rh[3];  // 1/h
  float om;     // cell volume
};

Box32 box_of(const ins_grid* G) {
  Box32 b;
  const GridDev& g = G->g;
  b.D = g.D;
  double om = 1.0;
  for (int a = 0; a < 3; ++a) {
    b.N[a] = a < g.D ? g.N[a] : 1;
    b.n[a] = a < g.D ? g.N[a] - 2 : 1;
    b.sx[a] = g.sx[a];
    b.rh[a] = a < g.D ? (float)(1.0 / G->h[a]) : 0.f;
    if (a < g.D) om *= G->h[a];
  }
  b.sc = g.sc;
  b.om = (float)om;
  return b;
}

int require_periodic_uniform(const ins_grid* G, const char* what) {
  if (!(G->all_periodic && G->uniform)) {
    ins_set_error("%s: all-periodic uniform boxes only (other grids: ins_poisson_wrap_f32 around psolver_direct / psolver_cg)", what);
    return INS_ERR_UNSUPPORTED;
  }
  return INS_OK;
}
inline bool periodic_uniform(const ins_grid* G) { return G->all_periodic && G->uniform; }

// ghost volumes of `ncomp` components: cell 0 <- cell N-2, cell N-1 <- cell 1, direction after direction (corners consistent),
// boundary_conditions.jl:276-288 / 306-318
__global__ __launch_bounds__(256) void k32_bc_periodic(Box32 b, float* __restrict__ f, int ncomp, int dir) {
  const int d1 = (dir + 1) % 3, d2 = (dir + 2) % 3;
  const long long plane = (long long)b.N[d1] * b.N[d2];
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < plane * ncomp; t += (long long)gridDim.x * 256) {
    const int c = (int)(t / plane);
    const long long r = t % plane;
    const int i1 = (int)(r % b.N[d1]), i2 = (int)(r / b.N[d1]);
    float* base = f + c * b.sc + i1 * b.sx[d1] + i2 * b.sx[d2];
    base[0] = base[(long long)(b.N[dir] - 2) * b.sx[dir]];
    base[(long long)(b.N[dir] - 1) * b.sx[dir]] = base[b.sx[dir]];
  }
}

int bc_periodic(const ins_grid* G, float* f, int ncomp, hipStream_t s) {
  const Box32 b = box_of(G);
  for (int dir = 0; dir < b.D; ++dir) {
    const long long work = (long long)b.N[(dir + 1) % 3] * b.N[(dir + 2) % 3] * ncomp;
    hipLaunchKernelGGL(k32_bc_periodic, dim3((unsigned)std::min<long long>((work + 255) / 256, 4096)), dim3(256), 0, s, b, f, ncomp, dir);
  }
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// momentum! on a uniform periodic box, one cell per work-item (operators.jl:647-690 with every weight 1/2 and constant spacings);
// u needs valid ghosts.  FUSE-free: the wide 3-D boxes take ins_flux64.hip's kernel instead.
template <int D>
__global__ __launch_bounds__(256) void k32_momentum(Box32 b, float visc, const float* __restrict__ u, float* __restrict__ F) {
  const int i = 1 + blockIdx.x * 64 + threadIdx.x, j = 1 + blockIdx.y * 4 + threadIdx.y, k = D == 3 ? 1 + (int)blockIdx.z : 0;
  if (i > b.n[0] || j > b.n[1]) return;
  const long long c = i + j * b.sx[1] + k * b.sx[2];
#pragma unroll
  for (int al = 0; al < D; ++al) {
    const float* ua = u + al * b.sc;
    float f = 0.f;
#pragma unroll
    for (int be = 0; be < D; ++be) {
      const float* ub = u + be * b.sc;
      const long long sb = b.sx[be], sa = b.sx[al];
      const float uc = ua[c], um = ua[c - sb], up = ua[c + sb];
      const float uab1 = 0.5f * (um + uc), uab2 = 0.5f * (uc + up);
      const float uba1 = 0.5f * (ub[c - sb] + ub[c - sb + sa]), uba2 = 0.5f * (ub[c] + ub[c + sa]);
      const float d1 = (uc - um) * b.rh[be], d2 = (up - uc) * b.rh[be];
      f += (visc * (d2 - d1) - (uab2 * uba2 - uab1 * uba1)) * b.rh[be];
    }
    F[al * b.sc + c] = f;
  }
}

// pI = Ω div(u) with the periodic image of u[I - e_a] (no ghost fill of u needed)        operators.jl:117-125, 81-95, pressure.jl:320
template <int D>
__global__ __launch_bounds__(256) void k32_div(Box32 b, const float* __restrict__ u, float* __restrict__ pI) {
  const int ii = blockIdx.x * 64 + threadIdx.x, jj = blockIdx.y * 4 + threadIdx.y, kk = D == 3 ? (int)blockIdx.z : 0;
  if (ii >= b.n[0] || jj >= b.n[1]) return;
  const int I[3] = {ii + 1, jj + 1, D == 3 ? kk + 1 : 0};
  const long long c = I[0] + I[1] * b.sx[1] + I[2] * b.sx[2];
  float d = 0.f;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const float* ua = u + a * b.sc;
    const long long cm = I[a] == 1 ? c + (long long)(b.N[a] - 3) * b.sx[a] : c - b.sx[a];
    d += (ua[c] - ua[cm]) * b.rh[a];
  }
  pI[ii + (long long)b.n[0] * (jj + (long long)b.n[1] * kk)] = d * b.om;
}

// phat = -phat / (ax + ay + az) / prod(Np), phat[0] = 0                                  pressure.jl:326-341
template <int D>
__global__ __launch_bounds__(256) void k32_symbol(hipfftComplex* __restrict__ phat, const float* __restrict__ ax, const float* __restrict__ ay,
                                                  const float* __restrict__ az, int k0, int k1, float inv_n) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= k0 || j >= k1) return;
  const long long q = i + (long long)k0 * (j + (long long)k1 * k);
  float den = ax[i] + ay[j];
  if (D == 3) den += az[k];
  const float sc = (i == 0 && j == 0 && k == 0) ? 0.f : -inv_n / den;
  hipfftComplex v = phat[q];
  v.x *= sc;
  v.y *= sc;
  phat[q] = v;
}

// p (padded, with periodic ghosts) <- pI; one work-item per padded volume                pressure.jl:347, boundary_conditions.jl:306-318
template <int D, typename S = float>
__global__ __launch_bounds__(256) void k32_pad(Box32 b, const S* __restrict__ pI, float* __restrict__ p) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= b.N[0] || j >= b.N[1]) return;
  auto w = [](int I, int n) { return I == 0 ? n - 1 : (I == n + 1 ? 0 : I - 1); };
  const long long q = w(i, b.n[0]) + (long long)b.n[0] * (w(j, b.n[1]) + (long long)b.n[1] * (D == 3 ? w(k, b.n[2]) : 0));
  p[i + j * b.sx[1] + k * b.sx[2]] = (float)pI[q];
}

// u[I, a] -= (p[I + e_a] - p[I]) / h_a on the interior                                    operators.jl:225-233
template <int D>
__global__ __launch_bounds__(256) void k32_applypressure(Box32 b, float* __restrict__ u, const float* __restrict__ p) {
  const int i = 1 + blockIdx.x * 64 + threadIdx.x, j = 1 + blockIdx.y * 4 + threadIdx.y, k = D == 3 ? 1 + (int)blockIdx.z : 0;
  if (i > b.n[0] || j > b.n[1]) return;
  const long long c = i + j * b.sx[1] + k * b.sx[2];
  const float pc = p[c];
#pragma unroll
  for (int a = 0; a < D; ++a) u[a * b.sc + c] -= (p[c + b.sx[a]] - pc) * b.rh[a];
}

// The three kernels above as one pass (3-D): every interior volume reads its own velocity and p at itself and its three upper periodic
// images from the unpadded solver output, and writes u - ∇p and p to itself and to the ghost volumes it is the image of (the fp64 path's
// k_grad_ghost3, ins_poisson.hip).  512^3: pad 0.39 + gradient 0.82 + six ghost-strip kernels -> one pass.
template <typename S>
__global__ __launch_bounds__(256) void k32_grad_ghost3(Box32 b, float* __restrict__ u, float* __restrict__ p, const S* __restrict__ pI) {
  const int ii = blockIdx.x * 64 + threadIdx.x, jj = blockIdx.y * 4 + threadIdx.y, kk = blockIdx.z;
  if (ii >= b.n[0] || jj >= b.n[1]) return;
  const int w[3] = {ii, jj, kk};
  const int I[3] = {ii + 1, jj + 1, kk + 1};
  const long long qs[3] = {1, b.n[0], (long long)b.n[0] * b.n[1]};
  const long long q = ii + qs[1] * jj + qs[2] * kk;
  const long long c = I[0] + I[1] * b.sx[1] + I[2] * b.sx[2];
  const float pc = (float)pI[q];
  float un[3];
  int img[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const long long qn = (w[a] + 1 < b.n[a]) ? q + qs[a] : q - (long long)(b.n[a] - 1) * qs[a];
    un[a] = u[a * b.sc + c] - ((float)pI[qn] - pc) * b.rh[a];
    img[a] = I[a] == 1 ? b.N[a] - 1 : (I[a] == b.N[a] - 2 ? 0 : -1);
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    bool ok = true;
    long long cc = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const bool use = (m >> a) & 1;
      ok = ok && (!use || img[a] >= 0);
      cc += (long long)(use ? img[a] : I[a]) * b.sx[a];
    }
    if (ok) {
      u[cc] = un[0];
      u[cc + b.sc] = un[1];
      u[cc + 2 * b.sc] = un[2];
      p[cc] = pc;
    }
  }
}

// out = base + Σ coef_q k_q                                                             step_explicit_runge_kutta.jl:35-38
struct Comb32 {
  int n;
  float coef[INS_MAX_STAGES + 1];
  const float* k[INS_MAX_STAGES + 1];
};
__global__ __launch_bounds__(256) void k32_combine(long long n, const float* __restrict__ base, float* __restrict__ out, Comb32 cb) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    float v = base[t];
    for (int q = 0; q < cb.n; ++q) v += cb.coef[q] * cb.k[q][t];
    out[t] = v;
  }
}

__global__ __launch_bounds__(256) void k32_cvt(long long n, const double* __restrict__ a, float* __restrict__ b) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) b[t] = (float)a[t];
}

dim3 grid_over(const Box32& b, bool padded) {
  const int e0 = padded ? b.N[0] : b.n[0], e1 = padded ? b.N[1] : b.n[1], e2 = b.D == 3 ? (padded ? b.N[2] : b.n[2]) : 1;
  return dim3(cdiv(e0, 64), cdiv(e1, 4), e2);
}

}  // namespace

extern "C" int ins_apply_bc_u_f32(const ins_grid_t* G, float* u, void* stream) {
  INS_REQUIRE(G && u, "null argument");
  if (!periodic_uniform(G)) return ins_k32g_apply_bc_u(G, u, as_stream(stream));
  return bc_periodic(G, u, G->g.D, as_stream(stream));
}

extern "C" int ins_apply_bc_p_f32(const ins_grid_t* G, float* p, void* stream) {
  INS_REQUIRE(G && p, "null argument");
  if (!periodic_uniform(G)) return ins_k32g_apply_bc_p(G, p, as_stream(stream));
  return bc_periodic(G, p, 1, as_stream(stream));
}

extern "C" int ins_momentum_f32(const ins_grid_t* G, float visc, const float* u, float* F, void* stream) {
  INS_REQUIRE(G && u && F, "null argument");
  hipStream_t s = as_stream(stream);
  if (!periodic_uniform(G)) return ins_k32g_momentum(G, visc, u, F, s);
  if (ins_flux64_supported(G)) return ins_k_flux64_f32(G, (double)visc, u, F, nullptr, nullptr, 0, s);
  const Box32 b = box_of(G);
  if (b.D == 2)
    hipLaunchKernelGGL(k32_momentum<2>, grid_over(b, false), dim3(64, 4), 0, s, b, visc, u, F);
  else
    hipLaunchKernelGGL(k32_momentum<3>, grid_over(b, false), dim3(64, 4), 0, s, b, visc, u, F);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

extern "C" int ins_poisson_destroy_f32(ins_poisson32_t* ps) {
  if (!ps) return INS_OK;
  if (ps->plans) {
    (void)hipfftDestroy(ps->fwd);
    (void)hipfftDestroy(ps->inv);
  }
  if (ps->pI) (void)hipFree(ps->pI);
  if (ps->phat) (void)hipFree(ps->phat);
  for (float* a : ps->ahat)
    if (a) (void)hipFree(a);
  if (ps->ps64) (void)ins_poisson_destroy(ps->ps64);
  if (ps->phat32) (void)hipFree(ps->phat32);
  for (float* t : ps->tw32)
    if (t) (void)hipFree(t);
  if (ps->p64) (void)hipFree(ps->p64);
  if (ps->div32) (void)hipFree(ps->div32);
  delete ps;  // wrap64 is the caller's
  return INS_OK;
}

// A Float32 pressure solver around ANY fp64 solver of the same grid (psolver_direct, psolver_cg, psolver_spectral; pressure.jl:85-154, 209-351): the
// right-hand side is formed / widened in double, the fp64 solver runs unchanged, the pressure is rounded to float once.  `ps64` stays the caller's and must
// outlive the returned handle.  This is what carries the `_f32` family to wall-bounded and stretched grids (csrc/ins_f32g.hip).
extern "C" int ins_poisson_wrap_f32(const ins_grid_t* G, ins_poisson_t* ps64, ins_poisson32_t** out) {
  INS_REQUIRE(G && ps64 && out, "null argument");
  INS_REQUIRE(ps64->grid == G, "the fp64 solver was created for a different grid");
  const GridDev& g = G->g;
  for (int a = 0; a < g.D; ++a)
    if (g.bc[a][0] == INS_BC_HALO || g.bc[a][1] == INS_BC_HALO) {
      ins_set_error("ins_poisson_wrap_f32: slab (halo) grids run in fp64 only");
      return INS_ERR_UNSUPPORTED;
    }
  ins_poisson32* ps = new ins_poisson32();
  ps->grid = G;
  ps->wrap64 = ps64;
  for (int a = 0; a < g.D; ++a) ps->np[a] = g.ip_hi[a] - g.ip_lo[a];
  const bool ok = hipMalloc(&ps->p64, G->ncell * sizeof(double)) == hipSuccess && hipMemset(ps->p64, 0, G->ncell * sizeof(double)) == hipSuccess &&
                  hipMalloc(&ps->div32, G->ncell * sizeof(float)) == hipSuccess && hipMemset(ps->div32, 0, G->ncell * sizeof(float)) == hipSuccess;
  if (!ok) {
    ins_set_error("ins_poisson_wrap_f32: device allocation failed");
    ins_poisson_destroy_f32(ps);
    return INS_ERR_HIP;
  }
  *out = ps;
  return INS_OK;
}

// psolver_spectral(setup) with T = Float32                                               pressure.jl:289-351
extern "C" int ins_poisson_spectral_create_f32(const ins_grid_t* G, ins_poisson32_t** out) {
  INS_REQUIRE(G && out, "null argument");
  int rc = require_periodic_uniform(G, "ins_poisson_spectral_create_f32");
  if (rc) return rc;
  const GridDev& g = G->g;
  ins_poisson32* ps = new ins_poisson32();
  ps->grid = G;
  long long nreal = 1, ncplx = 1;
  double om = 1.0;
  for (int a = 0; a < g.D; ++a) {
    ps->np[a] = g.N[a] - 2;
    if (ps->np[a] % 2) {
      ins_set_error("spectral psolver: the number of volumes must be even in every direction (utils.jl:1-13)");
      delete ps;
      return INS_ERR_INVALID;
    }
    ps->kmax[a] = a == 0 ? ps->np[a] / 2 + 1 : ps->np[a];
    nreal *= ps->np[a];
    ncplx *= ps->kmax[a];
    om *= G->h[a];
  }
  bool ok = hipMalloc(&ps->pI, nreal * sizeof(float)) == hipSuccess && hipMalloc(&ps->phat, ncplx * sizeof(hipfftComplex)) == hipSuccess;
  for (int a = 0; ok && a < g.D; ++a) {  // âα(k) = 4 Ω sin²(π k / Npα) / Δxα², evaluated in double, stored in float
    std::vector<float> h(ps->kmax[a]);
    for (int k = 0; k < ps->kmax[a]; ++k) {
      const double sn = std::sin(M_PI * (double)k / ps->np[a]);
      h[k] = (float)(4.0 * om * sn * sn / (G->h[a] * G->h[a]));
    }
    ok = hipMalloc(&ps->ahat[a], h.size() * sizeof(float)) == hipSuccess &&
         hipMemcpy(ps->ahat[a], h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
  }
  if (ok) {
    hipfftResult r1, r2;
    if (g.D == 2) {
      r1 = hipfftPlan2d(&ps->fwd, ps->np[1], ps->np[0], HIPFFT_R2C);
      r2 = r1 == HIPFFT_SUCCESS ? hipfftPlan2d(&ps->inv, ps->np[1], ps->np[0], HIPFFT_C2R) : r1;
    } else {
      r1 = hipfftPlan3d(&ps->fwd, ps->np[2], ps->np[1], ps->np[0], HIPFFT_R2C);
      r2 = r1 == HIPFFT_SUCCESS ? hipfftPlan3d(&ps->inv, ps->np[2], ps->np[1], ps->np[0], HIPFFT_C2R) : r1;
    }
    if (r1 == HIPFFT_SUCCESS && r2 != HIPFFT_SUCCESS) (void)hipfftDestroy(ps->fwd);
    ps->plans = r1 == HIPFFT_SUCCESS && r2 == HIPFFT_SUCCESS;
    ok = ps->plans;
  }
  if (!ok) {
    ins_set_error("ins_poisson_spectral_create_f32: allocation or hipFFT plan creation failed");
    ins_poisson_destroy_f32(ps);
    return INS_ERR_FFT;
  }
  if (g.D == 3 && !ins_opt(OPT_INS_F32_HIPFFT_PROJECT)) {  // the fused fp64 passes for project! where they exist (power-of-two sides)
    ins_poisson* p64 = nullptr;
    if (ins_poisson_spectral_create(G, &p64) == INS_OK) {
      if (ins_k_spectral_own3d(p64))
        ps->ps64 = p64;
      else
        (void)ins_poisson_destroy(p64);
    }
    if (ps->ps64 && ins_k_spectral_own3d_f32(ps->ps64) && !ins_opt(OPT_INS_F32_FP64_SPECTRA)) {
      ps->kxs32 = (ps->kmax[0] + 15) & ~15;  // rows of the float2 spectrum padded to whole 128-B lines
      const size_t nb = (size_t)ps->kxs32 * ps->np[1] * ps->np[2] * 2 * sizeof(float);
      bool ok32 = hipMalloc(&ps->phat32, nb) == hipSuccess && hipMemset(ps->phat32, 0, nb) == hipSuccess;
      for (int a = 0; ok32 && a < 3; ++a) ok32 = ins_zsolve_twiddles_f32(ps->np[a], &ps->tw32[a]) == INS_OK;
      ps->own32 = ok32;
    }
  }
  *out = ps;
  return INS_OK;
}

static int solve32(ins_poisson32* ps, hipStream_t s) {  // pI -> pI
  const GridDev& g = ps->grid->g;
  INS_FFT_TRY(hipfftSetStream(ps->fwd, s));
  INS_FFT_TRY(hipfftSetStream(ps->inv, s));
  INS_FFT_TRY(hipfftExecR2C(ps->fwd, ps->pI, ps->phat));
  const float inv_n = 1.0f / ((float)ps->np[0] * (float)ps->np[1] * (float)ps->np[2]);
  const dim3 grid(cdiv(ps->kmax[0], 64), cdiv(ps->kmax[1], 4), g.D == 3 ? ps->kmax[2] : 1);
  if (g.D == 2)
    hipLaunchKernelGGL(k32_symbol<2>, grid, dim3(64, 4), 0, s, ps->phat, ps->ahat[0], ps->ahat[1], nullptr, ps->kmax[0], ps->kmax[1], inv_n);
  else
    hipLaunchKernelGGL(k32_symbol<3>, grid, dim3(64, 4), 0, s, ps->phat, ps->ahat[0], ps->ahat[1], ps->ahat[2], ps->kmax[0], ps->kmax[1], inv_n);
  INS_LAUNCH_CHECK();
  INS_FFT_TRY(hipfftExecC2R(ps->inv, ps->phat, ps->pI));
  return INS_OK;
}

// project!(u, setup; psolver, p) with T = Float32                                        pressure.jl:69-82
// u needs no valid ghosts on entry (periodic images are read); on return u and p carry their periodic ghosts.
extern "C" int ins_project_f32(const ins_grid_t* G, ins_poisson32_t* ps, float* u, float* p, void* stream) {
  INS_REQUIRE(G && ps && u && p, "null argument");
  INS_REQUIRE(ps->grid == G, "psolver was created for a different grid");
  hipStream_t s = as_stream(stream);
  if (ps->wrap64) return ins_k32g_project(G, ps->wrap64, ps->p64, u, p, s);
  const Box32 b = box_of(G);
  int rc;
  if (ps->own32) {
    if ((rc = ins_k_spectral_solve_f32(ps->ps64, u, ps->pI, ps->phat32, ps->kxs32, ps->tw32[0], ps->tw32[1], ps->tw32[2], s))) return rc;
    if (!ins_opt(OPT_INS_F32_SPLIT_GRADIENT)) {
      hipLaunchKernelGGL((k32_grad_ghost3<float>), grid_over(b, false), dim3(64, 4), 0, s, b, u, p, (const float*)ps->pI);
      INS_LAUNCH_CHECK();
      return INS_OK;
    }
    hipLaunchKernelGGL((k32_pad<3, float>), grid_over(b, true), dim3(64, 4), 0, s, b, (const float*)ps->pI, p);
    hipLaunchKernelGGL(k32_applypressure<3>, grid_over(b, false), dim3(64, 4), 0, s, b, u, p);
    INS_LAUNCH_CHECK();
    return bc_periodic(G, u, b.D, s);
  }
  if (ps->ps64) {
    if ((rc = ins_k_spectral_solve_from_u32(ps->ps64, u, s))) return rc;
    if (!ins_opt(OPT_INS_F32_SPLIT_GRADIENT)) {
      hipLaunchKernelGGL((k32_grad_ghost3<double>), grid_over(b, false), dim3(64, 4), 0, s, b, u, p, ins_k_spectral_pI(ps->ps64));
      INS_LAUNCH_CHECK();
      return INS_OK;
    }
    hipLaunchKernelGGL((k32_pad<3, double>), grid_over(b, true), dim3(64, 4), 0, s, b, ins_k_spectral_pI(ps->ps64), p);
    hipLaunchKernelGGL(k32_applypressure<3>, grid_over(b, false), dim3(64, 4), 0, s, b, u, p);
    INS_LAUNCH_CHECK();
    return bc_periodic(G, u, b.D, s);
  }
  if (b.D == 2) {
    hipLaunchKernelGGL(k32_div<2>, grid_over(b, false), dim3(64, 4), 0, s, b, u, ps->pI);
  } else {
    hipLaunchKernelGGL(k32_div<3>, grid_over(b, false), dim3(64, 4), 0, s, b, u, ps->pI);
  }
  INS_LAUNCH_CHECK();
  if ((rc = solve32(ps, s))) return rc;
  if (b.D == 2) {
    hipLaunchKernelGGL(k32_pad<2>, grid_over(b, true), dim3(64, 4), 0, s, b, ps->pI, p);
    hipLaunchKernelGGL(k32_applypressure<2>, grid_over(b, false), dim3(64, 4), 0, s, b, u, p);
  } else {
    hipLaunchKernelGGL(k32_pad<3>, grid_over(b, true), dim3(64, 4), 0, s, b, ps->pI, p);
    hipLaunchKernelGGL(k32_applypressure<3>, grid_over(b, false), dim3(64, 4), 0, s, b, u, p);
  }
  INS_LAUNCH_CHECK();
  return bc_periodic(G, u, b.D, s);
}

// psolver(p): solve L p = f on view(p, Ip) in place (pressure.jl:318-350); ghosts of p are refreshed
extern "C" int ins_poisson_solve_f32(ins_poisson32_t* ps, float* p, void* stream) {
  INS_REQUIRE(ps && p, "null argument");
  hipStream_t s = as_stream(stream);
  const ins_grid* G = ps->grid;
  const GridDev& g = G->g;
  if (ps->wrap64) return ins_k32g_solve(G, ps->wrap64, ps->p64, p, s);
  // strip the ghosts (2-D copies), solve, pad
  const size_t w = (size_t)ps->np[0] * sizeof(float);
  hipMemcpy3DParms c = {};
  c.srcPtr = make_hipPitchedPtr(p, (size_t)g.N[0] * sizeof(float), g.N[0], g.N[1]);
  c.srcPos = make_hipPos(sizeof(float), 1, g.D == 3 ? 1 : 0);
  c.dstPtr = make_hipPitchedPtr(ps->pI, w, ps->np[0], ps->np[1]);
  c.extent = make_hipExtent(w, ps->np[1], g.D == 3 ? ps->np[2] : 1);
  c.kind = hipMemcpyDeviceToDevice;
  INS_HIP_TRY(hipMemcpy3DAsync(&c, s));
  int rc = solve32(ps, s);
  if (rc) return rc;
  const Box32 b = box_of(G);
  if (b.D == 2)
    hipLaunchKernelGGL(k32_pad<2>, grid_over(b, true), dim3(64, 4), 0, s, b, ps->pI, p);
  else
    hipLaunchKernelGGL(k32_pad<3>, grid_over(b, true), dim3(64, 4), 0, s, b, ps->pI, p);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// ---------------------------------------------------------------------------------------------- explicit Runge-Kutta, T = Float32
extern "C" int ins_rk_destroy_f32(ins_rk32_t* rk) {
  if (rk && rk->ustart) (void)hipFree(rk->ustart);
  if (!rk) return INS_OK;
  for (float* k : rk->ku)
    if (k) (void)hipFree(k);
  for (float* b : rk->ub)
    if (b) (void)hipFree(b);
  if (rk->p) (void)hipFree(rk->p);
  if (rk->pu) (void)hipFree(rk->pu);
  delete rk;
  return INS_OK;
}

extern "C" int ins_rk_create_f32(const ins_grid_t* G, ins_poisson32_t* ps, int nstage, const double* A, const double* c, ins_rk32_t** out) {
  INS_REQUIRE(G && ps && A && c && out, "null argument");
  INS_REQUIRE(ps->grid == G, "psolver was created for a different grid");
  INS_REQUIRE(nstage >= 1 && nstage <= INS_MAX_STAGES, "unsupported number of stages");
  ins_rk32* rk = new ins_rk32();
  rk->grid = G;
  rk->ps = ps;
  rk->nstage = nstage;
  rk->A.assign(A, A + nstage * nstage);
  rk->c.assign(c, c + nstage);
  const size_t vbytes = (size_t)G->ncell * G->g.D * sizeof(float);
  rk->ku.assign(nstage, nullptr);
  bool ok = hipMalloc(&rk->p, G->ncell * sizeof(float)) == hipSuccess && hipMemset(rk->p, 0, G->ncell * sizeof(float)) == hipSuccess;
  for (int i = 0; ok && i < nstage; ++i) ok = hipMalloc(&rk->ku[i], vbytes) == hipSuccess && hipMemset(rk->ku[i], 0, vbytes) == hipSuccess;
  for (int b = 0; ok && b < 2; ++b) ok = hipMalloc(&rk->ub[b], vbytes) == hipSuccess && hipMemset(rk->ub[b], 0, vbytes) == hipSuccess;
  if (!ok) {
    ins_set_error("ins_rk_create_f32: device allocation failed");
    ins_rk_destroy_f32(rk);
    return INS_ERR_HIP;
  }
  *out = rk;
  return INS_OK;
}

// timestep!(method, stepper, Δt; cache) for closure_model = temp = bodyforce = nothing, T = Float32       step_explicit_runge_kutta.jl:4-59
// The caller's u is ustart for the whole step; on wide 3-D boxes the stage combination is the stencil kernel's epilogue.  As on the fp64 path (csrc/ins_rk.hip):
//   * stage-velocity basis: with in-register correction the uncorrected stage velocities stay in memory anyway, so the combination is written in terms of them
//     (V_i = (1 - Σβ) ustart + Σ β_im V_m + Δt A[i,i] k_i) and no stage force is stored or read — the `ku` arrays hold the stage velocities;
//   * chain bit 1: `u` holds the previous step's UNCORRECTED last stage velocity and the solver's pI its pressure: the first stage corrects in registers and
//     stores the corrected field as this step's ustart; bit 2: the last stage only solves (the next step of the chain corrects).
static int rk32_step(ins_rk32* rk, float visc, float* u, float dt, hipStream_t s, int chain) {
  const ins_grid* G = rk->grid;
  const int ns = rk->nstage, D = G->g.D;
  const long long nvec = G->ncell * D;
  const bool general = !periodic_uniform(G);  // walls / stretched spacings: the operator kernels of ins_f32g.hip, the reference's sequence stage by stage
  const bool wide = !general && ins_flux64_supported(G);
  const bool raw_in = chain & 1, raw_out = chain & 2;
  int rc;
  if (general && !rk->ps->wrap64) {
    ins_set_error("ins_rk_step_f32: this grid needs a solver made by ins_poisson_wrap_f32");
    return INS_ERR_UNSUPPORTED;
  }
  if (!raw_in && (rc = general ? ins_k32g_apply_bc_u(G, u, s) : bc_periodic(G, u, D, s))) return rc;  // :19
  // Wide power-of-two boxes: the fp64 path's stage structure in float — stages >= 2 read the previous stage's UNCORRECTED u* and
  // its pressure and apply u = u* - ∇p in registers (CORR = 1), the solve forms Ω·div(u*) from the float field inside its x
  // pass; only the last stage materialises u (padded p, gradient-subtract, ghosts).
  const bool incorr = wide && rk->ps->ps64 && ns > 1 && G->uniform_exact && G->g.N[0] >= 8 && G->g.N[1] >= 8 && G->g.N[2] >= 8 &&
                      !ins_opt(OPT_INS_DISABLE_INKERNEL_CORR);
  const long long ncell_in = (long long)(G->g.N[0] - 2) * (G->g.N[1] - 2) * (G->g.N[2] - 2);
  const bool own32 = rk->ps->own32;  // float2 spectra: the solver's float pI is what the next stage's stencil kernel reads
  if (incorr && !own32 && !rk->pu) INS_HIP_TRY(hipMalloc(&rk->pu, ncell_in * sizeof(float)));
  const float* pcorr = own32 ? rk->ps->pI : rk->pu;
  bool vbasis = incorr && !ins_opt(OPT_INS_RK_KEEP_K);
  for (int i = 0; vbasis && i < ns; ++i) vbasis = rk->A[i * ns + i] != 0.0;
  if (chain && !(incorr && own32)) {
    ins_set_error("ins_rk_steps_f32: chained steps need the in-register correction on float2 spectra");
    return INS_ERR_UNSUPPORTED;
  }
  const float* cur = u;
  for (int i = 0; i < ns; ++i) {
    float* outp = (i == ns - 1 && ns > 1) ? u : (vbasis ? rk->ku[i] : rk->ub[i & 1]);
    if (wide) {
      RkEpi epi;
      memset(&epi, 0, sizeof(epi));
      if (vbasis) {
        double beta[INS_MAX_STAGES];
        for (int m = i - 1; m >= 0; --m) {  // β_i · A[0:i,0:i] = A[i,0:i], A lower triangular
          double v = rk->A[i * ns + m];
          for (int j = m + 1; j < i; ++j) v -= beta[j] * rk->A[j * ns + m];
          beta[m] = v / rk->A[m * ns + m];
        }
        for (int m = 0; m < i; ++m) {
          if (beta[m] == 0.0) continue;
          epi.c0m1 -= beta[m];
          if (m == i - 1) {  // V_{i-1} is this stage's stencil input
            epi.self_in = beta[m];
            continue;
          }
          epi.coef[epi.n] = beta[m];
          epi.k[epi.n] = reinterpret_cast<const double*>(rk->ku[m]);  // float arrays behind RkEpi's untyped pointers (the kernels cast back)
          ++epi.n;
        }
      } else {
        for (int j = 0; j < i; ++j) {
          const double coef = (double)dt * rk->A[i * ns + j];
          if (coef == 0.0) continue;
          epi.coef[epi.n] = coef;
          epi.k[epi.n] = reinterpret_cast<const double*>(rk->ku[j]);
          ++epi.n;
        }
        for (int i2 = i + 1; i2 < ns; ++i2)
          if (rk->A[i2 * ns + i] != 0.0) epi.write_k = 1;
      }
      epi.coef_self = (double)dt * rk->A[i * ns + i];
      epi.ustart = i == 0 ? nullptr : reinterpret_cast<const double*>(raw_in ? rk->ustart : u);
      epi.ustar = reinterpret_cast<double*>(outp);
      if (i == 0 && raw_in) epi.ustart_out = reinterpret_cast<double*>(rk->ustart);
      const bool corr = incorr && (i > 0 || raw_in);
      if ((rc = ins_k_flux64_f32(G, (double)visc, cur, vbasis ? rk->ub[0] : rk->ku[i], &epi, corr ? pcorr : nullptr, corr ? 1 : 0, s))) return rc;   // :21, :35-38
    } else {
      if ((rc = ins_momentum_f32(G, visc, cur, rk->ku[i], s))) return rc;       // :21
      Comb32 cb;
      cb.n = 0;
      for (int j = 0; j <= i; ++j) {
        const float coef = dt * (float)rk->A[i * ns + j];
        if (coef == 0.f) continue;
        cb.coef[cb.n] = coef;
        cb.k[cb.n] = rk->ku[j];
        ++cb.n;
      }
      hipLaunchKernelGGL(k32_combine, dim3((unsigned)std::min<long long>((nvec + 255) / 256, 8192)), dim3(256), 0, s, nvec, u, outp, cb);  // :35-38
      INS_LAUNCH_CHECK();
      if (general && (rc = ins_k32g_apply_bc_u(G, outp, s))) return rc;          // :47
    }
    if (incorr && (i < ns - 1 || raw_out) && own32) {  // solve only, float2 spectra: p lands in the solver's float pI
      ins_poisson32* q = rk->ps;
      if ((rc = ins_k_spectral_solve_f32(q->ps64, outp, q->pI, q->phat32, q->kxs32, q->tw32[0], q->tw32[1], q->tw32[2], s))) return rc;
    } else if (incorr && i < ns - 1) {  // solve only: p (unpadded, float) for the next stage's in-register correction
      if ((rc = ins_k_spectral_solve_from_u32(rk->ps->ps64, outp, s))) return rc;
      hipLaunchKernelGGL(k32_cvt, dim3((unsigned)std::min<long long>((ncell_in + 255) / 256, 8192)), dim3(256), 0, s, ncell_in,
                         ins_k_spectral_pI(rk->ps->ps64), rk->pu);
      INS_LAUNCH_CHECK();
    } else if ((rc = ins_project_f32(G, rk->ps, outp, rk->p, s))) {             // :48-49 (periodic images: no ghost fill before)
      return rc;
    }
    if (general && (rc = ins_k32g_apply_bc_u(G, outp, s))) return rc;            // :49
    cur = outp;
  }
  if (ns == 1) INS_HIP_TRY(hipMemcpyAsync(u, rk->ub[0], nvec * sizeof(float), hipMemcpyDeviceToDevice, s));
  return INS_OK;
}

extern "C" int ins_rk_step_f32(ins_rk32_t* rk, float visc, float* u, float dt, void* stream) {
  INS_REQUIRE(rk && u, "null argument");
  return rk32_step(rk, visc, u, dt, as_stream(stream), 0);
}

// nsteps steps of size dt (the fixed-Δt loop of solve_unsteady, solver.jl:74-83) with the final correction of every step but the last folded into the next
// step's first stage kernel, as ins_rk_steps_f64; u is valid before and after the call.  Falls back to single steps where the chain does not apply.
extern "C" int ins_rk_steps_f32(ins_rk32_t* rk, float visc, float* u, float dt, int nsteps, void* stream) {
  INS_REQUIRE(rk && u && nsteps >= 0, "bad argument");
  const ins_grid* G = rk->grid;
  hipStream_t s = as_stream(stream);
  const int ns = rk->nstage;
  bool ok = ins_flux64_supported(G) && rk->ps->ps64 && rk->ps->own32 && ns > 1 && G->uniform_exact && G->g.N[0] >= 8 && G->g.N[1] >= 8 && G->g.N[2] >= 8 &&
            !ins_opt(OPT_INS_DISABLE_INKERNEL_CORR) && !ins_opt(OPT_INS_DISABLE_STEP_CHAIN) && !ins_opt(OPT_INS_RK_KEEP_K);
  for (int i = 0; ok && i < ns; ++i) ok = rk->A[i * ns + i] != 0.0;
  if (!ok || nsteps < 2) {
    for (int n = 0; n < nsteps; ++n) {
      int rc = rk32_step(rk, visc, u, dt, s, 0);
      if (rc) return rc;
    }
    return INS_OK;
  }
  if (!rk->ustart) INS_HIP_TRY(hipMalloc(&rk->ustart, (size_t)G->ncell * G->g.D * sizeof(float)));
  for (int n = 0; n < nsteps; ++n) {
    int rc = rk32_step(rk, visc, u, dt, s, (n > 0 ? 1 : 0) | (n < nsteps - 1 ? 2 : 0));
    if (rc) return rc;
  }
  return INS_OK;
}

// maximum(abs, divergence(u)) over Ip (diagnostic; blocking)                              operators.jl:106-125
extern "C" int ins_max_abs_divergence_f32(const ins_grid_t* G, ins_poisson32_t* ps, const float* u, float* out, void* stream) {
  INS_REQUIRE(G && ps && u && out, "null argument");
  hipStream_t s = as_stream(stream);
  if (ps->wrap64) {  // any grid: divergence! on Ip of a padded scratch (zero elsewhere), maximum on the host
    int rc = ins_k32g_divergence(G, u, ps->div32, s);
    if (rc) return rc;
    std::vector<float> h(G->ncell);
    INS_HIP_TRY(hipMemcpyAsync(h.data(), ps->div32, G->ncell * sizeof(float), hipMemcpyDeviceToHost, s));
    INS_HIP_TRY(hipStreamSynchronize(s));
    float m = 0.f;
    for (float v : h) m = std::fmax(m, std::fabs(v));
    *out = m;
    return INS_OK;
  }
  const Box32 b = box_of(G);
  if (b.D == 2)
    hipLaunchKernelGGL(k32_div<2>, grid_over(b, false), dim3(64, 4), 0, s, b, u, ps->pI);
  else
    hipLaunchKernelGGL(k32_div<3>, grid_over(b, false), dim3(64, 4), 0, s, b, u, ps->pI);
  INS_LAUNCH_CHECK();
  const long long n = (long long)ps->np[0] * ps->np[1] * ps->np[2];
  std::vector<float> h(n);
  INS_HIP_TRY(hipMemcpyAsync(h.data(), ps->pI, n * sizeof(float), hipMemcpyDeviceToHost, s));
  INS_HIP_TRY(hipStreamSynchronize(s));
  float m = 0.f;
  for (float v : h) m = std::fmax(m, std::fabs(v));
  *out = m / b.om;
  return INS_OK;
}
