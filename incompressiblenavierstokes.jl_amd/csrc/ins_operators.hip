// Generic (any BC, 2-D/3-D, stretched grids) staggered-grid operators: one work-item per volume,
// x along the 64-lane wavefront so every global access is a unit-stride row segment.
// These are the reference-faithful twins of operators.jl; the tiled 3-D fast paths live in
// ins_fast3d.hip and are parity-tested against these and against the CPU oracle.
#include "ins_internal.h"

namespace {

struct Launch3 {
  dim3 grid, block;
};

// Box of nx*ny*nz work-items, x fastest; 64x4 threads so each wavefront owns one x-row segment.
inline Launch3 box_launch(int nx, int ny, int nz) {
  Launch3 l;
  l.block = dim3(64, 4, 1);
  l.grid = dim3(cdiv(nx, 64), cdiv(ny, 4), (unsigned)nz);
  return l;
}

template <int D>
__device__ __forceinline__ bool in_range(const int (&I)[3], const int* lo, const int* hi) {
  bool ok = true;
#pragma unroll
  for (int b = 0; b < D; ++b) ok = ok && (I[b] >= lo[b]) && (I[b] < hi[b]);
  return ok;
}

// --------------------------------------------------------------------------------------------
// convection / diffusion / fused                       operators.jl:389-415, 549-573, 647-690
//   MODE bit0 = convection, bit1 = diffusion.  OVERWRITE: F = value (0 outside the DOF range,
//   i.e. `fill!(F, 0)` of momentum!, operators.jl:971, fused in) instead of F += value.
// --------------------------------------------------------------------------------------------
template <int D, int MODE, bool OVERWRITE>
__global__ __launch_bounds__(256) void k_convdiff(GridDev g, double visc, const double* __restrict__ u,
                                                  double* __restrict__ F) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= g.N[0] || j >= g.N[1]) return;
  const int I[3] = {i, j, k};
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  bool inside = true;
#pragma unroll
  for (int a = 0; a < D; ++a) inside = inside && I[a] >= 1 && I[a] <= g.N[a] - 2;

#pragma unroll
  for (int al = 0; al < D; ++al) {
    const bool dof = inside && in_range<D>(I, g.iu_lo[al], g.iu_hi[al]);
    double* Fa = F + al * g.sc;
    if (!dof) {
      if (OVERWRITE) Fa[c] = 0.0;
      continue;
    }
    const double* ua = u + al * g.sc;
    const long long sa = g.sx[al];
    const double uc = ua[c];
    double f = OVERWRITE ? 0.0 : Fa[c];
#pragma unroll
    for (int be = 0; be < D; ++be) {
      const long long sb = g.sx[be];
      const int ib = I[be];
      const double um = ua[c - sb], up = ua[c + sb];
      const double r = (al == be ? g.rdxu[be] : g.rdx[be])[ib];
      double term = 0.0;
      if (MODE & 2) {
        const double ma = al == be ? g.mdx[be][ib] : g.mdxu[be][ib - 1];
        const double mb = al == be ? g.mdx[be][ib + 1] : g.mdxu[be][ib];
        const double d1 = (uc - um) * ma;
        const double d2 = (up - uc) * mb;
        term = visc * (d2 - d1);
      }
      if (MODE & 1) {
        const double* ub = u + be * g.sc;
        const double* A1 = g.A1[be][al];  // weights of component be in direction al (reverse interpolation)
        const double* A2 = g.A2[be][al];
        const int ia = I[al];
        const double uab1 = (um + uc) * 0.5;
        const double uab2 = (uc + up) * 0.5;
        const double uba1 = A2[ia - (al == be)] * ub[c - sb] + A1[ia + (al != be)] * ub[c - sb + sa];
        const double uba2 = A2[ia] * ub[c] + A1[ia + 1] * ub[c + sa];
        term -= (uab2 * uba2 - uab1 * uba1);
      }
      f += term * r;
    }
    Fa[c] = f;
  }
}

// K1 + K6 for every grid the tiled 3-D kernels do not take (2-D, tiny boxes): momentum! (fill + convection-diffusion) with the stage
// combination of step_explicit_runge_kutta.jl:35-38 as its epilogue, over the WHOLE padded array like the reference's broadcasts:
//   u*[c] = (ustart ? ustart[c] : u[c]) + Σ_q coef_q k_q[c] + coef_self F[c],   k_i[c] = F[c] when a later stage reads it.
// One pass instead of three (kernel, k_combine, the ustart snapshot).  `epi.ustar` must not alias `u` (neighbours are read).
template <int D>
__global__ __launch_bounds__(256) void k_convdiff_rk(GridDev g, double visc, const double* __restrict__ u, double* __restrict__ F, RkEpi epi) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= g.N[0] || j >= g.N[1]) return;
  const int I[3] = {i, j, k};
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  bool inside = true;
#pragma unroll
  for (int a = 0; a < D; ++a) inside = inside && I[a] >= 1 && I[a] <= g.N[a] - 2;
#pragma unroll
  for (int al = 0; al < D; ++al) {
    const bool dof = inside && in_range<D>(I, g.iu_lo[al], g.iu_hi[al]);
    const double* ua = u + al * g.sc;
    const double uc = ua[c];
    double f = 0.0;
    if (dof) {
      const long long sa = g.sx[al];
#pragma unroll
      for (int be = 0; be < D; ++be) {
        const long long sb = g.sx[be];
        const int ib = I[be];
        const double um = ua[c - sb], up = ua[c + sb];
        const double r = (al == be ? g.rdxu[be] : g.rdx[be])[ib];
        const double ma = al == be ? g.mdx[be][ib] : g.mdxu[be][ib - 1];
        const double mb = al == be ? g.mdx[be][ib + 1] : g.mdxu[be][ib];
        double term = visc * ((up - uc) * mb - (uc - um) * ma);
        const double* ub = u + be * g.sc;
        const double* A1 = g.A1[be][al];
        const double* A2 = g.A2[be][al];
        const int ia = I[al];
        const double uab1 = (um + uc) * 0.5;
        const double uab2 = (uc + up) * 0.5;
        const double uba1 = A2[ia - (al == be)] * ub[c - sb] + A1[ia + (al != be)] * ub[c - sb + sa];
        const double uba2 = A2[ia] * ub[c] + A1[ia + 1] * ub[c + sa];
        term -= (uab2 * uba2 - uab1 * uba1);
        f += term * r;
      }
    }
    const long long ca = al * g.sc + c;
    double sv = epi.ustart ? epi.ustart[ca] : uc;
    for (int q = 0; q < epi.n; ++q) sv += epi.coef[q] * epi.k[q][ca];
    epi.ustar[ca] = sv + epi.coef_self * f;
    if (epi.write_k) F[ca] = f;
  }
}

template <int MODE, bool OVERWRITE>
int launch_convdiff(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s) {
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.N[0], g.N[1], g.N[2]);
  if (g.D == 2)
    hipLaunchKernelGGL((k_convdiff<2, MODE, OVERWRITE>), l.grid, l.block, 0, s, g, visc, u, F);
  else
    hipLaunchKernelGGL((k_convdiff<3, MODE, OVERWRITE>), l.grid, l.block, 0, s, g, visc, u, F);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// --------------------------------------------------------------------------------------------
// divergence                                                              operators.jl:117-125
// --------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_divergence(GridDev g, const double* __restrict__ u, double* __restrict__ div) {
  const int i = g.ip_lo[0] + blockIdx.x * 64 + threadIdx.x;
  const int j = g.ip_lo[1] + blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? g.ip_lo[2] + (int)blockIdx.z : 0;
  if (i >= g.ip_hi[0] || j >= g.ip_hi[1]) return;
  const int I[3] = {i, j, k};
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  double d = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double* ua = u + a * g.sc;
    d += (ua[c] - ua[c - g.sx[a]]) * g.rdx[a][I[a]];
  }
  div[c] = d;
}

// scalewithvolume!                                                          operators.jl:81-95
template <int D>
__global__ __launch_bounds__(256) void k_scalewithvolume(GridDev g, double* __restrict__ p) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= g.N[0] || j >= g.N[1]) return;
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  double om = g.dx[0][i] * g.dx[1][j];
  if (D == 3) om = om * g.dx[2][k];
  p[c] *= om;
}

// pressuregradient! / applypressure!                                 operators.jl:170-178, 225-233
template <int D, bool APPLY>
__global__ __launch_bounds__(256) void k_pressuregradient(GridDev g, const double* __restrict__ p, double* __restrict__ G) {
  const int i = 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = 1 + blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? 1 + (int)blockIdx.z : 0;
  if (i > g.N[0] - 2 || j > g.N[1] - 2) return;
  const int I[3] = {i, j, k};
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  const double pc = p[c];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    if (in_range<D>(I, g.iu_lo[a], g.iu_hi[a])) {
      const double gr = (p[c + g.sx[a]] - pc) * g.rdxu[a][I[a]];
      double* Ga = G + a * g.sc;
      if (APPLY)
        Ga[c] -= gr;
      else
        Ga[c] = gr;
    }
  }
}

// laplacian!                                                           operators.jl:328-363
template <int D>
__global__ __launch_bounds__(256) void k_laplacian(GridDev g, const double* __restrict__ p, double* __restrict__ L) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= g.N[0] || j >= g.N[1]) return;
  const int I[3] = {i, j, k};
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  if (!in_range<D>(I, g.ip_lo, g.ip_hi)) {
    L[c] = 0.0;  // `L .= 0` (operators.jl:359)
    return;
  }
  double om = g.dx[0][i] * g.dx[1][j];
  if (D == 3) om = om * g.dx[2][k];
  const double pc = p[c];
  double lap = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const int ia = I[a];
    const bool first = ia == g.ip_lo[a], last = ia == g.ip_hi[a] - 1;
    const double pp = p[c + g.sx[a]], pm = p[c - g.sx[a]];
    const double rr = g.rdxu[a][ia], rl = g.rdxu[a][ia - 1];
    double right = (pp - pc) * rr;
    double left = (pc - pm) * rl;
    // if / elseif chain of operators.jl:334-350
    if (first && g.bc[a][0] == INS_BC_PRESSURE)
      left = pc * rl;
    else if (last && g.bc[a][1] == INS_BC_PRESSURE)
      right = (-pc) * rr;
    else if (first && g.bc[a][0] == INS_BC_DIRICHLET)
      left = 0.0;
    else if (last && g.bc[a][1] == INS_BC_DIRICHLET)
      right = 0.0;
    lap += om * g.rdx[a][ia] * (right - left);
  }
  L[c] = lap;
}

// kinetic_energy!                                                    operators.jl:1521-1545
template <int D>
__global__ __launch_bounds__(256) void k_kinetic_energy(GridDev g, const double* __restrict__ u, double* __restrict__ ke,
                                                        int interpolate_first) {
  const int i = g.ip_lo[0] + blockIdx.x * 64 + threadIdx.x;
  const int j = g.ip_lo[1] + blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? g.ip_lo[2] + (int)blockIdx.z : 0;
  if (i >= g.ip_hi[0] || j >= g.ip_hi[1]) return;
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  double e = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double up = u[a * g.sc + c], um = u[a * g.sc + c - g.sx[a]];
    e += interpolate_first ? (up + um) * (up + um) : up * up + um * um;
  }
  ke[c] = interpolate_first ? e / 8 : e / 4;
}

// buf = Δu[α] / |u[α]| on Iu[α]; elsewhere +inf                                solver.jl:115-118
template <int D>
__global__ __launch_bounds__(256) void k_cfl(GridDev g, const double* __restrict__ u, int al, double* __restrict__ buf) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= g.N[0] || j >= g.N[1]) return;
  const int I[3] = {i, j, k};
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  buf[c] = g.dxu[al][I[al]] / fabs(u[al * g.sc + c]);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// internal launchers
// ------------------------------------------------------------------------------------------------
int ins_k_momentum_rk_fused_generic(const ins_grid* G, double visc, const double* u, double* k_out, const RkEpi& epi, hipStream_t s) {
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.N[0], g.N[1], g.N[2]);
  if (g.D == 2)
    hipLaunchKernelGGL(k_convdiff_rk<2>, l.grid, l.block, 0, s, g, visc, u, k_out, epi);
  else
    hipLaunchKernelGGL(k_convdiff_rk<3>, l.grid, l.block, 0, s, g, visc, u, k_out, epi);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

int ins_k_momentum_generic(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s) {
  return launch_convdiff<3, true>(G, visc, u, F, s);
}

int ins_k_diffusion_overwrite(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s) {
  return launch_convdiff<2, true>(G, visc, u, F, s);  // fill!(F, 0) + diffusion!(F, u) in one write-only pass
}

int ins_k_divergence(const ins_grid* G, const double* u, double* div, hipStream_t s) {
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.ip_hi[0] - g.ip_lo[0], g.ip_hi[1] - g.ip_lo[1], g.ip_hi[2] - g.ip_lo[2]);
  if (g.D == 2)
    hipLaunchKernelGGL(k_divergence<2>, l.grid, l.block, 0, s, g, u, div);
  else
    hipLaunchKernelGGL(k_divergence<3>, l.grid, l.block, 0, s, g, u, div);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

int ins_k_scalewithvolume(const ins_grid* G, double* p, hipStream_t s) {
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.N[0], g.N[1], g.N[2]);
  if (g.D == 2)
    hipLaunchKernelGGL(k_scalewithvolume<2>, l.grid, l.block, 0, s, g, p);
  else
    hipLaunchKernelGGL(k_scalewithvolume<3>, l.grid, l.block, 0, s, g, p);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

template <bool APPLY>
static int launch_pg(const ins_grid* G, const double* p, double* Gf, hipStream_t s) {
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.N[0] - 2, g.N[1] - 2, g.D == 3 ? g.N[2] - 2 : 1);
  if (g.D == 2)
    hipLaunchKernelGGL((k_pressuregradient<2, APPLY>), l.grid, l.block, 0, s, g, p, Gf);
  else
    hipLaunchKernelGGL((k_pressuregradient<3, APPLY>), l.grid, l.block, 0, s, g, p, Gf);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

int ins_k_applypressure(const ins_grid* G, double* u, const double* p, hipStream_t s) { return launch_pg<true>(G, p, u, s); }

int ins_k_laplacian(const ins_grid* G, const double* p, double* L, hipStream_t s) {
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.N[0], g.N[1], g.N[2]);
  if (g.D == 2)
    hipLaunchKernelGGL(k_laplacian<2>, l.grid, l.block, 0, s, g, p, L);
  else
    hipLaunchKernelGGL(k_laplacian<3>, l.grid, l.block, 0, s, g, p, L);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
#define INS_ARGS2(G, a, b) INS_REQUIRE((G) && (a) && (b), "null argument")

extern "C" int ins_scalewithvolume_f64(const ins_grid_t* G, double* p, void* stream) {
  INS_REQUIRE(G && p, "null argument");
  return ins_k_scalewithvolume(G, p, as_stream(stream));
}

extern "C" int ins_divergence_f64(const ins_grid_t* G, const double* u, double* div, void* stream) {
  INS_ARGS2(G, u, div);
  return ins_k_divergence(G, u, div, as_stream(stream));
}

extern "C" int ins_pressuregradient_f64(const ins_grid_t* G, const double* p, double* Gf, void* stream) {
  INS_ARGS2(G, p, Gf);
  return launch_pg<false>(G, p, Gf, as_stream(stream));
}

extern "C" int ins_applypressure_f64(const ins_grid_t* G, double* u, const double* p, void* stream) {
  INS_ARGS2(G, u, p);
  return ins_k_applypressure(G, u, p, as_stream(stream));
}

extern "C" int ins_laplacian_f64(const ins_grid_t* G, const double* p, double* L, void* stream) {
  INS_ARGS2(G, p, L);
  INS_REQUIRE(p != L, "laplacian! cannot run in place");
  return ins_k_laplacian(G, p, L, as_stream(stream));
}

extern "C" int ins_convection_f64(const ins_grid_t* G, const double* u, double* F, void* stream) {
  INS_ARGS2(G, u, F);
  INS_REQUIRE(u != F, "convection! cannot run in place");
  return launch_convdiff<1, false>(G, 0.0, u, F, as_stream(stream));
}

extern "C" int ins_diffusion_f64(const ins_grid_t* G, double visc, const double* u, double* F, void* stream) {
  INS_ARGS2(G, u, F);
  INS_REQUIRE(u != F, "diffusion! cannot run in place");
  return launch_convdiff<2, false>(G, visc, u, F, as_stream(stream));
}

extern "C" int ins_convectiondiffusion_f64(const ins_grid_t* G, double visc, const double* u, double* F, void* stream) {
  INS_ARGS2(G, u, F);
  INS_REQUIRE(u != F, "convectiondiffusion! cannot run in place");
  return launch_convdiff<3, false>(G, visc, u, F, as_stream(stream));
}

extern "C" int ins_momentum_f64(const ins_grid_t* G, double visc, const double* u, double* F, void* stream) {
  INS_ARGS2(G, u, F);
  INS_REQUIRE(u != F, "momentum! cannot run in place");
  return ins_k_momentum(G, visc, u, F, as_stream(stream));
}

extern "C" int ins_kinetic_energy_f64(const ins_grid_t* G, const double* u, double* ke, int interpolate_first, void* stream) {
  INS_ARGS2(G, u, ke);
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.ip_hi[0] - g.ip_lo[0], g.ip_hi[1] - g.ip_lo[1], g.ip_hi[2] - g.ip_lo[2]);
  hipStream_t s = as_stream(stream);
  if (g.D == 2)
    hipLaunchKernelGGL(k_kinetic_energy<2>, l.grid, l.block, 0, s, g, u, ke, interpolate_first);
  else
    hipLaunchKernelGGL(k_kinetic_energy<3>, l.grid, l.block, 0, s, g, u, ke, interpolate_first);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// Scratch scalar field for the blocking diagnostics (allocated per call: these are not on the step path).
struct ScratchField {
  double* p = nullptr;
  ~ScratchField() {
    if (p) (void)hipFree(p);
  }
};

extern "C" int ins_total_kinetic_energy_f64(const ins_grid_t* G, const double* u, int interpolate_first, double* out, void* stream) {
  INS_ARGS2(G, u, out);
  ScratchField ke;
  INS_HIP_TRY(hipMalloc(&ke.p, G->ncell * sizeof(double)));
  hipStream_t s = as_stream(stream);
  INS_HIP_TRY(hipMemsetAsync(ke.p, 0, G->ncell * sizeof(double), s));
  int rc = ins_kinetic_energy_f64(G, u, ke.p, interpolate_first, stream);
  if (rc) return rc;
  rc = ins_k_scalewithvolume(G, ke.p, s);
  if (rc) return rc;
  return ins_k_reduce(G, 3, ke.p, nullptr, G->g.ip_lo, G->g.ip_hi, out, s);
}

extern "C" int ins_cfl_timestep_f64(const ins_grid_t* G, double Re, const double* u, double* out, void* stream) {
  INS_ARGS2(G, u, out);
  const GridDev& g = G->g;
  ScratchField buf;
  INS_HIP_TRY(hipMalloc(&buf.p, G->ncell * sizeof(double)));
  hipStream_t s = as_stream(stream);
  double dt = INFINITY;
  for (int a = 0; a < g.D; ++a) {
    double damin = INFINITY;
    for (int i = g.iu_lo[a][a]; i < g.iu_hi[a][a]; ++i) damin = fmin(damin, G->desc.dxu[a][i]);
    const double dt_diff = Re * damin * damin / 2;
    Launch3 l = box_launch(g.N[0], g.N[1], g.N[2]);
    if (g.D == 2)
      hipLaunchKernelGGL(k_cfl<2>, l.grid, l.block, 0, s, g, u, a, buf.p);
    else
      hipLaunchKernelGGL(k_cfl<3>, l.grid, l.block, 0, s, g, u, a, buf.p);
    INS_LAUNCH_CHECK();
    double dt_conv;
    int rc = ins_k_reduce(G, 2, buf.p, nullptr, g.iu_lo[a], g.iu_hi[a], &dt_conv, s);
    if (rc) return rc;
    dt = fmin(dt, fmin(dt_diff, dt_conv));
  }
  *out = dt;
  return INS_OK;
}

extern "C" int ins_max_abs_divergence_f64(const ins_grid_t* G, const double* u, double* out, void* stream) {
  INS_ARGS2(G, u, out);
  ScratchField d;
  INS_HIP_TRY(hipMalloc(&d.p, G->ncell * sizeof(double)));
  hipStream_t s = as_stream(stream);
  int rc = ins_k_divergence(G, u, d.p, s);
  if (rc) return rc;
  return ins_k_reduce(G, 1, d.p, nullptr, G->g.ip_lo, G->g.ip_hi, out, s);
}
