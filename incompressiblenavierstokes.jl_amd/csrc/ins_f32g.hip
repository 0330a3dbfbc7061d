// The `_f32` family on ANY grid the fp64 family takes (2-D / 3-D; Dirichlet, Symmetric, Pressure and periodic sides; stretched spacings): the reference is
// generic in the element type on every setup (docs/src/manual/precision.md:3-16), not only on the periodic boxes of csrc/ins_f32.hip.
//
//   fields      : Float32 arrays in the reference layout; the grid handle is the fp64 one — metric tables are read as doubles and rounded to float where
//                 they enter the arithmetic (one table for both families; the tables are a few KB and stay in cache);
//   operators   : one work-item per volume, x along the wavefront, the same index rules as csrc/ins_operators.hip / ins_bc.hip (which cite the reference
//                 lines); all arithmetic in float;
//   projection  : Ω·div(u) is formed from the float field in DOUBLE, the fp64 solver the caller wrapped (direct = fast diagonalisation, CG, spectral) runs
//                 unchanged, and the pressure is rounded to float once: a mixed-precision projection whose pressure is at least as accurate as a
//                 Float32 factorisation's (the reference's T = Float32 runs its sparse LU in Float32, pressure.jl:101-154).
// Slab (HALO) sides are not taken: the multi-GPU path is fp64.
#include <cmath>

#include "ins_internal.h"

namespace {

template <int D>
__device__ __forceinline__ bool in_range32(const int (&I)[3], const int* lo, const int* hi) {
  bool ok = true;
#pragma unroll
  for (int b = 0; b < D; ++b) ok = ok && (I[b] >= lo[b]) && (I[b] < hi[b]);
  return ok;
}

// apply_bc_u!                                   boundary_conditions.jl:159-206, 276-288, 344-375, 414-428, 472-482 (constant boundary data)
template <int D>
__global__ __launch_bounds__(256) void k32g_bc_u(GridDev g, float* __restrict__ u, int be) {
  const int o0 = be == 0 ? 1 : 0, o1 = be == 2 ? 1 : 2;
  const int q0 = blockIdx.x * 256 + threadIdx.x;
  const int q1 = D == 3 ? (int)blockIdx.y : 0;
  const int al = blockIdx.z;
  if (q0 >= g.N[o0]) return;
  const long long base = q0 * g.sx[o0] + (D == 3 ? q1 * g.sx[o1] : 0);
  const long long sb = g.sx[be];
  float* ua = u + al * g.sc;
  const int bcl = g.bc[be][0], bcr = g.bc[be][1];
  if (bcl == INS_BC_PERIODIC) {
    const int ia = g.ip_lo[be] - 1, ib = g.ip_hi[be];
    ua[base + ia * sb] = ua[base + (ib - 1) * sb];
    ua[base + ib * sb] = ua[base + (ia + 1) * sb];
    return;
  }
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int bc = side ? bcr : bcl;
    const int i = side ? g.iu_hi[al][be] : g.iu_lo[al][be] - 1;
    const int jn = side ? i - 1 : i + 1;
    float* dst = ua + base + i * sb;
    if (bc == INS_BC_DIRICHLET)
      *dst = (float)g.bc_u[be][side][al];
    else if (bc == INS_BC_SYMMETRIC)
      *dst = (al == be) ? 0.f : ua[base + jn * sb];
    else if (bc == INS_BC_PRESSURE)
      *dst = ua[base + jn * sb];
  }
}

// apply_bc_p!                                   boundary_conditions.jl:306-318, 388, 445-453, 497-502
template <int D>
__global__ __launch_bounds__(256) void k32g_bc_p(GridDev g, float* __restrict__ p, int be) {
  const int o0 = be == 0 ? 1 : 0, o1 = be == 2 ? 1 : 2;
  const int q0 = blockIdx.x * 256 + threadIdx.x;
  const int q1 = D == 3 ? (int)blockIdx.y : 0;
  if (q0 >= g.N[o0]) return;
  const long long base = q0 * g.sx[o0] + (D == 3 ? q1 * g.sx[o1] : 0);
  const long long sb = g.sx[be];
  const int bcl = g.bc[be][0], bcr = g.bc[be][1];
  const int ia = g.ip_lo[be] - 1, ib = g.ip_hi[be];
  if (bcl == INS_BC_PERIODIC) {
    p[base + ia * sb] = p[base + (ib - 1) * sb];
    p[base + ib * sb] = p[base + (ia + 1) * sb];
    return;
  }
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int bc = side ? bcr : bcl;
    const int i = side ? ib : ia;
    const int jn = side ? i - 1 : i + 1;
    if (bc == INS_BC_SYMMETRIC)
      p[base + i * sb] = p[base + jn * sb];
    else if (bc == INS_BC_PRESSURE)
      p[base + i * sb] = 0.f;
  }
}

// momentum! = fill!(F, 0) + convectiondiffusion!                                         operators.jl:647-690, 971
template <int D>
__global__ __launch_bounds__(256) void k32g_momentum(GridDev g, float visc, const float* __restrict__ u, float* __restrict__ F) {
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
    float* Fa = F + al * g.sc;
    if (!(inside && in_range32<D>(I, g.iu_lo[al], g.iu_hi[al]))) {
      Fa[c] = 0.f;
      continue;
    }
    const float* ua = u + al * g.sc;
    const long long sa = g.sx[al];
    const float uc = ua[c];
    float f = 0.f;
#pragma unroll
    for (int be = 0; be < D; ++be) {
      const long long sb = g.sx[be];
      const int ib = I[be], ia = I[al];
      const float um = ua[c - sb], up = ua[c + sb];
      const float r = (float)(al == be ? g.rdxu[be] : g.rdx[be])[ib];
      const float ma = (float)(al == be ? g.mdx[be][ib] : g.mdxu[be][ib - 1]);
      const float mb = (float)(al == be ? g.mdx[be][ib + 1] : g.mdxu[be][ib]);
      const float* ub = u + be * g.sc;
      const double* A1 = g.A1[be][al];
      const double* A2 = g.A2[be][al];
      const float uab1 = (um + uc) * 0.5f, uab2 = (uc + up) * 0.5f;
      const float uba1 = (float)A2[ia - (al == be)] * ub[c - sb] + (float)A1[ia + (al != be)] * ub[c - sb + sa];
      const float uba2 = (float)A2[ia] * ub[c] + (float)A1[ia + 1] * ub[c + sa];
      f += (visc * ((up - uc) * mb - (uc - um) * ma) - (uab2 * uba2 - uab1 * uba1)) * r;
    }
    Fa[c] = f;
  }
}

// divergence! on Ip (operators.jl:117-125); SCALE: times the volume (scalewithvolume!, operators.jl:81-95).  P = double: the differences are taken in double
// (the right-hand side of the fp64 solver); P = float: the diagnostic.
template <int D, typename P, bool SCALE>
__global__ __launch_bounds__(256) void k32g_div(GridDev g, const float* __restrict__ u, P* __restrict__ out) {
  const int i = g.ip_lo[0] + blockIdx.x * 64 + threadIdx.x;
  const int j = g.ip_lo[1] + blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? g.ip_lo[2] + (int)blockIdx.z : 0;
  if (i >= g.ip_hi[0] || j >= g.ip_hi[1]) return;
  const int I[3] = {i, j, k};
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  P d = 0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const float* ua = u + a * g.sc;
    d += ((P)ua[c] - (P)ua[c - g.sx[a]]) * (P)g.rdx[a][I[a]];
  }
  if (SCALE) {
    P om = (P)g.dx[0][i] * (P)g.dx[1][j];
    if (D == 3) om *= (P)g.dx[2][k];
    d *= om;
  }
  out[c] = d;
}

// applypressure!                                                                          operators.jl:225-233
template <int D>
__global__ __launch_bounds__(256) void k32g_applypressure(GridDev g, float* __restrict__ u, const float* __restrict__ p) {
  const int i = 1 + blockIdx.x * 64 + threadIdx.x;
  const int j = 1 + blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? 1 + (int)blockIdx.z : 0;
  if (i > g.N[0] - 2 || j > g.N[1] - 2) return;
  const int I[3] = {i, j, k};
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  const float pc = p[c];
#pragma unroll
  for (int a = 0; a < D; ++a)
    if (in_range32<D>(I, g.iu_lo[a], g.iu_hi[a])) u[a * g.sc + c] -= (p[c + g.sx[a]] - pc) * (float)g.rdxu[a][I[a]];
}

__global__ __launch_bounds__(256) void k32g_widen(long long n, const float* __restrict__ a, double* __restrict__ b) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) b[t] = (double)a[t];
}
__global__ __launch_bounds__(256) void k32g_round(long long n, const double* __restrict__ a, float* __restrict__ b) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) b[t] = (float)a[t];
}

inline dim3 flat_grid(long long n) { return dim3((unsigned)std::min<long long>((n + 255) / 256, 8192)); }

int no_halo(const ins_grid* G, const char* what) {
  for (int a = 0; a < G->g.D; ++a)
    if (G->g.bc[a][0] == INS_BC_HALO || G->g.bc[a][1] == INS_BC_HALO) {
      ins_set_error("%s: slab (halo) grids run in fp64 only", what);
      return INS_ERR_UNSUPPORTED;
    }
  return INS_OK;
}

}  // namespace

int ins_k32g_apply_bc_u(const ins_grid* G, float* u, hipStream_t s) {
  const GridDev& g = G->g;
  int rc = no_halo(G, "apply_bc_u (f32)");
  if (rc) return rc;
  for (int be = 0; be < g.D; ++be) {  // direction after direction: edges and corners come out as in the reference (boundary_conditions.jl:162-165)
    const int o0 = be == 0 ? 1 : 0, o1 = be == 2 ? 1 : 2;
    dim3 grid(cdiv(g.N[o0], 256), g.D == 3 ? g.N[o1] : 1, g.D);
    if (g.D == 2)
      hipLaunchKernelGGL(k32g_bc_u<2>, grid, dim3(256), 0, s, g, u, be);
    else
      hipLaunchKernelGGL(k32g_bc_u<3>, grid, dim3(256), 0, s, g, u, be);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}

int ins_k32g_apply_bc_p(const ins_grid* G, float* p, hipStream_t s) {
  const GridDev& g = G->g;
  int rc = no_halo(G, "apply_bc_p (f32)");
  if (rc) return rc;
  for (int be = 0; be < g.D; ++be) {
    if (g.bc[be][0] == INS_BC_DIRICHLET && g.bc[be][1] == INS_BC_DIRICHLET) continue;  // no-op (boundary_conditions.jl:388)
    const int o0 = be == 0 ? 1 : 0, o1 = be == 2 ? 1 : 2;
    dim3 grid(cdiv(g.N[o0], 256), g.D == 3 ? g.N[o1] : 1, 1);
    if (g.D == 2)
      hipLaunchKernelGGL(k32g_bc_p<2>, grid, dim3(256), 0, s, g, p, be);
    else
      hipLaunchKernelGGL(k32g_bc_p<3>, grid, dim3(256), 0, s, g, p, be);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}

int ins_k32g_momentum(const ins_grid* G, float visc, const float* u, float* F, hipStream_t s) {
  const GridDev& g = G->g;
  dim3 grid(cdiv(g.N[0], 64), cdiv(g.N[1], 4), g.D == 3 ? g.N[2] : 1);
  if (g.D == 2)
    hipLaunchKernelGGL(k32g_momentum<2>, grid, dim3(64, 4), 0, s, g, visc, u, F);
  else
    hipLaunchKernelGGL(k32g_momentum<3>, grid, dim3(64, 4), 0, s, g, visc, u, F);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// div[Ip] = divergence(u) (padded float array; volumes outside Ip are left as they are)
int ins_k32g_divergence(const ins_grid* G, const float* u, float* div, hipStream_t s) {
  const GridDev& g = G->g;
  dim3 grid(cdiv(g.ip_hi[0] - g.ip_lo[0], 64), cdiv(g.ip_hi[1] - g.ip_lo[1], 4), g.D == 3 ? g.ip_hi[2] - g.ip_lo[2] : 1);
  if (g.D == 2)
    hipLaunchKernelGGL((k32g_div<2, float, false>), grid, dim3(64, 4), 0, s, g, u, div);
  else
    hipLaunchKernelGGL((k32g_div<3, float, false>), grid, dim3(64, 4), 0, s, g, u, div);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// psolver(p) with a Float32 p around the fp64 solver: widen, solve (pressure.jl:22 on view(p, Ip)), round.  p64: padded fp64 scratch.
int ins_k32g_solve(const ins_grid* G, ins_poisson* ps64, double* p64, float* p, hipStream_t s) {
  hipLaunchKernelGGL(k32g_widen, flat_grid(G->ncell), dim3(256), 0, s, G->ncell, p, p64);
  INS_LAUNCH_CHECK();
  int rc = ins_k_poisson_solve(ps64, p64, s);
  if (rc) return rc;
  hipLaunchKernelGGL(k32g_round, flat_grid(G->ncell), dim3(256), 0, s, G->ncell, p64, p);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// project!(u, setup; psolver, p), T = Float32: divergence! + scalewithvolume! (in double, from the float field), the fp64 solve, p rounded once, apply_bc_p!,
// applypressure! (pressure.jl:69-82).  As in the reference the caller fills the ghost volumes of u before and after.
int ins_k32g_project(const ins_grid* G, ins_poisson* ps64, double* p64, float* u, float* p, hipStream_t s) {
  const GridDev& g = G->g;
  int rc = no_halo(G, "project (f32)");
  if (rc) return rc;
  dim3 gi(cdiv(g.ip_hi[0] - g.ip_lo[0], 64), cdiv(g.ip_hi[1] - g.ip_lo[1], 4), g.D == 3 ? g.ip_hi[2] - g.ip_lo[2] : 1);
  if (g.D == 2)
    hipLaunchKernelGGL((k32g_div<2, double, true>), gi, dim3(64, 4), 0, s, g, u, p64);
  else
    hipLaunchKernelGGL((k32g_div<3, double, true>), gi, dim3(64, 4), 0, s, g, u, p64);
  INS_LAUNCH_CHECK();
  if ((rc = ins_k_poisson_solve(ps64, p64, s))) return rc;
  hipLaunchKernelGGL(k32g_round, flat_grid(G->ncell), dim3(256), 0, s, G->ncell, p64, p);
  INS_LAUNCH_CHECK();
  if ((rc = ins_k32g_apply_bc_p(G, p, s))) return rc;
  dim3 gp(cdiv(g.N[0] - 2, 64), cdiv(g.N[1] - 2, 4), g.D == 3 ? g.N[2] - 2 : 1);
  if (g.D == 2)
    hipLaunchKernelGGL(k32g_applypressure<2>, gp, dim3(64, 4), 0, s, g, u, p);
  else
    hipLaunchKernelGGL(k32g_applypressure<3>, gp, dim3(64, 4), 0, s, g, u, p);
  INS_LAUNCH_CHECK();
  return INS_OK;
}
