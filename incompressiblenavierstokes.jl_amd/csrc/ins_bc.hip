// Ghost-volume fill (boundary_conditions.jl:159-206, 276-318, 344-388, 414-502) and the blocking
// box reductions used by diagnostics and the CG solver.
#include "ins_internal.h"

namespace {

// One launch per direction β (the reference sweeps β = 1..D in order so that edges and corners come
// out right, boundary_conditions.jl:162-165); work-items span the full padded extent of the other
// directions (`boundary()`, boundary_conditions.jl:97-103) times the D components.
//
// Plane coordinates (q0, q1) enumerate the two directions != β in memory order.
template <int D>
__global__ __launch_bounds__(256) void k_bc_u(GridDev g, double* __restrict__ u, int be, int dudt, const double* const* planes) {
  const int o0 = be == 0 ? 1 : 0;             // fastest direction != be
  const int o1 = be == 2 ? 1 : 2;             // slowest direction != be (3-D only)
  const int q0 = blockIdx.x * 256 + threadIdx.x;
  const int q1 = D == 3 ? (int)blockIdx.y : 0;
  const int al = blockIdx.z;
  if (q0 >= g.N[o0]) return;
  const long long base = q0 * g.sx[o0] + (D == 3 ? q1 * g.sx[o1] : 0);
  const long long sb = g.sx[be];
  double* ua = u + al * g.sc;
  const int bcl = g.bc[be][0], bcr = g.bc[be][1];

  if (bcl == INS_BC_PERIODIC) {  // boundary_conditions.jl:276-288 (both sides in one go)
    const int ia = g.ip_lo[be] - 1, ib = g.ip_hi[be];
    ua[base + ia * sb] = ua[base + (ib - 1) * sb];
    ua[base + ib * sb] = ua[base + (ia + 1) * sb];
    return;
  }
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int bc = side ? bcr : bcl;
    if (bc == INS_BC_HALO) continue;  // filled by the caller (slab neighbour exchange)
    const int i = side ? g.iu_hi[al][be] : g.iu_lo[al][be] - 1;
    const int jn = side ? i - 1 : i + 1;
    double* dst = ua + base + i * sb;
    if (bc == INS_BC_DIRICHLET) {  // boundary_conditions.jl:344-375
      const double* pl = planes ? planes[(be * 2 + side) * 3 + al] : nullptr;
      double v;
      if (pl)
        v = pl[q0 + (D == 3 ? (long long)q1 * g.N[o0] : 0)];
      else
        v = dudt ? 0.0 : g.bc_u[be][side][al];
      *dst = v;
    } else if (bc == INS_BC_SYMMETRIC) {  // boundary_conditions.jl:414-428
      *dst = (al == be) ? 0.0 : ua[base + jn * sb];
    } else if (bc == INS_BC_PRESSURE) {  // boundary_conditions.jl:472-482
      *dst = ua[base + jn * sb];
    }
  }
}

template <int D>
__global__ __launch_bounds__(256) void k_bc_p(GridDev g, double* __restrict__ p, int be, long long fstride = 0) {
  p += (long long)blockIdx.z * fstride;  // several scalar fields in one launch (the stress components of the closure, ins_k_apply_bc_p_fields)
  const int o0 = be == 0 ? 1 : 0;
  const int o1 = be == 2 ? 1 : 2;
  const int q0 = blockIdx.x * 256 + threadIdx.x;
  const int q1 = D == 3 ? (int)blockIdx.y : 0;
  if (q0 >= g.N[o0]) return;
  const long long base = q0 * g.sx[o0] + (D == 3 ? q1 * g.sx[o1] : 0);
  const long long sb = g.sx[be];
  const int bcl = g.bc[be][0], bcr = g.bc[be][1];
  const int ia = g.ip_lo[be] - 1, ib = g.ip_hi[be];
  if (bcl == INS_BC_PERIODIC) {  // boundary_conditions.jl:306-318
    p[base + ia * sb] = p[base + (ib - 1) * sb];
    p[base + ib * sb] = p[base + (ia + 1) * sb];
    return;
  }
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int bc = side ? bcr : bcl;
    const int i = side ? ib : ia;
    const int jn = side ? i - 1 : i + 1;
    if (bc == INS_BC_SYMMETRIC)  // boundary_conditions.jl:445-453
      p[base + i * sb] = p[base + jn * sb];
    else if (bc == INS_BC_PRESSURE)  // boundary_conditions.jl:497-502
      p[base + i * sb] = 0.0;
    // Dirichlet: no-op (boundary_conditions.jl:388); HALO: caller
  }
}

// ------------------------------------------------------------------------------------------------
// Box reductions: per-block partials (wavefront shuffles + LDS), finished on the host — these calls
// are blocking by contract (they return a scalar), like the reference's dot/sum/minimum.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_reduce(double v, int op) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(v, off, 64);
    v = op == 1 ? fmax(v, o) : op == 2 ? fmin(v, o) : v + o;
  }
  return v;
}

__global__ __launch_bounds__(256) void k_reduce(GridDev g, int op, const double* __restrict__ a, const double* __restrict__ b,
                                                int lo0, int lo1, int lo2, int n0, int n1, int n2, double* __restrict__ partial) {
  __shared__ double lds[4];
  const long long total = (long long)n0 * n1 * n2;
  double acc = op == 2 ? INFINITY : 0.0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int i = (int)(t % n0);
    const long long r = t / n0;
    const int j = (int)(r % n1), k = (int)(r / n1);
    const long long c = (lo0 + i) + (lo1 + j) * g.sx[1] + (lo2 + k) * g.sx[2];
    const double x = a[c];
    if (op == 0)
      acc += x * b[c];
    else if (op == 1)
      acc = fmax(acc, fabs(x));
    else if (op == 2)
      acc = fmin(acc, x);
    else if (op == 3)
      acc += x;
    else
      acc += x * x;
  }
  acc = wave_reduce(acc, op);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = lds[0];
    for (int q = 1; q < 4; ++q) v = op == 1 ? fmax(v, lds[q]) : op == 2 ? fmin(v, lds[q]) : v + lds[q];
    partial[blockIdx.x] = v;
  }
}

}  // namespace

int ins_k_apply_bc_u(const ins_grid* G, double* u, int dudt, const double* const* planes, hipStream_t s) {
  const GridDev& g = G->g;
  for (int be = 0; be < g.D; ++be) {
    if (g.bc[be][0] == INS_BC_HALO && g.bc[be][1] == INS_BC_HALO) continue;
    const int o0 = be == 0 ? 1 : 0, o1 = be == 2 ? 1 : 2;
    dim3 grid(cdiv(g.N[o0], 256), g.D == 3 ? g.N[o1] : 1, g.D);
    if (g.D == 2)
      hipLaunchKernelGGL(k_bc_u<2>, grid, dim3(256), 0, s, g, u, be, dudt, planes);
    else
      hipLaunchKernelGGL(k_bc_u<3>, grid, dim3(256), 0, s, g, u, be, dudt, planes);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}

// apply_bc_p! on nf scalar fields laid out back to back (field f at p + f·ncell): one launch per direction for all of them
int ins_k_apply_bc_p_fields(const ins_grid* G, double* p, int nf, hipStream_t s) {
  const GridDev& g = G->g;
  for (int be = 0; be < g.D; ++be) {
    const int l = g.bc[be][0], r = g.bc[be][1];
    const bool noop = (l == INS_BC_DIRICHLET || l == INS_BC_HALO) && (r == INS_BC_DIRICHLET || r == INS_BC_HALO);
    if (noop) continue;
    const int o0 = be == 0 ? 1 : 0, o1 = be == 2 ? 1 : 2;
    dim3 grid(cdiv(g.N[o0], 256), g.D == 3 ? g.N[o1] : 1, nf);
    if (g.D == 2)
      hipLaunchKernelGGL(k_bc_p<2>, grid, dim3(256), 0, s, g, p, be, G->ncell);
    else
      hipLaunchKernelGGL(k_bc_p<3>, grid, dim3(256), 0, s, g, p, be, G->ncell);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}
int ins_k_apply_bc_p(const ins_grid* G, double* p, hipStream_t s) { return ins_k_apply_bc_p_fields(G, p, 1, s); }

// op: 0 sum(a*b), 1 max|a|, 2 min(a), 3 sum(a), 4 sum(a*a).  Blocking.
int ins_k_reduce(const ins_grid* G, int op, const double* a, const double* b, const int lo[3], const int hi[3], double* out,
                 hipStream_t s) {
  const GridDev& g = G->g;
  const int n0 = hi[0] - lo[0], n1 = hi[1] - lo[1], n2 = g.D == 3 ? hi[2] - lo[2] : 1;
  const int l2 = g.D == 3 ? lo[2] : 0;
  const long long total = (long long)n0 * n1 * n2;
  int nblk = (int)((total + 255) / 256);
  if (nblk > 2048) nblk = 2048;
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(k_reduce, dim3(nblk), dim3(256), 0, s, g, op, a, b, lo[0], lo[1], l2, n0, n1, n2, G->red_dev);
  INS_LAUNCH_CHECK();
  INS_HIP_TRY(hipMemcpyAsync(G->red_host, G->red_dev, nblk * sizeof(double), hipMemcpyDeviceToHost, s));
  INS_HIP_TRY(hipStreamSynchronize(s));
  double v = op == 2 ? INFINITY : 0.0;
  for (int i = 0; i < nblk; ++i) v = op == 1 ? fmax(v, G->red_host[i]) : op == 2 ? fmin(v, G->red_host[i]) : v + G->red_host[i];
  *out = v;
  return INS_OK;
}

extern "C" int ins_apply_bc_u_f64(const ins_grid_t* G, double* u, int dudt, const double* const* planes, void* stream) {
  INS_REQUIRE(G && u, "null argument");
  if (!planes) return ins_k_apply_bc_u(G, u, dudt, nullptr, as_stream(stream));
  // `planes` is a host array of 18 device pointers; stage it in device memory for the kernels.
  const double** dplanes = nullptr;
  INS_HIP_TRY(hipMalloc(&dplanes, 18 * sizeof(double*)));
  hipError_t e = hipMemcpy(dplanes, planes, 18 * sizeof(double*), hipMemcpyHostToDevice);
  int rc = INS_OK;
  if (e != hipSuccess) {
    ins_set_error("hipMemcpy(planes): %s", hipGetErrorString(e));
    rc = INS_ERR_HIP;
  } else {
    rc = ins_k_apply_bc_u(G, u, dudt, dplanes, as_stream(stream));
    if (rc == INS_OK && hipStreamSynchronize(as_stream(stream)) != hipSuccess) rc = INS_ERR_HIP;
  }
  (void)hipFree(dplanes);
  return rc;
}

extern "C" int ins_apply_bc_p_f64(const ins_grid_t* G, double* p, void* stream) {
  INS_REQUIRE(G && p, "null argument");
  return ins_k_apply_bc_p(G, p, as_stream(stream));
}
