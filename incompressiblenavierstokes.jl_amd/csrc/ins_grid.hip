// Grid handle: device copies of the reference's 1-D metric vectors (grid.jl:100-276) plus the
// reciprocal / masked-reciprocal tables the stencil kernels use instead of fp64 divisions.
#include <algorithm>
#include <cmath>
#include <cstdarg>

#include "ins_internal.h"

static thread_local char g_err[1024] = "";

void ins_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ins_version(void) { return 100; }
extern "C" const char* ins_last_error(void) { return g_err; }

extern "C" int ins_set_device(int device) {
  INS_HIP_TRY(hipSetDevice(device));
  return INS_OK;
}

extern "C" int ins_sync(void* stream) {
  INS_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  return INS_OK;
}

extern "C" int ins_grid_create(const ins_grid_desc_t* d, ins_grid_t** out) {
  INS_REQUIRE(d && out, "null argument");
  INS_REQUIRE(d->D == 2 || d->D == 3, "D must be 2 or 3");
  const int D = d->D;
  for (int a = 0; a < D; ++a) {
    INS_REQUIRE(d->N[a] >= 3, "N[a] must be >= 3 (one DOF plus two ghosts)");
    INS_REQUIRE(d->dx[a] && d->dxu[a], "null metric vector");
    INS_REQUIRE(d->ip_lo[a] >= 1 && d->ip_hi[a] <= d->N[a] - 1 && d->ip_lo[a] < d->ip_hi[a], "bad Ip range");
    for (int b = 0; b < D; ++b) {
      INS_REQUIRE(d->A1[a][b] && d->A2[a][b], "null interpolation weights");
      INS_REQUIRE(d->iu_lo[a][b] >= 1 && d->iu_hi[a][b] <= d->N[b] - 1 && d->iu_lo[a][b] <= d->iu_hi[a][b], "bad Iu range");
    }
    for (int s = 0; s < 2; ++s) INS_REQUIRE(d->bc[a][s] >= INS_BC_PERIODIC && d->bc[a][s] <= INS_BC_HALO, "bad BC code");
    INS_REQUIRE((d->bc[a][0] == INS_BC_PERIODIC) == (d->bc[a][1] == INS_BC_PERIODIC), "periodic BC must be on both sides");
  }

  ins_grid* G = new ins_grid();
  G->desc = *d;
  // Table order in the slab: per direction {dx, dxu, rdx, rdxu, mdx, mdxu}, then A1[a][b], A2[a][b].
  std::vector<double>& h = G->host;
  size_t off_dx[3][6];
  size_t off_A[3][3][2];
  for (int a = 0; a < D; ++a) {
    const int n = d->N[a];
    for (int t = 0; t < 6; ++t) {
      off_dx[a][t] = h.size();
      for (int i = 0; i < n; ++i) {
        const double w = (t % 2 == 0) ? d->dx[a][i] : d->dxu[a][i];
        double v;
        if (t < 2)
          v = w;
        else if (t < 4)
          v = 1.0 / w;
        else
          v = (w > 2 * INS_EPS) ? 1.0 / w : 0.0;
        h.push_back(v);
      }
    }
  }
  for (int a = 0; a < D; ++a)
    for (int b = 0; b < D; ++b)
      for (int t = 0; t < 2; ++t) {
        off_A[a][b][t] = h.size();
        const double* src = t == 0 ? d->A1[a][b] : d->A2[a][b];
        // A[a][b] is indexed along direction b and has N[b] entries (grid.jl:227-248)
        for (int i = 0; i < d->N[b]; ++i) h.push_back(src[i]);
      }
  G->dev_count = h.size();
  hipError_t e = hipMalloc(&G->dev, h.size() * sizeof(double));
  if (e != hipSuccess) {
    delete G;
    ins_set_error("hipMalloc(grid tables): %s", hipGetErrorString(e));
    return INS_ERR_HIP;
  }
  e = hipMemcpy(G->dev, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(G->dev);
    delete G;
    ins_set_error("hipMemcpy(grid tables): %s", hipGetErrorString(e));
    return INS_ERR_HIP;
  }

  GridDev& g = G->g;
  memset(&g, 0, sizeof(g));
  g.D = D;
  long long stride = 1;
  for (int a = 0; a < 3; ++a) {
    g.N[a] = a < D ? d->N[a] : 1;
    g.sx[a] = stride;
    stride *= g.N[a];
  }
  g.sc = stride;
  G->ncell = stride;
  for (int a = 0; a < D; ++a) {
    g.dx[a] = G->dev + off_dx[a][0];
    g.dxu[a] = G->dev + off_dx[a][1];
    g.rdx[a] = G->dev + off_dx[a][2];
    g.rdxu[a] = G->dev + off_dx[a][3];
    g.mdx[a] = G->dev + off_dx[a][4];
    g.mdxu[a] = G->dev + off_dx[a][5];
    G->desc.dx[a] = h.data() + off_dx[a][0];
    G->desc.dxu[a] = h.data() + off_dx[a][1];
    for (int b = 0; b < D; ++b) {
      g.A1[a][b] = G->dev + off_A[a][b][0];
      g.A2[a][b] = G->dev + off_A[a][b][1];
      G->desc.A1[a][b] = h.data() + off_A[a][b][0];
      G->desc.A2[a][b] = h.data() + off_A[a][b][1];
      g.iu_lo[a][b] = d->iu_lo[a][b];
      g.iu_hi[a][b] = d->iu_hi[a][b];
    }
    g.ip_lo[a] = d->ip_lo[a];
    g.ip_hi[a] = d->ip_hi[a];
    for (int s = 0; s < 2; ++s) {
      g.bc[a][s] = d->bc[a][s];
      for (int c = 0; c < 3; ++c) g.bc_u[a][s][c] = d->bc_u[a][s][c];
    }
  }
  // dummy (degenerate) third direction for D = 2 so generic index math stays in range
  for (int a = D; a < 3; ++a) {
    g.ip_lo[a] = 0;
    g.ip_hi[a] = 1;
    for (int b = 0; b < 3; ++b) {
      g.iu_lo[b][a] = 0;
      g.iu_hi[b][a] = 1;
    }
  }

  // default_psolver's classification (pressure.jl:85-98): isapprox with rtol = sqrt(eps)
  G->all_periodic = true;
  G->uniform = true;
  for (int a = 0; a < D; ++a) {
    if (d->bc[a][0] != INS_BC_PERIODIC || d->bc[a][1] != INS_BC_PERIODIC) G->all_periodic = false;
    G->h[a] = d->dx[a][0];
    for (int i = 0; i < d->N[a]; ++i) {
      const double x = d->dx[a][i], y = d->dx[a][0];
      if (std::fabs(x - y) > 1.4901161193847656e-08 * std::fmax(std::fabs(x), std::fabs(y))) G->uniform = false;
    }
  }
  // Flux-kernel dispatch flags.  all_dof: Iu[α] == interior box for every α.  uniform_exact: the metric
  // records the flux kernel reads (index range 0..N-2, see ins_fast3d_flux.hip) are bitwise constant, so
  // the constant-record kernel returns exactly what the table-driven one does.
  G->all_dof = true;
  G->uniform_exact = true;
  for (int a = 0; a < D; ++a)
    for (int b = 0; b < D; ++b)
      if (d->iu_lo[a][b] != 1 || d->iu_hi[a][b] != d->N[b] - 1) G->all_dof = false;
  // "Constant" = equal to within the rounding of the coordinates they were differenced from: the spacings of range(0, 2π, n+1) or
  // range(0, 1, 385) scatter by ~N ulp (each x_i carries half an ulp of the box length), and taking one value for all of them perturbs
  // every coefficient by <= 4 N eps <= 2.5e-13 relative — inside the 1e-12 parity tolerance — while the constant-record kernels
  // (in-register pressure correction, stage-velocity basis, chained steps) make the step 1.5x faster than the table-driven ones
  // (256³ on [0, 2π]³: 4.2 -> 2.8 ms).  Boxes finer than that bound keep the tables; INS_UNIFORM_BITWISE=1 restores the bitwise test.
  const bool bitwise = ins_opt(OPT_INS_UNIFORM_BITWISE) != 0;
  int nmax = 1;
  for (int a = 0; a < D; ++a) nmax = std::max(nmax, (int)d->N[a]);
  const double tol = bitwise ? 0.0 : std::min(2.5e-13, 4.0 * nmax * 2.220446049250313e-16);
  auto close = [&](double x, double y) { return std::fabs(x - y) <= tol * std::fabs(y); };
  for (int a = 0; a < D && G->uniform_exact; ++a) {
    const int n = d->N[a];
    for (int i = 0; i <= n - 2 && G->uniform_exact; ++i) {
      bool same = close(d->dx[a][i + 1], d->dx[a][2]) && close(d->dxu[a][i], d->dxu[a][1]) && close(d->dx[a][i], d->dx[a][1]);
      for (int b = 0; b < D; ++b) same = same && close(d->A2[b][a][i], 0.5) && close(d->A1[b][a][i + 1], 0.5);
      if (!same) G->uniform_exact = false;
    }
  }
  e = hipMalloc(&G->red_dev, 4096 * sizeof(double));
  if (e == hipSuccess) e = hipHostMalloc(&G->red_host, 4096 * sizeof(double));
  if (e != hipSuccess) {
    ins_grid_destroy(G);
    ins_set_error("grid scratch allocation: %s", hipGetErrorString(e));
    return INS_ERR_HIP;
  }
  *out = G;
  return INS_OK;
}

extern "C" int ins_grid_is_uniform_exact(const ins_grid_t* G) { return G && G->uniform_exact && G->all_dof; }

extern "C" int ins_grid_destroy(ins_grid_t* G) {
  if (!G) return INS_OK;
  if (G->dev) (void)hipFree(G->dev);
  if (G->red_dev) (void)hipFree(G->red_dev);
  if (G->rec_dev) (void)hipFree(G->rec_dev);
  if (G->rec_diff_dev) (void)hipFree(G->rec_diff_dev);
  if (G->red_host) (void)hipHostFree(G->red_host);
  delete G;
  return INS_OK;
}
