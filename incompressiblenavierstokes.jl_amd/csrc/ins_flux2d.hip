// K1 (+ K6) for 2-D uniform periodic grids: momentum! = fill + convection-diffusion (operators.jl:647-690, 967-976) in flux form with the
// Runge-Kutta stage combination (step_explicit_runge_kutta.jl:35-38) as its epilogue.  The 2-D sibling of ins_flux64.hip:
//   * lanes run along x; a wavefront owns 62 columns (+ one halo lane each side) of a y-chunk and marches through its rows,
//   * x neighbours come from DPP wavefront shifts, the rows j-1, j, j+1 live in registers, the y-face fluxes are carried from row to row,
//   * every face flux is computed once and used by both cells it separates (the reference computes each twice),
// so a cell costs 2 loads + the epilogue operands + 2 stores instead of the ~30 eight-byte loads of the one-cell-per-work-item
// kernel, which is bound by the vector-L1 rate.  Algorithmic traffic 32 B/cell (+ 16 per epilogue operand).
// The input has valid ghost cells (apply_bc_u! ran): no wrap logic.  Same arithmetic as the reference up to the order of the
// face-flux differences (<= 1e-15 relative; parity tolerance 1e-12).
#include "ins_internal.h"

namespace {

template <unsigned CTRL>
__device__ __forceinline__ double dpp(double v) {  // lanes without a source lane keep their own value
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_next(double v) { return dpp<0x130>(v); }  // wave_shl:1  lane l <- l+1
__device__ __forceinline__ double from_prev(double v) { return dpp<0x138>(v); }  // wave_shr:1  lane l <- l-1

struct Flux2dArgs {
  const double* u;
  const double* p;   // CORR: the unpadded pressure of the projection of `u` (n0 x n1), u = u* - ∇p applied row by row in registers
  double* F;
  int N0, N1;
  long long sc;
  int yc;            // rows per chunk
  double vx, vy;     // ν/Δx, ν/Δy
  double rx, ry;     // 1/Δx, 1/Δy
  RkEpi epi;
};

// CORR (the 2-D sibling of the 3-D kernels' in-register pressure correction): `u` is the previous stage's UNCORRECTED u* (interior volumes only) and `p` the
// unpadded pressure of its projection; every row is corrected as it is loaded, u = u* - ∇p (applypressure!, operators.jl:225-233), rows and columns
// outside the interior are read through their periodic images.  The stage loop then needs no gradient-subtract / ghost-fill pass between its stages.
template <bool FUSE, bool CORR = false>
__global__ __launch_bounds__(256) void k_flux2d(Flux2dArgs a) {
  const int lane = threadIdx.x;
  const int x0 = 1 + (int)blockIdx.x * 62;                       // first output column of this wavefront
  const int j0 = 1 + ((int)blockIdx.y * 4 + (int)threadIdx.y) * a.yc;  // first output row
  if (j0 > a.N1 - 2) return;
  const int j1 = min(j0 + a.yc, a.N1 - 1);
  const int ci = min(x0 - 1 + lane, a.N0 - 1);
  const bool xout = lane >= 1 && lane <= 62 && x0 - 1 + lane <= a.N0 - 2;
  const int n0 = a.N0 - 2, n1 = a.N1 - 2;
  auto wrap = [](int q, int n) { return q < 0 ? q + n : (q >= n ? q - n : q); };
  const int ii = CORR ? wrap(ci - 1, n0) : 0, iip = CORR ? wrap(ii + 1, n0) : 0;  // interior column of this lane and its right neighbour
  const double* u0 = a.u + (CORR ? ii + 1 : ci);
  const double* v0 = a.u + a.sc + (CORR ? ii + 1 : ci);
  auto row = [&](int j) { return (long long)j * a.N0; };
  double pcur = 0.0;  // CORR: p(ii, row being loaded), carried from the previous load
  // one row of the two components, corrected when CORR (j: padded row index, possibly a ghost row -> its periodic image)
  auto load_row = [&](int j, double& uu, double& vv, bool first) {
    if (!CORR) {
      uu = u0[row(j)];
      vv = v0[row(j)];
      return;
    }
    const int jj = wrap(j - 1, n1), jjp = wrap(jj + 1, n1);
    if (first) pcur = a.p[ii + (long long)n0 * jj];
    const double pnext = a.p[ii + (long long)n0 * jjp], px = a.p[iip + (long long)n0 * jj];
    uu = u0[row(jj + 1)] - (px - pcur) * a.rx;
    vv = v0[row(jj + 1)] - (pnext - pcur) * a.ry;
    pcur = pnext;
  };
  double um, vm, uc, vc;
  load_row(j0 - 1, um, vm, true);
  load_row(j0, uc, vc, false);
  // y-face fluxes between rows j0-1 and j0
  double gyu = a.vy * (uc - um) - 0.5 * (um + uc) * 0.5 * (vm + from_next(vm));
  double gyv = a.vy * (vc - vm) - 0.25 * (vm + vc) * (vm + vc);
  for (int j = j0; j < j1; ++j) {
    double up, vp;
    load_row(j + 1, up, vp, false);
    // epilogue operands first: their loads fly during the flux arithmetic
    double su = 0.0, sv = 0.0;
    const long long c = row(j) + ci;
    if (FUSE) {
      su = a.epi.ustart ? a.epi.ustart[c] : uc;
      sv = a.epi.ustart ? a.epi.ustart[a.sc + c] : vc;
      for (int q = 0; q < a.epi.n; ++q) {
        su += a.epi.coef[q] * a.epi.k[q][c];
        sv += a.epi.coef[q] * a.epi.k[q][a.sc + c];
      }
    }
    // right-face x fluxes of this lane's cell (left faces = the left neighbour's right faces)
    const double un = from_next(uc), vn = from_next(vc);
    const double gxu = a.vx * (un - uc) - 0.25 * (uc + un) * (uc + un);
    const double gxv = a.vx * (vn - vc) - 0.5 * (vc + vn) * 0.5 * (uc + up);
    // y-face fluxes between rows j and j+1
    const double gyu2 = a.vy * (up - uc) - 0.5 * (uc + up) * 0.5 * (vc + vn);
    const double gyv2 = a.vy * (vp - vc) - 0.25 * (vc + vp) * (vc + vp);
    const double fu = (gxu - from_prev(gxu)) * a.rx + (gyu2 - gyu) * a.ry;
    const double fv = (gxv - from_prev(gxv)) * a.rx + (gyv2 - gyv) * a.ry;
    if (xout) {
      if (FUSE) {
        a.epi.ustar[c] = su + a.epi.coef_self * fu;
        a.epi.ustar[a.sc + c] = sv + a.epi.coef_self * fv;
        if (CORR && a.epi.ustart_out) {  // first stage of a chained step: the corrected input is this step's ustart
          a.epi.ustart_out[c] = uc;
          a.epi.ustart_out[a.sc + c] = vc;
        }
      }
      if (!FUSE || a.epi.write_k) {
        a.F[c] = fu;
        a.F[a.sc + c] = fv;
      }
    }
    gyu = gyu2;
    gyv = gyv2;
    um = uc;
    vm = vc;
    uc = up;
    vc = vp;
  }
}

}  // namespace

bool ins_flux2d_supported(const ins_grid* G) {
  const bool off = ins_opt(OPT_INS_DISABLE_FLUX2D) != 0;  // A/B switch
  // uniform_exact: spacings and interpolation weights constant to the rounding of the coordinates (ins_grid.hip), so one h per direction is exact enough
  return !off && G->g.D == 2 && G->all_periodic && G->all_dof && G->uniform_exact && G->g.N[0] >= 4 && G->g.N[1] >= 4;
}

// epi == nullptr: plain momentum! into F (ghost ring of F untouched: the caller's F has a zero ring or does not read it)
int ins_k_flux2d(const ins_grid* G, double visc, const double* u, double* F, const RkEpi* epi, hipStream_t s, const double* pI) {
  const GridDev& g = G->g;
  Flux2dArgs a;
  a.u = u;
  a.p = pI;
  a.F = F;
  a.N0 = g.N[0];
  a.N1 = g.N[1];
  a.sc = g.sc;
  const int n1 = g.N[1] - 2;
  // rows per chunk: long chunks amortise the two start-up rows, but a small grid needs its rows spread over enough wavefronts
  const int ntx = (int)cdiv(g.N[0] - 2, 62);
  const int want_chunks = std::max(1, 2048 / ntx);
  a.yc = std::min(64, std::max(4, n1 / want_chunks));
  a.rx = 1.0 / G->h[0];
  a.ry = 1.0 / G->h[1];
  a.vx = visc * a.rx;
  a.vy = visc * a.ry;
  if (epi) a.epi = *epi;
  else memset(&a.epi, 0, sizeof(a.epi));
  dim3 block(64, 4, 1), grid(cdiv(g.N[0] - 2, 62), cdiv(cdiv(n1, a.yc), 4), 1);
  if (pI && !epi) {
    ins_set_error("ins_k_flux2d: the correcting form exists for the fused stage kernel only");
    return INS_ERR_INVALID;
  }
  if (pI)
    hipLaunchKernelGGL((k_flux2d<true, true>), grid, block, 0, s, a);
  else if (epi)
    hipLaunchKernelGGL(k_flux2d<true>, grid, block, 0, s, a);
  else
    hipLaunchKernelGGL(k_flux2d<false>, grid, block, 0, s, a);
  INS_LAUNCH_CHECK();
  return INS_OK;
}
