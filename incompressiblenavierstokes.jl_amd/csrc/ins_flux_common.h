// Shared by the stage kernels of uniform periodic boxes (ins_flux64.hip: one x-column per lane; ins_flux128.hip: two): the constant metric records,
// the kernel argument block and the face-flux expression.
#pragma once
#include "ins_internal.h"

namespace {

// The constant metric record of one direction.  The uniform half weights ¼ = ½·½ of the convective flux are folded into the
// constants (4× the diffusion coefficients, ¼× the width reciprocals): power-of-two scalings, so every result is bitwise
// what the unscaled expression gives, and twelve multiplications per cell disappear.
struct Dir {
  double vs, vo;  // 4ν/Δ (α == β), 4ν/Δu (α != β)
  double rs, ro;  // ¼/Δu, ¼/Δ
  double gs;      // 1/Δu (pressure gradient, CORR)
};
// the same record in the arithmetic type T of the kernel (double, or float for the `_f32` entry points: the host computes the
// constants in double and rounds once)
template <typename T>
struct DirT {
  T vs, vo, rs, ro, gs;
  __device__ DirT(const Dir& d) : vs((T)d.vs), vo((T)d.vo), rs((T)d.rs), ro((T)d.ro), gs((T)d.gs) {}
};

struct FluxArgs {  // field pointers are T* of the kernel instantiation (RkEpi's pointers likewise)
  const void* u;
  const void* pI;
  void* F;
  long long sc;  // component stride (elements)
  int N0, N1, N2;
  int zc, ntx, nty, ntz;
  // plane range of this launch: chunk t covers [k_lo + t zc, min(.. + zc, k_hi)); kB > 0: two chunks, [k_lo, k_lo + zc) and [kB, kB + zc)
  // (the host runs the planes that read no ghost plane beside the halo exchange, then the two thin boundary ranges)
  int k_lo, k_hi, kB;
  int nt;   // cache-policy experiment on the result stores (INS_FLUX64_NT)
  int bar;  // one workgroup barrier per plane: the y-stacked wavefronts of a workgroup stay on the same plane (their shared halo rows are then cache hits)
  Dir X, Y, Z;
  RkEpi epi;
  int tm;      // temperature stage inside the kernel (EXTRA instantiation, CORR = 0)
  TempEpi te;
};

// 4 × face flux:  4ν(up - uc)/Δb - (uc + up)(ub0 + ub1)        [ν(up - uc)/Δb - ½(uc + up)·½(ub0 + ub1), times 4]
template <typename T>
__device__ __forceinline__ T flux(T uc, T up, T ub0, T ub1, T vd4) {
  return (up - uc) * vd4 - (uc + up) * (ub0 + ub1);
}

static Dir make_dir(const ins_grid* G, int d, double visc) {
  // the constant record ins_fast3d_flux.hip's UNIFORM kernels read (index 1): same fp64 operations, on the host
  const double dxu = G->desc.dxu[d][1], dx1 = G->desc.dx[d][1], dx2 = G->desc.dx[d][2];
  Dir r;
  r.vs = 4.0 * (visc * (dx2 > 2 * INS_EPS ? 1.0 / dx2 : 0.0));
  r.vo = 4.0 * (visc * (dxu > 2 * INS_EPS ? 1.0 / dxu : 0.0));
  r.gs = 1.0 / dxu;
  r.rs = 0.25 * r.gs;
  r.ro = 0.25 * (1.0 / dx1);
  return r;
}

}  // namespace
