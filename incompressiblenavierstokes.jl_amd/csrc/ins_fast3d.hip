// K1 — fused 3-D momentum-RHS stencil for gfx950 (convection_diffusion_kernel! + fill!(F,0),
// operators.jl:647-690, 971), any boundary conditions, stretched grids.
//
// Shape: a 256-thread workgroup (4 wavefronts = 4 y-rows of 64 x-lanes) owns a 64 x (4R) tile of
// (x, y) and marches through a chunk of z planes.  The three velocity components of planes
// k-1, k, k+1 (+ 1-cell halo incl. the in-plane diagonals I - e_b + e_a) live in a 4-slot LDS plane
// ring, so every u value is fetched from global memory once per workgroup; plane k+2 is prefetched
// into registers while plane k is computed and lands in the 4th slot, which keeps the loop at ONE
// barrier per plane.  x-direction metrics sit in VGPRs for the whole march; y/z metrics are
// wavefront-uniform and come through scalar loads.  Algorithmic traffic: 48 B / cell.
#include <cstdlib>

#include "ins_internal.h"

namespace {

constexpr int TX = 64;

template <int R>
struct Tile {
  static constexpr int TY = 4 * R;
  static constexpr int TYH = TY + 2;
  static constexpr int PITCH = TX + 2;
  static constexpr int PLANE = TYH * PITCH;   // one component, one plane (with halo)
  static constexpr int SLOT = 3 * PLANE;      // three components
  static constexpr int NLOAD = (SLOT + 255) / 256;
};

// Metrics of one direction d at index idx (reciprocal / masked-reciprocal tables, see ins_internal.h)
// and the interpolation weights A[be][d] read with an index along d (operators.jl:672-675).
struct DirMet {
  double rdx, rdxu, mdx_i, mdx_ip, mdxu_im, mdxu_i;
  double w[3][4];  // w[be] = {A2[idx-(d==be)], A1[idx+(d!=be)], A2[idx], A1[idx+1]}
};

template <int d>
__device__ __forceinline__ void load_dirmet(const GridDev& g, int idx, DirMet& m) {
  m.rdx = g.rdx[d][idx];
  m.rdxu = g.rdxu[d][idx];
  m.mdx_i = g.mdx[d][idx];
  m.mdx_ip = g.mdx[d][idx + 1];
  m.mdxu_im = g.mdxu[d][idx - 1];
  m.mdxu_i = g.mdxu[d][idx];
#pragma unroll
  for (int be = 0; be < 3; ++be) {
    const double* A1 = g.A1[be][d];
    const double* A2 = g.A2[be][d];
    m.w[be][2] = A2[idx];
    m.w[be][3] = A1[idx + 1];
    if (be == d) {
      m.w[be][0] = A2[idx - 1];
      m.w[be][1] = A1[idx];
    } else {
      m.w[be][0] = m.w[be][2];
      m.w[be][1] = m.w[be][3];
    }
  }
}

template <int R>
struct PlaneView {
  const double* pm;  // plane k-1
  const double* pc;  // plane k
  const double* pp;  // plane k+1
  int o;             // cy * PITCH + cx of this cell
  template <int C, int DX, int DY, int DZ>
  __device__ __forceinline__ double at() const {
    const double* base = DZ < 0 ? pm : (DZ > 0 ? pp : pc);
    return base[C * Tile<R>::PLANE + o + DY * Tile<R>::PITCH + DX];
  }
};

// One (al, be) term of operators.jl:666-685.
template <int R, int AL, int BE>
__device__ __forceinline__ double pair_term(const PlaneView<R>& v, const DirMet (&M)[3], double visc, double uc) {
  constexpr int bx = BE == 0, by = BE == 1, bz = BE == 2;
  constexpr int ax = AL == 0, ay = AL == 1, az = AL == 2;
  const double um = v.template at<AL, -bx, -by, -bz>();
  const double up = v.template at<AL, bx, by, bz>();
  const double ub_m = v.template at<BE, -bx, -by, -bz>();
  const double ub_ma = v.template at<BE, ax - bx, ay - by, az - bz>();
  const double ub_0 = v.template at<BE, 0, 0, 0>();
  const double ub_a = v.template at<BE, ax, ay, az>();
  const DirMet& mb = M[BE];
  const DirMet& ma = M[AL];
  const double r = AL == BE ? mb.rdxu : mb.rdx;
  const double da = AL == BE ? mb.mdx_i : mb.mdxu_im;
  const double db = AL == BE ? mb.mdx_ip : mb.mdxu_i;
  const double d1 = (uc - um) * da;
  const double d2 = (up - uc) * db;
  const double uab1 = (um + uc) * 0.5;
  const double uab2 = (uc + up) * 0.5;
  const double uba1 = ma.w[BE][0] * ub_m + ma.w[BE][1] * ub_ma;
  const double uba2 = ma.w[BE][2] * ub_0 + ma.w[BE][3] * ub_a;
  return (visc * (d2 - d1) - (uab2 * uba2 - uab1 * uba1)) * r;
}

template <int R>
__global__ __launch_bounds__(256) void k_momentum_tiled(GridDev g, double visc, const double* __restrict__ u,
                                                        double* __restrict__ F, int zc) {
  using T = Tile<R>;
  __shared__ double lds[4 * T::SLOT];
  const int tx = threadIdx.x;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = ty * 64 + tx;
  const int i0 = 1 + blockIdx.x * TX;
  const int j0 = 1 + blockIdx.y * T::TY;
  const int k0 = 1 + blockIdx.z * zc;
  const int k1 = min(k0 + zc, g.N[2] - 1);  // planes [k0, k1) are computed
  const int N0 = g.N[0], N1 = g.N[1];
  const long long sz = g.sx[2];

  // Per-thread load descriptors of the tile+halo image: LDS offset is tid + q*256 (layout [c][row][col]).
  long long goff[T::NLOAD];
#pragma unroll
  for (int q = 0; q < T::NLOAD; ++q) {
    const int e = tid + q * 256;
    const int c = e / T::PLANE;
    const int rem = e - c * T::PLANE;
    const int r = rem / T::PITCH;
    const int cx = rem - r * T::PITCH;
    const int gi = i0 - 1 + cx, gj = j0 - 1 + r;
    const bool valid = e < T::SLOT && gi < N0 && gj < N1;
    goff[q] = valid ? (long long)c * g.sc + gi + (long long)gj * N0 : -1;
  }
  double regs[T::NLOAD];
  auto load_plane = [&](int k) {
#pragma unroll
    for (int q = 0; q < T::NLOAD; ++q) regs[q] = goff[q] >= 0 ? u[goff[q] + k * sz] : 0.0;
  };
  auto store_plane = [&](int slot) {
    double* dst = lds + slot * T::SLOT + tid;
#pragma unroll
    for (int q = 0; q < T::NLOAD; ++q)
      if (tid + q * 256 < T::SLOT) dst[q * 256] = regs[q];
  };

  // x metrics: fixed for the whole march (per lane)
  const int i = i0 + tx;
  const bool xin = i <= N0 - 2;
  DirMet M[3];
  load_dirmet<0>(g, xin ? i : 1, M[0]);
  bool xdof[3];
#pragma unroll
  for (int al = 0; al < 3; ++al) xdof[al] = xin && i >= g.iu_lo[al][0] && i < g.iu_hi[al][0];

  load_plane(k0 - 1);
  store_plane(0);
  load_plane(k0);
  store_plane(1);
  load_plane(k0 + 1);

  for (int k = k0; k < k1; ++k) {
    const int rel = k - k0 + 1;  // slot of plane k
    store_plane((rel + 1) & 3);
    if (k + 2 <= k1) load_plane(k + 2);
    __syncthreads();
    PlaneView<R> v;
    v.pm = lds + ((rel - 1) & 3) * T::SLOT;
    v.pc = lds + (rel & 3) * T::SLOT;
    v.pp = lds + ((rel + 1) & 3) * T::SLOT;
    load_dirmet<2>(g, k, M[2]);
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int row = ty + 4 * rr;
      const int j = j0 + row;
      if (j > N1 - 2) continue;  // wavefront-uniform
      load_dirmet<1>(g, j, M[1]);
      v.o = (row + 1) * T::PITCH + tx + 1;
      const long long c = i + (long long)j * N0 + k * sz;
      double f[3];
      {
        const bool dof = xdof[0] && j >= g.iu_lo[0][1] && j < g.iu_hi[0][1] && k >= g.iu_lo[0][2] && k < g.iu_hi[0][2];
        const double uc = v.template at<0, 0, 0, 0>();
        double s = pair_term<R, 0, 0>(v, M, visc, uc);
        s += pair_term<R, 0, 1>(v, M, visc, uc);
        s += pair_term<R, 0, 2>(v, M, visc, uc);
        f[0] = dof ? s : 0.0;
      }
      {
        const bool dof = xdof[1] && j >= g.iu_lo[1][1] && j < g.iu_hi[1][1] && k >= g.iu_lo[1][2] && k < g.iu_hi[1][2];
        const double uc = v.template at<1, 0, 0, 0>();
        double s = pair_term<R, 1, 0>(v, M, visc, uc);
        s += pair_term<R, 1, 1>(v, M, visc, uc);
        s += pair_term<R, 1, 2>(v, M, visc, uc);
        f[1] = dof ? s : 0.0;
      }
      {
        const bool dof = xdof[2] && j >= g.iu_lo[2][1] && j < g.iu_hi[2][1] && k >= g.iu_lo[2][2] && k < g.iu_hi[2][2];
        const double uc = v.template at<2, 0, 0, 0>();
        double s = pair_term<R, 2, 0>(v, M, visc, uc);
        s += pair_term<R, 2, 1>(v, M, visc, uc);
        s += pair_term<R, 2, 2>(v, M, visc, uc);
        f[2] = dof ? s : 0.0;
      }
      if (xin) {
        F[c] = f[0];
        F[c + g.sc] = f[1];
        F[c + 2 * g.sc] = f[2];
      }
    }
  }
}

// Zero the ghost shell of a vector field (the part of `fill!(F, 0)` that K1's interior sweep does not cover).
__global__ __launch_bounds__(256) void k_zero_shell(GridDev g, double* __restrict__ F) {
  const int N0 = g.N[0], N1 = g.N[1], N2 = g.N[2];
  const long long nface_z = (long long)N0 * N1, nface_y = (long long)N0 * N2, nface_x = (long long)N1 * N2;
  const long long total = 2 * (nface_z + nface_y + nface_x);
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    long long r = t;
    long long c;
    if (r < 2 * nface_z) {
      const int side = r >= nface_z;
      r -= side * nface_z;
      c = r + (side ? (long long)(N2 - 1) * g.sx[2] : 0);
    } else if ((r -= 2 * nface_z) < 2 * nface_y) {
      const int side = r >= nface_y;
      r -= side * nface_y;
      const int ii = (int)(r % N0), kk = (int)(r / N0);
      c = ii + (side ? (long long)(N1 - 1) * g.sx[1] : 0) + kk * g.sx[2];
    } else {
      r -= 2 * nface_y;
      const int side = r >= nface_x;
      r -= side * nface_x;
      const int jj = (int)(r % N1), kk = (int)(r / N1);
      c = (side ? N0 - 1 : 0) + jj * g.sx[1] + kk * g.sx[2];
    }
    F[c] = 0.0;
    F[c + g.sc] = 0.0;
    F[c + 2 * g.sc] = 0.0;
  }
}

}  // namespace

static int g_tile_rows = 2;   // rows per thread (R)
static int g_zchunk = 32;     // planes per workgroup

// Tuning knobs for experiments (not part of the public ABI).
extern "C" void ins_tune_fast3d(int rows, int zchunk) {
  if (rows == 1 || rows == 2) g_tile_rows = rows;
  if (zchunk >= 1) g_zchunk = zchunk;
}

bool ins_fast3d_supported(const ins_grid* G) {
  const GridDev& g = G->g;
  if (g.D != 3) return false;
  if (ins_opt(OPT_INS_DISABLE_FAST3D)) return false;
  return g.N[0] >= 4 && g.N[1] >= 4 && g.N[2] >= 4;
}

int ins_k_momentum_flux3d(const ins_grid* G, double visc, const double* u, double* F, bool zero_shell, hipStream_t s);

int ins_k_momentum_fast3d_opts(const ins_grid* G, double visc, const double* u, double* F, bool zero_shell, hipStream_t s) {
  const bool use_lds = ins_opt(OPT_INS_K1_LDS) != 0;  // A/B switch for experiments
  if (!use_lds) return ins_k_momentum_flux3d(G, visc, u, F, zero_shell, s);
  const GridDev& g = G->g;
  const int nx = g.N[0] - 2, ny = g.N[1] - 2, nz = g.N[2] - 2;
  const int zc = g_zchunk;
  dim3 block(64, 4, 1);
  if (g_tile_rows == 1) {
    dim3 grid(cdiv(nx, TX), cdiv(ny, Tile<1>::TY), cdiv(nz, zc));
    hipLaunchKernelGGL(k_momentum_tiled<1>, grid, block, 0, s, g, visc, u, F, zc);
  } else {
    dim3 grid(cdiv(nx, TX), cdiv(ny, Tile<2>::TY), cdiv(nz, zc));
    hipLaunchKernelGGL(k_momentum_tiled<2>, grid, block, 0, s, g, visc, u, F, zc);
  }
  INS_LAUNCH_CHECK();
  if (zero_shell) {
    const long long total = 2LL * ((long long)g.N[0] * g.N[1] + (long long)g.N[0] * g.N[2] + (long long)g.N[1] * g.N[2]);
    hipLaunchKernelGGL(k_zero_shell, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, s, g, F);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}

int ins_k_momentum_fast3d(const ins_grid* G, double visc, const double* u, double* F, hipStream_t s) {
  return ins_k_momentum_fast3d_opts(G, visc, u, F, true, s);
}
