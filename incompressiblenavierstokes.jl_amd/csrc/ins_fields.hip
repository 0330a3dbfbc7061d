// Step-adjacent field operators: vorticity / interpolation / Q / D / λ2 diagnostics, the
// temperature equation (convection-diffusion, dissipation, gravity, ghost fill) and the Smagorinsky stress and its divergence.
// Same conventions as ins_operators.hip: one work-item per volume, x along the 64-lane wavefront (unit-stride row segments),
// any boundary conditions, 2-D and 3-D, stretched grids.  All are single HBM passes; reciprocal metric tables replace the
// reference's divisions (<= 1 ulp per term, inside the 1e-12 parity tolerance).
#include "ins_internal.h"

namespace {

struct Launch3 {
  dim3 grid, block;
};
inline Launch3 box_launch(int nx, int ny, int nz) {
  Launch3 l;
  l.block = dim3(64, 4, 1);
  l.grid = dim3(cdiv(nx, 64), cdiv(ny, 4), (unsigned)nz);
  return l;
}

// work-item -> volume of the box [lo, hi)
#define INS_BOX_INDEX(lo0, lo1, lo2, hi0, hi1)                   \
  const int i = (lo0) + blockIdx.x * 64 + threadIdx.x;           \
  const int j = (lo1) + blockIdx.y * 4 + threadIdx.y;            \
  const int k = D == 3 ? (lo2) + (int)blockIdx.z : 0;            \
  if (i >= (hi0) || j >= (hi1)) return;                          \
  const int I[3] = {i, j, k};                                    \
  const long long c = i + j * g.sx[1] + k * g.sx[2];             \
  (void)I

// --------------------------------------------------------------------------------------------
// vorticity!                                             operators.jl:985-1020 (ndrange = N .- 1)
// --------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_vorticity(GridDev g, const double* __restrict__ u, double* __restrict__ w) {
  INS_BOX_INDEX(0, 0, 0, g.N[0] - 1, g.N[1] - 1);
  if (D == 2) {
    const double* u0 = u;
    const double* u1 = u + g.sc;
    w[c] = (u1[c + g.sx[0]] - u1[c]) * g.rdxu[0][i] - (u0[c + g.sx[1]] - u0[c]) * g.rdxu[1][j];
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int ap = (a + 1) % 3, am = (a + 2) % 3;
      const double* up = u + ap * g.sc;
      const double* um = u + am * g.sc;
      w[a * g.sc + c] = (um[c + g.sx[ap]] - um[c]) * g.rdxu[ap][I[ap]] - (up[c + g.sx[am]] - up[c]) * g.rdxu[am][I[am]];
    }
  }
}

// interpolate_u_p!                                                     operators.jl:1311-1326
template <int D>
__global__ __launch_bounds__(256) void k_interp_u_p(GridDev g, const double* __restrict__ u, double* __restrict__ up) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double* ua = u + a * g.sc;
    up[a * g.sc + c] = (ua[c - g.sx[a]] + ua[c]) / 2;
  }
}

// interpolate_ω_p!                                                     operators.jl:1336-1370
template <int D>
__global__ __launch_bounds__(256) void k_interp_w_p(GridDev g, const double* __restrict__ w, double* __restrict__ wp) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  if (D == 2) {
    wp[c] = (w[c - g.sx[0] - g.sx[1]] + w[c]) / 2;
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int ap = (a + 1) % 3, am = (a + 2) % 3;
      const double* wa = w + a * g.sc;
      wp[a * g.sc + c] = (wa[c - g.sx[ap] - g.sx[am]] + wa[c]) / 2;
    }
  }
}

// Dfield! (after pressuregradient!)                                    operators.jl:1385-1422
template <int D>
__global__ __launch_bounds__(256) void k_Dfield(GridDev g, const double* __restrict__ G, double* __restrict__ d, double eps) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  double gg = 0.0, lap = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double* Ga = G + a * g.sc;
    const double gm = Ga[c - g.sx[a]], gc = Ga[c];
    gg += (gm + gc) * (gm + gc);
    lap += (gc - gm) * g.rdx[a][I[a]];
  }
  lap = lap > 0 ? fmax(lap, eps) : fmin(lap, -eps);
  d[c] = sqrt(gg) / 2 / lap;
}

// Qfield!                                                              operators.jl:1440-1460
template <int D>
__global__ __launch_bounds__(256) void k_Qfield(GridDev g, const double* __restrict__ u, double* __restrict__ Q) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  double q = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double* ua = u + a * g.sc;
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const double* ub = u + b * g.sc;
      q -= (ua[c] - ua[c - g.sx[b]]) * g.rdx[b][I[b]] * (ub[c] - ub[c - g.sx[a]]) * g.rdx[a][I[a]] / 2;
    }
  }
  Q[c] = q;
}

// ∇(u, I, Δ, Δu): velocity gradient at the pressure point I                operators.jl:1023-1034, 1069-1085
template <int D>
__device__ __forceinline__ void gradu(const GridDev& g, const double* __restrict__ u, long long c, const int (&I)[3], double (&G)[D][D]) {
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double* ua = u + a * g.sc;
    const long long sa = g.sx[a];
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const long long sb = g.sx[b];
      if (a == b) {
        G[a][b] = (ua[c] - ua[c - sb]) * g.rdx[b][I[b]];
      } else {
        const double r1 = g.rdxu[b][I[b]], r0 = g.rdxu[b][I[b] - 1];
        G[a][b] = ((ua[c + sb] - ua[c]) * r1 + (ua[c - sa + sb] - ua[c - sa]) * r1 + (ua[c] - ua[c - sb]) * r0 +
                   (ua[c - sa] - ua[c - sa - sb]) * r0) /
                  4;
      }
    }
  }
}

// dissipation_from_strain!                                                operators.jl:836-854
template <int D>
__global__ __launch_bounds__(256) void k_strain_dissipation(GridDev g, double visc, const double* __restrict__ u, double* __restrict__ e) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  double G[D][D];
  gradu<D>(g, u, c, I, G);
  double ss = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const double s = (G[a][b] + G[b][a]) / 2;
      ss += s * s;
    }
  e[c] = 2 * visc * ss;
}

// eig2field!: middle eigenvalue of S² + R² (3-D)                         operators.jl:1472-1492
// Closed form for a symmetric 3x3 matrix (trigonometric solution of the characteristic cubic).
__global__ __launch_bounds__(256) void k_eig2(GridDev g, const double* __restrict__ u, double* __restrict__ lam) {
  constexpr int D = 3;
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  double G[3][3];
  gradu<3>(g, u, c, I, G);
  double S[3][3], R[3][3], M[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      S[a][b] = (G[a][b] + G[b][a]) / 2;
      R[a][b] = (G[a][b] - G[b][a]) / 2;
    }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      double m = 0.0;
#pragma unroll
      for (int q = 0; q < 3; ++q) m += S[a][q] * S[q][b] + R[a][q] * R[q][b];
      M[a][b] = m;
    }
  const double p1 = M[0][1] * M[0][1] + M[0][2] * M[0][2] + M[1][2] * M[1][2];
  const double q = (M[0][0] + M[1][1] + M[2][2]) / 3;
  const double p2 = (M[0][0] - q) * (M[0][0] - q) + (M[1][1] - q) * (M[1][1] - q) + (M[2][2] - q) * (M[2][2] - q) + 2 * p1;
  double e2;
  if (p2 <= 0.0) {
    e2 = q;  // multiple of the identity
  } else {
    const double p = sqrt(p2 / 6), ip = 1.0 / p;
    const double b00 = (M[0][0] - q) * ip, b11 = (M[1][1] - q) * ip, b22 = (M[2][2] - q) * ip;
    const double b01 = M[0][1] * ip, b02 = M[0][2] * ip, b12 = M[1][2] * ip;
    double r = (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02)) / 2;
    r = fmin(1.0, fmax(-1.0, r));
    const double phi = acos(r) / 3;
    const double e1 = q + 2 * p * cos(phi);                             // largest
    const double e3 = q + 2 * p * cos(phi + 2.0943951023931954923084);  // smallest (phi + 2π/3)
    e2 = 3 * q - e1 - e3;
    // one Newton step on det(M - λ) = 0: the arccosine loses digits when two eigenvalues are close to each other
    const double a00 = M[0][0] - e2, a11 = M[1][1] - e2, a22 = M[2][2] - e2;
    const double m0 = a11 * a22 - M[1][2] * M[1][2], m1 = a00 * a22 - M[0][2] * M[0][2], m2 = a00 * a11 - M[0][1] * M[0][1];
    const double det = a00 * m0 - M[0][1] * (M[0][1] * a22 - M[1][2] * M[0][2]) + M[0][2] * (M[0][1] * M[1][2] - a11 * M[0][2]);
    const double dd = -(m0 + m1 + m2);  // d det / dλ
    if (fabs(dd) > 1e-8 * p * p) e2 -= det / dd;
  }
  lam[c] = e2;
}

// --------------------------------------------------------------------------------------------
// temperature equation
// --------------------------------------------------------------------------------------------
// avg(ϕ, Δ, I, α)                                                               operators.jl:59-62
__device__ __forceinline__ double avg_at(const GridDev& g, const double* __restrict__ phi, long long c, int ia, int a) {
  const double d0 = g.dx[a][ia], d1 = g.dx[a][ia + 1];
  return (d1 * phi[c] + d0 * phi[c + g.sx[a]]) / (d0 + d1);
}

// convection_diffusion_temp!  (c += ...)                                       operators.jl:712-737
template <int D>
__global__ __launch_bounds__(256) void k_convdiff_temp(GridDev g, double a4, const double* __restrict__ u, const double* __restrict__ temp,
                                                       double* __restrict__ out) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  const double tc = temp[c];
  double acc = 0.0;
#pragma unroll
  for (int b = 0; b < D; ++b) {
    const long long sb = g.sx[b];
    const double* ub = u + b * g.sc;
    const double dT1 = (tc - temp[c - sb]) * g.rdxu[b][I[b] - 1];
    const double dT2 = (temp[c + sb] - tc) * g.rdxu[b][I[b]];
    const double uT1 = ub[c - sb] * avg_at(g, temp, c - sb, I[b] - 1, b);
    const double uT2 = ub[c] * avg_at(g, temp, c, I[b], b);
    acc += (-(uT2 - uT1) + a4 * (dT2 - dT1)) * g.rdx[b][I[b]];
  }
  out[c] += acc;
}

// dissipation!: interpolation of u · diffusion(u) to the pressure points  (diss += ...)   operators.jl:800-810
template <int D>
__global__ __launch_bounds__(256) void k_dissipation_interp(GridDev g, double coef, const double* __restrict__ u, const double* __restrict__ diff,
                                                            double* __restrict__ diss) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  double d = 0.0;
#pragma unroll
  for (int b = 0; b < D; ++b) {
    const double* ub = u + b * g.sc;
    const double* db = diff + b * g.sc;
    d += coef * (ub[c - g.sx[b]] * db[c - g.sx[b]] + ub[c] * db[c]) / 2;
  }
  diss[c] += d;
}

// gravity!  (F[:, gdir] += α2 avg(temp))   over the whole Iu[gdir]                     operators.jl:914-931
template <int D>
__global__ __launch_bounds__(256) void k_gravity(GridDev g, int gdir, double a2, const double* __restrict__ temp, double* __restrict__ F) {
  INS_BOX_INDEX(g.iu_lo[gdir][0], g.iu_lo[gdir][1], g.iu_lo[gdir][2], g.iu_hi[gdir][0], g.iu_hi[gdir][1]);
  F[gdir * g.sc + c] += a2 * avg_at(g, temp, c, gdir == 0 ? i : (gdir == 1 ? j : k), gdir);
}

// apply_bc_temp!                               boundary_conditions.jl:236-246, 338-339, 391-405, 466-467, 512-513
// One work-item per point of the full padded plane (boundary(), :97-103).  Dirichlet value: constant, or a plane buffer.
struct TempBC {
  int bc[2];
  double val[2];
  const double* plane[2];
};
template <int D>
__global__ __launch_bounds__(256) void k_bc_temp(GridDev g, double* __restrict__ temp, int be, TempBC t) {
  const int o0 = be == 0 ? 1 : 0;
  const int o1 = be == 2 ? 1 : 2;
  const int q0 = blockIdx.x * 256 + threadIdx.x;
  const int q1 = D == 3 ? (int)blockIdx.y : 0;
  if (q0 >= g.N[o0]) return;
  const long long base = q0 * g.sx[o0] + (D == 3 ? q1 * g.sx[o1] : 0);
  const long long sb = g.sx[be];
  const int ia = g.ip_lo[be] - 1, ib = g.ip_hi[be];
  if (t.bc[0] == INS_BC_PERIODIC) {
    temp[base + ia * sb] = temp[base + (ib - 1) * sb];
    temp[base + ib * sb] = temp[base + (ia + 1) * sb];
    return;
  }
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const int bc = t.bc[side];
    const int i = side ? ib : ia;
    const int jn = side ? i - 1 : i + 1;
    if (bc == INS_BC_DIRICHLET)
      temp[base + i * sb] = t.plane[side] ? t.plane[side][q0 + (long long)q1 * g.N[o0]] : t.val[side];
    else if (bc == INS_BC_SYMMETRIC || bc == INS_BC_PRESSURE)
      temp[base + i * sb] = temp[base + jn * sb];
  }
}

// --------------------------------------------------------------------------------------------
// Smagorinsky closure.  The stress tensor is symmetric: stored as D(D+1)/2 scalar fields
// [xx, yy, (zz), xy, (xz, yz)] instead of the reference's array of D x D SMatrix.
// --------------------------------------------------------------------------------------------
template <int D>
__host__ __device__ constexpr int sym_index(int a, int b) {
  if (a == b) return a;
  if (D == 2) return 2;
  const int lo = a < b ? a : b, hi = a < b ? b : a;
  return lo == 0 ? (hi == 1 ? 3 : 4) : 5;
}

// smagtensor!                                                                 operators.jl:1135-1150
template <int D>
__global__ __launch_bounds__(256) void k_smagtensor(GridDev g, double theta, const double* __restrict__ u, double* __restrict__ sig) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  double G[D][D];
  gradu<D>(g, u, c, I, G);
  double ss = 0.0, d2 = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    d2 += g.dx[a][I[a]] * g.dx[a][I[a]];
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const double s = (G[a][b] + G[b][a]) / 2;
      ss += s * s;
    }
  }
  const double eddy = theta * theta * d2 * sqrt(2 * ss);  // gridsize² = Σ Δα²
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = a; b < D; ++b) sig[sym_index<D>(a, b) * g.sc + c] = 2 * eddy * ((G[a][b] + G[b][a]) / 2);
}

// divoftensor!                                                                operators.jl:1203-1236
template <int D>
__global__ __launch_bounds__(256) void k_divoftensor(GridDev g, const double* __restrict__ sig, double* __restrict__ s) {
  INS_BOX_INDEX(0, 0, 0, g.N[0], g.N[1]);
#pragma unroll
  for (int a = 0; a < D; ++a) {
    bool dof = true;
#pragma unroll
    for (int b = 0; b < D; ++b) dof = dof && I[b] >= g.iu_lo[a][b] && I[b] < g.iu_hi[a][b];
    if (!dof) continue;
    const long long sa = g.sx[a];
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const long long sb = g.sx[b];
      const double* t = sig + sym_index<D>(a, b) * g.sc;
      double s2, s1;
      if (a == b) {
        s2 = t[c + sb];
        s1 = t[c];
      } else {
        s2 = (t[c] + t[c + sb] + t[c + sa + sb] + t[c + sa]) / 4;
        s1 = (t[c - sb] + t[c] + t[c + sa - sb] + t[c + sa]) / 4;
      }
      acc += (s2 - s1) * (a == b ? g.rdxu[b] : g.rdx[b])[I[b]];
    }
    s[a * g.sc + c] = acc;
  }
}

#define INS_LAUNCH_D(KERNEL, L, S, ...)                                                \
  do {                                                                                  \
    if (g.D == 2)                                                                       \
      hipLaunchKernelGGL((KERNEL<2>), (L).grid, (L).block, 0, S, __VA_ARGS__);          \
    else                                                                                \
      hipLaunchKernelGGL((KERNEL<3>), (L).grid, (L).block, 0, S, __VA_ARGS__);          \
    INS_LAUNCH_CHECK();                                                                 \
  } while (0)

inline Launch3 ip_launch(const GridDev& g) {
  return box_launch(g.ip_hi[0] - g.ip_lo[0], g.ip_hi[1] - g.ip_lo[1], g.D == 3 ? g.ip_hi[2] - g.ip_lo[2] : 1);
}

}  // namespace

extern "C" int ins_vorticity_f64(const ins_grid_t* G, const double* u, double* w, void* stream) {
  INS_REQUIRE(G && u && w, "null argument");
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.N[0] - 1, g.N[1] - 1, g.D == 3 ? g.N[2] - 1 : 1);
  INS_LAUNCH_D(k_vorticity, l, as_stream(stream), g, u, w);
  return INS_OK;
}

extern "C" int ins_interpolate_u_p_f64(const ins_grid_t* G, const double* u, double* up, void* stream) {
  INS_REQUIRE(G && u && up, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_interp_u_p, l, as_stream(stream), g, u, up);
  return INS_OK;
}

extern "C" int ins_interpolate_w_p_f64(const ins_grid_t* G, const double* w, double* wp, void* stream) {
  INS_REQUIRE(G && w && wp, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_interp_w_p, l, as_stream(stream), g, w, wp);
  return INS_OK;
}

extern "C" int ins_dfield_f64(const ins_grid_t* G, const double* p, double* Gp, double* d, double eps, void* stream) {
  INS_REQUIRE(G && p && Gp && d, "null argument");
  const GridDev& g = G->g;
  int rc = ins_pressuregradient_f64(G, p, Gp, stream);
  if (rc != INS_OK) return rc;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_Dfield, l, as_stream(stream), g, (const double*)Gp, d, eps);
  return INS_OK;
}

extern "C" int ins_qfield_f64(const ins_grid_t* G, const double* u, double* Q, void* stream) {
  INS_REQUIRE(G && u && Q, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_Qfield, l, as_stream(stream), g, u, Q);
  return INS_OK;
}

extern "C" int ins_dissipation_from_strain_f64(const ins_grid_t* G, double visc, const double* u, double* e, void* stream) {
  INS_REQUIRE(G && u && e, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_strain_dissipation, l, as_stream(stream), g, visc, u, e);
  return INS_OK;
}

extern "C" int ins_eig2field_f64(const ins_grid_t* G, const double* u, double* lam, void* stream) {
  INS_REQUIRE(G && u && lam, "null argument");
  const GridDev& g = G->g;
  INS_REQUIRE(g.D == 3, "eig2 only implemented in 3D");  // operators.jl:1477
  Launch3 l = ip_launch(g);
  hipLaunchKernelGGL(k_eig2, l.grid, l.block, 0, as_stream(stream), g, u, lam);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

extern "C" int ins_apply_bc_temp_f64(const ins_grid_t* G, const int32_t* bc, const double* val, const double* const* planes, double* temp,
                                     void* stream) {
  INS_REQUIRE(G && bc && val && temp, "null argument");
  const GridDev& g = G->g;
  for (int be = 0; be < g.D; ++be) {
    TempBC t;
    for (int side = 0; side < 2; ++side) {
      t.bc[side] = bc[2 * be + side];
      t.val[side] = val[2 * be + side];
      t.plane[side] = planes ? planes[2 * be + side] : nullptr;
      INS_REQUIRE(t.bc[side] == INS_BC_PERIODIC || t.bc[side] == INS_BC_DIRICHLET || t.bc[side] == INS_BC_SYMMETRIC || t.bc[side] == INS_BC_PRESSURE,
                  "temperature boundary condition");
    }
    INS_REQUIRE((t.bc[0] == INS_BC_PERIODIC) == (t.bc[1] == INS_BC_PERIODIC), "periodic on both sides");
    const int o0 = be == 0 ? 1 : 0, o1 = be == 2 ? 1 : 2;
    dim3 grid(cdiv(g.N[o0], 256), g.D == 3 ? g.N[o1] : 1, 1);
    if (g.D == 2)
      hipLaunchKernelGGL(k_bc_temp<2>, grid, dim3(256), 0, as_stream(stream), g, temp, be, t);
    else
      hipLaunchKernelGGL(k_bc_temp<3>, grid, dim3(256), 0, as_stream(stream), g, temp, be, t);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}

extern "C" int ins_convection_diffusion_temp_f64(const ins_grid_t* G, double a4, const double* u, const double* temp, double* c, void* stream) {
  INS_REQUIRE(G && u && temp && c, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_convdiff_temp, l, as_stream(stream), g, a4, u, temp, c);
  return INS_OK;
}

extern "C" int ins_dissipation_f64(const ins_grid_t* G, double visc, double coef, const double* u, double* diff, double* diss, void* stream) {
  INS_REQUIRE(G && u && diff && diss, "null argument");
  const GridDev& g = G->g;
  // fill!(diff, 0); diffusion!(diff, u, setup)                                  operators.jl:797-798
  INS_HIP_TRY(hipMemsetAsync(diff, 0, (size_t)G->ncell * g.D * sizeof(double), as_stream(stream)));
  int rc = ins_diffusion_f64(G, visc, u, diff, stream);
  if (rc != INS_OK) return rc;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_dissipation_interp, l, as_stream(stream), g, coef, u, (const double*)diff, diss);
  return INS_OK;
}

extern "C" int ins_gravity_f64(const ins_grid_t* G, int gdir, double a2, const double* temp, double* F, void* stream) {
  INS_REQUIRE(G && temp && F, "null argument");
  const GridDev& g = G->g;
  INS_REQUIRE(gdir >= 0 && gdir < g.D, "gravity direction");
  Launch3 l = box_launch(g.iu_hi[gdir][0] - g.iu_lo[gdir][0], g.iu_hi[gdir][1] - g.iu_lo[gdir][1],
                         g.D == 3 ? g.iu_hi[gdir][2] - g.iu_lo[gdir][2] : 1);
  INS_LAUNCH_D(k_gravity, l, as_stream(stream), g, gdir, a2, temp, F);
  return INS_OK;
}

extern "C" int ins_smagtensor_f64(const ins_grid_t* G, double theta, const double* u, double* sig, void* stream) {
  INS_REQUIRE(G && u && sig, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_smagtensor, l, as_stream(stream), g, theta, u, sig);
  return INS_OK;
}

extern "C" int ins_divoftensor_f64(const ins_grid_t* G, const double* sig, double* s, void* stream) {
  INS_REQUIRE(G && sig && s, "null argument");
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.N[0], g.N[1], g.D == 3 ? g.N[2] : 1);
  INS_LAUNCH_D(k_divoftensor, l, as_stream(stream), g, sig, s);
  return INS_OK;
}
