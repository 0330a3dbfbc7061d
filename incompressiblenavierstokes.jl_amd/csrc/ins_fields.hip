// Step-adjacent field operators: vorticity / interpolation / Q / D / λ2 diagnostics, the
// temperature equation (convection-diffusion, dissipation, gravity, ghost fill) and the Smagorinsky stress and its divergence.
// Same conventions as ins_operators.hip: one work-item per volume, x along the 64-lane wavefront (unit-stride row segments),
// any boundary conditions, 2-D and 3-D, stretched grids.  All are single HBM passes; reciprocal metric tables replace the
// reference's divisions (<= 1 ulp per term, inside the 1e-12 parity tolerance).
#include "ins_internal.h"

namespace {

// Box of nx*ny*nz work-items as a 1-D grid of 64x4 workgroups with an XCD-aware order: workgroup ids are dealt round-robin to the
// 8 XCDs, so id & 7 selects one of 8 y-ranges and, inside it, tiles run x fastest, then y, then z.  Every XCD then walks ITS slab of
// rows plane after plane, and the k-1 / k+1 planes a stencil re-reads are still in that XCD's own 4 MB L2 (3 planes x 1/8 of the
// rows x 3 components = 0.6 MB at 256^3) instead of coming from HBM three times.
struct Launch3 {
  dim3 grid, block;
  int ntx, nty, nty_l;
};
inline Launch3 box_launch(int nx, int ny, int nz) {
  Launch3 l;
  l.block = dim3(64, 4, 1);
  l.ntx = (int)cdiv(nx, 64);
  l.nty = (int)cdiv(ny, 4);
  l.nty_l = (l.nty + 7) / 8;
  l.grid = dim3(8u * l.ntx * l.nty_l * (unsigned)nz, 1, 1);
  return l;
}

// work-item -> volume of the box [lo, hi)
#define INS_BOX_INDEX(lo0, lo1, lo2, hi0, hi1)                              \
  int seq_ = (int)(blockIdx.x >> 3);                                         \
  const int tx_ = seq_ % L.ntx;                                              \
  seq_ /= L.ntx;                                                             \
  const int ty_ = (int)(blockIdx.x & 7) * L.nty_l + seq_ % L.nty_l;          \
  if (ty_ >= L.nty) return;                                                  \
  const int i = (lo0) + tx_ * 64 + threadIdx.x;                              \
  const int j = (lo1) + ty_ * 4 + threadIdx.y;                               \
  const int k = D == 3 ? (lo2) + seq_ / L.nty_l : 0;                         \
  if (i >= (hi0) || j >= (hi1)) return;                                      \
  const int I[3] = {i, j, k};                                                \
  const long long c = i + j * g.sx[1] + k * g.sx[2];                         \
  (void)I

struct BoxMap {
  int ntx, nty, nty_l;
};

// --------------------------------------------------------------------------------------------
// vorticity!                                             operators.jl:985-1020 (ndrange = N .- 1)
// --------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_vorticity(GridDev g, BoxMap L, const double* __restrict__ u, double* __restrict__ w) {
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
__global__ __launch_bounds__(256) void k_interp_u_p(GridDev g, BoxMap L, const double* __restrict__ u, double* __restrict__ up) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double* ua = u + a * g.sc;
    up[a * g.sc + c] = (ua[c - g.sx[a]] + ua[c]) / 2;
  }
}

// interpolate_ω_p!                                                     operators.jl:1336-1370
template <int D>
__global__ __launch_bounds__(256) void k_interp_w_p(GridDev g, BoxMap L, const double* __restrict__ w, double* __restrict__ wp) {
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
__global__ __launch_bounds__(256) void k_Dfield(GridDev g, BoxMap L, const double* __restrict__ G, double* __restrict__ d, double eps) {
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
__global__ __launch_bounds__(256) void k_Qfield(GridDev g, BoxMap L, const double* __restrict__ u, double* __restrict__ Q) {
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

// ---- per-cell results from the velocity gradient G ------------------------------------------------------------------
template <int D>
__device__ __forceinline__ double strain_norm2(const double (&G)[D][D]) {  // Σ S_ab S_ab
  double ss = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const double s = (G[a][b] + G[b][a]) / 2;
      ss += s * s;
    }
  return ss;
}

// middle eigenvalue of S² + R² (operators.jl:1484-1488): the characteristic cubic of a symmetric 3x3 matrix in its depressed form, the middle
// root by Newton on a well-conditioned substitute equation, plus one Newton step on the determinant
__device__ __forceinline__ double eig2_of(const double (&G)[3][3]) {
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
  if (p2 <= 0.0) return q;  // multiple of the identity
  const double p = sqrt(p2 / 6), ip = 1.0 / p;
  const double b00 = (M[0][0] - q) * ip, b11 = (M[1][1] - q) * ip, b22 = (M[2][2] - q) * ip;
  const double b01 = M[0][1] * ip, b02 = M[0][2] * ip, b12 = M[1][2] * ip;
  double r = (b00 * (b11 * b22 - b12 * b12) - b01 * (b01 * b22 - b12 * b02) + b02 * (b01 * b12 - b11 * b02)) / 2;
  r = fmin(1.0, fmax(-1.0, r));
  // The eigenvalues are q + 2p x over the roots of 4x³ - 3x = r; the middle one is x = ∓(1/2 - δ) for r ≷ 0 with δ in [0, 1/2] the root of
  // δ²(6 - 4δ) = 1 - |r|.  Newton on that (quadratic also at the double root δ -> 0, where it is the iteration of a square root) from
  // δ0 = sqrt(c / (6 - 4 sqrt(c/6))) reaches double precision in three steps (reciprocals from v_rcp_f64: their error scales the shrinking
  // correction) — no arccosine and no cosines: this kernel was bound by them (0.36 ms at 256³, tools/fields_bench.py).
  const double c = 1.0 - fabs(r);
  double dl = sqrt(c / (6.0 - 4.0 * sqrt(c * (1.0 / 6.0))));
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const double h = (6.0 - 4.0 * dl) * dl * dl - c;
    dl -= h * __builtin_amdgcn_rcp(12.0 * dl * (1.0 - dl));
  }
  dl = c > 0.0 ? dl : 0.0;
  const double xm = r >= 0.0 ? dl - 0.5 : 0.5 - dl;
  double e2 = q + 2 * p * xm;
  // one Newton step on det(M - λ) = 0: 1 - |r| loses digits when two eigenvalues are close to each other
  const double a00 = M[0][0] - e2, a11 = M[1][1] - e2, a22 = M[2][2] - e2;
  const double m0 = a11 * a22 - M[1][2] * M[1][2], m1 = a00 * a22 - M[0][2] * M[0][2], m2 = a00 * a11 - M[0][1] * M[0][1];
  const double det = a00 * m0 - M[0][1] * (M[0][1] * a22 - M[1][2] * M[0][2]) + M[0][2] * (M[0][1] * M[1][2] - a11 * M[0][2]);
  const double dd = -(m0 + m1 + m2);  // d det / dλ
  if (fabs(dd) > 1e-8 * p * p) e2 -= det / dd;
  return e2;
}

template <int D>
__host__ __device__ constexpr int sym_index(int a, int b) {  // [xx, yy, (zz), xy, (xz, yz)]
  if (a == b) return a;
  if (D == 2) return 2;
  const int lo = a < b ? a : b, hi = a < b ? b : a;
  return lo == 0 ? (hi == 1 ? 3 : 4) : 5;
}

// OP 0: dissipation_from_strain! (operators.jl:836-854, par = 1/Re)   OP 1: eig2field! (:1472-1492)
// OP 2: smagtensor! (:1135-1150, par = θ; D(D+1)/2 outputs)
template <int D, int OP>
__device__ __forceinline__ void gradient_result(const GridDev& g, const double (&G)[D][D], const int (&I)[3], long long c, double par,
                                                double* __restrict__ out) {
  if (OP == 0) {
    out[c] = 2 * par * strain_norm2<D>(G);
  } else if (OP == 1) {
    if constexpr (D == 3) out[c] = eig2_of(G);
  } else {
    double d2 = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) d2 += g.dx[a][I[a]] * g.dx[a][I[a]];  // gridsize² = Σ Δα²
    const double eddy = par * par * d2 * sqrt(2 * strain_norm2<D>(G));
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = a; b < D; ++b) out[sym_index<D>(a, b) * g.sc + c] = 2 * eddy * ((G[a][b] + G[b][a]) / 2);
  }
}

// one work-item per pressure point, every neighbour from global memory (2-D grids)
template <int D, int OP>
__global__ __launch_bounds__(256) void k_gradient_op(GridDev g, BoxMap L, double par, const double* __restrict__ u, double* __restrict__ out) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  double G[D][D];
  gradu<D>(g, u, c, I, G);
  gradient_result<D, OP>(g, G, I, c, par, out);
}

// tensorbasis!: symmetry tensor basis B[1..nb] and invariants V[1..nv] of Silvis et al. (tensorbasis.jl:16-72), nb, nv = 3, 2 in 2-D and
// 11, 5 in 3-D.  Stored as scalar fields: B[(ib * D*D + a + D*b) * ncell + c] (the SMatrix' column-major element order), V[iv * ncell + c].
template <int D>
struct Mat {
  double m[D][D];
};
template <int D>
__device__ __forceinline__ Mat<D> mmul(const Mat<D>& x, const Mat<D>& y) {
  Mat<D> r;
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      double v = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) v += x.m[a][q] * y.m[q][b];
      r.m[a][b] = v;
    }
  return r;
}
template <int D>
__device__ __forceinline__ Mat<D> madd(const Mat<D>& x, const Mat<D>& y, double sy) {
  Mat<D> r;
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) r.m[a][b] = x.m[a][b] + sy * y.m[a][b];
  return r;
}
template <int D>
__device__ __forceinline__ double mtrace(const Mat<D>& x) {
  double t = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) t += x.m[a][a];
  return t;
}
template <int D>
__device__ __forceinline__ void put(double* __restrict__ B, const GridDev& g, long long c, int ib, const Mat<D>& x) {
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) B[(long long)(ib * D * D + a + D * b) * g.sc + c] = x.m[a][b];
}

template <int D>
__global__ __launch_bounds__(256) void k_tensorbasis(GridDev g, BoxMap L, const double* __restrict__ u, double* __restrict__ B, double* __restrict__ V) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  double G[D][D];
  gradu<D>(g, u, c, I, G);
  Mat<D> S, R, Id;
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      S.m[a][b] = (G[a][b] + G[b][a]) / 2;
      R.m[a][b] = (G[a][b] - G[b][a]) / 2;
      Id.m[a][b] = a == b ? 1.0 : 0.0;
    }
  const Mat<D> SR = mmul<D>(S, R), RS = mmul<D>(R, S);
  put<D>(B, g, c, 0, Id);
  put<D>(B, g, c, 1, S);
  put<D>(B, g, c, 2, madd<D>(SR, RS, -1.0));
  if (D == 2) {
    double ss = 0.0, rr = 0.0;  // dot(S, S), dot(R, R)                                   tensorbasis.jl:49-50
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        ss += S.m[a][b] * S.m[a][b];
        rr += R.m[a][b] * R.m[a][b];
      }
    V[c] = ss;
    V[g.sc + c] = rr;
  } else {
    const Mat<D> SS = mmul<D>(S, S), RR = mmul<D>(R, R);
    put<D>(B, g, c, 3, SS);
    put<D>(B, g, c, 4, RR);
    put<D>(B, g, c, 5, madd<D>(mmul<D>(SS, R), mmul<D>(R, SS), -1.0));           // S S R - R S S
    put<D>(B, g, c, 6, madd<D>(mmul<D>(S, RR), mmul<D>(RR, S), 1.0));            // S R R + R R S
    put<D>(B, g, c, 7, madd<D>(mmul<D>(RS, RR), mmul<D>(RR, SR), -1.0));         // R S R R - R R S R
    put<D>(B, g, c, 8, madd<D>(mmul<D>(SR, SS), mmul<D>(SS, RS), -1.0));         // S R S S - S S R S
    const Mat<D> SSRR = mmul<D>(SS, RR);
    put<D>(B, g, c, 9, madd<D>(SSRR, mmul<D>(RR, SS), 1.0));                     // S S R R + R R S S
    put<D>(B, g, c, 10, madd<D>(mmul<D>(R, SSRR), mmul<D>(mmul<D>(RR, SS), R), -1.0));  // R S S R R - R R S S R
    V[c] = mtrace<D>(SS);
    V[g.sc + c] = mtrace<D>(RR);
    V[2 * g.sc + c] = mtrace<D>(mmul<D>(SS, S));
    V[3 * g.sc + c] = mtrace<D>(mmul<D>(S, RR));
    V[4 * g.sc + c] = mtrace<D>(SSRR);
  }
}

// 3-D: a 64x4 tile marches through a z-chunk with a ring of three (tile + 1 halo) planes of u in LDS.  The ~40 neighbour values a
// cell needs then come from LDS; with one global load per neighbour the kernel was bound by L2 bandwidth (320 B/cell out of L2 for
// 32 algorithmic bytes: 0.32 ms at 256^3); the ring brings 3 x 6 x 66 / 256 = 4.6 loads per cell and plane.
constexpr int GT_X = 64, GT_Y = 4;
template <int OP>
__global__ __launch_bounds__(256) void k_gradient_march(GridDev g, BoxMap L, int zc, double par, const double* __restrict__ u, double* __restrict__ out) {
  __shared__ double T[3][3][GT_Y + 2][GT_X + 2];  // [ring slot][component][row][column]
  int seq = (int)(blockIdx.x >> 3);
  const int tx = seq % L.ntx;
  seq /= L.ntx;
  const int ty = (int)(blockIdx.x & 7) * L.nty_l + seq % L.nty_l;
  if (ty >= L.nty) return;  // whole workgroup
  const int chunk = seq / L.nty_l;
  const int i0 = g.ip_lo[0] + tx * GT_X, j0 = g.ip_lo[1] + ty * GT_Y;
  const int k0 = g.ip_lo[2] + chunk * zc, k1 = min(k0 + zc, g.ip_hi[2]);
  const int lx = threadIdx.x, ly = threadIdx.y, tid = ly * 64 + lx;
  const int i = i0 + lx, j = j0 + ly;
  const bool active = i < g.ip_hi[0] && j < g.ip_hi[1];
  auto load_plane = [&](int kk) {
    const int slot = kk % 3;
    const long long zoff = (long long)kk * g.sx[2];
    for (int q = tid; q < 3 * (GT_Y + 2) * (GT_X + 2); q += 256) {
      const int a = q / ((GT_Y + 2) * (GT_X + 2));
      const int r = (q / (GT_X + 2)) % (GT_Y + 2), cc = q % (GT_X + 2);
      const int gi = min(i0 - 1 + cc, g.N[0] - 1), gj = min(j0 - 1 + r, g.N[1] - 1);  // tiles may overhang the box: stay in memory
      T[slot][a][r][cc] = u[a * g.sc + gi + gj * g.sx[1] + zoff];
    }
  };
  load_plane(k0 - 1);
  load_plane(k0);
  for (int k = k0; k < k1; ++k) {
    load_plane(k + 1);
    __syncthreads();
    if (active) {
      const int sm = (k + 2) % 3, s0 = k % 3, sp = (k + 1) % 3;  // ring slots of planes k-1, k, k+1
      // value of component a at offset (ox, oy, oz) from this cell
      auto U = [&](int a, int ox, int oy, int oz) { return T[oz < 0 ? sm : (oz > 0 ? sp : s0)][a][ly + 1 + oy][lx + 1 + ox]; };
      auto at = [&](int a, int da, int sa_, int db, int sb_) {  // offsets sa_·e_da + sb_·e_db
        const int ox = (da == 0 ? sa_ : 0) + (db == 0 ? sb_ : 0);
        const int oy = (da == 1 ? sa_ : 0) + (db == 1 ? sb_ : 0);
        const int oz = (da == 2 ? sa_ : 0) + (db == 2 ? sb_ : 0);
        return U(a, ox, oy, oz);
      };
      const int I[3] = {i, j, k};
      double G[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          if (a == b) {
            G[a][b] = (at(a, a, 0, b, 0) - at(a, a, 0, b, -1)) * g.rdx[b][I[b]];
          } else {
            const double r1 = g.rdxu[b][I[b]], r0 = g.rdxu[b][I[b] - 1];
            G[a][b] = ((at(a, a, 0, b, 1) - at(a, a, 0, b, 0)) * r1 + (at(a, a, -1, b, 1) - at(a, a, -1, b, 0)) * r1 +
                       (at(a, a, 0, b, 0) - at(a, a, 0, b, -1)) * r0 + (at(a, a, -1, b, 0) - at(a, a, -1, b, -1)) * r0) /
                      4;
          }
        }
      gradient_result<3, OP>(g, G, I, i + j * g.sx[1] + k * g.sx[2], par, out);
    }
    __syncthreads();  // the next iteration overwrites the slot of plane k-1
  }
}

// 3-D, register rows (the stage kernel's formulation, ins_fast3d_flux.hip): lanes run along x (lanes 0 and 63 are halo columns, 62 outputs
// per wavefront), every work-item keeps R + 2 rows of the three components of THREE planes in registers and marches through a z-chunk;
// x neighbours come from DPP wave shifts, y neighbours from the work-item's own rows, z neighbours from the plane registers.  Per plane
// a work-item issues 3 (R + 2) loads for R cells (R = 4: 4.6 per cell with the halo lanes) against ~40 for the plain kernel — the plain and
// the LDS-ring kernels were bound by the vector-L1 / texture-addresser rate (250 M accesses per launch at 256³), not by HBM.
// One barrier per plane keeps the four y-stacked wavefronts of a workgroup on one plane (shared halo rows are cache hits).
// (no `old` operand: the lane without a source lane reads zero — lanes 0 / 63 are halo columns whose shifted values are never used; update_dpp(old = v, v)
// costs a register copy per 32-bit half on top of the DPP move)
__device__ __forceinline__ double lane_next(double v) {  // lane l receives lane l+1
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_prev(double v) {  // lane l receives lane l-1
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int GR_XO = 62;
// CORRP (all-periodic boxes, extended stage loop without a gradient-subtract pass): `u` is the UNCORRECTED stage velocity (interior volumes
// only) and pI the unpadded pressure of its projection; every volume is read through its periodic image and corrected in registers as its plane
// arrives, u = u* - ∇p (applypressure!, operators.jl:225-233).  Lane 63's x-component would need p of a 65th column: 61 outputs per wavefront.
template <int OP, int R, bool CORRP = false>
__global__ __launch_bounds__(256) void k_gradient_rows(GridDev g, BoxMap L, int zc_, double par, const double* __restrict__ u, double* __restrict__ out,
                                                       const double* __restrict__ pI = nullptr) {
  constexpr int XO = CORRP ? GR_XO - 1 : GR_XO;
  const bool bar = zc_ > 0;  // zc_ < 0: no barrier per plane (INS_FIELDS_NOBAR)
  const int zc = bar ? zc_ : -zc_;
  int seq = (int)(blockIdx.x >> 3);
  const int tx = seq % L.ntx;
  seq /= L.ntx;
  const int ty = (int)(blockIdx.x & 7) * L.nty_l + seq % L.nty_l;
  if (ty >= L.nty) return;  // whole workgroup
  const int chunk = seq / L.nty_l;
  const int lane = threadIdx.x, wy = threadIdx.y;
  const int i = g.ip_lo[0] - 1 + tx * XO + lane;  // lane 0 = left halo column
  const int ic = min(i, g.N[0] - 1);
  const int jb = g.ip_lo[1] + (ty * 4 + wy) * R;  // first output row of this wavefront
  const int k0 = g.ip_lo[2] + chunk * zc, k1 = min(k0 + zc, g.ip_hi[2]);
  const bool wave_on = i - lane < g.ip_hi[0] && jb < g.ip_hi[1];  // wave-uniform
  const bool xout = lane >= 1 && lane <= XO && i < g.ip_hi[0];
  const int n0 = g.ip_hi[0] - g.ip_lo[0], n1 = g.ip_hi[1] - g.ip_lo[1], n2 = g.ip_hi[2] - g.ip_lo[2];
  auto img = [](int idx, int lo, int n) {  // padded index -> padded index of the periodic image inside [lo, lo + n)
    int q = idx - lo;
    q = q < 0 ? q + n : (q >= n ? q - n : q);
    return lo + q;
  };
  long long rowoff[R + 2];
  long long prow[CORRP ? R + 3 : 1];  // element offset of (row, column) inside an unpadded pI plane
#pragma unroll
  for (int rr = 0; rr < R + 2; ++rr)
    rowoff[rr] = CORRP ? (long long)img(min(jb - 1 + rr, g.N[1]), g.ip_lo[1], n1) * g.sx[1] + img(min(i, g.N[0]), g.ip_lo[0], n0)
                       : (long long)min(jb - 1 + rr, g.N[1] - 1) * g.sx[1] + ic;
  if (CORRP) {
#pragma unroll
    for (int rr = 0; rr < R + 3; ++rr)
      prow[rr] = (long long)(img(min(jb - 1 + rr, g.N[1] + 1), g.ip_lo[1], n1) - g.ip_lo[1]) * n0 + (img(min(i, g.N[0]), g.ip_lo[0], n0) - g.ip_lo[0]);
  }
  double P[3][3][R + 2];  // [plane slot][component][row]
  double pnext[CORRP ? R + 3 : 1];  // p of the plane after the one loaded last
  auto load_p = [&](double (&q)[CORRP ? R + 3 : 1], int kk) {
    const double* base = pI + (long long)(img(min(kk, g.N[2] + 1), g.ip_lo[2], n2) - g.ip_lo[2]) * n0 * n1;
#pragma unroll
    for (int rr = 0; rr < R + 3; ++rr) q[rr] = base[prow[rr]];
  };
  const double gx = g.rdxu[0][g.ip_lo[0]], gy = g.rdxu[1][g.ip_lo[1]], gz = g.rdxu[2][g.ip_lo[2]];  // CORRP: uniform box
  auto load_plane = [&](double (&Q)[3][R + 2], int kk) {
    const double* base = u + (long long)(CORRP ? img(min(kk, g.N[2]), g.ip_lo[2], n2) : min(kk, g.N[2] - 1)) * g.sx[2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int rr = 0; rr < R + 2; ++rr) Q[a][rr] = base[a * g.sc + rowoff[rr]];
    if constexpr (CORRP) {
      double pc[R + 3];
#pragma unroll
      for (int rr = 0; rr < R + 3; ++rr) pc[rr] = pnext[rr];
      load_p(pnext, kk + 1);
#pragma unroll
      for (int rr = 0; rr < R + 2; ++rr) {
        Q[0][rr] -= (lane_next(pc[rr]) - pc[rr]) * gx;
        Q[1][rr] -= (pc[rr + 1] - pc[rr]) * gy;
        Q[2][rr] -= (pnext[rr] - pc[rr]) * gz;
      }
    }
  };
  // per-lane x metrics
  const double rdx_x = g.rdx[0][ic], rux1 = g.rdxu[0][ic], rux0 = g.rdxu[0][max(ic - 1, 0)];
  auto body = [&](const double (&M)[3][R + 2], const double (&C)[3][R + 2], const double (&N)[3][R + 2], int k) {
    const double rdx_z = g.rdx[2][k], ruz1 = g.rdxu[2][k], ruz0 = g.rdxu[2][k - 1];
#pragma unroll
    for (int rr = 1; rr <= R; ++rr) {
      const int j = jb - 1 + rr;
      const int jc = min(j, g.N[1] - 2);
      const double rdx_y = g.rdx[1][jc], ruy1 = g.rdxu[1][jc], ruy0 = g.rdxu[1][jc - 1];
      // value of component a at offset (ox, oy, oz) from this cell
      auto U = [&](int a, int ox, int oy, int oz) {
        const double v = oz < 0 ? M[a][rr + oy] : (oz > 0 ? N[a][rr + oy] : C[a][rr + oy]);
        return ox < 0 ? lane_prev(v) : (ox > 0 ? lane_next(v) : v);
      };
      auto at = [&](int a, int da, int sa_, int db, int sb_) {  // offsets sa_·e_da + sb_·e_db
        const int ox = (da == 0 ? sa_ : 0) + (db == 0 ? sb_ : 0);
        const int oy = (da == 1 ? sa_ : 0) + (db == 1 ? sb_ : 0);
        const int oz = (da == 2 ? sa_ : 0) + (db == 2 ? sb_ : 0);
        return U(a, ox, oy, oz);
      };
      const double rd[3] = {rdx_x, rdx_y, rdx_z}, r1s[3] = {rux1, ruy1, ruz1}, r0s[3] = {rux0, ruy0, ruz0};
      double G[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          if (a == b) {
            G[a][b] = (at(a, a, 0, b, 0) - at(a, a, 0, b, -1)) * rd[b];
          } else {
            const double r1 = r1s[b], r0 = r0s[b];
            G[a][b] = ((at(a, a, 0, b, 1) - at(a, a, 0, b, 0)) * r1 + (at(a, a, -1, b, 1) - at(a, a, -1, b, 0)) * r1 +
                       (at(a, a, 0, b, 0) - at(a, a, 0, b, -1)) * r0 + (at(a, a, -1, b, 0) - at(a, a, -1, b, -1)) * r0) /
                      4;
          }
        }
      if (xout && j < g.ip_hi[1]) {
        const int I[3] = {i, j, k};
        gradient_result<3, OP>(g, G, I, i + j * g.sx[1] + k * g.sx[2], par, out);
      }
    }
  };
  if (!wave_on) {  // keeps the workgroup's barrier count
    for (int k = k0; bar && k < k1; ++k) __builtin_amdgcn_s_barrier();
    return;
  }
  // three plane slots (a fourth, loading plane k + 2 during plane k, measured slower: 144 more bytes of registers per lane cost more
  // occupancy than the earlier loads gain — strain dissipation 0.207 -> 0.211 ms, smagtensor 0.449 -> 0.456 at 256^3)
  if constexpr (CORRP) load_p(pnext, k0 - 1);
  load_plane(P[0], k0 - 1);
  load_plane(P[1], k0);
  int k = k0;
  while (true) {  // rotation unrolled so that every register index is static
    if (bar) __builtin_amdgcn_s_barrier();
    load_plane(P[2], k + 1);
    body(P[0], P[1], P[2], k);
    if (++k >= k1) break;
    if (bar) __builtin_amdgcn_s_barrier();
    load_plane(P[0], k + 1);
    body(P[1], P[2], P[0], k);
    if (++k >= k1) break;
    if (bar) __builtin_amdgcn_s_barrier();
    load_plane(P[1], k + 1);
    body(P[2], P[0], P[1], k);
    if (++k >= k1) break;
  }
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
__global__ __launch_bounds__(256) void k_convdiff_temp(GridDev g, BoxMap L, double a4, const double* __restrict__ u, const double* __restrict__ temp,
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
__global__ __launch_bounds__(256) void k_dissipation_interp(GridDev g, BoxMap L, double coef, const double* __restrict__ u, const double* __restrict__ diff,
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

// One stage of the temperature equation in one pass (extended stage loop, ins_rk_ext.hip; step_explicit_runge_kutta.jl:23-27, 39-44):
//   ktemp_i = convection_diffusion_temp(u, temp) + coef · Σ_β (w_β[I-e_β] + w_β[I]) / 2        (w = u · diffusion(u), stored by the stage kernel)
//   temp_out = tempstart + Σ_j c_j ktemp_j + c_self ktemp_i
// instead of fill!, convection_diffusion_temp!, diffusion!, the interpolation kernel and the combination (five passes).  temp_out is another
// array than temp (neighbours of temp are read here).
struct TempStage {
  int n;
  double coef[INS_MAX_STAGES];
  const double* k[INS_MAX_STAGES];
  double c_self;
  const double* tempstart;
  double* ktemp_out;  // nullable: no later stage reads ktemp_i
  double* temp_out;
};
// pI != nullptr (all-periodic boxes): u is the UNCORRECTED stage velocity (interior volumes only) and pI the unpadded pressure of its projection;
// the two face velocities a volume needs are corrected here, u = u* - ∇p through the periodic image (applypressure!, operators.jl:225-233), so
// the stage loop needs no gradient-subtract pass between its stages.
template <int D>
__global__ __launch_bounds__(256) void k_temp_stage(GridDev g, BoxMap L, double a4, double coef, const double* __restrict__ u, const double* __restrict__ temp,
                                                    const double* __restrict__ w, TempStage ts, const double* __restrict__ pI,
                                                    const double* __restrict__ diff) {
  INS_BOX_INDEX(g.ip_lo[0], g.ip_lo[1], g.ip_lo[2], g.ip_hi[0], g.ip_hi[1]);
  const double tc = temp[c];
  double acc = 0.0, d = 0.0;
  const int n[3] = {g.ip_hi[0] - g.ip_lo[0], g.ip_hi[1] - g.ip_lo[1], D == 3 ? g.ip_hi[2] - g.ip_lo[2] : 1};
  const long long qs[3] = {1, n[0], (long long)n[0] * n[1]};
  const long long q = (I[0] - g.ip_lo[0]) + qs[1] * (I[1] - g.ip_lo[1]) + (D == 3 ? qs[2] * (I[2] - g.ip_lo[2]) : 0);
  const double pc = pI ? pI[q] : 0.0;
#pragma unroll
  for (int b = 0; b < D; ++b) {
    const long long sb = g.sx[b];
    const double* ub = u + b * g.sc;
    double u1, u2;
    if (pI) {
      const int w0 = I[b] - g.ip_lo[b];
      const bool first = w0 == 0, lastv = w0 == n[b] - 1;
      const double pp = pI[lastv ? q - (long long)(n[b] - 1) * qs[b] : q + qs[b]];
      const double pm = pI[first ? q + (long long)(n[b] - 1) * qs[b] : q - qs[b]];
      u2 = ub[c] - (pp - pc) * g.rdxu[b][I[b]];
      u1 = ub[first ? c + (long long)(n[b] - 1) * sb : c - sb] - (pc - pm) * g.rdxu[b][first ? g.ip_hi[b] - 1 : I[b] - 1];
    } else {
      u1 = ub[c - sb];
      u2 = ub[c];
    }
    const double dT1 = (tc - temp[c - sb]) * g.rdxu[b][I[b] - 1];
    const double dT2 = (temp[c + sb] - tc) * g.rdxu[b][I[b]];
    const double uT1 = u1 * avg_at(g, temp, c - sb, I[b] - 1, b);
    const double uT2 = u2 * avg_at(g, temp, c, I[b], b);
    acc += (-(uT2 - uT1) + a4 * (dT2 - dT1)) * g.rdx[b][I[b]];
    if (diff) {  // dissipation! as the reference forms it: u · diffusion(u) at the two faces (`diff` is zero off the DOFs: diffusion! never writes there)
      const double* db = diff + b * g.sc;
      d += coef * (ub[c - sb] * db[c - sb] + u2 * db[c]) / 2;
    } else if (w) {
      // diffusion! writes the DOFs of u only and dissipation! starts from fill!(diff, 0) (operators.jl:793-794): the lower-face term of the
      // first volume of a direction reads a ghost volume of `diff`, i.e. zero — also in a periodic direction (not its periodic image)
      const double* wb = w + b * g.sc;
      const double wm = I[b] - 1 >= g.iu_lo[b][b] ? wb[c - sb] : 0.0;
      d += coef * (wm + wb[c]) / 2;
    }
  }
  acc += d;
  double t = ts.tempstart[c];
  for (int q = 0; q < ts.n; ++q) t += ts.coef[q] * ts.k[q][c];
  t += ts.c_self * acc;
  if (ts.ktemp_out) ts.ktemp_out[c] = acc;
  ts.temp_out[c] = t;
}

// gravity!  (F[:, gdir] += α2 avg(temp))   over the whole Iu[gdir]                     operators.jl:914-931
template <int D>
__global__ __launch_bounds__(256) void k_gravity(GridDev g, BoxMap L, int gdir, double a2, const double* __restrict__ temp, double* __restrict__ F) {
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
// [xx, yy, (zz), xy, (xz, yz)] instead of the reference's array of D x D SMatrix (smagtensor!: gradient_result<D, 2>).
// --------------------------------------------------------------------------------------------
// divoftensor!                                                                operators.jl:1203-1236
template <int D>
__global__ __launch_bounds__(256) void k_divoftensor(GridDev g, BoxMap L, const double* __restrict__ sig, double* __restrict__ s) {
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

// divoftensor! in the register-row formulation of k_gradient_rows (3-D): lanes along x (lanes 0 and 63 are halo columns), R + 2 rows per
// work-item, march through a z-chunk.  What a volume needs of each stress component is small — σxx: I, I+ex; σyy: I, I+ey; σzz: I, I+ez;
// σxy: the 3 x 3 (x, y) neighbourhood in its own plane; σxz, σyz: three planes — so a work-item holds 11 R + 9 values and loads 6 R + 5 per
// plane for 3 R results (R = 4: 2.4 loads per result; the plain kernel issues 10 per result and ran at the vector-L1 rate: 0.68 ms at 256³).
// Same expressions in the same order as k_divoftensor.  Needs every DOF inside [1, N - 1) (no PressureBC side: the host checks).
template <int R>
__global__ __launch_bounds__(256) void k_divoftensor_rows(GridDev g, BoxMap L, int zc_, const double* __restrict__ sig, double* __restrict__ s) {
  const bool bar = zc_ > 0;  // zc_ < 0: no barrier per plane (INS_FIELDS_NOBAR)
  const int zc = bar ? zc_ : -zc_;
  int seq = (int)(blockIdx.x >> 3);
  const int tx = seq % L.ntx;
  seq /= L.ntx;
  const int ty = (int)(blockIdx.x & 7) * L.nty_l + seq % L.nty_l;
  if (ty >= L.nty) return;  // whole workgroup
  const int chunk = seq / L.nty_l;
  const int lane = threadIdx.x, wy = threadIdx.y;
  const int N0 = g.N[0], N1 = g.N[1], N2 = g.N[2];
  const int i = tx * GR_XO + lane;  // lane 0 = left halo column (outputs start at volume 1)
  const int ic = min(i, N0 - 1);
  const int jb = 1 + (ty * 4 + wy) * R;  // first output row of this wavefront
  const int k0 = 1 + chunk * zc, k1 = min(k0 + zc, N2 - 1);
  const bool wave_on = i - lane < N0 - 1 && jb < N1 - 1;  // wave-uniform
  if (!wave_on) {  // keeps the workgroup's barrier count
    for (int k = k0; bar && k < k1; ++k) __builtin_amdgcn_s_barrier();
    return;
  }
  const bool xout = lane >= 1 && lane <= GR_XO && i < N0 - 1;
  long long rowoff[R + 2];
#pragma unroll
  for (int rr = 0; rr < R + 2; ++rr) rowoff[rr] = (long long)min(jb - 1 + rr, N1 - 1) * g.sx[1] + ic;
  const double* sxx = sig + sym_index<3>(0, 0) * g.sc;
  const double* syy = sig + sym_index<3>(1, 1) * g.sc;
  const double* szz = sig + sym_index<3>(2, 2) * g.sc;
  const double* sxy = sig + sym_index<3>(0, 1) * g.sc;
  const double* sxz = sig + sym_index<3>(0, 2) * g.sc;
  const double* syz = sig + sym_index<3>(1, 2) * g.sc;
  // rows: index rr = 0 .. R + 1 is row jb - 1 + rr; outputs are rr = 1 .. R
  double XX[R + 2], YY[R + 2], XY[R + 2];          // plane k (XX: rows 1..R used, YY: 1..R+1)
  double ZZ[2][R + 2], XZ[3][R + 2], YZ[3][R + 2];  // planes k (, k-1), k+1 (ZZ, XZ: rows 1..R used)
  auto ld = [&](const double* comp, int kk, int rr) { return comp[(long long)min(kk, N2 - 1) * g.sx[2] + rowoff[rr]]; };
  auto load_upper = [&](int kk) {  // plane kk into the upper slots
#pragma unroll
    for (int rr = 1; rr <= R; ++rr) {
      ZZ[1][rr] = ld(szz, kk, rr);
      XZ[2][rr] = ld(sxz, kk, rr);
    }
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr) YZ[2][rr] = ld(syz, kk, rr);
  };
  auto load_own = [&](int kk) {
#pragma unroll
    for (int rr = 1; rr <= R; ++rr) XX[rr] = ld(sxx, kk, rr);
#pragma unroll
    for (int rr = 1; rr <= R + 1; ++rr) YY[rr] = ld(syy, kk, rr);
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr) XY[rr] = ld(sxy, kk, rr);
  };
  auto rotate = [&]() {
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr) {
      ZZ[0][rr] = ZZ[1][rr];
      XZ[0][rr] = XZ[1][rr];
      XZ[1][rr] = XZ[2][rr];
      YZ[0][rr] = YZ[1][rr];
      YZ[1][rr] = YZ[2][rr];
    }
  };
  bool dofx[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) dofx[a] = i >= g.iu_lo[a][0] && i < g.iu_hi[a][0];
  const double rux = g.rdxu[0][ic], rdx_x = g.rdx[0][ic];
  // planes k0 - 1 and k0 into the lower / middle slots
  load_upper(k0 - 1);
  rotate();
  load_upper(k0);
  rotate();
  for (int k = k0; k < k1; ++k) {
    if (bar) __builtin_amdgcn_s_barrier();  // the y-stacked wavefronts stay on one plane: their shared halo rows are cache hits
    load_upper(k + 1);
    load_own(k);
    const double ruz = g.rdxu[2][k], rdx_z = g.rdx[2][k];
    bool dofz[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) dofz[a] = k >= g.iu_lo[a][2] && k < g.iu_hi[a][2];
#pragma unroll
    for (int rr = 1; rr <= R; ++rr) {
      const int j = jb - 1 + rr;
      const int jc = min(j, N1 - 2);
      const double ruy = g.rdxu[1][jc], rdx_y = g.rdx[1][jc];
      // s_x: σxx along x; σxy: (I, I+ey, I+ex+ey, I+ex) - (I-ey, I, I+ex-ey, I+ex) over 4; σxz likewise with z
      const double xy_c = XY[rr], xy_u = XY[rr + 1], xy_d = XY[rr - 1];
      const double xz_c = XZ[1][rr], xz_u = XZ[2][rr], xz_d = XZ[0][rr];
      double ax = (lane_next(XX[rr]) - XX[rr]) * rux;
      ax += ((xy_c + xy_u + lane_next(xy_u) + lane_next(xy_c)) / 4 - (xy_d + xy_c + lane_next(xy_d) + lane_next(xy_c)) / 4) * rdx_y;
      ax += ((xz_c + xz_u + lane_next(xz_u) + lane_next(xz_c)) / 4 - (xz_d + xz_c + lane_next(xz_d) + lane_next(xz_c)) / 4) * rdx_z;
      // s_y: σxy: (I, I+ex, I+ey+ex, I+ey) - (I-ex, I, I+ey-ex, I+ey) over 4; σyy along y; σyz with z
      const double yz_c = YZ[1][rr], yz_u = YZ[2][rr], yz_d = YZ[0][rr], yz_cn = YZ[1][rr + 1], yz_un = YZ[2][rr + 1], yz_dn = YZ[0][rr + 1];
      double ay = ((xy_c + lane_next(xy_c) + lane_next(xy_u) + xy_u) / 4 - (lane_prev(xy_c) + xy_c + lane_prev(xy_u) + xy_u) / 4) * rdx_x;
      ay += (YY[rr + 1] - YY[rr]) * ruy;
      ay += ((yz_c + yz_u + yz_un + yz_cn) / 4 - (yz_d + yz_c + yz_dn + yz_cn) / 4) * rdx_z;
      // s_z: σxz: (I, I+ex, I+ez+ex, I+ez) - (I-ex, I, I+ez-ex, I+ez) over 4; σyz: (I, I+ey, I+ez+ey, I+ez) - (I-ey, I, I+ez-ey, I+ez) over 4; σzz along z
      const double yz_cd = YZ[1][rr - 1], yz_ud = YZ[2][rr - 1];
      double az = ((xz_c + lane_next(xz_c) + lane_next(xz_u) + xz_u) / 4 - (lane_prev(xz_c) + xz_c + lane_prev(xz_u) + xz_u) / 4) * rdx_x;
      az += ((yz_c + yz_cn + yz_un + yz_u) / 4 - (yz_cd + yz_c + yz_ud + yz_u) / 4) * rdx_y;
      az += (ZZ[1][rr] - ZZ[0][rr]) * ruz;
      if (xout && j < N1 - 1) {
        const long long c = i + j * g.sx[1] + k * g.sx[2];
        if (dofx[0] && dofz[0] && j >= g.iu_lo[0][1] && j < g.iu_hi[0][1]) s[c] = ax;
        if (dofx[1] && dofz[1] && j >= g.iu_lo[1][1] && j < g.iu_hi[1][1]) s[g.sc + c] = ay;
        if (dofx[2] && dofz[2] && j >= g.iu_lo[2][1] && j < g.iu_hi[2][1]) s[2 * g.sc + c] = az;
      }
    }
    rotate();
  }
}

#define INS_LAUNCH_D(KERNEL, L, S, ...)                                                \
  do {                                                                                  \
    if (g.D == 2)                                                                       \
      hipLaunchKernelGGL((KERNEL<2>), (L).grid, (L).block, 0, S, g, BoxMap{(L).ntx, (L).nty, (L).nty_l}, __VA_ARGS__);          \
    else                                                                                \
      hipLaunchKernelGGL((KERNEL<3>), (L).grid, (L).block, 0, S, g, BoxMap{(L).ntx, (L).nty, (L).nty_l}, __VA_ARGS__);          \
    INS_LAUNCH_CHECK();                                                                 \
  } while (0)

inline Launch3 ip_launch(const GridDev& g) {
  return box_launch(g.ip_hi[0] - g.ip_lo[0], g.ip_hi[1] - g.ip_lo[1], g.D == 3 ? g.ip_hi[2] - g.ip_lo[2] : 1);
}

template <int OP>
int launch_gradient_op(const ins_grid* G, double par, const double* u, double* out, hipStream_t s, const double* pI = nullptr) {
  const GridDev& g = G->g;
  if (pI) {  // uncorrected input + its pressure (all-periodic uniform 3-D boxes, the caller checks): the correcting register-row kernel
    if constexpr (OP == 2) {
      constexpr int R = 2;
      const int nx = g.ip_hi[0] - g.ip_lo[0], ny = g.ip_hi[1] - g.ip_lo[1], nz = g.ip_hi[2] - g.ip_lo[2];
      const int zc = ins_opt(OPT_INS_FIELDS_ZC) > 0 ? (int)ins_opt(OPT_INS_FIELDS_ZC) : (nz >= 128 ? 32 : (nz >= 32 ? 16 : (nz >= 8 ? 8 : nz)));
      Launch3 l;
      l.block = dim3(64, 4, 1);
      l.ntx = (int)cdiv(nx, GR_XO - 1);
      l.nty = (int)cdiv(ny, 4 * R);
      l.nty_l = (l.nty + 7) / 8;
      l.grid = dim3(8u * l.ntx * l.nty_l * cdiv(nz, zc), 1, 1);
      hipLaunchKernelGGL((k_gradient_rows<OP, R, true>), l.grid, l.block, 0, s, g, BoxMap{l.ntx, l.nty, l.nty_l}, zc, par, u, out, pI);
      INS_LAUNCH_CHECK();
      return INS_OK;
    }
    ins_set_error("in-register pressure correction: stress tensor only");
    return INS_ERR_UNSUPPORTED;
  }
  const bool march = !ins_opt(OPT_INS_FIELDS_NO_MARCH);  // A/B switch
  if (g.D == 2) {
    if constexpr (OP != 1) {
      Launch3 l = ip_launch(g);
      hipLaunchKernelGGL((k_gradient_op<2, OP>), l.grid, l.block, 0, s, g, BoxMap{l.ntx, l.nty, l.nty_l}, par, u, out);
    }
  } else if (ins_opt(OPT_INS_FIELDS_ROWS) >= 0 && g.ip_hi[0] - g.ip_lo[0] >= 32 && g.N[2] >= 4) {
    // register rows + DPP (INS_FIELDS_ROWS=-1: the older kernels).  256^3: strain dissipation 0.282 -> 0.207 ms, smagtensor 0.460 -> 0.449;
    // eig2 0.362 (plain kernel, arccosine form) -> 0.339 (plain, Newton form of eig2_of) -> 0.313 here
    // rows per work-item: 2 (with 4 rows the stress-tensor kernel needs all 256 VGPRs and runs one wavefront per SIMD); INS_FIELDS_ROWS = 2, 3, 4 overrides
    const int ro = (int)ins_opt(OPT_INS_FIELDS_ROWS);
    const int R = (ro >= 2 && ro <= 4) ? ro : 2;  // 256^3: strain dissipation 0.202 (4 rows) -> 0.186 ms, smagtensor 0.45 -> 0.33
    const int nx = g.ip_hi[0] - g.ip_lo[0], ny = g.ip_hi[1] - g.ip_lo[1], nz = g.ip_hi[2] - g.ip_lo[2];
    const int zca = ins_opt(OPT_INS_FIELDS_ZC) > 0 ? (int)ins_opt(OPT_INS_FIELDS_ZC) : (nz >= 128 ? 32 : (nz >= 32 ? 16 : (nz >= 8 ? 8 : nz)));
    const int zc = ins_opt(OPT_INS_FIELDS_NOBAR) ? -zca : zca;
    Launch3 l;
    l.block = dim3(64, 4, 1);
    l.ntx = (int)cdiv(nx, GR_XO);
    l.nty = (int)cdiv(ny, 4 * R);
    l.nty_l = (l.nty + 7) / 8;
    l.grid = dim3(8u * l.ntx * l.nty_l * cdiv(nz, zca), 1, 1);
    if (R == 2)
      hipLaunchKernelGGL((k_gradient_rows<OP, 2>), l.grid, l.block, 0, s, g, BoxMap{l.ntx, l.nty, l.nty_l}, zc, par, u, out);
    else if (R == 3)
      hipLaunchKernelGGL((k_gradient_rows<OP, 3>), l.grid, l.block, 0, s, g, BoxMap{l.ntx, l.nty, l.nty_l}, zc, par, u, out);
    else
      hipLaunchKernelGGL((k_gradient_rows<OP, 4>), l.grid, l.block, 0, s, g, BoxMap{l.ntx, l.nty, l.nty_l}, zc, par, u, out);
  } else if (!march || OP == 1) {  // eig2: its arithmetic dominates; the plain kernel measured faster (0.35 vs 0.43 ms at 256^3)
    Launch3 l = ip_launch(g);
    hipLaunchKernelGGL((k_gradient_op<3, OP>), l.grid, l.block, 0, s, g, BoxMap{l.ntx, l.nty, l.nty_l}, par, u, out);
  } else {
    const int nz = g.ip_hi[2] - g.ip_lo[2];
    const int zc = nz >= 64 ? 32 : (nz >= 16 ? 8 : nz);  // planes per z-chunk: (zc + 2) / zc re-read at chunk ends
    Launch3 l = box_launch(g.ip_hi[0] - g.ip_lo[0], g.ip_hi[1] - g.ip_lo[1], (int)cdiv(nz, zc));
    hipLaunchKernelGGL((k_gradient_march<OP>), l.grid, l.block, 0, s, g, BoxMap{l.ntx, l.nty, l.nty_l}, zc, par, u, out);
  }
  INS_LAUNCH_CHECK();
  return INS_OK;
}

}  // namespace

extern "C" int ins_vorticity_f64(const ins_grid_t* G, const double* u, double* w, void* stream) {
  INS_REQUIRE(G && u && w, "null argument");
  const GridDev& g = G->g;
  Launch3 l = box_launch(g.N[0] - 1, g.N[1] - 1, g.D == 3 ? g.N[2] - 1 : 1);
  INS_LAUNCH_D(k_vorticity, l, as_stream(stream), u, w);
  return INS_OK;
}

extern "C" int ins_interpolate_u_p_f64(const ins_grid_t* G, const double* u, double* up, void* stream) {
  INS_REQUIRE(G && u && up, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_interp_u_p, l, as_stream(stream), u, up);
  return INS_OK;
}

extern "C" int ins_interpolate_w_p_f64(const ins_grid_t* G, const double* w, double* wp, void* stream) {
  INS_REQUIRE(G && w && wp, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_interp_w_p, l, as_stream(stream), w, wp);
  return INS_OK;
}

extern "C" int ins_dfield_f64(const ins_grid_t* G, const double* p, double* Gp, double* d, double eps, void* stream) {
  INS_REQUIRE(G && p && Gp && d, "null argument");
  const GridDev& g = G->g;
  int rc = ins_pressuregradient_f64(G, p, Gp, stream);
  if (rc != INS_OK) return rc;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_Dfield, l, as_stream(stream), (const double*)Gp, d, eps);
  return INS_OK;
}

extern "C" int ins_qfield_f64(const ins_grid_t* G, const double* u, double* Q, void* stream) {
  INS_REQUIRE(G && u && Q, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_Qfield, l, as_stream(stream), u, Q);
  return INS_OK;
}

extern "C" int ins_dissipation_from_strain_f64(const ins_grid_t* G, double visc, const double* u, double* e, void* stream) {
  INS_REQUIRE(G && u && e, "null argument");
  return launch_gradient_op<0>(G, visc, u, e, as_stream(stream));
}

extern "C" int ins_eig2field_f64(const ins_grid_t* G, const double* u, double* lam, void* stream) {
  INS_REQUIRE(G && u && lam, "null argument");
  INS_REQUIRE(G->g.D == 3, "eig2 only implemented in 3D");  // operators.jl:1477
  return launch_gradient_op<1>(G, 0.0, u, lam, as_stream(stream));
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

// internal (ins_rk_ext.hip): w nullable (no dissipation term); ks / coefs: the previous ktemp_j with non-zero coefficient
int ins_k_diffusion_flux3d(const ins_grid* G, double visc, const double* u, double* F, bool zero_shell, hipStream_t s);
int ins_k_temp_stage(const ins_grid* G, double a4, double coef, const double* u, const double* temp, const double* w, const double* tempstart, int n,
                     const double* coefs, const double* const* ks, double c_self, double* ktemp_out, double* temp_out, hipStream_t s, const double* pI,
                     const double* diff) {
  const GridDev& g = G->g;
  TempStage ts;
  ts.n = 0;
  for (int q = 0; q < n && ts.n < INS_MAX_STAGES; ++q) {
    if (coefs[q] == 0.0) continue;
    ts.coef[ts.n] = coefs[q];
    ts.k[ts.n] = ks[q];
    ++ts.n;
  }
  ts.c_self = c_self;
  ts.tempstart = tempstart;
  ts.ktemp_out = ktemp_out;
  ts.temp_out = temp_out;
  // (a register-row form of this kernel — 2 rows per work-item, 158 VGPRs — measured 1-2 % faster on the all-walls 256^3 temperature loop, with 4
  // rows and 227 VGPRs 6 % slower: not kept)
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_temp_stage, l, s, a4, coef, u, temp, w, ts, pI, diff);
  return INS_OK;
}

extern "C" int ins_convection_diffusion_temp_f64(const ins_grid_t* G, double a4, const double* u, const double* temp, double* c, void* stream) {
  INS_REQUIRE(G && u && temp && c, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_convdiff_temp, l, as_stream(stream), a4, u, temp, c);
  return INS_OK;
}

extern "C" int ins_dissipation_f64(const ins_grid_t* G, double visc, double coef, const double* u, double* diff, double* diss, void* stream) {
  INS_REQUIRE(G && u && diff && diss, "null argument");
  const GridDev& g = G->g;
  // fill!(diff, 0); diffusion!(diff, u, setup)                                  operators.jl:797-798
  // (one write-only pass: zero outside the DOF ranges, the diffusion term inside)
  INS_REQUIRE(u != diff, "diffusion! cannot run in place");
  // 3-D: the tiled face-flux kernel with zero interpolation weights (0.47 -> 0.25 ms at 256^3); else the generic kernel
  int rc = g.D == 3 ? ins_k_diffusion_flux3d(G, visc, u, diff, true, as_stream(stream)) : INS_ERR_UNSUPPORTED;
  if (rc == INS_ERR_UNSUPPORTED) rc = ins_k_diffusion_overwrite(G, visc, u, diff, as_stream(stream));
  if (rc != INS_OK) return rc;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_dissipation_interp, l, as_stream(stream), coef, u, (const double*)diff, diss);
  return INS_OK;
}

extern "C" int ins_gravity_f64(const ins_grid_t* G, int gdir, double a2, const double* temp, double* F, void* stream) {
  INS_REQUIRE(G && temp && F, "null argument");
  const GridDev& g = G->g;
  INS_REQUIRE(gdir >= 0 && gdir < g.D, "gravity direction");
  Launch3 l = box_launch(g.iu_hi[gdir][0] - g.iu_lo[gdir][0], g.iu_hi[gdir][1] - g.iu_lo[gdir][1],
                         g.D == 3 ? g.iu_hi[gdir][2] - g.iu_lo[gdir][2] : 1);
  INS_LAUNCH_D(k_gravity, l, as_stream(stream), gdir, a2, temp, F);
  return INS_OK;
}

// smagtensor! of u = ustar - ∇p formed in registers (ins_rk_ext.hip; all-periodic uniform 3-D boxes with at least 32 columns)
int ins_k_smagtensor_corr(const ins_grid* G, double theta, const double* ustar, const double* pI, double* sig, hipStream_t s) {
  return launch_gradient_op<2>(G, theta, ustar, sig, s, pI);
}

extern "C" int ins_smagtensor_f64(const ins_grid_t* G, double theta, const double* u, double* sig, void* stream) {
  INS_REQUIRE(G && u && sig, "null argument");
  return launch_gradient_op<2>(G, theta, u, sig, as_stream(stream));
}

extern "C" int ins_divoftensor_f64(const ins_grid_t* G, const double* sig, double* s, void* stream) {
  INS_REQUIRE(G && sig && s, "null argument");
  const GridDev& g = G->g;
  bool inside = g.D == 3 && g.N[0] - 2 >= 32 && g.N[2] >= 4 && ins_opt(OPT_INS_FIELDS_ROWS) >= 0;
  for (int a = 0; inside && a < 3; ++a)
    for (int b = 0; b < 3; ++b) inside = inside && g.iu_lo[a][b] >= 1 && g.iu_hi[a][b] <= g.N[b] - 1;
  if (inside) {  // register rows + DPP (256^3: 0.68 -> ? ms); a PressureBC side puts DOFs into the ghost layer: plain kernel
    constexpr int R = 4;
    const int nx = g.N[0] - 2, ny = g.N[1] - 2, nz = g.N[2] - 2;
    const int zc = ins_opt(OPT_INS_FIELDS_ZC) > 0 ? (int)ins_opt(OPT_INS_FIELDS_ZC) : (nz >= 128 ? 32 : (nz >= 32 ? 16 : (nz >= 8 ? 8 : nz)));
    Launch3 l;
    l.block = dim3(64, 4, 1);
    l.ntx = (int)cdiv(nx, GR_XO);
    l.nty = (int)cdiv(ny, 4 * R);
    l.nty_l = (l.nty + 7) / 8;
    l.grid = dim3(8u * l.ntx * l.nty_l * cdiv(nz, zc), 1, 1);
    hipLaunchKernelGGL((k_divoftensor_rows<R>), l.grid, l.block, 0, as_stream(stream), g, BoxMap{l.ntx, l.nty, l.nty_l}, ins_opt(OPT_INS_FIELDS_NOBAR) ? -zc : zc, sig, s);
    INS_LAUNCH_CHECK();
    return INS_OK;
  }
  Launch3 l = box_launch(g.N[0], g.N[1], g.D == 3 ? g.N[2] : 1);
  INS_LAUNCH_D(k_divoftensor, l, as_stream(stream), sig, s);
  return INS_OK;
}

extern "C" int ins_tensorbasis_f64(const ins_grid_t* G, const double* u, double* B, double* V, void* stream) {
  INS_REQUIRE(G && u && B && V, "null argument");
  const GridDev& g = G->g;
  Launch3 l = ip_launch(g);
  INS_LAUNCH_D(k_tensorbasis, l, as_stream(stream), u, B, V);
  return INS_OK;
}
