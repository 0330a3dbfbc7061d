// Pressure-Poisson solvers and the projection (pressure.jl).
//   spectral: rocFFT (through the hipFFT API) D2Z / Z2D on the UNPADDED pressure block + one fused
//             k-space kernel (symbol division, mean-mode zero, 1/prod(Np) normalisation).
//   cg      : Jacobi-preconditioned CG with fused vector-update + reduction kernels.
//   project : for the spectral solver the ghost strip / re-pad copies, scalewithvolume! and the periodic
//             apply_bc_p! are folded into the divergence and gradient kernels.
#include <cmath>
#include <cstdlib>

#include "ins_internal.h"

namespace {

// K2: pI[i,j,k] = Ω_I * div(u)[Ip.lo + (i,j,k)]     (divergence! + scalewithvolume! + copyto!(pI, view(p, Ip)),
//                                                    operators.jl:117-125, 81-95, pressure.jl:320)
// WRAP: read u[I - e_a] through the periodic image instead of the ghost volume, so the stage velocity needs no
// ghost fill before the projection (apply_bc_u! at step_explicit_runge_kutta.jl:48 folded away).
template <int D, bool WRAP = false>
__global__ __launch_bounds__(256) void k_div_to_pI(GridDev g, const double* __restrict__ u, double* __restrict__ pI, int n0, int n1) {
  const int ii = blockIdx.x * 64 + threadIdx.x;
  const int jj = blockIdx.y * 4 + threadIdx.y;
  const int kk = D == 3 ? (int)blockIdx.z : 0;
  if (ii >= n0 || jj >= n1) return;
  const int I[3] = {g.ip_lo[0] + ii, g.ip_lo[1] + jj, D == 3 ? g.ip_lo[2] + kk : 0};
  const long long c = I[0] + I[1] * g.sx[1] + I[2] * g.sx[2];
  double d = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double* ua = u + a * g.sc;
    const long long cm = (WRAP && I[a] == 1) ? c + (long long)(g.N[a] - 3) * g.sx[a] : c - g.sx[a];
    d += (ua[c] - ua[cm]) * g.rdx[a][I[a]];
  }
  double om = g.dx[0][I[0]] * g.dx[1][I[1]];
  if (D == 3) om = om * g.dx[2][I[2]];
  pI[ii + (long long)n0 * (jj + (long long)n1 * kk)] = d * om;
}

// The same right-hand side written in the FOLDED form of the direct solver's symmetric directions (csrc/ins_fdm.hip: e_i = f_i + f_{n-1-i} at i,
// o_i = f_i - f_{n-1-i} at n/2 + i, i < n/2, along every direction of `mask`): a work-item evaluates the divergence at the 2^k mirror images
// of its volume and writes their sums / differences, so the solver needs no fold pass of its own.
template <int D, int MASK>  // MASK is a template parameter: the image loops unroll and v[][][] stays in registers (run-time bounds put it in scratch: 176 -> ? us at 256^3)
__global__ __launch_bounds__(256) void k_div_to_pI_fold(GridDev g, const double* __restrict__ u, double* __restrict__ pI, int n0, int n1, int n2) {
  constexpr int c0 = (MASK & 1) ? 2 : 1, c1 = (MASK & 2) ? 2 : 1, c2 = (D == 3 && (MASK & 4)) ? 2 : 1;
  const int h0 = c0 == 2 ? n0 / 2 : n0, h1 = c1 == 2 ? n1 / 2 : n1, h2 = c2 == 2 ? n2 / 2 : n2;
  const int ii = blockIdx.x * 64 + threadIdx.x;
  const int jj = blockIdx.y * 4 + threadIdx.y;
  const int kk = D == 3 ? (int)blockIdx.z : 0;
  if (ii >= h0 || jj >= h1 || kk >= h2) return;
  double v[c2][c1][c0];
  // metrics of the images: one read per direction and image instead of one per volume
  double rd[3][2], dd[3][2];
  int Ia[3][2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    Ia[0][m] = g.ip_lo[0] + (m ? n0 - 1 - ii : ii);
    Ia[1][m] = g.ip_lo[1] + (m ? n1 - 1 - jj : jj);
    Ia[2][m] = D == 3 ? g.ip_lo[2] + (m ? n2 - 1 - kk : kk) : 0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      if (m >= (a == 0 ? c0 : (a == 1 ? c1 : c2))) continue;
      rd[a][m] = g.rdx[a][Ia[a][m]];
      dd[a][m] = g.dx[a][Ia[a][m]];
    }
  }
#pragma unroll
  for (int cz = 0; cz < c2; ++cz)
#pragma unroll
    for (int cy = 0; cy < c1; ++cy)
#pragma unroll
      for (int cx = 0; cx < c0; ++cx) {
        const int m[3] = {cx, cy, cz};
        const long long c = Ia[0][cx] + Ia[1][cy] * g.sx[1] + Ia[2][cz] * g.sx[2];
        double d = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
          const double* ua = u + a * g.sc;
          d += (ua[c] - ua[c - g.sx[a]]) * rd[a][m[a]];
        }
        double om = dd[0][cx] * dd[1][cy];
        if (D == 3) om = om * dd[2][cz];
        v[cz][cy][cx] = d * om;
      }
  if (c0 == 2)
#pragma unroll
    for (int cz = 0; cz < c2; ++cz)
#pragma unroll
      for (int cy = 0; cy < c1; ++cy) {
        const double p = v[cz][cy][0], q = v[cz][cy][c0 - 1];
        v[cz][cy][0] = p + q;
        v[cz][cy][c0 - 1] = p - q;
      }
  if (c1 == 2)
#pragma unroll
    for (int cz = 0; cz < c2; ++cz)
#pragma unroll
      for (int cx = 0; cx < c0; ++cx) {
        const double p = v[cz][0][cx], q = v[cz][c1 - 1][cx];
        v[cz][0][cx] = p + q;
        v[cz][c1 - 1][cx] = p - q;
      }
  if (c2 == 2)
#pragma unroll
    for (int cy = 0; cy < c1; ++cy)
#pragma unroll
      for (int cx = 0; cx < c0; ++cx) {
        const double p = v[0][cy][cx], q = v[c2 - 1][cy][cx];
        v[0][cy][cx] = p + q;
        v[c2 - 1][cy][cx] = p - q;
      }
#pragma unroll
  for (int cz = 0; cz < c2; ++cz)
#pragma unroll
    for (int cy = 0; cy < c1; ++cy)
#pragma unroll
      for (int cx = 0; cx < c0; ++cx)
        pI[(cx ? n0 / 2 + ii : ii) + (long long)n0 * ((cy ? n1 / 2 + jj : jj) + (long long)n1 * (cz ? n2 / 2 + kk : kk))] = v[cz][cy][cx];
}

// value of the solution at interior offsets (w0, w1, w2) when the solver's buffer holds it in folded form (inverse of the fold above)
__device__ __forceinline__ double unfold_at(const double* __restrict__ pI, int w0, int w1, int w2, int n0, int n1, int n2, int mask) {
  const bool f0 = mask & 1, f1 = mask & 2, f2 = mask & 4;
  const int h0 = n0 / 2, h1 = n1 / 2, h2 = n2 / 2;
  const int i = (f0 && w0 >= h0) ? n0 - 1 - w0 : w0, j = (f1 && w1 >= h1) ? n1 - 1 - w1 : w1, k = (f2 && w2 >= h2) ? n2 - 1 - w2 : w2;
  const double s0 = (f0 && w0 >= h0) ? -1.0 : 1.0, s1 = (f1 && w1 >= h1) ? -1.0 : 1.0, s2 = (f2 && w2 >= h2) ? -1.0 : 1.0;
  double acc = 0.0;
  for (int cz = 0; cz < (f2 ? 2 : 1); ++cz)
    for (int cy = 0; cy < (f1 ? 2 : 1); ++cy)
      for (int cx = 0; cx < (f0 ? 2 : 1); ++cx) {
        const double sg = (cx ? s0 : 1.0) * (cy ? s1 : 1.0) * (cz ? s2 : 1.0);
        acc += sg * pI[(cx ? h0 + i : i) + (long long)n0 * ((cy ? h1 + j : j) + (long long)n1 * (cz ? h2 + k : k))];
      }
  return acc;
}

// copyto!(pI, view(p, Ip))  /  copyto!(view(p, Ip), pI)                        pressure.jl:320, 347
template <int D, bool PACK>
__global__ __launch_bounds__(256) void k_pack(GridDev g, double* __restrict__ p, double* __restrict__ pI, int n0, int n1,
                                              const double* __restrict__ shift = nullptr) {
  const int ii = blockIdx.x * 64 + threadIdx.x;
  const int jj = blockIdx.y * 4 + threadIdx.y;
  const int kk = D == 3 ? (int)blockIdx.z : 0;
  if (ii >= n0 || jj >= n1) return;
  const long long c = (g.ip_lo[0] + ii) + (g.ip_lo[1] + jj) * g.sx[1] + (D == 3 ? (g.ip_lo[2] + kk) * g.sx[2] : 0);
  const long long q = ii + (long long)n0 * (jj + (long long)n1 * kk);
  if (PACK)
    pI[q] = p[c];
  else
    p[c] = shift ? pI[q] - *shift : pI[q];
}

// psolver_direct inside project!: copyto!(view(p, Ip), pI) - mean + apply_bc_p! + applypressure! in one pass over the box
// Ip grown by one ghost layer (pressure.jl:150, boundary_conditions.jl:306-318 / 388 / 445-453 / 497-502, operators.jl:225-233).
// Ghost values follow apply_bc_p!'s rules composed over the directions: periodic image, copy of the adjacent volume (Symmetric),
// zero (Pressure); volumes behind a Dirichlet side are left untouched as in the reference (never read: no velocity DOF touches them).
__device__ __forceinline__ int map_p(int I, int lo, int hi, int bcl, int bcr) {  // -> offset in [0, hi-lo), -1: zero, -2: untouched
  if (I >= lo && I < hi) return I - lo;
  const bool left = I < lo;
  const int bc = left ? bcl : bcr;
  if (bc == INS_BC_PERIODIC) return left ? hi - lo - 1 : 0;
  if (bc == INS_BC_SYMMETRIC) return left ? 0 : hi - lo - 1;
  if (bc == INS_BC_PRESSURE) return -1;
  return -2;
}

template <int D, bool GRAD = true>
__global__ __launch_bounds__(256) void k_unpack_grad_bc(GridDev g, double* __restrict__ u, double* __restrict__ p, const double* __restrict__ pI,
                                                        int n0, int n1, const double* __restrict__ shift, int fmask = 0, int n2 = 1) {
  const int I0 = g.ip_lo[0] - 1 + blockIdx.x * 64 + threadIdx.x;
  const int I1 = g.ip_lo[1] - 1 + blockIdx.y * 4 + threadIdx.y;
  const int I2 = D == 3 ? g.ip_lo[2] - 1 + (int)blockIdx.z : 0;
  if (I0 > g.ip_hi[0] || I1 > g.ip_hi[1]) return;
  const int I[3] = {I0, I1, I2};
  const long long qs[3] = {1, n0, (long long)n0 * n1};
  int w[3] = {0, 0, 0};
  bool zero = false, skip = false;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    w[a] = map_p(I[a], g.ip_lo[a], g.ip_hi[a], g.bc[a][0], g.bc[a][1]);
    zero = zero || w[a] == -1;
    skip = skip || w[a] == -2;
  }
  if (skip) return;
  const double sh = shift ? *shift : 0.0;
  long long q = 0;
#pragma unroll
  for (int a = 0; a < D; ++a) q += (long long)(w[a] < 0 ? 0 : w[a]) * qs[a];
  auto val = [&](const int (&ww)[3], long long qq) {  // solution at interior offsets ww (flat index qq): unfolded on the fly when the solver kept it folded
    return fmask ? unfold_at(pI, ww[0] < 0 ? 0 : ww[0], ww[1] < 0 ? 0 : ww[1], ww[2] < 0 ? 0 : ww[2], n0, n1, n2, fmask) : pI[qq];
  };
  const double pc = zero ? 0.0 : val(w, q) - sh;
  const long long c = I[0] + I[1] * g.sx[1] + I[2] * g.sx[2];
  p[c] = pc;
  if (!GRAD) return;  // solve only: the gradient-subtract is left to the next stage's stencil kernel (ins_fast3d_flux.hip, CORR = 3)
#pragma unroll
  for (int a = 0; a < D; ++a) {
    bool dof = true;
#pragma unroll
    for (int b = 0; b < D; ++b) dof = dof && I[b] >= g.iu_lo[a][b] && I[b] < g.iu_hi[a][b];
    if (!dof) continue;
    const int wn = map_p(I[a] + 1, g.ip_lo[a], g.ip_hi[a], g.bc[a][0], g.bc[a][1]);
    bool zn = wn == -1;
#pragma unroll
    for (int b = 0; b < D; ++b) zn = zn || (b != a && w[b] == -1);
    const long long qn = q + (long long)((wn < 0 ? 0 : wn) - (w[a] < 0 ? 0 : w[a])) * qs[a];  // direction a replaced, the others kept
    int wnn[3] = {w[0], w[1], w[2]};
    wnn[a] = wn;
    const double pn = zn ? 0.0 : val(wnn, qn) - sh;
    u[a * g.sc + c] -= (pn - pc) * g.rdxu[a][I[a]];
  }
}

// The same pass for a 3-D solution kept FOLDED along any subset of its directions: a work-item owns the 2, 4 or 8 mirror images of its volume,
// so each folded value is read once for all of them (the kernel above gathers 2^k values per pressure it needs: 32 per volume with the
// gradient when all three directions are folded).  Neighbours: six more such reads — (i±1, j, k), (i, j±1, k), (i, j, k±1) — serve all
// images (the +x neighbour of the image n0-1-i is the image of i-1).  Ghost pressures are written by the owner of the adjacent interior
// volume, composed over the directions as map_p does.
struct UnfoldQ {
  double v[2][2][2];  // [image in z][image in y][image in x]
};
template <bool GRAD>
__global__ __launch_bounds__(256) void k_unfold_grad3(GridDev g, double* __restrict__ u, double* __restrict__ p, const double* __restrict__ pI,
                                                      int n0, int n1, int n2, const double* __restrict__ shift, int fmask) {
  const bool f[3] = {(fmask & 1) != 0, (fmask & 2) != 0, (fmask & 4) != 0};
  const int n[3] = {n0, n1, n2};
  const int h[3] = {f[0] ? n0 / 2 : n0, f[1] ? n1 / 2 : n1, f[2] ? n2 / 2 : n2};
  const int id[3] = {(int)(blockIdx.x * 64 + threadIdx.x), (int)(blockIdx.y * 4 + threadIdx.y), (int)blockIdx.z};
  if (id[0] >= h[0] || id[1] >= h[1]) return;
  const double sh = shift ? *shift : 0.0;
  const long long s1 = n0, s2 = (long long)n0 * n1;
  const long long hs[3] = {h[0], s1 * h[1], s2 * h[2]};  // offset of the odd half along each direction
  auto fetch = [&](int i, int j, int k) {
    const double* b = pI + i + s1 * j + s2 * k;
    double a[2][2][2];
#pragma unroll
    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
      for (int cy = 0; cy < 2; ++cy)
#pragma unroll
        for (int cx = 0; cx < 2; ++cx) {
          const bool have = (!cx || f[0]) && (!cy || f[1]) && (!cz || f[2]);
          a[cz][cy][cx] = have ? b[(cx ? hs[0] : 0) + (cy ? hs[1] : 0) + (cz ? hs[2] : 0)] : 0.0;
        }
    UnfoldQ q;  // the summation order of unfold_at: odd-half terms that do not exist are exact zeros
#pragma unroll
    for (int mz = 0; mz < 2; ++mz)
#pragma unroll
      for (int my = 0; my < 2; ++my)
#pragma unroll
        for (int mx = 0; mx < 2; ++mx) {
          double acc = 0.0;
#pragma unroll
          for (int cz = 0; cz < 2; ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
              for (int cx = 0; cx < 2; ++cx) {
                const bool neg = ((mx & cx) ^ (my & cy) ^ (mz & cz)) != 0;
                acc += neg ? -a[cz][cy][cx] : a[cz][cy][cx];
              }
          q.v[mz][my][mx] = acc - sh;
        }
    return q;
  };
  const UnfoldQ c = fetch(id[0], id[1], id[2]);
  UnfoldQ P[3] = {c, c, c}, M[3] = {c, c, c};  // the neighbours at index + 1 / index - 1 along each direction
  bool per[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) per[a] = g.bc[a][0] == INS_BC_PERIODIC;
  if (GRAD) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int q[3] = {id[0], id[1], id[2]};
      if (id[a] + 1 < h[a] || (!f[a] && per[a])) {  // unfolded periodic direction: past the last volume comes the first
        q[a] = id[a] + 1 < h[a] ? id[a] + 1 : 0;
        P[a] = fetch(q[0], q[1], q[2]);
      }
      if (f[a] && id[a] >= 1) {
        q[a] = id[a] - 1;
        M[a] = fetch(q[0], q[1], q[2]);
      }
    }
  }
  // (no side is a PressureBC here — the host routes those grids to the kernel above —, so a pressure just outside the box is the copy of the
  // adjacent interior value behind a Symmetric side and is read by no DOF behind a Dirichlet one)
#pragma unroll
  for (int mz = 0; mz < 2; ++mz) {
    if (mz && !f[2]) break;
#pragma unroll
    for (int my = 0; my < 2; ++my) {
      if (my && !f[1]) break;
#pragma unroll
      for (int mx = 0; mx < 2; ++mx) {
        if (mx && !f[0]) break;
        const int m[3] = {mx, my, mz};
        int w[3], I[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          w[a] = m[a] ? n[a] - 1 - id[a] : id[a];
          I[a] = g.ip_lo[a] + w[a];
        }
        const double pc = c.v[mz][my][mx];
        // ---- p: the volume itself and every ghost volume that takes its value from it
        int tgt[3][3], nt[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          nt[a] = 1;
          tgt[a][0] = I[a];
          if (w[a] == 0) {
            if (g.bc[a][1] == INS_BC_PERIODIC) tgt[a][nt[a]++] = g.ip_hi[a];
            if (g.bc[a][0] == INS_BC_SYMMETRIC) tgt[a][nt[a]++] = g.ip_lo[a] - 1;
          }
          if (w[a] == n[a] - 1 && nt[a] < 3) {
            if (g.bc[a][0] == INS_BC_PERIODIC) tgt[a][nt[a]++] = g.ip_lo[a] - 1;
            if (g.bc[a][1] == INS_BC_SYMMETRIC) tgt[a][nt[a]++] = g.ip_hi[a];
          }
        }
        for (int t2 = 0; t2 < nt[2]; ++t2)
          for (int t1 = 0; t1 < nt[1]; ++t1)
            for (int t0 = 0; t0 < nt[0]; ++t0) p[tgt[0][t0] + tgt[1][t1] * g.sx[1] + tgt[2][t2] * g.sx[2]] = pc;
        if (!GRAD) continue;
        // ---- u -= ∇p on the degrees of freedom of this volume.  The neighbour at coordinate + 1 along a: image 0 — the next index (or, past
        // the middle of a folded direction, image 1 of this work-item's own values); image 1 — image 1 of the previous index; past the last
        // volume: the first volume's value when the direction is periodic, the copy pc behind a Symmetric side, unread behind a Dirichlet one.
        const long long cc = I[0] + I[1] * g.sx[1] + I[2] * g.sx[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          int mm[3] = {mx, my, mz};
          double pn;
          if (m[a] == 0) {
            if (id[a] + 1 < h[a]) {
              pn = P[a].v[mm[2]][mm[1]][mm[0]];
            } else if (f[a]) {
              mm[a] = 1;
              pn = c.v[mm[2]][mm[1]][mm[0]];
            } else {
              pn = per[a] ? P[a].v[mm[2]][mm[1]][mm[0]] : pc;
            }
          } else {
            if (id[a] >= 1) {
              pn = M[a].v[mm[2]][mm[1]][mm[0]];
            } else {
              mm[a] = 0;
              pn = per[a] ? c.v[mm[2]][mm[1]][mm[0]] : pc;
            }
          }
          bool dof = true;
#pragma unroll
          for (int b = 0; b < 3; ++b) dof = dof && I[b] >= g.iu_lo[a][b] && I[b] < g.iu_hi[a][b];
          if (dof) u[a * g.sc + cc] -= (pn - pc) * g.rdxu[a][I[a]];
        }
      }
    }
  }
}

// K3: phat = -phat / (ax + ay + az) * (1/prod(Np));  phat[0] = 0                 pressure.jl:326-341
// (the normalisation of `ldiv!(pI, plan, phat)` is folded in: hipFFT's Z2D is unnormalised)
template <int D>
__global__ __launch_bounds__(256) void k_symbol(hipfftDoubleComplex* __restrict__ phat, const double* __restrict__ ax,
                                                const double* __restrict__ ay, const double* __restrict__ az, int k0, int k1,
                                                double inv_n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= k0 || j >= k1) return;
  const long long q = i + (long long)k0 * (j + (long long)k1 * k);
  double den = ax[i] + ay[j];
  if (D == 3) den = den + az[k];
  hipfftDoubleComplex v = phat[q];
  const double s = (q == 0) ? 0.0 : -inv_n / den;
  v.x *= s;
  v.y *= s;
  phat[q] = v;
}

// K4 (periodic, spectral): u[I,α] -= (pI[wrap(I+eα)] - pI[I]) / Δu[α][Iα] on the interior, and the padded,
// ghost-filled `p` is written in the same pass (copyto!(view(p,Ip),pI) + apply_bc_p! + applypressure!,
// pressure.jl:347, boundary_conditions.jl:306-318, operators.jl:225-233).
template <int D>
__global__ __launch_bounds__(256) void k_grad_from_pI(GridDev g, double* __restrict__ u, double* __restrict__ p,
                                                      const double* __restrict__ pI, int n0, int n1, int n2) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y * 4 + threadIdx.y;
  const int k = D == 3 ? (int)blockIdx.z : 0;
  if (i >= g.N[0] || j >= g.N[1]) return;
  const int n[3] = {n0, n1, n2};
  const int I[3] = {i, j, k};
  int w[3];
  bool interior = true;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    int q = I[a] - 1;  // Ip.lo == 1 for periodic
    interior = interior && q >= 0 && q < n[a];
    q = q < 0 ? q + n[a] : (q >= n[a] ? q - n[a] : q);
    w[a] = q;
  }
  if (D == 2) w[2] = 0;
  const long long q = w[0] + (long long)n0 * (w[1] + (long long)n1 * w[2]);
  const long long c = i + j * g.sx[1] + k * g.sx[2];
  const double pc = pI[q];
  p[c] = pc;
  if (!interior) return;
  const long long qs[3] = {1, n0, (long long)n0 * n1};
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const long long qn = (w[a] + 1 < n[a]) ? q + qs[a] : q - (long long)(n[a] - 1) * qs[a];
    u[a * g.sc + c] -= (pI[qn] - pc) * g.rdxu[a][I[a]];
  }
}

// K4 for the fused periodic RK path (3-D): one work-item per interior volume subtracts the pressure gradient
// (operators.jl:225-233, p read from the unpadded pI with periodic wrap) and ALSO writes the periodic ghost
// images of the updated velocity (boundary_conditions.jl:276-288) — up to 7 images for a corner volume — so no
// separate apply_bc_u! pass is needed before the next stage.  KEEP_P: also store the padded, ghost-filled p.
template <int D, bool KEEP_P>
__global__ __launch_bounds__(256) void k_grad_ghost(GridDev g, double* __restrict__ u, double* __restrict__ p, const double* __restrict__ pI, int n0,
                                                    int n1, int n2) {
  const int ii = blockIdx.x * 64 + threadIdx.x;
  const int jj = blockIdx.y * 4 + threadIdx.y;
  const int kk = D == 3 ? (int)blockIdx.z : 0;
  if (ii >= n0 || jj >= n1) return;
  const int n[3] = {n0, n1, n2};
  const int w[3] = {ii, jj, kk};
  const int I[3] = {ii + 1, jj + 1, D == 3 ? kk + 1 : 0};
  const long long q = ii + (long long)n0 * (jj + (long long)n1 * kk);
  const long long qs[3] = {1, n0, (long long)n0 * n1};
  long long c = I[0] + I[1] * g.sx[1];
  if (D == 3) c += I[2] * g.sx[2];
  const double pc = pI[q];
  double un[D];
  int img[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const long long qn = (w[a] + 1 < n[a]) ? q + qs[a] : q - (long long)(n[a] - 1) * qs[a];
    un[a] = u[a * g.sc + c] - (pI[qn] - pc) * g.rdxu[a][I[a]];
    img[a] = I[a] == 1 ? g.N[a] - 1 : (I[a] == g.N[a] - 2 ? 0 : -1);
  }
#pragma unroll
  for (int m = 0; m < (1 << D); ++m) {
    bool ok = true;
    long long cc = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const bool use = (m >> a) & 1;
      ok = ok && (!use || img[a] >= 0);
      cc += (long long)(use ? img[a] : I[a]) * g.sx[a];
    }
    if (ok) {
#pragma unroll
      for (int a = 0; a < D; ++a) u[cc + a * g.sc] = un[a];
      if (KEEP_P) p[cc] = pc;
    }
  }
}
// uin: the uncorrected field (u itself for the in-place form; another array keeps the uncorrected stage velocity intact: ins_rk_ext.hip)
template <bool KEEP_P>
__global__ __launch_bounds__(256) void k_grad_ghost3(GridDev g, double* u, double* __restrict__ p, const double* __restrict__ pI, int n0, int n1, int n2,
                                                     const double* uin) {
  const int ii = blockIdx.x * 64 + threadIdx.x;
  const int jj = blockIdx.y * 4 + threadIdx.y;
  const int kk = blockIdx.z;
  if (ii >= n0 || jj >= n1) return;
  const int n[3] = {n0, n1, n2};
  const int w[3] = {ii, jj, kk};
  const int I[3] = {ii + 1, jj + 1, kk + 1};
  const long long q = ii + (long long)n0 * (jj + (long long)n1 * kk);
  const long long qs[3] = {1, n0, (long long)n0 * n1};
  const long long c = I[0] + I[1] * g.sx[1] + I[2] * g.sx[2];
  const double pc = pI[q];
  double un[3];
  int img[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const long long qn = (w[a] + 1 < n[a]) ? q + qs[a] : q - (long long)(n[a] - 1) * qs[a];
    un[a] = uin[a * g.sc + c] - (pI[qn] - pc) * g.rdxu[a][I[a]];
    img[a] = I[a] == 1 ? g.N[a] - 1 : (I[a] == g.N[a] - 2 ? 0 : -1);
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    bool ok = true;
    long long cc = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const bool use = (m >> a) & 1;
      ok = ok && (!use || img[a] >= 0);
      cc += (long long)(use ? img[a] : I[a]) * g.sx[a];
    }
    if (ok) {
      u[cc] = un[0];
      u[cc + g.sc] = un[1];
      u[cc + 2 * g.sc] = un[2];
      if (KEEP_P) p[cc] = pc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// CG kernels (pressure.jl:222-285).  Whole-array updates exactly as the reference's broadcasts; the
// reductions run over Ip.  Each fused kernel leaves per-block partial sums in `partial`.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* lds) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int w = (threadIdx.y * 64 + threadIdx.x) >> 6;
  if (threadIdx.x == 0) lds[w] = v;
  __syncthreads();
  return lds[0] + lds[1] + lds[2] + lds[3];
}

template <int D>
__device__ __forceinline__ bool cell(const GridDev& g, int (&I)[3], long long& c, bool& inp) {
  I[0] = blockIdx.x * 64 + threadIdx.x;
  I[1] = blockIdx.y * 4 + threadIdx.y;
  I[2] = D == 3 ? (int)blockIdx.z : 0;
  const bool ok = I[0] < g.N[0] && I[1] < g.N[1];
  c = I[0] + I[1] * g.sx[1] + I[2] * g.sx[2];
  inp = ok;
#pragma unroll
  for (int a = 0; a < D; ++a) inp = inp && I[a] >= g.ip_lo[a] && I[a] < g.ip_hi[a];
  return ok;
}

__device__ __forceinline__ void store_partial(double v, double* partial) {
  if (threadIdx.x == 0 && threadIdx.y == 0) partial[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = v;
}

// p[Ip] -= shift
template <int D>
__global__ __launch_bounds__(256) void k_shift(GridDev g, double* __restrict__ p, double shift) {
  int I[3];
  long long c;
  bool inp;
  if (cell<D>(g, I, c, inp) && inp) p[c] -= shift;
}

// dinv = 1/d with d of pressure.jl:196-199
template <int D>
__global__ __launch_bounds__(256) void k_cg_diag(GridDev g, double* __restrict__ dinv) {
  int I[3];
  long long c;
  bool inp;
  if (!cell<D>(g, I, c, inp)) return;
  double v = 0.0;
  if (inp) {
    double om = g.dx[0][I[0]] * g.dx[1][I[1]];
    if (D == 3) om = om * g.dx[2][I[2]];
    double d = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) d -= om / g.dx[a][I[a]] * (1 / g.dxu[a][I[a]] + 1 / g.dxu[a][I[a] - 1]);
    v = 1.0 / d;
  }
  dinv[c] = v;
}

// r = p; p = 0; partial Σ r²       (q .= 0; L = lap(q) = 0; r .= p .- L; residual; p .= 0 — pressure.jl:239-249)
template <int D>
__global__ __launch_bounds__(256) void k_cg_init(GridDev g, double* __restrict__ p, double* __restrict__ r, double* __restrict__ q,
                                                 double* __restrict__ L, double* __restrict__ partial) {
  __shared__ double lds[4];
  int I[3];
  long long c;
  bool inp;
  const bool ok = cell<D>(g, I, c, inp);
  double s = 0.0;
  if (ok) {
    const double v = p[c];
    r[c] = v;
    p[c] = 0.0;
    q[c] = 0.0;
    L[c] = 0.0;
    if (inp) s = v * v;
  }
  s = block_sum(s, lds);
  store_partial(s, partial);
}

// L[Ip] = -r/d (preconditioner);  partial Σ L·r                                   pressure.jl:252-255
template <int D>
__global__ __launch_bounds__(256) void k_cg_precond(GridDev g, const double* __restrict__ r, const double* __restrict__ dinv,
                                                    double* __restrict__ L, double* __restrict__ partial) {
  __shared__ double lds[4];
  int I[3];
  long long c;
  bool inp;
  const bool ok = cell<D>(g, I, c, inp);
  double s = 0.0;
  if (ok && inp) {
    const double rv = r[c];
    const double z = -rv * dinv[c];
    L[c] = z;
    s = z * rv;
  }
  s = block_sum(s, lds);
  store_partial(s, partial);
}

// q = L + β q  (whole array)                                                       pressure.jl:260
__global__ __launch_bounds__(256) void k_cg_dir(long long n, double beta, const double* __restrict__ L, double* __restrict__ q) {
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) q[t] = L[t] + beta * q[t];
}

// partial Σ q·L over Ip                                                            pressure.jl:266
template <int D>
__global__ __launch_bounds__(256) void k_cg_dot(GridDev g, const double* __restrict__ a, const double* __restrict__ b,
                                                double* __restrict__ partial) {
  __shared__ double lds[4];
  int I[3];
  long long c;
  bool inp;
  const bool ok = cell<D>(g, I, c, inp);
  double s = (ok && inp) ? a[c] * b[c] : 0.0;
  s = block_sum(s, lds);
  store_partial(s, partial);
}

// p += α q; r -= α L (whole arrays); partial Σ r² over Ip                      pressure.jl:270-275
template <int D>
__global__ __launch_bounds__(256) void k_cg_update(GridDev g, double alpha, double* __restrict__ p, double* __restrict__ r,
                                                   const double* __restrict__ q, const double* __restrict__ L,
                                                   double* __restrict__ partial) {
  __shared__ double lds[4];
  int I[3];
  long long c;
  bool inp;
  const bool ok = cell<D>(g, I, c, inp);
  double s = 0.0;
  if (ok) {
    p[c] += alpha * q[c];
    const double rv = r[c] - alpha * L[c];
    r[c] = rv;
    if (inp) s = rv * rv;
  }
  s = block_sum(s, lds);
  store_partial(s, partial);
}


// ------------------------------------------------------------------------------------------------
// The same iteration with every scalar on the device (pressure.jl:244-280 reads three scalars per iteration on the host: two dots
// and a norm).  Flat grid-stride kernels leave NPART block partials; a one-block kernel folds them and derives α / β / the residual /
// the stopping flag in device memory (slot layout below); the vector kernels read what they need from there and do nothing once
// the flag is set.  The host looks at the flag once per batch of iterations, so a solve costs iterations/batch synchronisations
// instead of 3 x iterations, and the stopping rule (and hence the iterate) is exactly the reference's.  With a communicator
// (z-slabs) the folded sums are all-reduced across ranks between the two steps and q's ghost planes travel before laplacian!.
// ------------------------------------------------------------------------------------------------
enum { CG_RHO = 0, CG_RHO_PREV, CG_QL, CG_SS, CG_TOL, CG_DONE, CG_ITERS, CG_ALPHA, CG_BETA, CG_RES, CG_SUM, CG_NSCAL = 16 };
constexpr int CG_NPART = 1024;

__device__ __forceinline__ bool flat_cell(const GridDev& g, long long t, bool& inp) {
  const int i0 = (int)(t % g.N[0]);
  const long long r = t / g.N[0];
  const int i1 = (int)(r % g.N[1]), i2 = (int)(r / g.N[1]);
  inp = i0 >= g.ip_lo[0] && i0 < g.ip_hi[0] && i1 >= g.ip_lo[1] && i1 < g.ip_hi[1] && (g.D == 2 || (i2 >= g.ip_lo[2] && i2 < g.ip_hi[2]));
  return true;
}
__device__ __forceinline__ void flat_partial(double v, double* partial) {
  __shared__ double lds[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
// sc[CG_SUM] = Σ partial (one block)
__global__ __launch_bounds__(1024) void k_cgd_fold(const double* __restrict__ partial, int n, double* __restrict__ sc) {
  __shared__ double lds[16];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) v += partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += lds[w];
    sc[CG_SUM] = t;
  }
}
// stage 0: after init (Σ r²); 1: after the preconditioner (ρ, β); 2: after q·L (α); 3: after the update (Σ r², iteration count, stopping flag)
__global__ void k_cgd_scalars(double* __restrict__ sc, int stage, double reltol, double abstol, double maxiter) {
  const double v = sc[CG_SUM];
  if (stage == 0) {
    const double res = sqrt(v);
    sc[CG_SS] = v;
    sc[CG_RES] = res;
    sc[CG_TOL] = fmax(reltol * res, abstol);
    sc[CG_RHO_PREV] = 1.0;
    sc[CG_ITERS] = 0.0;
    sc[CG_DONE] = (0.0 < maxiter && res > sc[CG_TOL]) ? 0.0 : 1.0;
    return;
  }
  if (sc[CG_DONE] != 0.0) return;
  if (stage == 1) {
    sc[CG_RHO] = v;
    sc[CG_BETA] = v / sc[CG_RHO_PREV];
  } else if (stage == 2) {
    sc[CG_QL] = v;
    sc[CG_ALPHA] = sc[CG_RHO] / v;
  } else {
    const double res = sqrt(v);
    sc[CG_SS] = v;
    sc[CG_RES] = res;
    sc[CG_RHO_PREV] = sc[CG_RHO];
    const double it = sc[CG_ITERS] + 1.0;
    sc[CG_ITERS] = it;
    sc[CG_DONE] = (it < maxiter && res > sc[CG_TOL]) ? 0.0 : 1.0;
  }
}
__global__ __launch_bounds__(256) void k_cgd_init(GridDev g, long long n, double* __restrict__ p, double* __restrict__ r, double* __restrict__ q,
                                                  double* __restrict__ L, double* __restrict__ partial) {
  double s = 0.0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    bool inp;
    flat_cell(g, t, inp);
    const double v = p[t];
    r[t] = v;
    p[t] = 0.0;
    q[t] = 0.0;
    L[t] = 0.0;
    if (inp) s += v * v;
  }
  flat_partial(s, partial);
}
__global__ __launch_bounds__(256) void k_cgd_precond(GridDev g, long long n, const double* __restrict__ sc, const double* __restrict__ r,
                                                     const double* __restrict__ dinv, double* __restrict__ L, double* __restrict__ partial) {
  if (sc[CG_DONE] != 0.0) return;
  double s = 0.0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    bool inp;
    flat_cell(g, t, inp);
    if (inp) {
      const double rv = r[t], z = -rv * dinv[t];
      L[t] = z;
      s += z * rv;
    }
  }
  flat_partial(s, partial);
}
__global__ __launch_bounds__(256) void k_cgd_dir(long long n, const double* __restrict__ sc, const double* __restrict__ L, double* __restrict__ q) {
  if (sc[CG_DONE] != 0.0) return;
  const double beta = sc[CG_BETA];
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) q[t] = L[t] + beta * q[t];
}
__global__ __launch_bounds__(256) void k_cgd_dot(GridDev g, long long n, const double* __restrict__ sc, const double* __restrict__ a,
                                                 const double* __restrict__ b, double* __restrict__ partial) {
  if (sc[CG_DONE] != 0.0) return;
  double s = 0.0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    bool inp;
    flat_cell(g, t, inp);
    if (inp) s += a[t] * b[t];
  }
  flat_partial(s, partial);
}
__global__ __launch_bounds__(256) void k_cgd_update(GridDev g, long long n, const double* __restrict__ sc, double* __restrict__ p, double* __restrict__ r,
                                                    const double* __restrict__ q, const double* __restrict__ L, double* __restrict__ partial) {
  if (sc[CG_DONE] != 0.0) return;
  const double alpha = sc[CG_ALPHA];
  double s = 0.0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
    bool inp;
    flat_cell(g, t, inp);
    p[t] += alpha * q[t];
    const double rv = r[t] - alpha * L[t];
    r[t] = rv;
    if (inp) s += rv * rv;
  }
  flat_partial(s, partial);
}

struct BoxLaunch {
  dim3 grid, block;
  int nblk;
};

inline BoxLaunch full_box(const GridDev& g) {
  BoxLaunch l;
  l.block = dim3(64, 4, 1);
  l.grid = dim3(cdiv(g.N[0], 64), cdiv(g.N[1], 4), (unsigned)g.N[2]);
  l.nblk = (int)(l.grid.x * l.grid.y * l.grid.z);
  return l;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// spectral
// ------------------------------------------------------------------------------------------------
extern "C" int ins_poisson_spectral_create(const ins_grid_t* G, ins_poisson_t** out) {
  INS_REQUIRE(G && out, "null argument");
  // assert_uniform_periodic (utils.jl:1-13)
  INS_REQUIRE(G->all_periodic, "Spectral psolver requires periodic boundary conditions.");
  INS_REQUIRE(G->uniform, "Spectral psolver requires uniform grid spacing.");
  const GridDev& g = G->g;
  for (int a = 0; a < g.D; ++a) INS_REQUIRE(g.N[a] % 2 == 0, "Spectral psolver requires even number of volumes.");
  ins_poisson* ps = new ins_poisson();
  ps->kind = POISSON_SPECTRAL;
  ps->grid = G;
  const int D = g.D;
  long long nreal = 1, ncplx = 1;
  double om = 1.0;
  for (int a = 0; a < D; ++a) {
    ps->np[a] = g.ip_hi[a] - g.ip_lo[a];
    ps->kmax[a] = a == 0 ? ps->np[a] / 2 + 1 : ps->np[a];
    nreal *= ps->np[a];
    ncplx *= ps->kmax[a];
    om *= G->h[a];
  }
  int rc = INS_OK;
  auto fail = [&](int code) {
    ins_poisson_destroy(ps);
    return code;
  };
  ps->ownfft = D == 3 && ins_ownfft_supported_mixed(ps->np);
  // 2-D power-of-two boxes: own x passes around the fused solve kernel run along y (FFT · symbol · inverse FFT in one pass):
  // three kernels instead of rocFFT's eight plus its three staging copies per solve
  const int np2[3] = {ps->np[0], 16, ps->np[1]};
  const bool own2d = D == 2 && ins_ownfft_supported(np2);
  if (own2d) ps->ownfft = true;
  if (ps->ownfft) {  // rows of phat padded to whole 128-B lines: the y / z tiles then never straddle a line (profiles/r01f_pmc_traffic.json)
    ps->kxs = ins_opt(OPT_INS_PHAT_DENSE) ? ps->kmax[0] : ((ps->kmax[0] + 7) & ~7);
    ncplx = (long long)ps->kxs * ps->kmax[1] * ps->kmax[2];
  }
  if (hipMalloc(&ps->pI, nreal * sizeof(double)) != hipSuccess || hipMalloc(&ps->phat, ncplx * sizeof(hipfftDoubleComplex)) != hipSuccess ||
      hipMemset(ps->phat, 0, ncplx * sizeof(hipfftDoubleComplex)) != hipSuccess) {
    ins_set_error("hipMalloc(pI/phat) failed for %lld cells", nreal);
    return fail(INS_ERR_HIP);
  }
  // ahat[α][k] = 4 Ω sinpi(k/Np[α])² / Δx[α]²                                      pressure.jl:305-311
  for (int a = 0; a < D; ++a) {
    std::vector<double> ah(ps->kmax[a]);
    for (int k = 0; k < ps->kmax[a]; ++k) {
      const double sn = std::sin(M_PI * ((double)k / ps->np[a]));
      ah[k] = 4 * om * sn * sn / (G->h[a] * G->h[a]);
    }
    if (ps->ownfft && D == 3 && a == 1) {  // the own y pass leaves ky in its storage order: digit-reversed (ins_fft.hip) or that of the register passes (k_line3)
      // the four-pass route (opt-in) rides on the LDS y passes; otherwise the y passes run on the register passes wherever they exist
      ps->y3 = ins_line3_supported(ps->np[1]) && ins_ownfft_yz_partitions(ps->kmax[0], ps->np[1], ps->np[2]) == 0;
      std::vector<double> perm(ah.size());
      if (ps->y3)
        ins_line3_permute_symbol(ps->np[1], ah.data(), perm.data());
      else
        ins_ownfft_permute_symbol(ps->np[1], ah.data(), perm.data());
      ah.swap(perm);
    }
    if (hipMalloc(&ps->ahat[a], ah.size() * sizeof(double)) != hipSuccess ||
        hipMemcpy(ps->ahat[a], ah.data(), ah.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
      ins_set_error("ahat upload failed");
      return fail(INS_ERR_HIP);
    }
  }
  // plan_rfft(pI): all D dims, real -> half-complex along x (pressure.jl:316).  hipFFT takes lengths slowest first.
  // plan_rfft(pI): all D dims, real -> half-complex along x (pressure.jl:316); hipFFT takes lengths slowest first.
  // 3-D with a power-of-two nz: batched 2-D (x,y) plans + ONE fused z kernel (FFT · symbol · inverse FFT) = 5 passes.
  if (own2d) {
    if ((rc = ins_zsolve_twiddles(ps->np[0], &ps->tw_x))) return fail(rc);
    if ((rc = ins_zsolve_twiddles(ps->np[1], &ps->tw))) return fail(rc);
    // the solve kernel adds ay[line / kxs] to the symbol: one "row" of lines here, contributing nothing
    if (hipMalloc(&ps->ahat[2], sizeof(double)) != hipSuccess || hipMemset(ps->ahat[2], 0, sizeof(double)) != hipSuccess) return fail(INS_ERR_HIP);
    ps->zfused = true;
  } else if (ps->ownfft) {  // no rocFFT plans at all
    if ((rc = ins_zsolve_twiddles(ps->np[0], &ps->tw_x))) return fail(rc);
    if ((rc = ins_zsolve_twiddles(ps->np[1], &ps->tw_y))) return fail(rc);
    if ((rc = ins_zsolve_twiddles(ps->np[2], &ps->tw))) return fail(rc);
    ps->zfused = true;
    // four passes instead of five where the box allows it: the z direction as tridiagonal systems carried by the two y passes (ins_fft.hip, k_yz_*)
    ps->yz_P = ins_ownfft_yz_partitions(ps->kmax[0], ps->np[1], ps->np[2]);
    if (ps->yz_P) {
      const long long nc = ins_ownfft_yz_scratch(ps->kmax[0], ps->np[1], ps->np[2], ps->yz_P);
      if (hipMalloc(&ps->yz_scratch, nc * 2 * sizeof(double)) != hipSuccess) {
        ins_set_error("hipMalloc(yz scratch) failed");
        return fail(INS_ERR_HIP);
      }
    }
  } else {
    int nfull[3] = {ps->np[2], ps->np[1], ps->np[0]};
    ps->zfused = D == 3 && ins_zsolve_supported(ps->np[2]);
    rc = ps->zfused ? ins_fft_make_real_plans(&ps->plan_fwd, &ps->plan_inv, 2, nfull + 1, ps->np[2])
                    : ins_fft_make_real_plans(&ps->plan_fwd, &ps->plan_inv, D, nfull + (3 - D), 1);
    if (rc) return fail(rc);
    ps->plans = true;
    if (ps->zfused && (rc = ins_zsolve_twiddles(ps->np[2], &ps->tw))) return fail(rc);
  }
  *out = ps;
  return INS_OK;
}

// pI -> pI through the five own passes; from_u != nullptr: the right-hand side Ω·div(u) is formed inside pass 1
static int ownfft_transform(ins_poisson* ps, const double* from_u, hipStream_t s, int src_code = 1) {
  const int n0 = ps->np[0], n1 = ps->np[1], n2 = ps->np[2], kxn = ps->kmax[0], kxs = ps->kxs;
  double* ph = reinterpret_cast<double*>(ps->phat);
  int rc;
  if (ps->grid->g.D == 2) {  // x forward, fused solve along y (lines = kx, "planes" = ky), x inverse
    if (ins_ownfft_xy_supported(n0, n1))  // up to 64 x 64 volumes: the whole solve as one launch
      return ins_k_ownfft_xysolve2d(ps->grid, ps->pI, 0, ps->pI, n0, n1, ps->tw_x, ps->tw, ps->ahat[0], ps->ahat[1], s);
    if ((rc = ins_k_ownfft_xfwd(ps->grid, ps->pI, false, ph, n0, n1, 1, ps->tw_x, s, kxs))) return rc;
    if ((rc = ins_k_zsolve(ph, n1, (long long)kxs, ps->ahat[0], kxn, ps->ahat[2], ps->ahat[1], ps->tw, 1.0 / ((double)n0 * n1), true, s, kxs))) return rc;
    return ins_k_ownfft_xinv(ph, ps->pI, n0, n1, 1, ps->tw_x, s, kxs);
  }
  // small planes (16 .. 64 on both sides): the x and the y pass of a plane as one kernel each way — four launches per solve... three: xy forward, z, xy inverse
  if (!ps->y3 && !ps->yz_P && (src_code == 1 || !from_u) && ins_ownfft_xy_supported(n0, n1)) {
    if ((rc = ins_k_ownfft_xy(ps->grid, from_u ? from_u : ps->pI, from_u ? 1 : 0, ph, nullptr, n0, n1, n2, ps->tw_x, ps->tw_y, false, s, kxs))) return rc;
    const double inv_n = 1.0 / ((double)n0 * n1 * n2);
    if ((rc = ins_k_zsolve(ph, n2, (long long)kxs * n1, ps->ahat[0], kxn, ps->ahat[1], ps->ahat[2], ps->tw, inv_n, true, s, kxs))) return rc;
    return ins_k_ownfft_xy(nullptr, nullptr, 0, ph, ps->pI, n0, n1, n2, ps->tw_x, ps->tw_y, true, s, kxs);
  }
  if ((rc = ins_k_ownfft_xfwd(ps->grid, from_u ? from_u : ps->pI, from_u ? src_code : 0, ph, n0, n1, n2, ps->tw_x, s, kxs))) return rc;
  if (ps->yz_P && !ins_opt(OPT_INS_DISABLE_YZ_FUSED)) {
    const ins_grid* G = ps->grid;
    const double c = G->h[0] * G->h[1] / G->h[2];  // Ω/Δz²
    if ((rc = ins_k_ownfft_yz_solve(ph, kxn, n1, n2, kxs, ps->yz_P, ps->ahat[0], ps->ahat[1], c, -1.0 / ((double)n0 * n1), ps->tw_y, ps->yz_scratch, s))) return rc;
    return ins_k_ownfft_xinv(ph, ps->pI, n0, n1, n2, ps->tw_x, s, kxs);
  }
  if ((rc = ps->y3 ? ins_k_line3_y(ph, kxn, n1, n2, ps->tw_y, false, s, kxs) : ins_k_ownfft_y(ph, kxn, n1, n2, ps->tw_y, false, s, kxs))) return rc;
  const double inv_n = 1.0 / ((double)n0 * n1 * n2);
  if ((rc = ins_k_zsolve(ph, n2, (long long)kxs * n1, ps->ahat[0], kxn, ps->ahat[1], ps->ahat[2], ps->tw, inv_n, true, s, kxs))) return rc;
  if ((rc = ps->y3 ? ins_k_line3_y(ph, kxn, n1, n2, ps->tw_y, true, s, kxs) : ins_k_ownfft_y(ph, kxn, n1, n2, ps->tw_y, true, s, kxs))) return rc;
  return ins_k_ownfft_xinv(ph, ps->pI, n0, n1, n2, ps->tw_x, s, kxs);
}

static int spectral_transform(ins_poisson* ps, hipStream_t s) {
  const GridDev& g = ps->grid->g;
  if (ps->ownfft) return ownfft_transform(ps, nullptr, s);
  INS_FFT_TRY(hipfftSetStream(ps->plan_fwd, s));
  INS_FFT_TRY(hipfftSetStream(ps->plan_inv, s));
  INS_FFT_TRY(hipfftExecD2Z(ps->plan_fwd, ps->pI, ps->phat));
  double inv_n = 1.0;
  for (int a = 0; a < g.D; ++a) inv_n /= ps->np[a];
  if (ps->zfused) {
    int rc = ins_k_zsolve(reinterpret_cast<double*>(ps->phat), ps->np[2], (long long)ps->kmax[0] * ps->kmax[1], ps->ahat[0], ps->kmax[0],
                          ps->ahat[1], ps->ahat[2], ps->tw, inv_n, true, s);
    if (rc) return rc;
    INS_FFT_TRY(hipfftExecZ2D(ps->plan_inv, ps->phat, ps->pI));
    return INS_OK;
  }
  dim3 block(64, 4, 1), grid(cdiv(ps->kmax[0], 64), cdiv(ps->kmax[1], 4), g.D == 3 ? ps->kmax[2] : 1);
  if (g.D == 2)
    hipLaunchKernelGGL(k_symbol<2>, grid, block, 0, s, ps->phat, ps->ahat[0], ps->ahat[1], (const double*)nullptr, ps->kmax[0],
                       ps->kmax[1], inv_n);
  else
    hipLaunchKernelGGL(k_symbol<3>, grid, block, 0, s, ps->phat, ps->ahat[0], ps->ahat[1], ps->ahat[2], ps->kmax[0], ps->kmax[1],
                       inv_n);
  INS_LAUNCH_CHECK();
  INS_FFT_TRY(hipfftExecZ2D(ps->plan_inv, ps->phat, ps->pI));
  return INS_OK;
}

static int spectral_solve(ins_poisson* ps, double* p, hipStream_t s) {
  const GridDev& g = ps->grid->g;
  dim3 block(64, 4, 1), grid(cdiv(ps->np[0], 64), cdiv(ps->np[1], 4), g.D == 3 ? ps->np[2] : 1);
  if (g.D == 2)
    hipLaunchKernelGGL((k_pack<2, true>), grid, block, 0, s, g, p, ps->pI, ps->np[0], ps->np[1]);
  else
    hipLaunchKernelGGL((k_pack<3, true>), grid, block, 0, s, g, p, ps->pI, ps->np[0], ps->np[1]);
  INS_LAUNCH_CHECK();
  int rc = spectral_transform(ps, s);
  if (rc) return rc;
  if (g.D == 2)
    hipLaunchKernelGGL((k_pack<2, false>), grid, block, 0, s, g, p, ps->pI, ps->np[0], ps->np[1]);
  else
    hipLaunchKernelGGL((k_pack<3, false>), grid, block, 0, s, g, p, ps->pI, ps->np[0], ps->np[1]);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// ------------------------------------------------------------------------------------------------
// CG
// ------------------------------------------------------------------------------------------------
extern "C" int ins_poisson_cg_create(const ins_grid_t* G, double abstol, double reltol, int64_t maxiter, ins_poisson_t** out) {
  INS_REQUIRE(G && out, "null argument");
  const GridDev& g = G->g;
  for (int a = 0; a < 2 && a < g.D; ++a)
    INS_REQUIRE(g.bc[a][0] != INS_BC_HALO && g.bc[a][1] != INS_BC_HALO, "slabs are cut along z only");
  ins_poisson* ps = new ins_poisson();
  ps->kind = POISSON_CG;
  ps->grid = G;
  ps->abstol = abstol;
  ps->reltol = reltol;
  long long ndof = 1;
  for (int a = 0; a < g.D; ++a) ndof *= g.ip_hi[a] - g.ip_lo[a];
  ps->maxiter = maxiter > 0 ? maxiter : ndof;
  ps->ndof = ndof;
  for (int a = 0; a < g.D; ++a)
    if (g.bc[a][0] == INS_BC_PRESSURE || g.bc[a][1] == INS_BC_PRESSURE) ps->singular = false;
  const size_t bytes = G->ncell * sizeof(double);
  if (hipMalloc(&ps->r, bytes) != hipSuccess || hipMalloc(&ps->L, bytes) != hipSuccess || hipMalloc(&ps->q, bytes) != hipSuccess ||
      hipMalloc(&ps->dinv, bytes) != hipSuccess) {
    ins_set_error("hipMalloc(cg scratch) failed");
    ins_poisson_destroy(ps);
    return INS_ERR_HIP;
  }
  BoxLaunch l = full_box(g);
  if (g.D == 2)
    hipLaunchKernelGGL(k_cg_diag<2>, l.grid, l.block, 0, 0, g, ps->dinv);
  else
    hipLaunchKernelGGL(k_cg_diag<3>, l.grid, l.block, 0, 0, g, ps->dinv);
  if (hipDeviceSynchronize() != hipSuccess) {
    ins_set_error("cg diag kernel failed");
    ins_poisson_destroy(ps);
    return INS_ERR_HIP;
  }
  *out = ps;
  return INS_OK;
}

// Finish a per-block partial sum on the host.  Blocking.
static int finish_sum(const ins_grid* G, double* partial_dev, int nblk, hipStream_t s, double* out) {
  std::vector<double> h(nblk);
  INS_HIP_TRY(hipMemcpyAsync(h.data(), partial_dev, nblk * sizeof(double), hipMemcpyDeviceToHost, s));
  INS_HIP_TRY(hipStreamSynchronize(s));
  double v = 0.0;
  for (int i = 0; i < nblk; ++i) v += h[i];
  *out = v;
  return INS_OK;
}


// Device-resident scalars (default); INS_CG_HOSTSYNC=1 keeps the reference's three host reads per iteration (cg_solve_hostsync below).
static int cg_solve_device(ins_poisson* ps, double* p, hipStream_t s) {
  const ins_grid* G = ps->grid;
  const GridDev& g = G->g;
  const long long n = G->ncell;
  if (!ps->cg_scal) {
    INS_HIP_TRY(hipMalloc(&ps->cg_scal, (CG_NSCAL + CG_NPART) * sizeof(double)));
    INS_HIP_TRY(hipHostMalloc(&ps->cg_host, CG_NSCAL * sizeof(double)));
  }
  double* sc = ps->cg_scal;
  double* partial = sc + CG_NSCAL;
  const unsigned nb = (unsigned)std::min<long long>((n + 255) / 256, CG_NPART);
  int rc;
  const double maxit = (double)ps->maxiter;
  auto fold = [&](int stage) -> int {
    hipLaunchKernelGGL(k_cgd_fold, dim3(1), dim3(1024), 0, s, partial, (int)nb, sc);
    if (ps->comm) {
      int r2 = ins_comm_allreduce_internal(ps->comm, sc + CG_SUM, 1, 0, s);
      if (r2) return r2;
    }
    hipLaunchKernelGGL(k_cgd_scalars, dim3(1), dim3(1), 0, s, sc, stage, ps->reltol, ps->abstol, maxit);
    INS_LAUNCH_CHECK();
    return INS_OK;
  };
  const bool bordered = ps->bordered && ps->singular;
  if (bordered) {
    INS_REQUIRE(!ps->comm, "bordered CG on slabs is not implemented");
    double sum;
    if ((rc = ins_k_reduce(G, 3, p, nullptr, g.ip_lo, g.ip_hi, &sum, s))) return rc;
    if (g.D == 2)
      hipLaunchKernelGGL(k_shift<2>, full_box(g).grid, full_box(g).block, 0, s, g, p, sum / (double)ps->ndof);
    else
      hipLaunchKernelGGL(k_shift<3>, full_box(g).grid, full_box(g).block, 0, s, g, p, sum / (double)ps->ndof);
  }
  hipLaunchKernelGGL(k_cgd_init, dim3(nb), dim3(256), 0, s, g, n, p, ps->r, ps->q, ps->L, partial);
  if ((rc = fold(0))) return rc;
  const long long batch = ins_opt(OPT_INS_CG_BATCH) > 0 ? ins_opt(OPT_INS_CG_BATCH) : 32;
  long long issued = 0;
  while (true) {
    INS_HIP_TRY(hipMemcpyAsync(ps->cg_host, sc, CG_NSCAL * sizeof(double), hipMemcpyDeviceToHost, s));
    INS_HIP_TRY(hipStreamSynchronize(s));
    if (ps->cg_host[CG_DONE] != 0.0 || issued >= ps->maxiter) break;
    for (long long b = 0; b < batch && issued < ps->maxiter; ++b, ++issued) {
      hipLaunchKernelGGL(k_cgd_precond, dim3(nb), dim3(256), 0, s, g, n, sc, ps->r, ps->dinv, ps->L, partial);
      if ((rc = fold(1))) return rc;
      hipLaunchKernelGGL(k_cgd_dir, dim3(nb), dim3(256), 0, s, n, sc, ps->L, ps->q);
      if ((rc = ins_k_apply_bc_p(G, ps->q, s))) return rc;
      if (ps->comm && (rc = ins_comm_halo_scalar_internal(ps->comm, G, ps->q, s))) return rc;
      if ((rc = ins_k_laplacian(G, ps->q, ps->L, s))) return rc;
      hipLaunchKernelGGL(k_cgd_dot, dim3(nb), dim3(256), 0, s, g, n, sc, ps->q, ps->L, partial);
      if ((rc = fold(2))) return rc;
      hipLaunchKernelGGL(k_cgd_update, dim3(nb), dim3(256), 0, s, g, n, sc, p, ps->r, ps->q, ps->L, partial);
      if ((rc = fold(3))) return rc;
    }
  }
  if (bordered) {
    double sum;
    if ((rc = ins_k_reduce(G, 3, p, nullptr, g.ip_lo, g.ip_hi, &sum, s))) return rc;
    if (g.D == 2)
      hipLaunchKernelGGL(k_shift<2>, full_box(g).grid, full_box(g).block, 0, s, g, p, sum / (double)ps->ndof);
    else
      hipLaunchKernelGGL(k_shift<3>, full_box(g).grid, full_box(g).block, 0, s, g, p, sum / (double)ps->ndof);
    INS_LAUNCH_CHECK();
  }
  ps->last_iter = (long long)ps->cg_host[CG_ITERS];
  ps->last_res = ps->cg_host[CG_RES];
  return INS_OK;
}

static int cg_solve(ins_poisson* ps, double* p, hipStream_t s) {
  if (!ins_opt(OPT_INS_CG_HOSTSYNC)) return cg_solve_device(ps, p, s);
  INS_REQUIRE(!ps->comm, "the host-synchronising CG has no slab form");
  const ins_grid* G = ps->grid;
  const GridDev& g = G->g;
  BoxLaunch l = full_box(g);
  double* partial = nullptr;
  INS_HIP_TRY(hipMalloc(&partial, l.nblk * sizeof(double)));
  struct Guard {
    double* p;
    ~Guard() { (void)hipFree(p); }
  } guard{partial};
  int rc;
  double ss;
#define LAUNCH_D(K, ...)                                                        \
  do {                                                                          \
    if (g.D == 2)                                                               \
      hipLaunchKernelGGL(K<2>, l.grid, l.block, 0, s, __VA_ARGS__);             \
    else                                                                        \
      hipLaunchKernelGGL(K<3>, l.grid, l.block, 0, s, __VA_ARGS__);             \
    INS_LAUNCH_CHECK();                                                         \
  } while (0)

  const bool bordered = ps->bordered && ps->singular;
  if (bordered) {
    double sum;
    if ((rc = ins_k_reduce(G, 3, p, nullptr, g.ip_lo, g.ip_hi, &sum, s))) return rc;
    LAUNCH_D(k_shift, g, p, sum / (double)ps->ndof);
  }
  LAUNCH_D(k_cg_init, g, p, ps->r, ps->q, ps->L, partial);
  if ((rc = finish_sum(G, partial, l.nblk, s, &ss))) return rc;
  double residual = std::sqrt(ss);
  const double tolerance = std::fmax(ps->reltol * residual, ps->abstol);
  double rho_prev = 1.0;
  long long it = 0;
  while (it < ps->maxiter && residual > tolerance) {
    double rho, qL;
    LAUNCH_D(k_cg_precond, g, ps->r, ps->dinv, ps->L, partial);
    if ((rc = finish_sum(G, partial, l.nblk, s, &rho))) return rc;
    const double beta = rho / rho_prev;
    hipLaunchKernelGGL(k_cg_dir, dim3(std::min<long long>((G->ncell + 255) / 256, 4096)), dim3(256), 0, s, G->ncell, beta, ps->L, ps->q);
    INS_LAUNCH_CHECK();
    if ((rc = ins_k_apply_bc_p(G, ps->q, s))) return rc;
    if ((rc = ins_k_laplacian(G, ps->q, ps->L, s))) return rc;
    LAUNCH_D(k_cg_dot, g, ps->q, ps->L, partial);
    if ((rc = finish_sum(G, partial, l.nblk, s, &qL))) return rc;
    const double alpha = rho / qL;
    LAUNCH_D(k_cg_update, g, alpha, p, ps->r, ps->q, ps->L, partial);
    if ((rc = finish_sum(G, partial, l.nblk, s, &ss))) return rc;
    rho_prev = rho;
    residual = std::sqrt(ss);
    ++it;
  }
  if (bordered) {
    double sum;
    if ((rc = ins_k_reduce(G, 3, p, nullptr, g.ip_lo, g.ip_hi, &sum, s))) return rc;
    LAUNCH_D(k_shift, g, p, sum / (double)ps->ndof);
  }
#undef LAUNCH_D
  ps->last_iter = it;
  ps->last_res = residual;
  return INS_OK;
}

// ------------------------------------------------------------------------------------------------
// z-slab decomposition: the solver's dots and norm are summed over the ranks of `comm` (one scalar all-reduce each) and the ghost planes of the
// search direction travel before every laplacian! ("Config 5 (CG): halo + 3 scalar all-reduces per iteration").
// maxiter and the Jacobi preconditioner are local; the convergence test uses the global residual, so all ranks stop together.
extern "C" int ins_poisson_cg_set_comm(ins_poisson_t* ps, ins_comm_t* comm) {
  INS_REQUIRE(ps, "null argument");
  INS_REQUIRE(ps->kind == POISSON_CG, "ins_poisson_cg_set_comm applies to the CG solver only");
  ps->comm = comm;
  return INS_OK;
}

extern "C" int ins_poisson_cg_bordered(ins_poisson_t* ps, int enable) {
  INS_REQUIRE(ps, "null argument");
  INS_REQUIRE(ps->kind == POISSON_CG, "bordered mode applies to the CG solver only");
  ps->bordered = enable != 0;
  return INS_OK;
}

extern "C" int ins_poisson_fft_engine(const ins_poisson_t* ps, int32_t* engine) {
  if (!ps || !engine) {
    ins_set_error("ins_poisson_fft_engine: null argument");
    return INS_ERR_INVALID;
  }
  *engine = ps->kind != POISSON_SPECTRAL ? -1 : (ps->ownfft ? 1 : (ps->zfused ? 2 : 0));
  return INS_OK;
}

extern "C" int ins_poisson_yz_partitions(const ins_poisson_t* ps, int32_t* partitions) {
  if (!ps || !partitions) {
    ins_set_error("ins_poisson_yz_partitions: null argument");
    return INS_ERR_INVALID;
  }
  *partitions = ps->kind == POISSON_SPECTRAL ? ps->yz_P : 0;
  return INS_OK;
}

extern "C" int ins_poisson_destroy(ins_poisson_t* ps) {
  if (!ps) return INS_OK;
  if (ps->plans) {
    (void)hipfftDestroy(ps->plan_fwd);
    (void)hipfftDestroy(ps->plan_inv);
    ins_fft_solver_released();
  }
  if (ps->tw) (void)hipFree(ps->tw);
  if (ps->tw_x) (void)hipFree(ps->tw_x);
  if (ps->tw_y) (void)hipFree(ps->tw_y);
  if (ps->yz_scratch) (void)hipFree(ps->yz_scratch);
  if (ps->pI) (void)hipFree(ps->pI);
  if (ps->phat) (void)hipFree(ps->phat);
  for (int a = 0; a < 3; ++a)
    if (ps->ahat[a]) (void)hipFree(ps->ahat[a]);
  if (ps->r) (void)hipFree(ps->r);
  if (ps->cg_scal) (void)hipFree(ps->cg_scal);
  if (ps->cg_host) (void)hipHostFree(ps->cg_host);
  if (ps->L) (void)hipFree(ps->L);
  if (ps->q) (void)hipFree(ps->q);
  if (ps->dinv) (void)hipFree(ps->dinv);
  if (ps->fdm) (void)ins_fdm_destroy(ps->fdm);
  delete ps;
  return INS_OK;
}

// psolver_direct: pack Ip -> FDM solve (ins_fdm.hip) -> unpack                     pressure.jl:101-154
static int fdm_solve(ins_poisson* ps, double* p, hipStream_t s) {
  const GridDev& g = ps->grid->g;
  double* buf = ins_fdm_buffer(ps->fdm);
  dim3 block(64, 4, 1), grid(cdiv(ps->np[0], 64), cdiv(ps->np[1], 4), g.D == 3 ? ps->np[2] : 1);
  if (g.D == 2)
    hipLaunchKernelGGL((k_pack<2, true>), grid, block, 0, s, g, p, buf, ps->np[0], ps->np[1]);
  else
    hipLaunchKernelGGL((k_pack<3, true>), grid, block, 0, s, g, p, buf, ps->np[0], ps->np[1]);
  INS_LAUNCH_CHECK();
  int rc = ins_fdm_solve(ps->fdm, s);
  if (rc) return rc;
  const double* shift = ins_fdm_mean(ps->fdm);  // e'p = 0 of the bordered system, applied while unpacking
  if (g.D == 2)
    hipLaunchKernelGGL((k_pack<2, false>), grid, block, 0, s, g, p, buf, ps->np[0], ps->np[1], shift);
  else
    hipLaunchKernelGGL((k_pack<3, false>), grid, block, 0, s, g, p, buf, ps->np[0], ps->np[1], shift);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

extern "C" int ins_poisson_fdm_create(const ins_grid_t* G, const double* const* V, const double* const* lam, ins_poisson_t** out) {
  INS_REQUIRE(G && V && lam && out, "null argument");
  const GridDev& g = G->g;
  int n[3] = {1, 1, 1};
  bool singular = true;
  for (int a = 0; a < g.D; ++a) {
    INS_REQUIRE(g.bc[a][0] != INS_BC_HALO && g.bc[a][1] != INS_BC_HALO, "psolver_direct on a slab grid is not implemented");
    INS_REQUIRE(V[a] && lam[a], "null eigen-decomposition");
    n[a] = g.ip_hi[a] - g.ip_lo[a];
    if (g.bc[a][0] == INS_BC_PRESSURE || g.bc[a][1] == INS_BC_PRESSURE) singular = false;
  }
  ins_poisson* ps = new ins_poisson();
  ps->kind = POISSON_FDM;
  ps->grid = G;
  ps->singular = singular;
  ps->ndof = (long long)n[0] * n[1] * n[2];
  for (int a = 0; a < 3; ++a) ps->np[a] = n[a];
  const double* Vp[3] = {V[0], V[1], g.D == 3 ? V[2] : nullptr};
  const double* lp[3] = {lam[0], lam[1], g.D == 3 ? lam[2] : nullptr};
  int rc = ins_fdm_create(g.D, n, Vp, lp, singular ? 1 : 0, &ps->fdm);
  if (rc) {
    delete ps;
    return rc;
  }
  if (g.D == 3 && g.bc[2][0] == INS_BC_PERIODIC && g.bc[2][1] == INS_BC_PERIODIC) {  // periodic z: Fourier modes if the spacing is constant
    const ins_grid_desc_t& d = G->desc;
    bool uni = true;
    const double hz = d.dx[2][1];
    const double ztol = 4.0 * d.N[2] * 2.220446049250313e-16 * hz;
    for (int k = 0; k < d.N[2]; ++k) uni = uni && std::fabs(d.dx[2][k] - hz) <= ztol && (k == d.N[2] - 1 || std::fabs(d.dxu[2][k] - hz) <= ztol);  // (the last Δu is Δ/2 by construction, grid.jl:196)
    if (uni && (rc = ins_fdm_enable_zfft(ps->fdm, hz, lam[2]))) {
      ins_poisson_destroy(ps);
      return rc;
    }
    if (uni && g.bc[0][0] == INS_BC_PERIODIC && g.bc[0][1] == INS_BC_PERIODIC) {  // periodic x too (channel flows)
      const double hx = d.dx[0][1];
      const double xtol = 4.0 * d.N[0] * 2.220446049250313e-16 * hx;
      bool unix_ = true;
      for (int k = 0; k < d.N[0]; ++k) unix_ = unix_ && std::fabs(d.dx[0][k] - hx) <= xtol && (k == d.N[0] - 1 || std::fabs(d.dxu[0][k] - hx) <= xtol);
      if (unix_ && (rc = ins_fdm_enable_xfft(ps->fdm, hx, lam[0]))) {
        ins_poisson_destroy(ps);
        return rc;
      }
    }
  }
  if (g.D == 3 && g.bc[0][0] == INS_BC_PERIODIC && g.bc[0][1] == INS_BC_PERIODIC && g.bc[1][0] == INS_BC_PERIODIC && g.bc[1][1] == INS_BC_PERIODIC &&
      !(g.bc[2][0] == INS_BC_PERIODIC && g.bc[2][1] == INS_BC_PERIODIC)) {  // periodic x and y, walls / open sides in z
    const ins_grid_desc_t& d = G->desc;
    bool uni = true;
    double hh[2];
    for (int a = 0; a < 2; ++a) {
      hh[a] = d.dx[a][1];
      const double tol = 4.0 * d.N[a] * 2.220446049250313e-16 * hh[a];
      for (int k = 0; k < d.N[a]; ++k) uni = uni && std::fabs(d.dx[a][k] - hh[a]) <= tol && (k == d.N[a] - 1 || std::fabs(d.dxu[a][k] - hh[a]) <= tol);
    }
    if (uni && (rc = ins_fdm_enable_xyfft(ps->fdm, hh[0], hh[1], lam[0], lam[1]))) {
      ins_poisson_destroy(ps);
      return rc;
    }
  }
  if (g.D == 3 && g.bc[2][0] != INS_BC_PERIODIC) {  // uniform z between walls: cosine modes (the enable call checks the eigenvalues it was given)
    const ins_grid_desc_t& d = G->desc;
    const double hz = d.dx[2][1];
    const double ztol = 4.0 * d.N[2] * 2.220446049250313e-16 * hz;
    bool uni = true;
    for (int k = 1; k < d.N[2] - 1; ++k) uni = uni && std::fabs(d.dx[2][k] - hz) <= ztol;
    if (uni && (rc = ins_fdm_enable_zdct(ps->fdm, hz, lam[2]))) {
      ins_poisson_destroy(ps);
      return rc;
    }
  }
  *out = ps;
  return INS_OK;
}

int ins_k_poisson_solve(ins_poisson* ps, double* p, hipStream_t s) {
  switch (ps->kind) {
    case POISSON_SPECTRAL: return spectral_solve(ps, p, s);
    case POISSON_FDM: return fdm_solve(ps, p, s);
    default: return cg_solve(ps, p, s);
  }
}

extern "C" int ins_poisson_solve_f64(ins_poisson_t* ps, double* p, void* stream) {
  INS_REQUIRE(ps && p, "null argument");
  return ins_k_poisson_solve(ps, p, as_stream(stream));
}

// experiment / test hook (not in the public header): which directions of a direct solver run as folded half-size GEMMs (bit a = direction a)
int ins_fdm_modes(const ins_fdm* F);
// test hook: which directions of the direct solver run in trigonometric modes (1 Fourier z, 2 Fourier x, 4 Fourier x and y, 8 cosine z)
extern "C" int ins_dbg_fdm_modes(const ins_poisson_t* ps) { return (ps && ps->kind == POISSON_FDM) ? ins_fdm_modes(ps->fdm) : -1; }
extern "C" int ins_dbg_fdm_fold_mask(const ins_poisson_t* ps) { return (ps && ps->kind == POISSON_FDM) ? ins_fdm_fold_mask(ps->fdm) : -1; }

extern "C" int ins_poisson_last_info(const ins_poisson_t* ps, int64_t* iterations, double* residual) {
  INS_REQUIRE(ps, "null argument");
  if (iterations) *iterations = ps->last_iter;
  if (residual) *residual = ps->last_res;
  return INS_OK;
}

// project!(u, setup; psolver, p)                                                   pressure.jl:69-82
// project! without its last statement: p <- solution of L p = Ω div(u) (padded, ghost pressures per apply_bc_p!), u untouched.
// Direct solver inside its fused project form only (the caller checks ins_k_project_fdm_fused).
bool ins_k_project_fdm_fused(const ins_poisson* ps) { return ps->kind == POISSON_FDM && !ins_opt(OPT_INS_DISABLE_FDM_FUSED); }
// Ω·div(u) into the direct solver's buffer; fm != 0: in the folded form of its symmetric directions (no separate fold pass)
static int fdm_rhs(const ins_grid* G, ins_poisson* ps, const double* u, double* buf, int fm, hipStream_t s) {
  const GridDev& g = G->g;
  const dim3 block(64, 4, 1);
  if (fm) {
    const int n0 = ps->np[0], n1 = ps->np[1], n2 = g.D == 3 ? ps->np[2] : 1;
    const dim3 grid(cdiv((fm & 1) ? n0 / 2 : n0, 64), cdiv((fm & 2) ? n1 / 2 : n1, 4), (g.D == 3 && (fm & 4)) ? n2 / 2 : n2);
#define INS_DIVFOLD(DD, MM) \
  case MM: hipLaunchKernelGGL((k_div_to_pI_fold<DD, MM>), grid, block, 0, s, g, u, buf, n0, n1, n2); break;
    if (g.D == 2) {
      switch (fm & 3) {
        INS_DIVFOLD(2, 1)
        INS_DIVFOLD(2, 2)
        INS_DIVFOLD(2, 3)
      }
    } else {
      switch (fm & 7) {
        INS_DIVFOLD(3, 1)
        INS_DIVFOLD(3, 2)
        INS_DIVFOLD(3, 3)
        INS_DIVFOLD(3, 4)
        INS_DIVFOLD(3, 5)
        INS_DIVFOLD(3, 6)
        INS_DIVFOLD(3, 7)
      }
    }
#undef INS_DIVFOLD
  } else {
    const dim3 grid(cdiv(ps->np[0], 64), cdiv(ps->np[1], 4), g.D == 3 ? ps->np[2] : 1);
    if (g.D == 2)
      hipLaunchKernelGGL(k_div_to_pI<2>, grid, block, 0, s, g, u, buf, ps->np[0], ps->np[1]);
    else
      hipLaunchKernelGGL(k_div_to_pI<3>, grid, block, 0, s, g, u, buf, ps->np[0], ps->np[1]);
  }
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// copy-back (- mean) + apply_bc_p! (+ applypressure! when u != nullptr) from the direct solver's buffer, folded (fm) or not
static int fdm_unpack(const ins_grid* G, ins_poisson* ps, double* u, double* p, const double* buf, int fm, hipStream_t s) {
  const GridDev& g = G->g;
  const dim3 block(64, 4, 1);
  const int n0 = ps->np[0], n1 = ps->np[1], n2 = g.D == 3 ? ps->np[2] : 1;
  bool pressure_side = false;  // a PressureBC puts a DOF of u into the ghost layer: the general kernel covers it
  for (int a = 0; a < g.D; ++a) pressure_side = pressure_side || g.bc[a][0] == INS_BC_PRESSURE || g.bc[a][1] == INS_BC_PRESSURE;
  if (g.D == 3 && fm && !pressure_side && !((fm & 1) && (n0 & 1)) && !((fm & 2) && (n1 & 1)) && !((fm & 4) && (n2 & 1)) && !ins_opt(OPT_INS_DISABLE_FDM_UNFOLD4)) {
    const dim3 grid(cdiv((fm & 1) ? n0 / 2 : n0, 64), cdiv((fm & 2) ? n1 / 2 : n1, 4), (fm & 4) ? n2 / 2 : n2);
    if (u)
      hipLaunchKernelGGL(k_unfold_grad3<true>, grid, block, 0, s, g, u, p, buf, n0, n1, n2, ins_fdm_mean(ps->fdm), fm);
    else
      hipLaunchKernelGGL(k_unfold_grad3<false>, grid, block, 0, s, g, u, p, buf, n0, n1, n2, ins_fdm_mean(ps->fdm), fm);
    INS_LAUNCH_CHECK();
    return INS_OK;
  }
  const dim3 gridp(cdiv(n0 + 2, 64), cdiv(n1 + 2, 4), g.D == 3 ? n2 + 2 : 1);
  if (g.D == 2) {
    if (u)
      hipLaunchKernelGGL((k_unpack_grad_bc<2, true>), gridp, block, 0, s, g, u, p, buf, n0, n1, ins_fdm_mean(ps->fdm), fm, 1);
    else
      hipLaunchKernelGGL((k_unpack_grad_bc<2, false>), gridp, block, 0, s, g, u, p, buf, n0, n1, ins_fdm_mean(ps->fdm), fm, 1);
  } else {
    if (u)
      hipLaunchKernelGGL((k_unpack_grad_bc<3, true>), gridp, block, 0, s, g, u, p, buf, n0, n1, ins_fdm_mean(ps->fdm), fm, n2);
    else
      hipLaunchKernelGGL((k_unpack_grad_bc<3, false>), gridp, block, 0, s, g, u, p, buf, n0, n1, ins_fdm_mean(ps->fdm), fm, n2);
  }
  INS_LAUNCH_CHECK();
  return INS_OK;
}

int ins_k_project_fdm_solve_only(const ins_grid* G, ins_poisson* ps, const double* u, double* p, hipStream_t s) {
  int rc, fm = 0;
  double* buf = ins_fdm_buffer(ps->fdm);
  if (ins_fdm_takes_u(ps->fdm)) {
    if ((rc = ins_fdm_solve(ps->fdm, s, G, u))) return rc;
  } else {
    fm = ins_opt(OPT_INS_DISABLE_FDM_FOLDFUSE) ? 0 : ins_fdm_fold_mask(ps->fdm);
    if ((rc = fdm_rhs(G, ps, u, buf, fm, s))) return rc;
    if ((rc = ins_fdm_solve(ps->fdm, s, nullptr, nullptr, fm != 0))) return rc;
  }
  return fdm_unpack(G, ps, nullptr, p, buf, fm, s);
}

int ins_k_project(const ins_grid* G, ins_poisson* ps, double* u, double* p, hipStream_t s) {
  const GridDev& g = G->g;
  int rc;
  if (ps->kind == POISSON_SPECTRAL) {
    dim3 block(64, 4, 1), grid(cdiv(ps->np[0], 64), cdiv(ps->np[1], 4), g.D == 3 ? ps->np[2] : 1);
    if (g.D == 2)
      hipLaunchKernelGGL(k_div_to_pI<2>, grid, block, 0, s, g, u, ps->pI, ps->np[0], ps->np[1]);
    else
      hipLaunchKernelGGL(k_div_to_pI<3>, grid, block, 0, s, g, u, ps->pI, ps->np[0], ps->np[1]);
    INS_LAUNCH_CHECK();
    if ((rc = spectral_transform(ps, s))) return rc;
    dim3 gridp(cdiv(g.N[0], 64), cdiv(g.N[1], 4), (unsigned)g.N[2]);
    if (g.D == 2)
      hipLaunchKernelGGL(k_grad_from_pI<2>, gridp, block, 0, s, g, u, p, ps->pI, ps->np[0], ps->np[1], 1);
    else
      hipLaunchKernelGGL(k_grad_from_pI<3>, gridp, block, 0, s, g, u, p, ps->pI, ps->np[0], ps->np[1], ps->np[2]);
    INS_LAUNCH_CHECK();
    return INS_OK;
  }
  if (ps->kind == POISSON_FDM && !ins_opt(OPT_INS_DISABLE_FDM_FUSED)) {
    // direct solver: Ω·div(u) straight into the solver's buffer, and copy-back - mean + apply_bc_p! + applypressure! in one pass
    double* buf = ins_fdm_buffer(ps->fdm);
    int fm = 0;
    if (ins_fdm_takes_u(ps->fdm)) {  // periodic x: the divergence is formed inside the solver's x pass
      if ((rc = ins_fdm_solve(ps->fdm, s, G, u))) return rc;
    } else {
      fm = ins_opt(OPT_INS_DISABLE_FDM_FOLDFUSE) ? 0 : ins_fdm_fold_mask(ps->fdm);
      if ((rc = fdm_rhs(G, ps, u, buf, fm, s))) return rc;
      if ((rc = ins_fdm_solve(ps->fdm, s, nullptr, nullptr, fm != 0))) return rc;
    }
    return fdm_unpack(G, ps, u, p, buf, fm, s);
  }
  if ((rc = ins_k_divergence(G, u, p, s))) return rc;
  if ((rc = ins_k_scalewithvolume(G, p, s))) return rc;
  if ((rc = ins_k_poisson_solve(ps, p, s))) return rc;
  if ((rc = ins_k_apply_bc_p(G, p, s))) return rc;
  return ins_k_applypressure(G, u, p, s);
}

// project! for the fused periodic RK stage: u holds valid INTERIOR values only; on return its interior is
// divergence-free and its ghost volumes are filled.  3-D, all-periodic, spectral solver.
// uout != nullptr: u stays as it is (uncorrected) and the corrected field with its ghost volumes goes to uout.
int ins_k_project_periodic_fused(const ins_grid* G, ins_poisson* ps, double* u, double* p, bool keep_p, hipStream_t s, double* uout) {
  const GridDev& g = G->g;
  dim3 block(64, 4, 1), grid(cdiv(ps->np[0], 64), cdiv(ps->np[1], 4), ps->np[2]);
  int rc;
  double* dst = uout ? uout : u;
  if (ps->ownfft) {  // K2 lives inside the x-forward pass
    rc = ownfft_transform(ps, u, s);
  } else {
    hipLaunchKernelGGL((k_div_to_pI<3, true>), grid, block, 0, s, g, u, ps->pI, ps->np[0], ps->np[1]);
    INS_LAUNCH_CHECK();
    rc = spectral_transform(ps, s);
  }
  if (rc) return rc;
  if (keep_p)
    hipLaunchKernelGGL(k_grad_ghost3<true>, grid, block, 0, s, g, dst, p, ps->pI, ps->np[0], ps->np[1], ps->np[2], (const double*)u);
  else
    hipLaunchKernelGGL(k_grad_ghost3<false>, grid, block, 0, s, g, dst, p, ps->pI, ps->np[0], ps->np[1], ps->np[2], (const double*)u);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// The same for 2-D (own x passes + the fused solve kernel along y): x forward with the divergence formed inside, solve, x inverse,
// gradient-subtract with the periodic ghost images.  Four launches per projection.
int ins_k_project_periodic_fused_2d(const ins_grid* G, ins_poisson* ps, double* u, double* p, bool keep_p, hipStream_t s) {
  const GridDev& g = G->g;
  const int n0 = ps->np[0], n1 = ps->np[1], kxn = ps->kmax[0], kxs = ps->kxs;
  double* ph = reinterpret_cast<double*>(ps->phat);
  int rc;
  if (ins_ownfft_xy_supported(n0, n1)) {
    if ((rc = ins_k_ownfft_xysolve2d(G, u, 1, ps->pI, n0, n1, ps->tw_x, ps->tw, ps->ahat[0], ps->ahat[1], s))) return rc;
  } else {
    if ((rc = ins_k_ownfft_xfwd(G, u, 3, ph, n0, n1, 1, ps->tw_x, s, kxs))) return rc;
    if ((rc = ins_k_zsolve(ph, n1, (long long)kxs, ps->ahat[0], kxn, ps->ahat[2], ps->ahat[1], ps->tw, 1.0 / ((double)n0 * n1), true, s, kxs))) return rc;
    if ((rc = ins_k_ownfft_xinv(ph, ps->pI, n0, n1, 1, ps->tw_x, s, kxs))) return rc;
  }
  dim3 block(64, 4, 1), grid(cdiv(n0, 64), cdiv(n1, 4), 1);
  if (keep_p)
    hipLaunchKernelGGL((k_grad_ghost<2, true>), grid, block, 0, s, g, u, p, ps->pI, n0, n1, 1);
  else
    hipLaunchKernelGGL((k_grad_ghost<2, false>), grid, block, 0, s, g, u, p, ps->pI, n0, n1, 1);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// first half of the 2-D fused projection only: pI <- solution of L p = Ω div(u) (u: interior volumes valid); the next stage kernel corrects in registers
int ins_k_project_periodic_solve_only_2d(const ins_grid* G, ins_poisson* ps, const double* u, hipStream_t s) {
  const int n0 = ps->np[0], n1 = ps->np[1], kxn = ps->kmax[0], kxs = ps->kxs;
  double* ph = reinterpret_cast<double*>(ps->phat);
  int rc;
  if (ins_ownfft_xy_supported(n0, n1)) return ins_k_ownfft_xysolve2d(G, u, 1, ps->pI, n0, n1, ps->tw_x, ps->tw, ps->ahat[0], ps->ahat[1], s);
  if ((rc = ins_k_ownfft_xfwd(G, u, 3, ph, n0, n1, 1, ps->tw_x, s, kxs))) return rc;
  if ((rc = ins_k_zsolve(ph, n1, (long long)kxs, ps->ahat[0], kxn, ps->ahat[2], ps->ahat[1], ps->tw, 1.0 / ((double)n0 * n1), true, s, kxs))) return rc;
  return ins_k_ownfft_xinv(ph, ps->pI, n0, n1, 1, ps->tw_x, s, kxs);
}

bool ins_poisson_own2d(const ins_poisson* ps) { return ps->kind == POISSON_SPECTRAL && ps->ownfft && ps->grid->g.D == 2; }

// First half of the fused periodic projection only: pI <- solution of L p = Ω div(u) (u: interior volumes valid).
// The gradient-subtract is left to the next stage's stencil kernel (k_momentum_flux<..., CORR>).
int ins_k_project_periodic_solve_only(const ins_grid* G, ins_poisson* ps, const double* u, hipStream_t s) {
  const GridDev& g = G->g;
  if (ps->ownfft) return ownfft_transform(ps, u, s);
  dim3 block(64, 4, 1), grid(cdiv(ps->np[0], 64), cdiv(ps->np[1], 4), ps->np[2]);
  hipLaunchKernelGGL((k_div_to_pI<3, true>), grid, block, 0, s, g, u, ps->pI, ps->np[0], ps->np[1]);
  INS_LAUNCH_CHECK();
  return spectral_transform(ps, s);
}

// The `_f32` family's pressure equation on power-of-two boxes: right-hand side Ω·div(u) from the FLOAT field u32 (periodic images), the five fp64
// passes, solution in ps->pI (double, unpadded).  false: this solver has no own passes (the caller keeps hipFFT).
bool ins_k_spectral_own3d(const ins_poisson* ps) { return ps->kind == POISSON_SPECTRAL && ps->ownfft && ps->grid->g.D == 3; }
int ins_k_spectral_solve_from_u32(ins_poisson* ps, const float* u32, hipStream_t s) {
  return ownfft_transform(ps, reinterpret_cast<const double*>(u32), s, 5);
}
const double* ins_k_spectral_pI(const ins_poisson* ps) { return ps->pI; }

// The same five passes on float2 spectra (the `_f32` family, ins_f32.hip): `ps` supplies the symbol vectors (double, ây in the digit-reversed order of
// the y pass) and the grid; the float work arrays and float2 twiddles belong to the caller.  u32 != nullptr: right-hand side Ω·div(u) from the
// float velocity field inside the x pass (periodic images); else the right-hand side is in pI32.  Solution in pI32 (unpadded).
bool ins_zsolve_f32_supported(int nz);
int ins_k_zsolve_f32(float* data, int nz, long long nl, const double* ax, int kxn, const double* ay, const double* az, const float* tw, double inv_n,
                     bool zero_mean, hipStream_t s, int kxs);
int ins_k_ownfft_xfwd_f32(const ins_grid* G, const float* src, int from_u, float* phat, int n0, int n1, int n2, const float* tw, hipStream_t s, int kxs);
int ins_k_ownfft_xinv_f32(const float* phat, float* pI, int n0, int n1, int n2, const float* tw, hipStream_t s, int kxs);
int ins_k_ownfft_y_f32(float* phat, int kxn, int n1, int n2, const float* tw, bool inverse, hipStream_t s, int kxs);
bool ins_k_spectral_own3d_f32(const ins_poisson* ps) { return ins_k_spectral_own3d(ps) && ins_zsolve_f32_supported(ps->np[2]); }
int ins_k_spectral_solve_f32(ins_poisson* ps, const float* u32, float* pI32, float* phat32, int kxs32, const float* twx, const float* twy,
                             const float* twz, hipStream_t s) {
  const int n0 = ps->np[0], n1 = ps->np[1], n2 = ps->np[2], kxn = ps->kmax[0];
  int rc;
  if ((rc = ins_k_ownfft_xfwd_f32(ps->grid, u32 ? u32 : pI32, u32 ? 1 : 0, phat32, n0, n1, n2, twx, s, kxs32))) return rc;
  if ((rc = ps->y3 ? ins_k_line3_y_f32(phat32, kxn, n1, n2, twy, false, s, kxs32) : ins_k_ownfft_y_f32(phat32, kxn, n1, n2, twy, false, s, kxs32))) return rc;
  const double inv_n = 1.0 / ((double)n0 * n1 * n2);
  if ((rc = ins_k_zsolve_f32(phat32, n2, (long long)kxs32 * n1, ps->ahat[0], kxn, ps->ahat[1], ps->ahat[2], twz, inv_n, true, s, kxs32))) return rc;
  if ((rc = ps->y3 ? ins_k_line3_y_f32(phat32, kxn, n1, n2, twy, true, s, kxs32) : ins_k_ownfft_y_f32(phat32, kxn, n1, n2, twy, true, s, kxs32))) return rc;
  return ins_k_ownfft_xinv_f32(phat32, pI32, n0, n1, n2, twx, s, kxs32);
}

extern "C" int ins_project_f64(const ins_grid_t* G, ins_poisson_t* ps, double* u, double* p, void* stream) {
  INS_REQUIRE(G && ps && u && p, "null argument");
  INS_REQUIRE(ps->grid == G, "psolver was created for a different grid");
  return ins_k_project(G, ps, u, p, as_stream(stream));
}
