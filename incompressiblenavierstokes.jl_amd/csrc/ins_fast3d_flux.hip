// K1 (flux form) — fused 3-D momentum-RHS stencil for gfx950: convection_diffusion_kernel! +
// fill!(F, 0) (operators.jl:647-690, 971) for any boundary conditions and stretched grids.
//
// The reference evaluates, per cell and per (α, β), the difference of two face fluxes
//     φ_αβ(I) = ν (u^α[I+e_β] - u^α[I]) / Δb  -  ½(u^α[I] + u^α[I+e_β]) · (A₂ u^β[I] + A₁ u^β[I+e_α]),
// f^α[I] += (φ_αβ(I) - φ_αβ(I - e_β)) / Δu_αβ, and so computes every face flux twice.  The lower-face
// expression of cell I is term-for-term the upper-face expression of cell I - e_β (same widths, same
// weights: operators.jl:668-675), so this kernel evaluates each face flux ONCE:
//   * lanes of a wavefront run along x; the x-neighbour's values and the x-flux of the left neighbour
//     move between lanes with DPP wave shifts (v_mov_b32_dpp wave_shl/wave_shr) — no LDS, no barriers;
//     lanes 0 and 63 are halo columns, lanes 1..62 produce output;
//   * each thread keeps R+2 consecutive y-rows of the three components in registers, so y-fluxes are
//     shared between its rows;
//   * the workgroup marches along z; the upper z-flux of plane k is carried in registers and becomes
//     the lower z-flux of plane k+1; plane k+2 is prefetched while plane k is computed.
// Metrics come from per-direction records (reciprocal widths, ν-scaled masked reciprocals, half
// interpolation weights): per-lane for x, scalar (SGPR) loads for y and z; on exactly-uniform grids
// the records are constants.  Algorithmic traffic: 48 B / cell (read u, write F).
#include <algorithm>
#include <cstdlib>

#include "ins_internal.h"

// (struct Rec, the per-direction metric record, lives in ins_internal.h: ins_flux64m.hip reads the same tables)

namespace {

__global__ __launch_bounds__(256) void k_build_recs(GridDev g, double visc, Rec* __restrict__ r0, Rec* __restrict__ r1,
                                                    Rec* __restrict__ r2, int diffusion_only) {
  const int d = blockIdx.y;
  Rec* out = d == 0 ? r0 : (d == 1 ? r1 : r2);
  const int n = g.N[d];
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
    const int ip = min(idx + 1, n - 1);
    Rec r;
    r.vs = visc * g.mdx[d][ip];
    r.vo = visc * g.mdxu[d][idx];
    r.a0 = 0.5 * g.A2[0][d][idx];
    r.b0 = 0.5 * g.A1[0][d][ip];
    r.a1 = 0.5 * g.A2[1][d][idx];
    r.b1 = 0.5 * g.A1[1][d][ip];
    r.a2 = 0.5 * g.A2[2][d][idx];
    r.b2 = 0.5 * g.A1[2][d][ip];
    r.rs = g.rdxu[d][idx];
    r.ro = g.rdx[d][idx];
    for (int q = 0; q < 6; ++q) r.pad[q] = 0.0;
    if (diffusion_only) r.a0 = r.b0 = r.a1 = r.b1 = r.a2 = r.b2 = 0.0;  // the convective part of every face flux is then exactly zero
    out[idx] = r;
  }
}

__device__ __forceinline__ double from_next(double v) {  // lane l receives lane l+1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_prev(double v) {  // lane l receives lane l-1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Face flux: ν(up - uc)/Δb - ½(uc + up)(A₂ ub0 + A₁ ub1)
__device__ __forceinline__ double flux(double uc, double up, double ub0, double ub1, double ha, double hb, double vd) {
  const double uba = ha * ub0 + hb * ub1;
  return (up - uc) * vd - (uc + up) * uba;
}

constexpr int XOUT = 62;  // output columns per wavefront (lanes 1..62)


template <int R>
struct Plane {
  double v[3][R + 2];
};

// XW: wavefronts of a workgroup side by side in x (1, 2 or 4); the other 4/XW stack in y.
// KMAJOR: k-major sweep — XCD e (= blockIdx & 7) owns the e-th y-range; inside an XCD tiles run x fastest,
// then y, then z-chunk, so the whole chip works inside a few-plane window (DRAM-friendly streaming) and each
// XCD's L2 re-serves its slab of the chunk-boundary planes and halo rows.
// CORR (fused periodic RK path, stages >= 2): `u` is the PREVIOUS stage's uncorrected velocity u* (interior
// volumes only, no valid ghosts) and `pI` the unpadded pressure of its projection; every plane is corrected in
// registers as it arrives, u = u* - (p[I+e_α] - p[I]) / Δu (applypressure!, operators.jl:225-233), with all
// neighbours addressed through the periodic image (apply_bc_u!, boundary_conditions.jl:276-288).  That removes
// the gradient-subtract pass (K4) of every stage but the last.  Lanes 62 and 63 are halo columns then
// (lane 63 only supplies p to lane 62), so 61 columns are produced per wavefront.
// CORR = 2: the same on a z-slab — x, y through the periodic image, z through exchanged ghost planes: `u` has valid z-ghost
// planes of u*, and `pI` is the EXTENDED pressure buffer [1 ghost plane below | nzl local planes | 2 ghost planes above].
template <int R, bool UNIFORM, bool MASKED, int XW, bool FUSE, int CORR>
__global__ __launch_bounds__(256, 2) void k_momentum_flux(GridDev g, const Rec* __restrict__ rx, const Rec* __restrict__ ry,
                                                       const Rec* __restrict__ rz, const double* __restrict__ u,
                                                       double* __restrict__ F, int zc, int ntx, int nty, int ntz, RkEpi epi,
                                                       const double* __restrict__ pI, int bar) {
  constexpr int XO = CORR ? XOUT - 1 : XOUT;
  // XCD-aware order: consecutive block ids round-robin over the 8 XCDs; give each XCD a contiguous run of
  // tiles (y fastest) so halo rows/columns shared by neighbouring tiles are hits in that XCD's L2.
  int txi, tyi, t;
  {
    const int nty_local = (nty + 7) >> 3;
    int seq = (int)(blockIdx.x >> 3);
    if (seq >= ntx * nty_local * ntz) return;
    txi = seq % ntx;
    seq /= ntx;
    tyi = (int)(blockIdx.x & 7) * nty_local + seq % nty_local;
    t = seq / nty_local;
    if (tyi >= nty) return;
  }
  const int tzi = t;

  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int wx = wave % XW, wy = wave / XW;
  const int N0 = g.N[0], N1 = g.N[1], N2 = g.N[2];
  const int i = (txi * XW + wx) * XO + lane;  // lane 0 = left halo column
  const int ic = min(i, N0 - 1);
  const int jb = 1 + (tyi * (4 / XW) + wy) * R; // first output row of this wavefront
  const int k0 = 1 + tzi * zc;
  const int k1 = min(k0 + zc, N2 - 1);          // planes [k0, k1)
  if (i - lane > N0 - 2 || jb > N1 - 2) {       // whole wavefront outside the domain: it only keeps the barrier count (one per plane)
    if (bar)
      for (int k = k0; k < k1; ++k) __builtin_amdgcn_s_barrier();
    return;
  }
  const long long sz = g.sx[2];
  const bool xout = lane >= 1 && lane <= XO && i <= N0 - 2;
  const int n0 = N0 - 2, n1 = N1 - 2, n2 = N2 - 2;
  auto wrap = [](int idx, int n) {  // padded index -> interior index in [0, n) through the periodic image
    int q = idx - 1;
    q = q < 0 ? q + n : (q >= n ? q - n : q);
    return q;
  };

  long long rowoff[R + 2];  // element offset of (row, column) inside a padded plane
  long long prow[R + 3];    // ... inside an unpadded pI plane (CORR only)
  // CORR == 3 (masked grids, Periodic / Dirichlet sides): the stencil input is the previous stage's UNCORRECTED u* with its boundary data
  // in place (apply_bc_u! has run on it) and pI is the PADDED pressure of that stage's projection.  Periodic directions are read
  // through the periodic image (their ghost volumes hold uncorrected copies), the others at their padded index; the correction is
  // applied to degrees of freedom only (applypressure!, operators.jl:225-233), so boundary data stays what apply_bc_u! wrote.
  const bool perx = CORR == 3 && g.bc[0][0] == INS_BC_PERIODIC, pery = CORR == 3 && g.bc[1][0] == INS_BC_PERIODIC,
             perz = CORR == 3 && g.bc[2][0] == INS_BC_PERIODIC;
  const int colx = CORR == 3 ? (perx ? wrap(min(i, N0), n0) + 1 : ic) : 0;
  auto rowy = [&](int j) { return pery ? wrap(min(j, N1 + 1), n1) + 1 : min(j, N1 - 1); };
  auto planez = [&](int kk) { return perz ? wrap(min(kk, N2 + 1), n2) + 1 : min(max(kk, 0), N2 - 1); };
#pragma unroll
  for (int rr = 0; rr < R + 2; ++rr) {
    if (CORR == 3)
      rowoff[rr] = (long long)rowy(jb - 1 + rr) * N0 + colx;
    else if (CORR)
      rowoff[rr] = (long long)(wrap(min(jb - 1 + rr, N1), n1) + 1) * N0 + (wrap(min(i, N0), n0) + 1);
    else
      rowoff[rr] = (long long)min(jb - 1 + rr, N1 - 1) * N0 + ic;
  }
  if (CORR == 3) {
#pragma unroll
    for (int rr = 0; rr < R + 3; ++rr) prow[rr] = (long long)rowy(jb - 1 + rr) * N0 + colx;
  } else if (CORR) {
#pragma unroll
    // one index beyond the padded range is still a valid periodic image: the right ghost column / row / plane is
    // corrected with p of ITS right neighbour (image index 2)
    for (int rr = 0; rr < R + 3; ++rr) prow[rr] = (long long)wrap(min(jb - 1 + rr, N1 + 1), n1) * n0 + wrap(min(i, N0), n0);
  }

  // kk = padded plane index (CORR 1: wrapped into the interior; CORR 2: ghost planes are valid, clamp the unused overshoot)
  auto load_plane = [&](Plane<R>& P, int kk) {
    const double* base = u + (long long)(CORR == 3 ? planez(kk) : (CORR == 1 ? wrap(kk, n2) + 1 : (CORR == 2 ? min(kk, N2 - 1) : kk))) * sz;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int rr = 0; rr < R + 2; ++rr) P.v[c][rr] = base[c * g.sc + rowoff[rr]];
  };
  auto load_p = [&](double (&P)[R + 3], int kk) {
    const double* base = CORR == 3 ? pI + (long long)planez(kk) * sz : pI + (long long)(CORR == 2 ? min(kk, N2) : wrap(kk, n2)) * n0 * n1;
#pragma unroll
    for (int rr = 0; rr < R + 3; ++rr) P[rr] = base[prow[rr]];
  };

  const Rec X = rx[UNIFORM ? 1 : min(i, N0 - 2)];
  // u = u* - ∇p for one register plane (padded plane index kk; Pc = p rows of plane kk, Pn = of plane kk+1)
  auto correct = [&](Plane<R>& P, const double (&Pc)[R + 3], const double (&Pn)[R + 3], int kk) {
    if constexpr (CORR == 3) {
      const int kz = planez(kk);  // the volume this register plane holds (image index in a periodic direction)
      const Rec Zc = rz[min(max(kz, 1), N2 - 2)];
      const Rec Xc = rx[min(max(colx, 1), N0 - 2)];
      bool dx[3], dz[3];
#pragma unroll
      for (int al = 0; al < 3; ++al) {
        dx[al] = colx >= g.iu_lo[al][0] && colx < g.iu_hi[al][0];
        dz[al] = kz >= g.iu_lo[al][2] && kz < g.iu_hi[al][2];
      }
#pragma unroll
      for (int rr = 0; rr < R + 2; ++rr) {
        const int jy = rowy(jb - 1 + rr);
        const Rec Y = ry[min(max(jy, 1), N1 - 2)];
        const double pc = Pc[rr];
        const double gx = (from_next(pc) - pc) * Xc.rs, gy = (Pc[rr + 1] - pc) * Y.rs, gz = (Pn[rr] - pc) * Zc.rs;
        if (dx[0] && dz[0] && jy >= g.iu_lo[0][1] && jy < g.iu_hi[0][1]) P.v[0][rr] -= gx;
        if (dx[1] && dz[1] && jy >= g.iu_lo[1][1] && jy < g.iu_hi[1][1]) P.v[1][rr] -= gy;
        if (dx[2] && dz[2] && jy >= g.iu_lo[2][1] && jy < g.iu_hi[2][1]) P.v[2][rr] -= gz;
      }
      return;
    }
    const Rec Z = rz[UNIFORM ? 1 : min(max(kk, 1), N2 - 2)];
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr) {
      const Rec Y = ry[UNIFORM ? 1 : min(max(jb - 1 + rr, 1), N1 - 2)];
      const double pc = Pc[rr];
      P.v[0][rr] -= (from_next(pc) - pc) * X.rs;
      P.v[1][rr] -= (Pc[rr + 1] - pc) * Y.rs;
      P.v[2][rr] -= (Pn[rr] - pc) * Z.rs;
    }
  };

  bool dofx[3];
  if (MASKED) {
#pragma unroll
    for (int al = 0; al < 3; ++al) dofx[al] = i >= g.iu_lo[al][0] && i < g.iu_hi[al][0];
  }

  double zprev[3][R];

  // z-fluxes through the upper face of plane k (C = plane k, Nx = plane k+1)
  auto zflux = [&](const Plane<R>& C, const Plane<R>& Nx, int k, double (&out)[3][R]) {
    const Rec Z = rz[UNIFORM ? 1 : k];
#pragma unroll
    for (int rr = 1; rr <= R; ++rr) {
      const Rec Y = ry[UNIFORM ? 1 : min(jb - 1 + rr, N1 - 2)];
      const double Wc = C.v[2][rr];
      out[0][rr - 1] = flux(C.v[0][rr], Nx.v[0][rr], Wc, from_next(Wc), X.a2, X.b2, Z.vo);
      out[1][rr - 1] = flux(C.v[1][rr], Nx.v[1][rr], Wc, C.v[2][rr + 1], Y.a2, Y.b2, Z.vo);
      out[2][rr - 1] = flux(Wc, Nx.v[2][rr], Wc, Nx.v[2][rr], Z.a2, Z.b2, Z.vs);
    }
  };

  auto body = [&](const Plane<R>& C, const Plane<R>& Nx, int k) {
    const Rec Z = rz[UNIFORM ? 1 : k];
    const bool kin_u = !MASKED || (k >= g.iu_lo[0][2] && k < g.iu_hi[0][2]);
    const bool kin_v = !MASKED || (k >= g.iu_lo[1][2] && k < g.iu_hi[1][2]);
    const bool kin_w = !MASKED || (k >= g.iu_lo[2][2] && k < g.iu_hi[2][2]);
    // y-fluxes through the face between rows rr and rr+1, rr = 0..R
    double fyu[R + 1], fyv[R + 1], fyw[R + 1];
#pragma unroll
    for (int rr = 0; rr <= R; ++rr) {
      const Rec Y = ry[UNIFORM ? 1 : min(jb - 1 + rr, N1 - 2)];
      const double Vc = C.v[1][rr];
      fyu[rr] = flux(C.v[0][rr], C.v[0][rr + 1], Vc, from_next(Vc), X.a1, X.b1, Y.vo);
      fyv[rr] = flux(Vc, C.v[1][rr + 1], Vc, C.v[1][rr + 1], Y.a1, Y.b1, Y.vs);
      fyw[rr] = flux(C.v[2][rr], C.v[2][rr + 1], Vc, Nx.v[1][rr], Z.a1, Z.b1, Y.vo);
    }
    double znew[3][R];
    zflux(C, Nx, k, znew);
#pragma unroll
    for (int rr = 1; rr <= R; ++rr) {
      const int j = jb - 1 + rr;
      const Rec Y = ry[UNIFORM ? 1 : min(j, N1 - 2)];
      const double Uc = C.v[0][rr], Vc = C.v[1][rr], Wc = C.v[2][rr];
      const double Un = from_next(Uc), Vn = from_next(Vc), Wn = from_next(Wc);
      const double fxu = flux(Uc, Un, Uc, Un, X.a0, X.b0, X.vs);
      const double fxv = flux(Vc, Vn, Uc, C.v[0][rr + 1], Y.a0, Y.b0, X.vo);
      const double fxw = flux(Wc, Wn, Uc, Nx.v[0][rr], Z.a0, Z.b0, X.vo);
      double fu = (fxu - from_prev(fxu)) * X.rs;
      double fv = (fxv - from_prev(fxv)) * X.ro;
      double fw = (fxw - from_prev(fxw)) * X.ro;
      fu += (fyu[rr] - fyu[rr - 1]) * Y.ro;
      fv += (fyv[rr] - fyv[rr - 1]) * Y.rs;
      fw += (fyw[rr] - fyw[rr - 1]) * Y.ro;
      fu += (znew[0][rr - 1] - zprev[0][rr - 1]) * Z.ro;
      fv += (znew[1][rr - 1] - zprev[1][rr - 1]) * Z.ro;
      fw += (znew[2][rr - 1] - zprev[2][rr - 1]) * Z.rs;
      zprev[0][rr - 1] = znew[0][rr - 1];
      zprev[1][rr - 1] = znew[1][rr - 1];
      zprev[2][rr - 1] = znew[2][rr - 1];
      if (MASKED) {
        const bool ju = j >= g.iu_lo[0][1] && j < g.iu_hi[0][1];
        const bool jv = j >= g.iu_lo[1][1] && j < g.iu_hi[1][1];
        const bool jw = j >= g.iu_lo[2][1] && j < g.iu_hi[2][1];
        fu = (dofx[0] && ju && kin_u) ? fu : 0.0;
        fv = (dofx[1] && jv && kin_v) ? fv : 0.0;
        fw = (dofx[2] && jw && kin_w) ? fw : 0.0;
      }
      if (xout && j <= N1 - 2) {
        const long long c = i + (long long)j * N0 + (long long)k * sz;
        if constexpr (FUSE && CORR == 0) {
          if (epi.extra) {  // the stage force is F + E (closure term: ins_rk_ext.hip); E is zero off the degrees of freedom
            fu += epi.extra[c];
            fv += epi.extra[c + g.sc];
            fw += epi.extra[c + 2 * g.sc];
          }
          if (epi.gtemp) {  // gravity!: F[I, gdir] += α2 avg(temp, Δ, I, gdir) on Iu[gdir] (operators.jl:914-931, avg: :59-62)
            const int gd = epi.gdir;
            const int I3[3] = {i, j, k};
            bool dof = true;
#pragma unroll
            for (int b = 0; b < 3; ++b) dof = dof && I3[b] >= g.iu_lo[gd][b] && I3[b] < g.iu_hi[gd][b];
            if (dof) {
              const double d0 = g.dx[gd][I3[gd]], d1 = g.dx[gd][I3[gd] + 1];
              const double gv = epi.ga2 * ((d1 * epi.gtemp[c] + d0 * epi.gtemp[c + g.sx[gd]]) / (d0 + d1));
              if (gd == 0) fu += gv;
              if (gd == 1) fv += gv;
              if (gd == 2) fw += gv;
            }
          }
        }
        if (FUSE) {
          double su, sv, sw;
          if (epi.ustart) {
            const double c0 = 1.0 + epi.c0m1;  // exactly 1 in the k-basis
            su = c0 * epi.ustart[c];
            sv = c0 * epi.ustart[c + g.sc];
            sw = c0 * epi.ustart[c + 2 * g.sc];
          } else {
            su = Uc;
            sv = Vc;
            sw = Wc;
            if (CORR && epi.ustart_out) {  // first stage of a chained step: the corrected stencil input is this step's ustart (ins_rk_steps_f64)
              epi.ustart_out[c] = Uc;
              epi.ustart_out[c + g.sc] = Vc;
              epi.ustart_out[c + 2 * g.sc] = Wc;
            }
          }
          for (int q = 0; q < epi.n; ++q) {
            const double* kq = epi.k[q];
            su += epi.coef[q] * kq[c];
            sv += epi.coef[q] * kq[c + g.sc];
            sw += epi.coef[q] * kq[c + 2 * g.sc];
          }
          su += epi.coef_self * fu;
          sv += epi.coef_self * fv;
          sw += epi.coef_self * fw;
          epi.ustar[c] = su;
          epi.ustar[c + g.sc] = sv;
          epi.ustar[c + 2 * g.sc] = sw;
        }
        if (!FUSE || epi.write_k) {
          F[c] = fu;
          F[c + g.sc] = fv;
          F[c + 2 * g.sc] = fw;
        }
      }
    }
  };

  Plane<R> A, B, Cc;
  if (!CORR) {
    load_plane(A, k0 - 1);
    load_plane(B, k0);
    load_plane(Cc, min(k0 + 1, N2 - 1));
    zflux(A, B, k0 - 1, zprev);
    int k = k0;
    // 3-buffer rotation, unrolled so every register index is static: compute plane k from (cur, next)
    // while the load of plane k+2 into the third buffer is in flight.
    while (true) {
      load_plane(A, min(k + 2, N2 - 1));
      if (bar) __builtin_amdgcn_s_barrier();  // the wavefronts of a workgroup stay on one plane: shared halo lines are cache hits
      body(B, Cc, k);
      if (++k >= k1) break;
      load_plane(B, min(k + 2, N2 - 1));
      if (bar) __builtin_amdgcn_s_barrier();  // the wavefronts of a workgroup stay on one plane: shared halo lines are cache hits
      body(Cc, A, k);
      if (++k >= k1) break;
      load_plane(Cc, min(k + 2, N2 - 1));
      if (bar) __builtin_amdgcn_s_barrier();  // the wavefronts of a workgroup stay on one plane: shared halo lines are cache hits
      body(A, B, k);
      if (++k >= k1) break;
    }
  } else {
    // Same rotation with the pressure correction applied to each plane when it becomes `next`:
    // invariant at the top of iteration k: cur = corrected plane k, nxt = RAW plane k+1, Pa = p(k+1), Pb = p(k+2).
    double Pa[R + 3], Pb[R + 3];
    load_p(Pa, k0 - 1);
    load_p(Pb, k0);
    load_plane(A, k0 - 1);
    load_plane(B, k0);
    correct(A, Pa, Pb, k0 - 1);
    load_p(Pa, k0 + 1);
    correct(B, Pb, Pa, k0);
    load_plane(Cc, k0 + 1);
    load_p(Pb, k0 + 2);
    zflux(A, B, k0 - 1, zprev);
    int k = k0;
    while (true) {
      correct(Cc, Pa, Pb, k + 1);      // plane k+1 with p(k+1), p(k+2)
      load_plane(A, k + 2);
      load_p(Pa, k + 3);
      if (bar) __builtin_amdgcn_s_barrier();  // the wavefronts of a workgroup stay on one plane: shared halo lines are cache hits
      body(B, Cc, k);
      if (++k >= k1) break;
      correct(A, Pb, Pa, k + 1);
      load_plane(B, k + 2);
      load_p(Pb, k + 3);
      if (bar) __builtin_amdgcn_s_barrier();  // the wavefronts of a workgroup stay on one plane: shared halo lines are cache hits
      body(Cc, A, k);
      if (++k >= k1) break;
      correct(B, Pa, Pb, k + 1);
      load_plane(Cc, k + 2);
      load_p(Pa, k + 3);
      if (bar) __builtin_amdgcn_s_barrier();  // the wavefronts of a workgroup stay on one plane: shared halo lines are cache hits
      body(A, B, k);
      if (++k >= k1) break;
      // second half of the period-6 rotation (the two p buffers have swapped roles)
      correct(Cc, Pb, Pa, k + 1);
      load_plane(A, k + 2);
      load_p(Pb, k + 3);
      if (bar) __builtin_amdgcn_s_barrier();  // the wavefronts of a workgroup stay on one plane: shared halo lines are cache hits
      body(B, Cc, k);
      if (++k >= k1) break;
      correct(A, Pa, Pb, k + 1);
      load_plane(B, k + 2);
      load_p(Pa, k + 3);
      if (bar) __builtin_amdgcn_s_barrier();  // the wavefronts of a workgroup stay on one plane: shared halo lines are cache hits
      body(Cc, A, k);
      if (++k >= k1) break;
      correct(B, Pb, Pa, k + 1);
      load_plane(Cc, k + 2);
      load_p(Pb, k + 3);
      if (bar) __builtin_amdgcn_s_barrier();  // the wavefronts of a workgroup stay on one plane: shared halo lines are cache hits
      body(A, B, k);
      if (++k >= k1) break;
    }
  }
}

// Zero the ghost shell of a vector field (the part of `fill!(F, 0)` the interior sweep does not cover).
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

// Defaults (0 = pick per grid, see below); measured on MI355X, profiles/r01_k1_scan.txt.
// run-time options (ins_options.hip; environment variable of the same name, or ins_set_option)
#define g_rows ((int)ins_opt(OPT_INS_FLUX_ROWS))
#define g_zchunk ((int)ins_opt(OPT_INS_FLUX_ZC))
#define g_xw ((int)ins_opt(OPT_INS_FLUX_XW))

// Tuning knobs for experiments (not part of the public ABI).
extern "C" void ins_tune_flux3d(int rows, int zchunk, int xw) {
  ins_set_option("INS_FLUX_ROWS", (rows >= 2 && rows <= 4) ? rows : 0);
  ins_set_option("INS_FLUX_ZC", zchunk >= 1 ? zchunk : 0);
  ins_set_option("INS_FLUX_XW", (xw == 1 || xw == 2 || xw == 4) ? xw : 0);
}

int ins_flux3d_prepare(const ins_grid* G, double visc, hipStream_t s) {
  ins_grid* M = const_cast<ins_grid*>(G);  // metric-record cache keyed by ν
  const GridDev& g = G->g;
  if (!M->rec_dev) {
    const size_t n = (size_t)g.N[0] + g.N[1] + g.N[2];
    INS_HIP_TRY(hipMalloc(&M->rec_dev, n * sizeof(Rec)));
    M->rec_visc = -1.0;
  }
  if (M->rec_visc != visc) {
    Rec* r0 = reinterpret_cast<Rec*>(M->rec_dev);
    Rec* r1 = r0 + g.N[0];
    Rec* r2 = r1 + g.N[1];
    const int nmax = std::max(g.N[0], std::max(g.N[1], g.N[2]));
    hipLaunchKernelGGL(k_build_recs, dim3(cdiv(nmax, 256), 3), dim3(256), 0, s, g, visc, r0, r1, r2, 0);
    INS_LAUNCH_CHECK();
    M->rec_visc = visc;
  }
  return INS_OK;
}

template <int R, int XW, bool FUSE>
static int launch_flux(const ins_grid* G, const double* u, double* F, const RkEpi& epi, const double* pI, int corr_mode, hipStream_t s,
                       const void* recs = nullptr) {
  const GridDev& g = G->g;
  const Rec* r0 = reinterpret_cast<const Rec*>(recs ? recs : G->rec_dev);
  const Rec* r1 = r0 + g.N[0];
  const Rec* r2 = r1 + g.N[1];
  const int zc = g_zchunk ? g_zchunk : ((g.N[2] >= 384 || (pI && corr_mode == 3 && g.N[2] >= 128)) ? 8 : 4);
  const int bar = ins_opt(OPT_INS_FLUX_BAR) ? 1 : 0;  // measured neutral on the masked cavity kernel (6.31 vs 6.37 ms/step): off by default
  const bool corr = pI != nullptr;
  const bool masked = !G->all_dof;
  constexpr int RC = R > 3 ? 3 : R;              // register-heavy variants cap the rows per thread: masked 3,
  constexpr int RK = R > 2 ? 2 : R;              // correcting 2 (236 VGPRs; 3 rows need all 256 and run slower)
  const int reff = corr ? RK : (masked ? RC : R);
  const int xo = corr ? XOUT - 1 : XOUT;
  const int ntx = cdiv(g.N[0] - 2, xo * XW), nty = cdiv(g.N[1] - 2, (4 / XW) * reff), ntz = cdiv(g.N[2] - 2, zc);
  dim3 block(64, 4, 1);
  const unsigned nb = (unsigned)(8LL * ntx * ((nty + 7) / 8) * ntz);
  if (corr && corr_mode == 3) {  // masked / stretched grids with Periodic and Dirichlet sides: padded p, DOF-masked correction
    if constexpr (FUSE) {
      constexpr int R3 = R > 2 ? 2 : R;
      const int nty3 = cdiv(g.N[1] - 2, (4 / XW) * R3), ntx3 = cdiv(g.N[0] - 2, (XOUT - 1) * XW);
      const unsigned nb3 = (unsigned)(8LL * ntx3 * ((nty3 + 7) / 8) * ntz);
      hipLaunchKernelGGL((k_momentum_flux<R3, false, true, XW, true, 3>), dim3(nb3), block, 0, s, g, r0, r1, r2, u, F, zc, ntx3, nty3, ntz, epi, pI, bar);
      INS_LAUNCH_CHECK();
      return INS_OK;
    } else {
      ins_set_error("in-kernel pressure correction needs the fused epilogue");
      return INS_ERR_UNSUPPORTED;
    }
  } else if (corr) {
    if (!(FUSE && G->uniform_exact && !masked)) {
      ins_set_error("in-kernel pressure correction needs the fused path on an exactly uniform periodic grid");
      return INS_ERR_UNSUPPORTED;
    }
    if constexpr (FUSE)
      {
        if (corr_mode == 2)
          hipLaunchKernelGGL((k_momentum_flux<RK, true, false, XW, true, 2>), dim3(nb), block, 0, s, g, r0, r1, r2, u, F, zc, ntx, nty, ntz, epi, pI, bar);
        else
          hipLaunchKernelGGL((k_momentum_flux<RK, true, false, XW, true, 1>), dim3(nb), block, 0, s, g, r0, r1, r2, u, F, zc, ntx, nty, ntz, epi, pI, bar);
      }
  } else if (G->uniform_exact && !masked)
    hipLaunchKernelGGL((k_momentum_flux<R, true, false, XW, FUSE, 0>), dim3(nb), block, 0, s, g, r0, r1, r2, u, F, zc, ntx, nty, ntz, epi, pI, bar);
  else if (!masked)
    hipLaunchKernelGGL((k_momentum_flux<R, false, false, XW, FUSE, 0>), dim3(nb), block, 0, s, g, r0, r1, r2, u, F, zc, ntx, nty, ntz, epi, pI, bar);
  else  // masked: the epilogue leaves u* = ustart (+ 0) on volumes that are no DOF, apply_bc_u! sets them afterwards as always
    hipLaunchKernelGGL((k_momentum_flux<RC, false, true, XW, FUSE, 0>), dim3(nb), block, 0, s, g, r0, r1, r2, u, F, zc, ntx, nty, ntz, epi, pI, bar);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

template <bool FUSE>
static int launch_flux_any(const ins_grid* G, const double* u, double* F, const RkEpi& epi, const double* pI, int corr_mode, hipStream_t s,
                           const void* recs = nullptr) {
  // wavefronts side by side in x: 4 when the row needs >= 8 of them, else 2 (fewer mostly-empty workgroups)
  // (the correcting kernels produce one column less per wavefront; a 256-wide row then takes 5 wavefronts, and workgroups of 2 or 4 side by
  // side would launch 6 or 8: the side-by-side count is the largest of 4, 2, 1 that wastes no wavefront — cavity 256^3: 6.26 -> 6.05 ms/step)
  const int waves_x = cdiv(G->g.N[0] - 2, pI ? XOUT - 1 : XOUT);
  int xw = g_xw;
  if (!xw) {
    xw = waves_x >= 8 ? 4 : (waves_x >= 2 ? 2 : 1);
    while (xw > 1 && cdiv(waves_x, xw) * xw > waves_x) xw >>= 1;
  }
  const int rows = g_rows ? g_rows : 4;  // rows per thread (masked / correcting variants cap themselves at 3: registers)
#define INS_FLUX_CASE(RR)                                                          \
  if (rows == RR) {                                                                \
    if (xw == 4) return launch_flux<RR, 4, FUSE>(G, u, F, epi, pI, corr_mode, s, recs);  \
    if (xw == 2) return launch_flux<RR, 2, FUSE>(G, u, F, epi, pI, corr_mode, s, recs);  \
    return launch_flux<RR, 1, FUSE>(G, u, F, epi, pI, corr_mode, s, recs);               \
  }
  INS_FLUX_CASE(2)
  INS_FLUX_CASE(3)
  INS_FLUX_CASE(4)
#undef INS_FLUX_CASE
  return INS_ERR_INVALID;
}

// 64-outputs-per-wavefront specialisation for periodic, exactly-uniform boxes (ins_flux64.hip)
bool ins_flux64_supported(const ins_grid* G);
int ins_k_flux64(const ins_grid* G, double visc, const double* u, double* F, const RkEpi* epi, const double* pI, int corr_mode, hipStream_t s,
                 int part = 0);

int ins_k_momentum_flux3d(const ins_grid* G, double visc, const double* u, double* F, bool zero_shell, hipStream_t s) {
  int rc;
  // plain K1: the 64-wide kernel (workgroup barrier per plane, 8 wavefronts per workgroup on large boxes) runs at the flat-copy rate;
  // INS_FLUX64_62_FROM=n routes boxes from n cells per row on to the 62-wide kernel (round-1 routing: 448) for A/B
  const long long from62 = ins_opt(OPT_INS_FLUX64_62_FROM);
  if (ins_flux64_supported(G) && (from62 <= 0 || G->g.N[0] - 2 < from62)) {
    if ((rc = ins_k_flux64(G, visc, u, F, nullptr, nullptr, 0, s))) return rc;
  } else {
    if ((rc = ins_flux3d_prepare(G, visc, s))) return rc;
    RkEpi epi;
    memset(&epi, 0, sizeof(epi));
    if ((rc = launch_flux_any<false>(G, u, F, epi, nullptr, 0, s))) return rc;
  }
  if (zero_shell) {
    const GridDev& g = G->g;
    const long long total = 2LL * ((long long)g.N[0] * g.N[1] + (long long)g.N[0] * g.N[2] + (long long)g.N[1] * g.N[2]);
    hipLaunchKernelGGL(k_zero_shell, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, s, g, F);
    INS_LAUNCH_CHECK();
  }
  return INS_OK;
}

// fill!(F, 0) + diffusion!(F, u) (operators.jl:537-573) on the tiled kernel: the same face-flux sweep with records whose interpolation weights are
// zero, so every face flux is its diffusive part alone (dissipation! needs diffusion(u) as a field of its own: ins_rk_ext.hip, ins_fields.hip).
// zero_shell: also zero the ghost shell of F (a caller's array; the library's own scratch fields keep theirs zero).
bool ins_fast3d_supported(const ins_grid* G);
int ins_k_diffusion_flux3d(const ins_grid* G, double visc, const double* u, double* F, bool zero_shell, hipStream_t s) {
  const GridDev& g = G->g;
  if (!ins_fast3d_supported(G)) return INS_ERR_UNSUPPORTED;
  ins_grid* M = const_cast<ins_grid*>(G);
  if (!M->rec_diff_dev) {
    const size_t n = (size_t)g.N[0] + g.N[1] + g.N[2];
    INS_HIP_TRY(hipMalloc(&M->rec_diff_dev, n * sizeof(Rec)));
    M->rec_diff_visc = -1.0;
  }
  if (M->rec_diff_visc != visc) {
    Rec* r0 = reinterpret_cast<Rec*>(M->rec_diff_dev);
    Rec* r1 = r0 + g.N[0];
    Rec* r2 = r1 + g.N[1];
    const int nmax = std::max(g.N[0], std::max(g.N[1], g.N[2]));
    hipLaunchKernelGGL(k_build_recs, dim3(cdiv(nmax, 256), 3), dim3(256), 0, s, g, visc, r0, r1, r2, 1);
    INS_LAUNCH_CHECK();
    M->rec_diff_visc = visc;
  }
  RkEpi epi;
  memset(&epi, 0, sizeof(epi));
  int rc = launch_flux_any<false>(G, u, F, epi, nullptr, 0, s, M->rec_diff_dev);
  if (rc || !zero_shell) return rc;
  const long long total = 2LL * ((long long)g.N[0] * g.N[1] + (long long)g.N[0] * g.N[2] + (long long)g.N[1] * g.N[2]);
  hipLaunchKernelGGL(k_zero_shell, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, s, g, F);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// K1 + K6: k_i = momentum(u_in) (stored when epi.write_k) and the stage velocity u* (interior) in one pass.
// 64 outputs per wavefront on stretched / masked grids (ins_flux64m.hip)
bool ins_flux64m_supported(const ins_grid* G);
int ins_k_flux64m(const ins_grid* G, double visc, const double* u, double* k_out, const RkEpi& epi, const double* p_padded, hipStream_t s);

int ins_k_momentum_rk_fused(const ins_grid* G, double visc, const double* u_in, double* k_out, const RkEpi& epi, hipStream_t s) {
  if (ins_flux64_supported(G)) return ins_k_flux64(G, visc, u_in, k_out, &epi, nullptr, 0, s);
  // (INS_FLUX64M_SKIP_FIRST=1 keeps the 62-wide kernel for the non-correcting first stage: cavity 256^3 4.15 vs 4.11 ms/step)
  if ((!ins_opt(OPT_INS_FLUX64M_SKIP_FIRST) || epi.wout) && ins_flux64m_supported(G)) return ins_k_flux64m(G, visc, u_in, k_out, epi, nullptr, s);
  int rc = ins_flux3d_prepare(G, visc, s);
  if (rc) return rc;
  return launch_flux_any<true>(G, u_in, k_out, epi, nullptr, 0, s);
}

// Same, but u_in is the previous stage's UNCORRECTED u* (interior only) and pI its unpadded pressure: the
// projection's gradient-subtract is applied in registers (periodic, exactly-uniform grids).
int ins_k_momentum_rk_fused_corr(const ins_grid* G, double visc, const double* ustar_prev, const double* pI, double* k_out, const RkEpi& epi,
                                 hipStream_t s) {
  if (ins_flux64_supported(G)) return ins_k_flux64(G, visc, ustar_prev, k_out, &epi, pI, 1, s);
  int rc = ins_flux3d_prepare(G, visc, s);
  if (rc) return rc;
  return launch_flux_any<true>(G, ustar_prev, k_out, epi, pI, 1, s);
}

bool ins_fast3d_supported(const ins_grid* G);
// Masked / stretched grids whose sides are all Periodic or Dirichlet (cavities, channels): ustar_prev is the previous stage's uncorrected
// u* with apply_bc_u! applied, p_padded that stage's pressure (padded layout; only interior values and periodic images are read).
bool ins_corr3_supported(const ins_grid* G) {
  const GridDev& g = G->g;
  if (g.D != 3 || !ins_fast3d_supported(G) || g.N[0] < 8 || g.N[1] < 8 || g.N[2] < 8) return false;
  for (int a = 0; a < 3; ++a)
    for (int sd = 0; sd < 2; ++sd)
      if (g.bc[a][sd] != INS_BC_PERIODIC && g.bc[a][sd] != INS_BC_DIRICHLET) return false;
  return true;
}
int ins_k_momentum_rk_fused_corr3(const ins_grid* G, double visc, const double* ustar_prev, const double* p_padded, double* k_out, const RkEpi& epi,
                                  hipStream_t s) {
  if (ins_flux64m_supported(G)) return ins_k_flux64m(G, visc, ustar_prev, k_out, epi, p_padded, s);
  int rc = ins_flux3d_prepare(G, visc, s);
  if (rc) return rc;
  return launch_flux_any<true>(G, ustar_prev, k_out, epi, p_padded, 3, s);
}

// Slab flavour: z neighbours from ghost planes; p_ext = [1 plane below | local planes | 2 planes above] (unpadded in x, y).
// part (z-chunk subsets, see ins_flux64.hip): 1 = chunks that need no ghost plane, 2 = the rest, 0 = all.  Grids the 64-wide kernel does
// not take run everything in part 2 (and part 0).
int ins_k_momentum_rk_fused_corr_slab(const ins_grid* G, double visc, const double* ustar_prev, const double* p_ext, double* k_out,
                                      const RkEpi& epi, hipStream_t s, int part) {
  if (ins_flux64_supported(G)) return ins_k_flux64(G, visc, ustar_prev, k_out, &epi, p_ext, 2, s, part);
  if (part == 1) return INS_OK;
  int rc = ins_flux3d_prepare(G, visc, s);
  if (rc) return rc;
  return launch_flux_any<true>(G, ustar_prev, k_out, epi, p_ext, 2, s);
}
