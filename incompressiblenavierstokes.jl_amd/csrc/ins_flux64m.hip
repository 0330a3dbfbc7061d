// K1 + K6 (+ in-register pressure correction) with 64 outputs per wavefront on STRETCHED and MASKED grids: the scheme of ins_flux64.hip
// (convection_diffusion_kernel! + fill!(F, 0), operators.jl:647-690, 971; RK epilogue step_explicit_runge_kutta.jl:35-38; correction
// applypressure!, operators.jl:225-233) with the metric records and degree-of-freedom masks of ins_fast3d_flux.hip.
//
// The 62-wide kernel of ins_fast3d_flux.hip serves these grids with three register planes, 64-bit per-row addresses and one halo lane
// on each side: its correcting variant needs 255 VGPRs (+ scratch) and produces 61 columns per wavefront — a 256-wide cavity row takes 5
// wavefronts — and runs at 0.34 of 8 TB/s (0.64 ms per stage at 256^3).  Here, as in ins_flux64.hip:
//   * all 64 lanes produce output; the halo columns x0-1 and x0+64 of all rows arrive in one packed load per component and plane
//     (lanes 0..7 / 16..23), enter the wave shifts as the DPP `old` operand, and lane 0's left-face x-fluxes are evaluated from them;
//   * two register planes: a row is re-loaded with plane k+2 as soon as plane k has consumed it;
//   * wave-uniform row starts in SGPRs + one 32-bit lane offset (buffer loads): no 64-bit address registers.
// What is new against ins_flux64.hip: per-lane x record (10 doubles), scalar y / z records per row / plane, the record of column x0-1
// for lane 0's left faces, DOF masks on the results, boundary data instead of periodic images behind Dirichlet sides (clamped reads: the
// ghost / boundary volumes of the input hold valid data), and CORR = 3: the input is the previous stage's UNCORRECTED u* with
// apply_bc_u! applied, `p` the PADDED pressure of its projection; planes and the packed halo columns are corrected in registers on their
// degrees of freedom only, periodic directions read through the periodic image (their ghost volumes hold uncorrected copies).
#include <algorithm>
#include <cstring>

#include "ins_internal.h"
#include "ins_wave64.h"

int ins_flux3d_prepare(const ins_grid* G, double visc, hipStream_t s);
bool ins_fast3d_supported(const ins_grid* G);

namespace {

struct FluxMArgs {
  const double* u;
  const double* p;  // CORR = 3: padded pressure
  double* F;
  long long sc;
  int N0, N1, N2;
  int zc, ntx, nty, ntz;
  int bar;
  const Rec *rx, *ry, *rz;
  int per[3];  // CORR = 3: direction read through the periodic image
  int lo[3][3], hi[3][3];  // Iu[α] = [lo[α][β], hi[α][β]) (padded indices)
  const double* gdx;       // WT: Δ of the gravity direction (avg(temp, Δ, I, gdir), operators.jl:59-62)
  RkEpi epi;
};

// face flux: ν(up - uc)/Δb - ½(uc + up)(A₂ ub0 + A₁ ub1)   (weights pre-halved in the record)
__device__ __forceinline__ double fluxm(double uc, double up, double ub0, double ub1, double ha, double hb, double vd) {
  const double uba = ha * ub0 + hb * ub1;
  return (up - uc) * vd - (uc + up) * uba;
}

template <int R>
struct PlaneM {
  double v[3][R + 2];
  double h[3];  // packed halo columns: lane r = row r of column x0-1, lane 16+r = row r of column x0+64
};

// WT (extended stage loop on wall-bounded grids, ins_rk_ext.hip; CORR = 0): gravity from epi.gtemp and the closure force epi.extra are added to the
// stage force, and
// w_α = u_α · diffusion(u)_α is stored to epi.wout — the diffusive part of every face flux is accumulated a second time on its own.
template <int R, int XW, int CORR, bool WT = false>
__global__ __launch_bounds__(256, 2) void k_flux64m(FluxMArgs a) {  // (256, 3) = 168 VGPRs + 284 B scratch measured neutral (cavity 4.27 vs 4.29 ms/step); 8 wavefronts (two
                                                                    // stacked groups of four, barrier per plane) slower: 4.20 vs 4.10 (round 3)
  constexpr unsigned EB = 8;
  constexpr int NW = 4;
  static_assert(R + 2 + (CORR ? 1 : 0) <= 8, "packed halo rows live in 8-lane groups");
  int txi, tyi, tzi;
  {
    const int nty_local = (a.nty + 7) >> 3;
    int seq = (int)(blockIdx.x >> 3);
    if (seq >= a.ntx * nty_local * a.ntz) return;
    txi = seq % a.ntx;
    seq /= a.ntx;
    tyi = (int)(blockIdx.x & 7) * nty_local + seq % nty_local;
    tzi = seq / nty_local;
    if (tyi >= a.nty) return;
  }
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int wx = wave % XW, wy = wave / XW;
  const int N0 = a.N0, N1 = a.N1, N2 = a.N2;
  const int n0 = N0 - 2, n1 = N1 - 2, n2 = N2 - 2;
  const int x0 = (txi * XW + wx) * 64;         // interior (0-based) column of lane 0
  const int jb0 = (tyi * (NW / XW) + wy) * R;  // interior row of the first output row
  const int k0 = 1 + tzi * a.zc;               // padded plane index of the first output plane
  const int k1 = min(k0 + a.zc, N2 - 1);
  if (x0 >= n0 || jb0 >= n1) {  // wavefront outside the box: it only keeps the workgroup's barrier count (one per plane)
    if (a.bar)
      for (int k = k0; k < k1; ++k) __builtin_amdgcn_s_barrier();
    return;
  }
  const long long sz = (long long)N0 * N1;
  const int ci = x0 + lane;
  const bool xout = ci < n0;
  const bool perx = CORR && a.per[0], pery = CORR && a.per[1], perz = CORR && a.per[2];
  // padded index of a (possibly out-of-range) interior index: periodic image (CORR) or clamped into the padded range
  auto pcol_of = [&](int c) { return perx ? wrapi(c, n0) + 1 : min(max(c + 1, 0), N0 - 1); };
  auto prow_of = [&](int j) { return pery ? wrapi(j, n1) + 1 : min(max(j + 1, 0), N1 - 1); };
  auto planez = [&](int kk) { return perz ? wrapi(kk - 1, n2) + 1 : min(max(kk, 0), N2 - 1); };  // kk: padded plane index

  unsigned urow[R + 3];  // byte offset of the padded row inside a plane (R + 3: one more row of p for the y-gradient)
#pragma unroll
  for (int rr = 0; rr < R + 3; ++rr) urow[rr] = (unsigned)(prow_of(jb0 - 1 + rr) * N0) * EB;
  const unsigned ubytes = (unsigned)sz * EB;
  const unsigned ucol = (unsigned)pcol_of(ci) * EB;
  const int hr = lane & 7, hgrp = (lane >> 3) & 3;
  unsigned uhoff, qhoff = 0;  // packed halo loads: in-plane byte offset of this lane's (row, column)
  {
    const int ru = hr <= R + 1 ? hr : 0;
    uhoff = (unsigned)(prow_of(jb0 - 1 + ru) * N0 + (hgrp == 2 ? pcol_of(x0 + 64) : pcol_of(x0 - 1))) * EB;
    if (CORR) {
      const int rq = hr <= R + 2 ? hr : 0;
      const int cq = hgrp == 0 ? x0 - 1 : (hgrp == 1 ? x0 : (hgrp == 2 ? x0 + 64 : x0 + 65));
      qhoff = (unsigned)(prow_of(jb0 - 1 + rq) * N0 + pcol_of(cq)) * EB;
    }
  }

  // ---- metric records and masks -----------------------------------------------------------------------------------------
  const Rec X = a.rx[min(ci + 1, N0 - 2)];                                   // this lane's column (nominal padded index)
  const Rec XL = a.rx[min(x0, N0 - 2)];                                      // column x0 - 1: lane 0's left faces
  bool dofx[3];
#pragma unroll
  for (int al = 0; al < 3; ++al) dofx[al] = ci + 1 >= a.lo[al][0] && ci + 1 < a.hi[al][0];
  // pressure correction: the volume a register holds is the IMAGE (periodic directions) — its gradient metric and DOF status
  double gX = 0.0, gXh = 0.0, gYh = 0.0;
  bool mX[3] = {false, false, false}, mH[3] = {false, false, false};
  if (CORR) {
    const int colx = pcol_of(ci);
    gX = a.rx[min(max(colx, 1), N0 - 2)].rs;
    const int hcol = hgrp == 2 ? pcol_of(x0 + 64) : pcol_of(x0 - 1);
    const int hrow = prow_of(jb0 - 1 + (hr <= R + 1 ? hr : 0));
    gXh = a.rx[min(max(hcol, 1), N0 - 2)].rs;
    gYh = a.ry[min(max(hrow, 1), N1 - 2)].rs;
#pragma unroll
    for (int al = 0; al < 3; ++al) {
      mX[al] = colx >= a.lo[al][0] && colx < a.hi[al][0];
      mH[al] = hcol >= a.lo[al][0] && hcol < a.hi[al][0] && hrow >= a.lo[al][1] && hrow < a.hi[al][1];
    }
  }

  // y records of the R + 1 rows whose upper faces this wavefront evaluates, and the y-gradient metrics of its R + 2 register rows: they do
  // not change from plane to plane (wave-uniform: SGPRs, or lanes of a spill VGPR — cheaper than a scalar load per row and plane)
  Rec Yr[R + 1];
#pragma unroll
  for (int rr = 0; rr <= R; ++rr) Yr[rr] = a.ry[min(jb0 + rr, N1 - 2)];
  double gYr[R + 2];
  int jyr[R + 2];
#pragma unroll
  for (int rr = 0; rr < R + 2; ++rr) {
    jyr[rr] = prow_of(jb0 - 1 + rr);
    gYr[rr] = CORR ? a.ry[min(max(jyr[rr], 1), N1 - 2)].rs : 0.0;
  }

  auto load_plane = [&](PlaneM<R>& P, int kk) {
    const double* base = a.u + (long long)(CORR ? planez(kk) : kk) * sz;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const rsrc_t rs = plane_rsrc(base + c * a.sc, ubytes);
#pragma unroll
      for (int rr = 0; rr < R + 2; ++rr) P.v[c][rr] = ldb<double>(rs, ucol, urow[rr]);
      P.h[c] = ldb<double>(rs, uhoff, 0);
    }
  };
  auto load_p = [&](double (&P)[R + 3], double& PH, int kk) {
    const rsrc_t rs = plane_rsrc(a.p + (long long)planez(kk) * sz, ubytes);
#pragma unroll
    for (int rr = 0; rr < R + 3; ++rr) P[rr] = ldb<double>(rs, ucol, urow[rr]);
    PH = ldb<double>(rs, qhoff, 0);
  };
  // u = u* - ∇p on the degrees of freedom of one register plane (padded plane index kk) and of its packed halo columns
  auto correct = [&](PlaneM<R>& P, const double (&Pc)[R + 3], double PHc, const double (&Pn)[R + 3], double PHn, int kk) {
    const int kz = planez(kk);
    const double gZ = a.rz[min(max(kz, 1), N2 - 2)].rs;
    bool mZ[3];
#pragma unroll
    for (int al = 0; al < 3; ++al) mZ[al] = kz >= a.lo[al][2] && kz < a.hi[al][2];
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr) {
      const int jy = jyr[rr];
      const double gY = gYr[rr];
      const double pc = Pc[rr];
      const double gx = (next_h(pc, rdlane(PHc, 16 + rr)) - pc) * gX, gy = (Pc[rr + 1] - pc) * gY, gz = (Pn[rr] - pc) * gZ;
      if (mX[0] && mZ[0] && jy >= a.lo[0][1] && jy < a.hi[0][1]) P.v[0][rr] -= gx;
      if (mX[1] && mZ[1] && jy >= a.lo[1][1] && jy < a.hi[1][1]) P.v[1][rr] -= gy;
      if (mX[2] && mZ[2] && jy >= a.lo[2][1] && jy < a.hi[2][1]) P.v[2][rr] -= gz;
    }
    // packed: lanes 0..7 hold p of column x0-1, 8..15 of x0, 16..23 of x0+64, 24..31 of x0+65 (rows along the lanes)
    const double hx = (dpp_old<0x108>(PHc, PHc) - PHc) * gXh;  // row_shl:8 — p of the next column, same row
    const double hy = (dpp_old<0x101>(PHc, PHc) - PHc) * gYh;  // row_shl:1 — p of the next row, same column
    const double hz = (PHn - PHc) * gZ;
    if (mH[0] && mZ[0]) P.h[0] -= hx;
    if (mH[1] && mZ[1]) P.h[1] -= hy;
    if (mH[2] && mZ[2]) P.h[2] -= hz;
  };

  double zprev[3][R];
  double dzprev[3][WT ? R : 1];  // WT: the diffusive parts of the same fluxes
  auto zflux0 = [&](const PlaneM<R>& C, const PlaneM<R>& Nx, int k) {  // upper-face z-fluxes of the plane below the chunk
    const Rec Z = a.rz[k];
#pragma unroll
    for (int rr = 1; rr <= R; ++rr) {
      const Rec& Y = Yr[rr];
      const double Wc = C.v[2][rr];
      if constexpr (WT) {
        dzprev[0][rr - 1] = (Nx.v[0][rr] - C.v[0][rr]) * Z.vo;
        dzprev[1][rr - 1] = (Nx.v[1][rr] - C.v[1][rr]) * Z.vo;
        dzprev[2][rr - 1] = (Nx.v[2][rr] - Wc) * Z.vs;
      }
      zprev[0][rr - 1] = fluxm(C.v[0][rr], Nx.v[0][rr], Wc, next_h(Wc, rdlane(C.h[2], 16 + rr)), X.a2, X.b2, Z.vo);
      zprev[1][rr - 1] = fluxm(C.v[1][rr], Nx.v[1][rr], Wc, C.v[2][rr + 1], Y.a2, Y.b2, Z.vo);
      zprev[2][rr - 1] = fluxm(Wc, Nx.v[2][rr], Wc, Nx.v[2][rr], Z.a2, Z.b2, Z.vs);
    }
  };

  // output rows of this wavefront (clamped: rows / columns past the box are computed but never stored)
  unsigned orow[R];
#pragma unroll
  for (int rr = 0; rr < R; ++rr) orow[rr] = (unsigned)((min(jb0 + rr, n1 - 1) + 1) * N0) * EB;
  const unsigned ocol = (unsigned)(min(ci, n0 - 1) + 1) * EB;

  // RK epilogue, first half (issued at the top of the plane): s = c0 ustart + Σ_q coef_q k_q
  auto epi_load = [&](const PlaneM<R>& C, int k, double (&sacc)[3][R]) {
    const long long pk = (long long)k * sz;
    if (a.epi.ustart) {
      const double* b = a.epi.ustart + pk;
      const double c0 = 1.0 + a.epi.c0m1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const rsrc_t rs = plane_rsrc(b + c * a.sc, ubytes);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) sacc[c][rr] = c0 * ldb<double>(rs, ocol, orow[rr]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) sacc[c][rr] = C.v[c][rr + 1];
    }
    for (int q = 0; q < a.epi.n; ++q) {
      const double* kq = a.epi.k[q] + pk;
      const double cq = a.epi.coef[q];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const rsrc_t rs = plane_rsrc(kq + c * a.sc, ubytes);
        double kv[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) kv[rr] = ldb<double>(rs, ocol, orow[rr]);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) sacc[c][rr] += cq * kv[rr];
      }
    }
  };
  auto emit = [&](int rr, int k, double fu, double fv, double fw, double s0, double s1, double s2) {
    const long long pk = (long long)k * sz;
    const int j = jb0 + rr - 1;  // interior row
    if (xout && j < n1) {
      const unsigned rowb = orow[rr - 1], co = ocol;
      double* o = a.epi.ustar + pk;
      stb(plane_rsrc(o, ubytes), co, rowb, s0 + a.epi.coef_self * fu);
      stb(plane_rsrc(o + a.sc, ubytes), co, rowb, s1 + a.epi.coef_self * fv);
      stb(plane_rsrc(o + 2 * a.sc, ubytes), co, rowb, s2 + a.epi.coef_self * fw);
      if (a.epi.write_k) {
        double* f = a.F + pk;
        stb(plane_rsrc(f, ubytes), co, rowb, fu);
        stb(plane_rsrc(f + a.sc, ubytes), co, rowb, fv);
        stb(plane_rsrc(f + 2 * a.sc, ubytes), co, rowb, fw);
      }
    }
  };

  // One output plane.  C = plane k, Nx = plane k+1 (both complete, corrected).  As soon as a row of C has been consumed its registers
  // are re-loaded with the same row of plane `kload` (= k+2), so the prefetch is in flight during the whole of plane k.
  auto body = [&](PlaneM<R>& C, const PlaneM<R>& Nx, int k, int kload) {
    const double* nb = a.u + (long long)(CORR ? planez(kload) : kload) * sz;
    const rsrc_t n0r = plane_rsrc(nb, ubytes), n1r = plane_rsrc(nb + a.sc, ubytes), n2r = plane_rsrc(nb + 2 * a.sc, ubytes);
    const double ch0 = C.h[0], ch1 = C.h[1], ch2 = C.h[2];
    const Rec Z = a.rz[k];
    bool kin[3];
#pragma unroll
    for (int al = 0; al < 3; ++al) kin[al] = k >= a.lo[al][2] && k < a.hi[al][2];
    double sacc[3][R];
    epi_load(C, k, sacc);
    double fyu_o = 0, fyv_o = 0, fyw_o = 0;
    double dyu_o = 0, dyv_o = 0, dyw_o = 0;
#pragma unroll
    for (int rr = 0; rr <= R; ++rr) {
      const Rec& Y = Yr[rr];  // row rr (nominal padded index jb0 + rr)
      const double Uc = C.v[0][rr], Vc = C.v[1][rr], Wc = C.v[2][rr];
      const double Vn = next_h(Vc, rdlane(ch1, 16 + rr));
      // y-fluxes through the face between rows rr and rr+1
      const double fyu = fluxm(Uc, C.v[0][rr + 1], Vc, Vn, X.a1, X.b1, Y.vo);
      const double fyv = fluxm(Vc, C.v[1][rr + 1], Vc, C.v[1][rr + 1], Y.a1, Y.b1, Y.vs);
      const double fyw = fluxm(Wc, C.v[2][rr + 1], Vc, Nx.v[1][rr], Z.a1, Z.b1, Y.vo);
      const double dyu = WT ? (C.v[0][rr + 1] - Uc) * Y.vo : 0.0, dyv = WT ? (C.v[1][rr + 1] - Vc) * Y.vs : 0.0, dyw = WT ? (C.v[2][rr + 1] - Wc) * Y.vo : 0.0;
      if (rr >= 1) {
        const double Un = next_h(Uc, rdlane(ch0, 16 + rr)), Wn = next_h(Wc, rdlane(ch2, 16 + rr));
        const double fxu = fluxm(Uc, Un, Uc, Un, X.a0, X.b0, X.vs);
        const double fxv = fluxm(Vc, Vn, Uc, C.v[0][rr + 1], Y.a0, Y.b0, X.vo);
        const double fxw = fluxm(Wc, Wn, Uc, Nx.v[0][rr], Z.a0, Z.b0, X.vo);
        // left-face fluxes of lane 0 from the halo column x0-1 with ITS record (all other lanes take their left neighbour's right face)
        const double sU = rdlane(ch0, rr), sV = rdlane(ch1, rr), sW = rdlane(ch2, rr);
        const double sUu = rdlane(ch0, rr + 1), sUn = rdlane(Nx.h[0], rr);
        const double lxu = fluxm(sU, Uc, sU, Uc, XL.a0, XL.b0, XL.vs);
        const double lxv = fluxm(sV, Vc, sU, sUu, Y.a0, Y.b0, XL.vo);
        const double lxw = fluxm(sW, Wc, sU, sUn, Z.a0, Z.b0, XL.vo);
        double fu = (fxu - prev_h(fxu, lxu)) * X.rs;
        double fv = (fxv - prev_h(fxv, lxv)) * X.ro;
        double fw = (fxw - prev_h(fxw, lxw)) * X.ro;
        fu += (fyu - fyu_o) * Y.ro;
        fv += (fyv - fyv_o) * Y.rs;
        fw += (fyw - fyw_o) * Y.ro;
        const double zu = fluxm(Uc, Nx.v[0][rr], Wc, Wn, X.a2, X.b2, Z.vo);
        const double zv = fluxm(Vc, Nx.v[1][rr], Wc, C.v[2][rr + 1], Y.a2, Y.b2, Z.vo);
        const double zw = fluxm(Wc, Nx.v[2][rr], Wc, Nx.v[2][rr], Z.a2, Z.b2, Z.vs);
        fu += (zu - zprev[0][rr - 1]) * Z.ro;
        fv += (zv - zprev[1][rr - 1]) * Z.ro;
        fw += (zw - zprev[2][rr - 1]) * Z.rs;
        zprev[0][rr - 1] = zu;
        zprev[1][rr - 1] = zv;
        zprev[2][rr - 1] = zw;
        const int j = jb0 + rr;  // padded row
        const bool du_ = dofx[0] && kin[0] && j >= a.lo[0][1] && j < a.hi[0][1];
        const bool dv_ = dofx[1] && kin[1] && j >= a.lo[1][1] && j < a.hi[1][1];
        const bool dw_ = dofx[2] && kin[2] && j >= a.lo[2][1] && j < a.hi[2][1];
        fu = du_ ? fu : 0.0;
        fv = dv_ ? fv : 0.0;
        fw = dw_ ? fw : 0.0;
        if constexpr (WT) {
          // diffusion!(F, u) alone: the diffusive parts (up - uc)·ν/Δb of the same nine faces, differenced with the same reciprocals
          const double dxu = (Un - Uc) * X.vs, dxv = (Vn - Vc) * X.vo, dxw = (Wn - Wc) * X.vo;
          const double lu = (Uc - sU) * XL.vs, lv = (Vc - sV) * XL.vo, lw = (Wc - sW) * XL.vo;
          const double dzu = (Nx.v[0][rr] - Uc) * Z.vo, dzv = (Nx.v[1][rr] - Vc) * Z.vo, dzw = (Nx.v[2][rr] - Wc) * Z.vs;
          double Du = (dxu - prev_h(dxu, lu)) * X.rs + (dyu - dyu_o) * Y.ro + (dzu - dzprev[0][rr - 1]) * Z.ro;
          double Dv = (dxv - prev_h(dxv, lv)) * X.ro + (dyv - dyv_o) * Y.rs + (dzv - dzprev[1][rr - 1]) * Z.ro;
          double Dw = (dxw - prev_h(dxw, lw)) * X.ro + (dyw - dyw_o) * Y.ro + (dzw - dzprev[2][rr - 1]) * Z.rs;
          dzprev[0][rr - 1] = dzu;
          dzprev[1][rr - 1] = dzv;
          dzprev[2][rr - 1] = dzw;
          const long long pk = (long long)k * sz;
          if (a.epi.wout && xout && jb0 + rr - 1 < n1) {
            double* w = a.epi.wout + pk;
            stb(plane_rsrc(w, ubytes), ocol, orow[rr - 1], du_ ? Uc * Du : 0.0);
            stb(plane_rsrc(w + a.sc, ubytes), ocol, orow[rr - 1], dv_ ? Vc * Dv : 0.0);
            stb(plane_rsrc(w + 2 * a.sc, ubytes), ocol, orow[rr - 1], dw_ ? Wc * Dw : 0.0);
          }
          if (a.epi.extra) {  // the stage force is F + E (closure term: ins_rk_ext.hip); E is zero off the degrees of freedom
            const double* e = a.epi.extra + pk;
            fu += ldb<double>(plane_rsrc(e, ubytes), ocol, orow[rr - 1]);
            fv += ldb<double>(plane_rsrc(e + a.sc, ubytes), ocol, orow[rr - 1]);
            fw += ldb<double>(plane_rsrc(e + 2 * a.sc, ubytes), ocol, orow[rr - 1]);
          }
          if (a.epi.gtemp) {  // gravity!: F[I, gdir] += α2 avg(temp, Δ, I, gdir) on Iu[gdir] (operators.jl:914-931)
            const int gd = a.epi.gdir;
            const int idx = gd == 0 ? min(ci, n0 - 1) + 1 : (gd == 1 ? min(j, n1) : k);  // clamped like the store addresses (masked lanes / rows)
            const bool dof = gd == 0 ? du_ : (gd == 1 ? dv_ : dw_);
            const rsrc_t r0 = plane_rsrc(a.epi.gtemp + pk, ubytes), r1 = plane_rsrc(gd == 2 ? a.epi.gtemp + pk + sz : a.epi.gtemp + pk, ubytes);
            const double t0 = ldb<double>(r0, ocol, orow[rr - 1]);
            const double t1 = ldb<double>(r1, ocol + (gd == 0 ? EB : 0u), orow[rr - 1] + (gd == 1 ? (unsigned)N0 * EB : 0u));
            const double d0 = a.gdx[idx], d1 = a.gdx[idx + 1];
            const double gv = dof ? a.epi.ga2 * ((d1 * t0 + d0 * t1) / (d0 + d1)) : 0.0;
            if (gd == 0) fu += gv;
            if (gd == 1) fv += gv;
            if (gd == 2) fw += gv;
          }
        }
        emit(rr, k, fu, fv, fw, sacc[0][rr - 1], sacc[1][rr - 1], sacc[2][rr - 1]);
      }
      fyu_o = fyu;
      fyv_o = fyv;
      fyw_o = fyw;
      if constexpr (WT) {
        dyu_o = dyu;
        dyv_o = dyv;
        dyw_o = dyw;
      }
      // row rr of plane k is dead: its registers receive plane `kload`
      C.v[0][rr] = ldb<double>(n0r, ucol, urow[rr]);
      C.v[1][rr] = ldb<double>(n1r, ucol, urow[rr]);
      C.v[2][rr] = ldb<double>(n2r, ucol, urow[rr]);
    }
    C.v[0][R + 1] = ldb<double>(n0r, ucol, urow[R + 1]);
    C.v[1][R + 1] = ldb<double>(n1r, ucol, urow[R + 1]);
    C.v[2][R + 1] = ldb<double>(n2r, ucol, urow[R + 1]);
    C.h[0] = ldb<double>(n0r, uhoff, 0);
    C.h[1] = ldb<double>(n1r, uhoff, 0);
    C.h[2] = ldb<double>(n2r, uhoff, 0);
  };

  // Two register planes.  Loads past the chunk re-read plane k1 / p(k1+1) (cache hits) instead of branching.
  PlaneM<R> P0, P1;
  if (!CORR) {
    load_plane(P0, k0 - 1);
    load_plane(P1, k0);
    zflux0(P0, P1, k0 - 1);
    load_plane(P0, min(k0 + 1, k1));
    int k = k0;
    while (true) {
      if (a.bar) __builtin_amdgcn_s_barrier();
      body(P1, P0, k, min(k + 2, k1));
      if (++k >= k1) break;
      if (a.bar) __builtin_amdgcn_s_barrier();
      body(P0, P1, k, min(k + 2, k1));
      if (++k >= k1) break;
    }
  } else {
    // invariant at the top of iteration k: cur = corrected plane k, nxt = RAW plane k+1, Pa = p(k+1), Pb = p(k+2)
    double Pa[R + 3], Pb[R + 3], Ha, Hb;
    load_p(Pa, Ha, k0 - 1);
    load_p(Pb, Hb, k0);
    load_plane(P0, k0 - 1);
    load_plane(P1, k0);
    correct(P0, Pa, Ha, Pb, Hb, k0 - 1);
    load_p(Pa, Ha, k0 + 1);
    correct(P1, Pb, Hb, Pa, Ha, k0);
    zflux0(P0, P1, k0 - 1);
    load_plane(P0, min(k0 + 1, k1));
    load_p(Pb, Hb, min(k0 + 2, k1 + 1));
    int k = k0;
    while (true) {
      if (a.bar) __builtin_amdgcn_s_barrier();
      correct(P0, Pa, Ha, Pb, Hb, k + 1);  // plane k+1 with p(k+1), p(k+2)
      load_p(Pa, Ha, min(k + 3, k1 + 1));
      body(P1, P0, k, min(k + 2, k1));
      if (++k >= k1) break;
      if (a.bar) __builtin_amdgcn_s_barrier();
      correct(P1, Pb, Hb, Pa, Ha, k + 1);
      load_p(Pb, Hb, min(k + 3, k1 + 1));
      body(P0, P1, k, min(k + 2, k1));
      if (++k >= k1) break;
    }
  }
}

template <int R, int XW, int CORR, bool WT = false>
int launchm(FluxMArgs& a, hipStream_t s) {
  a.ntx = cdiv(a.N0 - 2, 64 * XW);
  a.nty = cdiv(a.N1 - 2, (4 / XW) * R);
  a.ntz = cdiv(a.N2 - 2, a.zc);
  const unsigned nb = (unsigned)(8LL * a.ntx * ((a.nty + 7) / 8) * a.ntz);
  hipLaunchKernelGGL((k_flux64m<R, XW, CORR, WT>), dim3(nb), dim3(64, 4, 1), 0, s, a);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

}  // namespace

// 3-D grids the tiled kernels take (ins_fast3d_supported) with rows wide enough for a full wavefront window; corr: additionally every side
// Periodic or Dirichlet (the in-register correction touches degrees of freedom only; ins_corr3_supported checks the sides).
bool ins_flux64m_supported(const ins_grid* G) {
  const GridDev& g = G->g;
  return !ins_opt(OPT_INS_DISABLE_FLUX64M) && g.D == 3 && ins_fast3d_supported(G) && g.N[0] - 2 >= 66 && g.N[1] - 2 >= 8 && g.N[2] - 2 >= 4;
}

// Stage kernel with the RK epilogue; p_padded != nullptr: `u` is the previous stage's uncorrected u* (boundary data applied), corrected in registers.
int ins_k_flux64m(const ins_grid* G, double visc, const double* u, double* k_out, const RkEpi& epi, const double* p_padded, hipStream_t s) {
  const GridDev& g = G->g;
  const bool wt = epi.gtemp || epi.wout || epi.extra;  // extended loop on wall-bounded grids: gravity and the closure force in, u·diffusion(u) out
  if (epi.self_in != 0.0 || epi.ustart_out || (wt && p_padded)) {
    ins_set_error("ins_k_flux64m: epilogue term not supported on stretched / masked grids");
    return INS_ERR_UNSUPPORTED;
  }
  int rc = ins_flux3d_prepare(G, visc, s);
  if (rc) return rc;
  FluxMArgs a;
  memset(&a, 0, sizeof(a));
  a.u = u;
  a.p = p_padded;
  a.F = k_out;
  a.sc = g.sc;
  a.N0 = g.N[0];
  a.N1 = g.N[1];
  a.N2 = g.N[2];
  a.rx = reinterpret_cast<const Rec*>(G->rec_dev);
  a.ry = a.rx + g.N[0];
  a.rz = a.ry + g.N[1];
  for (int d = 0; d < 3; ++d) a.per[d] = g.bc[d][0] == INS_BC_PERIODIC;
  for (int al = 0; al < 3; ++al)
    for (int be = 0; be < 3; ++be) {
      a.lo[al][be] = g.iu_lo[al][be];
      a.hi[al][be] = g.iu_hi[al][be];
    }
  a.epi = epi;
  a.gdx = epi.gtemp ? g.dx[epi.gdir] : nullptr;
  const int n2 = g.N[2] - 2;
  const int zco = (int)ins_opt(OPT_INS_FLUX64M_ZC);
  a.zc = zco > 0 ? zco : (n2 >= 128 ? 32 : (n2 >= 32 ? 8 : 4));
  const int waves_x = cdiv(g.N[0] - 2, 64);
  const int xwo = (int)ins_opt(OPT_INS_FLUX64M_XW);
  const int xw = (xwo == 1 || xwo == 2 || xwo == 4) ? xwo : (waves_x >= 8 ? 2 : (waves_x >= 4 ? 4 : (waves_x >= 2 ? 2 : 1)));
  a.bar = (ins_opt(OPT_INS_FLUX64M_NOBAR) || xw == 4) ? 0 : 1;  // the barrier keeps y-stacked wavefronts on one plane; four side by side share no rows
  const bool corr = p_padded != nullptr;
  const int ro = (int)ins_opt(OPT_INS_FLUX64M_ROWS);
  const int rows = corr ? 2 : ((ro == 2 || ro == 3) ? ro : 3);
#define INS_F64M(RR, CC)                            \
  if (rows == RR) {                                 \
    if (xw == 4) return launchm<RR, 4, CC>(a, s);   \
    if (xw == 2) return launchm<RR, 2, CC>(a, s);   \
    return launchm<RR, 1, CC>(a, s);                \
  }
  if (wt) {
    if (xw == 4) return launchm<2, 4, 0, true>(a, s);
    if (xw == 2) return launchm<2, 2, 0, true>(a, s);
    return launchm<2, 1, 0, true>(a, s);
  }
  if (corr) {
    INS_F64M(2, 3)
  } else {
    INS_F64M(2, 0)
    INS_F64M(3, 0)
  }
#undef INS_F64M
  return INS_ERR_INVALID;
}
