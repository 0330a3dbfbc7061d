// Smagorinsky closure force in one pass (all-periodic uniform 3-D boxes): s = divoftensor(apply_bc_p(smagtensor(u, θ)))
//   smagorinsky_closure  operators.jl:1284-1300  =  smagtensor! :1135-1150  ->  apply_bc_p! (every component)  ->  divoftensor! :1203-1236
// The three-kernel sequence writes the six stress fields and reads them back (12 field streams per stage on top of 3 in, 3 out); here the
// stress lives in registers only.  Register-row formulation of k_gradient_rows (ins_fields.hip): lanes along x, R + 4 rows of the three
// velocity components of THREE planes per work-item, marching through a z-chunk; every volume is addressed through its periodic image, so
// the ghost fill of σ is implicit.  A wavefront window of 64 columns gives the stress on lanes 1..62 and the force on lanes 2..61
// (60 outputs), R + 2 stress rows for R output rows.  The stress of plane m is formed once, when the velocity planes m-1, m, m+1 are in
// registers, and spent at once on the three force planes it enters:
//   s(m-1) += upper-plane terms (σxz, σyz, σzz of m)  -> complete, stored
//   s(m)   += own-plane terms (σxx, σyy, σxy; the x / y differences of σxz, σyz; -σzz/Δz)
//   s(m+1)  = -lower-plane terms (σxz, σyz of m)
// so only 5 R partial sums are carried between planes instead of three planes of stress.  (The z-averages of divoftensor! are linear:
// (c + u + un + cn)/4 - (d + c + dn + cn)/4 is evaluated as (u + un)/4 - (d + dn)/4 — rounding-level differences, inside the 1e-12 operator
// tolerance of tests/test_gpu_fields.py.)
// CORRP: `u` is the uncorrected stage velocity and pI the unpadded pressure of its projection; every plane is corrected as it arrives,
// u = u* - ∇p (applypressure!, operators.jl:225-233), so the extended stage loop needs no gradient-subtract pass (ins_rk_ext.hip).
#include "ins_wave64.h"

namespace {

struct SmagArgs {
  const double* u;
  const double* pI;
  double* s;
  int n[3];
  long long sy, sz, sc;  // padded strides
  double rh[3];          // 1/Δ
  double theta, d2;      // Σ Δα²
  int ntx, nty, nty_l, zc, bar;
  // GEN (non-periodic sides and / or stretched grids): metric tables (padded index), periodicity and the DOF boxes of the three components
  int per[3];
  int gz[3][2];  // what a ghost volume of σ holds at side [d][lo / hi]: 0 its periodic image, 1 the adjacent interior volume (Symmetric), 2 zero (Dirichlet)
  const double *dx[3], *rdx[3], *rdxu[3];
  int lo[3][3], hi[3][3];
};

// Wave shifts without an `old` operand (the lane with no source lane reads zero: lanes 0 / 63 are halo lanes here, their shifted values are
// never used) — update_dpp(old = v, v) costs a register copy per 32-bit half before the DPP move, and this kernel is bound by instruction issue.
template <int CTRL>
__device__ __forceinline__ double shift_dpp(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lnext(double v) { return shift_dpp<0x130>(v); }  // lane l <- l+1
__device__ __forceinline__ double lprev(double v) { return shift_dpp<0x138>(v); }  // lane l <- l-1

constexpr int SF_XO = 60;

// GEN: any mix of Periodic / Dirichlet / Symmetric sides on a stretched grid.  apply_bc_p!(σ) gives a ghost volume the stress of its periodic image
// (boundary_conditions.jl:306-318), of the adjacent interior volume at a Symmetric side (:445-453), and leaves it as allocated — zero — at a Dirichlet
// side (:388), direction by direction.  A wrapped index is a wrapped address as before; a copy comes from the neighbour lane (x), the neighbour row (y)
// or, in z, by spending the first / last plane's stress a second time in the role of the ghost plane below / above; zero is zero.
// STR (stretched grid): gradient and divergence with the metric tables
// (four one-sided differences per off-diagonal entry, as the reference; the row metrics are read once per wavefront: stores to s may alias the
// tables as far as the compiler knows, so it would re-read them every plane); stores masked to the degrees of freedom.
template <int R, bool CORRP, bool GEN = false, bool STR = false>
__global__ __launch_bounds__(256, 2) void k_smagforce(SmagArgs a) {
  int seq = (int)(blockIdx.x >> 3);
  const int tx = seq % a.ntx;
  seq /= a.ntx;
  const int ty = (int)(blockIdx.x & 7) * a.nty_l + seq % a.nty_l;
  if (ty >= a.nty) return;  // whole workgroup
  const int chunk = seq / a.nty_l;
  const int lane = threadIdx.x, wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int n0 = a.n[0], n1 = a.n[1], n2 = a.n[2];
  const int xi = tx * SF_XO - 2 + lane;  // 0-based interior column (before wrapping)
  const int jb = (ty * 4 + wy) * R;      // first output row
  const int k0 = chunk * a.zc, k1 = min(k0 + a.zc, n2);
  if (jb >= n1) {  // wave-uniform: keeps the workgroup's barrier count
    for (int m = k0 - 1; a.bar && m <= k1; ++m) __builtin_amdgcn_s_barrier();
    return;
  }
  const bool xout = lane >= 2 && lane <= SF_XO + 1 && xi < n0;
  auto mod = [](int q, int n) {
    q %= n;
    return q < 0 ? q + n : q;
  };
  // buffer addressing (ins_wave64.h): descriptor = one plane of one component, soffset = row start (scalar), voffset = the lane's column
  constexpr unsigned EB = 8;
  const unsigned pbytes = (unsigned)(a.sz * EB), ppbytes = (unsigned)((long long)n0 * n1 * EB);
  auto mapi = [&](int q, int n, int d) {  // interior index -> the index whose data stands for it (wrap, or clamp into the ghost layer)
    if (!GEN || a.per[d]) return mod(q, n);
    return min(max(q, -1), n);
  };
  const int col = mapi(xi, n0, 0);
  const unsigned colb = (unsigned)(col + 1) * EB, pcol = (unsigned)col * EB;
  unsigned rowb[R + 4];              // velocity row rr = interior row jb - 2 + rr
  unsigned prow[CORRP ? R + 5 : 1];  // pressure rows jb - 2 .. jb + R + 2 (unpadded array)
#pragma unroll
  for (int rr = 0; rr < R + 4; ++rr) rowb[rr] = (unsigned)((mapi(jb - 2 + rr, n1, 1) + 1) * a.sy) * EB;
  if constexpr (CORRP) {
#pragma unroll
    for (int rr = 0; rr < R + 5; ++rr) prow[rr] = (unsigned)(mod(jb - 2 + rr, n1) * n0) * EB;
  }
  // Plane slots: the upper plane's z component enters no gradient, so it is loaded one plane later, when the plane has become the middle
  // one — two of the three z slots are live at any time.
  double PX[3][R + 4], PY[3][R + 4], PZ[3][R + 4];
  double pk[CORRP ? R + 4 : 1];  // CORRP: p of the plane loaded last
  if constexpr (CORRP) {
#pragma unroll
    for (int rr = 0; rr < R + 4; ++rr) pk[rr] = 0.0;  // (only the unused first z load sees it)
  }
  // x, y of plane kk into (QX, QY); z of plane kk - 1 into QZ (the slot of plane kk - 1)
  auto load_plane = [&](double (&QX)[R + 4], double (&QY)[R + 4], double (&QZ)[R + 4], int kk) {
    const double* bxy = a.u + (long long)(mapi(kk, n2, 2) + 1) * a.sz;
    const double* bz = a.u + 2 * a.sc + (long long)(mapi(kk - 1, n2, 2) + 1) * a.sz;
    const rsrc_t rx = plane_rsrc(bxy, pbytes), ry = plane_rsrc(bxy + a.sc, pbytes), rz = plane_rsrc(bz, pbytes);
#pragma unroll
    for (int rr = 0; rr < R + 4; ++rr) {
      QX[rr] = ldb<double>(rx, colb, rowb[rr]);
      QY[rr] = ldb<double>(ry, colb, rowb[rr]);
      QZ[rr] = ldb<double>(rz, colb, rowb[rr]);
    }
    if constexpr (CORRP) {  // u = u* - ∇p: x, y of plane kk with p(kk); z of plane kk - 1 with p(kk) and the kept p(kk - 1)
      const rsrc_t rp = plane_rsrc(a.pI + (long long)mod(kk, n2) * n0 * n1, ppbytes);
      double pn[R + 5];
#pragma unroll
      for (int rr = 0; rr < R + 5; ++rr) pn[rr] = ldb<double>(rp, pcol, prow[rr]);
#pragma unroll
      for (int rr = 0; rr < R + 4; ++rr) {
        QX[rr] -= (lnext(pn[rr]) - pn[rr]) * a.rh[0];  // lane 63 stays uncorrected: no stress on lanes 1..62 reads its x component
        QY[rr] -= (pn[rr + 1] - pn[rr]) * a.rh[1];
        QZ[rr] -= (pn[rr] - pk[rr]) * a.rh[2];
        pk[rr] = pn[rr];
      }
    }
  };
  double Ep[3][R], En[2][R];  // partial sums of plane m (Ep) and m + 1 (En) when plane m's stress has been spent
#pragma unroll
  for (int q = 0; q < R; ++q) {
    Ep[0][q] = Ep[1][q] = Ep[2][q] = 0.0;
    En[0][q] = En[1][q] = 0.0;
  }
  const double rq[3] = {a.rh[0] / 4, a.rh[1] / 4, a.rh[2] / 4};
  const double nu0 = a.theta * a.theta * a.d2;
  // GEN: per-lane x metrics at the lane's (mapped) column, kept inside the interior index range (ghost lanes never use theirs)
  const int Ix = STR ? min(max(col + 1, 1), n0) : 1;
  const double gx_d = STR ? a.rdx[0][Ix] : 0.0, gx_u1 = STR ? a.rdxu[0][Ix] : 0.0, gx_u0 = STR ? a.rdxu[0][Ix - 1] : 0.0, gx_w = STR ? a.dx[0][Ix] : 0.0;
  // STR: y metrics of the R + 2 stress rows and the R output rows (wave-uniform)
  double gy_d[STR ? R + 2 : 1], gy_u1[STR ? R + 2 : 1], gy_u0[STR ? R + 2 : 1], gy_w[STR ? R + 2 : 1], oy_q[STR ? R : 1], oy_u[STR ? R : 1];
  if constexpr (STR) {
#pragma unroll
    for (int q = 0; q < R + 2; ++q) {
      const int Jy = min(max(mapi(jb - 1 + q, n1, 1) + 1, 1), n1);
      gy_d[q] = a.rdx[1][Jy];
      gy_u1[q] = a.rdxu[1][Jy] / 4;
      gy_u0[q] = a.rdxu[1][Jy - 1] / 4;
      gy_w[q] = a.dx[1][Jy];
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int Jo = min(jb + q + 1, n1);  // padded output row
      oy_q[q] = a.rdx[1][Jo] / 4;
      oy_u[q] = a.rdxu[1][Jo];
    }
  }
  const bool ghl = GEN && !a.per[0] && xi == -1, ghr = GEN && !a.per[0] && xi == n0;  // this lane holds a ghost column at a non-periodic side
  const bool xz0 = GEN && ((ghl && a.gz[0][0] == 2) || (ghr && a.gz[0][1] == 2));   // ... whose stress is zero
  bool dofx[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) dofx[c] = !GEN || (xi + 1 >= a.lo[c][0] && xi + 1 < a.hi[c][0]);
  struct Sig {
    double xx, yy, zz, xy, xz, yz;
  };
  // planes: M = m - 1, C = m, N = m + 1 (no z component)
  auto process = [&](const double (&MX)[R + 4], const double (&MY)[R + 4], const double (&MZ)[R + 4], const double (&CX)[R + 4],
                     const double (&CY)[R + 4], const double (&CZ)[R + 4], const double (&NX)[R + 4], const double (&NY)[R + 4], int m) {
    const int Kz = STR ? min(max(mapi(m, n2, 2) + 1, 1), n2) : 1;
    const double gz_d = STR ? a.rdx[2][Kz] : 0.0, gz_u1 = STR ? a.rdxu[2][Kz] : 0.0, gz_u0 = STR ? a.rdxu[2][Kz - 1] : 0.0, gz_w = STR ? a.dx[2][Kz] : 0.0;
    auto sigma = [&](int q) {  // stress row q = interior row jb - 1 + q = velocity row q + 1
      const int uc = q + 1;
      auto U = [&](int c, int ox, int oy, int oz) {
        double v;
        if (c == 0) v = oz < 0 ? MX[uc + oy] : (oz > 0 ? NX[uc + oy] : CX[uc + oy]);
        if (c == 1) v = oz < 0 ? MY[uc + oy] : (oz > 0 ? NY[uc + oy] : CY[uc + oy]);
        if (c == 2) v = oz < 0 ? MZ[uc + oy] : CZ[uc + oy];  // oz > 0 never asked of the z component
        return ox < 0 ? lprev(v) : (ox > 0 ? lnext(v) : v);
      };
      auto at = [&](int c, int da, int sa_, int db, int sb_) {  // offsets sa_·e_da + sb_·e_db
        const int ox = (da == 0 ? sa_ : 0) + (db == 0 ? sb_ : 0);
        const int oy = (da == 1 ? sa_ : 0) + (db == 1 ? sb_ : 0);
        const int oz = (da == 2 ? sa_ : 0) + (db == 2 ? sb_ : 0);
        return U(c, ox, oy, oz);
      };
      // ∇(u, I, Δ, Δu)   operators.jl:1023-1034, 1069-1085.  On a uniform box the four one-sided differences of an off-diagonal entry share
      // their metric, so the middle values cancel: two central differences (the kernel is bound by its instruction count, not by HBM).
      double G[3][3];
      double d2 = a.d2;
      if constexpr (STR) {
        const double gd[3] = {gx_d, gy_d[q], gz_d}, g1[3] = {gx_u1 / 4, gy_u1[q], gz_u1 / 4}, g0[3] = {gx_u0 / 4, gy_u0[q], gz_u0 / 4};  // (gy_u*: quarters already)
        const double wy_ = gy_w[q];
        d2 = gx_w * gx_w + wy_ * wy_ + gz_w * gz_w;  // gridsize² = Σ Δα²
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            if (c == b)
              G[c][b] = (at(c, c, 0, b, 0) - at(c, c, 0, b, -1)) * gd[b];
            else {  // the reference's four one-sided differences, the two that share a metric taken together: ((s+ - s0) r1 + (s0 - s-) r0) / 4
              const double s0 = at(c, c, 0, b, 0) + at(c, c, -1, b, 0);
              const double sp = at(c, c, 0, b, 1) + at(c, c, -1, b, 1), sm = at(c, c, 0, b, -1) + at(c, c, -1, b, -1);
              G[c][b] = (sp - s0) * g1[b] + (s0 - sm) * g0[b];
            }
          }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            if (c == b)
              G[c][b] = (at(c, c, 0, b, 0) - at(c, c, 0, b, -1)) * a.rh[b];
            else
              G[c][b] = ((at(c, c, 0, b, 1) - at(c, c, 0, b, -1)) + (at(c, c, -1, b, 1) - at(c, c, -1, b, -1))) * rq[b];
          }
      }
      const double sxy = (G[0][1] + G[1][0]) / 2, sxz = (G[0][2] + G[2][0]) / 2, syz = (G[1][2] + G[2][1]) / 2;
      const double ss = G[0][0] * G[0][0] + G[1][1] * G[1][1] + G[2][2] * G[2][2] + 2 * (sxy * sxy + sxz * sxz + syz * syz);  // Σ S_ab S_ab
      const double eddy = (STR ? a.theta * a.theta * d2 : nu0) * sqrt(2 * ss);  // smagtensor!: νt = θ² Δ² sqrt(2 S:S)
      Sig o;
      o.xx = 2 * eddy * G[0][0];
      o.yy = 2 * eddy * G[1][1];
      o.zz = 2 * eddy * G[2][2];
      o.xy = 2 * eddy * sxy;
      o.xz = 2 * eddy * sxz;
      o.yz = 2 * eddy * syz;
      if constexpr (GEN) {
        if (!a.per[0]) {  // a ghost column carries the stress of the adjacent interior column (Symmetric) or none (Dirichlet)
          auto fix = [&](double v) {  // (the wave shifts run in every lane: a DPP read of a lane masked off by a branch returns nothing)
            const double vn = lnext(v), vp = lprev(v);
            return xz0 ? 0.0 : (ghl ? vn : (ghr ? vp : v));
          };
          // only σxy and σxz are read from a ghost column by a degree of freedom (the y / z force of the first and last column); σxx of a
          // ghost column would enter the x force of the wall face, which is no degree of freedom
          o.xy = fix(o.xy);
          o.xz = fix(o.xz);
        }
      }
      return o;
    };
    if constexpr (GEN) {
      if (!a.per[2] && (m < 0 || m >= n2)) return;  // ghost planes at walls: their stress is the first / last plane's, spent below
    }
    const bool zfirst = GEN && !a.per[2] && m == 0, zlast = GEN && !a.per[2] && m == n2 - 1;
    const double zf_c = (GEN && a.gz[2][0] == 2) ? 0.0 : 1.0, zl_c = (GEN && a.gz[2][1] == 2) ? 0.0 : 1.0;  // Dirichlet: the ghost plane's stress is zero
    // z metrics by role (padded indices): plane m - 1 is output plane index m, plane m is m + 1, plane m + 1 is m + 2
    const int Km = min(max(m, 0), n2 + 1), Kc = min(max(m + 1, 0), n2 + 1), Kp = min(max(m + 2, 0), n2 + 1);
    const double rzm = STR ? a.rdx[2][Km] / 4 : rq[2], rzc = STR ? a.rdx[2][Kc] / 4 : rq[2], rzp = STR ? a.rdx[2][Kp] / 4 : rq[2];
    const double zum = STR ? a.rdxu[2][Km] : a.rh[2], zuc = STR ? a.rdxu[2][Kc] : a.rh[2];
    bool dofzm[3], dofzc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dofzm[c] = !GEN || (m >= a.lo[c][2] && m < a.hi[c][2]);
      dofzc[c] = !GEN || (m + 1 >= a.lo[c][2] && m + 1 < a.hi[c][2]);
    }
    const bool store = m - 1 >= k0;  // plane m - 1 is complete now
    const double* spl = a.s + (long long)m * a.sz;  // padded plane index of interior plane m - 1
    const rsrc_t o0 = plane_rsrc(spl, pbytes), o1 = plane_rsrc(spl + a.sc, pbytes), o2 = plane_rsrc(spl + 2 * a.sc, pbytes);
    const double* spc = a.s + (long long)(m + 1) * a.sz;  // padded plane index of interior plane m (stored here only at a wall above)
    const rsrc_t c0 = plane_rsrc(spc, pbytes), c1 = plane_rsrc(spc + a.sc, pbytes), c2 = plane_rsrc(spc + 2 * a.sc, pbytes);
    // rows in rolling order: output row q is emitted when stress row q + 1 exists
    const Sig zero{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    Sig lo = sigma(0), mid = sigma(1);
    if constexpr (GEN) {
      if (!a.per[1] && jb == 0) lo = a.gz[1][0] == 2 ? zero : mid;  // ghost row below the first row
    }
#pragma unroll
    for (int q = 1; q <= R; ++q) {
      Sig hi = sigma(q + 1);
      const int row = jb + q - 1;
      if constexpr (GEN) {
        if (!a.per[1] && row + 1 == n1) hi = a.gz[1][1] == 2 ? zero : mid;  // ghost row above the last row
      }
      const double ryq = STR ? oy_q[q - 1] : rq[1], ryu = STR ? oy_u[q - 1] : a.rh[1];
      const double rxq = STR ? gx_d / 4 : rq[0], rxu = STR ? gx_u1 : a.rh[0];
      const double xz_n = lnext(mid.xz), xz_p = lprev(mid.xz);
      const double upr_x = mid.xz + xz_n, upr_y = mid.yz + hi.yz;  // (times a z metric by role)
      const double cz = (xz_n - xz_p) * rxq + (hi.yz - lo.yz) * ryq;
      const double own_x = (lnext(mid.xx) - mid.xx) * rxu + ((hi.xy + lnext(hi.xy)) - (lo.xy + lnext(lo.xy))) * ryq;
      const double own_y = ((lnext(mid.xy) + lnext(hi.xy)) - (lprev(mid.xy) + lprev(hi.xy))) * rxq + (hi.yy - mid.yy) * ryu;
      bool dofy[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) dofy[c] = !GEN || (row + 1 >= a.lo[c][1] && row + 1 < a.hi[c][1]);
      if (store && xout && row < n1) {
        if (dofx[0] && dofy[0] && dofzm[0]) stb(o0, colb, rowb[q + 1], Ep[0][q - 1] + upr_x * rzm);
        if (dofx[1] && dofy[1] && dofzm[1]) stb(o1, colb, rowb[q + 1], Ep[1][q - 1] + upr_y * rzm);
        if (dofx[2] && dofy[2] && dofzm[2]) stb(o2, colb, rowb[q + 1], Ep[2][q - 1] + (cz + mid.zz * zum));
      }
      Ep[0][q - 1] = (zfirst ? -(upr_x * rzc) * zf_c : En[0][q - 1]) + own_x;  // zfirst: the ghost plane below carries this plane's stress, or none
      Ep[1][q - 1] = (zfirst ? -(upr_y * rzc) * zf_c : En[1][q - 1]) + own_y;
      Ep[2][q - 1] = cz - mid.zz * zuc;
      En[0][q - 1] = -(upr_x * rzp);
      En[1][q - 1] = -(upr_y * rzp);
      if constexpr (GEN) {
        if (zlast && xout && row < n1) {  // the ghost plane above carries this plane's stress, or none: plane m is complete as well
          if (dofx[0] && dofy[0] && dofzc[0]) stb(c0, colb, rowb[q + 1], Ep[0][q - 1] + upr_x * rzc * zl_c);
          if (dofx[1] && dofy[1] && dofzc[1]) stb(c1, colb, rowb[q + 1], Ep[1][q - 1] + upr_y * rzc * zl_c);
          if (dofx[2] && dofy[2] && dofzc[2]) stb(c2, colb, rowb[q + 1], Ep[2][q - 1] + (cz + mid.zz * zuc) * zl_c);
        }
      }
      lo = mid;
      mid = hi;
    }
  };
  // before stress plane m: x, y of planes m - 1 and m; z of plane m - 1.  Each step loads x, y of plane m + 1 and z of plane m.
  load_plane(PX[0], PY[0], PZ[2], k0 - 2);  // (z of plane k0 - 3 into a free slot: not used)
  load_plane(PX[1], PY[1], PZ[0], k0 - 1);
  int m = k0 - 1;
  while (true) {  // rotation unrolled so that every register index is static
    if (a.bar) __builtin_amdgcn_s_barrier();  // (experiment switch INS_SMAGFORCE_BAR: the y-stacked wavefronts kept on one plane)
    load_plane(PX[2], PY[2], PZ[1], m + 1);
    process(PX[0], PY[0], PZ[0], PX[1], PY[1], PZ[1], PX[2], PY[2], m);
    if (++m > k1) break;
    if (a.bar) __builtin_amdgcn_s_barrier();
    load_plane(PX[0], PY[0], PZ[2], m + 1);
    process(PX[1], PY[1], PZ[1], PX[2], PY[2], PZ[2], PX[0], PY[0], m);
    if (++m > k1) break;
    if (a.bar) __builtin_amdgcn_s_barrier();
    load_plane(PX[1], PY[1], PZ[0], m + 1);
    process(PX[2], PY[2], PZ[2], PX[0], PY[0], PZ[0], PX[1], PY[1], m);
    if (++m > k1) break;
  }
}

}  // namespace

// all-periodic uniform boxes: the specialised form (central differences, no masks, on-the-fly correction available)
// (z-slab grids — INS_BC_HALO sides — are all_dof and uniform_exact too, but their z neighbours are exchanged ghost planes, not images inside the slab)
static bool smagforce_uniform(const ins_grid* G) { return G->all_periodic && G->all_dof && G->uniform_exact; }
static bool has_halo_side(const ins_grid* G) {
  for (int d = 0; d < G->g.D; ++d)
    for (int side = 0; side < 2; ++side)
      if (G->g.bc[d][side] == INS_BC_HALO) return true;
  return false;
}

bool ins_smagforce_supported(const ins_grid* G) {
  const GridDev& g = G->g;
  if (ins_opt(OPT_INS_DISABLE_SMAGFORCE) || g.D != 3 || g.N[0] - 2 < 4 || g.N[1] - 2 < 4 || g.N[2] - 2 < 4) return false;
  if (smagforce_uniform(G)) return true;
  if (ins_opt(OPT_INS_DISABLE_SMAGFORCE_GEN)) return false;
  for (int d = 0; d < 3; ++d)
    for (int side = 0; side < 2; ++side)
      // a PressureBC on the RIGHT side is a zero ghost stress (apply_bc_p!: p[I] .= 0, boundary_conditions.jl:497-501) with one more degree of freedom of the
      // normal component (the masks come from the grid's Iu); on the LEFT side it adds a second ghost layer (N = n + 3), which this kernel's indexing does
      // not cover: the three kernels.  Slab grids: refused by the entry points.
      if ((g.bc[d][side] == INS_BC_PRESSURE && side == 0) || g.bc[d][side] == INS_BC_HALO) return false;
  return true;
}
// the correcting form (uncorrected input + pressure) exists for all-periodic uniform boxes only
bool ins_smagforce_corr_supported(const ins_grid* G) { return ins_smagforce_supported(G) && smagforce_uniform(G); }

// s (degrees of freedom of the three components; everything else untouched) = closure force of u.  pI == nullptr: u is a velocity field with its
// boundary data applied (periodic directions: ghost volumes not read); else (all-periodic uniform boxes) u is an uncorrected stage velocity and
// pI the (unpadded) pressure of its projection.
int ins_k_smagforce(const ins_grid* G, double theta, const double* u, const double* pI, double* sout, hipStream_t s) {
  const GridDev& g = G->g;
  const int force = (int)ins_opt(OPT_INS_SMAGFORCE_FORCE_GEN);  // experiment: the generalised forms on a periodic uniform box (1: uniform, 2: stretched)
  const bool gen = !smagforce_uniform(G) || (force && !pI);
  if (!ins_smagforce_supported(G) || (gen && pI)) {
    ins_set_error("ins_k_smagforce: grid not supported");
    return INS_ERR_UNSUPPORTED;
  }
  constexpr int R = 2;  // 3 rows: 76 bytes of scratch per lane, 0.40 ms against 0.27 at 256^3
  SmagArgs a;
  memset(&a, 0, sizeof(a));
  a.u = u;
  a.pI = pI;
  a.s = sout;
  a.d2 = 0.0;
  for (int b = 0; b < 3; ++b) {
    a.n[b] = g.N[b] - 2;
    a.rh[b] = 1.0 / G->h[b];
    a.d2 += G->h[b] * G->h[b];
    a.per[b] = g.bc[b][0] == INS_BC_PERIODIC;
    for (int side = 0; side < 2; ++side) a.gz[b][side] = a.per[b] ? 0 : (g.bc[b][side] == INS_BC_SYMMETRIC ? 1 : 2);
    a.dx[b] = g.dx[b];
    a.rdx[b] = g.rdx[b];
    a.rdxu[b] = g.rdxu[b];
    for (int d = 0; d < 3; ++d) {
      a.lo[b][d] = g.iu_lo[b][d];
      a.hi[b][d] = g.iu_hi[b][d];
    }
  }
  a.sy = g.sx[1];
  a.sz = g.sx[2];
  a.sc = g.sc;
  a.theta = theta;
  a.bar = ins_opt(OPT_INS_SMAGFORCE_BAR) ? 1 : 0;  // 256^3: 0.274 ms without, 0.304 with (the wavefronts wait on their loads either way: 2 per SIMD)
  a.ntx = (int)cdiv(a.n[0], SF_XO);
  a.nty = (int)cdiv(a.n[1], 4 * R);
  a.nty_l = (a.nty + 7) / 8;
  int zc = ins_opt(OPT_INS_SMAGFORCE_ZC) > 0 ? ins_opt(OPT_INS_SMAGFORCE_ZC) : 32;
  while (zc > 4 && (long long)a.ntx * a.nty * cdiv(a.n[2], zc) < 1024) zc >>= 1;
  a.zc = zc;
  const unsigned nb = 8u * a.ntx * a.nty_l * (unsigned)cdiv(a.n[2], zc);
  if (gen && (!G->uniform_exact || force == 2))  // (uniform_exact: every width and every centre distance incl. the ghost layers equal to 2.5e-13)
    hipLaunchKernelGGL((k_smagforce<R, false, true, true>), dim3(nb), dim3(64, 4, 1), 0, s, a);
  else if (gen)  // uniform spacing with walls: the central-difference form, ghost rules and masks only
    hipLaunchKernelGGL((k_smagforce<R, false, true, false>), dim3(nb), dim3(64, 4, 1), 0, s, a);
  else if (pI)
    hipLaunchKernelGGL((k_smagforce<R, true>), dim3(nb), dim3(64, 4, 1), 0, s, a);
  else
    hipLaunchKernelGGL((k_smagforce<R, false>), dim3(nb), dim3(64, 4, 1), 0, s, a);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// smagorinsky_closure(setup)(u, θ)   operators.jl:1284-1300 as one kernel where the box allows it, else the reference's three steps
// (sigma: scratch of D(D+1)/2 scalar fields, used by the three-step route only)
int ins_k_apply_bc_p_fields(const ins_grid* G, double* p, int nf, hipStream_t s);
extern "C" int ins_smagorinsky_force_needs_sigma(const ins_grid_t* G) { return (G && ins_smagforce_supported(G)) ? 0 : 1; }
extern "C" int ins_smagorinsky_force_f64(const ins_grid_t* G, double theta, const double* u, double* sigma, double* s, void* stream) {
  INS_REQUIRE(G && u && s, "null argument");
  if (has_halo_side(G)) {  // the stress tensor's ghost planes would have to travel between the ranks inside this call
    ins_set_error("ins_smagorinsky_force_f64: z-slab grids (INS_BC_HALO sides) are not supported");
    return INS_ERR_UNSUPPORTED;
  }
  if (ins_smagforce_supported(G)) return ins_k_smagforce(G, theta, u, nullptr, s, as_stream(stream));
  INS_REQUIRE(sigma, "sigma scratch needed on this grid");
  const int D = G->g.D;
  int rc;
  if ((rc = ins_smagtensor_f64(G, theta, u, sigma, stream))) return rc;
  if ((rc = ins_k_apply_bc_p_fields(G, sigma, D * (D + 1) / 2, as_stream(stream)))) return rc;
  return ins_divoftensor_f64(G, sigma, s, stream);
}
