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
};

__device__ __forceinline__ double lnext(double v) { return next_h(v, v); }  // lane l <- l+1
__device__ __forceinline__ double lprev(double v) { return prev_h(v, v); }  // lane l <- l-1

constexpr int SF_XO = 60;

template <int R, bool CORRP>
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
  const int col = mod(xi, n0);
  const unsigned colb = (unsigned)(col + 1) * EB, pcol = (unsigned)col * EB;
  unsigned rowb[R + 4];              // velocity row rr = interior row jb - 2 + rr
  unsigned prow[CORRP ? R + 5 : 1];  // pressure rows jb - 2 .. jb + R + 2 (unpadded array)
#pragma unroll
  for (int rr = 0; rr < R + 4; ++rr) rowb[rr] = (unsigned)((mod(jb - 2 + rr, n1) + 1) * a.sy) * EB;
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
    const double* bxy = a.u + (long long)(mod(kk, n2) + 1) * a.sz;
    const double* bz = a.u + 2 * a.sc + (long long)(mod(kk - 1, n2) + 1) * a.sz;
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
  struct Sig {
    double xx, yy, zz, xy, xz, yz;
  };
  // planes: M = m - 1, C = m, N = m + 1 (no z component)
  auto process = [&](const double (&MX)[R + 4], const double (&MY)[R + 4], const double (&MZ)[R + 4], const double (&CX)[R + 4],
                     const double (&CY)[R + 4], const double (&CZ)[R + 4], const double (&NX)[R + 4], const double (&NY)[R + 4], int m) {
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
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          if (c == b)
            G[c][b] = (at(c, c, 0, b, 0) - at(c, c, 0, b, -1)) * a.rh[b];
          else
            G[c][b] = ((at(c, c, 0, b, 1) - at(c, c, 0, b, -1)) + (at(c, c, -1, b, 1) - at(c, c, -1, b, -1))) * rq[b];
        }
      const double sxy = (G[0][1] + G[1][0]) / 2, sxz = (G[0][2] + G[2][0]) / 2, syz = (G[1][2] + G[2][1]) / 2;
      const double ss = G[0][0] * G[0][0] + G[1][1] * G[1][1] + G[2][2] * G[2][2] + 2 * (sxy * sxy + sxz * sxz + syz * syz);  // Σ S_ab S_ab
      const double eddy = nu0 * sqrt(2 * ss);  // smagtensor!: νt = θ² Δ² sqrt(2 S:S)
      Sig o;
      o.xx = 2 * eddy * G[0][0];
      o.yy = 2 * eddy * G[1][1];
      o.zz = 2 * eddy * G[2][2];
      o.xy = 2 * eddy * sxy;
      o.xz = 2 * eddy * sxz;
      o.yz = 2 * eddy * syz;
      return o;
    };
    const bool store = m - 1 >= k0;  // plane m - 1 is complete now
    const double* spl = a.s + (long long)m * a.sz;  // padded plane index of interior plane m - 1
    const rsrc_t o0 = plane_rsrc(spl, pbytes), o1 = plane_rsrc(spl + a.sc, pbytes), o2 = plane_rsrc(spl + 2 * a.sc, pbytes);
    // rows in rolling order: output row q is emitted when stress row q + 1 exists
    Sig lo = sigma(0), mid = sigma(1);
#pragma unroll
    for (int q = 1; q <= R; ++q) {
      const Sig hi = sigma(q + 1);
      const double xz_n = lnext(mid.xz), xz_p = lprev(mid.xz);
      const double up_x = (mid.xz + xz_n) * rq[2];
      const double up_y = (mid.yz + hi.yz) * rq[2];
      const double cz = (xz_n - xz_p) * rq[0] + (hi.yz - lo.yz) * rq[1];
      const double zt = mid.zz * a.rh[2];
      const double own_x = (lnext(mid.xx) - mid.xx) * a.rh[0] + ((hi.xy + lnext(hi.xy)) - (lo.xy + lnext(lo.xy))) * rq[1];
      const double own_y = ((lnext(mid.xy) + lnext(hi.xy)) - (lprev(mid.xy) + lprev(hi.xy))) * rq[0] + (hi.yy - mid.yy) * a.rh[1];
      const int row = jb + q - 1;
      if (store && xout && row < n1) {
        stb(o0, colb, rowb[q + 1], Ep[0][q - 1] + up_x);
        stb(o1, colb, rowb[q + 1], Ep[1][q - 1] + up_y);
        stb(o2, colb, rowb[q + 1], Ep[2][q - 1] + (cz + zt));
      }
      Ep[0][q - 1] = En[0][q - 1] + own_x;
      Ep[1][q - 1] = En[1][q - 1] + own_y;
      Ep[2][q - 1] = cz - zt;
      En[0][q - 1] = -up_x;
      En[1][q - 1] = -up_y;
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

bool ins_smagforce_supported(const ins_grid* G) {
  const GridDev& g = G->g;
  return !ins_opt(OPT_INS_DISABLE_SMAGFORCE) && g.D == 3 && G->all_dof && G->uniform_exact && g.N[0] - 2 >= 4 && g.N[1] - 2 >= 4 && g.N[2] - 2 >= 4;
}

// s (interior volumes of the three components; ghost volumes untouched) = closure force of u.  pI == nullptr: u is a velocity field (its ghost
// volumes are not read); else u is an uncorrected stage velocity and pI the (unpadded) pressure of its projection.
int ins_k_smagforce(const ins_grid* G, double theta, const double* u, const double* pI, double* sout, hipStream_t s) {
  const GridDev& g = G->g;
  if (!ins_smagforce_supported(G)) {
    ins_set_error("ins_k_smagforce: all-periodic uniform 3-D boxes only");
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
  if (pI)
    hipLaunchKernelGGL((k_smagforce<2, true>), dim3(nb), dim3(64, 4, 1), 0, s, a);
  else
    hipLaunchKernelGGL((k_smagforce<2, false>), dim3(nb), dim3(64, 4, 1), 0, s, a);
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// smagorinsky_closure(setup)(u, θ)   operators.jl:1284-1300 as one kernel where the box allows it, else the reference's three steps
// (sigma: scratch of D(D+1)/2 scalar fields, used by the three-step route only)
int ins_k_apply_bc_p_fields(const ins_grid* G, double* p, int nf, hipStream_t s);
extern "C" int ins_smagorinsky_force_f64(const ins_grid_t* G, double theta, const double* u, double* sigma, double* s, void* stream) {
  INS_REQUIRE(G && u && s, "null argument");
  if (ins_smagforce_supported(G)) return ins_k_smagforce(G, theta, u, nullptr, s, as_stream(stream));
  INS_REQUIRE(sigma, "sigma scratch needed on this grid");
  const int D = G->g.D;
  int rc;
  if ((rc = ins_smagtensor_f64(G, theta, u, sigma, stream))) return rc;
  if ((rc = ins_k_apply_bc_p_fields(G, sigma, D * (D + 1) / 2, as_stream(stream)))) return rc;
  return ins_divoftensor_f64(G, sigma, s, stream);
}
