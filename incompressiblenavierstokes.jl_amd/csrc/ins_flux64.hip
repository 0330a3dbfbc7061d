// K1 (flux form, 64 outputs per wavefront) — the periodic / exactly-uniform specialisation of ins_fast3d_flux.hip
// (convection_diffusion_kernel! + fill!(F, 0), operators.jl:647-690, 971; optional RK epilogue and in-register
// pressure correction exactly as documented there).
//
// ins_fast3d_flux.hip spends lanes 0 and 63 of every wavefront on halo columns, so a 256-wide row needs 5 wavefronts
// (62+62+62+62+8) and a 512-wide row 9.  Here all 64 lanes produce output:
//   * the two halo columns (x0-1 and x0+64) of ALL R+2 rows arrive in ONE extra load per component and plane, rows
//     packed across lanes (lanes 0..7: left column, lanes 16..23: right column);
//   * `v_readlane` moves a packed halo value to an SGPR; it enters the wave shift as the DPP `old` operand, i.e. the value
//     lane 63 (wave_shl) or lane 0 (wave_shr) keeps when it has no source lane;
//   * lane 0 has no left neighbour to take the left-face x-fluxes from: they are evaluated from the halo scalars (three
//     more flux evaluations per row, wave-wide; the kernel is memory-bound);
//   * with the pressure correction (CORR) the packed halo columns are corrected in packed form from a packed load of the
//     four pressure columns x0-1, x0, x0+64, x0+65 (DPP row shifts inside the 16-lane rows).
// So every column of u is loaded once per (R+2)-row window and a 256 (512) wide row takes exactly 4 (8) wavefronts.
// Everything else (R+2 register rows, z-march with carried z-flux, 3-buffer plane prefetch, k-major XCD-partitioned tile
// order) is the scheme of ins_fast3d_flux.hip.  Addressing: wave-uniform row bases (SGPR) + one 32-bit lane offset, so the
// loads use the saddr form and no 64-bit address registers; the grid enters as 12 scalars instead of the GridDev tables.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "ins_internal.h"
#include "ins_wave64.h"
#include "ins_flux_common.h"

namespace {

template <typename T, int R>
struct Plane {
  T v[3][R + 2];
  T h[3];  // packed halo columns: lane r = row r of column x0-1, lane 16+r = row r of column x0+64
  T raw[3][R];  // CORR: the output rows before the pressure correction (a term of the stage-velocity basis, ins_rk.hip)
};

// temperature rows of a register plane (EXTRA, a.tm): T rides along as a fourth component; vm1 = the y-component at the row BELOW the halo
// row (the Laplacian of v at the halo row needs it: dissipation! reads u·diffusion(u) of the volume below)
template <typename T, int R>
struct TExt {
  T t[R + 2];
  T th;  // packed halo columns of T
  T vm1;
};

// NW wavefronts per workgroup: XW side by side in x, NW/XW stacked in y.  CORR as in ins_fast3d_flux.hip: 0 = `u` has valid ghost volumes;
// 1 = `u` is the previous stage's uncorrected u* (interior only), pI its unpadded pressure, every neighbour through the
// periodic image; 2 = z-slab: x, y periodic images, z through exchanged ghost planes, pI = [1 | nzl | 2] extended buffer.
// EXTRA (extended stage loop, ins_rk_ext.hip): epi.extra is a vector field added to the stage force before it is used and stored
// (k_i = F_i + closure(u_i) + gravity(temp_i), step_explicit_runge_kutta.jl:21-34).
template <typename T, int R, int XW, bool FUSE, int CORR, bool SKEL = false, int NW = 4, bool EXTRA = false>
__global__ __launch_bounds__(64 * NW, NW == 16 ? 4 : 2) void k_flux64(FluxArgs a) {
  constexpr unsigned EB = (unsigned)sizeof(T);  // element bytes
  const T* const a_u = static_cast<const T*>(a.u);
  const T* const a_pI = static_cast<const T*>(a.pI);
  T* const a_F = static_cast<T*>(a.F);
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
  const int x0 = (txi * XW + wx) * 64;  // interior (0-based) column of lane 0
  const int jb0 = (tyi * (NW / XW) + wy) * R;  // interior row of the first output row
  const int k0 = a.kB ? (tzi ? a.kB : a.k_lo) : a.k_lo + tzi * a.zc;  // padded plane index of the first output plane
  const int k1 = a.kB ? k0 + a.zc : min(k0 + a.zc, a.k_hi);
  if (x0 >= n0 || jb0 >= n1) {  // wavefront outside the box: it only keeps the workgroup's barrier count (one per plane)
    if (a.bar)
      for (int k = k0; k < k1; ++k) __builtin_amdgcn_s_barrier();
    return;
  }
  const long long sz = (long long)N0 * N1;
  const int ci = x0 + lane;
  const bool xout = ci < n0;
  const DirT<T> X(a.X), Y(a.Y), Z(a.Z);

  // padded row / column of a (possibly out-of-range) interior index
  auto prow_of = [&](int jr) { return CORR ? wrapi(jr, n1) + 1 : min(jr + 1, N1 - 1); };
  auto pcol_of = [&](int c) { return CORR ? wrapi(c, n0) + 1 : min(c + 1, N0 - 1); };

  // ---- wave-uniform row starts (elements inside a plane) and per-lane byte offsets --------------------------------------
  unsigned urow[R + 2];  // u: byte offset of the padded row inside a plane
  unsigned qrow[R + 3];  // p: byte offset of the interior row inside an unpadded plane (CORR)
#pragma unroll
  for (int rr = 0; rr < R + 2; ++rr) urow[rr] = (unsigned)(prow_of(jb0 - 1 + rr) * N0) * EB;
  if (CORR) {
#pragma unroll
    for (int rr = 0; rr < R + 3; ++rr) qrow[rr] = (unsigned)((prow_of(jb0 - 1 + rr) - 1) * n0) * EB;
  }
  const unsigned ubytes = (unsigned)sz * EB, qbytes = (unsigned)(n0 * n1) * EB;
  const unsigned ucol = (unsigned)pcol_of(ci) * EB;            // main lanes: own column
  const unsigned qcol = CORR ? (unsigned)(pcol_of(ci) - 1) * EB : 0u;
  unsigned uhoff, qhoff = 0;  // packed halo loads: in-plane byte offset of this lane's (row, column)
  {
    const int r = lane & 7, grp = (lane >> 3) & 3;
    const int ru = r <= R + 1 ? r : 0;
    const int colu = (grp == 2) ? pcol_of(x0 + 64) : ((EXTRA && grp == 1) ? pcol_of(x0 - 2) : pcol_of(x0 - 1));  // EXTRA: lanes 8..15 hold column x0-2
    uhoff = (unsigned)(prow_of(jb0 - 1 + ru) * N0 + colu) * EB;
    if (CORR) {
      const int rq = r <= R + 2 ? r : 0;
      const int cq = grp == 0 ? x0 - 1 : (grp == 1 ? x0 : (grp == 2 ? x0 + 64 : x0 + 65));
      qhoff = (unsigned)((prow_of(jb0 - 1 + rq) - 1) * n0 + (pcol_of(cq) - 1)) * EB;
    }
  }

  auto uplane = [&](int kk) {  // padded plane index -> plane actually read
    return CORR == 1 ? wrapi(kk - 1, n2) + 1 : (CORR == 2 ? min(kk, N2 - 1) : kk);
  };
  auto load_plane = [&](Plane<T, R>& P, int kk) {
    const T* base = a_u + (long long)uplane(kk) * sz;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const rsrc_t rs = plane_rsrc(base + c * a.sc, ubytes);
#pragma unroll
      for (int rr = 0; rr < R + 2; ++rr) P.v[c][rr] = ldb<T>(rs, ucol, urow[rr]);
      P.h[c] = ldb<T>(rs, uhoff, 0);
    }
  };
  const unsigned urowm1 = (unsigned)(prow_of(jb0 - 2) * N0) * EB;
  auto load_text = [&](TExt<T, R>& E, int kk) {
    const long long po = (long long)uplane(kk) * sz;
    const rsrc_t rt = plane_rsrc(static_cast<const T*>((const void*)a.te.temp) + po, ubytes);
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr) E.t[rr] = ldb<T>(rt, ucol, urow[rr]);
    E.th = ldb<T>(rt, uhoff, 0);
    E.vm1 = ldb<T>(plane_rsrc(a_u + po + a.sc, ubytes), ucol, urowm1);
  };
  auto load_p = [&](T (&P)[R + 3], T& PH, int kk) {
    const rsrc_t rs = plane_rsrc(a_pI + (long long)(CORR == 2 ? min(kk, N2) : wrapi(kk - 1, n2)) * n0 * n1, qbytes);
#pragma unroll
    for (int rr = 0; rr < R + 3; ++rr) P[rr] = ldb<T>(rs, qcol, qrow[rr]);
    PH = ldb<T>(rs, qhoff, 0);
  };
  // u = u* - ∇p (applypressure!, operators.jl:225-233) for one register plane and its packed halo columns
  auto correct = [&](Plane<T, R>& P, const T (&Pc)[R + 3], T PHc, const T (&Pn)[R + 3], T PHn) {
    if (a.epi.self_in != 0.0) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) P.raw[c][rr] = P.v[c][rr + 1];
    }
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr) {
      const T pc = Pc[rr];
      P.v[0][rr] -= (next_h(pc, rdlane(PHc, 16 + rr)) - pc) * X.gs;
      P.v[1][rr] -= (Pc[rr + 1] - pc) * Y.gs;
      P.v[2][rr] -= (Pn[rr] - pc) * Z.gs;
    }
    P.h[0] -= (dpp_old<0x108>(PHc, PHc) - PHc) * X.gs;  // row_shl:8 — p of the next column, same row
    P.h[1] -= (dpp_old<0x101>(PHc, PHc) - PHc) * Y.gs;  // row_shl:1 — p of the next row, same column
    P.h[2] -= (PHn - PHc) * Z.gs;
  };

  T zprev[3][R];
  auto zflux0 = [&](const Plane<T, R>& C, const Plane<T, R>& Nx) {  // upper-face z-fluxes of the plane below the chunk
#pragma unroll
    for (int rr = 1; rr <= R; ++rr) {
      const T Wc = C.v[2][rr];
      zprev[0][rr - 1] = flux(C.v[0][rr], Nx.v[0][rr], Wc, next_h(Wc, rdlane(C.h[2], 16 + rr)), Z.vo);
      zprev[1][rr - 1] = flux(C.v[1][rr], Nx.v[1][rr], Wc, C.v[2][rr + 1], Z.vo);
      zprev[2][rr - 1] = flux(Wc, Nx.v[2][rr], Wc, Nx.v[2][rr], Z.vs);
    }
  };

  // output rows of this wavefront (clamped: rows / columns past the box are computed but never stored)
  unsigned orow[R];
#pragma unroll
  for (int rr = 0; rr < R; ++rr) orow[rr] = (unsigned)((min(jb0 + rr, n1 - 1) + 1) * N0) * EB;
  const unsigned ocol = (unsigned)(min(ci, n0 - 1) + 1) * EB;

  // RK epilogue, first half: s = ustart + Σ_q coef_q k_q for all R rows of plane k.  Issued at the top of the plane so the
  // loads fly during the flux arithmetic (they used to sit right before the stores: one exposed round trip per row).
  T eacc[3][EXTRA ? R : 1];
  T gacc[EXTRA ? R : 1];  // gravity!: α2 avg(temp) at the gdir-face (operators.jl:914-931; uniform grid: the plain mean)
  T Pz[3][EXTRA ? R : 1];  // the output rows of plane k-1 (for the Laplacian behind epi.wout)
  const T nux = X.vs * X.rs, nuy = Y.vs * Y.rs, nuz = Z.vs * Z.rs;  // ν/Δ² per direction (4ν/Δ · ¼/Δ: exact scalings)
  auto epi_load = [&](const Plane<T, R>& C, int k, T (&sacc)[3][R]) {
    const long long pk = (long long)k * sz;
    if constexpr (EXTRA) {
      if (a.epi.extra) {
        const T* b = static_cast<const T*>((const void*)a.epi.extra) + pk;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const rsrc_t rs = plane_rsrc(b + c * a.sc, ubytes);
#pragma unroll
          for (int rr = 0; rr < R; ++rr) eacc[c][rr] = ldb<T>(rs, ocol, orow[rr]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) eacc[c][rr] = 0;
      }
      if (a.epi.gtemp) {  // temp is a padded scalar field with valid ghost volumes (apply_bc_temp! has run)
        const T* tb = static_cast<const T*>((const void*)a.epi.gtemp) + pk;
        const int gd = a.epi.gdir;
        const rsrc_t r0 = plane_rsrc(tb, ubytes), r1 = plane_rsrc(gd == 2 ? tb + sz : tb, ubytes);
        const unsigned dcol = gd == 0 ? EB : 0u, drow = gd == 1 ? (unsigned)N0 * EB : 0u;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) gacc[rr] = (T)a.epi.ga2 * ((T)0.5 * (ldb<T>(r0, ocol, orow[rr]) + ldb<T>(r1, ocol + dcol, orow[rr] + drow)));
      }
    }
    const bool ntl = (a.nt & 8) != 0;  // INS_FLUX64_NT bit 3: non-temporal loads of the epilogue terms (read once; they share L2 with the re-read velocity planes)
    if (a.epi.ustart) {
      const T* b = static_cast<const T*>((const void*)a.epi.ustart) + pk;
      const T c0 = (T)(1.0 + a.epi.c0m1);  // exactly 1 in the k-basis
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const rsrc_t rs = plane_rsrc(b + c * a.sc, ubytes);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) sacc[c][rr] = c0 * (ntl ? ldb_aux<2>(rs, ocol, orow[rr], (T)0) : ldb<T>(rs, ocol, orow[rr]));
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) sacc[c][rr] = C.v[c][rr + 1];
    }
    if (a.epi.self_in != 0.0) {  // the stencil input is itself a term: no second trip to memory for it
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) sacc[c][rr] += (T)a.epi.self_in * (CORR ? C.raw[c][rr] : C.v[c][rr + 1]);
    }
    for (int q = 0; q < a.epi.n; ++q) {
      const T* kq = static_cast<const T*>((const void*)a.epi.k[q]) + pk;
      const T cq = (T)a.epi.coef[q];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const rsrc_t rs = plane_rsrc(kq + c * a.sc, ubytes);
        T kv[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) kv[rr] = ntl ? ldb_aux<2>(rs, ocol, orow[rr], (T)0) : ldb<T>(rs, ocol, orow[rr]);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) sacc[c][rr] += cq * kv[rr];
      }
    }
  };
  // second half: u* = s + coef_self f, and k_i = f when a later stage needs it
  // the three components of one result (INS_FLUX64_NT: cache-policy bits on these stores, an experiment)
  auto st3 = [&](T* o, unsigned co, unsigned rowb, T v0, T v1, T v2) {
    const rsrc_t r0 = plane_rsrc(o, ubytes), r1 = plane_rsrc(o + a.sc, ubytes), r2 = plane_rsrc(o + 2 * a.sc, ubytes);
    const int snt = a.nt & 7;
    if (snt == 0) {
      stb(r0, co, rowb, v0);
      stb(r1, co, rowb, v1);
      stb(r2, co, rowb, v2);
    } else if (snt == 1) {
      stb_aux<2>(r0, co, rowb, v0);
      stb_aux<2>(r1, co, rowb, v1);
      stb_aux<2>(r2, co, rowb, v2);
    } else if (snt == 2) {
      stb_aux<17>(r0, co, rowb, v0);
      stb_aux<17>(r1, co, rowb, v1);
      stb_aux<17>(r2, co, rowb, v2);
    } else {
      stb_aux<19>(r0, co, rowb, v0);
      stb_aux<19>(r1, co, rowb, v1);
      stb_aux<19>(r2, co, rowb, v2);
    }
  };
  auto emit = [&](int rr, int k, T fu, T fv, T fw, T s0, T s1, T s2) {
    const long long pk = (long long)k * sz;
    if constexpr (EXTRA) {
      fu += eacc[0][rr - 1];
      fv += eacc[1][rr - 1];
      fw += eacc[2][rr - 1];
      if (a.epi.gtemp) {
        const T gv = gacc[rr - 1];
        if (a.epi.gdir == 0) fu += gv;
        if (a.epi.gdir == 1) fv += gv;
        if (a.epi.gdir == 2) fw += gv;
      }
    }
    const int j = jb0 + rr - 1;  // interior row
    if (xout && j < n1) {
      const unsigned rowb = orow[rr - 1], co = ocol;
      if (FUSE) {
        if (CORR && a.epi.ustart_out) {  // chained steps: s is the corrected stencil input = this step's ustart (no term was added to it)
          T* w = static_cast<T*>((void*)a.epi.ustart_out) + pk;
          stb(plane_rsrc(w, ubytes), co, rowb, s0);
          stb(plane_rsrc(w + a.sc, ubytes), co, rowb, s1);
          stb(plane_rsrc(w + 2 * a.sc, ubytes), co, rowb, s2);
        }
        T* o = static_cast<T*>((void*)a.epi.ustar) + pk;
        st3(o, co, rowb, s0 + (T)a.epi.coef_self * fu, s1 + (T)a.epi.coef_self * fv, s2 + (T)a.epi.coef_self * fw);
      }
      if (!FUSE || a.epi.write_k) st3(a_F + pk, co, rowb, fu, fv, fw);
    }
  };

  // One output plane.  C = plane k, Nx = plane k+1 (both complete, corrected).  As soon as a row of C has been consumed its
  // registers are re-loaded with the same row of plane `kload` (= k+2, the next plane this buffer has to hold), so the
  // prefetch of plane k+2 is in flight during the whole of plane k without a third register plane.
  // temperature stage (EXTRA without in-kernel correction, a.tm): state carried from plane to plane
  constexpr bool TMK = EXTRA && CORR == 0;
  T Tz[TMK ? R : 1];      // T of plane k-1 at the output rows
  T wwprev[TMK ? R : 1];  // w·diffusion(w) of plane k-1 at the output rows (zero below the first plane: operators.jl:793-807 reads a ghost of `diff`)
  T PzV0 = 0;               // v at the halo row of plane k-1
  T hU_prev = 0;            // packed halo columns of u, plane k-1
  auto body = [&](Plane<T, R>& C, const Plane<T, R>& Nx, TExt<T, R>& CT, const TExt<T, R>& NT, int k, int kload) {
    const T* nb = a_u + (long long)uplane(kload) * sz;
    const rsrc_t n0r = plane_rsrc(nb, ubytes), n1r = plane_rsrc(nb + a.sc, ubytes), n2r = plane_rsrc(nb + 2 * a.sc, ubytes);
    const T ch0 = C.h[0], ch1 = C.h[1], ch2 = C.h[2];
    T sacc[3][R];
    if (FUSE) epi_load(C, k, sacc);
    const bool tm = TMK && a.tm;
    const rsrc_t ntr = plane_rsrc(static_cast<const T*>((const void*)a.te.temp) + (long long)uplane(kload) * sz, ubytes);
    T tacc[TMK ? R : 1];
    const T cth = CT.th;
    if constexpr (TMK) {
      if (tm) {  // temp_out = tempstart + Σ_j coef_j ktemp_j + c_self ktemp_i: the loads fly during the flux arithmetic
        const long long pk = (long long)k * sz;
        const rsrc_t rs0 = plane_rsrc(static_cast<const T*>((const void*)a.te.tempstart) + pk, ubytes);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) tacc[rr] = ldb<T>(rs0, ocol, orow[rr]);
        for (int q = 0; q < a.te.n; ++q) {
          const rsrc_t rq = plane_rsrc(static_cast<const T*>((const void*)a.te.k[q]) + pk, ubytes);
#pragma unroll
          for (int rr = 0; rr < R; ++rr) tacc[rr] += (T)a.te.coef[q] * ldb<T>(rq, ocol, orow[rr]);
        }
      }
    }
    T fyu_o = 0, fyv_o = 0, fyw_o = 0;
    T Uo = 0, Vo = 0, Wo = 0;  // row rr - 1 of this plane (EXTRA: its registers already hold the next plane)
    T To = 0, wv_o = 0;        // T and v·diffusion(v) of row rr - 1
#pragma unroll
    for (int rr = 0; rr <= R; ++rr) {
      const T Uc = C.v[0][rr], Vc = C.v[1][rr], Wc = C.v[2][rr];
      if (SKEL) {  // timing experiment: same loads, stores and epilogue, trivial arithmetic (tools/scan_flux64.sh)
        if (rr >= 1) {
          const T fu = Uc + fyu_o + C.v[0][rr + 1] + Nx.v[0][rr] + zprev[0][rr - 1] + ch0;
          const T fv = Vc + fyv_o + C.v[1][rr + 1] + Nx.v[1][rr] + zprev[1][rr - 1] + ch1;
          const T fw = Wc + fyw_o + C.v[2][rr + 1] + Nx.v[2][rr] + zprev[2][rr - 1] + ch2;
          emit(rr, k, fu, fv, fw, FUSE ? sacc[0][rr - 1] : (T)0, FUSE ? sacc[1][rr - 1] : (T)0, FUSE ? sacc[2][rr - 1] : (T)0);
        }
        fyu_o = Uc;
        fyv_o = Vc;
        fyw_o = Wc;
        C.v[0][rr] = ldb<T>(n0r, ucol, urow[rr]);
        C.v[1][rr] = ldb<T>(n1r, ucol, urow[rr]);
        C.v[2][rr] = ldb<T>(n2r, ucol, urow[rr]);
        continue;
      }
      const T Vn = next_h(Vc, rdlane(ch1, 16 + rr));
      // y-fluxes through the face between rows rr and rr+1
      const T fyu = flux(Uc, C.v[0][rr + 1], Vc, Vn, Y.vo);
      const T fyv = flux(Vc, C.v[1][rr + 1], Vc, C.v[1][rr + 1], Y.vs);
      const T fyw = flux(Wc, C.v[2][rr + 1], Vc, Nx.v[1][rr], Y.vo);
      if (rr >= 1) {
        const T Un = next_h(Uc, rdlane(ch0, 16 + rr)), Wn = next_h(Wc, rdlane(ch2, 16 + rr));
        const T fxu = flux(Uc, Un, Uc, Un, X.vs);
        const T fxv = flux(Vc, Vn, Uc, C.v[0][rr + 1], X.vo);
        const T fxw = flux(Wc, Wn, Uc, Nx.v[0][rr], X.vo);
        // left-face fluxes of lane 0 from the halo column x0-1 (all other lanes take their left neighbour's right face)
        const T sU = rdlane(ch0, rr), sV = rdlane(ch1, rr), sW = rdlane(ch2, rr);
        const T sUu = rdlane(ch0, rr + 1), sUn = rdlane(Nx.h[0], rr);
        const T lxu = flux(sU, Uc, sU, Uc, X.vs);
        const T lxv = flux(sV, Vc, sU, sUu, X.vo);
        const T lxw = flux(sW, Wc, sU, sUn, X.vo);
        T fu = (fxu - prev_h(fxu, lxu)) * X.rs;
        T fv = (fxv - prev_h(fxv, lxv)) * X.ro;
        T fw = (fxw - prev_h(fxw, lxw)) * X.ro;
        fu += (fyu - fyu_o) * Y.ro;
        fv += (fyv - fyv_o) * Y.rs;
        fw += (fyw - fyw_o) * Y.ro;
        const T zu = flux(Uc, Nx.v[0][rr], Wc, Wn, Z.vo);
        const T zv = flux(Vc, Nx.v[1][rr], Wc, C.v[2][rr + 1], Z.vo);
        const T zw = flux(Wc, Nx.v[2][rr], Wc, Nx.v[2][rr], Z.vs);
        fu += (zu - zprev[0][rr - 1]) * Z.ro;
        fv += (zv - zprev[1][rr - 1]) * Z.ro;
        fw += (zw - zprev[2][rr - 1]) * Z.rs;
        zprev[0][rr - 1] = zu;
        zprev[1][rr - 1] = zv;
        zprev[2][rr - 1] = zw;
        if constexpr (EXTRA) {
          // w_α = u_α · diffusion(u)_α at this volume (dissipation!, operators.jl:791-797): the seven-point Laplacian from values already in
          // registers (x: wave shifts, y: the rows above / below, z: plane k+1 and the saved plane k-1)
          if (a.epi.wout || tm) {
            const T Up = prev_h(Uc, sU), Vp = prev_h(Vc, sV), Wp = prev_h(Wc, sW);
            const T wu = Uc * ((Un + Up - 2 * Uc) * nux + (C.v[0][rr + 1] + Uo - 2 * Uc) * nuy + (Nx.v[0][rr] + Pz[0][rr - 1] - 2 * Uc) * nuz);
            const T wv = Vc * ((Vn + Vp - 2 * Vc) * nux + (C.v[1][rr + 1] + Vo - 2 * Vc) * nuy + (Nx.v[1][rr] + Pz[1][rr - 1] - 2 * Vc) * nuz);
            const T ww = Wc * ((Wn + Wp - 2 * Wc) * nux + (C.v[2][rr + 1] + Wo - 2 * Wc) * nuy + (Nx.v[2][rr] + Pz[2][rr - 1] - 2 * Wc) * nuz);
            const int j = jb0 + rr - 1;
            if (a.epi.wout && xout && j < n1) {
              T* w = static_cast<T*>((void*)a.epi.wout) + (long long)k * sz;
              stb(plane_rsrc(w, ubytes), ocol, orow[rr - 1], wu);
              stb(plane_rsrc(w + a.sc, ubytes), ocol, orow[rr - 1], wv);
              stb(plane_rsrc(w + 2 * a.sc, ubytes), ocol, orow[rr - 1], ww);
            }
            if constexpr (TMK) if (tm) {
              // ---- temperature stage at this volume.  Lower-face terms of dissipation!: the left neighbour's w_x by a wave shift — lane 0
              // takes it from the halo column x0-1, whose Laplacian is formed from wave-uniform values (columns x0-2, x0-1 of the packed
              // halo, lane 0's own column, the halo columns of planes k±1) —, the row below from the previous iteration, the plane below
              // from the previous plane; all of them zero below the first volume of a direction (a ghost volume of `diff`).
              T wuL = 0;
              if (x0 > 0) {
                const T U0 = rdlane(Uc, 0), Um2 = rdlane(ch0, 8 + rr), sUd = rdlane(ch0, rr - 1), sUp = rdlane(hU_prev, rr);
                wuL = sU * ((U0 + Um2 - 2 * sU) * nux + (sUu + sUd - 2 * sU) * nuy + (sUn + sUp - 2 * sU) * nuz);
              }
              const T dc = (T)a.te.dcoef;
              const T dd = dc * (prev_h(wu, wuL) + wu) / 2 + dc * (wv_o + wv) / 2 + dc * (wwprev[rr - 1] + ww) / 2;
              // convection_diffusion_temp! (operators.jl:723-737) on the uniform grid: avg = mean, Δ = Δu
              const T Tc = CT.t[rr];
              const T Txp = next_h(Tc, rdlane(cth, 16 + rr)), Txm = prev_h(Tc, rdlane(cth, rr));
              const T Typ = CT.t[rr + 1], Tym = To, Tzp = NT.t[rr], Tzm = Tz[rr - 1];
              const T a4 = (T)a.te.a4;
              T cd = (-(Uc * ((Tc + Txp) / 2) - Up * ((Txm + Tc) / 2)) + a4 * ((Txp - Tc) * X.gs - (Tc - Txm) * X.gs)) * X.gs;
              cd += (-(Vc * ((Tc + Typ) / 2) - Vo * ((Tym + Tc) / 2)) + a4 * ((Typ - Tc) * Y.gs - (Tc - Tym) * Y.gs)) * Y.gs;
              cd += (-(Wc * ((Tc + Tzp) / 2) - Pz[2][rr - 1] * ((Tzm + Tc) / 2)) + a4 * ((Tzp - Tc) * Z.gs - (Tc - Tzm) * Z.gs)) * Z.gs;
              const T kt = cd + dd;
              if (xout && j < n1) {
                const long long pk = (long long)k * sz;
                if (a.te.ktemp_out) stb(plane_rsrc(static_cast<T*>((void*)a.te.ktemp_out) + pk, ubytes), ocol, orow[rr - 1], kt);
                stb(plane_rsrc(static_cast<T*>((void*)a.te.temp_out) + pk, ubytes), ocol, orow[rr - 1], tacc[rr - 1] + (T)a.te.c_self * kt);
              }
              wv_o = wv;
              wwprev[rr - 1] = ww;
              Tz[rr - 1] = Tc;
            }
          }
          Pz[0][rr - 1] = Uc;
          Pz[1][rr - 1] = Vc;
          Pz[2][rr - 1] = Wc;
        }
        emit(rr, k, fu, fv, fw, FUSE ? sacc[0][rr - 1] : (T)0, FUSE ? sacc[1][rr - 1] : (T)0, FUSE ? sacc[2][rr - 1] : (T)0);
      }
      fyu_o = fyu;
      fyv_o = fyv;
      fyw_o = fyw;
      if constexpr (TMK) {
        if (tm) {
          if (rr == 0) {  // v·diffusion(v) at the halo row (the row below the first output row; a ghost row of `diff` when it is row -1 of the box)
            const T Vp = prev_h(Vc, rdlane(ch1, 0));
            wv_o = jb0 > 0 ? Vc * ((Vn + Vp - 2 * Vc) * nux + (C.v[1][1] + CT.vm1 - 2 * Vc) * nuy + (Nx.v[1][0] + PzV0 - 2 * Vc) * nuz) : (T)0;
            PzV0 = Vc;
          }
          To = CT.t[rr];
          CT.t[rr] = ldb<T>(ntr, ucol, urow[rr]);
        }
      }
      if constexpr (EXTRA) {
        Uo = Uc;
        Vo = Vc;
        Wo = Wc;
      }
      // row rr of plane k is dead: its registers receive plane `kload`
      C.v[0][rr] = ldb<T>(n0r, ucol, urow[rr]);
      C.v[1][rr] = ldb<T>(n1r, ucol, urow[rr]);
      C.v[2][rr] = ldb<T>(n2r, ucol, urow[rr]);
    }
    C.v[0][R + 1] = ldb<T>(n0r, ucol, urow[R + 1]);
    C.v[1][R + 1] = ldb<T>(n1r, ucol, urow[R + 1]);
    C.v[2][R + 1] = ldb<T>(n2r, ucol, urow[R + 1]);
    C.h[0] = ldb<T>(n0r, uhoff, 0);
    C.h[1] = ldb<T>(n1r, uhoff, 0);
    C.h[2] = ldb<T>(n2r, uhoff, 0);
    if constexpr (TMK) {
      if (tm) {
        hU_prev = ch0;
        CT.t[R + 1] = ldb<T>(ntr, ucol, urow[R + 1]);
        CT.th = ldb<T>(ntr, uhoff, 0);
        CT.vm1 = ldb<T>(n1r, ucol, urowm1);
      }
    }
  };

  // Two register planes.  Loads past the chunk re-read plane k1 / p(k1+1) (cache hits) instead of branching.
  Plane<T, R> P0, P1;
  TExt<T, R> TE0, TE1;
  if (!CORR) {
    load_plane(P0, k0 - 1);
    load_plane(P1, k0);
    zflux0(P0, P1);
    if constexpr (EXTRA) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) Pz[c][rr] = P0.v[c][rr + 1];
      if (a.tm) {
        load_text(TE0, k0 - 1);
        load_text(TE1, k0);
        PzV0 = P0.v[1][0];
        hU_prev = P0.h[0];
        // w·diffusion(w) of plane k0-1 (zero when that is the ghost plane): its z-neighbour below, plane k0-2, is read here once per chunk
        const rsrc_t rw = plane_rsrc(a_u + (long long)uplane(max(k0 - 2, 0)) * sz + 2 * a.sc, ubytes);
#pragma unroll
        for (int rr = 1; rr <= R; ++rr) {
          const T Wc = P0.v[2][rr], Wm2 = ldb<T>(rw, ucol, urow[rr]);
          const T Wn = next_h(Wc, rdlane(P0.h[2], 16 + rr)), Wp = prev_h(Wc, rdlane(P0.h[2], rr));
          wwprev[rr - 1] = k0 > 1 ? Wc * ((Wn + Wp - 2 * Wc) * nux + (P0.v[2][rr + 1] + P0.v[2][rr - 1] - 2 * Wc) * nuy + (P1.v[2][rr] + Wm2 - 2 * Wc) * nuz) : (T)0;
          Tz[rr - 1] = TE0.t[rr];
        }
      }
    }
    load_plane(P0, min(k0 + 1, k1));
    if constexpr (EXTRA) {
      if (a.tm) load_text(TE0, min(k0 + 1, k1));
    }
    int k = k0;
    while (true) {
      if (a.bar) __builtin_amdgcn_s_barrier();
      body(P1, P0, TE1, TE0, k, min(k + 2, k1));
      if (++k >= k1) break;
      if (a.bar) __builtin_amdgcn_s_barrier();
      body(P0, P1, TE0, TE1, k, min(k + 2, k1));
      if (++k >= k1) break;
    }
  } else {
    // invariant at the top of iteration k: cur = corrected plane k, nxt = RAW plane k+1, Pa = p(k+1), Pb = p(k+2)
    T Pa[R + 3], Pb[R + 3], Ha, Hb;
    load_p(Pa, Ha, k0 - 1);
    load_p(Pb, Hb, k0);
    load_plane(P0, k0 - 1);
    load_plane(P1, k0);
    correct(P0, Pa, Ha, Pb, Hb);
    load_p(Pa, Ha, k0 + 1);
    correct(P1, Pb, Hb, Pa, Ha);
    zflux0(P0, P1);
    if constexpr (EXTRA) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) Pz[c][rr] = P0.v[c][rr + 1];
    }
    load_plane(P0, min(k0 + 1, k1));
    load_p(Pb, Hb, min(k0 + 2, k1 + 1));
    int k = k0;
    while (true) {
      if (a.bar) __builtin_amdgcn_s_barrier();
      correct(P0, Pa, Ha, Pb, Hb);  // plane k+1 with p(k+1), p(k+2)
      load_p(Pa, Ha, min(k + 3, k1 + 1));
      body(P1, P0, TE1, TE0, k, min(k + 2, k1));
      if (++k >= k1) break;
      if (a.bar) __builtin_amdgcn_s_barrier();
      correct(P1, Pb, Hb, Pa, Ha);
      load_p(Pb, Hb, min(k + 3, k1 + 1));
      body(P0, P1, TE0, TE1, k, min(k + 2, k1));
      if (++k >= k1) break;
    }
  }
}

// tile-shape knobs: run-time options (ins_options.hip; environment variable of the same name, or ins_set_option)
#define g_disable ((int)ins_opt(OPT_INS_DISABLE_FLUX64))
#define g_rows ((int)ins_opt(OPT_INS_FLUX64_ROWS))
#define g_rows_corr ((int)ins_opt(OPT_INS_FLUX64_ROWS_CORR))
#define g_zchunk ((int)ins_opt(OPT_INS_FLUX64_ZC))
#define g_zchunk_corr ((int)ins_opt(OPT_INS_FLUX64_ZC_CORR))
#define g_xw ((int)ins_opt(OPT_INS_FLUX64_XW))
#define g_lds ((int)ins_opt(OPT_INS_FLUX64_LDS))    // experiment: dynamic LDS bytes per workgroup (caps workgroups per CU)
#define g_skel ((int)ins_opt(OPT_INS_FLUX64_SKEL))  // timing experiment only: wrong results by design

template <typename T, int R, int XW, bool FUSE, int NW>
int launch_range(const ins_grid* G, FluxArgs& a, int corr_mode, hipStream_t s) {
  const unsigned nb = (unsigned)(8LL * a.ntx * ((a.nty + 7) / 8) * a.ntz);
  if (nb == 0) return INS_OK;
  const dim3 block(64, NW, 1);
  if constexpr (NW <= 8 && R == 4 && sizeof(T) == 8) {
    if (g_skel && corr_mode == 0) {
      hipLaunchKernelGGL((k_flux64<T, R, XW, FUSE, 0, true, NW>), dim3(nb), block, (size_t)g_lds, s, a);
      INS_LAUNCH_CHECK();
      return INS_OK;
    }
  }
  if (a.tm && corr_mode != 0) {
    ins_set_error("temperature stage inside the stage kernel: no in-kernel pressure correction");
    return INS_ERR_UNSUPPORTED;
  }
  if (a.epi.extra || a.epi.gtemp || a.epi.wout || a.tm) {
    if constexpr (FUSE && R == 2 && NW == 4 && sizeof(T) == 8) {
      if (corr_mode == 0) {
        hipLaunchKernelGGL((k_flux64<T, R, XW, true, 0, false, NW, true>), dim3(nb), block, (size_t)g_lds, s, a);
        INS_LAUNCH_CHECK();
        return INS_OK;
      }
      if (corr_mode == 1) {
        hipLaunchKernelGGL((k_flux64<T, R, XW, true, 1, false, NW, true>), dim3(nb), block, (size_t)g_lds, s, a);
        INS_LAUNCH_CHECK();
        return INS_OK;
      }
    }
    ins_set_error("stage kernel with an extra force term: fp64, fused epilogue, 2 rows, 4 wavefronts, no slab correction");
    return INS_ERR_UNSUPPORTED;
  }
  if (corr_mode == 0)
    hipLaunchKernelGGL((k_flux64<T, R, XW, FUSE, 0, false, NW>), dim3(nb), block, (size_t)g_lds, s, a);
  else if constexpr (FUSE && R <= 5) {
    if (corr_mode == 1)
      hipLaunchKernelGGL((k_flux64<T, R, XW, true, 1, false, NW>), dim3(nb), block, (size_t)g_lds, s, a);
    else
      hipLaunchKernelGGL((k_flux64<T, R, XW, true, 2, false, NW>), dim3(nb), block, (size_t)g_lds, s, a);
  } else {
    ins_set_error("in-kernel pressure correction needs the fused epilogue and <= 5 rows per thread");
    return INS_ERR_UNSUPPORTED;
  }
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// part 0: every plane; 1: the planes that read no ghost plane, [1 + ZB, nzl + 1 - ZB); 2: the two boundary ranges of ZB planes
constexpr int ZB = 4;
template <typename T, int R, int XW, bool FUSE, int NW = 4>
int launch(const ins_grid* G, FluxArgs& a, int corr_mode, int part, hipStream_t s) {
  const GridDev& g = G->g;
  a.ntx = cdiv(g.N[0] - 2, 64 * XW);
  a.nty = cdiv(g.N[1] - 2, (NW / XW) * R);
  const int nzl = g.N[2] - 2;
  a.k_lo = 1;
  a.k_hi = nzl + 1;
  a.kB = 0;
  if (part != 0 && nzl <= 2 * ZB) {  // too thin to split: everything in part 2
    if (part == 1) return INS_OK;
    part = 0;
  }
  if (part == 1) {
    a.k_lo = 1 + ZB;
    a.k_hi = nzl + 1 - ZB;
  } else if (part == 2) {
    a.zc = ZB;
    a.kB = nzl + 1 - ZB;
  }
  a.ntz = part == 2 ? 2 : cdiv(a.k_hi - a.k_lo, a.zc);
  return launch_range<T, R, XW, FUSE, NW>(G, a, corr_mode, s);
}

}  // namespace

// Tuning knobs for experiments in one process on one allocation (not part of the public ABI); -1 keeps a value.
extern "C" void ins_tune_flux64(int disable, int rows, int rows_corr, int zchunk, int xw, int skel, int burst_unused, int lds) {
  if (disable >= 0) ins_set_option("INS_DISABLE_FLUX64", disable);
  if (rows >= 0) ins_set_option("INS_FLUX64_ROWS", rows);
  if (rows_corr >= 0) ins_set_option("INS_FLUX64_ROWS_CORR", rows_corr);
  if (zchunk >= 0) ins_set_option("INS_FLUX64_ZC", zchunk);
  if (xw >= 0) ins_set_option("INS_FLUX64_XW", xw);
  if (skel >= 0) ins_set_option("INS_FLUX64_SKEL", skel);
  (void)burst_unused;
  if (lds >= 0) ins_set_option("INS_FLUX64_LDS", lds);
}

// 3-D, every interior volume a DOF (all-periodic box or periodic slab), bitwise-constant metric records, room for the
// periodic wrap of a full wavefront window.
bool ins_flux64_supported(const ins_grid* G) {
  const GridDev& g = G->g;
  return !g_disable && g.D == 3 && G->all_dof && G->uniform_exact && g.N[0] - 2 >= 66 && g.N[1] - 2 >= 8 && g.N[2] - 2 >= 4;
}

// corr_mode 0: u has valid ghost volumes.  1 / 2: see k_flux64.  fuse: RK epilogue `epi`.
template <typename T>
static int flux64_dispatch(const ins_grid* G, double visc, const T* u, T* F, const RkEpi* epi, const T* pI, int corr_mode, hipStream_t s, int part) {
  const GridDev& g = G->g;
  FluxArgs a;
  memset(&a, 0, sizeof(a));
  a.u = u;
  a.pI = pI;
  a.F = F;
  a.sc = g.sc;
  a.N0 = g.N[0];
  a.N1 = g.N[1];
  a.N2 = g.N[2];
  const int n2 = g.N[2] - 2;
  a.X = make_dir(G, 0, visc);
  a.Y = make_dir(G, 1, visc);
  a.Z = make_dir(G, 2, visc);
  if (epi) a.epi = *epi;
  if (epi && epi->tstage) {
    a.te = *epi->tstage;
    a.tm = 1;
    a.epi.tstage = nullptr;
  }
  const int waves_x = cdiv(g.N[0] - 2, 64);
  // wavefronts side by side: 4 for 256-wide rows (2.85 vs 2.90 ms/step with 2), 2 for 512-wide ones (23.7 vs 24.3 ms/step)
  const int xwo = (corr_mode && ins_opt(OPT_INS_FLUX64_XW_CORR)) ? (int)ins_opt(OPT_INS_FLUX64_XW_CORR) : g_xw;
  int xw = xwo ? xwo : (waves_x >= 8 ? 2 : (waves_x >= 4 ? 4 : (waves_x >= 2 ? 2 : 1)));
  if (!xwo)  // rows of 3 or 6 wavefronts (192, 384 columns): side-by-side counts that leave no wavefront outside the box
    while (xw > 1 && cdiv(waves_x, xw) * xw > waves_x) xw >>= 1;
  int rows = corr_mode ? (g_rows_corr ? g_rows_corr : 2) : (g_rows ? g_rows : 4);
  rows = std::min(std::max(rows, 2), corr_mode ? 5 : 6);
  constexpr bool F32 = sizeof(T) == 4;  // the fp32 family is built for the default shapes only (2 rows correcting, 4 otherwise)
  if (F32) rows = corr_mode ? (ins_opt(OPT_INS_F32_CORR_ROWS) == 4 ? 4 : 2) : 4;
  // Workgroup shape and z-chunk.  The wavefronts of a workgroup share halo rows / columns; a workgroup barrier per plane keeps them on
  // the same plane, so those shared lines are cache hits instead of HBM re-reads (512^3 plain K1, same box: 1.45 -> 1.37 ms with 4
  // wavefronts, 1.28 ms with 8 wavefronts = 128 x 16 cells per plane and workgroup: the flat-copy rate of that box, profiles/r02_k1_lab.txt;
  // without the barrier 8 wavefronts are SLOWER than 4: 1.53 ms).  Every z-chunk re-reads two planes ((zc+2)/zc), so chunks are as long as
  // the tile count allows: 8 wavefronts and 64-plane chunks when that still gives every CU a workgroup (256^3: the correcting kernel, 2 rows
  // per thread; 512^3: every kernel), else 4 wavefronts (256^3 plain K1: 8 wavefronts x 64 planes would leave half the CUs idle, 0.30 vs
  // 0.205 ms; x 32 planes 0.213 ms).  Same-box steps: 256^3 2.69 (4 wavefronts) -> 2.64 ms, 512^3 23.7 -> 22.35 ms (profiles/r02a_step_lab.txt).
  const int nwo = (int)ins_opt(OPT_INS_FLUX64_NW);
  const int zco = (corr_mode && g_zchunk_corr) ? g_zchunk_corr : g_zchunk;
  auto tiles = [&](int nw_, int zc_) {
    return (long long)cdiv(g.N[0] - 2, 64 * xw) * cdiv(g.N[1] - 2, (nw_ / xw) * rows) * cdiv(n2, zc_);
  };
  const long long mintiles = ins_opt(OPT_INS_FLUX64_MINTILES) > 0 ? ins_opt(OPT_INS_FLUX64_MINTILES) : 256;  // one workgroup per CU
  int nw = nwo == 16 ? 16 : (nwo == 8 ? 8 : (nwo == 4 ? 4 : 0));
  int zc = zco;
  if (!nw) {
    nw = 4;
    if (xw <= 8 && !g_lds) {
      for (int z : {64}) {
        const int zt = zco ? zco : z;
        if (n2 >= zt && tiles(8, zt) >= mintiles) {
          nw = 8;
          zc = zt;
          break;
        }
      }
    }
  }
  if (!zc) {
    const bool small_plane = (long long)(g.N[0] - 2) * (g.N[1] - 2) <= 256LL * 256;
    zc = n2 >= 256 && small_plane ? 64 : (n2 >= 128 ? 32 : (n2 >= 64 ? 16 : (n2 >= 32 ? 8 : 4)));
    // boxes below 256 planes: chunks as long as two workgroups per CU allow (128^3: 32-plane chunks are 128 workgroups for 256 CUs —
    // 0.61 ms/step; 8-plane chunks 0.45)
    if (n2 < 256)
      while (zc > 4 && tiles(nw, zc) < 2 * mintiles) zc >>= 1;
  }
  if (rows != 2 && nw == 16) nw = 8;
  if (epi && (epi->extra || epi->gtemp || epi->wout || epi->tstage)) {  // one instantiation serves the extended stage loop
    rows = 2;
    nw = 4;
    if (!zco) zc = n2 >= 128 ? 32 : (n2 >= 64 ? 16 : (n2 >= 32 ? 8 : 4));
  }
  a.zc = zc;
  a.bar = ins_opt(OPT_INS_FLUX64_NOBAR) ? 0 : 1;
  // cache-policy bits of the stage kernel's once-only streams: bit 0 = nt on the result stores, bit 3 = nt on the loads of the epilogue terms (ustart, stage terms).
  // Default: both (256^3 step 2.538 -> 2.518 ms same-box, 512^3 21.09 -> 20.99: the re-read velocity planes keep more of L2); INS_FLUX64_NT=16: plain; 1 / 2 / 3 / 8: the
  // single experiments (profiles/r03_nt_lab.txt)
  {
    const long long o = ins_opt(OPT_INS_FLUX64_NT);
    a.nt = o == 0 ? ((epi && !F32) ? 9 : 0) : (o == 16 ? 0 : (int)o);  // the plain kernel (no epilogue) lost with nt stores in round 1's experiment: stage kernels only; fp32 not measured
  }
#define INS_F64_CASE(RR, FUSE)                                                        \
  if constexpr (!F32 || RR == 2 || RR == 4) {                                         \
    if (rows == RR) {                                                                 \
      if constexpr (RR == 2 && !F32) {                                                \
        if (nw == 16) {                                                               \
          if (xw == 4) return launch<T, RR, 4, FUSE, 16>(G, a, corr_mode, part, s);   \
          if (xw == 2) return launch<T, RR, 2, FUSE, 16>(G, a, corr_mode, part, s);   \
          return launch<T, RR, 1, FUSE, 16>(G, a, corr_mode, part, s);                \
        }                                                                             \
      }                                                                               \
      if (nw >= 8) {                                                                  \
        if (xw == 4) return launch<T, RR, 4, FUSE, 8>(G, a, corr_mode, part, s);      \
        if (xw == 2) return launch<T, RR, 2, FUSE, 8>(G, a, corr_mode, part, s);      \
        return launch<T, RR, 1, FUSE, 8>(G, a, corr_mode, part, s);                   \
      }                                                                               \
      if (xw == 4) return launch<T, RR, 4, FUSE>(G, a, corr_mode, part, s);           \
      if (xw == 2) return launch<T, RR, 2, FUSE>(G, a, corr_mode, part, s);           \
      return launch<T, RR, 1, FUSE>(G, a, corr_mode, part, s);                        \
    }                                                                                 \
  }
  if (epi) {
    INS_F64_CASE(2, true)
    INS_F64_CASE(3, true)
    INS_F64_CASE(4, true)
    INS_F64_CASE(5, true)
    INS_F64_CASE(6, true)
  } else {
    INS_F64_CASE(2, false)
    INS_F64_CASE(3, false)
    INS_F64_CASE(4, false)
    INS_F64_CASE(5, false)
    INS_F64_CASE(6, false)
  }
#undef INS_F64_CASE
  return INS_ERR_INVALID;
}

// two x-columns per lane (16 B per lane and memory instruction) wherever the box allows it: ins_flux128.hip
bool ins_flux128_supported(const ins_grid* G, const RkEpi* epi, int corr_mode, bool f32);
int ins_k_flux128(const ins_grid* G, double visc, const double* u, double* F, const RkEpi* epi, const double* pI, int corr_mode, hipStream_t s, int part);
int ins_k_flux128_f32(const ins_grid* G, double visc, const float* u, float* F, const RkEpi* epi, const float* pI, int corr_mode, hipStream_t s, int part);

int ins_k_flux64(const ins_grid* G, double visc, const double* u, double* F, const RkEpi* epi, const double* pI, int corr_mode, hipStream_t s,
                 int part) {
  if (ins_flux128_supported(G, epi, corr_mode, false)) return ins_k_flux128(G, visc, u, F, epi, pI, corr_mode, s, part);
  return flux64_dispatch<double>(G, visc, u, F, epi, pI, corr_mode, s, part);
}
// fp32 family (`_f32` entry points): same kernels instantiated for float; RkEpi's pointers are float arrays then
int ins_k_flux64_f32(const ins_grid* G, double visc, const float* u, float* F, const RkEpi* epi, const float* pI, int corr_mode, hipStream_t s,
                     int part) {
  if (ins_flux128_supported(G, epi, corr_mode, true) && !ins_opt(OPT_INS_F32_ONE_COLUMN)) return ins_k_flux128_f32(G, visc, u, F, epi, pI, corr_mode, s, part);
  return flux64_dispatch<float>(G, visc, u, F, epi, pI, corr_mode, s, part);
}
