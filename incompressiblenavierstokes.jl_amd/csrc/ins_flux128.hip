// K1 / stage kernel of uniform periodic boxes with TWO x-columns per lane: 128 outputs per wavefront, 16 B per lane and memory instruction in fp64
// (buffer_load_b128 / buffer_store_b128), 8 B in fp32.  Same arithmetic, same operand order and the same scheme as ins_flux64.hip
// (convection_diffusion_kernel! + fill!(F, 0), operators.jl:647-690, 971; RK epilogue step_explicit_runge_kutta.jl:35-38; in-register
// applypressure! of the previous stage, operators.jl:225-233) — what changes is the lane layout:
//   * lane l holds columns x0 + 2l ("a") and x0 + 2l + 1 ("b") of every register row.  The right neighbour of a is b (same lane), the right neighbour of b
//     is the next lane's a (ONE wave shift per value and row instead of one per column), the left face of a is the next-lower lane's right face of b;
//   * the halo columns x0 - 1 and x0 + 128 of all R + 2 rows arrive packed in one extra load per component and plane, as in ins_flux64.hip;
//   * why: the 8-B-per-lane kernels ran at the rate of an 8-B-per-lane copy with the same access pattern (tools/k1like.hip: 4.65 TB/s against 4.97 TB/s for
//     16 B per lane, same box, same tile order, same barrier) — the vector memory pipeline issues half as many instructions per byte.
// Pairs start at interior column 0 = padded column 1, i.e. 8 B off a 16-B boundary; the probe measured no difference to aligned pairs (rows are 2064 /
// 4112 B long, every row start has a different offset inside its 128-B line anyway).  Rows must hold an even number of volumes (a pair never straddles
// the box end); other boxes keep ins_flux64.hip.
#include <algorithm>
#include <cstring>

#include "ins_internal.h"
#include "ins_wave64.h"
#include "ins_flux_common.h"

namespace {

template <typename T>
struct P2 {
  T a, b;
};
template <typename T>
__device__ __forceinline__ P2<T> ldb2(rsrc_t r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ P2<double> ldb2<double>(rsrc_t r, unsigned voff, unsigned soff) {
  const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return {__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z)};
}
template <>
__device__ __forceinline__ P2<float> ldb2<float>(rsrc_t r, unsigned voff, unsigned soff) {
  const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return {__int_as_float((int)v.x), __int_as_float((int)v.y)};
}
__device__ __forceinline__ void stb2(rsrc_t r, unsigned voff, unsigned soff, double a, double b) {
  v4u v;
  v.x = (unsigned)__double2loint(a);
  v.y = (unsigned)__double2hiint(a);
  v.z = (unsigned)__double2loint(b);
  v.w = (unsigned)__double2hiint(b);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}
__device__ __forceinline__ void stb2(rsrc_t r, unsigned voff, unsigned soff, float a, float b) {
  v2u v;
  v.x = (unsigned)__float_as_int(a);
  v.y = (unsigned)__float_as_int(b);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

template <typename T, int R>
struct Plane2 {
  P2<T> v[3][R + 2];
  T h[3];  // packed halo columns: lane r = row r of column x0-1, lane 16+r = row r of column x0+128
};

// NW wavefronts per workgroup: XW side by side in x (128 columns each), NW/XW stacked in y (R rows each).  CORR: 0 = `u` has valid ghost volumes;
// 1 = `u` is the previous stage's uncorrected u* (interior only), pI its unpadded pressure, every neighbour through the periodic image;
// 2 = z-slab: x, y periodic images, z through exchanged ghost planes, pI = [1 | nzl | 2] extended buffer.
// (the correcting form with two rows of pairs needs ~340 registers in fp64: it runs as 4-wavefront workgroups with ONE wavefront per SIMD — 256 VGPRs + AGPRs as
//  spill space, no scratch; with two wavefronts per SIMD it spills 117 registers to scratch and runs 1.5x slower)
template <typename T, int R, int XW, bool FUSE, int CORR, int NW>
__global__ __launch_bounds__(64 * NW, (CORR && R == 2 && NW == 4 && sizeof(T) == 8) ? 1 : 2) void k_flux128(FluxArgs a) {
  constexpr unsigned EB = (unsigned)sizeof(T);
  constexpr int WC = 128;  // columns per wavefront
  const T* const a_u = static_cast<const T*>(a.u);
  const T* const a_pI = static_cast<const T*>(a.pI);
  T* const a_F = static_cast<T*>(a.F);
  static_assert(R + 3 <= 8, "packed halo rows live in 8-lane groups");
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
  const int x0 = (txi * XW + wx) * WC;         // interior (0-based) column of lane 0's first element
  const int jb0 = (tyi * (NW / XW) + wy) * R;  // interior row of the first output row
  const int k0 = a.kB ? (tzi ? a.kB : a.k_lo) : a.k_lo + tzi * a.zc;  // padded plane index of the first output plane
  const int k1 = a.kB ? k0 + a.zc : min(k0 + a.zc, a.k_hi);
  if (x0 >= n0 || jb0 >= n1) {  // wavefront outside the box: it only keeps the workgroup's barrier count (one per plane)
    if (a.bar)
      for (int k = k0; k < k1; ++k) __builtin_amdgcn_s_barrier();
    return;
  }
  const long long sz = (long long)N0 * N1;
  const int ci = x0 + 2 * lane;  // interior column of element a (n0 is even: a and b are inside the box together)
  const bool xout = ci < n0;
  const DirT<T> X(a.X), Y(a.Y), Z(a.Z);

  auto prow_of = [&](int jr) { return CORR ? wrapi(jr, n1) + 1 : min(jr + 1, N1 - 1); };
  auto pcol_of = [&](int c) { return CORR ? wrapi(c, n0) + 1 : min(c + 1, N0 - 1); };
  const int cic = xout ? ci : 0;  // lanes past the box read (and never store) the first pair

  unsigned urow[R + 2];  // u: byte offset of the padded row inside a plane
  unsigned qrow[R + 3];  // p: byte offset of the interior row inside an unpadded plane (CORR)
#pragma unroll
  for (int rr = 0; rr < R + 2; ++rr) urow[rr] = (unsigned)(prow_of(jb0 - 1 + rr) * N0) * EB;
  if (CORR) {
#pragma unroll
    for (int rr = 0; rr < R + 3; ++rr) qrow[rr] = (unsigned)((prow_of(jb0 - 1 + rr) - 1) * n0) * EB;
  }
  const unsigned ubytes = (unsigned)sz * EB, qbytes = (unsigned)(n0 * n1) * EB;
  const unsigned ucol = (unsigned)(cic + 1) * EB;  // padded column of element a
  const unsigned qcol = (unsigned)cic * EB;
  unsigned uhoff, qhoff = 0;  // packed halo loads: in-plane byte offset of this lane's (row, column)
  {
    const int r = lane & 7, grp = (lane >> 3) & 3;
    const int ru = r <= R + 1 ? r : 0;
    const int colu = (grp == 2) ? pcol_of(x0 + WC) : pcol_of(x0 - 1);
    uhoff = (unsigned)(prow_of(jb0 - 1 + ru) * N0 + colu) * EB;
    if (CORR) {
      const int rq = r <= R + 2 ? r : 0;
      const int cq = grp == 0 ? x0 - 1 : (grp == 1 ? x0 : (grp == 2 ? x0 + WC : x0 + WC + 1));
      qhoff = (unsigned)((prow_of(jb0 - 1 + rq) - 1) * n0 + (pcol_of(cq) - 1)) * EB;
    }
  }

  auto uplane = [&](int kk) { return CORR == 1 ? wrapi(kk - 1, n2) + 1 : (CORR == 2 ? min(kk, N2 - 1) : kk); };
  auto load_plane = [&](Plane2<T, R>& P, int kk) {
    const T* base = a_u + (long long)uplane(kk) * sz;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const rsrc_t rs = plane_rsrc(base + c * a.sc, ubytes);
#pragma unroll
      for (int rr = 0; rr < R + 2; ++rr) P.v[c][rr] = ldb2<T>(rs, ucol, urow[rr]);
      P.h[c] = ldb<T>(rs, uhoff, 0);
    }
  };
  auto load_p = [&](P2<T> (&P)[R + 3], T& PH, int kk) {
    const rsrc_t rs = plane_rsrc(a_pI + (long long)(CORR == 2 ? min(kk, N2) : wrapi(kk - 1, n2)) * n0 * n1, qbytes);
#pragma unroll
    for (int rr = 0; rr < R + 3; ++rr) P[rr] = ldb2<T>(rs, qcol, qrow[rr]);
    PH = ldb<T>(rs, qhoff, 0);
  };
  // u = u* - ∇p (applypressure!, operators.jl:225-233) for one register plane and its packed halo columns
  auto correct = [&](Plane2<T, R>& P, const P2<T> (&Pc)[R + 3], T PHc, const P2<T> (&Pn)[R + 3], T PHn) {
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr) {
      const T pa = Pc[rr].a, pb = Pc[rr].b;
      P.v[0][rr].a -= (pb - pa) * X.gs;
      P.v[0][rr].b -= (next_h(pa, rdlane(PHc, 16 + rr)) - pb) * X.gs;
      P.v[1][rr].a -= (Pc[rr + 1].a - pa) * Y.gs;
      P.v[1][rr].b -= (Pc[rr + 1].b - pb) * Y.gs;
      P.v[2][rr].a -= (Pn[rr].a - pa) * Z.gs;
      P.v[2][rr].b -= (Pn[rr].b - pb) * Z.gs;
    }
    P.h[0] -= (dpp_old<0x108>(PHc, PHc) - PHc) * X.gs;  // row_shl:8 — p of the next column, same row
    P.h[1] -= (dpp_old<0x101>(PHc, PHc) - PHc) * Y.gs;  // row_shl:1 — p of the next row, same column
    P.h[2] -= (PHn - PHc) * Z.gs;
  };

  P2<T> zprev[3][R];
  auto zflux0 = [&](const Plane2<T, R>& C, const Plane2<T, R>& Nx) {  // upper-face z-fluxes of the plane below the chunk
#pragma unroll
    for (int rr = 1; rr <= R; ++rr) {
      const T Wa = C.v[2][rr].a, Wb = C.v[2][rr].b;
      const T Wn = next_h(Wa, rdlane(C.h[2], 16 + rr));
      zprev[0][rr - 1].a = flux(C.v[0][rr].a, Nx.v[0][rr].a, Wa, Wb, Z.vo);
      zprev[0][rr - 1].b = flux(C.v[0][rr].b, Nx.v[0][rr].b, Wb, Wn, Z.vo);
      zprev[1][rr - 1].a = flux(C.v[1][rr].a, Nx.v[1][rr].a, Wa, C.v[2][rr + 1].a, Z.vo);
      zprev[1][rr - 1].b = flux(C.v[1][rr].b, Nx.v[1][rr].b, Wb, C.v[2][rr + 1].b, Z.vo);
      zprev[2][rr - 1].a = flux(Wa, Nx.v[2][rr].a, Wa, Nx.v[2][rr].a, Z.vs);
      zprev[2][rr - 1].b = flux(Wb, Nx.v[2][rr].b, Wb, Nx.v[2][rr].b, Z.vs);
    }
  };

  // output rows of this wavefront (clamped: rows / columns past the box are computed but never stored)
  unsigned orow[R];
#pragma unroll
  for (int rr = 0; rr < R; ++rr) orow[rr] = (unsigned)((min(jb0 + rr, n1 - 1) + 1) * N0) * EB;
  const unsigned ocol = ucol;

  // RK epilogue, first half: s = ustart + Σ_q coef_q k_q for all R rows of plane k, issued at the top of the plane (the loads fly during the flux
  // arithmetic).  The stencil input itself as a term (epi.self_in, stage-velocity basis) is read back from memory here — its uncorrected value left the
  // registers when the plane was corrected, and a second copy of it would not fit beside two register planes of pairs; the row was fetched two planes ago.
  auto epi_load = [&](const Plane2<T, R>& C, int k, P2<T> (&sacc)[3][R]) {
    const long long pk = (long long)k * sz;
    if (a.epi.ustart) {
      const T* b = static_cast<const T*>((const void*)a.epi.ustart) + pk;
      const T c0 = (T)(1.0 + a.epi.c0m1);  // exactly 1 in the k-basis
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const rsrc_t rs = plane_rsrc(b + c * a.sc, ubytes);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          const P2<T> v = ldb2<T>(rs, ocol, orow[rr]);
          sacc[c][rr].a = c0 * v.a;
          sacc[c][rr].b = c0 * v.b;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) sacc[c][rr] = C.v[c][rr + 1];
    }
    if (a.epi.self_in != 0.0) {
      const T cs = (T)a.epi.self_in;
      if (CORR) {
        const T* b = a_u + (long long)uplane(k) * sz;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const rsrc_t rs = plane_rsrc(b + c * a.sc, ubytes);
#pragma unroll
          for (int rr = 0; rr < R; ++rr) {
            const P2<T> v = ldb2<T>(rs, ucol, urow[rr + 1]);
            sacc[c][rr].a += cs * v.a;
            sacc[c][rr].b += cs * v.b;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) {
            sacc[c][rr].a += cs * C.v[c][rr + 1].a;
            sacc[c][rr].b += cs * C.v[c][rr + 1].b;
          }
      }
    }
    for (int q = 0; q < a.epi.n; ++q) {
      const T* kq = static_cast<const T*>((const void*)a.epi.k[q]) + pk;
      const T cq = (T)a.epi.coef[q];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const rsrc_t rs = plane_rsrc(kq + c * a.sc, ubytes);
        P2<T> kv[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) kv[rr] = ldb2<T>(rs, ocol, orow[rr]);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          sacc[c][rr].a += cq * kv[rr].a;
          sacc[c][rr].b += cq * kv[rr].b;
        }
      }
    }
  };
  // second half: u* = s + coef_self f, and k_i = f when a later stage needs it
  auto emit = [&](int rr, int k, const P2<T>& fu, const P2<T>& fv, const P2<T>& fw, const P2<T>& s0, const P2<T>& s1, const P2<T>& s2) {
    const long long pk = (long long)k * sz;
    const int j = jb0 + rr - 1;  // interior row
    if (xout && j < n1) {
      const unsigned rowb = orow[rr - 1], co = ocol;
      if (FUSE) {
        if (CORR && a.epi.ustart_out) {  // chained steps: s is the corrected stencil input = this step's ustart (no term was added to it)
          T* w = static_cast<T*>((void*)a.epi.ustart_out) + pk;
          stb2(plane_rsrc(w, ubytes), co, rowb, s0.a, s0.b);
          stb2(plane_rsrc(w + a.sc, ubytes), co, rowb, s1.a, s1.b);
          stb2(plane_rsrc(w + 2 * a.sc, ubytes), co, rowb, s2.a, s2.b);
        }
        T* o = static_cast<T*>((void*)a.epi.ustar) + pk;
        const T cs = (T)a.epi.coef_self;
        stb2(plane_rsrc(o, ubytes), co, rowb, s0.a + cs * fu.a, s0.b + cs * fu.b);
        stb2(plane_rsrc(o + a.sc, ubytes), co, rowb, s1.a + cs * fv.a, s1.b + cs * fv.b);
        stb2(plane_rsrc(o + 2 * a.sc, ubytes), co, rowb, s2.a + cs * fw.a, s2.b + cs * fw.b);
      }
      if (!FUSE || a.epi.write_k) {
        T* o = a_F + pk;
        stb2(plane_rsrc(o, ubytes), co, rowb, fu.a, fu.b);
        stb2(plane_rsrc(o + a.sc, ubytes), co, rowb, fv.a, fv.b);
        stb2(plane_rsrc(o + 2 * a.sc, ubytes), co, rowb, fw.a, fw.b);
      }
    }
  };

  // One output plane.  C = plane k, Nx = plane k+1 (both complete, corrected).  As soon as a row of C has been consumed its registers are re-loaded
  // with the same row of plane `kload` (= k+2): the prefetch of plane k+2 is in flight during the whole of plane k without a third register plane.
  auto body = [&](Plane2<T, R>& C, const Plane2<T, R>& Nx, int k, int kload) {
    const T* nb = a_u + (long long)uplane(kload) * sz;
    const rsrc_t n0r = plane_rsrc(nb, ubytes), n1r = plane_rsrc(nb + a.sc, ubytes), n2r = plane_rsrc(nb + 2 * a.sc, ubytes);
    const T ch0 = C.h[0], ch1 = C.h[1], ch2 = C.h[2];
    P2<T> sacc[3][FUSE ? R : 1];
    if constexpr (FUSE) epi_load(C, k, sacc);
    P2<T> fyu_o = {0, 0}, fyv_o = {0, 0}, fyw_o = {0, 0};
#pragma unroll
    for (int rr = 0; rr <= R; ++rr) {
      const T Ua = C.v[0][rr].a, Ub = C.v[0][rr].b, Va = C.v[1][rr].a, Vb = C.v[1][rr].b, Wa = C.v[2][rr].a, Wb = C.v[2][rr].b;
      const T Vn = next_h(Va, rdlane(ch1, 16 + rr));  // v at the column right of b
      // y-fluxes through the face between rows rr and rr+1
      P2<T> fyu, fyv, fyw;
      fyu.a = flux(Ua, C.v[0][rr + 1].a, Va, Vb, Y.vo);
      fyu.b = flux(Ub, C.v[0][rr + 1].b, Vb, Vn, Y.vo);
      fyv.a = flux(Va, C.v[1][rr + 1].a, Va, C.v[1][rr + 1].a, Y.vs);
      fyv.b = flux(Vb, C.v[1][rr + 1].b, Vb, C.v[1][rr + 1].b, Y.vs);
      fyw.a = flux(Wa, C.v[2][rr + 1].a, Va, Nx.v[1][rr].a, Y.vo);
      fyw.b = flux(Wb, C.v[2][rr + 1].b, Vb, Nx.v[1][rr].b, Y.vo);
      if (rr >= 1) {
        const T Un = next_h(Ua, rdlane(ch0, 16 + rr)), Wn = next_h(Wa, rdlane(ch2, 16 + rr));
        // x-fluxes through the right faces of a and of b
        const T fxu_a = flux(Ua, Ub, Ua, Ub, X.vs), fxu_b = flux(Ub, Un, Ub, Un, X.vs);
        const T fxv_a = flux(Va, Vb, Ua, C.v[0][rr + 1].a, X.vo), fxv_b = flux(Vb, Vn, Ub, C.v[0][rr + 1].b, X.vo);
        const T fxw_a = flux(Wa, Wb, Ua, Nx.v[0][rr].a, X.vo), fxw_b = flux(Wb, Wn, Ub, Nx.v[0][rr].b, X.vo);
        // left face of lane 0's a from the halo column x0-1 (every other lane takes the next-lower lane's right face of b)
        const T sU = rdlane(ch0, rr), sV = rdlane(ch1, rr), sW = rdlane(ch2, rr);
        const T sUu = rdlane(ch0, rr + 1), sUn = rdlane(Nx.h[0], rr);
        const T lxu = flux(sU, Ua, sU, Ua, X.vs);
        const T lxv = flux(sV, Va, sU, sUu, X.vo);
        const T lxw = flux(sW, Wa, sU, sUn, X.vo);
        P2<T> fu, fv, fw;
        fu.a = (fxu_a - prev_h(fxu_b, lxu)) * X.rs;
        fu.b = (fxu_b - fxu_a) * X.rs;
        fv.a = (fxv_a - prev_h(fxv_b, lxv)) * X.ro;
        fv.b = (fxv_b - fxv_a) * X.ro;
        fw.a = (fxw_a - prev_h(fxw_b, lxw)) * X.ro;
        fw.b = (fxw_b - fxw_a) * X.ro;
        fu.a += (fyu.a - fyu_o.a) * Y.ro;
        fu.b += (fyu.b - fyu_o.b) * Y.ro;
        fv.a += (fyv.a - fyv_o.a) * Y.rs;
        fv.b += (fyv.b - fyv_o.b) * Y.rs;
        fw.a += (fyw.a - fyw_o.a) * Y.ro;
        fw.b += (fyw.b - fyw_o.b) * Y.ro;
        P2<T> zu, zv, zw;
        zu.a = flux(Ua, Nx.v[0][rr].a, Wa, Wb, Z.vo);
        zu.b = flux(Ub, Nx.v[0][rr].b, Wb, Wn, Z.vo);
        zv.a = flux(Va, Nx.v[1][rr].a, Wa, C.v[2][rr + 1].a, Z.vo);
        zv.b = flux(Vb, Nx.v[1][rr].b, Wb, C.v[2][rr + 1].b, Z.vo);
        zw.a = flux(Wa, Nx.v[2][rr].a, Wa, Nx.v[2][rr].a, Z.vs);
        zw.b = flux(Wb, Nx.v[2][rr].b, Wb, Nx.v[2][rr].b, Z.vs);
        fu.a += (zu.a - zprev[0][rr - 1].a) * Z.ro;
        fu.b += (zu.b - zprev[0][rr - 1].b) * Z.ro;
        fv.a += (zv.a - zprev[1][rr - 1].a) * Z.ro;
        fv.b += (zv.b - zprev[1][rr - 1].b) * Z.ro;
        fw.a += (zw.a - zprev[2][rr - 1].a) * Z.rs;
        fw.b += (zw.b - zprev[2][rr - 1].b) * Z.rs;
        zprev[0][rr - 1] = zu;
        zprev[1][rr - 1] = zv;
        zprev[2][rr - 1] = zw;
        if constexpr (FUSE)
          emit(rr, k, fu, fv, fw, sacc[0][rr - 1], sacc[1][rr - 1], sacc[2][rr - 1]);
        else
          emit(rr, k, fu, fv, fw, fu, fu, fu);
      }
      fyu_o = fyu;
      fyv_o = fyv;
      fyw_o = fyw;
      // row rr of plane k is dead: its registers receive plane `kload`
      C.v[0][rr] = ldb2<T>(n0r, ucol, urow[rr]);
      C.v[1][rr] = ldb2<T>(n1r, ucol, urow[rr]);
      C.v[2][rr] = ldb2<T>(n2r, ucol, urow[rr]);
    }
    C.v[0][R + 1] = ldb2<T>(n0r, ucol, urow[R + 1]);
    C.v[1][R + 1] = ldb2<T>(n1r, ucol, urow[R + 1]);
    C.v[2][R + 1] = ldb2<T>(n2r, ucol, urow[R + 1]);
    C.h[0] = ldb<T>(n0r, uhoff, 0);
    C.h[1] = ldb<T>(n1r, uhoff, 0);
    C.h[2] = ldb<T>(n2r, uhoff, 0);
  };

  // Two register planes.  Loads past the chunk re-read plane k1 / p(k1+1) (cache hits) instead of branching.
  Plane2<T, R> P0, P1;
  if (!CORR) {
    load_plane(P0, k0 - 1);
    load_plane(P1, k0);
    zflux0(P0, P1);
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
    P2<T> Pa[R + 3], Pb[R + 3];
    T Ha, Hb;
    load_p(Pa, Ha, k0 - 1);
    load_p(Pb, Hb, k0);
    load_plane(P0, k0 - 1);
    load_plane(P1, k0);
    correct(P0, Pa, Ha, Pb, Hb);
    load_p(Pa, Ha, k0 + 1);
    correct(P1, Pb, Hb, Pa, Ha);
    zflux0(P0, P1);
    load_plane(P0, min(k0 + 1, k1));
    load_p(Pb, Hb, min(k0 + 2, k1 + 1));
    int k = k0;
    while (true) {
      if (a.bar) __builtin_amdgcn_s_barrier();
      correct(P0, Pa, Ha, Pb, Hb);  // plane k+1 with p(k+1), p(k+2)
      load_p(Pa, Ha, min(k + 3, k1 + 1));
      body(P1, P0, k, min(k + 2, k1));
      if (++k >= k1) break;
      if (a.bar) __builtin_amdgcn_s_barrier();
      correct(P1, Pb, Hb, Pa, Ha);
      load_p(Pb, Hb, min(k + 3, k1 + 1));
      body(P0, P1, k, min(k + 2, k1));
      if (++k >= k1) break;
    }
  }
}

template <typename T, int R, int XW, bool FUSE, int NW>
int launch_range(FluxArgs& a, int corr_mode, hipStream_t s) {
  const unsigned nb = (unsigned)(8LL * a.ntx * ((a.nty + 7) / 8) * a.ntz);
  if (nb == 0) return INS_OK;
  const dim3 block(64, NW, 1);
  if (corr_mode == 0)
    hipLaunchKernelGGL((k_flux128<T, R, XW, FUSE, 0, NW>), dim3(nb), block, 0, s, a);
  else if constexpr (FUSE) {
    if (corr_mode == 1)
      hipLaunchKernelGGL((k_flux128<T, R, XW, true, 1, NW>), dim3(nb), block, 0, s, a);
    else
      hipLaunchKernelGGL((k_flux128<T, R, XW, true, 2, NW>), dim3(nb), block, 0, s, a);
  } else {
    ins_set_error("in-kernel pressure correction needs the fused epilogue");
    return INS_ERR_UNSUPPORTED;
  }
  INS_LAUNCH_CHECK();
  return INS_OK;
}

// part 0: every plane; 1: the planes that read no ghost plane, [1 + ZB, nzl + 1 - ZB); 2: the two boundary ranges of ZB planes (as ins_flux64.hip)
constexpr int ZB = 4;
template <typename T, int R, int XW, bool FUSE, int NW>
int launch(const ins_grid* G, FluxArgs& a, int corr_mode, int part, hipStream_t s) {
  const GridDev& g = G->g;
  a.ntx = cdiv(g.N[0] - 2, 128 * XW);
  a.nty = cdiv(g.N[1] - 2, (NW / XW) * R);
  const int nzl = g.N[2] - 2;
  a.k_lo = 1;
  a.k_hi = nzl + 1;
  a.kB = 0;
  if (part != 0 && nzl <= 2 * ZB) {
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
  return launch_range<T, R, XW, FUSE, NW>(a, corr_mode, s);
}

template <typename T>
int flux128_dispatch(const ins_grid* G, double visc, const T* u, T* F, const RkEpi* epi, const T* pI, int corr_mode, hipStream_t s, int part) {
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
  const int n0 = g.N[0] - 2, n1 = g.N[1] - 2, n2 = g.N[2] - 2;
  a.X = make_dir(G, 0, visc);
  a.Y = make_dir(G, 1, visc);
  a.Z = make_dir(G, 2, visc);
  if (epi) a.epi = *epi;
  const int waves_x = cdiv(n0, 128);
  // wavefronts side by side: a whole row per workgroup up to 512 columns; counts that leave no wavefront outside the box (3 wavefronts: 384 columns)
  const int xwo = (int)ins_opt(OPT_INS_FLUX128_XW);
  int xw = (xwo == 1 || xwo == 2 || xwo == 4) ? xwo : (waves_x >= 4 ? 4 : (waves_x >= 2 ? 2 : 1));
  if (!xwo)
    while (xw > 1 && cdiv(waves_x, xw) * xw > waves_x) xw >>= 1;
  // rows per lane: 2 (x 2 columns: the register budget of four single-column rows); the correcting kernel runs 1 unless INS_FLUX128_ROWS_CORR=2
  int rows = 2;
  if (corr_mode && ins_opt(OPT_INS_FLUX128_ROWS_CORR) != 2) {
    // fp32: two rows of pairs where 8-wavefront workgroups with 64-plane chunks still give every CU a tile (512^3: RK44 step 14.2 -> 13.6 ms), one row on
    // smaller boxes (256^3: 1.95 -> 1.91 ms with one row, 2.00 with two)
    const bool big = (long long)cdiv(n0, 128 * xw) * cdiv(n1, (8 / xw) * 2) * cdiv(n2, 64) >= 256;
    if (!(sizeof(T) == 4 && ins_opt(OPT_INS_FLUX128_ROWS_CORR) == 0 && big)) rows = 1;
  }
  if (!corr_mode && ins_opt(OPT_INS_FLUX128_ROWS) == 1) rows = 1;
  // workgroup size and z-chunk: 8 wavefronts and 64-plane chunks when that gives every CU a workgroup, else 4 wavefronts and shorter chunks
  // (the rule of ins_flux64.hip: every chunk re-reads two planes, the wavefronts of a workgroup share halo rows under the per-plane barrier)
  // (INS_FLUX64_NW / _ZC / _ZC_CORR, the knobs of the one-column kernel, are honoured when the two-column ones are unset: tests force tile shapes with them)
  const int nwo = ins_opt(OPT_INS_FLUX128_NW) ? (int)ins_opt(OPT_INS_FLUX128_NW) : std::min((int)ins_opt(OPT_INS_FLUX64_NW), 8);
  const int zco = ins_opt(OPT_INS_FLUX128_ZC) ? (int)ins_opt(OPT_INS_FLUX128_ZC)
                                              : (int)((corr_mode && ins_opt(OPT_INS_FLUX64_ZC_CORR)) ? ins_opt(OPT_INS_FLUX64_ZC_CORR) : ins_opt(OPT_INS_FLUX64_ZC));
  const long long mintiles = ins_opt(OPT_INS_FLUX64_MINTILES) > 0 ? ins_opt(OPT_INS_FLUX64_MINTILES) : 256;
  auto tiles = [&](int nw_, int zc_) { return (long long)cdiv(n0, 128 * xw) * cdiv(n1, (nw_ / xw) * rows) * cdiv(n2, zc_); };
  int nw = (nwo == 8 || nwo == 4) ? nwo : 0;
  if (nw && nw < xw) nw = xw;
  int zc = zco;
  if (!nw) {
    nw = std::max(4, xw);
    const int zt = zco ? zco : 64;
    if (n2 >= zt && tiles(8, zt) >= mintiles) {
      nw = 8;
      zc = zt;
    }
  }
  if (!zc) {
    zc = n2 >= 128 ? 32 : (n2 >= 64 ? 16 : (n2 >= 32 ? 8 : 4));
    if (n2 >= 256 && tiles(nw, 64) >= mintiles) zc = 64;
    if (n2 < 256)
      while (zc > 4 && tiles(nw, zc) < 2 * mintiles) zc >>= 1;
  }
  if (corr_mode && rows == 2 && sizeof(T) == 8 && nw == 8 && !nwo) nw = 4;  // (see k_flux128: the register budget of the correcting form)
  a.zc = zc;
  a.bar = ins_opt(OPT_INS_FLUX64_NOBAR) ? 0 : 1;
#define INS_F128_CASE(RR, FUSE)                                               \
  if (rows == RR) {                                                           \
    if (nw >= 8) {                                                            \
      if (xw == 4) return launch<T, RR, 4, FUSE, 8>(G, a, corr_mode, part, s); \
      if (xw == 2) return launch<T, RR, 2, FUSE, 8>(G, a, corr_mode, part, s); \
      return launch<T, RR, 1, FUSE, 8>(G, a, corr_mode, part, s);             \
    }                                                                         \
    if (xw == 4) return launch<T, RR, 4, FUSE, 4>(G, a, corr_mode, part, s);  \
    if (xw == 2) return launch<T, RR, 2, FUSE, 4>(G, a, corr_mode, part, s);  \
    return launch<T, RR, 1, FUSE, 4>(G, a, corr_mode, part, s);               \
  }
  if (epi) {
    INS_F128_CASE(2, true)
    INS_F128_CASE(1, true)
  } else {
    INS_F128_CASE(2, false)
    INS_F128_CASE(1, false)
  }
#undef INS_F128_CASE
  return INS_ERR_INVALID;
}

}  // namespace

// Boxes the two-column kernels take: what ins_flux64.hip takes (3-D, every interior volume a DOF, bitwise-constant metric records) with an even number of
// volumes per row and room for the periodic wrap of a full 128-column window; no extended-loop epilogue terms (those stay on ins_flux64.hip's EXTRA form).
// The correcting forms (corr_mode != 0) are built and parity-tested but not the default: measured same-box they equal the one-column kernel with one row of
// pairs (256^3 step 2.596 vs 2.610 ms, 512^3 21.84 vs 21.81) and lose with two (2.68 ms at one wavefront per SIMD, 3.99 ms with scratch spills) — the stage
// kernels are bound below L2, not by vector-memory issue (profiles/r03_k1_pmc_512.txt).  INS_FLUX128_CORR=1 selects them.
// In fp32 the correcting form with two rows of pairs fits (218 VGPRs, two wavefronts per SIMD) and is the default: 8 B per lane instead of 4.
bool ins_flux128_supported(const ins_grid* G, const RkEpi* epi, int corr_mode, bool f32) {
  const GridDev& g = G->g;
  if (ins_opt(OPT_INS_DISABLE_FLUX128) || ins_opt(OPT_INS_DISABLE_FLUX64) || ins_opt(OPT_INS_FLUX64_SKEL)) return false;
  if (corr_mode && !f32 && !ins_opt(OPT_INS_FLUX128_CORR)) return false;
  if (epi && (epi->extra || epi->gtemp || epi->wout || epi->tstage)) return false;
  const int n0 = g.N[0] - 2;
  return g.D == 3 && G->all_dof && G->uniform_exact && n0 >= 130 && n0 % 2 == 0 && g.N[1] - 2 >= 8 && g.N[2] - 2 >= 4;
}

int ins_k_flux128(const ins_grid* G, double visc, const double* u, double* F, const RkEpi* epi, const double* pI, int corr_mode, hipStream_t s, int part) {
  return flux128_dispatch<double>(G, visc, u, F, epi, pI, corr_mode, s, part);
}
int ins_k_flux128_f32(const ins_grid* G, double visc, const float* u, float* F, const RkEpi* epi, const float* pI, int corr_mode, hipStream_t s, int part) {
  return flux128_dispatch<float>(G, visc, u, F, epi, pI, corr_mode, s, part);
}
