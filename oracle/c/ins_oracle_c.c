/* CPU restatement (plain C + OpenMP) of the reference's per-stage passes for PERIODIC 3-D boxes.
 * TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/ins_oracle.py header): used by tests as a second checker and by
 * bench.py's `cpu_baseline` leg.  The pass structure is the reference's own (unfused), one full-array pass per
 * statement of step_explicit_runge_kutta.jl:17-50; the FFT itself is done by the caller (scipy pocketfft).
 * Arrays are the reference layout: u[(N0,N1,N2,3)] column-major, x fastest. */
#include <stddef.h>
#include <string.h>

#define IDX(i, j, k) ((size_t)(i) + (size_t)N0 * ((size_t)(j) + (size_t)N1 * (size_t)(k)))

/* apply_bc_u! / apply_bc_p! for PeriodicBC, ncomp components          boundary_conditions.jl:276-288, 306-318 */
void oc_bc_periodic(double* f, int N0, int N1, int N2, int ncomp) {
  const size_t sc = (size_t)N0 * N1 * N2;
  for (int c = 0; c < ncomp; ++c) {
    double* u = f + c * sc;
#pragma omp parallel for collapse(2)
    for (int k = 0; k < N2; ++k)
      for (int j = 0; j < N1; ++j) {
        u[IDX(0, j, k)] = u[IDX(N0 - 2, j, k)];
        u[IDX(N0 - 1, j, k)] = u[IDX(1, j, k)];
      }
#pragma omp parallel for collapse(2)
    for (int k = 0; k < N2; ++k)
      for (int i = 0; i < N0; ++i) {
        u[IDX(i, 0, k)] = u[IDX(i, N1 - 2, k)];
        u[IDX(i, N1 - 1, k)] = u[IDX(i, 1, k)];
      }
#pragma omp parallel for collapse(2)
    for (int j = 0; j < N1; ++j)
      for (int i = 0; i < N0; ++i) {
        u[IDX(i, j, 0)] = u[IDX(i, j, N2 - 2)];
        u[IDX(i, j, N2 - 1)] = u[IDX(i, j, 1)];
      }
  }
}

/* momentum!: fill!(F, 0) then convection_diffusion_kernel!             operators.jl:967-976, 647-690
 * dx/dxu: Δ[α], Δu[α]; A1/A2[β*3+α]: A[β][α][1|2] (indexed along α). */
void oc_momentum(double* F, const double* u, double visc, int N0, int N1, int N2, const double* const dx[3], const double* const dxu[3],
                 const double* const A1[9], const double* const A2[9]) {
  const int N[3] = {N0, N1, N2};
  const size_t sc = (size_t)N0 * N1 * N2;
  const size_t st[3] = {1, (size_t)N0, (size_t)N0 * N1};
  const double eps2 = 2 * 2.220446049250313e-16;
  memset(F, 0, 3 * sc * sizeof(double));
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 1; k < N2 - 1; ++k)
    for (int j = 1; j < N1 - 1; ++j)
      for (int i = 1; i < N0 - 1; ++i) {
        const int I[3] = {i, j, k};
        const size_t c = IDX(i, j, k);
        for (int al = 0; al < 3; ++al) {
          const double* ua = u + al * sc;
          double f = F[al * sc + c];
          for (int be = 0; be < 3; ++be) {
            const double* ub = u + be * sc;
            const double dab = (al == be ? dxu[be] : dx[be])[I[be]];
            const double da = al == be ? dx[be][I[be]] : dxu[be][I[be] - 1];
            const double db = al == be ? dx[be][I[be] + 1] : dxu[be][I[be]];
            const double uab1 = (ua[c - st[be]] + ua[c]) / 2;
            const double uab2 = (ua[c] + ua[c + st[be]]) / 2;
            const double* a1 = A1[be * 3 + al];
            const double* a2 = A2[be * 3 + al];
            const double uba1 = a2[I[al] - (al == be)] * ub[c - st[be]] + a1[I[al] + (al != be)] * ub[c - st[be] + st[al]];
            const double uba2 = a2[I[al]] * ub[c] + a1[I[al] + 1] * ub[c + st[al]];
            double d1 = (ua[c] - ua[c - st[be]]) / da;
            double d2 = (ua[c + st[be]] - ua[c]) / db;
            d1 = da > eps2 ? d1 : 0.0;
            d2 = db > eps2 ? d2 : 0.0;
            f += (visc * (d2 - d1) - (uab2 * uba2 - uab1 * uba1)) / dab;
          }
          F[al * sc + c] = f;
        }
      }
  (void)N;
}

/* y = x  /  y += a*x   (the stage-combination broadcasts, step_explicit_runge_kutta.jl:35-38) */
void oc_copy(double* y, const double* x, size_t n) {
#pragma omp parallel for
  for (size_t t = 0; t < n; ++t) y[t] = x[t];
}
void oc_axpy(double* y, double a, const double* x, size_t n) {
#pragma omp parallel for
  for (size_t t = 0; t < n; ++t) y[t] += a * x[t];
}

/* divergence! then scalewithvolume! (two passes, as the reference)      operators.jl:106-125, 81-95 */
void oc_divergence(double* div, const double* u, int N0, int N1, int N2, const double* const dx[3]) {
  const size_t sc = (size_t)N0 * N1 * N2;
  const size_t st[3] = {1, (size_t)N0, (size_t)N0 * N1};
#pragma omp parallel for collapse(2)
  for (int k = 1; k < N2 - 1; ++k)
    for (int j = 1; j < N1 - 1; ++j)
      for (int i = 1; i < N0 - 1; ++i) {
        const int I[3] = {i, j, k};
        const size_t c = IDX(i, j, k);
        double d = 0.0;
        for (int a = 0; a < 3; ++a) d += (u[a * sc + c] - u[a * sc + c - st[a]]) / dx[a][I[a]];
        div[c] = d;
      }
}
void oc_scalewithvolume(double* p, int N0, int N1, int N2, const double* const dx[3]) {
#pragma omp parallel for collapse(2)
  for (int k = 0; k < N2; ++k)
    for (int j = 0; j < N1; ++j)
      for (int i = 0; i < N0; ++i) p[IDX(i, j, k)] *= dx[0][i] * dx[1][j] * dx[2][k];
}

/* copyto!(pI, view(p, Ip)) / copyto!(view(p, Ip), pI)                    pressure.jl:320, 347 */
void oc_strip(double* pI, const double* p, int N0, int N1, int N2) {
  const int n0 = N0 - 2, n1 = N1 - 2;
#pragma omp parallel for collapse(2)
  for (int k = 1; k < N2 - 1; ++k)
    for (int j = 1; j < N1 - 1; ++j)
      for (int i = 1; i < N0 - 1; ++i) pI[(size_t)(i - 1) + (size_t)n0 * ((size_t)(j - 1) + (size_t)n1 * (size_t)(k - 1))] = p[IDX(i, j, k)];
}
void oc_pad(double* p, const double* pI, int N0, int N1, int N2) {
  const int n0 = N0 - 2, n1 = N1 - 2;
#pragma omp parallel for collapse(2)
  for (int k = 1; k < N2 - 1; ++k)
    for (int j = 1; j < N1 - 1; ++j)
      for (int i = 1; i < N0 - 1; ++i) p[IDX(i, j, k)] = pI[(size_t)(i - 1) + (size_t)n0 * ((size_t)(j - 1) + (size_t)n1 * (size_t)(k - 1))];
}

/* phat = -phat / (ax + ay + az); phat[0] = 0   on the [kz][ky][kx] complex array     pressure.jl:326-341 */
void oc_symbol(double* phat, const double* ax, const double* ay, const double* az, int kxn, int n1, int n2) {
#pragma omp parallel for collapse(2)
  for (int k = 0; k < n2; ++k)
    for (int j = 0; j < n1; ++j)
      for (int i = 0; i < kxn; ++i) {
        const size_t q = 2 * ((size_t)i + (size_t)kxn * ((size_t)j + (size_t)n1 * (size_t)k));
        const double den = ax[i] + ay[j] + az[k];
        if (i == 0 && j == 0 && k == 0) {
          phat[q] = 0.0;
          phat[q + 1] = 0.0;
        } else {
          phat[q] = -phat[q] / den;
          phat[q + 1] = -phat[q + 1] / den;
        }
      }
}

/* applypressure!                                                          operators.jl:225-233 */
void oc_applypressure(double* u, const double* p, int N0, int N1, int N2, const double* const dxu[3]) {
  const size_t sc = (size_t)N0 * N1 * N2;
  const size_t st[3] = {1, (size_t)N0, (size_t)N0 * N1};
#pragma omp parallel for collapse(2)
  for (int k = 1; k < N2 - 1; ++k)
    for (int j = 1; j < N1 - 1; ++j)
      for (int i = 1; i < N0 - 1; ++i) {
        const int I[3] = {i, j, k};
        const size_t c = IDX(i, j, k);
        for (int a = 0; a < 3; ++a) u[a * sc + c] -= (p[c + st[a]] - p[c]) / dxu[a][I[a]];
      }
}
