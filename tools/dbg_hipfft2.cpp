// Does rocfft_cleanup()+rocfft_setup() clear the poisoned plan cache?
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <rocfft/rocfft.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
struct Pair { hipfftHandle f, i; };
static Pair make(int b, int ny, int nx) { Pair p; int n2[2] = {ny, nx};
  hipfftPlanMany(&p.f, 2, n2, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, b);
  hipfftPlanMany(&p.i, 2, n2, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, b); return p; }
static void kill(Pair p) { hipfftDestroy(p.f); hipfftDestroy(p.i); }
static double fwd_err(Pair p, int batch, int ny, int nx) {
  const int kxn = nx / 2 + 1;
  std::vector<double> in((size_t)batch * ny * nx);
  unsigned s = 99u; for (auto& v : in) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
  double* din; hipfftDoubleComplex* dout;
  hipMalloc(&din, in.size() * 8); hipMalloc(&dout, (size_t)batch * ny * kxn * 16);
  hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice);
  hipfftExecD2Z(p.f, din, dout); hipDeviceSynchronize();
  std::vector<std::complex<double>> out((size_t)batch * ny * kxn);
  hipMemcpy(out.data(), dout, out.size() * 16, hipMemcpyDeviceToHost);
  double num = 0, den = 0; const int b = batch - 1;
  for (int ky = 0; ky < ny; ++ky) for (int kx = 0; kx < kxn; ++kx) {
    std::complex<double> acc = 0;
    for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
      const double ph = -2 * M_PI * ((double)ky * y / ny + (double)kx * x / nx);
      acc += in[((size_t)b * ny + y) * nx + x] * std::complex<double>(cos(ph), sin(ph)); }
    num += std::norm(acc - out[((size_t)b * ny + ky) * kxn + kx]); den += std::norm(acc); }
  hipFree(din); hipFree(dout); return sqrt(num / den);
}
int main() {
  Pair A = make(16, 32, 32); printf("A [32x32] alive: err %.2e\n", fwd_err(A, 16, 32, 32));
  Pair B = make(16, 16, 64); printf("B [16x64] created while A alive: err %.2e\n", fwd_err(B, 16, 16, 64));
  kill(A); kill(B);
  Pair B2 = make(16, 16, 64); printf("B' after destroying A and B: err %.2e\n", fwd_err(B2, 16, 16, 64));
  kill(B2);
  rocfft_cleanup(); rocfft_setup();
  Pair B3 = make(16, 16, 64); printf("B'' after rocfft_cleanup+setup: err %.2e\n", fwd_err(B3, 16, 16, 64));
  Pair A2 = make(16, 32, 32); printf("A' [32x32] created while B'' alive: err %.2e\n", fwd_err(A2, 16, 32, 32));
  Pair C = make(16, 32, 64); printf("C [32x64] created while A',B'' alive: err %.2e\n", fwd_err(C, 16, 32, 64));
  printf("B'' re-run: err %.2e\n", fwd_err(B3, 16, 16, 64));
  return 0;
}
