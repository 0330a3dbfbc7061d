// rocFFT (ROCm 7.2, gfx950) reproducer: batched 2-D real transforms go wrong depending on which plans were
// created earlier in the process.  Each case creates D2Z (+ optionally Z2D) plans and checks the forward result
// of a few batches against a naive DFT.   usage: dbg_hipfft [with_inverse_plan=1] [destroy=1]
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double fwd_err(int batch, int ny, int nx, bool inv_plan, bool destroy) {
  const int kxn = nx / 2 + 1;
  std::vector<double> in((size_t)batch * ny * nx);
  unsigned s = 12345u + batch * 7 + ny * 131 + nx;
  for (auto& v : in) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
  double* din; hipfftDoubleComplex* dout;
  hipMalloc(&din, in.size() * 8); hipMalloc(&dout, (size_t)batch * ny * kxn * 16);
  hipMemcpy(din, in.data(), in.size() * 8, hipMemcpyHostToDevice);
  hipfftHandle pf, pi = 0; int n2[2] = {ny, nx};
  hipfftPlanMany(&pf, 2, n2, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_D2Z, batch);
  if (inv_plan) hipfftPlanMany(&pi, 2, n2, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_Z2D, batch);
  hipfftExecD2Z(pf, din, dout); hipDeviceSynchronize();
  std::vector<std::complex<double>> out((size_t)batch * ny * kxn);
  hipMemcpy(out.data(), dout, out.size() * 16, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (int b = 0; b < batch; b += std::max(1, batch / 2))
    for (int ky = 0; ky < ny; ++ky)
      for (int kx = 0; kx < kxn; ++kx) {
        std::complex<double> acc = 0;
        for (int y = 0; y < ny; ++y)
          for (int x = 0; x < nx; ++x) {
            const double ph = -2 * M_PI * ((double)ky * y / ny + (double)kx * x / nx);
            acc += in[((size_t)b * ny + y) * nx + x] * std::complex<double>(cos(ph), sin(ph));
          }
        num += std::norm(acc - out[((size_t)b * ny + ky) * kxn + kx]); den += std::norm(acc);
      }
  if (destroy) { hipfftDestroy(pf); if (inv_plan) hipfftDestroy(pi); }
  hipFree(din); hipFree(dout);
  return sqrt(num / den);
}
int main(int argc, char** argv) {
  const bool inv_plan = argc > 1 ? atoi(argv[1]) : 1, destroy = argc > 2 ? atoi(argv[2]) : 1;
  struct C { int b, ny, nx; } cases[] = {{16, 32, 32}, {16, 16, 64}, {16, 6, 10}, {16, 16, 64}};
  for (auto c : cases) printf("inv_plan=%d destroy=%d  batch %d ny %d nx %d : forward relerr vs DFT %.3e\n", (int)inv_plan, (int)destroy, c.b, c.ny, c.nx, fwd_err(c.b, c.ny, c.nx, inv_plan, destroy));
  return 0;
}
