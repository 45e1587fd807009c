// Developer probe: relative error of v_rsq_f64 (the seed of the pivot 1/sqrt in the diagonal-block kernels) and of the
// refinements built on it.   hipcc --offload-arch=gfx950 -O2 rsq_probe.hip -o rsq_probe && ./rsq_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* p, double* y0, double* y3, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = p[i];
  const double s = __builtin_amdgcn_rsq(x);
  y0[i] = s;
  const double t = x * s;
  const double e = fma(-t, s, 1.0);
  y3[i] = fma(s * e, fma(0.375, e, 0.5), s);  // third order: s (1 + e/2 + 3 e^2 / 8)
}

int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), a(n), b(n);
  unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    const double u = (double)(st >> 11) / 9007199254740992.0;
    h[i] = std::ldexp(0.5 + u, (int)(st % 41) - 20);
  }
  double *dp, *d0, *d3;
  hipMalloc(&dp, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d3, n * 8);
  hipMemcpy(dp, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dp, d0, d3, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d3, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e3 = 0;
  for (int i = 0; i < n; ++i) {
    const long double r = 1.0L / sqrtl((long double)h[i]);
    e0 = std::fmax(e0, (double)fabsl((a[i] - r) / r));
    e3 = std::fmax(e3, (double)fabsl((b[i] - r) / r));
  }
  printf("v_rsq_f64 max rel err %.3e (2^%.1f); third-order refinement %.3e (%.2f ulp)\n", e0, std::log2(e0), e3, e3 / 1.11e-16);
  return 0;
}
