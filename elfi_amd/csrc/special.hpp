// Normal cdf and Owen's T function on the device, for the acquisition rules built on the variance of the unnormalised
// approximate posterior (elfi/methods/bo/acquisition.py:392-463, 795-821: MaxVar, RandMaxVar, ExpIntVar).  The reference
// evaluates scipy.stats.skewnorm.cdf there; with z = (x - loc) / scale that is  Phi(z) - 2 T(z, a)  [Owen 1956],
// T(h, a) = (1 / 2 pi) int_0^a exp(-h^2 (1 + x^2) / 2) / (1 + x^2) dx.  Every shape parameter on this path lies in [0, 1]
// (sigma_n / sqrt(sigma_n^2 + 2 v), sqrt((A - d) / (A + d))), so the integral is taken directly: Gauss-Legendre panels
// whose width shrinks with |h| (the integrand falls like exp(-h^2 x^2 / 2)), ten points each -- relative accuracy about
// 1e-13 or better, no case analysis.
#pragma once

#include <hip/hip_runtime.h>

namespace elfihip {

__device__ __forceinline__ double norm_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }

__device__ __forceinline__ double norm_pdf(double z) { return 0.39894228040143267794 * exp(-0.5 * z * z); }

// T(h, a) for 0 <= a <= 1 (any h)
__device__ inline double owens_t(double h, double a) {
  // 10-point Gauss-Legendre on [-1, 1]: abscissae (positive half) and weights
  const double gx[5] = {0.14887433898163121088, 0.43339539412924719080, 0.67940956829902440623,
                        0.86506336668898451073, 0.97390652851717172008};
  const double gw[5] = {0.29552422471475287017, 0.26926671930999635509, 0.21908636251598204400,
                        0.14945134915058059315, 0.06667134430868813759};
  if (!(a > 0.0)) return 0.0;
  const double ah = fabs(h) * a;
  int panels = 1 + (int)(0.75 * ah);       // panel width <= 4 / (3 |h|); checked against scipy.special.owens_t over
  if (panels > 96) panels = 96;            // |h| <= 37: 2e-13 relative (the rounding of exp at arguments near -700)
  const double w = a / panels, hh = -0.5 * h * h;
  double sum = 0.0;
  for (int p = 0; p < panels; ++p) {
    const double mid = (p + 0.5) * w, half = 0.5 * w;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const double x1 = mid - half * gx[i], x2 = mid + half * gx[i];
      const double q1 = 1.0 + x1 * x1, q2 = 1.0 + x2 * x2;
      s += gw[i] * (exp(hh * q1) / q1 + exp(hh * q2) / q2);
    }
    sum += s * half;
  }
  return sum * 0.15915494309189533577;  // 1 / 2 pi
}

}  // namespace elfihip
