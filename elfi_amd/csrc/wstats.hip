// Weighted column variance of an SMC-ABC population on gfx950.
//
// Replaces weighted_var (elfi/methods/utils.py:108-139), which SMC calls once per round on the accepted parameter
// sample to set the proposal covariance (elfi/methods/inference/samplers.py:521-534, "2 * diag(weighted_var)"):
//     V1 = sum w;  V2 = sum w^2;  xbar = sum_i w_i x_i / V1;  s2 = sum_i w_i (x_i - xbar)^2 / (V1 - V2 / V1)
// Two streaming passes over the (n, m) sample (the mean is needed before the squared deviations), 8 (m + 1) bytes per
// row each; SURVEY.md section 8f, rank 2.
//
// Determinism: thread (row group rg, column c) always owns the rows rg, rg + rpi * G, ... of its workgroup's slot, the
// row groups of a workgroup are added in order, then the workgroups in order: no atomics, same bits every run.
#include "common.hpp"

#pragma clang fp contract(off)

namespace elfihip {

struct WvarArgs {
  const double* X;
  const double* w;   // NULL: unit weights
  int64_t n, ldx;
  int m;
  int rpi;           // row groups per workgroup (256 / m, at least 1)
  const double* head;  // pass 2: [V1, V2, xbar[0..m)]
  double* partial;   // (gridDim, m + 2): pass 1 [sum w, sum w^2, sum w x_c], pass 2 [-, -, sum w (x_c - xbar_c)^2]
};

template <int PASS>
__global__ __launch_bounds__(256) void wvar_partial_kernel(WvarArgs A) {
  __shared__ double red[256];
  const int tid = threadIdx.x, m = A.m;
  const int cstride = m <= 256 ? m : 256;
  const int rg = m <= 256 ? tid / m : 0;
  const int c0 = m <= 256 ? tid - rg * m : tid;
  const bool active = rg < A.rpi;
  const int64_t rstride = (int64_t)gridDim.x * A.rpi;
  double* out = A.partial + (int64_t)blockIdx.x * (m + 2);
  // columns; the two weight sums ride along as "columns" m and m + 1 of pass 1
  const int ncol = PASS == 1 ? m + 2 : m;
  for (int cb = 0; cb < ncol; cb += cstride) {
    const int c = cb + c0;
    double acc = 0.0;
    if (active && c < ncol) {
      const double xb = (PASS == 2) ? A.head[2 + c] : 0.0;
      for (int64_t r = (int64_t)blockIdx.x * A.rpi + rg; r < A.n; r += rstride) {
        const double wr = A.w ? A.w[r] : 1.0;
        if (PASS == 1) {
          if (c < m)
            acc += wr * A.X[r * A.ldx + c];
          else
            acc += (c == m) ? wr : wr * wr;
        } else {
          const double dv = A.X[r * A.ldx + c] - xb;
          acc += wr * (dv * dv);
        }
      }
    }
    __syncthreads();
    red[tid] = acc;
    __syncthreads();
    if (rg == 0 && c < ncol) {  // row groups in order
      double s = red[tid];
      for (int g = 1; g < A.rpi; ++g) s += red[g * cstride + c0];
      if (PASS == 1)
        out[c < m ? 2 + c : c - m] = s;
      else
        out[2 + c] = s;
    }
  }
}

// PASS 1: head = [V1, V2, xbar]; PASS 2: s2[c] = sum / (V1 - V2 / V1)
template <int PASS>
__global__ void wvar_finish_kernel(const double* partial, int nblk, int m, double* head, double* s2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (PASS == 1) {
    if (c >= m + 2) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * (m + 2) + c];
    head[c] = s;  // sums first; the means are formed by the second launch below (needs V1 = head[0])
  } else {
    if (c >= m) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * (m + 2) + 2 + c];
    const double V1 = head[0], V2 = head[1];
    s2[c] = s / (V1 - V2 / V1);
  }
}

__global__ void wvar_mean_kernel(double* head, int m) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < m) head[2 + c] = head[2 + c] / head[0];
}

// dX (n, m) with row pitch ldx, dw (n) or NULL, ds2 (m): all device pointers.
static int wvar_dev_impl(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx, const double* dw,
                         double* ds2) {
  hipStream_t st = ctx->stream;
  int rpi = m <= 256 ? 256 / m : 1;
  if (rpi < 1) rpi = 1;
  int64_t want = (n + (int64_t)rpi * 16 - 1) / ((int64_t)rpi * 16);  // about 16 rows per thread at least
  int grid = (int)std::min<int64_t>(std::max<int64_t>(want, 1), (int64_t)ctx->cu_count * 4);
  const size_t bytes = ((size_t)grid + 1) * (size_t)(m + 2) * sizeof(double);
  ELFIHIP_CHECK_HIP(ctx, ctx->scratch.reserve(bytes));
  double* head = ctx->scratch.as<double>();
  double* partial = head + (m + 2);
  WvarArgs A;
  A.X = dX;
  A.w = dw;
  A.n = n;
  A.ldx = ldx;
  A.m = m;
  A.rpi = rpi;
  A.head = head;
  A.partial = partial;
  const int fb = (m + 2 + 63) / 64;
  hipLaunchKernelGGL(wvar_partial_kernel<1>, dim3(grid), dim3(256), 0, st, A);
  hipLaunchKernelGGL(wvar_finish_kernel<1>, dim3(fb), dim3(64), 0, st, partial, grid, m, head, ds2);
  hipLaunchKernelGGL(wvar_mean_kernel, dim3(fb), dim3(64), 0, st, head, m);
  hipLaunchKernelGGL(wvar_partial_kernel<2>, dim3(grid), dim3(256), 0, st, A);
  hipLaunchKernelGGL(wvar_finish_kernel<2>, dim3(fb), dim3(64), 0, st, partial, grid, m, head, ds2);
  return launch_status(ctx, "weighted variance kernels");
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_weighted_var_dev(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx, const double* dw,
                             double* ds2) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 1 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m, (long long)ldx);
  ELFIHIP_REQUIRE(ctx, dX && ds2, "NULL data pointer");
  DeviceGuard g(ctx->device);
  return wvar_dev_impl(ctx, dX, n, m, ldx, dw, ds2);
}

int elfihip_weighted_var(elfihip_ctx* ctx, const double* X, int64_t n, int m, int64_t ldx, const double* w, double* s2) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 1 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m, (long long)ldx);
  ELFIHIP_REQUIRE(ctx, X && s2, "NULL data pointer");
  DeviceGuard g(ctx->device);
  hipStream_t st = ctx->stream;
  const size_t xb = (size_t)n * m * sizeof(double), wb = w ? (size_t)n * sizeof(double) : 0;
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve(xb + wb));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)m * sizeof(double)));
  double* dX = ctx->in.as<double>();
  double* dw = w ? dX + (size_t)n * m : nullptr;
  if (ldx == m)
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dX, X, xb, hipMemcpyHostToDevice, st));
  else
    ELFIHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(dX, (size_t)m * sizeof(double), X, (size_t)ldx * sizeof(double),
                                            (size_t)m * sizeof(double), (size_t)n, hipMemcpyHostToDevice, st));
  if (w) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dw, w, wb, hipMemcpyHostToDevice, st));
  ELFIHIP_TRY(wvar_dev_impl(ctx, dX, n, m, m, dw, ctx->out.as<double>()));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(s2, ctx->out.as<double>(), (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  return ELFIHIP_OK;
}

}  // extern "C"
