// Gaussian-mixture proposal density of SMC-ABC on gfx950.
//
// Replaces GMDistribution.pdf / logpdf (elfi/methods/utils.py:142-198), which the SMC sampler
// evaluates for every new particle against the whole previous population
// (elfi/methods/inference/samplers.py:508-549): the reference loops over the N mixture components
// on the host and calls scipy.stats.multivariate_normal.pdf on all M points for each of them,
// O(N M d^2) with N Python-level iterations.  SURVEY.md section 8f, rank 2.
//
//   pdf(x) = sum_i w_i exp(-0.5 (c + |(x - m_i) U|^2)),   c = rank log(2 pi) + log pdet(cov)
//
// with U = eigenvectors * sqrt(1 / eigenvalues) of the shared covariance, exactly the
// factorisation SciPy's multivariate_normal uses ([SciPy] _PSD); the deviation x - m_i is formed
// first and then whitened, as SciPy does, and the components are accumulated in index order, so
// the only difference to the reference is the device exp() (relative 1e-13 class).
// One thread per evaluation point; the component means stream through LDS in tiles of 256.  The
// kernel is VALU/exp-bound (M N (d^2 + d) FMAs + M N exps), HBM traffic is negligible.
#include "common.hpp"
#include "philox.hpp"

namespace elfihip {

// ---- draws from the mixture (GMDistribution.rvs, elfi/methods/utils.py:199-262: the SMC proposal of a batch) ----------
// Draw i: a component by the inverse of the cumulative weights at U_i (Philox counter i of stream `stream`), plus A z_i
// with A A^T = cov and z_i the d standard normals of counters i ceil(d / 2) ... of stream `stream + 1`.  The reference
// draws component indices with RandomState.choice and the perturbations with scipy's multivariate_normal on the host
// (10^6 proposals per batch: 60 % of an SMC batch once simulator and distance run on the GPU).
struct GmRvsArgs {
  const double* means;   // (N, d)
  const double* cumw;    // (N) cumulative normalised weights, last entry 1
  const double* A;       // (d, d) row-major factor of the covariance
  double* out;           // (n, d)
  int64_t n, N;
  int d;
  uint64_t seed, stream;
};

__global__ __launch_bounds__(256) void gm_rvs_kernel(GmRvsArgs G) {
  const int d = G.d, dh = (d + 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < G.n; i += (int64_t)gridDim.x * 256) {
    uint32_t r[4];
    philox4x32_10((uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)G.stream, (uint32_t)(G.stream >> 32), (uint32_t)G.seed,
                  (uint32_t)(G.seed >> 32), r);
    const double u = (double)(((uint64_t)r[0] << 21) | (r[1] >> 11)) * 0x1.0p-53;
    int64_t lo = 0, hi = G.N - 1;   // first index with cumw > u
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (G.cumw[mid] > u)
        hi = mid;
      else
        lo = mid + 1;
    }
    double z[64];
    for (int jj = 0; jj < dh; ++jj) {
      double z0, z1;
      normal_pair(G.seed, G.stream + 1, (uint64_t)i * (uint64_t)dh + (uint64_t)jj, z0, z1);
      z[2 * jj] = z0;
      if (2 * jj + 1 < d) z[2 * jj + 1] = z1;
    }
    for (int a = 0; a < d; ++a) {
      double x = G.means[lo * d + a];
      for (int c = 0; c < d; ++c) x += G.A[a * d + c] * z[c];
      G.out[i * d + a] = x;
    }
  }
}

struct GmArgs {
  const double* x;       // (M, d)
  const double* means;   // (N, d)
  const double* w;       // (N)
  const double* U;       // (d, d) row-major: U[k][j]
  double* out;           // (M)
  int64_t M, N;
  int d;
  double c;              // rank log(2 pi) + log pdet
};

template <int DP>
__global__ __launch_bounds__(256) void gm_pdf_kernel(GmArgs G) {
  __shared__ double Us[DP * DP];
  __shared__ double ms[256 * DP];
  __shared__ double ws[256];
  const int d = G.d, tid = threadIdx.x;
  for (int e = tid; e < DP * DP; e += 256) {
    const int k = e / DP, j = e - k * DP;
    Us[e] = (k < d && j < d) ? G.U[k * d + j] : 0.0;
  }
  const int64_t row = (int64_t)blockIdx.x * 256 + tid;
  double x[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) x[k] = (row < G.M && k < d) ? G.x[row * d + k] : 0.0;
  double acc = 0.0;
  for (int64_t i0 = 0; i0 < G.N; i0 += 256) {
    const int cnt = (int)((G.N - i0) < 256 ? (G.N - i0) : 256);
    __syncthreads();
    for (int e = tid; e < cnt * DP; e += 256) {
      const int i = e / DP, k = e - i * DP;
      ms[e] = k < d ? G.means[(i0 + i) * d + k] : 0.0;
    }
    if (tid < cnt) ws[tid] = G.w[i0 + tid];
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
      double dev[DP];
#pragma unroll
      for (int k = 0; k < DP; ++k) dev[k] = x[k] - ms[i * DP + k];
      double maha = 0.0;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        double z = 0.0;
#pragma unroll
        for (int k = 0; k < DP; ++k) z += dev[k] * Us[k * DP + j];
        maha += z * z;
      }
      acc += ws[i] * exp(-0.5 * (G.c + maha));
    }
  }
  if (row < G.M) G.out[row] = acc;
}

// 16 < d <= 64: |U^T (x - m)|^2 = |U^T x - U^T m|^2, so both point sets are transformed ONCE (d^2 flops per row) and a
// pair costs d subtractions instead of d^2 products -- with d values per point in registers and 64 components per LDS
// tile.  The additions come in another order than SciPy's dot(dev, prec_U) (compared at 1e-12).
__global__ __launch_bounds__(256) void gm_transform_kernel(const double* in, const double* U, double* out, int64_t R, int d) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= R * d) return;
  const int64_t r = e / d;
  const int j = (int)(e - r * d);
  double z = 0.0;
  for (int k = 0; k < d; ++k) z += in[r * d + k] * U[k * d + j];
  out[e] = z;
}

template <int DP>
__global__ __launch_bounds__(256) void gm_pdf_wide_kernel(GmArgs G, const double* xt, const double* mt) {
  __shared__ double ms[64 * DP];
  __shared__ double ws[64];
  const int d = G.d, tid = threadIdx.x;
  const int64_t row = (int64_t)blockIdx.x * 256 + tid;
  double x[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) x[k] = (row < G.M && k < d) ? xt[row * d + k] : 0.0;
  double acc = 0.0;
  for (int64_t i0 = 0; i0 < G.N; i0 += 64) {
    const int cnt = (int)((G.N - i0) < 64 ? (G.N - i0) : 64);
    __syncthreads();
    for (int e = tid; e < cnt * DP; e += 256) {
      const int i = e / DP, k = e - i * DP;
      ms[e] = k < d ? mt[(i0 + i) * d + k] : 0.0;
    }
    if (tid < cnt) ws[tid] = G.w[i0 + tid];
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
      double maha = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double z = x[k] - ms[i * DP + k];
        maha += z * z;
      }
      acc += ws[i] * exp(-0.5 * (G.c + maha));
    }
  }
  if (row < G.M) G.out[row] = acc;
}

static int gm_pdf_dev_impl(elfihip_ctx* ctx, GmArgs G) {
  ELFIHIP_REQUIRE(ctx, G.M >= 0 && G.N >= 1 && G.d >= 1 && G.d <= 64, "bad shape M=%lld N=%lld d=%d (d <= 64)",
                  (long long)G.M, (long long)G.N, G.d);
  if (G.M == 0) return ELFIHIP_OK;
  const unsigned grid = (unsigned)((G.M + 255) / 256);
  if (G.d > 16) {
    const size_t nx = (size_t)G.M * G.d, nm = (size_t)G.N * G.d;
    ELFIHIP_CHECK_HIP(ctx, ctx->scratch.reserve((nx + nm) * sizeof(double)));
    double* xt = ctx->scratch.as<double>();
    double* mt = xt + nx;
    hipLaunchKernelGGL(gm_transform_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, ctx->stream, G.x, G.U, xt, G.M, G.d);
    hipLaunchKernelGGL(gm_transform_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, ctx->stream, G.means, G.U, mt, G.N,
                       G.d);
    if (G.d <= 32)
      hipLaunchKernelGGL((gm_pdf_wide_kernel<32>), dim3(grid), dim3(256), 0, ctx->stream, G, xt, mt);
    else
      hipLaunchKernelGGL((gm_pdf_wide_kernel<64>), dim3(grid), dim3(256), 0, ctx->stream, G, xt, mt);
    return launch_status(ctx, "gm_pdf_wide_kernel");
  }
  if (G.d <= 2)
    hipLaunchKernelGGL((gm_pdf_kernel<2>), dim3(grid), dim3(256), 0, ctx->stream, G);
  else if (G.d <= 4)
    hipLaunchKernelGGL((gm_pdf_kernel<4>), dim3(grid), dim3(256), 0, ctx->stream, G);
  else if (G.d <= 8)
    hipLaunchKernelGGL((gm_pdf_kernel<8>), dim3(grid), dim3(256), 0, ctx->stream, G);
  else
    hipLaunchKernelGGL((gm_pdf_kernel<16>), dim3(grid), dim3(256), 0, ctx->stream, G);
  return launch_status(ctx, "gm_pdf_kernel");
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_gm_pdf(elfihip_ctx* ctx, const double* x, int64_t M, int d, const double* means, int64_t N,
                   const double* weights, const double* U, double log_norm, double* out) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, M >= 0 && N >= 1 && d >= 1 && d <= 64, "bad shape M=%lld N=%lld d=%d (d <= 64)", (long long)M,
                  (long long)N, d);
  ELFIHIP_REQUIRE(ctx, means && weights && U && (M == 0 || (x && out)), "NULL data pointer");
  if (M == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  const size_t nx = (size_t)M * d, nm = (size_t)N * d, nu = (size_t)d * d;
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((nx + nm + (size_t)N + nu) * sizeof(double)));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)M * sizeof(double)));
  double* dx = ctx->in.as<double>();
  double* dm = dx + nx;
  double* dw = dm + nm;
  double* dU = dw + N;
  hipStream_t st = ctx->stream;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dx, x, nx * sizeof(double), hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dm, means, nm * sizeof(double), hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dw, weights, (size_t)N * sizeof(double), hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dU, U, nu * sizeof(double), hipMemcpyHostToDevice, st));
  GmArgs G;
  G.x = dx;
  G.means = dm;
  G.w = dw;
  G.U = dU;
  G.out = ctx->out.as<double>();
  G.M = M;
  G.N = N;
  G.d = d;
  G.c = log_norm;
  ELFIHIP_TRY(gm_pdf_dev_impl(ctx, G));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->out.p, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  return ELFIHIP_OK;
}

int elfihip_gm_rvs(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, int d, const double* means, int64_t N,
                   const double* cumw, const double* A, double* out) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && N >= 1 && d >= 1 && d <= 64, "bad shape n=%lld N=%lld d=%d (d <= 64)", (long long)n,
                  (long long)N, d);
  ELFIHIP_REQUIRE(ctx, means && cumw && A && (n == 0 || out), "NULL data pointer");
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  const size_t nm = (size_t)N * d, na = (size_t)d * d, nx = (size_t)n * d;
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((nm + (size_t)N + na) * sizeof(double)));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve(nx * sizeof(double)));
  double* dm = ctx->in.as<double>();
  double* dc = dm + nm;
  double* dA = dc + N;
  hipStream_t st = ctx->stream;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dm, means, nm * sizeof(double), hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dc, cumw, (size_t)N * sizeof(double), hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dA, A, na * sizeof(double), hipMemcpyHostToDevice, st));
  GmRvsArgs G;
  G.means = dm;
  G.cumw = dc;
  G.A = dA;
  G.out = ctx->out.as<double>();
  G.n = n;
  G.N = N;
  G.d = d;
  G.seed = seed;
  G.stream = stream;
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 16);
  hipLaunchKernelGGL(gm_rvs_kernel, dim3(grid), dim3(256), 0, st, G);
  ELFIHIP_TRY(launch_status(ctx, "gm_rvs_kernel"));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->out.p, nx * sizeof(double), hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  return ELFIHIP_OK;
}

}  // extern "C"
