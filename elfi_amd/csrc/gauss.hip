// The Gaussian example model on the GPU: draws, simulator, summaries and distance in one pass.
//
// Replaces, for elfi/examples/gauss.py:
//     gauss(mu, sigma, n_obs)  = ss.norm.rvs(loc=mu, scale=sigma, size=(batch, n_obs))     gauss.py:11-35
//                              = random_state.standard_normal(size) * sigma + mu           [SciPy rv_continuous.rvs]
//     ss_mean(y) = np.mean(y, axis=1),  ss_var(y) = np.var(y, axis=1)                       gauss.py:142-173
//     elfi.Distance('euclidean', ss_mean, ss_var)                                           gauss.py:133
// One kernel takes standard normals z (n, n_obs) -- handed in by the caller (drawn from the reference's MT19937
// stream: that is what bit-parity with the reference needs) or drawn here from a counter-based generator -- forms
// y = z * sigma + mu in LDS, sums every row in NumPy's pairwise order (summaries.hip's scheme: eight lanes per row,
// FMA contraction off), and writes the two summaries and the distance to the observed ones: 16 bytes of parameters
// in, 24 bytes out per simulation; y never exists in memory unless asked for.
//
// Draws: Philox4x32-10 (Salmon et al., SC'11; the generator behind cuRAND / rocRAND / NumPy's Philox), counter =
// (pair index, stream), key = seed; a pair of 53-bit uniforms -> Box-Muller -> elements 2p and 2p+1 of the (n, n_obs)
// matrix in row-major order.  The stream is a pure function of (seed, stream, element index): the same numbers for
// every launch geometry, any element reproducible on its own.  (Not the reference's numbers -- NumPy's MT19937 cannot
// be reproduced in parallel; runs that must match the reference draw on the host and pass z in.)
#include "common.hpp"

#include <cmath>

#include "philox.hpp"

#pragma clang fp contract(off)

namespace elfihip {

__global__ __launch_bounds__(256) void random_bits_kernel(uint64_t seed, uint64_t stream, int64_t nblocks, uint32_t* out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= nblocks) return;
  uint32_t r[4];
  philox4x32_10((uint32_t)p, (uint32_t)((uint64_t)p >> 32), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), r);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[4 * p + j] = r[j];
}

// out[e] = z_e * scale + loc for e in [0, n), z from pair e / 2 (a grid-stride loop over pairs)
__global__ __launch_bounds__(256) void randn_kernel(uint64_t seed, uint64_t stream, int64_t n, double loc, double scale,
                                                    double* out) {
  const int64_t npair = (n + 1) / 2;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npair; p += (int64_t)gridDim.x * 256) {
    double z0, z1;
    normal_pair(seed, stream, (uint64_t)p, z0, z1);
    if (2 * p + 1 < n && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
      *reinterpret_cast<double2*>(out + 2 * p) = make_double2(z0 * scale + loc, z1 * scale + loc);
    } else {
      out[2 * p] = z0 * scale + loc;
      if (2 * p + 1 < n) out[2 * p + 1] = z1 * scale + loc;
    }
  }
}

// out[i][j] = loc[i] + scale[j] z_e, e = i m + j (m even: a pair of normals stays inside a row) -- the draws of randn_kernel
// with the same seed / stream, shaped into the rows of a synthetic Gaussian simulator
__global__ __launch_bounds__(256) void randn_rows_kernel(uint64_t seed, uint64_t stream, int64_t n, int m, const double* loc,
                                                         const double* scale, double* out) {
  const int64_t npair = n * (int64_t)m / 2;
  const int h = m / 2;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npair; p += (int64_t)gridDim.x * 256) {
    double z0, z1;
    normal_pair(seed, stream, (uint64_t)p, z0, z1);
    const int64_t i = p / h;
    const int j = 2 * (int)(p - i * h);
    const double l = loc[i];
    *reinterpret_cast<double2*>(out + 2 * p) = make_double2(z0 * scale[j] + l, z1 * scale[j + 1] + l);
  }
}

// ---- prior draws (include/elfihip.h: elfihip_prior_draw) --------------------------------------------------------
// U_e = 53-bit uniform in [0, 1): element e of stream (seed, stream) is the first (e even) or second (e odd) 53-bit word of
// Philox counter e / 2 -- the words normal_pair() turns into a Box-Muller pair.  Every transform keeps the reference's
// operation order (multiply and add round separately, as NumPy's do).
__device__ __forceinline__ void uniform_pair(uint64_t seed, uint64_t stream, uint64_t p, double& u0, double& u1) {
  uint32_t r[4];
  philox4x32_10((uint32_t)p, (uint32_t)(p >> 32), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), r);
  const uint64_t a = ((uint64_t)r[0] << 21) | (r[1] >> 11), b = ((uint64_t)r[2] << 21) | (r[3] >> 11);
  u0 = (double)a * 0x1.0p-53;
  u1 = (double)b * 0x1.0p-53;
}

// (plain operators: this file is compiled with fp contract off, the __dmul_rn / __dadd_rn header wrappers are not -- their
// products and sums were fused into FMAs after inlining)
template <int KIND>
__device__ __forceinline__ double prior_transform(double u, double a0, double a1, double c) {
  if (KIND == 0) {   // ss.uniform.rvs(loc, scale): U * scale + loc
    const double t = u * a1;
    return t + a0;
  }
  if (KIND == 1) {   // ma2.py:116-117: np.where(u < 0.5, np.sqrt(2. * u) * b - b, -np.sqrt(2. * (1. - u)) * b + b)
    const double b = a0;
    if (u < 0.5) {
      const double r = sqrt(2.0 * u), t = r * b;
      return t - b;
    }
    const double w = 1.0 - u, r = -sqrt(2.0 * w), t = r * b;
    return t + b;
  }
  // ma2.py:163-165: locs = np.maximum(-a - t1, -a + t1); scales = a - locs; ss.uniform.rvs(loc=locs, scale=scales)
  const double a = a0, l0 = -a - c, l1 = -a + c;
  const double loc = l0 > l1 ? l0 : l1;
  const double sc = a - loc, t = u * sc;
  return t + loc;
}

template <int KIND>
__global__ __launch_bounds__(256) void prior_draw_kernel(uint64_t seed, uint64_t stream, int64_t n, double a0, double a1,
                                                         const double* cond, double* out) {
  const int64_t npair = (n + 1) / 2;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npair; p += (int64_t)gridDim.x * 256) {
    double u0, u1;
    uniform_pair(seed, stream, (uint64_t)p, u0, u1);
    const bool two = 2 * p + 1 < n;
    const double c0 = KIND == 2 ? cond[2 * p] : 0.0, c1 = (KIND == 2 && two) ? cond[2 * p + 1] : 0.0;
    out[2 * p] = prior_transform<KIND>(u0, a0, a1, c0);
    if (two) out[2 * p + 1] = prior_transform<KIND>(u1, a0, a1, c1);
  }
}

// NumPy's pairwise sum of n <= 128 terms by eight lanes (summaries.hip: np_pairwise8); lane j of an aligned group of 8
template <class F>
__device__ __forceinline__ double pairwise8(F f, int n, int j) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += f(i);
    return r;
  }
  const int nfull = n - (n % 8);
  double r = f(j);
  for (int i = 8 + j; i < nfull; i += 8) r += f(i);
  r = lanes8_sum(r);   // (DPP, bit-identical to the xor butterfly: common.hpp)
  for (int i = nfull; i < n; ++i) r += f(i);
  return r;
}

// the general length: one lane per row, NumPy's recursion (blocks of at most 128, halves rounded to multiples of 8)
template <class F>
__device__ double pairwise1(F f, int lo, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += f(lo + i);
    return r;
  }
  if (n <= 128) {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f(lo + j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += f(lo + i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += f(lo + i);
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise1(f, lo, n2) + pairwise1(f, lo + n2, n - n2);
}

struct GaussArgs {
  const double* Z;     // (n, ldz) standard normals, or NULL: drawn here
  int64_t ldz;
  uint64_t seed, stream;
  const double* mu;    // (n)
  const double* sigma; // (n)
  int64_t n;
  int L, Lp;           // observations per simulation, LDS row pitch
  int rows;            // rows per tile
  double obs_mean, obs_var;
  double* Y;           // optional (n, L): the simulator output
  double* S1;          // (n) ss_mean
  double* S2;          // (n) ss_var
  double* D;           // (n) distance
};

template <bool WIDE>
__global__ __launch_bounds__(128) void gauss_kernel(GaussArgs G) {
  extern __shared__ __align__(16) double tile[];
  const int tid = threadIdx.x, L = G.L, Rt = G.rows;
  const int64_t ntiles = (G.n + Rt - 1) / Rt;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * Rt;
    const int rows = (int)((G.n - row0) < Rt ? (G.n - row0) : Rt);
    __syncthreads();
    if (G.Z) {
      for (int e = tid; e < rows * L; e += 128) {
        const int r = e / L, i = e - r * L;
        const double y = G.Z[(row0 + r) * G.ldz + i] * G.sigma[row0 + r] + G.mu[row0 + r];
        tile[r * G.Lp + i] = y;
        if (G.Y) G.Y[(row0 + r) * (int64_t)L + i] = y;
      }
    } else {
      // elements [e0, e1) of the row-major (n, L) matrix; pairs that straddle the tile's ends are drawn by both tiles
      const int64_t e0 = row0 * L, e1 = e0 + (int64_t)rows * L;
      for (int64_t p = e0 / 2 + tid; 2 * p < e1; p += 128) {
        double z[2];
        normal_pair(G.seed, G.stream, (uint64_t)p, z[0], z[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int64_t e = 2 * p + h;
          if (e >= e0 && e < e1) {
            const int r = (int)((e - e0) / L), i = (int)((e - e0) - (int64_t)r * L);
            const double y = z[h] * G.sigma[row0 + r] + G.mu[row0 + r];
            tile[r * G.Lp + i] = y;
            if (G.Y) G.Y[e] = y;
          }
        }
      }
    }
    __syncthreads();
    if (!WIDE) {
      const int j = tid & 7;
      for (int r0 = 0; r0 < rows; r0 += 16) {
        const int r = r0 + (tid >> 3);
        const bool live = r < rows;
        const double* row = tile + (size_t)(live ? r : 0) * G.Lp;
        const double mean = pairwise8([&](int i) { return row[i]; }, L, j) / (double)L;
        const double var = pairwise8([&](int i) { const double d = row[i] - mean; return d * d; }, L, j) / (double)L;
        if (live && j == 0) {
          const int64_t gi = row0 + r;
          G.S1[gi] = mean;
          G.S2[gi] = var;
          const double d1 = mean - G.obs_mean, d2 = var - G.obs_var;   // cdist euclidean over the two summaries
          G.D[gi] = sqrt(d1 * d1 + d2 * d2);
        }
      }
    } else if (tid < rows) {
      const double* row = tile + (size_t)tid * G.Lp;
      const double mean = pairwise1([&](int i) { return row[i]; }, 0, L) / (double)L;
      const double var = pairwise1([&](int i) { const double d = row[i] - mean; return d * d; }, 0, L) / (double)L;
      const int64_t gi = row0 + tid;
      G.S1[gi] = mean;
      G.S2[gi] = var;
      const double d1 = mean - G.obs_mean, d2 = var - G.obs_var;
      G.D[gi] = sqrt(d1 * d1 + d2 * d2);
    }
  }
}

static int gauss_dev_impl(elfihip_ctx* ctx, const double* dZ, int64_t ldz, uint64_t seed, uint64_t stream, int64_t n,
                          int n_obs, const double* dmu, const double* dsigma, double obs_mean, double obs_var, double* dY,
                          double* dS1, double* dS2, double* dD) {
  ELFIHIP_REQUIRE(ctx, n >= 0 && n_obs >= 1 && (!dZ || ldz >= n_obs), "bad shape n=%lld n_obs=%d ldz=%lld", (long long)n,
                  n_obs, (long long)ldz);
  ELFIHIP_REQUIRE(ctx, n == 0 || (dmu && dsigma && dS1 && dS2 && dD), "NULL data pointer");
  if (n == 0) return ELFIHIP_OK;
  GaussArgs G;
  G.Z = dZ;
  G.ldz = ldz;
  G.seed = seed;
  G.stream = stream;
  G.mu = dmu;
  G.sigma = dsigma;
  G.n = n;
  G.L = n_obs;
  G.Lp = n_obs | 1;
  G.obs_mean = obs_mean;
  G.obs_var = obs_var;
  G.Y = dY;
  G.S1 = dS1;
  G.S2 = dS2;
  G.D = dD;
  const bool wide = n_obs > 128;
  // rows per tile: 16 KiB of observations (eight lanes per row: a multiple of 16 rows), at most 128 rows for the
  // lane-per-row form
  int rows = (int)((16 * 1024) / ((size_t)G.Lp * sizeof(double)));
  rows = wide ? std::max(1, std::min(rows, 128)) : std::max(16, rows / 16 * 16);
  G.rows = rows;
  const size_t lds = (size_t)rows * G.Lp * sizeof(double);
  ELFIHIP_REQUIRE(ctx, lds <= 160 * 1024, "simulations of %d observations do not fit the LDS tile", n_obs);
  const int64_t ntiles = (n + rows - 1) / rows;
  const int64_t g = std::min<int64_t>(ntiles, (int64_t)ctx->cu_count * 8);
  if (wide) {
    if (lds > 64 * 1024)
      ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_kernel<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(gauss_kernel<true>, dim3((unsigned)g), dim3(128), lds, ctx->stream, G);
  } else {
    hipLaunchKernelGGL(gauss_kernel<false>, dim3((unsigned)g), dim3(128), lds, ctx->stream, G);
  }
  return launch_status(ctx, "gauss_kernel");
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_random_bits_dev(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t nblocks, uint32_t* dout) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, nblocks >= 0 && (nblocks == 0 || dout), "bad arguments");
  if (nblocks == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  hipLaunchKernelGGL(random_bits_kernel, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, ctx->stream, seed, stream,
                     nblocks, dout);
  return launch_status(ctx, "random_bits_kernel");
}

int elfihip_randn_dev(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, double loc, double scale, double* dout) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && (n == 0 || dout), "bad arguments");
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  const int64_t npair = (n + 1) / 2;
  const int64_t grid = std::min<int64_t>((npair + 255) / 256, (int64_t)ctx->cu_count * 16);
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, seed, stream, n, loc, scale, dout);
  return launch_status(ctx, "randn_kernel");
}

int elfihip_prior_draw_dev(elfihip_ctx* ctx, int kind, uint64_t seed, uint64_t stream, int64_t n, const double* a,
                           const double* dcond, double* dout) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, kind >= 0 && kind <= 2, "unknown prior kind %d", kind);
  ELFIHIP_REQUIRE(ctx, n >= 0 && a && (n == 0 || dout), "bad arguments");
  ELFIHIP_REQUIRE(ctx, kind != ELFIHIP_PRIOR_MA2_T2 || n == 0 || dcond, "the conditional prior needs t1");
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  const int64_t npair = (n + 1) / 2;
  const unsigned grid = (unsigned)std::min<int64_t>((npair + 255) / 256, (int64_t)ctx->cu_count * 16);
  const double a0 = a[0], a1 = kind == ELFIHIP_PRIOR_UNIFORM ? a[1] : 0.0;
  if (kind == ELFIHIP_PRIOR_UNIFORM)
    hipLaunchKernelGGL((prior_draw_kernel<0>), dim3(grid), dim3(256), 0, ctx->stream, seed, stream, n, a0, a1, dcond, dout);
  else if (kind == ELFIHIP_PRIOR_MA2_T1)
    hipLaunchKernelGGL((prior_draw_kernel<1>), dim3(grid), dim3(256), 0, ctx->stream, seed, stream, n, a0, a1, dcond, dout);
  else
    hipLaunchKernelGGL((prior_draw_kernel<2>), dim3(grid), dim3(256), 0, ctx->stream, seed, stream, n, a0, a1, dcond, dout);
  return launch_status(ctx, "prior_draw_kernel");
}

int elfihip_prior_draw(elfihip_ctx* ctx, int kind, uint64_t seed, uint64_t stream, int64_t n, const double* a,
                       const double* cond, double* out) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && a && (n == 0 || out), "bad arguments");
  ELFIHIP_REQUIRE(ctx, kind != ELFIHIP_PRIOR_MA2_T2 || n == 0 || cond, "the conditional prior needs t1");
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  const size_t bytes = (size_t)n * sizeof(double);
  ELFIHIP_CHECK_HIP(ctx, ctx->par.reserve(2 * bytes));
  double* dcond = ctx->par.as<double>();
  double* dout = dcond + n;
  if (kind == ELFIHIP_PRIOR_MA2_T2)
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dcond, cond, bytes, hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_TRY(elfihip_prior_draw_dev(ctx, kind, seed, stream, n, a, kind == ELFIHIP_PRIOR_MA2_T2 ? dcond : nullptr, dout));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_randn_rows(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, int m, const double* loc,
                       const double* scale, double* out) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 2 && (m & 1) == 0, "bad shape n=%lld m=%d (m even)", (long long)n, m);
  ELFIHIP_REQUIRE(ctx, n == 0 || (loc && scale && out), "NULL data pointer");
  ++ctx->rows_epoch;
  ctx->rows_n = 0;
  ctx->rows_m = m;
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  ELFIHIP_CHECK_HIP(ctx, ctx->rows.reserve((size_t)n * m * sizeof(double)));
  ELFIHIP_CHECK_HIP(ctx, ctx->par.reserve(((size_t)n + m) * sizeof(double)));
  double* dloc = ctx->par.as<double>();
  double* dsc = dloc + n;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dloc, loc, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dsc, scale, (size_t)m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const int64_t npair = n * (int64_t)m / 2;
  const int64_t grid = std::min<int64_t>((npair + 255) / 256, (int64_t)ctx->cu_count * 16);
  hipLaunchKernelGGL(randn_rows_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, seed, stream, n, m, dloc, dsc,
                     ctx->rows.as<double>());
  ELFIHIP_TRY(launch_status(ctx, "randn_rows_kernel"));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->rows.p, (size_t)n * m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->rows_n = n;   // the device copy stays for the distance call that follows (elfihip_adaptive_push_kept)
  return ELFIHIP_OK;
}

int elfihip_gauss_distance_dev(elfihip_ctx* ctx, const double* dZ, int64_t ldz, uint64_t seed, uint64_t stream, int64_t n,
                               int n_obs, const double* dmu, const double* dsigma, double obs_mean, double obs_var,
                               double* dY, double* dS1, double* dS2, double* dD) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return gauss_dev_impl(ctx, dZ, ldz, seed, stream, n, n_obs, dmu, dsigma, obs_mean, obs_var, dY, dS1, dS2, dD);
}

int elfihip_gauss_distance(elfihip_ctx* ctx, const double* Z, uint64_t seed, uint64_t stream, int64_t n, int n_obs,
                           const double* mu, const double* sigma, double obs_mean, double obs_var, double* Y, double* S1,
                           double* S2, double* D) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && n_obs >= 1, "bad shape n=%lld n_obs=%d", (long long)n, n_obs);
  ELFIHIP_REQUIRE(ctx, n == 0 || (mu && sigma && S1 && S2 && D), "NULL data pointer");
  if (n == 0) return keep_distances(ctx, nullptr, 0, 1);   // an empty batch is still a call: the kept copy's name moves on
  DeviceGuard g(ctx->device);
  const size_t nz = Z ? (size_t)n * n_obs : 0, ny = Y ? (size_t)n * n_obs : 0;
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((nz + 2 * (size_t)n) * sizeof(double)));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((ny + 3 * (size_t)n) * sizeof(double)));
  double* dZ = ctx->in.as<double>();
  double* dmu = dZ + nz;
  double* dsg = dmu + n;
  double* dS = ctx->out.as<double>();
  double* dY = dS + 3 * (size_t)n;
  if (Z) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dZ, Z, nz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dmu, mu, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dsg, sigma, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_TRY(gauss_dev_impl(ctx, Z ? dZ : nullptr, n_obs, seed, stream, n, n_obs, dmu, dsg, obs_mean, obs_var,
                             Y ? dY : nullptr, dS, dS + n, dS + 2 * n));
  ELFIHIP_TRY(keep_distances(ctx, dS + 2 * n, n, 1));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(S1, dS, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(S2, dS + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(D, dS + 2 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (Y) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(Y, dY, ny * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

}  // extern "C"
