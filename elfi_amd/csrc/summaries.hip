// Row-wise summary statistics of simulator output on gfx950, and the fused MA2 path.
//
// Replaces the NumPy row reductions ELFI's example models use as Summary operations
// (SURVEY.md section 8 row a6/a7):
//     autocov(x, lag) = np.mean(x[:, lag:] * x[:, :-lag], axis=1)      elfi/examples/ma2.py:40-59
//     ss_mean(y)      = np.mean(y, axis=1)                               elfi/examples/gauss.py:142-156
//     ss_var(y)       = np.var(y, axis=1)                                elfi/examples/gauss.py:159-173
//     MA2(t1, t2)     : x = w[:, 2:] + t1 * w[:, 1:-1] + t2 * w[:, :-2]  elfi/examples/ma2.py:11-37
// HBM-bound streaming reductions (8 L bytes in, 8 bytes out per row).  A workgroup streams a tile
// of rows with coalesced 16-byte loads into LDS (tile_stream.hpp); each lane then owns one row and
// sums it in exactly NumPy's order -- np.add.reduce along a contiguous axis is a pairwise sum:
// eight interleaved accumulators up to 128 elements, recursive halving (rounded to a multiple of 8)
// above -- with FMA contraction off, so results are BIT-IDENTICAL to NumPy.
//
// The fused MA2 kernel takes the white-noise matrix w (drawn on the host from the reference's
// MT19937 stream, which is what bit-parity requires) and produces both autocovariance summaries
// and the euclidean distance to the observed summaries in one pass over w: 8 (L+2) bytes in, 24
// bytes out per simulation, instead of the reference's seven full-size temporaries.
#include "common.hpp"
#include <cstdlib>

#include "philox.hpp"
#include "tile_stream.hpp"

#pragma clang fp contract(off)

namespace elfihip {

enum { SUM_MEAN = 0, SUM_VAR = 1, SUM_AUTOCOV = 2, SUM_MA2 = 3 };

// NumPy's pairwise_sum over a[0..n): f(i) yields element i.
template <class F>
__device__ double np_pairwise(F f, int lo, int n) {
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
  return np_pairwise(f, lo, n2) + np_pairwise(f, lo + n2, n - n2);
}

// The same sum for n <= 128 computed by EIGHT lanes: NumPy's eight interleaved accumulators are
// independent chains (r[j] takes elements j, j+8, ...), so lane j of an aligned 8-lane group owns
// r[j]; the fixed combine tree ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) is three xor-shuffles (fp addition
// is commutative, so both partners of a pair hold the same bits) and the tail is added in order by
// every lane.  Bit-identical to np_pairwise, eight times the lanes per row.
template <class F>
__device__ __forceinline__ double np_pairwise8(F f, int n, int j) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += f(i);
    return r;
  }
  const int nfull = n - (n % 8);
  // all (at most 16) elements of this lane's chain are fetched first, back to back, then added in
  // order: the loads pipeline instead of sitting one LDS latency apart on the dependent adds
  double e[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = 8 * k + j;
    e[k] = f(i < nfull ? i : j);  // clamped index keeps the load unconditional; the value is unused beyond nfull
  }
  double r = e[0];
#pragma unroll
  for (int k = 1; k < 16; ++k)
    if (8 * k < nfull) r += e[k];  // uniform condition
  r = lanes8_sum(r);   // (DPP, bit-identical to the xor butterfly: common.hpp)
  for (int i = nfull; i < n; ++i) r += f(i);
  return r;
}

struct SumArgs {
  RowArgs R;        // X, n, ldx, m (= row length L), mp, vec2, div_h, R (rows per tile)
  int lag;          // AUTOCOV
  const double* t1; // MA2: per-row parameters (n)
  const double* t2;
  double obs1, obs2;
  double* out1;     // MEAN/VAR/AUTOCOV: result; MA2: S1
  double* out2;     // MA2: S2
  double* out3;     // MA2: distance
  // MA2 with the white noise DRAWN HERE (R.X == NULL): element e of the row-major (n, L) noise matrix is normal number e
  // of the stream (seed, stream) of philox.hpp -- what elfihip_randn_dev would have written to memory
  uint64_t seed, stream;
};

template <int KIND, int U, bool PIPE>
__global__ __launch_bounds__(256) void row_summary_kernel(SumArgs S) {
  extern __shared__ __align__(16) double lds[];
  const RowArgs& A = S.R;
  const int tid = threadIdx.x, L = A.m, Rt = A.R;
  double* tile = lds;
  const int64_t ntiles = (A.n + Rt - 1) / Rt;
  double2 v[U];
  int64_t t = blockIdx.x;
  const bool draw = KIND == SUM_MA2 && A.X == nullptr;
  if (PIPE && !draw && t < ntiles) tile_fetch<U>(A, t * Rt, (int)((A.n - t * Rt) < Rt ? (A.n - t * Rt) : Rt), v);
  for (; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * Rt;
    const int rows = (int)((A.n - row0) < Rt ? (A.n - row0) : Rt);
    __syncthreads();
    if (draw) {
      // elements [e0, e1) of the noise matrix; pairs that straddle the tile's ends are drawn by both tiles
      const int64_t e0 = row0 * L, e1 = e0 + (int64_t)rows * L;
      for (int64_t p = e0 / 2 + tid; 2 * p < e1; p += blockDim.x) {
        double z[2];
        normal_pair(S.seed, S.stream, (uint64_t)p, z[0], z[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int64_t e = 2 * p + h;
          if (e >= e0 && e < e1) {
            const int r = (int)((e - e0) / L), i = (int)((e - e0) - (int64_t)r * L);
            tile[r * A.mp + i] = z[h];
          }
        }
      }
    } else if (PIPE) {
      tile_commit<U>(A, tile, rows, v);
      const int64_t tn = t + gridDim.x;
      if (tn < ntiles) tile_fetch<U>(A, tn * Rt, (int)((A.n - tn * Rt) < Rt ? (A.n - tn * Rt) : Rt), v);
    } else {
      load_tile<8>(A, tile, row0, rows);
    }
    __syncthreads();
    const int red_len = KIND == SUM_AUTOCOV ? L - S.lag : (KIND == SUM_MA2 ? L - 3 : L);  // longest reduction
    if (PIPE || red_len <= 128) {  // the pipelined kernel is only launched for L <= 128
      // eight lanes per row (np_pairwise8): 16 rows at a time with 128 threads
      const int j = tid & 7, gpr = blockDim.x >> 3;
      const int rounds = (rows + gpr - 1) / gpr;
      for (int q = 0; q < rounds; ++q) {
        const int r = q * gpr + (tid >> 3);
        const bool live = r < rows;
        const double* row = tile + (size_t)(live ? r : 0) * A.mp;
        const int64_t gi = row0 + r;
        if constexpr (KIND == SUM_MEAN) {
          const double v = np_pairwise8([&](int i) { return row[i]; }, L, j) / (double)L;
          if (live && j == 0) S.out1[gi] = v;
        } else if constexpr (KIND == SUM_VAR) {
          const double mean = np_pairwise8([&](int i) { return row[i]; }, L, j) / (double)L;
          const double v = np_pairwise8([&](int i) { const double d = row[i] - mean; return d * d; }, L, j) / (double)L;
          if (live && j == 0) S.out1[gi] = v;
        } else if constexpr (KIND == SUM_AUTOCOV) {
          const int lag = S.lag, cnt = L - lag;
          const double v = np_pairwise8([&](int i) { return row[i + lag] * row[i]; }, cnt, j) / (double)cnt;
          if (live && j == 0) S.out1[gi] = v;
        } else {  // SUM_MA2: row holds w (L = n_obs + 2); x_i = (w[i+2] + t1 w[i+1]) + t2 w[i]
          const double a = live ? S.t1[gi] : 0.0, b = live ? S.t2[gi] : 0.0;
          const int nobs = L - 2;
          // x replaces w in place: every lane first forms ITS x values (i = j, j+8, ...) in registers
          // from w, then stores them.  The eight lanes of a row sit in one wavefront, whose LDS
          // operations execute in program order, so all reads of w precede the first store.
          double* xr = tile + (size_t)(live ? r : 0) * A.mp;
          // (two halves of 64 elements: the stores of the first half only touch w[0..63], which the
          // second half no longer reads -- half the registers)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            double xv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int i = 64 * h + 8 * k + j, ic = i < nobs ? i : 0;
              xv[k] = (xr[ic + 2] + a * xr[ic + 1]) + b * xr[ic];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int i = 64 * h + 8 * k + j;
              if (live && i < nobs) xr[i] = xv[k];
            }
          }
          const double s1 = np_pairwise8([&](int i) { return xr[i + 1] * xr[i]; }, nobs - 1, j) / (double)(nobs - 1);
          const double s2 = np_pairwise8([&](int i) { return xr[i + 2] * xr[i]; }, nobs - 2, j) / (double)(nobs - 2);
          if (live && j == 0) {
            S.out1[gi] = s1;
            S.out2[gi] = s2;
            const double d1 = s1 - S.obs1, d2 = s2 - S.obs2;   // cdist euclidean over the two summaries
            S.out3[gi] = sqrt(d1 * d1 + d2 * d2);
          }
        }
      }
    } else if constexpr (!PIPE) {
     if (tid < rows) {
      const double* row = tile + (size_t)tid * A.mp;
      const int64_t gi = row0 + tid;
      if constexpr (KIND == SUM_MEAN) {
        S.out1[gi] = np_pairwise([&](int i) { return row[i]; }, 0, L) / (double)L;
      } else if constexpr (KIND == SUM_VAR) {
        const double mean = np_pairwise([&](int i) { return row[i]; }, 0, L) / (double)L;
        S.out1[gi] = np_pairwise([&](int i) { const double d = row[i] - mean; return d * d; }, 0, L) / (double)L;
      } else if constexpr (KIND == SUM_AUTOCOV) {
        const int lag = S.lag, cnt = L - lag;
        S.out1[gi] = np_pairwise([&](int i) { return row[i + lag] * row[i]; }, 0, cnt) / (double)cnt;
      } else {  // SUM_MA2: row holds w (L = n_obs + 2); x_i = (w[i+2] + t1 w[i+1]) + t2 w[i], in place
        const double a = S.t1[gi], b = S.t2[gi];
        const int nobs = L - 2;
        double* x = tile + (size_t)tid * A.mp;
        for (int i = 0; i < nobs; ++i) x[i] = (x[i + 2] + a * x[i + 1]) + b * x[i];
        const double s1 = np_pairwise([&](int i) { return x[i + 1] * x[i]; }, 0, nobs - 1) / (double)(nobs - 1);
        const double s2 = np_pairwise([&](int i) { return x[i + 2] * x[i]; }, 0, nobs - 2) / (double)(nobs - 2);
        S.out1[gi] = s1;
        S.out2[gi] = s2;
        const double d1 = s1 - S.obs1, d2 = s2 - S.obs2;   // cdist euclidean over the two summaries
        S.out3[gi] = sqrt(d1 * d1 + d2 * d2);
      }
     }
    }
  }
}

template <int KIND>
static int launch_summary(elfihip_ctx* ctx, SumArgs S) {
  RowArgs& A = S.R;
  const int L = A.m;
  const bool pipe = (A.vec2 || (KIND == SUM_MA2 && A.X == nullptr)) && L <= 128;   // (noise drawn in the kernel: no loads to vectorise)
  // 8 KiB in flight per workgroup (128 threads x 4 x 16 bytes), eight workgroups per CU: measured best for these
  // streaming kernels (L = 100: mean 5.6 TB/s against 4.9 with 32 KiB tiles and 4.5 with 4 KiB)
  int T;
  // the fused MA2 path does several LDS passes per row: whole rounds of 16 rows (eight lanes per row, all 128 lanes busy)
  // matter more to it than a small tile, so it streams 16 KiB per workgroup (2 10^6 x 102: 0.92 -> 0.79 ms; 32 KiB: 1.00)
  constexpr int U = KIND == SUM_MA2 ? 8 : 4;
  if (pipe) {
    T = 128;
    int R = 2 * T * U / L;
    if (R > T) R = T;
    if (KIND == SUM_MA2 && R >= 16) R = R / 16 * 16;
    A.R = R;
  } else {
    // generic path: as many rows per tile as fit a 48 KiB tile, at most 64
    int R = (int)((48 * 1024) / ((size_t)A.mp * sizeof(double)));
    if (R > 64) R = 64;
    if (R < 1) R = 1;
    A.R = R;
    T = 64;
  }
  const size_t lds = (size_t)A.R * A.mp * sizeof(double);
  if (lds > 160 * 1024) return fail(ctx, ELFIHIP_ERR_ARG, "rows of %d values do not fit the LDS tile", L);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  int64_t g = (int64_t)ctx->cu_count * per_cu;
  const int64_t ntiles = (A.n + A.R - 1) / A.R;
  if (g > ntiles) g = ntiles;
  if (pipe) {
    auto k = row_summary_kernel<KIND, U, true>;
    if (lds > 64 * 1024)
      ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)g), dim3(T), lds, ctx->stream, S);
  } else {
    auto k = row_summary_kernel<KIND, 1, false>;
    if (lds > 64 * 1024)
      ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)g), dim3(T), lds, ctx->stream, S);
  }
  return launch_status(ctx, "row_summary_kernel");
}

static RowArgs summary_row_args(const double* dX, int64_t n, int L, int64_t ldx) {
  RowArgs A;
  A.X = dX;
  A.n = n;
  A.ldx = ldx;
  A.y = nullptr;
  A.aux = nullptr;
  A.out = nullptr;
  A.p = 2.0;
  A.inv_p = 0.5;
  A.m = L;
  A.mp = L | 1;
  A.K = 0;
  A.R = 0;
  A.nt = 0;
  A.vec2 = (L % 2 == 0) && (ldx % 2 == 0) && tile_aligned16(dX);
  A.div_h = make_fastdiv((uint32_t)(A.vec2 ? L / 2 : L));
  return A;
}

static int summary_dev_impl(elfihip_ctx* ctx, int kind, const double* dX, int64_t n, int L, int64_t ldx, int lag,
                            double* dout) {
  ELFIHIP_REQUIRE(ctx, n >= 0 && L >= 1 && ldx >= L, "bad shape n=%lld L=%d ldx=%lld", (long long)n, L, (long long)ldx);
  ELFIHIP_REQUIRE(ctx, kind != SUM_AUTOCOV || (lag >= 1 && lag < L), "lag %d outside [1, %d)", lag, L);
  ELFIHIP_REQUIRE(ctx, n == 0 || (dX && dout), "NULL data pointer");
  if (n == 0) return ELFIHIP_OK;
  SumArgs S;
  S.R = summary_row_args(dX, n, L, ldx);
  S.R.nt = ctx->dist_form != 1;
  S.lag = lag;
  S.t1 = S.t2 = nullptr;
  S.obs1 = S.obs2 = 0.0;
  S.out1 = dout;
  S.out2 = S.out3 = nullptr;
  switch (kind) {
    case SUM_MEAN:
      return launch_summary<SUM_MEAN>(ctx, S);
    case SUM_VAR:
      return launch_summary<SUM_VAR>(ctx, S);
    case SUM_AUTOCOV:
      return launch_summary<SUM_AUTOCOV>(ctx, S);
  }
  return fail(ctx, ELFIHIP_ERR_ARG, "unknown summary kind %d", kind);
}

static int ma2_dev_impl(elfihip_ctx* ctx, const double* dW, int64_t n, int n_obs, int64_t ldw, const double* dt1,
                        const double* dt2, double obs1, double obs2, double* dS1, double* dS2, double* dD,
                        bool draw = false, uint64_t seed = 0, uint64_t stream = 0) {
  ELFIHIP_REQUIRE(ctx, n >= 0 && n_obs >= 3 && ldw >= n_obs + 2, "bad shape n=%lld n_obs=%d ldw=%lld", (long long)n,
                  n_obs, (long long)ldw);
  ELFIHIP_REQUIRE(ctx, n == 0 || ((dW || draw) && dt1 && dt2 && dS1 && dS2 && dD), "NULL data pointer");
  ELFIHIP_REQUIRE(ctx, !draw || n_obs + 2 <= 128, "the drawing form handles up to 126 observations per simulation");
  if (n == 0) return ELFIHIP_OK;
  SumArgs S;
  S.R = summary_row_args(draw ? nullptr : dW, n, n_obs + 2, ldw);
  S.R.nt = ctx->dist_form != 1;
  S.seed = seed;
  S.stream = stream;
  S.lag = 0;
  S.t1 = dt1;
  S.t2 = dt2;
  S.obs1 = obs1;
  S.obs2 = obs2;
  S.out1 = dS1;
  S.out2 = dS2;
  S.out3 = dD;
  return launch_summary<SUM_MA2>(ctx, S);
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_row_summary_dev(elfihip_ctx* ctx, int kind, const double* dX, int64_t n, int L, int64_t ldx, int lag,
                            double* dout) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return summary_dev_impl(ctx, kind, dX, n, L, ldx, lag, dout);
}

int elfihip_row_summary(elfihip_ctx* ctx, int kind, const double* X, int64_t n, int L, int64_t ldx, int lag,
                        double* out) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && L >= 1 && ldx >= L, "bad shape n=%lld L=%d ldx=%lld", (long long)n, L, (long long)ldx);
  ELFIHIP_REQUIRE(ctx, n == 0 || (X && out), "NULL data pointer");
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((size_t)n * L * sizeof(double)));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)n * sizeof(double)));
  double* dX = ctx->in.as<double>();
  ELFIHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(dX, (size_t)L * sizeof(double), X, (size_t)ldx * sizeof(double),
                                          (size_t)L * sizeof(double), (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_TRY(summary_dev_impl(ctx, kind, dX, n, L, L, lag, ctx->out.as<double>()));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_ma2_distance_dev(elfihip_ctx* ctx, const double* dW, int64_t n, int n_obs, int64_t ldw, const double* dt1,
                             const double* dt2, double obs1, double obs2, double* dS1, double* dS2, double* dD) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return ma2_dev_impl(ctx, dW, n, n_obs, ldw, dt1, dt2, obs1, obs2, dS1, dS2, dD);
}

int elfihip_ma2_draw_distance_dev(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, int n_obs, const double* dt1,
                                  const double* dt2, double obs1, double obs2, double* dS1, double* dS2, double* dD) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return ma2_dev_impl(ctx, nullptr, n, n_obs, n_obs + 2, dt1, dt2, obs1, obs2, dS1, dS2, dD, true, seed, stream);
}

// host forms: W != NULL reads the caller's white noise, W == NULL draws it in the kernel (seed, stream)
static int ma2_host(elfihip_ctx* ctx, const double* W, uint64_t seed, uint64_t stream, int64_t n, int n_obs, const double* t1,
                    const double* t2, double obs1, double obs2, double* S1, double* S2, double* D) {
  ELFIHIP_REQUIRE(ctx, n >= 0 && n_obs >= 3, "bad shape n=%lld n_obs=%d", (long long)n, n_obs);
  ELFIHIP_REQUIRE(ctx, n == 0 || (t1 && t2 && S1 && S2 && D), "NULL data pointer");
  if (n == 0) return keep_distances(ctx, nullptr, 0, 1);   // an empty batch is still a call: the kept copy's name moves on
  DeviceGuard g(ctx->device);
  const int L = n_obs + 2;
  const size_t nw = W ? (size_t)n * L : 0;
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((nw + 2 * (size_t)n) * sizeof(double)));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve(3 * (size_t)n * sizeof(double)));
  double* dW = ctx->in.as<double>();
  double* dt1 = dW + nw;
  double* dt2 = dt1 + n;
  double* dS = ctx->out.as<double>();
  if (W) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dW, W, nw * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dt1, t1, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dt2, t2, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (W)
    ELFIHIP_TRY(ma2_dev_impl(ctx, dW, n, n_obs, L, dt1, dt2, obs1, obs2, dS, dS + n, dS + 2 * n));
  else
    ELFIHIP_TRY(ma2_dev_impl(ctx, nullptr, n, n_obs, L, dt1, dt2, obs1, obs2, dS, dS + n, dS + 2 * n, true, seed, stream));
  ELFIHIP_TRY(keep_distances(ctx, dS + 2 * n, n, 1));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(S1, dS, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(S2, dS + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(D, dS + 2 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_ma2_distance(elfihip_ctx* ctx, const double* W, int64_t n, int n_obs, const double* t1, const double* t2,
                         double obs1, double obs2, double* S1, double* S2, double* D) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n == 0 || W, "NULL data pointer");
  return ma2_host(ctx, W, 0, 0, n, n_obs, t1, t2, obs1, obs2, S1, S2, D);
}

int elfihip_ma2_draw_distance(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, int n_obs, const double* t1,
                              const double* t2, double obs1, double obs2, double* S1, double* S2, double* D) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  return ma2_host(ctx, nullptr, seed, stream, n, n_obs, t1, t2, obs1, obs2, S1, S2, D);
}

}  // extern "C"
