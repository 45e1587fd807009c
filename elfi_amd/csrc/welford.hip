// Running column mean / sum of squared deviations for AdaptiveDistance on gfx950.
//
// Replaces AdaptiveDistance.add_data (elfi/model/elfi_model.py:1104-1125):
//     N += b; d1 = x - mean; mean += sum(d1, axis=0)/N; d2 = x - mean; M2 += sum(d1*d2, axis=0)
// The update needs the NEW mean before the second sum, so it is two streaming passes
// over the batch (the second one is served from L2 / Infinity Cache for ordinary batch
// sizes).  Both passes are HBM-bound column reductions: 8*m bytes per row.
//
// Determinism: lane t always owns column t % m, rows are assigned to (block, row-group)
// in a fixed pattern, and partial sums are combined in a fixed order (row-groups inside
// a workgroup, then workgroups by index) -- no atomics, so repeated runs and different
// grid schedules of the same launch shape give identical bits.
#include "common.hpp"
#include "internal.hpp"

#pragma clang fp contract(off)

namespace elfihip {

struct WelfordArgs {
  const double* X;
  int64_t n, ldx;
  int m;
  int rpi;           // row-groups per workgroup iteration (blockDim / m, at least 1)
  const double* mean_old;  // state + 1
  const double* mean_new;  // scratch (pass 2)
  double* partial;   // (gridDim, m)
};

// PASS 1: sum_rows (x - mean_old).  PASS 2: sum_rows (x - mean_old) * (x - mean_new).
template <int PASS>
__global__ void welford_partial_kernel(WelfordArgs A) {
  extern __shared__ __align__(16) double red[];  // blockDim doubles
  const int T = blockDim.x, tid = threadIdx.x, m = A.m;
  const int cstride = m <= T ? m : T;
  const int rg = m <= T ? tid / m : 0;
  const int c0 = m <= T ? tid - rg * m : tid;
  const bool active = rg < A.rpi;
  const int64_t rstride = (int64_t)gridDim.x * A.rpi;
  for (int c = c0; c < m; c += cstride) {
    double acc = 0.0;
    if (active) {
      const double mo = A.mean_old[c];
      const double mn = PASS == 2 ? A.mean_new[c] : 0.0;
      const double* __restrict__ col = A.X + c;
      int64_t r = (int64_t)blockIdx.x * A.rpi + rg;
      for (; r + 7 * rstride < A.n; r += 8 * rstride) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = col[(r + u * rstride) * A.ldx];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          double d1 = v[u] - mo;
          if constexpr (PASS == 1)
            acc += d1;
          else
            acc += d1 * (v[u] - mn);
        }
      }
      for (; r < A.n; r += rstride) {
        double x = col[r * A.ldx];
        double d1 = x - mo;
        if constexpr (PASS == 1)
          acc += d1;
        else
          acc += d1 * (x - mn);
      }
    }
    if (m <= T) {  // every lane makes exactly one trip of the column loop here
      red[tid] = acc;
      __syncthreads();
      if (tid < m) {
        double s = red[tid];
        for (int g = 1; g < A.rpi; ++g) s += red[g * m + tid];
        A.partial[(size_t)blockIdx.x * m + tid] = s;
      }
    } else {
      A.partial[(size_t)blockIdx.x * m + c] = acc;
    }
  }
}

// state = [N, mean (m), M2 (m)].  PASS 1 writes mean_new to scratch; PASS 2 commits.
// One workgroup per 32 columns: thread (c, j) sums the block partials b = j, j+8, ... of column c with
// four independent accumulators (the loads pipeline), the eight j-sums are combined in fixed order.
template <int PASS>
__global__ __launch_bounds__(256) void welford_finish_kernel(const double* partial, int nblocks, int m, double nrows,
                                                             double* state, double* mean_new) {
  __shared__ double red[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), j = threadIdx.x >> 5;
  double a4[4] = {0, 0, 0, 0};
  if (c < m) {
    int b = j;
    for (; b + 24 < nblocks; b += 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a4[u] += partial[(size_t)(b + 8 * u) * m + c];
    }
    for (; b < nblocks; b += 8) a4[0] += partial[(size_t)b * m + c];
  }
  red[j][threadIdx.x & 31] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  __syncthreads();
  if (j == 0 && c < m) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][threadIdx.x];
    if constexpr (PASS == 1) {
      const double N = state[0] + nrows;
      mean_new[c] = state[1 + c] + s / N;
    } else {
      state[1 + m + c] += s;
      state[1 + c] = mean_new[c];
    }
  }
  if constexpr (PASS == 2) {
    // one writer for the count, after every reader of state[0] in pass 1 has long finished
    if (blockIdx.x == 0 && threadIdx.x == 0) state[0] += nrows;
  }
}

int welford_dev_impl(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx, double* dstate) {
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m,
                  (long long)ldx);
  ELFIHIP_REQUIRE(ctx, dstate && (n == 0 || dX), "NULL data pointer");
  if (n == 0) return ELFIHIP_OK;
  const int T = 256;
  WelfordArgs A;
  A.X = dX;
  A.n = n;
  A.ldx = ldx;
  A.m = m;
  A.rpi = m <= T ? T / m : 1;
  int64_t groups = (n + A.rpi - 1) / A.rpi;
  int64_t g = (int64_t)ctx->cu_count * 4;  // enough loads in flight; fewer partials for the finish kernels
  if (g > groups) g = groups;
  if (g < 1) g = 1;
  ELFIHIP_CHECK_HIP(ctx, ctx->scratch.reserve(((size_t)g * m + m) * sizeof(double)));
  A.partial = ctx->scratch.as<double>();
  double* mean_new = A.partial + (size_t)g * m;
  A.mean_old = dstate + 1;
  A.mean_new = mean_new;
  const int fb = (m + 31) / 32;
  hipLaunchKernelGGL((welford_partial_kernel<1>), dim3((unsigned)g), dim3(T), T * sizeof(double), ctx->stream, A);
  hipLaunchKernelGGL((welford_finish_kernel<1>), dim3(fb), dim3(256), 0, ctx->stream, A.partial, (int)g, m,
                     (double)n, dstate, mean_new);
  hipLaunchKernelGGL((welford_partial_kernel<2>), dim3((unsigned)g), dim3(T), T * sizeof(double), ctx->stream, A);
  hipLaunchKernelGGL((welford_finish_kernel<2>), dim3(fb), dim3(256), 0, ctx->stream, A.partial, (int)g, m,
                     (double)n, dstate, mean_new);
  return launch_status(ctx, "welford kernels");
}

}  // namespace elfihip

using namespace elfihip;

namespace elfihip {
// Chan merge of `world` (count, mean[m], M2[m]) states in rank order, one thread per column: the device twin of the
// host merge in elfi_amd/sharding.py (same operations in the same order, so both give the same bits), for the multi-GPU
// adaptive distance: all-gather of the rank states -> this kernel -> the merged state and, optionally, the cdist
// weights 1 / scale^2 = N / M2 (elfi/model/elfi_model.py:1124,1129-1132) without a host round trip.
__global__ void welford_merge_kernel(const double* states, int world, int m, double* merged, double* w2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const int ns = 1 + 2 * m;
  double N = 0.0, mean = 0.0, M2 = 0.0;
  for (int r = 0; r < world; ++r) {
    const double* st = states + (size_t)r * ns;
    const double nb = st[0];
    if (nb == 0.0) continue;
    if (N == 0.0) {
      N = nb;
      mean = st[1 + j];
      M2 = st[1 + m + j];
      continue;
    }
    const double tot = N + nb;
    const double delta = st[1 + j] - mean;
    M2 = M2 + st[1 + m + j] + delta * delta * (N * (nb / tot));
    mean = mean + delta * (nb / tot);
    N = tot;
  }
  if (j == 0) merged[0] = N;
  merged[1 + j] = mean;
  merged[1 + m + j] = M2;
  if (w2) w2[j] = 1.0 / (M2 / N);
}
}  // namespace elfihip

extern "C" {

int elfihip_welford_merge_dev(elfihip_ctx* ctx, const double* dstates, int world, int m, double* dmerged, double* dw2) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, world >= 1 && m >= 1 && dstates && dmerged, "bad arguments (world=%d m=%d)", world, m);
  DeviceGuard g(ctx->device);
  hipLaunchKernelGGL(welford_merge_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream, dstates, world, m,
                     dmerged, dw2);
  return launch_status(ctx, "welford_merge_kernel");
}

int elfihip_welford_update_dev(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx,
                               double* dstate) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return welford_dev_impl(ctx, dX, n, m, ldx, dstate);
}

int elfihip_welford_update(elfihip_ctx* ctx, const double* X, int64_t n, int m, int64_t ldx, int64_t* count,
                           double* mean, double* M2) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m,
                  (long long)ldx);
  ELFIHIP_REQUIRE(ctx, count && mean && M2 && (n == 0 || X), "NULL data pointer");
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  const size_t ns = 1 + 2 * (size_t)m;
  std::vector<double> st(ns);
  st[0] = (double)*count;
  memcpy(&st[1], mean, (size_t)m * sizeof(double));
  memcpy(&st[1 + m], M2, (size_t)m * sizeof(double));
  ELFIHIP_CHECK_HIP(ctx, ctx->par.reserve(ns * sizeof(double)));
  double* dstate = ctx->par.as<double>();
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dstate, st.data(), ns * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((size_t)n * m * sizeof(double)));
  double* dX = ctx->in.as<double>();
  if (ldx == m)
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dX, X, (size_t)n * m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  else
    ELFIHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(dX, (size_t)m * sizeof(double), X, (size_t)ldx * sizeof(double),
                                            (size_t)m * sizeof(double), (size_t)n, hipMemcpyHostToDevice,
                                            ctx->stream));
  ELFIHIP_TRY(welford_dev_impl(ctx, dX, n, m, m, dstate));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(st.data(), dstate, ns * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *count += n;
  memcpy(mean, &st[1], (size_t)m * sizeof(double));
  memcpy(M2, &st[1 + m], (size_t)m * sizeof(double));
  return ELFIHIP_OK;
}

}  // extern "C"
