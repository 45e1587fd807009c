// One AdaptiveDistance batch in ONE read of the rows (gfx950).
//
// What the reference does with a batch of an adaptive-distance round, in three NumPy sweeps over the (n, m) summaries
// and one argsort:
//   * AdaptiveDistance.nested_distance (elfi/model/elfi_model.py:1135-1151): K cdist calls -- the weights of the EARLIER
//     rounds, the first function unweighted -- column-stacked to (n, K);
//   * AdaptiveDistance.add_data (elfi_model.py:1104-1125), called by Rejection._merge_batch for every batch
//     (elfi/methods/inference/samplers.py:213-216): the running column mean / sum of squared deviations that the NEXT
//     update_distance (:1127-1133) turns into weights -- it never feeds the distances of the batch it is computed from;
//   * Rejection._merge_batch (samplers.py:218-237): accept a row if EVERY nested column is <= its threshold
//     (AdaptiveDistanceSMC hands a list, samplers.py:657-660), rank by the last column.
// The three are independent given the rows, so one kernel does them while a tile of rows is on the chip: 8 m bytes read
// per row, 8 K written (the distances, when the caller wants them), nothing else -- against three reads of the matrix
// by welford.hip (two passes) + distance.hip (round 3: 3.35 ms per round of 10^7 x 64 where one read takes 0.85 ms).
//
//   * rows stream exactly as in distance.hip (software-pipelined 16-byte loads -> LDS tile with an odd pitch); lane r sums
//     row r left to right for each weight vector: the distances are bit-identical to dist_multiw_pipe_kernel / cdist;
//   * column statistics: thread (g, c) owns column c of the rows g, g + G, ... of every tile its workgroup sees.  Per
//     tile it makes the two passes over its rows IN LDS (sum -> tile mean -> sum of squared deviations about it: the
//     textbook two-pass form, no cancellation) and folds (count, mean, M2) into its running triple with Chan's
//     pairwise update -- the same formula welford_merge_kernel uses across ranks.  At the end the G row groups of a
//     workgroup are merged in order, one (1 + 2m)-double partial per workgroup leaves the chip, and
//     adaptive_finish_kernel merges the partials in a fixed order: no atomics, bit-reproducible for a launch shape.
//     The result equals the reference's batched Welford update up to rounding (it is the more accurate of the two: the
//     reference's first batch sums x (x - mean) about a zero mean); tests hold it to an a-priori bound.
//   * selection: the per-column acceptance and the running k-th distance of the sampler state are applied to the
//     distances while they are in registers (RejectFilter, tile_stream.hpp); accepted rows are counted.
#include "common.hpp"
#include "tile_stream.hpp"
#include "internal.hpp"

#include <algorithm>

#pragma clang fp contract(off)

namespace elfihip {

constexpr int ADA_T = 128;      // threads per workgroup (two waves), as dist_multiw_pipe_kernel
constexpr int ADA_U = 16;       // 16-byte loads per thread and tile

struct AdaptArgs {
  RowArgs A;                         // rows, observed y, W (K, m) in aux, out (n, K) or NULL, the selection filter
  const double* acc;                 // K acceptance thresholds (NULL: every row is acceptable)
  unsigned long long* acc_count;     // rows accepted (with acc)
  double* partial;                   // (gridDim.x, 1 + 2m) workgroup statistics (NULL: none)
};

struct ColStat {
  double n, mean, M2;
};

// Chan et al.: (n, mean, M2) <- (n, mean, M2) U (nb, mb, qb); the operations and their order are those of
// welford_merge_kernel (welford.hip) and elfi_amd/sharding.py:merge_welford.
__device__ __forceinline__ void chan_merge(ColStat& a, double nb, double mb, double qb) {
  if (nb == 0.0) return;
  if (a.n == 0.0) {
    a.n = nb;
    a.mean = mb;
    a.M2 = qb;
    return;
  }
  const double tot = a.n + nb;
  const double delta = mb - a.mean;
  const double f = nb / tot;
  a.M2 = a.M2 + qb + delta * delta * (a.n * f);
  a.mean = a.mean + delta * f;
  a.n = tot;
}

template <int U>
__global__ __launch_bounds__(ADA_T) void adaptive_pass_kernel(AdaptArgs P) {
  extern __shared__ __align__(16) double lds[];
  const RowArgs& A = P.A;
  const int T = ADA_T, tid = threadIdx.x, m = A.m, K = A.K, R = A.R, mp = A.mp;
  double* tile = lds;
  const int tile_doubles = R * mp > 3 * T ? R * mp : 3 * T;   // the tile doubles as the end-of-kernel reduction space
  double* ys = tile + tile_doubles;
  double* ws = ys + m;        // (K, m)
  double* accs = ws + (size_t)K * m;   // (K)
  for (int j = tid; j < m; j += T) ys[j] = A.y[j];
  for (int j = tid; j < K * m; j += T) ws[j] = A.aux[j];
  if (P.acc)
    for (int j = tid; j < K; j += T) accs[j] = P.acc[j];
  const int64_t ntiles = (A.n + R - 1) / R;
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  const double thr = A.F.thr ? *A.F.thr : inf;
  // column statistics: thread (g, c)
  const int G = T / m;                    // row groups (the launcher guarantees m <= T)
  const int g = tid / m, c = tid - g * m;
  const bool sact = P.partial != nullptr && g < G;
  ColStat st = {0.0, 0.0, 0.0};
  unsigned long long nacc = 0;
  double2 v[U];
  int64_t t = blockIdx.x;
  if (t < ntiles) tile_fetch<U>(A, t * R, (int)((A.n - t * R) < R ? (A.n - t * R) : R), v);
  for (; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * R;
    const int rows = (int)((A.n - row0) < R ? (A.n - row0) : R);
    __syncthreads();   // tile free (every reader of the previous one is done); ys / ws / accs visible on the first trip
    tile_commit<U>(A, tile, rows, v);
    const int64_t tn = t + gridDim.x;
    if (tn < ntiles) tile_fetch<U>(A, tn * R, (int)((A.n - tn * R) < R ? (A.n - tn * R) : R), v);
    __syncthreads();
    // ---- nested distances: lane r owns row r (cdist's left-to-right order per weight vector)
    double dlast = 0.0;
    bool ok = tid < rows;
    if (tid < rows) {
      const double* row = tile + (size_t)tid * mp;
      for (int k0 = 0; k0 < K; k0 += 4) {
        const int kn = K - k0 < 4 ? K - k0 : 4;
        const double* w0 = ws + (size_t)k0 * m;
        const double* w1 = ws + (size_t)(k0 + (kn > 1 ? 1 : 0)) * m;
        const double* w2 = ws + (size_t)(k0 + (kn > 2 ? 2 : 0)) * m;
        const double* w3 = ws + (size_t)(k0 + (kn > 3 ? 3 : 0)) * m;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
        for (int j = 0; j < m; ++j) {
          const double d = row[j] - ys[j];
          const double d2 = d * d;
          s0 = s0 + w0[j] * d2;
          s1 = s1 + w1[j] * d2;
          s2 = s2 + w2[j] * d2;
          s3 = s3 + w3[j] * d2;
        }
        const double r0 = sqrt(s0), r1 = sqrt(s1), r2 = sqrt(s2), r3 = sqrt(s3);
        if (A.out) {
          double* o = A.out + (row0 + tid) * K + k0;
          o[0] = r0;
          if (kn > 1) o[1] = r1;
          if (kn > 2) o[2] = r2;
          if (kn > 3) o[3] = r3;
        }
        if (P.acc) {   // samplers.py:219-225: every nested column against its threshold (a NaN distance is not accepted)
          ok = ok && r0 <= accs[k0];
          if (kn > 1) ok = ok && r1 <= accs[k0 + 1];
          if (kn > 2) ok = ok && r2 <= accs[k0 + 2];
          if (kn > 3) ok = ok && r3 <= accs[k0 + 3];
        }
        dlast = kn > 3 ? r3 : (kn > 2 ? r2 : (kn > 1 ? r1 : r0));
      }
    }
    if (P.acc) nacc += (unsigned long long)__popcll(__ballot(ok));   // wave-uniform
    if (A.F.thr) reject_offer(A.F, ok && dlast < thr, dlast, A.F.row_base + row0 + tid);
    // ---- column statistics of the tile: two passes over this thread's rows in LDS, then Chan's update
    if (sact && g < rows) {
      const double* col = tile + c;
      double s = 0.0;
      int cnt = 0;
#pragma unroll 8
      for (int r = g; r < rows; r += G) {
        s += col[(size_t)r * mp];
        ++cnt;
      }
      const double mt = s / (double)cnt;
      double q = 0.0;
#pragma unroll 8
      for (int r = g; r < rows; r += G) {
        const double e = col[(size_t)r * mp] - mt;
        q += e * e;
      }
      chan_merge(st, (double)cnt, mt, q);
    }
  }
  if (P.partial) {
    __syncthreads();   // the tile is free: its first 3 T doubles carry the row groups' triples
    double* red = tile;
    if (sact) {
      red[3 * tid] = st.n;
      red[3 * tid + 1] = st.mean;
      red[3 * tid + 2] = st.M2;
    }
    __syncthreads();
    if (tid < m) {
      ColStat a = {0.0, 0.0, 0.0};
      for (int gg = 0; gg < G; ++gg) {
        const double* e = red + 3 * (gg * m + tid);
        chan_merge(a, e[0], e[1], e[2]);
      }
      double* o = P.partial + (size_t)blockIdx.x * (1 + 2 * m);
      if (tid == 0) o[0] = a.n;
      o[1 + tid] = a.mean;
      o[1 + m + tid] = a.M2;
    }
  }
  if (P.acc && (tid & 63) == 0 && nacc) atomicAdd(P.acc_count, nacc);
}

// One workgroup per column: the `nparts` workgroup partials of the pass(es) -> the batch's (count, mean, M2) in `bst`
// (1 + 2m).  Thread j merges partials j, j + 256, ... in that order, then the 256 triples are merged by a fixed tree.
__global__ __launch_bounds__(256) void adaptive_finish_kernel(const double* partial, int nparts, int m, double* bst) {
  __shared__ double rn[256], rm[256], rq[256];
  const int c = blockIdx.x, j = threadIdx.x, ns = 1 + 2 * m;
  ColStat a = {0.0, 0.0, 0.0};
  for (int p = j; p < nparts; p += 256) {
    const double* e = partial + (size_t)p * ns;
    chan_merge(a, e[0], e[1 + c], e[1 + m + c]);
  }
  rn[j] = a.n;
  rm[j] = a.mean;
  rq[j] = a.M2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (j < s) {
      chan_merge(a, rn[j + s], rm[j + s], rq[j + s]);
      rn[j] = a.n;
      rm[j] = a.mean;
      rq[j] = a.M2;
    }
    __syncthreads();
  }
  if (j == 0) {
    if (c == 0) bst[0] = a.n;
    bst[1 + c] = a.mean;
    bst[1 + m + c] = a.M2;
  }
}

// state (1 + 2m: count, mean, M2) <- state U batch: what AdaptiveDistance.add_data leaves after the batch
// (elfi_model.py:1116-1124), in Chan's form.  One workgroup; every thread reads the old count before it is replaced.
__global__ void adaptive_fold_kernel(double* state, const double* bst, int m) {
  const double n_old = state[0];
  __syncthreads();
  for (int c = threadIdx.x; c < m; c += blockDim.x) {
    ColStat a = {n_old, state[1 + c], state[1 + m + c]};
    if (n_old == 0.0) a.mean = 0.0, a.M2 = 0.0;
    chan_merge(a, bst[0], bst[1 + c], bst[1 + m + c]);
    state[1 + c] = a.mean;
    state[1 + m + c] = a.M2;
    if (c == 0) state[0] = a.n;
  }
}

static bool ada_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static size_t ada_lds_bytes(int m, int K, int R) {
  const int mp = m | 1;
  const size_t tile = std::max<size_t>((size_t)R * mp, 3 * (size_t)ADA_T);
  return (tile + (size_t)m + (size_t)K * m + (size_t)K) * sizeof(double);
}

static int ada_rows_per_tile(int m) {
  int R = 2 * ADA_T * ADA_U / m;
  return R > ADA_T ? ADA_T : R;
}

bool adaptive_pass_supported(const double* dX, int m, int64_t ldx, int K) {
  if (m < 1 || m > ADA_T || (m & 1) || (ldx & 1) || !ada_aligned16(dX)) return false;
  return ada_lds_bytes(m, K, ada_rows_per_tile(m)) <= 64 * 1024;
}

int adaptive_max_parts(const elfihip_ctx* ctx) { return ctx->cu_count * 8; }

// The fused pass over rows [0, n) of dX.  partial: room for adaptive_max_parts() x (1 + 2m) doubles or NULL;
// *nparts receives the number of partials written.
int adaptive_pass_impl(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx, const double* dy,
                       const double* dW, int K, double* dout, const RejectFilter* F, const double* dacc,
                       unsigned long long* dacc_count, double* partial, int* nparts) {
  if (nparts) *nparts = 0;
  if (n <= 0) return ELFIHIP_OK;
  AdaptArgs P;
  RowArgs& A = P.A;
  A.X = dX;
  A.n = n;
  A.ldx = ldx;
  A.y = dy;
  A.aux = dW;
  A.out = dout;
  A.p = 2.0;
  A.inv_p = 0.5;
  A.m = m;
  A.mp = m | 1;
  A.K = K;
  A.vec2 = 1;
  A.R = ada_rows_per_tile(m);
  A.nt = 0;
  A.div_h = make_fastdiv((uint32_t)(m / 2));
  A.F = F ? *F : RejectFilter{nullptr, nullptr, nullptr, nullptr, 0u, 0ll};
  P.acc = dacc;
  P.acc_count = dacc_count;
  P.partial = partial;
  const size_t lds = ada_lds_bytes(m, K, A.R);
  const int64_t ntiles = (n + A.R - 1) / A.R;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  int64_t g = (int64_t)ctx->cu_count * per_cu;
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  hipLaunchKernelGGL((adaptive_pass_kernel<ADA_U>), dim3((unsigned)g), dim3(ADA_T), lds, ctx->stream, P);
  if (nparts && partial) *nparts = (int)g;
  return launch_status(ctx, "adaptive_pass_kernel");
}

// partials -> batch statistics -> folded into the running store dstate (1 + 2m); bst: 1 + 2m doubles of scratch that
// keep the batch's own statistics.
int adaptive_stats_finish(elfihip_ctx* ctx, const double* partial, int nparts, int m, double* bst, double* dstate) {
  if (nparts <= 0) return ELFIHIP_OK;
  hipLaunchKernelGGL(adaptive_finish_kernel, dim3((unsigned)m), dim3(256), 0, ctx->stream, partial, nparts, m, bst);
  if (dstate) hipLaunchKernelGGL(adaptive_fold_kernel, dim3(1), dim3(128), 0, ctx->stream, dstate, bst, m);
  return launch_status(ctx, "adaptive statistics kernels");
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_adaptive_push_dev(elfihip_ctx* ctx, elfihip_reject* state, const double* dX, int64_t n, int m, int64_t ldx,
                              const double* dy, const double* dW, int K, double* dout, double* dwelford,
                              int64_t row_base) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m, (long long)ldx);
  ELFIHIP_REQUIRE(ctx, K >= 1 && K <= 64, "K=%d outside [1,64]", K);
  ELFIHIP_REQUIRE(ctx, n == 0 || (dX && dy && dW), "NULL data pointer");
  ELFIHIP_REQUIRE(ctx, !state || reject_ctx(state) == ctx, "the sampler state belongs to another context");
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  return adaptive_push_impl(ctx, state, dX, n, m, ldx, dy, dW, K, dout, dwelford, row_base);
}

int elfihip_adaptive_push(elfihip_ctx* ctx, elfihip_reject* state, const double* X, int64_t n, int m, int64_t ldx,
                          const double* y, const double* W, int K, double* out, int64_t* count, double* mean,
                          double* M2, int64_t row_base) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m, (long long)ldx);
  ELFIHIP_REQUIRE(ctx, K >= 1 && K <= 64, "K=%d outside [1,64]", K);
  ELFIHIP_REQUIRE(ctx, y && W && (n == 0 || X), "NULL data pointer");
  ELFIHIP_REQUIRE(ctx, (!count && !mean && !M2) || (count && mean && M2), "count / mean / M2 come together");
  ELFIHIP_REQUIRE(ctx, !state || reject_ctx(state) == ctx, "the sampler state belongs to another context");
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  hipStream_t st = ctx->stream;
  const size_t ns = 1 + 2 * (size_t)m;
  // parameters: y (m), W (K m), running store (1 + 2m)
  std::vector<double> hst(ns, 0.0);
  if (count) {
    hst[0] = (double)*count;
    memcpy(&hst[1], mean, (size_t)m * sizeof(double));
    memcpy(&hst[1 + m], M2, (size_t)m * sizeof(double));
  }
  ELFIHIP_CHECK_HIP(ctx, ctx->par.reserve(((size_t)m + (size_t)K * m + ns) * sizeof(double)));
  double* dy = ctx->par.as<double>();
  double* dW = dy + m;
  double* dstate = dW + (size_t)K * m;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dy, y, (size_t)m * sizeof(double), hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dW, W, (size_t)K * m * sizeof(double), hipMemcpyHostToDevice, st));
  if (count) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dstate, hst.data(), ns * sizeof(double), hipMemcpyHostToDevice, st));
  // the batch itself: packed (n, m), pitch m rounded up to even so that the rows stay 16-byte aligned
  const int64_t ldd = (m + 1) & ~1;
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((size_t)n * ldd * sizeof(double)));
  double* dX = ctx->in.as<double>();
  ELFIHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(dX, (size_t)ldd * sizeof(double), X, (size_t)ldx * sizeof(double),
                                          (size_t)m * sizeof(double), (size_t)n, hipMemcpyHostToDevice, st));
  double* dout = nullptr;
  if (out) {
    ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)n * K * sizeof(double)));
    dout = ctx->out.as<double>();
  }
  ELFIHIP_TRY(adaptive_push_impl(ctx, state, dX, n, m, ldd, dy, dW, K, dout, count ? dstate : nullptr, row_base));
  if (out) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, dout, (size_t)n * K * sizeof(double), hipMemcpyDeviceToHost, st));
  if (count) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(hst.data(), dstate, ns * sizeof(double), hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  if (count) {
    *count = (int64_t)hst[0];
    memcpy(mean, &hst[1], (size_t)m * sizeof(double));
    memcpy(M2, &hst[1 + m], (size_t)m * sizeof(double));
  }
  return ELFIHIP_OK;
}

}  // extern "C"
