// The k smallest of n distances, on the GPU, so that only k rows ever leave it.
//
// Replaces the selection inside Rejection._merge_batch
// (elfi/methods/inference/samplers.py:209-237): the reference appends every batch (batch_size
// rows) to its n_samples best so far and runs a full np.argsort over n_samples + batch_size
// distances on the host, i.e. it needs ALL distances of the batch on the host.  Keeping the k
// smallest of the batch is equivalent (the final sample is the n_samples smallest overall) and
// needs k values + k row indices.  SURVEY.md section 8f, rank 1.
//
// Order: ascending by (distance, row index); NaN sorts last (as in np.argsort).  Ties at the cut
// are resolved towards the lower row index -- np.argsort's default quicksort leaves that order
// unspecified.
//
// Method: MSD radix select on the order-preserving 64-bit image of the doubles, 8 bits per pass:
// per pass one histogram kernel over the keys that still match the prefix (LDS bins, one global
// atomic per bin per workgroup -- integer counts, so the result is deterministic) and one
// single-workgroup kernel that picks the digit; then a stable two-pass compaction (per-block
// counts, scan, write) of the keys below the k-th key plus as many equal ones as needed.
// HBM-bound and tiny next to the distance kernel: 8 bytes per distance per pass.
#include <algorithm>
#include <numeric>

#include "common.hpp"

namespace elfihip {

struct SelState {
  unsigned long long prefix;   // bits decided so far (high bits), rest zero
  unsigned long long k_rem;    // rank still to find inside the prefix class (1-based)
  unsigned long long n_lt;     // keys strictly below the prefix class
  unsigned int hist[256];
};

__device__ __forceinline__ unsigned long long key_of(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  if (v != v) return ~0ull;                                   // NaN last
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

__global__ __launch_bounds__(256) void sel_hist_kernel(const double* d, int64_t n, int64_t stride, int pass,
                                                       SelState* st) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int shift = 56 - 8 * pass;
  const unsigned long long prefix = st->prefix;
  const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned long long k = key_of(d[i * stride]);
    if ((k & himask) == prefix) atomicAdd(&h[(k >> shift) & 255], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void sel_pick_kernel(int pass, SelState* st) {
  if (threadIdx.x != 0) return;
  const int shift = 56 - 8 * pass;
  unsigned long long rem = st->k_rem, below = 0;
  int digit = 255;
  for (int b = 0; b < 256; ++b) {
    const unsigned long long c = st->hist[b];
    if (rem <= c) {
      digit = b;
      break;
    }
    rem -= c;
    below += c;
  }
  st->prefix |= (unsigned long long)digit << shift;
  st->k_rem = rem;
  st->n_lt += below;
  for (int b = 0; b < 256; ++b) st->hist[b] = 0;
}

// counts[b] = {#keys < kth, #keys == kth} of block b's contiguous slice
__global__ __launch_bounds__(256) void sel_count_kernel(const double* d, int64_t n, int64_t stride, int64_t per_block,
                                                        const SelState* st, unsigned int* counts) {
  __shared__ unsigned int s[2];
  if (threadIdx.x < 2) s[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long kth = st->prefix;
  const int64_t lo = (int64_t)blockIdx.x * per_block;
  const int64_t hi = lo + per_block < n ? lo + per_block : n;
  unsigned int lt = 0, eq = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const unsigned long long k = key_of(d[i * stride]);
    lt += k < kth;
    eq += k == kth;
  }
  if (lt) atomicAdd(&s[0], lt);
  if (eq) atomicAdd(&s[1], eq);
  __syncthreads();
  if (threadIdx.x < 2) counts[2 * blockIdx.x + threadIdx.x] = s[threadIdx.x];
}

// exclusive scan of the per-block counts (single workgroup, sequential over blocks per class)
__global__ void sel_scan_kernel(unsigned int* counts, int nblocks) {
  if (threadIdx.x < 2) {
    unsigned int run = 0;
    for (int b = 0; b < nblocks; ++b) {
      const unsigned int c = counts[2 * b + threadIdx.x];
      counts[2 * b + threadIdx.x] = run;
      run += c;
    }
  }
}

// stable write: row order inside a block slice is kept (chunks of 256 rows, ballot ranks)
__global__ __launch_bounds__(256) void sel_write_kernel(const double* d, int64_t n, int64_t stride, int64_t per_block,
                                                        const SelState* st, const unsigned int* offs, int64_t k,
                                                        double* vals, int64_t* idx) {
  __shared__ unsigned int wl[4], we[4], base[2];
  const unsigned long long kth = st->prefix;
  const int64_t n_lt = (int64_t)st->n_lt;  // == total #keys < kth
  const int64_t lo = (int64_t)blockIdx.x * per_block;
  const int64_t hi = lo + per_block < n ? lo + per_block : n;
  if (threadIdx.x < 2) base[threadIdx.x] = offs[2 * blockIdx.x + threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t c0 = lo; c0 < hi; c0 += 256) {
    const int64_t i = c0 + threadIdx.x;
    double v = 0.0;
    bool is_lt = false, is_eq = false;
    if (i < hi) {
      v = d[i * stride];
      const unsigned long long kk = key_of(v);
      is_lt = kk < kth;
      is_eq = kk == kth;
    }
    const unsigned long long bl = __ballot(is_lt), be = __ballot(is_eq);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) {
      wl[w] = (unsigned int)__popcll(bl);
      we[w] = (unsigned int)__popcll(be);
    }
    __syncthreads();
    unsigned int pl = 0, pe = 0;
    for (int q = 0; q < w; ++q) {
      pl += wl[q];
      pe += we[q];
    }
    if (is_lt) {
      const int64_t pos = (int64_t)base[0] + pl + __popcll(bl & below);
      vals[pos] = v;
      idx[pos] = i;
    } else if (is_eq) {
      const int64_t pos = n_lt + (int64_t)base[1] + pe + __popcll(be & below);
      if (pos < k) {
        vals[pos] = v;
        idx[pos] = i;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      base[0] += wl[0] + wl[1] + wl[2] + wl[3];
      base[1] += we[0] + we[1] + we[2] + we[3];
    }
    __syncthreads();
  }
}

static int topk_dev_impl(elfihip_ctx* ctx, const double* dD, int64_t n, int64_t stride, int64_t k, double* dvals,
                         int64_t* didx) {
  ELFIHIP_REQUIRE(ctx, n >= 0 && k >= 0 && stride >= 1, "bad arguments n=%lld k=%lld stride=%lld", (long long)n,
                  (long long)k, (long long)stride);
  if (k > n) k = n;
  if (k == 0) return ELFIHIP_OK;
  ELFIHIP_REQUIRE(ctx, dD && dvals && didx, "NULL data pointer");
  hipStream_t st = ctx->stream;
  int nblocks = (int)std::min<int64_t>((n + 4095) / 4096, (int64_t)ctx->cu_count * 8);
  if (nblocks < 1) nblocks = 1;
  int64_t per_block = (n + nblocks - 1) / nblocks;
  per_block = (per_block + 255) / 256 * 256;
  nblocks = (int)((n + per_block - 1) / per_block);
  const size_t bytes = sizeof(SelState) + 2 * (size_t)nblocks * sizeof(unsigned int);
  ELFIHIP_CHECK_HIP(ctx, ctx->scratch.reserve(bytes));
  SelState* ds = ctx->scratch.as<SelState>();
  unsigned int* counts = reinterpret_cast<unsigned int*>(ds + 1);
  SelState init;
  memset(&init, 0, sizeof init);
  init.k_rem = (unsigned long long)k;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(ds, &init, sizeof init, hipMemcpyHostToDevice, st));
  const int hist_blocks = (int)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->cu_count * 8);
  for (int pass = 0; pass < 8; ++pass) {
    hipLaunchKernelGGL(sel_hist_kernel, dim3(hist_blocks), dim3(256), 0, st, dD, n, stride, pass, ds);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(64), 0, st, pass, ds);
  }
  hipLaunchKernelGGL(sel_count_kernel, dim3(nblocks), dim3(256), 0, st, dD, n, stride, per_block, ds, counts);
  hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(64), 0, st, counts, nblocks);
  hipLaunchKernelGGL(sel_write_kernel, dim3(nblocks), dim3(256), 0, st, dD, n, stride, per_block, ds, counts, k, dvals,
                     didx);
  return launch_status(ctx, "top-k selection kernels");
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_topk_smallest_dev(elfihip_ctx* ctx, const double* dD, int64_t n, int64_t stride, int64_t k, double* dvals,
                              int64_t* didx) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return topk_dev_impl(ctx, dD, n, stride, k, dvals, didx);
}

int elfihip_topk_smallest(elfihip_ctx* ctx, const double* D, int64_t n, int64_t stride, int64_t k, double* vals,
                          int64_t* idx) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && k >= 0 && stride >= 1, "bad arguments n=%lld k=%lld stride=%lld", (long long)n,
                  (long long)k, (long long)stride);
  if (k > n) k = n;
  if (k == 0) return ELFIHIP_OK;
  ELFIHIP_REQUIRE(ctx, D && vals && idx, "NULL data pointer");
  DeviceGuard g(ctx->device);
  const size_t in_bytes = (size_t)n * stride * sizeof(double);
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve(in_bytes));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)k * (sizeof(double) + sizeof(int64_t))));
  double* dD = ctx->in.as<double>();
  double* dv = ctx->out.as<double>();
  int64_t* di = reinterpret_cast<int64_t*>(dv + k);
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dD, D, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_TRY(topk_dev_impl(ctx, dD, n, stride, k, dv, di));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(vals, dv, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(idx, di, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // the k survivors are in row order inside the two classes; final order by (distance, row)
  std::vector<int64_t> perm((size_t)k);
  std::iota(perm.begin(), perm.end(), 0);
  auto is_nan = [](double v) { return v != v; };
  std::sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) {
    const double va = vals[a], vb = vals[b];
    if (is_nan(va) != is_nan(vb)) return is_nan(vb);
    if (!is_nan(va) && va != vb) return va < vb;
    return idx[a] < idx[b];
  });
  std::vector<double> tv((size_t)k);
  std::vector<int64_t> ti((size_t)k);
  for (int64_t q = 0; q < k; ++q) {
    tv[(size_t)q] = vals[perm[(size_t)q]];
    ti[(size_t)q] = idx[perm[(size_t)q]];
  }
  memcpy(vals, tv.data(), (size_t)k * sizeof(double));
  memcpy(idx, ti.data(), (size_t)k * sizeof(int64_t));
  return ELFIHIP_OK;
}

}  // extern "C"
