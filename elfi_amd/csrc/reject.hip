// The rejection sampler's running best-k on the GPU, fused with the distance pass.
//
// Replaces Rejection._merge_batch (elfi/methods/inference/samplers.py:209-237): the reference copies every batch
// behind its n_samples best rows and argsorts n_samples + batch_size distances on the host, batch after batch.  What
// that merge keeps is the n_samples smallest distances seen so far.  Here that state lives in device memory -- k
// (distance, row) pairs, ascending, ties to the earlier row -- and a batch is folded in by ONE distance pass:
//   * once the state holds k finite distances its k-th is a threshold, and the distance kernel itself appends the rows
//     below it to a candidate list (RejectFilter, distance.hip: one wave-aggregated atomic per wave that has a hit --
//     after j batches of the same distribution about k / j rows per batch qualify);
//   * a single-workgroup kernel merges state and candidates by ranks and keeps the first k -- not after every batch
//     but, once the threshold has settled, after every REJ_MERGE_EVERY-th (the list simply keeps growing in between, against a threshold that is then a
//     few batches old: still an upper bound of the current k-th distance, so nothing is missed), because a 15 us
//     one-workgroup launch between two 48 us distance passes is a third of their time, and running it on a second
//     stream costs a 10 us event hand-over per batch on this stack (both measured); result / reset / state_dev /
//     flush merge what is pending first;
//   * only while the state is still filling up (the first batch) the batch goes through the radix selection of topk.hip.
// Only the k best rows ever leave the GPU; row numbers are global (row_base + row in the batch), so the host fetches the
// parameters / summaries of the accepted rows from its own batch store (as ELFI's OutputPool keeps them).
//
// The state never fails for valid input (round 2's fixed 65 536-entry list could overflow -- a small first batch, a
// round change without reset -- and reported it only at result(), when the batches were gone):
//   * the list holds 8 x (largest batch pushed) entries and merges happen at least every 8th push, so it cannot overflow;
//   * a push expected to offer many candidates (n k / rows seen > 8192: the threshold is still weak) takes the radix
//     selection of its k best instead of the list -- exact either way, this only keeps one-workgroup merges short.
// k > 2048 ("host-merge" states, up to 2^20): the candidates of every push -- a few hundred rows once the threshold has
// settled -- are merged into a sorted host copy of the state, the k-th distance goes back as the device threshold.
// An acceptance threshold (the objective of Rejection.sample(threshold=...), samplers.py:219-225: EVERY nested column of
// a row must be <= threshold) is applied by the candidate pass; accepted rows are counted on the device.
#include "internal.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

struct elfihip_reject {
  elfihip_ctx* ctx = nullptr;
  int64_t k = 0;
  int64_t filled = 0;        // rows offered so far, capped at k: the filter is armed once this reaches k
  elfihip::DevBuf mem;       // everything below lives here
  double* best_val = nullptr;      // (k) ascending
  long long* best_row = nullptr;   // (k)
  double* thr = nullptr;           // device scalar: best_val[k-1]
  double* cand_val = nullptr;      // (cap) candidates offered since the last merge
  long long* cand_row = nullptr;   // (cap)
  unsigned int* count = nullptr;   // how many
  unsigned int* status = nullptr;  // bit 0: more candidates were offered than the list holds
  unsigned int cap = 0;
  int unmerged = 0;                // pushes since the last merge
  int64_t armed_pushes = 0;        // pushes since the state became full (sets the merge interval)
  void* export_dst = nullptr;      // optional: every merge also leaves the packed state (k values, k rows) here
  elfihip::DevBuf cand_mem;        // the candidate list (cap pairs), grown to 8 x the largest batch
  int64_t rows_seen = 0;           // rows offered so far
  int64_t pending_rows = 0;        // rows of the filtered pushes since the last merge: bounds the list's length
  // acceptance threshold (samplers.py:219-225)
  bool has_accept = false;
  double accept = 0.0;
  int acc_ncols = 0;                         // > 0: one threshold per nested column (AdaptiveDistanceSMC, samplers.py:657-660)
  elfihip::DevBuf acc_mem;                   // REJ_ACC_COLS thresholds on the device (a scalar threshold fills them all)
  double* acc_dev = nullptr;
  unsigned long long* acc_count = nullptr;   // device: rows accepted so far
  unsigned long long acc_seen = 0;           // ... as of the previous meta() read
  elfihip::DevBuf mask_mem;                  // accept-and-select pushes: this batch's count + the masked ranking column
  // k > REJ_MAX_K: sorted host copy of the state, merged on the host after every push
  bool host_mode = false, host_dirty = false;
  std::vector<double> hval;
  std::vector<long long> hrow;
};

namespace elfihip {

constexpr int64_t REJ_MAX_K = 2048;       // state entries (LDS-resident during a merge)
constexpr int REJ_CHUNK = 1024;           // candidates merged per round
constexpr int REJ_MERGE_EVERY = 8;        // pushes per merge
constexpr unsigned int REJ_CAP = 1u << 16;   // smallest candidate list
constexpr int64_t REJ_MAX_K_HOST = 1 << 20;  // host-merge states
constexpr double REJ_HEAVY = 8192.0;         // expected candidates from which a push takes the radix selection
constexpr int REJ_ACC_COLS = 64;             // nested columns an acceptance condition can cover (= kMaxK of distance.hip)
constexpr int64_t REJ_ACC_SELECT_MIN = 1 << 15;   // batch rows from which an acceptance push selects instead of listing
// provisional threshold of a large first batch (adaptive_push_impl)
constexpr int64_t REJ_PROV_MIN_ROWS = 1 << 20;
constexpr unsigned int REJ_PROV_MAX_CAND = 1u << 16;

__device__ __forceinline__ bool rej_less(double av, long long ar, double bv, long long br) {
  return av < bv || (av == bv && ar < br);
}

// state <- the k smallest of state U candidates.  ncand < 0: read the count from S.count (and clear it); row_offset is
// added to the candidates' row numbers (the radix selection of the first batch reports batch-local rows).
struct RejArgs {
  double* best_val;
  long long* best_row;
  double* thr;
  const double* cand_val;
  const long long* cand_row;
  unsigned int* count;
  unsigned int* status;
  unsigned int cap;
  int k;
  int ncand;
  long long row_offset;
  double* export_val;   // packed copy of the new state for the caller (may be NULL)
  unsigned int ok_lo, ok_hi;   // ncand < 0: a list shorter than ok_lo or longer than ok_hi is left alone (the state too)
};

constexpr size_t REJ_MERGE_LDS = REJ_MAX_K * 16 + 2 * REJ_CHUNK * 16;   // state, the chunk, the exchange buffer: 64 KiB

// one compare-exchange of the bitonic network: this thread keeps the smaller (take_min) or the larger of its pair
__device__ __forceinline__ void rej_cx(double& v, long long& r, double pv, long long pr, bool take_min) {
  if (rej_less(pv, pr, v, r) == take_min) {
    v = pv;
    r = pr;
  }
}

__global__ __launch_bounds__(1024) void reject_merge_kernel(RejArgs S) {
  // Merge by ranks, REJ_CHUNK candidates at a time.  (distance, row) is a strict total order (rows are unique), so an
  // element's place in the merged sequence is the number of elements before it:
  //   1. the chunk is sorted, ONE pair per thread in registers (bitonic network over the next power of two, +inf /
  //      max-row padding): partners less than 64 apart are lanes of the same wave and trade through lane shuffles, no
  //      barrier; only the strides >= 64 -- 10 of the 55 steps of a 1024-chunk -- go through LDS, alternating between two
  //      buffers so that a step costs one barrier;
  //   2. candidate i of the sorted chunk moves to  i + (state entries below it)      -- binary search in the state;
  //      state entry e           moves to  e + (chunk candidates below it)           -- binary search in the chunk;
  //   3. places >= k fall off.  Values travel in registers across the barrier, the scatter is in place.
  // History: a 4096-pair bitonic sort of state + candidates together measured 50 us per merge; ranks of the candidates
  // of one state slot through a linked list (rounds 2-3) cost one dependent LDS round trip per candidate of the slot --
  // about one when a full state meets a few candidates, but the whole chunk when the state is still empty: every
  // candidate falls into slot 0, 1024 list steps each, 57 us per chunk (round 4: 171 us for the 2400 candidates of an SMC
  // round's first batch).  Rounds 4-5 sorted the chunk in LDS, two pairs per thread and a barrier per step (35 us for a
  // merge of a few hundred candidates, 52 us for the two chunks of an SMC round's first batch).  Round 6 also tried 2048
  // candidates per chunk, two pairs per thread: 37 us for 1700 candidates against 34 us for the two 1024-chunks, and every
  // small merge 1-4 us slower -- not kept.
  extern __shared__ __align__(16) unsigned char rej_sm[];
  double* bv = reinterpret_cast<double*>(rej_sm);
  long long* br = reinterpret_cast<long long*>(bv + REJ_MAX_K);
  double* cv = reinterpret_cast<double*>(br + REJ_MAX_K);
  long long* cr = reinterpret_cast<long long*>(cv + REJ_CHUNK);
  double* xv_ = reinterpret_cast<double*>(cr + REJ_CHUNK);       // the second exchange buffer
  long long* xr_ = reinterpret_cast<long long*>(xv_ + REJ_CHUNK);
  const int t = threadIdx.x, k = S.k;
  unsigned int c = S.ncand >= 0 ? (unsigned int)S.ncand : *S.count;
  if (S.ncand < 0 && (c < S.ok_lo || c > S.ok_hi)) return;   // (uniform) the caller's verdict on this list, taken here as well
  if (c > S.cap) {
    if (t == 0) atomicOr(S.status, 1u);   // the list is incomplete: the state can no longer be trusted (reported by result)
    c = S.cap;
  }
  for (int e = t; e < k; e += 1024) {
    bv[e] = S.best_val[e];
    br[e] = S.best_row[e];
  }
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  const long long maxrow = 0x7fffffffffffffffll;
  for (unsigned int c0 = 0; c0 < c; c0 += REJ_CHUNK) {
    const int nc = (int)min((unsigned int)REJ_CHUNK, c - c0);
    int P = 1;
    while (P < nc) P <<= 1;
    double mv = inf;
    long long mr = maxrow;
    if (t < nc) {
      mv = S.cand_val[c0 + t];
      mr = S.cand_row[c0 + t] + S.row_offset;
      if (!(mv == mv)) {   // a NaN never enters the state
        mv = inf;
        mr = maxrow;
      }
    }
    int flip = 0;
    for (int size = 2; size <= P; size <<= 1) {
      const bool asc = (t & size) == 0;
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        const bool take_min = ((t & stride) == 0) == asc;
        double pv;
        long long pr;
        if (stride >= 64) {
          double* ev = flip ? xv_ : cv;
          long long* er = flip ? xr_ : cr;
          flip ^= 1;
          if (t < P) {
            ev[t] = mv;
            er[t] = mr;
          }
          __syncthreads();   // (uniform: P is the same for every thread)
          pv = mv, pr = mr;
          if (t < P) {
            pv = ev[t ^ stride];
            pr = er[t ^ stride];
          }
        } else {
          pv = __shfl_xor(mv, stride, 64);
          pr = __shfl_xor(mr, stride, 64);
        }
        rej_cx(mv, mr, pv, pr, take_min);
      }
    }
    __syncthreads();   // the state of the previous round is in place; every reader of the exchange buffers is done
    if (t < P) {
      cv[t] = mv;
      cr[t] = mr;
    }
    __syncthreads();
    // places
    double sv[2];
    long long sr[2];
    int sp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = t + 1024 * u;
      sp[u] = k;
      sv[u] = 0.0;
      sr[u] = 0;
      if (e < k) {
        sv[u] = bv[e];
        sr[u] = br[e];
        int lo = 0, hi = nc;   // candidates below this entry: first candidate that is not below it
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (rej_less(cv[mid], cr[mid], sv[u], sr[u]))
            lo = mid + 1;
          else
            hi = mid;
        }
        sp[u] = e + lo;
      }
    }
    int cp = k;
    double xv = 0.0;
    long long xr = 0;
    if (t < nc) {
      xv = cv[t];
      xr = cr[t];
      if (xr != maxrow || xv != inf) {   // (padding and NaN candidates sort last and take no part)
        int lo = 0, hi = k;   // state entries below this candidate
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (rej_less(bv[mid], br[mid], xv, xr))
            lo = mid + 1;
          else
            hi = mid;
        }
        cp = t + lo;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (sp[u] < k) {
        bv[sp[u]] = sv[u];
        br[sp[u]] = sr[u];
      }
    if (cp < k) {
      bv[cp] = xv;
      br[cp] = xr;
    }
  }
  __syncthreads();
  for (int e = t; e < k; e += 1024) {
    S.best_val[e] = bv[e];
    S.best_row[e] = br[e];
    if (S.export_val) {
      S.export_val[e] = bv[e];
      reinterpret_cast<long long*>(S.export_val + k)[e] = br[e];
    }
  }
  if (t == 0) {
    *S.thr = bv[k - 1];
    if (S.ncand < 0) *S.count = 0u;
  }
}

// The candidate pass for distance kernels without the fused filter: d (n rows, `stride` apart, pointing at the ranking
// column = the last of `ncols` nested columns) against the threshold.  accept != NULL: a row takes part only if EVERY
// one of its columns is <= *accept (samplers.py:219-225), and the rows that do are counted.
__global__ __launch_bounds__(256) void reject_filter_kernel(const double* d, int64_t n, int64_t stride, int ncols,
                                                            RejectFilter F, int use_accept, const double* acc,
                                                            unsigned long long* acc_count) {
  const double thr = *F.thr;
  const int64_t nround = (n + (int64_t)gridDim.x * 256 - 1) / ((int64_t)gridDim.x * 256);
  unsigned long long mine = 0;
  for (int64_t r = 0; r < nround; ++r) {
    const int64_t i = (r * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    const double v = i < n ? d[i * stride] : 0.0;
    bool ok = i < n;
    if (use_accept && ok)
      for (int c = 0; c < ncols; ++c) ok = ok && d[i * stride - c] <= acc[ncols - 1 - c];
    if (use_accept) mine += __popcll(__ballot(ok));   // uniform branch: every lane of the wave holds the wave's count
    reject_offer(F, ok && v < thr, v, F.row_base + i);
  }
  if (use_accept && (threadIdx.x & 63) == 0 && mine) atomicAdd(acc_count, mine);
}

// An acceptance threshold and no useful k-th distance yet (a new SMC round: samplers.py:474-487 builds a fresh Rejection
// per round, and its thresholds accept a quarter to a half of the proposals): every accepted row would be a candidate --
// 3 10^5 of a batch of 10^6, merged 1024 at a time by one workgroup (17-80 ms per push, measured through
// HipAdaptiveDistanceSMC).  Instead the ranking column is MASKED (+inf where a row is not acceptable), the accepted rows
// are counted, and the batch's min(k, accepted) best come from the radix selection.
__global__ __launch_bounds__(256) void reject_mask_kernel(const double* d, int64_t n, int64_t stride, int ncols,
                                                          const double* acc, double* masked,
                                                          unsigned long long* batch_count, unsigned long long* acc_count) {
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  const int64_t nround = (n + (int64_t)gridDim.x * 256 - 1) / ((int64_t)gridDim.x * 256);
  unsigned long long mine = 0;
  for (int64_t r = 0; r < nround; ++r) {
    const int64_t i = (r * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    bool ok = i < n;
    if (ok)
      for (int c = 0; c < ncols; ++c) ok = ok && d[i * stride - c] <= acc[ncols - 1 - c];
    mine += __popcll(__ballot(ok));   // uniform: every lane of the wave holds the wave's count
    if (i < n) masked[i] = ok ? d[i * stride] : inf;
  }
  if ((threadIdx.x & 63) == 0 && mine) {
    atomicAdd(batch_count, mine);
    atomicAdd(acc_count, mine);
  }
}

__global__ void reject_init_kernel(double* best_val, long long* best_row, double* thr, unsigned int* count,
                                   unsigned int* status, unsigned long long* acc_count, int k) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  if (e < k) {
    best_val[e] = inf;
    best_row[e] = 0x7fffffffffffffffll;
  }
  if (e == 0) {
    *thr = inf;
    *count = 0u;
    *status = 0u;
    *acc_count = 0ull;
  }
}

static RejArgs merge_args(elfihip_reject* h, int ncand, long long row_offset) {
  RejArgs S;
  S.best_val = h->best_val;
  S.best_row = h->best_row;
  S.thr = h->thr;
  S.cand_val = h->cand_val;
  S.cand_row = h->cand_row;
  S.count = h->count;
  S.status = h->status;
  S.cap = h->cap;
  S.k = (int)h->k;
  S.ncand = ncand;
  S.row_offset = row_offset;
  S.export_val = reinterpret_cast<double*>(h->export_dst);
  S.ok_lo = 0u;
  S.ok_hi = ~0u;
  return S;
}

static int host_merge(elfihip_reject* h, unsigned int ncand, long long row_offset);

// merge whatever the list holds (asynchronous on the context's stream; host-merge states synchronise)
static int reject_flush(elfihip_reject* h) {
  if (h->unmerged == 0) return ELFIHIP_OK;
  h->unmerged = 0;
  h->pending_rows = 0;
  if (h->host_mode) return host_merge(h, ~0u, 0);
  hipLaunchKernelGGL(reject_merge_kernel, dim3(1), dim3(1024), REJ_MERGE_LDS, h->ctx->stream, merge_args(h, -1, 0));
  return launch_status(h->ctx, "reject_merge_kernel");
}

static int reject_reset_impl(elfihip_reject* h) {
  hipLaunchKernelGGL(reject_init_kernel, dim3((unsigned)((std::min<int64_t>(h->k, REJ_MAX_K) + 255) / 256)), dim3(256), 0,
                     h->ctx->stream, h->best_val, h->best_row, h->thr, h->count, h->status, h->acc_count,
                     (int)std::min<int64_t>(h->k, REJ_MAX_K));
  h->filled = 0;
  h->unmerged = 0;
  h->armed_pushes = 0;
  h->rows_seen = 0;
  h->pending_rows = 0;
  h->acc_seen = 0;
  h->hval.clear();
  h->hrow.clear();
  h->host_dirty = h->host_mode;
  return launch_status(h->ctx, "reject_init_kernel");
}

// The list holds 8 x the largest batch (>= 65 536 entries): with a merge every 8th push at the latest it cannot
// overflow.  Growing it merges what is pending, waits, and reallocates.
static int ensure_cap(elfihip_reject* h, int64_t n) {
  const int64_t want = std::max<int64_t>(REJ_CAP, REJ_MERGE_EVERY * n);
  if (want <= (int64_t)h->cap) return ELFIHIP_OK;
  elfihip_ctx* ctx = h->ctx;
  ELFIHIP_REQUIRE(ctx, want < (int64_t)1 << 31, "batch of %lld rows is too large for the sampler state", (long long)n);
  ELFIHIP_TRY(reject_flush(h));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  hipError_t e = h->cand_mem.reserve((size_t)want * 16);
  if (e != hipSuccess) return fail(ctx, ELFIHIP_ERR_NOMEM, "candidate list allocation failed: %s", hipGetErrorString(e));
  const size_t have = h->cand_mem.cap / 16;
  h->cap = (unsigned int)std::min<size_t>(have, 0x7fffffffu);
  h->cand_val = h->cand_mem.as<double>();
  h->cand_row = reinterpret_cast<long long*>(h->cand_val + h->cap);
  return ELFIHIP_OK;
}

// ---- host-merge states (k > REJ_MAX_K): the candidates of a push come down, are sorted and merged into the sorted
// host copy; the new k-th distance goes back as the device threshold.  ncand == ~0u: read the list's counter.
static int host_merge(elfihip_reject* h, unsigned int ncand, long long row_offset) {
  elfihip_ctx* ctx = h->ctx;
  hipStream_t st = ctx->stream;
  unsigned int c = ncand;
  if (ncand == ~0u) {
    ELFIHIP_TRY(mail_post(ctx, MailSrc{{h->count, nullptr, nullptr, nullptr}, {4, 0, 0, 0}, 1}));
    ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
    c = (unsigned int)mail_read(ctx, 0);
    ELFIHIP_CHECK_HIP(ctx, hipMemsetAsync(h->count, 0, sizeof c, st));
    if (c > h->cap) return fail(ctx, ELFIHIP_ERR_STATE, "internal: candidate list overflow (%u > %u)", c, h->cap);
  }
  std::vector<double> cv(c);
  std::vector<long long> cr(c);
  if (c) {
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(cv.data(), h->cand_val, (size_t)c * 8, hipMemcpyDeviceToHost, st));
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(cr.data(), h->cand_row, (size_t)c * 8, hipMemcpyDeviceToHost, st));
  }
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  std::vector<size_t> perm;
  perm.reserve(c);
  for (size_t i = 0; i < c; ++i)
    if (cv[i] == cv[i]) perm.push_back(i);   // a NaN never enters the state
  for (size_t i : perm) cr[i] += row_offset;
  std::sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return cv[a] < cv[b] || (cv[a] == cv[b] && cr[a] < cr[b]); });
  const size_t k = (size_t)h->k, ns = h->hval.size();
  std::vector<double> mv;
  std::vector<long long> mr;
  mv.reserve(std::min(k, ns + perm.size()));
  mr.reserve(std::min(k, ns + perm.size()));
  size_t a = 0, b = 0;
  while (mv.size() < k && (a < ns || b < perm.size())) {
    bool take_state;
    if (a >= ns)
      take_state = false;
    else if (b >= perm.size())
      take_state = true;
    else {
      const size_t q = perm[b];
      take_state = h->hval[a] < cv[q] || (h->hval[a] == cv[q] && h->hrow[a] <= cr[q]);
    }
    if (take_state) {
      mv.push_back(h->hval[a]);
      mr.push_back(h->hrow[a]);
      ++a;
    } else {
      mv.push_back(cv[perm[b]]);
      mr.push_back(cr[perm[b]]);
      ++b;
    }
  }
  h->hval.swap(mv);
  h->hrow.swap(mr);
  const double thr = h->hval.size() == k ? h->hval[k - 1] : std::numeric_limits<double>::infinity();
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(h->thr, &thr, sizeof thr, hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));   // `thr` is a stack variable
  h->host_dirty = true;
  return ELFIHIP_OK;
}

// device copy of a host-merge state (state_dev / export): uploaded when asked for
static int host_upload(elfihip_reject* h) {
  if (!h->host_mode || !h->host_dirty) return ELFIHIP_OK;
  elfihip_ctx* ctx = h->ctx;
  const size_t k = (size_t)h->k;
  std::vector<double> v(h->hval);
  std::vector<long long> r(h->hrow);
  v.resize(k, std::numeric_limits<double>::infinity());
  r.resize(k, std::numeric_limits<long long>::max());
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(h->best_val, v.data(), k * 8, hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(h->best_row, r.data(), k * 8, hipMemcpyHostToDevice, ctx->stream));
  if (h->export_dst) {
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(h->export_dst, v.data(), k * 8, hipMemcpyHostToDevice, ctx->stream));
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(reinterpret_cast<char*>(h->export_dst) + k * 8, r.data(), k * 8,
                                          hipMemcpyHostToDevice, ctx->stream));
  }
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  h->host_dirty = false;
  return ELFIHIP_OK;
}

// Fold a batch into the state.  run(F, &filtered) launches the distance pass with the filter F (or without: F == nullptr)
// on the context's stream; dsel / stride address the batch's ranking distances (the last of `ncols` nested columns) for
// the passes that need them.
template <class Run>
static int reject_push(elfihip_reject* h, int64_t n, const double* dsel, int64_t stride, int ncols, long long row_base,
                       Run run) {
  elfihip_ctx* ctx = h->ctx;
  hipStream_t st = ctx->stream;
  ELFIHIP_TRY(ensure_cap(h, n));
  const bool full = h->host_mode ? (int64_t)h->hval.size() >= h->k : h->filled >= h->k;
  // rows this push is expected to offer against the current threshold (batches of one distribution): n k / rows seen
  const double expect = full ? (double)n * (double)h->k / (double)std::max<int64_t>(h->rows_seen, 1) : 1e300;
  const bool select = !h->has_accept && (!full || expect > REJ_HEAVY);
  h->rows_seen += n;
  if (h->has_accept && !h->host_mode && n >= REJ_ACC_SELECT_MIN && (!full || expect > REJ_HEAVY)) {
    // acceptance condition, state still filling up (or very many rows would qualify): mask, count, select (above)
    ELFIHIP_TRY(reject_flush(h));
    bool dummy = false;
    ELFIHIP_TRY(run(nullptr, &dummy));
    ELFIHIP_CHECK_HIP(ctx, h->mask_mem.reserve(((size_t)n + 2) * sizeof(double)));
    unsigned long long* bcount = h->mask_mem.as<unsigned long long>();
    double* masked = h->mask_mem.as<double>() + 2;
    ELFIHIP_CHECK_HIP(ctx, hipMemsetAsync(bcount, 0, sizeof(unsigned long long), st));
    int g = (int)((n + 255) / 256);
    if (g > ctx->cu_count * 8) g = ctx->cu_count * 8;
    hipLaunchKernelGGL(reject_mask_kernel, dim3(g), dim3(256), 0, st, dsel, n, stride, ncols, h->acc_dev, masked, bcount,
                       h->acc_count);
    ELFIHIP_TRY(mail_post(ctx, MailSrc{{bcount, nullptr, nullptr, nullptr}, {8, 0, 0, 0}, 1}));
    ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
    const unsigned long long accepted = mail_read(ctx, 0);
    const int64_t kb = (int64_t)accepted < h->k ? (int64_t)accepted : h->k;
    if (kb > 0) {
      ELFIHIP_TRY(topk_dev_impl(ctx, masked, n, 1, kb, h->cand_val, reinterpret_cast<int64_t*>(h->cand_row), true));
      hipLaunchKernelGGL(reject_merge_kernel, dim3(1), dim3(1024), REJ_MERGE_LDS, st, merge_args(h, (int)kb, row_base));
    }
    return launch_status(ctx, "acceptance push: mask, select, merge");
  }
  if (select) {
    // every row could enter (state still filling up) or very many would: plain distance pass, radix selection of the
    // batch's k best (batch-local rows), merge.  What the list holds from earlier pushes is merged first -- the
    // selection writes its result there.
    ELFIHIP_TRY(reject_flush(h));
    bool dummy = false;
    ELFIHIP_TRY(run(nullptr, &dummy));
    const int64_t kb = n < h->k ? n : h->k;
    if (kb > 0) {
      ELFIHIP_TRY(topk_dev_impl(ctx, dsel, n, stride, kb, h->cand_val, reinterpret_cast<int64_t*>(h->cand_row), true));
      if (h->host_mode)
        ELFIHIP_TRY(host_merge(h, (unsigned int)kb, row_base));
      else
        hipLaunchKernelGGL(reject_merge_kernel, dim3(1), dim3(1024), REJ_MERGE_LDS, st, merge_args(h, (int)kb, row_base));
    }
    h->filled = h->filled + n < h->k ? h->filled + n : h->k;
    return launch_status(ctx, "reject_merge_kernel");
  }
  if (h->pending_rows + n > (int64_t)h->cap) ELFIHIP_TRY(reject_flush(h));   // (cannot happen with a merge every 8th push)
  RejectFilter F;
  F.thr = h->thr;
  F.cval = h->cand_val;
  F.crow = h->cand_row;
  F.count = h->count;
  F.cap = h->cap;
  F.row_base = row_base;
  bool filtered = false;
  // an acceptance threshold needs every column of a row: the separate candidate pass applies it
  ELFIHIP_TRY(run(h->has_accept ? nullptr : &F, &filtered));
  if ((!filtered || h->has_accept) && n > 0) {
    int g = (int)((n + 255) / 256);
    if (g > ctx->cu_count * 8) g = ctx->cu_count * 8;
    hipLaunchKernelGGL(reject_filter_kernel, dim3(g), dim3(256), 0, st, dsel, n, stride, ncols, F, h->has_accept ? 1 : 0,
                       h->acc_dev, h->acc_count);
  }
  h->pending_rows += n;
  // rows that ENTERED the state: every offered row without an acceptance condition; with one, the accepted rows -- the
  // count elfihip_reject_meta reads back (until then the state keeps merging after every push, which is always correct)
  if (!full && !h->has_accept) h->filled = h->filled + n < h->k ? h->filled + n : h->k;
  // Merge interval: the p-th push after the state became full offers about k / p candidates (batches of one
  // distribution), so merging every p / 2 pushes -- at most every REJ_MERGE_EVERY-th -- keeps a merge at about k / 2
  // candidates: early on, while the threshold still falls quickly, after every push.  Host-merge states and states
  // that are still filling merge after every push.
  ++h->armed_pushes;
  int64_t interval = h->armed_pushes / 2;
  interval = interval < 1 ? 1 : (interval > REJ_MERGE_EVERY ? REJ_MERGE_EVERY : interval);
  if (h->host_mode || !full) interval = 1;
  if (++h->unmerged >= interval) return reject_flush(h);
  return launch_status(ctx, "distance pass with selection");
}

elfihip_ctx* reject_ctx(elfihip_reject* h) { return h->ctx; }

int reject_push_rows_impl(elfihip_reject* h, int metric, const double* dX, int64_t n, int m, int64_t ldx,
                          const double* dy, const double* daux, double p, double* dout, int64_t row_base) {
  elfihip_ctx* ctx = h->ctx;
  return reject_push(h, n, dout, 1, 1, (long long)row_base, [&](const RejectFilter* F, bool* filtered) {
    return dist_rows_dev_impl(ctx, metric, dX, n, m, ldx, dy, daux, p, dout, F, filtered);
  });
}

// ---- one AdaptiveDistance batch: distances + column statistics + selection in one read (adaptive.hip) -------------
// The j best rows of a batch's prefix sit at the head of the candidate list with batch-local row numbers: make the
// numbers global, make the j-th distance the threshold of the pass over the rest, and let the list continue behind them.
__global__ __launch_bounds__(256) void reject_seed_kernel(double* thr, unsigned int* count, const double* cand_val,
                                                          long long* cand_row, int j, long long row_base) {
  // (the selection hands the j smallest over in ROW order, not sorted: the threshold is their maximum)
  __shared__ double red[256];
  double m = -__builtin_huge_val();
  for (int e = threadIdx.x; e < j; e += blockDim.x) {
    cand_row[e] += row_base;
    const double v = cand_val[e];
    m = v > m ? v : m;
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s_ = 128; s_ > 0; s_ >>= 1) {
    if ((int)threadIdx.x < s_) red[threadIdx.x] = red[threadIdx.x + s_] > red[threadIdx.x] ? red[threadIdx.x + s_] : red[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *thr = red[0];
    *count = (unsigned int)j;
  }
}

__global__ void reject_unseed_kernel(double* thr, unsigned int* count, const double* best_val, int k) {
  *thr = best_val[k - 1];
  *count = 0u;
}

// h == nullptr: distances and statistics only (what the AdaptiveDistance node computes when a batch is generated).
// dwelford: the running (count, mean, M2) store of add_data, 1 + 2m doubles on the device (nullptr: no statistics).
// dout: (n, K) or nullptr (the distances are then kept only as far as the selection needs them).
int adaptive_push_impl(elfihip_ctx* ctx, elfihip_reject* h, const double* dX, int64_t n, int m, int64_t ldx,
                       const double* dy, const double* dW, int K, double* dout, double* dwelford, int64_t row_base) {
  hipStream_t st = ctx->stream;
  const bool fused = adaptive_pass_supported(dX, m, ldx, K);
  const size_t ns = 1 + 2 * (size_t)m;
  const int maxp = adaptive_max_parts(ctx);
  double *partial = nullptr, *bst = nullptr;
  if (dwelford && fused) {
    ELFIHIP_CHECK_HIP(ctx, ctx->stat.reserve((2 * (size_t)maxp * ns + ns) * sizeof(double)));
    partial = ctx->stat.as<double>();
    bst = partial + 2 * (size_t)maxp * ns;
  }
  int nparts = 0;
  // the pass over rows [lo, hi): distances to o (may be nullptr when fused), statistics into the next partial slots
  auto pass = [&](int64_t lo, int64_t hi, const RejectFilter* F, bool with_acc, double* o) -> int {
    const double* X = dX + lo * ldx;
    if (fused) {
      int np = 0;
      ELFIHIP_TRY(adaptive_pass_impl(ctx, X, hi - lo, m, ldx, dy, dW, K, o, F, with_acc ? h->acc_dev : nullptr,
                                     with_acc ? h->acc_count : nullptr, partial ? partial + (size_t)nparts * ns : nullptr,
                                     &np));
      nparts += np;
      return ELFIHIP_OK;
    }
    return dist_multiw_dev_impl(ctx, X, hi - lo, m, ldx, dy, dW, K, o, nullptr, nullptr);
  };
  auto finish_stats = [&]() -> int {
    if (!dwelford) return ELFIHIP_OK;
    if (fused) return adaptive_stats_finish(ctx, partial, nparts, m, bst, dwelford);
    return welford_dev_impl(ctx, dX, n, m, ldx, dwelford);   // (odd m, wide rows, many weight vectors: the two-pass form)
  };
  // distances the selection can read when the caller keeps none
  auto scratch_out = [&](int64_t rows, double** o) -> int {
    ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)rows * K * sizeof(double)));
    *o = ctx->out.as<double>();
    return ELFIHIP_OK;
  };
  if (!h) {
    double* o = dout;
    if (!fused && !o) ELFIHIP_TRY(scratch_out(n, &o));
    ELFIHIP_TRY(pass(0, n, nullptr, false, o));
    return finish_stats();
  }
  ELFIHIP_REQUIRE(ctx, h->acc_ncols == 0 || h->acc_ncols == K, "%d acceptance thresholds but K = %d nested distances",
                  h->acc_ncols, K);
  ELFIHIP_TRY(ensure_cap(h, n));
  const bool full = h->host_mode ? (int64_t)h->hval.size() >= h->k : h->filled >= h->k;
  const bool acc_select = h->has_accept && !h->host_mode && n >= REJ_ACC_SELECT_MIN &&
                          (!full || (double)n * (double)h->k / (double)std::max<int64_t>(h->rows_seen, 1) > REJ_HEAVY);
  if (!fused || acc_select) {
    // distances (and, fused, the statistics) first, then the state's own candidate pass over them
    double* o = dout;
    if (!o) ELFIHIP_TRY(scratch_out(n, &o));
    ELFIHIP_TRY(pass(0, n, nullptr, false, o));
    ELFIHIP_TRY(reject_push(h, n, o + (K - 1), K, K, (long long)row_base, [&](const RejectFilter*, bool* filtered) {
      *filtered = false;
      return ELFIHIP_OK;
    }));
    return finish_stats();
  }
  const double expect = full ? (double)n * (double)h->k / (double)std::max<int64_t>(h->rows_seen, 1) : 1e300;
  const bool select = !h->has_accept && (!full || expect > REJ_HEAVY);
  h->rows_seen += n;
  auto merge_list = [&](int ncand, long long row_offset) -> int {
    if (h->host_mode) return host_merge(h, ncand < 0 ? ~0u : (unsigned int)ncand, row_offset);
    hipLaunchKernelGGL(reject_merge_kernel, dim3(1), dim3(1024), REJ_MERGE_LDS, st, merge_args(h, ncand, row_offset));
    return launch_status(ctx, "reject_merge_kernel");
  };
  if (select && !full && n >= REJ_PROV_MIN_ROWS && 64 * h->k <= n) {
    // A large first batch (an SMC round starts from an empty state: samplers.py:474-487).  Every row could enter, so
    // there is no threshold for the kernel to filter with, and a radix selection over all n distances costs as much
    // as the pass itself (10^7 x 3: 0.65 ms).  Instead: the j-th smallest distance T of a PREFIX of s rows (j a few
    // standard deviations -- five -- above the k s / n of the batch's k best that fall into the prefix of exchangeable rows) is
    // with overwhelming probability above the batch's k-th smallest, and then the prefix's j best + the rows of the
    // rest below T -- about j n / s of them -- contain the batch's k best.  That is CHECKED (the list's length is read
    // back): if fewer than k rows qualified (the rows were not exchangeable: sorted input), or far too many, the
    // selection of all n distances runs instead.  Exact either way.
    ELFIHIP_TRY(reject_flush(h));
    // (s = n / 16: j n / s = 1700 +- 160 candidates for k = 1000, i.e. two 1024-chunks of the merge; with n / 32 -- rounds 4-5 --
    // they were 2050 +- 250, a third chunk every other round)
    const int64_t s = std::min<int64_t>(std::max<int64_t>(n / 16, 16384), n / 2);
    const double mu = (double)h->k * (double)s / (double)n;
    int64_t j = (int64_t)std::ceil(mu + 5.0 * std::sqrt(mu) + 4.0);
    if (j > h->k) j = h->k;
    double* pre = dout;
    if (!pre) ELFIHIP_TRY(scratch_out(s, &pre));
    ELFIHIP_TRY(pass(0, s, nullptr, false, pre));
    // (the prefix's selection as ONE resident launch: its barrier time-out flag is read back with the list's length below,
    // and a time-out takes the route of a failed check)
    ELFIHIP_TRY(topk_dev_impl(ctx, pre + (K - 1), s, K, j, h->cand_val, reinterpret_cast<int64_t*>(h->cand_row), false));
    const void* sel_err_dev = topk_resident_err_dev(ctx);   // (NULL: the nine-launch form ran -- nothing to time out)
    hipLaunchKernelGGL(reject_seed_kernel, dim3(1), dim3(256), 0, st, h->thr, h->count, h->cand_val, h->cand_row, (int)j,
                       (long long)row_base);
    RejectFilter F;
    F.thr = h->thr;
    F.cval = h->cand_val;
    F.crow = h->cand_row;
    F.count = h->count;
    F.cap = h->cap;
    F.row_base = (long long)row_base + s;
    ELFIHIP_TRY(pass(s, n, &F, false, dout ? dout + s * K : nullptr));
    // the list's length and the selection's time-out flag in ONE read-back
    ELFIHIP_TRY(mail_post(ctx, MailSrc{{h->count, sel_err_dev ? sel_err_dev : h->count, nullptr, nullptr}, {4, 4, 0, 0}, 2}));
    // An EMPTY state (every SMC round starts with one) is merged -- and the batch's statistics are finished -- BEFORE the
    // verdict is known: the host waits for the mail kernel alone (mail_wait), the device works on while it wakes up, and a
    // wrong merge is undone by emptying the state again.  (Rounds 4-5 synchronised the stream here: 20-30 us of idle device
    // per round between the pass and the merge.)
    const bool early = !h->host_mode && h->filled == 0;
    bool stats_done = false;
    const int64_t c_hi = std::min<int64_t>(std::max<int64_t>(REJ_PROV_MAX_CAND, 64 * j), (int64_t)h->cap);
    if (early) {
      // (the merge applies the verdict's bounds on the list's length itself: a list that fails them -- sorted input: a
      // handful of candidates, or every row of the batch -- is not touched, nor is the state)
      RejArgs A = merge_args(h, -1, 0);
      A.ok_lo = (unsigned int)h->k;
      A.ok_hi = (unsigned int)c_hi;
      hipLaunchKernelGGL(reject_merge_kernel, dim3(1), dim3(1024), REJ_MERGE_LDS, st, A);
      ELFIHIP_TRY(launch_status(ctx, "reject_merge_kernel"));
      ELFIHIP_TRY(finish_stats());
      stats_done = true;
      ELFIHIP_TRY(mail_wait(ctx));
    } else {
      ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
    }
    const unsigned int c = (unsigned int)mail_read(ctx, 0);
    const unsigned int sel_err = sel_err_dev ? (unsigned int)mail_read(ctx, 1) : 0u;
    if (sel_err == 0 && (int64_t)c >= h->k && (int64_t)c <= c_hi) {
      if (!early) ELFIHIP_TRY(merge_list(-1, 0));
    } else {
      // the prefix did not represent the batch: selection over all n distances (recomputed when the caller kept none)
      if (early)   // the state as reject_reset leaves it (only a timed-out selection with a list of plausible length has been merged)
        hipLaunchKernelGGL(reject_init_kernel, dim3((unsigned)((std::min<int64_t>(h->k, REJ_MAX_K) + 255) / 256)), dim3(256), 0, st,
                           h->best_val, h->best_row, h->thr, h->count, h->status, h->acc_count,
                           (int)std::min<int64_t>(h->k, REJ_MAX_K));
      hipLaunchKernelGGL(reject_unseed_kernel, dim3(1), dim3(1), 0, st, h->thr, h->count, h->best_val,
                         (int)std::min<int64_t>(h->k, REJ_MAX_K));
      if (h->host_mode) {
        const double inf = std::numeric_limits<double>::infinity();
        const double thr = (int64_t)h->hval.size() >= h->k ? h->hval[h->k - 1] : inf;
        ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(h->thr, &thr, sizeof thr, hipMemcpyHostToDevice, st));
        ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
      }
      double* all = dout;
      if (!all) {
        ELFIHIP_TRY(scratch_out(n, &all));
        ELFIHIP_TRY(dist_multiw_dev_impl(ctx, dX, n, m, ldx, dy, dW, K, all, nullptr, nullptr));
      }
      const int64_t kb = n < h->k ? n : h->k;
      ELFIHIP_TRY(topk_dev_impl(ctx, all + (K - 1), n, K, kb, h->cand_val, reinterpret_cast<int64_t*>(h->cand_row), true));
      ELFIHIP_TRY(merge_list((int)kb, (long long)row_base));
    }
    h->filled = h->k;
    h->unmerged = 0;
    h->pending_rows = 0;
    return stats_done ? ELFIHIP_OK : finish_stats();
  }
  if (select) {
    // (as reject_push: the state is still filling up, or very many rows would qualify) plain pass, radix selection of
    // the batch's k best, merge
    ELFIHIP_TRY(reject_flush(h));
    double* o = dout;
    if (!o) ELFIHIP_TRY(scratch_out(n, &o));
    ELFIHIP_TRY(pass(0, n, nullptr, false, o));
    const int64_t kb = n < h->k ? n : h->k;
    ELFIHIP_TRY(topk_dev_impl(ctx, o + (K - 1), n, K, kb, h->cand_val, reinterpret_cast<int64_t*>(h->cand_row), true));
    ELFIHIP_TRY(merge_list((int)kb, (long long)row_base));
    h->filled = h->filled + n < h->k ? h->filled + n : h->k;
    return finish_stats();
  }
  // the kernel lists the acceptable rows below the state's k-th distance itself
  if (h->pending_rows + n > (int64_t)h->cap) ELFIHIP_TRY(reject_flush(h));
  RejectFilter F;
  F.thr = h->thr;
  F.cval = h->cand_val;
  F.crow = h->cand_row;
  F.count = h->count;
  F.cap = h->cap;
  F.row_base = (long long)row_base;
  ELFIHIP_TRY(pass(0, n, &F, h->has_accept, dout));
  h->pending_rows += n;
  if (!full && !h->has_accept) h->filled = h->filled + n < h->k ? h->filled + n : h->k;
  ++h->armed_pushes;
  int64_t interval = h->armed_pushes / 2;
  interval = interval < 1 ? 1 : (interval > REJ_MERGE_EVERY ? REJ_MERGE_EVERY : interval);
  if (h->host_mode || !full) interval = 1;
  if (++h->unmerged >= interval) ELFIHIP_TRY(reject_flush(h));
  return finish_stats();
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_reject_free(elfihip_reject* h) {
  if (!h) return ELFIHIP_OK;
  DeviceGuard g(h->ctx->device);
  (void)hipStreamSynchronize(h->ctx->stream);
  h->mem.release();
  h->cand_mem.release();
  h->mask_mem.release();
  h->acc_mem.release();
  delete h;
  return ELFIHIP_OK;
}

int elfihip_reject_create(elfihip_ctx* ctx, int64_t k, elfihip_reject** out) {
  if (!ctx || !out) return fail(ctx, ELFIHIP_ERR_ARG, "NULL argument");
  *out = nullptr;
  ELFIHIP_REQUIRE(ctx, k >= 1 && k <= REJ_MAX_K_HOST, "k = %lld outside [1, %lld]", (long long)k, (long long)REJ_MAX_K_HOST);
  DeviceGuard g(ctx->device);
  // (the merge kernel keeps the state and a chunk of candidates in REJ_MERGE_LDS = 64 KiB of dynamic LDS, the most a launch gets without an attribute)
  elfihip_reject* h = new elfihip_reject();
  h->ctx = ctx;
  h->k = k;
  h->host_mode = k > REJ_MAX_K;
  const size_t bytes = (size_t)k * 16 + 64;
  hipError_t e = h->mem.reserve(bytes);
  if (e == hipSuccess) e = h->cand_mem.reserve((size_t)REJ_CAP * 16);
  if (e == hipSuccess) e = h->acc_mem.reserve(REJ_ACC_COLS * sizeof(double));
  if (e != hipSuccess) {
    h->mem.release();
    h->cand_mem.release();
    delete h;
    return fail(ctx, ELFIHIP_ERR_NOMEM, "sampler state allocation failed: %s", hipGetErrorString(e));
  }
  char* p = reinterpret_cast<char*>(h->mem.p);
  h->best_val = reinterpret_cast<double*>(p);
  p += (size_t)k * 8;
  h->best_row = reinterpret_cast<long long*>(p);
  p += (size_t)k * 8;
  h->thr = reinterpret_cast<double*>(p);
  h->count = reinterpret_cast<unsigned int*>(p + 8);
  h->status = reinterpret_cast<unsigned int*>(p + 12);
  h->acc_count = reinterpret_cast<unsigned long long*>(p + 16);
  h->cap = (unsigned int)std::min<size_t>(h->cand_mem.cap / 16, 0x7fffffffu);
  h->cand_val = h->cand_mem.as<double>();
  h->cand_row = reinterpret_cast<long long*>(h->cand_val + h->cap);
  h->acc_dev = h->acc_mem.as<double>();
  int rc = reject_reset_impl(h);
  if (rc != ELFIHIP_OK) {
    elfihip_reject_free(h);
    return rc;
  }
  *out = h;
  return ELFIHIP_OK;
}

int elfihip_reject_reset(elfihip_reject* h) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  DeviceGuard g(h->ctx->device);
  return reject_reset_impl(h);
}

int elfihip_reject_flush(elfihip_reject* h) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  DeviceGuard g(h->ctx->device);
  return reject_flush(h);
}

int elfihip_reject_push_rows_dev(elfihip_reject* h, int metric, const double* dX, int64_t n, int m, int64_t ldx,
                                 const double* dy, const double* daux, double p, double* dout, int64_t row_base) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  elfihip_ctx* ctx = h->ctx;
  ELFIHIP_REQUIRE(ctx, n >= 0 && (n == 0 || dout), "the batch's distances need a destination (dout)");
  DeviceGuard g(ctx->device);
  return reject_push_rows_impl(h, metric, dX, n, m, ldx, dy, daux, p, dout, row_base);
}

int elfihip_reject_push_multiw_dev(elfihip_reject* h, const double* dX, int64_t n, int m, int64_t ldx, const double* dy,
                                   const double* dW, int K, double* dout, int64_t row_base) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  elfihip_ctx* ctx = h->ctx;
  ELFIHIP_REQUIRE(ctx, n >= 0 && K >= 1 && (n == 0 || dout), "the batch's distances need a destination (dout)");
  DeviceGuard g(ctx->device);
  // nested distances are ranked by their LAST column (samplers.py:233)
  return reject_push(h, n, dout ? dout + (K - 1) : nullptr, K, K, (long long)row_base,
                     [&](const RejectFilter* F, bool* filtered) {
                       return dist_multiw_dev_impl(ctx, dX, n, m, ldx, dy, dW, K, dout, F, filtered);
                     });
}

int elfihip_reject_push_dev(elfihip_reject* h, const double* dD, int64_t n, int64_t stride, int64_t row_base) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  elfihip_ctx* ctx = h->ctx;
  ELFIHIP_REQUIRE(ctx, n >= 0 && stride >= 1 && (n == 0 || dD), "bad arguments");
  DeviceGuard g(ctx->device);
  return reject_push(h, n, dD, stride, 1, (long long)row_base, [&](const RejectFilter*, bool* filtered) {
    *filtered = false;   // the distances exist already: candidates come from the separate pass
    return ELFIHIP_OK;
  });
}

int elfihip_reject_push_kept(elfihip_reject* h, uint64_t epoch, int64_t row_base) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  elfihip_ctx* ctx = h->ctx;
  if (epoch != ctx->keep_epoch || !ctx->keep_valid || (ctx->keep_n > 0 && !ctx->keep.p))
    return fail(ctx, ELFIHIP_ERR_STATE, "no kept distances under that name: a later call replaced them, or no copy could "
                "be kept (epoch %llu, asked for %llu)", (unsigned long long)ctx->keep_epoch, (unsigned long long)epoch);
  const int64_t n = ctx->keep_n;
  const int K = ctx->keep_cols;
  ELFIHIP_REQUIRE(ctx, K >= 1 && K <= REJ_ACC_COLS, "ncols = %d outside [1, %d]", K, REJ_ACC_COLS);
  ELFIHIP_REQUIRE(ctx, h->acc_ncols == 0 || h->acc_ncols == K, "%d acceptance thresholds but %d nested columns",
                  h->acc_ncols, K);
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  const double* dD = ctx->keep.as<double>();
  ELFIHIP_TRY(reject_push(h, n, dD + (K - 1), K, K, (long long)row_base, [&](const RejectFilter*, bool* filtered) {
    *filtered = false;
    return ELFIHIP_OK;
  }));
  // the kept copy belongs to the context: what reads it is queued before the next distance call can replace it (same stream)
  return ELFIHIP_OK;
}

int elfihip_reject_state_dev(elfihip_reject* h, double** dvals, int64_t** drows) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  DeviceGuard g(h->ctx->device);
  ELFIHIP_TRY(reject_flush(h));   // work queued on the context's stream after this call sees every batch merged
  ELFIHIP_TRY(host_upload(h));    // (host-merge states: the device copy is brought up to date)
  if (dvals) *dvals = h->best_val;
  if (drows) *drows = reinterpret_cast<int64_t*>(h->best_row);
  return ELFIHIP_OK;
}

int elfihip_reject_export_dev(elfihip_reject* h, void* ddst) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  h->export_dst = ddst;   // from the next merge on; NULL stops it
  return ELFIHIP_OK;
}

int elfihip_reject_result(elfihip_reject* h, double* vals, int64_t* rows, int64_t* count) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  elfihip_ctx* ctx = h->ctx;
  ELFIHIP_REQUIRE(ctx, vals && rows, "NULL result pointer");
  DeviceGuard g(ctx->device);
  ELFIHIP_TRY(reject_flush(h));
  if (h->host_mode) {
    const size_t c = h->hval.size();
    for (size_t i = 0; i < (size_t)h->k; ++i) {
      vals[i] = i < c ? h->hval[i] : std::numeric_limits<double>::infinity();
      rows[i] = i < c ? (int64_t)h->hrow[i] : std::numeric_limits<int64_t>::max();
    }
    if (count) *count = (int64_t)c;
    return ELFIHIP_OK;
  }
  unsigned int status = 0;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(vals, h->best_val, (size_t)h->k * 8, hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(rows, h->best_row, (size_t)h->k * 8, hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(&status, h->status, sizeof status, hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (status & 1u)   // cannot happen: the list holds 8 x the largest batch and is merged every 8th push at the latest
    return fail(ctx, ELFIHIP_ERR_STATE, "internal: the candidate list (%u entries) overflowed", h->cap);
  if (count) {
    // entries in use: slots that hold a row (a NaN distance never enters the state; the reference would list such rows
    // last, after every finite distance; rows above an acceptance threshold never enter either)
    int64_t c = 0;
    while (c < h->k && rows[c] != std::numeric_limits<int64_t>::max()) ++c;
    *count = c;
  }
  return ELFIHIP_OK;
}

int elfihip_reject_set_accept(elfihip_reject* h, int enable, double threshold) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  ELFIHIP_REQUIRE(h->ctx, !enable || threshold == threshold, "the acceptance threshold is NaN");
  ELFIHIP_REQUIRE(h->ctx, h->rows_seen == 0, "set the acceptance threshold before the first push (or after a reset)");
  h->has_accept = enable != 0;
  h->accept = threshold;
  h->acc_ncols = 0;
  if (h->has_accept) {
    DeviceGuard g(h->ctx->device);
    double v[REJ_ACC_COLS];
    for (double& x : v) x = threshold;
    ELFIHIP_CHECK_HIP(h->ctx, hipMemcpyAsync(h->acc_dev, v, sizeof v, hipMemcpyHostToDevice, h->ctx->stream));
    ELFIHIP_CHECK_HIP(h->ctx, hipStreamSynchronize(h->ctx->stream));   // `v` is a stack array
  }
  return ELFIHIP_OK;
}

int elfihip_reject_set_accept_cols(elfihip_reject* h, int ncols, const double* thresholds) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  ELFIHIP_REQUIRE(h->ctx, ncols >= 1 && ncols <= REJ_ACC_COLS && thresholds, "ncols = %d outside [1, %d]", ncols, REJ_ACC_COLS);
  ELFIHIP_REQUIRE(h->ctx, h->rows_seen == 0, "set the acceptance thresholds before the first push (or after a reset)");
  double v[REJ_ACC_COLS];
  for (int c = 0; c < REJ_ACC_COLS; ++c) {
    v[c] = c < ncols ? thresholds[c] : std::numeric_limits<double>::infinity();
    ELFIHIP_REQUIRE(h->ctx, v[c] == v[c], "acceptance threshold %d is NaN", c);
  }
  DeviceGuard g(h->ctx->device);
  ELFIHIP_CHECK_HIP(h->ctx, hipMemcpyAsync(h->acc_dev, v, sizeof v, hipMemcpyHostToDevice, h->ctx->stream));
  ELFIHIP_CHECK_HIP(h->ctx, hipStreamSynchronize(h->ctx->stream));
  h->has_accept = true;
  h->accept = thresholds[ncols - 1];
  h->acc_ncols = ncols;
  return ELFIHIP_OK;
}

int elfihip_reject_push(elfihip_reject* h, const double* D, int64_t n, int ncols, int64_t row_base) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  elfihip_ctx* ctx = h->ctx;
  ELFIHIP_REQUIRE(ctx, n >= 0 && ncols >= 1 && (n == 0 || D), "bad arguments");
  ELFIHIP_REQUIRE(ctx, ncols <= REJ_ACC_COLS, "ncols = %d above %d", ncols, REJ_ACC_COLS);
  ELFIHIP_REQUIRE(ctx, h->acc_ncols == 0 || h->acc_ncols == ncols, "%d acceptance thresholds but %d nested columns",
                  h->acc_ncols, ncols);
  if (n == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  const size_t bytes = (size_t)n * ncols * sizeof(double);
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve(bytes));
  double* dD = ctx->in.as<double>();
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dD, D, bytes, hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_TRY(reject_push(h, n, dD + (ncols - 1), ncols, ncols, (long long)row_base, [&](const RejectFilter*, bool* filtered) {
    *filtered = false;
    return ELFIHIP_OK;
  }));
  // the staging buffer is the context's: merged before the next call may overwrite it (selection route: already done)
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_reject_meta(elfihip_reject* h, double* kth, int64_t* in_use, int64_t* accepted_last, int64_t* accepted_total) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  elfihip_ctx* ctx = h->ctx;
  DeviceGuard g(ctx->device);
  ELFIHIP_TRY(reject_flush(h));
  ELFIHIP_TRY(mail_post(ctx, MailSrc{{h->thr, h->acc_count, nullptr, nullptr}, {8, 8, 0, 0}, 2}));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const unsigned long long thr_bits = mail_read(ctx, 0), acc = mail_read(ctx, 1);
  double thr;
  memcpy(&thr, &thr_bits, sizeof thr);
  if (kth) *kth = thr;                      // +inf while fewer than k rows have entered
  if (in_use) *in_use = h->host_mode ? (int64_t)h->hval.size() : -1;   // device states: ask elfihip_reject_result
  if (accepted_last) *accepted_last = (int64_t)(acc - h->acc_seen);
  if (accepted_total) *accepted_total = (int64_t)acc;
  h->acc_seen = acc;
  if (h->has_accept && !h->host_mode) h->filled = (int64_t)acc < h->k ? (int64_t)acc : h->k;
  return ELFIHIP_OK;
}

}  // extern "C"
