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
// Method: MSD radix select on the order-preserving 64-bit image of the doubles, 11 bits per pass (six
// passes: 5 x 11 + 9): one kernel per pass builds the histogram of the keys that still match the prefix (LDS
// bins, one global atomic per non-empty bin per workgroup -- integer counts, so the result is deterministic;
// a wave whose lanes all carry the same digit, the normal case in the exponent bits, adds its count with a
// single LDS atomic) and the LAST workgroup to finish picks the digit (parallel scan of the 2048 bins);
// then a stable two-pass compaction (per-block counts, scan, write) of the keys below the k-th key plus as
// many equal ones as needed.  Nine launches, 8 bytes per distance per pass -- this form is the fall-back.
// The default is the SAME passes inside one resident kernel (sel_persistent_kernel below: grid barriers instead of
// launches, and a prefix class of at most 1024 keys is resolved in one step): 10^6 keys, k = 1000: 0.107 -> 0.061 ms;
// 4096 keys: 0.076 -> 0.020 ms.
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <numeric>

#include "common.hpp"
#include "internal.hpp"


// The hand-offs between workgroups in this file (write-through stores, s_waitcnt vmcnt(0), relaxed agent-scope atomics, one
// acquire at the consumer) rely on the gfx9 family counting stores in vmcnt and on sc1 atomics writing through.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "built for gfx950 (MI355X): the cross-workgroup hand-offs here are not valid on this target"
#endif

namespace elfihip {

constexpr int SEL_BITS = 11, SEL_BINS = 1 << SEL_BITS, SEL_PASSES = 6;

struct SelState {
  unsigned long long prefix;   // bits decided so far (high bits), rest zero
  unsigned long long k_rem;    // rank still to find inside the prefix class (1-based)
  unsigned long long n_lt;     // keys strictly below the prefix class
  unsigned int done;           // workgroups of the current pass that have added their bins
  unsigned int hist[SEL_BINS];
};

__device__ __forceinline__ unsigned long long key_of(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  if (v != v) return ~0ull;                                   // NaN last
  if (v == 0.0) b = 0ull;                                     // -0.0 and +0.0 are one key (NumPy compares them equal)
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// pass p looks at bits [shift, shift + width): 53/11, 42/11, 31/11, 20/11, 9/11, 0/9
__device__ __forceinline__ int sel_shift(int pass) { return pass < 5 ? 53 - SEL_BITS * pass : 0; }
__device__ __forceinline__ int sel_width(int pass) { return pass < 5 ? SEL_BITS : 9; }

__global__ __launch_bounds__(256) void sel_hist_kernel(const double* d, int64_t n, int64_t stride, int pass,
                                                       SelState* st) {
  __shared__ unsigned int h[SEL_BINS];
  __shared__ unsigned int part[256];
  __shared__ int last;
  for (int b = threadIdx.x; b < SEL_BINS; b += 256) h[b] = 0;
  __syncthreads();
  const int shift = sel_shift(pass), width = sel_width(pass);
  const unsigned int dmask = (1u << width) - 1u;
  const unsigned long long prefix = st->prefix;
  const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + width));
  // eight rows per thread and trip, all eight loads in flight before the first is looked at (the scan is
  // latency-bound otherwise: a wave would wait a full memory round trip per 512 bytes)
  constexpr int U = 8;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 * U; i0 < n; i0 += (int64_t)gridDim.x * 256 * U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 256 + threadIdx.x;
      v[u] = i < n ? d[i * stride] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 256 + threadIdx.x;
      const unsigned long long k = key_of(v[u]);
      const bool in = i < n && (k & himask) == prefix;
      const unsigned int digit = (unsigned int)(k >> shift) & dmask;
      const unsigned long long members = __ballot(in);
      if (members == 0) continue;
      // the wave's first member's digit; one atomic for the whole wave when every member shares it
      const int lead = __ffsll((long long)members) - 1;
      const unsigned int d0 = (unsigned int)__builtin_amdgcn_readlane((int)digit, lead);
      if (__ballot(in && digit == d0) == members) {
        if ((int)(threadIdx.x & 63) == lead) atomicAdd(&h[d0], (unsigned int)__popcll(members));
      } else if (in) {
        atomicAdd(&h[digit], 1u);
      }
    }
  }
  __syncthreads();
  // one wave adds the workgroup's bins to the global histogram, orders them before its arrival mark (a single
  // device-scope fence per workgroup: the fence is the expensive part) and learns whether it arrived last
  if (threadIdx.x < 64) {
    for (int b = threadIdx.x; b < SEL_BINS; b += 64)
      if (h[b]) atomicAdd(&st->hist[b], h[b]);
    __threadfence();
    if (threadIdx.x == 0) last = (atomicAdd(&st->done, 1u) == gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  // the last workgroup to arrive picks the digit of this pass
  constexpr int PER = SEL_BINS / 256;  // bins per thread
  unsigned int c[PER], sum = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    c[q] = atomicAdd(&st->hist[threadIdx.x * PER + q], 0u);  // coherent read of the other workgroups' adds
    sum += c[q];
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {  // exclusive scan of 256 partial sums: short and sequential
    unsigned int run = 0;
    for (int t = 0; t < 256; ++t) {
      const unsigned int v = part[t];
      part[t] = run;
      run += v;
    }
  }
  __syncthreads();
  const unsigned long long rem = st->k_rem;
  const unsigned long long below0 = part[threadIdx.x];
  if (rem > below0 && rem <= below0 + sum) {  // exactly one thread owns the k-th key's bin range
    unsigned long long below = below0;
    int digit = threadIdx.x * PER;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (rem > below + c[q]) {
        below += c[q];
        digit = threadIdx.x * PER + q + 1;
      } else {
        break;
      }
    }
    st->prefix = prefix | ((unsigned long long)digit << shift);
    st->k_rem = rem - below;
    st->n_lt += below;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < SEL_BINS; b += 256) st->hist[b] = 0;
  if (threadIdx.x == 0) st->done = 0;
}

// the state of a selection before its first pass
__global__ __launch_bounds__(256) void sel_init_kernel(SelState* st, unsigned long long k) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < SEL_BINS) st->hist[t] = 0u;
  if (t == 0) {
    st->prefix = 0ull;
    st->k_rem = k;
    st->n_lt = 0ull;
    st->done = 0u;
  }
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
__global__ __launch_bounds__(256) void sel_scan_kernel(unsigned int* counts, int nblocks) {
  // both classes at once: thread t owns the contiguous run of blocks [t * per, t * per + per)
  __shared__ unsigned int tot[2][256];
  const int per = (nblocks + 255) / 256;
  const int b0 = threadIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  unsigned int s0 = 0, s1 = 0;
  for (int b = b0; b < b1; ++b) {
    s0 += counts[2 * b];
    s1 += counts[2 * b + 1];
  }
  tot[0][threadIdx.x] = s0;
  tot[1][threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned int run = 0;
    for (int t = 0; t < 256; ++t) {
      const unsigned int v = tot[threadIdx.x][t];
      tot[threadIdx.x][t] = run;
      run += v;
    }
  }
  __syncthreads();
  unsigned int r0 = tot[0][threadIdx.x], r1 = tot[1][threadIdx.x];
  for (int b = b0; b < b1; ++b) {
    const unsigned int c0 = counts[2 * b], c1 = counts[2 * b + 1];
    counts[2 * b] = r0;
    counts[2 * b + 1] = r1;
    r0 += c0;
    r1 += c1;
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

// ---- the same selection in ONE launch ----
// The nine kernels above cost about 12 us each at 10^6 keys (launch + fence + last-workgroup hand-over), the work in
// them about 2.  Here one workgroup per CU stays resident for the whole selection: the six histogram passes, the count
// and the stable write are separated by grid barriers (a monotonic arrival counter in device memory, one fence per
// workgroup and barrier), every workgroup picks the digit of a pass redundantly from the global histogram (one
// histogram per pass, zeroed once per call, so nothing has to be reset or broadcast), and the per-workgroup write
// offsets are prefix sums over at most 512 count pairs that every workgroup forms for itself.
// All workgroups are resident at once (grid <= number of CUs, 256 threads, 9 KiB of LDS); a workgroup that cannot be
// placed yet because other kernels fill the device only delays the others.  The spin is bounded: after 2^24 polls a
// workgroup raises `err` and leaves, so a lost workgroup can never hang the device.
constexpr int SEL_MAX_GRID = 512;

constexpr int SEL_SMALL = 1024;  // a prefix class this small is resolved in one step instead of further passes

struct SelWork {
  unsigned int hist[SEL_PASSES][SEL_BINS];
  unsigned int counts[2 * SEL_MAX_GRID];
  unsigned long long cand[SEL_SMALL];
  unsigned int ncand;
  unsigned int bar;
  unsigned int err;
};

// Grid barrier WITHOUT cache-maintenance fences.  Everything the workgroups exchange goes through agent-scope accesses on
// both sides -- histogram bins and counters by device-scope atomics, candidate keys and per-workgroup counts by
// write-through (sc1) stores, every reader by sc1 loads (__hip_atomic_load relaxed / agent), which never hit the CU's L1
// -- so an arrival only has to wait for the wave's own memory operations (s_waitcnt vmcnt(0): stores, atomics) and bump
// the counter.  MI355X_MICROARCH.md's price list: a counter barrier between two __threadfence() (this kernel until round
// 2: buffer_wbl2 + buffer_inv of a whole L2 / L1 per workgroup and barrier) 9-9.3 us, against 3-4.5 us for the
// fan-in + broadcast alone; five of them per selection.
__device__ __forceinline__ void sel_grid_barrier(SelWork* w, unsigned int target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY wave: its write-through stores and atomics have landed
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&w->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int spins = 0;
    while (__hip_atomic_load(&w->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) {
        atomicExch(&w->err, 1u);
        break;
      }
    }
  }
  __syncthreads();
}

constexpr int SEL_NT = 1024;  // threads of a resident workgroup: few workgroups (few contenders per hot histogram bin
                              // and at the barrier counter), many waves each

__global__ __launch_bounds__(SEL_NT) void sel_persistent_kernel(const double* d, int64_t n, int64_t stride,
                                                             int64_t per_block, int64_t k, SelWork* w, double* vals,
                                                             int64_t* idx) {
  __shared__ unsigned int h[SEL_BINS];
  __shared__ unsigned int part[256];
  __shared__ unsigned long long picked[4];
  __shared__ unsigned long long cand[SEL_SMALL];
  __shared__ unsigned int wl[SEL_NT / 64], we[SEL_NT / 64], base[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned int G = gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per_block;
  const int64_t hi = lo + per_block < n ? lo + per_block : n;
  unsigned long long prefix = 0, rem = (unsigned long long)k, n_lt = 0;
  unsigned int bar_no = 0;  // barriers passed so far (the same in every workgroup: control flow depends on shared data only)
  for (int pass = 0; pass < SEL_PASSES; ++pass) {
    for (int b = tid; b < SEL_BINS; b += SEL_NT) h[b] = 0;
    __syncthreads();
    const int shift = sel_shift(pass), width = sel_width(pass);
    const unsigned int dmask = (1u << width) - 1u;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + width));
    constexpr int U = 8;
    for (int64_t i0 = lo; i0 < hi; i0 += SEL_NT * U) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * SEL_NT + tid;
        v[u] = i < hi ? d[i * stride] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * SEL_NT + tid;
        const unsigned long long kk = key_of(v[u]);
        const bool in = i < hi && (kk & himask) == prefix;
        const unsigned int digit = (unsigned int)(kk >> shift) & dmask;
        const unsigned long long members = __ballot(in);
        if (members == 0) continue;
        const int lead = __ffsll((long long)members) - 1;
        const unsigned int d0 = (unsigned int)__builtin_amdgcn_readlane((int)digit, lead);
        if (__ballot(in && digit == d0) == members) {
          if (lane == lead) atomicAdd(&h[d0], (unsigned int)__popcll(members));
        } else if (in) {
          atomicAdd(&h[digit], 1u);
        }
      }
    }
    __syncthreads();
    for (int b = tid; b < SEL_BINS; b += SEL_NT)
      if (h[b]) atomicAdd(&w->hist[pass][b], h[b]);
    sel_grid_barrier(w, G * ++bar_no);
    // every workgroup picks the digit of this pass from the complete histogram
    constexpr int PER = SEL_BINS / 256;  // the first 256 threads, 8 bins each
    unsigned int c[PER], sum = 0;
    if (tid < 256) {
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        c[q] = __hip_atomic_load(&w->hist[pass][tid * PER + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += c[q];
      }
      part[tid] = sum;
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan of the 256 partial sums: four per lane, then across the wave
      unsigned int v0 = part[4 * tid], v1 = part[4 * tid + 1], v2 = part[4 * tid + 2], v3 = part[4 * tid + 3];
      const unsigned int mine = v0 + v1 + v2 + v3;
      unsigned int incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (tid >= off) incl += o;
      }
      unsigned int run = incl - mine;
      part[4 * tid] = run;
      run += v0;
      part[4 * tid + 1] = run;
      run += v1;
      part[4 * tid + 2] = run;
      run += v2;
      part[4 * tid + 3] = run;
    }
    __syncthreads();
    if (tid < 256) {
      const unsigned long long below0 = part[tid];
      if (rem > below0 && rem <= below0 + sum) {  // exactly one thread owns the k-th key's bin range
        unsigned long long below = below0;
        int digit = tid * PER;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          if (rem > below + c[q]) {
            below += c[q];
            digit = tid * PER + q + 1;
          } else {
            break;
          }
        }
        picked[0] = prefix | ((unsigned long long)digit << shift);
        picked[1] = rem - below;
        picked[2] = n_lt + below;
        picked[3] = c[digit - tid * PER];  // size of the chosen class
      }
    }
    __syncthreads();
    prefix = picked[0];
    rem = picked[1];
    n_lt = picked[2];
    const unsigned long long csize = picked[3];
    __syncthreads();
    if (pass + 1 < SEL_PASSES && csize <= SEL_SMALL) {
      // few keys left in the class: collect them (order is irrelevant, their values decide) and let every workgroup find
      // the rem-th smallest among them by rank counting -- one step instead of the remaining passes
      const unsigned long long cmask = ~0ull << shift;
      for (int64_t i0 = lo; i0 < hi; i0 += SEL_NT * U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t i = i0 + u * SEL_NT + tid;
          v[u] = i < hi ? d[i * stride] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t i = i0 + u * SEL_NT + tid;
          const unsigned long long kk = key_of(v[u]);
          if (i < hi && (kk & cmask) == prefix) {   // write-through store: visible to the other XCDs without a fence
            const unsigned int at = atomicAdd(&w->ncand, 1u);
            if (at < (unsigned int)SEL_SMALL) __hip_atomic_store(&w->cand[at], kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      sel_grid_barrier(w, G * ++bar_no);
      const int m = (int)csize;
      for (int j = tid; j < m; j += SEL_NT)
        cand[j] = __hip_atomic_load(&w->cand[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (tid < m) {
        const unsigned long long mine = cand[tid];
        unsigned int lt = 0, eq_before = 0;
        for (int j = 0; j < m; ++j) {
          const unsigned long long o = cand[j];
          lt += o < mine;
          eq_before += (o == mine) && j < tid;
        }
        if (lt + eq_before == (unsigned int)(rem - 1)) {  // exactly one entry has rank rem - 1
          picked[0] = mine;
          picked[2] = n_lt + lt;
        }
      }
      __syncthreads();
      prefix = picked[0];
      n_lt = picked[2];
      __syncthreads();
      break;
    }
  }
  const unsigned long long kth = prefix;
  // ---- counts of this workgroup's slice, then everybody's offsets
  {
    unsigned int lt = 0, eq = 0;
    for (int64_t i = lo + tid; i < hi; i += SEL_NT) {
      const unsigned long long kk = key_of(d[i * stride]);
      lt += kk < kth;
      eq += kk == kth;
    }
    if (tid < 2) base[tid] = 0;
    __syncthreads();
    if (lt) atomicAdd(&base[0], lt);
    if (eq) atomicAdd(&base[1], eq);
    __syncthreads();
    if (tid < 2) __hip_atomic_store(&w->counts[2 * blockIdx.x + tid], base[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  sel_grid_barrier(w, G * ++bar_no);
  {
    unsigned int o0 = 0, o1 = 0;
    for (unsigned int b = tid; b < blockIdx.x; b += SEL_NT) {
      o0 += __hip_atomic_load(&w->counts[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      o1 += __hip_atomic_load(&w->counts[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < 2) base[tid] = 0;
    __syncthreads();
    if (o0) atomicAdd(&base[0], o0);
    if (o1) atomicAdd(&base[1], o1);
    __syncthreads();
  }
  // ---- stable write (sel_write_kernel's loop)
  for (int64_t c0 = lo; c0 < hi; c0 += SEL_NT) {
    const int64_t i = c0 + tid;
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
      wl[wv] = (unsigned int)__popcll(bl);
      we[wv] = (unsigned int)__popcll(be);
    }
    __syncthreads();
    unsigned int pl = 0, pe = 0;
    for (int q = 0; q < wv; ++q) {
      pl += wl[q];
      pe += we[q];
    }
    if (is_lt) {
      const int64_t pos = (int64_t)base[0] + pl + __popcll(bl & below);
      vals[pos] = v;
      idx[pos] = i;
    } else if (is_eq) {
      const int64_t pos = (int64_t)n_lt + (int64_t)base[1] + pe + __popcll(be & below);
      if (pos < k) {
        vals[pos] = v;
        idx[pos] = i;
      }
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int tl = 0, te = 0;
      for (int q = 0; q < SEL_NT / 64; ++q) {
        tl += wl[q];
        te += we[q];
      }
      base[0] += tl;
      base[1] += te;
    }
    __syncthreads();
  }
}

// ---- the resident selection with the slice's keys kept ON THE CHIP (round 6) ----
// sel_persistent_kernel reads its slice from memory in every phase -- up to four histogram passes, the collection of a small
// class, the count, the write -- and its stable write walks the slice 1024 keys at a time with three workgroup barriers per
// step: 10^6 keys, k = 1000 took 56 us, of which the arithmetic is a few.  For slices of at most SEL_RU x SEL_NT = 16 384
// keys (n <= 2 x 10^6 on this chip: every selection the sampler makes except the fall-back over a whole 10^7-row batch) the
// workgroup keeps the 64-bit keys of its slice in 128 KiB of LDS for the whole selection: ONE read of the distances, every
// later phase from LDS, and -- wave w owning the CONTIGUOUS rows [1024 w, 1024 w + 1024) of the slice, lane l the rows
// 64 u + l, each lane reading back only words it wrote itself (no barrier guards the slice) -- a stable write whose
// positions are wave-ballot prefix counts: one barrier for the sixteen wave totals instead of forty-eight.  The values of
// the k survivors are fetched again from the input (the keys are not invertible for -0.0 and NaN payloads).
// Why LDS and not registers: the first form of this kernel held sixteen values per thread in registers, which needs every
// loop over them unrolled -- 30 KB of code, and a kernel that runs each instruction once or twice pays for FETCHING it:
// the first histogram pass took 12.5 us against 3.4 us for the second pass through the same (then cached) code
// (wall-clock stamps, scripts/native/sel_probe.hip).  With the keys in LDS the loops stay rolled.
// Same passes, same barriers between workgroups, same result (row order inside "< k-th", then "== k-th") as
// sel_persistent_kernel.
#ifdef ELFIHIP_SEL_STAMP   // developer probe (scripts/native/sel_probe.hip): wall-clock stamps (100 MHz) per workgroup and phase
__device__ unsigned long long g_sel_stamp[512 * 32];
#define SEL_STAMP() do { if (threadIdx.x == 0 && sel_si < 32) g_sel_stamp[blockIdx.x * 32 + sel_si] = __builtin_amdgcn_s_memrealtime(); ++sel_si; } while (0)
#else
#define SEL_STAMP() do { } while (0)
#endif

constexpr int SEL_RU = 16;
constexpr size_t SEL_SLICE_LDS = (size_t)SEL_RU * SEL_NT * sizeof(unsigned long long);   // 128 KiB of dynamic LDS

__global__ __launch_bounds__(SEL_NT) void sel_resident_lds_kernel(const double* d, int64_t n, int64_t stride, int64_t k,
                                                                  SelWork* w, double* vals, int64_t* idx) {
  extern __shared__ __align__(16) unsigned long long sel_keys[];   // [wave][u][lane]
  __shared__ unsigned int h[SEL_BINS];
  __shared__ unsigned int part[256];
  __shared__ unsigned long long picked[4];
  __shared__ unsigned long long cand[SEL_SMALL];
  __shared__ unsigned int wl[SEL_NT / 64], we[SEL_NT / 64], base[2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef ELFIHIP_SEL_STAMP
  int sel_si = 0;
#endif
  SEL_STAMP();   // 0: start
  const unsigned int G = gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * (SEL_RU * SEL_NT);
  const int64_t hi = lo + SEL_RU * SEL_NT < n ? lo + SEL_RU * SEL_NT : n;
  const int64_t first = lo + (int64_t)wv * (64 * SEL_RU) + lane;   // this lane's rows: first + 64 u, u < nv
  const int64_t room = hi - first;
  const int nv = room <= 0 ? 0 : (int)(room + 63 >> 6 < SEL_RU ? room + 63 >> 6 : SEL_RU);
  const int nvw = __builtin_amdgcn_readfirstlane(nv);   // lane 0 owns the wave's lowest rows: its count bounds the wave's
  unsigned long long* mine = sel_keys + wv * (64 * SEL_RU) + lane;   // key u at mine[64 u]
  {
    double v[SEL_RU];
#pragma unroll
    for (int u = 0; u < SEL_RU; ++u) v[u] = u < nv ? d[(first + 64 * u) * stride] : 0.0;   // all loads in flight together
#pragma unroll
    for (int u = 0; u < SEL_RU; ++u) mine[64 * u] = key_of(v[u]);
  }
  SEL_STAMP();   // 1: keys in LDS
#ifdef ELFIHIP_SEL_STAMP
  __syncthreads();
  SEL_STAMP();   // 2 (stamped build only): every wave's keys are in LDS
#endif
  unsigned long long prefix = 0, rem = (unsigned long long)k, n_lt = 0;
  unsigned int bar_no = 0;
  for (int pass = 0; pass < SEL_PASSES; ++pass) {
    for (int b = tid; b < SEL_BINS; b += SEL_NT) h[b] = 0;
    __syncthreads();
    const int shift = sel_shift(pass), width = sel_width(pass);
    const unsigned int dmask = (1u << width) - 1u;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + width));
    if (pass == 0) {
      // The exponent bits: a workgroup's 16 384 keys share a handful of digits, and LDS atomics on a handful of words are
      // serialised -- one atomic per key made this pass 8 us against 2 us for a pass whose digits spread (stamps:
      // scripts/native/sel_probe.hip).  Every THREAD therefore counts up to four digits of its own sixteen keys in
      // registers (compares and adds, nothing crosses lanes) and adds them once; a key with a fifth digit adds itself.
      // (Measured against it, first + second pass: one atomic per wave, digit and 64 keys for up to four digits 16 + 2.8 us;
      // four digits counted per WAVE in scalar registers 9.2 + 6.9 us -- the ballots cost more than they save.)
      unsigned int pd0 = ~0u, pd1 = ~0u, pd2 = ~0u, pd3 = ~0u, pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0;
      for (int u = 0; u < nv; ++u) {
        const unsigned int digit = (unsigned int)(mine[64 * u] >> shift) & dmask;
        if (digit == pd0)
          ++pc0;
        else if (digit == pd1)
          ++pc1;
        else if (digit == pd2)
          ++pc2;
        else if (digit == pd3)
          ++pc3;
        else if (pd0 == ~0u)
          pd0 = digit, pc0 = 1;
        else if (pd1 == ~0u)
          pd1 = digit, pc1 = 1;
        else if (pd2 == ~0u)
          pd2 = digit, pc2 = 1;
        else if (pd3 == ~0u)
          pd3 = digit, pc3 = 1;
        else
          atomicAdd(&h[digit], 1u);
      }
      if (pc0) atomicAdd(&h[pd0], pc0);
      if (pc1) atomicAdd(&h[pd1], pc1);
      if (pc2) atomicAdd(&h[pd2], pc2);
      if (pc3) atomicAdd(&h[pd3], pc3);
    } else {
#pragma unroll 2
      for (int u = 0; u < nvw; ++u) {
        const unsigned long long ku = mine[64 * u];
        const bool in = u < nv && (ku & himask) == prefix;
        const unsigned int digit = (unsigned int)(ku >> shift) & dmask;
        const unsigned long long members = __ballot(in);
        if (members == 0) continue;
        const int lead = __ffsll((long long)members) - 1;
        const unsigned int d0 = (unsigned int)__builtin_amdgcn_readlane((int)digit, lead);
        if (__ballot(in && digit == d0) == members) {   // one atomic for the wave when every member shares the digit
          if (lane == lead) atomicAdd(&h[d0], (unsigned int)__popcll(members));
        } else if (in) {
          atomicAdd(&h[digit], 1u);
        }
      }
    }
    __syncthreads();
    SEL_STAMP();   // pass: workgroup histogram complete
    for (int b = tid; b < SEL_BINS; b += SEL_NT)
      if (h[b]) atomicAdd(&w->hist[pass][b], h[b]);
    SEL_STAMP();   // pass: histogram flushed
    sel_grid_barrier(w, G * ++bar_no);
    SEL_STAMP();   // pass: barrier passed
    // every workgroup picks the digit of this pass from the complete histogram (as sel_persistent_kernel)
    constexpr int PER = SEL_BINS / 256;
    unsigned int c[PER], sum = 0;
    if (tid < 256) {
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        c[q] = __hip_atomic_load(&w->hist[pass][tid * PER + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += c[q];
      }
      part[tid] = sum;
    }
    __syncthreads();
    if (tid < 64) {
      unsigned int v0 = part[4 * tid], v1 = part[4 * tid + 1], v2 = part[4 * tid + 2], v3 = part[4 * tid + 3];
      const unsigned int mine_s = v0 + v1 + v2 + v3;
      unsigned int incl = mine_s;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned int o = __shfl_up(incl, off, 64);
        if (tid >= off) incl += o;
      }
      unsigned int run = incl - mine_s;
      part[4 * tid] = run;
      run += v0;
      part[4 * tid + 1] = run;
      run += v1;
      part[4 * tid + 2] = run;
      run += v2;
      part[4 * tid + 3] = run;
    }
    __syncthreads();
    if (tid < 256) {
      const unsigned long long below0 = part[tid];
      if (rem > below0 && rem <= below0 + sum) {
        unsigned long long below = below0;
        int digit = tid * PER;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          if (rem > below + c[q]) {
            below += c[q];
            digit = tid * PER + q + 1;
          } else {
            break;
          }
        }
        picked[0] = prefix | ((unsigned long long)digit << shift);
        picked[1] = rem - below;
        picked[2] = n_lt + below;
        picked[3] = c[digit - tid * PER];
      }
    }
    __syncthreads();
    prefix = picked[0];
    rem = picked[1];
    n_lt = picked[2];
    const unsigned long long csize = picked[3];
    __syncthreads();
    SEL_STAMP();   // pass: digit picked
    if (pass + 1 < SEL_PASSES && csize <= SEL_SMALL) {
      const unsigned long long cmask = ~0ull << shift;
      // the class's keys of this workgroup: places inside the workgroup from an LDS counter, ONE global reservation per
      // workgroup (a global atomic per key put up to 1024 of them on one address)
      unsigned int mine_n = 0;
#pragma unroll 2
      for (int u = 0; u < nv; ++u) mine_n += (mine[64 * u] & cmask) == prefix ? 1u : 0u;
      if (tid < 2) base[tid] = 0;
      __syncthreads();
      unsigned int at = mine_n ? atomicAdd(&base[0], mine_n) : 0u;
      __syncthreads();
      if (tid == 0 && base[0]) base[1] = atomicAdd(&w->ncand, base[0]);
      __syncthreads();
      if (mine_n) {
        at += base[1];
        for (int u = 0; u < nv; ++u) {
          const unsigned long long ku = mine[64 * u];
          if ((ku & cmask) == prefix && at < (unsigned int)SEL_SMALL)   // write-through store: visible to the other XCDs without a
            __hip_atomic_store(&w->cand[at++], ku, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // fence (the bound only
                                                                                                    // matters if the input changes under the kernel)
        }
      }
      SEL_STAMP();   // small class: keys stored
      sel_grid_barrier(w, G * ++bar_no);
      SEL_STAMP();   // small class: barrier passed
      const int m = (int)csize;
      for (int j = tid; j < m; j += SEL_NT)
        cand[j] = __hip_atomic_load(&w->cand[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (tid < m) {
        const unsigned long long me = cand[tid];
        unsigned int lt = 0, eq_before = 0;
        for (int j = 0; j < m; ++j) {
          const unsigned long long o = cand[j];
          lt += o < me;
          eq_before += (o == me) && j < tid;
        }
        if (lt + eq_before == (unsigned int)(rem - 1)) {
          picked[0] = me;
          picked[2] = n_lt + lt;
        }
      }
      __syncthreads();
      prefix = picked[0];
      n_lt = picked[2];
      __syncthreads();
      SEL_STAMP();   // small class: ranked
      break;
    }
  }
  const unsigned long long kth = prefix;
  // ---- this workgroup's counts: wave totals from ballots, then everybody's offsets
  unsigned int wlt = 0, weq = 0;   // wave-uniform
#pragma unroll 2
  for (int u = 0; u < nvw; ++u) {
    const unsigned long long ku = mine[64 * u];
    wlt += (unsigned int)__popcll(__ballot(u < nv && ku < kth));
    weq += (unsigned int)__popcll(__ballot(u < nv && ku == kth));
  }
  if (lane == 0) {
    wl[wv] = wlt;
    we[wv] = weq;
  }
  __syncthreads();
  unsigned int pl = 0, pe = 0, tl = 0, te = 0;   // keys of the waves before this one; of the whole workgroup
  for (int q = 0; q < SEL_NT / 64; ++q) {
    if (q < wv) pl += wl[q], pe += we[q];
    tl += wl[q];
    te += we[q];
  }
  if (tid == 0) {
    __hip_atomic_store(&w->counts[2 * blockIdx.x], tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&w->counts[2 * blockIdx.x + 1], te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  SEL_STAMP();   // counts stored
  sel_grid_barrier(w, G * ++bar_no);
  SEL_STAMP();   // counts: barrier passed
  {
    unsigned int o0 = 0, o1 = 0;
    for (unsigned int b = tid; b < blockIdx.x; b += SEL_NT) {
      o0 += __hip_atomic_load(&w->counts[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      o1 += __hip_atomic_load(&w->counts[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 2) base[tid] = 0;
    __syncthreads();
    if (o0) atomicAdd(&base[0], o0);
    if (o1) atomicAdd(&base[1], o1);
    __syncthreads();
  }
  SEL_STAMP();   // offsets formed
  // ---- stable write: slice order = wave order, then u, then lane
  int64_t at_l = (int64_t)base[0] + pl, at_e = (int64_t)n_lt + (int64_t)base[1] + pe;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll 2
  for (int u = 0; u < nvw; ++u) {
    const int64_t i = first + 64 * u;
    const unsigned long long ku = mine[64 * u];
    const bool is_lt = u < nv && ku < kth, is_eq = u < nv && ku == kth;
    const unsigned long long bl = __ballot(is_lt), be = __ballot(is_eq);
    if (is_lt) {
      const int64_t pos = at_l + __popcll(bl & below);
      vals[pos] = d[i * stride];
      idx[pos] = i;
    } else if (is_eq) {
      const int64_t pos = at_e + __popcll(be & below);
      if (pos < k) {
        vals[pos] = d[i * stride];
        idx[pos] = i;
      }
    }
    at_l += __popcll(bl);
    at_e += __popcll(be);
  }
  SEL_STAMP();   // written
}

// force_multi: use the nine-launch form (the host entry point does when the resident form reports a timed-out barrier)
int topk_dev_impl(elfihip_ctx* ctx, const double* dD, int64_t n, int64_t stride, int64_t k, double* dvals,
                  int64_t* didx, bool force_multi) {
  ELFIHIP_REQUIRE(ctx, n >= 0 && k >= 0 && stride >= 1, "bad arguments n=%lld k=%lld stride=%lld", (long long)n,
                  (long long)k, (long long)stride);
  if (k > n) k = n;
  if (k == 0) return ELFIHIP_OK;
  ELFIHIP_REQUIRE(ctx, dD && dvals && didx, "NULL data pointer");
  hipStream_t st = ctx->stream;
  const bool multi = ctx->topk_form == 1;   // elfihip_topk_set_form: the tests and the timing script exercise both forms
  if (!multi && !force_multi) {
    // 16 keys per thread and pass at least; at most one workgroup per two CUs, so that two selections running at the
    // same time (two contexts on one GPU) are still resident together
    const int64_t max_grid = (int64_t)std::min(std::max(ctx->cu_count / 2, 1), SEL_MAX_GRID);
    int grid = (int)std::min<int64_t>((n + 16 * SEL_NT - 1) / (16 * SEL_NT), max_grid);
    if (grid < 1) grid = 1;
    int64_t per = (n + grid - 1) / grid;
    per = (per + SEL_NT - 1) / SEL_NT * SEL_NT;
    grid = (int)((n + per - 1) / per);
    ELFIHIP_CHECK_HIP(ctx, ctx->scratch.reserve(sizeof(SelWork)));
    SelWork* w = ctx->scratch.as<SelWork>();
    ELFIHIP_CHECK_HIP(ctx, hipMemsetAsync(w, 0, sizeof(SelWork), st));
    const int64_t reg_grid = (n + SEL_RU * SEL_NT - 1) / (SEL_RU * SEL_NT);
    if (reg_grid <= max_grid && ctx->topk_form != 2) {   // slices of 16 384 keys: the keys stay in LDS (form 2: the memory form, for the tests)
      static bool lds_enabled[64] = {};
      if (ctx->device < 0 || ctx->device >= 64 || !lds_enabled[ctx->device]) {
        ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sel_resident_lds_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)SEL_SLICE_LDS));
        if (ctx->device >= 0 && ctx->device < 64) lds_enabled[ctx->device] = true;
      }
      hipLaunchKernelGGL(sel_resident_lds_kernel, dim3((unsigned)reg_grid), dim3(SEL_NT), SEL_SLICE_LDS, st, dD, n, stride, k, w,
                         dvals, didx);
    } else
      hipLaunchKernelGGL(sel_persistent_kernel, dim3(grid), dim3(SEL_NT), 0, st, dD, n, stride, per, k, w, dvals, didx);
    return launch_status(ctx, "top-k selection kernel");
  }
  int nblocks = (int)std::min<int64_t>((n + 4095) / 4096, (int64_t)ctx->cu_count * 8);
  if (nblocks < 1) nblocks = 1;
  int64_t per_block = (n + nblocks - 1) / nblocks;
  per_block = (per_block + 255) / 256 * 256;
  nblocks = (int)((n + per_block - 1) / per_block);
  const size_t bytes = sizeof(SelState) + 2 * (size_t)nblocks * sizeof(unsigned int);
  ELFIHIP_CHECK_HIP(ctx, ctx->scratch.reserve(bytes));
  SelState* ds = ctx->scratch.as<SelState>();
  unsigned int* counts = reinterpret_cast<unsigned int*>(ds + 1);
  // (a kernel, not a copy of a host struct: an asynchronous copy from this function's stack may be read after it returned
  // when the stream is busy -- found by scripts/soak_round.py once the sampler stopped draining the stream before its fall-back)
  hipLaunchKernelGGL(sel_init_kernel, dim3(SEL_BINS / 256), dim3(256), 0, st, ds, (unsigned long long)k);
  // one workgroup per CU at most: each pays a device-scope fence, and a launch of this size is latency-bound anyway
  // (10^6 keys: 0.16 ms with 8 workgroups per CU, 0.10 ms with one)
  const int hist_blocks = (int)std::min<int64_t>((n + 2047) / 2048, (int64_t)ctx->cu_count);
  for (int pass = 0; pass < SEL_PASSES; ++pass)
    hipLaunchKernelGGL(sel_hist_kernel, dim3(hist_blocks), dim3(256), 0, st, dD, n, stride, pass, ds);
  hipLaunchKernelGGL(sel_count_kernel, dim3(nblocks), dim3(256), 0, st, dD, n, stride, per_block, ds, counts);
  hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(256), 0, st, counts, nblocks);
  hipLaunchKernelGGL(sel_write_kernel, dim3(nblocks), dim3(256), 0, st, dD, n, stride, per_block, ds, counts, k, dvals,
                     didx);
  return launch_status(ctx, "top-k selection kernels");
}

// The resident form's barrier time-out flag of the LATEST selection on this context lives here on the device (NULL: the
// nine-launch form is configured): non-zero = that selection's result is not valid.  Callers read it back together with
// whatever else they need (mail_post), in stream order, and synchronise before they look.
const void* topk_resident_err_dev(elfihip_ctx* ctx) {
  if (ctx->topk_form == 1 || !ctx->scratch.p) return nullptr;
  return reinterpret_cast<const char*>(ctx->scratch.p) + offsetof(SelWork, err);
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_topk_smallest_dev(elfihip_ctx* ctx, const double* dD, int64_t n, int64_t stride, int64_t k, double* dvals,
                              int64_t* didx) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return topk_dev_impl(ctx, dD, n, stride, k, dvals, didx, false);
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
  for (int attempt = 0; attempt < 2; ++attempt) {
    ELFIHIP_TRY(topk_dev_impl(ctx, dD, n, stride, k, dv, di, attempt == 1));
    unsigned int err = 0;
    if (attempt == 0)  // the resident form's barrier time-out flag (offset of SelWork::err in the scratch buffer)
      ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(&err, reinterpret_cast<const char*>(ctx->scratch.p) + offsetof(SelWork, err),
                                            sizeof err, hipMemcpyDeviceToHost, ctx->stream));
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(vals, dv, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(idx, di, (size_t)k * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (err == 0) break;  // otherwise: once more with one launch per pass
  }
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
