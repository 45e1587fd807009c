// Batched row-vs-observed distances for gfx950 (MI355X).
//
// Replaces the scipy.spatial.distance.cdist(X (n,m), Y (1,m), ...) call that
// elfi.Distance / AdaptiveDistance make once per batch
// (elfi/model/elfi_model.py:1037,1084 via elfi/model/utils.py:37-52).
//
// These kernels are HBM-bound: 8*m bytes in and 8 bytes out per distance, 2-3 flops
// per byte.  Design:
//   * row-major input (the np.column_stack layout): a workgroup streams a tile of
//     R = blockDim rows as ONE contiguous span with 16-byte loads per lane (fully
//     coalesced), drops it into LDS with an odd row pitch (m|1 doubles, so a
//     column walk by 64 lanes is bank-conflict free for ds_read_b64), and then lane r
//     sums row r left to right.  The LDS transpose is what lets every lane own a
//     whole row, so the accumulation order is exactly SciPy's (sequential over j) and
//     the result is bit-identical to cdist for the +,*,abs,max metrics.
//   * column-major input (ELFI's separate summary arrays, before column_stack): lane
//     i walks column j at row i -- coalesced by construction, no LDS needed.
//   * very wide rows (m >= 300): one wavefront per row with a shuffle tree (order
//     differs from SciPy by a few ulp; documented tolerance).
// FMA contraction is off in this file: a fused d*d+s would round differently from the
// reference's separate multiply and add.
#include "common.hpp"
#include "tile_stream.hpp"
#include "internal.hpp"

#pragma clang fp contract(off)

namespace elfihip {

constexpr int kMaxTileM = 299;  // widest row the LDS-tile kernel takes (64 rows * 301 * 8 B < 160 KiB)
constexpr int kMaxK = 64;

// Cube root for the order-3 Minkowski distance: exponent split by frexp, a single-precision seed (exp2 / log2, ~1e-6) and
// ONE Halley step (cubic: ~1e-18) -- some sixty instructions against the several hundred of pow(s, 1.0 / 3.0); within
// 1 ulp of the correctly rounded root, inside the 1e-14 the general orders are held to against SciPy's pow().
// (The device library's cbrt() measured SLOWER than pow() here: 1.25 10^6 x 64 rows 0.128 -> 0.205 ms.)
__device__ __forceinline__ double cbrt_halley(double s) {
  if (!(s > 0.0) || !(s < __builtin_huge_val())) return s;   // 0, NaN, +inf (negative sums do not occur)
  int e;
  double mant = frexp(s, &e);                 // s = mant 2^e, mant in [0.5, 1)
  int q = e / 3, r = e - 3 * q;
  if (r < 0) {
    r += 3;
    q -= 1;
  }
  mant = ldexp(mant, r);                      // in [0.5, 4)
  double y = (double)__builtin_exp2f(__builtin_log2f((float)mant) * (1.0f / 3.0f));
  const double y3 = y * y * y;
  y = y * ((y3 + 2.0 * mant) / (2.0 * y3 + mant));
  return ldexp(y, q);
}

// ---- per-metric term / finish --------------------------------------------------
template <int METRIC, bool W>
struct Op {
  __device__ static __forceinline__ double init() { return 0.0; }
  __device__ static __forceinline__ double step(double s, double x, double y, double a, double p) {
    double d = x - y;
    if constexpr (METRIC == ELFIHIP_EUCLIDEAN) {
      double t = d * d;
      if constexpr (W) t = a * t;  // SciPy: w * (d*d)
      return s + t;
    } else if constexpr (METRIC == ELFIHIP_SQEUCLIDEAN) {
      if constexpr (W) return s + (a * d) * d;  // SciPy associates the other way here
      return s + d * d;
    } else if constexpr (METRIC == ELFIHIP_CITYBLOCK) {
      double t = fabs(d);
      if constexpr (W) t = a * t;
      return s + t;
    } else if constexpr (METRIC == ELFIHIP_CHEBYSHEV) {
      double t = fabs(d);
      if constexpr (W) t = (a == 0.0) ? 0.0 : t;  // SciPy: zero-weight columns are ignored
      return t > s ? t : s;
    } else if constexpr (METRIC == ELFIHIP_MINKOWSKI) {
      // p = 3 and p = 4 (the integer orders the repository's examples use beyond 1 and 2) by multiplication:
      // within 1 ulp of pow() per term and several times cheaper; any other order through pow()
      const double ad = fabs(d);
      double t;
      if (p == 3.0)
        t = (ad * ad) * ad;
      else if (p == 4.0)
        t = (ad * ad) * (ad * ad);
      else
        t = pow(ad, p);
      if constexpr (W) t = a * t;
      return s + t;
    } else {  // ELFIHIP_SEUCLIDEAN, a = V_j
      return s + (d * d) / a;
    }
  }
  __device__ static __forceinline__ double combine(double a, double b) {
    if constexpr (METRIC == ELFIHIP_CHEBYSHEV)
      return a > b ? a : b;
    else
      return a + b;
  }
  __device__ static __forceinline__ double finish(double s, double inv_p) {
    if constexpr (METRIC == ELFIHIP_EUCLIDEAN || METRIC == ELFIHIP_SEUCLIDEAN)
      return sqrt(s);
    else if constexpr (METRIC == ELFIHIP_MINKOWSKI) {
      // the root of the integer orders 3 and 4 without pow() (SciPy takes pow(s, 1.0 / p), whose exponent is itself
      // rounded: the roots below agree with it to 1-2 ulp, inside the 1e-14 the general orders are held to); at
      // m = 2 the pow() per ROW was what the kernel spent its time on (4 10^6 rows: 0.058 ms against 0.018 for euclidean)
      if (inv_p == 1.0 / 3.0) return cbrt_halley(s);
      if (inv_p == 0.25) return sqrt(sqrt(s));
      return pow(s, inv_p);
    } else
      return s;
  }
};

// Pipelined form of dist_rows_kernel: requires vec2 and T * U >= T * m / 2 (whole tile per batch).
template <int METRIC, bool W, int U>
__global__ __launch_bounds__(256) void dist_rows_pipe_kernel(RowArgs A) {
  extern __shared__ __align__(16) double lds[];
  const int T = blockDim.x, tid = threadIdx.x, m = A.m;
  double* tile = lds;
  const int R = A.R;
  double* ys = tile + (size_t)R * A.mp;
  double* as = ys + m;
  for (int j = tid; j < m; j += T) {
    ys[j] = A.y[j];
    if constexpr (W) as[j] = A.aux[j];
  }
  const int64_t ntiles = (A.n + R - 1) / R;
  const double thr = A.F.thr ? *A.F.thr : 0.0;   // fused selection: the sampler state's current k-th best distance
  double2 v[U];
  int64_t t = blockIdx.x;
  if (t < ntiles) tile_fetch<U>(A, t * R, (int)((A.n - t * R) < R ? (A.n - t * R) : R), v);
  for (; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * R;
    const int rows = (int)((A.n - row0) < R ? (A.n - row0) : R);
    __syncthreads();  // tile free (previous readers done); ys/as visible on the first trip
    tile_commit<U>(A, tile, rows, v);
    const int64_t tn = t + gridDim.x;
    if (tn < ntiles) tile_fetch<U>(A, tn * R, (int)((A.n - tn * R) < R ? (A.n - tn * R) : R), v);
    __syncthreads();
    double dist = 0.0;
    if (tid < rows) {
      const double* row = tile + (size_t)tid * A.mp;
      double s = Op<METRIC, W>::init();
#pragma unroll 8
      for (int j = 0; j < m; ++j) s = Op<METRIC, W>::step(s, row[j], ys[j], W ? as[j] : 1.0, A.p);
      dist = Op<METRIC, W>::finish(s, A.inv_p);
      A.out[row0 + tid] = dist;
    }
    if (A.F.thr) reject_offer(A.F, tid < rows && dist < thr, dist, A.F.row_base + row0 + tid);
  }
}

// ---- LDS-DMA form of the row stream (round 5; m = 16 / 32 / 64 with 16-byte aligned rows) -----------------------------------
// The tile no longer passes through registers: `global_load_lds_dwordx4` writes 1 KiB per wave-instruction straight into
// LDS, so a wave can keep a RING of D slots (16 KiB each) in flight at no register cost and without any workgroup barrier --
// a wave only ever reads slots it issued itself, behind its own counted `s_waitcnt vmcnt` (MI355X_MICROARCH.md: nothing
// else orders a ds_read behind an LDS-DMA).  The DMA image is lane-linear (base + lane * 16), i.e. rows at pitch m with no
// padding; the bank conflicts of "lane r reads row r" are removed by an XOR swizzle of the 16-byte granules that is applied
// to the SOURCE address when the slot is filled and again when the row is read (cdna_hip_programming.md rule 21).  Lane r
// still sums row r left to right, so the results stay bit-identical to cdist.  Non-temporal loads: the rows are read once.
// Measured on 10^6 x 32 (scripts/native/glds_probe.hip, profiles/r05_glds_probe.md): 41.8-42.1 us = 6.3 TB/s (40.7 on the
// best box), against 46.5-47.2 us for the register-staged pipeline in the same binary.
template <int METRIC, bool W, int MM, int ROWS, int D>
__global__ __launch_bounds__(64) void dist_rows_dma_kernel(RowArgs A) {
  extern __shared__ __align__(16) double lds[];
  constexpr int SLOT = ROWS * MM;                // doubles
  constexpr int H = MM / 2;
  constexpr int PIECES = ROWS * H / 64;
  static_assert(ROWS <= 64 && (ROWS * H) % 64 == 0, "a slot is a whole number of 1 KiB DMA pieces, one row per lane");
  const int lane = threadIdx.x;
  double* ring = lds;
  double* ys = lds + (size_t)D * SLOT;
  double* as = ys + MM;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  if (lane < MM) {
    ys[lane] = A.y[lane];
    if constexpr (W) as[lane] = A.aux[lane];
  }
  const int64_t nslots = (A.n + ROWS - 1) / ROWS;
  const int64_t stride = gridDim.x;
  const double thr = A.F.thr ? *A.F.thr : 0.0;   // fused selection: the sampler state's current k-th best distance
  unsigned off[PIECES];   // byte offset of this lane's granule of piece i from the slot's first row (the launcher checks
#pragma unroll            // that 64 rows of pitch ldx stay below 2^31 bytes)
  for (int i = 0; i < PIECES; ++i) {
    const int G = i * 64 + lane;
    const int row = G / H, g = G % H;
    off[i] = (unsigned)(((int64_t)row * A.ldx + 2 * (g ^ dma_swizzle_key<MM>(row))) * 8);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // nothing of the prologue is outstanding: the counted
  int64_t t = blockIdx.x;                                        // waits below see the DMA pieces and the stores only
#pragma unroll
  for (int k = 0; k < D - 1; ++k) {
    const int64_t tk = t + k * stride;
    if (tk < nslots) dma_issue_slot<MM, ROWS>(A, off, tk * ROWS, lds_base + (unsigned)(k * SLOT * 8), lane);
  }
  int cur = 0;
  for (; t < nslots; t += stride) {
    // keep the ring full: the slot read in the previous trip is free (its reads were waited for at the trip's end)
    const int64_t tn = t + (int64_t)(D - 1) * stride;
    int nxt = cur + D - 1;
    if (nxt >= D) nxt -= D;
    if (tn < nslots) {
      dma_issue_slot<MM, ROWS>(A, off, tn * ROWS, lds_base + (unsigned)(nxt * SLOT * 8), lane);
      // loads return in order: once at most the D - 1 younger slots' pieces are outstanding, slot `cur` has landed (a
      // store of an earlier trip still in flight only makes this wait longer, never shorter)
      wait_vmcnt<PIECES * (D - 1)>();
    } else {
      wait_vmcnt<0>();
    }
    const int64_t row0 = t * ROWS;
    double dist = 0.0;
    const bool mine = (ROWS == 64 || lane < ROWS) && row0 + lane < A.n;
    if (ROWS == 64 || lane < ROWS) {
      const double* row = ring + (size_t)cur * SLOT + (size_t)lane * MM;
      const int key = dma_swizzle_key<MM>(lane);
      double s = Op<METRIC, W>::init();
#pragma unroll
      for (int c = 0; c < H; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(row + 2 * (c ^ key));
        const double2 yv = *reinterpret_cast<const double2*>(ys + 2 * c);
        double2 av = make_double2(1.0, 1.0);
        if constexpr (W) av = *reinterpret_cast<const double2*>(as + 2 * c);
        s = Op<METRIC, W>::step(s, v.x, yv.x, av.x, A.p);
        s = Op<METRIC, W>::step(s, v.y, yv.y, av.y, A.p);
      }
      dist = Op<METRIC, W>::finish(s, A.inv_p);
      if (mine) A.out[row0 + lane] = dist;
    }
    if (A.F.thr) reject_offer(A.F, mine && dist < thr, dist, A.F.row_base + row0 + lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are done before it is refilled
    cur = cur + 1 == D ? 0 : cur + 1;
  }
}

// Pipelined K-weight form (AdaptiveDistance.nested_distance).
template <int U>
__global__ __launch_bounds__(256) void dist_multiw_pipe_kernel(RowArgs A) {
  extern __shared__ __align__(16) double lds[];
  const int T = blockDim.x, tid = threadIdx.x, m = A.m, K = A.K;
  double* tile = lds;
  const int R = A.R;
  double* ys = tile + (size_t)R * A.mp;
  double* ws = ys + m;  // (K, m)
  for (int j = tid; j < m; j += T) ys[j] = A.y[j];
  for (int j = tid; j < K * m; j += T) ws[j] = A.aux[j];
  const int64_t ntiles = (A.n + R - 1) / R;
  const double thr = A.F.thr ? *A.F.thr : 0.0;   // fused selection, by the LAST nested distance (samplers.py:233)
  double2 v[U];
  int64_t t = blockIdx.x;
  if (t < ntiles) tile_fetch<U>(A, t * R, (int)((A.n - t * R) < R ? (A.n - t * R) : R), v);
  for (; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * R;
    const int rows = (int)((A.n - row0) < R ? (A.n - row0) : R);
    __syncthreads();
    tile_commit<U>(A, tile, rows, v);
    const int64_t tn = t + gridDim.x;
    if (tn < ntiles) tile_fetch<U>(A, tn * R, (int)((A.n - tn * R) < R ? (A.n - tn * R) : R), v);
    __syncthreads();
    double dlast = 0.0;
    if (tid < rows) {
      const double* row = tile + (size_t)tid * A.mp;
      // four weight vectors per sweep over the row: (x-y)^2 is formed once per element and feeds four
      // independent left-to-right sums (each still in cdist's order)
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
        double* o = A.out + (row0 + tid) * K + k0;
        const double r0 = sqrt(s0), r1 = sqrt(s1), r2 = sqrt(s2), r3 = sqrt(s3);
        o[0] = r0;
        if (kn > 1) o[1] = r1;
        if (kn > 2) o[2] = r2;
        if (kn > 3) o[3] = r3;
        dlast = kn > 3 ? r3 : (kn > 2 ? r2 : (kn > 1 ? r1 : r0));
      }
    }
    if (A.F.thr) reject_offer(A.F, tid < rows && dlast < thr, dlast, A.F.row_base + row0 + tid);
  }
}

// One distance per row, SciPy accumulation order.
template <int METRIC, bool W, int U>
__global__ void dist_rows_kernel(RowArgs A) {
  extern __shared__ __align__(16) double lds[];
  const int T = blockDim.x, tid = threadIdx.x, m = A.m;
  double* tile = lds;
  double* ys = tile + (size_t)T * A.mp;
  double* as = ys + m;
  for (int j = tid; j < m; j += T) {
    ys[j] = A.y[j];
    if constexpr (W) as[j] = A.aux[j];
  }
  const int64_t ntiles = (A.n + T - 1) / T;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * T;
    const int rows = (int)((A.n - row0) < T ? (A.n - row0) : T);
    __syncthreads();  // tile free (previous readers done); ys/as visible on the first trip
    load_tile<U>(A, tile, row0, rows);
    __syncthreads();
    if (tid < rows) {
      const double* row = tile + (size_t)tid * A.mp;
      double s = Op<METRIC, W>::init();
#pragma unroll 4
      for (int j = 0; j < m; ++j) s = Op<METRIC, W>::step(s, row[j], ys[j], W ? as[j] : 1.0, A.p);
      A.out[row0 + tid] = Op<METRIC, W>::finish(s, A.inv_p);
    }
  }
}

// Mahalanobis: sqrt(d' VI d); VI (m*m, row-major) is read through the scalar/L1 path.
template <int U>
__global__ void dist_rows_mahalanobis_kernel(RowArgs A) {
  extern __shared__ __align__(16) double lds[];
  const int T = blockDim.x, tid = threadIdx.x, m = A.m;
  double* tile = lds;
  double* ys = tile + (size_t)T * A.mp;
  for (int j = tid; j < m; j += T) ys[j] = A.y[j];
  const double* __restrict__ VI = A.aux;
  const int64_t ntiles = (A.n + T - 1) / T;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * T;
    const int rows = (int)((A.n - row0) < T ? (A.n - row0) : T);
    __syncthreads();
    load_tile<U>(A, tile, row0, rows);
    __syncthreads();
    if (tid < rows) {
      double* row = tile + (size_t)tid * A.mp;
      for (int j = 0; j < m; ++j) row[j] = row[j] - ys[j];  // own row only: no hazard
      double s = 0.0;
      for (int i = 0; i < m; ++i) {
        double ti = 0.0;
        const double* vi = VI + (size_t)i * m;
        for (int k = 0; k < m; ++k) ti += row[k] * vi[k];
        s += row[i] * ti;
      }
      A.out[row0 + tid] = sqrt(s);
    }
  }
}

// Mahalanobis for 8 <= m <= 64 on the matrix cores: 2 m^2 flop per row is GEMM-shaped work (delta (rows x m) times VI) and the
// lane-per-row form above reads m^2 LDS words per row (0.72 ms for 10^6 x 32, 6.3 ms for 1.25 10^6 x 64 -- 0.05 and 0.01
// of the HBM roofline).  Here a wave owns 16 rows of the tile: T = delta VI as v_mfma_f64_16x16x4 tiles (A operand: the
// rows' differences from LDS, B operand: VI from LDS, both zero padded to the MFMA shape), then s_r = sum_c T[r][c]
// delta[r][c] folded in the accumulator layout and reduced over the 16 lanes of a row.  One MFMA per row at m = 32:
// 26 us of matrix-pipe time for 10^6 rows, below the 47 us the rows take to stream.  The order of the additions
// is the matrix core's, not SciPy's BLAS calls' (whose order is unspecified too): compared at 1e-13.
constexpr int MAHA_ROWS = 64;   // rows per tile: 16 per wave, 4 waves
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void dist_rows_mahalanobis_mfma_kernel(RowArgs A) {
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, m = A.m;
  const int mk = (m + 3) & ~3, mc = (m + 15) & ~15;     // k and column extents of the padded product
  const int dp = mc | 1;                                // pitch of the difference rows (>= mc: the fold reads the padding)
  double* dl = lds;                                     // MAHA_ROWS x dp: x - y, zero beyond m
  double* vi = dl + MAHA_ROWS * dp;                     // mk x mc: VI, zero padded
  for (int e = tid; e < mk * mc; e += 256) {
    const int k = e / mc, c = e - k * mc;
    vi[e] = (k < m && c < m) ? A.aux[(size_t)k * m + c] : 0.0;
  }
  const int64_t ntiles = (A.n + MAHA_ROWS - 1) / MAHA_ROWS;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * MAHA_ROWS;
    const int rows = (int)((A.n - row0) < MAHA_ROWS ? (A.n - row0) : MAHA_ROWS);
    __syncthreads();
    for (int e = tid; e < MAHA_ROWS * dp; e += 256) {
      const int r = e / dp, c = e - r * dp;
      dl[e] = (r < rows && c < m) ? A.X[(row0 + r) * A.ldx + c] - A.y[c] : 0.0;
    }
    __syncthreads();
    const double* da = dl + (16 * w + (l & 15)) * dp + (l >> 4);   // A operand: row l & 15, k = 4 s + (l >> 4)
    double part[4] = {0.0, 0.0, 0.0, 0.0};
    for (int ct0 = 0; ct0 < mc / 16; ct0 += 4) {
      v4d acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = (v4d){0.0, 0.0, 0.0, 0.0};
      for (int s_ = 0; s_ < mk / 4; ++s_) {
        const double a = da[4 * s_];
        const double* vb = vi + (4 * s_ + (l >> 4)) * mc + 16 * ct0 + (l & 15);   // B operand: k = 4 s + (l >> 4), column l & 15
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (16 * (ct0 + j) < mc) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, vb[16 * j], acc[j], 0, 0, 0);
      }
      // accumulator element i of lane l is T[row (l >> 4) + 4 i][column 16 ct + (l & 15)]
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (16 * (ct0 + j) < mc) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            part[i] += acc[j][i] * dl[(16 * w + (l >> 4) + 4 * i) * dp + 16 * (ct0 + j) + (l & 15)];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double v = lanes16_sum(part[i]);
      const int r = 16 * w + (l >> 4) + 4 * i;
      if ((l & 15) == 0 && r < rows) A.out[row0 + r] = sqrt(v);
    }
  }
}

// K weighted euclidean distances per row (AdaptiveDistance.nested_distance); out (n,K).
template <int U>
__global__ void dist_multiw_kernel(RowArgs A) {
  extern __shared__ __align__(16) double lds[];
  const int T = blockDim.x, tid = threadIdx.x, m = A.m, K = A.K;
  double* tile = lds;
  double* ys = tile + (size_t)T * A.mp;
  double* ws = ys + m;  // (K, m)
  for (int j = tid; j < m; j += T) ys[j] = A.y[j];
  for (int j = tid; j < K * m; j += T) ws[j] = A.aux[j];
  const int64_t ntiles = (A.n + T - 1) / T;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * T;
    const int rows = (int)((A.n - row0) < T ? (A.n - row0) : T);
    __syncthreads();
    load_tile<U>(A, tile, row0, rows);
    __syncthreads();
    if (tid < rows) {
      const double* row = tile + (size_t)tid * A.mp;
      // four weight vectors per sweep over the row: (x-y)^2 is formed once per element and feeds four
      // independent left-to-right sums (each still in cdist's order)
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
        double* o = A.out + (row0 + tid) * K + k0;
        o[0] = sqrt(s0);
        if (kn > 1) o[1] = sqrt(s1);
        if (kn > 2) o[2] = sqrt(s2);
        if (kn > 3) o[3] = sqrt(s3);
      }
    }
  }
}

// Column-major input: lane i reads C[j*ldc + i]; two rows per lane when aligned.
struct ColArgs {
  const double* C;
  int64_t n, ldc;
  const double* y;
  const double* aux;
  double* out;
  double p, inv_p;
  int m;
  int vec2;
  int nt;   // non-temporal loads of the columns (each element is read once)
};

typedef double v2d_cols __attribute__((ext_vector_type(2)));

template <int METRIC, bool W>
__global__ void dist_cols_kernel(ColArgs A) {
  extern __shared__ __align__(16) double lds[];
  const int m = A.m;
  double* ys = lds;
  double* as = ys + m;
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    ys[j] = A.y[j];
    if constexpr (W) as[j] = A.aux[j];
  }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (A.vec2) {
    const int64_t npair = A.n >> 1;
    for (int64_t i2 = gid; i2 < npair; i2 += stride) {
      double s0 = Op<METRIC, W>::init(), s1 = s0;
      const double* __restrict__ c = A.C + 2 * i2;
      int j = 0;
      for (; j + 8 <= m; j += 8) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double* src = c + (int64_t)(j + u) * A.ldc;
          if (A.nt) {
            const v2d_cols t = __builtin_nontemporal_load(reinterpret_cast<const v2d_cols*>(src));
            v[u] = make_double2(t.x, t.y);
          } else {
            v[u] = *reinterpret_cast<const double2*>(src);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s0 = Op<METRIC, W>::step(s0, v[u].x, ys[j + u], W ? as[j + u] : 1.0, A.p);
          s1 = Op<METRIC, W>::step(s1, v[u].y, ys[j + u], W ? as[j + u] : 1.0, A.p);
        }
      }
      for (; j < m; ++j) {
        double2 v = *reinterpret_cast<const double2*>(c + (int64_t)j * A.ldc);
        s0 = Op<METRIC, W>::step(s0, v.x, ys[j], W ? as[j] : 1.0, A.p);
        s1 = Op<METRIC, W>::step(s1, v.y, ys[j], W ? as[j] : 1.0, A.p);
      }
      double2 o;
      o.x = Op<METRIC, W>::finish(s0, A.inv_p);
      o.y = Op<METRIC, W>::finish(s1, A.inv_p);
      *reinterpret_cast<double2*>(A.out + 2 * i2) = o;
    }
    if ((A.n & 1) && gid == 0) {  // odd tail row
      const int64_t i = A.n - 1;
      double s = Op<METRIC, W>::init();
      for (int j = 0; j < m; ++j)
        s = Op<METRIC, W>::step(s, A.C[(int64_t)j * A.ldc + i], ys[j], W ? as[j] : 1.0, A.p);
      A.out[i] = Op<METRIC, W>::finish(s, A.inv_p);
    }
  } else {
    for (int64_t i = gid; i < A.n; i += stride) {
      double s = Op<METRIC, W>::init();
      const double* __restrict__ c = A.C + i;
      int j = 0;
      for (; j + 8 <= m; j += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = c[(int64_t)(j + u) * A.ldc];
#pragma unroll
        for (int u = 0; u < 8; ++u) s = Op<METRIC, W>::step(s, v[u], ys[j + u], W ? as[j + u] : 1.0, A.p);
      }
      for (; j < m; ++j) s = Op<METRIC, W>::step(s, c[(int64_t)j * A.ldc], ys[j], W ? as[j] : 1.0, A.p);
      A.out[i] = Op<METRIC, W>::finish(s, A.inv_p);
    }
  }
}

// Very wide rows: one wavefront per row, lanes stride over the columns, butterfly combine.
template <int METRIC, bool W>
__global__ void dist_rows_wide_kernel(RowArgs A) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < A.n; r += nwaves) {
    const double* __restrict__ x = A.X + r * A.ldx;
    double s = Op<METRIC, W>::init();
    for (int j = lane; j < A.m; j += 64) s = Op<METRIC, W>::step(s, x[j], A.y[j], W ? A.aux[j] : 1.0, A.p);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s = Op<METRIC, W>::combine(s, __shfl_xor(s, off, 64));
    if (lane == 0) A.out[r] = Op<METRIC, W>::finish(s, A.inv_p);
  }
}

// ---- host-side launch logic ------------------------------------------------------
static int pick_block(int m, size_t extra_doubles, size_t* lds_bytes) {
  const int mp = m | 1;
  const int cand[3] = {256, 128, 64};
  for (int c = 0; c < 3; ++c) {
    size_t b = ((size_t)cand[c] * mp + extra_doubles) * sizeof(double);
    if (b <= 41 * 1024 || cand[c] == 64) {
      *lds_bytes = b;
      return cand[c];
    }
  }
  return 64;
}

static int grid_for(const elfihip_ctx* ctx, int64_t ntiles, size_t lds_bytes, int T) {
  int per_cu = (int)((160 * 1024) / (lds_bytes ? lds_bytes : 1));
  int by_waves = 32 / (T / 64);
  if (per_cu > by_waves) per_cu = by_waves;
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  int64_t g = (int64_t)ctx->cu_count * per_cu;
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  return (int)g;
}

template <class KernelT>
static int set_lds(elfihip_ctx* ctx, KernelT k, size_t lds) {
  if (lds > 64 * 1024)
    ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return ELFIHIP_OK;
}
// ---- narrow rows (round 6): m = 2 or 4 summaries, 16-byte aligned ----------------------------------------------------------
// A row is one or two 16-byte granules: nothing to stage.  The tile kernels above put 128 rows (2 KiB at m = 2) through LDS
// per pair of barriers with one lane in four idle and were launch- and barrier-bound at configs[0]'s own shape (4 10^6 x 2:
// minkowski 0.23, mahalanobis 0.28, the K-weight form 0.26 of HBM -- profiles/r05_kernel_table.md).  Here lane r of a
// 256-thread workgroup OWNS rows r, r + 256, ...: U rows (U 16- or 32-byte non-temporal loads) in flight per lane,
// consecutive lanes on consecutive rows (a wave-instruction covers 1 KiB of contiguous rows), the row summed left to right in
// registers exactly as the tile kernels sum it (bit-identical), one 8-byte store per row (512 contiguous bytes per wave).
template <int M>
__device__ __forceinline__ void narrow_load(const RowArgs& A, int64_t r, double (&x)[M]) {
  typedef double v2d_nt __attribute__((ext_vector_type(2)));
  const v2d_nt* src = reinterpret_cast<const v2d_nt*>(A.X + r * A.ldx);
#pragma unroll
  for (int h = 0; h < M / 2; ++h) {
    const v2d_nt t = __builtin_nontemporal_load(src + h);
    x[2 * h] = t.x;
    x[2 * h + 1] = t.y;
  }
}

template <int METRIC, bool W, int M, int U>
__global__ __launch_bounds__(256) void dist_rows_narrow_kernel(RowArgs A) {
  const int tid = threadIdx.x;
  double yv[M], av[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    yv[j] = A.y[j];
    av[j] = W ? A.aux[j] : 1.0;
  }
  const double thr = A.F.thr ? *A.F.thr : 0.0;   // fused selection: the sampler state's current k-th best distance
  const int64_t per = 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * per; base < A.n; base += (int64_t)gridDim.x * per) {
    double x[U][M];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u * 256 + tid;
      narrow_load<M>(A, r < A.n ? r : A.n - 1, x[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u * 256 + tid;
      double s = Op<METRIC, W>::init();
#pragma unroll
      for (int j = 0; j < M; ++j) s = Op<METRIC, W>::step(s, x[u][j], yv[j], av[j], A.p);
      const double dist = Op<METRIC, W>::finish(s, A.inv_p);
      if (r < A.n) A.out[r] = dist;
      if (A.F.thr) reject_offer(A.F, r < A.n && dist < thr, dist, A.F.row_base + r);
    }
  }
}

// Mahalanobis on narrow rows: VI (M x M) in registers, the sums in dist_rows_mahalanobis_kernel's order (bit-identical to it).
template <int M, int U>
__global__ __launch_bounds__(256) void dist_rows_mahalanobis_narrow_kernel(RowArgs A) {
  const int tid = threadIdx.x;
  double yv[M], vi[M][M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    yv[j] = A.y[j];
#pragma unroll
    for (int k = 0; k < M; ++k) vi[j][k] = A.aux[j * M + k];
  }
  const int64_t per = 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * per; base < A.n; base += (int64_t)gridDim.x * per) {
    double x[U][M];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u * 256 + tid;
      narrow_load<M>(A, r < A.n ? r : A.n - 1, x[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u * 256 + tid;
      double d[M];
#pragma unroll
      for (int j = 0; j < M; ++j) d[j] = x[u][j] - yv[j];
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        double ti = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) ti += d[k] * vi[i][k];
        s += d[i] * ti;
      }
      if (r < A.n) A.out[r] = sqrt(s);
    }
  }
}

// K weighted euclidean distances per narrow row (AdaptiveDistance.nested_distance): the weights of up to 8 vectors in
// registers, every sum left to right as dist_multiw_pipe_kernel forms it (bit-identical); the K results of a row are
// adjacent in `out` (a wave writes 512 K contiguous bytes).
template <int M, int U>
__global__ __launch_bounds__(256) void dist_multiw_narrow_kernel(RowArgs A) {
  constexpr int KMAX = 8;
  __shared__ __align__(16) double stage_all[4 * 64 * KMAX];   // per wave: the K results of 64 rows on their way to contiguous stores
  const int tid = threadIdx.x, K = A.K;
  double* stage = stage_all + (tid >> 6) * 64 * KMAX;
  const bool staged = (reinterpret_cast<uintptr_t>(A.out) & 15u) == 0;
  double yv[M], wv[KMAX][M];
#pragma unroll
  for (int j = 0; j < M; ++j) yv[j] = A.y[j];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int j = 0; j < M; ++j) wv[k][j] = k < K ? A.aux[k * M + j] : 0.0;
  const double thr = A.F.thr ? *A.F.thr : 0.0;   // fused selection, by the LAST nested distance (samplers.py:233)
  const int64_t per = 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * per; base < A.n; base += (int64_t)gridDim.x * per) {
    double x[U][M];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u * 256 + tid;
      narrow_load<M>(A, r < A.n ? r : A.n - 1, x[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u * 256 + tid;
      double d2[M];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const double d = x[u][j] - yv[j];
        d2[j] = d * d;
      }
      double dlast = 0.0, dk[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        dk[k] = 0.0;
        if (k < K) {
          double sk = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) sk = sk + wv[k][j] * d2[j];
          dlast = sqrt(sk);
          dk[k] = dlast;
          if (!staged && r < A.n) A.out[r * K + k] = dlast;
        }
      }
      if (staged) {
        const int64_t r0 = base + u * 256 + (tid & ~63);   // first row of this wave's 64
        const int64_t left = A.n - r0;
        if (left > 0) wave_store_rows<KMAX>(stage, A.out + r0 * K, dk, K, tid & 63, left < 64 ? (int)left : 64);
      }
      if (A.F.thr) reject_offer(A.F, r < A.n && dlast < thr, dlast, A.F.row_base + r);
    }
  }
}

static inline bool narrow_rows(const elfihip_ctx* ctx, const RowArgs& A) {
  return A.vec2 && (A.m == 2 || A.m == 4) && ctx->dist_form != 1;   // (form 1: the tile kernels of rounds 1-5, for comparison)
}
static inline unsigned narrow_grid(const elfihip_ctx* ctx, int64_t n, int U) {
  int64_t g = (n + 256 * U - 1) / (256 * U), cap = (int64_t)ctx->cu_count * 8;
  if (g > cap) g = cap;
  return (unsigned)(g < 1 ? 1 : g);
}

// 16-byte loads per thread of the pipelined row kernels.  Rows that cost a few flops per element (everything but
// general Minkowski and the K-weight sums) stream best in tiles of 32 to 64 rows (8 to 16 KiB per workgroup); the
// heavier per-row work wants all 128 lanes of the workgroup on rows (tiles of 128 rows).
static int pipe_unroll(int m, bool light) {
  if (!light) return m <= 16 ? 8 : 16;
  return m <= 32 ? 4 : (m <= 64 ? 8 : 16);
}

template <int METRIC, bool W>
static int launch_rows(elfihip_ctx* ctx, RowArgs A, bool* filtered) {
  if (A.m > kMaxTileM) {
    int64_t rows_per_block = 4;
    int64_t g = (A.n + rows_per_block - 1) / rows_per_block;
    int64_t cap = (int64_t)ctx->cu_count * 8;
    if (g > cap) g = cap;
    hipLaunchKernelGGL((dist_rows_wide_kernel<METRIC, W>), dim3((unsigned)g), dim3(256), 0, ctx->stream, A);
    return launch_status(ctx, "dist_rows_wide_kernel");
  }
  if (narrow_rows(ctx, A)) {
    constexpr int U = 4;
    if (A.m == 2)
      hipLaunchKernelGGL((dist_rows_narrow_kernel<METRIC, W, 2, U>), dim3(narrow_grid(ctx, A.n, U)), dim3(256), 0, ctx->stream, A);
    else
      hipLaunchKernelGGL((dist_rows_narrow_kernel<METRIC, W, 4, U>), dim3(narrow_grid(ctx, A.n, U)), dim3(256), 0, ctx->stream, A);
    if (filtered) *filtered = A.F.thr != nullptr;   // this form offers its candidates itself, too
    return launch_status(ctx, "dist_rows_narrow_kernel");
  }
  size_t lds;
  const int T = pick_block(A.m, 2 * (size_t)A.m, &lds);
  const int64_t ntiles = (A.n + T - 1) / T;
  const int g = grid_for(ctx, ntiles, lds, T);
  if (A.vec2 && ctx->dist_form != 1 && METRIC != ELFIHIP_MINKOWSKI && METRIC != ELFIHIP_SEUCLIDEAN &&
      (A.m == 16 || A.m == 32 || A.m == 64) && A.ldx <= (1 << 21)) {
    // LDS-DMA form: one-wave workgroups, each with a ring of two 16 KiB slots (64 rows of 32 summaries, 32 rows of 64; four
    // 8 KiB slots of 64 rows at 16 summaries), four workgroups per CU.  Measured on 10^6 x 32 / 5 10^5 x 64, plain | weighted
    // (scripts/native/glds_probe.hip, profiles/r05_glds_probe.md): ring of 2 x 4 per CU 41.9 | 42.0 and 40.7 | 41.1 us; ring
    // of 4 x 2 per CU 41.8 | 42.6 and 40.0 | 51.2 (two waves per CU cannot hide the weighted 64-column row sums); 8 KiB
    // slots 42.3-43.7; without `nt` 46-48; the register-staged pipeline 46.6-47.2.  Metrics with a division or pow() per
    // element (seuclidean, general Minkowski) keep the register-staged form below: they want all lanes of more waves.
    const int rows = A.m == 64 ? 32 : 64;
    const int D = A.m == 16 ? 4 : 2;
    const size_t ldsd = ((size_t)D * rows * A.m + 2 * (size_t)A.m) * sizeof(double);
    int64_t gd = (int64_t)ctx->cu_count * 4;
    const int64_t nslots = (A.n + rows - 1) / rows;
    if (gd > nslots) gd = nslots;
    if (gd < 1) gd = 1;
    if (A.m == 16)
      hipLaunchKernelGGL((dist_rows_dma_kernel<METRIC, W, 16, 64, 4>), dim3((unsigned)gd), dim3(64), ldsd, ctx->stream, A);
    else if (A.m == 32)
      hipLaunchKernelGGL((dist_rows_dma_kernel<METRIC, W, 32, 64, 2>), dim3((unsigned)gd), dim3(64), ldsd, ctx->stream, A);
    else
      hipLaunchKernelGGL((dist_rows_dma_kernel<METRIC, W, 64, 32, 2>), dim3((unsigned)gd), dim3(64), ldsd, ctx->stream, A);
    if (filtered) *filtered = A.F.thr != nullptr;   // this form offers its candidates itself, too
    return launch_status(ctx, "dist_rows_dma_kernel");
  }
  if (A.vec2 && A.m <= 128) {
    // pipelined form: 128 threads, U register pairs each = one whole tile of R rows.  Small tiles win: 8 KiB in
    // flight per workgroup (32 rows of 32) with 8 workgroups per CU streams 10^6 x 32 in 47.5 us, the 32 KiB tile
    // (U = 16) in 49.6 us, 4 KiB (U = 2) in 59 us; a pure read of the buffer takes 42.5 us
    // (scripts/native/stream_probe.hip).  Also measured: 256-thread tiles and non-temporal loads, no gain.
    const int Tp = 128, U = pipe_unroll(A.m, METRIC != ELFIHIP_MINKOWSKI);
    int R = 2 * Tp * U / A.m;
    if (R > Tp) R = Tp;
    A.R = R;
    const size_t ldsp = ((size_t)R * A.mp + 2 * (size_t)A.m) * sizeof(double);
    const int gp = grid_for(ctx, (A.n + R - 1) / R, ldsp, Tp);
    if (U == 4)
      hipLaunchKernelGGL((dist_rows_pipe_kernel<METRIC, W, 4>), dim3(gp), dim3(Tp), ldsp, ctx->stream, A);
    else if (U == 8)
      hipLaunchKernelGGL((dist_rows_pipe_kernel<METRIC, W, 8>), dim3(gp), dim3(Tp), ldsp, ctx->stream, A);
    else
      hipLaunchKernelGGL((dist_rows_pipe_kernel<METRIC, W, 16>), dim3(gp), dim3(Tp), ldsp, ctx->stream, A);
    if (filtered) *filtered = A.F.thr != nullptr;   // the pipelined form offers its candidates itself
    return launch_status(ctx, "dist_rows_pipe_kernel");
  }
  if (T == 64) {
    ELFIHIP_TRY(set_lds(ctx, dist_rows_kernel<METRIC, W, 16>, lds));
    hipLaunchKernelGGL((dist_rows_kernel<METRIC, W, 16>), dim3(g), dim3(T), lds, ctx->stream, A);
  } else {
    hipLaunchKernelGGL((dist_rows_kernel<METRIC, W, 8>), dim3(g), dim3(T), lds, ctx->stream, A);
  }
  return launch_status(ctx, "dist_rows_kernel");
}

template <int METRIC, bool W>
static int launch_cols(elfihip_ctx* ctx, ColArgs A) {
  const int T = 256;
  int64_t work = A.vec2 ? ((A.n + 1) >> 1) : A.n;
  int64_t g = (work + T - 1) / T;
  int64_t cap = (int64_t)ctx->cu_count * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  size_t lds = 2 * (size_t)A.m * sizeof(double);
  hipLaunchKernelGGL((dist_cols_kernel<METRIC, W>), dim3((unsigned)g), dim3(T), lds, ctx->stream, A);
  return launch_status(ctx, "dist_cols_kernel");
}

// SciPy folds minkowski p=1 / p=2 / p=inf into cityblock / euclidean / chebyshev.
static int canonical_metric(elfihip_ctx* ctx, int metric, double p, const double* aux, int* out_metric) {
  switch (metric) {
    case ELFIHIP_EUCLIDEAN:
    case ELFIHIP_SQEUCLIDEAN:
    case ELFIHIP_CITYBLOCK:
    case ELFIHIP_CHEBYSHEV:
      *out_metric = metric;
      return ELFIHIP_OK;
    case ELFIHIP_MINKOWSKI:
      if (!(p > 0.0)) return fail(ctx, ELFIHIP_ERR_ARG, "minkowski needs p > 0 (got %g)", p);
      if (p == 1.0)
        *out_metric = ELFIHIP_CITYBLOCK;
      else if (p == 2.0)
        *out_metric = ELFIHIP_EUCLIDEAN;
      else if (p > 1.7e308)
        *out_metric = ELFIHIP_CHEBYSHEV;
      else
        *out_metric = ELFIHIP_MINKOWSKI;
      return ELFIHIP_OK;
    case ELFIHIP_SEUCLIDEAN:
      if (!aux) return fail(ctx, ELFIHIP_ERR_ARG, "seuclidean needs V");
      *out_metric = metric;
      return ELFIHIP_OK;
    case ELFIHIP_MAHALANOBIS:
      if (!aux) return fail(ctx, ELFIHIP_ERR_ARG, "mahalanobis needs VI");
      *out_metric = metric;
      return ELFIHIP_OK;
    default:
      return fail(ctx, ELFIHIP_ERR_ARG, "unknown metric id %d", metric);
  }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static RowArgs make_row_args(const double* dX, int64_t n, int m, int64_t ldx, const double* dy,
                             const double* daux, double p, double* dout) {
  RowArgs A;
  A.X = dX;
  A.n = n;
  A.ldx = ldx;
  A.y = dy;
  A.aux = daux;
  A.out = dout;
  A.p = p;
  A.inv_p = p != 0.0 ? 1.0 / p : 0.0;
  A.m = m;
  A.mp = m | 1;
  A.K = 0;
  A.R = 0;
  A.nt = 0;
  A.F = RejectFilter{nullptr, nullptr, nullptr, nullptr, 0u, 0ll};
  A.vec2 = (m % 2 == 0) && (ldx % 2 == 0) && aligned16(dX);
  A.div_h = make_fastdiv((uint32_t)(A.vec2 ? m / 2 : m));
  return A;
}

// The same product with VI in REGISTERS and the rows streamed like the other distance kernels (round 3: the LDS form above
// re-read its B operand from LDS for every MFMA and filled its tile with 8-byte loads behind an integer division --
// 0.59 ms for 1.25 10^6 x 64, 0.14 of the HBM roofline and an eighth of what the matrix pipes allow).  VI does not change
// between tiles: lane l keeps VI[4 s + (l >> 4)][16 j + (l & 15)] for every k step s and column tile j (KC = 4: 64
// doubles) from the first tile to the last; the rows arrive by the software-pipelined 16-byte loads of tile_stream.hpp
// (next tile in flight while this one is multiplied) as RAW x, and x - y is formed when an operand is read.
// KC = 16-column tiles of the padded row (the k extent is padded to the same 16 KC; padded entries are exact zeros).
template <int KC>
__global__ __launch_bounds__(256, 2) void dist_rows_mahalanobis_reg_kernel(RowArgs A) {   // two workgroups per CU: <= 256 registers
  extern __shared__ __align__(16) double lds[];
  constexpr int U = 8;   // 256 threads x 8 x 16 bytes = one 64 x 64 tile
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, m = A.m, mp = A.mp;
  double* tile = lds;    // MAHA_ROWS x mp raw rows (+ 64 doubles of zeros behind: operand reads beyond the last row)
  double* ys = tile + MAHA_ROWS * mp + 64;   // y padded to 16 KC entries (in LDS: 16 KC registers fewer per lane)
  for (int e = tid; e < 64; e += 256) tile[MAHA_ROWS * mp + e] = 0.0;
  for (int e = tid; e < 16 * KC; e += 256) ys[e] = e < m ? A.y[e] : 0.0;
  double b[4 * KC][KC], yc[KC];
#pragma unroll
  for (int s_ = 0; s_ < 4 * KC; ++s_) {
    const int k = 4 * s_ + (l >> 4);
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const int c = 16 * j + (l & 15);
      b[s_][j] = (k < m && c < m) ? A.aux[(size_t)k * m + c] : 0.0;
    }
  }
#pragma unroll
  for (int j = 0; j < KC; ++j) yc[j] = (16 * j + (l & 15)) < m ? A.y[16 * j + (l & 15)] : 0.0;
  const int64_t ntiles = (A.n + MAHA_ROWS - 1) / MAHA_ROWS;
  double2 v[U];
  int64_t t = blockIdx.x;
  if (t < ntiles) tile_fetch<U>(A, t * MAHA_ROWS, (int)((A.n - t * MAHA_ROWS) < MAHA_ROWS ? (A.n - t * MAHA_ROWS) : MAHA_ROWS), v);
  for (; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * MAHA_ROWS;
    const int rows = (int)((A.n - row0) < MAHA_ROWS ? (A.n - row0) : MAHA_ROWS);
    __syncthreads();   // tile free
    tile_commit<U>(A, tile, rows, v);
    const int64_t tn = t + gridDim.x;
    if (tn < ntiles) tile_fetch<U>(A, tn * MAHA_ROWS, (int)((A.n - tn * MAHA_ROWS) < MAHA_ROWS ? (A.n - tn * MAHA_ROWS) : MAHA_ROWS), v);
    __syncthreads();
    // rows >= `rows` of a short last tile hold the previous tile's values: finite or not, they only reach their own
    // (discarded) results -- every lane's operand is its own row's
    const double* xa = tile + (16 * w + (l & 15)) * mp + (l >> 4);   // A operand: row l & 15, k = 4 s + (l >> 4)
    // one column tile at a time (ONE accumulator tile live: with all KC of them the KC = 4 instance spills beside its 64
    // registers of VI); the A operand is re-read from LDS per column tile, 16 KC ds_read_b64 against 4 KC^2 MFMAs
    double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s_ = 0; s_ < 4 * KC; ++s_) {
        const int k = 4 * s_ + (l >> 4);
        const double a = k < m ? xa[4 * s_] - ys[k] : 0.0;    // (masked: the padding must not carry a neighbour's NaN)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s_][j], acc, 0, 0, 0);
      }
      // fold with delta: accumulator element i of lane l is T[row (l >> 4) + 4 i][column 16 j + (l & 15)]
      const int c = 16 * j + (l & 15);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double dlt = c < m ? tile[(16 * w + (l >> 4) + 4 * i) * mp + c] - yc[j] : 0.0;
        part[i] += acc[i] * dlt;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double q = lanes16_sum(part[i]);
      const int r = 16 * w + (l >> 4) + 4 * i;
      if ((l & 15) == 0 && r < rows) A.out[row0 + r] = sqrt(q);
    }
  }
}

// ---- 50 <= m <= 64 (four 16-column tiles of VI): the waves SHARE VI instead of each holding all of it ----------------
// The register form above at KC = 4 keeps 64 doubles of VI per lane beside the staging registers of the next tile: the
// compiler spilled the row addresses, and every reload (`scratch_load; s_waitcnt vmcnt(0)`) waited for ALL loads in flight
// -- the eight 16-byte loads of a tile went out one memory round trip after the other, and with the addresses repaired
// the reload moved behind the prefetch and made the MFMA loop wait for it: 12 / 9.5 us per 64-row tile, waves waiting
// 68 % of their cycles, matrix pipes busy 0.28 / 0.36 (profiles/r04_mahalanobis_pmc.md).  Here wave w owns column tile w
// of VI (16 doubles per lane) and multiplies ALL 64 rows of the tile by it -- the same 64 MFMAs per wave and tile -- and
// the four waves' shares of delta^T VI delta meet in LDS: under 128 registers, no scratch, four workgroups per CU.
// Row loads: thread (row t >> 5, column pair t & 31), eight rows apart per step; a lane beyond the row's last pair / the
// tile's last row re-reads the last valid one (no predicated loads; the commit drops it).  The tile holds delta = x - y
// (subtracted at the commit: the operand reads are the MFMA operands themselves).
// KC = column tiles of VI = 1, 2 or 4 (m <= 16, <= 32, 50 .. 64): wave w owns column tile w % KC and the KC row groups
// from (w / KC) KC on -- 4 KC^2 MFMAs per wave and tile whatever KC.  LDS row pitch 16 KC + 2 doubles: lane (row l & 15,
// k-offset l >> 4) of an operand read lands in 8-byte bank (pitch row + k-offset) mod 32, and with pitch = 2 (mod 32) (or 18)
// each half-wave covers the 32 banks once (the odd pitch m | 1 of the other kernels puts row + k-offset there: four lanes
// per bank).
template <int KC>
__global__ __launch_bounds__(256, KC == 4 ? 3 : 4) void dist_rows_mahalanobis_split_kernel(RowArgs A) {   // (KC = 4 at four per CU: 3 spills, 0.282 against 0.274 ms)
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, m = A.m, h = m >> 1;
  constexpr int P = 16 * KC + 2;     // LDS row pitch
  constexpr int KS = 4 * KC;         // k-steps of four
  constexpr int CPB = 8 * KC;        // column pairs of a padded row: thread (row t / CPB, pair t % CPB), 256 / CPB rows per step
  constexpr int RPS = 256 / CPB, U = MAHA_ROWS / RPS;
  double* tile = lds;                // MAHA_ROWS x P: delta = x - y, zero from column m on (written once, below)
  double* red = tile + MAHA_ROWS * P;   // [column tile][row]: the waves' shares of a row's quadratic form
  for (int e = tid; e < MAHA_ROWS * P; e += 256) tile[e] = 0.0;
  const int jt = w % KC, g0 = (w / KC) * KC;
  const int c = 16 * jt + (l & 15);          // this lane's column of VI
  double b[KS];
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) {
    const int k = 4 * s_ + (l >> 4);
    b[s_] = (k < m && c < m) ? A.aux[(size_t)k * m + c] : 0.0;
  }
  const int cp = tid % CPB, r0 = tid / CPB;
  const int cpc = cp < h ? cp : h - 1;
  const uint32_t col = 2u * (uint32_t)cpc;
  const double y0 = A.y[2 * cpc], y1 = A.y[2 * cpc + 1];
  const int64_t ntiles = (A.n + MAHA_ROWS - 1) / MAHA_ROWS;
  double2 v[U];
  // straight-line: no branch around the loads (with the loads of a tile on one of several paths the compiler's wait
  // counters are merged at the join and the first MFMA of the loop waits for the prefetch it should overlap with)
  auto fetch = [&](int64_t tt) {
    const int64_t row0 = tt * MAHA_ROWS;
    const int rows = (int)((A.n - row0) < MAHA_ROWS ? (A.n - row0) : MAHA_ROWS);
    const char* __restrict__ Xt = reinterpret_cast<const char*>(A.X + row0 * A.ldx);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + RPS * u;
      v[u] = *reinterpret_cast<const double2*>(Xt + ((uint32_t)(r < rows ? r : rows - 1) * (uint32_t)A.ldx + col) * 8u);
    }
  };
  int64_t t = blockIdx.x;
  if (t < ntiles) fetch(t);
  for (; t < ntiles; t += gridDim.x) {
    const int64_t row0 = t * MAHA_ROWS;
    const int rows = (int)((A.n - row0) < MAHA_ROWS ? (A.n - row0) : MAHA_ROWS);
    __syncthreads();   // tile free, red read (and, the first time, the zeros in place)
    if (cp < h) {
#pragma unroll
      for (int u = 0; u < U; ++u)   // (rows beyond a short last tile get copies of its last row: finite, discarded)
        *reinterpret_cast<double2*>(tile + (r0 + RPS * u) * P + 2 * cp) = make_double2(v[u].x - y0, v[u].y - y1);
    }
    const int64_t tn = t + gridDim.x;
    fetch(tn < ntiles ? tn : t);   // (beyond the last tile: this one again, dropped)
    __syncthreads();
    // the wave's KC 16-row groups side by side: independent accumulator chains (one chain of dependent MFMAs leaves the
    // matrix pipe idle between a result and the next issue whenever the SIMD's other waves are waiting too)
    const double* xa = tile + (16 * g0 + (l & 15)) * P + (l >> 4);   // A operand of group g0 + g: row 16 (g0 + g) + (l & 15), k = 4 s + (l >> 4)
    v4d acc[KC];
#pragma unroll
    for (int g = 0; g < KC; ++g) acc[g] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
      for (int g = 0; g < KC; ++g)
        acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[16 * g * P + 4 * s_], b[s_], acc[g], 0, 0, 0);
    // fold with delta: accumulator element i of lane l is T[row 16 (g0 + g) + (l >> 4) + 4 i][column c] (delta is 0 from column m on)
#pragma unroll
    for (int g = 0; g < KC; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 16 * (g0 + g) + (l >> 4) + 4 * i;
        const double q = lanes16_sum(acc[g][i] * tile[r * P + c]);   // (DPP: common.hpp; __shfl_xor here was a third of a tile's time)
        if ((l & 15) == 0) red[jt * MAHA_ROWS + r] = q;
      }
    __syncthreads();
    if (tid < rows) {
      double q = red[tid];
      if (KC == 2) q += red[MAHA_ROWS + tid];
      if (KC == 4) q = ((q + red[MAHA_ROWS + tid]) + red[2 * MAHA_ROWS + tid]) + red[3 * MAHA_ROWS + tid];
      A.out[row0 + tid] = sqrt(q);
    }
  }
}

// F / filtered: fused selection (reject.hip).  *filtered tells the caller whether the kernel that ran offered the
// candidates itself; otherwise the caller filters dout in a separate pass.
int dist_rows_dev_impl(elfihip_ctx* ctx, int metric, const double* dX, int64_t n, int m, int64_t ldx, const double* dy,
                       const double* daux, double p, double* dout, const RejectFilter* F, bool* filtered) {
  if (filtered) *filtered = false;
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1, "bad shape n=%lld m=%d", (long long)n, m);
  ELFIHIP_REQUIRE(ctx, ldx >= m, "ldx (%lld) < m (%d)", (long long)ldx, m);
  ELFIHIP_REQUIRE(ctx, n == 0 || (dX && dy && dout), "NULL data pointer");
  int cm;
  ELFIHIP_TRY(canonical_metric(ctx, metric, p, daux, &cm));
  if (n == 0) return ELFIHIP_OK;
  RowArgs A = make_row_args(dX, n, m, ldx, dy, daux, p, dout);
  A.nt = ctx->dist_form != 1;   // rows are read once: non-temporal loads (the register-staged forms: 16-byte nt loads)
  if (F) A.F = *F;
  const bool w = daux != nullptr;
  if (cm == ELFIHIP_MAHALANOBIS) {
    ELFIHIP_REQUIRE(ctx, m <= kMaxTileM, "mahalanobis supports m <= %d", kMaxTileM);
    if (narrow_rows(ctx, A)) {
      constexpr int U = 4;
      if (m == 2)
        hipLaunchKernelGGL((dist_rows_mahalanobis_narrow_kernel<2, U>), dim3(narrow_grid(ctx, n, U)), dim3(256), 0, ctx->stream, A);
      else
        hipLaunchKernelGGL((dist_rows_mahalanobis_narrow_kernel<4, U>), dim3(narrow_grid(ctx, n, U)), dim3(256), 0, ctx->stream, A);
      return launch_status(ctx, "dist_rows_mahalanobis_narrow_kernel");
    }
    if (m >= 8 && m <= 64 && A.vec2 && ldx < (1 << 22)) {   // even m, 16-byte aligned rows: VI in registers, pipelined row loads
      const int kc = (m + 15) / 16;
      const size_t lb = ((size_t)MAHA_ROWS * A.mp + 64 + 64) * sizeof(double);
      const int g = grid_for(ctx, (n + MAHA_ROWS - 1) / MAHA_ROWS, lb, 256);
      if (kc != 3) {   // the waves share VI (kc = 3 would leave a wave without a column tile)
        const int kt = kc == 4 ? 4 : kc;
        const size_t lb4 = ((size_t)MAHA_ROWS * (16 * kt + 2) + (size_t)kt * MAHA_ROWS) * sizeof(double);
        int g4 = grid_for(ctx, (n + MAHA_ROWS - 1) / MAHA_ROWS, lb4, 256);
        const int per_cu = kt == 4 ? 3 : (kt == 2 ? 5 : 8);   // workgroups per CU by registers (measured: m = 32 0.128 ms with 3, 0.075 with 5)
        if (g4 > per_cu * ctx->cu_count) g4 = per_cu * ctx->cu_count;
        if (kt == 4)
          hipLaunchKernelGGL((dist_rows_mahalanobis_split_kernel<4>), dim3(g4), dim3(256), lb4, ctx->stream, A);
        else if (kt == 2)
          hipLaunchKernelGGL((dist_rows_mahalanobis_split_kernel<2>), dim3(g4), dim3(256), lb4, ctx->stream, A);
        else
          hipLaunchKernelGGL((dist_rows_mahalanobis_split_kernel<1>), dim3(g4), dim3(256), lb4, ctx->stream, A);
        return launch_status(ctx, "dist_rows_mahalanobis_split_kernel");
      }
      hipLaunchKernelGGL((dist_rows_mahalanobis_reg_kernel<3>), dim3(g), dim3(256), lb, ctx->stream, A);
      return launch_status(ctx, "dist_rows_mahalanobis_reg_kernel");
    }
    if (m >= 8 && m <= 64) {   // narrower rows: the padding to the 16-wide tile costs more than the lane-per-row form
      const int mk = (m + 3) & ~3, mc = (m + 15) & ~15;
      const size_t lb = ((size_t)MAHA_ROWS * (mc | 1) + (size_t)mk * mc) * sizeof(double);
      const int g = grid_for(ctx, (n + MAHA_ROWS - 1) / MAHA_ROWS, lb, 256);
      ELFIHIP_TRY(set_lds(ctx, dist_rows_mahalanobis_mfma_kernel, lb));
      hipLaunchKernelGGL(dist_rows_mahalanobis_mfma_kernel, dim3(g), dim3(256), lb, ctx->stream, A);
      return launch_status(ctx, "dist_rows_mahalanobis_mfma_kernel");
    }
    size_t lds;
    const int T = pick_block(m, (size_t)m, &lds);
    const int g = grid_for(ctx, (n + T - 1) / T, lds, T);
    ELFIHIP_TRY(set_lds(ctx, dist_rows_mahalanobis_kernel<8>, lds));
    hipLaunchKernelGGL((dist_rows_mahalanobis_kernel<8>), dim3(g), dim3(T), lds, ctx->stream, A);
    return launch_status(ctx, "dist_rows_mahalanobis_kernel");
  }
#define ELFIHIP_DISPATCH_ROWS(M)                                                  \
  case M:                                                                         \
    return w ? launch_rows<M, true>(ctx, A, filtered) : launch_rows<M, false>(ctx, A, filtered);
  switch (cm) {
    ELFIHIP_DISPATCH_ROWS(ELFIHIP_EUCLIDEAN)
    ELFIHIP_DISPATCH_ROWS(ELFIHIP_SQEUCLIDEAN)
    ELFIHIP_DISPATCH_ROWS(ELFIHIP_CITYBLOCK)
    ELFIHIP_DISPATCH_ROWS(ELFIHIP_CHEBYSHEV)
    ELFIHIP_DISPATCH_ROWS(ELFIHIP_MINKOWSKI)
    case ELFIHIP_SEUCLIDEAN:
      return launch_rows<ELFIHIP_SEUCLIDEAN, true>(ctx, A, filtered);
  }
#undef ELFIHIP_DISPATCH_ROWS
  return fail(ctx, ELFIHIP_ERR_ARG, "unhandled metric %d", cm);
}

static int dist_cols_dev_impl(elfihip_ctx* ctx, int metric, const double* dC, int64_t n, int m,
                              int64_t ldc, const double* dy, const double* daux, double p, double* dout) {
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1, "bad shape n=%lld m=%d", (long long)n, m);
  ELFIHIP_REQUIRE(ctx, ldc >= n, "ldc (%lld) < n (%lld)", (long long)ldc, (long long)n);
  ELFIHIP_REQUIRE(ctx, n == 0 || (dC && dy && dout), "NULL data pointer");
  ELFIHIP_REQUIRE(ctx, metric != ELFIHIP_MAHALANOBIS, "mahalanobis needs the row-major entry point");
  int cm;
  ELFIHIP_TRY(canonical_metric(ctx, metric, p, daux, &cm));
  if (n == 0) return ELFIHIP_OK;
  ColArgs A;
  A.C = dC;
  A.n = n;
  A.ldc = ldc;
  A.y = dy;
  A.aux = daux;
  A.out = dout;
  A.p = p;
  A.inv_p = p != 0.0 ? 1.0 / p : 0.0;
  A.m = m;
  A.vec2 = (ldc % 2 == 0) && aligned16(dC) && aligned16(dout);
  A.nt = ctx->dist_form != 1;
  const bool w = daux != nullptr;
#define ELFIHIP_DISPATCH_COLS(M)                                                  \
  case M:                                                                         \
    return w ? launch_cols<M, true>(ctx, A) : launch_cols<M, false>(ctx, A);
  switch (cm) {
    ELFIHIP_DISPATCH_COLS(ELFIHIP_EUCLIDEAN)
    ELFIHIP_DISPATCH_COLS(ELFIHIP_SQEUCLIDEAN)
    ELFIHIP_DISPATCH_COLS(ELFIHIP_CITYBLOCK)
    ELFIHIP_DISPATCH_COLS(ELFIHIP_CHEBYSHEV)
    ELFIHIP_DISPATCH_COLS(ELFIHIP_MINKOWSKI)
    case ELFIHIP_SEUCLIDEAN:
      return launch_cols<ELFIHIP_SEUCLIDEAN, true>(ctx, A);
  }
#undef ELFIHIP_DISPATCH_COLS
  return fail(ctx, ELFIHIP_ERR_ARG, "unhandled metric %d", cm);
}

int dist_multiw_dev_impl(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx, const double* dy,
                         const double* dW, int K, double* dout, const RejectFilter* F, bool* filtered) {
  if (filtered) *filtered = false;
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1, "bad shape n=%lld m=%d", (long long)n, m);
  ELFIHIP_REQUIRE(ctx, K >= 1 && K <= kMaxK, "K=%d outside [1,%d]", K, kMaxK);
  ELFIHIP_REQUIRE(ctx, ldx >= m, "ldx (%lld) < m (%d)", (long long)ldx, m);
  ELFIHIP_REQUIRE(ctx, n == 0 || (dX && dy && dW && dout), "NULL data pointer");
  if (n == 0) return ELFIHIP_OK;
  RowArgs A = make_row_args(dX, n, m, ldx, dy, dW, 2.0, dout);
  A.nt = ctx->dist_form != 1;
  A.K = K;
  if (F) A.F = *F;
  if (narrow_rows(ctx, A) && K <= 8) {
    constexpr int U = 4;
    if (m == 2)
      hipLaunchKernelGGL((dist_multiw_narrow_kernel<2, U>), dim3(narrow_grid(ctx, n, U)), dim3(256), 0, ctx->stream, A);
    else
      hipLaunchKernelGGL((dist_multiw_narrow_kernel<4, U>), dim3(narrow_grid(ctx, n, U)), dim3(256), 0, ctx->stream, A);
    if (filtered) *filtered = A.F.thr != nullptr;
    return launch_status(ctx, "dist_multiw_narrow_kernel");
  }
  size_t lds;
  int T = pick_block(m, (size_t)m + (size_t)K * m, &lds);
  ELFIHIP_REQUIRE(ctx, lds <= 160 * 1024, "m=%d with K=%d weight vectors does not fit LDS", m, K);
  const int g = grid_for(ctx, (n + T - 1) / T, lds, T);
  if (A.vec2 && m <= 128) {
    const int Tp = 128, U = 16;
    int R = 2 * Tp * U / m;
    if (R > Tp) R = Tp;
    A.R = R;
    const size_t ldsp = ((size_t)R * A.mp + (size_t)m + (size_t)K * m) * sizeof(double);
    if (ldsp <= 64 * 1024) {
      const int gp = grid_for(ctx, (n + R - 1) / R, ldsp, Tp);
      hipLaunchKernelGGL((dist_multiw_pipe_kernel<16>), dim3(gp), dim3(Tp), ldsp, ctx->stream, A);
      if (filtered) *filtered = A.F.thr != nullptr;
      return launch_status(ctx, "dist_multiw_pipe_kernel");
    }
  }
  ELFIHIP_TRY(set_lds(ctx, dist_multiw_kernel<8>, lds));
  hipLaunchKernelGGL((dist_multiw_kernel<8>), dim3(g), dim3(T), lds, ctx->stream, A);
  return launch_status(ctx, "dist_multiw_kernel");
}

static size_t aux_len(int metric, int m) {
  return metric == ELFIHIP_MAHALANOBIS ? (size_t)m * m : (size_t)m;
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_dist_rows_dev(elfihip_ctx* ctx, int metric, const double* dX, int64_t n, int m, int64_t ldx,
                          const double* dy, const double* daux, double p, double* dout) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return dist_rows_dev_impl(ctx, metric, dX, n, m, ldx, dy, daux, p, dout, nullptr, nullptr);
}

int elfihip_dist_cols_dev(elfihip_ctx* ctx, int metric, const double* dC, int64_t n, int m, int64_t ldc,
                          const double* dy, const double* daux, double p, double* dout) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return dist_cols_dev_impl(ctx, metric, dC, n, m, ldc, dy, daux, p, dout);
}

int elfihip_dist_multiw_dev(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx,
                            const double* dy, const double* dW, int K, double* dout) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  return dist_multiw_dev_impl(ctx, dX, n, m, ldx, dy, dW, K, dout, nullptr, nullptr);
}

// ---- host-pointer entry points: stage, launch, copy back, synchronise ---------------
static int stage_params(elfihip_ctx* ctx, const double* y, const double* aux, int m, size_t naux,
                        double** dy, double** daux) {
  ELFIHIP_CHECK_HIP(ctx, ctx->par.reserve(((size_t)m + naux) * sizeof(double)));
  *dy = ctx->par.as<double>();
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(*dy, y, (size_t)m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  *daux = nullptr;
  if (aux) {
    *daux = *dy + m;
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(*daux, aux, naux * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  return ELFIHIP_OK;
}

static int stage_rows(elfihip_ctx* ctx, const double* X, int64_t n, int m, int64_t ldx, double** dX) {
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((size_t)n * m * sizeof(double)));
  *dX = ctx->in.as<double>();
  if (n == 0) return ELFIHIP_OK;
  if (ldx == m)
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(*dX, X, (size_t)n * m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  else
    ELFIHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(*dX, (size_t)m * sizeof(double), X, (size_t)ldx * sizeof(double),
                                            (size_t)m * sizeof(double), (size_t)n, hipMemcpyHostToDevice,
                                            ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_dist_rows(elfihip_ctx* ctx, int metric, const double* X, int64_t n, int m, int64_t ldx,
                      const double* y, const double* aux, double p, double* out) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m,
                  (long long)ldx);
  ELFIHIP_REQUIRE(ctx, y && (n == 0 || (X && out)), "NULL data pointer");
  DeviceGuard g(ctx->device);
  double *dX, *dy, *daux;
  ELFIHIP_TRY(stage_params(ctx, y, aux, m, aux ? aux_len(metric, m) : 0, &dy, &daux));
  ELFIHIP_TRY(stage_rows(ctx, X, n, m, ldx, &dX));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)(n ? n : 1) * sizeof(double)));
  ELFIHIP_TRY(dist_rows_dev_impl(ctx, metric, dX, n, m, m, dy, daux, p, ctx->out.as<double>(), nullptr, nullptr));
  ELFIHIP_TRY(keep_distances(ctx, ctx->out.as<double>(), n, 1));
  if (n)
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

// Host-pointer form of the fused distance + selection step (reject.hip): one ABC batch in, its distances out, the
// sampler state updated on the way.
int elfihip_reject_push_rows(elfihip_reject* h, int metric, const double* X, int64_t n, int m, int64_t ldx,
                             const double* y, const double* aux, double p, double* out, int64_t row_base) {
  if (!h) return fail(nullptr, ELFIHIP_ERR_ARG, "state is NULL");
  elfihip_ctx* ctx = reject_ctx(h);
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m,
                  (long long)ldx);
  ELFIHIP_REQUIRE(ctx, y && (n == 0 || X), "NULL data pointer");   // out == NULL: the distances stay on the device
  DeviceGuard g(ctx->device);
  double *dX, *dy, *daux;
  ELFIHIP_TRY(stage_params(ctx, y, aux, m, aux ? aux_len(metric, m) : 0, &dy, &daux));
  ELFIHIP_TRY(stage_rows(ctx, X, n, m, ldx, &dX));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)(n ? n : 1) * sizeof(double)));
  ELFIHIP_TRY(reject_push_rows_impl(h, metric, dX, n, m, m, dy, daux, p, ctx->out.as<double>(), row_base));
  if (n && out)
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_dist_cols(elfihip_ctx* ctx, int metric, const double* const* cols, int m, int64_t n,
                      const double* y, const double* aux, double p, double* out) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1, "bad shape n=%lld m=%d", (long long)n, m);
  ELFIHIP_REQUIRE(ctx, y && cols && (n == 0 || out), "NULL data pointer");
  ELFIHIP_REQUIRE(ctx, metric != ELFIHIP_MAHALANOBIS, "mahalanobis needs the row-major entry point");
  DeviceGuard g(ctx->device);
  double *dy, *daux;
  ELFIHIP_TRY(stage_params(ctx, y, aux, m, aux ? (size_t)m : 0, &dy, &daux));
  const int64_t ldc = (n + 1) & ~(int64_t)1;  // even pitch keeps every column 16-byte aligned
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((size_t)(ldc ? ldc : 2) * m * sizeof(double)));
  double* dC = ctx->in.as<double>();
  for (int j = 0; j < m && n; ++j) {
    ELFIHIP_REQUIRE(ctx, cols[j] != nullptr, "column %d is NULL", j);
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dC + (size_t)j * ldc, cols[j], (size_t)n * sizeof(double),
                                          hipMemcpyHostToDevice, ctx->stream));
  }
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)(ldc ? ldc : 2) * sizeof(double)));
  ELFIHIP_TRY(dist_cols_dev_impl(ctx, metric, dC, n, m, ldc ? ldc : 2, dy, daux, p, ctx->out.as<double>()));
  ELFIHIP_TRY(keep_distances(ctx, ctx->out.as<double>(), n, 1));
  if (n)
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_dist_multiw(elfihip_ctx* ctx, const double* X, int64_t n, int m, int64_t ldx, const double* y,
                        const double* W, int K, double* out) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m,
                  (long long)ldx);
  ELFIHIP_REQUIRE(ctx, K >= 1 && K <= kMaxK, "K=%d outside [1,%d]", K, kMaxK);
  ELFIHIP_REQUIRE(ctx, y && W && (n == 0 || (X && out)), "NULL data pointer");
  DeviceGuard g(ctx->device);
  double *dX, *dy, *dW;
  ELFIHIP_TRY(stage_params(ctx, y, W, m, (size_t)K * m, &dy, &dW));
  ELFIHIP_TRY(stage_rows(ctx, X, n, m, ldx, &dX));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)(n ? n : 1) * K * sizeof(double)));
  ELFIHIP_TRY(dist_multiw_dev_impl(ctx, dX, n, m, m, dy, dW, K, ctx->out.as<double>(), nullptr, nullptr));
  ELFIHIP_TRY(keep_distances(ctx, ctx->out.as<double>(), n, K));
  if (n)
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->out.p, (size_t)n * K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

}  // extern "C"
