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

constexpr int ADA_T = 256;      // threads per workgroup: four waves
constexpr int ADA_U = 8;        // 16-byte loads per thread and tile (generic streaming path)
constexpr int ADA_RMAX = 128;   // rows per tile at most (two waves of row owners)
#ifndef ADA_OCC
#define ADA_OCC 4                // waves per SIMD the register allocation aims at (<= 128 registers: four workgroups per CU)
#endif

struct AdaptArgs {
  RowArgs A;                         // rows, observed y, W (K, m) in aux, out (n, K) or NULL, the selection filter
  const double* acc;                 // K acceptance thresholds (NULL: every row is acceptable)
  unsigned long long* acc_count;     // rows accepted (with acc)
  double* partial;                   // (gridDim.x, 1 + 2m) workgroup statistics (NULL: none)
};

struct ColStat {
  double n, mean, M2;
};

#ifdef ELFIHIP_ADA_STAMP   // developer probe (scripts/native/ada_probe.hip): shader-clock stamps per wave, tile and phase
__device__ long long g_ada_stamp[8 * 4 * 16 * 8];
#define ADA_STAMP(slot)                                                                                   \
  do {                                                                                                    \
    if (blockIdx.x < 8 && iter >= 64 && iter < 80 && (threadIdx.x & 63) == 0)                             \
      g_ada_stamp[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + iter - 64) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define ADA_STAMP(slot) \
  do {                  \
  } while (0)
#endif

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

// Uniform operands of the row sums (the weights w_kj, the observed y_j: the same for every lane) come through the
// constant address space, i.e. scalar loads into SGPRs that the f64 VALU instructions take directly, in blocks of B
// elements with the next block requested before the current one is summed (scalar loads return out of order: a wait is a
// wait for all of them, so a block is requested right after the wait that delivered its predecessor; two blocks of
// B = 16 doubles are 64 of a wave's 102 SGPRs).  Measured per tile of 64 x 64 and ONE column
// (scripts/native/ada_probe.hip, cycles): broadcast ds_reads of y_j / w_kj as distance.hip's kernels do 5700 (the LDS
// round trip of every unrolled group of four); scalar loads, four elements per wait 5400 (a ~300-cycle round trip per
// four elements); operands spread over the lanes + v_readlane 8400 (the SGPR hand-over stalls the VALU).
typedef const double __attribute__((address_space(4))) * cdptr;
__device__ __forceinline__ cdptr as_constant(const double* p) { return (cdptr) reinterpret_cast<uintptr_t>(p); }

// sqrt(sum_j w_j d_j^2), left to right (cdist's order; multiply and add round separately).  PRE: `row` holds the
// differences d_j = x_j - y_j (formed once, when the tile was written); otherwise x_j, and y comes with the weights.
// The row's elements are read from LDS four at a time, one group ahead of their use (the read past the row's last
// group lands in LDS the workgroup owns and is never used).
template <int B, bool PRE>
__device__ __forceinline__ double row_distance(const double* row, cdptr ys, cdptr w, int m) {
  double s = 0.0;
  double wn[B], yn[PRE ? 1 : B];
#pragma unroll
  for (int i = 0; i < B; ++i) {
    wn[i] = w[i];
    if constexpr (!PRE) yn[i] = ys[i];
  }
  double xn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xn[i] = row[i];
  int j0 = 0;
  for (; j0 + B <= m; j0 += B) {
    double wc[B], yc[PRE ? 1 : B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      wc[i] = wn[i];
      if constexpr (!PRE) yc[i] = yn[i];
    }
    const int jn = j0 + 2 * B <= m ? j0 + B : j0;   // (the last block re-requests itself: nothing is read beyond y / W)
#pragma unroll
    for (int i = 0; i < B; ++i) {
      wn[i] = w[jn + i];
      if constexpr (!PRE) yn[i] = ys[jn + i];
    }
#pragma unroll
    for (int i0 = 0; i0 < B; i0 += 4) {
      double x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = xn[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) xn[i] = row[j0 + i0 + 4 + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double d = x[i];
        if constexpr (!PRE) d = d - yc[i0 + i];
        const double d2 = d * d;
        s = s + wc[i0 + i] * d2;
      }
    }
  }
  for (int j = j0; j < m; ++j) {   // m not a multiple of the block
    double d = row[j];
    if constexpr (!PRE) d = d - ys[j];
    const double d2 = d * d;
    s = s + w[j] * d2;
  }
  return sqrt(s);
}

// Column statistics of CH rows (p[0], p[step], ...) held in registers: sum -> mean (CH is a power of two: the scaling is
// exact) -> sum of squared deviations about it; fixed association.
template <int CH>
__device__ __forceinline__ void chunk_stats(const double (&x)[CH], double& mean, double& q) {
  if constexpr (CH >= 4) {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < CH; i += 4) {
      a[0] += x[i];
      a[1] += x[i + 1];
      a[2] += x[i + 2];
      a[3] += x[i + 3];
    }
    mean = ((a[0] + a[1]) + (a[2] + a[3])) * (1.0 / CH);
    double b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < CH; i += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double e = x[i + u] - mean;
        b[u] += e * e;
      }
    }
    q = (b[0] + b[1]) + (b[2] + b[3]);
  } else {
    double a = 0.0;
#pragma unroll
    for (int i = 0; i < CH; ++i) a += x[i];
    mean = a * (1.0 / CH);
    q = 0.0;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const double e = x[i] - mean;
      q += e * e;
    }
  }
}

// NU > 0: m / 2 divides the workgroup size, so load u of a thread is the same element pair QS rows further down and a
// tile is exactly NU loads per thread -- source and LDS offsets are one add per load, and the loads of a full tile carry
// no predicate (one straight run of NU global_load_dwordx4).  A thread then holds NU rows of the SAME two columns of every
// tile in registers when it writes them to LDS: it subtracts the observed pair once (the tile holds d = x - y) and takes
// the two columns' statistics from those registers -- two-pass sums of the NU values, Chan's update of its running
// triples -- so the statistics cost no LDS read at all.  NU == 0: any even m through the generic tile_fetch /
// tile_commit of tile_stream.hpp (an integer division per load and tile, ADA_U predicated loads); the tile holds x, the
// statistics make their two passes over LDS.
//
// Work split (measured with scripts/native/ada_probe.hip; round 4): with the whole row's K distances in one lane the
// wave that owned the rows was the critical path of a two-wave workgroup (K = 3: 1.35 ms per 10^7 x 64 against 1.03 at
// K = 1), and its time was operand latency, not arithmetic.  Four waves per workgroup: lane r of a wave owns row r (or
// r + 64: tiles of 128 rows have two row blocks), the K columns are dealt out over the 4 (2) waves of a row block, so
// every sum keeps cdist's order and a wave sweeps its row once per column it holds.  The verdicts of the waves on their
// columns (acceptance) meet in LDS.
template <int NU>
__global__ __launch_bounds__(ADA_T, ADA_OCC) void adaptive_pass_kernel(AdaptArgs P) {
  extern __shared__ __align__(16) double lds[];
  constexpr int U = NU > 0 ? NU : ADA_U;
  constexpr bool POW2 = NU > 0;
  const RowArgs& A = P.A;
  const int T = ADA_T, tid = threadIdx.x, m = A.m, K = A.K, R = A.R, mp = A.mp;
  double* tile = lds;
  const int tile_doubles = R * mp > 6 * T ? R * mp : 6 * T;   // the tile doubles as the end-of-kernel reduction space
  double* accs = tile + tile_doubles + 8;   // (K) acceptance thresholds (8 doubles of slack behind the tile: row_distance reads ahead)
  int* flag = reinterpret_cast<int*>(accs + K);   // (4, 128) acceptance of a column group's columns, row by row
  const cdptr ys = as_constant(A.y), ws = as_constant(A.aux);   // (m), (K, m): uniform operands through the scalar cache
  if (P.acc)
    for (int j = tid; j < K; j += T) accs[j] = P.acc[j];
  const int64_t ntiles = (A.n + R - 1) / R;
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  const double thr = A.F.thr ? *A.F.thr : inf;
  // (wave-uniform values the compiler must see as uniform: they select the wave's columns, whose weights are scalar loads)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int RB = R > 64 ? 2 : 1;           // row blocks of 64
  const int CG = 4 / RB;                   // column groups: waves per row block
  const int rb = wave % RB, cg = wave / RB;
  const int rl = rb * 64 + (tid & 63);     // the row this lane owns
  const bool owner = cg == (K - 1) % CG;   // this wave holds the LAST column of its rows: it counts and offers
  unsigned long long nacc = 0;
  // strided streaming: element pair jj0 (columns 2 jj0, 2 jj0 + 1) of row q0 + u QS
  const int h = m >> 1;
  const int q0 = POW2 ? tid / h : 0, jj0 = POW2 ? tid - q0 * h : 0, QS = POW2 ? T / h : 1;
  const int64_t soff = (int64_t)q0 * A.ldx + 2 * jj0, sstep = (int64_t)QS * A.ldx;
  const int doff = q0 * mp + 2 * jj0, dstep = QS * mp;
  double2 yp = make_double2(0.0, 0.0);     // the observed pair of this thread's columns
  if constexpr (POW2) yp = make_double2(A.y[2 * jj0], A.y[2 * jj0 + 1]);
  // column statistics.  POW2: this thread's two columns (st0, st1), from registers.  Generic: thread (g, c) = column c,
  // rows g, g + G, ... of every tile, from LDS (st0).
  const int G = T / m;          // row groups (the launcher guarantees m <= T / 2)
  const int g = tid / m, c = tid - g * m;
  const bool stats = P.partial != nullptr;
  ColStat st0 = {0.0, 0.0, 0.0}, st1 = {0.0, 0.0, 0.0};
  double2 v[U];
  auto fetch = [&](int64_t row0, int rows) {
    if constexpr (POW2) {
      const double* __restrict__ src = A.X + row0 * A.ldx + soff;
      if (rows == R) {
        if (A.nt) {
          typedef double v2d_nt __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const v2d_nt tv = __builtin_nontemporal_load(reinterpret_cast<const v2d_nt*>(src + u * sstep));
            v[u] = make_double2(tv.x, tv.y);
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const double2*>(src + u * sstep);
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          v[u] = make_double2(0.0, 0.0);
          if (q0 + u * QS < rows) v[u] = *reinterpret_cast<const double2*>(src + u * sstep);
        }
      }
    } else {
      tile_fetch<U>(A, row0, rows, v);
    }
  };
  // tile <- the registers (POW2: as differences from the observed pair), statistics of the registers on the way
  auto commit = [&](int rows) {
    if constexpr (POW2) {
      if (rows == R) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          double* dst = tile + doff + u * dstep;
          dst[0] = v[u].x - yp.x;
          dst[1] = v[u].y - yp.y;
        }
        if (stats) {
          double a[U], b[U], ma, qa, mb, qb;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            a[u] = v[u].x;
            b[u] = v[u].y;
          }
          chunk_stats<U>(a, ma, qa);
          chunk_stats<U>(b, mb, qb);
          chan_merge(st0, (double)U, ma, qa);
          chan_merge(st1, (double)U, mb, qb);
        }
      } else {
        int cnt = 0;
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (q0 + u * QS < rows) {
            double* dst = tile + doff + u * dstep;
            dst[0] = v[u].x - yp.x;
            dst[1] = v[u].y - yp.y;
            sa += v[u].x;
            sb += v[u].y;
            ++cnt;
          }
        if (stats && cnt > 0) {
          const double ma = sa / (double)cnt, mb = sb / (double)cnt;
          double qa = 0.0, qb = 0.0;
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (q0 + u * QS < rows) {
              const double ea = v[u].x - ma, eb = v[u].y - mb;
              qa += ea * ea;
              qb += eb * eb;
            }
          chan_merge(st0, (double)cnt, ma, qa);
          chan_merge(st1, (double)cnt, mb, qb);
        }
      }
    } else {
      tile_commit<U>(A, tile, rows, v);
    }
  };
  int64_t t = blockIdx.x;
  if (t < ntiles) fetch(t * R, (int)((A.n - t * R) < R ? (A.n - t * R) : R));
#ifdef ELFIHIP_ADA_STAMP
  int iter = -1;
#endif
  for (; t < ntiles; t += gridDim.x) {
#ifdef ELFIHIP_ADA_STAMP
    ++iter;
#endif
    const int64_t row0 = t * R;
    const int rows = (int)((A.n - row0) < R ? (A.n - row0) : R);
    ADA_STAMP(0);
    __syncthreads();   // tile free (every reader of the previous one is done); accs visible on the first trip
    ADA_STAMP(1);
    commit(rows);
    ADA_STAMP(2);
    const int64_t tn = t + gridDim.x;
    if (tn < ntiles) fetch(tn * R, (int)((A.n - tn * R) < R ? (A.n - tn * R) : R));
    ADA_STAMP(3);
    __syncthreads();
    ADA_STAMP(4);
    // ---- nested distances of this lane's row under its wave's columns cg, cg + CG, ...
    double dlast = 0.0;
    bool ok = rl < rows;
    if (rl < rows) {
      const double* row = tile + (size_t)rl * mp;
      for (int k = cg; k < K; k += CG) {
        double r;
        if constexpr (POW2)
          r = row_distance<16, true>(row, ys, ws + (size_t)k * m, m);
        else
          r = row_distance<8, false>(row, ys, ws + (size_t)k * m, m);
        if (A.out) A.out[(row0 + rl) * K + k] = r;
        if (P.acc) ok = ok && r <= accs[k];   // samplers.py:219-225: every nested column against its threshold (a NaN distance is not accepted)
        dlast = r;   // (the owner's last column is K - 1)
      }
    }
    if (P.acc && K > 1) {   // the other column groups' verdicts on this row (uniform branch: all threads arrive)
      if (!owner) flag[cg * ADA_RMAX + rl] = ok ? 1 : 0;
      __syncthreads();
      if (owner) {
        const int ng = K < CG ? K : CG;   // groups that hold columns
        for (int q = 0; q < ng; ++q)
          if (q != cg) ok = ok && flag[q * ADA_RMAX + rl] != 0;
      }
    }
    if (owner) {
      if (P.acc) nacc += (unsigned long long)__popcll(__ballot(ok));   // wave-uniform
      if (A.F.thr) reject_offer(A.F, ok && dlast < thr, dlast, A.F.row_base + row0 + rl);
    }
    ADA_STAMP(5);
    if constexpr (!POW2) {
      // ---- column statistics of the tile: two passes over this thread's rows in LDS, then Chan's update
      if (stats && g < G && g < rows) {
        const double* col = tile + c;
        double sum = 0.0;
        int cnt = 0;
#pragma unroll 8
        for (int r = g; r < rows; r += G) {
          sum += col[(size_t)r * mp];
          ++cnt;
        }
        const double mt = sum / (double)cnt;
        double q = 0.0;
#pragma unroll 8
        for (int r = g; r < rows; r += G) {
          const double e = col[(size_t)r * mp] - mt;
          q += e * e;
        }
        chan_merge(st0, (double)cnt, mt, q);
      }
    }
    ADA_STAMP(6);
  }
  if (stats) {
    __syncthreads();   // the tile is free: it carries the threads' triples, column by column
    double* red = tile;
    // entry (column col, slot sl): red[3 (sl * m + col) ..]; POW2: slot = q0 (QS per column), generic: slot = g (G per column)
    const int nslot = POW2 ? QS : G;
    if constexpr (POW2) {
      double* e0 = red + 3 * ((size_t)q0 * m + 2 * jj0);
      e0[0] = st0.n, e0[1] = st0.mean, e0[2] = st0.M2;
      e0[3] = st1.n, e0[4] = st1.mean, e0[5] = st1.M2;
    } else if (g < G) {
      double* e0 = red + 3 * ((size_t)g * m + c);
      e0[0] = st0.n, e0[1] = st0.mean, e0[2] = st0.M2;
    }
    __syncthreads();
    if (tid < m) {
      ColStat a = {0.0, 0.0, 0.0};
      for (int sl = 0; sl < nslot; ++sl) {
        const double* e = red + 3 * ((size_t)sl * m + tid);
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

static bool ada_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- narrow rows (round 6): m = 2 or 4 summaries -------------------------------------------------------------------------------
// A row is one or two 16-byte granules: lane r of a 256-thread workgroup OWNS rows r, r + 256, ... (U rows = U 16- or 32-byte
// non-temporal loads in flight per lane, a wave-instruction covers 1 KiB of contiguous rows; no LDS tile, no barrier per tile).
// Rounds 4-5 sent m = 2 through the separate K-weight and two-pass statistics kernels (three reads: 0.126 ms for 4 10^6 x 2,
// K = 3 = 0.16 of HBM) because the tile kernel above streams such rows at a quarter of their rate.
//   * distances: every sum left to right over the row's M elements with the weights in registers -- the expressions of
//     dist_multiw_narrow_kernel (distance.hip): bit-identical to it and to cdist;
//   * column statistics: the lane's U rows of an iteration -> two-pass (mean, M2) per column in registers (chunk_stats) ->
//     Chan's update of the lane's running triples; at the end the 256 lanes' triples are merged by a fixed tree in LDS and
//     one (1 + 2m) partial per workgroup leaves for adaptive_finish_kernel: no atomics, bit-reproducible for a launch shape;
//   * selection: per-column acceptance and the sampler state's k-th distance on the distances while they are in registers.
constexpr int ADA_NARROW_KMAX = 8;
constexpr int ADA_NARROW_U = 8;

template <int M, int U>
__global__ __launch_bounds__(256) void adaptive_narrow_kernel(AdaptArgs P) {
  __shared__ double rn[256], rm[256 * M], rq[256 * M];
  __shared__ __align__(16) double stage_all[4 * 64 * ADA_NARROW_KMAX];   // per wave: the K results of 64 rows -> contiguous stores
  const RowArgs& A = P.A;
  const int tid = threadIdx.x, K = A.K;
  double* stage = stage_all + (tid >> 6) * 64 * ADA_NARROW_KMAX;
  const bool staged = A.out && (reinterpret_cast<uintptr_t>(A.out) & 15u) == 0;
  double yv[M], wv[ADA_NARROW_KMAX][M], accs[ADA_NARROW_KMAX];
#pragma unroll
  for (int j = 0; j < M; ++j) yv[j] = A.y[j];
#pragma unroll
  for (int k = 0; k < ADA_NARROW_KMAX; ++k) {
    accs[k] = (P.acc && k < K) ? P.acc[k] : 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) wv[k][j] = k < K ? A.aux[k * M + j] : 0.0;
  }
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  const double thr = A.F.thr ? *A.F.thr : inf;
  const bool stats = P.partial != nullptr;
  ColStat st[M];
#pragma unroll
  for (int j = 0; j < M; ++j) st[j] = {0.0, 0.0, 0.0};
  unsigned long long nacc = 0;
  typedef double v2d_nt __attribute__((ext_vector_type(2)));
  const int64_t per = 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * per; base < A.n; base += (int64_t)gridDim.x * per) {
    double x[U][M];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u * 256 + tid;
      const v2d_nt* src = reinterpret_cast<const v2d_nt*>(A.X + (r < A.n ? r : A.n - 1) * A.ldx);
#pragma unroll
      for (int h = 0; h < M / 2; ++h) {
        const v2d_nt t = __builtin_nontemporal_load(src + h);
        x[u][2 * h] = t.x;
        x[u][2 * h + 1] = t.y;
      }
    }
    if (stats) {
      if (base + per <= A.n) {   // every lane holds U rows (wave-uniform)
#pragma unroll
        for (int j = 0; j < M; ++j) {
          double col[U], mean, q;
#pragma unroll
          for (int u = 0; u < U; ++u) col[u] = x[u][j];
          chunk_stats<U>(col, mean, q);
          chan_merge(st[j], (double)U, mean, q);
        }
      } else {
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) cnt += (base + u * 256 + tid < A.n) ? 1 : 0;
        if (cnt > 0) {
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double sum = 0.0;
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (base + u * 256 + tid < A.n) sum += x[u][j];
            const double mean = sum / (double)cnt;
            double q = 0.0;
#pragma unroll
            for (int u = 0; u < U; ++u)
              if (base + u * 256 + tid < A.n) {
                const double e = x[u][j] - mean;
                q += e * e;
              }
            chan_merge(st[j], (double)cnt, mean, q);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = base + u * 256 + tid;
      const bool live = r < A.n;
      double d2[M];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const double d = x[u][j] - yv[j];
        d2[j] = d * d;
      }
      double dlast = 0.0, dk[ADA_NARROW_KMAX];
      bool ok = live;
#pragma unroll
      for (int k = 0; k < ADA_NARROW_KMAX; ++k) {
        dk[k] = 0.0;
        if (k < K) {
          double sk = 0.0;
#pragma unroll
          for (int j = 0; j < M; ++j) sk = sk + wv[k][j] * d2[j];
          dlast = sqrt(sk);
          dk[k] = dlast;
          if (A.out && !staged && live) A.out[r * K + k] = dlast;
          if (P.acc) ok = ok && dlast <= accs[k];   // samplers.py:219-225 (a NaN distance is not accepted)
        }
      }
      if (staged) {
        const int64_t r0 = base + u * 256 + (tid & ~63);   // first row of this wave's 64
        const int64_t left = A.n - r0;
        if (left > 0) wave_store_rows<ADA_NARROW_KMAX>(stage, A.out + r0 * K, dk, K, tid & 63, left < 64 ? (int)left : 64);
      }
      if (P.acc) nacc += (unsigned long long)__popcll(__ballot(ok));   // wave-uniform
      if (A.F.thr) reject_offer(A.F, ok && dlast < thr, dlast, A.F.row_base + r);
    }
  }
  if (stats) {
    // the 256 lanes' triples, column by column, by a fixed tree (the same merges in the same order at every launch)
    rn[tid] = st[0].n;   // (the count is the same for every column of a lane)
#pragma unroll
    for (int j = 0; j < M; ++j) {
      rm[j * 256 + tid] = st[j].mean;
      rq[j * 256 + tid] = st[j].M2;
    }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      double nb = 0.0;
      if (tid < s) nb = rn[tid + s];
      if (tid < s) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
          ColStat a = {rn[tid], rm[j * 256 + tid], rq[j * 256 + tid]};
          chan_merge(a, nb, rm[j * 256 + tid + s], rq[j * 256 + tid + s]);
          rm[j * 256 + tid] = a.mean;
          rq[j * 256 + tid] = a.M2;
          if (j == M - 1) st[0].n = a.n;
        }
      }
      __syncthreads();
      if (tid < s) rn[tid] = st[0].n;
      __syncthreads();
    }
    if (tid < M) {
      double* o = P.partial + (size_t)blockIdx.x * (1 + 2 * M);
      if (tid == 0) o[0] = rn[0];
      o[1 + tid] = rm[tid * 256];
      o[1 + M + tid] = rq[tid * 256];
    }
  }
  if (P.acc && (tid & 63) == 0 && nacc) atomicAdd(P.acc_count, nacc);
}



// ---- the same pass with the rows brought in by LDS-DMA (round 5; m = 16 / 32 / 64, K <= ADA_DMA_KMAX) ----------------------
// NOT the default (elfihip_dist_set_form(ctx, 2) selects it; kept for measurement): 10^7 x 64, K = 3 takes 1.35 ms here
// against 1.10 ms for adaptive_pass_kernel (profiles/r05_adaptive_dma.md).  The row stream itself is the faster one (the
// distance-only kernel: 0.79 against 0.68 of 8 TB/s), but a one-wave workgroup does everything a slot needs one thing after
// the other -- the K-column sweep of its rows 0.70 ms, the column statistics (read back from LDS, where the register-staged
// kernel takes them from its staging registers for nothing) 0.37 ms, the distances' stores 0.28 ms (vmcnt counts stores and
// DMA pieces together: the counted wait that frees the next slot also waits for the write acknowledgements) -- and four
// waves per CU cannot hide 1.35 ms of it behind 0.85 ms of memory time.
// The row stream of dist_rows_dma_kernel (distance.hip / tile_stream.hpp): one-wave workgroups, each with a ring of slots
// filled by `global_load_lds_dwordx4 ... nt`, no workgroup barrier, four workgroups per CU.  What changes against
// adaptive_pass_kernel, whose tile passes through registers:
//   * lane r owns row r of a slot and forms ALL K nested distances in ONE sweep over the row: d = x - y and d^2 once per
//     element, K independent left-to-right sums s_k = s_k + w_kj d^2 (each in cdist's order: bit-identical distances), the
//     observed row and the weight rows as broadcast LDS reads;
//   * the column statistics are read back from LDS: lane (g, c) owns column c of the rows g, g + G, ... of the slot
//     (G = 64 / m row groups), takes the textbook two-pass sums of its CH = ROWS / G values in registers and folds them into
//     its running triple with Chan's update; the G triples of a column meet in LDS at the end; one (1 + 2m) partial per
//     workgroup for adaptive_finish_kernel, as before.  A column walk touches every word of a row once (the swizzle
//     permutes 16-byte granules inside the row): conflict-free.
constexpr int ADA_DMA_KMAX = 8;

// One sweep over a row for KN weight vectors, eight 16-byte granules (16 elements) at a time: ALL LDS reads of a chunk --
// the row's granules, the observed pairs, the KN weight pairs -- are issued before the first of them is waited for (a
// scheduling barrier keeps the compiler from interleaving them with the arithmetic: left alone it read two or three operands,
// waited, computed, and a slot of 32 x 64 cost 11 500 cycles of exposed LDS latency -- 1.46 ms per 10^7 x 64, K = 3).  A
// one-wave workgroup has the SIMD's registers to itself, so the 8 (2 + KN) pairs of a chunk live in registers.
template <int MM, int KN>
__device__ __forceinline__ void dma_row_sums(const double* row, int key, const double* ys, const double* wk, double (&s)[KN]) {
  constexpr int H = MM / 2;
  constexpr int CG = 8;
  static_assert(H % CG == 0, "whole chunks");
#pragma unroll
  for (int k = 0; k < KN; ++k) s[k] = 0.0;
#pragma unroll
  for (int c0 = 0; c0 < H; c0 += CG) {
    double2 v[CG], yv[CG], wv[KN][CG];
#pragma unroll
    for (int i = 0; i < CG; ++i) v[i] = *reinterpret_cast<const double2*>(row + 2 * ((c0 + i) ^ key));
#pragma unroll
    for (int i = 0; i < CG; ++i) yv[i] = *reinterpret_cast<const double2*>(ys + 2 * (c0 + i));
#pragma unroll
    for (int k = 0; k < KN; ++k)
#pragma unroll
      for (int i = 0; i < CG; ++i) wv[k][i] = *reinterpret_cast<const double2*>(wk + k * MM + 2 * (c0 + i));
    __builtin_amdgcn_sched_barrier(0);
    double q[2 * CG];
#pragma unroll
    for (int i = 0; i < CG; ++i) {
      const double d0 = v[i].x - yv[i].x, d1 = v[i].y - yv[i].y;
      q[2 * i] = d0 * d0;
      q[2 * i + 1] = d1 * d1;
    }
#pragma unroll
    for (int i = 0; i < CG; ++i)
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        s[k] = s[k] + wv[k][i].x * q[2 * i];
        s[k] = s[k] + wv[k][i].y * q[2 * i + 1];
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// KNMAX sums at most per lane and sweep; kn of them in use
template <int MM>
__device__ __forceinline__ void dma_row_sweep(const double* row, int key, const double* ys, const double* wk, int kn,
                                              double (&r)[4]) {
  if (kn == 4) {
    double s4[4];
    dma_row_sums<MM, 4>(row, key, ys, wk, s4);
    r[0] = sqrt(s4[0]), r[1] = sqrt(s4[1]), r[2] = sqrt(s4[2]), r[3] = sqrt(s4[3]);
  } else if (kn == 3) {
    double s3[3];
    dma_row_sums<MM, 3>(row, key, ys, wk, s3);
    r[0] = sqrt(s3[0]), r[1] = sqrt(s3[1]), r[2] = sqrt(s3[2]);
  } else if (kn == 2) {
    double s2[2];
    dma_row_sums<MM, 2>(row, key, ys, wk, s2);
    r[0] = sqrt(s2[0]), r[1] = sqrt(s2[1]);
  } else if (kn == 1) {
    double s1[1];
    dma_row_sums<MM, 1>(row, key, ys, wk, s1);
    r[0] = sqrt(s1[0]);
  }
}

template <int MM, int ROWS, int D>
__global__ __launch_bounds__(64) void adaptive_dma_kernel(AdaptArgs P) {
  extern __shared__ __align__(16) double lds[];
  constexpr int SLOT = ROWS * MM;                // doubles
  constexpr int H = MM / 2;
  constexpr int PIECES = ROWS * H / 64;
  constexpr int G = 64 / MM;                     // row groups of the column statistics
  constexpr int CH = ROWS / G;                   // rows per lane and slot there (a power of two)
  constexpr int HALVES = 64 / ROWS;              // lanes per row in the distance sweep: 2 at 32-row slots (m = 64)
  const RowArgs& A = P.A;
  const int lane = threadIdx.x, K = A.K;
  double* ring = lds;
  double* ys = lds + (size_t)D * SLOT;           // (MM)
  double* ws = ys + MM;                          // (K, MM)
  double* accs = ws + (size_t)K * MM;            // (K)
  double* ostage = accs + ((K + 1) & ~1);        // (ROWS, K): a slot's distances on their way to coalesced stores
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  if (lane < MM) ys[lane] = A.y[lane];
  for (int j = lane; j < K * MM; j += 64) ws[j] = A.aux[j];
  if (P.acc)
    for (int j = lane; j < K; j += 64) accs[j] = P.acc[j];
  const int64_t nslots = (A.n + ROWS - 1) / ROWS;
  const int64_t stride = gridDim.x;
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  const double thr = A.F.thr ? *A.F.thr : inf;
  const bool stats = P.partial != nullptr;
  unsigned off[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int Gi = i * 64 + lane;
    const int row = Gi / H, g = Gi % H;
    off[i] = (unsigned)(((int64_t)row * A.ldx + 2 * (g ^ dma_swizzle_key<MM>(row))) * 8);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  int64_t t = blockIdx.x;
#pragma unroll
  for (int k = 0; k < D - 1; ++k) {
    const int64_t tk = t + k * stride;
    if (tk < nslots) dma_issue_slot<MM, ROWS>(A, off, tk * ROWS, lds_base + (unsigned)(k * SLOT * 8), lane);
  }
  // the distance sweep: lane = (half, row); with two lanes per row (32-row slots) the K columns are split between them --
  // the first ceil(K / 2) to half 0, the rest (the LAST column among them) to half 1 -- so each sweeps the row once for its
  // own columns and every sum keeps cdist's order
  const int rrow = lane % ROWS, half = lane / ROWS;
  const int ka = HALVES == 2 ? (K + 1) / 2 : K;          // columns per half: half 0 takes [0, ka), half 1 [ka, K) -- and, so
  const int kbeg = half == 0 ? 0 : ka;                   // that both halves run the SAME sweep (lanes of one wave), half 1
  const int kend = half == 0 ? ka : K;                   // repeats its last column when K is odd (clamped below, not stored)
  // statistics: lane (sg, sc) = rows sg, sg + G, ... and column sc of every slot
  const int sg = lane / MM, sc = lane % MM;
  ColStat st = {0.0, 0.0, 0.0};
  unsigned long long nacc = 0;
  int cur = 0;
  int64_t pend_row0 = 0;
  int pend_rows = 0, pend_buf = 0, obuf = 0;
  // (rows, K) doubles of a slot are contiguous in the output: 16-byte stores from the staging buffer
  auto flush_out = [&](int64_t row0_, int rows_, int buf) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const double* src = ostage + (size_t)buf * ROWS * K;
    const int total = rows_ * K;                     // doubles
    double* dst = A.out + row0_ * K;
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
      for (int e = 2 * lane; e < total; e += 128) {
        if (e + 1 < total)
          *reinterpret_cast<double2*>(dst + e) = *reinterpret_cast<const double2*>(src + e);
        else
          dst[e] = src[e];
      }
    } else {
      for (int e = lane; e < total; e += 64) dst[e] = src[e];
    }
  };
  for (; t < nslots; t += stride) {
    const int64_t tn = t + (int64_t)(D - 1) * stride;
    int nxt = cur + D - 1;
    if (nxt >= D) nxt -= D;
    if (tn < nslots) {
      dma_issue_slot<MM, ROWS>(A, off, tn * ROWS, lds_base + (unsigned)(nxt * SLOT * 8), lane);
      wait_vmcnt<PIECES * (D - 1)>();
    } else {
      wait_vmcnt<0>();
    }
    // the previous slot's distances leave now: their stores are the oldest operations in flight at the NEXT counted wait,
    // a whole trip away (issued at the end of their own trip they sat between two DMA batches, and the wait that frees the
    // next slot had to wait for their write acknowledgements as well: 0.25 ms of 1.2 per 10^7 x 64 rows)
    if (A.out && pend_rows > 0) flush_out(pend_row0, pend_rows, pend_buf);
    const int64_t row0 = t * ROWS;
    const int rows = (int)((A.n - row0) < ROWS ? (A.n - row0) : ROWS);
    const double* slot = ring + (size_t)cur * SLOT;
    double* ost = ostage + (size_t)obuf * ROWS * K;
    // ---- the K nested distances of this lane's row (its share of the columns), acceptance
    bool ok = rrow < rows;
    double dlast = 0.0;
    {
      const double* row = slot + (size_t)rrow * MM;
      const int key = dma_swizzle_key<MM>(rrow);
      for (int j0 = 0; j0 < ka; j0 += 4) {               // wave-uniform trip count and kn
        const int kn = ka - j0 < 4 ? ka - j0 : 4;
        int k0 = kbeg + j0;
        if (k0 + kn > K) k0 = K - kn;                      // half 1, odd K: the sweep slides back over a column it has done
        if (k0 < 0) k0 = 0;                                // (K = 1: half 1 repeats half 0's only column, nothing is kept)
        double r[4];
        dma_row_sweep<MM>(row, key, ys, ws + (size_t)k0 * MM, kn, r);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < kn && k0 + k >= kbeg && k0 + k < kend) {
            if (A.out) ost[rrow * K + k0 + k] = r[k];
            if (P.acc) ok = ok && r[k] <= accs[k0 + k];   // samplers.py:219-225 (a NaN distance is not accepted)
            dlast = r[k];
          }
      }
    }
    if constexpr (HALVES == 2) {
      // the row's verdict is the AND of its two lanes'; the last column lives in half 1 (half 0 when K = 1)
      const int other = __shfl_xor((int)ok, 32, 64);
      ok = ok && other != 0;
      if (K == 1) {
        ok = ok && half == 0;
      } else {
        ok = ok && half == 1;     // one lane per row counts and offers: the one that holds column K - 1
      }
    }
    if (P.acc) nacc += (unsigned long long)__popcll(__ballot(ok));
    if (A.F.thr) reject_offer(A.F, ok && dlast < thr, dlast, A.F.row_base + row0 + rrow);
    pend_row0 = row0;
    pend_rows = rows;
    pend_buf = obuf;
    obuf ^= 1;
    // ---- column statistics of the slot
    if (stats) {
      const unsigned char* base = reinterpret_cast<const unsigned char*>(slot);
      if (rows == ROWS) {
        double x[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int r_ = sg + i * G;
          x[i] = *reinterpret_cast<const double*>(base + (size_t)r_ * (MM * 8) + (((sc >> 1) ^ dma_swizzle_key<MM>(r_)) << 4) +
                                                  ((sc & 1) << 3));
        }
        double mean, q;
        chunk_stats<CH>(x, mean, q);
        chan_merge(st, (double)CH, mean, q);
      } else {
        int cnt = 0;
        double sum = 0.0;
        for (int r_ = sg; r_ < rows; r_ += G) {
          sum += *reinterpret_cast<const double*>(base + (size_t)r_ * (MM * 8) + (((sc >> 1) ^ dma_swizzle_key<MM>(r_)) << 4) +
                                                  ((sc & 1) << 3));
          ++cnt;
        }
        if (cnt > 0) {
          const double mt = sum / (double)cnt;
          double q = 0.0;
          for (int r_ = sg; r_ < rows; r_ += G) {
            const double e = *reinterpret_cast<const double*>(base + (size_t)r_ * (MM * 8) +
                                                               (((sc >> 1) ^ dma_swizzle_key<MM>(r_)) << 4) + ((sc & 1) << 3)) - mt;
            q += e * e;
          }
          chan_merge(st, (double)cnt, mt, q);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are done before it is refilled
    cur = cur + 1 == D ? 0 : cur + 1;
  }
  if (A.out && pend_rows > 0) flush_out(pend_row0, pend_rows, pend_buf);
  if (stats) {
    // the G row groups of a column, merged in order; the ring is free (nothing is in flight after the last trip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    double* red = ring;
    double* e0 = red + 3 * ((size_t)sg * MM + sc);
    e0[0] = st.n, e0[1] = st.mean, e0[2] = st.M2;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane < MM) {
      ColStat a = {0.0, 0.0, 0.0};
      for (int g2 = 0; g2 < G; ++g2) {
        const double* e = red + 3 * ((size_t)g2 * MM + lane);
        chan_merge(a, e[0], e[1], e[2]);
      }
      double* o = P.partial + (size_t)blockIdx.x * (1 + 2 * MM);
      if (lane == 0) o[0] = a.n;
      o[1 + lane] = a.mean;
      o[1 + MM + lane] = a.M2;
    }
  }
  if (P.acc && lane == 0 && nacc) atomicAdd(P.acc_count, nacc);
}

bool adaptive_dma_supported(const elfihip_ctx* ctx, const double* dX, int m, int64_t ldx, int K) {
  return ctx->dist_form == 2 && (m == 16 || m == 32 || m == 64) && K >= 1 && K <= ADA_DMA_KMAX && !(ldx & 1) &&
         ldx <= (1 << 21) && ada_aligned16(dX);
}

// One workgroup per column: the `nparts` workgroup partials of the pass(es) -> the batch's (count, mean, M2) in `bst`
// (1 + 2m).  Thread j merges partials j, j + 256, ... in that order, then the 256 triples are merged by a fixed tree.
// state != NULL: the fold into the running store (what adaptive_fold_kernel did as a launch of its own until round 6) by the
// column's workgroup itself: every workgroup reads the OLD count before it arrives on `done`; the last of the m to arrive
// -- every other one has read the count by then -- writes the new count and resets the counter.
__global__ __launch_bounds__(256) void adaptive_finish_kernel(const double* partial, int nparts, int m, double* bst,
                                                              double* state, unsigned* done) {
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
    if (state) {
      // state (1 + 2m: count, mean, M2) <- state U batch: what AdaptiveDistance.add_data leaves after the batch
      // (elfi_model.py:1116-1124), in Chan's form (the operations of adaptive_fold_kernel, in its order)
      const double n_old = state[0];
      ColStat f = {n_old, state[1 + c], state[1 + m + c]};
      if (n_old == 0.0) f.mean = 0.0, f.M2 = 0.0;
      chan_merge(f, a.n, a.mean, a.M2);
      state[1 + c] = f.mean;
      state[1 + m + c] = f.M2;
      const unsigned old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1u == (unsigned)m) {
        state[0] = f.n;
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

static size_t ada_lds_bytes(int m, int K, int R) {
  const int mp = m | 1;
  const size_t tile = std::max<size_t>((size_t)R * mp, 6 * (size_t)ADA_T);
  return (tile + 8 + (size_t)K + 2 * ADA_RMAX) * sizeof(double);   // + read-ahead slack, K thresholds, 4 x 128 ints of acceptance flags
}

static int ada_rows_per_tile(int m) {
  int R = 2 * ADA_T * ADA_U / m;      // 32 KiB of rows
  return R > ADA_RMAX ? ADA_RMAX : R;
}

static bool adaptive_narrow(const double* dX, int m, int64_t ldx, int K) {
  return (m == 2 || m == 4) && K >= 1 && K <= ADA_NARROW_KMAX && !(ldx & 1) && ada_aligned16(dX);
}

bool adaptive_pass_supported(const double* dX, int m, int64_t ldx, int K) {
  if (adaptive_narrow(dX, m, ldx, K)) return true;   // lane-owned rows (adaptive_narrow_kernel)
  // (m = 2 with more than eight weight vectors: the generic addressing streams such rows at a quarter of the separate
  // kernels' rate -- they take those)
  if (m < 4 || m > ADA_T / 2 || (m & 1) || (ldx & 1) || !ada_aligned16(dX)) return false;
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
  A.nt = ctx->dist_form != 1;   // the rows are read once
  A.div_h = make_fastdiv((uint32_t)(m / 2));
  A.F = F ? *F : RejectFilter{nullptr, nullptr, nullptr, nullptr, 0u, 0ll};
  P.acc = dacc;
  P.acc_count = dacc_count;
  P.partial = partial;
  if (adaptive_narrow(dX, m, ldx, K)) {
    constexpr int U = ADA_NARROW_U;
    int64_t gn = (n + 256 * U - 1) / (256 * U);
    if (gn > (int64_t)ctx->cu_count * 4) gn = (int64_t)ctx->cu_count * 4;   // (four workgroups per CU: fewer partials for the finish launch)
    if (m == 2)
      hipLaunchKernelGGL((adaptive_narrow_kernel<2, U>), dim3((unsigned)gn), dim3(256), 0, ctx->stream, P);
    else
      hipLaunchKernelGGL((adaptive_narrow_kernel<4, U>), dim3((unsigned)gn), dim3(256), 0, ctx->stream, P);
    if (nparts && partial) *nparts = (int)gn;
    return launch_status(ctx, "adaptive_narrow_kernel");
  }
  if (adaptive_dma_supported(ctx, dX, m, ldx, K)) {
    // LDS-DMA form: rings of two 16 KiB slots (four 8 KiB slots at m = 16) per one-wave workgroup, four workgroups per CU
    const int rows = m == 64 ? 32 : 64;
    const int D = m == 16 ? 4 : 2;
    const size_t ldsd = ((size_t)D * rows * m + (size_t)(1 + K) * m + (size_t)((K + 1) & ~1) + 2 * (size_t)rows * K + 2) * sizeof(double);
    int per_cu = (int)((160 * 1024) / ldsd);
    if (per_cu > 4) per_cu = 4;
    int64_t gd = (int64_t)ctx->cu_count * per_cu;
    const int64_t nslots = (n + rows - 1) / rows;
    if (gd > nslots) gd = nslots;
    if (gd < 1) gd = 1;
    if (m == 16)
      hipLaunchKernelGGL((adaptive_dma_kernel<16, 64, 4>), dim3((unsigned)gd), dim3(64), ldsd, ctx->stream, P);
    else if (m == 32)
      hipLaunchKernelGGL((adaptive_dma_kernel<32, 64, 2>), dim3((unsigned)gd), dim3(64), ldsd, ctx->stream, P);
    else
      hipLaunchKernelGGL((adaptive_dma_kernel<64, 32, 2>), dim3((unsigned)gd), dim3(64), ldsd, ctx->stream, P);
    if (nparts && partial) *nparts = (int)gd;
    return launch_status(ctx, "adaptive_dma_kernel");
  }
  const size_t lds = ada_lds_bytes(m, K, A.R);
  const int64_t ntiles = (n + A.R - 1) / A.R;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  int64_t g = (int64_t)ctx->cu_count * per_cu;
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  const int h = m / 2;
  const int nu = (ADA_T % h == 0 && A.R % (ADA_T / h) == 0) ? A.R / (ADA_T / h) : 0;   // loads per thread and tile when m / 2 divides the workgroup
#define ELFIHIP_ADA_LAUNCH(NU) \
  hipLaunchKernelGGL((adaptive_pass_kernel<NU>), dim3((unsigned)g), dim3(ADA_T), lds, ctx->stream, P)
  switch (nu) {
    case 8: ELFIHIP_ADA_LAUNCH(8); break;
    case 4: ELFIHIP_ADA_LAUNCH(4); break;
    case 2: ELFIHIP_ADA_LAUNCH(2); break;
    case 1: ELFIHIP_ADA_LAUNCH(1); break;
    default: ELFIHIP_ADA_LAUNCH(0); break;
  }
#undef ELFIHIP_ADA_LAUNCH
  if (nparts && partial) *nparts = (int)g;
  return launch_status(ctx, "adaptive_pass_kernel");
}

// partials -> batch statistics -> folded into the running store dstate (1 + 2m); bst: 1 + 2m doubles of scratch that
// keep the batch's own statistics.
int adaptive_stats_finish(elfihip_ctx* ctx, const double* partial, int nparts, int m, double* bst, double* dstate) {
  if (nparts <= 0) return ELFIHIP_OK;
  if (dstate && !ctx->fold_cnt) {
    ELFIHIP_CHECK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->fold_cnt), 64));
    ELFIHIP_CHECK_HIP(ctx, hipMemsetAsync(ctx->fold_cnt, 0, 64, ctx->stream));
  }
  hipLaunchKernelGGL(adaptive_finish_kernel, dim3((unsigned)m), dim3(256), 0, ctx->stream, partial, nparts, m, bst, dstate,
                     ctx->fold_cnt);
  return launch_status(ctx, "adaptive statistics kernel");
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

// host form; X == nullptr: the rows are the context's kept copy (ctx->rows, pitch m)
static int adaptive_push_host(elfihip_ctx* ctx, elfihip_reject* state, const double* X, int64_t n, int m, int64_t ldx,
                              const double* y, const double* W, int K, double* out, int64_t* count, double* mean,
                              double* M2, int64_t row_base) {
  ELFIHIP_REQUIRE(ctx, K >= 1 && K <= 64, "K=%d outside [1,64]", K);
  ELFIHIP_REQUIRE(ctx, y && W, "NULL data pointer");
  ELFIHIP_REQUIRE(ctx, (!count && !mean && !M2) || (count && mean && M2), "count / mean / M2 come together");
  ELFIHIP_REQUIRE(ctx, !state || reject_ctx(state) == ctx, "the sampler state belongs to another context");
  if (n == 0) return out ? keep_distances(ctx, nullptr, 0, K) : ELFIHIP_OK;   // (an empty batch is still a distance call)
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
  double* dX;
  int64_t ldd;
  if (X) {
    // the batch itself: packed (n, m), pitch m rounded up to even so that the rows stay 16-byte aligned
    ldd = (m + 1) & ~1;
    ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve((size_t)n * ldd * sizeof(double)));
    dX = ctx->in.as<double>();
    ELFIHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(dX, (size_t)ldd * sizeof(double), X, (size_t)ldx * sizeof(double),
                                            (size_t)m * sizeof(double), (size_t)n, hipMemcpyHostToDevice, st));
  } else {
    dX = ctx->rows.as<double>();
    ldd = m;
  }
  double* dout = nullptr;
  if (out) {
    ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve((size_t)n * K * sizeof(double)));
    dout = ctx->out.as<double>();
  }
  ELFIHIP_TRY(adaptive_push_impl(ctx, state, dX, n, m, ldd, dy, dW, K, dout, count ? dstate : nullptr, row_base));
  if (out) {
    ELFIHIP_TRY(keep_distances(ctx, dout, n, K));
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, dout, (size_t)n * K * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  if (count) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(hst.data(), dstate, ns * sizeof(double), hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  if (count) {
    *count = (int64_t)hst[0];
    memcpy(mean, &hst[1], (size_t)m * sizeof(double));
    memcpy(M2, &hst[1 + m], (size_t)m * sizeof(double));
  }
  return ELFIHIP_OK;
}

int elfihip_adaptive_push(elfihip_ctx* ctx, elfihip_reject* state, const double* X, int64_t n, int m, int64_t ldx,
                          const double* y, const double* W, int K, double* out, int64_t* count, double* mean,
                          double* M2, int64_t row_base) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, n >= 0 && m >= 1 && ldx >= m, "bad shape n=%lld m=%d ldx=%lld", (long long)n, m, (long long)ldx);
  ELFIHIP_REQUIRE(ctx, n == 0 || X, "NULL data pointer");
  return adaptive_push_host(ctx, state, X, n, m, ldx, y, W, K, out, count, mean, M2, row_base);
}

int elfihip_adaptive_push_kept(elfihip_ctx* ctx, elfihip_reject* state, uint64_t rows_epoch, const double* y,
                               const double* W, int K, double* out, int64_t* count, double* mean, double* M2,
                               int64_t row_base) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  if (rows_epoch != ctx->rows_epoch || (ctx->rows_n > 0 && !ctx->rows.p))
    return fail(ctx, ELFIHIP_ERR_STATE, "the kept rows are those of a later call (epoch %llu, asked for %llu)",
                (unsigned long long)ctx->rows_epoch, (unsigned long long)rows_epoch);
  return adaptive_push_host(ctx, state, nullptr, ctx->rows_n, ctx->rows_m, ctx->rows_m, y, W, K, out, count, mean, M2,
                            row_base);
}

}  // extern "C"
