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
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const double2*>(src + u * sstep);
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
  const size_t tile = std::max<size_t>((size_t)R * mp, 6 * (size_t)ADA_T);
  return (tile + 8 + (size_t)K + 2 * ADA_RMAX) * sizeof(double);   // + read-ahead slack, K thresholds, 4 x 128 ints of acceptance flags
}

static int ada_rows_per_tile(int m) {
  int R = 2 * ADA_T * ADA_U / m;      // 32 KiB of rows
  return R > ADA_RMAX ? ADA_RMAX : R;
}

bool adaptive_pass_supported(const double* dX, int m, int64_t ldx, int K) {
  // (m = 2 streams through the generic addressing at a quarter of the separate kernels' rate: 4 10^6 x 2, K = 3: 0.135 ms
  // fused against 0.071 + 0.046)
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
