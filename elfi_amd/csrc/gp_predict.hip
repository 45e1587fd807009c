// GP posterior mean / variance / gradients and the LCB acquisition at many points at once.
//
// Replaces GPyRegression.predict / predictive_gradients
// (elfi/methods/bo/gpy_regression.py:98-147,179-223; closed forms :127-140,:206-218) and
// LCBSC.evaluate / evaluate_gradient (elfi/methods/bo/acquisition.py:262-301).  The reference
// evaluates ONE point per call (three O(n^2) products with the dense K^-1 each, see
// SURVEY.md 3.3); here S points are evaluated together in groups of 16 columns:
//
//   kr[s][i] = s_f exp(-|x_s - X_i|^2 / 2 l^2)                       (S x n, VALU + exp)
//   mu_s     = sum_i (kr + s_b) alpha_i
//   v        = L^-1 (kr + s_b)^T        v[i][s] = sum_{k<=i} WT[k][i] kb[s][k]   (n x 16, MFMA f64)
//   var_s    = s_f + s_b - sum_i v[i][s]^2
//   u        = L^-T v = K^-1 kb^T       u[i][s] = sum_{k>=i} WT[i][k] v[k][s]    (n x 16, MFMA f64)
//   dmu_s    = sum_i alpha_i dk_si,   dvar_s = -2 sum_i u[i][s] dk_si,
//              dk_si = -(kr[s][i] / l^2) (x_s - X_i)                  (RBF part only, as GPy)
//
// The two triangular products stream a triangle of n^2/2 doubles once each: HBM/L3-bound for 16
// columns, so they are split over (32-row block, 256-deep k chunk) pairs -- about a thousand
// workgroups of <= 64 KiB at n = 4096, enough for the dispatcher to balance the triangle over the
// 256 CUs -- and reduced in a fixed order (deterministic).  Both products read their matrix
// row-wise along k (v from L^-T, u from its mirror image L^-1, gp->WL), so one kernel serves both.
// The strictly-lower part of WT (strictly-upper of WL) is zero, so no masking on the diagonal.
#include <chrono>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include <cmath>

#include "gp.hpp"
#include "mfma_f64.hpp"
#include "special.hpp"


// The hand-offs between workgroups in this file (write-through stores, s_waitcnt vmcnt(0), relaxed agent-scope atomics, one
// acquire at the consumer) rely on the gfx9 family counting stores in vmcnt and on sc1 atomics writing through.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "built for gfx950 (MI355X): the cross-workgroup hand-offs here are not valid on this target"
#endif

namespace elfihip {

constexpr int PC = 16;    // columns (query points) per pass
constexpr int KCH = 2;    // 128-blocks of k per chunk of the triangular products
constexpr int KC = KCH * NB;  // = 256 k per workgroup
constexpr int RB = 32;    // output rows per workgroup of the triangular products
constexpr int MAX_GROUP = 8;  // 16-point passes handled by one set of launches


// Query points of a single-pass call travel in the kernel arguments (the argument block is written by the host and
// read from device memory): [point][dimension, padded to 24] and the squared norms -- 3.2 KiB of the 4 KiB an
// argument block may hold, enough for the 20 parameters of BASELINE configs[4].
constexpr int QUERY_ARGS_MAX_DP = 24;
struct QueryArgs {
  double x[PC][QUERY_ARGS_MAX_DP];
  double x2[PC];
};

#ifdef ELFIHIP_TRI_STAMP   // developer probe (scripts/tri_timeline.py): wall-clock stamps (100 MHz) per workgroup and phase
__device__ unsigned long long g_tri_stamp[8192 * 8];
#define PSTAMP(slot) do { const unsigned pidx_ = (MODE == 1 ? 4096u : 0u) + blockIdx.x + blockIdx.y * gridDim.x; \
    if (threadIdx.x == 0 && pidx_ < 8192) g_tri_stamp[pidx_ * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define KSTAMP(slot) do { const unsigned kidx_ = 7168u + blockIdx.x + 16u * blockIdx.y; \
    if (threadIdx.x == 0 && blockIdx.z == 0 && kidx_ < 8192) g_tri_stamp[kidx_ * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PSTAMP(slot) do { } while (0)
#define KSTAMP(slot) do { } while (0)
#endif

// ---- kr[s][i], partial mu ----------------------------------------------------------
__global__ __launch_bounds__(256) void kstar_kernel(const double* X, const double* x2, const double* alpha,
                                                    const double* xs, const double* xs2, double* kr, double* kb,
                                                    double* mu_part, int64_t n, int64_t np, int dp, double var,
                                                    double neg_half_inv_ls2, double bias, double* xs_copy,
                                                    int from_args, QueryArgs q, double* kbt) {
  __shared__ double red[256];
  __shared__ double sx[256 + 1];  // this workgroup's query point and its squared norm
  const int s = blockIdx.y;
  KSTAMP(0);
  // several 16-point passes in one launch (blockIdx.z): per-pass slices of the query points and outputs
  xs += (int64_t)blockIdx.z * PC * dp;
  xs2 += (int64_t)blockIdx.z * PC;
  kr += (int64_t)blockIdx.z * PC * np;
  if (kb) kb += (int64_t)blockIdx.z * PC * np;
  if (kbt) kbt += (int64_t)blockIdx.z * PC * np;   // [point][k]: the k-contiguous form the dense products read (gp_dense.hip)
  mu_part += (int64_t)blockIdx.z * PC * gridDim.x;
  if (from_args) {
    // single-pass call: the point comes with the arguments (uniform index: two wide scalar loads); workgroup
    // column 0 leaves a device copy of it for the gradient kernel
    double row[QUERY_ARGS_MAX_DP];
#pragma unroll
    for (int c = 0; c < QUERY_ARGS_MAX_DP; ++c) row[c] = q.x[s][c];
    if (threadIdx.x < QUERY_ARGS_MAX_DP) {
      double x = 0.0;
#pragma unroll
      for (int c = 0; c < QUERY_ARGS_MAX_DP; ++c) x = (int)threadIdx.x == c ? row[c] : x;
      sx[threadIdx.x] = x;
      if (blockIdx.x == 0 && (int)threadIdx.x < dp) xs_copy[s * dp + threadIdx.x] = x;
    }
    if (threadIdx.x == 0) sx[256] = q.x2[s];
  } else {
    // xs may be pinned host memory here (multi-pass calls without an upload): one parallel read per workgroup
    if ((int)threadIdx.x < dp) {
      const double x = xs[s * dp + threadIdx.x];
      sx[threadIdx.x] = x;
      if (xs_copy && blockIdx.x == 0) xs_copy[(int64_t)blockIdx.z * PC * dp + s * dp + threadIdx.x] = x;
    }
    if (threadIdx.x == 0) sx[256] = xs2[s];
  }
  __syncthreads();
  KSTAMP(1);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double contrib = 0.0;
  if (i < np) {
    double k = 0.0;
    if (i < n) {
      double dot = 0.0;
      for (int c = 0; c < dp; ++c) dot += sx[c] * X[i * dp + c];
      double r2 = (sx[256] + x2[i]) + (-2.0 * dot);
      r2 = r2 < 0.0 ? 0.0 : r2;
      k = var * exp(r2 * neg_half_inv_ls2);
      contrib = (k + bias) * alpha[i];
    }
    kr[(int64_t)s * np + i] = k;
    const double kbv = i < n ? k + bias : 0.0;  // right-hand side of the first triangular product
    if (kb) kb[i * PC + s] = kbv;               // [k][s] for the streaming kernel
    if (kbt) kbt[(int64_t)s * np + i] = kbv;
  }
  // fixed order: butterfly inside the wave, then the four waves in order (one barrier instead of nine)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) contrib += __shfl_xor(contrib, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = contrib;
  __syncthreads();
  KSTAMP(2);
  if (threadIdx.x == 0) mu_part[s * gridDim.x + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ---- triangular skinny products on the matrix cores ------------------------------------
// MODE 0 (upper):  out[i][s] = sum_{k <= i} W[k][i] bin[k][s],  W = L^-T (gp->WT),  bin = kb     (v = L^-1 kb)
// MODE 1 (lower):  out[i][s] = sum_{k >= i} W[k][i] bin[k][s],  W = L^-1 (gp->WL),  bin = v      (u = L^-T v)
// MODE 2 (dense):  out[i][s] = sum_k        W[k][i] bin[k][s],  W = V_P (gp->VP),   bin = v      (V_P^T v)
// MODE 3 (full):   out[i][s] = sum_k        W[k][i] bin[k][s],  W = K^-1 (gp->Kinv, symmetric), bin = kb   (u = K^-1 kb)
//                  -- ONE product instead of the two dependent triangular ones: the same 8 n^2 bytes in one launch, no
//                  second launch ramp / round trip / hand-off (measured: gp_predict.hip, ensure_kinv_sym below)
// Workgroup (rb, kc): rows i in [32 rb, 32 rb + 32), k in chunk kc clipped to the triangle (a multiple
// of 32 long).  Writes part[kc][i][s]; chunks outside the triangle are never read by the reduction.
struct TriArgs {
  const double* W;
  const double* bin;   // [k][s], np x 16 per pass
  double* part;
  int64_t lda, np;
  int64_t nout;  // rows of the output (np for the triangular products, the padded point count for the dense one)
  int nrb, nkc;
  int npass;   // 16-point passes in this launch (see the blockIdx mapping in the kernel)
  // FUSED form (one launch instead of product + reduction / product + gradient sums): the LAST workgroup to arrive at
  // a row block sums that block's chunk partials and runs the epilogue the separate launch ran
  unsigned* cnt;          // (nrb) arrival counters, zero between launches (the last arriver resets its counter)
  double* vout;           // MODE 0: v[pass][i][s]
  double* sq_part;        // MODE 0: sum_i v^2 per 16-row block [pass][np / 16][16]
  const double* X;        // MODE 1: gradient sums of the row block
  const double* alpha;
  const double* xs;       // [pass][16][dp]
  const double* kr;       // [pass][16][np]
  double* g_part;         // [pass][16][nrb][2 dp]
  int64_t n;
  int dp;
  double bias;            // MODE 3: kb = kr + bias, for the variance term sum_i kb_i u_i of the row block
};

typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) unsigned int gu32_t;

// One 8-byte WRITE-THROUGH store (global_store_dwordx2 ... sc1): the value leaves this XCD's L2 with the store itself, so
// publishing a workgroup's partials needs no release fence (buffer_wbl2 of a whole L2: 1.7-6.5 us per workgroup, times
// the workgroups per CU -- what made round 2's last-arriver reduction slower than the launch it saved).
__device__ __forceinline__ void store_wt(double* p, double v) {
  __hip_atomic_store((gu64_t*)(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline double sum_partials(const double* __restrict__ part, int64_t np, int64_t e, int lo, int hi);
template <int ROWS_PER_THREAD, int QN, bool FULL = false>
__device__ __forceinline__ void grad_rows(const double* __restrict__ X, const double* __restrict__ alpha,
                                          const double* __restrict__ xs, const double* __restrict__ kr,
                                          const double* __restrict__ part, int nkc, double* __restrict__ g_part, int chunk,
                                          int nchunks, int64_t i0, int64_t n, int64_t np, int dp, double (*red)[PC][32],
                                          double bias = 0.0, double* __restrict__ sq_part = nullptr);

template <int MODE, bool FUSE>
__device__ __forceinline__ void tri_apply_body(const TriArgs& T, const int rb, const int kc, double* Bs, int* s_last_p) {
  // No LDS staging of the matrix: a lane's 16-byte load IS its MFMA operand.  Lane (kq = l >> 4, ip = l & 15)
  // of wave w loads W[k][i0 + 2 ip .. + 1] for k = k0 + 16 q + 4 w + kq, q = 0 .. len/16: one instruction covers
  // 4 rows x 256 contiguous bytes, and its two doubles feed two v_mfma_f64_16x16x4 (even rows i / odd rows i)
  // against B[k][s] from LDS.  All (up to 16) loads of a lane are issued before the first MFMA, so a workgroup
  // has its whole <= 64 KiB in flight at once; the four waves' partial tiles are added in fixed order at the end.
  // (Bs: KC * PC doubles of LDS -- the right-hand sides of this chunk [k][s]; later the wave partials)
  // One workgroup serves ALL passes of the launch for its (row block, chunk): the matrix operands stay in registers
  // and only the right-hand sides change -- with a workgroup per pass every pass pulled its 64 KiB of the matrix
  // through L2 again (16 passes at n = 8192: 8.6 GB per evaluation round, L2-bound at 1.6 ms against 0.45 ms of
  // matrix-pipe time).
  if (rb >= T.nrb) return;
  PSTAMP(0);
  const int64_t i0 = (int64_t)rb * RB;
  int64_t k0 = (int64_t)kc * KC, k1 = k0 + KC;
  if (MODE == 1) {
    if (k0 < i0) k0 = i0;
    if (k1 > T.np) k1 = T.np;
  } else if (MODE == 0) {
    if (k1 > i0 + RB) k1 = i0 + RB;
  } else {
    if (k1 > T.np) k1 = T.np;
  }
  if (k0 >= k1) return;  // chunk lies outside the triangle
  const int len = (int)(k1 - k0);
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  typedef double v2d __attribute__((ext_vector_type(2)));
  const int nbr = len / 32, nq = len / 16;  // 16-byte pieces of B per thread, MFMA pairs per wave
  // Straight-line code for every chunk length: pieces past the end of a clipped chunk re-read its last piece
  // (cache hit) and are multiplied by zero right-hand sides.
  v2d breg[8], wr[16];
  auto load_b = [&](int pass) {
    const double* bin = T.bin + ((int64_t)pass * T.np + k0) * PC;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int pp = p < nbr ? p : nbr - 1;
      const v2d x = *reinterpret_cast<const v2d*>(bin + pp * 512 + 2u * (unsigned)t);
      breg[p] = p < nbr ? x : (v2d){0.0, 0.0};
    }
  };
  load_b(0);
  // uniform row base + 32-bit lane offset
  // (a tiled copy of the matrix -- every workgroup's 64 KiB one contiguous run -- was measured through this kernel's
  // addressing alone and changes nothing: 21.7 / 25.2 us per fused product at n = 4096 either way; the products are
  // bound by launch ramp + one memory round trip per workgroup + the last arriver's epilogue, not by DRAM locality)
  const double* wbase = T.W + k0 * T.lda + i0;
  const unsigned lane_off = (unsigned)(4 * w + (l >> 4)) * (unsigned)T.lda + 2u * (l & 15);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int qq = q < nq ? q : nq - 1;
    wr[q] = *reinterpret_cast<const v2d*>(wbase + (int64_t)16 * qq * T.lda + lane_off);
  }
  for (int pass = 0; pass < T.npass; ++pass) {
    if (pass > 0) __syncthreads();   // the previous pass's partial sums have been read
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<v2d*>(Bs + p * 512 + 2 * t) = breg[p];
    __syncthreads();
    PSTAMP(2);
    if (pass + 1 < T.npass) load_b(pass + 1);   // in flight during this pass's products
    v4d acc0 = (v4d){0, 0, 0, 0}, acc1 = (v4d){0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const double b = Bs[(16 * q + 4 * w + (l >> 4)) * PC + (l & 15)];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[q].x, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[q].y, b, acc1, 0, 0, 0);
    }
    __syncthreads();
    // wave partials -> LDS as [w][i local][s] (i local = 2 * (MFMA row) + even/odd), then fixed-order sums
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ip = (l >> 4) + 4 * r;
      Bs[(w * RB + 2 * ip) * PC + (l & 15)] = acc0[r];
      Bs[(w * RB + 2 * ip + 1) * PC + (l & 15)] = acc1[r];
    }
    __syncthreads();
    double* out = T.part + (((int64_t)pass * T.nkc + kc) * T.nout + i0) * PC;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = t + 256 * h;  // (i local, s) = (e >> 4, e & 15)
      const double v = ((Bs[e] + Bs[RB * PC + e]) + Bs[2 * RB * PC + e]) + Bs[3 * RB * PC + e];
      if (FUSE)
        store_wt(out + e, v);
      else
        out[e] = v;
    }
  }
  if (!FUSE || MODE == 2) return;
  int& s_last = *s_last_p;
  PSTAMP(3);
  // ---- arrival at the row block (MI355X_MICROARCH.md, hand-off price list: write-through payload, every storing wave
  // drains, ONE lane arrives on the block's counter with a relaxed device-scope atomic; the last arriver takes ONE
  // agent-scope acquire -- its CU's L1 may hold the other workgroups' lines from an earlier launch -- then plain loads)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    // chunks inside the triangle for this row block = workgroups that arrive
    const unsigned expected = MODE == 0 ? (unsigned)((i0 + RB - 1) / KC + 1)
                                        : (MODE == 1 ? (unsigned)(T.nkc - (int)(i0 / KC)) : (unsigned)T.nkc);
    gu32_t* c = (gu32_t*)(T.cnt + rb);   // global address space said explicitly: no flat atomics
    const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old + 1u == expected;
    if (last) {
      __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    s_last = last;
  }
  __syncthreads();
  PSTAMP(4);
  if (!s_last) return;
  if (MODE == 0) {
    // what tri_reduce_kernel does for these 32 rows (same order of summation, same per-16-row blocks of v^2)
    const int ib = (int)(i0 / NB);
    for (int pass = 0; pass < T.npass; ++pass) {
      const double* part = T.part + (int64_t)pass * T.nkc * T.np * PC;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t e = i0 * PC + t + 256 * h;
        const double v = sum_partials(part, T.np, e, 0, ib / KCH + 1);
        T.vout[(int64_t)pass * T.np * PC + e] = v;
        __syncthreads();
        Bs[t] = v * v;
        __syncthreads();
        if (t < PC) {
          double q = 0.0;
          for (int r = 0; r < 16; ++r) q += Bs[r * PC + t];
          T.sq_part[((int64_t)pass * (T.np / 16) + (i0 / 16 + h)) * PC + t] = q;
        }
      }
    }
  } else if (MODE == 1) {
    // what grad_kernel does, for this block's 32 rows (chunk = row block)
    double(*red)[PC][32] = reinterpret_cast<double(*)[PC][32]>(Bs);
    for (int pass = 0; pass < T.npass; ++pass)
      grad_rows<2, 2>(T.X, T.alpha, T.xs + (int64_t)pass * PC * T.dp, T.kr + (int64_t)pass * PC * T.np,
                   T.part + (int64_t)pass * T.nkc * T.np * PC, T.nkc,
                   T.g_part + (int64_t)pass * PC * T.nrb * 2 * T.dp, rb, T.nrb, i0, T.n, T.np, T.dp, red);
  } else {
    // MODE 3: u of this block's rows is complete -- the gradient sums as above and the block's share of kb . u
    double(*red)[PC][32] = reinterpret_cast<double(*)[PC][32]>(Bs);
    for (int pass = 0; pass < T.npass; ++pass)
      grad_rows<2, 2, true>(T.X, T.alpha, T.xs + (int64_t)pass * PC * T.dp, T.kr + (int64_t)pass * PC * T.np,
                            T.part + (int64_t)pass * T.nkc * T.np * PC, T.nkc,
                            T.g_part + (int64_t)pass * PC * T.nrb * 2 * T.dp, rb, T.nrb, i0, T.n, T.np, T.dp, red, T.bias,
                            T.sq_part + (int64_t)pass * (T.np / 16) * PC);
  }
  PSTAMP(5);
}

constexpr int RB_PER_KC = KC / RB;

// Tiles of a fused product as a ONE-dimensional grid, CHUNK by chunk (round 6).  The (row block, chunk) grid of rounds 2-5
// launched 2048 workgroups at n = 4096 of which 960 lay outside the triangle and left at once, and ran the second product in
// ASCENDING chunk order -- where every row block needs the LAST chunk, so all 128 last-arriver epilogues ran together after
// the last tile.  Now only live tiles are launched and the second product runs its chunks in DESCENDING order: row blocks
// 8 kc .. 8 kc + 7 are complete once chunk kc has run and their epilogues run under the next chunks' tiles.  Measured
// (profiles/r06_lockstep_timeline.md): 62.4 -> 61.6 us per lock-step at n = 4096, 52.2 -> 50.6 with K^-1 -- the tail is still
// ONE epilogue (5.5-6.4 us behind the last tile), which is a chain of dependent memory and barrier round trips, not work.
// (Row block by row block instead -- measured -- is 15-40 % SLOWER: the tiles in flight then share their 256-byte column
// offsets, i.e. their memory channels; chunk by chunk the 128 row blocks of a chunk cover whole 32 KiB rows.)
// Same tiles, same partial slots, same order of summation as the two-dimensional grid.
//   MODE 0 (k <= i): chunk kc holds the row blocks rb >= 8 kc, kc ascending (row blocks complete from the top anyway)
//   MODE 1 (k >= i): chunk kc holds the row blocks rb < 8 (kc + 1), kc DESCENDING          MODE 3 (full): nrb tiles per chunk
template <int MODE>
__device__ __forceinline__ bool tri_tile_of(int b, int nrb, int nkc, int* rb, int* kc) {
  if (MODE == 3) {
    *kc = b / nrb;
    *rb = b - *kc * nrb;
    return *kc < nkc;
  }
  for (int c = 0; c < nkc; ++c) {
    const int k = MODE == 0 ? c : nkc - 1 - c;
    const int lim = RB_PER_KC * (k + 1);
    const int cnt = MODE == 0 ? nrb - RB_PER_KC * k : (lim < nrb ? lim : nrb);
    if (b < cnt) {
      *kc = k;
      *rb = MODE == 0 ? RB_PER_KC * k + b : b;
      return true;
    }
    b -= cnt;
  }
  return false;
}

static int tri_tiles(int mode, int nrb, int nkc) {
  if (mode == 3) return nrb * nkc;
  int total = 0;
  for (int rb = 0; rb < nrb; ++rb) total += mode == 0 ? rb / RB_PER_KC + 1 : nkc - rb / RB_PER_KC;
  return total;
}

// Three workgroups per CU (<= 168 registers).  Four were measured twice: with 128 registers and spills (round 3: 61 -> 74 us per
// lock-step) and, round 6, without spills (the right-hand sides DMA'd straight into LDS, 112 registers): 1024 tiles in flight
// instead of 768 lengthen every tile's life from 7.1 to 10.8 us -- the products already pull 6.5 TB/s while their tiles are
// in flight -- and the lock-step went from 61.6 to 67.5 us (profiles/r06_lockstep_timeline.md).
template <int MODE, bool FUSE>
__global__ __launch_bounds__(256, 3) void tri_apply_kernel(TriArgs T) {
  __shared__ __align__(16) double Bs[KC * PC];
  __shared__ int s_last;
  if (FUSE && MODE != 2) {
    int rb, kc;
    if (!tri_tile_of<MODE>((int)blockIdx.x, T.nrb, T.nkc, &rb, &kc)) return;
    tri_apply_body<MODE, FUSE>(T, rb, kc, Bs, &s_last);
  } else {
    tri_apply_body<MODE, FUSE>(T, (int)blockIdx.x, (int)blockIdx.y, Bs, &s_last);
  }
}

// Sum of the chunk partials of element e = i * 16 + s over chunks [lo, hi), in the one fixed order every consumer
// uses: chunk kc goes to accumulator kc & 3 in increasing kc, then (a0 + a1) + (a2 + a3).  Sixteen loads are issued
// together (chunks outside the range contribute an exact 0.0), so the sum costs one memory round trip per 16 chunks.
__device__ inline double sum_partials(const double* __restrict__ part, int64_t np, int64_t e, int lo, int hi) {
  double acc[4] = {0, 0, 0, 0};
  for (int base = lo & ~3; base < hi; base += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int kc = base + u;
      v[u] = (kc >= lo && kc < hi) ? part[(int64_t)kc * np * PC + e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u & 3] += v[u];
  }
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// v[i][s] = sum_kc part[kc][i][s] (fixed order); optional per-block partials of sum_i v^2.
__global__ __launch_bounds__(256) void tri_reduce_kernel(const double* part, double* out, double* sq_part,
                                                         int64_t np, int nkc, int kc_lo_is_row, int want_sq) {
  __shared__ double red[256];
  part += (int64_t)blockIdx.y * nkc * np * PC;   // per-pass slices (blockIdx.y = pass)
  out += (int64_t)blockIdx.y * np * PC;
  if (want_sq) sq_part += (int64_t)blockIdx.y * gridDim.x * PC;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // element (i, s), 16 rows per block
  double v = 0.0;
  if (e < np * PC) {
    const int64_t i = e / PC;
    const int ib = (int)(i / NB);
    // chunks inside the triangle: first product (kc_lo_is_row == 0): kc*KCH <= ib ; second: kc*KCH+KCH > ib
    const int lo = kc_lo_is_row == 1 ? ib / KCH : 0;              // kc_lo_is_row == 2: dense product, every chunk
    const int hi = kc_lo_is_row == 0 ? ib / KCH + 1 : nkc;
    v = sum_partials(part, np, e, lo, hi);
    out[e] = v;
  }
  if (want_sq) {
    red[threadIdx.x] = v * v;
    __syncthreads();
    // columns are e % 16: reduce the 16 rows of this block per column, fixed order
    if (threadIdx.x < PC) {
      double s = 0.0;
      for (int r = 0; r < 16; ++r) s += red[r * PC + threadIdx.x];
      sq_part[(int64_t)blockIdx.x * PC + threadIdx.x] = s;
    }
  }
}

// ---- gradients ---------------------------------------------------------------------
// g_part[s][chunk][0..dp) = sum_i alpha_i kr_si (x_s - X_i),  [dp..2dp) = sum_i u_is kr_si (x_s - X_i)
// Workgroup = 64 evidence rows x the 16 query points; thread (s = t & 15, ig = t >> 4) owns rows ig, ig + 16,
// ig + 32, ig + 48 of the chunk.  u is summed here from the chunk partials of the second triangular product
// (same order as tri_reduce_kernel), which saves that kernel on the prediction path.  Four dimensions at a time:
// register sums over the thread's rows, two butterfly steps over the wave's row groups, the four waves in order.
constexpr int GR = 64;
// The sums of RPT x 16 evidence rows from i0 on (thread (s, ig) owns rows ig + 16 r), written as chunk `chunk` of
// `nchunks`: the body of grad_kernel (RPT = 4) and of the fused second product's last arriver (RPT = 2, gp_predict.hip above).
// FULL (the K^-1 product): every chunk contributes to a row, and the rows' kb_i u_i are summed per 16-row block into
// sq_part[(i0 / 16 + r)][s] -- the slot the first triangular product fills with sum v^2, so finish_kernel is unchanged.
template <int RPT, int QN, bool FULL>   // QN: groups of four dimensions per round (4: grad_kernel; 2 keeps the fused product at 3 workgroups per CU)
__device__ __forceinline__ void grad_rows(const double* __restrict__ X, const double* __restrict__ alpha,
                                          const double* __restrict__ xs, const double* __restrict__ kr,
                                          const double* __restrict__ part, int nkc, double* __restrict__ g_part, int chunk,
                                          int nchunks, int64_t i0, int64_t n, int64_t np, int dp, double (*red)[PC][32],
                                          double bias, double* __restrict__ sq_part) {
  static_assert(!FULL || (RPT == 2 && QN == 2), "the variance slots of the K^-1 form sit behind two groups of four dimensions");
  const int t = threadIdx.x, s = t & 15, ig = t >> 4, w = t >> 6;
  double c1[RPT], c2[RPT], pv[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int64_t i = i0 + ig + 16 * r;
    c1[r] = 0.0;
    c2[r] = 0.0;
    pv[r] = 0.0;
    if (i < n) {
      const double k = kr[(int64_t)s * np + i];
      const double u = sum_partials(part, np, i * PC + s, FULL ? 0 : (int)(i / KC), nkc);
      c1[r] = alpha[i] * k;
      c2[r] = u * k;
      pv[r] = u * (k + bias);
    }
  }
  if (FULL) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      pv[r] += __shfl_xor(pv[r], 16, 64);
      pv[r] += __shfl_xor(pv[r], 32, 64);
    }
  }
  typedef double v4 __attribute__((ext_vector_type(4)));
  // sixteen dimensions per round (dp <= 16: one round, two barriers): register sums over the thread's rows, two
  // butterfly steps over the wave's row groups, the four waves in order
  for (int a0 = 0; a0 < dp; a0 += 4 * QN) {
    v4 g1[QN], g2[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      g1[q] = (v4){0, 0, 0, 0};
      g2[q] = (v4){0, 0, 0, 0};
      if (a0 + 4 * q < dp) {
        const v4 x4 = *reinterpret_cast<const v4*>(xs + s * dp + a0 + 4 * q);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const v4 diff = x4 - *reinterpret_cast<const v4*>(X + (i0 + ig + 16 * r) * dp + a0 + 4 * q);  // rows < np exist (zeros)
          g1[q] += c1[r] * diff;
          g2[q] += c2[r] * diff;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g1[q][j] += __shfl_xor(g1[q][j], 16, 64);
          g2[q][j] += __shfl_xor(g2[q][j], 16, 64);
          g1[q][j] += __shfl_xor(g1[q][j], 32, 64);
          g2[q][j] += __shfl_xor(g2[q][j], 32, 64);
        }
      }
    }
    __syncthreads();  // red free
    if ((t & 63) < 16) {
#pragma unroll
      for (int q = 0; q < QN; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          red[w][s][4 * q + j] = g1[q][j];
          red[w][s][16 + 4 * q + j] = g2[q][j];
        }
      if (FULL && a0 == 0) {   // slots 8, 9 of a point: free beside two groups of four dimensions
#pragma unroll
        for (int r = 0; r < RPT; ++r) red[w][s][8 + r] = pv[r];
      }
    }
    __syncthreads();
    for (int e = t; e < PC * 32; e += 256) {
      const int ss = e >> 5, j = e & 31;
      const int a = a0 + (j & 15);
      if ((j & 15) < 4 * QN && a < dp) {
        const double v = ((red[0][ss][j] + red[1][ss][j]) + red[2][ss][j]) + red[3][ss][j];
        g_part[((int64_t)ss * nchunks + chunk) * 2 * dp + (j < 16 ? a : dp + a)] = v;
      } else if (FULL && a0 == 0 && j >= 8 && j < 8 + RPT) {
        // the four waves hold rows ig = 4 w .. 4 w + 3 of each 16-row block: in order
        const double v = ((red[0][ss][j] + red[1][ss][j]) + red[2][ss][j]) + red[3][ss][j];
        sq_part[(i0 / 16 + (j - 8)) * PC + ss] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void grad_kernel(const double* __restrict__ X, const double* __restrict__ alpha,
                                                   const double* __restrict__ xs, const double* __restrict__ kr,
                                                   const double* __restrict__ part, int nkc,
                                                   double* __restrict__ g_part, int64_t n, int64_t np, int dp) {
  __shared__ double red[4][PC][32];
  xs += (int64_t)blockIdx.y * PC * dp;                 // per-pass slices (blockIdx.y = pass)
  kr += (int64_t)blockIdx.y * PC * np;
  part += (int64_t)blockIdx.y * nkc * np * PC;
  g_part += (int64_t)blockIdx.y * PC * gridDim.x * 2 * dp;
  grad_rows<4, 4, false>(X, alpha, xs, kr, part, nkc, g_part, (int)blockIdx.x, (int)gridDim.x, (int64_t)blockIdx.x * GR, n, np, dp,
                         red, 0.0, nullptr);
}

// ---- final assembly: mu, var, dmu, dvar, LCB value and gradient ---------------------------
// out layout per pass: mu[16] var[16] val[16] dmu[16*dp] dvar[16*dp] grad[16*dp]
// One workgroup per query point (blockIdx.x = column s, blockIdx.y = pass): every sum below is a fixed-order
// reduction of that column's partials.  S_left = real points from the first pass of this launch on.
__global__ __launch_bounds__(256) void finish_kernel(const double* mu_part, int nblk_k, const double* var_part,
                                                     int nblk_v, const double* g_part, int ngc, double* out, int dp,
                                                     int S_left, double prior_var, double noise_add, double inv_ls2,
                                                     double beta, int with_grad, double* host_out,
                                                     unsigned long long* done_flags, unsigned long long done_value) {
  __shared__ double red[256];
  __shared__ double gs[8][32];
  __shared__ double gtot[2 * 256];
  const int t = threadIdx.x, s = blockIdx.x;
  mu_part += (int64_t)blockIdx.y * PC * nblk_k;
  var_part += (int64_t)blockIdx.y * nblk_v * PC;
  g_part += (int64_t)blockIdx.y * PC * ngc * 2 * dp;
  out += (int64_t)blockIdx.y * (3 * PC + 3 * PC * dp);
  const int S = S_left - (int)blockIdx.y * PC;  // columns >= S are padding
  double* mu = out;
  double* var = out + PC;
  double* val = out + 2 * PC;
  double* dmu = out + 3 * PC;
  double* dvar = dmu + PC * dp;
  double* grad = dvar + PC * dp;
  if (s >= S) {
    if (t == 0) {
      mu[s] = 0;
      var[s] = 0;
      val[s] = 0;
    }
    return;
  }
  double m = 0.0, q = 0.0;
  for (int b = t; b < nblk_k; b += 256) m += mu_part[s * nblk_k + b];
  for (int b = t; b < nblk_v; b += 256) q += var_part[(int64_t)b * PC + s];
  // both sums at once: butterfly inside the wave, the four waves in order (two barriers instead of twenty)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m += __shfl_xor(m, off, 64);
    q += __shfl_xor(q, off, 64);
  }
  if ((t & 63) == 0) {
    red[t >> 6] = m;
    red[4 + (t >> 6)] = q;
  }
  __syncthreads();
  m = ((red[0] + red[1]) + red[2]) + red[3];
  q = ((red[4] + red[5]) + red[6]) + red[7];
  double v = prior_var - q;
  v = v > 1e-15 ? v : 1e-15;  // [GPy-upstream] predict clips the variance at 1e-15
  // copy-free calls: the same values go to pinned host memory straight from the registers (re-reading `out` first cost
  // a store -> load round trip through memory at the very end of every lock-step); every writer is in wave 0
  double* hout = host_out ? host_out + (int64_t)blockIdx.y * (3 * PC + 3 * PC * dp) : nullptr;
  if (t == 0) {
    const double vv = v + noise_add, lc = m - sqrt(beta * v);
    mu[s] = m;
    var[s] = vv;
    val[s] = lc;
    if (hout) {
      hout[s] = m;
      hout[PC + s] = vv;
      hout[2 * PC + s] = lc;
    }
  }
  if (with_grad) {
    // 32 values of the 2 dp gradient sums at a time: thread (value t & 31, slice t >> 5) adds chunks slice, slice + 8,
    // ... in order, then the eight slices are added in order
    const double* gp1 = g_part + (int64_t)s * ngc * 2 * dp;
    for (int a0 = 0; a0 < 2 * dp; a0 += 32) {
      const int idx = a0 + (t & 31), cs = t >> 5;
      double acc = 0.0;
      if (idx < 2 * dp) {
        // sixteen chunk values in flight per round (the loop with one load per iteration waits a memory round trip per
        // chunk: 16 of them at n = 4096), added in chunk order
        for (int c = cs; c < ngc; c += 128) {   // (n = 4096: 128 chunks = ONE round trip for all of a thread's 16 values)
          double gv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) gv[u] = (c + 8 * u < ngc) ? gp1[(int64_t)(c + 8 * u) * 2 * dp + idx] : 0.0;
#pragma unroll
          for (int u = 0; u < 16; ++u) acc += gv[u];
        }
      }
      __syncthreads();
      gs[cs][t & 31] = acc;
      __syncthreads();
      if (t < 32 && idx < 2 * dp) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += gs[k][t];
        gtot[idx] = tot;
      }
    }
    __syncthreads();
    const double sc = sqrt(beta / v);
    for (int a = t; a < dp; a += 256) {
      const double dm = -inv_ls2 * gtot[a];
      const double dv = 2.0 * inv_ls2 * gtot[dp + a];  // -2 * sum u_i dk_i, dk_i = -(k/l^2)(x - X_i)
      const double gr = dm - 0.5 * dv * sc;
      dmu[s * dp + a] = dm;
      dvar[s * dp + a] = dv;
      grad[s * dp + a] = gr;
      if (hout) {
        hout[3 * PC + s * dp + a] = dm;
        hout[3 * PC + PC * dp + s * dp + a] = dv;
        hout[3 * PC + 2 * PC * dp + s * dp + a] = gr;
      }
    }
  }
  if (host_out) {
    // one wave makes the column's results visible to the host and raises the column's flag (the host polls the flags
    // of the columns it asked for); dp <= 64: every store above came from wave 0
    done_flags += (int64_t)blockIdx.y * PC;
    if (dp > 64) {   // (writers beyond wave 0: each makes its own stores visible, then the flag)
      __threadfence_system();
      __syncthreads();
    }
    if (t < 64) {
      __threadfence_system();
      if (t == 0) __hip_atomic_store(done_flags + s, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- MaxVar surface from the assembled prediction (elfi/methods/bo/acquisition.py:392-463) ----
// Variance of the unnormalised approximate posterior at a point with GP mean m, noiseless variance v:
//   value = p^2 W,   W = Phi(z) - Phi(z)^2 - 2 T(z, b),   z = (eps - m) / sqrt(s_n + v),   b = sqrt(s_n / (s_n + 2 v))
// (Phi(z) - 2 T(z, b) is the skew-normal cdf the reference takes from SciPy), p the prior density; and its gradient by
// the chain rule through z and b with dT/dh = -phi(h) (Phi(a h) - 1/2), dT/da = exp(-h^2 (1 + a^2) / 2) / (2 pi (1 + a^2)).
// One thread per query point; reads mu / var / dmu / dvar of the pass from `out`, writes its val / grad slots.
__global__ void maxvar_kernel(double* out, const double* prior_pdf, const double* prior_glog, int64_t S, int d, int dp,
                              double eps, double s2n, int64_t s0) {
  const int q = threadIdx.x;
  out += (int64_t)blockIdx.x * (3 * PC + 3 * PC * dp);
  const int64_t s = s0 + (int64_t)blockIdx.x * PC + q;
  if (q >= PC || s >= S) return;
  const double m = out[q], v = out[PC + q];
  const double* dmu = out + 3 * PC + q * dp;
  const double* dvar = dmu + PC * dp;
  double* grad = out + 3 * PC + 2 * PC * dp + q * dp;
  const double sv = s2n + v, sdev = sqrt(sv), z = (eps - m) / sdev;
  const double sb = s2n + 2.0 * v, b = sqrt(s2n) / sqrt(sb);
  const double Pz = norm_cdf(z), pz = norm_pdf(z);
  const double W = (Pz - Pz * Pz) - 2.0 * owens_t(z, b);
  const double p = prior_pdf[s];
  out[2 * PC + q] = p * p * W;
  const double dT_dh = -pz * (norm_cdf(z * b) - 0.5);
  const double dT_da = exp(-0.5 * z * z * (1.0 + b * b)) / (6.28318530717958647692 * (1.0 + b * b));
  const double dW_dz = (1.0 - 2.0 * Pz) * pz - 2.0 * dT_dh;
  const double dz_dm = -1.0 / sdev, dz_dv = -(eps - m) / (2.0 * sv * sdev), db_dv = -sqrt(s2n) / (sb * sqrt(sb));
  for (int a = 0; a < d; ++a) {
    const double dz = dz_dm * dmu[a] + dz_dv * dvar[a];
    const double dW = dW_dz * dz - 2.0 * dT_da * (db_dv * dvar[a]);
    grad[a] = 2.0 * p * W * (p * prior_glog[s * d + a]) + p * p * dW;
  }
}

// ---- ExpIntVar loss from the posterior covariances (elfi/methods/bo/acquisition.py:795-821) ----
// For candidate s with noiseless variance v_s and covariances c_is to the M integration points (GP mean m_i, noiseless
// variance v_i, weight w_i = omega_i prior_i^2):
//   loss_s = 2 sum_i w_i T(z_i, a_is),   z_i = (eps - m_i) / sqrt(A_i),  A_i = s_n + v_i,  d_is = c_is^2 / (s_n + v_s),
//   a_is = sqrt((A_i - d_is) / (A_i + d_is))
// ((Phi(z_i) - skew-normal cdf) / 2 of the reference is T(z_i, a_is) itself).  One workgroup per candidate, fixed-order sum.
__global__ __launch_bounds__(256) void expintvar_kernel(const double* cov, const double* outs, int64_t outsz,
                                                        const double* w_int, const double* mean_int,
                                                        const double* var_int, int64_t M, int64_t S, double eps,
                                                        double s2n, double* loss) {
  __shared__ double red[256];
  const int64_t s = blockIdx.x;
  const double v_s = outs[(s / PC) * outsz + PC + (s % PC)];   // noiseless variance of the candidate
  const double den = s2n + v_s;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < M; i += 256) {
    const double c = cov[i * S + s];
    const double A = s2n + var_int[i];
    const double dl = c * c / den;
    double r = (A - dl) / (A + dl);
    r = r < 0.0 ? 0.0 : r;
    acc += w_int[i] * owens_t((eps - mean_int[i]) / sqrt(A), sqrt(r));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[s] = 2.0 * red[0];
}

// WL = WT^T in 64 x 64 tiles through LDS (upper tiles of WT -> lower tiles of WL).
__global__ __launch_bounds__(256) void mirror_kernel(const double* WT, double* WL, int64_t lda) {
  __shared__ double tile[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bi > bj) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ty + 4 * r;
    tile[row][tx] = WT[((int64_t)bi * 64 + row) * lda + (int64_t)bj * 64 + tx];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ty + 4 * r;
    WL[((int64_t)bj * 64 + row) * lda + (int64_t)bi * 64 + tx] = tile[tx][row];
  }
}

// Prior covariance between two point sets, k(a, b) = s_f exp(-|a - b|^2 / 2 l^2) + s_b -- what GPy's kern.K(X, X2)
// returns for ELFI's rbf + bias kernel ([GPy-upstream] Stationary._unscaled_dist: (|a|^2 + |b|^2) - 2 a.b clipped at 0,
// exactly 0 on the diagonal when X2 is None).  One thread per entry.
__global__ __launch_bounds__(256) void kernel_matrix_kernel(const double* A, const double* B, int64_t na, int64_t nb, int d,
                                                            double var, double neg_half_inv_ls2, double bias, int same,
                                                            double* out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= na * nb) return;
  const int64_t i = e / nb, j = e - i * nb;
  double a2 = 0.0, b2 = 0.0, dot = 0.0;
  for (int c = 0; c < d; ++c) {
    const double a = A[i * d + c], b = B[j * d + c];
    a2 += a * a;
    b2 += b * b;
    dot += a * b;
  }
  double r2 = (a2 + b2) + (-2.0 * dot);
  r2 = r2 < 0.0 ? 0.0 : r2;
  if (same && i == j) r2 = 0.0;
  out[e] = var * exp(r2 * neg_half_inv_ls2) + bias;
}

// L^-1 (row-wise) for the second triangular product: mirrored from L^-T once per factorisation, on first use
// (hyper-parameter searches factorise many times without predicting).
static int ensure_wl(elfihip_gp* gp) {
  if (gp->wl_valid) return ELFIHIP_OK;
  const unsigned nt = (unsigned)(gp->np / 64);
  hipLaunchKernelGGL(mirror_kernel, dim3(nt, nt), dim3(256), 0, gp->ctx->stream, gp->WT, gp->WL, gp->lda);
  ELFIHIP_TRY(launch_status(gp->ctx, "mirror_kernel"));
  gp->wl_valid = true;
  return ELFIHIP_OK;
}

int ensure_wl_public(elfihip_gp* gp) { return ensure_wl(gp); }

// kstar / finish for `npass` 16-point passes in one launch each, for the dense path (gp_dense.hip)
void launch_kstar_passes(elfihip_gp* gp, const double* xs, const double* xs2, double* kr, double* kbt, double* mu_part,
                         int nblk_k, unsigned npass) {
  const double inv_ls2 = 1.0 / (gp->ls * gp->ls);
  hipLaunchKernelGGL(kstar_kernel, dim3(nblk_k, PC, npass), dim3(256), 0, gp->ctx->stream, gp->X, gp->x2, gp->alpha, xs, xs2,
                     kr, (double*)nullptr, mu_part, gp->n, gp->np, gp->dp, gp->var, -0.5 * inv_ls2, gp->bias,
                     (double*)nullptr, 0, QueryArgs(), kbt);
}

void launch_finish_passes(elfihip_gp* gp, const double* mu_part, int nblk_k, const double* var_part, int nblk_v,
                          const double* g_part, int ngc, double* out, int s_left, int noiseless, double beta, int mode,
                          unsigned npass, double* host_out, unsigned long long* done_flags, unsigned long long done_value) {
  const double inv_ls2 = 1.0 / (gp->ls * gp->ls);
  hipLaunchKernelGGL(finish_kernel, dim3(PC, npass), dim3(256), 0, gp->ctx->stream, mu_part, nblk_k, var_part, nblk_v, g_part,
                     ngc, out, gp->dp, s_left, gp->var + gp->bias, noiseless ? 0.0 : gp->noise, inv_ls2, beta, mode,
                     host_out, done_flags, done_value);
}

// One triangular product for `g` passes: part[pass][kc][i][s] from bin[pass][k][s].
// fused = the reduction (first product) / the gradient sums (second product) by the last workgroup of every row block
// full: ONE product with the symmetric K^-1 (fused form only; xs as for the lower product)
static TriArgs tri_args(const elfihip_gp* gp, const PredictWs& W, bool lower, const double* bin, unsigned g, const double* xs,
                        bool full) {
  TriArgs T;
  T.W = full ? gp->Kinv : (lower ? gp->WL : gp->WT);
  T.bias = gp->bias;
  T.bin = bin;
  T.part = W.part;
  T.lda = gp->lda;
  T.np = gp->np;
  T.nout = gp->np;
  T.nrb = (int)(gp->np / RB);
  T.nkc = W.nkc;
  T.npass = (int)g;
  T.cnt = gp->tri_cnt ? gp->tri_cnt + (lower ? gp->cap / RB : 0) : nullptr;
  T.vout = W.v;
  T.sq_part = W.var_part;
  T.X = gp->X;
  T.alpha = gp->alpha;
  T.xs = xs;
  T.kr = W.kr;
  T.g_part = W.g_part;
  T.n = gp->n;
  T.dp = gp->dp;
  return T;
}

static void launch_tri(const elfihip_gp* gp, const PredictWs& W, bool lower, const double* bin, unsigned g, bool fused = false,
                       const double* xs = nullptr, bool full = false) {
  const TriArgs T = tri_args(gp, W, lower, bin, g, xs, full);
  const dim3 grid((unsigned)T.nrb, (unsigned)W.nkc);
  if (full)
    hipLaunchKernelGGL((tri_apply_kernel<3, true>), dim3((unsigned)tri_tiles(3, T.nrb, T.nkc)), dim3(256), 0, gp->ctx->stream, T);
  else if (fused && lower)
    hipLaunchKernelGGL((tri_apply_kernel<1, true>), dim3((unsigned)tri_tiles(1, T.nrb, T.nkc)), dim3(256), 0, gp->ctx->stream, T);
  else if (fused)
    hipLaunchKernelGGL((tri_apply_kernel<0, true>), dim3((unsigned)tri_tiles(0, T.nrb, T.nkc)), dim3(256), 0, gp->ctx->stream, T);
  else if (lower)
    hipLaunchKernelGGL((tri_apply_kernel<1, false>), grid, dim3(256), 0, gp->ctx->stream, T);
  else
    hipLaunchKernelGGL((tri_apply_kernel<0, false>), grid, dim3(256), 0, gp->ctx->stream, T);
}

// The fused lock-step (four launches instead of six) is the default; elfihip_gp_set_lockstep_form(gp, 1) keeps the
// six-launch form (same numbers for mean / variance, gradient sums in 64-row instead of 32-row chunks).
static bool lockstep_fused(const elfihip_gp* gp) { return gp->lockstep_form != 1; }

// ---- K^-1 for the acquisition lock-step -------------------------------------------------------------------------
// u = K^-1 kb gives the variance (kb . u) and its gradient from ONE product over the n x n symmetric matrix instead of two
// dependent products over the two triangles (what GPy's posterior does with its woodbury_inv, gpy_regression.py:127-140).
// The bytes are the same; the second launch's ramp, round trip and hand-off are not there.  Measured
// (profiles/r04_lockstep_pmc.md; 10 points, d = 10): 61.8 -> 52.3 us per lock-step at n = 4096 (the product: 33 us for
// 134 MB against 21 + 25 for 2 x 67), 44.0 -> 36.0 at n = 2048; value and gradient agree with the triangular form to 1e-13.
// The price: forming K^-1 = L^-T L^-1 after a factorisation costs 0.33 / 0.74 ms at n = 2048 / 4096 -- what 40-80
// lock-steps save -- so the matrix is made only for a factorisation that has already served KINV_AFTER_STEPS lock-steps
// (a factorisation that is extended point by point between hyper-parameter searches serves hundreds; one that is rebuilt for
// every acquisition never pays), and it is then carried through extends by the rank-one bordering below (an extend at
// n = 4048: 63 -> 129 us, at 2000: 43 -> 50).  k(x,x) - kb . u
// cancels with an error of eps cond(K) k(x,x) where the triangular form's sum of squares has eps k(x,x): the form is used
// while the estimate of cond(K) below is <= KINV_MAX_COND (the variance then agrees with the triangular form to 1e-10 k(x,x)).
constexpr int64_t KINV_AFTER_STEPS = 64;
constexpr double KINV_MAX_COND = 1e5;
constexpr double KINV_CHECK_TOL = 1e-9;   // of k(x,x): the first K^-1 lock-step of a factorisation against the triangular form

// The strictly-upper part of K^-1 from the lower one (the gradient kernel writes the lower 128 x 128 tiles), 64 x 64 tiles
// through LDS, in place.
__global__ __launch_bounds__(256) void kinv_mirror_kernel(double* K, int64_t lda) {
  __shared__ double tile[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bi < bj) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ty + 4 * r;
    tile[row][tx] = K[((int64_t)bi * 64 + row) * lda + (int64_t)bj * 64 + tx];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ty + 4 * r;
    if (bi > bj || row < tx) K[((int64_t)bj * 64 + row) * lda + (int64_t)bi * 64 + tx] = tile[tx][row];
  }
}

// K^-1 of the bordered matrix: K^-1 (padded with the new row / column) + w w^T, w = the new column of L^-T = [-u / d ; 1 / d]
// (extend_write_kernel).  Entry (i, j) and (j, i) get the same product, so the matrix stays symmetric to the bit.
__global__ __launch_bounds__(256) void kinv_border_kernel(double* K, int64_t lda, int64_t n, const double* u, const double* red) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > n) return;
  const double d = red[8];
  const double wj = j < n ? -u[j * PC] / d : 1.0 / d;
  const int64_t i0 = (int64_t)blockIdx.y * 16;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int64_t i = i0 + r;
    if (i > n) break;
    const double wi = i < n ? -u[i * PC] / d : 1.0 / d;
    double* e = K + i * lda + j;
    *e = (i < n && j < n) ? *e + wi * wj : wi * wj;
  }
}

// (max L_ii / min L_ii)^2 only bounds cond(K) from BELOW (lambda_max can be n times the largest diagonal entry): the gate
// uses n times that ratio, capped by the bound the hyper-parameters give for free -- Ky = K_psd + s I with a constant
// diagonal v + b + s, so lambda_min >= s and lambda_max <= trace = n (v + b + s).  The result is an ESTIMATE, not a bound
// (lambda_min can lie far below min L_ii^2): it only keeps hopeless matrices from being formed; what makes the K^-1 form
// fail SAFE is the validation of its first lock-step against the triangular products (predict_impl, KINV_CHECK_TOL)
static double kinv_cond_estimate(const elfihip_gp* gp) {
  if (!(gp->diag_min > 0.0)) return __builtin_huge_val();
  const double r = gp->diag_max / gp->diag_min;
  const double s = gp->noise + GP_JITTER + gp->jitter;
  const double upper = (double)gp->n * (gp->var + gp->bias + s) / s;
  const double est = (double)gp->n * r * r;
  return est < upper ? est : upper;
}
static bool kinv_conditioned(const elfihip_gp* gp) { return kinv_cond_estimate(gp) <= KINV_MAX_COND; }

// Called before an LCB lock-step is enqueued: true when this call multiplies with K^-1 (forming it first if it is due).
static int lockstep_kinv(elfihip_gp* gp, bool* use) {
  *use = false;
  if (gp->lockstep_form == 1 || gp->lockstep_form == 2) return ELFIHIP_OK;
  ++gp->lcb_steps;
  if (!kinv_conditioned(gp)) return ELFIHIP_OK;
  if (gp->kinv_bad_full == gp->full_gen) return ELFIHIP_OK;   // this factorisation's K^-1 failed its validation
  if (!gp->kinv_sym) {
    // (a K^-1 the caller formed already -- HipGPRegression after a hyper-parameter search -- is used at once)
    if (gp->lockstep_form == 0 && gp->lcb_steps <= KINV_AFTER_STEPS && !gp->has_kinv) return ELFIHIP_OK;
    if (!gp->has_kinv) ELFIHIP_TRY(form_kinv_impl(gp));
    const unsigned nt = (unsigned)(gp->np / 64);
    hipLaunchKernelGGL(kinv_mirror_kernel, dim3(nt, nt), dim3(256), 0, gp->ctx->stream, gp->Kinv, gp->lda);
    ELFIHIP_TRY(launch_status(gp->ctx, "kinv_mirror_kernel"));
    gp->kinv_sym = true;
    gp->kinv_checked = false;   // a new K^-1: its first lock-step is validated (predict_impl)
  }
  *use = true;
  return ELFIHIP_OK;
}

static int ensure_tri_counters(elfihip_gp* gp) {
  if (gp->tri_cnt) return ELFIHIP_OK;
  const size_t bytes = (size_t)2 * (gp->cap / RB) * sizeof(unsigned);
  ELFIHIP_CHECK_HIP(gp->ctx, hipMalloc(reinterpret_cast<void**>(&gp->tri_cnt), bytes));
  ELFIHIP_CHECK_HIP(gp->ctx, hipMemsetAsync(gp->tri_cnt, 0, bytes, gp->ctx->stream));
  return ELFIHIP_OK;
}

static int ensure_ws(elfihip_gp* gp, PredictWs* W, int64_t npass) {
  elfihip_ctx* ctx = gp->ctx;
  const int64_t np = gp->np;
  const int nb = (int)(np / NB);
  W->nblk_k = (int)((np + 255) / 256);
  W->nkc = (nb + KCH - 1) / KCH;
  // gradient chunks: 64-row workgroups of grad_kernel, or the 32-row blocks of the fused second product
  W->ngc = lockstep_fused(gp) ? (int)(np / RB) : (int)((gp->n + GR - 1) / GR);
  size_t off = 0;
  auto take = [&](size_t doubles) {
    size_t o = off;
    off += (doubles + 15) & ~(size_t)15;
    return o;
  };
  // passes run `group` at a time in one set of launches; the scratch below is per pass of a group
  const size_t per_pass = (size_t)(W->nkc + 4) * np * PC * sizeof(double);
  int64_t group = (int64_t)(((size_t)768 << 20) / per_pass);
  group = group < 1 ? 1 : (group > MAX_GROUP ? MAX_GROUP : group);
  if (group > npass) group = npass;
  W->group = (int)group;
  const size_t g = (size_t)group;
  const size_t o_xs = take((size_t)npass * PC * gp->dp), o_xs2 = take((size_t)npass * PC),
               o_kr = take(g * PC * np), o_kb = take(g * PC * np), o_part = take(g * W->nkc * np * PC), o_v = take(g * np * PC),
               o_u = take(g * np * PC), o_mu = take(g * PC * W->nblk_k), o_var = take(g * (np * PC / 256) * PC + 16),
               o_g = take(g * PC * W->ngc * 2 * gp->dp), o_out = take((size_t)npass * (3 * PC + 3 * PC * gp->dp));
  ELFIHIP_CHECK_HIP(ctx, gp->ws.reserve(off * sizeof(double)));
  double* base = gp->ws.as<double>();
  W->xs = base + o_xs;
  W->xs2 = base + o_xs2;
  W->kr = base + o_kr;
  W->kb = base + o_kb;
  W->part = base + o_part;
  W->v = base + o_v;
  W->u = base + o_u;
  W->mu_part = base + o_mu;
  W->var_part = base + o_var;
  W->g_part = base + o_g;
  W->out = base + o_out;
  return ELFIHIP_OK;
}

// ---- a prediction call in four steps: host preparation, input fill, device enqueue, result read.
// (Replaying the enqueue part from a hipGraph was measured and is slower here: +90 us per launch on
// this stack for a 9-node graph with two copy nodes, against ~25 us of plain launch cost.)
int predict_prepare(elfihip_gp* gp, int64_t S, PredictPlan* P) {
  elfihip_ctx* ctx = gp->ctx;
  if (!gp->factored)
    return fail(ctx, ELFIHIP_ERR_STATE, "GP is not factorised (call elfihip_gp_factorize after changing data)");
  P->npass = (S + PC - 1) / PC;
  PredictWs W;
  ELFIHIP_TRY(ensure_ws(gp, &W, P->npass));
  P->ws = W;
  const int dp = gp->dp;
  P->outsz = (size_t)3 * PC + 3 * PC * dp;
  P->n_in = (size_t)P->npass * PC * dp + (size_t)P->npass * PC;
  P->n_out = (size_t)P->npass * P->outsz;
  // pinned, device-visible staging: [completion flag | query points | results].  Calls of one pass (S <= 16 points
  // of <= 24 dimensions: every step of the acquisition search) skip both copies: the points ride in the kernel
  // arguments, the last kernel writes the results into this buffer and raises the flag the host polls.
  constexpr size_t HDR = (size_t)MAX_GROUP * PC;  // completion flags, one per query column of a group
  if (gp->h_cap < P->n_in + P->n_out + HDR) {
    if (gp->h_stage) ELFIHIP_CHECK_HIP(ctx, hipHostFree(gp->h_stage));
    gp->h_stage = nullptr;
    gp->h_cap = 0;
    const size_t want = 2 * (P->n_in + P->n_out) + 1024;
    ELFIHIP_CHECK_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&gp->h_stage), want * sizeof(double),
                                         hipHostMallocMapped | hipHostMallocCoherent));
    gp->h_cap = want;
    for (size_t f = 0; f < HDR; ++f) reinterpret_cast<unsigned long long*>(gp->h_stage)[f] = 0;
    gp->done_seq = 0;
  }
  // copy-free when the call is one group of passes: points from the kernel arguments (one pass, <= 24 dimensions) or
  // read by the first kernel straight from this buffer, results and flags written by the last kernel
  P->direct = P->npass <= W.group;
  P->by_args = P->direct && P->npass == 1 && gp->dp <= QUERY_ARGS_MAX_DP;
  P->flag = reinterpret_cast<unsigned long long*>(gp->h_stage);
  P->n_flags = (int)S;
  P->hx = gp->h_stage + HDR;
  P->hout = P->hx + P->n_in;
  return ELFIHIP_OK;
}

void predict_fill(const elfihip_gp* gp, const PredictPlan& P, const double* Xs, int64_t S) {
  const int dp = gp->dp, d = gp->d;
  std::fill(P.hx, P.hx + P.n_in, 0.0);
  double* hx2 = P.hx + (size_t)P.npass * PC * dp;
  for (int64_t s = 0; s < S; ++s) {
    double q = 0.0;
    for (int c = 0; c < d; ++c) {
      const double x = Xs[s * d + c];
      P.hx[(size_t)s * dp + c] = x;
      q += x * x;
    }
    hx2[s] = q;
  }
}

// All points go up in one copy; the 16-point passes run `group` at a time inside each launch (the
// group scratch is reused in stream order); all results come down in one copy; no synchronisation here.
// S_active: number of real points (columns beyond it are computed on zero inputs and ignored).
int predict_enqueue(elfihip_gp* gp, const PredictPlan& P, int64_t S_active, int mode, int noiseless, double beta,
                    const MaxVarEpilogue* mv, bool with_kinv) {
  elfihip_ctx* ctx = gp->ctx;
  hipStream_t st = ctx->stream;
  const PredictWs& W = P.ws;
  const int dp = gp->dp;
  const int64_t np = gp->np;
  const double inv_ls2 = 1.0 / (gp->ls * gp->ls);
  // W.xs and W.xs2 are adjacent in the workspace (PC * dp is a multiple of the 16-double granule)
  if (!P.direct) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(W.xs, P.hx, P.n_in * sizeof(double), hipMemcpyHostToDevice, st));
  if (mode == 1) ELFIHIP_TRY(ensure_wl(gp));
  const bool fused = lockstep_fused(gp);
  if (fused) ELFIHIP_TRY(ensure_tri_counters(gp));
  const int rblocks = (int)(np * PC / 256);
  static thread_local QueryArgs qa;  // only filled (and read by the kernel) for single-pass calls
  for (int64_t pass0 = 0; pass0 < P.npass; pass0 += W.group) {
    const unsigned g = (unsigned)((P.npass - pass0) < W.group ? (P.npass - pass0) : W.group);
    int s_left = (int)(S_active - pass0 * PC);
    if (s_left < 0) s_left = 0;
    const double* xs = W.xs + (size_t)pass0 * PC * dp;  // device copy (filled by the upload or by kstar_kernel)
    const double* xs2 = W.xs2 + (size_t)pass0 * PC;
    double* out = W.out + (size_t)pass0 * P.outsz;
    const bool from_host = P.direct && !P.by_args;  // direct calls are a single group: pass0 == 0
    if (P.by_args) {
      for (int s = 0; s < PC; ++s) {
        for (int c = 0; c < QUERY_ARGS_MAX_DP; ++c) qa.x[s][c] = c < dp ? P.hx[(size_t)s * dp + c] : 0.0;
        qa.x2[s] = P.hx[(size_t)PC * dp + s];
      }
    }
    const bool prof = gp->profile && P.npass <= W.group;   // phase timing of single-group calls
    if (prof) prof_mark(gp, 0);
    hipLaunchKernelGGL(kstar_kernel, dim3(W.nblk_k, PC, g), dim3(256), 0, st, gp->X, gp->x2, gp->alpha,
                       from_host ? P.hx : xs, from_host ? P.hx + (size_t)P.npass * PC * dp : xs2, W.kr, W.kb,
                       W.mu_part, gp->n, np, dp, gp->var, -0.5 * inv_ls2, gp->bias,
                       P.direct ? W.xs : (double*)nullptr, P.by_args ? 1 : 0, qa, (double*)nullptr);
    if (prof) prof_mark(gp, 1);
    if (fused && with_kinv && mode == 1) {
      launch_tri(gp, W, true, W.kb, g, true, xs, true);   // u = K^-1 kb, the block sums of kb . u and the gradient sums
      if (prof) {
        prof_mark(gp, 2);
        prof_mark(gp, 3);
      }
    } else if (fused) {
      launch_tri(gp, W, false, W.kb, g, true);
      if (prof) prof_mark(gp, 2);
      if (mode == 1) {
        launch_tri(gp, W, true, W.v, g, true, xs);
        if (prof) prof_mark(gp, 3);
      }
    } else {
      launch_tri(gp, W, false, W.kb, g);
      hipLaunchKernelGGL(tri_reduce_kernel, dim3(rblocks, g), dim3(256), 0, st, W.part, W.v, W.var_part, np, W.nkc, 0,
                         1);
      if (prof) prof_mark(gp, 2);
      if (mode == 1) {
        launch_tri(gp, W, true, W.v, g);
        if (prof) prof_mark(gp, 3);
        hipLaunchKernelGGL(grad_kernel, dim3(W.ngc, g), dim3(256), 0, st, gp->X, gp->alpha, xs, W.kr, W.part, W.nkc,
                           W.g_part, gp->n, np, dp);
      }
    }
    hipLaunchKernelGGL(finish_kernel, dim3(PC, g), dim3(256), 0, st, W.mu_part, W.nblk_k, W.var_part, rblocks, W.g_part,
                       W.ngc, out, dp, s_left, gp->var + gp->bias, noiseless ? 0.0 : gp->noise, inv_ls2, beta, mode,
                       P.direct ? P.hout : (double*)nullptr, P.flag, (unsigned long long)(gp->done_seq + 1));
    if (prof) prof_mark(gp, mode == 1 ? 4 : 3);
    if (mv)
      hipLaunchKernelGGL(maxvar_kernel, dim3(g), dim3(64), 0, st, out, mv->prior_pdf, mv->prior_glog, S_active, gp->d, dp,
                         mv->eps, gp->noise, pass0 * PC);
  }
  if (P.direct)
    ++gp->done_seq;
  else
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(P.hout, W.out, P.n_out * sizeof(double), hipMemcpyDeviceToHost, st));
  return ELFIHIP_OK;
}

// Completion of an enqueued call.  Single-pass calls poll the flag in pinned memory (a few hundred ns after the
// last kernel's store, against the ~10 us wake-up of a stream wait); if it does not arrive within the budget the
// stream wait takes over and reports whatever went wrong.
int predict_wait(elfihip_gp* gp, const PredictPlan& P) {
  elfihip_ctx* ctx = gp->ctx;
  if (P.direct) {
    const unsigned long long want = gp->done_seq;
    const auto t0 = std::chrono::steady_clock::now();
    int s = 0;  // columns 0 .. s-1 have reported
    for (unsigned spin = 0;; ++spin) {
      while (s < P.n_flags && __atomic_load_n(P.flag + s, __ATOMIC_ACQUIRE) == want) ++s;
      if (s == P.n_flags) return ELFIHIP_OK;
      if ((spin & 1023u) == 1023u &&
          std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200))
        break;
    }
  }
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

void predict_read(const elfihip_gp* gp, const PredictPlan& P, int64_t S, double* mu, double* var, double* dmu,
                  double* dvar, double* val, double* grad) {
  const int dp = gp->dp, d = gp->d;
  for (int64_t s = 0; s < S; ++s) {
    const double* o = P.hout + (size_t)(s / PC) * P.outsz;
    const int q = (int)(s % PC);
    if (mu) mu[s] = o[q];
    if (var) var[s] = o[PC + q];
    if (val) val[s] = o[2 * PC + q];
    for (int c = 0; c < d; ++c) {
      if (dmu) dmu[s * d + c] = o[3 * PC + q * dp + c];
      if (dvar) dvar[s * d + c] = o[3 * PC + PC * dp + q * dp + c];
      if (grad) grad[s * d + c] = o[3 * PC + 2 * PC * dp + q * dp + c];
    }
  }
}

// mode: 0 = mean/var only, 1 = + gradients (and LCB)
int predict_impl(elfihip_gp* gp, const double* Xs, int64_t S, int mode, int noiseless, double beta,
                        double* mu, double* var, double* dmu, double* dvar, double* val, double* grad) {
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, S >= 0, "negative S");
  if (S == 0) return ELFIHIP_OK;
  ELFIHIP_REQUIRE(ctx, Xs, "Xs is NULL");
  // many points: the products are bound by the matrix pipes, not by the read of the factor -> dense tiles (gp_dense.hip)
  if (S >= dense_min_points(gp) && gp->dp <= 24)
    return predict_dense_impl(gp, Xs, S, mode, noiseless, beta, mu, var, dmu, dvar, val, grad);
  PredictPlan P;
  ELFIHIP_TRY(predict_prepare(gp, S, &P));
  predict_fill(gp, P, Xs, S);
  // an acquisition lock-step (value AND gradient of the LCB at a handful of points): with K^-1 when that is due
  bool with_kinv = false;
  if (mode == 1 && val && grad && P.npass <= P.ws.group) ELFIHIP_TRY(lockstep_kinv(gp, &with_kinv));
  if (with_kinv && !gp->kinv_checked) {
    // the first K^-1 lock-step of a factorisation: the same points through the triangular products as well.  k(x,x) - kb . u
    // cancels with eps cond(K) k(x,x); the two variances have to agree to KINV_CHECK_TOL k(x,x) or this factorisation keeps
    // the triangular form (two extra lock-steps per factorisation that uses K^-1)
    std::vector<double> vt((size_t)S), vk((size_t)S);
    for (int pass = 0; pass < 2; ++pass) {
      ELFIHIP_TRY(predict_enqueue(gp, P, S, mode, noiseless, beta, nullptr, pass == 1));
      ELFIHIP_TRY(launch_status(ctx, "predict kernels"));
      ELFIHIP_TRY(predict_wait(gp, P));
      predict_read(gp, P, S, nullptr, pass == 0 ? vt.data() : vk.data(), nullptr, nullptr, nullptr, nullptr);
    }
    double worst = 0.0;
    bool finite = true;
    for (int64_t s = 0; s < S; ++s) {
      const double e = std::fabs(vk[(size_t)s] - vt[(size_t)s]);
      finite = finite && std::isfinite(e);
      if (e > worst) worst = e;
    }
    double tol = KINV_CHECK_TOL;
    if (const char* e = std::getenv("ELFIHIP_KINV_CHECK_TOL")) tol = std::atof(e);   // (tests: 0 rejects every K^-1)
    if (finite && worst <= tol * (gp->var + gp->bias) && tol > 0.0) {
      gp->kinv_checked = true;
    } else {
      gp->kinv_bad_full = gp->full_gen;
      gp->kinv_sym = false;     // (extends stop carrying it)
      gp->has_kinv = false;
      with_kinv = false;
    }
  }
  ELFIHIP_TRY(predict_enqueue(gp, P, S, mode, noiseless, beta, nullptr, with_kinv));
  ELFIHIP_TRY(launch_status(ctx, "predict kernels"));
  ELFIHIP_TRY(predict_wait(gp, P));
  if (gp->profile && P.npass <= P.ws.group) {
    // the polled flags precede the last event: wait for it, then file the four phases of this call
    const int last = mode == 1 ? 4 : 3;
    ELFIHIP_CHECK_HIP(ctx, hipEventSynchronize(gp->pev[last]));
    prof_add(gp, ELFIHIP_PHASE_KSTAR, 0, 1);
    prof_add(gp, ELFIHIP_PHASE_TRI_FIRST, 1, 2);
    if (mode == 1) {
      prof_add(gp, ELFIHIP_PHASE_TRI_SECOND, 2, 3);
      prof_add(gp, ELFIHIP_PHASE_GRAD_FINISH, 3, 4);
    } else {
      prof_add(gp, ELFIHIP_PHASE_GRAD_FINISH, 2, 3);
    }
  }
  predict_read(gp, P, S, mu, var, dmu, dvar, val, grad);
  return ELFIHIP_OK;
}


// ---- incremental extension of the factorisation by one evidence point ------------------------
// Adding (x, y) with unchanged hyper-parameters borders Ky by one row/column:
//   k = K(X, x) + s_b,   l = L^-1 k,   d = sqrt(k(x,x) + s_n + jitter - l.l)        new row of L: [l^T, d]
//   u = L^-T l,   new column of L^-T: [-u/d ; 1/d]
//   z_new = (y - l.z)/d,   alpha += (new column of L^-T) z_new,   logdet += 2 log d,   y'K^-1 y += z_new^2
// l and u are the two triangular products the predictor already runs (one column instead of 16), so
// an update is two passes over L^-T (8 n^2 bytes) instead of the 2 n^3/3 flops of a rebuild.
__global__ __launch_bounds__(256) void extend_scalars_kernel(const double* v, const double* z, const double* sq_part,
                                                             int nblk_v, double knn, double ynew, int64_t n,
                                                             double* red, int* info, int pivot_index) {
  __shared__ double s0[256], s1[256];
  double ll = 0.0, lz = 0.0;
  for (int b = threadIdx.x; b < nblk_v; b += 256) ll += sq_part[(int64_t)b * PC];  // column 0 of the per-block sums
  for (int64_t i = threadIdx.x; i < n; i += 256) lz += v[i * PC] * z[i];
  s0[threadIdx.x] = ll;
  s1[threadIdx.x] = lz;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      s0[threadIdx.x] += s0[threadIdx.x + off];
      s1[threadIdx.x] += s1[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double d2 = knn - s0[0];
    if (!(d2 > 0.0)) atomicCAS(info, 0, pivot_index);
    const double d = sqrt(d2 > 0.0 ? d2 : 1.0);
    red[8] = d;
    red[9] = (ynew - s1[0]) / d;  // z_new
  }
}

__global__ __launch_bounds__(256) void extend_write_kernel(const double* v, const double* u, const double* red,
                                                           double* A, double* WT, double* WL, double* alpha, int64_t lda,
                                                           int64_t n, int64_t np, const double* xs, double x2new,
                                                           double ynew, double* X, double* x2, double* y, int dp,
                                                           const int* info, double* host_report,
                                                           unsigned long long* done_flag, unsigned long long done_value) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double d = red[8], zn = red[9];
  if (j == n) {
    // the evidence arrays themselves (row n): padded x, |x|^2, y -- and the report the host waits for
    for (int c = 0; c < dp; ++c) X[n * dp + c] = xs[c];
    x2[n] = x2new;
    y[n] = ynew;
    host_report[0] = d;
    host_report[1] = zn;
    host_report[2] = (double)*info;
    __threadfence_system();
    __hip_atomic_store(done_flag, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (j < n) {
    A[n * lda + j] = v[j * PC];              // new row of L
    const double w = -u[j * PC] / d;         // new column of L^-T
    WT[j * lda + n] = w;
    WL[n * lda + j] = w;                     // ... and its mirror image, row n of L^-1
    alpha[j] += w * zn;
  } else if (j == n) {
    A[n * lda + n] = d;
    WT[n * lda + n] = 1.0 / d;
    WL[n * lda + n] = 1.0 / d;
    alpha[n] = zn / d;
    A[np * lda + n] = zn;                    // z = L^-1 y lives in row np of A
  }
}

static int extend_one(elfihip_gp* gp, const double* x, double ynew) {
  elfihip_ctx* ctx = gp->ctx;
  hipStream_t st = ctx->stream;
  PredictWs W;
  ELFIHIP_TRY(ensure_ws(gp, &W, 1));
  const int dp = gp->dp, d = gp->d;
  const int64_t np = gp->np, n = gp->n;
  const double inv_ls2 = 1.0 / (gp->ls * gp->ls);
  // No copy in either direction for d <= 24: the point rides in the first kernel's arguments (which leaves the device
  // copy the last kernel files into the evidence arrays), and the last kernel reports (d, z_new, info) into pinned
  // memory and raises the flag the host polls.  Wider points are uploaded once.
  PredictPlan P;
  ELFIHIP_TRY(predict_prepare(gp, 1, &P));
  double q = 0.0;
  for (int c = 0; c < d; ++c) q += x[c] * x[c];
  const bool by_args = dp <= QUERY_ARGS_MAX_DP;
  static thread_local QueryArgs qa;
  if (by_args) {
    for (int s_ = 0; s_ < PC; ++s_) {
      for (int c = 0; c < QUERY_ARGS_MAX_DP; ++c) qa.x[s_][c] = (s_ == 0 && c < d) ? x[c] : 0.0;
      qa.x2[s_] = s_ == 0 ? q : 0.0;
    }
  } else {
    predict_fill(gp, P, x, 1);
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(W.xs, P.hx, P.n_in * sizeof(double), hipMemcpyHostToDevice, st));
  }
  ELFIHIP_TRY(ensure_wl(gp));
  hipLaunchKernelGGL(kstar_kernel, dim3(W.nblk_k, PC), dim3(256), 0, st, gp->X, gp->x2, gp->alpha, W.xs, W.xs2, W.kr,
                     W.kb, W.mu_part, n, np, dp, gp->var, -0.5 * inv_ls2, gp->bias, by_args ? W.xs : (double*)nullptr,
                     by_args ? 1 : 0, qa, (double*)nullptr);
  const int rblocks = (int)(np * PC / 256);
  launch_tri(gp, W, false, W.kb, 1);
  hipLaunchKernelGGL(tri_reduce_kernel, dim3(rblocks), dim3(256), 0, st, W.part, W.v, W.var_part, np, W.nkc, 0, 1);
  const double* z = gp->A + np * gp->lda;
  hipLaunchKernelGGL(extend_scalars_kernel, dim3(1), dim3(256), 0, st, W.v, z, W.var_part, rblocks,
                     gp->var + gp->bias + gp->noise + GP_JITTER, ynew, n, gp->red, gp->info, (int)n + 1);
  launch_tri(gp, W, true, W.v, 1);
  hipLaunchKernelGGL(tri_reduce_kernel, dim3(rblocks), dim3(256), 0, st, W.part, W.u, (double*)nullptr, np, W.nkc, 1, 0);
  hipLaunchKernelGGL(extend_write_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, W.v, W.u, gp->red,
                     gp->A, gp->WT, gp->WL, gp->alpha, gp->lda, n, np, W.xs, q, ynew, gp->X, gp->x2, gp->y, dp,
                     gp->info, P.hout, P.flag, (unsigned long long)(gp->done_seq + 1));
  ++gp->done_seq;
  // K^-1 in use by the lock-step: bordered with the same u and d (W.u and gp->red stay as they are until the next call)
  const bool carry_kinv = gp->kinv_sym;
  if (carry_kinv)
    hipLaunchKernelGGL(kinv_border_kernel, dim3((unsigned)((n + 1 + 255) / 256), (unsigned)((n + 1 + 15) / 16)), dim3(256), 0,
                       st, gp->Kinv, gp->lda, n, W.u, gp->red);
  ELFIHIP_TRY(launch_status(ctx, "extend kernels"));
  P.direct = true;
  P.n_flags = 1;
  ELFIHIP_TRY(predict_wait(gp, P));
  const double sc[2] = {P.hout[0], P.hout[1]};
  const int info = (int)P.hout[2];
  if (info != 0) {
    gp->factored = false;
    return fail(ctx, ELFIHIP_ERR_NOT_PD, "covariance matrix is not positive definite (pivot %d <= 0)", info);
  }
  gp->logdet += 2.0 * log(sc[0]);
  gp->yKy += sc[1] * sc[1];
  gp->diag_min = sc[0] < gp->diag_min ? sc[0] : gp->diag_min;
  gp->diag_max = sc[0] > gp->diag_max ? sc[0] : gp->diag_max;
  gp->n = n + 1;
  gp->has_kinv = carry_kinv;   // (the lower tiles the gradient kernel would write are part of the bordered matrix)
  gp->kinv_sym = carry_kinv;
  ++gp->fact_gen;
  return ELFIHIP_OK;
}

// ---- posterior covariance between a fixed point set and query points (ExpIntVar) ---------------------
// cov(p_i, q_s) = k(p_i, q_s) - k(p_i, X) K^-1 k(X, q_s) = k(p_i, q_s) - V_P[:, i] . v_s,   V = L^-1 k(X, .)
// (what ExpIntVar.evaluate builds with cho_solve per call, elfi/methods/bo/acquisition.py:800-808).  V_P is made
// once per point set by the batched first triangular product and kept k-major, so the per-query part is the
// same streaming kernel as the predictor's products in its dense mode.
__global__ __launch_bounds__(256) void scatter_v_kernel(const double* v, double* VP, int64_t ldvp, int64_t np,
                                                        int64_t col0) {
  // VP[k][col0 + 16 pass + s] = v[pass][k][s]
  v += (int64_t)blockIdx.y * np * PC;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < np * PC) VP[(e / PC) * ldvp + col0 + (int64_t)blockIdx.y * PC + (e % PC)] = v[e];
}

__global__ __launch_bounds__(256) void cross_finish_kernel(const double* Pint, const double* xs, const double* dot,
                                                           double* cov, int64_t M, int64_t m_pad, int dp, int64_t S,
                                                           int64_t s0, double var, double neg_half_inv_ls2,
                                                           double bias) {
  // one thread per (point i, query column s) of pass blockIdx.y: cov[i][s0 + 16 pass + s]
  xs += (int64_t)blockIdx.y * PC * dp;
  dot += (int64_t)blockIdx.y * m_pad * PC;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = e / PC;
  const int sq = (int)(e % PC);
  const int64_t col = s0 + (int64_t)blockIdx.y * PC + sq;
  if (i >= M || col >= S) return;
  double r2 = 0.0;
  for (int c = 0; c < dp; ++c) {
    const double t = Pint[i * dp + c] - xs[sq * dp + c];
    r2 += t * t;
  }
  cov[i * S + col] = (var * exp(r2 * neg_half_inv_ls2) + bias) - dot[i * PC + sq];
}

// first triangular product for `g` passes whose points are already in W.xs / W.xs2: W.v, W.var_part, W.mu_part
static void enqueue_v(elfihip_gp* gp, const PredictWs& W, int64_t pass0, unsigned g) {
  hipStream_t st = gp->ctx->stream;
  const int dp = gp->dp;
  const int64_t np = gp->np;
  const double inv_ls2 = 1.0 / (gp->ls * gp->ls);
  hipLaunchKernelGGL(kstar_kernel, dim3(W.nblk_k, PC, g), dim3(256), 0, st, gp->X, gp->x2, gp->alpha,
                     W.xs + (size_t)pass0 * PC * dp, W.xs2 + (size_t)pass0 * PC, W.kr, W.kb, W.mu_part, gp->n, np, dp,
                     gp->var, -0.5 * inv_ls2, gp->bias, (double*)nullptr, 0, QueryArgs(), (double*)nullptr);
  launch_tri(gp, W, false, W.kb, g);
  hipLaunchKernelGGL(tri_reduce_kernel, dim3((unsigned)(np * PC / 256), g), dim3(256), 0, st, W.part, W.v, W.var_part,
                     np, W.nkc, 0, 1);
}

static int upload_points(elfihip_gp* gp, const double* Xs, int64_t S, PredictPlan* P) {
  ELFIHIP_TRY(predict_prepare(gp, S, P));
  P->direct = false;
  predict_fill(gp, *P, Xs, S);
  ELFIHIP_CHECK_HIP(gp->ctx, hipMemcpyAsync(P->ws.xs, P->hx, P->n_in * sizeof(double), hipMemcpyHostToDevice,
                                            gp->ctx->stream));
  return ELFIHIP_OK;
}

static int set_integration_points_impl(elfihip_gp* gp, const double* Pts, int64_t M) {
  elfihip_ctx* ctx = gp->ctx;
  hipStream_t st = ctx->stream;
  ELFIHIP_REQUIRE(ctx, M >= 1 && Pts, "bad arguments");
  const int64_t np = gp->np, m_pad = round_up(M, RB);
  const int dp = gp->dp;
  if (gp->VP) ELFIHIP_CHECK_HIP(ctx, hipFree(gp->VP));
  if (gp->Pint) ELFIHIP_CHECK_HIP(ctx, hipFree(gp->Pint));
  gp->VP = gp->Pint = nullptr;
  gp->n_int = 0;
  ELFIHIP_CHECK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&gp->VP), (size_t)np * m_pad * sizeof(double)));
  ELFIHIP_CHECK_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&gp->Pint), (size_t)m_pad * dp * sizeof(double)));
  ELFIHIP_CHECK_HIP(ctx, hipMemsetAsync(gp->VP, 0, (size_t)np * m_pad * sizeof(double), st));
  ELFIHIP_CHECK_HIP(ctx, hipMemsetAsync(gp->Pint, 0, (size_t)m_pad * dp * sizeof(double), st));
  PredictPlan P;
  ELFIHIP_TRY(upload_points(gp, Pts, M, &P));
  // the padded points are exactly the rows the upload staged ([pass][16][dp], zero padded)
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(gp->Pint, P.ws.xs, (size_t)M * dp * sizeof(double), hipMemcpyDeviceToDevice, st));
  const PredictWs& W = P.ws;
  for (int64_t pass0 = 0; pass0 < P.npass; pass0 += W.group) {
    const unsigned g = (unsigned)((P.npass - pass0) < W.group ? (P.npass - pass0) : W.group);
    enqueue_v(gp, W, pass0, g);
    // columns beyond m_pad (the last pass of a point count that is not a multiple of 32 rounds up to 16) stay inside
    // the allocation: m_pad is a multiple of 32 >= 16 * npass
    hipLaunchKernelGGL(scatter_v_kernel, dim3((unsigned)(np * PC / 256), g), dim3(256), 0, st, W.v, gp->VP, m_pad, np,
                       pass0 * PC);
  }
  ELFIHIP_TRY(launch_status(ctx, "integration points"));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  gp->n_int = M;
  gp->m_pad = m_pad;
  gp->vp_gen = gp->fact_gen;
  return ELFIHIP_OK;
}

static int cross_cov_impl(elfihip_gp* gp, const double* Q, int64_t S, double* cov, double* var_q,
                          const ExpIntVarArgs* ei = nullptr, double* loss = nullptr) {
  elfihip_ctx* ctx = gp->ctx;
  hipStream_t st = ctx->stream;
  ELFIHIP_REQUIRE(ctx, S >= 1 && Q && (cov || loss), "bad arguments");
  if (!(gp->n_int > 0 && gp->vp_gen == gp->fact_gen))
    return fail(ctx, ELFIHIP_ERR_STATE,
                "no integration points for the current factorisation (call elfihip_gp_set_integration_points)");
  const int64_t np = gp->np, M = gp->n_int, m_pad = gp->m_pad;
  const int dp = gp->dp;
  PredictPlan P;
  ELFIHIP_TRY(upload_points(gp, Q, S, &P));
  const PredictWs& W = P.ws;
  // scratch of the dense product: partials [g][nkc][m_pad][16], result [g][m_pad][16], covariance (M, S)
  const size_t n_part = (size_t)W.group * W.nkc * m_pad * PC, n_dot = (size_t)W.group * m_pad * PC;
  ELFIHIP_CHECK_HIP(ctx, gp->ws2.reserve((n_part + n_dot + (size_t)M * S) * sizeof(double)));
  double* part2 = gp->ws2.as<double>();
  double* dot = part2 + n_part;
  double* cov_dev = dot + n_dot;
  const double inv_ls2 = 1.0 / (gp->ls * gp->ls);
  const int rblocks = (int)(np * PC / 256);
  for (int64_t pass0 = 0; pass0 < P.npass; pass0 += W.group) {
    const unsigned g = (unsigned)((P.npass - pass0) < W.group ? (P.npass - pass0) : W.group);
    enqueue_v(gp, W, pass0, g);
    int s_left = (int)(S - pass0 * PC);
    hipLaunchKernelGGL(finish_kernel, dim3(PC, g), dim3(256), 0, st, W.mu_part, W.nblk_k, W.var_part, rblocks, W.g_part,
                       W.ngc, W.out + (size_t)pass0 * P.outsz, dp, s_left, gp->var + gp->bias, 0.0, inv_ls2, 0.0, 0,
                       (double*)nullptr, (unsigned long long*)nullptr, 0ull);
    TriArgs T;
    T.W = gp->VP;
    T.bin = W.v;
    T.part = part2;
    T.lda = m_pad;
    T.np = np;
    T.nout = m_pad;
    T.nrb = (int)(m_pad / RB);
    T.nkc = W.nkc;
    T.npass = (int)g;
    hipLaunchKernelGGL((tri_apply_kernel<2, false>), dim3((unsigned)T.nrb, (unsigned)W.nkc), dim3(256), 0, st, T);
    hipLaunchKernelGGL(tri_reduce_kernel, dim3((unsigned)(m_pad * PC / 256), g), dim3(256), 0, st, part2, dot,
                       (double*)nullptr, m_pad, W.nkc, 2, 0);
    hipLaunchKernelGGL(cross_finish_kernel, dim3((unsigned)((M * PC + 255) / 256), g), dim3(256), 0, st, gp->Pint,
                       W.xs + (size_t)pass0 * PC * dp, dot, cov_dev, M, m_pad, dp, S, pass0 * PC, gp->var,
                       -0.5 * inv_ls2, gp->bias);
  }
  ELFIHIP_TRY(launch_status(ctx, "cross covariance"));
  std::vector<double> stage;
  if (ei) {
    // weights / means / variances of the integration points go up in one copy, the S losses come down
    stage.resize((size_t)3 * M);
    std::copy(ei->w_int, ei->w_int + M, stage.begin());
    std::copy(ei->mean_int, ei->mean_int + M, stage.begin() + M);
    std::copy(ei->var_int, ei->var_int + M, stage.begin() + 2 * M);
    ELFIHIP_CHECK_HIP(ctx, ctx->par.reserve(((size_t)3 * M + (size_t)S) * sizeof(double)));
    double* dpar = ctx->par.as<double>();
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dpar, stage.data(), (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(expintvar_kernel, dim3((unsigned)S), dim3(256), 0, st, cov_dev, W.out, (int64_t)P.outsz, dpar,
                       dpar + M, dpar + 2 * M, M, S, ei->eps, gp->noise, dpar + 3 * M);
    ELFIHIP_TRY(launch_status(ctx, "expintvar_kernel"));
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(loss, dpar + 3 * M, (size_t)S * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  if (cov)
    ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(cov, cov_dev, (size_t)M * S * sizeof(double), hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(P.hout, W.out, P.n_out * sizeof(double), hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  if (var_q) predict_read(gp, P, S, nullptr, var_q, nullptr, nullptr, nullptr, nullptr);
  return ELFIHIP_OK;
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

#ifdef ELFIHIP_TRI_STAMP
int elfihip_debug_tri_stamps(unsigned long long* out, int clear) {
  if (clear) {
    static unsigned long long zero[8192 * 8];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(elfihip::g_tri_stamp), zero, sizeof(zero));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(elfihip::g_tri_stamp), sizeof(unsigned long long) * 8192 * 8);
}
#endif

int elfihip_gp_lockstep_info(const elfihip_gp* gp, int* kinv_in_use, int64_t* steps, double* cond_estimate) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  if (kinv_in_use) *kinv_in_use = gp->factored && gp->kinv_sym ? 1 : 0;
  if (steps) *steps = gp->lcb_steps;
  if (cond_estimate) *cond_estimate = kinv_cond_estimate(gp);
  return ELFIHIP_OK;
}

int elfihip_gp_set_lockstep_form(elfihip_gp* gp, int form) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, form >= 0 && form <= 3,
                  "form must be 0 (fused epilogues; K^-1 product once it pays), 1 (six launches), 2 (fused, triangular "
                  "products only) or 3 (K^-1 product from the first acquisition lock-step on)");
  gp->lockstep_form = form;
  return ELFIHIP_OK;
}

int elfihip_gp_predict(elfihip_gp* gp, const double* Xs, int64_t S, int noiseless, double* mu, double* var) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  DeviceGuard g(gp->ctx->device);
  return predict_impl(gp, Xs, S, 0, noiseless, 0.0, mu, var, nullptr, nullptr, nullptr, nullptr);
}

int elfihip_gp_predict_grad(elfihip_gp* gp, const double* Xs, int64_t S, double* mu, double* var, double* dmu,
                            double* dvar) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  DeviceGuard g(gp->ctx->device);
  return predict_impl(gp, Xs, S, 1, 1, 0.0, mu, var, dmu, dvar, nullptr, nullptr);
}

int elfihip_gp_lcb(elfihip_gp* gp, const double* Xs, int64_t S, double beta, double* val, double* grad) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, beta >= 0, "beta must be non-negative");
  DeviceGuard g(gp->ctx->device);
  return predict_impl(gp, Xs, S, grad ? 1 : 0, 1, beta, nullptr, nullptr, nullptr, nullptr, val, grad);
}

int elfihip_gp_maxvar(elfihip_gp* gp, const double* Xs, int64_t S, double eps, const double* prior_pdf,
                      const double* prior_grad_logpdf, double* val, double* grad) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, S >= 0 && (S == 0 || (Xs && prior_pdf && prior_grad_logpdf && val && grad)), "bad arguments");
  if (S == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  PredictPlan P;
  ELFIHIP_TRY(predict_prepare(gp, S, &P));
  P.direct = false;   // staged call: the epilogue kernel runs between the assembly and the download
  P.by_args = false;
  predict_fill(gp, P, Xs, S);
  const size_t npr = (size_t)S * (1 + (size_t)gp->d);
  ELFIHIP_CHECK_HIP(ctx, ctx->par.reserve(npr * sizeof(double)));
  double* dpr = ctx->par.as<double>();
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dpr, prior_pdf, (size_t)S * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dpr + S, prior_grad_logpdf, (size_t)S * gp->d * sizeof(double),
                                        hipMemcpyHostToDevice, ctx->stream));
  MaxVarEpilogue mv;
  mv.eps = eps;
  mv.prior_pdf = dpr;
  mv.prior_glog = dpr + S;
  ELFIHIP_TRY(predict_enqueue(gp, P, S, 1, 1, 0.0, &mv, false));
  ELFIHIP_TRY(launch_status(ctx, "maxvar kernels"));
  ELFIHIP_TRY(predict_wait(gp, P));
  predict_read(gp, P, S, nullptr, nullptr, nullptr, nullptr, val, grad);
  return ELFIHIP_OK;
}

int elfihip_gp_expintvar(elfihip_gp* gp, const double* Q, int64_t S, double eps, const double* w_int,
                         const double* mean_int, const double* var_int, double* loss) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  if (!gp->factored)
    return fail(gp->ctx, ELFIHIP_ERR_STATE, "GP is not factorised (call elfihip_gp_factorize first)");
  ELFIHIP_REQUIRE(gp->ctx, S >= 1 && Q && w_int && mean_int && var_int && loss, "bad arguments");
  DeviceGuard g(gp->ctx->device);
  ExpIntVarArgs ei;
  ei.eps = eps;
  ei.w_int = w_int;
  ei.mean_int = mean_int;
  ei.var_int = var_int;
  return cross_cov_impl(gp, Q, S, nullptr, nullptr, &ei, loss);
}

int elfihip_gp_extend(elfihip_gp* gp, const double* X_new, const double* y_new, int64_t k, double* log_marginal) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, k >= 0 && (k == 0 || (X_new && y_new)), "bad arguments");
  ELFIHIP_REQUIRE(ctx, gp->n + k <= gp->cap, "evidence count %lld exceeds the GP capacity %lld",
                  (long long)(gp->n + k), (long long)gp->cap);
  DeviceGuard g(ctx->device);
  int64_t done = 0;
  // bordering works inside the current padded size; crossing a 128 boundary (or an unfactorised
  // GP) takes the ordinary path: append the rest and rebuild
  // ... and so does a factor that carries jitchol jitter (GPy rebuilds on every update: the plain Cholesky gets its
  // chance again) and a bordered pivot that is not positive (the rebuild below then walks the jitter ladder)
  while (done < k && gp->factored && gp->n > 0 && gp->n < gp->np && gp->jitter == 0.0) {
    const int rc = extend_one(gp, X_new + done * gp->d, y_new[done]);
    if (rc == ELFIHIP_ERR_NOT_PD) break;
    ELFIHIP_TRY(rc);
    ++done;
  }
  if (done < k) {
    // (same hyper-parameters, evidence only appended: the rebuild starts at the rung the current factor needed)
    const int rung = (gp->factored && gp->jitter > 0.0) ? gp->jitter_tries : 0;
    ELFIHIP_TRY(elfihip_gp_append(gp, X_new + done * gp->d, y_new + done, k - done));
    gp->jit_start = rung;
    ELFIHIP_TRY(elfihip_gp_factorize(gp, nullptr));
  }
  if (log_marginal)
    *log_marginal = 0.5 * (-(double)gp->n * 1.8378770664093453 /* log(2 pi) */ - gp->logdet - gp->yKy);
  return ELFIHIP_OK;
}

int elfihip_gp_set_integration_points(elfihip_gp* gp, const double* P, int64_t M) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  if (!gp->factored)
    return fail(gp->ctx, ELFIHIP_ERR_STATE, "GP is not factorised (call elfihip_gp_factorize first)");
  DeviceGuard g(gp->ctx->device);
  return set_integration_points_impl(gp, P, M);
}

int elfihip_gp_cross_cov(elfihip_gp* gp, const double* Q, int64_t S, double* cov, double* var_q) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  if (!gp->factored)
    return fail(gp->ctx, ELFIHIP_ERR_STATE, "GP is not factorised (call elfihip_gp_factorize first)");
  DeviceGuard g(gp->ctx->device);
  return cross_cov_impl(gp, Q, S, cov, var_q);
}

int elfihip_gp_kernel_matrix(elfihip_gp* gp, const double* A, int64_t na, const double* B, int64_t nb, double* out) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, na >= 0 && nb >= 0 && (na == 0 || A) && (na * (B ? nb : na) == 0 || out), "bad arguments");
  const bool same = B == nullptr;
  if (same) nb = na;
  if (na == 0 || nb == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  const int d = gp->d;
  const size_t n_in = (size_t)(na + (same ? 0 : nb)) * d, n_out = (size_t)na * nb;
  ELFIHIP_CHECK_HIP(ctx, ctx->in.reserve(n_in * sizeof(double)));
  ELFIHIP_CHECK_HIP(ctx, ctx->out.reserve(n_out * sizeof(double)));
  double* dA = ctx->in.as<double>();
  double* dB = same ? dA : dA + (size_t)na * d;
  hipStream_t st = ctx->stream;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dA, A, (size_t)na * d * sizeof(double), hipMemcpyHostToDevice, st));
  if (!same) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(dB, B, (size_t)nb * d * sizeof(double), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(kernel_matrix_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, dA, dB, na, nb, d, gp->var,
                     -0.5 / (gp->ls * gp->ls), gp->bias, same ? 1 : 0, ctx->out.as<double>());
  ELFIHIP_TRY(launch_status(ctx, "kernel_matrix_kernel"));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->out.p, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  return ELFIHIP_OK;
}

}  // extern "C"
