// GP regression fit on gfx950: Gram matrix, blocked Cholesky, L^-T and K^-1 y in one sweep.
//
// Replaces what GPy does every time reference ELFI calls GPyRegression.update
// (elfi/methods/bo/gpy_regression.py:286-315 rebuilds GPy.models.GPRegression; GPy then
// runs ExactGaussianInference [GPy-upstream]): K = s_f exp(-r^2 / 2 l^2) + s_b,
// Ky = K + (s_n + 1e-8) I, L = chol(Ky), K^-1 (dpotri), alpha = K^-1 y, log-marginal.
//
// One right-looking sweep over 128-wide block columns produces everything:
//   * the working matrix is [Ky ; y^T ; I]: eliminating block column k of Ky applies the same
//     column operations to the appended rows, so row y^T turns into z = L^-1 y and the
//     identity turns into L^-T (upper triangular) -- no separate triangular inversion
//     and no triangular solves, whose 32-step dependency chain is what makes the
//     reference's per-point predict an O(n^2) BLAS-2 affair.
//   * per block column: (1) potf2_tiles: one workgroup factors the 128x128 diagonal block and
//     its inverse, every 16x16 tile in registers; (2) trsm: every row block below / above multiplies its panel by
//     W11^T on the matrix cores (a 128x128x128 GEMM per workgroup); (3) update: one launch
//     of 128x128 f64-MFMA tiles does the SYRK on the trailing lower triangle AND the GEMM
//     on the y row and on the growing L^-T rows.
//   * alpha = L^-T z is a triangular GEMV; logdet from the diagonal.
// Flops: n^3/3 (Cholesky) + n^3/3 (L^-T) on v_mfma_f64_16x16x4_f64.
#include "gp.hpp"
#include "mfma_f64.hpp"
#include "sweep_sched.hpp"

#include <map>
#include <memory>
#include <mutex>

namespace elfihip {

// ------------------------------------------------------------------------- Gram matrix
// K[i][j] for a 64x64 tile (lower tile pairs only), X.X^T on the matrix cores (k = padded d),
// exp epilogue.  r^2 follows [GPy-upstream] Stationary._unscaled_dist: (|xi|^2 + |xj|^2) - 2 xi.xj,
// clipped at 0, exactly 0 on the diagonal.  Rows/cols >= n are the identity (padding).
// The y row block (rows np..np+127) is written by the tiles of the last tile row.
struct GramArgs {
  const double* X;
  const double* x2;
  const double* y;
  double* A;
  int64_t lda, n, np;
  int dp;
  double var, neg_half_inv_ls2, bias, diag_add;
  int* info;   // the factorisation's pivot report and the fused sweep's arrival counters behind it: cleared here, by
  int ninfo;   // the first kernel of a rebuild
};

__global__ __launch_bounds__(256) void gram_kernel(GramArgs G) {
  extern __shared__ __align__(16) double sm[];
  const int pitch = G.dp + 1;  // odd pitch in doubles: bank-conflict free for column walks
  double* Xi = sm;
  double* Xj = sm + 64 * pitch;
  // decode lower-triangular tile pair (ti >= tj) from the linear block index
  const int64_t nt = G.np / 64;
  int64_t b = blockIdx.x;
  if (b == 0)
    for (int i = threadIdx.x; i < G.ninfo; i += 256) G.info[i] = 0;
  int64_t ti = (int64_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > b) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
  const int64_t tj = b - ti * (ti + 1) / 2;
  if (ti >= nt) return;
  const int64_t i0 = ti * 64, j0 = tj * 64;
  for (int e = threadIdx.x; e < 64 * G.dp; e += 256) {
    int r = e / G.dp, c = e - r * G.dp;
    Xi[r * pitch + c] = G.X[(i0 + r) * G.dp + c];
    Xj[r * pitch + c] = G.X[(j0 + r) * G.dp + c];
  }
  __syncthreads();
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = w >> 1, wc = w & 1;
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = (v4d){0, 0, 0, 0};
  for (int k0 = 0; k0 < G.dp; k0 += 4) {
    double av[2], bv[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      av[a] = Xi[(wr * 32 + a * 16 + (l & 15)) * pitch + k0 + (l >> 4)];
      bv[a] = Xj[(wc * 32 + a * 16 + (l & 15)) * pitch + k0 + (l >> 4)];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + wr * 32 + a * 16 + (l >> 4) + 4 * r;
        const int64_t j = j0 + wc * 32 + c * 16 + (l & 15);
        double v;
        if (i >= G.n || j >= G.n) {
          v = (i == j) ? 1.0 : 0.0;
        } else {
          double r2 = (G.x2[i] + G.x2[j]) + (-2.0 * acc[a][c][r]);
          r2 = r2 < 0.0 ? 0.0 : r2;   // (a NaN stays a NaN: it must reach the pivot test, not turn into distance 0)
          if (i == j) r2 = 0.0;
          v = G.var * exp(r2 * G.neg_half_inv_ls2) + G.bias;
          if (i == j) v += G.diag_add;
        }
        G.A[i * G.lda + j] = v;
      }
  // y row block: [y^T ; 0] under block column tj (written once, by the tiles of the diagonal)
  if (ti == tj) {
    for (int e = threadIdx.x; e < NB * 64; e += 256) {
      int r = e >> 6, c = e & 63;
      int64_t j = j0 + c;
      G.A[(G.np + r) * G.lda + j] = (r == 0 && j < G.n) ? G.y[j] : 0.0;
    }
  }
}

__global__ void x2_kernel(const double* X, double* x2, int64_t n, int64_t cap, int dp) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  double s = 0.0;
  if (i < n)
    for (int c = 0; c < dp; ++c) s += X[i * dp + c] * X[i * dp + c];
  x2[i] = s;
}

// ------------------------------------------------------------- diagonal block: L11 and L11^-1
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// s = sqrt(p), y = 1/sqrt(p) from v_rsq_f64 + Newton (no division, no libm call on the
// pivot dependency chain); both within an ulp or two.
__device__ __forceinline__ void sqrt_rsqrt(double p, double& s, double& y) {
  y = __builtin_amdgcn_rsq(p);
  const double h = 0.5 * p;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  s = p * y;
  const double e = fma(-s, s, p);
  s = fma(0.5 * e, y, s);
  const double e2 = fma(-s, y, 1.0);
  y = fma(e2, y, y);
}

// The 128 x 128 diagonal block of [Ky ; I] -> [L11 ; L11^-T], 16 columns ("panel") at a time, with every tile in
// REGISTERS.
//
// Wave R (of eight "tile waves") owns row block R of both halves: slot C <= R holds tile (R, C) of the block of Ky, slot
// C > R tile (R, C) of the identity rows (zero until panel R has been eliminated; tile (R, R) of the identity never
// exists as data).  A tile X lives TRANSPOSED in the f64 16x16x4 MFMA accumulator layout (lane (lr, lc), register r:
// X^T[4 r + lr][lc] = X[lc][4 r + lr]) -- which is at the same time the layout of MFMA operand B (k slice r) of X^T and
// of operand A (k slice r) of X.  Hence, with M = L_pp^-1 (16 x 16) of the panel's diagonal tile:
//   * panel solve  P = X L_pp^-T,  P^T = M X^T:  four MFMAs with A = M (from LDS) and B = the tile's registers as they
//     are; the result P^T is again in the accumulator layout;
//   * rank-16 update  X(R, C) -= P(R) P(C)^T,  X^T -= P(C) P(R)^T:  A = -P(C)^T registers of wave C (through LDS, already
//     negated), B = P(R)^T registers of the wave itself -- no layout conversion anywhere, and the DIAGONAL tile (R, R)
//     takes both operands from the wave's own registers.
// So the chain from one diagonal tile to the next never waits for another wave's panel.  Per panel p:
//   factor wave (the ninth; it holds no tiles, so its sixteen row registers do not compete with the slots):
//                  tile (p, p), one lane per row, rows 16-31 carrying the identity: elimination in registers
//                  -> M into LDS -> barrier B(p)
//   tile wave R:   [U(p-1) on tile (R, p)] -> P(R)^T = M X(R, p)^T -> publish -P(R)^T -> U(p) on its own diagonal tile
//                  -> (wave p+1: diagonal tile into LDS for the factor wave, which polls a flag word for it)
//                  -> U(p-1) on its remaining tiles, the panel to global memory -> barrier B(p+1)
// U(q) of an off-diagonal tile needs another wave's panel and is therefore applied one panel late (at panel q+1), which
// is early enough: tile (R, C) is first READ at panel C.  (In the code the slots are RELATIVE to the panel and rotate, so
// that every panel runs the same instructions; waves that share a SIMD with the wave of the next diagonal tile let it
// finish its twelve MFMAs first.)  LDS: M (2 x 2 KiB), the negated panels (2 x 16 KiB), 4 KiB for the diagonal tile on
// its way to one lane per row above the identity rows, 21 KiB of staging tiles for the stores to global memory.
// Round 2's form (three elimination waves holding one row of the whole 144-row panel per lane, thirteen update waves,
// two LDS round trips per panel) took 35 us per block, 40 % of it elimination; this one has the elimination of ONE
// 16 x 16 tile (with its identity rows), 12 MFMAs, one LDS flag and one barrier on the chain.
constexpr int PT_MA = 0;                        // 2 x 256: MA[16 t + c] = M[c][t]
constexpr int PT_NP = 2 * 256;                  // 2 x 8 x 256: [parity][C][register][lane]
constexpr int PT_CV = PT_NP + 2 * 8 * 256;      // 32 x 17: the diagonal tile, below it (written once) the identity
constexpr int PT_FLAG = PT_CV + 32 * 17;         // one word: the diagonal tile in CV belongs to panel <flag>
constexpr int PT_ST = PT_FLAG + 2;              // 9 x (16 x 18): a wave's solved tile, row-major, on its way to coalesced stores
constexpr int POTF2T_LDS_DOUBLES = PT_ST + 9 * 16 * 18;
constexpr int POTF2T_THREADS = 11 * 64;
constexpr int WAIT_VMCNT0 = 0x0F70;   // s_waitcnt vmcnt(0) alone (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)

__device__ __forceinline__ void lds_barrier() {
  // s_barrier that waits for this wave's LDS traffic only: global stores stay in flight (__syncthreads would drain them)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef __attribute__((address_space(1))) double gdouble;    // global memory and LDS, said explicitly: inside the
typedef __attribute__((address_space(3))) double ldouble;    // out-of-line instance below generic pointers would become
typedef __attribute__((address_space(3))) int lint;          // FLAT accesses (which count on vmcnt AND lgkmcnt)
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2d_t ldouble2;
typedef __attribute__((address_space(1))) v2d_t gdouble2;

// One 8-byte WRITE-THROUGH store (global_store_dwordx2 ... sc1): the value leaves this XCD's L2 with the store, so a
// workgroup that publishes results to workgroups of the same (or a concurrently running) launch needs no release fence
typedef __attribute__((address_space(1))) unsigned long long gu64_fit_t;
__device__ __forceinline__ void store_wt_f64(double* p, double v) {
  __hip_atomic_store((gu64_fit_t*)(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// stores of the diagonal block / a panel piece: plain, or write-through for the overlapped sweep (sweep_overlap below)
template <bool WT>
__device__ __forceinline__ void pst(gdouble* p, double v) {
  if (WT)
    __hip_atomic_store((gu64_fit_t*)(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    *p = v;
}
template <bool WT>
__device__ __forceinline__ void pst2(gdouble2* p, v2d_t v) {
  if (WT) {
    pst<true>((gdouble*)p, v.x);
    pst<true>((gdouble*)p + 1, v.y);
  } else {
    *p = v;
  }
}

#ifdef ELFIHIP_POTF2_STAMP   // developer probe (scripts/native/potf2_probe.hip): cycle stamps per wave, panel and phase
__device__ long long g_potf2_stamp[16 * 8 * 8];
#define STAMP(p, slot) do { if ((threadIdx.x & 63) == 0) g_potf2_stamp[(R * 8 + (p)) * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(p, slot) do { } while (0)
#endif

template <int NT, bool WT = false>
__device__ __forceinline__ void potf2_tiles_body(double* Akk_, int64_t lda, double* Wkk_, int64_t ldw, double* W11_, int* info,
                                                 int kblock, double* sm) {
  static_assert(NT >= POTF2T_THREADS, "eleven wave slots");
  // Waves are dealt to the four SIMDs of the CU in turn (wave w -> SIMD class w mod 4).  Wave 3 factors and has its SIMD
  // to itself (waves 7, 11, ... leave at once; a finished wave no longer counts at s_barrier): beside waves that keep
  // the SIMD's matrix pipe busy the elimination wave is hardly issued at all (measured: no progress during the 3 000
  // cycles of a panel's deferred updates).  The eight row blocks go to the other three classes so that the MFMA work of
  // every panel is as even as a fixed assignment allows, counting the 1 500 cycles the class of the next diagonal tile's
  // wave waits for it (rows {0, 7}, {1, 2, 6}, {3, 4, 5}: the matrix tiles of the late
  // rows are busy in the early panels, the identity tiles of the early rows in the late ones).
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= 11 || wid == 7 || wid == 8) return;
  //                      wid:  0  1  2  3  4  5  6  7  8  9  10
  constexpr unsigned long long ROW_OF_WAVE = 0x5'6'f'f'4'2'7'8'3'1'0ull;   // nibbles; 8: the factor wave
  const int R = (int)((ROW_OF_WAVE >> (4 * wid)) & 15);
  constexpr unsigned CLASS_OF_ROW = 0x0'1'2'2'2'1'1'0u;                     // nibble R: SIMD class of row block R's wave
  gdouble* Akk = (gdouble*)Akk_;
  gdouble* Wkk = (gdouble*)Wkk_;
  gdouble* W11 = (gdouble*)W11_;
  const int l = threadIdx.x & 63;
#ifdef ELFIHIP_POTF2_STAMP
  if ((threadIdx.x & 63) == 0) g_potf2_stamp[(R * 8 + 0) * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
#endif
  ldouble* MA = (ldouble*)sm + PT_MA;
  ldouble* NP = (ldouble*)sm + PT_NP;
  ldouble* CV = (ldouble*)sm + PT_CV;
  volatile lint* flag = (volatile lint*)((ldouble*)sm + PT_FLAG);
  if (R == 8) {
    // ================================================================== the factor wave
    const int trow = l & 15;
    const bool is_tile = l < 16, is_aug = l >= 16 && l < 32;
    if (l == 0) *flag = 0;
    int bad = 0;
    // tile (0, 0) straight from global memory.  The explicit wait tells the compiler's wait-count pass that no load is
    // outstanding when the loop is entered: otherwise it guards the first use of these registers in the loop with
    // s_waitcnt vmcnt(0), which also drains the global STORES of the previous panel -- a memory round trip per panel
    double a[16];
    {
      gdouble* src = Akk + (int64_t)trow * lda;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double v = src[c];
        a[c] = is_tile ? v : ((is_aug && c == trow) ? 1.0 : 0.0);   // (entries right of the diagonal are never read)
      }
      if (is_aug) {   // rows 16-31 of CV: the identity rows every later tile is eliminated with
#pragma unroll
        for (int c = 0; c < 16; ++c) CV[(16 + trow) * 17 + c] = c == trow ? 1.0 : 0.0;
      }
    }
    __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
#pragma unroll 1
    for (int p = 0; p < NB / 16; ++p) {
      ldouble* MAp = MA + (p & 1) * 256;
      STAMP(p, 0);
      if (p > 0) {
        // wave p has put its diagonal tile into CV and then raised the flag (LDS operations of one wave complete in
        // order); polling it instead of a barrier keeps the other waves' panel solves off this chain
        int spins = 0;
        while (*flag != p && ++spins < (1 << 22)) {
        }
        if (spins >= (1 << 22) && bad == 0) bad = kblock * NB + 16 * p + 1;   // cannot happen: a pivot report, not a hang
        STAMP(p, 5);
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = CV[(l & 31) * 17 + c];   // no masks: tile rows, then the identity rows
      }
      STAMP(p, 1);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        double acc0 = a[c], acc1 = 0.0;
#pragma unroll
        for (int j = 0; j < c; ++j) {
          const double lcj = readlane_f64(a[j], c);
          if (j & 1)
            acc1 = fma(-a[j], lcj, acc1);
          else
            acc0 = fma(-a[j], lcj, acc0);
        }
        a[c] = acc0 + acc1;
        // pivot: a[c] / sqrt(p) with ONE third-order step on the 24-bit v_rsq_f64 seed y0 (scripts/native/rsq_probe.hip:
        // 1.25 ulp): e = 1 - p y0^2, 1/sqrt(p) = y0 (1 + e/2 + 3 e^2/8).  Four dependent operations from the seed to
        // the scaled column instead of seven with two Newton steps -- this chain is what a panel's time is made of.
        const double pc = readlane_f64(a[c], c);
        const double y0 = __builtin_amdgcn_rsq(pc);
        const double ay0 = a[c] * y0;
        const double tc_ = pc * y0;
        const double ec = fma(-tc_, y0, 1.0);
        const double sc_ = fma(0.375, ec, 0.5);
        if (!(pc > 0.0) && bad == 0) bad = kblock * NB + 16 * p + c + 1;
        a[c] = fma(ay0 * ec, sc_, ay0);  // on tile row c this is p / sqrt(p): the diagonal of the factor
      }
      STAMP(p, 2);
      // M for everybody: identity row t = l - 16 holds L^-T[t][c] = M[c][t]
      if (is_aug) {
        ldouble* mrow = MAp + 16 * trow;
#pragma unroll
        for (int c = 0; c < 16; c += 2) *(ldouble2*)(mrow + c) = (v2d_t){a[c], a[c + 1]};
      }
      STAMP(p, 3);
      lds_barrier();   // B(p)
      STAMP(p, 4);
      // off the chain: the diagonal tile of L11 (lower triangle) to global memory, through LDS so that a store
      // instruction covers 8 rows x 128 bytes (one lane per row: 16 partial lines per instruction, and the CU's
      // memory pipe, not the elimination, sets the pace -- 1.5 us per panel)
      {
        ldouble* LT = (ldouble*)sm + PT_ST + 8 * 16 * 18;
        if (is_tile) {
#pragma unroll
          for (int c = 0; c < 16; c += 2) *(ldouble2*)(LT + trow * 18 + c) = (v2d_t){a[c], a[c + 1]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int sj = l & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = (l >> 3) + 8 * h;
          const v2d_t v = *(ldouble2*)(LT + i * 18 + 2 * sj);
          gdouble* dst = Akk + (int64_t)(16 * p + i) * lda + 16 * p + 2 * sj;
          if (2 * sj + 1 <= i)
            pst2<WT>((gdouble2*)dst, v);
          else if (2 * sj <= i)
            pst<WT>(dst, v.x);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (bad != 0 && l == 0) atomicCAS(info, 0, bad);
    return;
  }
  // ==================================================================== the tile waves
  // Register slots RELATIVE to the panel: `cur` = tile (R, p), q[j] = tile (R, p + 1 + j), rotated after every panel, and
  // the diagonal tile (R, R) in registers of its own (its q slot stays zero: tile (R, R) of the identity is never data).
  // Every panel therefore runs the SAME instructions (a slot chosen by `if (p == C)` copies was eight pieces of code
  // each executed once per launch: instruction-cache misses at ~100 cycles per instruction, 6 000 cycles per panel).
  const int lr = l >> 4, lc = l & 15;
  v4d cur = (v4d){0.0, 0.0, 0.0, 0.0}, dg = cur;
  v4d q[7];
  {
    gdouble* row = Akk + (int64_t)(16 * R + lc) * lda;
    if (R > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cur[r] = row[4 * r + lr];
        const int j = 4 * r + lr;   // symmetric fill from the lower triangle
        dg[r] = j <= lc ? row[16 * R + j] : Akk[(int64_t)(16 * R + j) * lda + 16 * R + lc];
      }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      q[j] = (v4d){0.0, 0.0, 0.0, 0.0};
      if (1 + j < R) {
#pragma unroll
        for (int r = 0; r < 4; ++r) q[j][r] = row[16 * (1 + j) + 4 * r + lr];
      }
    }
  }
  v4d yp = (v4d){0.0, 0.0, 0.0, 0.0};   // P_{p-1}(R)^T
  __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);   // (as in the factor wave: nothing but stores outstanding inside the loop)
  lds_barrier();                        // B(0)
#pragma unroll 1
  for (int p = 0; p < NB / 16; ++p) {
    const ldouble* MAp = MA + (p & 1) * 256;
    const ldouble* NPr = NP + ((p - 1) & 1) * 8 * 256;   // panels of step p-1
    ldouble* NPw = NP + (p & 1) * 8 * 256;
    STAMP(p, 0);
    // ---- on the chain: U(p-1) on tile (R, p), the panel solve, U(p) on the wave's own diagonal tile
    // (the wave whose diagonal tile is factored next goes first at the SIMD's matrix pipe, which it shares with two
    // other waves and their deferred updates)
    if (R != p + 1 && p + 1 < NB / 16 && ((CLASS_OF_ROW >> (4 * R)) & 15) == ((CLASS_OF_ROW >> (4 * (p + 1))) & 15)) {
      // the wave whose diagonal tile is factored next has its SIMD's matrix pipe to itself until it has handed the tile
      // over (beside the deferred updates of the waves it shares the SIMD with, its twelve MFMAs took 3 000 cycles)
      int spins = 0;
      while (*flag != p + 1 && ++spins < (1 << 22)) {
      }
    }
    v4d y;
    if (R == p) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = MAp[16 * lc + 4 * r + lr];   // P(p)^T of the identity rows = M
    } else {
      double ma[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) ma[kk] = MAp[64 * kk + l];
      if (p > 0) {
        double na[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) na[kk] = NPr[p * 256 + 64 * kk + l];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) cur = __builtin_amdgcn_mfma_f64_16x16x4f64(na[kk], yp[kk], cur, 0, 0, 0);
      }
      y = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) y = __builtin_amdgcn_mfma_f64_16x16x4f64(ma[kk], cur[kk], y, 0, 0, 0);
    }
    if (R > p) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) dg = __builtin_amdgcn_mfma_f64_16x16x4f64(-y[kk], y[kk], dg, 0, 0, 0);
      if (R == p + 1) {   // complete: to the factor wave, one row per lane there
#pragma unroll
        for (int r = 0; r < 4; ++r) CV[(4 * r + lr) * 17 + lc] = dg[r];
        asm volatile("" ::: "memory");
        if (l == 0) *flag = p + 1;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) NPw[R * 256 + 64 * r + l] = -y[r];
    }
    STAMP(p, 1);
    // ---- off the chain: U(p-1) on the remaining tiles right of column p (matrix tiles C < R; identity tiles C > R
    // once panel R has been eliminated, i.e. R <= p-1)
    if (p > 0) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int C = p + 1 + j;
        if (C < 8 && C != R && (C < R || R <= p - 1)) {
          const ldouble* np = NPr + C * 256 + l;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) q[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(np[64 * kk], yp[kk], q[j], 0, 0, 0);
        }
      }
    }
    STAMP(p, 2);
    // ---- the solved panel to global memory: L11 rows (R > p); L^-T rows (R <= p) into WT's diagonal block and,
    // transposed, into W11 = L11^-1 (entries on and right of the diagonal only).  Row-major destinations go through the
    // wave's LDS staging tile: 8 rows x 128 bytes per store instruction instead of 16 rows x 32 bytes.
    {
      ldouble* ST = (ldouble*)sm + PT_ST + R * (16 * 18);
#pragma unroll
      for (int r = 0; r < 4; ++r) ST[lc * 18 + 4 * r + lr] = y[r];
      if (R <= p) {   // W11[col][row]: the accumulator layout is already coalesced along the rows
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gcol = 16 * p + 4 * r + lr, grow = 16 * R + lc;
          if (gcol >= grow) pst<WT>(W11 + gcol * NB + grow, y[r]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int sj = l & 7;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = (l >> 3) + 8 * h;
        const v2d_t v = *(ldouble2*)(ST + i * 18 + 2 * sj);
        const int grow = 16 * R + i, gcol = 16 * p + 2 * sj;
        if (R > p) {
          pst2<WT>((gdouble2*)(Akk + (int64_t)grow * lda + gcol), v);
        } else {
          gdouble* dst = Wkk + (int64_t)grow * ldw + gcol;
          if (gcol >= grow)
            pst2<WT>((gdouble2*)dst, v);
          else if (gcol + 1 >= grow)
            pst<WT>(dst + 1, v.y);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // ---- next panel: rotate the slots
    yp = y;
    cur = q[0];
#pragma unroll
    for (int j = 0; j < 6; ++j) q[j] = q[j + 1];
    q[6] = (v4d){0.0, 0.0, 0.0, 0.0};
    STAMP(p, 3);
    if (p + 1 < NB / 16) lds_barrier();   // B(p+1)
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void potf2_tiles_kernel(double* Akk, int64_t lda, double* Wkk, int64_t ldw, double* W11,
                                                         int* info, int kblock) {
  extern __shared__ __align__(16) double sm[];
  potf2_tiles_body<NT>(Akk, lda, Wkk, ldw, W11, info, kblock, sm);
}

// --------------------------------------------------------------- panel solve on the matrix cores
// Row block list for block column k: A row blocks k+1..nb-1, the y block, then WT row blocks 0..k-1.
// Each workgroup: P <- P * W11^T, i.e. P[x][c] = sum_j P[x][j] W11[c][j]  (NT GEMM, K = 128, in place).
struct PanelArgs {
  double* A;
  double* WT;
  const double* W11;
  int64_t lda;
  int k, nb;      // k: the panel being solved (panel_block / trsm)
  int ku0, kun;   // trailing updates apply the panel group [ku0, ku0 + kun): K = 128 kun per pass over C
};

__device__ __forceinline__ double* panel_block(const PanelArgs& P, int idx) {
  const int nbelow = P.nb - 1 - P.k;
  if (idx < nbelow) return P.A + ((int64_t)(P.k + 1 + idx) * NB) * P.lda + (int64_t)P.k * NB;
  if (idx == nbelow) return P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)P.k * NB;  // y block
  return P.WT + ((int64_t)(idx - nbelow - 1) * NB) * P.lda + (int64_t)P.k * NB;
}

__global__ __launch_bounds__(256) void trsm_gemm_kernel(PanelArgs P) {
  // 4 workgroups per 128-row block, 32 rows each: a workgroup reads only the rows it
  // overwrites, so the in-place update is race free, and 4*nb workgroups fill the chip.
  extern __shared__ __align__(16) double lds[];
  double* Pb = panel_block(P, blockIdx.x >> 2) + (int64_t)(blockIdx.x & 3) * 32 * P.lda;
  GemmAcc32 acc;
  acc.zero();
  gemm_tile32_nt(acc, Pb, P.lda, P.W11, NB, 0, NB, lds);
  acc32_foreach(acc, [&](int row, int col, double v) { Pb[(int64_t)row * P.lda + col] = v; });
}

// The panel solve for the fused sweep, where it sits on the critical chain of every step: ONE memory round trip per
// workgroup and only the lower triangle of W11.  A workgroup owns 16 rows of a row block (8 workgroups per block: one
// per CU at n = 4096); each lane loads the MFMA operands of its wave straight from global memory -- the 16 x 128 strip of
// the panel (the same for the four waves) and the rows of W11 = L11^-1 of its two 16-column tiles -- all loads issued
// before the first wait, no LDS.  W11 is lower triangular: column tile t needs k < 16 (t + 1) only, and wave w takes
// tiles w and 7 - w (nine 16-deep k blocks each).  k is permuted as in lookahead_tile_kernel (lane group q of MFMA
// 2o / 2o+1 holds k = 8o + 2q / + 1), so a 16-byte load feeds two MFMAs.  In place: the barrier separates the strip's
// last read from its first overwrite.  (The LDS-staged 32-row form above: 8.3 us per launch at n = 4096.)
template <int W, bool WT = false>
__device__ __forceinline__ void trsm16_wave(double* Pb, int64_t lda, const double* W11, int l,
                                            const unsigned* wait_cnt = nullptr, int* info = nullptr,
                                            unsigned wait_target = 28u * 8u + 8u * 4u) {
  constexpr int T0 = W, T1 = 7 - W;              // the wave's column tiles
  constexpr int O0 = 2 * (T0 + 1), O1 = 2 * (T1 + 1);   // k octets they need
  const int q2 = 2 * (l >> 4);
  const double* pa = Pb + (int64_t)(l & 15) * lda + q2;
  const double* pb0 = W11 + (int64_t)(16 * T0 + (l & 15)) * NB + q2;
  const double* pb1 = W11 + (int64_t)(16 * T1 + (l & 15)) * NB + q2;
  double2 a[O1], b0[O0], b1[O1];
#pragma unroll
  for (int o = 0; o < O1; ++o) a[o] = *reinterpret_cast<const double2*>(pa + 8 * o);
#pragma unroll
  for (int o = 0; o < O0; ++o) b0[o] = *reinterpret_cast<const double2*>(pb0 + 8 * o);
#pragma unroll
  for (int o = 0; o < O1; ++o) b1[o] = *reinterpret_cast<const double2*>(pb1 + 8 * o);
  v4d c0 = (v4d){0.0, 0.0, 0.0, 0.0}, c1 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int o = 0; o < O1; ++o) {
    if (o < O0) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o].x, b0[o].x, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o].y, b0[o].y, c0, 0, 0, 0);
    }
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o].x, b1[o].x, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o].y, b1[o].y, c1, 0, 0, 0);
  }
  __syncthreads();   // every wave holds its copy of the strip: it may be overwritten
  if (wait_cnt) {
    // panel_look_kernel: rows of row block k+1 -- the tile waves of the same launch read them RAW; write after they have
    // (one polling lane per wave: 2048 lanes polling one address stood in the tile waves' way)
    if (l == 0) {
      int spins = 0;
      while (__hip_atomic_load(wait_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_target) {
        if (++spins >= (1 << 22)) {
          atomicCAS(info, 0, -1 /* STEP_INFO_TIMEOUT */);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  double* po = Pb + (int64_t)(l >> 4) * lda + (l & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (WT) {
      store_wt_f64(po + (int64_t)(4 * r) * lda + 16 * T0, c0[r]);
      store_wt_f64(po + (int64_t)(4 * r) * lda + 16 * T1, c1[r]);
    } else {
      po[(int64_t)(4 * r) * lda + 16 * T0] = c0[r];
      po[(int64_t)(4 * r) * lda + 16 * T1] = c1[r];
    }
  }
}

__global__ __launch_bounds__(256) void trsm16_kernel(PanelArgs P) {
  double* Pb = panel_block(P, blockIdx.x >> 3) + (int64_t)(blockIdx.x & 7) * 16 * P.lda;
  const int l = threadIdx.x & 63;
  switch (threadIdx.x >> 6) {
    case 0: trsm16_wave<0>(Pb, P.lda, P.W11, l); break;
    case 1: trsm16_wave<1>(Pb, P.lda, P.W11, l); break;
    case 2: trsm16_wave<2>(Pb, P.lda, P.W11, l); break;
    default: trsm16_wave<3>(Pb, P.lda, P.W11, l); break;
  }
}

// --------------------------------------------------------------- trailing update on the matrix cores
// Tiles for block column k (m = nb-1-k remaining block columns):
//   [0, tA)            Cholesky rows: (i, c), k < c <= i < nb      C -= P_i P_c^T   (SYRK)
//   [tA, tA+m)         y block:       (y, c)                        C -= P_y P_c^T
//   [tA+m, ...)        L^-T rows:     (r, c), r <= k                C  = beta C - P_r P_c^T, beta = 0 for r == k
//
// Look-ahead split: `single` = 1 updates only block column c0 = k+1 (what the next diagonal
// block and panel solve need); `single` = 0 updates block columns >= c0 (c0 = k+2 for the bulk
// launch that overlaps the next panel factorisation on another stream, c0 = k+1 for all).
__global__ __launch_bounds__(256) void trailing_update_kernel(PanelArgs P, int c0, int single) {
  extern __shared__ __align__(16) double lds[];
  int idx = blockIdx.x;
  const double* Ap;
  double* C;
  int cblk;
  bool same = false, beta0 = false;
  if (single) {
    cblk = c0;
    const int mrows = P.nb - c0;  // Cholesky row blocks c0..nb-1
    if (idx < mrows) {
      const int ib = c0 + idx;
      Ap = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)P.ku0 * NB;
      C = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)cblk * NB;
      same = (idx == 0);
    } else if (idx == mrows) {
      Ap = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)P.ku0 * NB;
      C = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)cblk * NB;
    } else {
      const int r = idx - mrows - 1;
      Ap = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)P.ku0 * NB;
      C = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)cblk * NB;
      beta0 = (r >= P.ku0);  // first group that touches this L^-T row: overwrite
    }
  } else {
    const int m = P.nb - c0;
    const int tA = m * (m + 1) / 2;
    if (idx < tA) {
      int i = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
      while (i * (i + 1) / 2 > idx) --i;
      while ((i + 1) * (i + 2) / 2 <= idx) ++i;
      const int c = idx - i * (i + 1) / 2;
      const int ib = c0 + i;
      cblk = c0 + c;
      Ap = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)P.ku0 * NB;
      C = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)cblk * NB;
      same = (i == c);
    } else if (idx < tA + m) {
      cblk = c0 + (idx - tA);
      Ap = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)P.ku0 * NB;
      C = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)cblk * NB;
    } else {
      idx -= tA + m;
      const int r = idx / m;
      cblk = c0 + (idx - r * m);
      Ap = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)P.ku0 * NB;
      C = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)cblk * NB;
      beta0 = (r >= P.ku0);  // first group that touches this L^-T row: overwrite
    }
  }
  const double* Bp = P.A + ((int64_t)cblk * NB) * P.lda + (int64_t)P.ku0 * NB;
  GemmAcc acc;
  acc.zero();
  if (same)
    gemm_tile_nt<true>(acc, Ap, P.lda, Bp, P.lda, 0, P.kun * NB, lds);
  else
    gemm_tile_nt<false>(acc, Ap, P.lda, Bp, P.lda, 0, P.kun * NB, lds);
  if (beta0)
    acc_foreach(acc, [&](int row, int col, double v) { C[(int64_t)row * P.lda + col] = -v; });
  else
    acc_foreach(acc, [&](int row, int col, double v) { C[(int64_t)row * P.lda + col] -= v; });
}

// Same update for ONE block column (the look-ahead column k+1) with 32-row workgroups: four
// times the workgroups at a quarter of the latency -- this launch sits on the critical path.
__global__ __launch_bounds__(256) void trailing_update_col_kernel(PanelArgs P, int cblk) {
  extern __shared__ __align__(16) double lds[];
  cblk += blockIdx.y;  // a launch with gridDim.y > 1 covers consecutive block columns (fine-grained bulk pass)
  const int rb = blockIdx.x >> 2, sub = blockIdx.x & 3;
  const int mrows = P.nb - cblk;
  if (rb >= mrows + 1 + P.ku0 + P.kun) return;
  const double* Ap;
  double* C;
  bool beta0 = false;
  if (rb < mrows) {
    const int ib = cblk + rb;
    Ap = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)P.ku0 * NB;
    C = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)cblk * NB;
  } else if (rb == mrows) {
    Ap = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)P.ku0 * NB;
    C = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)cblk * NB;
  } else {
    const int r = rb - mrows - 1;
    Ap = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)P.ku0 * NB;
    C = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)cblk * NB;
    beta0 = (r >= P.ku0);  // first group that touches this L^-T row: overwrite
  }
  Ap += (int64_t)sub * 32 * P.lda;
  C += (int64_t)sub * 32 * P.lda;
  const double* Bp = P.A + ((int64_t)cblk * NB) * P.lda + (int64_t)P.ku0 * NB;
  GemmAcc32 acc;
  acc.zero();
  gemm_tile32_nt(acc, Ap, P.lda, Bp, P.lda, 0, P.kun * NB, lds);
  if (beta0)
    acc32_foreach(acc, [&](int row, int col, double v) { C[(int64_t)row * P.lda + col] = -v; });
  else
    acc32_foreach(acc, [&](int row, int col, double v) { C[(int64_t)row * P.lda + col] -= v; });
}

// The look-ahead column again, for the launches that sit on the critical path: ONE memory round trip per workgroup.
// A workgroup owns a 32 x 32 tile of C (16 per 128-row block: blockIdx.x = (row block, 32-row sub block, 32-column
// slice)); its four waves split the k range [0, 128 KUN) and each lane loads the MFMA operands of its wave's quarter
// straight from global memory -- every load of the kernel is issued before the first wait (16 KUN doubles of A, as many
// of B and the C words per lane) -- instead of walking k through LDS one 16-deep slice per round trip.  k is permuted so
// that a 16-byte load feeds two MFMAs (lane group q of MFMA 2o / 2o+1 holds k = 8o + 2q / + 1, the same for both
// operands).  The wave partials meet in LDS and are summed in wave order (deterministic).  Compared with the 32 x 128
// form above the workgroup pulls 64 KUN KiB instead of 160 KUN KiB through its CU and four times as many CUs take part.
// Updates deeper than 256 (panel groups of four) keep two 32-deep pieces per wave in flight and refill a buffer as soon
// as its piece has been multiplied: two to three round trips instead of 24-32 (28-36 -> about 20 us per launch).
// lb: the workgroup's (row block, 32-row sub block, 32-column slice) = (lb >> 4, (lb >> 2) & 3, lb & 3);  WT: the tile
// leaves through write-through stores (overlapped sweep: another launch reads it while this one is still running)
template <int KUN, bool WT>
__device__ __forceinline__ void lookahead_tile_body(const PanelArgs& P, int cblk, int lb, double* lds) {
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int rb = lb >> 4, sub = (lb >> 2) & 3, cs = lb & 3;
  const int mrows = P.nb - cblk;
  if (rb >= mrows + 1 + P.ku0 + P.kun) return;
  const double* Ap;
  double* C;
  bool beta0 = false;
  if (rb < mrows) {
    const int ib = cblk + rb;
    Ap = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)P.ku0 * NB;
    C = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)cblk * NB;
  } else if (rb == mrows) {
    Ap = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)P.ku0 * NB;
    C = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)cblk * NB;
  } else {
    const int r = rb - mrows - 1;
    Ap = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)P.ku0 * NB;
    C = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)cblk * NB;
    beta0 = (r >= P.ku0);  // first group that touches this L^-T row: overwrite
  }
  Ap += (int64_t)sub * 32 * P.lda;
  C += (int64_t)sub * 32 * P.lda + cs * 32;
  const double* Bp = P.A + ((int64_t)cblk * NB + cs * 32) * P.lda + (int64_t)P.ku0 * NB;
  constexpr int KW = 32 * KUN;  // k range of one wave
  // the wave's range in pieces of 32 (4 octets of 8 k): two pieces are in flight; for KUN <= 2 that is everything, loaded
  // before the first wait; deeper updates refill a buffer as soon as its piece has been multiplied
  constexpr int NO = 4;
  const int kb = w * KW + 2 * (l >> 4);
  const double* pa = Ap + (int64_t)(l & 15) * P.lda + kb;
  const double* pb = Bp + (int64_t)(l & 15) * P.lda + kb;
  double2 a[2][2][NO], b[2][2][NO];  // [buffer][row tile][octet]
  auto load_piece = [&](int buf, int piece) {
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[buf][i][o] = *reinterpret_cast<const double2*>(pa + (int64_t)i * 16 * P.lda + 32 * piece + 8 * o);
        b[buf][i][o] = *reinterpret_cast<const double2*>(pb + (int64_t)i * 16 * P.lda + 32 * piece + 8 * o);
      }
  };
  load_piece(0, 0);
  if (KUN > 1) load_piece(1, 1);
  double cv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = t + 256 * u;
    cv[u] = beta0 ? 0.0 : C[(int64_t)(e >> 5) * P.lda + (e & 31)];
  }
  v4d acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int piece = 0; piece < KUN; ++piece) {
    const int buf = piece & 1;
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[buf][i][o].x, b[buf][j][o].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[buf][i][o].y, b[buf][j][o].y, acc[i][j], 0, 0, 0);
        }
    if (piece + 2 < KUN) load_piece(buf, piece + 2);
  }
  constexpr int TP = 33;  // pitch of a 32 x 32 partial in LDS
  double* part = lds + w * 32 * TP;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(i * 16 + (l >> 4) + 4 * r) * TP + j * 16 + (l & 15)] = acc[i][j][r];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = t + 256 * u;
    const int o = (e >> 5) * TP + (e & 31);
    const double s = ((lds[o] + lds[32 * TP + o]) + lds[2 * 32 * TP + o]) + lds[3 * 32 * TP + o];
    if (WT)
      store_wt_f64(C + (int64_t)(e >> 5) * P.lda + (e & 31), cv[u] - s);
    else
      C[(int64_t)(e >> 5) * P.lda + (e & 31)] = cv[u] - s;
  }
}

template <int KUN>
__global__ __launch_bounds__(256) void lookahead_tile_kernel(PanelArgs P, int cblk) {
  extern __shared__ __align__(16) double lds[];
  // gridDim.y consecutive block columns in one launch.  Workgroups go round-robin over the 8 XCDs: renumber so that the
  // four column slices of one 32-row sub block (the same rows of A) run on the same XCD and share its L2 (gridDim.x is
  // a multiple of 16)
  lookahead_tile_body<KUN, false>(P, cblk + blockIdx.y, (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3), lds);
}
constexpr size_t LOOKAHEAD_TILE_LDS = 4 * 32 * 33 * sizeof(double);

// --------------------------------------------------------------- panel solve AND diagonal tile in ONE launch (round 5)
// elfihip_gp_set_schedule(gp, 4, 0); kept for measurement, NOT the default: 11.5 us per launch at n = 4096 (10.7 at 1024)
// against 6.5 + 5.4 us for the two launches it replaces -- a rebuild of 1.660 against 1.640 ms (profiles/r05_fit_trace.md).
// A tile workgroup runs the panel solve's round trip (launch ramp, operand loads, 36 dependent MFMAs) and then the tile
// update's (LDS exchange, 8 MFMAs, partial sums, read-modify-write) one after the other: the launch is as long as the two
// it merges, less one kernel boundary, plus the write-after-read hand-off.  First form (four waves form both strips, 222
// registers): 14.9 us.
// Per step the fused sweep runs trsm16_kernel (6.4 us) -> lookahead_tile_kernel<1> (5.3 us) -> step_kernel: two launches on
// the critical chain of every block column during which the matrix pipes idle (profiles/r04_fit_n4096_trace.md).  The tile
// (k+1, k+1) -= P P^T only needs the 128 rows of the panel below the diagonal block, P = A21 W11^T -- and those rows can be
// formed from the RAW block A21 by whoever needs them.  panel_look_kernel does both in one launch:
//   workgroups [0, 36):  one 16 x 16 tile (i >= j) of the lower triangle of block (k+1, k+1) each (all the next diagonal
//                        block reads: potf2_tiles_body fills the diagonal tiles symmetrically from their lower halves).
//                        Four waves form the strips P_i and P_j (16 x 128 each, the SAME instruction sequence as
//                        trsm16_wave, so the values are bit-identical to the panel the other workgroups store), pass them
//                        through LDS into operand layout, and apply the K = 128 update split over the waves exactly as
//                        lookahead_tile_kernel<1> splits it (k = 32 w ...; partials summed in wave order): the factor is
//                        bit-identical to the three-launch form's.
//   workgroups [36, ..): trsm16_wave on 16 rows of the panel each, as before, in place.
// One dependency inside the launch, write-after-read only: the eight workgroups that overwrite rows of A21 (row block k+1)
// hold their stores until all 256 tile waves have their raw strips in registers (one relaxed device-scope arrival per
// wave after its loads have landed, one relaxed poll by the writers -- no data is handed over, so no fence).  The tile
// workgroups come first in the grid: they are resident before any writer can wait for them.
constexpr int LOOK_TILES = 36;
constexpr int LOOK_PP = 130;   // pitch (doubles) of a strip in LDS: even (16-byte operand reads)
constexpr size_t PANEL_LOOK_LDS = (2 * 16 * LOOK_PP + 4 * 16 * 17) * sizeof(double);

template <int W>
__device__ __forceinline__ void look_strip_wave(const double* As, int64_t lda, const double* W11, int l, double* Ps,
                                                unsigned* cnt) {
  // one 16 x 128 strip of the panel, P = A W11^T: trsm16_wave's instruction sequence (same values), result into LDS
  constexpr int T0 = W, T1 = 7 - W;
  constexpr int O0 = 2 * (T0 + 1), O1 = 2 * (T1 + 1);
  const int q2 = 2 * (l >> 4);
  const double* pa = As + (int64_t)(l & 15) * lda + q2;
  const double* pb0 = W11 + (int64_t)(16 * T0 + (l & 15)) * NB + q2;
  const double* pb1 = W11 + (int64_t)(16 * T1 + (l & 15)) * NB + q2;
  double2 a[O1], b0[O0], b1[O1];
#pragma unroll
  for (int o = 0; o < O1; ++o) a[o] = *reinterpret_cast<const double2*>(pa + 8 * o);
#pragma unroll
  for (int o = 0; o < O0; ++o) b0[o] = *reinterpret_cast<const double2*>(pb0 + 8 * o);
#pragma unroll
  for (int o = 0; o < O1; ++o) b1[o] = *reinterpret_cast<const double2*>(pb1 + 8 * o);
  // the raw strip is in registers: its rows may be overwritten now
  __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
  if (l == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v4d c0 = (v4d){0.0, 0.0, 0.0, 0.0}, c1 = c0;
#pragma unroll
  for (int o = 0; o < O1; ++o) {
    if (o < O0) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o].x, b0[o].x, c0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o].y, b0[o].y, c0, 0, 0, 0);
    }
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o].x, b1[o].x, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[o].y, b1[o].y, c1, 0, 0, 0);
  }
  double* po = Ps + (l >> 4) * LOOK_PP + (l & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    po[4 * r * LOOK_PP + 16 * T0] = c0[r];
    po[4 * r * LOOK_PP + 16 * T1] = c1[r];
  }
}

// tile waves that arrive at the counter: eight per off-diagonal tile (four per strip), four per diagonal tile
constexpr unsigned LOOK_ARRIVALS = 28u * 8u + 8u * 4u;

__global__ __launch_bounds__(512) void panel_look_kernel(PanelArgs P, unsigned* cnt, int* info) {
  extern __shared__ __align__(16) double lds[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, ww = w & 3, half = w >> 2;
  if (blockIdx.x >= LOOK_TILES) {
    // ---- the panel solve: two 16-row pieces per workgroup (waves 0-3 / 4-7); row block k+1's pieces wait for the tile
    // waves before they store
    const int piece = 2 * (blockIdx.x - LOOK_TILES) + half;
    const int npieces = 8 * P.nb;
    const int pc = piece < npieces ? piece : npieces - 1;   // (npieces is even: never taken; keeps the barrier count equal)
    double* Pb = panel_block(P, pc >> 3) + (int64_t)(pc & 7) * 16 * P.lda;
    unsigned* wait = (pc >> 3) == 0 ? cnt : nullptr;
    switch (ww) {
      case 0: trsm16_wave<0>(Pb, P.lda, P.W11, l, wait, info); break;
      case 1: trsm16_wave<1>(Pb, P.lda, P.W11, l, wait, info); break;
      case 2: trsm16_wave<2>(Pb, P.lda, P.W11, l, wait, info); break;
      default: trsm16_wave<3>(Pb, P.lda, P.W11, l, wait, info); break;
    }
    return;
  }
  // ---- tile (i, j), i >= j, of block (k+1, k+1): waves 0-3 form strip i, waves 4-7 strip j
  int i = 0, b = blockIdx.x;
  while (b > i) {
    b -= i + 1;
    ++i;
  }
  const int j = b;
  const bool diag = i == j;
  const double* Ablk = P.A + ((int64_t)(P.k + 1) * NB) * P.lda + (int64_t)P.k * NB;   // raw A21 rows of row block k+1
  double* C = P.A + ((int64_t)(P.k + 1) * NB + 16 * i) * P.lda + (int64_t)(P.k + 1) * NB + 16 * j;
  double* Pi = lds;
  double* Pj = diag ? lds : lds + 16 * LOOK_PP;
  double* part = lds + 2 * 16 * LOOK_PP;
  double cv = 0.0;
  if (t < 256) cv = C[(int64_t)(t >> 4) * P.lda + (t & 15)];   // old value of the tile entry this thread finishes
  if (half == 0 || !diag) {
    const double* As = Ablk + (int64_t)(16 * (half == 0 ? i : j)) * P.lda;
    double* Ps = half == 0 ? Pi : Pj;
    switch (ww) {
      case 0: look_strip_wave<0>(As, P.lda, P.W11, l, Ps, cnt); break;
      case 1: look_strip_wave<1>(As, P.lda, P.W11, l, Ps, cnt); break;
      case 2: look_strip_wave<2>(As, P.lda, P.W11, l, Ps, cnt); break;
      default: look_strip_wave<3>(As, P.lda, P.W11, l, Ps, cnt); break;
    }
  }
  __syncthreads();
  // tile -= P_i P_j^T: wave w < 4 takes k in [32 w, 32 w + 32), k permuted as everywhere (lane group q of MFMA 2o / 2o+1
  // holds k = 8o + 2q / + 1), partials summed in wave order -- lookahead_tile_kernel<1>'s arithmetic
  if (half == 0) {
    const int kb = 32 * w + 2 * (l >> 4);
    const double* pa = Pi + (l & 15) * LOOK_PP + kb;
    const double* pb = Pj + (l & 15) * LOOK_PP + kb;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const double2 a = *reinterpret_cast<const double2*>(pa + 8 * o);
      const double2 bb = *reinterpret_cast<const double2*>(pb + 8 * o);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bb.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bb.y, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w * 16 * 17 + ((l >> 4) + 4 * r) * 17 + (l & 15)] = acc[r];
  }
  __syncthreads();
  if (t < 256) {
    const int o = (t >> 4) * 17 + (t & 15);
    const double sum = ((part[o] + part[16 * 17 + o]) + part[2 * 16 * 17 + o]) + part[3 * 16 * 17 + o];
    C[(int64_t)(t >> 4) * P.lda + (t & 15)] = cv - sum;
  }
}

// out-of-line instance for the fused step kernel: inlined there, the block's 128 live registers would be allocated
// next to the update tile's
__device__ __noinline__ void potf2_tiles_call(double* Akk, int64_t lda, double* Wkk, int64_t ldw, double* W11, int* info,
                                              int kblock, double* sm) {
  potf2_tiles_body<1024>(Akk, lda, Wkk, ldw, W11, info, kblock, sm);
}
// (the same with write-through stores, for sweep_diag_kernel: inlined into that kernel's loop over the block columns the
// body spills 280 bytes per lane)
__device__ __noinline__ void potf2_tiles_call_wt(double* Akk, int64_t lda, double* Wkk, int64_t ldw, double* W11, int* info,
                                                 int kblock, double* sm) {
  potf2_tiles_body<1024, true>(Akk, lda, Wkk, ldw, W11, info, kblock, sm);
}

// --------------------------------------------------------------- one step of the sweep as ONE launch
// step_kernel(k): workgroup 0 factors diagonal block k+1 (potf2_tiles_body) WHILE the other workgroups update the
// trailing matrix -- the two halves of a step that do not depend on each other, in one launch on one stream: no second
// stream, no event hand-offs (about 6 us each on the critical chain), and the diagonal block is the launch's first
// workgroup, so it never waits for a compute unit to drain (as a separate one-workgroup kernel it needs a whole CU
// and measured 34 us alone, 57 us beside a running pass over the trailing matrix).
// Before it, per step: trsm_gemm_kernel (all row blocks of panel k) and the update of tile (k+1, k+1) alone
// (lookahead_tile_kernel<1>, 16 workgroups) -- the only tile the next diagonal block needs.
//
// Which tiles receive which columns of the factor in which step is decided on the host (sweep_sched.hpp): every step's
// update work is a list of UNITS -- the upper or lower 64 rows of a 128 x 128 tile receiving a contiguous range of
// 32-column k-tiles of the factor, K = 64 ... 416 -- dealt to the workgroups so that all of them, in every step, are busy
// for about as long as the diagonal block takes.  (A K = 128 update of a tile moves 512 KB for 4.2 MFLOP; updating every
// tile right of panel k in every step ran at the fabric's 5 TB/s, 25 us per tile against 13.6 us of matrix-pipe time,
// and made the first third of the steps three units long while the last third left most compute units idle.)
// Row kinds of a column's tiles: Cholesky rows i >= c (C -= P_i P_c^T), the y block, L^-T rows r (created --
// overwritten, not accumulated -- by the unit that brings block (r, r), the first non-zero block of that row; the
// strictly lower blocks WT(r, j < r) a unit's k range may start in are zero from allocation).
// 1024 threads per workgroup (what the diagonal block needs), so one workgroup per CU: the update workgroups are
// PERSISTENT -- one per CU beside workgroup 0 -- and walk their list with the next unit's first operand loads and old
// C values in flight while the current unit is multiplied.
struct StepArgs {
  PanelArgs P;              // P.k: the panel this launch solves (P.W11 = the inverse of diagonal block k) and applies
  double* W11;              // where the diagonal block k+1 puts ITS inverse (the other of the two buffers; NULL: the
                            // update alone, scripts/native/step_probe.hip)
  int* info;
  const SweepUnit* units;   // the schedule's unit table (all steps)
  const int32_t* wg_off;    // this step: offset of every update workgroup's first unit in `units`
  const SweepUnit* heads;   // this step: every update workgroup's first unit again (pad = its unit count), so that the
                            // first operand addresses are one load away from the workgroup number
  unsigned* cnt;            // this step's arrival counters {panel solved, strip (k+1, k) solved, tile (k+1, k+1) updated};
                            // NULL: round 2's form -- panel solve and diagonal tile are launches of their own
  int nmini;                // 16-row pieces of panel k (8 per row block)
  int nwg;                  // update workgroups launched
};

struct StepUnit {
  const double* Ap;   // unit's rows of the factor's columns (64 x K)
  const double* Bp;   // the block column's rows of the same columns (128 x K)
  double* C;          // unit's rows of the tile
  double keep;        // 1: C -= P P^T, 0: C = -P P^T (the unit that creates an L^-T tile)
  int nkt;            // 32-deep k-tiles: K / 32
  int kt0;            // the first of them
};

__device__ __forceinline__ StepUnit step_decode(const StepArgs& S, const int4 r) {
  const PanelArgs& P = S.P;   // r = {row, c | kt0 << 16, nkt | half << 16 | keep << 24, pad}
  const int row = r.x, c = r.y & 0xffff, kt0 = (r.y >> 16) & 0xffff;
  const int nkt = r.z & 0xffff, half = (r.z >> 16) & 0xff, keep = (r.z >> 24) & 0xff;
  const double* base = row <= P.nb ? P.A + ((int64_t)row * NB) * P.lda : P.WT + ((int64_t)(row - P.nb - 1) * NB) * P.lda;
  StepUnit U;
  U.Ap = base + (int64_t)kt0 * GK2 + (int64_t)half * 64 * P.lda;
  U.C = const_cast<double*>(base) + (int64_t)c * NB + (int64_t)half * 64 * P.lda;
  U.Bp = P.A + ((int64_t)c * NB) * P.lda + (int64_t)kt0 * GK2;
  U.keep = keep ? 1.0 : 0.0;
  U.nkt = nkt;
  U.kt0 = kt0;
  return U;
}

__device__ __forceinline__ StepUnit step_unit(const StepArgs& S, int i) {
  return step_decode(S, *reinterpret_cast<const int4*>(S.units + i));
}

#ifdef ELFIHIP_STEP_STAMP   // developer probe (scripts/native/step_timeline_probe.hip): shader-clock stamps of ONE step
__device__ long long g_step_stamp[1024 * 8];
__device__ int g_step_stamp_k = 16;
#define SSTAMP(slot) do { if (threadIdx.x == 0 && S.P.k == g_step_stamp_k) g_step_stamp[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SSTAMP(slot) do { } while (0)
#endif

// ---- hand-offs inside the fused step launch (guide: write-through stores, every storing wave drains, ONE relaxed
// device-scope arrival per workgroup, ONE agent-scope acquire by the consumer -- no release fence anywhere)
constexpr int STEP_SPIN_LIMIT = 1 << 22;   // polls (a few seconds): a hand-off that never comes is reported, not waited for
constexpr int STEP_INFO_TIMEOUT = -1;      // pivot report of a launch whose workgroups were not all resident

// The workgroup waits until *c >= target: ONE lane polls and the workgroup's FIRST WAVE ALONE takes the agent-scope acquire
// before the barrier.  (Round 3's form -- every thread polls, every wave acquires -- queued 16 x 255 cache invalidations
// per hand-off; buffer_inv is a cache operation, not a per-wave one, and a load issued behind a thousand of them took
// 8-25 us: scripts/native/overlap_probe.hip, profiles/r05_overlap.md.)  Called by every thread of the workgroup.
__device__ __forceinline__ void step_wait(const unsigned* c, unsigned target, int* info) {
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) {
      int spins = 0;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins >= STEP_SPIN_LIMIT) {
          atomicCAS(info, 0, STEP_INFO_TIMEOUT);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
// after the workgroup's write-through stores: drain, meet, one arrival (at one or two counters)
__device__ __forceinline__ void step_arrive(unsigned* c, unsigned n, unsigned* c2 = nullptr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(c, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c2) __hip_atomic_fetch_add(c2, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One 16-row piece of panel k, P <- P W11^T in place, by the sixteen waves of a step workgroup: wave (h, T) takes
// column tile T and the k octets [8 h, 8 h + 8) of the 2 (T + 1) the lower-triangular W11 leaves it (as trsm16_wave,
// whose single round trip this keeps: every operand is loaded before the first wait); the two halves meet in LDS.
// Stores are write-through: other workgroups of the SAME launch read the solved panel.
__device__ __forceinline__ void step_solve_piece(const PanelArgs& P, int piece, double* sm) {
  double* Pb = panel_block(P, piece >> 3) + (int64_t)(piece & 7) * 16 * P.lda;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int h = w >> 3, T = w & 7;
  const int O = 2 * (T + 1);
  const int q2 = 2 * (l >> 4);
  const double* pa = Pb + (int64_t)(l & 15) * P.lda + q2 + 64 * h;
  const double* pb = P.W11 + (int64_t)(16 * T + (l & 15)) * NB + q2 + 64 * h;
  double2 a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = (double2){0.0, 0.0};
    b[j] = (double2){0.0, 0.0};
    if (8 * h + j < O) {
      a[j] = *reinterpret_cast<const double2*>(pa + 8 * j);
      b[j] = *reinterpret_cast<const double2*>(pb + 8 * j);
    }
  }
  v4d c = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (8 * h + j < O) {
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j].x, b[j].x, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j].y, b[j].y, c, 0, 0, 0);
    }
  double* part = sm + (T * 64 + l) * 4;
  if (h == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) part[r] = c[r];
  }
  __syncthreads();   // every wave holds its part of the strip: it may be overwritten
  if (h == 0) {
    double* po = Pb + (int64_t)(l >> 4) * P.lda + (l & 15) + 16 * T;
#pragma unroll
    for (int r = 0; r < 4; ++r) store_wt_f64(po + (int64_t)(4 * r) * P.lda, c[r] + part[r]);
  }
  __syncthreads();   // LDS free for the next piece
}

// Two of the sixteen 32 x 32 pieces of tile (k+1, k+1) -= P P^T (P = the solved strip (k+1, k)) by waves 0-7 of a step
// workgroup (lookahead_tile_kernel<1> for the diagonal tile only: four waves per piece, 32 of the 128 k each, partial
// sums through LDS); write-through stores: the launch's first workgroup factors the tile next.
__device__ __forceinline__ void step_tile_pieces(const PanelArgs& P, int pair, double* sm) {
  const int t = threadIdx.x & 255, l = t & 63, w = t >> 6;
  const int g = threadIdx.x >> 8;                  // piece of the pair (waves 8-15: none)
  const int lb = 2 * pair + (g & 1);
  const int sub = (lb >> 2) & 3, cs = lb & 3;
  const int cblk = P.k + 1;
  constexpr int TP = 33;                           // pitch of a 32 x 32 partial in LDS
  double* lds = sm + (g & 1) * 4 * 32 * TP;
  const bool on = g < 2;
  const double* Ap = P.A + ((int64_t)cblk * NB + sub * 32) * P.lda + (int64_t)P.k * NB;
  const double* Bp = P.A + ((int64_t)cblk * NB + cs * 32) * P.lda + (int64_t)P.k * NB;
  double* C = P.A + ((int64_t)cblk * NB + sub * 32) * P.lda + (int64_t)cblk * NB + cs * 32;
  double cv[4];
  if (on) {
    const int kb = w * 32 + 2 * (l >> 4);
    const double* pa = Ap + (int64_t)(l & 15) * P.lda + kb;
    const double* pb = Bp + (int64_t)(l & 15) * P.lda + kb;
    double2 a[2][4], b[2][4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i][o] = *reinterpret_cast<const double2*>(pa + (int64_t)i * 16 * P.lda + 8 * o);
        b[i][o] = *reinterpret_cast<const double2*>(pb + (int64_t)i * 16 * P.lda + 8 * o);
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + 256 * u;
      cv[u] = C[(int64_t)(e >> 5) * P.lda + (e & 31)];
    }
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][o].x, b[j][o].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][o].y, b[j][o].y, acc[i][j], 0, 0, 0);
        }
    double* part = lds + w * 32 * TP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(i * 16 + (l >> 4) + 4 * r) * TP + j * 16 + (l & 15)] = acc[i][j][r];
  }
  __syncthreads();
  if (on) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + 256 * u;
      const int o = (e >> 5) * TP + (e & 31);
      const double sres = ((lds[o] + lds[32 * TP + o]) + lds[2 * 32 * TP + o]) + lds[3 * 32 * TP + o];
      store_wt_f64(C + (int64_t)(e >> 5) * P.lda + (e & 31), cv[u] - sres);
    }
  }
}

// The update workgroup's walk through its units of one step (wgi: its index among the update workgroups).  CHAINED: the
// panel is solved inside the same launch (step_chain_kernel).  WT: the tiles leave through write-through stores -- the
// overlapped sweep (sweep_update_kernel), where workgroups of a running launch on other XCDs read them in the next step.
template <bool CHAINED, bool WT>
__device__ __forceinline__ void step_units(const StepArgs& S, double* sm, int wgi) {
  const PanelArgs& P = S.P;
  constexpr bool chained = CHAINED;
  constexpr int RT = 1;                 // 16-row MFMA tiles per wave along the rows
  constexpr int ROWS = 64 * RT;         // rows of a unit
  constexpr int BUF = (ROWS + 128) * GLP2;  // doubles of one LDS stage: A rows, then B rows
  const int4 head = *reinterpret_cast<const int4*>(S.heads + wgi);
  int u = S.wg_off[wgi];
  if (head.w <= 0) return;
  const int uend = u + head.w;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wr = w >> 2, wc = w & 3;
  const int64_t lda = P.lda;
  // global -> LDS: thread t carries k pair 2 (t & 15) of rows (t >> 4) and (t >> 4) + 64
  const int64_t goff = (int64_t)(t >> 4) * lda + 2 * (t & 15);
  const int doff = (t >> 4) * GLP2 + 2 * (t & 15);
  // LDS -> MFMA operands: wave (wr, wc) owns rows [16 RT wr, +16 RT) x columns [32 wc, +32)
  const int faoff = (wr * 16 * RT + (l & 15)) * GLP2 + 2 * (l >> 4);
  const int fboff = (ROWS + wc * 32 + (l & 15)) * GLP2 + 2 * (l >> 4);
  const int64_t coff = (int64_t)(wr * 16 * RT + (l >> 4)) * lda + wc * 32 + (l & 15);

  // The k-tiles of a workgroup's units form ONE stream e = (unit, kt) through two LDS stages: while stage p is
  // multiplied, the next element (in registers since the previous step) is written to stage p^1 and the loads of the
  // element after it are issued -- one barrier per k-tile, and the LDS stores overlap other waves' MFMAs.
  StepUnit cur = step_decode(S, head);
  // k-tiles from 4 k on belong to the panel this launch solves: before the first of them is requested, the whole panel
  // must have arrived (a workgroup's units are ordered so that those come last)
  const int kt_new = chained ? 4 * P.k : (1 << 30);
  bool ready = !chained;
  auto need_panel = [&](const StepUnit& U, int kt) {
    if (chained && !ready && U.kt0 + kt >= kt_new) {
      SSTAMP(4);
      step_wait(S.cnt, (unsigned)S.nmini, S.info);
      SSTAMP(5);
      ready = true;
    }
  };
  double2 pa0, pa1, pb0, pb1;
  auto issue = [&](const double* A_, const double* B_) {
    pa0 = *reinterpret_cast<const double2*>(A_ + goff);
    if (RT == 2) pa1 = *reinterpret_cast<const double2*>(A_ + goff + 64 * lda);
    pb0 = *reinterpret_cast<const double2*>(B_ + goff);
    pb1 = *reinterpret_cast<const double2*>(B_ + goff + 64 * lda);
  };
  auto stage = [&](double* buf) {
    double* da = buf + doff;
    *reinterpret_cast<double2*>(da) = pa0;
    if (RT == 2) *reinterpret_cast<double2*>(da + 64 * GLP2) = pa1;
    *reinterpret_cast<double2*>(da + ROWS * GLP2) = pb0;
    *reinterpret_cast<double2*>(da + (ROWS + 64) * GLP2) = pb1;
  };
  pa1 = (double2){0.0, 0.0};
  need_panel(cur, 1);   // (the first two k-tiles are requested before the loop)
  issue(cur.Ap, cur.Bp);
  v4d cv[RT][2];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[i][j][r] = cur.C[coff + (int64_t)(i * 16 + 4 * r) * lda + j * 16];
  int p = 0;
  stage(sm);
  issue(cur.Ap + GK2, cur.Bp + GK2);
  __syncthreads();
  for (;;) {
    const int un = u + 1;
    const bool more = un < uend;
    const StepUnit nxt = more ? step_unit(S, un) : cur;
    v4d acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int kt = 0; kt < cur.nkt; ++kt) {
      const double* buf = sm + p * BUF;
      if (kt + 1 < cur.nkt || more) {
        stage(sm + (p ^ 1) * BUF);                       // element e+1: this unit's next k-tile or the next unit's first
        const int k2 = kt + 2 - cur.nkt;                 // element e+2
        if (k2 < 0) {
          need_panel(cur, kt + 2);
          issue(cur.Ap + (kt + 2) * GK2, cur.Bp + (kt + 2) * GK2);
        } else if (more) {
          need_panel(nxt, k2);
          issue(nxt.Ap + k2 * GK2, nxt.Bp + k2 * GK2);
        }
      }
      const double* fa = buf + faoff;
      const double* fb = buf + fboff;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        double2 a[RT], bb[2];
#pragma unroll
        for (int i = 0; i < RT; ++i) a[i] = *reinterpret_cast<const double2*>(fa + i * 16 * GLP2 + 8 * h);
#pragma unroll
        for (int j = 0; j < 2; ++j) bb[j] = *reinterpret_cast<const double2*>(fb + j * 16 * GLP2 + 8 * h);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, bb[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, bb[j].y, acc[i][j], 0, 0, 0);
          }
      }
      __syncthreads();
      p ^= 1;
    }
    double* cp = cur.C + coff;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (WT)
            store_wt_f64(cp + (int64_t)(i * 16 + 4 * r) * lda + j * 16, cur.keep * cv[i][j][r] - acc[i][j][r]);
          else
            cp[(int64_t)(i * 16 + 4 * r) * lda + j * 16] = cur.keep * cv[i][j][r] - acc[i][j][r];
    if (!more) {
      SSTAMP(6);
      break;
    }
    // the next unit's old values: in flight during its k loop
    const double* cn = nxt.C + coff;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[i][j][r] = cn[(int64_t)(i * 16 + 4 * r) * lda + j * 16];
    cur = nxt;
    u = un;
  }
}

// CHAINED: panel solve and diagonal tile inside this launch (S.cnt != NULL); the two forms are separate kernels so that
// the three-launch form's update loop compiles exactly as it did without the other's roles around it
template <bool CHAINED>
__device__ __forceinline__ void step_body(const StepArgs& S) {
  extern __shared__ __align__(16) double sm[];
  const PanelArgs& P = S.P;
  constexpr bool chained = CHAINED;
  SSTAMP(0);
  if (blockIdx.x == 0) {
    const int kk = P.k + 1;
    double* Akk = P.A + ((int64_t)kk * NB) * P.lda + (int64_t)kk * NB;
    double* Wkk = P.WT + ((int64_t)kk * NB) * P.lda + (int64_t)kk * NB;
    if (chained) step_wait(S.cnt + 2, 8u, S.info);   // tile (k+1, k+1) complete (eight workgroups, two pieces each)
    SSTAMP(1);
    if (S.W11) potf2_tiles_call(Akk, P.lda, Wkk, P.lda, S.W11, S.info, kk, sm);   // (NULL: the update alone, scripts/native/step_probe.hip)
    SSTAMP(2);
    return;
  }
  const int idx = blockIdx.x - 1;
  if (chained) {
    // ---- the panel solve, dealt over the update workgroups; the first eight take strip (k+1, k) and then the diagonal
    // tile: the chain to the next diagonal block is solve -> hop -> tile -> hop -> block, beside everybody's updates
    const int lead = S.nwg >= 16 ? 8 : 0;   // workgroups that keep out of the rest of the panel
    if (idx < 8) {
      step_solve_piece(P, idx, sm);
      step_arrive(S.cnt + 1, 1u, S.cnt);
      SSTAMP(1);
    }
    unsigned done = 0;
    if (idx >= lead)
      for (int e = 8 + (idx - lead); e < S.nmini; e += S.nwg - lead) {
        step_solve_piece(P, e, sm);
        ++done;
      }
    if (done) step_arrive(S.cnt, done);
    if (idx >= 8) SSTAMP(1);
    if (idx < 8) {
      step_wait(S.cnt + 1, 8u, S.info);
      SSTAMP(2);
      step_tile_pieces(P, idx, sm);
      step_arrive(S.cnt + 2, 1u);
      SSTAMP(3);
    }
  }
  step_units<CHAINED, false>(S, sm, idx);
}

__global__ __launch_bounds__(1024) void step_kernel(StepArgs S) { step_body<false>(S); }
__global__ __launch_bounds__(1024) void step_chain_kernel(StepArgs S) { step_body<true>(S); }

constexpr size_t STEP_LDS_BYTES = 2 * (64 + 128) * GLP2 * sizeof(double);  // two stages of a unit (> the diagonal block's)
static_assert(STEP_LDS_BYTES >= POTF2T_LDS_DOUBLES * sizeof(double) && STEP_LDS_BYTES <= 160 * 1024, "LDS budget");

// --------------------------------------------------------------- the sweep OVERLAPPED: chain beside the update (round 5)
// elfihip_gp_set_schedule(gp, 5, 0).  In the three-launch form the matrix pipes of 255 CUs idle while the chain's two
// small launches run (panel solve 6.5 us + diagonal tile 5.3 us + boundaries, per block column), and every step launch
// pays 7 us before its first unit multiplies.  Here the three roles of a step run CONCURRENTLY as three launches on three
// streams, ordered by counters in device memory instead of kernel boundaries or stream events:
//   sweep_update_kernel   ONE launch, persistent (stream `bulk`): the update workgroups (one per CU beside the diagonal
//                         block's) walk the same unit table, step by step; step k starts when panel k has been solved
//   sweep_diag_kernel     ONE launch, persistent, one workgroup (stream `hi`): diagonal block k+1 as soon as its tile is
//                         complete.  Only the CU the update leaves free can hold it (registers), as in the step launch
//   chain_ov_kernel       one launch per block column on the caller's stream, enqueued ahead: 8 nb workgroups solve one
//                         16-row piece of panel k each (trsm16_wave), 16 more update tile (k+1, k+1)
//                         (lookahead_tile_body<1>).  Its waves fit beside the update's (96 + 120 registers of a SIMD's
//                         512) and run at raised priority; launched while block k is still being factored, it polls.
// Hand-offs (per step, words of the counter block): potf2_done -> {solve pieces} -> panel_solved -> {update}, strip ->
// {tile pieces} -> tile_done -> {diagonal block}, upd_done -> {next step's solve pieces}.  Payloads are written through
// (sc1), the storing waves drain, ONE lane per workgroup arrives; a consumer polls with ONE lane and its FIRST WAVE ALONE
// takes the agent-scope acquire before the workgroup's barrier: buffer_inv is a cache operation, not a per-wave one, and
// a thousand of them queued per hand-off (every wave of every workgroup) are what made a load behind them take 8-25 us
// (scripts/native/overlap_probe.hip: 56 -> 21 us per hop pair).  The arithmetic, instruction for instruction, is the
// three-launch form's: the factor is bit-identical (tests/test_gp_gpu.py).
constexpr int OV_WORDS = 8;          // counter words per step: potf2_done, panel_solved, strip, tile_done, upd_done
constexpr int OV_POTF2 = 0, OV_PANEL = 1, OV_STRIP = 2, OV_TILE = 3, OV_UPD = 4;
// behind the steps' blocks: words of the launch as a whole
constexpr int OV_AUX_XRANK = 0;      // [16] update workgroups that have started, per XCD
constexpr int OV_AUX_NEXT = 16;      // the next update workgroup's index into the unit table
constexpr int OV_AUX_DIAG_XCC = 17;  // 1 + the XCD the diagonal block's workgroup runs on
constexpr int OV_AUX_RESIDENT = 18;  // workgroups of the two persistent launches that have started (and stay)
constexpr int OV_AUX_WORDS = 32;
constexpr int OV_TILE_WGS = 16;
constexpr int OV_SPIN_LIMIT = 1 << 20;

// one lane polls *c >= target (false: gave up -- the abort word info[1] is raised and every later wait returns at once)
__device__ __forceinline__ bool ov_poll(const unsigned* c, unsigned target, int* info, int who = 0) {
  int spins = 0;
  while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    ++spins;
    if (spins >= OV_SPIN_LIMIT || ((spins & 31) == 0 && __hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      // the first to give up says who it is, what it saw and what it waited for (info[2], info[3]: the error message)
      if (atomicCAS(info + 1, 0, 1) == 0) {
        info[2] = who;
        info[3] = (int)((__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 16) | (target & 0xffffu));
      }
      atomicCAS(info, 0, STEP_INFO_TIMEOUT);
      return false;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  return true;
}
// the workgroup waits for up to two counters; its first wave alone acquires
__device__ __forceinline__ bool ov_wait(const unsigned* c0, unsigned t0, const unsigned* c1, unsigned t1, int* info, int* s_ok,
                                        int who) {
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) {
      bool ok = ov_poll(c0, t0, info, who);
      if (ok && c1) ok = ov_poll(c1, t1, info, who + 1);
      *s_ok = ok ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return *s_ok != 0;
}
// after the workgroup's write-through stores: drain, meet, one arrival (at one or two counters)
__device__ __forceinline__ void ov_arrive(unsigned* c, unsigned* c2 = nullptr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c2) __hip_atomic_fetch_add(c2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct OvArgs {
  PanelArgs P;              // A, WT, lda, nb (k and W11 belong to the step)
  double* W11;              // the two buffers of the diagonal block's inverse, NB * NB each
  int* info;                // [0] pivot report, [1] abort
  unsigned* flags;          // OV_WORDS per step
  const SweepUnit* units;   // the schedule's tables for all steps
  const int32_t* wg_off;
  const SweepUnit* heads;
  unsigned* aux;            // OV_AUX_WORDS
  int nwg;                  // update workgroups
};

__device__ __forceinline__ unsigned ov_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }   // XCC_ID[3:0]

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5))) void sweep_update_kernel(OvArgs O) {
  extern __shared__ __align__(16) double sm[];
  __shared__ int s_ok;
  // One workgroup per CU (LDS), and the diagonal block's workgroup (sweep_diag_kernel, resident before this launch
  // starts: the gate on this stream) holds a CU of its own.  cu_count workgroups are launched and the first nwg =
  // cu_count - 4 to START take the indices into the unit table from a counter and stay; the others leave at once,
  // whenever they get a CU (at the latest when the sweep is over).  The dispatcher deals workgroups to the XCDs in
  // turn, 32 each, and does not start a launch at XCD 0: the XCD the diagonal block was sent to is one CU short for
  // this launch, whichever it is (measured: XCD 7), and a launch of exactly cu_count - 1 workgroups that all have to be
  // resident waits there for the other launch to END -- measured, as were two more workgroups that found no CU.
  __shared__ int s_wgi;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(O.aux + OV_AUX_XRANK + ov_xcc_id(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int w = (int)__hip_atomic_fetch_add(O.aux + OV_AUX_NEXT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w < O.nwg) __hip_atomic_fetch_add(O.aux + OV_AUX_RESIDENT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_wgi = w;
  }
  __syncthreads();
  const int wgi = s_wgi;
  if (wgi >= O.nwg) return;
  StepArgs S;
  S.P = O.P;
  S.P.kun = 1;
  S.W11 = nullptr;
  S.info = O.info;
  S.units = O.units;
  S.cnt = nullptr;
  S.nmini = 0;
  S.nwg = O.nwg;
  for (int k = 0; k + 1 < O.P.nb; ++k) {
    S.heads = O.heads + (size_t)k * O.nwg;
    const int4 head = *reinterpret_cast<const int4*>(S.heads + wgi);
    if (head.w <= 0) continue;   // no units in this step: neither waits nor arrives (the host counts the active ones)
    unsigned* fl = O.flags + (size_t)OV_WORDS * k;
    if (!ov_wait(fl + OV_PANEL, 8u * (unsigned)O.P.nb, nullptr, 0u, O.info, &s_ok, 1000 + k)) return;
    S.P.k = k;
    S.P.ku0 = k;
    S.wg_off = O.wg_off + (size_t)k * (O.nwg + 1);
    step_units<false, true>(S, sm, wgi);
    ov_arrive(fl + OV_UPD);
  }
}

__global__ __launch_bounds__(1024) void sweep_diag_kernel(OvArgs O) {
  extern __shared__ __align__(16) double sm[];
  __shared__ int s_ok;
  // the two wave slots the diagonal block never uses leave now: the workgroup's barriers count live waves only
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid >= 11 || wid == 7 || wid == 8) return;
  if (threadIdx.x == 0) {
    __hip_atomic_store(O.aux + OV_AUX_DIAG_XCC, 1u + ov_xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(O.aux + OV_AUX_RESIDENT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const PanelArgs& P = O.P;
  for (int k = 0; k + 1 < P.nb; ++k) {
    unsigned* fl = O.flags + (size_t)OV_WORDS * k;
    if (!ov_wait(fl + OV_TILE, (unsigned)OV_TILE_WGS, nullptr, 0u, O.info, &s_ok, 2000 + k)) return;
    const int kk = k + 1;
    double* Akk = P.A + ((int64_t)kk * NB) * P.lda + (int64_t)kk * NB;
    double* Wkk = P.WT + ((int64_t)kk * NB) * P.lda + (int64_t)kk * NB;
    potf2_tiles_call_wt(Akk, P.lda, Wkk, P.lda, O.W11 + (size_t)(kk & 1) * NB * NB, O.info, kk, sm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(fl + OV_WORDS + OV_POTF2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// P.k: the panel; fl: its counter words; upd_target: update workgroups with units in step k-1 (their arrivals at the
// previous step's OV_UPD say that block column k has received everything); ntile: 16, or 0 behind the last panel
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void chain_ov_kernel(PanelArgs P, unsigned* fl, unsigned upd_target, int ntile, int* info) {
  extern __shared__ __align__(16) double lds[];
  __shared__ int s_ok;
  __builtin_amdgcn_s_setprio(3);   // beside the update's waves (older, and never short of an MFMA to issue)
  // the tile workgroups first in the grid, then the pieces of row block k+1 they wait for: when the launch has more
  // workgroups than the chip has room beside the update (one per CU), the ones that start late are not on the chain
  if ((int)blockIdx.x < ntile) {
    if (!ov_wait(fl + OV_STRIP, 8u, nullptr, 0u, info, &s_ok, 5000 + P.k)) return;
    lookahead_tile_body<1, true>(P, P.k + 1, blockIdx.x, lds);
    ov_arrive(fl + OV_TILE);
    return;
  }
  const int b = blockIdx.x - ntile;
  if (P.k > 0 && !ov_wait(fl + OV_POTF2, 1u, fl - OV_WORDS + OV_UPD, upd_target, info, &s_ok, 3000 + 2 * P.k)) return;
  double* Pb = panel_block(P, b >> 3) + (int64_t)(b & 7) * 16 * P.lda;   // (row block k+1 first: pieces 0-7)
  const int l = threadIdx.x & 63;
  switch (threadIdx.x >> 6) {
    case 0: trsm16_wave<0, true>(Pb, P.lda, P.W11, l); break;
    case 1: trsm16_wave<1, true>(Pb, P.lda, P.W11, l); break;
    case 2: trsm16_wave<2, true>(Pb, P.lda, P.W11, l); break;
    default: trsm16_wave<3, true>(Pb, P.lda, P.W11, l); break;
  }
  ov_arrive(fl + OV_PANEL, (ntile && b < 8) ? fl + OV_STRIP : nullptr);
}

// One wave that polls a word: (1) on the update's stream before its launch -- the diagonal block's workgroup must be
// resident first, it needs an EMPTY CU; (2) on the caller's stream between the first chain launch and the others -- they
// may start only when every workgroup of the two persistent launches is resident.  (A chain launch that polls occupies
// registers: two of its workgroups on a CU leave no room for an update workgroup, which the chain launch is waiting for.)
__global__ __launch_bounds__(64) void ov_gate_kernel(const unsigned* word, unsigned target, int* info, int who) {
  if (threadIdx.x == 0) ov_poll(word, target, info, who);
}

// --------------------------------------------------------------- alpha, logdet, y^T K^-1 y
// ONE launch.  Workgroups [1, np / 4]: alpha_i = sum_{k >= i} WT[i][k] z_k, one wavefront per row, coalesced along k,
// eight loads of the row in flight per lane (the long rows at the top of the triangle are what the launch waits for:
// 24 us with one load per iteration); the sums of the eight strands are added in order.  Workgroup 0:
// red[0] = sum log L_ii (i < n), red[1] = sum z_i^2 (fixed order), red[2] = the pivot report, red[8..9] = min / max L_ii -- written straight to
// pinned host memory (two device-to-host copies, 25 us on this stack, replaced by the stream wait alone).
__global__ __launch_bounds__(256) void alpha_logdet_kernel(const double* WT, const double* A, const double* z, double* alpha,
                                                           double* red, const int* info, int64_t n, int64_t np, int64_t lda,
                                                           unsigned long long ticket) {
  if (blockIdx.x == 0) {   // first, so that it is under way while the long rows of the triangle stream
    __shared__ double s0[256], s1[256], s2[256], s3[256];
    double a = 0.0, b = 0.0, lo = 1e300, hi = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const double dg = A[i * lda + i];
      a += log(dg);
      b += z[i] * z[i];
      lo = dg < lo ? dg : lo;
      hi = dg > hi ? dg : hi;
    }
    s0[threadIdx.x] = a;
    s1[threadIdx.x] = b;
    s2[threadIdx.x] = lo;
    s3[threadIdx.x] = hi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) {
        s0[threadIdx.x] += s0[threadIdx.x + off];
        s1[threadIdx.x] += s1[threadIdx.x + off];
        s2[threadIdx.x] = s2[threadIdx.x + off] < s2[threadIdx.x] ? s2[threadIdx.x + off] : s2[threadIdx.x];
        s3[threadIdx.x] = s3[threadIdx.x + off] > s3[threadIdx.x] ? s3[threadIdx.x + off] : s3[threadIdx.x];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {   // `red` is pinned host memory: the rebuild's scalars need no copy
      red[0] = s0[0];
      red[1] = s1[0];
      red[2] = (double)*info;
      red[8] = s2[0];         // smallest / largest diagonal entry of L
      red[9] = s3[0];
      post_ticket(red + 15, ticket);   // the host waits for THIS, not for the rows of alpha still streaming (gp_factorize_attempt)
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)(blockIdx.x - 1) * 4 + (threadIdx.x >> 6);
  double s = 0.0;
  if (row < n) {
    const double* w = WT + row * lda;
    double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int64_t k = (row & ~(int64_t)63) + lane;
    for (; k + 7 * 64 < n; k += 512) {
      double wv[8], zv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        wv[u] = w[k + 64 * u];
        zv[u] = z[k + 64 * u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] += (k + 64 * u >= row) ? wv[u] * zv[u] : 0.0;
    }
    double tail = 0.0;
    for (; k < n; k += 64)
      if (k >= row) tail += w[k] * z[k];
    s = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) + tail;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  }
  if (lane == 0 && row < np) alpha[row] = s;
}

template <class K>
static int enable_lds(elfihip_ctx* ctx, K k, size_t bytes) {
  ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return ELFIHIP_OK;
}

// ---- the sweep, fused schedule: everything on the caller's stream, two small launches and one fused launch per step
//   potf2(0);  for k = 0 .. nb-1:  trsm(k) | tile (k+1, k+1) -= P P^T | step_kernel(k) = potf2(k+1) beside the trailing update
// Chain per step: 8 + 5 + max(30, update) us + three same-stream kernel boundaries (1.5-2 us each), against
// potf2 -> trsm -> look-ahead column -> two event hops (measured 82 us per step at n = 4096) of the stream schedule.
// The update work of every step comes from the schedule of sweep_sched.hpp, built once per (nb, update workgroups) in
// this process and uploaded once per GP object and nb.
static std::mutex g_sched_mutex;
static std::map<std::pair<int, int>, std::shared_ptr<const SweepSchedule>> g_sched_cache;

static std::shared_ptr<const SweepSchedule> sweep_schedule_for(int nb, int nwg, bool far_first) {
  std::lock_guard<std::mutex> lock(g_sched_mutex);
  auto key = std::make_pair(nb, far_first ? -nwg : nwg);
  auto it = g_sched_cache.find(key);
  if (it != g_sched_cache.end()) return it->second;
  auto S = std::make_shared<SweepSchedule>();
  sweep_build(nb, nwg, S.get(), far_first);
  g_sched_cache[key] = S;
  return S;
}

static int sweep_plan(elfihip_gp* gp, int nb, int nwg, bool far_first, hipStream_t st) {
  if (gp->sched_nb == nb && gp->sched_nwg == nwg && gp->sched_far_first == far_first) return ELFIHIP_OK;
  elfihip_ctx* ctx = gp->ctx;
  std::shared_ptr<const SweepSchedule> S = sweep_schedule_for(nb, nwg, far_first);
  // every workgroup's first unit of every step once more, contiguous per step (an empty workgroup: count 0)
  std::vector<SweepUnit> heads;
  for (const SweepStep& x : S->steps)
    for (int w = 0; w < nwg; ++w) {
      const int lo = S->wg_off[x.off0 + w], hi = S->wg_off[x.off0 + w + 1];
      SweepUnit h = {};
      if (hi > lo) h = S->units[lo];
      heads.push_back(h);
    }
  const size_t ub = S->units.size() * sizeof(SweepUnit), ob = S->wg_off.size() * sizeof(int32_t);
  const size_t hb = heads.size() * sizeof(SweepUnit);
  // the previous table may still be read by a sweep in flight on this stream
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  ELFIHIP_CHECK_HIP(ctx, gp->sched_mem.reserve(ub + ob + hb + 192));
  char* base = reinterpret_cast<char*>(gp->sched_mem.p);
  if (ub) ELFIHIP_CHECK_HIP(ctx, hipMemcpy(base, S->units.data(), ub, hipMemcpyHostToDevice));
  const size_t o_off = (ub + 63) / 64 * 64, h_off = (o_off + ob + 63) / 64 * 64;
  if (ob) ELFIHIP_CHECK_HIP(ctx, hipMemcpy(base + o_off, S->wg_off.data(), ob, hipMemcpyHostToDevice));
  if (hb) ELFIHIP_CHECK_HIP(ctx, hipMemcpy(base + h_off, heads.data(), hb, hipMemcpyHostToDevice));
  gp->sched_units = base;
  gp->sched_wgoff = base + o_off;
  gp->sched_heads = base + h_off;
  gp->sched_step_off.clear();
  gp->sched_step_nwg.clear();
  for (const SweepStep& x : S->steps) {
    gp->sched_step_off.push_back(x.off0);
    gp->sched_step_nwg.push_back(x.nwg);
  }
  gp->sched_nb = nb;
  gp->sched_nwg = nwg;
  gp->sched_far_first = far_first;
  return ELFIHIP_OK;
}

static int sweep_fused(elfihip_gp* gp, int nb, hipStream_t st, bool chained) {
  elfihip_ctx* ctx = gp->ctx;
  if (!ctx->step_lds_enabled) {
    ELFIHIP_TRY(enable_lds(ctx, step_kernel, STEP_LDS_BYTES));
    ELFIHIP_TRY(enable_lds(ctx, step_chain_kernel, STEP_LDS_BYTES));
    ctx->step_lds_enabled = true;
  }
  const int nwg = std::max(8, ctx->cu_count - 1);   // update workgroups: one per CU beside the diagonal block
  ELFIHIP_TRY(sweep_plan(gp, nb, nwg, chained, st));
  double* const W11buf[2] = {gp->W11, gp->W11 + (size_t)NB * NB};
  PanelArgs P;
  P.A = gp->A;
  P.WT = gp->WT;
  P.lda = gp->lda;
  P.nb = nb;
  P.kun = 1;
  hipLaunchKernelGGL(potf2_tiles_kernel<1024>, dim3(1), dim3(1024), POTF2T_LDS_DOUBLES * sizeof(double), st, gp->A,
                     gp->lda, gp->WT, gp->lda, W11buf[0], gp->info, 0);
  for (int k = 0; k < nb; ++k) {
    P.k = k;
    P.ku0 = k;
    P.W11 = W11buf[k & 1];
    const int nrows = (nb - 1 - k) + 1 + k;  // below + y block + L^-T rows above
    const int m = nb - 1 - k;
    const bool merged = !chained && m > 0 && gp->schedule == 4;   // measured: no faster than the two launches (below)
    if (merged)   // panel solve + tile (k+1, k+1) in one launch; word 3 of the step's counter block counts the tile waves
      hipLaunchKernelGGL(panel_look_kernel, dim3(LOOK_TILES + 4 * nrows), dim3(512), PANEL_LOOK_LDS, st, P,
                         reinterpret_cast<unsigned*>(gp->info) + 4 + 4 * k + 3, gp->info);
    else if (!chained || m == 0)
      hipLaunchKernelGGL(trsm16_kernel, dim3(8 * nrows), dim3(256), 0, st, P);
    if (m == 0) break;
    if (!chained && !merged) hipLaunchKernelGGL(lookahead_tile_kernel<1>, dim3(16), dim3(256), LOOKAHEAD_TILE_LDS, st, P, k + 1);
    StepArgs S;
    S.P = P;
    S.W11 = W11buf[(k + 1) & 1];
    S.info = gp->info;
    S.units = reinterpret_cast<const SweepUnit*>(gp->sched_units);
    S.wg_off = reinterpret_cast<const int32_t*>(gp->sched_wgoff) + gp->sched_step_off[k];
    S.heads = reinterpret_cast<const SweepUnit*>(gp->sched_heads) + (size_t)k * nwg;
    S.cnt = chained ? reinterpret_cast<unsigned*>(gp->info) + 4 + 4 * k : nullptr;
    S.nmini = 8 * nrows;
    S.nwg = nwg;
    // chained: every update workgroup has its share of the panel solve, with or without units
    const int grid = 1 + (chained ? nwg : gp->sched_step_nwg[k]);
    if (chained)
      hipLaunchKernelGGL(step_chain_kernel, dim3(grid), dim3(1024), STEP_LDS_BYTES, st, S);
    else
      hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(1024), STEP_LDS_BYTES, st, S);
  }
  return launch_status(ctx, "cholesky sweep (fused steps)");
}

// ---- the sweep, overlapped (kernels above): the persistent update and diagonal-block launches on the context's two
// auxiliary streams, the chain launches on the caller's; two stream events per REBUILD (fork after the first diagonal
// block, join before alpha), none per step.
static int sweep_overlap(elfihip_gp* gp, int nb, hipStream_t st) {
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_TRY(ctx_aux(ctx));
  if (!ctx->ov_lds_enabled) {
    ELFIHIP_TRY(enable_lds(ctx, sweep_update_kernel, STEP_LDS_BYTES));
    ctx->ov_lds_enabled = true;
  }
  const int nwg = std::max(8, ctx->cu_count - 4);   // update workgroups that stay (sweep_update_kernel)
  ELFIHIP_TRY(sweep_plan(gp, nb, nwg, false, st));
  hipStream_t su = ctx->bulk_stream, sd = ctx->hi_stream;
  PanelArgs P;
  P.A = gp->A;
  P.WT = gp->WT;
  P.lda = gp->lda;
  P.nb = nb;
  P.kun = 1;
  P.k = 0;
  P.ku0 = 0;
  P.W11 = gp->W11;
  unsigned* flags = reinterpret_cast<unsigned*>(gp->info) + gp->ov_flags_off;
  unsigned* aux = flags + (size_t)OV_WORDS * (gp->cap / NB + 1);
  hipLaunchKernelGGL(potf2_tiles_kernel<1024>, dim3(1), dim3(1024), POTF2T_LDS_DOUBLES * sizeof(double), st, gp->A,
                     gp->lda, gp->WT, gp->lda, gp->W11, gp->info, 0);
  // Once the persistent launches are on their streams, EVERY way out of this function joins them back into `st` -- on an
  // error path (a failed event record / wait below) by waiting for the two streams on the host: the launches give up their
  // spins by themselves (OV_SPIN_LIMIT), and the caller's next work on `st` (the memset of a retry, the next Gram matrix)
  // must not race with them on A / WT.
  struct Forked {
    hipStream_t a, b;
    bool armed = false;
    ~Forked() {
      if (armed) {
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
      }
    }
  } forked{su, sd};
  if (nb > 1) {
    ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev_a, st));
    ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(su, ctx->ev_a, 0));
    ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(sd, ctx->ev_a, 0));
    forked.armed = true;
    OvArgs O;
    O.P = P;
    O.W11 = gp->W11;
    O.info = gp->info;
    O.flags = flags;
    O.units = reinterpret_cast<const SweepUnit*>(gp->sched_units);
    O.wg_off = reinterpret_cast<const int32_t*>(gp->sched_wgoff);
    O.heads = reinterpret_cast<const SweepUnit*>(gp->sched_heads);
    O.nwg = nwg;
    O.aux = aux;
    hipLaunchKernelGGL(sweep_diag_kernel, dim3(1), dim3(1024), POTF2T_LDS_DOUBLES * sizeof(double), sd, O);
    hipLaunchKernelGGL(ov_gate_kernel, dim3(1), dim3(64), 0, su, aux + OV_AUX_DIAG_XCC, 1u, gp->info, 9001);
    hipLaunchKernelGGL(sweep_update_kernel, dim3(std::max(nwg, ctx->cu_count)), dim3(1024), STEP_LDS_BYTES, su, O);
  }
  for (int k = 0; k < nb; ++k) {
    P.k = k;
    P.ku0 = k;
    P.W11 = gp->W11 + (size_t)(k & 1) * NB * NB;
    const int ntile = k + 1 < nb ? OV_TILE_WGS : 0;
    const unsigned upd_target = k > 0 ? (unsigned)gp->sched_step_nwg[k - 1] : 0u;
    hipLaunchKernelGGL(chain_ov_kernel, dim3(8 * nb + ntile), dim3(256), LOOKAHEAD_TILE_LDS, st, P,
                       flags + (size_t)OV_WORDS * k, upd_target, ntile, gp->info);
    if (k == 0 && nb > 1)
      hipLaunchKernelGGL(ov_gate_kernel, dim3(1), dim3(64), 0, st, aux + OV_AUX_RESIDENT, (unsigned)nwg + 1u, gp->info, 9000);
  }
  if (nb > 1) {
    ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev_a, su));
    ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_a, 0));
    ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev_b, sd));
    ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_b, 0));
    forked.armed = false;   // joined in stream order
  }
  return launch_status(ctx, "cholesky sweep (overlapped)");
}

// ---- the sweep, stream schedule: critical chain on a high-priority stream, passes over the trailing matrix on a
// second one, panel groups (one pass with K = 128 G per G panels).  Ahead from about 40 block columns, where a
// K = 128 pass over the trailing matrix is HBM-limited (8 flop per byte).
static int sweep_streams(elfihip_gp* gp, int nb, hipStream_t st) {
  elfihip_ctx* ctx = gp->ctx;
  const size_t gemm_lds = GEMM_LDS_DOUBLES * sizeof(double);
  PanelArgs P;
  P.A = gp->A;
  P.WT = gp->WT;
  P.W11 = gp->W11;
  P.lda = gp->lda;
  P.nb = nb;
  // Two streams besides the caller's: `hi` (high priority) carries the critical path  potf2(k) -> trsm(k) -> update
  // of block column k+1, `bulk` carries the rest of the trailing update (block columns >= k+2), overlapping the next
  // panel factorisation.  (CU masks are not honoured on this stack -- DESIGN.md section 7 -- so the streams differ in
  // priority only: a critical kernel that meets a running bulk pass waits for its workgroups to retire, rocprof:
  // trsm 8 us alone, 25-30 us inside a pass.)
  // Hazards: the column-(k+1) update must follow the previous bulk update (same tiles), the bulk
  // update must follow trsm(k) (reads its panels).
  ELFIHIP_TRY(ctx_aux(ctx));
  hipStream_t hi = ctx->hi_stream, bulk = ctx->bulk_stream;
  ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev_a, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(hi, ctx->ev_a, 0));
  ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(bulk, ctx->ev_a, 0));
  bool bulk_pending = false;
  // Panel grouping.  Panels are eliminated in groups of G (1, 2 or 4).  Inside a group only the next
  // block column is brought up to date after each panel (left-looking: it receives all panels of the
  // group so far in one GEMM); when the group is complete the whole trailing matrix receives all G
  // panels in ONE pass (K = 128 G): the read-modify-write traffic on the trailing tiles -- what
  // limits a K = 128 update to 8 flop per byte -- drops by G, and so do the cross-stream hand-offs.
  // Order on `bulk` after a group: first the G-1 block columns the next group touches before its
  // own end (one launch and one event each, so the critical stream never waits for more than it
  // needs), then the rest.
  // measured rebuild times (round 1), G = 1 / 2 / 4:  n=8192: 15.3 / 12.2 / 11.7 ms; n=12288: 48.1 / 35.1 / 30.0 ms;
  // with the pass over C in 32-row workgroups n=4096: 2.72 / 2.70 / 2.83, n=4608: 3.25 / 3.27 / 3.30
  int group = gp->panel_group > 0 ? gp->panel_group : (nb >= 48 ? 4 : (nb >= 40 ? 2 : (nb >= 30 ? 4 : 1)));
  if (group != 2 && group != 4) group = 1;
  const size_t lds32 = GEMM32_LDS_DOUBLES * sizeof(double);
  // the pass over the trailing matrix in 32-row workgroups (the look-ahead column kernel over all block columns)
  // instead of 128 x 128 tiles: measured n=6144: 6.04 -> 5.77 ms, n=8192: 11.1 -> 10.3 ms, n=12288: 29.6 -> 27.0 ms;
  // also below 30 block columns, where short-lived workgroups let the critical kernels in sooner
  const bool fine_bulk = nb < 30 || nb >= 40;
  // every row block of block columns [cblk, cblk + ncol), 32-row workgroups
  auto col_update = [&](hipStream_t s_, int cblk, int ncol = 1) {
    const int rows = (nb - cblk) + 1 + (P.ku0 + P.kun);
    const dim3 grid(16 * rows, ncol), block(256);
    switch (P.kun) {
      case 1: hipLaunchKernelGGL(lookahead_tile_kernel<1>, grid, block, LOOKAHEAD_TILE_LDS, s_, P, cblk); break;
      case 2: hipLaunchKernelGGL(lookahead_tile_kernel<2>, grid, block, LOOKAHEAD_TILE_LDS, s_, P, cblk); break;
      case 3: hipLaunchKernelGGL(lookahead_tile_kernel<3>, grid, block, LOOKAHEAD_TILE_LDS, s_, P, cblk); break;
      default: hipLaunchKernelGGL(lookahead_tile_kernel<4>, grid, block, LOOKAHEAD_TILE_LDS, s_, P, cblk); break;
    }
  };
  int g0 = 0;              // first panel of the current group
  int urgent_pending = 0;  // window columns of this group still guarded by ev_u[0..)
  for (int k = 0; k < nb; ++k) {
    P.k = k;
    double* Akk = gp->A + ((int64_t)k * NB) * gp->lda + (int64_t)k * NB;
    double* Wkk = gp->WT + ((int64_t)k * NB) * gp->lda + (int64_t)k * NB;
    hipLaunchKernelGGL(potf2_tiles_kernel<1024>, dim3(1), dim3(1024), POTF2T_LDS_DOUBLES * sizeof(double), hi, Akk,
                       gp->lda, Wkk, gp->lda, gp->W11, gp->info, k);
    const int nrows = (nb - 1 - k) + 1 + k;  // below + y block + L^-T rows above
    hipLaunchKernelGGL(trsm_gemm_kernel, dim3(4 * nrows), dim3(256), lds32, hi, P);
    const int m = nb - 1 - k;  // block columns right of k
    if (m == 0) break;
    const int j = k - g0;      // position inside the group
    P.ku0 = g0;
    P.kun = j + 1;
    if (j < group - 1) {
      // column k+1 was in the previous group's window: its update from that group must have landed
      if (j < urgent_pending) ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(hi, ctx->ev_u[j], 0));
      col_update(hi, k + 1);
      continue;
    }
    // group complete
    if (bulk_pending) ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(hi, ctx->ev_b, 0));  // previous group's pass over C
    col_update(hi, k + 1);
    ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev_a, hi));  // panels g0..k solved, column k+1 current
    ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(bulk, ctx->ev_a, 0));
    urgent_pending = 0;
    for (int u = 0; u < group - 1 && k + 2 + u < nb; ++u) {  // the next group's window, one column at a time
      col_update(bulk, k + 2 + u);
      ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev_u[u], bulk));
      ++urgent_pending;
    }
    const int c0 = k + 1 + group;  // first block column left to the big pass
    if (c0 < nb) {
      const int mc = nb - c0;
      const int tiles = mc * (mc + 1) / 2 + mc + (k + 1) * mc;
      if (fine_bulk) {
        const int rows_max = (nb - c0) + 1 + (P.ku0 + P.kun);
        hipLaunchKernelGGL(trailing_update_col_kernel, dim3(4 * rows_max, mc), dim3(256), lds32, bulk, P, c0);
      } else {
        hipLaunchKernelGGL(trailing_update_kernel, dim3(tiles), dim3(256), gemm_lds, bulk, P, c0, 0);
      }
    }
    ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev_b, bulk));
    bulk_pending = true;
    g0 = k + 1;
  }
  ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev_a, hi));
  ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_a, 0));
  ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev_b, bulk));
  ELFIHIP_CHECK_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_b, 0));
  return launch_status(ctx, "cholesky sweep (streams)");
}

// One attempt: Gram matrix with `diag_add` on the diagonal, sweep, alpha + log-determinant.  *info_out = 0 on success,
// the 1-based index of the first non-positive pivot otherwise (the object is left unfactorised).
static int gp_factorize_attempt(elfihip_gp* gp, double diag_add, int* info_out) {
  elfihip_ctx* ctx = gp->ctx;
  hipStream_t st = ctx->stream;
  const int64_t np = gp->np;
  const int nb = (int)(np / NB);
  if (gp->wt_dirty) {
    // a sweep that met a non-positive (or NaN) pivot carries Inf / NaN through its products, also into the strictly lower
    // blocks of WT that every later sweep relies on being zero (0 * NaN = NaN): clear the matrix before the next attempt
    ELFIHIP_CHECK_HIP(ctx, hipMemsetAsync(gp->WT, 0, (size_t)gp->cap * gp->lda * sizeof(double), st));
    gp->wt_dirty = false;
  }
  prof_mark(gp, 0);
  {
    const int T = 256;
    hipLaunchKernelGGL(x2_kernel, dim3((unsigned)((gp->cap + T - 1) / T)), dim3(T), 0, st, gp->X, gp->x2, gp->n,
                       gp->cap, gp->dp);
    GramArgs G;
    G.X = gp->X;
    G.x2 = gp->x2;
    G.y = gp->y;
    G.A = gp->A;
    G.lda = gp->lda;
    G.n = gp->n;
    G.np = np;
    G.dp = gp->dp;
    G.var = gp->var;
    G.neg_half_inv_ls2 = -0.5 / (gp->ls * gp->ls);
    G.bias = gp->bias;
    G.diag_add = diag_add;
    G.info = gp->info;
    G.ninfo = gp->ninfo;
    const int64_t nt = np / 64;
    const size_t lds = 2 * 64 * (size_t)(gp->dp + 1) * sizeof(double);
    hipLaunchKernelGGL(gram_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), lds, st, G);
    ELFIHIP_TRY(launch_status(ctx, "gram_kernel"));
  }
  prof_mark(gp, 1);
  // schedule of the sweep: gp->schedule 1 = streams, 2 = fused steps (panel solve | diagonal tile | step launch per block
  // column), 3 = fused steps chained inside ONE launch per block column (measured slower: DESIGN.md section 7), 0 = by
  // size (elfihip_gp_set_schedule)
  const bool fused = gp->schedule == 2 || gp->schedule == 3 || gp->schedule == 4 || (gp->schedule == 0 && nb < FUSED_BELOW_NB);
  if (gp->schedule == 5 && nb < FUSED_BELOW_NB)
    ELFIHIP_TRY(sweep_overlap(gp, nb, st));
  else if (fused)
    ELFIHIP_TRY(sweep_fused(gp, nb, st, gp->schedule == 3));
  else
    ELFIHIP_TRY(sweep_streams(gp, nb, st));
  prof_mark(gp, 2);
  const double* z = gp->A + np * gp->lda;  // row np of A: z = L^-1 y
  hipLaunchKernelGGL(alpha_logdet_kernel, dim3((unsigned)(np / 4 + 1)), dim3(256), 0, st, gp->WT, gp->A, z, gp->alpha,
                     gp->h_fit, gp->info, gp->n, np, gp->lda, ++gp->fit_ticket);
  ELFIHIP_TRY(launch_status(ctx, "alpha/logdet"));
  prof_mark(gp, 3);
  // The rebuild's scalars (log-determinant, z'z, pivot report) are in page-locked memory as soon as the first workgroup of
  // alpha_logdet_kernel has stored them and its ticket: the host polls the ticket instead of draining the stream, so that what
  // the caller enqueues next (kernel rows of an acquisition, the gradient of a MAP search) reaches the device while the rows of
  // alpha are still being formed.  (A profiled rebuild reads event times: it drains the stream.)
  if (gp->profile)
    ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  else
    ELFIHIP_TRY(host_wait_ticket(ctx, reinterpret_cast<const volatile unsigned long long*>(gp->h_fit + 15), gp->fit_ticket));
  *info_out = (int)gp->h_fit[2];
  if (*info_out != 0) gp->wt_dirty = true;
  prof_add(gp, ELFIHIP_PHASE_GRAM, 0, 1);
  prof_add(gp, ELFIHIP_PHASE_SWEEP, 1, 2);
  prof_add(gp, ELFIHIP_PHASE_ALPHA, 2, 3);
  return ELFIHIP_OK;
}

int gp_factorize_impl(elfihip_gp* gp);

// The factorisation with GPy's `jitchol` semantics ([GPy-upstream] GPy/util/linalg.py: jitchol, reached from
// ExactGaussianInference -> pdinv behind GPyRegression.update / optimize, elfi/methods/bo/gpy_regression.py:286-323):
// a plain Cholesky first; if a pivot is not positive, retry with jitter = mean(diag Ky) * 1e-6 added to the diagonal,
// the jitter growing tenfold per failed try, at most five tries; then "not positive definite, even with jitter"
// (LinAlgError in the reference, ELFIHIP_ERR_NOT_PD here).  The factor, alpha, log-determinant and L^-T all belong to the
// jittered matrix, as GPy's posterior does.  The diagonal of Ky is constant for this kernel (var + bias + noise + 1e-8),
// so its mean is that value.
int gp_factorize_impl(elfihip_gp* gp) {
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, gp->n > 0, "GP has no evidence");
  const double diag0 = gp->noise + GP_JITTER;
  int info = 0;
  gp->jitter = 0.0;
  gp->jitter_tries = 0;
  // A rebuild that only APPENDS evidence to a factorisation that needed rung `first` (elfihip_gp_extend on a jittered factor:
  // GPy rebuilds on every update) starts AT that rung: pivot k of a right-looking sweep is a function of the leading k x k
  // block alone, so the plain attempt and the rungs below `first` would fail at the same pivot with the same arithmetic --
  // and each of them costs a full sweep plus, after the failure, a memset of the whole L^-T matrix (512 MB at capacity
  // 8192) per evidence point.  The result is the ladder's, (jitter, tries) included.
  int first = gp->jit_start;
  gp->jit_start = 0;
  if (first > gp->jitchol_maxtries) first = 0;
  if (first == 0) ELFIHIP_TRY(gp_factorize_attempt(gp, diag0, &info));
  if ((first > 0 || (info != 0 && info != STEP_INFO_TIMEOUT)) && gp->jitchol_maxtries > 0) {
    double jitter = (gp->var + gp->bias + diag0) * 1e-6;
    for (int t = 1; t < first; ++t) jitter *= 10.0;
    for (int t = first > 0 ? first : 1; t <= gp->jitchol_maxtries && std::isfinite(jitter); ++t, jitter *= 10.0) {
      gp->jitter_tries = t;
      ELFIHIP_TRY(gp_factorize_attempt(gp, diag0 + jitter, &info));
      if (info == 0 || info == STEP_INFO_TIMEOUT) {
        if (info == 0) gp->jitter = jitter;
        break;
      }
    }
  }
  const double red[2] = {gp->h_fit[0], gp->h_fit[1]};
  if (info == STEP_INFO_TIMEOUT) {
    gp->factored = false;
    int w[4] = {0, 0, 0, 0};
    (void)hipMemcpy(w, gp->info, sizeof(w), hipMemcpyDeviceToHost);
    return fail(ctx, ELFIHIP_ERR_HIP, "factorisation sweep: a hand-off inside a step launch timed out (the launch's "
                "workgroups were not all resident: another kernel holds compute units of this device) [waiter %d saw %d of %d]",
                w[2], (int)((unsigned)w[3] >> 16), w[3] & 0xffff);
  }
  if (info != 0) {
    gp->factored = false;
    if (gp->jitter_tries > 0)
      return fail(ctx, ELFIHIP_ERR_NOT_PD, "not positive definite, even with jitter (pivot %d <= 0 after %d tries)", info,
                  gp->jitter_tries);
    return fail(ctx, ELFIHIP_ERR_NOT_PD, "covariance matrix is not positive definite (pivot %d <= 0)", info);
  }
  gp->logdet = 2.0 * red[0];
  gp->yKy = red[1];
  gp->diag_min = gp->h_fit[8];
  gp->diag_max = gp->h_fit[9];
  gp->factored = true;
  gp->has_kinv = false;
  gp->kinv_sym = false;
  gp->lcb_steps = 0;
  gp->wl_valid = false;
  ++gp->fact_gen;
  ++gp->full_gen;
  return ELFIHIP_OK;
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_gp_create(elfihip_ctx* ctx, int d, int64_t capacity, elfihip_gp** out) {
  if (!ctx || !out) return fail(ctx, ELFIHIP_ERR_ARG, "NULL argument");
  *out = nullptr;
  ELFIHIP_REQUIRE(ctx, d >= 1 && d <= 256, "input dimension %d outside [1,256]", d);
  ELFIHIP_REQUIRE(ctx, capacity >= 1 && capacity <= (1 << 17), "capacity %lld outside [1,131072]",
                  (long long)capacity);
  DeviceGuard g(ctx->device);
  elfihip_gp* gp = new elfihip_gp();
  gp->ctx = ctx;
  gp->d = d;
  gp->dp = (int)round_up(d, 4);
  gp->cap = round_up(capacity, NB);
  gp->lda = gp->cap + 16;  // not a power of two: spreads rows over HBM channels; keeps 128-byte alignment
  const size_t mat = (size_t)gp->lda * sizeof(double);
  hipError_t e = hipSuccess;
  auto alloc = [&](double** p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(p), bytes);
    if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, ctx->stream);
  };
  alloc(&gp->X, (size_t)gp->cap * gp->dp * sizeof(double));
  alloc(&gp->x2, (size_t)gp->cap * sizeof(double));
  alloc(&gp->y, (size_t)gp->cap * sizeof(double));
  alloc(&gp->A, (size_t)(gp->cap + NB) * mat);
  alloc(&gp->WT, (size_t)gp->cap * mat);
  alloc(&gp->WL, (size_t)gp->cap * mat);
  alloc(&gp->W11, (2 * (size_t)NB * NB + 64) * sizeof(double));   // two: block k's is read while block k+1's is written
  alloc(&gp->alpha, (size_t)gp->cap * sizeof(double));
  alloc(&gp->red, 64 * sizeof(double));
  // pivot report and abort word; four counter words per step of the fused sweep; OV_WORDS per step of the overlapped one
  gp->ov_flags_off = 4 + 4 * (int)(gp->cap / NB);
  gp->ninfo = gp->ov_flags_off + OV_WORDS * ((int)(gp->cap / NB) + 1) + OV_AUX_WORDS;
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&gp->info), gp->ninfo * sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(gp->info, 0, gp->ninfo * sizeof(int), ctx->stream);
  if (e == hipSuccess)
    e = hipHostMalloc(reinterpret_cast<void**>(&gp->h_fit), 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) memset(gp->h_fit, 0, 16 * sizeof(double));   // (the ticket words start at 0: the first ticket is 1)
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    int rc = fail(ctx, e == hipErrorOutOfMemory ? ELFIHIP_ERR_NOMEM : ELFIHIP_ERR_HIP, "GP allocation failed: %s",
                  hipGetErrorString(e));
    elfihip_gp_free(gp);
    return rc;
  }
  *out = gp;
  return ELFIHIP_OK;
}

int elfihip_gp_free(elfihip_gp* gp) {
  if (!gp) return ELFIHIP_OK;
  DeviceGuard g(gp->ctx->device);
  (void)hipStreamSynchronize(gp->ctx->stream);
  for (double* p : {gp->X, gp->x2, gp->y, gp->A, gp->WT, gp->WL, gp->Kinv, gp->W11, gp->alpha, gp->red})
    if (p) (void)hipFree(p);
  if (gp->info) (void)hipFree(gp->info);
  if (gp->h_stage) (void)hipHostFree(gp->h_stage);
  if (gp->h_fit) (void)hipHostFree(gp->h_fit);
  for (auto e : gp->pev)
    if (e) (void)hipEventDestroy(e);
  if (gp->VP) (void)hipFree(gp->VP);
  if (gp->Pint) (void)hipFree(gp->Pint);
  if (gp->h_dense) (void)hipHostFree(gp->h_dense);
  if (gp->tri_cnt) (void)hipFree(gp->tri_cnt);
  gp->ws.release();
  gp->ws2.release();
  gp->ws_dense.release();
  gp->hyper_items.release();
  gp->sched_mem.release();
  delete gp;
  return ELFIHIP_OK;
}

int elfihip_gp_set_hyper(elfihip_gp* gp, double rbf_variance, double lengthscale, double bias_variance,
                         double noise_variance) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, rbf_variance > 0 && lengthscale > 0 && bias_variance >= 0 && noise_variance >= 0,
                  "hyper-parameters must be positive (var=%g ls=%g bias=%g noise=%g)", rbf_variance, lengthscale,
                  bias_variance, noise_variance);
  gp->var = rbf_variance;
  gp->ls = lengthscale;
  gp->bias = bias_variance;
  gp->noise = noise_variance;
  gp->factored = false;
  gp->has_kinv = false;
  gp->kinv_sym = false;
  return ELFIHIP_OK;
}

// One new evidence point travels in the kernel arguments (a BOLFI iteration appends exactly one): no staging copy, no
// synchronisation -- the two small copies below and their stream wait cost 30 us of every 1.7 ms update.
struct OneRow {
  double x[24];
  double y;
};
__global__ void append_row_kernel(OneRow r, double* X, double* y, int64_t at, int d, int dp) {
  const int t = threadIdx.x;
  if (t < dp) X[at * dp + t] = t < d ? r.x[t] : 0.0;
  if (t == 0) y[at] = r.y;
}

static int gp_copy_rows(elfihip_gp* gp, const double* X, const double* y, int64_t at, int64_t k) {
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, at + k <= gp->cap, "evidence count %lld exceeds the GP capacity %lld", (long long)(at + k),
                  (long long)gp->cap);
  if (k == 0) return ELFIHIP_OK;
  ELFIHIP_REQUIRE(ctx, X && y, "NULL data pointer");
  if (k == 1 && gp->dp <= 24) {
    OneRow r;
    for (int c = 0; c < 24; ++c) r.x[c] = c < gp->d ? X[c] : 0.0;
    r.y = y[0];
    hipLaunchKernelGGL(append_row_kernel, dim3(1), dim3(64), 0, ctx->stream, r, gp->X, gp->y, at, gp->d, gp->dp);
    return launch_status(ctx, "append_row_kernel");
  }
  ELFIHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(gp->X + at * gp->dp, (size_t)gp->dp * sizeof(double), X,
                                          (size_t)gp->d * sizeof(double), (size_t)gp->d * sizeof(double), (size_t)k,
                                          hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(gp->y + at, y, (size_t)k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // caller's buffers are free after return
  return ELFIHIP_OK;
}

int elfihip_gp_set_data(elfihip_gp* gp, const double* X, const double* y, int64_t n) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, n >= 0, "negative n");
  DeviceGuard g(gp->ctx->device);
  ELFIHIP_TRY(gp_copy_rows(gp, X, y, 0, n));
  gp->n = n;
  gp->np = round_up(n, NB);
  gp->factored = false;
  gp->has_kinv = false;
  gp->kinv_sym = false;
  return ELFIHIP_OK;
}

int elfihip_gp_append(elfihip_gp* gp, const double* X_new, const double* y_new, int64_t k) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, k >= 0, "negative k");
  DeviceGuard g(gp->ctx->device);
  ELFIHIP_TRY(gp_copy_rows(gp, X_new, y_new, gp->n, k));
  gp->n += k;
  gp->np = round_up(gp->n, NB);
  gp->factored = false;
  gp->has_kinv = false;
  gp->kinv_sym = false;
  return ELFIHIP_OK;
}

int elfihip_gp_factorize(elfihip_gp* gp, double* log_marginal) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  DeviceGuard g(gp->ctx->device);
  ELFIHIP_TRY(gp_factorize_impl(gp));
  if (log_marginal)
    *log_marginal = 0.5 * (-(double)gp->n * 1.8378770664093453 /* log(2 pi) */ - gp->logdet - gp->yKy);
  return ELFIHIP_OK;
}

int elfihip_gp_jitchol(elfihip_gp* gp, int maxtries, double* jitter, int* tries) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, maxtries <= 16, "maxtries %d above 16", maxtries);
  if (maxtries >= 0) gp->jitchol_maxtries = maxtries;
  if (jitter) *jitter = gp->factored ? gp->jitter : 0.0;
  if (tries) *tries = gp->jitter_tries;
  return ELFIHIP_OK;
}

int elfihip_gp_profile(elfihip_gp* gp, int enable, double* phase_ms, int64_t* phase_calls) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  elfihip_ctx* ctx = gp->ctx;
  DeviceGuard g(ctx->device);
  if (phase_ms)
    for (int i = 0; i < ELFIHIP_PHASE_COUNT; ++i) phase_ms[i] = gp->phase_ms[i];
  if (phase_calls)
    for (int i = 0; i < ELFIHIP_PHASE_COUNT; ++i) phase_calls[i] = gp->phase_calls[i];
  if (enable > 0) {
    for (auto& e : gp->pev)
      if (!e) ELFIHIP_CHECK_HIP(ctx, hipEventCreate(&e));
    for (int i = 0; i < ELFIHIP_PHASE_COUNT; ++i) {
      gp->phase_ms[i] = 0.0;
      gp->phase_calls[i] = 0;
    }
    gp->profile = true;
  } else if (enable == 0) {
    gp->profile = false;
  }
  return ELFIHIP_OK;
}

int elfihip_gp_set_schedule(elfihip_gp* gp, int schedule, int panel_group) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, schedule >= 0 && schedule <= 5, "schedule %d outside {0, ..., 5}", schedule);
  ELFIHIP_REQUIRE(gp->ctx, panel_group == 0 || panel_group == 1 || panel_group == 2 || panel_group == 4,
                  "panel_group %d outside {0, 1, 2, 4}", panel_group);
  gp->schedule = schedule;
  gp->panel_group = panel_group;
  return ELFIHIP_OK;
}

int elfihip_gp_size(const elfihip_gp* gp, int64_t* n, int64_t* capacity, int* d) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  if (n) *n = gp->n;
  if (capacity) *capacity = gp->cap;
  if (d) *d = gp->d;
  return ELFIHIP_OK;
}

// Copy internal state to the host (tests / ELFI attribute access).  which: 0 = L (n x n, lower,
// zeros above), 1 = L^-T (n x n, upper), 2 = alpha (n), 3 = X (n x d), 4 = y (n), 5 = K^-1 (n x n).
int elfihip_gp_get(elfihip_gp* gp, int which, double* out) {
  if (!gp || !out) return fail(gp ? gp->ctx : nullptr, ELFIHIP_ERR_ARG, "NULL argument");
  elfihip_ctx* ctx = gp->ctx;
  DeviceGuard g(ctx->device);
  const int64_t n = gp->n;
  const size_t row = (size_t)n * sizeof(double);
  switch (which) {
    case 0:
    case 1:
    case 5: {
      ELFIHIP_REQUIRE(ctx, gp->factored, "GP is not factorised");
      ELFIHIP_REQUIRE(ctx, which != 5 || gp->has_kinv, "K^-1 has not been formed");
      const double* src = which == 0 ? gp->A : (which == 1 ? gp->WT : gp->Kinv);
      ELFIHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(out, row, src, (size_t)gp->lda * sizeof(double), row, (size_t)n,
                                              hipMemcpyDeviceToHost, ctx->stream));
      ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) {
          if (which == 0 && j > i) out[i * n + j] = 0.0;
          if (which == 1 && j < i) out[i * n + j] = 0.0;
          if (which == 5 && j > i) out[i * n + j] = out[j * n + i];
        }
      if (which == 5)  // lower tiles were computed; mirror what was not
        for (int64_t i = 0; i < n; ++i)
          for (int64_t j = i + 1; j < n; ++j) out[i * n + j] = out[j * n + i];
      return ELFIHIP_OK;
    }
    case 2:
      ELFIHIP_REQUIRE(ctx, gp->factored, "GP is not factorised");
      ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, gp->alpha, row, hipMemcpyDeviceToHost, ctx->stream));
      break;
    case 3:
      ELFIHIP_CHECK_HIP(ctx, hipMemcpy2DAsync(out, (size_t)gp->d * sizeof(double), gp->X,
                                              (size_t)gp->dp * sizeof(double), (size_t)gp->d * sizeof(double),
                                              (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
      break;
    case 4:
      ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(out, gp->y, row, hipMemcpyDeviceToHost, ctx->stream));
      break;
    default:
      return fail(ctx, ELFIHIP_ERR_ARG, "unknown selector %d", which);
  }
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

}  // extern "C"
