// GP prediction for MANY points at once (S >= ~100): the two triangular products as dense matrix-core GEMMs.
//
// Same quantities as gp_predict.hip (GPyRegression.predict / predictive_gradients,
// elfi/methods/bo/gpy_regression.py:127-140,206-218; the lock-step of the multi-start search of
// elfi/methods/bo/utils.py:97-103 when BASELINE configs[4] asks for 256 starts; many-chain posterior sampling):
//
//   V = L^-1 KB^T     V[i][s] = sum_{k <= i} WL[i][k] KB[s][k]       WL = L^-1 (row-major, gp->WL)
//   U = L^-T V        U[i][s] = sum_{k >= i} WT[i][k] VT[s][k]       WT = L^-T (row-major, gp->WT)
//
// With S right-hand sides the products have S/4 flops per byte of the factor: from S ~ 50 on they are
// bound by the FP64 matrix pipes, not by HBM.  The streaming kernel of gp_predict.hip pushes 16 columns per pass through
// (row block, k chunk) workgroups and leaves split-k partials [pass][chunk][i][s] behind -- 33 MB written and re-read
// per pass and product at n = 8192 -- which is the right shape for S <= 16..64 (one read of the factor, nothing else
// matters) and the wrong one for S = 256 (0.17 of the matrix peak, round 2).  Here:
//   * both operands are read row-wise along k ("NT"), so one 64 x 64 x 32 MFMA tile loop serves both products; V is
//     written TRANSPOSED (VT[s][k]) so that it is the k-contiguous operand of the second product;
//   * FULL-k accumulation: a workgroup owns output tiles (64 rows x 64 points) from first to last k, no partials;
//   * the triangle is balanced by PAIRING: row block rb needs rb + 1 k blocks in the first product (nrb - rb in the
//     second), so the workgroup of pair p takes row blocks p and nrb - 1 - p -- every workgroup multiplies nrb + 1
//     blocks, 256 workgroups at n = 8192, S = 256 (one per CU);
//   * XCD-aware order: the column blocks of one pair (same rows of the factor) sit on the same XCD and share its L2;
//   * epilogues fused: the first product leaves sum_i v^2 per (row block, point) for the variance, the second turns its
//     tile of U straight into the gradient sums  sum_i alpha_i k_si (x_s - X_i),  sum_i u_is k_si (x_s - X_i)  per
//     (row block, point) -- U itself is never written.  kstar_kernel (kernel rows, mean partials) and finish_kernel
//     (assembly) are those of gp_predict.hip, unchanged.
// Four launches per evaluation round whatever S is.
#include "gp.hpp"
#include "mfma_f64.hpp"

#include <chrono>
#include <cstdlib>

namespace elfihip {

constexpr int DT = 64;            // points per column block (and the largest row-tile height)
constexpr int DK = 32;            // k-tile depth
constexpr int DLP = 36;           // LDS row pitch in doubles (as the step kernel's stages: conflict-free b128 reads)
constexpr int DEP = 65;           // pitch of the [row][point] epilogue tile
// LDS: two stages of the tile loop, (TM + 64) rows each; the gradient epilogue lays its tiles over them:
// U tile [TM][65] + kernel rows [64][TM + 1] + evidence rows [TM][24] + query points [64][24] + alpha [TM]
constexpr size_t dense_lds_bytes(int tm) {
  const size_t stages = (size_t)2 * (tm + DT) * DLP;
  const size_t epi = (size_t)tm * DEP + (size_t)DT * (tm + 1) + (size_t)tm * 24 + (size_t)DT * 24 + tm;
  return (stages > epi ? stages : epi) * sizeof(double);
}

// Wave layout of a TM x 64 output tile on 4 waves:  TM = 64: 2 x 2 waves of 32 x 32;  TM = 32: 2 x 2 waves of 16 x 32;
// TM = 16: 1 x 4 waves of 16 x 16.  Shorter tiles = more workgroups for the same product (fewer points, smaller n): the
// matrix pipes of a workgroup are less busy, but the whole chip takes part.
template <int TM>
struct DenseShape {
  static constexpr int WROWS = TM >= 32 ? TM / 2 : 16;
  static constexpr int WAVES_R = TM / WROWS;         // 2, 2, 1
  static constexpr int WAVES_C = 4 / WAVES_R;        // 2, 2, 4
  static constexpr int WCOLS = DT / WAVES_C;         // 32, 32, 16
  static constexpr int MI = WROWS / 16;              // 2, 1, 1
  static constexpr int NJ = WCOLS / 16;              // 2, 2, 1
  static constexpr int STAGE = (TM + DT) * DLP;      // doubles of one stage: A rows, then B rows
};

template <int TM>
struct DenseAcc {
  v4d c[DenseShape<TM>::MI][DenseShape<TM>::NJ];
};

// One k-tile of both operands in flight per register SET: thread t carries k pair 2 (t & 15) of rows (t >> 4) + 16 j
// (j < TM / 16 of A, j < 4 of B).  Plain local scalars named by token pasting: a struct passed by reference, or an array
// captured by a lambda, is sent to scratch memory by this compiler (272 bytes per lane measured) and the loads serialise.
#define DENSE_DECL(S) double2 S##a0 = {0.0, 0.0}, S##a1 = {0.0, 0.0}, S##a2 = {0.0, 0.0}, S##a3 = {0.0, 0.0}, S##b0, S##b1, S##b2, S##b3
#define DENSE_ISSUE(S, a_, b_)                                                        \
  do {                                                                                \
    const double* ia_ = (a_);                                                         \
    const double* ib_ = (b_);                                                         \
    S##a0 = *reinterpret_cast<const double2*>(ia_);                                   \
    if (TM >= 32) S##a1 = *reinterpret_cast<const double2*>(ia_ + (int64_t)16 * lda); \
    if (TM >= 64) S##a2 = *reinterpret_cast<const double2*>(ia_ + (int64_t)32 * lda); \
    if (TM >= 64) S##a3 = *reinterpret_cast<const double2*>(ia_ + (int64_t)48 * lda); \
    S##b0 = *reinterpret_cast<const double2*>(ib_);                                   \
    S##b1 = *reinterpret_cast<const double2*>(ib_ + (int64_t)16 * ldb);               \
    S##b2 = *reinterpret_cast<const double2*>(ib_ + (int64_t)32 * ldb);               \
    S##b3 = *reinterpret_cast<const double2*>(ib_ + (int64_t)48 * ldb);               \
  } while (0)
#define DENSE_STAGE(S, sb_)                                                           \
  do {                                                                                \
    double* sp_ = (sb_);                                                              \
    *reinterpret_cast<double2*>(sp_) = S##a0;                                         \
    if (TM >= 32) *reinterpret_cast<double2*>(sp_ + 16 * DLP) = S##a1;                \
    if (TM >= 64) *reinterpret_cast<double2*>(sp_ + 32 * DLP) = S##a2;                \
    if (TM >= 64) *reinterpret_cast<double2*>(sp_ + 48 * DLP) = S##a3;                \
    *reinterpret_cast<double2*>(sp_ + TM * DLP) = S##b0;                              \
    *reinterpret_cast<double2*>(sp_ + (TM + 16) * DLP) = S##b1;                       \
    *reinterpret_cast<double2*>(sp_ + (TM + 32) * DLP) = S##b2;                       \
    *reinterpret_cast<double2*>(sp_ + (TM + 48) * DLP) = S##b3;                       \
  } while (0)

template <int TM>
__device__ __forceinline__ void dense_mma(DenseAcc<TM>& acc, const double* fa, const double* fb) {
  typedef DenseShape<TM> SH;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    double2 a[SH::MI], b[SH::NJ];
#pragma unroll
    for (int i = 0; i < SH::MI; ++i) a[i] = *reinterpret_cast<const double2*>(fa + i * 16 * DLP + 8 * h);
#pragma unroll
    for (int j = 0; j < SH::NJ; ++j) b[j] = *reinterpret_cast<const double2*>(fb + j * 16 * DLP + 8 * h);
#pragma unroll
    for (int i = 0; i < SH::MI; ++i)
#pragma unroll
      for (int j = 0; j < SH::NJ; ++j) {
        acc.c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc.c[i][j], 0, 0, 0);
        acc.c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc.c[i][j], 0, 0, 0);
      }
  }
}

// acc += A(TM x K) B(64 x K)^T for k in [kbeg, kend), multiples of 32.  256 threads.  Two LDS stages, one barrier per
// k-tile, and TWO register sets: the loads of k-tile j are issued two k-tiles before it is written to its stage (a
// k-tile is about 1 us of matrix-pipe time at TM = 64, less than a loaded memory round trip: with one set in flight the
// waves waited at the stage write -- tri_first at n = 8192, S = 256: 0.341 ms with one set).
template <int TM>
__device__ __forceinline__ void gemm_tm_nt(DenseAcc<TM>& acc, const double* __restrict__ A, int64_t lda,
                                           const double* __restrict__ B, int64_t ldb, int kbeg, int kend, double* sm) {
  typedef DenseShape<TM> SH;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wr = w / SH::WAVES_C, wc = w % SH::WAVES_C;
  const double* pa = A + (int64_t)(t >> 4) * lda + 2 * (t & 15) + kbeg;
  const double* pb = B + (int64_t)(t >> 4) * ldb + 2 * (t & 15) + kbeg;
  double* s0 = sm + (t >> 4) * DLP + 2 * (t & 15);
  double* s1 = s0 + SH::STAGE;
  const int faoff = (wr * SH::WROWS + (l & 15)) * DLP + 2 * (l >> 4);
  const int fboff = (TM + wc * SH::WCOLS + (l & 15)) * DLP + 2 * (l >> 4);
  const int nkt = (kend - kbeg) / DK;
  DENSE_DECL(r0);
  DENSE_DECL(r1);
  DENSE_ISSUE(r0, pa, pb);
  if (nkt > 1) DENSE_ISSUE(r1, pa + DK, pb + DK);
  __syncthreads();   // whoever used the staging area before (previous tile's epilogue) is done
  DENSE_STAGE(r0, s0);
  if (nkt > 2) DENSE_ISSUE(r0, pa + 2 * DK, pb + 2 * DK);
  __syncthreads();
#pragma unroll 1
  for (int kt = 0; kt < nkt; kt += 2) {
    // even k-tile in stage 0; k-tile kt + 1 (set 1) goes to stage 1, set 1 is refilled with k-tile kt + 3
    if (kt + 1 < nkt) {
      DENSE_STAGE(r1, s1);
      if (kt + 3 < nkt) DENSE_ISSUE(r1, pa + (int64_t)(kt + 3) * DK, pb + (int64_t)(kt + 3) * DK);
    }
    dense_mma<TM>(acc, sm + faoff, sm + fboff);
    __syncthreads();
    if (kt + 1 >= nkt) break;
    // odd k-tile in stage 1; k-tile kt + 2 (set 0) goes to stage 0, set 0 is refilled with k-tile kt + 4
    if (kt + 2 < nkt) {
      DENSE_STAGE(r0, s0);
      if (kt + 4 < nkt) DENSE_ISSUE(r0, pa + (int64_t)(kt + 4) * DK, pb + (int64_t)(kt + 4) * DK);
    }
    dense_mma<TM>(acc, sm + SH::STAGE + faoff, sm + SH::STAGE + fboff);
    __syncthreads();
  }
}
#undef DENSE_DECL
#undef DENSE_ISSUE
#undef DENSE_STAGE

struct DenseArgs {
  const double* W;      // WL (first product) or WT (second)
  const double* Bm;     // KBT (first) or VT (second): [S_pad][np], k contiguous
  int64_t lda, np, n;
  int nrb, ncb;         // TM-row blocks of the factor, 64-point column blocks
  // first product
  double* VT;           // [S_pad][np]
  double* var_part;     // [pass][nrb][16]
  // second product
  const double* kr;     // [S_pad][np] kernel rows without the bias
  const double* X;      // (np, dp)
  const double* alpha;
  const double* xs;     // [S_pad][dp]
  double* g_part;       // [pass][16][nrb][2 dp]
  int dp;
};

// blockIdx -> (pair, column block): the ncb column blocks of a pair are consecutive in the LOGICAL order and the logical
// order runs down each XCD in turn (hardware: block b on XCD b % 8), so they share an L2.  gridDim.x is a multiple of 8.
__device__ __forceinline__ bool dense_decode(const DenseArgs& D, int* pair, int* cb) {
  const int lb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  *pair = lb / D.ncb;
  *cb = lb - *pair * D.ncb;
  return *pair < D.nrb / 2;
}

// MODE 0: V = L^-1 KB^T, written transposed + sum_i v^2.   MODE 1: U = L^-T V folded into the gradient sums.
template <int MODE, int TM>
__global__ __launch_bounds__(256) void dense_tri_kernel(DenseArgs D) {
  typedef DenseShape<TM> SH;
  extern __shared__ __align__(16) double sm[];
  int pair, cb;
  if (!dense_decode(D, &pair, &cb)) return;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int wr = w / SH::WAVES_C, wc = w % SH::WAVES_C;
  const int64_t s0 = (int64_t)cb * DT;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    // longer tile first in both products
    const int rb = MODE == 0 ? (half == 0 ? D.nrb - 1 - pair : pair) : (half == 0 ? pair : D.nrb - 1 - pair);
    const int64_t i0 = (int64_t)rb * TM;
    // k ranges in multiples of the 32-deep k-tile: [0, end of the diagonal 32-block) / [start of it, np)
    const int kbeg = MODE == 0 ? 0 : (int)(i0 / DK * DK);
    const int kend = MODE == 0 ? (int)((i0 + TM + DK - 1) / DK * DK) : (int)D.np;
    DenseAcc<TM> acc;
#pragma unroll
    for (int i = 0; i < SH::MI; ++i)
#pragma unroll
      for (int j = 0; j < SH::NJ; ++j) acc.c[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    gemm_tm_nt<TM>(acc, D.W + i0 * D.lda, D.lda, D.Bm + s0 * D.np, D.np, kbeg, kend, sm);
    // the tile as [i][s] in LDS (the GEMM's last barrier has passed: the staging area is free)
    double* T = sm;
#pragma unroll
    for (int i = 0; i < SH::MI; ++i)
#pragma unroll
      for (int j = 0; j < SH::NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          T[(wr * SH::WROWS + i * 16 + (l >> 4) + 4 * r) * DEP + wc * SH::WCOLS + j * 16 + (l & 15)] = acc.c[i][j][r];
    __syncthreads();
    if (MODE == 0) {
      // VT[s0 + s][i0 + i]: consecutive threads walk i (runs of TM doubles)
      for (int e = t; e < TM * DT; e += 256) {
        const int i = e % TM, s = e / TM;
        D.VT[(s0 + s) * D.np + i0 + i] = T[i * DEP + s];
      }
      if (t < DT) {   // sum_i v[i][s]^2 over the tile's rows, fixed order
        double q2 = 0.0;
        for (int i = 0; i < TM; ++i) {
          const double v = T[i * DEP + t];
          q2 += v * v;
        }
        const int64_t sg = s0 + t;
        D.var_part[((sg >> 4) * D.nrb + rb) * 16 + (sg & 15)] = q2;
      }
    } else {
      // gradient sums of this (row block, 64 points): needs k_si, X_i, alpha_i, x_s
      const int dp = D.dp;
      constexpr int KP = TM + 1;
      double* KR = sm + TM * DEP;           // [s][i], pitch TM + 1
      double* XI = KR + DT * KP;            // [i][24]
      double* XS = XI + TM * 24;            // [s][24]   (dp <= 24 by the caller's check)
      double* AL = XS + DT * 24;            // [i]
      for (int e = t; e < TM * DT; e += 256) {
        const int i = e % TM, s = e / TM;
        KR[s * KP + i] = D.kr[(s0 + s) * D.np + i0 + i];
      }
      for (int e = t; e < TM * dp; e += 256) {
        const int r = e / dp, c = e - r * dp;
        XI[r * 24 + c] = D.X[(i0 + r) * dp + c];
      }
      for (int e = t; e < DT * dp; e += 256) {
        const int r = e / dp, c = e - r * dp;
        XS[r * 24 + c] = D.xs[(s0 + r) * dp + c];
      }
      if (t < TM) AL[t] = (i0 + t) < D.n ? D.alpha[i0 + t] : 0.0;
      __syncthreads();
      // thread (s = t & 63, q = t >> 6): dimensions a = q, q + 4, ... ; sums over the tile's rows in order
      const int s = t & 63, q = t >> 6;
      double g1[6], g2[6], xa[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        g1[j] = 0.0;
        g2[j] = 0.0;
        xa[j] = (q + 4 * j) < dp ? XS[s * 24 + q + 4 * j] : 0.0;
      }
#pragma unroll 2
      for (int i = 0; i < TM; ++i) {
        const double k = KR[s * KP + i];
        const double c1 = AL[i] * k, c2 = T[i * DEP + s] * k;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const double diff = xa[j] - XI[i * 24 + ((q + 4 * j) < dp ? q + 4 * j : 0)];
          g1[j] += c1 * diff;
          g2[j] += c2 * diff;
        }
      }
      const int64_t sg = s0 + s;
      double* gp_ = D.g_part + (((sg >> 4) * 16 + (sg & 15)) * D.nrb + rb) * 2 * dp;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (q + 4 * j < dp) {
          gp_[q + 4 * j] = g1[j];
          gp_[dp + q + 4 * j] = g2[j];
        }
    }
    // (gemm_tm_nt opens with a barrier: the epilogue's LDS reads are over before the next tile stages)
  }
}

template <int TM>
static int dense_launch(elfihip_gp* gp, DenseArgs D, int mode, const double* kbt, const double* vt) {
  elfihip_ctx* ctx = gp->ctx;
  hipStream_t st = ctx->stream;
  constexpr size_t lds = dense_lds_bytes(TM);
  const unsigned bit = TM == 64 ? 1u : (TM == 32 ? 2u : 4u);   // per context (= per device of this process)
  if (!(ctx->dense_lds_mask & bit)) {
    ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_tri_kernel<0, TM>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(dense_tri_kernel<1, TM>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->dense_lds_mask |= bit;
  }
  D.nrb = (int)(D.np / TM);
  const unsigned grid = (unsigned)round_up((int64_t)D.ncb * (D.nrb / 2), 8);
  const bool prof = gp->profile;
  if (prof) prof_mark(gp, 0);
  D.W = gp->WL;
  D.Bm = kbt;
  hipLaunchKernelGGL((dense_tri_kernel<0, TM>), dim3(grid), dim3(256), lds, st, D);
  if (prof) prof_mark(gp, 1);
  if (mode == 1) {
    D.W = gp->WT;
    D.Bm = vt;
    hipLaunchKernelGGL((dense_tri_kernel<1, TM>), dim3(grid), dim3(256), lds, st, D);
  }
  if (prof) prof_mark(gp, 2);
  return ELFIHIP_OK;
}

// Row-tile height: the tallest of 64 / 32 / 16 that gives at least TWO workgroups per CU (pairs x column blocks): two
// co-resident workgroups cover each other's barriers and epilogues.  Measured (ms per call incl. kernel rows, assembly
// and copies; tile rows 64 / 32 / 16; profiles/r03_dense.md):
//   n = 8192: S = 128: 0.79 / 0.51 / 0.51   S = 192: 0.78 / 0.75 / 0.67   S = 256: 0.795 / 0.765 / 0.88
//   n = 4096: S = 128: 0.41 / 0.27 / 0.19   S = 192: 0.42 / 0.28 / 0.28   S = 256: 0.42 / 0.29 / 0.29
// -- the rule picks the fastest (or a tie) in every one of these cases.
static int dense_tile_rows(const elfihip_gp* gp, int ncb) {
  const int64_t want = (int64_t)2 * gp->ctx->cu_count;
  for (int tm : {64, 32})
    if ((int64_t)ncb * (gp->np / tm / 2) >= want) return tm;
  return 16;
}

int64_t dense_min_points(const elfihip_gp* gp) { return gp->dense_min > 0 ? gp->dense_min : 112; }

static int predict_dense_round(elfihip_gp* gp, const double* Xs, int64_t S, int mode, int noiseless, double beta, double* mu,
                               double* var, double* dmu, double* dvar, double* val, double* grad);

// mode: 0 = mean / variance, 1 = + gradients (and LCB); same outputs as predict_impl.  Any number of points: rounds of at
// most `cap` points reuse one workspace (3 np doubles per point: kr, kbt, VT -- 512 MiB at most) and one set of flags; a
// round's grid stays far below the 65535-block limit of its y / z dimensions (a posterior evaluated on a 10^5-point grid
// would otherwise ask for 5 GB of workspace, and 10^6 points for an illegal launch).
int predict_dense_impl(elfihip_gp* gp, const double* Xs, int64_t S, int mode, int noiseless, double beta, double* mu,
                       double* var, double* dmu, double* dvar, double* val, double* grad) {
  const int d = gp->d;
  int64_t cap = ((int64_t)512 << 20) / (24 * (gp->np > 0 ? gp->np : 1));
  cap = cap < 256 ? 256 : (cap > 4096 ? 4096 : cap);
  cap -= cap % 64;
  for (int64_t s0 = 0; s0 < S; s0 += cap) {
    const int64_t sc = S - s0 < cap ? S - s0 : cap;
    ELFIHIP_TRY(predict_dense_round(gp, Xs + s0 * d, sc, mode, noiseless, beta, mu ? mu + s0 : nullptr,
                                    var ? var + s0 : nullptr, dmu ? dmu + s0 * d : nullptr, dvar ? dvar + s0 * d : nullptr,
                                    val ? val + s0 : nullptr, grad ? grad + s0 * d : nullptr));
  }
  return ELFIHIP_OK;
}

static int predict_dense_round(elfihip_gp* gp, const double* Xs, int64_t S, int mode, int noiseless, double beta, double* mu,
                               double* var, double* dmu, double* dvar, double* val, double* grad) {
  elfihip_ctx* ctx = gp->ctx;
  hipStream_t st = ctx->stream;
  if (!gp->factored)
    return fail(ctx, ELFIHIP_ERR_STATE, "GP is not factorised (call elfihip_gp_factorize after changing data)");
  const int dp = gp->dp, d = gp->d;
  ELFIHIP_REQUIRE(ctx, dp <= 24, "the dense predictor handles up to 24 input dimensions");
  const int64_t np = gp->np;
  const int64_t S_pad = round_up(S, DT);
  const int npass = (int)(S_pad / 16);
  const int ncb = (int)(S_pad / DT);
  const int nrb_max = (int)(np / 16);   // chunks of the epilogue sums at the shortest row tile
  const int nblk_k = (int)((np + 255) / 256);
  const size_t outsz = (size_t)3 * 16 + 3 * 16 * dp;
  // workspace (doubles): xs | xs2 | kr | kbt | VT | mu_part | var_part | g_part | out
  size_t off = 0;
  auto take = [&](size_t doubles) {
    size_t o = off;
    off += (doubles + 15) & ~(size_t)15;
    return o;
  };
  const size_t o_xs = take((size_t)S_pad * dp), o_xs2 = take((size_t)S_pad), o_kr = take((size_t)S_pad * np),
               o_kbt = take((size_t)S_pad * np), o_vt = take((size_t)S_pad * np),
               o_mu = take((size_t)npass * 16 * nblk_k), o_var = take((size_t)npass * nrb_max * 16),
               o_g = take((size_t)npass * 16 * nrb_max * 2 * dp), o_out = take((size_t)npass * outsz);
  ELFIHIP_CHECK_HIP(ctx, gp->ws_dense.reserve(off * sizeof(double)));
  double* base = gp->ws_dense.as<double>();
  // pinned staging: points + norms up, results down (one copy each way)
  const size_t n_in = (size_t)S_pad * dp + (size_t)S_pad, n_out = (size_t)npass * outsz;
  // [completion flags, one per point | points + norms | results]: pinned, device-visible, coherent -- the input upload stays
  // one copy (the dense products read the points many times), the RESULTS are written to host memory by the assembly kernel
  // itself, which raises a flag per point the host polls: no download, no stream wait (a blocking stream wait wakes up
  // some 10-20 us late on this stack, once per round of the multi-start search)
  const size_t n_flags = (size_t)S_pad;
  if (gp->hd_cap < n_flags + n_in + n_out) {
    if (gp->h_dense) ELFIHIP_CHECK_HIP(ctx, hipHostFree(gp->h_dense));
    gp->h_dense = nullptr;
    gp->hd_cap = 0;
    const size_t want = 2 * (n_flags + n_in + n_out) + 1024;
    ELFIHIP_CHECK_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&gp->h_dense), want * sizeof(double),
                                         hipHostMallocMapped | hipHostMallocCoherent));
    gp->hd_cap = want;
    gp->hd_flags = 0;
    gp->hd_seq = 0;
  }
  if (gp->hd_flags < n_flags) {   // a larger call than any before: its new flag words start from zero
    for (size_t f = gp->hd_flags; f < n_flags; ++f) reinterpret_cast<unsigned long long*>(gp->h_dense)[f] = 0;
    gp->hd_flags = n_flags;
  }
  unsigned long long* hflag = reinterpret_cast<unsigned long long*>(gp->h_dense);
  double* hx = gp->h_dense + n_flags;
  double* hout = hx + n_in;
  std::fill(hx, hx + n_in, 0.0);
  for (int64_t s = 0; s < S; ++s) {
    double q = 0.0;
    for (int c = 0; c < d; ++c) {
      const double x = Xs[s * d + c];
      hx[(size_t)s * dp + c] = x;
      q += x * x;
    }
    hx[(size_t)S_pad * dp + s] = q;
  }
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(base + o_xs, hx, n_in * sizeof(double), hipMemcpyHostToDevice, st));
  // (xs2 directly follows xs: S_pad * dp is a multiple of the workspace's 16-double granule)
  ELFIHIP_TRY(ensure_wl_public(gp));   // L^-1 row-wise is the FIRST product's matrix here
  launch_kstar_passes(gp, base + o_xs, base + o_xs2, base + o_kr, base + o_kbt, base + o_mu, nblk_k, (unsigned)npass);
  const int tm = gp->dense_tm > 0 ? gp->dense_tm : dense_tile_rows(gp, ncb);
  const int nrb = (int)(np / tm);
  DenseArgs D;
  D.lda = gp->lda;
  D.np = np;
  D.n = gp->n;
  D.nrb = nrb;
  D.ncb = ncb;
  D.VT = base + o_vt;
  D.var_part = base + o_var;
  D.kr = base + o_kr;
  D.X = gp->X;
  D.alpha = gp->alpha;
  D.xs = base + o_xs;
  D.g_part = base + o_g;
  D.dp = dp;
  const bool prof = gp->profile;
  switch (tm) {
    case 64: ELFIHIP_TRY(dense_launch<64>(gp, D, mode, base + o_kbt, base + o_vt)); break;
    case 32: ELFIHIP_TRY(dense_launch<32>(gp, D, mode, base + o_kbt, base + o_vt)); break;
    default: ELFIHIP_TRY(dense_launch<16>(gp, D, mode, base + o_kbt, base + o_vt)); break;
  }
  const unsigned long long seq = ++gp->hd_seq;
  launch_finish_passes(gp, base + o_mu, nblk_k, base + o_var, nrb, base + o_g, nrb, base + o_out, (int)S, noiseless, beta, mode,
                       (unsigned)npass, hout, hflag, seq);
  ELFIHIP_TRY(launch_status(ctx, "dense prediction"));
  {
    // poll the points' flags (the assembly workgroup of a point writes its results, fences, then raises the flag);
    // if they do not arrive within the budget the stream wait takes over and reports whatever went wrong
    const auto t0 = std::chrono::steady_clock::now();
    int64_t done = 0;
    bool ok = false;
    for (unsigned spin = 0;; ++spin) {
      while (done < S && __atomic_load_n(hflag + done, __ATOMIC_ACQUIRE) == seq) ++done;
      if (done == S) {
        ok = true;
        break;
      }
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) break;
    }
    if (!ok) ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  }
  if (prof) {
    prof_add(gp, ELFIHIP_PHASE_TRI_FIRST, 0, 1);
    if (mode == 1) prof_add(gp, ELFIHIP_PHASE_TRI_SECOND, 1, 2);
  }
  for (int64_t s = 0; s < S; ++s) {
    const double* o = hout + (size_t)(s / 16) * outsz;
    const int q = (int)(s % 16);
    if (mu) mu[s] = o[q];
    if (var) var[s] = o[16 + q];
    if (val) val[s] = o[2 * 16 + q];
    for (int c = 0; c < d; ++c) {
      if (dmu) dmu[s * d + c] = o[3 * 16 + q * dp + c];
      if (dvar) dvar[s * d + c] = o[3 * 16 + 16 * dp + q * dp + c];
      if (grad) grad[s * d + c] = o[3 * 16 + 2 * 16 * dp + q * dp + c];
    }
  }
  return ELFIHIP_OK;
}

}  // namespace elfihip

extern "C" int elfihip_gp_set_dense_threshold(elfihip_gp* gp, int64_t min_points, int tile_rows) {
  if (!gp) return elfihip::fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  if (!(tile_rows == 0 || tile_rows == 16 || tile_rows == 32 || tile_rows == 64))
    return elfihip::fail(gp->ctx, ELFIHIP_ERR_ARG, "tile_rows must be 0, 16, 32 or 64");
  gp->dense_min = min_points > 0 ? min_points : 0;
  gp->dense_tm = tile_rows;
  return ELFIHIP_OK;
}
