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
//   * per block column: (1) potf2_aug: one workgroup factors the 128x128 diagonal block and
//     its inverse in LDS; (2) trsm: every row block below / above multiplies its panel by
//     W11^T on the matrix cores (a 128x128x128 GEMM per workgroup); (3) update: one launch
//     of 128x128 f64-MFMA tiles does the SYRK on the trailing lower triangle AND the GEMM
//     on the y row and on the growing L^-T rows.
//   * alpha = L^-T z is a triangular GEMV; logdet from the diagonal.
// Flops: n^3/3 (Cholesky) + n^3/3 (L^-T) on v_mfma_f64_16x16x4_f64.
#include "gp.hpp"
#include "mfma_f64.hpp"

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
};

__global__ __launch_bounds__(256) void gram_kernel(GramArgs G) {
  extern __shared__ __align__(16) double sm[];
  const int pitch = G.dp + 1;  // odd pitch in doubles: bank-conflict free for column walks
  double* Xi = sm;
  double* Xj = sm + 64 * pitch;
  // decode lower-triangular tile pair (ti >= tj) from the linear block index
  const int64_t nt = G.np / 64;
  int64_t b = blockIdx.x;
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
          r2 = r2 > 0.0 ? r2 : 0.0;
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

constexpr int SP = 129;   // pitch of the 128x128 working block S in LDS
constexpr int PP = 18;    // pitch of the 16-wide panel copy
constexpr int POTF2_LDS_DOUBLES = NB * SP + NB + 16 + 16 * 17 + 144 * PP;

// S holds two triangles at once: entries (i, c <= i) are the block of Ky being turned into
// L11; entries (r, c > r) are the rows of the appended identity being turned into L11^-T
// (its diagonal lives in bd[]).  Eliminating 16 columns at a time:
//   phase 1  one wave factors the 16x16 diagonal tile in registers (row per lane, pivots and
//            multipliers broadcast with v_readlane) -- this is the serial pivot chain;
//   phase 2  every other row (112-c0 rows below + c0+16 identity rows above = 128 rows)
//            forward-substitutes its 16 panel entries against that tile;
//   phase 3  rank-16 update of everything to the right of the panel.
__global__ __launch_bounds__(256) void potf2_aug_kernel(double* Akk, int64_t lda, double* Wkk, int64_t ldw,
                                                        double* W11, int* info, int kblock) {
  extern __shared__ __align__(16) double sm[];
  double* S = sm;
  double* bd = S + NB * SP;
  double* rinvs = bd + NB;
  double* Lt = rinvs + 16;   // 16 x 17
  double* P2 = Lt + 16 * 17; // 144 x PP
  const int tid = threadIdx.x;

  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 7, c = e & 127;
    S[i * SP + c] = (c <= i) ? Akk[(int64_t)i * lda + c] : 0.0;
  }
  if (tid < NB) bd[tid] = 1.0;
  __syncthreads();

  for (int p = 0; p < NB / 16; ++p) {
    const int c0 = 16 * p;
    // ---- phase 1: 16x16 tile, wave 0, lane i < 16 owns row c0 + i
    if (tid < 64) {
      const int li = tid & 15;
      double a[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = (c <= li) ? S[(c0 + li) * SP + c0 + c] : 0.0;
      double rinv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double pj = readlane_f64(a[j], j);
        double sj, yj;
        if (pj > 0.0) {
          sqrt_rsqrt(pj, sj, yj);
        } else {  // not positive definite (or NaN): record, keep going with zeros
          sj = 0.0;
          yj = 0.0;
          if (tid == 0) atomicCAS(info, 0, kblock * NB + c0 + j + 1);
        }
        rinv[j] = yj;
        a[j] = (li == j) ? sj : a[j] * yj;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
          const double lcj = readlane_f64(a[j], c);
          a[c] = fma(-a[j], lcj, a[c]);
        }
      }
      if (tid < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const double v = (c <= li) ? a[c] : 0.0;
          Lt[li * 17 + c] = v;
          if (c <= li) S[(c0 + li) * SP + c0 + c] = v;
        }
        rinvs[li] = rinv[li];
      }
    }
    __syncthreads();
    // ---- phase 2: forward substitution of the other 128 rows' panel entries
    const int ntop = NB - 16 - c0;  // rows below the tile
    if (tid < NB) {
      const bool top = tid < ntop;
      const int row = top ? (c0 + 16 + tid) : (tid - ntop);  // S row index (bottom: identity row r)
      double x[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = c0 + c;
        if (top)
          x[c] = S[row * SP + col];
        else
          x[c] = col > row ? S[row * SP + col] : (col == row ? bd[row] : 0.0);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        x[j] *= rinvs[j];
#pragma unroll
        for (int c = j + 1; c < 16; ++c) x[c] = fma(-x[j], Lt[c * 17 + j], x[c]);
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = c0 + c;
        P2[tid * PP + c] = x[c];
        if (top || col > row)
          S[row * SP + col] = x[c];
        else if (col == row)
          bd[row] = x[c];
      }
    }
    __syncthreads();
    // ---- phase 3: rank-16 update to the right of the panel
    const int nq = NB - 16 - c0;  // remaining columns
    if (nq > 0) {
      const int xr = tid & 127, half = tid >> 7;
      const bool top = xr < ntop;
      const int row = top ? (c0 + 16 + xr) : (xr - ntop);
      double a[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = P2[xr * PP + c];
      const int qmax = top ? (xr + 1) : nq;  // lower triangle only for the Cholesky rows
      for (int q = half; q < qmax; q += 2) {
        const double* bq = P2 + q * PP;  // L[c0+16+q][panel]
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc = fma(a[c], bq[c], acc);
        S[row * SP + c0 + 16 + q] -= acc;
      }
    }
    __syncthreads();
  }

  // ---- write out: L11 (lower), L11^-T into WT's diagonal block (upper, zero below), W11 = L11^-1
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e >> 7, c = e & 127;
    if (c <= i) Akk[(int64_t)i * lda + c] = S[i * SP + c];
    Wkk[(int64_t)i * ldw + c] = c > i ? S[i * SP + c] : (c == i ? bd[i] : 0.0);
    // W11[i][c] = L11^-1[i][c] = (L11^-T)[c][i]
    W11[i * NB + c] = c < i ? S[c * SP + i] : (c == i ? bd[i] : 0.0);
  }
}

// --------------------------------------------------------------- panel solve on the matrix cores
// Row block list for block column k: A row blocks k+1..nb-1, the y block, then WT row blocks 0..k-1.
// Each workgroup: P <- P * W11^T, i.e. P[x][c] = sum_j P[x][j] W11[c][j]  (NT GEMM, K = 128, in place).
struct PanelArgs {
  double* A;
  double* WT;
  const double* W11;
  int64_t lda;
  int k, nb;
};

__device__ __forceinline__ double* panel_block(const PanelArgs& P, int idx) {
  const int nbelow = P.nb - 1 - P.k;
  if (idx < nbelow) return P.A + ((int64_t)(P.k + 1 + idx) * NB) * P.lda + (int64_t)P.k * NB;
  if (idx == nbelow) return P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)P.k * NB;  // y block
  return P.WT + ((int64_t)(idx - nbelow - 1) * NB) * P.lda + (int64_t)P.k * NB;
}

__global__ __launch_bounds__(256) void trsm_gemm_kernel(PanelArgs P) {
  extern __shared__ __align__(16) double lds[];
  double* Pb = panel_block(P, blockIdx.x);
  GemmAcc acc;
  acc.zero();
  gemm_tile_nt(acc, Pb, P.lda, P.W11, NB, 0, NB, lds, false);
  acc_foreach(acc, [&](int row, int col, double v) { Pb[(int64_t)row * P.lda + col] = v; });
}

// --------------------------------------------------------------- trailing update on the matrix cores
// Tiles for block column k (m = nb-1-k remaining block columns):
//   [0, tA)            Cholesky rows: (i, c), k < c <= i < nb      C -= P_i P_c^T   (SYRK)
//   [tA, tA+m)         y block:       (y, c)                        C -= P_y P_c^T
//   [tA+m, ...)        L^-T rows:     (r, c), r <= k                C  = beta C - P_r P_c^T, beta = 0 for r == k
__global__ __launch_bounds__(256) void trailing_update_kernel(PanelArgs P) {
  extern __shared__ __align__(16) double lds[];
  const int m = P.nb - 1 - P.k;
  const int tA = m * (m + 1) / 2;
  int idx = blockIdx.x;
  const double* Ap;
  double* C;
  int cblk;
  bool same = false, beta0 = false;
  if (idx < tA) {
    int i = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
    while (i * (i + 1) / 2 > idx) --i;
    while ((i + 1) * (i + 2) / 2 <= idx) ++i;
    const int c = idx - i * (i + 1) / 2;
    const int ib = P.k + 1 + i;
    cblk = P.k + 1 + c;
    Ap = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)P.k * NB;
    C = P.A + ((int64_t)ib * NB) * P.lda + (int64_t)cblk * NB;
    same = (i == c);
  } else if (idx < tA + m) {
    cblk = P.k + 1 + (idx - tA);
    Ap = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)P.k * NB;
    C = P.A + ((int64_t)P.nb * NB) * P.lda + (int64_t)cblk * NB;
  } else {
    idx -= tA + m;
    const int r = idx / m;
    cblk = P.k + 1 + (idx - r * m);
    Ap = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)P.k * NB;
    C = P.WT + ((int64_t)r * NB) * P.lda + (int64_t)cblk * NB;
    beta0 = (r == P.k);
  }
  const double* Bp = P.A + ((int64_t)cblk * NB) * P.lda + (int64_t)P.k * NB;
  GemmAcc acc;
  acc.zero();
  gemm_tile_nt(acc, Ap, P.lda, Bp, P.lda, 0, NB, lds, same);
  if (beta0)
    acc_foreach(acc, [&](int row, int col, double v) { C[(int64_t)row * P.lda + col] = -v; });
  else
    acc_foreach(acc, [&](int row, int col, double v) { C[(int64_t)row * P.lda + col] -= v; });
}

// --------------------------------------------------------------- alpha, logdet, y^T K^-1 y
// alpha_i = sum_{k >= i} WT[i][k] z_k : one wavefront per row, coalesced along k.
__global__ void alpha_kernel(const double* WT, const double* z, double* alpha, int64_t n, int64_t np, int64_t lda) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= np) return;
  double s = 0.0;
  if (row < n) {
    const double* w = WT + row * lda;
    for (int64_t k = (row & ~(int64_t)63) + lane; k < n; k += 64)
      if (k >= row) s += w[k] * z[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  }
  if (lane == 0) alpha[row] = s;
}

// red[0] = sum log L_ii (i < n), red[1] = sum z_i^2.  Single workgroup, fixed order.
__global__ void logdet_kernel(const double* A, const double* z, double* red, int64_t n, int64_t lda) {
  __shared__ double s0[256], s1[256];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    a += log(A[i * lda + i]);
    b += z[i] * z[i];
  }
  s0[threadIdx.x] = a;
  s1[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      s0[threadIdx.x] += s0[threadIdx.x + off];
      s1[threadIdx.x] += s1[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    red[0] = s0[0];
    red[1] = s1[0];
  }
}

template <class K>
static int enable_lds(elfihip_ctx* ctx, K k, size_t bytes) {
  ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return ELFIHIP_OK;
}

int gp_factorize_impl(elfihip_gp* gp) {
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, gp->n > 0, "GP has no evidence");
  hipStream_t st = ctx->stream;
  const int64_t np = gp->np;
  const int nb = (int)(np / NB);
  ELFIHIP_CHECK_HIP(ctx, hipMemsetAsync(gp->info, 0, sizeof(int), st));
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
    G.diag_add = gp->noise + GP_JITTER;
    const int64_t nt = np / 64;
    const size_t lds = 2 * 64 * (size_t)(gp->dp + 1) * sizeof(double);
    hipLaunchKernelGGL(gram_kernel, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), lds, st, G);
    ELFIHIP_TRY(launch_status(ctx, "gram_kernel"));
  }
  const size_t potf2_lds = POTF2_LDS_DOUBLES * sizeof(double);
  const size_t gemm_lds = GEMM_LDS_DOUBLES * sizeof(double);
  ELFIHIP_TRY(enable_lds(ctx, potf2_aug_kernel, potf2_lds));
  PanelArgs P;
  P.A = gp->A;
  P.WT = gp->WT;
  P.W11 = gp->W11;
  P.lda = gp->lda;
  P.nb = nb;
  for (int k = 0; k < nb; ++k) {
    P.k = k;
    double* Akk = gp->A + ((int64_t)k * NB) * gp->lda + (int64_t)k * NB;
    double* Wkk = gp->WT + ((int64_t)k * NB) * gp->lda + (int64_t)k * NB;
    hipLaunchKernelGGL(potf2_aug_kernel, dim3(1), dim3(256), potf2_lds, st, Akk, gp->lda, Wkk, gp->lda, gp->W11,
                       gp->info, k);
    const int nrows = (nb - 1 - k) + 1 + k;  // below + y block + L^-T rows above
    hipLaunchKernelGGL(trsm_gemm_kernel, dim3(nrows), dim3(256), gemm_lds, st, P);
    const int m = nb - 1 - k;
    if (m > 0) {
      const int tiles = m * (m + 1) / 2 + m + (k + 1) * m;
      hipLaunchKernelGGL(trailing_update_kernel, dim3(tiles), dim3(256), gemm_lds, st, P);
    }
  }
  ELFIHIP_TRY(launch_status(ctx, "cholesky sweep"));
  const double* z = gp->A + np * gp->lda;  // row np of A: z = L^-1 y
  hipLaunchKernelGGL(alpha_kernel, dim3((unsigned)((np * 64 + 255) / 256)), dim3(256), 0, st, gp->WT, z, gp->alpha,
                     gp->n, np, gp->lda);
  hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, st, gp->A, z, gp->red, gp->n, gp->lda);
  ELFIHIP_TRY(launch_status(ctx, "alpha/logdet"));
  double red[2];
  int info = 0;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(red, gp->red, sizeof red, hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(&info, gp->info, sizeof info, hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  if (info != 0) {
    gp->factored = false;
    return fail(ctx, ELFIHIP_ERR_NOT_PD, "covariance matrix is not positive definite (pivot %d <= 0)", info);
  }
  gp->logdet = 2.0 * red[0];
  gp->yKy = red[1];
  gp->factored = true;
  gp->has_kinv = false;
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
  alloc(&gp->W11, (size_t)NB * NB * sizeof(double));
  alloc(&gp->alpha, (size_t)gp->cap * sizeof(double));
  alloc(&gp->red, 64 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&gp->info), sizeof(int));
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
  for (double* p : {gp->X, gp->x2, gp->y, gp->A, gp->WT, gp->Kinv, gp->W11, gp->alpha, gp->red})
    if (p) (void)hipFree(p);
  if (gp->info) (void)hipFree(gp->info);
  gp->ws.release();
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
  return ELFIHIP_OK;
}

static int gp_copy_rows(elfihip_gp* gp, const double* X, const double* y, int64_t at, int64_t k) {
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, at + k <= gp->cap, "evidence count %lld exceeds the GP capacity %lld", (long long)(at + k),
                  (long long)gp->cap);
  if (k == 0) return ELFIHIP_OK;
  ELFIHIP_REQUIRE(ctx, X && y, "NULL data pointer");
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
