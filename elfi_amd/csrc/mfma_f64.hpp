// f64 matrix-core building block for gfx950: one 128x128 output tile per 256-thread
// workgroup, computed with v_mfma_f64_16x16x4_f64.
//
//   acc[i][j] += sum_k A[i][k] * B[j][k]          ("NT": both operands row-major, k contiguous)
//
// Layout facts used here (cdna_hip_programming.md section 3, f64 exception included):
//   A operand  lane l holds A[i = l & 15][k = l >> 4]      (one f64 per lane)
//   B operand  lane l holds B[k = l >> 4][j = l & 15]
//   C/D        4 f64 per lane: element r is D[row = (l >> 4) + 4 r][col = l & 15]
//
// Workgroup = 4 waves as 2 x 2, each wave owns a 64x64 sub-tile = 4 x 4 MFMA tiles
// (16 accumulators of 4 f64 = 128 VGPRs).  Operand tiles are staged global -> registers
// -> LDS as [row][k] with a 20-double pitch: 16-byte ds_write_b128 stores are conflict
// free (8 consecutive lanes = one 128-byte row) and the fragment loads are 16-byte
// ds_read_b128 of a k-PAIR per lane, conflict free for the 4 x 16 lane groups of that
// instruction (slot = 10*row + q + 4h hits 16 distinct slots).  A lane's pair (k, k+1)
// feeds two consecutive MFMAs, so MFMA step (h, e) contracts k in {8h + 2q + e}: a
// permutation of k that A and B share, which leaves the product unchanged.
// The next k-tile is prefetched into registers while the current one is multiplied.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace elfihip {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int GT = 128;        // output tile edge
constexpr int GK = 16;         // k-tile depth
constexpr int GLP = 20;        // LDS row pitch (doubles)
constexpr int GEMM_LDS_DOUBLES = 2 * GT * GLP;  // A tile + B tile
constexpr int GEMM_THREADS = 256;

struct GemmAcc {
  v4d c[4][4];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
};

// Multiply the staged k-tile: 16 ds_read_b128 and 64 MFMAs per wave.
__device__ __forceinline__ void mma_ktile(GemmAcc& acc, const double* As, const double* Bs) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = w >> 1, wc = w & 1;
  const double* a0 = As + (wr * 64 + (l & 15)) * GLP + 2 * (l >> 4);
  const double* b0 = Bs + (wc * 64 + (l & 15)) * GLP + 2 * (l >> 4);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    double2 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = *reinterpret_cast<const double2*>(a0 + i * 16 * GLP + 8 * h);
      b[i] = *reinterpret_cast<const double2*>(b0 + i * 16 * GLP + 8 * h);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc.c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc.c[i][j], 0, 0, 0);
        acc.c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc.c[i][j], 0, 0, 0);
      }
  }
}

// acc += A(128 x K) * B(128 x K)^T for k in [kbeg, kend), both multiples of 16, kbeg < kend.
// `lds` holds GEMM_LDS_DOUBLES doubles.  SAME: the B tile is the A tile (diagonal SYRK tile).
template <bool SAME>
__device__ __forceinline__ void gemm_tile_nt(GemmAcc& acc, const double* __restrict__ A, int64_t lda,
                                             const double* __restrict__ B, int64_t ldb, int kbeg, int kend,
                                             double* lds) {
  double* As = lds;
  double* Bs = SAME ? lds : lds + GT * GLP;
  const int t = threadIdx.x;
  const double* pa = A + (int64_t)(t >> 3) * lda + 2 * (t & 7) + kbeg;
  const double* pb = B + (int64_t)(t >> 3) * ldb + 2 * (t & 7) + kbeg;
  double* da = As + (t >> 3) * GLP + 2 * (t & 7);
  double* db = Bs + (t >> 3) * GLP + 2 * (t & 7);
  double2 a0, a1, a2, a3, b0, b1, b2, b3;
  a0 = *reinterpret_cast<const double2*>(pa);
  a1 = *reinterpret_cast<const double2*>(pa + 32 * lda);
  a2 = *reinterpret_cast<const double2*>(pa + 64 * lda);
  a3 = *reinterpret_cast<const double2*>(pa + 96 * lda);
  if (!SAME) {
    b0 = *reinterpret_cast<const double2*>(pb);
    b1 = *reinterpret_cast<const double2*>(pb + 32 * ldb);
    b2 = *reinterpret_cast<const double2*>(pb + 64 * ldb);
    b3 = *reinterpret_cast<const double2*>(pb + 96 * ldb);
  }
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();  // previous tile fully consumed
    *reinterpret_cast<double2*>(da) = a0;
    *reinterpret_cast<double2*>(da + 32 * GLP) = a1;
    *reinterpret_cast<double2*>(da + 64 * GLP) = a2;
    *reinterpret_cast<double2*>(da + 96 * GLP) = a3;
    if (!SAME) {
      *reinterpret_cast<double2*>(db) = b0;
      *reinterpret_cast<double2*>(db + 32 * GLP) = b1;
      *reinterpret_cast<double2*>(db + 64 * GLP) = b2;
      *reinterpret_cast<double2*>(db + 96 * GLP) = b3;
    }
    __syncthreads();
    if (k0 + GK < kend) {  // prefetch the next k-tile; it lands while the MFMAs below run
      pa += GK;
      a0 = *reinterpret_cast<const double2*>(pa);
      a1 = *reinterpret_cast<const double2*>(pa + 32 * lda);
      a2 = *reinterpret_cast<const double2*>(pa + 64 * lda);
      a3 = *reinterpret_cast<const double2*>(pa + 96 * lda);
      if (!SAME) {
        pb += GK;
        b0 = *reinterpret_cast<const double2*>(pb);
        b1 = *reinterpret_cast<const double2*>(pb + 32 * ldb);
        b2 = *reinterpret_cast<const double2*>(pb + 64 * ldb);
        b3 = *reinterpret_cast<const double2*>(pb + 96 * ldb);
      }
    }
    mma_ktile(acc, As, Bs);
  }
}

// Visit every accumulator element with its (row, col) inside the 128 x 128 tile.
template <class F>
__device__ __forceinline__ void acc_foreach(const GemmAcc& acc, F f) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = w >> 1, wc = w & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * 64 + i * 16 + (l >> 4) + 4 * r;
        const int col = wc * 64 + j * 16 + (l & 15);
        f(row, col, acc.c[i][j][r]);
      }
}


// ---- skinny variant: 32 x 128 output tile (panel solve: many workgroups, short latency) ----
// 4 waves side by side, each 32 rows x 32 columns = 2 x 2 MFMA tiles.
struct GemmAcc32 {
  v4d c[2][2];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) c[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
};

constexpr int GEMM32_LDS_DOUBLES = (32 + GT) * GLP;

// acc += A(32 x K) * B(128 x K)^T, k in [kbeg, kend).
__device__ __forceinline__ void gemm_tile32_nt(GemmAcc32& acc, const double* __restrict__ A, int64_t lda,
                                               const double* __restrict__ B, int64_t ldb, int kbeg, int kend,
                                               double* lds) {
  double* As = lds;
  double* Bs = lds + 32 * GLP;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const double* pa = A + (int64_t)(t >> 3) * lda + 2 * (t & 7) + kbeg;
  const double* pb = B + (int64_t)(t >> 3) * ldb + 2 * (t & 7) + kbeg;
  double* da = As + (t >> 3) * GLP + 2 * (t & 7);
  double* db = Bs + (t >> 3) * GLP + 2 * (t & 7);
  double2 a0, b0, b1, b2, b3;
  a0 = *reinterpret_cast<const double2*>(pa);
  b0 = *reinterpret_cast<const double2*>(pb);
  b1 = *reinterpret_cast<const double2*>(pb + 32 * ldb);
  b2 = *reinterpret_cast<const double2*>(pb + 64 * ldb);
  b3 = *reinterpret_cast<const double2*>(pb + 96 * ldb);
  const double* fa = As + (l & 15) * GLP + 2 * (l >> 4);
  const double* fb = Bs + (w * 32 + (l & 15)) * GLP + 2 * (l >> 4);
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();
    *reinterpret_cast<double2*>(da) = a0;
    *reinterpret_cast<double2*>(db) = b0;
    *reinterpret_cast<double2*>(db + 32 * GLP) = b1;
    *reinterpret_cast<double2*>(db + 64 * GLP) = b2;
    *reinterpret_cast<double2*>(db + 96 * GLP) = b3;
    __syncthreads();
    if (k0 + GK < kend) {
      pa += GK;
      pb += GK;
      a0 = *reinterpret_cast<const double2*>(pa);
      b0 = *reinterpret_cast<const double2*>(pb);
      b1 = *reinterpret_cast<const double2*>(pb + 32 * ldb);
      b2 = *reinterpret_cast<const double2*>(pb + 64 * ldb);
      b3 = *reinterpret_cast<const double2*>(pb + 96 * ldb);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double2 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const double2*>(fa + i * 16 * GLP + 8 * h);
        b[i] = *reinterpret_cast<const double2*>(fb + i * 16 * GLP + 8 * h);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc.c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc.c[i][j], 0, 0, 0);
          acc.c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc.c[i][j], 0, 0, 0);
        }
    }
  }
}

template <class F>
__device__ __forceinline__ void acc32_foreach(const GemmAcc32& acc, F f) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) f(i * 16 + (l >> 4) + 4 * r, w * 32 + j * 16 + (l & 15), acc.c[i][j][r]);
}

// ---- staging constants of the fused step kernel (gp_fit.hip: step_kernel): k-tiles are 32 deep (two 16-byte loads of A
// and of B per thread and k-tile).  LDS rows have a 36-double pitch: 16-byte slot = 18 row + q + 4 h, i.e. (2 row + q)
// mod 16 -- distinct for the 4 x 16 lane groups of ds_read_b128 -- and the ds_write_b128 stores of 8 consecutive lanes
// fill one 128-byte stretch of a row.
constexpr int GK2 = 32;
constexpr int GLP2 = 36;
}  // namespace elfihip
