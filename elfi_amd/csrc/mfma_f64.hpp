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

// Global -> register stage of one 128 x 16 operand tile (rows r0.., k-columns k0..k0+15).
// Thread t loads rows (t >> 3) + 32 p, k-pair (t & 7): 8 lanes cover one 128-byte row.
struct StageRegs {
  double2 v[4];
};

__device__ __forceinline__ void stage_load(StageRegs& s, const double* __restrict__ P, int64_t ld, int k0) {
  const int t = threadIdx.x;
  const double* src = P + (int64_t)(t >> 3) * ld + k0 + 2 * (t & 7);
#pragma unroll
  for (int p = 0; p < 4; ++p) s.v[p] = *reinterpret_cast<const double2*>(src + (int64_t)(32 * p) * ld);
}

__device__ __forceinline__ void stage_store(const StageRegs& s, double* tile) {
  const int t = threadIdx.x;
  double* dst = tile + (t >> 3) * GLP + 2 * (t & 7);
#pragma unroll
  for (int p = 0; p < 4; ++p) *reinterpret_cast<double2*>(dst + 32 * p * GLP) = s.v[p];
}

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

// acc += A(128 x K) * B(128 x K)^T for k in [kbeg, kend), both multiples of 16.
// `lds` holds GEMM_LDS_DOUBLES doubles.  same_ab: B tile is the A tile (diagonal SYRK tile).
__device__ __forceinline__ void gemm_tile_nt(GemmAcc& acc, const double* __restrict__ A, int64_t lda,
                                             const double* __restrict__ B, int64_t ldb, int kbeg, int kend,
                                             double* lds, bool same_ab) {
  double* As = lds;
  double* Bs = same_ab ? lds : lds + GT * GLP;
  StageRegs ra, rb;
  if (kbeg < kend) {
    stage_load(ra, A, lda, kbeg);
    if (!same_ab) stage_load(rb, B, ldb, kbeg);
  }
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    __syncthreads();  // previous tile fully consumed
    stage_store(ra, As);
    if (!same_ab) stage_store(rb, Bs);
    __syncthreads();
    if (k0 + GK < kend) {
      stage_load(ra, A, lda, k0 + GK);
      if (!same_ab) stage_load(rb, B, ldb, k0 + GK);
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

}  // namespace elfihip
