// Streaming row tiles of a row-major f64 matrix through LDS (shared by the distance and the
// summary kernels): a workgroup reads a tile of rows as one contiguous span with coalesced loads,
// drops it into LDS with an odd row pitch, and then every lane owns one whole row, so per-row
// arithmetic can follow the reference's sequential order exactly.
#pragma once

#include "common.hpp"

namespace elfihip {

struct FastDiv {
  uint32_t mul, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint64_t q = (1ull << 32) / d;
  f.mul = q > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)q;
  return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t x, FastDiv f) {
  // floor(2^32/d) under-estimates by at most one; a single fix-up makes it exact.
  uint32_t q = __umulhi(x, f.mul);
  if (x - q * f.d >= f.d) ++q;
  return q;
}

// Fused selection (reject.hip): a distance kernel that is handed a filter also appends every row whose distance is
// below the running threshold -- the current k-th best distance of the sampler state, what Rejection._merge_batch
// (elfi/methods/inference/samplers.py:209-237) would keep of the batch -- to a candidate list, so that the selection
// after the distance costs one small merge instead of passes over all n distances.  thr == nullptr: no filter.
struct RejectFilter {
  const double* thr;          // device scalar: rows with d < *thr are candidates
  double* cval;               // candidate distances
  long long* crow;            // candidate row numbers (row_base + row inside this batch)
  unsigned int* count;        // candidates offered so far (may exceed cap: the list holds the first cap)
  unsigned int cap;
  long long row_base;
};

// Append (d, row) for the lanes with `hit`: one atomic per wave (ballot + prefix count), all lanes of the wave must call.
__device__ __forceinline__ void reject_offer(const RejectFilter& F, bool hit, double d, long long row) {
  const unsigned long long mask = __ballot(hit);
  if (mask == 0) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  unsigned int base = 0;
  if (lane == leader) base = atomicAdd(F.count, (unsigned int)__popcll(mask));
  base = __shfl(base, leader, 64);
  if (hit) {
    const unsigned int idx = base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
    if (idx < F.cap) {
      F.cval[idx] = d;
      F.crow[idx] = row;
    }
  }
}

struct RowArgs {
  const double* X;
  int64_t n;
  int64_t ldx;
  const double* y;
  const double* aux;
  double* out;
  double p, inv_p;
  int m, mp;      // mp = m | 1: LDS row pitch in doubles
  int K;          // multi-weight: number of weight rows in aux
  int vec2;       // 16-byte loads are legal (m, ldx even; X 16-byte aligned)
  int R;          // pipelined kernels: rows per tile (<= blockDim.x); T * U >= R * m / 2
  int nt;         // pipelined kernels: non-temporal loads
  FastDiv div_h;  // by m/2 (vec2) or m
  RejectFilter F; // fused selection (thr == nullptr: off)
};

// Stream one tile of `rows` rows starting at row0 into LDS (pitch mp).
template <int U>
__device__ __forceinline__ void load_tile(const RowArgs& A, double* tile, int64_t row0, int rows) {
  const int T = blockDim.x, tid = threadIdx.x;
  const double* __restrict__ X = A.X + row0 * A.ldx;
  if (A.vec2) {
    const uint32_t h = (uint32_t)A.m >> 1;
    const uint32_t npairs = (uint32_t)rows * h;
    for (uint32_t base = 0; base < npairs; base += (uint32_t)(T * U)) {
      double2 v[U];
      uint32_t r[U], jj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t idx = base + (uint32_t)(u * T + tid);
        bool ok = idx < npairs;
        uint32_t q = fastdiv(ok ? idx : 0u, A.div_h);
        r[u] = q;
        jj[u] = (ok ? idx : 0u) - q * h;
        if (ok)
          v[u] = *reinterpret_cast<const double2*>(X + (int64_t)q * A.ldx + 2 * jj[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t idx = base + (uint32_t)(u * T + tid);
        if (idx < npairs) {
          double* dst = tile + r[u] * (uint32_t)A.mp + 2 * jj[u];
          dst[0] = v[u].x;
          dst[1] = v[u].y;
        }
      }
    }
  } else {
    const uint32_t m = (uint32_t)A.m;
    const uint32_t nel = (uint32_t)rows * m;
    for (uint32_t base = 0; base < nel; base += (uint32_t)(T * U)) {
      double v[U];
      uint32_t r[U], j[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t idx = base + (uint32_t)(u * T + tid);
        bool ok = idx < nel;
        uint32_t q = fastdiv(ok ? idx : 0u, A.div_h);
        r[u] = q;
        j[u] = (ok ? idx : 0u) - q * m;
        if (ok) v[u] = X[(int64_t)q * A.ldx + j[u]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint32_t idx = base + (uint32_t)(u * T + tid);
        if (idx < nel) tile[r[u] * (uint32_t)A.mp + j[u]] = v[u];
      }
    }
  }
}

// ---- software-pipelined tile streaming (16-byte loads, whole tile in one batch) -------------
// fetch: issue the global loads of one tile into registers (nothing waits on them here);
// commit: drop the registers into the LDS tile.  With the loads of tile t+1 issued before the
// row sums of tile t are computed, every workgroup keeps a full tile of HBM requests in flight
// while it does its LDS/VALU work, instead of alternating between the two.
template <int U>
__device__ __forceinline__ void tile_fetch(const RowArgs& A, int64_t row0, int rows, double2 (&v)[U]) {
  const int T = blockDim.x, tid = threadIdx.x;
  const double* __restrict__ X = A.X + row0 * A.ldx;
  const uint32_t h = (uint32_t)A.m >> 1;
  const uint32_t npairs = (uint32_t)rows * h;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t idx = (uint32_t)(u * T + tid);
    const bool ok = idx < npairs;
    const uint32_t q = fastdiv(ok ? idx : 0u, A.div_h);
    const uint32_t jj = (ok ? idx : 0u) - q * h;
    v[u] = make_double2(0.0, 0.0);
    if (ok) {
      const double2* src = reinterpret_cast<const double2*>(X + (int64_t)q * A.ldx + 2 * jj);
      if (A.nt) {  // streamed once: do not keep the lines in L2 / MALL (ONE 16-byte load: the builtin on the scalar members
                   // made two 8-byte loads of it, which is why round 2 measured no gain from it)
        typedef double v2d_nt __attribute__((ext_vector_type(2)));
        const v2d_nt t = __builtin_nontemporal_load(reinterpret_cast<const v2d_nt*>(src));
        v[u] = make_double2(t.x, t.y);
      } else {
        v[u] = *src;
      }
    }
  }
}

template <int U>
__device__ __forceinline__ void tile_commit(const RowArgs& A, double* tile, int rows, const double2 (&v)[U]) {
  const int T = blockDim.x, tid = threadIdx.x;
  const uint32_t h = (uint32_t)A.m >> 1;
  const uint32_t npairs = (uint32_t)rows * h;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t idx = (uint32_t)(u * T + tid);
    if (idx < npairs) {
      const uint32_t q = fastdiv(idx, A.div_h);
      const uint32_t jj = idx - q * h;
      double* dst = tile + q * (uint32_t)A.mp + 2 * jj;
      dst[0] = v[u].x;
      dst[1] = v[u].y;
    }
  }
}


#if defined(__HIPCC__)
// ---- LDS-DMA streaming of row slots (round 5; used by distance.hip and adaptive.hip) ----------------------------------------
// A slot is ROWS rows of MM doubles = ROWS * MM / 128 DMA pieces of 1 KiB (64 lanes x 16 bytes); its LDS image is
// lane-linear with the 16-byte granules of a row XOR-swizzled on the SOURCE address and again on the read (see
// dist_rows_dma_kernel in distance.hip for the design and the measurements).
template <bool NT>
__device__ __forceinline__ void dma16_to_lds(const void* gsrc, unsigned lds_dst) {
  unsigned keep;   // M0 carries the LDS destination of the DMA; the compiler owns M0, so save / restore in one statement
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// granule swizzle: rows at pitch MM doubles; RP rows share one 256-byte bank sweep, the key changes every RP rows
template <int MM>
__device__ __forceinline__ int dma_swizzle_key(int row) {
  constexpr int H = MM / 2;                      // 16-byte granules per row
  constexpr int RP = H >= 16 ? 1 : 16 / H;
  constexpr int MASK = (H >= 16 ? 16 : H) - 1;
  return (row / RP) & MASK;
}
// ... with the source given as a wave-uniform base (SGPR pair) + a per-lane 32-bit byte offset: the offsets of a slot's pieces
// are the same for every slot, so a slot costs five scalar instructions per piece and no vector arithmetic (with per-lane
// 64-bit addresses -- a multiply by the row pitch and a clamp per piece -- address generation was a third of a wave's time
// and the kernel, at two waves per CU, was bound by it: 47-53 us in the library against 41 us for the probe's constant pitch)
template <bool NT>
__device__ __forceinline__ void dma16_to_lds_off(const void* base, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int MM, int ROWS>
__device__ __forceinline__ void dma_issue_slot(const RowArgs& A, const unsigned (&off)[ROWS * (MM / 2) / 64], int64_t row0,
                                               unsigned lds_slot, int lane) {
  constexpr int H = MM / 2;
  constexpr int PIECES = ROWS * H / 64;
  if (row0 + ROWS <= A.n) {
    const uint64_t p = (uint64_t)(A.X + row0 * A.ldx);   // wave-uniform: make the compiler keep it in SGPRs
    const uint64_t b = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)p) |
                       ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) << 32);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) dma16_to_lds_off<true>((const void*)b, off[i], lds_slot + (unsigned)i * 1024u);
  } else {
    // the ragged last slot: rows beyond n read row n - 1 (a valid address), their results are discarded
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int G = i * 64 + lane;
      const int row = G / H, g = G % H;
      int64_t gr = row0 + row;
      if (gr >= A.n) gr = A.n - 1;
      dma16_to_lds<true>(A.X + gr * A.ldx + 2 * (g ^ dma_swizzle_key<MM>(row)), lds_slot + (unsigned)i * 1024u);
    }
  }
}

// The K results of each of a wave's 64 consecutive rows -> `dst` (= out + first row * K, 16-byte aligned) as ONE contiguous
// block: the lanes deal their K values into the wave's LDS stage ([row][k], as the block lies in memory) and the wave writes
// the block back out in 16-byte pieces, 1 KiB per store instruction.  (Lane r storing its own K doubles is K instructions of
// 64 eight-byte pieces at stride 8 K: at K = 3 the narrow-row kernels wrote their 96 MB at 3 TB/s.)  nv: live rows of the
// block (rows >= nv are not stored); stage: 64 * KMAX doubles owned by this wave.
template <int KMAX>
__device__ __forceinline__ void wave_store_rows(double* stage, double* dst, const double (&vals)[KMAX], int K, int lane, int nv) {
  typedef double v2d_st __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) stage[lane * K + k] = vals[k];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int total = nv * K, npieces = total >> 1;
#pragma unroll
  for (int i = 0; i < (KMAX + 1) / 2; ++i) {
    const int p = lane + 64 * i;
    if (p < npieces) *reinterpret_cast<v2d_st*>(dst + 2 * p) = *reinterpret_cast<const v2d_st*>(stage + 2 * p);
  }
  if ((total & 1) && lane == 0) dst[total - 1] = stage[total - 1];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();   // the stage is free again
}

#endif

static inline bool tile_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace elfihip
