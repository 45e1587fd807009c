// Developer probe (round 5): the row-distance stream of configs[1] (10^6 x 32 f64, euclidean) with the tile brought
// into LDS by LDS-DMA (global_load_lds_dwordx4) instead of through registers.  Every WAVE owns a ring of D slots of
// ROWS x 32 doubles; the slot image is lane-linear (what LDS-DMA writes), the column pairs of a row are XOR-swizzled
// on the SOURCE address so that the 64 lanes' ds_read_b128 of "their" row do not meet in a bank; lane r sums row r
// left to right (cdist's order).  No workgroup barrier: a wave reads only what it issued itself, behind a counted vmcnt.
//   hipcc --offload-arch=gfx950 -O3 -o glds_probe glds_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#pragma clang fp contract(off)

#define CK(x)                                                                \
  do {                                                                       \
    hipError_t e_ = (x);                                                     \
    if (e_ != hipSuccess) {                                                  \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int M = 32;   // reference / register pipeline width; the LDS-DMA kernel is templated on its own width

__global__ __launch_bounds__(256) void ref_kernel(const double* X, int64_t n, const double* y, double* out) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
  for (int j = 0; j < M; ++j) {
    const double d = X[r * M + j] - y[j];
    s = s + d * d;
  }
  out[r] = sqrt(s);
}

template <bool NT>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// LDS-DMA with the source as (wave-uniform base in SGPRs) + (per-lane 32-bit byte offset): the offsets of a slot's pieces
// are the same for every slot, so they are computed ONCE and a slot costs no vector arithmetic at all
template <bool NT>
__device__ __forceinline__ void glds16_off(const void* base, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

template <int MM, int ROWS, int D, bool NT, bool WGT>
__global__ __launch_bounds__(256) void glds_kernel(const double* __restrict__ X, int64_t n, int64_t ldx,
                                                   const double* __restrict__ y, const double* __restrict__ wgt,
                                                   double* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  constexpr int SLOT = ROWS * MM * 8;
  constexpr int H = MM / 2;
  constexpr int PIECES = ROWS * H / 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  unsigned char* mine = lds_raw + (size_t)wave * D * SLOT;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)mine);
  double* ys = reinterpret_cast<double*>(lds_raw + (size_t)nw * D * SLOT);
  double* ws = ys + MM;
  const int64_t nslots = (n + ROWS - 1) / ROWS;
  const int64_t stride = (int64_t)gridDim.x * nw;
  int64_t t = (int64_t)blockIdx.x * nw + wave;
  for (int j = threadIdx.x; j < MM; j += blockDim.x) {
    ys[j] = y[j];
    if constexpr (WGT) ws[j] = wgt[j];
  }
  __syncthreads();
  unsigned off[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int G = i * 64 + lane;
    const int row = G / H, g = G % H;
    off[i] = (unsigned)(((int64_t)row * ldx + 2 * (g ^ (row & 15))) * 8);
  }
  auto issue = [&](int64_t tk, int slot) {
    const int64_t row0 = tk * ROWS;
    const unsigned dst = lds_base + (unsigned)slot * SLOT;
    if (row0 + ROWS <= n) {
      const double* base = X + row0 * ldx;   // wave-uniform
      const uint64_t b = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((uint64_t)base)) |
                         ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((uint64_t)base >> 32)) << 32);
#pragma unroll
      for (int i = 0; i < PIECES; ++i) glds16_off<NT>((const void*)b, off[i], dst + (unsigned)i * 1024u);
    } else {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {   // ragged last slot: rows beyond n read row n - 1, results discarded
        const int G = i * 64 + lane;
        const int row = G / H, g = G % H;
        int64_t gr = row0 + row;
        if (gr >= n) gr = n - 1;
        glds16<NT>(X + gr * ldx + 2 * (g ^ (row & 15)), dst + (unsigned)i * 1024u);
      }
    }
  };
#pragma unroll
  for (int k = 0; k < D - 1; ++k) {
    const int64_t tk = t + k * stride;
    if (tk < nslots) issue(tk, k);
  }
  int cur = 0;
  for (; t < nslots; t += stride) {
    const int64_t tn = t + (int64_t)(D - 1) * stride;
    int nxt = cur + D - 1;
    if (nxt >= D) nxt -= D;
    if (tn < nslots) {
      issue(tn, nxt);
      wait_vm<PIECES * (D - 1)>();
    } else {
      wait_vm<0>();
    }
    const unsigned char* slot = mine + (size_t)cur * SLOT;
    double s = 0.0;
    if (ROWS == 64 || lane < ROWS) {
      const unsigned char* row = slot + (size_t)lane * (MM * 8);
#pragma unroll
      for (int c = 0; c < H; ++c) {
        const v2d v = *reinterpret_cast<const v2d*>(row + ((c ^ (lane & 15)) << 4));
        const v2d yv = *reinterpret_cast<const v2d*>(ys + 2 * c);
        const double d0 = v.x - yv.x;
        const double d1 = v.y - yv.y;
        if constexpr (WGT) {
          const v2d wv = *reinterpret_cast<const v2d*>(ws + 2 * c);
          s = s + wv.x * (d0 * d0);
          s = s + wv.y * (d1 * d1);
        } else {
          s = s + d0 * d0;
          s = s + d1 * d1;
        }
      }
      const int64_t r = t * ROWS + lane;
      if (r < n) out[r] = sqrt(s);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are done before it is refilled
    cur = cur + 1 == D ? 0 : cur + 1;
  }
}

template <int MM, bool WGT>
__global__ __launch_bounds__(256) void refm_kernel(const double* X, int64_t n, const double* y, const double* wgt, double* out) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
  for (int j = 0; j < MM; ++j) {
    const double d = X[r * MM + j] - y[j];
    if (WGT) s = s + wgt[j] * (d * d); else s = s + d * d;
  }
  out[r] = sqrt(s);
}

// register-staged pipeline as the library has it today (tile_stream.hpp), fixed to m = 32, for an A/B in one binary
template <int U>
__global__ __launch_bounds__(128) void regpipe_kernel(const double* __restrict__ X, int64_t n,
                                                      const double* __restrict__ y, double* __restrict__ out) {
  constexpr int T = 128, R = 2 * T * U / M, MP = M | 1;
  __shared__ double tile[R * MP + M];
  double* ys = tile + R * MP;
  const int tid = threadIdx.x;
  if (tid < M) ys[tid] = y[tid];
  const int64_t ntiles = (n + R - 1) / R;
  v2d v[U];
  auto fetch = [&](int64_t tt) {
    const double* base = X + tt * R * M;
    const int64_t left = (n - tt * R) * (M / 2);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = u * T + tid;
      v[u] = idx < left ? *reinterpret_cast<const v2d*>(base + 2 * idx) : (v2d){0.0, 0.0};
    }
  };
  int64_t t = blockIdx.x;
  if (t < ntiles) fetch(t);
  for (; t < ntiles; t += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = u * T + tid, q = idx / (M / 2), jj = idx % (M / 2);
      tile[q * MP + 2 * jj] = v[u].x;
      tile[q * MP + 2 * jj + 1] = v[u].y;
    }
    if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
    __syncthreads();
    const int64_t r = t * R + tid;
    if (tid < R && r < n) {
      const double* row = tile + tid * MP;
      double s = 0.0;
#pragma unroll 8
      for (int j = 0; j < M; ++j) {
        const double d = row[j] - ys[j];
        s = s + d * d;
      }
      out[r] = sqrt(s);
    }
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
  const size_t bytes = (size_t)n * M * 8;
  const int NBUF = 3;  // 3 x 256 MB in rotation: beyond the 256 MiB Infinity Cache
  std::vector<double> hX((size_t)n * M), hy(128);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    return (double)(st >> 11) * (1.0 / 9007199254740992.0) * 4.0 - 2.0;
  };
  for (auto& v : hX) v = rnd();
  for (auto& v : hy) v = rnd();
  std::vector<double*> bufs(NBUF);
  for (auto& b : bufs) {
    CK(hipMalloc(&b, bytes));
    CK(hipMemcpy(b, hX.data(), bytes, hipMemcpyHostToDevice));
  }
  double *dy, *dref, *dout, *dw;
  std::vector<double> hw(128);
  for (auto& v : hw) v = 0.5 + 0.25 * (rnd() + 2.0);
  CK(hipMalloc(&dw, 128 * 8));
  CK(hipMemcpy(dw, hw.data(), 128 * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dy, 128 * 8));
  CK(hipMemcpy(dy, hy.data(), 128 * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dref, n * 8));
  CK(hipMalloc(&dout, n * 8));
  hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, bufs[0], n, dy, dref);
  CK(hipDeviceSynchronize());
  std::vector<double> href(n), hout(n);
  CK(hipMemcpy(href.data(), dref, n * 8, hipMemcpyDeviceToHost));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int64_t cur_n = n;
  int cur_m = M;
  auto run = [&](const char* name, auto launch) {
    const int64_t n = cur_n;
    CK(hipMemset(dout, 0xff, n * 8));
    launch(bufs[0]);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hout.data(), dout, n * 8, hipMemcpyDeviceToHost));
    const bool same = memcmp(hout.data(), href.data(), n * 8) == 0;
    size_t bad = 0;
    if (!same) for (int64_t i = 0; i < n; ++i) bad += hout[i] != href[i];
    for (int w = 0; w < 3; ++w) launch(bufs[w % NBUF]);
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    const int reps = 30, rounds = 3;
    for (int k = 0; k < rounds; ++k) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) launch(bufs[r % NBUF]);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      best = ms < best ? ms : best;
      sum += ms;
    }
    const double alg = (double)n * (8.0 * cur_m + 8.0);
    printf("%-46s best %6.1f us mean %6.1f us  %5.0f GB/s  frac8TB %.3f  %s", name, best * 1e3, sum / rounds * 1e3,
           alg / (best * 1e-3) / 1e9, alg / (best * 1e-3) / 8e12, same ? "bit-exact" : "MISMATCH");
    if (!same) printf(" (%zu rows)", bad);
    printf("\n");
    fflush(stdout);
  };
  char name[128];
  for (int per_cu : {8, 16}) {
    snprintf(name, sizeof name, "regpipe U=4 (library today)  %2d wg/CU", per_cu);
    run(name, [&](double* b) { hipLaunchKernelGGL(regpipe_kernel<4>, dim3(256 * per_cu), dim3(128), 0, 0, b, n, dy, dout); });
  }
#define VARIANT(MM, ROWS, D, NT, WGT, WAVES, PERCU)                                                                 \
  do {                                                                                                              \
    const size_t lds = (size_t)(WAVES) * (D) * (ROWS) * (MM) * 8 + 2 * (MM) * 8;                                      \
    if (lds * (PERCU) <= 160 * 1024) {                                                                              \
      const int64_t nn = n * M / (MM);                                                                              \
      hipLaunchKernelGGL((refm_kernel<MM, WGT>), dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, 0, bufs[0], nn, dy, dw, dref); \
      CK(hipDeviceSynchronize());                                                                                   \
      CK(hipMemcpy(href.data(), dref, nn * 8, hipMemcpyDeviceToHost));                                              \
      cur_n = nn;                                                                                                   \
      cur_m = MM;                                                                                                   \
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(glds_kernel<MM, ROWS, D, NT, WGT>),                      \
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                \
      snprintf(name, sizeof name, "glds m=%d rows=%d D=%d nt=%d w=%d waves/wg=%d wg/CU=%d", MM, ROWS, D, (int)NT,    \
               (int)WGT, WAVES, PERCU);                                                                             \
      run(name, [&](double* b) {                                                                                    \
        hipLaunchKernelGGL((glds_kernel<MM, ROWS, D, NT, WGT>), dim3(256 * (PERCU)), dim3(64 * (WAVES)), lds, 0, b, nn, \
                           (int64_t)(MM), dy, dw, dout);                                                            \
      });                                                                                                           \
    }                                                                                                               \
  } while (0)
  VARIANT(32, 64, 4, true, false, 1, 2);
  VARIANT(32, 64, 4, true, true, 1, 2);
  VARIANT(32, 64, 3, true, false, 1, 3);
  VARIANT(32, 64, 3, true, true, 1, 3);
  VARIANT(32, 64, 2, true, false, 1, 4);
  VARIANT(32, 64, 2, true, true, 1, 4);
  VARIANT(32, 64, 2, true, false, 2, 2);
  VARIANT(32, 64, 2, true, false, 4, 1);
  VARIANT(32, 64, 2, true, true, 4, 1);
  VARIANT(32, 32, 4, true, false, 1, 4);
  VARIANT(32, 32, 4, true, true, 1, 4);
  VARIANT(32, 32, 2, true, false, 1, 8);
  VARIANT(32, 32, 2, true, true, 1, 8);
  VARIANT(32, 32, 3, true, false, 1, 6);
  VARIANT(32, 32, 4, true, false, 4, 1);
  VARIANT(32, 32, 4, true, true, 4, 1);
  VARIANT(32, 16, 4, true, false, 1, 8);
  VARIANT(32, 16, 4, true, true, 1, 8);
  VARIANT(32, 16, 8, true, false, 1, 4);
  // 64 summaries (configs[3]'s width)
  VARIANT(64, 32, 4, true, false, 1, 2);
  VARIANT(64, 32, 4, true, true, 1, 2);
  VARIANT(64, 32, 2, true, false, 1, 4);
  VARIANT(64, 32, 2, true, true, 1, 4);
  VARIANT(64, 16, 4, true, false, 1, 4);
  VARIANT(64, 16, 4, true, true, 1, 4);
  VARIANT(64, 16, 2, true, false, 1, 8);
  VARIANT(64, 16, 2, true, true, 1, 8);
  VARIANT(64, 16, 3, true, true, 1, 6);
  VARIANT(64, 8, 4, true, false, 1, 8);
  VARIANT(64, 8, 4, true, true, 1, 8);
  return 0;
}
