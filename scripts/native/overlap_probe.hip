// Developer probe (round 5): can the sweep's chain (panel solve -> diagonal tile, two small launches per block column on
// one stream) run BESIDE a persistent update kernel on a second stream, with flags in device memory instead of kernel
// boundaries or stream events between the two?  Models one rebuild at n = 4096:
//   stream U: ONE persistent launch, 256 workgroups x 1024 threads, 104 KiB LDS (the step kernel's shape).  Workgroup 0
//             plays the diagonal block: wait tile_done[k-1], spend POTF2 us, publish potf2_done[k].  Workgroups 1..255
//             play the update: wait panel_solved[k], spend UPD us (MFMAs), drain, arrive at upd_done[k].
//   stream H: per step two ordinary launches, enqueued ahead of time: solve<<<248 x 256>>> (one lane per workgroup polls
//             potf2_done[k] and upd_done[k-1], then loads, a few MFMAs, write-through stores, arrival at panel_solved[k])
//             and tile<<<16 x 256>>> (same, arrival at tile_done[k]).
// Prints the period of the chain per step (workgroup 0's wall-clock stamps) for a few update lengths.
//   hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                \
  do {                                                                       \
    hipError_t e_ = (x);                                                     \
    if (e_ != hipSuccess) {                                                  \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));         \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int NS = 31;          // steps
constexpr int SOLVE_WGS = 248;
constexpr int TILE_WGS = 16;
constexpr int SPIN_LIMIT = 1 << 20;

struct Flags {
  unsigned potf2_done[64];
  unsigned panel_solved[64];
  unsigned tile_done[64];
  unsigned upd_done[64];
  int timeout;
  long long stamp[64];     // wall clock (100 MHz) at the start of every diagonal block
  long long stamp_end[64];
  long long cs[64][2][8];
  unsigned long long seg_max[64][2][4];  // per launch: max over workgroups of (acquire -> issued), (issued -> arrived), ticks  // chain launch (step, solve/tile): workgroup 0 entered, flags seen, arrived
};

__device__ int g_long_sleep;
__device__ __forceinline__ long long now() { return (long long)wall_clock64(); }

__device__ __forceinline__ bool wait_ge(const unsigned* c, unsigned target, int* timeout) {
  int spins = 0;
  while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    if (++spins >= SPIN_LIMIT || ((spins & 63) == 0 && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      __hip_atomic_store(timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    if (g_long_sleep) __builtin_amdgcn_s_sleep(64); else __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

__device__ __forceinline__ void store_wt(double* p, double v) {
  __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double burn_until(long long t_end, double seed) {
  // keep the matrix pipe busy until the wall clock says stop
  v4d acc = (v4d){seed, 0, 0, 0};
  while (now() < t_end) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(seed, 1e-9, acc, 0, 0, 0);
  }
  return acc[0] + acc[1];
}

__global__ __launch_bounds__(1024) void persistent_kernel(Flags* F, double* scratch, int potf2_ticks, int upd_ticks, int busy0, int one_fence) {
  extern __shared__ double sm[];
  __shared__ int ok;
  if (blockIdx.x == 0) {
    for (int k = 1; k <= NS; ++k) {
      if (threadIdx.x == 0) ok = wait_ge(&F->tile_done[k - 1], TILE_WGS, &F->timeout);
      if (one_fence ? threadIdx.x < 64 : true) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __syncthreads();
      if (!ok) return;
      const long long t0 = now();
      if (threadIdx.x == 0) F->stamp[k] = t0;
      double v;
      if (busy0)
        v = burn_until(t0 + potf2_ticks, 1.0 + threadIdx.x);
      else {
        while (now() < t0 + potf2_ticks) __builtin_amdgcn_s_sleep(4);
        v = 1.0;
      }
      store_wt(scratch + threadIdx.x, v);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        F->stamp_end[k] = now();
        __hip_atomic_store(&F->potf2_done[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  for (int k = 0; k < NS; ++k) {
    if (threadIdx.x == 0) ok = wait_ge(&F->panel_solved[k], SOLVE_WGS, &F->timeout);
    if (one_fence ? threadIdx.x < 64 : true) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    if (!ok) return;
    const double v = burn_until(now() + upd_ticks, 1.0 + threadIdx.x);
    store_wt(scratch + (size_t)blockIdx.x * 1024 + threadIdx.x, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&F->upd_done[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// a chain launch: poll, one memory round trip, 36 dependent MFMAs, write-through stores, arrival
#define CS(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) st[i] = now(); } while (0)
__global__ __launch_bounds__(256) void chain_kernel(Flags* F, const double* src, double* dst, int k, int is_tile, int nupd, int prio, int one_fence) {
  __shared__ int ok;
  long long st[8];
  long long w3 = 0, w4 = 0;
  if (prio) __builtin_amdgcn_s_setprio(3);
  CS(0);
  if (threadIdx.x == 0) {
    bool o = true;
    if (!is_tile) {
      o = wait_ge(&F->potf2_done[k], 1u, &F->timeout);
      if (o && k > 0) o = wait_ge(&F->upd_done[k - 1], (unsigned)nupd, &F->timeout);
    }
    ok = o;
  }
  CS(1);
  if (one_fence ? threadIdx.x < 64 : true) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  CS(2);
  __syncthreads();
  if (!ok) return;
  CS(3);
  if (threadIdx.x == 0) w3 = now();
  const size_t off = ((size_t)(k * 512 + blockIdx.x) * 256 + threadIdx.x) * 2;
  const double a = src[off], b = src[off + 1];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long wl = 0, wm = 0;
  {
    const int dep = __builtin_amdgcn_readfirstlane(__double2hiint(a + b));
    if (threadIdx.x == 0) wl = now() + (dep & 0);
  }
  v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 36; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  {
    const int dep = __builtin_amdgcn_readfirstlane(__double2hiint(acc[0]));
    if (threadIdx.x == 0) wm = now() + (dep & 0);
  }
  store_wt(dst + off, acc[0]);
  store_wt(dst + off + 1, acc[1]);
  CS(4);
  if (threadIdx.x == 0) w4 = now();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CS(5);
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_fetch_add(is_tile ? &F->tile_done[k] : &F->panel_solved[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  CS(6);
  if (threadIdx.x == 0) {
    atomicMax(&F->seg_max[k][is_tile][0], (unsigned long long)(w4 - w3));
    atomicMax(&F->seg_max[k][is_tile][1], (unsigned long long)(now() - w4));
    atomicMax(&F->seg_max[k][is_tile][2], (unsigned long long)(wl - w3));
    atomicMax(&F->seg_max[k][is_tile][3], (unsigned long long)(wm - wl));
  }
  if (threadIdx.x == 0 && blockIdx.x == 0)
    for (int i = 0; i < 7; ++i) F->cs[k][is_tile][i] = st[i];
}

int main(int argc, char** argv) {
  const double potf2_us = argc > 1 ? atof(argv[1]) : 27.0;
  const int prio = argc > 2 ? atoi(argv[2]) : 0;
  const int long_sleep = argc > 3 ? atoi(argv[3]) : 0;
  const int one_fence = argc > 4 ? atoi(argv[4]) : 1;
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_long_sleep), &long_sleep, sizeof(int)));
  Flags* F;
  CK(hipMalloc(&F, sizeof(Flags)));
  double *src, *dst, *scratch;
  const size_t words = (size_t)64 * 512 * 256 * 2;
  CK(hipMalloc(&src, words * 8));
  CK(hipMalloc(&dst, words * 8));
  CK(hipMalloc(&scratch, (size_t)256 * 1024 * 8));
  CK(hipMemset(src, 0, words * 8));
  hipStream_t U, H;
  CK(hipStreamCreateWithFlags(&U, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&H, hipStreamNonBlocking));
  const size_t lds = 104 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int nupd = 255;
  for (int busy0 = 0; busy0 < 2; ++busy0)
    for (double upd_us : {0.0, 20.0, 30.0, 36.0}) {
      double best_ms = 1e9, best_period = 0.0, best_gap = 0.0;
      int timeouts = 0;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(F, 0, sizeof(Flags), H));
        CK(hipStreamSynchronize(H));
        CK(hipEventRecord(e0, U));
        hipLaunchKernelGGL(persistent_kernel, dim3(256), dim3(1024), lds, U, F, scratch, (int)(potf2_us * 100), (int)(upd_us * 100), busy0, one_fence);
        for (int k = 0; k < NS; ++k) {
          // step 0: the first diagonal block is "already factored"
          if (k == 0) {
            unsigned one = 1;
            CK(hipMemcpyAsync(&F->potf2_done[0], &one, 4, hipMemcpyHostToDevice, H));
          }
          hipLaunchKernelGGL(chain_kernel, dim3(SOLVE_WGS), dim3(256), 0, H, F, src, dst, k, 0, nupd, prio, one_fence);
          hipLaunchKernelGGL(chain_kernel, dim3(TILE_WGS), dim3(256), 0, H, F, src, dst, k, 1, nupd, prio, one_fence);
        }
        CK(hipEventRecord(e1, U));
        CK(hipStreamSynchronize(H));
        CK(hipStreamSynchronize(U));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        Flags h;
        CK(hipMemcpy(&h, F, sizeof(Flags), hipMemcpyDeviceToHost));
        timeouts += h.timeout;
        if (rep > 0 && ms < best_ms && !h.timeout) {
          best_ms = ms;
          best_period = (h.stamp[NS] - h.stamp[5]) / 100.0 / (NS - 5);
          double g = 0;
          for (int k = 6; k <= NS; ++k) g += (h.stamp[k] - h.stamp_end[k - 1]) / 100.0;
          best_gap = g / (NS - 5);
        }
        if ((upd_us == 30.0 || upd_us == 0.0) && rep == 3 && busy0 == 1) {
            for (int k = 10; k < 12; ++k) {
              printf("   step %d (us after block %d started): block end %+.2f, next block starts %+.2f\n", k, k,
                     (h.stamp_end[k] - h.stamp[k]) / 100.0, (h.stamp[k + 1] - h.stamp[k]) / 100.0);
              for (int j = 0; j < 2; ++j) {
                printf("      %s: slowest workgroup: acquire -> issued %.2f us (loads %.2f, 36 MFMAs %.2f), issued -> arrived %.2f us\n", j ? "tile " : "solve",
                       h.seg_max[k][j][0] / 100.0, h.seg_max[k][j][2] / 100.0, h.seg_max[k][j][3] / 100.0, h.seg_max[k][j][1] / 100.0);
                printf("      %s: entered %+.2f, waited %+.2f, barrier %+.2f, acquire %+.2f, loads+MFMAs+stores issued %+.2f, drained %+.2f, arrived %+.2f\n",
                       j ? "tile " : "solve", (h.cs[k][j][0] - h.stamp[k]) / 100.0, (h.cs[k][j][1] - h.stamp[k]) / 100.0,
                       (h.cs[k][j][2] - h.stamp[k]) / 100.0, (h.cs[k][j][3] - h.stamp[k]) / 100.0, (h.cs[k][j][4] - h.stamp[k]) / 100.0,
                       (h.cs[k][j][5] - h.stamp[k]) / 100.0, (h.cs[k][j][6] - h.stamp[k]) / 100.0);
              }
            }
          }
      }
      printf("diagonal block %.0f us (%s), update %.0f us per step: launch %.3f ms, chain period %.2f us per step, "
             "block end -> next block start %.2f us, timeouts %d\n",
             potf2_us, busy0 ? "MFMA busy" : "sleeping", upd_us, best_ms, best_period, best_gap, timeouts);
      fflush(stdout);
    }
  return 0;
}
