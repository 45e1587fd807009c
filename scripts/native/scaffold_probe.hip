// Developer probe for the resident-sweep scaffold: ticket loop + dependency spins + barrier mix,
// with trivial task bodies.   hipcc --offload-arch=gfx950 -O3 scaffold_probe.hip -o scaffold_probe; ./scaffold_probe MODE
//   mode 1  ticket loop, no dependencies          mode 2  task i waits for task i-1 (chain across workgroups)
//   mode 3  mode 2 + a role-split body with LDS-only barriers (waves 0-2 / 3-15, as the diagonal-block body)
//   mode 4  mode 3 with the body reached through a __noinline__ call
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool wait_ge(const int* p, int want, int* err) {
  unsigned int spins = 0;
  while (ld(p) < want) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0 && (ld(err) != 0 || spins > (1u << 18))) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return true;
}

template <bool INL>
__device__ void role_body_impl(double* sm, double* out, int task) {
  const int tid = threadIdx.x, w = tid >> 6;
  if (w >= 3) {
    sm[tid] = task + tid;
    lds_barrier();
    for (int p = 0; p < 8; ++p) {
      sm[tid] += 1.0;
      lds_barrier();
      sm[tid] += sm[(tid + 64) & 1023];
      lds_barrier();
    }
    out[tid] = sm[tid];
  } else {
    lds_barrier();
    for (int p = 0; p < 8; ++p) {
      sm[tid] = p;
      lds_barrier();
      lds_barrier();
    }
  }
}
__device__ __noinline__ void role_body_call(double* sm, double* out, int task) { role_body_impl<false>(sm, out, task); }

__global__ __launch_bounds__(1024) void scaffold(int mode, int ntasks, int* sync, double* out) {
  extern __shared__ __align__(16) double sm[];
  __shared__ int sh_task, sh_ok;
  const int tid = threadIdx.x;
  int* ticket = sync;
  int* err = sync + 1;
  int* done = sync + 2;
  // ONE `if (tid == 0)` region per iteration, between two barriers: publishing the finished task and drawing the next
  // one in separate regions (tail and head of the loop) lets the compiler thread lane 0 from one into the other across
  // the back edge, and the waves then disagree on the number of barriers they execute (the kernel never returns).
  bool have_prev = false;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      if (have_prev) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int t = atomicAdd(ticket, 1);
      sh_task = t;
      sh_ok = (t >= ntasks || mode < 2 || wait_ge(done, t, err)) ? 1 : 0;
    }
    __syncthreads();
    const int ti = sh_task;
    if (ti >= ntasks || !sh_ok) return;
    __threadfence();
    if (mode == 3)
      role_body_impl<true>(sm, out + (size_t)ti * 1024, ti);
    else if (mode == 4)
      role_body_call(sm, out + (size_t)ti * 1024, ti);
    else
      out[(size_t)ti * 1024 + tid] = ti;
    __threadfence();
    have_prev = true;
  }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 1, ntasks = argc > 2 ? atoi(argv[2]) : 64, grid = argc > 3 ? atoi(argv[3]) : 8;
  int* sync;
  double* out;
  hipMalloc(&sync, 64);
  hipMalloc(&out, (size_t)ntasks * 1024 * 8);
  hipMemset(sync, 0, 64);
  hipLaunchKernelGGL(scaffold, dim3(grid), dim3(1024), 57 * 1024, 0, mode, ntasks, sync, out);
  hipError_t e = hipDeviceSynchronize();
  int h[3];
  hipMemcpy(h, sync, sizeof h, hipMemcpyDeviceToHost);
  printf("mode %d tasks %d grid %d: %s  ticket %d err %d done %d\n", mode, ntasks, grid, hipGetErrorString(e), h[0], h[1], h[2]);
  return 0;
}
