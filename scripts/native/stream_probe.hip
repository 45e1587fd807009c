// Developer probe: how fast can ONE contiguous f64 array be read on this GPU, by load width / unroll / grid size?
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>

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

typedef double v2d __attribute__((ext_vector_type(2)));

// grid-stride over 16-byte pieces, U pieces per thread in flight; the xor keeps the loads alive
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const v2d* x, size_t npieces, double* sink) {
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t base = (size_t)blockIdx.x * 256 * U; base < npieces; base += stride) {
    v2d v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256 + threadIdx.x;
      v[u] = i < npieces ? x[i] : (v2d){0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 123.456) sink[0] = acc;
}

// contiguous chunk per workgroup (each workgroup owns npieces / grid consecutive pieces)
template <int U>
__global__ __launch_bounds__(256) void read_chunk_kernel(const v2d* x, size_t npieces, double* sink) {
  double acc = 0.0;
  const size_t per = (npieces + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < npieces ? lo + per : npieces;
  for (size_t base = lo; base < hi; base += 256 * U) {
    v2d v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * 256 + threadIdx.x;
      v[u] = i < hi ? x[i] : (v2d){0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  if (acc == 123.456) sink[0] = acc;
}

int main() {
  const size_t bytes = (size_t)256 << 20;  // one 10^6 x 32 f64 batch
  const int NBUF = 3;                       // rotate: beyond the 256 MiB Infinity Cache
  std::vector<v2d*> bufs(NBUF);
  for (auto& b : bufs) {
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(b, 1, bytes));
  }
  double* sink;
  CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t npieces = bytes / 16;
  auto time_it = [&](const char* name, auto launch) {
    for (int w = 0; w < 3; ++w) launch(bufs[w % NBUF]);
    CK(hipDeviceSynchronize());
    const int reps = 30;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch(bufs[r % NBUF]);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %7.1f us  %6.0f GB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
  };
  char name[96];
  for (int per_cu : {2, 4, 8, 16}) {
    const int grid = 256 * per_cu;
    snprintf(name, sizeof name, "grid-stride  U=4  %2d workgroups/CU", per_cu);
    time_it(name, [&](v2d* b) { hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, b, npieces, sink); });
    snprintf(name, sizeof name, "grid-stride  U=8  %2d workgroups/CU", per_cu);
    time_it(name, [&](v2d* b) { hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, b, npieces, sink); });
    snprintf(name, sizeof name, "grid-stride  U=16 %2d workgroups/CU", per_cu);
    time_it(name, [&](v2d* b) { hipLaunchKernelGGL(read_kernel<16>, dim3(grid), dim3(256), 0, 0, b, npieces, sink); });
    snprintf(name, sizeof name, "chunk/wg     U=8  %2d workgroups/CU", per_cu);
    time_it(name, [&](v2d* b) { hipLaunchKernelGGL(read_chunk_kernel<8>, dim3(grid), dim3(256), 0, 0, b, npieces, sink); });
  }
  return 0;
}
