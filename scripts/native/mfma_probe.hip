// Developer probe: what the f64 matrix pipe delivers on this chip under the access patterns of the step kernel.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_probe mfma_probe.hip && ./mfma_probe
// V0 pure MFMA, 16 waves / CU; V1 pure MFMA, 4 waves / CU; V2 + operands from LDS (the step kernel's 2 x 2 wave tile);
// V3 + one barrier per 32 MFMAs; V4 + LDS stores of a k-tile and two barriers per 32 MFMAs (single-stage staging);
// V5 = V2 with 8 waves per CU holding 4 x 2 tiles (more MFMAs per LDS read, two waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int GLP2 = 36;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int V>
__global__ __launch_bounds__(1024) void probe(double* out, int iters) {
  extern __shared__ __align__(16) double sm[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  for (int i = t; i < 2 * 256 * GLP2; i += blockDim.x) sm[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  v4d acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  const int wr = (w >> 2) & 3, wc = w & 3;
  const double* fa = sm + (wr * 32 + (l & 15)) * GLP2 + 2 * (l >> 4);
  const double* fb = sm + (128 + wc * 32 + (l & 15)) * GLP2 + 2 * (l >> 4);
  double* da = sm + 256 * GLP2 + (t >> 4) * GLP2 + 2 * (t & 15);
  double2 r0 = {1.0, 2.0}, r1 = {3.0, 4.0};
  for (int it = 0; it < iters; ++it) {
    if (V == 4) {
      __syncthreads();
      *reinterpret_cast<double2*>(da) = r0;
      *reinterpret_cast<double2*>(da + 64 * GLP2) = r1;
      *reinterpret_cast<double2*>(da + 128 * GLP2) = r0;
      *reinterpret_cast<double2*>(da + 192 * GLP2) = r1;
      __syncthreads();
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      double2 a[2], b[2];
      if (V >= 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const double2*>(fa + i * 16 * GLP2 + 8 * h);
          b[i] = *reinterpret_cast<const double2*>(fb + i * 16 * GLP2 + 8 * h);
        }
      } else {
        a[0] = r0; a[1] = r1; b[0] = r1; b[1] = r0;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
        }
    }
    if (V == 3) __syncthreads();
    if (V < 2) { r0.x += 1e-12; }
  }
  double s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * blockDim.x + t] = s;
}

// 8 waves, each 64 x 32 (4 x 2 MFMA tiles)
__global__ __launch_bounds__(512) void probe8(double* out, int iters) {
  extern __shared__ __align__(16) double sm[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  for (int i = t; i < 2 * 256 * GLP2; i += blockDim.x) sm[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  v4d acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  const int wr = w >> 2, wc = w & 3;
  const double* fa = sm + (wr * 64 + (l & 15)) * GLP2 + 2 * (l >> 4);
  const double* fb = sm + (128 + wc * 32 + (l & 15)) * GLP2 + 2 * (l >> 4);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      double2 a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const double2*>(fa + i * 16 * GLP2 + 8 * h);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const double2*>(fb + j * 16 * GLP2 + 8 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
        }
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * blockDim.x + t] = s;
}

template <class K>
static int run(const char* name, K kern, int threads, int blocks, int mfma_per_iter_per_wave, double* out) {
  const size_t lds = 2 * 256 * GLP2 * sizeof(double);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 4000;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, 200);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * (threads / 64) * iters * mfma_per_iter_per_wave * 2048.0;
  printf("%-58s %8.3f ms  %6.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double* out;
  CK(hipMalloc(&out, (size_t)cus * 4 * 1024 * sizeof(double)));
  run("V0 pure MFMA, 16 waves/CU", probe<0>, 1024, cus, 32, out);
  run("V1 pure MFMA, 4 waves/CU", probe<1>, 256, cus, 32, out);
  run("V1b pure MFMA, 8 waves/CU", probe<1>, 512, cus, 32, out);
  run("V2 LDS operands 2x2 tile, 16 waves/CU", probe<2>, 1024, cus, 32, out);
  run("V3 V2 + one barrier per 32 MFMAs", probe<3>, 1024, cus, 32, out);
  run("V4 V2 + LDS stores + two barriers per 32 MFMAs", probe<4>, 1024, cus, 32, out);
  run("V5 LDS operands 4x2 tile, 8 waves/CU", probe8, 512, cus, 64, out);
  run("V0 on 248 CUs", probe<0>, 1024, 248, 32, out);
  return 0;
}
