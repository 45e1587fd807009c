// Developer probe: do CU-masked streams keep two kernels on disjoint CUs on this stack, and which
// (XCC, SE, CU) does a mask bit select?   hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                  \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__global__ void where_kernel(unsigned* out, long long spin_cycles, int lds_bytes_touch) {
  extern __shared__ char lds[];
  if (lds_bytes_touch > 0 && threadIdx.x == 0) lds[lds_bytes_touch - 1] = 1;
  const long long t0 = clock64();
  while (clock64() - t0 < spin_cycles) {
  }
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
}

static void report(const char* name, const std::vector<unsigned>& h, int nwg) {
  std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int i = 0; i < nwg; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  printf("%s: %zu XCCs used:", name, per_xcc.size());
  size_t total = 0;
  for (auto& kv : per_xcc) {
    printf(" xcc%u:%zu", kv.first, kv.second.size());
    total += kv.second.size();
  }
  printf("  (distinct CUs %zu)\n", total);
}

int main(int argc, char** argv) {
  const int reserved = argc > 1 ? atoi(argv[1]) : 2;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int words = (cus + 31) / 32;
  std::vector<uint32_t> crit(words, 0u), bulk(words, 0u);
  for (int i = 0; i < cus; ++i) ((i % 8) >= 8 - reserved ? crit : bulk)[i / 32] |= 1u << (i % 32);
  hipStream_t s_hi, s_bulk, s_plain;
  CK(hipExtStreamCreateWithCUMask(&s_hi, words, crit.data()));
  CK(hipExtStreamCreateWithCUMask(&s_bulk, words, bulk.data()));
  CK(hipStreamCreateWithFlags(&s_plain, hipStreamNonBlocking));
  const int nwg = 4096;
  unsigned* d;
  CK(hipMalloc(&d, 2 * nwg * sizeof(unsigned)));
  std::vector<unsigned> h(2 * nwg);
  struct {
    const char* name;
    hipStream_t s;
  } runs[] = {{"plain stream", s_plain}, {"critical mask", s_hi}, {"bulk mask", s_bulk}};
  for (auto& r : runs) {
    hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(256), 1024, r.s, d, 20000LL, 0);
    CK(hipStreamSynchronize(r.s));
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    report(r.name, h, nwg);
  }
  // latency of a small kernel on the critical stream while a long kernel fills the bulk stream
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  unsigned* d2;
  CK(hipMalloc(&d2, 2 * nwg * sizeof(unsigned)));
  for (int big_lds = 0; big_lds < 2; ++big_lds) {
    const int lds_small = big_lds ? 150 * 1024 : 1024;
    if (big_lds) CK(hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int busy = 0; busy < 2; ++busy) {
      for (int which = 0; which < 2; ++which) {
        hipStream_t sb = which ? s_bulk : s_plain, sh = which ? s_hi : s_plain;
        if (!which && busy) {
          // plain streams: second plain stream for the small kernel
          static hipStream_t s_plain2 = nullptr;
          if (!s_plain2) CK(hipStreamCreateWithFlags(&s_plain2, hipStreamNonBlocking));
          sh = s_plain2;
        }
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          if (busy) hipLaunchKernelGGL(where_kernel, dim3(8192), dim3(256), 40 * 1024, sb, d, 100000LL, 40 * 1024);  // ~40 us per WG
          CK(hipEventRecord(e0, sh));
          hipLaunchKernelGGL(where_kernel, dim3(big_lds ? 1 : 64), dim3(256), lds_small, sh, d2, 2000LL, lds_small);
          CK(hipEventRecord(e1, sh));
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
          CK(hipDeviceSynchronize());
        }
        printf("small kernel (%s LDS) on %s streams, bulk %s: %.1f us\n", big_lds ? "150 KiB" : "1 KiB",
               which ? "masked" : "plain", busy ? "busy" : "idle", best * 1e3f);
      }
    }
  }
  return 0;
}
