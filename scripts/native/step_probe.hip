// Developer probe: what a unit of the fused sweep's trailing update costs (step_kernel, gp_fit.hip) as a function of its
// depth -- the cost model of sweep_sched.hpp (fixed part + per k-tile).  Every update workgroup gets `per` units of
// `nkt` k-tiles on distinct tiles of a 32 x 32 block matrix; the diagonal block is switched off.
//   cd scripts/native && hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I../../elfi_amd/csrc \
//       -I../../include -c -o /tmp/step_probe.o step_probe.hip && \
//   hipcc --offload-arch=gfx950 -o step_probe /tmp/step_probe.o $(ls ../../elfi_amd/csrc/build/*.o | grep -v gp_fit.o)
#include "../../elfi_amd/csrc/gp_fit.hip"
#include <cstdio>
#include <vector>
using namespace elfihip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int nb = 32, nwg = 248;
  const int64_t lda = nb * NB, rows = (nb + 1) * NB;
  double *A, *WT;
  CK(hipMalloc(&A, rows * lda * 8));
  CK(hipMalloc(&WT, (int64_t)nb * NB * lda * 8));
  CK(hipMemset(A, 0, rows * lda * 8));
  CK(hipMemset(WT, 0, (int64_t)nb * NB * lda * 8));
  CK(hipFuncSetAttribute((const void*)step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)STEP_LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  SweepUnit *dU, *dH; int32_t* dO;
  CK(hipMalloc(&dU, 16 * 8192)); CK(hipMalloc(&dH, 16 * nwg)); CK(hipMalloc(&dO, 4 * (nwg + 1)));
  printf("%5s %4s | %8s  per unit\n", "nkt", "per", "us");
  for (int nkt : {2, 3, 4, 6, 8, 13, 16, 26, 52})
    for (int per : {1, 2, 4}) {
      std::vector<SweepUnit> U, H;
      std::vector<int32_t> O;
      int tile = 0;
      for (int w = 0; w < nwg; ++w) {
        O.push_back((int32_t)U.size());
        for (int i = 0; i < per; ++i, ++tile) {
          // distinct tile halves in the lower triangle of block columns 16..31 (k range [0, nkt) lies left of them)
          const int c = 16 + (tile / 2) % 16, row = c + ((tile / 2) / 16) % (nb - c);
          SweepUnit u;
          u.row = row; u.c = (int16_t)c; u.kt0 = 0; u.nkt = (int16_t)nkt; u.half = (uint8_t)(tile & 1); u.keep = 1; u.pad = 0;
          if (i == 0) { u.pad = per; H.push_back(u); }
          U.push_back(u);
        }
      }
      O.push_back((int32_t)U.size());
      CK(hipMemcpy(dU, U.data(), U.size() * 16, hipMemcpyHostToDevice));
      CK(hipMemcpy(dO, O.data(), O.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dH, H.data(), H.size() * 16, hipMemcpyHostToDevice));
      StepArgs S = {};
      S.P.A = A; S.P.WT = WT; S.P.W11 = nullptr; S.P.lda = lda; S.P.k = 0; S.P.nb = nb; S.P.ku0 = 0; S.P.kun = 1;
      S.W11 = nullptr; S.info = nullptr; S.units = dU; S.wg_off = dO; S.heads = dH;
      float best = 1e9f;
      for (int rep = 0; rep < 12; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(step_kernel, dim3(1 + nwg), dim3(1024), STEP_LDS_BYTES, 0, S);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
      }
      printf("%5d %4d | %8.2f  %6.2f\n", nkt, per, best * 1e3f, best * 1e3f / per);
    }
  return 0;
}
