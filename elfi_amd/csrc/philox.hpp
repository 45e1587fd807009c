// Counter-based normals on the device: Philox4x32-10 (Salmon et al., Random123; known answers in tests/test_gauss_gpu.py)
// + Box-Muller in f64.  Normal number e of stream (seed, stream) is the same wherever it is drawn: by
// elfihip_randn_dev into memory, or inside the fused example kernels (gauss.hip, summaries.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace elfihip {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// standard normals number 2p and 2p+1 of stream (seed, stream)
__device__ __forceinline__ void normal_pair(uint64_t seed, uint64_t stream, uint64_t p, double& z0, double& z1) {
  uint32_t r[4];
  philox4x32_10((uint32_t)p, (uint32_t)(p >> 32), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), r);
  // 53-bit uniforms in (0, 1]: ((hi << 21) | (lo >> 11)) + 1 scaled by 2^-53 -- the logarithm's argument is never zero
  const uint64_t a = ((uint64_t)r[0] << 21) | (r[1] >> 11), b = ((uint64_t)r[2] << 21) | (r[3] >> 11);
  const double u1 = (double)(a + 1) * 0x1.0p-53, u2 = (double)b * 0x1.0p-53;
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincospi(2.0 * u2, &sn, &cs);
  z0 = rad * cs;
  z1 = rad * sn;
}

}  // namespace elfihip
