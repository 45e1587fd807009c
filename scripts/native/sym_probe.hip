// Developer probe: the ONE-TRIANGLE symmetric product  u = S b  (S = K^-1, n x n symmetric, 16 right-hand sides) that rounds 5
// and 6 left as analysis -- every 256 x 32 tile of the LOWER triangle applied twice while it is in registers: as S[k][i] for
// the rows i of its row block (the product tri_apply_kernel<3> makes from the FULL matrix today) and, re-dealt through LDS, as
// S[k][i] = S[i][k] for the rows k of its chunk.  Stand-alone (no library code), for ONE question: what does the product
// cost when the matrix is read once instead of twice?
//   sh scripts/native/build_sym_probe.sh && scripts/native/sym_probe [n]
// Shape of the kernel (see DESIGN.md section 7.2 for why single-tile workgroups cannot do this: 32 KB of transposed partials
// per 64 KiB tile):
//   * a workgroup (256 threads) owns a SEGMENT: one 256-row chunk of k, G consecutive 32-row blocks of i left of (or on) the
//     diagonal.  The transposed sums u[k] += S[k][i] b[i] of its G tiles stay in registers (a wave: 64 k x 16 columns) and
//     leave once per segment; the direct sums u[i] += S[k][i] b[k] leave once per tile, as today.
//   * tiles that cross the diagonal take the direct product for k >= i0 only and the transposed one for k >= i0 + 32 only
//     (right-hand sides / operand rows zeroed), so that every entry of the lower triangle below the 32 x 32 diagonal blocks
//     counts twice and the diagonal blocks once, read from the full storage.
//   * a second launch sums the partials (the fused form would give that to the last workgroup to arrive, as tri_apply does).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int KC = 256, RB = 32, PC = 16;
constexpr int TP = 34;   // pitch (doubles) of the tile's copy in LDS

struct Seg {
  int kc, rb0, nrb;   // chunk, first row block, row blocks (<= G)
};

__device__ __forceinline__ void store_wt(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// part1[kc][i][s]: direct sums of chunk kc for row i; part2[seg][k local][s]: transposed sums of a segment
template <int G>
__global__ __launch_bounds__(256) void sym_apply_kernel(const double* __restrict__ S, int64_t lda, int64_t n,
                                                        const double* __restrict__ b, const Seg* __restrict__ segs,
                                                        double* __restrict__ part1, double* __restrict__ part2) {
  extern __shared__ __align__(16) double lds[];
  double* Bk = lds;                    // [256][16] right-hand sides of the chunk's rows k
  double* Bi = Bk + KC * PC;           // [32][16] right-hand sides of the tile's rows i
  double* Tt = Bi + RB * PC;           // [256][TP] the tile, [k][i]; afterwards the four waves' direct partials
  const Seg sg = segs[blockIdx.x];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int64_t k0 = (int64_t)sg.kc * KC;
  for (int e = t; e < KC * PC; e += 256) Bk[e] = b[k0 * PC + e];
  v4d acc2[4];   // transposed sums: rows k0 + 64 w + 16 g + (l >> 4) + 4 r, column l & 15
#pragma unroll
  for (int g = 0; g < 4; ++g) acc2[g] = (v4d){0, 0, 0, 0};
  const unsigned lane_off = (unsigned)(4 * w + (l >> 4)) * (unsigned)lda + 2u * (l & 15);
  v2d wr[16], wn[16];
  auto load_tile = [&](int rb, v2d* dst) {
    const double* wbase = S + k0 * lda + (int64_t)rb * RB;
#pragma unroll
    for (int q = 0; q < 16; ++q) dst[q] = *reinterpret_cast<const v2d*>(wbase + (int64_t)16 * q * lda + lane_off);
  };
  load_tile(sg.rb0, wr);
  for (int j = 0; j < sg.nrb; ++j) {
    const int rb = sg.rb0 + j;
    const int64_t i0 = (int64_t)rb * RB;
    if (j + 1 < sg.nrb) load_tile(rb + 1, wn);   // in flight during this tile's products
    __syncthreads();   // the previous tile's LDS traffic is over (first trip: Bk is complete)
    for (int e = t; e < RB * PC; e += 256) Bi[e] = b[i0 * PC + e];
    // ---- the direct product: u[i] += sum_k S[k][i] b[k], rows k >= i0 only
    v4d acc0 = (v4d){0, 0, 0, 0}, acc1 = (v4d){0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int kl = 16 * q + 4 * w + (l >> 4);
      const double bb = k0 + kl >= i0 ? Bk[kl * PC + (l & 15)] : 0.0;
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[q].x, bb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[q].y, bb, acc1, 0, 0, 0);
    }
    // ---- the tile into LDS as [k][i] for the transposed product
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int kl = 16 * q + 4 * w + (l >> 4);
      *reinterpret_cast<v2d*>(Tt + kl * TP + 2 * (l & 15)) = wr[q];
    }
    __syncthreads();
    // ---- the transposed product: u[k] += sum_i S[k][i] b[i], rows k >= i0 + 32 only
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int kl = 64 * w + 16 * g + (l & 15);
      const bool live = k0 + kl >= i0 + RB;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const double a = live ? Tt[kl * TP + 4 * jj + (l >> 4)] : 0.0;
        const double bb = Bi[(4 * jj + (l >> 4)) * PC + (l & 15)];
        acc2[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc2[g], 0, 0, 0);
      }
    }
    __syncthreads();   // Tt has been read: it carries the waves' direct partials now
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ip = (l >> 4) + 4 * r;
      Tt[(w * RB + 2 * ip) * PC + (l & 15)] = acc0[r];
      Tt[(w * RB + 2 * ip + 1) * PC + (l & 15)] = acc1[r];
    }
    __syncthreads();
    double* out = part1 + ((int64_t)sg.kc * n + i0) * PC;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = t + 256 * h;
      store_wt(out + e, ((Tt[e] + Tt[RB * PC + e]) + Tt[2 * RB * PC + e]) + Tt[3 * RB * PC + e]);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) wr[q] = wn[q];
  }
  double* o2 = part2 + (int64_t)blockIdx.x * KC * PC;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) store_wt(o2 + (64 * w + 16 * g + (l >> 4) + 4 * r) * PC + (l & 15), acc2[g][r]);
}

// u[i][s] = sum over the chunks at or below row i of part1 + sum over the segments of row i's own chunk of part2
__global__ __launch_bounds__(256) void sym_reduce_kernel(const double* __restrict__ part1, const double* __restrict__ part2,
                                                         const int* __restrict__ seg_first, int64_t n, int nkc,
                                                         double* __restrict__ u) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (i, s)
  if (e >= n * PC) return;
  const int64_t i = e / PC;
  const int kc_i = (int)(i / KC);
  double s = 0.0;
  for (int kc = kc_i; kc < nkc; ++kc) s += part1[(int64_t)kc * n * PC + e];
  for (int sg = seg_first[kc_i]; sg < seg_first[kc_i + 1]; ++sg)
    s += part2[((int64_t)sg * KC + (i - (int64_t)kc_i * KC)) * PC + (e % PC)];
  u[e] = s;
}

// the product from the FULL matrix with the same tiles, one tile per workgroup (what tri_apply_kernel<3> does, without its
// epilogue): the baseline this probe is compared with on the same box
__global__ __launch_bounds__(256) void full_apply_kernel(const double* __restrict__ S, int64_t lda, int64_t n,
                                                         const double* __restrict__ b, int nrb, double* __restrict__ part1) {
  __shared__ __align__(16) double Bs[KC * PC];
  const int kc = blockIdx.x / nrb, rb = blockIdx.x % nrb;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int64_t k0 = (int64_t)kc * KC, i0 = (int64_t)rb * RB;
  for (int e = t; e < KC * PC; e += 256) Bs[e] = b[k0 * PC + e];
  const double* wbase = S + k0 * lda + i0;
  const unsigned lane_off = (unsigned)(4 * w + (l >> 4)) * (unsigned)lda + 2u * (l & 15);
  v2d wr[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) wr[q] = *reinterpret_cast<const v2d*>(wbase + (int64_t)16 * q * lda + lane_off);
  __syncthreads();
  v4d acc0 = (v4d){0, 0, 0, 0}, acc1 = (v4d){0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const double bb = Bs[(16 * q + 4 * w + (l >> 4)) * PC + (l & 15)];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[q].x, bb, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[q].y, bb, acc1, 0, 0, 0);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ip = (l >> 4) + 4 * r;
    Bs[(w * RB + 2 * ip) * PC + (l & 15)] = acc0[r];
    Bs[(w * RB + 2 * ip + 1) * PC + (l & 15)] = acc1[r];
  }
  __syncthreads();
  double* out = part1 + ((int64_t)kc * n + i0) * PC;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int e = t + 256 * h;
    store_wt(out + e, ((Bs[e] + Bs[RB * PC + e]) + Bs[2 * RB * PC + e]) + Bs[3 * RB * PC + e]);
  }
}

__global__ __launch_bounds__(256) void full_reduce_kernel(const double* __restrict__ part1, int64_t n, int nkc,
                                                          double* __restrict__ u) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * PC) return;
  double s = 0.0;
  for (int kc = 0; kc < nkc; ++kc) s += part1[(int64_t)kc * n * PC + e];
  u[e] = s;
}

template <int G>
static int run(int64_t n) {
  const int nkc = (int)(n / KC), nrb = (int)(n / RB);
  std::mt19937_64 g(3);
  std::normal_distribution<double> N(0.0, 1.0);
  std::vector<double> S((size_t)n * n), b((size_t)n * PC);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j <= i; ++j) S[i * n + j] = S[j * n + i] = N(g) / std::sqrt((double)n);
  for (auto& x : b) x = N(g);
  // segments: chunk kc, row blocks 0 .. 8 kc + 7 in groups of G (the long chunks first: they finish last otherwise)
  std::vector<Seg> segs;
  std::vector<int> first(nkc + 1, 0);
  for (int kc = 0; kc < nkc; ++kc) {
    first[kc] = (int)segs.size();
    const int last_rb = std::min(nrb, 8 * kc + 8);
    for (int rb0 = 0; rb0 < last_rb; rb0 += G) segs.push_back(Seg{kc, rb0, std::min(G, last_rb - rb0)});
  }
  first[nkc] = (int)segs.size();
  double *dS, *db, *dp1, *dp2, *du, *du0;
  Seg* dsg;
  int* dfirst;
  CK(hipMalloc(&dS, S.size() * 8)); CK(hipMalloc(&db, b.size() * 8)); CK(hipMalloc(&dp1, (size_t)nkc * n * PC * 8));
  CK(hipMalloc(&dp2, segs.size() * KC * PC * 8)); CK(hipMalloc(&du, n * PC * 8)); CK(hipMalloc(&du0, n * PC * 8));
  CK(hipMalloc(&dsg, segs.size() * sizeof(Seg))); CK(hipMalloc(&dfirst, first.size() * 4));
  CK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsg, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
  CK(hipMemcpy(dfirst, first.data(), first.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dp1, 0, (size_t)nkc * n * PC * 8));
  const size_t lds = (size_t)(KC * PC + RB * PC + KC * TP) * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(sym_apply_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float best_a = 1e9f, best_r = 1e9f, best_fa = 1e9f, best_fr = 1e9f;
  const unsigned rgrid = (unsigned)((n * PC + 255) / 256);
  for (int rep = 0; rep < 30; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(sym_apply_kernel<G>, dim3((unsigned)segs.size()), dim3(256), lds, 0, dS, n, n, db, dsg, dp1, dp2);
    CK(hipEventRecord(e1));
    hipLaunchKernelGGL(sym_reduce_kernel, dim3(rgrid), dim3(256), 0, 0, dp1, dp2, dfirst, n, nkc, du);
    CK(hipEventRecord(e2));
    CK(hipDeviceSynchronize());
    float a, r;
    CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&r, e1, e2));
    best_a = std::min(best_a, a); best_r = std::min(best_r, r);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(full_apply_kernel, dim3((unsigned)(nkc * nrb)), dim3(256), 0, 0, dS, n, n, db, nrb, dp1);
    CK(hipEventRecord(e1));
    hipLaunchKernelGGL(full_reduce_kernel, dim3(rgrid), dim3(256), 0, 0, dp1, n, nkc, du0);
    CK(hipEventRecord(e2));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&r, e1, e2));
    best_fa = std::min(best_fa, a); best_fr = std::min(best_fr, r);
    if (rep == 0) CK(hipMemset(dp1, 0, (size_t)nkc * n * PC * 8));   // (the full product filled the upper chunks too)
  }
  // one more one-triangle run on clean partials for the check
  CK(hipMemset(dp1, 0, (size_t)nkc * n * PC * 8));
  hipLaunchKernelGGL(sym_apply_kernel<G>, dim3((unsigned)segs.size()), dim3(256), lds, 0, dS, n, n, db, dsg, dp1, dp2);
  hipLaunchKernelGGL(sym_reduce_kernel, dim3(rgrid), dim3(256), 0, 0, dp1, dp2, dfirst, n, nkc, du);
  std::vector<double> u(n * PC), u0(n * PC);
  CK(hipMemcpy(u.data(), du, u.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(u0.data(), du0, u0.size() * 8, hipMemcpyDeviceToHost));
  double worst = 0.0, worst0 = 0.0, big = 0.0;
  for (int64_t i = 0; i < n; i += 37)
    for (int s = 0; s < PC; ++s) {
      long double ref = 0.0L;
      for (int64_t k = 0; k < n; ++k) ref += (long double)S[k * n + i] * b[k * PC + s];
      worst = std::max(worst, std::fabs(u[i * PC + s] - (double)ref));
      worst0 = std::max(worst0, std::fabs(u0[i * PC + s] - (double)ref));
      big = std::max(big, std::fabs((double)ref));
    }
  printf("n = %lld, G = %d: %zu segments.  one triangle: product %.1f us + sums %.1f us = %.1f us;  full matrix: %.1f + %.1f = %.1f us;"
         "  max |u - ref| %.2g (one triangle) %.2g (full), max |ref| %.2g\n",
         (long long)n, G, segs.size(), best_a * 1e3, best_r * 1e3, (best_a + best_r) * 1e3, best_fa * 1e3, best_fr * 1e3,
         (best_fa + best_fr) * 1e3, worst, worst0, big);
  hipFree(dS); hipFree(db); hipFree(dp1); hipFree(dp2); hipFree(du); hipFree(du0); hipFree(dsg); hipFree(dfirst);
  return 0;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
  if (run<2>(n)) return 1;
  if (run<4>(n)) return 1;
  if (run<8>(n)) return 1;
  return 0;
}
