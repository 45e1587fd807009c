// Gradient of the GP log marginal likelihood w.r.t. the four hyper-parameters, on gfx950.
//
// Replaces what GPy computes on every objective evaluation of GPyRegression.optimize()
// (elfi/methods/bo/gpy_regression.py:317-323 -> GPy model.optimize, [GPy-upstream]
// ExactGaussianInference + kern.update_gradients_full):
//     K^-1   = L^-T L^-1                      (GPy: dpotri)
//     dL/dK  = 0.5 (alpha alpha^T - K^-1)
//     d logZ / d s_f = sum(dL/dK * K_rbf) / s_f
//     d logZ / d l   = sum(dL/dK * K_rbf * r^2) / l^3
//     d logZ / d s_b = sum(dL/dK)
//     d logZ / d s_n = trace(dL/dK)
// The factorisation already holds L^-T (upper triangular, gp_fit.hip), so K^-1 is one SYRK on
// the matrix cores: tile (I, J), I >= J, is sum_{k >= I} WT[I][k] WT[J][k]^T.  The four
// contractions are fused into that SYRK's epilogue -- K_rbf and r^2 are recomputed from X for
// the tile (a d-long dot product per entry, noise next to the n/2-long one just finished), so
// K^-1 is never written to memory unless the caller asks for it (GPy's `woodbury_inv`).
// Flops: n^3/3 on v_mfma_f64_16x16x4_f64; HBM: WT read once per tile row pair through L2/MALL.
#include <algorithm>
#include <vector>

#include "gp.hpp"
#include "mfma_f64.hpp"

namespace elfihip {

// One workgroup = one (tile, k chunk).  Tile (I, J) sums k >= I: block row 0 is n deep, the last one 128.  With
// about as many tiles as workgroup slots the kernel would last as long as its longest tile, so tiles are cut
// into chunks of about a third of (total depth / CUs), at least 512; the contractions are linear in the tile, so every chunk runs the
// epilogue on its partial tile (the alpha alpha^T term goes with a tile's first chunk) and the partial sums are
// added in fixed order afterwards.  When K^-1 itself is wanted the tiles are not cut.
struct HyperItem {
  int ti, tj, k0, k1;  // tile, k range [k0, k1); k0 == ti * NB marks the tile's first chunk
};

struct HyperArgs {
  const HyperItem* items;
  const double* WT;
  const double* X;
  const double* x2;
  const double* alpha;
  double* Kinv;   // optional (cap, lda) output, lower tiles
  double* part;   // (items, 4) partial sums of every (tile, k chunk)
  int64_t lda, n, np;
  int dp, x_in_lds;
  double var, neg_half_inv_ls2;
};

__global__ __launch_bounds__(256) void kinv_grad_kernel(HyperArgs H) {
  extern __shared__ __align__(16) double lds[];
  const int64_t b = blockIdx.x;
  const HyperItem it = H.items[b];
  const int64_t ti = it.ti, tj = it.tj;
  const int64_t i0 = ti * NB, j0 = tj * NB;
  const bool first_chunk = it.k0 == (int)i0;
  GemmAcc acc;
  acc.zero();
  const double* Ai = H.WT + i0 * H.lda;
  const double* Bj = H.WT + j0 * H.lda;
  if (ti == tj)
    gemm_tile_nt<true>(acc, Ai, H.lda, Bj, H.lda, it.k0, it.k1, lds);
  else
    gemm_tile_nt<false>(acc, Ai, H.lda, Bj, H.lda, it.k0, it.k1, lds);
  if (H.Kinv) {
    double* C = H.Kinv + i0 * H.lda + j0;
    acc_foreach(acc, [&](int row, int col, double v) { C[(int64_t)row * H.lda + col] = v; });
  }
  // ---- epilogue: the four contractions of dL/dK with the kernel derivatives
  __syncthreads();  // GEMM staging area is free
  const int dp = H.dp;
  const double* xi_base;
  const double* xj_base;
  int pitch;
  if (H.x_in_lds) {
    pitch = dp + 1;
    double* Xi = lds;
    double* Xj = lds + NB * pitch;
    for (int e = threadIdx.x; e < NB * dp; e += 256) {
      const int r = e / dp, c = e - r * dp;
      Xi[r * pitch + c] = H.X[(i0 + r) * dp + c];
      Xj[r * pitch + c] = H.X[(j0 + r) * dp + c];
    }
    __syncthreads();
    xi_base = Xi;
    xj_base = Xj;
  } else {
    pitch = dp;
    xi_base = H.X + i0 * dp;
    xj_base = H.X + j0 * dp;
  }
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = w >> 1, wc = w & 1;
  double s_var = 0.0, s_ls = 0.0, s_bias = 0.0, s_noise = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * 64 + i * 16 + (l >> 4) + 4 * r;
      const int64_t gi = i0 + row;
      const double* xi = xi_base + row * pitch;
      double dot[4] = {0.0, 0.0, 0.0, 0.0};
      for (int c = 0; c < dp; ++c) {
        const double a = xi[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) dot[j] = fma(a, xj_base[(wc * 64 + j * 16 + (l & 15)) * pitch + c], dot[j]);
      }
      const double ai = gi < H.n ? H.alpha[gi] : 0.0;
      const double x2i = H.x2[gi];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wc * 64 + j * 16 + (l & 15);
        const int64_t gj = j0 + col;
        if (gi < H.n && gj <= gi) {
          double r2 = (x2i + H.x2[gj]) + (-2.0 * dot[j]);
          r2 = r2 < 0.0 ? 0.0 : r2;
          if (gi == gj) r2 = 0.0;
          const double krbf = H.var * exp(r2 * H.neg_half_inv_ls2);
          const double D = 0.5 * ((first_chunk ? ai * H.alpha[gj] : 0.0) - acc.c[i][j][r]);
          const double wt = gi == gj ? 1.0 : 2.0;  // the strict upper triangle mirrors the lower
          s_var += wt * D * krbf;
          s_ls += wt * D * krbf * r2;
          s_bias += wt * D;
          if (gi == gj) s_noise += D;
        }
      }
    }
  // fixed-order workgroup reduction: butterfly inside the wave, then the four waves in order
  __syncthreads();
  double v[4] = {s_var, s_ls, s_bias, s_noise};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off, 64);
    if (l == 0) lds[w * 4 + q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    H.part[b * 4 + q] = ((lds[q] + lds[4 + q]) + lds[8 + q]) + lds[12 + q];
  }
}

// red[2..5] = sum over tiles of part[t][0..3], fixed order.
__global__ __launch_bounds__(256) void hyper_reduce_kernel(const double* part, int64_t ntiles, double* red, unsigned long long ticket) {
  __shared__ double s[256][4];
  double a[4] = {0, 0, 0, 0};
  for (int64_t t = threadIdx.x; t < ntiles; t += 256)
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] += part[t * 4 + q];
#pragma unroll
  for (int q = 0; q < 4; ++q) s[threadIdx.x][q] = a[q];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off)
#pragma unroll
      for (int q = 0; q < 4; ++q) s[threadIdx.x][q] += s[threadIdx.x + off][q];
    __syncthreads();
  }
  if (threadIdx.x < 4) red[2 + threadIdx.x] = s[0][threadIdx.x];
  if (threadIdx.x == 0) post_ticket(red + 12, ticket);   // (red = h_fit + 2: word 14; lanes 0-3 are one wave: their stores precede the fence)
}

static int hyper_grad_impl(elfihip_gp* gp, bool store_kinv, double* grad) {
  elfihip_ctx* ctx = gp->ctx;
  if (!gp->factored)
    return fail(ctx, ELFIHIP_ERR_STATE, "GP is not factorised (call elfihip_gp_factorize first)");
  hipStream_t st = ctx->stream;
  const int64_t np = gp->np;
  const int64_t nt = np / NB;
  if (store_kinv && !gp->Kinv) {
    const size_t bytes = (size_t)gp->cap * gp->lda * sizeof(double);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&gp->Kinv), bytes);
    if (e != hipSuccess)
      return fail(ctx, e == hipErrorOutOfMemory ? ELFIHIP_ERR_NOMEM : ELFIHIP_ERR_HIP, "K^-1 allocation failed: %s",
                  hipGetErrorString(e));
    ELFIHIP_CHECK_HIP(ctx, hipMemsetAsync(gp->Kinv, 0, bytes, st));
  }
  // work list: (tile, k chunk) with chunks of about total depth / (2 workgroups per CU), longest first
  static thread_local std::vector<HyperItem> items;
  items.clear();
  const bool cached = gp->hyper_items_key == (store_kinv ? -np : np);
  int64_t depth = 0;
  for (int64_t ti = 0; ti < nt; ++ti) depth += (ti + 1) * (np - ti * NB);
  // measured (nlml_grad, ms): n=2048: 0.40 uncut, 0.26 with chunks of 512; n=4096: 0.83 uncut, 0.66 with 1024;
  // n=8192: 3.3 uncut, 3.6-3.9 with 2048-1024 (enough tiles to balance; every extra chunk repeats the epilogue)
  // (round 3: chunks down to 128 deep -- at the sizes a BOLFI run spends most of its searches on, n <= 2048, chunks of
  // 512 left 10-200 workgroups with one long tile each: 0.095 / 0.105 / 0.119 ms at n = 512 / 1024 / 2048)
  int64_t kchunk = round_up(depth / ((int64_t)ctx->cu_count * 3) + 1, NB);
  if (kchunk < NB) kchunk = NB;
  if (store_kinv || kchunk > np) kchunk = np;
  for (int64_t ti = 0; ti < nt && !cached; ++ti)
    for (int64_t tj = 0; tj <= ti; ++tj) {
      const int64_t len = np - ti * NB, nch = (len + kchunk - 1) / kchunk;
      const int64_t per = round_up((len + nch - 1) / nch, NB);  // equal chunks of a tile
      for (int64_t k0 = ti * NB; k0 < np; k0 += per)
        items.push_back(HyperItem{(int)ti, (int)tj, (int)k0, (int)std::min<int64_t>(k0 + per, np)});
    }
  std::stable_sort(items.begin(), items.end(),
                   [](const HyperItem& a, const HyperItem& c) { return a.k1 - a.k0 > c.k1 - c.k0; });
  const int64_t nitems = cached ? gp->hyper_items_n : (int64_t)items.size();
  const size_t part_bytes = (size_t)nitems * 4 * sizeof(double);
  ELFIHIP_CHECK_HIP(ctx, gp->ws.reserve(part_bytes));
  // the work list depends on the padded size only: uploaded once per size (a search evaluates ~100 gradients at one size)
  const int64_t list_key = store_kinv ? -np : np;
  if (gp->hyper_items_key != list_key) {
    ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));   // a launch in flight may still read the old list
    ELFIHIP_CHECK_HIP(ctx, gp->hyper_items.reserve((size_t)nitems * sizeof(HyperItem)));
    ELFIHIP_CHECK_HIP(ctx, hipMemcpy(gp->hyper_items.p, items.data(), (size_t)nitems * sizeof(HyperItem), hipMemcpyHostToDevice));
    gp->hyper_items_key = list_key;
    gp->hyper_items_n = nitems;
  }
  HyperItem* d_items = gp->hyper_items.as<HyperItem>();
  HyperArgs H;
  H.items = d_items;
  H.WT = gp->WT;
  H.X = gp->X;
  H.x2 = gp->x2;
  H.alpha = gp->alpha;
  H.Kinv = store_kinv ? gp->Kinv : nullptr;
  H.part = gp->ws.as<double>();
  H.lda = gp->lda;
  H.n = gp->n;
  H.np = np;
  H.dp = gp->dp;
  H.var = gp->var;
  H.neg_half_inv_ls2 = -0.5 / (gp->ls * gp->ls);
  size_t lds = GEMM_LDS_DOUBLES * sizeof(double);
  const size_t xlds = 2 * (size_t)NB * (gp->dp + 1) * sizeof(double);
  H.x_in_lds = xlds <= 96 * 1024;
  if (H.x_in_lds && xlds > lds) lds = xlds;
  ELFIHIP_CHECK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kinv_grad_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  prof_mark(gp, 0);
  hipLaunchKernelGGL(kinv_grad_kernel, dim3((unsigned)nitems), dim3(256), lds, st, H);
  hipLaunchKernelGGL(hyper_reduce_kernel, dim3(1), dim3(256), 0, st, H.part, nitems, gp->h_fit + 2, ++gp->hyper_ticket);   // pinned: no copy
  prof_mark(gp, 1);
  ELFIHIP_TRY(launch_status(ctx, "kinv_grad_kernel"));
  if (gp->profile)   // (event times are read below)
    ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  else
    ELFIHIP_TRY(host_wait_ticket(ctx, reinterpret_cast<const volatile unsigned long long*>(gp->h_fit + 14), gp->hyper_ticket));
  const double s[4] = {gp->h_fit[4], gp->h_fit[5], gp->h_fit[6], gp->h_fit[7]};
  prof_add(gp, ELFIHIP_PHASE_KINV_GRAD, 0, 1);
  if (store_kinv) {
    gp->has_kinv = true;
    gp->kinv_sym = false;   // the lower tiles are new: the upper ones are mirrored again before a product reads them
  }
  if (grad) {
    grad[0] = s[0] / gp->var;
    grad[1] = s[1] / (gp->ls * gp->ls * gp->ls);
    grad[2] = s[2];
    grad[3] = s[3];
  }
  return ELFIHIP_OK;
}

int form_kinv_impl(elfihip_gp* gp) { return hyper_grad_impl(gp, true, nullptr); }

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_gp_nlml_grad(elfihip_gp* gp, double* log_marginal, double* grad) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, grad != nullptr, "grad is NULL");
  DeviceGuard g(gp->ctx->device);
  ELFIHIP_TRY(hyper_grad_impl(gp, false, grad));
  if (log_marginal)
    *log_marginal = 0.5 * (-(double)gp->n * 1.8378770664093453 /* log(2 pi) */ - gp->logdet - gp->yKy);
  return ELFIHIP_OK;
}

int elfihip_gp_form_kinv(elfihip_gp* gp) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  DeviceGuard g(gp->ctx->device);
  return hyper_grad_impl(gp, true, nullptr);
}

}  // extern "C"
