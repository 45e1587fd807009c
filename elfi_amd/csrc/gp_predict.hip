// GP posterior mean / variance / gradients and the LCB acquisition at many points at once.
//
// Replaces GPyRegression.predict / predictive_gradients
// (elfi/methods/bo/gpy_regression.py:98-147,179-223; closed forms :127-140,:206-218) and
// LCBSC.evaluate / evaluate_gradient (elfi/methods/bo/acquisition.py:262-301).  The reference
// evaluates ONE point per call (three O(n^2) products with the dense K^-1 each, see
// SURVEY.md 3.3); here S points are evaluated together in groups of 16 columns:
//
//   kr[s][i] = s_f exp(-|x_s - X_i|^2 / 2 l^2)                       (S x n, VALU + exp)
//   mu_s     = sum_i (kr + s_b) alpha_i
//   v        = L^-1 (kr + s_b)^T        v[i][s] = sum_{k<=i} WT[k][i] kb[s][k]   (n x 16, MFMA f64)
//   var_s    = s_f + s_b - sum_i v[i][s]^2
//   u        = L^-T v = K^-1 kb^T       u[i][s] = sum_{k>=i} WT[i][k] v[k][s]    (n x 16, MFMA f64)
//   dmu_s    = sum_i alpha_i dk_si,   dvar_s = -2 sum_i u[i][s] dk_si,
//              dk_si = -(kr[s][i] / l^2) (x_s - X_i)                  (RBF part only, as GPy)
//
// The two triangular products stream L^-T once each (8 n^2 / 2 bytes): HBM/L3-bound for 16
// columns, so they are split over (row block, k chunk) pairs to fill the chip and reduced
// in a fixed order (deterministic).  WT's strictly-lower part is zero, so no masking.
#include <type_traits>

#include "gp.hpp"
#include "mfma_f64.hpp"

namespace elfihip {

constexpr int PC = 16;    // columns (query points) per pass
constexpr int KCH = 2;    // 128-blocks of k per workgroup (fewer partial sums to reduce than with 1)
constexpr int SLAB = 32;  // k-slab staged per step
constexpr int MAX_GROUP = 8;  // 16-point passes handled by one set of launches


// ---- kr[s][i], partial mu ----------------------------------------------------------
__global__ __launch_bounds__(256) void kstar_kernel(const double* X, const double* x2, const double* alpha,
                                                    const double* xs, const double* xs2, double* kr, double* mu_part,
                                                    int64_t n, int64_t np, int dp, double var,
                                                    double neg_half_inv_ls2, double bias) {
  __shared__ double red[256];
  const int s = blockIdx.y;
  // several 16-point passes in one launch (blockIdx.z): per-pass slices of the query points and outputs
  xs += (int64_t)blockIdx.z * PC * dp;
  xs2 += (int64_t)blockIdx.z * PC;
  kr += (int64_t)blockIdx.z * PC * np;
  mu_part += (int64_t)blockIdx.z * PC * gridDim.x;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double contrib = 0.0;
  if (i < np) {
    double k = 0.0;
    if (i < n) {
      double dot = 0.0;
      for (int c = 0; c < dp; ++c) dot += xs[s * dp + c] * X[i * dp + c];
      double r2 = (xs2[s] + x2[i]) + (-2.0 * dot);
      r2 = r2 > 0.0 ? r2 : 0.0;
      k = var * exp(r2 * neg_half_inv_ls2);
      contrib = (k + bias) * alpha[i];
    }
    kr[(int64_t)s * np + i] = k;
  }
  red[threadIdx.x] = contrib;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) mu_part[s * gridDim.x + blockIdx.x] = red[0];
}

// ---- triangular skinny products on the matrix cores ------------------------------------
// TRANS = true :  out[i][s] += sum_k WT[k][i] * (kr[s][k] + bias [k < n])      (v = L^-1 kb)
// TRANS = false:  out[i][s] += sum_k WT[i][k] * vin[k][s]                        (u = L^-T v)
// Workgroup (ib, kc): rows i in block ib, k in blocks [kc*KCH, kc*KCH+KCH) clipped to the
// triangle.  Writes part[kc][i][s] (zeros if the chunk is outside the triangle).
struct TriArgs {
  const double* WT;
  const double* kr;    // TRANS
  const double* vin;   // !TRANS
  double* part;
  int64_t lda, n, np;
  int nb, nkc;
  int npass;   // 16-point passes in this launch (see the blockIdx mapping in the kernel)
  double bias;
};

template <bool TRANS>
__global__ __launch_bounds__(256) void tri_apply_kernel(TriArgs T) {
  // One workgroup per (row block ib, chunk of KCH k blocks) inside the triangle; its part of L^-T is
  // streamed as 32-deep slabs with the loads of the next two slabs in flight (registers) while the
  // current one is multiplied -- these kernels are pure HBM/MALL streaming (S <= 16).
  // LDS: W slab + B slab.  TRANS: W as [k][i] pitch 144, B = kb as [s][k] pitch 34.
  //      !TRANS: W as [i][k] pitch 34, B = v as [k][s] pitch 16.
  extern __shared__ __align__(16) double sm[];
  constexpr int WP = TRANS ? 144 : 34;
  double* Ws = sm;
  double* Bs = sm + (TRANS ? SLAB * 144 : 128 * 34);
  // blockIdx.x = (ib / 8 * npass + pass) * 8 + ib % 8: workgroups are dealt to the 8 XCDs round-robin, so the
  // passes of one ib land on the same XCD, 8 dispatch slots apart, and share the L^-T block in that XCD's L2
  const int ib_lo = blockIdx.x & 7, bq = blockIdx.x >> 3;
  const int ib = (bq / T.npass) * 8 + ib_lo, pass = bq % T.npass, kc = blockIdx.y;
  if (ib >= T.nb) return;
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  T.kr += (int64_t)pass * PC * T.np;                               // per-pass slices
  if (!TRANS) T.vin += (int64_t)pass * T.np * PC;
  T.part += (int64_t)pass * T.nkc * T.np * PC;
  // k-block range of this chunk, clipped to the triangle (TRANS: k <= i, else k >= i)
  int kb0 = kc * KCH, kb1 = kb0 + KCH;
  if (TRANS) {
    if (kb1 > ib + 1) kb1 = ib + 1;
  } else {
    if (kb0 < ib) kb0 = ib;
    if (kb1 > T.nb) kb1 = T.nb;
  }
  if (kb0 >= kb1) return;  // chunk lies outside the triangle (the reduction never reads it)
  const int nslab = (kb1 - kb0) * (NB / SLAB);  // 4 or 8
  v4d acc[2];
  acc[0] = (v4d){0, 0, 0, 0};
  acc[1] = (v4d){0, 0, 0, 0};
  const int64_t i0 = (int64_t)ib * NB, kbase = (int64_t)kb0 * NB;
  // per-thread source / destination of pair p inside a slab (slab s adds s * step to the source)
  auto src = [&](int p) -> const double* {
    if (TRANS) return T.WT + (kbase + (t >> 6) + 4 * p) * T.lda + i0 + 2 * (t & 63);  // row k of the slab, 1 KiB per row
    return T.WT + (i0 + (t >> 4) + 16 * p) * T.lda + kbase + 2 * (t & 15);            // row i, 256 B of the slab per row
  };
  auto dst = [&](int p) -> int {
    if (TRANS) return ((t >> 6) + 4 * p) * 144 + 2 * (t & 63);
    return ((t >> 4) + 16 * p) * 34 + 2 * (t & 15);
  };
  const int64_t step = TRANS ? (int64_t)SLAB * T.lda : (int64_t)SLAB;
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d r0[8], r1[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) r0[p] = *reinterpret_cast<const v2d*>(src(p));
#pragma unroll
  for (int p = 0; p < 8; ++p) r1[p] = *reinterpret_cast<const v2d*>(src(p) + step);
  // four slabs, written out (compile-time slab index keeps r0 / r1 in registers)
#define ELFIHIP_SLAB_STEP(sl, cur)                                                                        \
  {                                                                                                      \
    const int64_t k0 = kbase + (int64_t)(sl)*SLAB;                                                       \
    __syncthreads();                                                                                     \
    _Pragma("unroll") for (int p = 0; p < 8; ++p) *reinterpret_cast<v2d*>(Ws + dst(p)) = cur[p];    \
    if ((sl) + 2 < nslab) {                                                                              \
      _Pragma("unroll") for (int p = 0; p < 8; ++p) cur[p] =                                             \
          *reinterpret_cast<const v2d*>(src(p) + ((sl) + 2) * step);                                     \
    }                                                                                                    \
    if (TRANS) {                                                                                         \
      _Pragma("unroll") for (int e0 = 0; e0 < PC * SLAB; e0 += 256) {                                    \
        const int e = e0 + t;                                                                            \
        const int s_ = e >> 5, kk = e & 31;                                                              \
        const int64_t k = k0 + kk;                                                                       \
        Bs[s_ * 34 + kk] = (k < T.n) ? (T.kr[(int64_t)s_ * T.np + k] + T.bias) : 0.0;                    \
      }                                                                                                  \
    } else {                                                                                             \
      _Pragma("unroll") for (int e0 = 0; e0 < SLAB * PC; e0 += 256) Bs[e0 + t] = T.vin[k0 * PC + e0 + t]; \
    }                                                                                                    \
    __syncthreads();                                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < SLAB / 4; ++ks) {                                            \
      const int kq = 4 * ks + (l >> 4);                                                                  \
      const double bq = TRANS ? Bs[(l & 15) * 34 + kq] : Bs[kq * PC + (l & 15)];                         \
      _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                    \
        const int i = (2 * w + m) * 16 + (l & 15);                                                       \
        const double aq = TRANS ? Ws[kq * WP + i] : Ws[i * WP + kq];                                     \
        acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bq, acc[m], 0, 0, 0);                          \
      }                                                                                                  \
    }                                                                                                    \
  }
  ELFIHIP_SLAB_STEP(0, r0)
  ELFIHIP_SLAB_STEP(1, r1)
  ELFIHIP_SLAB_STEP(2, r0)
  ELFIHIP_SLAB_STEP(3, r1)
  if (nslab > 4) {  // workgroup-uniform: second k block of the chunk
    ELFIHIP_SLAB_STEP(4, r0)
    ELFIHIP_SLAB_STEP(5, r1)
    ELFIHIP_SLAB_STEP(6, r0)
    ELFIHIP_SLAB_STEP(7, r1)
  }
#undef ELFIHIP_SLAB_STEP
  static_assert(KCH * (NB / SLAB) == 8, "slab sequence above is written out for up to eight slabs");
  double* out = T.part + ((int64_t)kc * T.np + i0) * PC;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (2 * w + m) * 16 + (l >> 4) + 4 * r;
      out[row * PC + (l & 15)] = acc[m][r];
    }
}

// v[i][s] = sum_kc part[kc][i][s] (fixed order); optional per-block partials of sum_i v^2.
__global__ __launch_bounds__(256) void tri_reduce_kernel(const double* part, double* out, double* sq_part,
                                                         int64_t np, int nkc, int kc_lo_is_row, int want_sq) {
  __shared__ double red[256];
  part += (int64_t)blockIdx.y * nkc * np * PC;   // per-pass slices (blockIdx.y = pass)
  out += (int64_t)blockIdx.y * np * PC;
  if (want_sq) sq_part += (int64_t)blockIdx.y * gridDim.x * PC;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // element (i, s), 16 rows per block
  double v = 0.0;
  if (e < np * PC) {
    const int64_t i = e / PC;
    const int ib = (int)(i / NB);
    // chunks that can be non-zero: TRANS (kc_lo_is_row == 0): kc*KCH <= ib ; else kc*KCH+KCH > ib
    const int lo = kc_lo_is_row ? ib / KCH : 0;
    const int hi = kc_lo_is_row ? nkc : ib / KCH + 1;
    double v4[4] = {0, 0, 0, 0};
    int kc = lo;
    for (; kc + 4 <= hi; kc += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) v4[u] += part[(int64_t)(kc + u) * np * PC + e];
    }
    for (; kc < hi; ++kc) v4[0] += part[(int64_t)kc * np * PC + e];
    v = (v4[0] + v4[1]) + (v4[2] + v4[3]);
    out[e] = v;
  }
  if (want_sq) {
    red[threadIdx.x] = v * v;
    __syncthreads();
    // columns are e % 16: reduce the 16 rows of this block per column, fixed order
    if (threadIdx.x < PC) {
      double s = 0.0;
      for (int r = 0; r < 16; ++r) s += red[r * PC + threadIdx.x];
      sq_part[(int64_t)blockIdx.x * PC + threadIdx.x] = s;
    }
  }
}

// ---- gradients ---------------------------------------------------------------------
// g_part[s][chunk][0..dp) = sum_i alpha_i kr_si (x_s - X_i),  [dp..2dp) = sum_i u_is kr_si (x_s - X_i)
// One evidence row per thread (256 rows per workgroup); per dimension a wave butterfly, then the four
// wave partials in fixed order.
__global__ __launch_bounds__(256) void grad_kernel(const double* X, const double* alpha, const double* xs,
                                                   const double* kr, const double* u, double* g_part, int64_t n,
                                                   int64_t np, int dp, int rows_per_block) {
  __shared__ double red[4][2];
  const int s = blockIdx.y;
  xs += (int64_t)blockIdx.z * PC * dp;                 // per-pass slices (blockIdx.z = pass)
  kr += (int64_t)blockIdx.z * PC * np;
  u += (int64_t)blockIdx.z * np * PC;
  g_part += (int64_t)blockIdx.z * PC * gridDim.x * 2 * dp;
  const int64_t i = (int64_t)blockIdx.x * rows_per_block + threadIdx.x;
  double* outp = g_part + ((int64_t)s * gridDim.x + blockIdx.x) * 2 * dp;
  double c1 = 0.0, c2 = 0.0;
  if (i < n) {
    const double k = kr[(int64_t)s * np + i];
    c1 = alpha[i] * k;
    c2 = u[i * PC + s] * k;
  }
  const int64_t ic = i < n ? i : 0;
  for (int a = 0; a < dp; ++a) {
    const double diff = xs[s * dp + a] - X[ic * dp + a];
    double v1 = c1 * diff, v2 = c2 * diff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      v1 += __shfl_xor(v1, off, 64);
      v2 += __shfl_xor(v2, off, 64);
    }
    __syncthreads();  // red free
    if ((threadIdx.x & 63) == 0) {
      red[threadIdx.x >> 6][0] = v1;
      red[threadIdx.x >> 6][1] = v2;
    }
    __syncthreads();
    if (threadIdx.x < 2) outp[threadIdx.x * dp + a] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) +
                                                      red[3][threadIdx.x];
  }
}

// ---- final assembly: mu, var, dmu, dvar, LCB value and gradient ---------------------------
// out layout per pass: mu[16] var[16] val[16] dmu[16*dp] dvar[16*dp] grad[16*dp]
__global__ __launch_bounds__(256) void finish_kernel(const double* mu_part, int nblk_k, const double* var_part,
                                                     int nblk_v, const double* g_part, int ngc, double* out, int dp,
                                                     int S_left, double prior_var, double noise_add, double inv_ls2,
                                                     double beta, int with_grad) {
  // one workgroup per pass (blockIdx.x); S_left = real points from the first pass of this launch on
  mu_part += (int64_t)blockIdx.x * PC * nblk_k;
  var_part += (int64_t)blockIdx.x * nblk_v * PC;
  g_part += (int64_t)blockIdx.x * PC * ngc * 2 * dp;
  out += (int64_t)blockIdx.x * (3 * PC + 3 * PC * dp);
  const int S = S_left - (int)blockIdx.x * PC;   // columns >= S are padding
  // 16 groups of 16 lanes: lane (s, j) sums the partial blocks b = j, j+16, ... of column s;
  // the 16 group sums are then added in a fixed order.
  __shared__ double red_m[16][PC], red_q[16][PC];
  const int s = threadIdx.x & 15, j = threadIdx.x >> 4;
  double m = 0.0, q = 0.0;
  for (int b = j; b < nblk_k; b += 16) m += mu_part[s * nblk_k + b];
  {
    double q4[4] = {0, 0, 0, 0};  // independent accumulators: the loads pipeline
    int b = j;
    for (; b + 48 < nblk_v; b += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) q4[u] += var_part[(int64_t)(b + 16 * u) * PC + s];
    }
    for (; b < nblk_v; b += 16) q4[0] += var_part[(int64_t)b * PC + s];
    q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
  }
  red_m[j][s] = m;
  red_q[j][s] = q;
  __syncthreads();
  double* mu = out;
  double* var = out + PC;
  double* val = out + 2 * PC;
  double* dmu = out + 3 * PC;
  double* dvar = dmu + PC * dp;
  double* grad = dvar + PC * dp;
  __shared__ double vfin[PC];
  if (j == 0) {
    if (s >= S) {
      mu[s] = 0;
      var[s] = 0;
      val[s] = 0;
      vfin[s] = 1.0;
    } else {
      m = 0.0;
      q = 0.0;
      for (int g = 0; g < 16; ++g) {
        m += red_m[g][s];
        q += red_q[g][s];
      }
      double v = prior_var - q;
      v = v > 1e-15 ? v : 1e-15;  // [GPy-upstream] predict clips the variance at 1e-15
      mu[s] = m;
      var[s] = v + noise_add;
      val[s] = m - sqrt(beta * v);
      vfin[s] = v;
    }
  }
  __syncthreads();
  if (with_grad && s < S) {
    // lane (s, j) assembles dimensions a = j, j + 16, ...: chunk partials in fixed order, four
    // independent accumulators so the loads pipeline
    const double sc = sqrt(beta / vfin[s]);
    for (int a = j; a < dp; a += 16) {
      const double* gp1 = g_part + (int64_t)s * ngc * 2 * dp + a;
      double g1[4] = {0, 0, 0, 0}, g2[4] = {0, 0, 0, 0};
      int c = 0;
      for (; c + 4 <= ngc; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          g1[u] += gp1[(int64_t)(c + u) * 2 * dp];
          g2[u] += gp1[(int64_t)(c + u) * 2 * dp + dp];
        }
      }
      for (; c < ngc; ++c) {
        g1[0] += gp1[(int64_t)c * 2 * dp];
        g2[0] += gp1[(int64_t)c * 2 * dp + dp];
      }
      const double s1 = (g1[0] + g1[1]) + (g1[2] + g1[3]), s2 = (g2[0] + g2[1]) + (g2[2] + g2[3]);
      const double dm = -inv_ls2 * s1;
      const double dv = 2.0 * inv_ls2 * s2;  // -2 * sum u_i dk_i, dk_i = -(k/l^2)(x - X_i)
      dmu[s * dp + a] = dm;
      dvar[s * dp + a] = dv;
      grad[s * dp + a] = dm - 0.5 * dv * sc;
    }
  }
}

static int ensure_ws(elfihip_gp* gp, PredictWs* W, int64_t npass) {
  elfihip_ctx* ctx = gp->ctx;
  const int64_t np = gp->np;
  const int nb = (int)(np / NB);
  W->nblk_k = (int)((np + 255) / 256);
  W->nkc = (nb + KCH - 1) / KCH;
  const int rows_per_block = 256;
  W->ngc = (int)((gp->n + rows_per_block - 1) / rows_per_block);
  size_t off = 0;
  auto take = [&](size_t doubles) {
    size_t o = off;
    off += (doubles + 15) & ~(size_t)15;
    return o;
  };
  // passes run `group` at a time in one set of launches; the scratch below is per pass of a group
  const size_t per_pass = (size_t)(W->nkc + 3) * np * PC * sizeof(double);
  int64_t group = (int64_t)(((size_t)768 << 20) / per_pass);
  group = group < 1 ? 1 : (group > MAX_GROUP ? MAX_GROUP : group);
  if (group > npass) group = npass;
  W->group = (int)group;
  const size_t g = (size_t)group;
  const size_t o_xs = take((size_t)npass * PC * gp->dp), o_xs2 = take((size_t)npass * PC),
               o_kr = take(g * PC * np), o_part = take(g * W->nkc * np * PC), o_v = take(g * np * PC),
               o_u = take(g * np * PC), o_mu = take(g * PC * W->nblk_k), o_var = take(g * (np * PC / 256) * PC + 16),
               o_g = take(g * PC * W->ngc * 2 * gp->dp), o_out = take((size_t)npass * (3 * PC + 3 * PC * gp->dp));
  ELFIHIP_CHECK_HIP(ctx, gp->ws.reserve(off * sizeof(double)));
  double* base = gp->ws.as<double>();
  W->xs = base + o_xs;
  W->xs2 = base + o_xs2;
  W->kr = base + o_kr;
  W->part = base + o_part;
  W->v = base + o_v;
  W->u = base + o_u;
  W->mu_part = base + o_mu;
  W->var_part = base + o_var;
  W->g_part = base + o_g;
  W->out = base + o_out;
  return ELFIHIP_OK;
}

// ---- a prediction call in four steps: host preparation, input fill, device enqueue, result read.
// (Replaying the enqueue part from a hipGraph was measured and is slower here: +90 us per launch on
// this stack for a 9-node graph with two copy nodes, against ~25 us of plain launch cost.)
int predict_prepare(elfihip_gp* gp, int64_t S, PredictPlan* P) {
  elfihip_ctx* ctx = gp->ctx;
  if (!gp->factored)
    return fail(ctx, ELFIHIP_ERR_STATE, "GP is not factorised (call elfihip_gp_factorize after changing data)");
  P->npass = (S + PC - 1) / PC;
  PredictWs W;
  ELFIHIP_TRY(ensure_ws(gp, &W, P->npass));
  P->ws = W;
  const int dp = gp->dp;
  P->outsz = (size_t)3 * PC + 3 * PC * dp;
  P->n_in = (size_t)P->npass * PC * dp + (size_t)P->npass * PC;
  P->n_out = (size_t)P->npass * P->outsz;
  if (gp->h_cap < P->n_in + P->n_out) {  // pinned staging: the copies are true async DMA, no bounce buffer
    if (gp->h_stage) ELFIHIP_CHECK_HIP(ctx, hipHostFree(gp->h_stage));
    gp->h_stage = nullptr;
    gp->h_cap = 0;
    const size_t want = 2 * (P->n_in + P->n_out) + 1024;
    ELFIHIP_CHECK_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&gp->h_stage), want * sizeof(double), hipHostMallocDefault));
    gp->h_cap = want;
  }
  P->hx = gp->h_stage;
  P->hout = gp->h_stage + P->n_in;
  return ELFIHIP_OK;
}

void predict_fill(const elfihip_gp* gp, const PredictPlan& P, const double* Xs, int64_t S) {
  const int dp = gp->dp, d = gp->d;
  std::fill(P.hx, P.hx + P.n_in, 0.0);
  double* hx2 = P.hx + (size_t)P.npass * PC * dp;
  for (int64_t s = 0; s < S; ++s) {
    double q = 0.0;
    for (int c = 0; c < d; ++c) {
      const double x = Xs[s * d + c];
      P.hx[(size_t)s * dp + c] = x;
      q += x * x;
    }
    hx2[s] = q;
  }
}

// All points go up in one copy; the 16-point passes run `group` at a time inside each launch (the
// group scratch is reused in stream order); all results come down in one copy; no synchronisation here.
// S_active: number of real points (columns beyond it are computed on zero inputs and ignored).
int predict_enqueue(elfihip_gp* gp, const PredictPlan& P, int64_t S_active, int mode, int noiseless, double beta) {
  elfihip_ctx* ctx = gp->ctx;
  hipStream_t st = ctx->stream;
  const PredictWs& W = P.ws;
  const int dp = gp->dp;
  const int64_t np = gp->np;
  const int nb = (int)(np / NB);
  const double inv_ls2 = 1.0 / (gp->ls * gp->ls);
  // W.xs and W.xs2 are adjacent in the workspace (PC * dp is a multiple of the 16-double granule)
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(W.xs, P.hx, P.n_in * sizeof(double), hipMemcpyHostToDevice, st));
  const size_t lds_t = (SLAB * 144 + PC * 34) * sizeof(double);
  const size_t lds_n = (128 * 34 + SLAB * PC) * sizeof(double);
  const int rblocks = (int)(np * PC / 256);
  const unsigned nb8 = (unsigned)((nb + 7) / 8 * 8);
  for (int64_t pass0 = 0; pass0 < P.npass; pass0 += W.group) {
    const unsigned g = (unsigned)((P.npass - pass0) < W.group ? (P.npass - pass0) : W.group);
    int s_left = (int)(S_active - pass0 * PC);
    if (s_left < 0) s_left = 0;
    const double* xs = W.xs + (size_t)pass0 * PC * dp;
    const double* xs2 = W.xs2 + (size_t)pass0 * PC;
    double* out = W.out + (size_t)pass0 * P.outsz;
    hipLaunchKernelGGL(kstar_kernel, dim3(W.nblk_k, PC, g), dim3(256), 0, st, gp->X, gp->x2, gp->alpha, xs, xs2,
                       W.kr, W.mu_part, gp->n, np, dp, gp->var, -0.5 * inv_ls2, gp->bias);
    TriArgs T;
    T.WT = gp->WT;
    T.kr = W.kr;
    T.vin = nullptr;
    T.part = W.part;
    T.lda = gp->lda;
    T.n = gp->n;
    T.np = np;
    T.nb = nb;
    T.nkc = W.nkc;
    T.npass = (int)g;
    T.bias = gp->bias;
    hipLaunchKernelGGL((tri_apply_kernel<true>), dim3(nb8 * g, W.nkc), dim3(256), lds_t, st, T);
    hipLaunchKernelGGL(tri_reduce_kernel, dim3(rblocks, g), dim3(256), 0, st, W.part, W.v, W.var_part, np, W.nkc, 0,
                       1);
    if (mode == 1) {
      T.vin = W.v;
      hipLaunchKernelGGL((tri_apply_kernel<false>), dim3(nb8 * g, W.nkc), dim3(256), lds_n, st, T);
      hipLaunchKernelGGL(tri_reduce_kernel, dim3(rblocks, g), dim3(256), 0, st, W.part, W.u, (double*)nullptr, np,
                         W.nkc, 1, 0);
      hipLaunchKernelGGL(grad_kernel, dim3(W.ngc, PC, g), dim3(256), 0, st, gp->X, gp->alpha, xs, W.kr, W.u,
                         W.g_part, gp->n, np, dp, 256);
    }
    hipLaunchKernelGGL(finish_kernel, dim3(g), dim3(256), 0, st, W.mu_part, W.nblk_k, W.var_part, rblocks, W.g_part,
                       W.ngc, out, dp, s_left, gp->var + gp->bias, noiseless ? 0.0 : gp->noise, inv_ls2, beta, mode);
  }
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(P.hout, W.out, P.n_out * sizeof(double), hipMemcpyDeviceToHost, st));
  return ELFIHIP_OK;
}

void predict_read(const elfihip_gp* gp, const PredictPlan& P, int64_t S, double* mu, double* var, double* dmu,
                  double* dvar, double* val, double* grad) {
  const int dp = gp->dp, d = gp->d;
  for (int64_t s = 0; s < S; ++s) {
    const double* o = P.hout + (size_t)(s / PC) * P.outsz;
    const int q = (int)(s % PC);
    if (mu) mu[s] = o[q];
    if (var) var[s] = o[PC + q];
    if (val) val[s] = o[2 * PC + q];
    for (int c = 0; c < d; ++c) {
      if (dmu) dmu[s * d + c] = o[3 * PC + q * dp + c];
      if (dvar) dvar[s * d + c] = o[3 * PC + PC * dp + q * dp + c];
      if (grad) grad[s * d + c] = o[3 * PC + 2 * PC * dp + q * dp + c];
    }
  }
}

// mode: 0 = mean/var only, 1 = + gradients (and LCB)
int predict_impl(elfihip_gp* gp, const double* Xs, int64_t S, int mode, int noiseless, double beta,
                        double* mu, double* var, double* dmu, double* dvar, double* val, double* grad) {
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, S >= 0, "negative S");
  if (S == 0) return ELFIHIP_OK;
  ELFIHIP_REQUIRE(ctx, Xs, "Xs is NULL");
  PredictPlan P;
  ELFIHIP_TRY(predict_prepare(gp, S, &P));
  predict_fill(gp, P, Xs, S);
  ELFIHIP_TRY(predict_enqueue(gp, P, S, mode, noiseless, beta));
  ELFIHIP_TRY(launch_status(ctx, "predict kernels"));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  predict_read(gp, P, S, mu, var, dmu, dvar, val, grad);
  return ELFIHIP_OK;
}


// ---- incremental extension of the factorisation by one evidence point ------------------------
// Adding (x, y) with unchanged hyper-parameters borders Ky by one row/column:
//   k = K(X, x) + s_b,   l = L^-1 k,   d = sqrt(k(x,x) + s_n + jitter - l.l)        new row of L: [l^T, d]
//   u = L^-T l,   new column of L^-T: [-u/d ; 1/d]
//   z_new = (y - l.z)/d,   alpha += (new column of L^-T) z_new,   logdet += 2 log d,   y'K^-1 y += z_new^2
// l and u are the two triangular products the predictor already runs (one column instead of 16), so
// an update is two passes over L^-T (8 n^2 bytes) instead of the 2 n^3/3 flops of a rebuild.
__global__ __launch_bounds__(256) void extend_scalars_kernel(const double* v, const double* z, const double* sq_part,
                                                             int nblk_v, double knn, double ynew, int64_t n,
                                                             double* red, int* info, int pivot_index) {
  __shared__ double s0[256], s1[256];
  double ll = 0.0, lz = 0.0;
  for (int b = threadIdx.x; b < nblk_v; b += 256) ll += sq_part[(int64_t)b * PC];  // column 0 of the per-block sums
  for (int64_t i = threadIdx.x; i < n; i += 256) lz += v[i * PC] * z[i];
  s0[threadIdx.x] = ll;
  s1[threadIdx.x] = lz;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      s0[threadIdx.x] += s0[threadIdx.x + off];
      s1[threadIdx.x] += s1[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double d2 = knn - s0[0];
    if (!(d2 > 0.0)) atomicCAS(info, 0, pivot_index);
    const double d = sqrt(d2 > 0.0 ? d2 : 1.0);
    red[8] = d;
    red[9] = (ynew - s1[0]) / d;  // z_new
  }
}

__global__ __launch_bounds__(256) void extend_write_kernel(const double* v, const double* u, const double* red,
                                                           double* A, double* WT, double* alpha, int64_t lda,
                                                           int64_t n, int64_t np) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double d = red[8], zn = red[9];
  if (j < n) {
    A[n * lda + j] = v[j * PC];              // new row of L
    const double w = -u[j * PC] / d;         // new column of L^-T
    WT[j * lda + n] = w;
    alpha[j] += w * zn;
  } else if (j == n) {
    A[n * lda + n] = d;
    WT[n * lda + n] = 1.0 / d;
    alpha[n] = zn / d;
    A[np * lda + n] = zn;                    // z = L^-1 y lives in row np of A
  }
}

static int extend_one(elfihip_gp* gp, const double* x, double ynew) {
  elfihip_ctx* ctx = gp->ctx;
  hipStream_t st = ctx->stream;
  PredictWs W;
  ELFIHIP_TRY(ensure_ws(gp, &W, 1));
  const int dp = gp->dp, d = gp->d;
  const int64_t np = gp->np, n = gp->n;
  const int nb = (int)(np / NB);
  const double inv_ls2 = 1.0 / (gp->ls * gp->ls);
  static thread_local std::vector<double> hx;
  hx.assign((size_t)PC * dp + PC, 0.0);
  double q = 0.0;
  for (int c = 0; c < d; ++c) {
    hx[c] = x[c];
    q += x[c] * x[c];
  }
  hx[(size_t)PC * dp] = q;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(W.xs, hx.data(), (size_t)PC * dp * sizeof(double), hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(W.xs2, hx.data() + (size_t)PC * dp, PC * sizeof(double), hipMemcpyHostToDevice, st));
  // the evidence arrays themselves (row n): padded x, |x|^2, y
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(gp->X + n * dp, hx.data(), (size_t)dp * sizeof(double), hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(gp->x2 + n, hx.data() + (size_t)PC * dp, sizeof(double), hipMemcpyHostToDevice, st));
  hx[(size_t)PC * dp + 1] = ynew;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(gp->y + n, hx.data() + (size_t)PC * dp + 1, sizeof(double), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(kstar_kernel, dim3(W.nblk_k, PC), dim3(256), 0, st, gp->X, gp->x2, gp->alpha, W.xs, W.xs2, W.kr,
                     W.mu_part, n, np, dp, gp->var, -0.5 * inv_ls2, gp->bias);
  TriArgs T;
  T.WT = gp->WT;
  T.kr = W.kr;
  T.vin = nullptr;
  T.part = W.part;
  T.lda = gp->lda;
  T.n = n;
  T.np = np;
  T.nb = nb;
  T.nkc = W.nkc;
  T.npass = 1;
  T.bias = gp->bias;
  const size_t lds_t = (SLAB * 144 + PC * 34) * sizeof(double);
  const size_t lds_n = (128 * 34 + SLAB * PC) * sizeof(double);
  const int rblocks = (int)(np * PC / 256);
  hipLaunchKernelGGL((tri_apply_kernel<true>), dim3((nb + 7) / 8 * 8, W.nkc), dim3(256), lds_t, st, T);
  hipLaunchKernelGGL(tri_reduce_kernel, dim3(rblocks), dim3(256), 0, st, W.part, W.v, W.var_part, np, W.nkc, 0, 1);
  const double* z = gp->A + np * gp->lda;
  hipLaunchKernelGGL(extend_scalars_kernel, dim3(1), dim3(256), 0, st, W.v, z, W.var_part, rblocks,
                     gp->var + gp->bias + gp->noise + GP_JITTER, ynew, n, gp->red, gp->info, (int)n + 1);
  T.vin = W.v;
  hipLaunchKernelGGL((tri_apply_kernel<false>), dim3((nb + 7) / 8 * 8, W.nkc), dim3(256), lds_n, st, T);
  hipLaunchKernelGGL(tri_reduce_kernel, dim3(rblocks), dim3(256), 0, st, W.part, W.u, (double*)nullptr, np, W.nkc, 1, 0);
  hipLaunchKernelGGL(extend_write_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, W.v, W.u, gp->red,
                     gp->A, gp->WT, gp->alpha, gp->lda, n, np);
  ELFIHIP_TRY(launch_status(ctx, "extend kernels"));
  double sc[2];
  int info = 0;
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(sc, gp->red + 8, sizeof sc, hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(&info, gp->info, sizeof info, hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  if (info != 0) {
    gp->factored = false;
    return fail(ctx, ELFIHIP_ERR_NOT_PD, "covariance matrix is not positive definite (pivot %d <= 0)", info);
  }
  gp->logdet += 2.0 * log(sc[0]);
  gp->yKy += sc[1] * sc[1];
  gp->n = n + 1;
  gp->has_kinv = false;
  return ELFIHIP_OK;
}

}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_gp_predict(elfihip_gp* gp, const double* Xs, int64_t S, int noiseless, double* mu, double* var) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  DeviceGuard g(gp->ctx->device);
  return predict_impl(gp, Xs, S, 0, noiseless, 0.0, mu, var, nullptr, nullptr, nullptr, nullptr);
}

int elfihip_gp_predict_grad(elfihip_gp* gp, const double* Xs, int64_t S, double* mu, double* var, double* dmu,
                            double* dvar) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  DeviceGuard g(gp->ctx->device);
  return predict_impl(gp, Xs, S, 1, 1, 0.0, mu, var, dmu, dvar, nullptr, nullptr);
}

int elfihip_gp_lcb(elfihip_gp* gp, const double* Xs, int64_t S, double beta, double* val, double* grad) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  ELFIHIP_REQUIRE(gp->ctx, beta >= 0, "beta must be non-negative");
  DeviceGuard g(gp->ctx->device);
  return predict_impl(gp, Xs, S, grad ? 1 : 0, 1, beta, nullptr, nullptr, nullptr, nullptr, val, grad);
}

int elfihip_gp_extend(elfihip_gp* gp, const double* X_new, const double* y_new, int64_t k, double* log_marginal) {
  if (!gp) return fail(nullptr, ELFIHIP_ERR_ARG, "gp is NULL");
  elfihip_ctx* ctx = gp->ctx;
  ELFIHIP_REQUIRE(ctx, k >= 0 && (k == 0 || (X_new && y_new)), "bad arguments");
  ELFIHIP_REQUIRE(ctx, gp->n + k <= gp->cap, "evidence count %lld exceeds the GP capacity %lld",
                  (long long)(gp->n + k), (long long)gp->cap);
  DeviceGuard g(ctx->device);
  int64_t done = 0;
  // bordering works inside the current padded size; crossing a 128 boundary (or an unfactorised
  // GP) takes the ordinary path: append the rest and rebuild
  while (done < k && gp->factored && gp->n > 0 && gp->n < gp->np) {
    ELFIHIP_TRY(extend_one(gp, X_new + done * gp->d, y_new[done]));
    ++done;
  }
  if (done < k) {
    ELFIHIP_TRY(elfihip_gp_append(gp, X_new + done * gp->d, y_new + done, k - done));
    ELFIHIP_TRY(elfihip_gp_factorize(gp, nullptr));
  }
  if (log_marginal)
    *log_marginal = 0.5 * (-(double)gp->n * 1.8378770664093453 /* log(2 pi) */ - gp->logdet - gp->yKy);
  return ELFIHIP_OK;
}

}  // extern "C"
