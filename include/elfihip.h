/*
 * elfihip.h -- C ABI of libelfihip.so: the MI355X (gfx950) hot path for ELFI.
 *
 * Drop-in boundary.  Every entry point replaces one piece of
 * arithmetic that reference ELFI (v0.8.7, /root/reference) delegates to SciPy / GPy /
 * NumPy on the host.  The reference site each function stands in for is cited next
 * to its declaration as `elfi/<file>:<lines>`.
 *
 * Conventions
 *  - plain C types only: pointers, sizes, doubles.  No torch / numpy types.
 *  - every function returns an int status (ELFIHIP_OK == 0); the text of the last
 *    failure on a context is available from elfihip_last_error().
 *  - "host" entry points take HOST pointers; the library stages through device
 *    buffers it owns and never keeps a caller pointer after it returns.
 *  - "_dev" twins take DEVICE pointers (every pointer argument, including the small
 *    y / w vectors), enqueue on the context's stream and return without
 *    synchronising; results are ordered with later work on the same stream.
 *  - a context is bound to one GPU and one HIP stream; it is not thread-safe, but
 *    distinct contexts may be used from distinct threads.
 *  - all floating point is IEEE binary64; row sums are accumulated in the same
 *    left-to-right order SciPy's cdist uses, with FMA contraction disabled, so the
 *    euclidean / cityblock / chebyshev families reproduce cdist bit for bit.
 */
#ifndef ELFIHIP_H
#define ELFIHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELFIHIP_VERSION 100 /* 0.1.0 */

/* status codes */
enum {
  ELFIHIP_OK = 0,
  ELFIHIP_ERR_ARG = 1,      /* bad argument / shape -> ValueError on the Python side   */
  ELFIHIP_ERR_HIP = 2,      /* HIP runtime failure  -> RuntimeError                      */
  ELFIHIP_ERR_NOT_PD = 3,   /* Cholesky met a non-positive pivot -> np.linalg.LinAlgError */
  ELFIHIP_ERR_STATE = 4,    /* call made in the wrong state (e.g. predict before factor) */
  ELFIHIP_ERR_NOMEM = 5
};

/* metrics of scipy.spatial.distance.cdist that elfi.Distance accepts by name
 * (elfi/model/elfi_model.py:1020-1037). */
enum {
  ELFIHIP_EUCLIDEAN = 0,   /* sqrt(sum w_j (x_j-y_j)^2)            */
  ELFIHIP_SQEUCLIDEAN = 1, /* sum w_j (x_j-y_j)^2                  */
  ELFIHIP_CITYBLOCK = 2,   /* sum w_j |x_j-y_j|                    */
  ELFIHIP_CHEBYSHEV = 3,   /* max_j |x_j-y_j|   (w ignored unless 0: SciPy drops w_j==0 columns) */
  ELFIHIP_MINKOWSKI = 4,   /* (sum w_j |x_j-y_j|^p)^(1/p)          */
  ELFIHIP_SEUCLIDEAN = 5,  /* sqrt(sum (x_j-y_j)^2 / V_j), aux = V */
  ELFIHIP_MAHALANOBIS = 6  /* sqrt(d' VI d), aux = VI (m x m row-major) */
};

typedef struct elfihip_ctx elfihip_ctx;
typedef struct elfihip_gp elfihip_gp;

/* ------------------------------------------------------------------ context */
int elfihip_version(void);
int elfihip_device_count(int* count);
/* device < 0: use the current HIP device. */
int elfihip_ctx_create(int device, elfihip_ctx** ctx);
int elfihip_ctx_destroy(elfihip_ctx* ctx);
/* Text of the last error on ctx (or the last context-less error if ctx == NULL). */
const char* elfihip_last_error(const elfihip_ctx* ctx);
/* Adopt an externally owned hipStream_t (e.g. torch's current stream); NULL restores
 * the context's own stream.  The own stream is NON-BLOCKING: it is not ordered against
 * the null stream or any other stream, so a caller that fills device buffers elsewhere
 * (the *_dev entry points) either adopts that stream here or synchronises it before the
 * call; the *_dev entry points themselves are asynchronous on the context's stream. */
int elfihip_ctx_set_stream(elfihip_ctx* ctx, void* hip_stream);
int elfihip_ctx_synchronize(elfihip_ctx* ctx);
/* Device properties the roofline needs: CU count, clock (kHz), memory clock (kHz),
 * bus width (bits), total global memory (bytes).  Any pointer may be NULL. */
/* Page-locked host memory for results that leave the device in bulk (the (n, m) rows of a device-side simulator: a
 * pageable destination takes the copy at ~1 GB/s on first touch, a pinned one at PCIe speed).  Not tied to a context. */
int elfihip_host_alloc(size_t bytes, void** out);
int elfihip_host_free(void* p);
int elfihip_device_info(elfihip_ctx* ctx, int* cu_count, int* clock_khz, int* mem_clock_khz,
                        int* mem_bus_bits, int64_t* total_mem, char* name, int name_len);

/* Kernel timing with HIP events on the context's stream (bench.py's roofline leg). */
int elfihip_timer_start(elfihip_ctx* ctx);
int elfihip_timer_stop(elfihip_ctx* ctx, float* elapsed_ms); /* synchronises on the stop event */

/* ----------------------------------------------------------------- distance
 * Replaces scipy.spatial.distance.cdist(X (n,m), Y (1,m), metric, p=, w=, V=, VI=)
 * as called from distance_as_discrepancy (elfi/model/utils.py:37-52) through
 * elfi.Distance (elfi/model/elfi_model.py:1037) and AdaptiveDistance (:1084).
 * out has n doubles -- the (n,1)->(n,) squeeze of utils.py:50-51 is built in.
 *
 * aux: w (m) for EUCLIDEAN/SQEUCLIDEAN/CITYBLOCK/MINKOWSKI/CHEBYSHEV (NULL = unweighted),
 *      V (m) for SEUCLIDEAN, VI (m*m, row-major) for MAHALANOBIS.  p only for MINKOWSKI.
 */

/* X row-major (n, m) with leading dimension ldx >= m (doubles). */
int elfihip_dist_rows(elfihip_ctx* ctx, int metric, const double* X, int64_t n, int m, int64_t ldx,
                      const double* y, const double* aux, double p, double* out);
int elfihip_dist_rows_dev(elfihip_ctx* ctx, int metric, const double* dX, int64_t n, int m,
                          int64_t ldx, const double* dy, const double* daux, double p,
                          double* dout);

/* How the row-major distance calls stream their matrix (no reference counterpart; results are bit-identical either way):
 * 0 (default) rows of 16 / 32 / 64 summaries with 16-byte aligned rows arrive in LDS by LDS-DMA (global_load_lds, rings
 * of two 16 KiB slots per wave, non-temporal) -- every other shape, and 1 always, takes the register-staged pipeline;
 * 2 = as 0, and the fused adaptive-distance pass (elfihip_adaptive_push) streams the same way (measured slower than its
 * register-staged form: kept for measurement). */
int elfihip_dist_set_form(elfihip_ctx* ctx, int form);
/* m separately stored summary columns of length n each (structure of arrays): the
 * form ELFI hands to distance_as_discrepancy BEFORE np.column_stack
 * (elfi/model/utils.py:39) -- lets the caller skip that host-side copy. */
int elfihip_dist_cols(elfihip_ctx* ctx, int metric, const double* const* cols, int m, int64_t n,
                      const double* y, const double* aux, double p, double* out);
/* Column-major device matrix: column j starts at dC + j*ldc, ldc >= n. */
int elfihip_dist_cols_dev(elfihip_ctx* ctx, int metric, const double* dC, int64_t n, int m,
                          int64_t ldc, const double* dy, const double* daux, double p,
                          double* dout);

/* AdaptiveDistance.nested_distance (elfi/model/elfi_model.py:1135-1151): K weighted
 * euclidean distances of the same rows in one pass.  W is (K, m) row-major holding the
 * cdist weights (i.e. (1/scale)^2, elfi_model.py:1132); the unweighted first function
 * (w=None, :1089) is passed as a row of ones, which is bit-identical.  out is (n, K)
 * row-major, as np.column_stack of the K cdist results. */
int elfihip_dist_multiw(elfihip_ctx* ctx, const double* X, int64_t n, int m, int64_t ldx,
                        const double* y, const double* W, int K, double* out);
int elfihip_dist_multiw_dev(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx,
                            const double* dy, const double* dW, int K, double* dout);

/* AdaptiveDistance.add_data (elfi/model/elfi_model.py:1104-1125): one batched
 * Welford/Chan update of the running column statistics with the rows of X (n, m):
 *   N += n; d1 = x - mean; mean += sum(d1)/N; d2 = x - mean; M2 += sum(d1*d2).
 * Host form: count/mean/M2 are host scalars/vectors updated in place.
 * Device form: dstate is a device vector of 1 + 2m doubles [N, mean (m), M2 (m)]
 * updated in place on the context's stream (no host round trip). */
int elfihip_welford_update(elfihip_ctx* ctx, const double* X, int64_t n, int m, int64_t ldx,
                           int64_t* count, double* mean, double* M2);
int elfihip_welford_update_dev(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx,
                               double* dstate);
/* Multi-GPU adaptive distance: merge the `world` rank states (each 1 + 2m doubles: count, mean, M2 -- what
 * elfihip_welford_update_dev maintains) that an all-gather left in device memory, in rank order, with Chan's pairwise
 * formula -- the merge the reference gets for free from seeing all batches in one process (elfi_model.py:1104-1125).
 * dmerged (1 + 2m) receives the global state, dw2 (m, may be NULL) the cdist weights 1 / scale^2 = N / M2
 * (elfi_model.py:1124,1129-1132).  Asynchronous on the context's stream. */
int elfihip_welford_merge_dev(elfihip_ctx* ctx, const double* dstates, int world, int m, double* dmerged, double* dw2);

/* ------------------------------------------------------------------ selection
 * The selection inside Rejection._merge_batch (elfi/methods/inference/samplers.py:209-237): the
 * reference argsorts n_samples + batch_size distances on the host per batch; keeping the k smallest
 * of the batch is equivalent and lets only k rows leave the GPU.  vals (k) / idx (k, row numbers)
 * come back ascending by (distance, row); NaN last; k is clipped to n.  `stride` (in doubles) lets
 * the _dev form read one column of an (n, K) nested-distance matrix in place; the _dev form leaves
 * the k survivors unsorted (row order inside "< k-th" then "== k-th"). */
/* Which implementation serves the selections of this context: 0 (default) the resident single-launch radix select
 * (slices of 16 384 keys held in registers for n <= 2 x 10^6 on MI355X, re-read from memory per phase above that),
 * falling back to the nine-launch form when its grid barrier times out; 1 the nine-launch form always; 2 the resident
 * form without the register variant (measurement and tests). */
int elfihip_topk_set_form(elfihip_ctx* ctx, int form);
int elfihip_topk_smallest(elfihip_ctx* ctx, const double* D, int64_t n, int64_t stride, int64_t k, double* vals,
                          int64_t* idx);
int elfihip_topk_smallest_dev(elfihip_ctx* ctx, const double* dD, int64_t n, int64_t stride, int64_t k,
                              double* dvals, int64_t* didx);

/* ------------------------------------------------------------------- sampler state: running best-k, fused with the distance
 * Replaces Rejection._merge_batch (elfi/methods/inference/samplers.py:209-237: copy the batch behind the n_samples best
 * rows, argsort n_samples + batch_size distances on the host, every batch).  An elfihip_reject keeps the k smallest
 * distances seen so far and their GLOBAL row numbers (row_base + row inside the batch; ties go to the earlier row) in
 * device memory, ascending.  A push computes a batch's distances (written to dout / out, as the Distance node must
 * return them) and folds the batch in during the same pass: once the state is full its k-th distance is a threshold and
 * the distance kernel itself lists the rows below it; a one-workgroup merge keeps the k best.
 * k <= 2048: the state lives on the device, pushes are asynchronous.  2048 < k <= 2^20 ("host-merge" states): the few
 * candidates of every push are merged into a sorted host copy, the k-th distance goes back as the device threshold
 * (a push synchronises).
 * _dev forms are asynchronous (distance pass on the context's stream); elfihip_reject_result synchronises and copies the state out
 * (vals, rows: k entries; *count = how many are real).  A state never fails for valid input: the candidate list holds
 * 8 x the largest batch pushed and is merged every 8th push at the latest; a push that would offer very many
 * candidates (a small first batch, a new SMC round without reset) takes the radix selection of its k best instead.
 * Nested distances (n, K) are ranked by their last column (samplers.py:233). */
typedef struct elfihip_reject elfihip_reject;
int elfihip_reject_create(elfihip_ctx* ctx, int64_t k, elfihip_reject** out);
int elfihip_reject_free(elfihip_reject* h);
int elfihip_reject_reset(elfihip_reject* h);
/* out may be NULL: the batch's distances are not copied back (only the k best rows ever leave the GPU) */
int elfihip_reject_push_rows(elfihip_reject* h, int metric, const double* X, int64_t n, int m, int64_t ldx,
                             const double* y, const double* aux, double p, double* out, int64_t row_base);
int elfihip_reject_push_rows_dev(elfihip_reject* h, int metric, const double* dX, int64_t n, int m, int64_t ldx,
                                 const double* dy, const double* daux, double p, double* dout, int64_t row_base);
int elfihip_reject_push_multiw_dev(elfihip_reject* h, const double* dX, int64_t n, int m, int64_t ldx, const double* dy,
                                   const double* dW, int K, double* dout, int64_t row_base);
/* distances that exist already (any Distance / Discrepancy node): dD[i * stride], i < n */
int elfihip_reject_push_dev(elfihip_reject* h, const double* dD, int64_t n, int64_t stride, int64_t row_base);
/* ... in host memory: D (n, ncols) row-major, ncols >= 1 nested columns, ranked by the last one -- what
 * Rejection._merge_batch receives as batch[discrepancy_name] (samplers.py:209-237).  Synchronises. */
int elfihip_reject_push(elfihip_reject* h, const double* D, int64_t n, int ncols, int64_t row_base);
/* The acceptance condition of a threshold objective (samplers.py:219-225: Rejection.sample(threshold=...)): a row takes
 * part only if EVERY one of its nested columns is <= threshold (pushes of (n, ncols) distances: elfihip_reject_push,
 * _push_multiw_dev; single-column pushes test their one column).  Set before the first push or after a reset;
 * enable = 0 removes it.  Accepted rows are counted on the device (elfihip_reject_meta). */
int elfihip_reject_set_accept(elfihip_reject* h, int enable, double threshold);
/* The same condition with ONE THRESHOLD PER NESTED COLUMN: what AdaptiveDistanceSMC hands its Rejection as `threshold`
 * -- the list [inf, threshold of population 1, ...] (samplers.py:657-660), compared column by column by
 * `batch[d] <= threshold` (samplers.py:222-223).  thresholds: ncols values, +inf allowed; every later push must carry
 * ncols nested columns. */
int elfihip_reject_set_accept_cols(elfihip_reject* h, int ncols, const double* thresholds);
/* What Rejection._update_state_meta / _update_objective_n_batches read after every batch (samplers.py:239-271): the
 * current k-th distance (+inf while fewer than k rows are in; it is the sampler's `threshold` state), rows accepted by
 * the pushes since the previous call and in total (acceptance threshold set).  in_use: entries of a host-merge state
 * (-1 for device states: elfihip_reject_result reports it).  Merges what is pending and synchronises; any output may be NULL. */
int elfihip_reject_meta(elfihip_reject* h, double* kth, int64_t* in_use, int64_t* accepted_last, int64_t* accepted_total);
/* Distances that are still on the device.  Every HOST-form call that returns distances (elfihip_dist_rows, _dist_cols,
 * _dist_multiw, _adaptive_push, _gauss_distance, _ma2_distance, _ma2_draw_distance) leaves a device copy of what it
 * returned in the context -- (n, ncols) row-major -- until the next such call; elfihip_kept_distances names it (epoch: a
 * counter of those calls).  elfihip_reject_push_kept folds that copy into the state as elfihip_reject_push would fold
 * the host array (acceptance, ranking by the last column), without the upload: what Rejection._merge_batch receives as
 * batch[discrepancy_name] (samplers.py:209-237) is the array the Distance node just returned.  Status ELFIHIP_ERR_STATE
 * when a later call has replaced the copy (the caller then pushes the host array). */
int elfihip_kept_distances(elfihip_ctx* ctx, uint64_t* epoch, int64_t* n, int* ncols);
int elfihip_reject_push_kept(elfihip_reject* h, uint64_t epoch, int64_t row_base);
/* device pointers to the state (k values ascending, k rows), e.g. as the send buffers of a gather */
int elfihip_reject_state_dev(elfihip_reject* h, double** dvals, int64_t** drows);
/* From the next merge on, every merge also leaves the state as one packed device buffer at ddst -- k doubles (values)
 * followed by k int64 (rows): the send buffer of a gather -- written by the merges themselves (no extra copy); NULL
 * stops it. */
int elfihip_reject_export_dev(elfihip_reject* h, void* ddst);
/* Candidates are merged into the state after every push at first and, once the threshold has settled, after every 8th
 * (a one-workgroup merge after every distance pass would cost a third of the pass); in between the list keeps growing against the last merged threshold, which still bounds the
 * current k-th distance from above, so nothing is missed.  elfihip_reject_flush merges what is pending now
 * (asynchronous); _result, _state_dev do so themselves. */
int elfihip_reject_flush(elfihip_reject* h);
/* vals / rows: k entries each; *count = entries in use.  A NaN distance never enters the state (the reference's argsort
 * would list such rows last, after every finite distance): while fewer than k finite distances have been seen, count is
 * their number. */
int elfihip_reject_result(elfihip_reject* h, double* vals, int64_t* rows, int64_t* count);

/* ------------------------------------------------------------------ one adaptive-distance batch in one read
 * Everything the reference does with a batch of an AdaptiveDistance round while its rows are read ONCE:
 *   - AdaptiveDistance.nested_distance (elfi/model/elfi_model.py:1135-1151): the K weighted euclidean distances under
 *     the weights of the EARLIER rounds (W (K, m): cdist weights, a row of ones for the unweighted first function) ->
 *     dout / out (n, K), bit-identical to elfihip_dist_multiw; dout / out may be NULL (nothing but the state's k rows
 *     leaves the GPU);
 *   - AdaptiveDistance.add_data (elfi_model.py:1104-1125), which Rejection._merge_batch calls for every batch
 *     (elfi/methods/inference/samplers.py:213-216): the batch folded into the running column statistics [N, mean (m),
 *     M2 (m)] that the NEXT update_distance turns into weights -- dwelford (device, 1 + 2m doubles, in place) or
 *     count / mean / M2 (host, in place); NULL: no statistics.  Tile-local two-pass sums merged with Chan's pairwise
 *     update in a fixed order: equal to the reference's batched Welford update up to rounding, bit-reproducible;
 *   - Rejection._merge_batch (samplers.py:218-237) against `state` (may be NULL): acceptance of every nested column
 *     (elfihip_reject_set_accept / _set_accept_cols), ranking by the last one, rows numbered row_base + row.  A first
 *     batch of >= 2^20 rows is selected against a provisional threshold taken from a prefix of the batch and verified
 *     (see csrc/reject.hip); the result is exact in every case.
 * Rows narrower than 129 doubles with even m / ldx and 16-byte aligned dX take the fused kernel (csrc/adaptive.hip);
 * other shapes run the separate passes (elfihip_welford_update_dev, elfihip_dist_multiw_dev, candidate pass).
 * _dev: asynchronous on the context's stream, except for the first-batch check, which reads four bytes back. */
int elfihip_adaptive_push_dev(elfihip_ctx* ctx, elfihip_reject* state, const double* dX, int64_t n, int m, int64_t ldx,
                              const double* dy, const double* dW, int K, double* dout, double* dwelford,
                              int64_t row_base);
int elfihip_adaptive_push(elfihip_ctx* ctx, elfihip_reject* state, const double* X, int64_t n, int m, int64_t ldx,
                          const double* y, const double* W, int K, double* out, int64_t* count, double* mean,
                          double* M2, int64_t row_base);

/* ... on rows that are still on the device: elfihip_randn_rows (below) leaves a device copy of the rows it returned --
 * elfihip_kept_rows names it -- and elfihip_adaptive_push_kept runs the pass on that copy (no upload of the batch);
 * ELFIHIP_ERR_STATE when a later elfihip_randn_rows call has replaced it. */
int elfihip_kept_rows(elfihip_ctx* ctx, uint64_t* epoch, int64_t* n, int* m);
int elfihip_adaptive_push_kept(elfihip_ctx* ctx, elfihip_reject* state, uint64_t rows_epoch, const double* y,
                               const double* W, int K, double* out, int64_t* count, double* mean, double* M2,
                               int64_t row_base);

/* ------------------------------------------------------------------ SMC proposal density
 * GMDistribution.pdf (elfi/methods/utils.py:142-183): density of a Gaussian mixture with shared
 * covariance at M points, out[r] = sum_i weights[i] * N(x_r; means[i], cov).  The caller passes the
 * covariance in the factored form SciPy's multivariate_normal uses: U (d x d, row-major) with
 * cov^+ = U U^T, and log_norm = rank*log(2 pi) + log pdet(cov); weights already normalised.  d <= 64 (above 16 both point sets are
 * transformed by U once and a pair costs d subtractions). */
int elfihip_gm_pdf(elfihip_ctx* ctx, const double* x, int64_t M, int d, const double* means, int64_t N,
                   const double* weights, const double* U, double log_norm, double* out);
/* n draws from the same mixture -- GMDistribution.rvs (elfi/methods/utils.py:199-262), the proposal of an SMC batch
 * (samplers.py:434-459): draw i picks the component at the inverse of the cumulative weights `cumw` (N values, last 1) at
 * the uniform of Philox counter i of stream (seed, stream) and adds A z_i, A (d, d) row-major with A A^T = cov, z_i the d
 * standard normals of counters i ceil(d / 2) ... of stream (seed, stream + 1) (the generator of elfihip_randn_dev).  The
 * validity check against the prior (prior_logpdf) stays with the caller.  Host pointers; out (n, d). */
int elfihip_gm_rvs(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, int d, const double* means, int64_t N,
                   const double* cumw, const double* A, double* out);

/* ------------------------------------------------------------------ SMC population statistics
 * weighted_var (elfi/methods/utils.py:108-139; caller samplers.py:521-534): unbiased weighted variance of every
 * column of the accepted sample, s2[c] = sum_i w_i (x_ic - xbar_c)^2 / (V1 - V2 / V1), xbar_c = sum_i w_i x_ic / V1,
 * V1 = sum w, V2 = sum w^2.  X (n, m) row-major with pitch ldx; w (n) or NULL for unit weights (the reference's
 * default); s2 (m).  Deterministic (fixed summation order). */
int elfihip_weighted_var(elfihip_ctx* ctx, const double* X, int64_t n, int m, int64_t ldx, const double* w, double* s2);
int elfihip_weighted_var_dev(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx, const double* dw,
                             double* ds2);

/* ------------------------------------------------------------------ summaries
 * Row-wise summary statistics that ELFI's example models install as elfi.Summary operations, with
 * NumPy's exact (pairwise) summation order, i.e. bit-identical results:
 *   ELFIHIP_SUM_MEAN     np.mean(y, axis=1)                           elfi/examples/gauss.py:142-156
 *   ELFIHIP_SUM_VAR      np.var(y, axis=1)                            elfi/examples/gauss.py:159-173
 *   ELFIHIP_SUM_AUTOCOV  np.mean(x[:, lag:] * x[:, :-lag], axis=1)    elfi/examples/ma2.py:40-59
 * X is row-major (n, L) with leading dimension ldx; out has n doubles; lag only for AUTOCOV. */
enum { ELFIHIP_SUM_MEAN = 0, ELFIHIP_SUM_VAR = 1, ELFIHIP_SUM_AUTOCOV = 2 };
int elfihip_row_summary(elfihip_ctx* ctx, int kind, const double* X, int64_t n, int L, int64_t ldx, int lag,
                        double* out);
int elfihip_row_summary_dev(elfihip_ctx* ctx, int kind, const double* dX, int64_t n, int L, int64_t ldx, int lag,
                            double* dout);
/* The whole MA2 example path after the random draw, fused (elfi/examples/ma2.py:11-59,62-92 +
 * elfi.Distance('euclidean', S1, S2)): from the white noise W (n, n_obs + 2) -- drawn by the caller
 * from the reference's MT19937 stream, random_state.randn(batch_size, n_obs + 2) -- and per-row
 * parameters t1, t2 (n): x = w[:, 2:] + t1 w[:, 1:-1] + t2 w[:, :-2], S1 = autocov(x, 1),
 * S2 = autocov(x, 2), D = sqrt((S1 - obs1)^2 + (S2 - obs2)^2), bit-identical to the reference's
 * NumPy/SciPy results, in one pass over W. */
int elfihip_ma2_distance(elfihip_ctx* ctx, const double* W, int64_t n, int n_obs, const double* t1, const double* t2,
                         double obs1, double obs2, double* S1, double* S2, double* D);
/* ... with the white noise drawn inside the kernel (Philox4x32-10, key `seed`, counter stream `stream`: the draws of
 * elfihip_randn_dev), host parameters in, host results out: the whole example as ONE node operation. */
int elfihip_ma2_draw_distance(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, int n_obs, const double* t1,
                              const double* t2, double obs1, double obs2, double* S1, double* S2, double* D);
int elfihip_ma2_distance_dev(elfihip_ctx* ctx, const double* dW, int64_t n, int n_obs, int64_t ldw, const double* dt1,
                             const double* dt2, double obs1, double obs2, double* dS1, double* dS2, double* dD);
/* The same with the white noise DRAWN IN THE KERNEL (synthetic-throughput runs; the reference draws it from MT19937 on the
 * host, examples/ma2.py:31-33: bit parity with a reference run needs that stream, so parity runs hand W in): element e
 * of the row-major (n, n_obs + 2) noise matrix is normal number e of the stream (seed, stream) of elfihip_randn_dev, so
 * the results equal elfihip_randn_dev + elfihip_ma2_distance_dev bit for bit while the noise never exists in memory.
 * n_obs <= 126. */
int elfihip_ma2_draw_distance_dev(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, int n_obs, const double* dt1,
                                  const double* dt2, double obs1, double obs2, double* dS1, double* dS2, double* dD);

/* The Gaussian example model, fused (elfi/examples/gauss.py:11-35 gauss, :142-173 ss_mean / ss_var, :133
 * elfi.Distance('euclidean', ss_mean, ss_var)): y = z * sigma + mu for n simulations of n_obs observations (what
 * ss.norm.rvs(loc, scale) computes from standard normals z), S1 = np.mean(y, axis=1), S2 = np.var(y, axis=1),
 * D = sqrt((S1 - obs_mean)^2 + (S2 - obs_var)^2) -- bit-identical to NumPy / SciPy on the same z -- in one pass.
 * Z (n, n_obs) holds the standard normals (drawn by the caller from the reference's MT19937 stream when results must
 * match the reference); Z == NULL draws them on the device: Philox4x32-10, counter = (pair index, stream), key = seed,
 * Box-Muller, element e of the row-major (n, n_obs) matrix from pair e / 2 -- a pure function of (seed, stream, e).
 * Y (optional, (n, n_obs)) receives the simulator output.  mu, sigma: n values each. */
int elfihip_gauss_distance(elfihip_ctx* ctx, const double* Z, uint64_t seed, uint64_t stream, int64_t n, int n_obs,
                           const double* mu, const double* sigma, double obs_mean, double obs_var, double* Y, double* S1,
                           double* S2, double* D);
int elfihip_gauss_distance_dev(elfihip_ctx* ctx, const double* dZ, int64_t ldz, uint64_t seed, uint64_t stream, int64_t n,
                               int n_obs, const double* dmu, const double* dsigma, double obs_mean, double obs_var,
                               double* dY, double* dS1, double* dS2, double* dD);
/* Device-resident synthetic inputs from the same generator: dout[e] = z_e * scale + loc, e < n (the simulator
 * outputs of BASELINE configs[1] / configs[3]); the raw Philox4x32-10 blocks (4 x uint32 per block, counter =
 * (block, stream), key = seed) for known-answer tests. */
int elfihip_randn_dev(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, double loc, double scale, double* dout);
/* The synthetic Gaussian simulator of BASELINE configs[1] / [3] as ONE host-form node operation: out (n, m) row-major,
 * out[i][j] = loc[i] + scale[j] z, z = the draws of elfihip_randn_dev(seed, stream) in row-major order (m even).  The rows
 * also stay on the device for the distance call that follows (elfihip_kept_rows, elfihip_adaptive_push_kept). */
int elfihip_randn_rows(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t n, int m, const double* loc,
                       const double* scale, double* out);
int elfihip_random_bits_dev(elfihip_ctx* ctx, uint64_t seed, uint64_t stream, int64_t nblocks, uint32_t* dout);
/* Prior draws on the device.  Replaces the draws behind elfi.Prior nodes -- rvs_from_distribution
 * (elfi/model/utils.py:6-35) -> scipy.stats rvs on the host, 64 % of the reference loop's time per batch of 10^6 once
 * simulator and distance run on the GPU.  U_e = the 53-bit uniform in [0, 1) made of the first
 * (e even) / second (e odd) pair of words of Philox counter e / 2 of stream (seed, stream) -- the generator of
 * elfihip_randn_dev -- transformed in the reference's own operation order:
 *   ELFIHIP_PRIOR_UNIFORM: out = U * a[1] + a[0]                  ss.uniform.rvs(loc = a[0], scale = a[1])
 *   ELFIHIP_PRIOR_MA2_T1:  CustomPrior1.rvs(b = a[0])             elfi/examples/ma2.py:102-118
 *   ELFIHIP_PRIOR_MA2_T2:  CustomPrior2.rvs(t1 = cond, a = a[0])  elfi/examples/ma2.py:147-166 (cond: n values)
 * a: host scalars.  Host form: cond / out host pointers; _dev form: device pointers, no synchronisation. */
#define ELFIHIP_PRIOR_UNIFORM 0
#define ELFIHIP_PRIOR_MA2_T1 1
#define ELFIHIP_PRIOR_MA2_T2 2
int elfihip_prior_draw(elfihip_ctx* ctx, int kind, uint64_t seed, uint64_t stream, int64_t n, const double* a,
                       const double* cond, double* out);
int elfihip_prior_draw_dev(elfihip_ctx* ctx, int kind, uint64_t seed, uint64_t stream, int64_t n, const double* a,
                           const double* dcond, double* dout);

/* ------------------------------------------------------------------- GP surrogate
 * Replaces the GPy model behind elfi.methods.bo.gpy_regression.GPyRegression
 * (elfi/methods/bo/gpy_regression.py:15-364): kernel RBF(variance, lengthscale) + Bias
 * (:260-280), Gaussian noise, exact inference.  One elfihip_gp holds the evidence
 * (X (n,d), y (n)) and, after elfihip_gp_factorize, L = chol(K + (noise + 1e-8) I),
 * L^-T and alpha = K^-1 y in device memory.  Host pointers in, host pointers out.
 */
/* capacity: the largest n this GP will hold (rounded up to a multiple of 128). */
int elfihip_gp_create(elfihip_ctx* ctx, int d, int64_t capacity, elfihip_gp** gp);
int elfihip_gp_free(elfihip_gp* gp);
/* kern.rbf.variance, kern.rbf.lengthscale, kern.bias.variance, Gaussian_noise.variance
 * (gpy_regression.py:151-155). */
int elfihip_gp_set_hyper(elfihip_gp* gp, double rbf_variance, double lengthscale, double bias_variance,
                         double noise_variance);
/* Replace / extend the evidence (GPyRegression.update, gpy_regression.py:286-315: np.r_[X_old, x]). */
int elfihip_gp_set_data(elfihip_gp* gp, const double* X, const double* y, int64_t n);
int elfihip_gp_append(elfihip_gp* gp, const double* X_new, const double* y_new, int64_t k);
/* Gram matrix + Cholesky + L^-T + alpha (what GPy's ExactGaussianInference does when
 * gpy_regression.py:283-284 / :311-312 construct GPRegression).  log_marginal may be NULL.
 * A failed plain Cholesky is retried the way GPy's `jitchol` does it ([GPy-upstream] GPy/util/linalg.py, reached
 * through ExactGaussianInference / pdinv from GPyRegression.update and .optimize, elfi/methods/bo/gpy_regression.py:286-323):
 * jitter = mean(diag Ky) * 1e-6 on the diagonal, ten times more per failed try, at most `maxtries` (5) tries; L, alpha,
 * log Z and L^-T then belong to the jittered matrix, as GPy's posterior object does.  ELFIHIP_ERR_NOT_PD only when the
 * last try fails too ("not positive definite, even with jitter": LinAlgError in the reference, which
 * GPyRegression.optimize catches and warns about, gpy_regression.py:320-323). */
int elfihip_gp_factorize(elfihip_gp* gp, double* log_marginal);
/* The ladder above: maxtries >= 0 sets the retries allowed (GPy: 5; 0 = fail at the first non-positive pivot),
 * maxtries < 0 leaves it.  jitter receives what the CURRENT factor carries on its diagonal besides noise + 1e-8 (0 when
 * the plain Cholesky went through), tries the retries the latest elfihip_gp_factorize made.  Pointers may be NULL.
 * elfihip_gp_extend refuses to border a factor that carries jitter (it refactors instead, as GPy would). */
int elfihip_gp_jitchol(elfihip_gp* gp, int maxtries, double* jitter, int* tries);
/* How elfihip_gp_factorize schedules its sweep over the 128-wide block columns (no reference counterpart: GPy hands
 * the factorisation to LAPACK).  schedule 0 = chosen by size (default), 1 = two-stream look-ahead with panel groups,
 * 2 = fused steps on the caller's stream (panel solve, diagonal tile, then ONE launch with the next diagonal block
 * beside the trailing update: three launches per block column), 3 = the same chained by arrival counters inside ONE
 * launch per block column (kept for measurement: the in-launch hand-offs cost more than the two launches they replace;
 * needs all of the launch's workgroups resident, i.e. the device to itself), 4 = fused steps with the panel solve and the
 * diagonal tile in ONE launch (two launches per block column; bit-identical to 2, and no faster: kept for measurement),
 * 5 = the three roles of a step as three CONCURRENT launches on three streams -- a persistent update launch, a
 * persistent one-workgroup launch for the diagonal blocks, one chain launch (panel solve + diagonal tile) per block
 * column -- ordered by counters in device memory (bit-identical to 2; measured slower, the update of step k and the
 * panel solve of step k+1 depend on each other in full and cannot overlap: kept for measurement; needs the device to
 * itself like 3, and reports a hand-off that does not arrive within about a third of a second as ELFIHIP_ERR_HIP);
 * panel_group 0 = by size, else 1 / 2 / 4 panels per pass over the trailing matrix (schedule 1).  Results agree to
 * rounding between schedules; each is deterministic. */
int elfihip_gp_set_schedule(elfihip_gp* gp, int schedule, int panel_group);
/* Which form the two triangular products of a prediction call take (no reference counterpart: the reference predicts
 * one point per call, gpy_regression.py:98-147,179-223; bo/utils.py:97-103 runs its starts one after the other).  Calls
 * with fewer than min_points query points stream the factor once per <= 128 points through (row block, k chunk)
 * workgroups -- HBM-bound, right for the 10 starts of the default LCBSC; calls with at least min_points points (default
 * 112: the 256 parallel starts of BASELINE configs[4], many-chain sampling) run both products as dense 64 x 64 MFMA
 * tiles with full-k accumulation -- matrix-pipe-bound (csrc/gp_dense.hip).  min_points <= 0 restores the default;
 * a huge value keeps every call on the streaming form.  tile_rows: rows of the factor per output tile of the dense form,
 * 0 = by size (the tallest of 64 / 32 / 16 that still gives about one workgroup per CU), else 64, 32 or 16.  Results agree
 * to rounding; each form is deterministic. */
int elfihip_gp_set_dense_threshold(elfihip_gp* gp, int64_t min_points, int tile_rows);
/* How a prediction call with few points (every step of the multi-start search, bo/utils.py:97-103) is launched:
 * form 0 (default) = four launches -- kernel rows | first triangular product | second | assembly -- the reduction of the
 * first product's chunk partials and the gradient sums of the second run as epilogues of the LAST workgroup to arrive at
 * each 32-row block (write-through partials, one relaxed device-scope arrival per workgroup, one acquire by the last
 * arriver); form 1 = the six launches of round 2 (product, reduction, product, gradient sums as kernels of their own).
 * Mean / variance are bit-identical between the forms, gradients agree to rounding (32- against 64-row chunks).
 * Acquisition lock-steps (elfihip_gp_lcb with a gradient, elfihip_gp_lcb_minimize) additionally have a THREE-launch form:
 * u = K^-1 kb from ONE product with the symmetric K^-1 instead of the two dependent triangular products (GPy's own
 * closed form with woodbury_inv, gpy_regression.py:127-140).  Under form 0 it takes over once a factorisation has served
 * 64 lock-steps -- or at once when the caller has formed K^-1 for this factorisation (elfihip_gp_form_kinv) -- (forming
 * K^-1 costs what 40-80 lock-steps save; afterwards elfihip_gp_extend borders the matrix, rank one, with every new
 * point) and while (max L_ii / min L_ii)^2 <= 1e5 (the variance k(x,x) - kb . u then agrees with
 * the triangular form to 1e-10 k(x,x)); form 2 = never (fused triangular products only); form 3 = from the first
 * lock-step on.  elfihip_gp_predict / _predict_grad always use the triangular products. */
int elfihip_gp_set_lockstep_form(elfihip_gp* gp, int form);
/* State of the above for the current factorisation: whether acquisition lock-steps multiply with K^-1 now, how many
 * lock-steps the factorisation has served, and the ESTIMATE of cond(K) that keeps hopeless matrices from being formed:
 * min(n (max L_ii / min L_ii)^2, n (var + bias + s) / s) with s = noise + 1e-8 + jitter (only the second is a true upper
 * bound; the first can lie below cond(K)).  What makes the K^-1 form fail safe is not this number: the first lock-step with
 * a newly formed K^-1 is also run through the triangular products and the two variances must agree to 1e-9 k(x,x), or the
 * factorisation keeps the triangular form.  Any pointer may be NULL. */
int elfihip_gp_lockstep_info(const elfihip_gp* gp, int* kinv_in_use, int64_t* steps, double* cond_estimate);
/* Device time per phase, for roofline accounting (bench.py; no reference counterpart).  While enabled, HIP events on the
 * GP's stream bracket the phases of elfihip_gp_factorize (Gram matrix | sweep | alpha + log-determinant), of single-group
 * prediction calls -- elfihip_gp_predict / _predict_grad / _lcb and every step of elfihip_gp_lcb_minimize -- (kernel row |
 * first triangular product + its reduction | second triangular product | gradient sums + final assembly) and of
 * elfihip_gp_nlml_grad (K^-1 tiles with the fused gradient contractions).  enable > 0 starts a fresh measurement,
 * 0 stops it, < 0 only reads; phase_ms / phase_calls (ELFIHIP_PHASE_COUNT entries each, may be NULL) receive the sums of
 * milliseconds and the number of timed calls per phase collected so far. */
enum {
  ELFIHIP_PHASE_GRAM = 0,
  ELFIHIP_PHASE_SWEEP = 1,
  ELFIHIP_PHASE_ALPHA = 2,
  ELFIHIP_PHASE_KSTAR = 3,
  ELFIHIP_PHASE_TRI_FIRST = 4,
  ELFIHIP_PHASE_TRI_SECOND = 5,
  ELFIHIP_PHASE_GRAD_FINISH = 6,
  ELFIHIP_PHASE_KINV_GRAD = 7,
  ELFIHIP_PHASE_COUNT = 8
};
int elfihip_gp_profile(elfihip_gp* gp, int enable, double* phase_ms, int64_t* phase_calls);
/* What GPy evaluates once per objective call of GPyRegression.optimize() (gpy_regression.py:317-323
 * -> [GPy-upstream] ExactGaussianInference + kern.update_gradients_full): the log marginal
 * likelihood and its gradient w.r.t. (rbf.variance, rbf.lengthscale, bias.variance,
 * Gaussian_noise.variance), grad[4], from dL/dK = 0.5 (alpha alpha^T - K^-1).  Priors and the
 * positivity transform are host-side scalars (elfi_amd/hyperopt.py).  Needs a factorised GP. */
int elfihip_gp_nlml_grad(elfihip_gp* gp, double* log_marginal, double* grad);
/* Materialise K^-1 (GPy's posterior.woodbury_inv, read by gpy_regression.py:158) for
 * elfihip_gp_get(gp, 5, ...).  Not needed by predict / gradients / nlml_grad. */
int elfihip_gp_form_kinv(elfihip_gp* gp);
/* GPyRegression.update with unchanged hyper-parameters (what elfi.BOLFI does for every batch between
 * two hyper-parameter optimisations, bolfi.py:219 -> gpy_regression.py:304-312): append k evidence
 * points and bring L, L^-T, alpha and the log marginal up to date by bordering -- one new row /
 * column per point, two passes over L^-T (O(n^2)) instead of the O(n^3) rebuild the reference
 * performs.  Same result as elfihip_gp_append + elfihip_gp_factorize to rounding.  Falls back to
 * exactly that when the GP is not factorised yet or the padded size (multiple of 128) grows. */
int elfihip_gp_extend(elfihip_gp* gp, const double* X_new, const double* y_new, int64_t k, double* log_marginal);
int elfihip_gp_size(const elfihip_gp* gp, int64_t* n, int64_t* capacity, int* d);
/* Copy state to the host: 0 = L (n*n lower), 1 = L^-T (n*n upper), 2 = alpha (n),
 * 3 = X (n*d), 4 = y (n), 5 = K^-1 (n*n, after elfihip_gp_form_kinv). */
int elfihip_gp_get(elfihip_gp* gp, int which, double* out);

/* GPyRegression.predict(x, noiseless) (gpy_regression.py:98-147; closed form :127-140):
 * mu_s = k_s^T alpha,  var_s = clip(s_f + s_b - k_s^T K^-1 k_s, 1e-15) (+ noise unless noiseless).
 * Xs is (S, d) row-major; mu, var get S doubles each. */
int elfihip_gp_predict(elfihip_gp* gp, const double* Xs, int64_t S, int noiseless, double* mu, double* var);
/* GPyRegression.predictive_gradients(x) (gpy_regression.py:179-223; closed form :206-218):
 * dmu (S,d), dvar (S,d).  mu/var (noiseless) are returned too when non-NULL. */
int elfihip_gp_predict_grad(elfihip_gp* gp, const double* Xs, int64_t S, double* mu, double* var,
                            double* dmu, double* dvar);
/* LCBSC.evaluate / evaluate_gradient (elfi/methods/bo/acquisition.py:262-301) for S points at
 * once: val_s = mu - sqrt(beta var),  grad_s = dmu - 0.5 dvar sqrt(beta / var)  (noiseless). */
int elfihip_gp_lcb(elfihip_gp* gp, const double* Xs, int64_t S, double beta, double* val, double* grad);

/* The inner optimisation of AcquisitionBase.acquire (elfi/methods/bo/acquisition.py:146-163 ->
 * minimize(), elfi/methods/bo/utils.py:40-111) for the LCBSC rule: minimise
 * a(x) = mu(x) - sqrt(beta var(x)) inside the box [lower, upper] from S start points (S, d).
 * The reference runs scipy L-BFGS-B from each start in turn, one GP prediction per evaluation;
 * here all starts advance in lock-step with ONE batched device evaluation per step; each start
 * is an L-BFGS-B state machine with scipy's defaults (csrc/lbfgsb.hpp, csrc/gp_acq.hip).
 * x_out (S,d) and f_out (S) receive every start's end point / value (the caller takes the
 * arg-min, utils.py:105-109); iters_out (S) and n_eval_out (total point evaluations) may be NULL. */
int elfihip_gp_lcb_minimize(elfihip_gp* gp, const double* starts, int64_t S, const double* lower,
                            const double* upper, double beta, int maxiter, double* x_out, double* f_out,
                            int* iters_out, int64_t* n_eval_out);
/* Options of that search (no environment switches): host_threads of the quasi-Newton algebra for searches of >= 64 starts
 * (0: min(16, hardware threads, CPUs the cgroup grants)); trace: 1 = one line per search on stderr (rounds, evaluations,
 * device / host milliseconds), 2 = + active points and device time of every round, 0 = nothing. */
int elfihip_gp_set_acq_options(elfihip_gp* gp, int host_threads, int trace);

/* ExpIntVar (elfi/methods/bo/acquisition.py:629-821) needs the GP's posterior covariance between its M
 * integration points and each candidate: cov(p_i, q) = k(p_i, q) - k(p_i, X) K^-1 k(X, q), which the
 * reference forms per evaluate() call with cho_factor + cho_solve of the n x n matrix (:800-808).
 * set_integration_points stores V_P = L^-1 k(X, P) for P (M, d) once (valid until the evidence or the
 * hyper-parameters change); cross_cov then returns cov (M, S) row-major and, optionally, the noiseless
 * predictive variance var_q (S) of the S candidates Q (S, d): one streaming pass over V_P per 128 candidates. */
int elfihip_gp_set_integration_points(elfihip_gp* gp, const double* P, int64_t M);
int elfihip_gp_cross_cov(elfihip_gp* gp, const double* Q, int64_t S, double* cov, double* var_q);
/* The PRIOR covariance between two point sets under the GP's current hyper-parameters, out (na, nb) row-major:
 * k(a, b) = rbf.variance exp(-|a - b|^2 / 2 lengthscale^2) + bias.variance -- GPy's `kern.K(X, X2)`, which the reference's
 * ExpIntVar reaches for through `model._gp.kern.K` (elfi/methods/bo/acquisition.py:754,770).  B == NULL: k(A, A) with an
 * exact diagonal.  Host pointers; no factorisation needed. */
int elfihip_gp_kernel_matrix(elfihip_gp* gp, const double* A, int64_t na, const double* B, int64_t nb, double* out);
/* MaxVar / RandMaxVar surface (elfi/methods/bo/acquisition.py:392-463): the variance of the unnormalised approximate
 * posterior prior(x)^2 [Phi_skew(eps) - Phi(eps)^2] at S points and its gradient, from ONE batched prediction with the
 * skew-normal / Owen's-T epilogue on the device (the reference: three single-point GP predictions and SciPy's skewnorm
 * per value/gradient pair).  prior_pdf (S) and prior_grad_logpdf (S, d) are the prior's density and the gradient of its
 * logarithm at the points (host callables of the caller's prior object); val (S), grad (S, d). */
int elfihip_gp_maxvar(elfihip_gp* gp, const double* Xs, int64_t S, double eps, const double* prior_pdf,
                      const double* prior_grad_logpdf, double* val, double* grad);
/* ExpIntVar loss (acquisition.py:795-821) of S candidates against the integration points fixed by
 * elfihip_gp_set_integration_points: 2 sum_i w_i T(z_i, a_is) with the posterior covariances taken from the device
 * factorisation (the reference: cho_factor + cho_solve of the n x n matrix per call).  w_int = omega_i prior(p_i)^2,
 * mean_int / var_int = GP mean and noiseless variance at the integration points (M each); loss (S). */
int elfihip_gp_expintvar(elfihip_gp* gp, const double* Q, int64_t S, double eps, const double* w_int,
                         const double* mean_int, const double* var_int, double* loss);

/* ---- the same multi-start search for objectives the HOST assembles -----------------------
 * scipy.optimize.minimize(method='L-BFGS-B') as the reference calls it from minimize()
 * (elfi/methods/bo/utils.py:97-103: jac given, bounds, maxiter, SciPy's default tolerances), turned
 * inside out so that S searches share batched evaluations: the acquisition rules other than LCBSC
 * (MaxVar.acquire, acquisition.py:349-384) combine device predictions with host-side formulas, so their
 * objective cannot live inside the library.  Usage: create with the S start points; then repeat
 *   n = pending(idx, x)        the searches that wait for an evaluation and where (n == 0: all done)
 *   feed(n, f, g)              objective values (n) and gradients (n, d) at those points, same order
 * and read every search's end point.  Pure host code (no device, no context). */
typedef struct elfihip_lbfgsb elfihip_lbfgsb;
int elfihip_lbfgsb_create(int d, int64_t S, const double* lower, const double* upper, const double* starts,
                          int maxiter, elfihip_lbfgsb** out);
/* idx (S) / x (S, d) receive the waiting searches' indices and points; returns their number, < 0 on error. */
int64_t elfihip_lbfgsb_pending(elfihip_lbfgsb* h, int64_t* idx, double* x);
int elfihip_lbfgsb_feed(elfihip_lbfgsb* h, int64_t n, const double* f, const double* g);
/* x (S, d), f (S), iters (S), status (S: 1 projected gradient, 2 relative reduction, 3 maxiter, 4 abnormal, 5 maxfun = 15000 evaluations);
 * any of f / iters / status may be NULL. */
int elfihip_lbfgsb_result(const elfihip_lbfgsb* h, double* x, double* f, int* iters, int* status);
int elfihip_lbfgsb_free(elfihip_lbfgsb* h);

/* ------------------------------------------------------------------- multi-GPU exchange (one process per GPU)
 * Replaces the pickled returns of ELFI's batch farm (elfi/client.py:268-274; the per-batch farm-out of
 * elfi/methods/parameter_inference.py:283-292): the ranks' small per-round results travel device to device over RCCL
 * (xGMI), and a factorised GP can be handed from one rank to the others.  RCCL (librccl.so) is loaded at run time by the
 * first of these calls; everything runs on the context's stream and does not synchronise.  Bootstrap as with NCCL: rank 0
 * calls elfihip_comm_unique_id (128 bytes), hands them to the other ranks by any out-of-band means (the launcher's
 * rendezvous, a file, MPI), every rank calls elfihip_comm_init_rank with its context (= its GPU). */
typedef struct elfihip_comm elfihip_comm;
int elfihip_comm_unique_id(elfihip_ctx* ctx, void* id128);
int elfihip_comm_init_rank(elfihip_ctx* ctx, const void* id128, int rank, int world_size, elfihip_comm** out);
int elfihip_comm_free(elfihip_comm* comm);
/* drecv (world_size * count) in rank order, on every rank / on `root` only (NULL elsewhere). */
int elfihip_comm_allgather_f64(elfihip_comm* comm, const double* dsend, int64_t count, double* drecv);
int elfihip_comm_gather_f64(elfihip_comm* comm, const double* dsend, int64_t count, double* drecv, int root);
int elfihip_comm_bcast_f64(elfihip_comm* comm, double* dbuf, int64_t count, int root);
/* The root's factorised GP (evidence, hyper-parameters, L, L^-T, K^-1 y, log-marginal terms) to every rank's GP object of
 * the same input dimension and capacity -- instead of the same rebuild on every GPU (the kernels are deterministic, so
 * replicas are bit-identical either way); synchronises the stream (a header travels first). */
int elfihip_comm_bcast_factor(elfihip_comm* comm, elfihip_gp* gp, int root);

#ifdef __cplusplus
}
#endif
#endif /* ELFIHIP_H */
