// Internal definition of the GP object behind the elfihip_gp_* entry points.
#pragma once

#include "common.hpp"

#include <vector>

namespace elfihip {
constexpr int NB = 128;          // block size of the factorisation (one MFMA GEMM tile)
constexpr int FUSED_BELOW_NB = 97;  // block columns below which the fused-step schedule is the default (gp_fit.hip):
                                    // measured ahead of the stream schedule at every size up to n = 12288 (15.5 against
                                    // 16.6 ms at n = 10240, 26.0 against 26.3 ms at 12288)
constexpr double GP_JITTER = 1e-8;  // [GPy-upstream] ExactGaussianInference: Ky = K + (noise + 1e-8) I
inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
}  // namespace elfihip

struct elfihip_gp {
  elfihip_ctx* ctx = nullptr;
  int d = 0, dp = 0;          // input dimension, padded to a multiple of 4 (MFMA k-step)
  int64_t cap = 0;            // capacity in evidence points (multiple of NB)
  int64_t n = 0, np = 0;      // current evidence count, padded to a multiple of NB
  int64_t lda = 0;            // row pitch (doubles) of A and WT
  double var = 1, ls = 1, bias = 0, noise = 1;
  bool factored = false, has_kinv = false, wl_valid = false;
  bool kinv_sym = false;      // K^-1 complete (both triangles) and current: formed after a factorisation, bordered by extends
  // K^-1 lock-steps are VALIDATED once per K^-1 (formed after a full factorisation, then carried through extends by
  // bordering): the first lock-step with it also runs through the triangular products and the two variances must agree
  // (gp_predict.hip: predict_impl).  kinv_checked: this K^-1 passed; kinv_bad_full: the full factorisation (full_gen) whose
  // K^-1 failed -- not formed or used again until the next full factorisation
  bool kinv_checked = false;
  long long full_gen = 0, kinv_bad_full = -1;
  int64_t lcb_steps = 0;      // acquisition lock-steps since the latest factorisation (extends do not reset it)
  double diag_min = 0, diag_max = 0;   // smallest / largest diagonal entry of L: (max / min)^2 bounds cond(K) from below
  double logdet = 0, yKy = 0;
  // GPy's jitchol ladder (gp_fit.hip: gp_factorize_impl): tries allowed after a failed plain Cholesky (0 = fail at once),
  // the jitter the current factor carries on its diagonal (0: none was needed) and the retries the latest call made
  int jitchol_maxtries = 5;
  double jitter = 0.0;
  int jitter_tries = 0;
  int jit_start = 0;        // rung the NEXT factorisation starts at (set by elfihip_gp_extend on a jittered factor; consumed there)
  bool wt_dirty = false;   // a failed attempt has run: the blocks of WT the sweep relies on being zero may hold NaN / Inf

  // device memory
  double* X = nullptr;      // (cap, dp) evidence inputs, zero padded
  double* x2 = nullptr;     // (cap)     |x_i|^2
  double* y = nullptr;      // (cap)
  double* A = nullptr;      // (cap + NB, lda): K -> L (lower); row block [np, np+NB) carries y -> z = L^-1 y
  double* WT = nullptr;     // (cap, lda): L^-T (upper triangular, strictly-lower part kept zero)
  double* WL = nullptr;     // (cap, lda): L^-1 = WT^T (lower), mirrored from WT on first use after a factorisation
                            //             (wl_valid); the second triangular product of the predictor reads it row-wise
  double* Kinv = nullptr;   // (cap, lda): K^-1 -- lower tiles from the gradient kernel (has_kinv); both triangles once the
                            //             acquisition lock-step uses it (kinv_sym, gp_predict.hip)
  double* W11 = nullptr;    // 2 x (NB, NB): inverse of the diagonal block being eliminated (lower), alternating
  double* alpha = nullptr;  // (cap) K^-1 y
  double* red = nullptr;    // small reduction scratch
  int* info = nullptr;      // device: 1-based index of the first non-positive pivot, 0 if none; from word 4 on the
  int ov_flags_off = 0;     // first word of the overlapped sweep's counters in `info`
  int ninfo = 0;            // arrival counters of the fused sweep's steps (four words each); ninfo words in all
  double* h_fit = nullptr;  // pinned, device-visible: sum log L_ii, z'z and the pivot report of the latest rebuild;
                            // words 15 / 14: tickets of the latest rebuild / hyper-gradient (host_wait_ticket)
  unsigned long long fit_ticket = 0, hyper_ticket = 0;
  // integration points of ExpIntVar (elfihip_gp_set_integration_points): V_P = L^-1 K(X, P) stored k-major
  double* VP = nullptr;     // (np, m_pad)
  double* Pint = nullptr;   // (m_pad, dp) the points, zero padded
  int64_t n_int = 0, m_pad = 0;
  unsigned long long fact_gen = 0, vp_gen = 0;  // factorisation counter / the one VP belongs to
  elfihip::DevBuf ws2;      // partials and result tile of the dense product V_P^T v
  elfihip::DevBuf hyper_items;   // work list of the gradient kernel, cached per padded size (gp_hyper.hip)
  int64_t hyper_items_key = 0, hyper_items_n = 0;
  // schedule of the factorisation sweep (elfihip_gp_set_schedule): 0 = by size, 1 = streams, 2 = fused steps;
  // panel_group 0 = by size, else 1 / 2 / 4 panels per pass over the trailing matrix (stream schedule)
  int schedule = 0, panel_group = 0;
  // fused schedule: the update plan of sweep_sched.hpp for sched_nb block columns on sched_nwg workgroups, uploaded
  elfihip::DevBuf sched_mem;
  const void* sched_units = nullptr;
  const void* sched_wgoff = nullptr;
  const void* sched_heads = nullptr;
  std::vector<int> sched_step_off, sched_step_nwg;   // per step: first offset entry, workgroups with work
  int sched_nb = 0, sched_nwg = 0;
  bool sched_far_first = false;   // a workgroup's units by descending column (the chained form of the step launch)
  // per-phase device timing (elfihip_gp_profile): HIP events around the phases of a fit / prediction / gradient call
  // while enabled; sums in milliseconds and call counts per phase (indices: ELFIHIP_PHASE_* in include/elfihip.h)
  bool profile = false;
  hipEvent_t pev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int64_t phase_calls[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // prediction workspace (grown on demand)
  elfihip::DevBuf ws;
  int64_t ws_S = 0;
  // pinned host staging for the query points / results of a prediction call (one H2D, one D2H)
  double* h_stage = nullptr;
  size_t h_cap = 0;  // doubles
  unsigned long long done_seq = 0;  // value of the completion flag after the latest single-pass prediction
  // 0: fused triangular products (four launches per prediction) and, for LCB lock-steps on a factorisation that has
  //    already served KINV_AFTER_STEPS of them, ONE product with K^-1 (three launches); 1: six launches; 2: fused
  //    triangular products only; 3: the K^-1 product from the first LCB lock-step on
  int lockstep_form = 0;
  // acquisition search (gp_acq.hip, elfihip_gp_set_acq_options): host threads of the quasi-Newton algebra (0 = by the
  // machine) and the trace level written to stderr (0 = none)
  int acq_host_threads = 0;
  int acq_trace = 0;
  unsigned* tri_cnt = nullptr;   // 2 x (cap / 32) arrival counters of the fused triangular products (gp_predict.hip)
  // dense predictor (gp_dense.hip): workspace and pinned staging of calls with many points
  int dense_tm = 0;        // row-tile height of the dense form: 0 = by size, else 64 / 32 / 16 (ELFIHIP_DENSE_TM, tests)
  int64_t dense_min = 0;   // points from which a call takes the dense form; 0 = default (elfihip_gp_set_dense_threshold)
  elfihip::DevBuf ws_dense;
  double* h_dense = nullptr;
  size_t hd_cap = 0;  // doubles
  size_t hd_flags = 0;               // flag words initialised so far
  unsigned long long hd_seq = 0;     // value the flags of the latest dense call take
};

namespace elfihip {
// profile helpers: record event `i` on the GP's stream when profiling; after a synchronisation add the time between
// two recorded events to a phase
inline void prof_mark(elfihip_gp* gp, int i) {
  if (gp->profile) (void)hipEventRecord(gp->pev[i], gp->ctx->stream);
}
inline void prof_add(elfihip_gp* gp, int phase, int from, int to) {
  float ms = 0.0f;
  if (gp->profile && hipEventElapsedTime(&ms, gp->pev[from], gp->pev[to]) == hipSuccess) {
    gp->phase_ms[phase] += (double)ms;
    gp->phase_calls[phase] += 1;
  }
}
struct PredictWs {
  double *xs, *xs2, *kr, *kb, *part, *v, *u, *mu_part, *var_part, *g_part, *out;
  int nblk_k;   // blocks of the kstar kernel along i
  int nkc;      // k chunks
  int ngc;      // i chunks of the gradient kernel
  int group;    // 16-point passes per set of launches (kr .. g_part hold this many slices)
};

// One prediction call split into host-side preparation, input fill, device enqueue and result read.
struct PredictPlan {
  PredictWs ws;
  int64_t npass = 0;
  size_t n_in = 0, n_out = 0, outsz = 0;  // doubles
  double* hx = nullptr;                   // pinned: query points + their squared norms
  double* hout = nullptr;                 // pinned: results
  unsigned long long* flag = nullptr;     // pinned: completion flags, one per query column (single-pass calls)
  int n_flags = 0;                        // columns in use
  bool direct = false;                    // no copies: the last kernel writes hout and the flags
  bool by_args = false;                   // ... and the points travel in the kernel arguments (else read from hx)
};
int predict_prepare(elfihip_gp* gp, int64_t S, PredictPlan* P);
void predict_fill(const elfihip_gp* gp, const PredictPlan& P, const double* Xs, int64_t S);
// optional epilogues: MaxVar surface from the assembled prediction (device pointers), ExpIntVar loss from the posterior
// covariances against the integration points (host pointers, M entries each)
struct MaxVarEpilogue {
  double eps;
  const double* prior_pdf;    // (S)
  const double* prior_glog;   // (S, d): gradient of the log prior density
};
struct ExpIntVarArgs {
  double eps;
  const double* w_int;      // omega_i * prior(p_i)^2
  const double* mean_int;   // GP mean at the integration points
  const double* var_int;    // noiseless GP variance at the integration points
};
int predict_enqueue(elfihip_gp* gp, const PredictPlan& P, int64_t S_active, int mode, int noiseless, double beta,
                    const MaxVarEpilogue* mv, bool with_kinv = false);
int predict_wait(elfihip_gp* gp, const PredictPlan& P);
void predict_read(const elfihip_gp* gp, const PredictPlan& P, int64_t S, double* mu, double* var, double* dmu,
                  double* dvar, double* val, double* grad);
// gp_predict.hip: mean / variance / gradients / LCB for S host points (one stream sync per call).
// mode 0 = values only, 1 = + gradients.  Any output pointer may be NULL.
int predict_impl(elfihip_gp* gp, const double* Xs, int64_t S, int mode, int noiseless, double beta, double* mu,
                 double* var, double* dmu, double* dvar, double* val, double* grad);
// gp_dense.hip: the same for many points (S >= dense_min_points()): both triangular products as dense 64 x 64 MFMA tiles
int predict_dense_impl(elfihip_gp* gp, const double* Xs, int64_t S, int mode, int noiseless, double beta, double* mu,
                       double* var, double* dmu, double* dvar, double* val, double* grad);
int64_t dense_min_points(const elfihip_gp* gp);
// gp_hyper.hip: K^-1 = L^-T L^-1 into gp->Kinv (lower 128 x 128 tiles; gp->has_kinv)
int form_kinv_impl(elfihip_gp* gp);
// pieces of gp_predict.hip the dense path shares
int ensure_wl_public(elfihip_gp* gp);
void launch_kstar_passes(elfihip_gp* gp, const double* xs, const double* xs2, double* kr, double* kbt, double* mu_part,
                         int nblk_k, unsigned npass);
void launch_finish_passes(elfihip_gp* gp, const double* mu_part, int nblk_k, const double* var_part, int nblk_v,
                          const double* g_part, int ngc, double* out, int s_left, int noiseless, double beta, int mode,
                          unsigned npass, double* host_out, unsigned long long* done_flags, unsigned long long done_value);
}  // namespace elfihip
