// Shared internals of libelfihip.so: context object, error plumbing, device workspace.
// Nothing here is part of the ABI (see include/elfihip.h for that).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/elfihip.h"

namespace elfihip {

// Grow-only device buffer owned by a context (host entry points stage through these).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      p = nullptr;
      cap = 0;
      if (e != hipSuccess) return e;
    }
    // round up so a slowly growing batch does not realloc every call
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return e;
    }
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

}  // namespace elfihip

struct elfihip_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;  // own_stream or an adopted one
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // look-ahead machinery of the GP factorisation (created on first use)
  hipStream_t hi_stream = nullptr;    // critical path of the sweep (high priority)
  hipStream_t bulk_stream = nullptr;  // passes over the trailing matrix
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  hipEvent_t ev_u[4] = {nullptr, nullptr, nullptr, nullptr};  // look-ahead window columns of the next panel group
  int cu_count = 0;
  // small read-backs (a counter, a threshold, a time-out flag): a one-workgroup kernel writes them into this page-locked,
  // device-visible mailbox and the host reads it after the stream's synchronisation it needs anyway -- a 4-byte
  // hipMemcpyAsync into a pageable stack variable is a staged blit of 30 us on the stream (profiles/r05_cfg4_trace.md:
  // 4.5 of them per adaptive-distance round)
  unsigned long long* mail = nullptr;
  unsigned long long mail_ticket = 0;   // number of the latest mail_post (word 7 of the mailbox carries it: mail_wait)
  unsigned* fold_cnt = nullptr;       // arrival counter of adaptive_finish_kernel's fold (adaptive.hip), zero between launches
  int dist_form = 0;                  // 0: LDS-DMA row stream where the shape allows; 1: register-staged pipeline (elfihip_dist_set_form)
  int topk_form = 0;                  // 0: resident selection (keys in registers up to 2 10^6 rows) with the nine-launch form as fallback; 1: nine-launch form; 2: resident, keys re-read from memory in every phase
  unsigned dense_lds_mask = 0;        // dense_tri_kernel<.,64/32/16>: dynamic-LDS limit raised (gp_dense.hip)
  bool ov_lds_enabled = false;        // sweep_update_kernel's (gp_fit.hip)
  bool step_lds_enabled = false;      // step_kernel's dynamic-LDS limit has been raised (gp_fit.hip)
  std::string err;
  // staging buffers for the host entry points
  elfihip::DevBuf in, out, par, scratch;
  elfihip::DevBuf stat;   // workgroup partials of the fused adaptive-distance pass (adaptive.hip; topk.hip owns `scratch`)
  // the distances the last host-form distance call returned, kept on the device for the sampler state
  // (elfihip_kept_distances / elfihip_reject_push_kept): (keep_n, keep_cols) row-major; the epoch names the call
  elfihip::DevBuf keep;
  uint64_t keep_epoch = 0;
  int64_t keep_n = 0;
  bool keep_valid = false;            // the copy the epoch names exists (false: the buffer could not be had)
  int keep_cols = 0;
  // ... and the rows the last device-side simulator call (elfihip_randn_rows) returned: (rows_n, rows_m) row-major with
  // pitch rows_m, for the distance call that follows (elfihip_kept_rows / elfihip_adaptive_push_kept)
  elfihip::DevBuf rows;
  uint64_t rows_epoch = 0;
  int64_t rows_n = 0;
  int rows_m = 0;
};

namespace elfihip {

extern thread_local std::string g_err;  // context-less errors

inline int fail(elfihip_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    g_err = buf;
  return code;
}

#define ELFIHIP_CHECK_HIP(ctx, expr)                                                          \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return ::elfihip::fail((ctx), ELFIHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,          \
                             hipGetErrorString(e__), __FILE__, __LINE__);                     \
  } while (0)

#define ELFIHIP_REQUIRE(ctx, cond, ...)                                          \
  do {                                                                           \
    if (!(cond)) return ::elfihip::fail((ctx), ELFIHIP_ERR_ARG, __VA_ARGS__);    \
  } while (0)

#define ELFIHIP_TRY(expr)            \
  do {                               \
    int rc__ = (expr);               \
    if (rc__ != ELFIHIP_OK) return rc__; \
  } while (0)

// RAII device guard: entry points run on the context's device and restore the caller's.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) {
      ok = false;
      return;
    }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

int ctx_aux(elfihip_ctx* ctx);  // ctx.hip: lazily creates hi_stream / ev_a / ev_b

// Up to four device words (4 or 8 bytes each) -> the context's mailbox, in stream order; ctx.hip.  mail_read() after the
// caller's hipStreamSynchronize returns word i (zero-extended).
struct MailSrc {
  const void* p[4];
  int bytes[4];
  int n;
};
int mail_post(elfihip_ctx* ctx, const MailSrc& S);
inline unsigned long long mail_read(const elfihip_ctx* ctx, int i) { return ctx->mail[i]; }
// Wait for the LATEST mail_post alone -- not for the kernels enqueued behind it: the mail kernel writes its ticket last
// (system-scope release) and the host polls that word of the page-locked mailbox.  A caller that has more work for the
// stream enqueues it first and then waits here, so that the device is busy while the host wakes up (a stream
// synchronisation in the middle of a round left it idle for 20-30 us).  Falls back to hipStreamSynchronize when the ticket
// does not arrive (a faulted stream reports its error there).
int mail_wait(elfihip_ctx* ctx);
// The same wait on any word of page-locked, device-visible memory that a kernel on the context's stream stores LAST, behind a
// system-scope fence (the rebuild's scalars: gp_fit.hip / gp_hyper.hip).
int host_wait_ticket(elfihip_ctx* ctx, const volatile unsigned long long* word, unsigned long long want);
// device side of such a ticket: after the thread's own stores to the page-locked words
__device__ __forceinline__ void post_ticket(double* slot, unsigned long long ticket) {
  __threadfence_system();
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot), ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Host-form distance calls leave a device copy of what they return (n x cols doubles at dsrc, on the context's stream).
// Every host-form distance call comes through here, also with n = 0 (an empty batch is a call: the epoch must move on, or
// the previous call's rows would be folded in again under the new call's name).  keep_valid says whether the copy named by
// the epoch really exists: when the buffer cannot be had, elfihip_reject_push_kept answers ELFIHIP_ERR_STATE and the
// sampler uploads the host array instead (round 4 reported OK with zero rows: the batch was silently dropped).
inline int keep_distances(elfihip_ctx* ctx, const double* dsrc, int64_t n, int cols) {
  ++ctx->keep_epoch;
  ctx->keep_n = 0;
  ctx->keep_cols = cols;
  ctx->keep_valid = false;
  if (n <= 0) {
    ctx->keep_valid = true;   // an empty batch: nothing to keep, nothing to fold in
    return ELFIHIP_OK;
  }
  if (ctx->keep.reserve((size_t)n * cols * sizeof(double)) != hipSuccess) {
    (void)hipGetLastError();
    return ELFIHIP_OK;        // no copy kept (keep_valid stays false): the sampler uploads
  }
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(ctx->keep.p, dsrc, (size_t)n * cols * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  ctx->keep_n = n;
  ctx->keep_valid = true;
  return ELFIHIP_OK;
}

inline int launch_status(elfihip_ctx* ctx, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess)
    return fail(ctx, ELFIHIP_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
  return ELFIHIP_OK;
}

#if defined(__HIPCC__)
// ---- sums across lanes by DPP (register moves) ------------------------------------------------------------------
// __shfl_xor of a double is two ds_bpermute_b32 -- a round trip through the LDS crossbar of a hundred cycles, and the
// steps of a butterfly depend on each other; inside 16 lanes the same partners are reachable by the data-parallel
// primitives: quad permutes for xor 1 / xor 2, the half-row mirror (lane i <-> 7 - i of each 8) and the row mirror
// (i <-> 15 - i of each 16).  After the two quad steps every lane of a quad holds the quad's sum, so the mirrored partner
// holds the very bits the xor-4 / xor-8 partner holds: the sums are bit-identical to the butterfly's.
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// every lane of an aligned group of 8 / 16 lanes gets the group's sum: (((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7))) ...
__device__ __forceinline__ double lanes8_sum(double q) {
  q += dpp_move_f64<0xB1>(q);    // quad_perm [1, 0, 3, 2]
  q += dpp_move_f64<0x4E>(q);    // quad_perm [2, 3, 0, 1]
  q += dpp_move_f64<0x141>(q);   // row_half_mirror
  return q;
}
__device__ __forceinline__ double lanes16_sum(double q) {
  q = lanes8_sum(q);
  q += dpp_move_f64<0x140>(q);   // row_mirror
  return q;
}
#endif

}  // namespace elfihip
