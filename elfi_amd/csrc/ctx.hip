// Context lifecycle, error text, stream adoption and event timing for libelfihip.so.
#include "common.hpp"

#include <atomic>
#include <chrono>

#include <cstdlib>

namespace elfihip {
thread_local std::string g_err;

// Streams of the GP factorisation's stream schedule: `hi` (high priority) carries the critical chain, `bulk` the
// rest of the trailing update.  (A CU-mask partition of the two was tried and removed: on this stack a stream made with
// hipExtStreamCreateWithCUMask still runs on all 256 CUs -- scripts/native/cumask_probe.hip prints the XCC / CU ids.)
__global__ void mail_kernel(MailSrc S, unsigned long long* box, unsigned long long ticket) {
  const int t = threadIdx.x;
  if (t < S.n)
    box[t] = S.bytes[t] == 8 ? *reinterpret_cast<const unsigned long long*>(S.p[t])
                             : (unsigned long long)*reinterpret_cast<const unsigned int*>(S.p[t]);
  __threadfence_system();   // (one wave: the words above are on their way to host memory before the ticket is)
  if (t == 0) __hip_atomic_store(box + 7, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int mail_post(elfihip_ctx* ctx, const MailSrc& S) {
  if (!ctx->mail) {
    void* p = nullptr;
    ELFIHIP_CHECK_HIP(ctx, hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent));
    ctx->mail = reinterpret_cast<unsigned long long*>(p);
    ctx->mail[7] = 0;
  }
  hipLaunchKernelGGL(mail_kernel, dim3(1), dim3(64), 0, ctx->stream, S, ctx->mail, ++ctx->mail_ticket);
  return launch_status(ctx, "mail_kernel");
}

int host_wait_ticket(elfihip_ctx* ctx, const volatile unsigned long long* word, unsigned long long want) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; *word != want; ++spins) {
    if ((spins & 1023u) == 1023u &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.25) {
      // not in a quarter of a second: let the stream say what happened (or finish: a long queue ahead of the writer)
      ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (*word != want) return fail(ctx, ELFIHIP_ERR_STATE, "internal: the ticket %llu never arrived", want);
      break;
    }
    __builtin_ia32_pause();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return ELFIHIP_OK;
}

int mail_wait(elfihip_ctx* ctx) { return host_wait_ticket(ctx, ctx->mail + 7, ctx->mail_ticket); }

int ctx_aux(elfihip_ctx* ctx) {
  if (ctx->hi_stream) return ELFIHIP_OK;
  int lo = 0, hi = 0;
  ELFIHIP_CHECK_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
  ELFIHIP_CHECK_HIP(ctx, hipStreamCreateWithPriority(&ctx->hi_stream, hipStreamNonBlocking, hi));
  ELFIHIP_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->bulk_stream, hipStreamNonBlocking));
  // device-scope release: these events only order kernels on this GPU, no host visibility needed
  const unsigned flags = hipEventDisableTiming | hipEventReleaseToDevice;
  ELFIHIP_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_a, flags));
  ELFIHIP_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_b, flags));
  for (auto& e : ctx->ev_u) ELFIHIP_CHECK_HIP(ctx, hipEventCreateWithFlags(&e, flags));
  return ELFIHIP_OK;
}
}  // namespace elfihip

using namespace elfihip;

extern "C" {

int elfihip_version(void) { return ELFIHIP_VERSION; }

int elfihip_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(nullptr, ELFIHIP_ERR_ARG, "out is NULL");
  *out = nullptr;
  if (bytes == 0) return ELFIHIP_OK;
  hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    *out = nullptr;
    return fail(nullptr, ELFIHIP_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  return ELFIHIP_OK;
}

int elfihip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
  return ELFIHIP_OK;
}

int elfihip_kept_rows(elfihip_ctx* ctx, uint64_t* epoch, int64_t* n, int* m) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  if (epoch) *epoch = ctx->rows_epoch;
  if (n) *n = ctx->rows_n;
  if (m) *m = ctx->rows_m;
  return ELFIHIP_OK;
}

int elfihip_kept_distances(elfihip_ctx* ctx, uint64_t* epoch, int64_t* n, int* ncols) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  if (epoch) *epoch = ctx->keep_epoch;
  if (n) *n = ctx->keep_n;
  if (ncols) *ncols = ctx->keep_cols;
  return ELFIHIP_OK;
}

int elfihip_device_count(int* count) {
  if (!count) return fail(nullptr, ELFIHIP_ERR_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(nullptr, ELFIHIP_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  *count = n;
  return ELFIHIP_OK;
}

int elfihip_ctx_create(int device, elfihip_ctx** out) {
  if (!out) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx out-pointer is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return fail(nullptr, ELFIHIP_ERR_HIP, "no HIP device available (%s)",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0) {
    e = hipGetDevice(&device);
    if (e != hipSuccess)
      return fail(nullptr, ELFIHIP_ERR_HIP, "hipGetDevice failed: %s", hipGetErrorString(e));
  }
  if (device >= n) return fail(nullptr, ELFIHIP_ERR_ARG, "device %d out of range (have %d)", device, n);

  elfihip_ctx* ctx = new elfihip_ctx();
  ctx->device = device;
  DeviceGuard g(device);
  if (!g.ok) {
    delete ctx;
    return fail(nullptr, ELFIHIP_ERR_HIP, "cannot select device %d", device);
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
  if (e != hipSuccess) {
    int rc = fail(nullptr, ELFIHIP_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    delete ctx;
    return rc;
  }
  ctx->cu_count = prop.multiProcessorCount;
  ctx->stream = ctx->own_stream;
  *out = ctx;
  return ELFIHIP_OK;
}

int elfihip_ctx_destroy(elfihip_ctx* ctx) {
  if (!ctx) return ELFIHIP_OK;
  {
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->in.release();
    ctx->out.release();
    ctx->par.release();
    ctx->scratch.release();
    ctx->stat.release();
    ctx->keep.release();
    ctx->rows.release();
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
    for (auto e : ctx->ev_u)
      if (e) (void)hipEventDestroy(e);
    if (ctx->hi_stream) (void)hipStreamDestroy(ctx->hi_stream);
    if (ctx->bulk_stream) (void)hipStreamDestroy(ctx->bulk_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->mail) (void)hipHostFree(ctx->mail);
    if (ctx->fold_cnt) (void)hipFree(ctx->fold_cnt);
  }
  delete ctx;
  return ELFIHIP_OK;
}

const char* elfihip_last_error(const elfihip_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_err.c_str();
}

int elfihip_ctx_set_stream(elfihip_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return ELFIHIP_OK;
}

int elfihip_dist_set_form(elfihip_ctx* ctx, int form) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, form >= 0 && form <= 2, "form %d outside {0: LDS-DMA row stream for the distance kernels, 1: register-staged "
                  "pipelines, 2: LDS-DMA also for the fused adaptive pass}", form);
  ctx->dist_form = form;
  return ELFIHIP_OK;
}

int elfihip_topk_set_form(elfihip_ctx* ctx, int form) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  ELFIHIP_REQUIRE(ctx, form >= 0 && form <= 2, "form %d outside {0: resident, 1: nine launches, 2: resident without the register form}", form);
  ctx->topk_form = form;
  return ELFIHIP_OK;
}

int elfihip_ctx_synchronize(elfihip_ctx* ctx) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_device_info(elfihip_ctx* ctx, int* cu_count, int* clock_khz, int* mem_clock_khz,
                        int* mem_bus_bits, int64_t* total_mem, char* name, int name_len) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  hipDeviceProp_t prop;
  ELFIHIP_CHECK_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (mem_clock_khz) *mem_clock_khz = prop.memoryClockRate;
  if (mem_bus_bits) *mem_bus_bits = prop.memoryBusWidth;
  if (total_mem) *total_mem = (int64_t)prop.totalGlobalMem;
  if (name && name_len > 0) {
    snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return ELFIHIP_OK;
}

int elfihip_timer_start(elfihip_ctx* ctx) {
  if (!ctx) return fail(nullptr, ELFIHIP_ERR_ARG, "ctx is NULL");
  DeviceGuard g(ctx->device);
  ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_timer_stop(elfihip_ctx* ctx, float* elapsed_ms) {
  if (!ctx || !elapsed_ms) return fail(ctx, ELFIHIP_ERR_ARG, "NULL argument");
  DeviceGuard g(ctx->device);
  ELFIHIP_CHECK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  ELFIHIP_CHECK_HIP(ctx, hipEventSynchronize(ctx->ev1));
  ELFIHIP_CHECK_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
  return ELFIHIP_OK;
}

}  // extern "C"
