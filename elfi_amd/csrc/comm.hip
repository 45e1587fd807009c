// Multi-GPU exchange behind the C ABI: one process per GPU, RCCL over xGMI.
//
// Replaces the pickled returns of ELFI's batch farm (elfi/client.py:268-274 collects every batch's outputs on the host
// through the client's result queue): the ranks' small per-step results -- a sampler state, Welford statistics, the
// optima of sharded acquisition starts -- are gathered device to device, and a factorised GP can be handed from one
// rank to the others instead of being refactorised there.  The data path itself has no collective (batches are
// independent, elfi/loader.py:164-169); these calls are the ONE exchange per round.
//
// RCCL is loaded at run time (dlopen of librccl.so, the library torch.distributed's "nccl" backend uses on ROCm), so
// libelfihip.so itself has no link-time dependency on it and single-GPU users never touch it; a missing library is an
// error of elfihip_comm_unique_id / elfihip_comm_init_rank, loudly.  Collectives run on the context's stream and do not
// synchronise.  Python programs normally use torch.distributed for the same exchanges (bench.py); this is the route
// for hosts that are not Python.
#include "gp.hpp"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace {

// the slice of the NCCL / RCCL API used here (nccl.h: stable since NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclSuccess = 0 };
enum { ncclFloat64 = 8, ncclInt8 = 0 };

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_error;

void load_rccl() {
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) {
    g_rccl_error = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "not found");
    return;
  }
  bool ok = true;
  auto sym = [&](const char* n) {
    void* p = dlsym(g_rccl.lib, n);
    if (!p) {
      ok = false;
      g_rccl_error = std::string("RCCL symbol missing: ") + n;
    }
    return p;
  };
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
  g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(sym("ncclBroadcast"));
  g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
  g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
  g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
  g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) {
    dlclose(g_rccl.lib);
    g_rccl.lib = nullptr;
  }
}

int need_rccl(elfihip_ctx* ctx) {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.lib) return elfihip::fail(ctx, ELFIHIP_ERR_STATE, "%s", g_rccl_error.c_str());
  return ELFIHIP_OK;
}

}  // namespace

struct elfihip_comm {
  elfihip_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

#define ELFIHIP_CHECK_RCCL(ctx, call)                                                                         \
  do {                                                                                                        \
    const int rc_ = (call);                                                                                   \
    if (rc_ != ncclSuccess)                                                                                   \
      return elfihip::fail(ctx, ELFIHIP_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc_) : "?"); \
  } while (0)

using namespace elfihip;

extern "C" {

int elfihip_comm_unique_id(elfihip_ctx* ctx, void* id128) {
  if (!ctx || !id128) return fail(ctx, ELFIHIP_ERR_ARG, "NULL argument");
  ELFIHIP_TRY(need_rccl(ctx));
  ncclUniqueId id;
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.GetUniqueId(&id));
  std::memcpy(id128, id.internal, sizeof id.internal);
  return ELFIHIP_OK;
}

int elfihip_comm_init_rank(elfihip_ctx* ctx, const void* id128, int rank, int world_size, elfihip_comm** out) {
  if (!ctx || !id128 || !out) return fail(ctx, ELFIHIP_ERR_ARG, "NULL argument");
  *out = nullptr;
  ELFIHIP_REQUIRE(ctx, world_size >= 1 && rank >= 0 && rank < world_size, "rank %d outside [0, %d)", rank, world_size);
  ELFIHIP_TRY(need_rccl(ctx));
  DeviceGuard g(ctx->device);
  ncclUniqueId id;
  std::memcpy(id.internal, id128, sizeof id.internal);
  elfihip_comm* c = new elfihip_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world_size;
  const int rc = g_rccl.CommInitRank(&c->comm, world_size, id, rank);
  if (rc != ncclSuccess) {
    delete c;
    return fail(ctx, ELFIHIP_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(rc));
  }
  *out = c;
  return ELFIHIP_OK;
}

int elfihip_comm_free(elfihip_comm* c) {
  if (!c) return ELFIHIP_OK;
  DeviceGuard g(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  delete c;
  return ELFIHIP_OK;
}

int elfihip_comm_allgather_f64(elfihip_comm* c, const double* dsend, int64_t count, double* drecv) {
  if (!c) return fail(nullptr, ELFIHIP_ERR_ARG, "comm is NULL");
  elfihip_ctx* ctx = c->ctx;
  ELFIHIP_REQUIRE(ctx, count >= 0 && (count == 0 || (dsend && drecv)), "bad arguments");
  if (count == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.AllGather(dsend, drecv, (size_t)count, ncclFloat64, c->comm, ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_comm_gather_f64(elfihip_comm* c, const double* dsend, int64_t count, double* drecv, int root) {
  if (!c) return fail(nullptr, ELFIHIP_ERR_ARG, "comm is NULL");
  elfihip_ctx* ctx = c->ctx;
  ELFIHIP_REQUIRE(ctx, count >= 0 && root >= 0 && root < c->world && (count == 0 || dsend) &&
                           (count == 0 || c->rank != root || drecv),
                  "bad arguments");
  if (count == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  // point-to-point under one group: every rank sends its block to the root, the root posts world receives.  A call
  // that fails inside the group must not leave it open (every later RCCL call of this thread would be queued into a
  // group nobody closes): remember the first failure, close the group, then report.
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.GroupStart());
  int rc = ncclSuccess;
  const char* what = "";
  if (c->rank == root)
    for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {
      rc = g_rccl.Recv(drecv + (size_t)r * count, (size_t)count, ncclFloat64, r, c->comm, ctx->stream);
      what = "ncclRecv";
    }
  if (rc == ncclSuccess) {
    rc = g_rccl.Send(dsend, (size_t)count, ncclFloat64, root, c->comm, ctx->stream);
    what = "ncclSend";
  }
  const int rc_end = g_rccl.GroupEnd();
  if (rc != ncclSuccess) return fail(ctx, ELFIHIP_ERR_HIP, "%s failed: %s", what, g_rccl.GetErrorString(rc));
  if (rc_end != ncclSuccess) return fail(ctx, ELFIHIP_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(rc_end));
  return ELFIHIP_OK;
}

int elfihip_comm_bcast_f64(elfihip_comm* c, double* dbuf, int64_t count, int root) {
  if (!c) return fail(nullptr, ELFIHIP_ERR_ARG, "comm is NULL");
  elfihip_ctx* ctx = c->ctx;
  ELFIHIP_REQUIRE(ctx, count >= 0 && root >= 0 && root < c->world && (count == 0 || dbuf), "bad arguments");
  if (count == 0) return ELFIHIP_OK;
  DeviceGuard g(ctx->device);
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.Broadcast(dbuf, dbuf, (size_t)count, ncclFloat64, root, c->comm, ctx->stream));
  return ELFIHIP_OK;
}

int elfihip_comm_bcast_factor(elfihip_comm* c, elfihip_gp* gp, int root) {
  if (!c || !gp) return fail(nullptr, ELFIHIP_ERR_ARG, "NULL argument");
  elfihip_ctx* ctx = c->ctx;
  ELFIHIP_REQUIRE(ctx, gp->ctx == ctx, "the GP belongs to another context");
  ELFIHIP_REQUIRE(ctx, root >= 0 && root < c->world, "root %d outside [0, %d)", root, c->world);
  DeviceGuard g(ctx->device);
  hipStream_t st = ctx->stream;
  // Header first (host values travel through a device scalar block): evidence count, hyper-parameters, log det,
  // y'K^-1 y, whether the root is factorised at all, and the root's LAYOUT (input dimension, its padding, capacity, row
  // pitch).  ncclBroadcast does not compare counts across ranks, and the bulk broadcasts below move whole rows at the
  // local pitch: a receiver laid out differently would be corrupted silently (or hang).  So every rank checks the
  // header against its own object, the verdicts are all-gathered, and ALL ranks return the same status before any
  // bulk transfer -- no rank is left waiting in a collective its peers never enter.
  ELFIHIP_REQUIRE(ctx, c->world <= 64, "bcast_factor supports up to 64 ranks");
  ELFIHIP_CHECK_HIP(ctx, ctx->par.reserve((12 + 2 * 64) * sizeof(double)));
  double* hdr = ctx->par.as<double>();          // 12 header doubles, then one verdict per rank, then this rank's own
  double* verdicts = hdr + 12;
  double h[12] = {(double)gp->n, gp->var, gp->ls, gp->bias, gp->noise, gp->logdet, gp->yKy, gp->factored ? 1.0 : 0.0,
                  (double)gp->d, (double)gp->dp, (double)gp->cap, (double)gp->lda};
  if (c->rank == root) ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(hdr, h, sizeof h, hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.Broadcast(hdr, hdr, 12, ncclFloat64, root, c->comm, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(h, hdr, sizeof h, hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  const int64_t n = (int64_t)h[0];
  // verdict of this rank: 0 fine, 1 root not factorised, 2 no room, 3 different layout
  double mine = 0.0;
  if (h[7] != 1.0 || n < 1)
    mine = 1.0;
  else if (n > gp->cap)
    mine = 2.0;
  else if ((int)h[8] != gp->d || (int)h[9] != gp->dp || (int64_t)h[10] != gp->cap || (int64_t)h[11] != gp->lda)
    mine = 3.0;
  double all[64];
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(verdicts + 64, &mine, sizeof mine, hipMemcpyHostToDevice, st));
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.AllGather(verdicts + 64, verdicts, 1, ncclFloat64, c->comm, st));
  ELFIHIP_CHECK_HIP(ctx, hipMemcpyAsync(all, verdicts, (size_t)c->world * sizeof(double), hipMemcpyDeviceToHost, st));
  ELFIHIP_CHECK_HIP(ctx, hipStreamSynchronize(st));
  for (int r = 0; r < c->world; ++r) {
    if (all[r] == 0.0) continue;
    if (all[r] == 1.0) return fail(ctx, ELFIHIP_ERR_STATE, "the root's GP (rank %d) is not factorised", root);
    if (all[r] == 2.0)
      return fail(ctx, ELFIHIP_ERR_ARG, "the root's GP holds %lld points, the GP of rank %d has no room for them", (long long)n, r);
    return fail(ctx, ELFIHIP_ERR_ARG,
                "the GP of rank %d is laid out differently from the root's (d %d, padded d %d, capacity %lld, pitch %lld): "
                "create every rank's GP with the same d and capacity", r, (int)h[8], (int)h[9], (long long)h[10], (long long)h[11]);
  }
  if (c->rank != root) {
    gp->n = n;
    gp->np = round_up(n, NB);
    gp->var = h[1];
    gp->ls = h[2];
    gp->bias = h[3];
    gp->noise = h[4];
    gp->logdet = h[5];
    gp->yKy = h[6];
  }
  const int64_t np = gp->np;
  // evidence, factor L (+ the z row block), L^-T, alpha: the rows in use, whole rows (equal pitch on all ranks: checked above)
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.Broadcast(gp->X, gp->X, (size_t)np * gp->dp, ncclFloat64, root, c->comm, st));
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.Broadcast(gp->x2, gp->x2, (size_t)np, ncclFloat64, root, c->comm, st));
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.Broadcast(gp->y, gp->y, (size_t)np, ncclFloat64, root, c->comm, st));
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.Broadcast(gp->A, gp->A, (size_t)(np + NB) * gp->lda, ncclFloat64, root, c->comm, st));
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.Broadcast(gp->WT, gp->WT, (size_t)np * gp->lda, ncclFloat64, root, c->comm, st));
  ELFIHIP_CHECK_RCCL(ctx, g_rccl.Broadcast(gp->alpha, gp->alpha, (size_t)np, ncclFloat64, root, c->comm, st));
  if (c->rank != root) {
    gp->factored = true;
    gp->has_kinv = false;
    gp->kinv_sym = false;
    gp->wl_valid = false;
    ++gp->fact_gen;
  }
  return ELFIHIP_OK;
}

}  // extern "C"
