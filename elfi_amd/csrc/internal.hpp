// Entry points one translation unit of libelfihip.so offers to another (not part of the ABI).
#pragma once

#include "common.hpp"
#include "tile_stream.hpp"

struct elfihip_reject;

namespace elfihip {

// distance.hip: device-pointer distance passes, optionally with the fused selection filter (see RejectFilter).
// *filtered reports whether the kernel that ran offered the candidates itself.
int dist_rows_dev_impl(elfihip_ctx* ctx, int metric, const double* dX, int64_t n, int m, int64_t ldx, const double* dy,
                       const double* daux, double p, double* dout, const RejectFilter* F, bool* filtered);
int dist_multiw_dev_impl(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx, const double* dy,
                         const double* dW, int K, double* dout, const RejectFilter* F, bool* filtered);

// topk.hip: k smallest of n strided doubles (ascending by (value, index)); force_multi = the nine-launch form, which
// needs no co-residency and cannot time out.
int topk_dev_impl(elfihip_ctx* ctx, const double* dD, int64_t n, int64_t stride, int64_t k, double* dvals,
                  int64_t* didx, bool force_multi);

const void* topk_resident_err_dev(elfihip_ctx* ctx);   // see topk.hip: device address of the resident selection's time-out flag, or NULL

// reject.hip: the sampler state; push of a device-resident batch (no device guard, no argument checks)
elfihip_ctx* reject_ctx(elfihip_reject* h);
int reject_push_rows_impl(elfihip_reject* h, int metric, const double* dX, int64_t n, int m, int64_t ldx,
                          const double* dy, const double* daux, double p, double* dout, int64_t row_base);


// welford.hip: AdaptiveDistance.add_data in the reference's own (two-pass) form, into dstate (1 + 2m)
int welford_dev_impl(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx, double* dstate);

// adaptive.hip: the fused pass of an adaptive-distance batch (K nested distances + column statistics + selection)
bool adaptive_pass_supported(const double* dX, int m, int64_t ldx, int K);
int adaptive_max_parts(const elfihip_ctx* ctx);
int adaptive_pass_impl(elfihip_ctx* ctx, const double* dX, int64_t n, int m, int64_t ldx, const double* dy,
                       const double* dW, int K, double* dout, const RejectFilter* F, const double* dacc,
                       unsigned long long* dacc_count, double* partial, int* nparts);
int adaptive_stats_finish(elfihip_ctx* ctx, const double* partial, int nparts, int m, double* bst, double* dstate);
// reject.hip: the batch against a sampler state (h may be NULL: distances and statistics only)
int adaptive_push_impl(elfihip_ctx* ctx, elfihip_reject* h, const double* dX, int64_t n, int m, int64_t ldx,
                       const double* dy, const double* dW, int K, double* dout, double* dwelford, int64_t row_base);

}  // namespace elfihip
