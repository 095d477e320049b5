// agpl_common.h -- context, error plumbing and small device helpers shared by the libagpl.so sources.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <utility>
#include <vector>

#include "../../include/agpl.h"

struct rocblas_handle_s; // fwd (rocblas_handle is a pointer to an opaque struct)

struct agpl_ctx {
    int device = 0;
    uint64_t seed = 0;
    int64_t point_offset = 0; // global index of local point 0 (agpl_ctx_set_point_offset): per-point Philox streams
                              // are keyed on point_offset + i, so that N sharded over ranks draws what one rank would
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void *rocblas = nullptr; // rocblas_handle, created lazily by agpl_update.hip
    hipStream_t aux_stream = nullptr; // high-priority side stream of the dense Cholesky's look-ahead (agpl_dense.hip), lazily created
    hipEvent_t aux_ev[2] = {nullptr, nullptr};
    // scratch (lazily grown)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    void *ws2 = nullptr; // small persistent scratch (reductions, info flags)
    size_t ws2_bytes = 0;
    unsigned *pg_retry = nullptr; // PG(1) kernels: [0] entries, [1] workgroups done (both zero between launches), [2 ..] the point list
    size_t pg_retry_entries = 0;
    unsigned *red_cnt = nullptr; // arrival counters of the slab reduction (one per 1 KB of G and per half block of g), zero between launches
    size_t red_cnt_entries = 0;
    double *logtheta_dev = nullptr; // categorical link parameters mirrored on device
    int logtheta_cap = 0;
    double logtheta_host[128];      // last uploaded values (skip the copy when unchanged)
    int logtheta_n = 0;
    // optional kernel timing (agpl_timing_*): event pairs per kernel family
    int accumulate_split = 0; // internal: 1 while a *_split / *_image / plan entry point runs its accumulation (split-float16), else 0 (f32-input MFMA)
    int ncu = 0;              // compute units of `device` (queried once, by the first queue-served launch)
    int strip_attr = 0;       // the accumulation kernels' dynamic-LDS attributes are set (once)
    int queue_attr = 0;       // marginal_factor_queue_kernel's dynamic-LDS attribute is set (once)
    int pipe_attr = 0;        // factor_pipe_kernel's dynamic-LDS attribute is set (once)
    const void *checked_image = nullptr; // the accumulate image whose header agpl_syrk_image_launch has validated last, and for which (N, M)
    int64_t checked_image_N = 0;
    int32_t checked_image_M = 0;
    double *elbo_part = nullptr;  // per-workgroup partial sums of the ELBO terms that ride the per-point kernel (1024 doubles)
    int live_plans = 0;           // agpl_plan objects created on this context and not yet destroyed (agpl_ctx_destroy refuses while > 0)
    bool debug_force_rescue = false; // agpl_debug_force_factor_rescue (test hook)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[4]; // 0 marginal, 1 syrk, 2 gibbs point pass, 3 aux_sample
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    // deferred outcome of the last agpl_gaussian_factor_async: info words in pinned host memory, ready once the event
    // has passed (agpl_pending_resolve waits for the event only, not for work enqueued behind it)
    bool pend = false;
    int pend_n = 0;          // info words (L, or 2 L for the two-block form)
    int pend_latents = 0;
    int *pend_host = nullptr; // hipHostMalloc (mapped), 128 ints
    bool pend_gamma_word = false; // pend_host[127] carries the sweep's bad-gamma word
    int *pend_host_dev = nullptr; // the same memory as the device addresses it
    hipEvent_t pend_ev = nullptr;
    char err[512] = {0};
};

#include "agpl_slices.h" // agpl_slice_plan / agpl_slice_range: how one accumulation cuts N into slices (host-testable, no HIP)

// reports (and clears) the deferred outcome of the last asynchronous factorisation; AGPL_OK when none is pending
int32_t agpl_pending_resolve(agpl_ctx *ctx);

// RAII-less helpers: record a start event now, and the matching stop event after the launch
int32_t agpl_timing_begin(agpl_ctx *ctx, int which);
int32_t agpl_timing_end(agpl_ctx *ctx, int which);

#define AGPL_FAIL(ctx, code, ...)                                   \
    do {                                                            \
        if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
        return (code);                                              \
    } while (0)

#define AGPL_HIP(ctx, call)                                                                   \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            AGPL_FAIL(ctx, AGPL_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                      __FILE__, __LINE__);                                                    \
    } while (0)

#define AGPL_LAUNCH_CHECK(ctx)                                                              \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess)                                                              \
            AGPL_FAIL(ctx, AGPL_ERR_HIP, "kernel launch failed: %s (%s:%d)",                \
                      hipGetErrorString(e__), __FILE__, __LINE__);                          \
    } while (0)

// grow-only scratch
int32_t agpl_ws_reserve(agpl_ctx *ctx, size_t bytes);
int32_t agpl_ws2_reserve(agpl_ctx *ctx, size_t bytes);
int32_t agpl_red_cnt_reserve(agpl_ctx *ctx, int64_t n); // the slab reduction's arrival counters (agpl_core.hip)
int32_t agpl_pg_retry_reserve(agpl_ctx *ctx, int64_t n); // the retry list of the PG(1) kernels for n points (agpl_core.hip)

// device-side view of a likelihood descriptor (logtheta mirrored to device memory)
struct agpl_lik_dev {
    int32_t kind;
    int32_t nlatent;
    double p[4];
    const double *logtheta; // device
    double sum_theta;       // categorical.jl:16-20
    double cat_const;       // categorical.jl:12-14 (bijective only)
};
int32_t agpl_lik_to_device(agpl_ctx *ctx, const agpl_lik_desc *lik, agpl_lik_dev *out);

// bytes of the plan's two kinds of image (agpl_split.hip, agpl_syrk.hip)
int64_t agpl_split_features_bytes(int64_t N, int32_t M);
int64_t agpl_accumulate_image_bytes(int64_t N, int32_t M);

static inline int64_t agpl_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
