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

// Points per accumulation slice (one workgroup's float32 accumulation run; one slab per slice and tile).  4096, and 8192 where the
// feature matrix is a single 256-tile wide: there the slab write + reduction is a larger share of a slice (C4-shape accumulate
// 2.92 against 3.27 ms on one box); at M >= 512 the longer slice measured slower (round 3).
__host__ __device__ constexpr int agpl_chunk_points(int M) { return M <= 256 ? 8192 : 4096; }
// Points per accumulation slice (float32 sums within a slice, float64 across slices): the figure above, doubled while the launch
// would still have >= 32 workgroups per CU -- N = 1e7 at M = 1024: 2442 slices of 36 x 64 KB are 5.8 GB of slabs written and read
// again per sweep (the reduction alone 0.98-1.05 ms); 1221 slices (47 workgroups per CU) halve that.  A function of (N, M, L) only:
// results repeat.
#ifndef AGPL_SLICE_MIN_WG
#define AGPL_SLICE_MIN_WG 32 // workgroups per CU the launch must keep after a doubling.  (Measured with 8: C2 then runs 8192-point slices at 14
                             // workgroups per CU -- the accumulation kernel loses 0.24-0.34 ms to its coarser tail, the reduction gains 0.14.)
#endif
inline int agpl_slice_points(int64_t N, int M, int L) {
    int chunk = agpl_chunk_points(M);
    const int64_t nb2 = (M + 255) / 256, pairs = nb2 * (nb2 + 1) / 2;
    while (chunk < 16384 && (int64_t)L * pairs * ((N + 2 * chunk - 1) / (2 * chunk)) >= AGPL_SLICE_MIN_WG * 256) chunk *= 2;
    return chunk;
}
// The slices of one accumulation (round 6): `nbig` slices of `chunk` points, then the rest of the points in slices of chunk / 4.
// A launch of equal workgroups of 0.2-0.45 ms each ends with a tail of about one of them on a mostly idle device -- measured as a
// fixed 0.21 ms per launch at M = 512 whatever N is (kernel time = 0.21 + 0.217 ms x rounds of 256 workgroups:
// profiles/NOTES_r06.md), which is 3 % of the launch at N = 1e7 and 23 % of it at a rank's N / 8.  So the last round's worth of
// workgroups (256 / (L x 256-tile pairs) slices) is cut four times finer; a slab is written per slice and tile whatever its length, so
// the finer tail adds about one round of slabs.  A function of (N, M, L) only: results repeat.
struct agpl_slices {
    int chunk, small; // points per big / small slice (multiples of 32)
    int nbig, ns;     // big slices; slices in all
};
#ifndef AGPL_SLICE_TAIL_DIV
#define AGPL_SLICE_TAIL_DIV 4 // (1: no finer tail -- the slices of rounds 1-5)
#endif
inline agpl_slices agpl_slice_plan(int64_t N, int M, int L) {
    agpl_slices o;
    o.chunk = agpl_slice_points(N, M, L);
    o.small = o.chunk / AGPL_SLICE_TAIL_DIV;
    const int64_t nfull = (N + o.chunk - 1) / o.chunk;
    const int64_t nb2 = (M + 255) / 256, wg_per_slice = (int64_t)L * nb2 * (nb2 + 1) / 2;
    int64_t tail = (256 + wg_per_slice - 1) / wg_per_slice; // big slices that make one round of workgroups
    // measured (profiles/NOTES_r06.md): worth 0.05-0.11 ms per launch at M >= 512 and for launches of up to two rounds at any M;
    // a longer launch of diagonal tiles only (M = 256: C4, ten latents, 4.8 rounds) ran 0.08 ms slower with it
    if (AGPL_SLICE_TAIL_DIV == 1 || (M < 512 && nfull * wg_per_slice > 512)) tail = 0;
    o.nbig = (int)(nfull > tail ? nfull - tail : 0);
    const int64_t rest = N - (int64_t)o.nbig * o.chunk;
    o.ns = o.nbig + (int)((rest + o.small - 1) / o.small);
    return o;
}
// points [nbeg, nend) of slice s
__host__ __device__ inline void agpl_slice_range(int s, int chunk, int nbig, int small, int64_t N, int64_t &nbeg, int64_t &nend) {
    if (s < nbig) {
        nbeg = (int64_t)s * chunk;
        nend = nbeg + chunk;
    } else {
        nbeg = (int64_t)nbig * chunk + (int64_t)(s - nbig) * small;
        nend = nbeg + small;
    }
    if (nend > N) nend = N;
}

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
