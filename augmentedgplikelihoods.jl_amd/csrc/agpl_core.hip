// agpl_core.hip -- context lifetime, stream binding, scratch memory, likelihood descriptor upload.
#include <math.h>

#include "agpl_common.h"

extern "C" void agpl_update_release(agpl_ctx *ctx); // agpl_update.hip (destroys the rocBLAS handle)

extern "C" int32_t agpl_version(void) { return AGPL_VERSION; }

extern "C" int32_t agpl_ctx_create(agpl_ctx **out, int32_t device_id, uint64_t seed) {
    if (!out) return AGPL_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AGPL_ERR_HIP;
    if (device_id < 0 || device_id >= ndev) return AGPL_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device_id) != hipSuccess) return AGPL_ERR_HIP;
    agpl_ctx *ctx = new (std::nothrow) agpl_ctx();
    if (!ctx) return AGPL_ERR_OUT_OF_MEMORY;
    ctx->device = device_id;
    ctx->seed = seed;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return AGPL_ERR_HIP;
    }
    ctx->own_stream = true;
    *out = ctx;
    return AGPL_OK;
}

extern "C" int32_t agpl_ctx_destroy(agpl_ctx *ctx) {
    if (!ctx) return AGPL_OK;
    if (ctx->live_plans > 0) // a plan enqueues on, and reports through, its context: destroy the plans first (include/agpl.h)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "%d plan(s) created on this context are still alive: agpl_plan_destroy them first",
                  ctx->live_plans);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    agpl_update_release(ctx);
    for (int w = 0; w < 4; ++w)
        for (auto &pr : ctx->ev[w]) ctx->ev_pool.push_back(pr);
    for (auto &pr : ctx->ev_pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->ws2) (void)hipFree(ctx->ws2);
    if (ctx->pg_retry) (void)hipFree(ctx->pg_retry);
    if (ctx->red_cnt) (void)hipFree(ctx->red_cnt);
    if (ctx->elbo_part) (void)hipFree(ctx->elbo_part);
    if (ctx->logtheta_dev) (void)hipFree(ctx->logtheta_dev);
    if (ctx->pend_host) (void)hipHostFree(ctx->pend_host);
    if (ctx->pend_ev) (void)hipEventDestroy(ctx->pend_ev);
    for (int i = 0; i < 2; ++i)
        if (ctx->aux_ev[i]) (void)hipEventDestroy(ctx->aux_ev[i]);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return AGPL_OK;
}

extern "C" int32_t agpl_ctx_set_stream(agpl_ctx *ctx, void *hip_stream) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    AGPL_HIP(ctx, hipSetDevice(ctx->device));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) AGPL_HIP(ctx, hipStreamDestroy(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream; // NULL = the device's default (null) stream
    ctx->own_stream = false;
    return AGPL_OK;
}

extern "C" int32_t agpl_ctx_set_seed(agpl_ctx *ctx, uint64_t seed) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    ctx->seed = seed;
    return AGPL_OK;
}

extern "C" int32_t agpl_ctx_set_point_offset(agpl_ctx *ctx, int64_t i0) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (i0 < 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "point offset %lld < 0", (long long)i0);
    ctx->point_offset = i0;
    return AGPL_OK;
}

extern "C" int32_t agpl_ctx_synchronize(agpl_ctx *ctx) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return agpl_pending_resolve(ctx); // also the place where a deferred factorisation failure surfaces
}

extern "C" const char *agpl_last_error(const agpl_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int32_t agpl_ws_reserve(agpl_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return AGPL_OK;
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) AGPL_HIP(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    if (hipMalloc(&ctx->ws, bytes) != hipSuccess)
        AGPL_FAIL(ctx, AGPL_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) for the workspace failed", bytes);
    ctx->ws_bytes = bytes;
    return AGPL_OK;
}

// Bytes 8192..16383 hold words that are ZERO between launches: the marginal kernel's item queues (8192, eight words) and the
// factor kernel's hand-off flags (8448, 4 words per latent).  The kernels that use them leave them zero again (no memset per
// sweep); a fresh allocation starts zeroed.
int32_t agpl_ws2_reserve(agpl_ctx *ctx, size_t bytes) {
    if (bytes < 16384) bytes = 16384;
    if (bytes <= ctx->ws2_bytes) return AGPL_OK;
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    void *grown = nullptr;
    if (hipMalloc(&grown, bytes) != hipSuccess)
        AGPL_FAIL(ctx, AGPL_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) for the small workspace failed", bytes);
    // the first 16 KB carry state ACROSS calls (a sweep's bad-gamma word waits there for the update that reports it, and that
    // update may be the call that grows this allocation): they move with it; a first allocation starts zeroed
    if (ctx->ws2) {
        AGPL_HIP(ctx, hipMemcpyAsync(grown, ctx->ws2, 16384, hipMemcpyDeviceToDevice, ctx->stream));
        AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        AGPL_HIP(ctx, hipFree(ctx->ws2));
    } else {
        AGPL_HIP(ctx, hipMemsetAsync(grown, 0, 16384, ctx->stream));
    }
    ctx->ws2 = grown;
    ctx->ws2_bytes = bytes;
    return AGPL_OK;
}

// The list of points a PG(1) kernel hands to its retry kernel (aux_sample_pg1_retry_kernel, agpl_ops.hip): two counter words that
// are zero between launches, then one 32-bit point index per entry -- every point can end up there (|f| >= 16 has no fitted
// branch mass), so the list holds n.
int32_t agpl_pg_retry_reserve(agpl_ctx *ctx, int64_t n) {
    if ((size_t)n <= ctx->pg_retry_entries) return AGPL_OK;
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pg_retry) AGPL_HIP(ctx, hipFree(ctx->pg_retry));
    ctx->pg_retry = nullptr;
    ctx->pg_retry_entries = 0;
    const size_t bytes = sizeof(unsigned) * ((size_t)n + 2);
    if (hipMalloc((void **)&ctx->pg_retry, bytes) != hipSuccess)
        AGPL_FAIL(ctx, AGPL_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) for the PG retry list failed", bytes);
    AGPL_HIP(ctx, hipMemsetAsync(ctx->pg_retry, 0, 2 * sizeof(unsigned), ctx->stream));
    ctx->pg_retry_entries = (size_t)n;
    return AGPL_OK;
}

// Arrival counters of reduce_slab_kernel (agpl_mfma.hip): the workgroup that finds its counter at ngroup - 1 sums the group partials
// and sets the counter back to zero, so that the words are zero between launches; a fresh allocation is zeroed once.
int32_t agpl_red_cnt_reserve(agpl_ctx *ctx, int64_t n) {
    if ((size_t)n <= ctx->red_cnt_entries) return AGPL_OK;
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->red_cnt) AGPL_HIP(ctx, hipFree(ctx->red_cnt));
    ctx->red_cnt = nullptr;
    ctx->red_cnt_entries = 0;
    const size_t bytes = sizeof(unsigned) * (size_t)n;
    if (hipMalloc((void **)&ctx->red_cnt, bytes) != hipSuccess)
        AGPL_FAIL(ctx, AGPL_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) for the reduction counters failed", bytes);
    AGPL_HIP(ctx, hipMemsetAsync(ctx->red_cnt, 0, bytes, ctx->stream));
    ctx->red_cnt_entries = (size_t)n;
    return AGPL_OK;
}

int32_t agpl_lik_to_device(agpl_ctx *ctx, const agpl_lik_desc *lik, agpl_lik_dev *out) {
    if (!lik) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null likelihood descriptor");
    if (lik->kind < AGPL_LIK_BERNOULLI_LOGISTIC || lik->kind > AGPL_LIK_HETEROGAUSS)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "unknown likelihood kind %d", lik->kind);
    out->kind = lik->kind;
    out->nlatent = lik->nlatent;
    for (int i = 0; i < 4; ++i) out->p[i] = lik->p[i];
    out->logtheta = nullptr;
    out->sum_theta = 0.0;
    out->cat_const = 0.0;
    const bool cat = lik->kind == AGPL_LIK_CATEGORICAL || lik->kind == AGPL_LIK_CATEGORICAL_BIJ;
    const int want_l = lik->kind == AGPL_LIK_HETEROGAUSS ? 2 : 1;
    if (!cat && lik->nlatent != want_l)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "nlatent = %d, expected %d", lik->nlatent, want_l);
    if (cat) {
        if (lik->nlatent < 1 || lik->nlatent > 64 || !lik->logtheta)
            AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "categorical needs 1 <= nlatent <= 64 and logtheta");
        const int K = lik->nlatent + (lik->kind == AGPL_LIK_CATEGORICAL_BIJ ? 1 : 0);
        if (K > ctx->logtheta_cap) {
            AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->logtheta_dev) AGPL_HIP(ctx, hipFree(ctx->logtheta_dev));
            ctx->logtheta_dev = nullptr;
            AGPL_HIP(ctx, hipMalloc((void **)&ctx->logtheta_dev, sizeof(double) * 128));
            ctx->logtheta_cap = 128;
        }
        if (ctx->logtheta_n != K || memcmp(ctx->logtheta_host, lik->logtheta, sizeof(double) * K) != 0) {
            // synchronous small copy: the descriptor's host memory need not outlive the call
            AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
            AGPL_HIP(ctx, hipMemcpy(ctx->logtheta_dev, lik->logtheta, sizeof(double) * K, hipMemcpyHostToDevice));
            memcpy(ctx->logtheta_host, lik->logtheta, sizeof(double) * K);
            ctx->logtheta_n = K;
        }
        out->logtheta = ctx->logtheta_dev;
        double s = 0.0;
        for (int k = 0; k < lik->nlatent; ++k) s += exp(lik->logtheta[k]); // categorical.jl:16-20
        if (lik->kind == AGPL_LIK_CATEGORICAL_BIJ) {
            out->cat_const = exp(lik->logtheta[lik->nlatent]) * 0.5; // categorical.jl:12-14
            s += out->cat_const;
        }
        out->sum_theta = s;
    }
    if (lik->kind == AGPL_LIK_NEGBINOMIAL && !(lik->p[0] > 0.0))
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "NegBinomial failures r must be > 0");
    if (lik->kind == AGPL_LIK_STUDENTT && !(lik->p[0] > 0.0 && lik->p[1] > 0.0))
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "StudentT needs nu > 0 and sigma > 0");
    return AGPL_OK;
}

// ---- optional kernel timing -------------------------------------------------------------------------
int32_t agpl_timing_begin(agpl_ctx *ctx, int which) {
    if (!ctx->timing) return AGPL_OK;
    std::pair<hipEvent_t, hipEvent_t> pr;
    if (!ctx->ev_pool.empty()) {
        pr = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
    } else {
        AGPL_HIP(ctx, hipEventCreate(&pr.first));
        AGPL_HIP(ctx, hipEventCreate(&pr.second));
    }
    AGPL_HIP(ctx, hipEventRecord(pr.first, ctx->stream));
    // the stop event is recorded here as well (agpl_timing_end records it again, the later record counts): a caller that
    // fails between begin and end leaves a well-formed pair of ~0 ms instead of an event that was never recorded
    AGPL_HIP(ctx, hipEventRecord(pr.second, ctx->stream));
    ctx->ev[which].push_back(pr);
    return AGPL_OK;
}
int32_t agpl_timing_end(agpl_ctx *ctx, int which) {
    if (!ctx->timing) return AGPL_OK;
    AGPL_HIP(ctx, hipEventRecord(ctx->ev[which].back().second, ctx->stream));
    return AGPL_OK;
}
extern "C" int32_t agpl_timing(agpl_ctx *ctx, int32_t which, double *total_ms_host, int64_t *launches_host) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (which == -1 || which == -2) { // switch the event pairs around the hot kernels on / off
        ctx->timing = which == -1;
        return AGPL_OK;
    }
    if (which < 0 || which > 3 || !total_ms_host || !launches_host) return AGPL_ERR_INVALID_ARGUMENT;
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (auto &pr : ctx->ev[which]) {
        float ms = 0.f;
        AGPL_HIP(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
        tot += ms;
        ctx->ev_pool.push_back(pr);
    }
    *total_ms_host = tot;
    *launches_host = (int64_t)ctx->ev[which].size();
    ctx->ev[which].clear();
    return AGPL_OK;
}

// ------------------------------------------------------------------------------------------------
// agpl_allreduce_nat: the one exchange step of an N-sharded sweep for hosts that drive RCCL themselves (the Python
// host uses torch.distributed, whose "nccl" backend is the same RCCL).  librccl is resolved at the first call
// (dlopen by SONAME: the instance already in the process if there is one), so libagpl.so carries no load-time
// dependency on it.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>

extern "C" int32_t agpl_allreduce_nat(agpl_ctx *ctx, void *rccl_comm, double *buf, int64_t count) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (!rccl_comm || !buf || count < 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    if (count == 0) return AGPL_OK;
    typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    static allreduce_fn fn = nullptr;
    if (!fn) {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED, "librccl is not available: %s", dlerror());
        fn = (allreduce_fn)dlsym(h, "ncclAllReduce");
        if (!fn) AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED, "ncclAllReduce not found in librccl");
    }
    // ncclFloat64 = 8, ncclSum = 0 (rccl.h); in place, on the context's stream
    const int rc = fn(buf, buf, (size_t)count, 8, 0, rccl_comm, ctx->stream);
    if (rc != 0) AGPL_FAIL(ctx, AGPL_ERR_HIP, "ncclAllReduce failed: ncclResult_t %d", rc);
    return AGPL_OK;
}
