// agpl_plan.hip -- the plan: everything that is static about one data set on one GPU, built once, and the sweep entry points on it.
//
// Rounds 1-3 grew four generations of sweep entry points (agpl_cavi_pass, _split, _factor_split, _factor_image; the same for
// the Gibbs pass and the accumulation), each with its own list of images and work arrays a host had to build in the right order
// and keep consistent; round 5 removed all but the float32 one.  A plan owns the images:
//   * the marginal image of Phi (split-float16, blocks by 128-point tile: the B operand of U Phi, agpl_split.hip) and the
//     accumulate image (point-major: both operands of Phi Diag(gamma) Phi', agpl_syrk.hip), BOTH scaled by the same 2^e chosen
//     from max |Phi| -- one domain (any finite feature range that can be scaled into float16: max |Phi| in 2^-24 .. 2^44, see
//     agpl_image_scale_exp) instead
//     of an unscaled marginal image that refused |x| >= 65504 and lost precision below 2^-14 beside a self-scaling accumulate image;
//   * a copy of the Nystrom residual d_i = k_ii - |phi_i|^2 (agpl_feature_residual);
//   * q(v) in factor form: U = chol(I + G)^-1 (float64, and split-float16 images of 2^15 U: |U| <= 1 always, so the scale is
//     fixed and the images keep normal float16 parts down to |U| ~ 2^-29), v = U (g + eta0), log det(I + G);
// and picks the kernels by shape.  After agpl_plan_create the float32 features are not read again: the CAVI sweep, the marginals
// and the Gibbs pass (projection phi_i' v from the accumulate image) read the images only.
// Reference: the loop bodies of examples/bernoulli/script.jl:29-39 (cavi!) and :76-87 (gibbs_sample) in the sparse form of
// docs/src/index.md:154-163; the ELBO pieces are those of aug_elbo, script.jl:65-70.
#include "agpl_common.h"

// internals of the other translation units
int32_t agpl_feature_range_check(agpl_ctx *ctx, int64_t N, int32_t M, const float *Phi, float limit, const char *what,
                                 unsigned *max_bits_out);                                                       // agpl_syrk.hip
int32_t agpl_image_scale_exp(agpl_ctx *ctx, unsigned hmx, int *eA_out);                                         // agpl_syrk.hip
int32_t agpl_accumulate_image_build(agpl_ctx *ctx, int64_t N, int32_t M, int32_t Msrc, const float *Phi, int eA, unsigned hmx,
                                    void *image_out);                                                           // agpl_syrk.hip
int32_t agpl_split_features_build(agpl_ctx *ctx, int64_t N, int32_t M, int32_t Msrc, const float *Phi, float scale, void *Phi_hi,
                                  void *Phi_lo);                                                                // agpl_split.hip
int32_t agpl_marginals_factor_internal(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const void *Phi_hi, const void *Phi_lo,
                                       const float *resid, const float *mu0, const void *U_hi, const void *U_lo, const float *v,
                                       float *mu_out, float *var_out, int image_scale_exp);                     // agpl_split.hip
int32_t agpl_cavi_pass_factor_internal(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M, const float *Phi,
                                       const void *Phi_hi, const void *Phi_lo, const void *acc_image, const float *resid,
                                       const float *mu0, const void *y, const void *U_hi, const void *U_lo, const float *v,
                                       double *G_out, double *g_out, float *c_out, float *gamma_out, float *beta_out,
                                       int image_scale_exp, double *elbo_terms_out);                            // agpl_update.hip
int32_t agpl_gaussian_factor_async_scaled(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                          const double *eta0, double *A_work, double *v_out, float *v32_out, void *U_hi,
                                          void *U_lo, double *logdet_out, int u_scale_exp);                         // agpl_update.hip
int32_t agpl_pack_factor_split_info(agpl_ctx *ctx, int32_t M, int32_t L, const double *A, void *U_hi, void *U_lo,
                                    const int *info, int *info_host, int ninfo, int u_scale_exp);                  // agpl_split.hip
int32_t agpl_gibbs_pass_internal(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M, const float *Phi,
                                 const void *acc_image, bool force_split, const float *kdiag, const float *mu0, const void *y,
                                 const double *v, uint32_t sweep, double *G_out, double *g_out, double *f_out, double *omega_out,
                                 int64_t *n_out, uint32_t *nuni_out);                                           // agpl_update.hip

// Feature counts (round 6): the caller's M is ANY positive count; the plan works on Mp = M rounded up to a multiple of 256 -- the images
// carry zero features M .. Mp - 1, so G and g have zero rows / columns there, I + G is the identity there and U = chol(I + G)^-1,
// v are the caller's in their leading M x M / M block (the rest: identity / zero).  The caller's arrays (G, g, eta0, the Gibbs
// draw v) are M-sized; for M != Mp they pass through the plan's Mp-sized staging copies (two small kernels per call).
int32_t agpl_pad_natural(agpl_ctx *ctx, int L, int Mc, int Mp, const double *G, const double *g, const double *e, const double *v,
                         double *Gp, double *gp, double *ep, double *vp);                                       // agpl_update.hip
int32_t agpl_unpad_natural(agpl_ctx *ctx, int L, int Mc, int Mp, const double *Gp, const double *gp, double *G, double *g); // agpl_update.hip

struct agpl_plan {
    agpl_ctx *ctx = nullptr;
    int64_t N = 0;
    int32_t M = 0, L = 0;  // M: the padded count Mp every kernel works on
    int32_t Mc = 0;        // the caller's feature count (<= M)
    double *Gp = nullptr, *gp = nullptr, *eta0p = nullptr, *vp = nullptr; // staging at Mp (Mc != M only)
    uint32_t flags = 0;
    int scale_exp = 0;     // both images hold 2^scale_exp Phi
    char *base = nullptr;  // the plan's device memory
    size_t bytes = 0;
    bool own = false;      // allocated here (storage == NULL at creation)
    // carved out of base
    void *Phi_hi = nullptr, *Phi_lo = nullptr, *Phi_acc = nullptr;
    float *resid = nullptr;
    void *U_hi = nullptr, *U_lo = nullptr;
    double *A_work = nullptr; // [L, M, M]: column-major lower triangle = U
    double *v = nullptr;      // [L, M]
    float *v32 = nullptr;     // [L, M]
    double *logdet = nullptr; // [L] log det(I + G)
    double *klpart = nullptr; // [L][kKlWaves][2] partial sums of the Gaussian KL
};

namespace {

constexpr int kKlBlocks = 16, kKlWaves = kKlBlocks * 4;
constexpr int kUExp = 15; // the plan's U images carry 2^15 U: |U[a][b]| <= 1 always (I + G >= I), so this never overflows float16

struct PlanLayout {
    size_t hi, lo, acc, resid, uhi, ulo, awork, v, v32, logdet, klpart, stage, total;
};
inline int32_t plan_padded(int32_t M) { return (M + 255) / 256 * 256; }
// M: the padded count; Mc: the caller's
PlanLayout plan_layout(int64_t N, int32_t M, int32_t Mc, int32_t L, uint32_t flags) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    PlanLayout o;
    // one of hi / lo; a plan without the marginal image (AGPL_PLAN_NO_MARGINALS: Gibbs sweeps only) keeps none
    const size_t img = (flags & AGPL_PLAN_NO_MARGINALS) ? 0 : (size_t)agpl_split_features_bytes(N, M);
    o.hi = 0;
    o.lo = al(o.hi + img);
    o.acc = al(o.lo + img);
    o.resid = al(o.acc + (size_t)agpl_accumulate_image_bytes(N, M));
    o.uhi = al(o.resid + sizeof(float) * (size_t)N);
    o.ulo = al(o.uhi + sizeof(_Float16) * (size_t)L * M * M);
    o.awork = al(o.ulo + sizeof(_Float16) * (size_t)L * M * M);
    o.v = al(o.awork + sizeof(double) * (size_t)L * M * M);
    o.v32 = al(o.v + sizeof(double) * (size_t)L * M);
    o.logdet = al(o.v32 + sizeof(float) * (size_t)L * M);
    o.klpart = al(o.logdet + sizeof(double) * (size_t)L);
    o.stage = al(o.klpart + sizeof(double) * (size_t)L * kKlWaves * 2);
    // staging of the caller's M-sized natural parameters at the padded size: G [L, M, M], g, eta0, v [L, M] each
    o.total = Mc == M ? o.stage : al(o.stage + sizeof(double) * (size_t)L * ((size_t)M * M + 3 * (size_t)M));
    return o;
}

// U = I, v = 0 (S = I, m = 0: examples/bernoulli/script.jl:41-42), log det = 0, as data and as images
__global__ void plan_identity_kernel(int M, int L, double *__restrict__ A, double *__restrict__ v, float *__restrict__ v32,
                                     double *__restrict__ logdet) {
    const int64_t total = (int64_t)L * M * M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i % ((int64_t)M * M);
        A[i] = (r / M == r % M) ? 1.0 : 0.0;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)L * M; i += (int64_t)gridDim.x * blockDim.x) {
        v[i] = 0.0;
        v32[i] = 0.f;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < L) logdet[threadIdx.x] = 0.0;
}

// The plan's copy of d_i = k_ii - |phi_i|^2.  d_i >= 0 in exact arithmetic (Nystrom residual of a positive semi-definite
// kernel); a float32 evaluation rounds below zero by ~1e-7 (k_ii + |phi_i|^2), which is a NEGATIVE marginal variance once the
// posterior is tighter than that: such round-off is clamped to 0.  Anything more negative than 1e-5 (|d_i| + |phi_i|^2) is not
// round-off (a wrong kdiag / resid array, a wrong sign): its smallest index goes to *bad.  One wave per point at a
// time, |phi_i|^2 from the float32 features (read once more, at plan creation only).  A NaN stays: the sweep reports it.
__global__ __launch_bounds__(256) void plan_residual_kernel(int64_t N, int M, const float *__restrict__ Phi,
                                                            const float *__restrict__ in, float *__restrict__ out,
                                                            unsigned long long *__restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = wave; i < N; i += nwaves) {
        float s = 0.f;
        if (!(M & 3)) {
            const float4 *row = (const float4 *)(Phi + i * M);
            for (int q = lane; q < M / 4; q += 64) {
                const float4 x = row[q];
                s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            }
        } else { // (a feature count that is not a multiple of 4: the rows are not 16-byte aligned)
            const float *row = Phi + i * M;
            for (int q = lane; q < M; q += 64) s += row[q] * row[q];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const float d = in[i];
            if (d < -1e-5f * (fabsf(d) + s)) atomicMin(bad, (unsigned long long)i);
            out[i] = d > 0.f ? d : (d == d ? 0.f : d);
        }
    }
}

// KL(q(v_l) || N(0, I)) = (tr S + m'm - M + log det(I + G)) / 2 straight from the inverse factor: S = U'U, m = U'v, i.e.
// tr S = sum_{a >= b} U[a][b]^2 and m_b = sum_{a >= b} U[a][b] v_a, with U[a][b] = A[b M + a] (column-major lower triangle).
// grid (kKlBlocks, L) x 4 waves: wave w of block k takes the columns b = 4 k + w (mod kKlWaves) in ascending order, the lanes
// walk down a column (contiguous rows), one shuffle tree per column; each wave leaves (its tr S share, its m'm share).
__global__ __launch_bounds__(256) void plan_kl_part_kernel(int M, const double *__restrict__ A, const double *__restrict__ v,
                                                           double *__restrict__ part) {
    const int l = blockIdx.y, lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const double *Al = A + (int64_t)l * M * M, *vl = v + (int64_t)l * M;
    double trs = 0.0, mm = 0.0;
    for (int b = wv; b < M; b += kKlWaves) {
        double t = 0.0, mb = 0.0;
        for (int a = b + lane; a < M; a += 64) {
            const double u = Al[(int64_t)b * M + a];
            t += u * u;
            mb += u * vl[a];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            t += __shfl_xor(t, o);
            mb += __shfl_xor(mb, o);
        }
        trs += t;
        mm += mb * mb;
    }
    if (lane == 0) {
        part[((int64_t)l * kKlWaves + wv) * 2 + 0] = trs;
        part[((int64_t)l * kKlWaves + wv) * 2 + 1] = mm;
    }
}
// fixed-order sum of the shares over waves and latents; out = sum_l (tr S_l + m_l'm_l - M + log det(I + G_l)) / 2
__global__ void plan_kl_final_kernel(int M, int L, const double *__restrict__ part, const double *__restrict__ logdet,
                                     double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double kl = 0.0;
    for (int l = 0; l < L; ++l) {
        double trs = 0.0, mm = 0.0;
        for (int w = 0; w < kKlWaves; ++w) {
            trs += part[((int64_t)l * kKlWaves + w) * 2 + 0];
            mm += part[((int64_t)l * kKlWaves + w) * 2 + 1];
        }
        kl += 0.5 * (trs + mm - (double)M + logdet[l]);
    }
    *out = kl;
}

int32_t plan_check(const agpl_plan *p, bool needs_marginals = false) {
    if (!p || !p->ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (needs_marginals && (p->flags & AGPL_PLAN_NO_MARGINALS))
        AGPL_FAIL(p->ctx, AGPL_ERR_INVALID_ARGUMENT, "this plan was created without the marginal image (AGPL_PLAN_NO_MARGINALS)");
    return AGPL_OK;
}

} // namespace

extern "C" int64_t agpl_plan_bytes(int64_t N, int32_t M, int32_t L, uint32_t flags) {
    if (N <= 0 || M <= 0 || M > (1 << 20) || L <= 0 || L > 64 || (flags & ~(uint32_t)AGPL_PLAN_NO_MARGINALS)) return 0;
    return (int64_t)plan_layout(N, plan_padded(M), M, L, flags).total;
}

extern "C" int32_t agpl_plan_create(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi, const float *resid,
                                    uint32_t flags, void *storage, agpl_plan **plan_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (!plan_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null plan_out");
    *plan_out = nullptr;
    if (N <= 0 || M <= 0 || M > (1 << 20) || L <= 0 || L > 64)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes N=%lld M=%d L=%d", (long long)N, M, L);
    const int32_t Mc = M; // the caller's feature count; every kernel below works on the padded one
    M = plan_padded(Mc);
    if (!Phi || !resid) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    if (flags & ~(uint32_t)AGPL_PLAN_NO_MARGINALS) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "unknown plan flags 0x%x", flags);
    // one range check and ONE scale for both images (a non-finite feature is AGPL_ERR_DOMAIN with its position)
    unsigned hmx = 0;
    int32_t rc = agpl_feature_range_check(ctx, N, Mc, Phi, __builtin_inff(), "the split-float16 images", &hmx);
    if (rc) return rc;
    int e = 0;
    rc = agpl_image_scale_exp(ctx, hmx, &e);
    if (rc) return rc;
    const PlanLayout lo = plan_layout(N, M, Mc, L, flags);
    agpl_plan *p = new agpl_plan;
    p->flags = flags;
    p->ctx = ctx;
    p->N = N;
    p->M = M;
    p->Mc = Mc;
    p->L = L;
    p->scale_exp = e;
    p->bytes = lo.total;
    if (storage) {
        p->base = (char *)storage;
    } else {
        if (hipMalloc((void **)&p->base, lo.total) != hipSuccess) {
            delete p;
            AGPL_FAIL(ctx, AGPL_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) for the plan failed", lo.total);
        }
        p->own = true;
    }
    p->Phi_hi = p->base + lo.hi;
    p->Phi_lo = p->base + lo.lo;
    p->Phi_acc = p->base + lo.acc;
    p->resid = (float *)(p->base + lo.resid);
    p->U_hi = p->base + lo.uhi;
    p->U_lo = p->base + lo.ulo;
    p->A_work = (double *)(p->base + lo.awork);
    p->v = (double *)(p->base + lo.v);
    p->v32 = (float *)(p->base + lo.v32);
    p->logdet = (double *)(p->base + lo.logdet);
    p->klpart = (double *)(p->base + lo.klpart);
    if (Mc != M) {
        p->Gp = (double *)(p->base + lo.stage);
        p->gp = p->Gp + (size_t)L * M * M;
        p->eta0p = p->gp + (size_t)L * M;
        p->vp = p->eta0p + (size_t)L * M;
    }
    auto fail = [&](int32_t code) {
        if (p->own) (void)hipFree(p->base);
        delete p;
        return code;
    };
    if (!(flags & AGPL_PLAN_NO_MARGINALS)) {
        rc = agpl_split_features_build(ctx, N, M, Mc, Phi, ldexpf(1.f, e), p->Phi_hi, p->Phi_lo);
        if (rc) return fail(rc);
    }
    rc = agpl_accumulate_image_build(ctx, N, M, Mc, Phi, e, hmx, p->Phi_acc);
    if (rc) return fail(rc);
    rc = agpl_ws2_reserve(ctx, 16384);
    if (rc) return fail(rc);
    unsigned long long *bad = (unsigned long long *)((char *)ctx->ws2 + 32); // (bytes 8..63 of the small scratch are nobody's)
    if (hipMemsetAsync(bad, 0xff, sizeof(*bad), ctx->stream) != hipSuccess) return fail(AGPL_ERR_HIP);
    plan_residual_kernel<<<2048, 256, 0, ctx->stream>>>(N, Mc, Phi, resid, p->resid, bad);
    if (hipGetLastError() != hipSuccess) return fail(AGPL_ERR_HIP);
    // q(v) = N(0, I) to start from (script.jl:41-42)
    plan_identity_kernel<<<1024, 256, 0, ctx->stream>>>(M, L, p->A_work, p->v, p->v32, p->logdet);
    if (hipGetLastError() != hipSuccess) return fail(AGPL_ERR_HIP);
    rc = agpl_pack_factor_split_info(ctx, M, L, p->A_work, p->U_hi, p->U_lo, nullptr, nullptr, 0, kUExp);
    if (rc) return fail(rc);
    // The image builds and the residual copy above READ the caller's Phi and resid: wait for them, so that "Phi is not read
    // after agpl_plan_create returns" holds for a host that frees it stream-ordered on another stream.  One
    // synchronisation per data set.
    unsigned long long hbad = ~0ull;
    if (hipMemcpyAsync(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "agpl_plan_create: waiting for the image builds failed: %s",
                 hipGetErrorString(hipGetLastError()));
        return fail(AGPL_ERR_HIP);
    }
    if (hbad != ~0ull) {
        snprintf(ctx->err, sizeof(ctx->err),
                 "resid[%llu] is negative beyond the float32 round-off of k_ii - |phi_i|^2 (tolerance 1e-5 (|d| + |phi|^2)): "
                 "not a Nystrom residual (agpl_feature_residual)", hbad);
        return fail(AGPL_ERR_DOMAIN);
    }
    ctx->live_plans += 1;
    *plan_out = p;
    return AGPL_OK;
}

extern "C" int32_t agpl_plan_destroy(agpl_plan *p) {
    if (!p) return AGPL_OK;
    // Lifetime rule (include/agpl.h): a plan is destroyed BEFORE its context; agpl_ctx_destroy refuses while plans are alive.
    if (p->ctx) {
        (void)hipStreamSynchronize(p->ctx->stream);
        // the accumulation skips the header check of the image it validated last: forget it, the memory may be reused
        if (p->ctx->checked_image == p->Phi_acc) p->ctx->checked_image = nullptr;
        p->ctx->live_plans -= 1;
    }
    if (p->own && p->base) (void)hipFree(p->base);
    delete p;
    return AGPL_OK;
}

extern "C" int32_t agpl_plan_info(const agpl_plan *p, int64_t *N, int32_t *M, int32_t *L, int32_t *scale_exp, int64_t *bytes) {
    if (!p) return AGPL_ERR_INVALID_ARGUMENT;
    if (N) *N = p->N;
    if (M) *M = p->Mc; // (the caller's count; the state arrays of agpl_plan_state are sized by M rounded up to a multiple of 256)
    if (L) *L = p->L;
    if (scale_exp) *scale_exp = p->scale_exp;
    if (bytes) *bytes = (int64_t)p->bytes;
    return AGPL_OK;
}

extern "C" int32_t agpl_plan_state(const agpl_plan *p, double **U_out, double **v_out, void **U_hi_out, void **U_lo_out,
                                   float **v32_out, double **logdet_out, const float **resid_out) {
    if (!p) return AGPL_ERR_INVALID_ARGUMENT;
    if (U_out) *U_out = p->A_work;
    if (v_out) *v_out = p->v;
    if (U_hi_out) *U_hi_out = p->U_hi;
    if (U_lo_out) *U_lo_out = p->U_lo;
    if (v32_out) *v32_out = p->v32;
    if (logdet_out) *logdet_out = p->logdet;
    if (resid_out) *resid_out = p->resid;
    return AGPL_OK;
}

extern "C" int32_t agpl_cavi_pass_plan(agpl_plan *p, const agpl_lik_desc *lik, const float *mu0, const void *y, double *G_out,
                                       double *g_out, float *c_out, float *gamma_out, float *beta_out, double *elbo_terms_out) {
    int32_t rc = plan_check(p, true);
    if (rc) return rc;
    if (!lik) AGPL_FAIL(p->ctx, AGPL_ERR_INVALID_ARGUMENT, "null likelihood descriptor");
    if (lik->nlatent != p->L)
        AGPL_FAIL(p->ctx, AGPL_ERR_INVALID_ARGUMENT, "the likelihood has %d latents, the plan was created for %d", lik->nlatent, p->L);
    if (!G_out || !g_out) AGPL_FAIL(p->ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    const bool pad = p->Mc != p->M;
    rc = agpl_cavi_pass_factor_internal(p->ctx, lik, p->N, p->M, nullptr, p->Phi_hi, p->Phi_lo, p->Phi_acc, p->resid, mu0, y,
                                        p->U_hi, p->U_lo, p->v32, pad ? p->Gp : G_out, pad ? p->gp : g_out, c_out, gamma_out,
                                        beta_out, p->scale_exp + kUExp, elbo_terms_out);
    if (rc || !pad) return rc;
    return agpl_unpad_natural(p->ctx, p->L, p->Mc, p->M, p->Gp, p->gp, G_out, g_out);
}

extern "C" int32_t agpl_plan_update(agpl_plan *p, const double *G, const double *g, const double *eta0, double *kl_out) {
    int32_t rc = plan_check(p);
    if (rc) return rc;
    agpl_ctx *ctx = p->ctx;
    if (!G || !g) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    if (p->Mc != p->M) { // the caller's M-sized natural parameters at the padded size (zero beyond M: I + G is the identity there)
        rc = agpl_pad_natural(ctx, p->L, p->Mc, p->M, G, g, eta0, nullptr, p->Gp, p->gp, eta0 ? p->eta0p : nullptr, nullptr);
        if (rc) return rc;
        G = p->Gp, g = p->gp, eta0 = eta0 ? p->eta0p : nullptr;
    }
    rc = agpl_gaussian_factor_async_scaled(ctx, p->M, p->L, G, g, eta0, p->A_work, p->v, p->v32, p->U_hi, p->U_lo, p->logdet, kUExp);
    if (rc) return rc;
    if (kl_out) {
        plan_kl_part_kernel<<<dim3(kKlBlocks, (unsigned)p->L), 256, 0, ctx->stream>>>(p->M, p->A_work, p->v, p->klpart);
        AGPL_LAUNCH_CHECK(ctx);
        plan_kl_final_kernel<<<1, 64, 0, ctx->stream>>>(p->M, p->L, p->klpart, p->logdet, kl_out);
        AGPL_LAUNCH_CHECK(ctx);
    }
    return AGPL_OK;
}

extern "C" int32_t agpl_marginals_plan(agpl_plan *p, const float *mu0, float *mu_out, float *var_out) {
    int32_t rc = plan_check(p, true);
    if (rc) return rc;
    return agpl_marginals_factor_internal(p->ctx, p->N, p->M, p->L, p->Phi_hi, p->Phi_lo, p->resid, mu0, p->U_hi, p->U_lo, p->v32,
                                          mu_out, var_out, p->scale_exp + kUExp);
}

extern "C" int32_t agpl_gibbs_pass_plan(agpl_plan *p, const agpl_lik_desc *lik, const float *mu0, const void *y,
                                        const double *v, uint32_t sweep, double *G_out, double *g_out, double *f_out,
                                        double *omega_out, int64_t *n_out, uint32_t *nuni_out) {
    int32_t rc = plan_check(p);
    if (rc) return rc;
    if (!lik) AGPL_FAIL(p->ctx, AGPL_ERR_INVALID_ARGUMENT, "null likelihood descriptor");
    if (lik->nlatent != p->L)
        AGPL_FAIL(p->ctx, AGPL_ERR_INVALID_ARGUMENT, "the likelihood has %d latents, the plan was created for %d", lik->nlatent, p->L);
    if (!v || !G_out || !g_out) AGPL_FAIL(p->ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    const bool pad = p->Mc != p->M;
    if (pad) {
        rc = agpl_pad_natural(p->ctx, p->L, p->Mc, p->M, nullptr, nullptr, nullptr, v, nullptr, nullptr, nullptr, p->vp);
        if (rc) return rc;
    }
    rc = agpl_gibbs_pass_internal(p->ctx, lik, p->N, p->M, nullptr, p->Phi_acc, true, p->resid, mu0, y, pad ? p->vp : v, sweep,
                                  pad ? p->Gp : G_out, pad ? p->gp : g_out, f_out, omega_out, n_out, nuni_out);
    if (rc || !pad) return rc;
    return agpl_unpad_natural(p->ctx, p->L, p->Mc, p->M, p->Gp, p->gp, G_out, g_out);
}
