/*
 * agpl.h -- C ABI of libagpl.so: the MI355X (gfx950) implementation of the inner inference loop of
 * AugmentedGPLikelihoods.jl (per-datapoint augmented-variable draw / expectation + the
 * diagonal-precision conditional Gaussian accumulation of one CAVI / Gibbs sweep).
 *
 * The reference has no FFI: its seam is Julia multiple dispatch on (lik, Omega | qOmega, y, f | qf)
 * with SoA containers.  Each entry point below names the reference method it stands
 * in for (paths relative to the reference repo root, v0.4.19); INTEGRATION.md shows the Julia
 * `ccall` methods a maintainer adds to route device arrays here.
 *
 * Conventions
 *   - every function returns an int32 status (AGPL_OK == 0, negative = error); the message of the
 *     last error on a context is available from agpl_last_error().
 *   - all array arguments are DEVICE pointers unless the name ends in _host; caller-owned; nothing is
 *     retained past return (the reference mutates Omega / qOmega in place and returns them:
 *     src/generic.jl:11, src/likelihoods/bernoulli.jl:23-24).
 *   - work is enqueued on the context's stream and NOT synchronised, except for the functions that
 *     return a host scalar (agpl_logtilt, agpl_expected_logtilt, agpl_aux_kldivergence).
 *   - multi-latent layout follows the reference: per-point containers are [L, N] column-major
 *     (L contiguous per point, src/likelihoods/categorical.jl:52-70); potentials / precisions are
 *     returned "transposed" as L contiguous vectors of N (src/utils.jl:24).
 *   - y: uint8 {0,1} for Bernoulli and one-hot categorical [L,N]; int32 counts for NegBinomial and
 *     Poisson; real (dtype) for StudentT, Laplace, heteroscedastic Gaussian.
 *   - one context per GPU; a context is not thread-safe (the reference is single-threaded).
 */
#ifndef AGPL_H
#define AGPL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGPL_VERSION 110

#if defined(__GNUC__)
#define AGPL_API __attribute__((visibility("default")))
#else
#define AGPL_API
#endif

typedef struct agpl_ctx agpl_ctx;

typedef enum {
    AGPL_OK = 0,
    AGPL_ERR_INVALID_ARGUMENT = -1, /* ArgumentError (e.g. negativemultinomial.jl:17-22), bad sizes    */
    AGPL_ERR_DOMAIN = -2,           /* DomainError   (polyagamma.jl:175)                             */
    AGPL_ERR_UNSUPPORTED = -3,      /* error(...)    (categorical.jl:165-170, polyagamma.jl:100-104) */
    AGPL_ERR_HIP = -4,              /* a HIP / rocSOLVER / rocBLAS call failed                        */
    AGPL_ERR_NOT_POSDEF = -5,       /* PosDefException of the M x M Cholesky                          */
    AGPL_ERR_OUT_OF_MEMORY = -6
} agpl_status;

/* likelihood families = the files of the reference's src/likelihoods directory */
typedef enum {
    AGPL_LIK_BERNOULLI_LOGISTIC = 0, /* bernoulli.jl        BernoulliLikelihood(LogisticLink)              */
    AGPL_LIK_NEGBINOMIAL = 1,        /* negativebinomial.jl NBParamFailure(r), p[0] = r                    */
    AGPL_LIK_STUDENTT = 2,           /* studentt.jl         StudentTLikelihood(nu, sigma), p[0]=nu p[1]=sigma */
    AGPL_LIK_CATEGORICAL = 3,        /* categorical.jl      LogisticSoftMaxLink(logtheta), nlatent = K      */
    AGPL_LIK_CATEGORICAL_BIJ = 4,    /* categorical.jl      BijectiveSimplexLink(...), nlatent = K-1        */
    AGPL_LIK_POISSON = 5,            /* poisson.jl          ScaledLogistic(lambda), p[0] = lambda           */
    AGPL_LIK_LAPLACE = 6,            /* laplace.jl          LaplaceLikelihood(beta), p[0] = beta            */
    AGPL_LIK_HETEROGAUSS = 7         /* heteroscedasticgaussian.jl InvScaledLogistic(lambda), nlatent = 2   */
} agpl_lik_kind;

/* constructor arguments of the likelihood (SURVEY.md 5 "config / flags"). */
typedef struct {
    int32_t kind;           /* agpl_lik_kind */
    int32_t nlatent;        /* nlatent(lik): generic.jl:87, categorical.jl:46-47                       */
    double p[4];            /* see agpl_lik_kind                                                       */
    const double *logtheta; /* HOST pointer, categorical only: K entries (K = nlatent [+1 if bijective]) */
} agpl_lik_desc;

typedef enum { AGPL_F32 = 0, AGPL_F64 = 1 } agpl_dtype;

/* ---- context: replaces GLOBAL_RNG / the rng argument (src/generic.jl:1-3,14-16,32-34) ---------- */
AGPL_API int32_t agpl_ctx_create(agpl_ctx **out, int32_t device_id, uint64_t seed);
AGPL_API int32_t agpl_ctx_destroy(agpl_ctx *ctx);
/* enqueue on an existing hipStream_t (e.g. the caller's current stream) instead of the context's own
 * non-blocking stream; NULL = the device's default (null) stream */
AGPL_API int32_t agpl_ctx_set_stream(agpl_ctx *ctx, void *hip_stream);
AGPL_API int32_t agpl_ctx_set_seed(agpl_ctx *ctx, uint64_t seed);
/* global index of this context's local point 0 (default 0).  The per-point Philox streams of agpl_aux_sample and
 * agpl_gibbs_pass are keyed (seed, point_offset + i, sweep): a rank that owns points [i0, i1) of N sharded
 * observations sets i0 here and, with the same seed on every rank, draws exactly what one process
 * holding all N points would (agpl_gibbs_draw_v is unaffected: every rank must draw the identical v).           */
AGPL_API int32_t agpl_ctx_set_point_offset(agpl_ctx *ctx, int64_t i0);
/* waits for the context's stream; also returns what an agpl_gaussian_factor_async still has to report */
AGPL_API int32_t agpl_ctx_synchronize(agpl_ctx *ctx);
AGPL_API const char *agpl_last_error(const agpl_ctx *ctx);
AGPL_API int32_t agpl_version(void);

/* ---- Gibbs half: aux_sample!(rng, Omega, lik, y, f)  src/generic.jl:5-12 ------------------------
 * Omega_i <- one draw from aux_full_conditional(lik, y_i, f_i):
 *   bernoulli.jl:13-15 PG(1,|f|); negativebinomial.jl:20-22 PG(y+r,|f|); studentt.jl:46-48 Gamma;
 *   categorical.jl:72-78 PG o NegativeMultinomial; poisson.jl:26-28 PG o Poisson; laplace.jl:40-42
 *   InverseGaussian; heteroscedasticgaussian.jl:28-32.
 * The PG draw is polyagamma.jl:121-257 (Devroye alternating series, one lane per point, per-lane
 * Philox4x32-10 stream keyed (ctx seed, point index, sweep)).  f, omega_out are float64.
 * n_out: int64 counts ([L,N] categorical, [N] poisson / heterogauss) or NULL.
 * nuni_out / nterms_out: optional uint32[N] bookkeeping (uniforms consumed, summed series index).
 * Limit: a point's PG(b, c) needs b = y + r (y + n) < 65535 -- its PG(1, c) draws are numbered with 16 bits of the Philox
 *   counter; a larger b returns AGPL_ERR_UNSUPPORTED (that point's outputs are NaN), also from agpl_gibbs_pass* and
 *   agpl_rand_polyagamma.  The reference draws any integer b (polyagamma.jl:129-134).                                  */
AGPL_API int32_t agpl_aux_sample(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                        const double *f, double *omega_out, int64_t *n_out, uint32_t sweep,
                        uint32_t *nuni_out, uint32_t *nterms_out);

/* rand(rng, PolyaGamma(b, c), n)  polyagamma.jl:121-126: n iid draws, stream index = draw index.   */
AGPL_API int32_t agpl_rand_polyagamma(agpl_ctx *ctx, double b, double c, int64_t n, uint32_t sweep,
                             double *out, uint32_t *nuni_out, uint32_t *nterms_out);

/* auglik_potential / auglik_precision (+ _and_precision, src/generic.jl:64-66):
 *   bernoulli.jl:27-33, negativebinomial.jl:35-41, studentt.jl:60-66, categorical.jl:112-119,
 *   poisson.jl:41-47, laplace.jl:54-60, heteroscedasticgaussian.jl:48-66 (fg = [2,N] latents).     */
AGPL_API int32_t agpl_potential_precision(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                 const double *omega, const int64_t *n_aux, const double *fg,
                                 double *beta_out, double *gamma_out);

/* ---- CAVI half --------------------------------------------------------------------------------
 * aux_posterior!(qOmega, lik, y, qf): bernoulli.jl:17-25, negativebinomial.jl:24-33,
 *   studentt.jl:50-58, categorical.jl:80-110, poisson.jl:30-39, laplace.jl:44-52,
 *   heteroscedasticgaussian.jl:34-46.   qf = Normal marginals as SoA (mu, var), [L,N].
 *   out1 = c (studentt: beta_i; laplace: mu_i), out2 = p [L,N] (categorical) / lambda (poisson,
 *   heterogauss) or NULL, out3 = psi (heterogauss) or NULL.  dtype selects float / double arrays.   */
AGPL_API int32_t agpl_aux_posterior(agpl_ctx *ctx, const agpl_lik_desc *lik, int32_t dtype, int64_t n,
                           const void *y, const void *mu, const void *var, void *out1, void *out2,
                           void *out3);

/* expected_auglik_potential / expected_auglik_precision (+ _and_precision generic.jl:68-72):
 *   bernoulli.jl:35-45, negativebinomial.jl:43-49, studentt.jl:68-74, categorical.jl:121-136,
 *   poisson.jl:49-60, laplace.jl:62-68, heteroscedasticgaussian.jl:68-104 (mu_g = mean of q(g)).
 *   q1,q2 = the aux_posterior outputs.                                                            */
AGPL_API int32_t agpl_expected_potential_precision(agpl_ctx *ctx, const agpl_lik_desc *lik, int32_t dtype,
                                          int64_t n, const void *y, const void *q1, const void *q2,
                                          const void *mu_g, void *beta_out, void *gamma_out);

/* ---- ELBO terms (N-reductions, float64, result to host) ----------------------------------------
 * logtilt generic.jl:40-46 ; expected_logtilt api.jl:219-223 ; aux_kldivergence generic.jl:56-62 ;
 * aug_loglik generic.jl:48-50 ; expected_aug_loglik generic.jl:52-54.                               */
AGPL_API int32_t agpl_logtilt(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                     const double *omega, const int64_t *n_aux, const double *f, double *out_host);
/* agpl_aug_loglik: aug_loglik(lik, Omega, y, f) = logtilt + logdensity_def(aux_prior(lik, y), Omega), src/generic.jl:48-50;
 *   the PG prior density is the 101-term series of src/SpecialDistributions/polyagamma.jl:37-91 (log-domain for
 *   omega < 1e-2), evaluated per point in float64.  agpl_aux_prior_logpdf is the second term alone.  Priors:
 *   Bernoulli PG(1,0) (bernoulli.jl:51-57), negative binomial PG(y+r,0) (negativebinomial.jl:67-73), Student-t
 *   Gamma(nu/2, 2 sigma^2/nu) (studentt.jl:85-91), Poisson PolyaGammaPoisson(y,0,lambda) (poisson.jl:67-76, joint density
 *   polyagammapoisson.jl:29-33: needs the counts n_aux), Laplace InverseGamma(1/2, (2 beta)^-2) (laplace.jl:90-96).
 *   Heteroscedastic Gaussian: agpl_aug_loglik is the likelihood's own method (heteroscedasticgaussian.jl:106-128; f = fg
 *   [2,N], n_aux required); it has no aux_prior (agpl_aux_prior_logpdf, agpl_logtilt: AGPL_ERR_UNSUPPORTED).
 *   Categorical: AGPL_ERR_UNSUPPORTED -- the reference's logdensity_def of PolyaGammaNegativeMultinomial is broken
 *   (polyagammanegativemultinomial.jl:33-39 sums over the NamedTuple's 2 fields; its tests are skipped). */
AGPL_API int32_t agpl_aux_prior_logpdf(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                       const double *omega, const int64_t *n_aux, double *out_host);
AGPL_API int32_t agpl_aug_loglik(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                 const double *omega, const int64_t *n_aux, const double *f, double *out_host);
AGPL_API int32_t agpl_expected_logtilt(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                              const double *q1, const double *q2, const double *mu,
                              const double *var, double *out_host);
AGPL_API int32_t agpl_aux_kldivergence(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                              const double *q1, const double *q2, double *out_host);
/* agpl_expected_aug_loglik: expected_aug_loglik(lik, qOmega, y, qf) = expected_logtilt + aux_kldivergence, src/generic.jl:52-54
 *   (the PLUS is the reference's; its example ELBOs subtract the KL themselves, examples/bernoulli/script.jl:65-70).
 *   Heteroscedastic Gaussian: the likelihood's own method, heteroscedasticgaussian.jl:130-145 (q1 = c, q2 = lambda of
 *   aux_posterior!; mu, var = q(f), q(g) as [2,N]; `var(first(qg))` read as var(q(g)), Appendix B).  Non-bijective
 *   categorical: AGPL_ERR_UNSUPPORTED (categorical.jl:165-170). */
AGPL_API int32_t agpl_expected_aug_loglik(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                          const double *q1, const double *q2, const double *mu, const double *var,
                                          double *out_host);

/* ---- the sparse Gaussian half of a sweep -------------------------------------------------------
 * Feature matrix Phi: float32, [M, N] column-major (M contiguous per point; ld = M), M % 128 == 0
 * (zero-pad features).  In the reference's terms Phi = K_ZX (docs/src/index.md:154-163) or any basis
 * of it; the host driver uses the whitened basis Phi = L^-1 K_ZX, K_Z = L L'.
 *
 * agpl_marginals: q(f_i) = N(mu_i, var_i), the `marginals(post_u(x))` call of
 *   examples/bernoulli/script.jl:32-33 (un-vendored ApproximateGPs) in feature form:
 *     mu[l,i]  = mu0[l,i] + phi_i' alpha_l          var[l,i] = kdiag[i] - phi_i' W_l phi_i
 *   Wpack: [L, M, M] float32 in the packed form written by agpl_gaussian_update (lower-triangular
 *   transpose with doubled off-diagonal: Wpack[b][a] = (b>a ? 2 W[a][b] : W[a][a]); upper part ignored).
 *   alpha: [L, M] float32.  kdiag: [N] float32.  mu0: [L][N] float32 or NULL.  mu/var out: [L][N] float32.
 *   f32-input MFMA (v_mfma_f32_32x32x2_f32), executes ~ (1 + 128/M) N M^2 flop per latent.          */
AGPL_API int32_t agpl_marginals(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi,
                       const float *kdiag, const float *mu0, const float *Wpack, const float *alpha,
                       float *mu_out, float *var_out);

/* agpl_accumulate: the only O(N) objects that cross a sweep:
 *     G_l = Phi Diag(gamma_l) Phi'  (= kappa Diag(r) kappa' of docs/src/index.md:154-163)
 *     g_l = Phi beta_l              (= kappa t)
 *   gamma, beta: [L][N] float32.  G_out [L, M, M], g_out [L, M] float64 (full symmetric).
 *   f32-input MFMA over 128 x 128 output tiles of the lower triangle, N split across workgroups,
 *   two-level accumulation (f32 over <= 2048 points, then f64), fixed-order f64 slab reduction:
 *   bitwise reproducible, no atomics.                                                             */
AGPL_API int32_t agpl_accumulate(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi,
                        const float *beta, const float *gamma, double *G_out, double *g_out);

/* agpl_gaussian_update: S = (I + G)^-1, m = S (g + eta0)  -- the update of
 *   examples/bernoulli/script.jl:35-36 (CAVI) / :82-83 (Gibbs) in the sparse whitened form of
 *   docs/src/index.md:154-163.  float64 Cholesky (rocSOLVER potrf/potri) on device.
 *   G [L,M,M], g [L,M], eta0 [L,M] or NULL.  Outputs (any may be NULL): S_out [L,M,M] f64,
 *   m_out [L,M] f64, Wpack_out [L,M,M] f32 (packed -S, so that var = kdiag + phi' S phi through
 *   agpl_marginals), alpha_out [L,M] f32 (= m).  Returns AGPL_ERR_NOT_POSDEF if I + G is not SPD.   */
AGPL_API int32_t agpl_gaussian_update(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                             const double *eta0, double *S_out, double *m_out, float *Wpack_out,
                             float *alpha_out);

/* agpl_gaussian_kl: sum over latents of KL(q(v_l) || p(v_l)) = (tr S + m'm - M + logdet(I + G)) / 2 for the
 *   q(v) that agpl_gaussian_update would produce from (G, g, eta0): the `kldivergence(u_post.approx.q,
 *   u_post.approx.fz)` term of aug_elbo (examples/bernoulli/script.jl:65-70) in the whitened basis, where the
 *   prior is N(0, I).  float64, synchronous, result to host.                                               */
AGPL_API int32_t agpl_gaussian_kl(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                  const double *eta0, double *kl_out_host);

/* agpl_pack_w: Wpack (float32, packed as above) from a symmetric float64 W [L,M,M] scaled by `scale`. */
AGPL_API int32_t agpl_pack_w(agpl_ctx *ctx, int32_t M, int32_t L, const double *W, double scale,
                    float *Wpack_out);

/* agpl_cavi_pass: one fused pass = agpl_marginals -> agpl_aux_posterior ->
 *   agpl_expected_potential_precision -> agpl_accumulate (the loop body of
 *   examples/bernoulli/script.jl:32-36 up to the M x M solve).  Single- and multi-latent.
 *   Optional per-point outputs (float32): c_out [L,N] (out1 of aux_posterior), gamma_out, beta_out [L][N]. */
AGPL_API int32_t agpl_cavi_pass(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M,
                       const float *Phi, const float *kdiag, const float *mu0, const void *y,
                       const float *Wpack, const float *alpha, double *G_out, double *g_out,
                       float *c_out, float *gamma_out, float *beta_out);

/* ---- Gibbs half of the sparse sweep (examples/bernoulli/script.jl:76-87 in sparse form) ---------------
 * agpl_gibbs_pass: for every point  f_il = mu0_il + phi_i' v_l + sqrt(kdiag_i) eps_il  (the draw of f given
 *   the inducing draw v under the sparse model; eps from the point's Philox stream (seed, i, sweep)), then
 *   aux_sample! (src/generic.jl:5-12) on the same stream, auglik_potential / auglik_precision of the draw, and
 *   the accumulation G_l = Phi Diag(gamma_l) Phi', g_l = Phi beta_l (agpl_accumulate).  One read of Phi for
 *   the projection + one for the accumulation.  v: [L, M] float64.  y real-valued: float64.
 *   Optional outputs: f_out [L,N] col-major f64, omega_out f64, n_out i64, nuni_out u32[N].            */
AGPL_API int32_t agpl_gibbs_pass(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M,
                                 const float *Phi, const float *kdiag, const float *mu0, const void *y,
                                 const double *v, uint32_t sweep, double *G_out, double *g_out,
                                 double *f_out, double *omega_out, int64_t *n_out, uint32_t *nuni_out);

/* agpl_gibbs_draw_v: v_l ~ N(m_l, S_l), S = (I + G)^-1, m = S (g + eta0): the `rand!(MvNormal(mu, Sigma), f)`
 *   of examples/bernoulli/script.jl:82-84 for the M inducing coordinates.  Cholesky C C' = I + G (rocSOLVER
 *   potrf), m by potrs, v = m + C^-T z with z_a = the normal of Philox stream (seed, l*M + a, sweep | 2^31).
 *   v_out [L,M] f64; m_out [L,M] f64 or NULL.  sweep < 2^31.                                            */
AGPL_API int32_t agpl_gibbs_draw_v(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                   const double *eta0, uint32_t sweep, double *v_out, double *m_out);

/* ---- the full-rank Gibbs step the reference executes (gibbs_sample, examples/bernoulli/script.jl:76-87) ---
 * agpl_dense_cholesky: L_out = lower Cholesky factor of the symmetric N x N matrix A (float64, rocSOLVER potrf;
 *   the `_chol_cov(fz)` of script.jl:77).  Stored in the triangle that LAPACK calls lower for a column-major
 *   array (= the upper triangle of a row-major view); the other triangle keeps A.  PosDefException -> -5.
 * agpl_dense_gibbs_step: Omega <- aux_sample!(lik, y, f) (:81); f ~ N(mu, Sigma) with
 *   Sigma = (K^-1 + Diag(gamma))^-1, mu = Sigma (beta + K^-1 mu0) (:82-84), evaluated with one Cholesky of
 *   B = I + D^1/2 K D^1/2 (written to B_work, N x N float64) and no inverse -- an exact draw (Matheron's rule).
 *   Normals: Philox streams (seed, 0..2N-1, sweep | 2^31).  Single-latent likelihoods.  f_inout: f in, new f out. */
AGPL_API int32_t agpl_dense_cholesky(agpl_ctx *ctx, int64_t N, const double *A, double *L_out);
AGPL_API int32_t agpl_dense_gibbs_step(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, const double *K,
                                       const double *Lk, const double *mu0, const void *y, double *f_inout,
                                       double *B_work, uint32_t sweep, double *omega_out, int64_t *n_out);

/* agpl_probe_mfma_f64: measured float64 MFMA rate of this device [TFLOP/s] (v_mfma_f64_16x16x4_f64 back to back on
 *   every SIMD for `iters` x 16 instructions per wave): the peak bench.py prices the C5 Cholesky against
 *   ("FP64 peak is not in the local guide -- measure, don't assume").  Synchronous.              */
AGPL_API int32_t agpl_probe_mfma_f64(agpl_ctx *ctx, int32_t iters, double *tflops_host);

/* agpl_debug_force_factor_rescue: test hook.  The one-launch M x M factorisation (agpl_gaussian_factor*, M <= 1024) runs as
 *   cooperating workgroups that assume each other resident; should other work hold their CUs, the launch reports it and a
 *   second launch queued behind it redoes the latent in one workgroup, without the host.  on != 0 makes every cooperative
 *   launch of this context take that path at once, so that it can be tested (it is otherwise never exercised on an idle
 *   device); the results are the cooperative launch's.  No reference counterpart (the reference calls LAPACK). */
AGPL_API int32_t agpl_debug_force_factor_rescue(agpl_ctx *ctx, int32_t on);

/* agpl_probe_mfma_f16: SUSTAINED float16 MFMA rate of this device [TFLOP/s]: v_mfma_f32_32x32x16_f16 in the instruction
 *   mix of one stage of the split accumulation's wave (12 per step into four 32 x 32 accumulators), six launches of
 *   `iters` steps per wave timed as one region after two that settle the clocks, `workgroups_per_cu` (1..4) 4-wave
 *   workgroups per CU.  mode 0: MFMA only; mode 1: with that stage's eight 16-byte LDS fragment reads; modes 2, 3: the
 *   same for v_mfma_f32_16x16x32_f16 in the mix of one stage of the shipped marginal kernel's wave (48 per step into
 *   sixteen 16 x 16 accumulators, hashed operands, at most 2 workgroups per CU; 3: with the stage's sixteen fragment reads)
 *   -- the shape the shipped
 *   contractions issue.  The ceiling bench.py reports next to the data-sheet peak (the clock under MFMA load is below the
 *   boost clock).  ms_host (may be NULL): average launch duration.  Synchronous.                                  */
AGPL_API int32_t agpl_probe_mfma_f16(agpl_ctx *ctx, int32_t iters, int32_t mode, int32_t workgroups_per_cu,
                                     double *tflops_host, double *ms_host);

/* Optional in-library timing of the two MFMA kernels (bench.py's roofline leg): when enabled, a hipEvent
 * pair is recorded on the context's stream around every launch of the marginal (which = 0), the
 * accumulation (which = 1), the Gibbs per-point (which = 2) and the aux_sample (which = 3) kernel.  agpl_timing_read synchronises the stream, returns the summed kernel
 * time [ms] and the number of launches since the last read, and resets the counters.                 */
AGPL_API int32_t agpl_timing_enable(agpl_ctx *ctx, int32_t on);
AGPL_API int32_t agpl_timing_read(agpl_ctx *ctx, int32_t which, double *total_ms_host, int64_t *launches_host);

/* ---- split-float16 marginal pass (agpl_split.hip): the same a11 computation on the fast matrix cores ----------
 * Every float32 operand x is carried as hi = f16(x), lo = f16(x - hi); W' Phi ~= hi*hi + hi*lo + lo*hi runs as
 * three v_mfma_f32_32x32x16_f16 per sub-product with float32 accumulation (representation error
 * <= max(2^-22 |x|, 3e-8) per operand, |x| < 6e4).  Operands live in blocked images (4 KB blocks of
 * [2 planes][128 rows][8 halves], one per (row block, 16-wide k-slice)) that are both the HBM and the LDS layout.
 *   agpl_split_features_bytes: bytes of ONE image (hi or lo) for N points, M features.
 *   agpl_split_features: Phi (float32 [M,N] col-major) -> Phi_hi, Phi_lo images (once per data set).  This image is
 *                        UNSCALED: AGPL_ERR_DOMAIN (point and feature in agpl_last_error) if a feature is not finite
 *                        or |x| >= 65504, the float16 range.  Synchronises the stream once.
 *   agpl_pack_w_split:   scale * W' (W symmetric float64 [L,M,M]) -> W_hi, W_lo images, L*M*M halves each
 *                        (each sweep, after agpl_gaussian_update: W = S, scale = -1).
 *   agpl_marginals_split / agpl_cavi_pass_split: drop-in twins of agpl_marginals / agpl_cavi_pass taking the
 *                        images (the float32 Phi is still read by the Hadamard epilogue and the accumulation). */
AGPL_API int64_t agpl_split_features_bytes(int64_t N, int32_t M);
AGPL_API int32_t agpl_split_features(agpl_ctx *ctx, int64_t N, int32_t M, const float *Phi, void *Phi_hi,
                                     void *Phi_lo);
AGPL_API int32_t agpl_pack_w_split(agpl_ctx *ctx, int32_t M, int32_t L, const double *W, double scale,
                                   void *W_hi, void *W_lo);
AGPL_API int32_t agpl_marginals_split(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi,
                                      const void *Phi_hi, const void *Phi_lo, const float *kdiag,
                                      const float *mu0, const void *W_hi, const void *W_lo, const float *alpha,
                                      float *mu_out, float *var_out);
AGPL_API int32_t agpl_cavi_pass_split(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M,
                                      const float *Phi, const void *Phi_hi, const void *Phi_lo,
                                      const float *kdiag, const float *mu0, const void *y, const void *W_hi,
                                      const void *W_lo, const float *alpha, double *G_out, double *g_out,
                                      float *c_out, float *gamma_out, float *beta_out);

/* ---- factor (one-pass) form of the split-float16 marginal pass -------------------------------------------------
 * I + G = R R' (Cholesky), U = R^-1 (lower triangular), T = U Phi:
 *     var_n = (k_nn - |phi_n|^2) + sum_a T[a,n]^2        mu_n = mu0_n + sum_a v_a T[a,n],   v = U (g + eta0)
 * which is the same q(f_n) as agpl_marginals (S = U'U, m = U'v; reference: the marginals of q(u) = N(m, S) pushed
 * through K_XZ K_ZZ^-1, examples/<lik>/script.jl `u_posterior` + the SVGP predictive of ApproximateGPs) but needs T only:
 * the launch reads the feature images once per 256-row block and never the float32 features.  M % 256 == 0.
 *
 * agpl_feature_residual : resid_n = k_nn - |phi_n|^2 (static; float64 accumulation).
 * agpl_gaussian_factor  : A_work [L,M,M] f64 scratch/out (on return its column-major lower triangle holds U);
 *                         v_out [L,M] f64 / v32_out [L,M] f32 / (U_hi, U_lo) images / logdet_out [L] f64 (device,
 *                         log det(I + G)) are optional.  AGPL_ERR_NOT_POSDEF as agpl_gaussian_update.
 * agpl_pack_factor_split: images of U from A_work (what agpl_gaussian_factor does when U_hi / U_lo are given).
 * agpl_marginals_factor_split / agpl_cavi_pass_factor_split: the factor-form twins of agpl_marginals_split /
 *                         agpl_cavi_pass_split (`resid` in place of kdiag, U images and v in place of W-pack and alpha).   */
AGPL_API int32_t agpl_feature_residual(agpl_ctx *ctx, int64_t N, int32_t M, const float *Phi, const float *kdiag,
                                       float *resid_out);
AGPL_API int32_t agpl_gaussian_factor(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                      const double *eta0, double *A_work, double *v_out, float *v32_out, void *U_hi,
                                      void *U_lo, double *logdet_out);
/* The same without waiting for the outcome: a failed factorisation (AGPL_ERR_NOT_POSDEF, ...) is reported by the next
 * agpl_cavi_pass_factor_split -- after that call has enqueued its own kernels, so the host never idles the GPU between
 * the update and the next pass of a sweep loop (examples/bernoulli/script.jl:34-38) --, by the next
 * agpl_gaussian_factor[_async], or by agpl_ctx_synchronize on this context.  Work enqueued behind a failed
 * factorisation computes on NaNs; only the moment of the report moves. */
AGPL_API int32_t agpl_gaussian_factor_async(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                      const double *eta0, double *A_work, double *v_out, float *v32_out, void *U_hi,
                                      void *U_lo, double *logdet_out);
AGPL_API int32_t agpl_pack_factor_split(agpl_ctx *ctx, int32_t M, int32_t L, const double *A, void *U_hi, void *U_lo);
AGPL_API int32_t agpl_marginals_factor_split(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const void *Phi_hi,
                                             const void *Phi_lo, const float *resid, const float *mu0,
                                             const void *U_hi, const void *U_lo, const float *v, float *mu_out,
                                             float *var_out);
AGPL_API int32_t agpl_cavi_pass_factor_split(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M,
                                             const float *Phi, const void *Phi_hi, const void *Phi_lo,
                                             const float *resid, const float *mu0, const void *y, const void *U_hi,
                                             const void *U_lo, const float *v, double *G_out, double *g_out,
                                             float *c_out, float *gamma_out, float *beta_out);

/* ---- split-float16 accumulation from a static point-major image (agpl_syrk.hip) --------------------------------
 * G_l = Phi Diag(gamma_l) Phi', g_l = Phi beta_l (the accumulators of docs/src/index.md:154-163: S = (K_Z^-1 +
 * kappa Diag(r) kappa')^-1, m = S (kappa t + ...), in the whitened basis; script.jl:35-36 in sparse form).  The reduction
 * index of this product is the POINT, the strided index of the float32 features; the image stores, once per data set,
 * hi = f16(s phi), lo = f16(s phi - hi) (s = 2^e chosen from max |Phi|, -30 <= e <= 30: max |Phi| in 2^-24 .. 2^44, else
 * AGPL_ERR_DOMAIN; error <= 2^-22 |phi| down to 2^-17 max |Phi|) in 4 KB blocks [point slice of 16][feature block of 128][hi | lo] =
 * [2 planes of 8 points][128 features][8 halves], whose 16-byte granule (one feature, 8 consecutive points) is one MFMA
 * operand fragment.  The accumulation then reads ONLY the image: A = the image (HBM -> LDS by DMA), B = gamma_n x the
 * image (rebuilt, scaled and re-split in registers), three float16 MFMA products per float32 product, float32
 * accumulation over 4096-point slices, the fixed-order float64 slab reduction of agpl_accumulate.
 *   agpl_accumulate_image_bytes : bytes of the image (256-byte header + blocks) for N points, M features (M % 128 == 0).
 *   agpl_accumulate_image       : Phi (float32 [M,N] col-major) -> image.  AGPL_ERR_DOMAIN (with the offending point and
 *                                 feature in agpl_last_error) if a feature is not finite.  Synchronises the stream once.
 *   agpl_accumulate_split       : agpl_accumulate on the float16 matrix cores:
 *                                 from the image when acc_image != NULL and M % 256 == 0 (Phi may then be NULL), else from
 *                                 the float32 Phi (psi = sqrt(gamma) phi split while staging; |sqrt(gamma) phi| < 6e4).
 *                                 gamma >= 0 (TestUtils.jl:88): from the image, a negative or non-finite gamma is
 *                                 AGPL_ERR_DOMAIN with its index (the call then synchronises the stream once; inside a
 *                                 sweep the same report comes with the next update, without a synchronisation).        */
AGPL_API int64_t agpl_accumulate_image_bytes(int64_t N, int32_t M);
AGPL_API int32_t agpl_accumulate_image(agpl_ctx *ctx, int64_t N, int32_t M, const float *Phi, void *image_out);
AGPL_API int32_t agpl_accumulate_split(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi,
                                       const void *acc_image, const float *beta, const float *gamma, double *G_out,
                                       double *g_out);

/* The sweep entry points on the image (the shipped path: bench.py, SparseCAVI / SparseGibbs defaults):
 *   agpl_cavi_pass_factor_image : agpl_cavi_pass_factor_split with the accumulation taken from Phi_acc (the image of
 *                                 agpl_accumulate_image) -- the float32 features are not an argument; M % 256 == 0.
 *   agpl_gibbs_pass_image       : agpl_gibbs_pass with the split-float16 accumulation, from Phi_acc when it is given
 *                                 and M % 256 == 0 (Phi is still read by the projection phi_i' v).
 * Every *_split / *_image entry point runs the split-float16 accumulation.
 * agpl_cavi_pass_factor_image is three launches up to the slabs (marginal partial sums; ONE per-point kernel: q(f_i),
 * aux_posterior!, expected potential / precision, written straight into the accumulation's gamma | beta records; the
 * accumulation) -- gamma_out / beta_out / c_out may be NULL and are then never materialised.  A gamma that is negative or not
 * finite (observations or marginals outside the likelihood's domain) is reported as AGPL_ERR_DOMAIN, with its flat index, by
 * the call that reports the outcome of the update enqueued behind this pass (agpl_gaussian_factor_async's rules).          */
AGPL_API int32_t agpl_cavi_pass_factor_image(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M,
                                             const void *Phi_hi, const void *Phi_lo, const void *Phi_acc,
                                             const float *resid, const float *mu0, const void *y, const void *U_hi,
                                             const void *U_lo, const float *v, double *G_out, double *g_out,
                                             float *c_out, float *gamma_out, float *beta_out);
AGPL_API int32_t agpl_gibbs_pass_image(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M,
                                       const float *Phi, const void *Phi_acc, const float *kdiag, const float *mu0,
                                       const void *y, const double *v, uint32_t sweep, double *G_out, double *g_out,
                                       double *f_out, double *omega_out, int64_t *n_out, uint32_t *nuni_out);

/* ---- the plan: the shipped sweep path (agpl_plan.hip) -------------------------------------------------------------------
 * Everything static about one data set on one context, built once: the two split-float16 images of Phi (marginal image by
 * 128-point tile, accumulate image point-major), BOTH carrying 2^e Phi with one e chosen from max |Phi| (domain: any finite
 * features with max |Phi| in 2^-24 .. 2^44, else AGPL_ERR_DOMAIN; a non-finite feature is AGPL_ERR_DOMAIN with its (point,
 * feature)), a copy of the
 * Nystrom residual, and q(v) in factor form -- U = chol(I + G)^-1, v = U (g + eta0), log det(I + G) -- which the plan's update
 * writes and its passes read; kernels are chosen by shape.  After creation the float32 features are not read by any plan entry
 * point (C2: 41 GB of images resident instead of 61 GB with the features; 20.5 GB for a Gibbs-only plan).  These entry points supersede
 * agpl_split_features, agpl_accumulate_image, agpl_pack_factor_split, agpl_marginals_factor_split, agpl_cavi_pass_split,
 * agpl_cavi_pass_factor_split, agpl_cavi_pass_factor_image and agpl_gibbs_pass_image (kept below for existing callers).
 *   agpl_plan_bytes     : device bytes a plan needs for (N, M, L, flags); 0 for sizes a plan does not take (M % 256 != 0, L > 64).
 *   agpl_plan_create    : Phi float32 [M, N] column-major; resid float32 [N] (agpl_feature_residual); flags: 0, or
 *                         AGPL_PLAN_NO_MARGINALS for a plan that serves Gibbs passes only (no marginal image: half the bytes);
 *                         storage: agpl_plan_bytes bytes of caller-owned device memory that stay valid for the plan's life, or
 *                         NULL (the library allocates and frees).  q(v) starts at N(0, I) (examples/bernoulli/script.jl:41-42).
 *                         Waits for its own kernels before returning (one synchronisation per data set): Phi and resid are not
 *                         read after the call returns and may be freed on any stream.  A resid entry more negative than the
 *                         float32 round-off of k_ii - |phi_i|^2 (d < -1e-5 (|d| + |phi|^2)) is AGPL_ERR_DOMAIN with its index;
 *                         round-off below zero is stored as 0.
 *   lifetime            : a plan enqueues on, and reports through, the context it was created on: agpl_plan_destroy it BEFORE
 *                         agpl_ctx_destroy (which returns AGPL_ERR_INVALID_ARGUMENT, and destroys nothing, while plans are alive).
 *   agpl_cavi_pass_plan : marginals of the plan's q(v) -> aux_posterior! -> expected potential / precision -> G, g
 *                         (script.jl:32-36 up to the M x M solve; agpl_cavi_pass's contract).  elbo_terms_out (device double, may be
 *                         NULL): sum over the points of expected_logtilt_i - aux_kldivergence_i for the q(v) the pass used, from
 *                         the same marginals in float64 -- the per-point part of aug_elbo (script.jl:65-70) rides the pass's one
 *                         per-point kernel (AGPL_ERR_UNSUPPORTED for the non-bijective categorical and the heteroscedastic
 *                         likelihood, whose terms the reference does not define).
 *   agpl_plan_update    : q(v) <- N(S (g + eta0), S), S = (I + G)^-1 (script.jl:35-36), kept as (U, v); asynchronous, outcome
 *                         reported as by agpl_gaussian_factor_async.  kl_out (device double, may be NULL): KL(q(v) || N(0, I)) of
 *                         the NEW q(v), from U and v (the kldivergence term of aug_elbo).  ELBO of a q(v) = the elbo_terms of the
 *                         pass that used it (summed over ranks) - the kl of the update that made it.
 *   agpl_marginals_plan : q(f_i) of the plan's q(v): mu, var float32 [L][N].
 *   agpl_gibbs_pass_plan: agpl_gibbs_pass with the plan's residual and accumulate image; the projection phi_i' v is formed from the
 *                         image too (x = (hi + lo) 2^-e: the features to 2^-22 relative, float64 accumulation in a fixed order),
 *                         so a Gibbs chain needs the float32 features at plan creation only.
 *   agpl_plan_factor    : device pointers to U (float64 [L, M, M], column-major lower triangle), v (float64 [L, M]) and the
 *                         residual, e.g. to form S = U'U, m = U'v.   agpl_plan_info: sizes, the images' scale exponent, bytes.
 *   agpl_plan_state     : device pointers to what an update rewrites besides U and v -- the images of 2^15 U (float16, L M M each),
 *                         v as float32 [L, M], log det(I + G) [L] -- for repeatability checks and checkpoints. */
typedef struct agpl_plan agpl_plan;
#define AGPL_PLAN_NO_MARGINALS 1u
AGPL_API int64_t agpl_plan_bytes(int64_t N, int32_t M, int32_t L, uint32_t flags);
AGPL_API int32_t agpl_plan_create(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi, const float *resid,
                                  uint32_t flags, void *storage, agpl_plan **plan_out);
AGPL_API int32_t agpl_plan_destroy(agpl_plan *plan);
AGPL_API int32_t agpl_plan_info(const agpl_plan *plan, int64_t *N, int32_t *M, int32_t *L, int32_t *scale_exp, int64_t *bytes);
AGPL_API int32_t agpl_plan_factor(const agpl_plan *plan, const double **U_out, const double **v_out, const float **resid_out);
AGPL_API int32_t agpl_plan_state(const agpl_plan *plan, const void **U_hi_out, const void **U_lo_out, const float **v32_out,
                                 const double **logdet_out);
AGPL_API int32_t agpl_cavi_pass_plan(agpl_plan *plan, const agpl_lik_desc *lik, const float *mu0, const void *y, double *G_out,
                                     double *g_out, float *c_out, float *gamma_out, float *beta_out, double *elbo_terms_out);
AGPL_API int32_t agpl_plan_update(agpl_plan *plan, const double *G, const double *g, const double *eta0, double *kl_out);
AGPL_API int32_t agpl_marginals_plan(agpl_plan *plan, const float *mu0, float *mu_out, float *var_out);
AGPL_API int32_t agpl_gibbs_pass_plan(agpl_plan *plan, const agpl_lik_desc *lik, const float *mu0, const void *y,
                                      const double *v, uint32_t sweep, double *G_out, double *g_out, double *f_out,
                                      double *omega_out, int64_t *n_out, uint32_t *nuni_out);

/* agpl_allreduce_nat: the exchange step of the N-sharded sweep: in-place float64 sum of the
 *   L (M^2 + M) natural-parameter accumulators over an RCCL communicator (ncclComm_t as void*), queued on the
 *   context's stream.  For hosts that own their communicator (the Julia / C++ callers of INTEGRATION.md); the
 *   Python host reaches the same RCCL through torch.distributed.  librccl is loaded at the first call.          */
AGPL_API int32_t agpl_allreduce_nat(agpl_ctx *ctx, void *rccl_comm, double *buf, int64_t count);

/* (agpl_set_accumulate_precision, v100: removed in v110.  The float32-named entry points -- agpl_accumulate, agpl_cavi_pass,
 * agpl_gibbs_pass -- always contract on the float32-input MFMA; the split-float16 arithmetic is the plan API's, and the
 * superseded *_split / *_image entry points'.  There is no per-context precision state any more.) */

/* bytes of scratch the context will hold for a given problem (allocated lazily, reused) */
AGPL_API int64_t agpl_workspace_bytes(int64_t N, int32_t M, int32_t L);

/* ---- features and synthetic workloads (the step before the path) -----------
 * agpl_se_features: K_ZX[a,i] = exp(-(x_i - z_a)^2 / (2 ell^2)) (with_lengthscale(SqExponentialKernel(),
 *   ell), examples/bernoulli/script.jl:15), float32 [ld, N] column-major, rows >= M zero-filled.     */
AGPL_API int32_t agpl_se_features(agpl_ctx *ctx, int64_t N, int32_t M, int32_t ld, const double *x,
                         const double *z, double ell, float *out);
/* agpl_transform_features: out[:, i] = A in[:, i]  with A [M, M] float32 row-major (e.g. A = L^-1 for
 *   whitening).  in/out: [M, N] column-major float32; out may not alias in.                        */
AGPL_API int32_t agpl_transform_features(agpl_ctx *ctx, int64_t N, int32_t M, const float *A,
                                const float *in, float *out);
/* agpl_synth_xy: x_i = -10 + 20 u_i, y_i ~ lik(f*(x_i)), a pure function of (seed, i) through Philox
 *   (bit-identical to oracle/agpl_oracle.c agplo_synth_*).  x_out float64 [n] or NULL.              */
AGPL_API int32_t agpl_synth_xy(agpl_ctx *ctx, const agpl_lik_desc *lik, uint64_t seed, int64_t i0, int64_t n,
                      double *x_out, void *y_out);

#ifdef __cplusplus
}
#endif
#endif /* AGPL_H */
