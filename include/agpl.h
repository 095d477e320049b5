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

#define AGPL_VERSION 121 /* 121 (round 6): same 45 entry points; a plan takes any feature count, PG(b, c) takes b < 2^22 */

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
/* waits for the context's stream; also returns what an asynchronous factorisation (agpl_gaussian_factor, agpl_plan_update) still has to report */
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
 * Limit: a point's PG(b, c) needs b = y + r (y + n) < 4194304 = 2^22 (round 6; 65535 before: draw j keeps 16 bits of the Philox
 *   counter's sub-stream id, j mod 65535, and draws beyond 65534 start their block counter at (j div 65535) << 20 -- the streams
 *   of b < 65535 are unchanged); a larger b returns AGPL_ERR_UNSUPPORTED (that point's outputs are NaN), also from
 *   agpl_gibbs_pass* and agpl_rand_polyagamma.  The reference draws any integer b (polyagamma.jl:129-134), one PG(1, c)
 *   draw after the other: 2^22 draws for ONE point are ~1 ms of a whole GPU.                                              */
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
 *   docs/src/index.md:154-163.  float64, on device: the hand-written inverse-factor kernels up to M = 2048 (S = U'U by
 *   one GEMM), rocSOLVER potrf/potri beyond and for feature counts they do not take.
 *   G [L,M,M], g [L,M], eta0 [L,M] or NULL.  Outputs (any may be NULL): S_out [L,M,M] f64,
 *   m_out [L,M] f64, Wpack_out [L,M,M] f32 (packed -S, so that var = kdiag + phi' S phi through
 *   agpl_marginals), alpha_out [L,M] f32 (= m).  Returns AGPL_ERR_NOT_POSDEF if I + G is not SPD.
 *   kl_out (device double, may be NULL): sum over latents of KL(q(v_l) || N(0, I)) = (tr S + m'm - M + logdet(I + G)) / 2 of the
 *   q(v) just formed -- the `kldivergence(u_post.approx.q, u_post.approx.fz)` term of aug_elbo (examples/bernoulli/script.jl:65-70)
 *   in the whitened basis (agpl_gaussian_kl of v110 folded in).  G = 0, g = 0 gives S = I, m = 0 and the packed -I a sweep
 *   starts from (script.jl:41-42; agpl_pack_w of v110).                                                                  */
AGPL_API int32_t agpl_gaussian_update(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                             const double *eta0, double *S_out, double *m_out, float *Wpack_out,
                             float *alpha_out, double *kl_out);


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
 *   of examples/bernoulli/script.jl:82-84 for the M inducing coordinates.  I + G = C C', U = C^-1 (the hand-written
 *   factor kernels up to M = 2048, any M: zero-padded inside; rocSOLVER beyond): m = U'(U r), v = U'(U r + z) with
 *   z_a = the normal of Philox stream (seed, l*M + a, sweep | 2^31).
 *   v_out [L,M] f64; m_out [L,M] f64 or NULL.  sweep < 2^31.                                            */
AGPL_API int32_t agpl_gibbs_draw_v(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                   const double *eta0, uint32_t sweep, double *v_out, double *m_out);

/* ---- the full-rank Gibbs step the reference executes (gibbs_sample, examples/bernoulli/script.jl:76-87) ---
 * agpl_dense_cholesky: L_out = lower Cholesky factor of the symmetric N x N matrix A (float64, rocSOLVER potrf;
 *   the `_chol_cov(fz)` of script.jl:77).  Stored in the triangle that LAPACK calls lower for a column-major
 *   array (= the upper triangle of a row-major view); the other triangle keeps A.  PosDefException -> -5.
 * agpl_dense_gibbs_step: Omega <- aux_sample!(lik, y, f) (:81); f ~ N(mu, Sigma) with
 *   Sigma = (K^-1 + Diag(gamma))^-1, mu = Sigma (beta + K^-1 mu0) (:82-84), evaluated with one Cholesky of
 *   B = I + D^1/2 K D^1/2 (B_work, N x N float64: scratch -- what it holds afterwards depends on the route) and no
 *   inverse of B -- an exact draw (Matheron's rule).  N % 1024 == 0, N >= 8192: block by block on the one-launch M x M
 *   factorisation of the sparse sweep, panels and trailing updates on the float64 matrix cores, one stream.
 *   Normals: Philox streams (seed, 0..2N-1, sweep | 2^31).  Single-latent likelihoods.  f_inout: f in, new f out. */
AGPL_API int32_t agpl_dense_cholesky(agpl_ctx *ctx, int64_t N, const double *A, double *L_out);
AGPL_API int32_t agpl_dense_gibbs_step(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, const double *K,
                                       const double *Lk, const double *mu0, const void *y, double *f_inout,
                                       double *B_work, uint32_t sweep, double *omega_out, int64_t *n_out);

/* ---- the M x M update in factor form (agpl_factor.hip) -------------------------------------------------------------------
 * I + G = R R' (Cholesky), U = R^-1 (lower triangular):  S = (I + G)^-1 = U'U,  m = S (g + eta0) = U'v,  v = U (g + eta0) --
 * the update of examples/bernoulli/script.jl:35-36 kept as (U, v), which is all the marginals need:
 *     var_n = (k_nn - |phi_n|^2) + sum_a T[a,n]^2,   mu_n = mu0_n + sum_a v_a T[a,n],   T = U Phi
 * (the same q(f_n) as agpl_marginals; reference: the marginals of q(u) = N(m, S) pushed through K_XZ K_ZZ^-1).
 * agpl_gaussian_factor : A_work [L,M,M] f64 scratch/out (on return its column-major lower triangle holds U); v_out [L,M] f64 and
 *                        logdet_out [L] f64 (device, log det(I + G)) are optional.  One hand-written launch for M <= 1024,
 *                        two block rows of it with the products on the float64 matrix cores for M <= 2048 (M % 128 == 0;
 *                        rocSOLVER beyond, and for counts the kernels do not take).  ASYNCHRONOUS: a failed factorisation (AGPL_ERR_NOT_POSDEF with the pivot row, ...)
 *                        is reported by the next agpl_gaussian_factor / agpl_plan_update / agpl_cavi_pass_plan on this context
 *                        -- after that call has enqueued its own kernels, so the host never idles the GPU inside a sweep loop
 *                        (script.jl:34-38) -- or by agpl_ctx_synchronize.  Work enqueued behind a failed factorisation computes
 *                        on NaNs; only the moment of the report moves.
 * agpl_feature_residual: resid_n = k_nn - |phi_n|^2 (static; float64 accumulation), the `resid` of agpl_plan_create.        */
AGPL_API int32_t agpl_gaussian_factor(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                      const double *eta0, double *A_work, double *v_out, double *logdet_out);
AGPL_API int32_t agpl_feature_residual(agpl_ctx *ctx, int64_t N, int32_t M, const float *Phi, const float *kdiag,
                                       float *resid_out);

/* ---- the plan: the shipped sweep path (agpl_plan.hip) -------------------------------------------------------------------
 * Split-float16 arithmetic: every float32 operand x is carried as hi = f16(s x), lo = f16(s x - hi) (s = 2^e chosen from
 * max |Phi|: error <= 2^-22 |x| down to 2^-17 max |Phi|); a float32 product runs as three float16 MFMA products (hi*hi + hi*lo
 * + lo*hi) with float32 accumulation over <= 8192 points and float64 reductions in a fixed order (bitwise reproducible, no
 * atomics).  Everything static about one data set on one context is built once: the two images of Phi (marginal image by
 * 128-point tile = the B operand of U Phi; accumulate image point-major = both operands of Phi Diag(gamma) Phi'), BOTH carrying
 * 2^e Phi with one e (domain: any finite features with max |Phi| in 2^-24 .. 2^44, else AGPL_ERR_DOMAIN; a non-finite feature is
 * AGPL_ERR_DOMAIN with its (point, feature)), a copy of the Nystrom residual, and q(v) in factor form -- U = chol(I + G)^-1,
 * v = U (g + eta0), log det(I + G) -- which the plan's update writes and its passes read; kernels are chosen by shape.  After
 * creation the float32 features are not read by any plan entry point (C2: 41 GB of images resident; 20.5 GB for a Gibbs-only plan).
 * (v110's per-generation entry points -- agpl_split_features, agpl_pack_w_split, agpl_marginals_split, agpl_cavi_pass_split,
 * agpl_pack_factor_split, agpl_marginals_factor_split, agpl_cavi_pass_factor_split, agpl_accumulate_image, agpl_accumulate_split,
 * agpl_cavi_pass_factor_image, agpl_gibbs_pass_image -- are gone in v120 with the kernels only they reached.)
 *   feature counts      : ANY M >= 1 (round 6; rounds 3-5 asked the caller to zero-pad to a multiple of 256).  The plan works on
 *                         Mp = M rounded up to a multiple of 256: its images carry zero features M .. Mp - 1, so G and g are zero
 *                         there, I + G is the identity there, and U = chol(I + G)^-1, v are the caller's in their leading M x M /
 *                         M block (identity / zero beyond).  Every array the CALLER passes or receives (Phi, G, g, eta0, the
 *                         Gibbs draw v) is M-sized; only the state arrays of agpl_plan_state are Mp-sized.  For M == Mp nothing
 *                         changes; otherwise a pass / update costs two small copy kernels more.
 *   agpl_plan_bytes     : device bytes a plan needs for (N, M, L, flags); 0 for sizes a plan does not take (N, M < 1, L > 64).
 *   agpl_plan_create    : Phi float32 [M, N] column-major (M contiguous floats per point, 16-byte aligned base; M % 4 != 0 is read
 *                         element-wise); resid float32 [N] (agpl_feature_residual); flags: 0, or
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
 *                         (script.jl:32-36 up to the M x M solve; agpl_cavi_pass's contract) in three launches up to the slabs
 *                         (marginal partial sums; ONE per-point kernel; the accumulation) -- gamma_out / beta_out / c_out may be
 *                         NULL and are then never materialised.  A gamma that is negative or not finite (observations or marginals
 *                         outside the likelihood's domain; TestUtils.jl:88) is AGPL_ERR_DOMAIN with its flat index, reported by the
 *                         call that reports the outcome of the update enqueued behind this pass.  elbo_terms_out (device double,
 *                         may be NULL): sum over the points of expected_logtilt_i - aux_kldivergence_i for the q(v) the pass used,
 *                         from the same marginals in float64 -- the per-point part of aug_elbo (script.jl:65-70) rides the pass's
 *                         one per-point kernel (AGPL_ERR_UNSUPPORTED for the non-bijective categorical and the heteroscedastic
 *                         likelihood, whose terms the reference does not define).
 *   agpl_plan_update    : q(v) <- N(S (g + eta0), S), S = (I + G)^-1 (script.jl:35-36), kept as (U, v); asynchronous, outcome
 *                         reported as by agpl_gaussian_factor.  kl_out (device double, may be NULL): KL(q(v) || N(0, I)) of
 *                         the NEW q(v), from U and v (the kldivergence term of aug_elbo).  ELBO of a q(v) = the elbo_terms of the
 *                         pass that used it (summed over ranks) - the kl of the update that made it.
 *   agpl_marginals_plan : q(f_i) of the plan's q(v): mu, var float32 [L][N].
 *   agpl_gibbs_pass_plan: agpl_gibbs_pass with the plan's residual and accumulate image; the projection phi_i' v is formed from the
 *                         image too (x = (hi + lo) 2^-e: the features to 2^-22 relative, float64 accumulation in a fixed order),
 *                         so a Gibbs chain needs the float32 features at plan creation only.
 *   agpl_plan_info      : sizes (M: the caller's), the images' scale exponent, bytes.
 *   agpl_plan_state     : device pointers to everything an update rewrites -- U (float64 [L, Mp, Mp], column-major lower triangle,
 *                         Mp = M rounded up to a multiple of 256: the caller's U is the leading M x M block, leading dimension Mp),
 *                         v (float64 [L, Mp]), the float16 images of 2^15 U (L Mp Mp halves each), v as float32 [L, Mp],
 *                         log det(I + G) [L] -- and to the plan's residual copy; any out pointer may be NULL.  To form S = U'U,
 *                         m = U'v; for repeatability checks; and as the CHECKPOINT of a sweep loop: copy the five arrays out
 *                         (with the context's seed and the host's sweep counter), and into the same pointers of a plan created
 *                         on the same data to resume bit for bit (tests/test_gpu_checkpoint.py). */
typedef struct agpl_plan agpl_plan;
#define AGPL_PLAN_NO_MARGINALS 1u
AGPL_API int64_t agpl_plan_bytes(int64_t N, int32_t M, int32_t L, uint32_t flags);
AGPL_API int32_t agpl_plan_create(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi, const float *resid,
                                  uint32_t flags, void *storage, agpl_plan **plan_out);
AGPL_API int32_t agpl_plan_destroy(agpl_plan *plan);
AGPL_API int32_t agpl_plan_info(const agpl_plan *plan, int64_t *N, int32_t *M, int32_t *L, int32_t *scale_exp, int64_t *bytes);
AGPL_API int32_t agpl_plan_state(const agpl_plan *plan, double **U_out, double **v_out, void **U_hi_out, void **U_lo_out,
                                 float **v32_out, double **logdet_out, const float **resid_out);
AGPL_API int32_t agpl_cavi_pass_plan(agpl_plan *plan, const agpl_lik_desc *lik, const float *mu0, const void *y, double *G_out,
                                     double *g_out, float *c_out, float *gamma_out, float *beta_out, double *elbo_terms_out);
AGPL_API int32_t agpl_plan_update(agpl_plan *plan, const double *G, const double *g, const double *eta0, double *kl_out);
AGPL_API int32_t agpl_marginals_plan(agpl_plan *plan, const float *mu0, float *mu_out, float *var_out);
AGPL_API int32_t agpl_gibbs_pass_plan(agpl_plan *plan, const agpl_lik_desc *lik, const float *mu0, const void *y,
                                      const double *v, uint32_t sweep, double *G_out, double *g_out, double *f_out,
                                      double *omega_out, int64_t *n_out, uint32_t *nuni_out);

/* ---- diagnostics (no reference counterpart) -----------------------------------------------------------------------------
 * agpl_probe_mfma: measured matrix-core rate of this device [TFLOP/s]; synchronous; ms_host (may be NULL) = average launch time.
 *   dtype AGPL_F64: v_mfma_f64_16x16x4_f64 back to back on every SIMD, `iters` x 16 instructions per wave (mode, workgroups_per_cu
 *     ignored): the peak bench.py prices the C5 Cholesky against ("FP64 peak is not in the local guide").
 *   dtype AGPL_F32 (float16 operands, float32 accumulate): the SUSTAINED rate of v_mfma_f32_32x32x16_f16 in the instruction mix
 *     of one stage of the accumulation's wave (12 per step into four 32 x 32 accumulators), six launches of `iters` steps per wave
 *     timed as one region after two that settle the clocks, `workgroups_per_cu` (1..4) 4-wave workgroups per CU.  mode 0: MFMA only;
 *     1: with that stage's eight 16-byte LDS fragment reads; 2, 3: the same for v_mfma_f32_16x16x32_f16 in the mix of one stage of
 *     the marginal kernel's wave (48 per step into sixteen 16 x 16 accumulators, hashed operands, <= 2 workgroups per CU; 3: with
 *     the stage's sixteen fragment reads).  The ceiling bench.py reports next to the data-sheet peak.
 * agpl_timing: in-library hipEvent timing of the hot kernels (bench.py's roofline leg).  which = -1 / -2: switch it on / off
 *   (outputs ignored); which = 0 marginal, 1 accumulation, 2 Gibbs point pass, 3 aux_sample kernel: synchronises the stream, returns
 *   the summed kernel time [ms] and the number of launches since the last read, and resets the counters.
 * agpl_debug_force_factor_rescue: test hook.  The one-launch M x M factorisation runs as cooperating workgroups that assume each
 *   other resident; should other work hold their CUs, the launch reports it and a second launch queued behind it redoes the latent
 *   in one workgroup, without the host.  on != 0 makes every cooperative launch of this context take that path at once, so that
 *   it can be tested; the results are the cooperative launch's. */
AGPL_API int32_t agpl_probe_mfma(agpl_ctx *ctx, int32_t dtype, int32_t iters, int32_t mode, int32_t workgroups_per_cu,
                                 double *tflops_host, double *ms_host);
AGPL_API int32_t agpl_timing(agpl_ctx *ctx, int32_t which, double *total_ms_host, int64_t *launches_host);
AGPL_API int32_t agpl_debug_force_factor_rescue(agpl_ctx *ctx, int32_t on);

/* agpl_allreduce_nat: the exchange step of the N-sharded sweep: in-place float64 sum of the
 *   L (M^2 + M) natural-parameter accumulators over an RCCL communicator (ncclComm_t as void*), queued on the
 *   context's stream.  For hosts that own their communicator (the Julia / C++ callers of INTEGRATION.md); the
 *   Python host reaches the same RCCL through torch.distributed.  librccl is loaded at the first call.          */
AGPL_API int32_t agpl_allreduce_nat(agpl_ctx *ctx, void *rccl_comm, double *buf, int64_t count);

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
