/*
 * agpl_oracle.c -- CPU restatement (plain C, Float64) of the inner inference loop of
 * AugmentedGPLikelihoods.jl v0.4.19.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (libagpl.so, HIP) never links,
 * includes or calls anything in oracle/.
 *
 * Parity status (see DESIGN.md "Oracle"): the reference is pure Julia and no Julia toolchain
 * exists in the build image, so the reference cannot be executed here and its tests hold no golden
 * vectors.  The restatement is pinned against every known-answer the reference
 * tests do hold (closed-form PG means, the hard-coded r(z=0), sample-mean-within-1e-2, the
 * density-series cross check, approx_expected_logistic saturation, the two full-conditional
 * identities of src/TestUtils.jl) -- tests/test_oracle_pins.py.  Sample *values* are "parity
 * unpinned" against Julia by construction: Julia draws from Xoshiro256++/MersenneTwister, this code
 * (and the HIP path) from Philox4x32-10, so equality with Julia is distributional only.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Third-party algorithms that are NOT under /root/reference (Distributions.jl 0.25 samplers,
 * StatsFuns normlogcdf, LogExpFunctions _logistic_bounds) are restated from their published
 * algorithms and marked "upstream, unpinned".
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define AGPLO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* constants: src/SpecialDistributions/polyagamma.jl:3-4                                      */
/* ------------------------------------------------------------------------------------------ */
static const double PG_T = 0.64;
static const double PI_ = 3.14159265358979323846;
#define PI2_8 (PI_ * PI_ / 8.0)
static const double LOGTWO = 0.69314718055994530942;
static const double LOG2PI = 1.83787706640934548356;

/* likelihood kinds -- mirror of include/agpl.h agpl_lik_kind (kept in sync by tests/test_abi.py) */
enum {
    LIK_BERNOULLI_LOGISTIC = 0,  /* src/likelihoods/bernoulli.jl */
    LIK_NEGBINOMIAL = 1,         /* src/likelihoods/negativebinomial.jl ; p[0] = failures r */
    LIK_STUDENTT = 2,            /* src/likelihoods/studentt.jl ; p[0] = nu, p[1] = sigma */
    LIK_CATEGORICAL = 3,         /* src/likelihoods/categorical.jl non-bijective; nlatent = K */
    LIK_CATEGORICAL_BIJ = 4,     /* bijective: nlatent = K-1, logtheta has K entries */
    LIK_POISSON = 5,             /* src/likelihoods/poisson.jl ; p[0] = lambda */
    LIK_LAPLACE = 6,             /* src/likelihoods/laplace.jl ; p[0] = beta */
    LIK_HETEROGAUSS = 7          /* src/likelihoods/heteroscedasticgaussian.jl ; p[0] = lambda */
};

typedef struct {
    int32_t kind;
    int32_t nlatent;
    double p[4];
    const double *logtheta; /* categorical only; K entries (K = nlatent, or nlatent+1 if bijective) */
} agplo_lik;

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 counter RNG (Salmon et al. 2011).  Replaces Julia's GLOBAL_RNG                */
/* (src/generic.jl:1-3).  One independent stream per (seed, element index, sweep).             */
/* ------------------------------------------------------------------------------------------ */
AGPLO_API void agplo_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2],
                                   uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
    uint32_t key[2];
    uint32_t ctr[4];
    uint32_t buf[4];
    int pos;
    uint32_t nuni; /* number of uniforms consumed so far: the integer bookkeeping compared bit-exactly */
} agplo_rng;

static void rng_init(agplo_rng *g, uint64_t seed, uint64_t stream, uint32_t sweep) {
    g->key[0] = (uint32_t)seed;
    g->key[1] = (uint32_t)(seed >> 32);
    g->ctr[0] = 0;
    g->ctr[1] = sweep;
    g->ctr[2] = (uint32_t)stream;
    g->ctr[3] = (uint32_t)(stream >> 32);
    g->pos = 4;
    g->nuni = 0;
}

/* uniform in (0,1): (k + 1/2) 2^-52 with 52 random bits -- exact in float64, never 0 or 1 (Julia's rand() is
 * [0,1) on a 2^-52 grid; the half-step shift only removes log(0)). */
static double rng_u01(agplo_rng *g) {
    if (g->pos >= 4) {
        agplo_philox4x32_10(g->ctr, g->key, g->buf);
        g->ctr[0] += 1u;
        g->pos = 0;
    }
    uint32_t w0 = g->buf[g->pos], w1 = g->buf[g->pos + 1];
    g->pos += 2;
    g->nuni += 1u;
    uint64_t k = ((uint64_t)(w0 >> 6) << 26) | (uint64_t)(w1 >> 6);
    return ((double)k + 0.5) * 0x1.0p-52;
}

/* randexp(rng): Exp(1) by inversion (Julia uses a ziggurat; distribution identical). */
static double rng_exp(agplo_rng *g) { return -log(rng_u01(g)); }

/* randn(rng): Box-Muller, cosine branch only, 2 uniforms per normal (Julia: ziggurat). */
static double rng_normal(agplo_rng *g) {
    double u1 = rng_u01(g);
    double u2 = rng_u01(g);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * PI_ * u2);
}

/* ------------------------------------------------------------------------------------------ */
/* upstream, unpinned: StatsFuns.normlogcdf (called at polyagamma.jl:186-187)                  */
/*   z < -1 : log(erfcx(-z/sqrt2)/2) - z^2/2 ;  else log1p(-erfc(z/sqrt2)/2)                   */
/* erfcx is not in C libm: log(erfc(t)/2) is used while erfc(t) is a normal double, and the     */
/* 5-term asymptotic expansion of log Phi beyond (|z| >= 35; truncation error < 1e-12).         */
/* ------------------------------------------------------------------------------------------ */
AGPLO_API double agplo_normlogcdf(double z) {
    if (z < -1.0) {
        if (z > -35.0) return log(0.5 * erfc(-z * M_SQRT1_2));
        double iz2 = 1.0 / (z * z);
        double ser = 1.0 - iz2 * (1.0 - 3.0 * iz2 * (1.0 - 5.0 * iz2 * (1.0 - 7.0 * iz2)));
        return -0.5 * z * z - log(-z) - 0.5 * LOG2PI + log(ser);
    }
    return log1p(-0.5 * erfc(z * M_SQRT1_2));
}

/* ------------------------------------------------------------------------------------------ */
/* Polya-Gamma: src/SpecialDistributions/polyagamma.jl                                         */
/* ------------------------------------------------------------------------------------------ */

/* mean(::PolyaGamma) polyagamma.jl:25-31 */
AGPLO_API double agplo_pg_mean(double b, double c) {
    if (c == 0.0) return b / 4.0;
    return b / (2.0 * c) * tanh(c / 2.0);
}

/* logtilt(omega,b,c) polyagamma.jl:108-110 ; logcosh from LogExpFunctions:
 * logcosh(x) = |x| + log1p(exp(-2|x|)) - log 2 */
static double logcosh_(double x) {
    double ax = fabs(x);
    return ax + log1p(exp(-2.0 * ax)) - LOGTWO;
}
AGPLO_API double agplo_pg_logtilt(double omega, double b, double c) {
    return b * logcosh_(c / 2.0) - c * c * omega / 2.0;
}

/* kldivergence(PG(b,c) || PG(b,0)) polyagamma.jl:99-106 */
AGPLO_API double agplo_pg_kl(double b, double c) {
    return agplo_pg_logtilt(agplo_pg_mean(b, c), b, c);
}

/* a(n,x) polyagamma.jl:167-177.  x <= 0 is a DomainError in the reference: NaN here. */
AGPLO_API double agplo_pg_a(int n, double x) {
    double k = (n + 0.5) * PI_;
    if (x > PG_T) return k * exp(-k * k * x / 2.0);
    if (x > 0.0) {
        double expnt = -3.0 / 2.0 * (log(PI_ / 2.0) + log(x)) - 2.0 * (n + 0.5) * (n + 0.5) / x;
        return k * exp(expnt);
    }
    return NAN;
}

/* mass_texpon(z,K) polyagamma.jl:179-192 */
AGPLO_API double agplo_pg_mass_texpon(double z, double K) {
    double t = PG_T;
    double b = sqrt(1.0 / t) * (t * z - 1.0);
    double a = -sqrt(1.0 / t) * (t * z + 1.0);
    double x0 = log(K) + K * t;
    double xb = x0 - z + agplo_normlogcdf(b);
    double xa = x0 + z + agplo_normlogcdf(a);
    double qdivp = (4.0 / PI_) * (exp(xb) + exp(xa));
    return 1.0 / (1.0 + qdivp);
}

/* rand_truncated_inverse_gaussian(rng,z) polyagamma.jl:195-221 */
static double rand_tig(agplo_rng *g, double z) {
    double mu = 1.0 / z;
    double x = 1.0 + PG_T;
    if (mu > PG_T) {
        double alpha = 0.0;
        while (alpha < rng_u01(g)) {
            double E = rng_exp(g);
            double Ep = rng_exp(g);
            while (E * E > (2.0 * Ep / PG_T)) {
                E = rng_exp(g);
                Ep = rng_exp(g);
            }
            double d = 1.0 + E * PG_T;
            x = PG_T / (d * d);
            alpha = exp(-z * z * x / 2.0);
        }
    } else {
        while (x > PG_T) {
            double nrm = rng_normal(g);
            double y = nrm * nrm;
            double muy = mu * y;
            x = mu + mu * muy / 2.0 - mu * sqrt(4.0 * muy + muy * muy) / 2.0;
            if (mu / (mu + x) < rng_u01(g)) x = mu * mu / x;
        }
    }
    return x;
}

/* sample_pg1(rng,c) polyagamma.jl:225-257.  *nterms accumulates the series index n at exit. */
static double sample_pg1(agplo_rng *g, double c, uint32_t *nterms) {
    double z = fabs(c) / 2.0;
    double r, K;
    if (z == 0.0) {
        r = 0.5776972428360435; /* polyagamma.jl:231 */
        K = PI2_8;
    } else {
        K = PI2_8 + z * z / 2.0;
        r = agplo_pg_mass_texpon(z, K);
    }
    for (;;) {
        double x;
        if (r > rng_u01(g))
            x = PG_T + rng_exp(g) / K;
        else
            x = rand_tig(g, z);
        double s = agplo_pg_a(0, x);
        double y = rng_u01(g) * s;
        int n = 0;
        for (;;) {
            n += 1;
            if (n & 1) {
                s -= agplo_pg_a(n, x);
                if (y <= s) {
                    *nterms += (uint32_t)n;
                    return x / 4.0;
                }
            } else {
                s += agplo_pg_a(n, x);
                if (y > s) break;
            }
        }
        *nterms += (uint32_t)n;
    }
}

/* upstream, unpinned: Distributions.jl 0.25 Gamma samplers.  shape >= 1: Marsaglia-Tsang (2000)
 * (GammaMTSampler); shape < 1: draw at shape+1 and multiply by exp(-E/shape) (GammaIPSampler). */
static double rand_gamma_mt(agplo_rng *g, double shape) {
    double d = shape - 1.0 / 3.0;
    double c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = rng_normal(g);
        double v = 1.0 + c * x;
        while (v <= 0.0) {
            x = rng_normal(g);
            v = 1.0 + c * x;
        }
        v = v * v * v;
        double u = rng_u01(g);
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2 || log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v;
    }
}
static double rand_gamma(agplo_rng *g, double shape) {
    if (!(shape > 0.0)) return NAN; /* NaN / invalid shape: no draw (the rejection loops would not end) */
    if (shape >= 1.0) return rand_gamma_mt(g, shape);
    double x = rand_gamma_mt(g, shape + 1.0);
    double e = rng_exp(g);
    return x * exp(-e / shape);
}

/* rand_gamma_sum(rng,d,e) polyagamma.jl:157-164 (200-term truncated Gamma series, non-integer b) */
static double rand_gamma_sum(agplo_rng *g, double c, double e) {
    double inv2pi2 = (1.0 / (2.0 * PI_)) * (1.0 / PI_);
    double w = (c * (1.0 / (2.0 * PI_)));
    w = w * w;
    double acc = 0.0;
    for (int k = 1; k <= 200; ++k) acc += rand_gamma(g, e) / ((k - 0.5) * (k - 0.5) + w);
    return inv2pi2 * acc;
}

/* rand(PolyaGamma(b, c)) polyagamma.jl:121-154: draw_sum of floor(b) PG(1, c) draws (:129-134) plus, for real b, the
 * truncated Gamma series on the residual (:137-154; identical draws when the residual is zero).
 * STREAM LAYOUT (shared bit for bit with agpl_random.h): every PG(1, c) draw lives on a SUB-STREAM of its point's
 * Philox stream -- counter word 3 carries (point index >> 32) in its low 8 bits and the sub-stream id above them;
 * draw j of latent k uses id 1 + (k << 16) + (j mod 65535) and starts its block counter (counter word 0) at
 * (j div 65535) << 20 (zero for j < 65535: the layout of rounds 2-5; round 6 lifted b < 65535 to b < 2^22 this way --
 * polyagamma.jl:129-134 sums any b), the residual series id 1 + (k << 16) + 0xFFFF; id 0 is the point's
 * main stream (noise, Gamma / Poisson / inverse-Gaussian draws).  The draws of one point are therefore independent
 * work items (the device deals them across the lanes of a wave), summed left to right in draw order, residual last. */
#define AGPLO_SUB_RESIDUAL 0xFFFFu
#define AGPLO_PG_MAX_B 4194304.0
static void rng_sub_at(const agplo_rng *g, uint32_t id, uint32_t block0, agplo_rng *s) {
    *s = *g;
    s->ctr[0] = block0;
    s->ctr[3] = (g->ctr[3] & 0xFFu) + (id << 8);
    s->pos = 4;
    s->nuni = 0;
}
static void rng_sub(const agplo_rng *g, uint32_t id, agplo_rng *s) {
    *s = *g;
    s->ctr[0] = 0;
    s->ctr[3] = (g->ctr[3] & 0xFFu) + (id << 8);
    s->pos = 4;
    s->nuni = 0;
}
static double rand_pg(agplo_rng *g, int latent, double b, double c, uint32_t *nterms) {
    /* NaN / Inf in, NaN out: the accept loops never terminate on a non-finite tilt (the reference would spin
     * or throw its DomainError from a(n, 0), polyagamma.jl:175) */
    if (!(b >= 0.0) || !(fabs(c) < INFINITY) || !(b < AGPLO_PG_MAX_B)) return NAN;
    if (b == 0.0) return 0.0;
    const long tb = (long)floor(b);
    const uint32_t base = 1u + ((uint32_t)latent << 16);
    double acc = 0.0;
    agplo_rng s;
    for (long j = 0; j < tb; ++j) {
        rng_sub_at(g, base + (uint32_t)(j % 65535), (uint32_t)(j / 65535) << 20, &s);
        acc += sample_pg1(&s, c, nterms);
        g->nuni += s.nuni;
    }
    double res = b - (double)tb;
    if (res == 0.0) return acc;
    rng_sub(g, base + AGPLO_SUB_RESIDUAL, &s);
    acc += rand_gamma_sum(&s, c, res);
    g->nuni += s.nuni;
    return acc;
}

/* upstream, unpinned: Distributions.jl 0.25 Poisson.  mu < 6: PoissonCountSampler (count unit-rate
 * exponential arrivals).  mu >= 6: the reference uses Ahrens-Dieter PD; restated here with
 * Hoermann's PTRS transformed rejection (1993) -- same distribution, DOCUMENTED DEVIATION. */
static int64_t rand_poisson(agplo_rng *g, double mu) {
    if (!(mu > 0.0)) return 0;
    if (mu < 6.0) {
        int64_t n = 0;
        double c = rng_exp(g);
        while (c < mu) {
            n += 1;
            c += rng_exp(g);
        }
        return n;
    }
    double slam = sqrt(mu), loglam = log(mu);
    double b = 0.931 + 2.53 * slam;
    double a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    double vr = 0.9277 - 3.6224 / (b - 2.0);
    for (;;) {
        double U = rng_u01(g) - 0.5;
        double V = rng_u01(g);
        double us = 0.5 - fabs(U);
        double kf = floor((2.0 * a / us + b) * U + mu + 0.43);
        if (us >= 0.07 && V <= vr) return (int64_t)kf;
        if (kf < 0.0 || (us < 0.013 && V > us)) continue;
        if ((log(V) + log(invalpha) - log(a / (us * us) + b)) <=
            (-mu + kf * loglam - lgamma(kf + 1.0)))
            return (int64_t)kf;
    }
}

/* upstream, unpinned: Distributions.jl 0.25 rand(InverseGaussian(mu,lambda)):
 * Michael-Schucany-Haas (1976). Called at src/likelihoods/laplace.jl:40-42. */
static double rand_invgaussian(agplo_rng *g, double mu, double lambda) {
    double z = rng_normal(g);
    double v = z * z;
    double w = mu * v;
    double x1 = mu + mu / (2.0 * lambda) * (w - sqrt(w * (4.0 * lambda + w)));
    double p1 = mu / (mu + x1);
    double u = rng_u01(g);
    return u >= p1 ? mu * mu / x1 : x1;
}

/* logpdf(PolyaGamma(b,c), x) polyagamma.jl:37-91 (off the hot path; used by aug_loglik and as
 * the KS oracle for the sampler).  Reference quirk kept: x == 0 falls through (Appendix B). */
static double pg_calc_series(double x, double b, int max_half_n) {
    int max_n = 2 * max_half_n;
    double prod = 1.0, acc = 0.0;
    /* series_m_prods[n] = prod_{m=1..n} (1 + (b-1)/m) */
    int m = 0;
    for (int n = 0; n <= max_n; n += 2) {
        while (m < n) {
            m += 1;
            prod *= 1.0 + (b - 1.0) / m;
        }
        double Rn = 2.0 * n + b;
        double exp_out = exp(Rn * Rn / (-8.0 * x));
        double c_nb = ((n + b) / (n + 1.0)) * (2.0 / Rn + 1.0);
        double inner = 1.0 - c_nb * exp((Rn + 1.0) / (-2.0 * x));
        acc += (n == 0 ? 1.0 : prod) * Rn * exp_out * inner;
    }
    return acc;
}
static double log1mexp_(double x) { /* LogExpFunctions.log1mexp, x < 0 */
    return x < -LOGTWO ? log1p(-exp(x)) : log(-expm1(x));
}
static double pg_calc_log_series(double x, double b, int max_half_n) {
    int max_n = 2 * max_half_n;
    int cnt = max_half_n + 1;
    double *lo = (double *)malloc(sizeof(double) * cnt);
    double logprod = 0.0, mx = -INFINITY;
    int m = 0, j = 0;
    for (int n = 0; n <= max_n; n += 2, ++j) {
        while (m < n) {
            m += 1;
            logprod += log(1.0 + (b - 1.0) / m);
        }
        double Rn = 2.0 * n + b;
        double log_exp_out = Rn * Rn / (-8.0 * x);
        double log_c_nb = log(n + b) - log(n + 1.0) + log(2.0 / Rn + 1.0);
        double log_inner = log1mexp_(log_c_nb + ((Rn + 1.0) / (-2.0 * x)));
        lo[j] = (n == 0 ? 0.0 : logprod) + log(Rn) + log_exp_out + log_inner;
        if (lo[j] > mx) mx = lo[j];
    }
    double s = 0.0;
    for (j = 0; j < cnt; ++j) s += exp(lo[j] - mx);
    free(lo);
    return mx + log(s);
}
AGPLO_API double agplo_pg_logpdf(double b, double c, double x) {
    if (b == 0.0) return x == 0.0 ? 0.0 : -INFINITY;
    double ext = agplo_pg_logtilt(x, b, c) + (b - 1.0) * LOGTWO - (LOG2PI + 3.0 * log(x)) / 2.0;
    if (x < 1e-2) return ext + pg_calc_log_series(x, b, 100);
    double ss = pg_calc_series(x, b, 100);
    if (!(ss > 2.2250738585072014e-308)) ss = 2.2250738585072014e-308; /* max(s, floatmin) */
    return ext + log(ss);
}

/* scalar entry points for the pins / distribution tests */
AGPLO_API void agplo_rand_pg_many(double b, double c, int64_t n, uint64_t seed, double *out,
                                  uint32_t *nuni_out, uint32_t *nterms_out) {
    for (int64_t i = 0; i < n; ++i) {
        agplo_rng g;
        rng_init(&g, seed, (uint64_t)i, 0);
        uint32_t nt = 0;
        out[i] = rand_pg(&g, 0, b, c, &nt);
        if (nuni_out) nuni_out[i] = g.nuni;
        if (nterms_out) nterms_out[i] = nt;
    }
}
AGPLO_API void agplo_rand_gamma_many(double shape, double scale, int64_t n, uint64_t seed,
                                     double *out) {
    for (int64_t i = 0; i < n; ++i) {
        agplo_rng g;
        rng_init(&g, seed, (uint64_t)i, 0);
        out[i] = scale * rand_gamma(&g, shape);
    }
}
AGPLO_API void agplo_rand_poisson_many(double mu, int64_t n, uint64_t seed, int64_t *out) {
    for (int64_t i = 0; i < n; ++i) {
        agplo_rng g;
        rng_init(&g, seed, (uint64_t)i, 0);
        out[i] = rand_poisson(&g, mu);
    }
}
AGPLO_API void agplo_rand_invgaussian_many(double mu, double lambda, int64_t n, uint64_t seed,
                                           double *out) {
    for (int64_t i = 0; i < n; ++i) {
        agplo_rng g;
        rng_init(&g, seed, (uint64_t)i, 0);
        out[i] = rand_invgaussian(&g, mu, lambda);
    }
}
AGPLO_API void agplo_uniforms(uint64_t seed, uint64_t stream, uint32_t sweep, int64_t n,
                              double *out) {
    agplo_rng g;
    rng_init(&g, seed, stream, sweep);
    for (int64_t i = 0; i < n; ++i) out[i] = rng_u01(&g);
}

/* ------------------------------------------------------------------------------------------ */
/* src/utils.jl                                                                                */
/* ------------------------------------------------------------------------------------------ */
/* second_moment utils.jl:1-7 */
static double second_moment(double mu, double var) { return mu * mu + var; }
static double second_moment_y(double mu, double var, double y) {
    return (mu - y) * (mu - y) + var;
}
/* approx_expected_logistic utils.jl:11-14; bounds = LogExpFunctions._logistic_bounds(::Float64)
 * (upstream, unpinned): (-744.4400719213812, 36.7368005696771) */
AGPLO_API double agplo_approx_expected_logistic(double mu, double c) {
    if (mu < -744.4400719213812) return 0.0;
    if (mu > 36.7368005696771) return 1.0;
    return exp(mu / 2.0) * (1.0 / cosh(c / 2.0)) / 2.0;
}
/* Float32 variant exercised by test/utils.jl:9-13; bounds (-103.27893f0, 16.635532f0) */
AGPLO_API float agplo_approx_expected_logistic_f32(float mu, float c) {
    if (mu < -103.27893f) return 0.0f;
    if (mu > 16.635532f) return 1.0f;
    return expf(mu / 2.0f) * (1.0f / coshf(c / 2.0f)) / 2.0f;
}
static double logistic_(double x) { return 1.0 / (1.0 + exp(-x)); }

/* ------------------------------------------------------------------------------------------ */
/* categorical link helpers: src/likelihoods/categorical.jl:12-30                              */
/* ------------------------------------------------------------------------------------------ */
static double cat_get_const(const agplo_lik *lik) { /* :12-14 */
    return exp(lik->logtheta[lik->nlatent]) * 0.5;
}
static double cat_sum_theta(const agplo_lik *lik) { /* :16-20 */
    double s = 0.0;
    for (int k = 0; k < lik->nlatent; ++k) s += exp(lik->logtheta[k]);
    if (lik->kind == LIK_CATEGORICAL_BIJ) s += cat_get_const(lik);
    return s;
}

/* ------------------------------------------------------------------------------------------ */
/* aux_sample!(rng, Omega, lik, y, f)  src/generic.jl:5-12, with aux_full_conditional of each   */
/* likelihood.  y layout: u8 (bernoulli, categorical one-hot [L,N]), i32 (negbin, poisson),     */
/* f64 (studentt, laplace, heterogauss).  f: [N] or [L,N] column-major (L contiguous per point) */
/* as categorical.jl:52-57.  omega_out: same shape as f ([N] for heterogauss); n_out: i64       */
/* (categorical [L,N], poisson/heterogauss [N]) or NULL.                                       */
/* nuni_out (u32[N], optional): uniforms consumed per point; nterms_out: summed series index.   */
/* ------------------------------------------------------------------------------------------ */
/* Global index of local point 0 for the per-point streams of agplo_aux_sample / agplo_gibbs_points (the twin of
 * agpl_ctx_set_point_offset: a shard [i0, i1) of N points draws on the streams (seed, i0 + i, sweep)). */
static int64_t g_point_offset = 0;
AGPLO_API void agplo_set_point_offset(int64_t i0) { g_point_offset = i0; }

AGPLO_API int agplo_aux_sample(const agplo_lik *lik, int64_t n, const void *yv, const double *f,
                               double *omega, int64_t *nout, uint64_t seed, uint32_t sweep,
                               uint32_t *nuni_out, uint32_t *nterms_out) {
    const int L = lik->nlatent;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < n; ++i) {
        agplo_rng g;
        rng_init(&g, seed, (uint64_t)(g_point_offset + i), sweep);
        uint32_t nt = 0;
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC: /* bernoulli.jl:13-15  PG(1,|f|) */
            omega[i] = rand_pg(&g, 0, 1.0, fabs(f[i]), &nt);
            break;
        case LIK_NEGBINOMIAL: { /* negativebinomial.jl:20-22  PG(y+r,|f|) */
            const int32_t *y = (const int32_t *)yv;
            omega[i] = rand_pg(&g, 0, (double)y[i] + lik->p[0], fabs(f[i]), &nt);
        } break;
        case LIK_STUDENTT: { /* studentt.jl:46-48  Gamma((nu+1)/2, scale 2/(nu/sigma^2+(y-f)^2)) */
            const double *y = (const double *)yv;
            double nu = lik->p[0], sg = lik->p[1];
            double d = y[i] - f[i];
            double scale = 2.0 / (nu / (sg * sg) + d * d);
            omega[i] = scale * rand_gamma(&g, (nu + 1.0) / 2.0);
        } break;
        case LIK_CATEGORICAL:
        case LIK_CATEGORICAL_BIJ: { /* categorical.jl:72-78 ; PGNM ntrand
                                       polyagammanegativemultinomial.jl:27-31 ;
                                       NegativeMultinomial _rand! negativemultinomial.jl:35-45 */
            const uint8_t *y = (const uint8_t *)yv;
            double sumth = cat_sum_theta(lik);
            double sp = 0.0;
            for (int k = 0; k < L; ++k) sp += exp(lik->logtheta[k]) * logistic_(f[i * L + k]) / sumth;
            double p0 = 1.0 - sp;
            if (!(sp < 1.0)) { bad |= 1; break; } /* ArgumentError negativemultinomial.jl:17-22 */
            double theta = (1.0 / p0 - 1.0) * rand_gamma(&g, 1.0); /* x0 = 1 */
            for (int k = 0; k < L; ++k) {
                double pk = exp(lik->logtheta[k]) * logistic_(f[i * L + k]) / sumth;
                double lam = pk * theta / (1.0 - p0);
                nout[i * L + k] = rand_poisson(&g, lam);
            }
            for (int k = 0; k < L; ++k)
                omega[i * L + k] =
                    rand_pg(&g, k, (double)(nout[i * L + k] + (int64_t)y[i * L + k]), fabs(f[i * L + k]), &nt);
        } break;
        case LIK_POISSON: { /* poisson.jl:26-28 ; PGPoisson ntrand polyagammapoisson.jl:23-27 */
            const int32_t *y = (const int32_t *)yv;
            double lam = lik->p[0] * logistic_(-f[i]);
            int64_t nn = rand_poisson(&g, lam);
            nout[i] = nn;
            omega[i] = rand_pg(&g, 0, (double)(nn + y[i]), fabs(f[i]), &nt);
        } break;
        case LIK_LAPLACE: { /* laplace.jl:40-42  IG(1/(2 beta |y-f|), 2 * (2 beta)^-2) */
            const double *y = (const double *)yv;
            double beta = lik->p[0];
            double lam = 1.0 / ((2.0 * beta) * (2.0 * beta));
            omega[i] = rand_invgaussian(&g, 1.0 / (2.0 * beta * fabs(y[i] - f[i])), 2.0 * lam);
        } break;
        case LIK_HETEROGAUSS: { /* heteroscedasticgaussian.jl:28-32 ; f = fg[2i], g = fg[2i+1] */
            const double *y = (const double *)yv;
            double ff = f[2 * i], gg = f[2 * i + 1];
            double lam = lik->p[0] * logistic_(-gg) * (ff - y[i]) * (ff - y[i]) / 2.0;
            int64_t nn = rand_poisson(&g, lam);
            nout[i] = nn;
            omega[i] = rand_pg(&g, 0, 0.5 + (double)nn, fabs(gg), &nt);
        } break;
        default:
            bad |= 2;
        }
        if (nuni_out) nuni_out[i] = g.nuni;
        if (nterms_out) nterms_out[i] = nt;
    }
    return bad ? -bad : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* aux_posterior!(qOmega, lik, y, qf)                                                          */
/*   bernoulli.jl:17-25, negativebinomial.jl:24-33, studentt.jl:50-58, categorical.jl:80-110,   */
/*   poisson.jl:30-39, laplace.jl:44-52, heteroscedasticgaussian.jl:34-46                       */
/* qf = (mu, var) SoA.  out1 = c (or beta_i for studentt, mu_i for laplace);                    */
/* out2 = p [L,N] (categorical), lambda (poisson, heterogauss) ; out3 = psi (heterogauss)       */
/* ------------------------------------------------------------------------------------------ */
AGPLO_API int agplo_aux_posterior(const agplo_lik *lik, int64_t n, const void *yv, const double *mu,
                                  const double *var, double *out1, double *out2, double *out3) {
    const int L = lik->nlatent;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC:
        case LIK_NEGBINOMIAL:
            out1[i] = sqrt(second_moment(mu[i], var[i]));
            break;
        case LIK_STUDENTT: {
            const double *y = (const double *)yv;
            double nu = lik->p[0], sg = lik->p[1];
            out1[i] = (nu / (sg * sg) + second_moment_y(mu[i], var[i], y[i])) / 2.0;
        } break;
        case LIK_CATEGORICAL:
        case LIK_CATEGORICAL_BIJ: {
            double den = lik->kind == LIK_CATEGORICAL ? (double)L : cat_get_const(lik) + (double)L;
            for (int k = 0; k < L; ++k) {
                double c = sqrt(second_moment(mu[i * L + k], var[i * L + k]));
                out1[i * L + k] = c;
                out2[i * L + k] = agplo_approx_expected_logistic(-mu[i * L + k], c) / den;
            }
        } break;
        case LIK_POISSON: {
            double c = sqrt(second_moment(mu[i], var[i]));
            out1[i] = c;
            out2[i] = lik->p[0] * agplo_approx_expected_logistic(-mu[i], c);
        } break;
        case LIK_LAPLACE: {
            const double *y = (const double *)yv;
            out1[i] = 1.0 / (2.0 * lik->p[0] * sqrt(second_moment_y(mu[i], var[i], y[i])));
        } break;
        case LIK_HETEROGAUSS: {
            const double *y = (const double *)yv;
            double psi = second_moment_y(mu[2 * i], var[2 * i], y[i]) / 2.0;
            double c = sqrt(second_moment(mu[2 * i + 1], var[2 * i + 1]));
            out3[i] = psi;
            out1[i] = c;
            out2[i] = lik->p[0] * agplo_approx_expected_logistic(-mu[2 * i + 1], c) * psi;
        } break;
        default:
            break;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* expected_auglik_potential / expected_auglik_precision                                       */
/*   bernoulli.jl:35-45, negativebinomial.jl:43-49, studentt.jl:68-74, categorical.jl:121-136,  */
/*   poisson.jl:49-60, laplace.jl:62-68, heteroscedasticgaussian.jl:68-104                      */
/* q1,q2 = the aux_posterior outputs.  beta_out/gamma_out: L contiguous vectors of N            */
/* (the transposed layout of utils.jl:24).  mu_g: mean of q(g) (heterogauss only).              */
/* ------------------------------------------------------------------------------------------ */
AGPLO_API int agplo_expected_potential_precision(const agplo_lik *lik, int64_t n, const void *yv,
                                                 const double *q1, const double *q2,
                                                 const double *mu_g, double *beta, double *gamma) {
    const int L = lik->nlatent;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC: {
            const uint8_t *y = (const uint8_t *)yv;
            double s = (double)y[i] - 0.5;
            beta[i] = (s > 0 ? 1.0 : (s < 0 ? -1.0 : 0.0)) / 2.0;
            gamma[i] = agplo_pg_mean(1.0, q1[i]);
        } break;
        case LIK_NEGBINOMIAL: {
            const int32_t *y = (const int32_t *)yv;
            beta[i] = ((double)y[i] - lik->p[0]) / 2.0;
            gamma[i] = agplo_pg_mean((double)y[i] + lik->p[0], q1[i]);
        } break;
        case LIK_STUDENTT: { /* mean(Gamma(alpha, 1/beta_i)) = alpha/beta_i ; studentt.jl:41-43 */
            const double *y = (const double *)yv;
            double w = ((lik->p[0] + 1.0) / 2.0) * (1.0 / q1[i]);
            gamma[i] = w;
            beta[i] = w * y[i];
        } break;
        case LIK_CATEGORICAL:
        case LIK_CATEGORICAL_BIJ: { /* tvmean polyagammanegativemultinomial.jl:41-49 ; NM mean
                                       negativemultinomial.jl:54 */
            const uint8_t *y = (const uint8_t *)yv;
            double sp = 0.0;
            for (int k = 0; k < L; ++k) sp += q2[i * L + k];
            double p0 = 1.0 - sp;
            for (int k = 0; k < L; ++k) {
                double nbar = 1.0 / p0 * q2[i * L + k];
                double yk = (double)y[i * L + k];
                beta[(int64_t)k * n + i] = (yk - nbar) / 2.0;
                gamma[(int64_t)k * n + i] = agplo_pg_mean(yk + nbar, q1[i * L + k]);
            }
        } break;
        case LIK_POISSON: { /* tvmean polyagammapoisson.jl:35-41 */
            const int32_t *y = (const int32_t *)yv;
            double nbar = q2[i];
            beta[i] = ((double)y[i] - nbar) / 2.0;
            gamma[i] = agplo_pg_mean((double)y[i] + nbar, q1[i]);
        } break;
        case LIK_LAPLACE: { /* mean(InverseGaussian(mu,lambda)) = mu */
            const double *y = (const double *)yv;
            gamma[i] = 2.0 * q1[i];
            beta[i] = 2.0 * q1[i] * y[i];
        } break;
        case LIK_HETEROGAUSS: { /* :94-104 */
            const double *y = (const double *)yv;
            double lsg = lik->p[0] * (1.0 - agplo_approx_expected_logistic(-mu_g[i], q1[i]));
            double nbar = q2[i];
            beta[i] = y[i] * lsg / 2.0;
            gamma[i] = lsg;
            beta[n + i] = (0.5 - nbar) / 2.0;
            gamma[n + i] = agplo_pg_mean(0.5 + nbar, q1[i]);
        } break;
        default:
            break;
        }
    }
    return 0;
}

/* auglik_potential / auglik_precision (sampled twins)
 *   bernoulli.jl:27-33, negativebinomial.jl:35-41, studentt.jl:60-66, categorical.jl:112-119,
 *   poisson.jl:41-47, laplace.jl:54-60, heteroscedasticgaussian.jl:48-66 (g = latent 2 of fg) */
AGPLO_API int agplo_potential_precision(const agplo_lik *lik, int64_t n, const void *yv,
                                        const double *omega, const int64_t *nn, const double *fg,
                                        double *beta, double *gamma) {
    const int L = lik->nlatent;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC: {
            const uint8_t *y = (const uint8_t *)yv;
            double s = (double)y[i] - 0.5;
            beta[i] = (s > 0 ? 1.0 : (s < 0 ? -1.0 : 0.0)) / 2.0;
            gamma[i] = omega[i];
        } break;
        case LIK_NEGBINOMIAL: {
            const int32_t *y = (const int32_t *)yv;
            beta[i] = ((double)y[i] - lik->p[0]) / 2.0;
            gamma[i] = omega[i];
        } break;
        case LIK_STUDENTT: {
            const double *y = (const double *)yv;
            beta[i] = y[i] * omega[i];
            gamma[i] = omega[i];
        } break;
        case LIK_CATEGORICAL:
        case LIK_CATEGORICAL_BIJ: {
            const uint8_t *y = (const uint8_t *)yv;
            for (int k = 0; k < L; ++k) {
                beta[(int64_t)k * n + i] = ((double)y[i * L + k] - (double)nn[i * L + k]) / 2.0;
                gamma[(int64_t)k * n + i] = omega[i * L + k];
            }
        } break;
        case LIK_POISSON: {
            const int32_t *y = (const int32_t *)yv;
            beta[i] = ((double)y[i] - (double)nn[i]) / 2.0;
            gamma[i] = omega[i];
        } break;
        case LIK_LAPLACE: {
            const double *y = (const double *)yv;
            beta[i] = 2.0 * omega[i] * y[i];
            gamma[i] = 2.0 * omega[i];
        } break;
        case LIK_HETEROGAUSS: {
            const double *y = (const double *)yv;
            double il = lik->p[0] * logistic_(fg[2 * i + 1]); /* inv(invlink(g)) = lambda*sigma(g) */
            beta[i] = y[i] * il;
            gamma[i] = il;
            beta[n + i] = (0.5 - (double)nn[i]) / 2.0;
            gamma[n + i] = omega[i];
        } break;
        default:
            break;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* logtilt (sum over points)  generic.jl:40-46 with the per-likelihood scalar forms             */
/*   bernoulli.jl:47-49, negativebinomial.jl:51-57, studentt.jl:76-78, categorical.jl:138-145,  */
/*   poisson.jl:62-65, laplace.jl:70-81                                                        */
/* ------------------------------------------------------------------------------------------ */
static double negbin_logconst(double y, double r) { /* negativebinomial.jl:51-52 */
    return lgamma(y + r) - lgamma(y + 1.0) - lgamma(r);
}
AGPLO_API double agplo_logtilt(const agplo_lik *lik, int64_t n, const void *yv, const double *omega,
                               const int64_t *nn, const double *f) {
    const int L = lik->nlatent;
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC: {
            const uint8_t *y = (const uint8_t *)yv;
            double s = y[i] ? 1.0 : -1.0;
            acc += -LOGTWO + (s * f[i] - f[i] * f[i] * omega[i]) / 2.0;
        } break;
        case LIK_NEGBINOMIAL: {
            const int32_t *y = (const int32_t *)yv;
            double r = lik->p[0], yy = (double)y[i];
            acc += negbin_logconst(yy, r) - (yy + r) * LOGTWO +
                   (f[i] * (yy - r) - f[i] * f[i] * omega[i]) / 2.0;
        } break;
        case LIK_STUDENTT: { /* logpdf(Normal(f, sqrt(1/omega)), y) */
            const double *y = (const double *)yv;
            double d = y[i] - f[i];
            acc += -0.5 * LOG2PI + 0.5 * log(omega[i]) - 0.5 * d * d * omega[i];
        } break;
        case LIK_CATEGORICAL:
        case LIK_CATEGORICAL_BIJ: {
            const uint8_t *y = (const uint8_t *)yv;
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < L; ++k) {
                double yk = (double)y[i * L + k], nk = (double)nn[i * L + k], fk = f[i * L + k];
                s1 += yk + nk;
                s2 += (yk - nk) * fk - fk * fk * omega[i * L + k];
            }
            acc += -s1 * LOGTWO + s2 / 2.0;
        } break;
        case LIK_POISSON: {
            const int32_t *y = (const int32_t *)yv;
            double yy = (double)y[i], nk = (double)nn[i];
            acc += yy * log(lik->p[0]) - (yy + nk) * LOGTWO - lgamma(yy + 1.0) +
                   ((yy - nk) * f[i] - f[i] * f[i] * omega[i]) / 2.0;
        } break;
        case LIK_LAPLACE: { /* loggamma(1/2) - log(sqrt(pi)) = 0 */
            const double *y = (const double *)yv;
            double d = y[i] - f[i];
            acc += lgamma(0.5) - 0.5 * log(PI_) - log(2.0 * lik->p[0]) - d * d * omega[i];
        } break;
        default:
            return NAN;
        }
    }
    return acc;
}

/* expected_logtilt: api.jl:219-223 + bernoulli.jl:59-65, negativebinomial.jl:59-65,
 * studentt.jl:80-83, categorical.jl:172-180, poisson.jl:76-85, laplace.jl:83-88.
 * q1,q2 = aux_posterior outputs. */
AGPLO_API double agplo_expected_logtilt(const agplo_lik *lik, int64_t n, const void *yv,
                                        const double *q1, const double *q2, const double *mu,
                                        const double *var) {
    const int L = lik->nlatent;
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC: {
            const uint8_t *y = (const uint8_t *)yv;
            double s = y[i] ? 1.0 : -1.0;
            double th = agplo_pg_mean(1.0, q1[i]);
            acc += -LOGTWO + (s * mu[i] - (mu[i] * mu[i] + var[i]) * th) / 2.0;
        } break;
        case LIK_NEGBINOMIAL: {
            const int32_t *y = (const int32_t *)yv;
            double r = lik->p[0], yy = (double)y[i];
            double th = agplo_pg_mean(yy + r, q1[i]);
            acc += negbin_logconst(yy, r) - (yy + r) * LOGTWO +
                   (mu[i] * (yy - r) - second_moment(mu[i], var[i]) * th) / 2.0;
        } break;
        case LIK_STUDENTT: {
            const double *y = (const double *)yv;
            double th = ((lik->p[0] + 1.0) / 2.0) / q1[i];
            double d = mu[i] - y[i];
            acc += -0.5 * LOG2PI + 0.5 * log(th) - 0.5 * d * d * th - var[i] * th / 2.0;
        } break;
        case LIK_CATEGORICAL:
        case LIK_CATEGORICAL_BIJ: {
            const uint8_t *y = (const uint8_t *)yv;
            double sp = 0.0;
            for (int k = 0; k < L; ++k) sp += q2[i * L + k];
            double p0 = 1.0 - sp, s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < L; ++k) {
                double yk = (double)y[i * L + k], nbar = q2[i * L + k] / p0;
                double w = agplo_pg_mean(yk + nbar, q1[i * L + k]);
                double m = mu[i * L + k], v = var[i * L + k];
                s1 += yk + nbar;
                s2 += ((yk - nbar) * m - (m * m + v) * w) / 2.0;
            }
            acc += -s1 * LOGTWO + s2;
        } break;
        case LIK_POISSON: {
            const int32_t *y = (const int32_t *)yv;
            double yy = (double)y[i], nbar = q2[i];
            double w = agplo_pg_mean(yy + nbar, q1[i]);
            acc += -(yy + nbar) * LOGTWO + ((yy - nbar) * mu[i] - (mu[i] * mu[i] + var[i]) * w) / 2.0 +
                   yy * log(lik->p[0]) - lgamma(yy + 1.0);
        } break;
        case LIK_LAPLACE: {
            const double *y = (const double *)yv;
            acc += lgamma(0.5) - 0.5 * log(PI_) - log(2.0 * lik->p[0]) -
                   second_moment_y(mu[i], var[i], y[i]) * q1[i];
        } break;
        default:
            return NAN;
        }
    }
    return acc;
}

/* aux_kldivergence generic.jl:56-62 with priors bernoulli.jl:51-57 (PG(1,0)),
 * negativebinomial.jl:67-73 (PG(y+r,0)), studentt.jl:85-91 (Gamma(nu/2, scale sigma^2/(nu/2))),
 * poisson.jl:67-74 + polyagammapoisson.jl:47-51, laplace.jl:96-104 (custom closed form),
 * categorical bijective polyagammanegativemultinomial.jl:56-65 + negativemultinomial.jl:72-82.
 * Non-bijective categorical: error() categorical.jl:165-170 -> NAN. */
static double digamma_(double x) {
    double r = 0.0;
    while (x < 6.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x -
           f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f / 132.0))));
}
AGPLO_API double agplo_aux_kl(const agplo_lik *lik, int64_t n, const void *yv, const double *q1,
                              const double *q2) {
    const int L = lik->nlatent;
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC:
            acc += agplo_pg_kl(1.0, q1[i]);
            break;
        case LIK_NEGBINOMIAL: {
            const int32_t *y = (const int32_t *)yv;
            acc += agplo_pg_kl((double)y[i] + lik->p[0], q1[i]);
        } break;
        case LIK_STUDENTT: { /* upstream, unpinned: Distributions kldivergence(Gamma,Gamma) */
            double nu = lik->p[0], sg = lik->p[1];
            double ap = (nu + 1.0) / 2.0, thp = 1.0 / q1[i]; /* q: shape ap, scale thp */
            double aq = nu / 2.0, thq = sg * sg / (nu / 2.0);  /* prior */
            acc += (ap - aq) * digamma_(ap) - lgamma(ap) + lgamma(aq) + aq * (log(thq) - log(thp)) +
                   ap * (thp - thq) / thq;
        } break;
        case LIK_POISSON: { /* KL(Po(q.lambda) || Po(lambda)) upstream closed form */
            const int32_t *y = (const int32_t *)yv;
            double lq = q2[i], lp = lik->p[0];
            double klp = lq > 0 ? lq * (log(lq) - log(lp)) - lq + lp : lp;
            acc += agplo_pg_kl((double)y[i] + lq, q1[i]) + klp;
        } break;
        case LIK_LAPLACE: { /* laplace.jl:96-104 with lambda = scale(prior) = (2 beta)^-2 */
            double lam = 1.0 / ((2.0 * lik->p[0]) * (2.0 * lik->p[0]));
            acc += log(2.0 * lam) / 2.0 - log(2.0 * PI_) / 2.0 - log(lam) / 2.0 + lgamma(0.5) +
                   lam / q1[i];
        } break;
        case LIK_CATEGORICAL_BIJ: {
            const uint8_t *y = (const uint8_t *)yv;
            double sp = 0.0;
            for (int k = 0; k < L; ++k) sp += q2[i * L + k];
            double p0 = 1.0 - sp;
            double pp = 1.0 / cat_sum_theta(lik); /* prior p_k, categorical.jl:153-157 */
            double p0p = 1.0 - L * pp;
            double s = 0.0;
            for (int k = 0; k < L; ++k) {
                double nbar = q2[i * L + k] / p0;
                acc += agplo_pg_kl((double)y[i * L + k] + nbar, q1[i * L + k]);
                s += q2[i * L + k] * (log(q2[i * L + k]) - log(pp));
            }
            acc += log(p0) - log(p0p) + s / p0;
        } break;
        default:
            return NAN;
        }
    }
    return acc;
}

/* logpdf of the upstream scalar families the priors / conditionals are built from (Distributions.jl 0.25 closed forms,
 * "upstream, unpinned"): Poisson, Gamma(shape, scale), InverseGamma(shape, scale), InverseGaussian(mu, lambda). */
static double poisson_logpdf(double lam, double n) {
    if (lam == 0.0) return n == 0.0 ? 0.0 : -INFINITY;
    return n * log(lam) - lam - lgamma(n + 1.0);
}
static double gamma_logpdf(double a, double th, double x) {
    return -lgamma(a) - a * log(th) + (a - 1.0) * log(x) - x / th;
}
static double invgamma_logpdf(double a, double th, double x) {
    return a * log(th) - lgamma(a) - (a + 1.0) * log(x) - th / x;
}
static double invgaussian_logpdf(double mu, double lam, double x) {
    return (log(lam) - (LOG2PI + 3.0 * log(x)) - lam * (x - mu) * (x - mu) / (mu * mu * x)) / 2.0;
}

/* logdensity_def(aux_prior(lik, y), Omega), summed (second half of aug_loglik generic.jl:48-50):
 *   bernoulli.jl:51-57 PG(1,0); negativebinomial.jl:67-73 PG(y+r,0); studentt.jl:85-91 Gamma(nu/2, scale sigma^2/(nu/2));
 *   poisson.jl:67-76 PolyaGammaPoisson(y,0,lambda) with the joint density of polyagammapoisson.jl:29-33
 *   (logpdf(Poisson(lambda), n) + logpdf(PG(y+n,0), omega)); laplace.jl:90-96 InverseGamma(1/2, (2 beta)^-2).
 * Categorical: the reference's logdensity_def of PolyaGammaNegativeMultinomial is broken -> NAN.
 * Heteroscedastic: there is no aux_prior; aug_loglik is its own method (agplo_aug_loglik). */
AGPLO_API double agplo_aux_prior_logpdf(const agplo_lik *lik, int64_t n, const void *yv,
                                        const double *omega, const int64_t *nn) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC:
            acc += agplo_pg_logpdf(1.0, 0.0, omega[i]);
            break;
        case LIK_NEGBINOMIAL: {
            const int32_t *y = (const int32_t *)yv;
            acc += agplo_pg_logpdf((double)y[i] + lik->p[0], 0.0, omega[i]);
        } break;
        case LIK_STUDENTT: { /* logpdf(Gamma(nu/2, scale 2 sigma^2/nu), omega) studentt.jl:91 */
            double a = lik->p[0] / 2.0, th = lik->p[1] * lik->p[1] / a;
            acc += gamma_logpdf(a, th, omega[i]);
        } break;
        case LIK_POISSON: { /* polyagammapoisson.jl:29-33 at (y, c = 0, lambda) */
            const int32_t *y = (const int32_t *)yv;
            if (!nn) return NAN;
            acc += poisson_logpdf(lik->p[0], (double)nn[i]) +
                   agplo_pg_logpdf((double)y[i] + (double)nn[i], 0.0, omega[i]);
        } break;
        case LIK_LAPLACE: { /* laplace.jl:96 */
            double lam = 1.0 / ((2.0 * lik->p[0]) * (2.0 * lik->p[0]));
            acc += invgamma_logpdf(0.5, lam, omega[i]);
        } break;
        default:
            return NAN;
        }
    }
    return acc;
}

/* aug_loglik(lik, Omega, y, f): generic.jl:48-50 (logtilt + logdensity_def(aux_prior)); the heteroscedastic likelihood
 * has its own method, heteroscedasticgaussian.jl:106-128, with f = fg [2,N] column-major (f_i = fg[2i], g_i = fg[2i+1]). */
AGPLO_API double agplo_aug_loglik(const agplo_lik *lik, int64_t n, const void *yv, const double *omega,
                                  const int64_t *nn, const double *f) {
    if (lik->kind == LIK_HETEROGAUSS) {
        const double *y = (const double *)yv;
        double acc = 0.0;
        if (!nn) return NAN;
        for (int64_t i = 0; i < n; ++i) {
            double ff = f[2 * i], gg = f[2 * i + 1], nk = (double)nn[i];
            acc += -(0.5 + nk) * LOGTWO + ((0.5 - nk) * gg - gg * gg * omega[i]) / 2.0 +
                   agplo_pg_logpdf(0.5 + nk, 0.0, omega[i]) +
                   poisson_logpdf(lik->p[0] / 2.0 * (y[i] - ff) * (y[i] - ff), nk);
        }
        return acc;
    }
    return agplo_logtilt(lik, n, yv, omega, nn, f) + agplo_aux_prior_logpdf(lik, n, yv, omega, nn);
}

/* logdensity_def(aux_full_conditional(lik, y, f), Omega), summed -- the second term of the full-conditional-Omega identity
 * of src/TestUtils.jl:107-116.  Conditionals: bernoulli.jl:13-15, negativebinomial.jl:20-22, studentt.jl:46-48,
 * poisson.jl:26-28, laplace.jl:40-42, heteroscedasticgaussian.jl:28-32 (joint densities polyagammapoisson.jl:29-33;
 * NTDist -> logpdf of the wrapped distribution, ntdist.jl). */
AGPLO_API double agplo_full_conditional_logpdf(const agplo_lik *lik, int64_t n, const void *yv, const double *f,
                                               const double *omega, const int64_t *nn) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC:
            acc += agplo_pg_logpdf(1.0, fabs(f[i]), omega[i]);
            break;
        case LIK_NEGBINOMIAL:
            acc += agplo_pg_logpdf((double)((const int32_t *)yv)[i] + lik->p[0], fabs(f[i]), omega[i]);
            break;
        case LIK_STUDENTT: {
            double nu = lik->p[0], sg = lik->p[1], d = ((const double *)yv)[i] - f[i];
            acc += gamma_logpdf((nu + 1.0) / 2.0, 2.0 / (nu / (sg * sg) + d * d), omega[i]);
        } break;
        case LIK_POISSON: {
            if (!nn) return NAN;
            double yy = (double)((const int32_t *)yv)[i], nk = (double)nn[i];
            acc += poisson_logpdf(lik->p[0] * logistic_(-f[i]), nk) + agplo_pg_logpdf(yy + nk, fabs(f[i]), omega[i]);
        } break;
        case LIK_LAPLACE: {
            double beta = lik->p[0], lam = 1.0 / ((2.0 * beta) * (2.0 * beta));
            acc += invgaussian_logpdf(1.0 / (2.0 * beta * fabs(((const double *)yv)[i] - f[i])), 2.0 * lam, omega[i]);
        } break;
        case LIK_HETEROGAUSS: {
            if (!nn) return NAN;
            double ff = f[2 * i], gg = f[2 * i + 1], yy = ((const double *)yv)[i], nk = (double)nn[i];
            acc += poisson_logpdf(lik->p[0] * logistic_(-gg) * (ff - yy) * (ff - yy) / 2.0, nk) +
                   agplo_pg_logpdf(0.5 + nk, fabs(gg), omega[i]);
        } break;
        default:
            return NAN;
        }
    }
    return acc;
}

/* expected_aug_loglik(lik, qOmega, y, qf): generic.jl:52-54 = expected_logtilt + aux_kldivergence (the PLUS sign is the
 * reference's; its ELBO examples subtract the KL themselves, examples/bernoulli/script.jl:65-70); the heteroscedastic
 * likelihood has its own method, heteroscedasticgaussian.jl:130-145: q1 = c, q2 = lambda of aux_posterior!, (mu, var) =
 * q(f), q(g) as [2,N].  `var(first(qg))` there is taken as var(qg) (first() of a scalar Normal does not exist upstream:
 * SURVEY App. B). */
AGPLO_API double agplo_expected_aug_loglik(const agplo_lik *lik, int64_t n, const void *yv, const double *q1,
                                           const double *q2, const double *mu, const double *var) {
    if (lik->kind == LIK_HETEROGAUSS) {
        const double *y = (const double *)yv;
        const double lam = lik->p[0];
        const double Cst = 0.5 * (log(lam) + log(2.0 / PI_));
        double acc = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double mf = mu[2 * i], vf = var[2 * i], g = mu[2 * i + 1], vg = var[2 * i + 1];
            double tn = q2[i], tw = agplo_pg_mean(0.5 + tn, q1[i]);
            double lp = lam / 2.0 * ((y[i] - mf) * (y[i] - mf) + vf);
            double klp = tn > 0 ? tn * (log(tn) - log(lp)) - tn + lp : lp;
            acc += Cst - (0.5 + tn) * LOGTWO + ((0.5 - tn) * g - (g * g + vg) * tw) / 2.0 +
                   agplo_pg_kl(0.5 + tn, q1[i]) + klp;
        }
        return acc;
    }
    return agplo_expected_logtilt(lik, n, yv, q1, q2, mu, var) + agplo_aux_kl(lik, n, yv, q1, q2);
}

/* ------------------------------------------------------------------------------------------ */
/* Sparse CAVI pass over N points and M features (docs/src/index.md:154-163 restated in the     */
/* feature basis Phi = columns phi_i in R^M):                                    */
/*   marginals   mu_i = mu0_i + phi_i' alpha ,  var_i = kdiag_i - phi_i' W phi_i   (a11)         */
/*   aux_posterior! + expected_auglik_{potential,precision}                        (a9, a10)    */
/*   G_l = sum_i gamma_il phi_i phi_i' ,  g_l = sum_i beta_il phi_i                (a12)         */
/* Phi: [M,N] column-major (M contiguous per point), float32 (bit-identical to what the device  */
/* reads) ; W: [M,M,L] symmetric f64 ; alpha: [M,L] ; G: [M,M,L] f64 ; g: [M,L].                */
/* y / outputs as in the operator functions.  gamma_out/beta_out [L][N] optional.               */
/* ------------------------------------------------------------------------------------------ */
AGPLO_API int agplo_cavi_pass(const agplo_lik *lik, int64_t N, int M, const float *Phi,
                              const double *kdiag, const double *mu0, const void *yv,
                              const double *W, const double *alpha, double *G, double *g,
                              double *mu_out, double *var_out, double *beta_out, double *gamma_out) {
    const int L = lik->nlatent;
    const int64_t MM = (int64_t)M * M;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    double *Gt = (double *)calloc((size_t)nthreads * L * (MM + M), sizeof(double));
    if (!Gt) return -1;
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *Gl = Gt + (size_t)tid * L * (MM + M);
        double *gl = Gl + (size_t)L * MM;
        double *phi = (double *)malloc(sizeof(double) * M);
        double *mu = (double *)malloc(sizeof(double) * L * 2);
        double *var = mu + L;
        double *q1 = (double *)malloc(sizeof(double) * L * 5);
        double *q2 = q1 + L, *bt = q1 + 2 * L, *gm = q1 + 3 * L, *q3 = q1 + 4 * L;
#pragma omp for schedule(static)
        for (int64_t i = 0; i < N; ++i) {
            const float *pf = Phi + i * (int64_t)M;
            for (int a = 0; a < M; ++a) phi[a] = (double)pf[a];
            for (int l = 0; l < L; ++l) {
                const double *Wl = W + (int64_t)l * MM;
                const double *al = alpha + (int64_t)l * M;
                double m = mu0 ? mu0[i * L + l] : 0.0, q = 0.0;
                for (int a = 0; a < M; ++a) {
                    const double *Wr = Wl + (int64_t)a * M;
                    double t = 0.0;
                    for (int b = 0; b < M; ++b) t += Wr[b] * phi[b];
                    q += phi[a] * t;
                    m += al[a] * phi[a];
                }
                mu[l] = m;
                var[l] = kdiag[i] - q;
            }
            /* per-point operator calls on a length-1 slice (same code path as the vector API) */
            const void *yi;
            switch (lik->kind) {
            case LIK_BERNOULLI_LOGISTIC: yi = (const uint8_t *)yv + i; break;
            case LIK_NEGBINOMIAL:
            case LIK_POISSON: yi = (const int32_t *)yv + i; break;
            case LIK_CATEGORICAL:
            case LIK_CATEGORICAL_BIJ: yi = (const uint8_t *)yv + i * L; break;
            default: yi = (const double *)yv + i; break;
            }
            agplo_aux_posterior(lik, 1, yi, mu, var, q1, q2, q3);
            agplo_expected_potential_precision(lik, 1, yi, q1, q2, L > 1 ? mu + 1 : NULL, bt, gm);
            for (int l = 0; l < L; ++l) {
                if (mu_out) mu_out[i * L + l] = mu[l];
                if (var_out) var_out[i * L + l] = var[l];
                if (beta_out) beta_out[(int64_t)l * N + i] = bt[l];
                if (gamma_out) gamma_out[(int64_t)l * N + i] = gm[l];
                double *Gll = Gl + (int64_t)l * MM;
                double *gll = gl + (int64_t)l * M;
                for (int a = 0; a < M; ++a) {
                    double ga = gm[l] * phi[a];
                    double *Gr = Gll + (int64_t)a * M;
                    for (int b = 0; b <= a; ++b) Gr[b] += ga * phi[b];
                    gll[a] += bt[l] * phi[a];
                }
            }
        }
        free(phi);
        free(mu);
        free(q1);
    }
    for (int64_t k = 0; k < (int64_t)L * MM; ++k) G[k] = 0.0;
    for (int64_t k = 0; k < (int64_t)L * M; ++k) g[k] = 0.0;
    for (int t = 0; t < nthreads; ++t) {
        const double *Gl = Gt + (size_t)t * L * (MM + M);
        const double *gl = Gl + (size_t)L * MM;
        for (int64_t k = 0; k < (int64_t)L * MM; ++k) G[k] += Gl[k];
        for (int64_t k = 0; k < (int64_t)L * M; ++k) g[k] += gl[k];
    }
    for (int l = 0; l < L; ++l) /* mirror the lower triangle */
        for (int a = 0; a < M; ++a)
            for (int b = a + 1; b < M; ++b)
                G[(int64_t)l * MM + (int64_t)a * M + b] = G[(int64_t)l * MM + (int64_t)b * M + a];
    free(Gt);
    return 0;
}

/* Weighted feature sums only (the a12 accumulation given beta/gamma, used by the Gibbs path):
 * G = Phi diag(gamma) Phi', g = Phi beta.  gamma/beta: [L][N]. */
AGPLO_API int agplo_accumulate(int64_t N, int M, int L, const float *Phi, const double *beta,
                               const double *gamma, double *G, double *g) {
    const int64_t MM = (int64_t)M * M;
    for (int64_t k = 0; k < (int64_t)L * MM; ++k) G[k] = 0.0;
    for (int64_t k = 0; k < (int64_t)L * M; ++k) g[k] = 0.0;
    for (int l = 0; l < L; ++l) {
        double *Gl = G + (int64_t)l * MM, *gl = g + (int64_t)l * M;
        for (int64_t i = 0; i < N; ++i) {
            const float *pf = Phi + i * (int64_t)M;
            double gm = gamma[(int64_t)l * N + i], bt = beta[(int64_t)l * N + i];
            for (int a = 0; a < M; ++a) {
                double ga = gm * (double)pf[a];
                for (int b = 0; b <= a; ++b) Gl[(int64_t)a * M + b] += ga * (double)pf[b];
                gl[a] += bt * (double)pf[a];
            }
        }
        for (int a = 0; a < M; ++a)
            for (int b = a + 1; b < M; ++b) Gl[(int64_t)a * M + b] = Gl[(int64_t)b * M + a];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Synthetic workload generator: every value is a pure function of (seed, index) */
/* through Philox, so host and device produce identical inputs without any transfer.            */
/*   x_i = -10 + 20 u(stream i, draw 0)           (domain of examples/bernoulli/script.jl:14)    */
/*   f*(x) = 2 sin(0.7 x) + cos(0.23 x)                                                         */
/*   bernoulli y_i = [u(stream i, draw 1) < logistic(f*)]                                        */
/* ------------------------------------------------------------------------------------------ */
AGPLO_API double agplo_synth_fstar(double x) { return 2.0 * sin(0.7 * x) + cos(0.23 * x); }

AGPLO_API void agplo_synth_x(uint64_t seed, int64_t i0, int64_t n, double *x) {
    for (int64_t i = 0; i < n; ++i) {
        agplo_rng g;
        rng_init(&g, seed, (uint64_t)(i0 + i), 0xD47Au);
        x[i] = -10.0 + 20.0 * rng_u01(&g);
    }
}
/* y for the PG families: kind = bernoulli (u8) or negbin (i32, Gamma-Poisson mixture, r = p[0]) */
AGPLO_API void agplo_synth_y(const agplo_lik *lik, uint64_t seed, int64_t i0, int64_t n, void *yv) {
    for (int64_t i = 0; i < n; ++i) {
        agplo_rng g;
        rng_init(&g, seed, (uint64_t)(i0 + i), 0xD47Au);
        double x = -10.0 + 20.0 * rng_u01(&g);
        double fs = agplo_synth_fstar(x);
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC:
            ((uint8_t *)yv)[i] = rng_u01(&g) < logistic_(fs) ? 1 : 0;
            break;
        case LIK_NEGBINOMIAL: { /* y ~ NB(r, p = sigma(f)): lambda ~ Gamma(r, p/(1-p)), y ~ Po */
            double p = logistic_(0.5 * fs);
            double lam = rand_gamma(&g, lik->p[0]) * p / (1.0 - p);
            ((int32_t *)yv)[i] = (int32_t)rand_poisson(&g, lam);
        } break;
        case LIK_STUDENTT: { /* y = f* + sigma * t_nu,  t = z / sqrt(chi2_nu / nu) */
            double z = rng_normal(&g);
            double ch = 2.0 * rand_gamma(&g, lik->p[0] / 2.0);
            ((double *)yv)[i] = fs + lik->p[1] * z / sqrt(ch / lik->p[0]);
        } break;
        case LIK_CATEGORICAL:
        case LIK_CATEGORICAL_BIJ: { /* class weights theta_k logistic(f*(x + 2k)); one-hot u8 [L,N] */
            const int L = lik->nlatent;
            double tot = lik->kind == LIK_CATEGORICAL_BIJ ? cat_get_const(lik) : 0.0;
            for (int k = 0; k < L; ++k)
                tot += exp(lik->logtheta[k]) * logistic_(agplo_synth_fstar(x + 2.0 * k));
            double u = rng_u01(&g) * tot, cum = 0.0;
            int cls = L;
            for (int k = 0; k < L; ++k) {
                cum += exp(lik->logtheta[k]) * logistic_(agplo_synth_fstar(x + 2.0 * k));
                if (u < cum) {
                    cls = k;
                    break;
                }
            }
            if (lik->kind == LIK_CATEGORICAL && cls == L) cls = L - 1;
            for (int k = 0; k < L; ++k) ((uint8_t *)yv)[i * L + k] = (k == cls) ? 1 : 0;
        } break;
        default:
            break;
        }
    }
}

/* squared-exponential features: K_ZX[a,i] = exp(-(x_i - z_a)^2 / (2 ell^2)), float32 output,
 * [M,N] column-major.  (with_lengthscale(SqExponentialKernel(), ell), examples/bernoulli/script.jl:15) */
AGPLO_API void agplo_se_kernel_f32(int64_t n, int M, const double *x, const double *z, double ell,
                                   float *out) {
    for (int64_t i = 0; i < n; ++i)
        for (int a = 0; a < M; ++a) {
            double d = (x[i] - z[a]) / ell;
            out[i * (int64_t)M + a] = (float)exp(-0.5 * d * d);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* Gibbs half of the sparse sweep (examples/bernoulli/script.jl:81-84 in sparse form):          */
/*   f_il = mu0_il + phi_i' v_l + sqrt(kdiag_i) eps_il ; Omega_i ~ aux_full_conditional ;        */
/*   beta, gamma = auglik_potential / auglik_precision (rounded to float32 as the device feeds    */
/*   them to the accumulation).  The float64 summation order of phi_i' v_l is the device's:       */
/*   16 partial sums (partial q takes features 4q.., 4q+64.., each float4 in order) combined by   */
/*   an xor butterfly 8,4,2,1.                                                                   */
/* ------------------------------------------------------------------------------------------ */
AGPLO_API void agplo_randn_many(uint64_t seed, uint64_t stream0, uint32_t sweep, int64_t n, double *out) {
    for (int64_t i = 0; i < n; ++i) {
        agplo_rng g;
        rng_init(&g, seed, stream0 + (uint64_t)i, sweep);
        out[i] = rng_normal(&g);
    }
}

static double project_device_order(const float *row, const double *vl, int M) {
    /* 16 partial sums: partial q takes features 4q.., 4q+64.., each float4 in order; xor butterfly 8,4,2,1 */
    double part[16], tmp[16];
    for (int q = 0; q < 16; ++q) {
        double acc = 0.0;
        for (int a = q << 2; a < M; a += 64) {
            acc += (double)row[a] * vl[a];
            acc += (double)row[a + 1] * vl[a + 1];
            acc += (double)row[a + 2] * vl[a + 2];
            acc += (double)row[a + 3] * vl[a + 3];
        }
        part[q] = acc;
    }
    for (int off = 8; off > 0; off >>= 1) {
        for (int q = 0; q < 16; ++q) tmp[q] = part[q] + part[q ^ off];
        memcpy(part, tmp, sizeof(part));
    }
    return part[0];
}

AGPLO_API int agplo_gibbs_points(const agplo_lik *lik, int64_t N, int M, const float *Phi, const double *kdiag,
                                 const double *mu0, const void *yv, const double *v, uint64_t seed,
                                 uint32_t sweep, double *f_out, double *omega, int64_t *nout,
                                 uint32_t *nuni_out, float *beta32, float *gamma32) {
    const int Lf = lik->nlatent;
    const int Lo = lik->kind == LIK_HETEROGAUSS ? 1 : lik->nlatent;
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < N; ++i) {
        agplo_rng g;
        rng_init(&g, seed, (uint64_t)(g_point_offset + i), sweep);
        double sd = sqrt(kdiag[i] > 0.0 ? kdiag[i] : 0.0); /* a float32 Nystrom residual can round below zero */
        double *fi = f_out + i * Lf;
        for (int l = 0; l < Lf; ++l) {
            double f = project_device_order(Phi + i * (int64_t)M, v + (int64_t)l * M, M) + sd * rng_normal(&g);
            if (mu0) f += mu0[(int64_t)l * N + i];
            fi[l] = f;
        }
        /* aux_sample! on the same stream: single-point call of the vector API would restart the stream, so the
         * per-kind draw is restated through a 1-point slice with the rng carried over */
        uint32_t nt = 0;
        double *om = omega + i * Lo;
        int64_t *nn = nout ? nout + i * Lo : NULL;
        switch (lik->kind) {
        case LIK_BERNOULLI_LOGISTIC:
            om[0] = rand_pg(&g, 0, 1.0, fabs(fi[0]), &nt);
            break;
        case LIK_NEGBINOMIAL:
            om[0] = rand_pg(&g, 0, (double)((const int32_t *)yv)[i] + lik->p[0], fabs(fi[0]), &nt);
            break;
        case LIK_STUDENTT: {
            double nu = lik->p[0], sg = lik->p[1], d = ((const double *)yv)[i] - fi[0];
            om[0] = (2.0 / (nu / (sg * sg) + d * d)) * rand_gamma(&g, (nu + 1.0) / 2.0);
        } break;
        case LIK_CATEGORICAL:
        case LIK_CATEGORICAL_BIJ: {
            const uint8_t *y = (const uint8_t *)yv;
            double sumth = cat_sum_theta(lik), sp = 0.0;
            for (int k = 0; k < Lf; ++k) sp += exp(lik->logtheta[k]) * logistic_(fi[k]) / sumth;
            double p0 = 1.0 - sp;
            if (!(sp < 1.0)) { bad |= 1; break; }
            double theta = (1.0 / p0 - 1.0) * rand_gamma(&g, 1.0);
            for (int k = 0; k < Lf; ++k)
                nn[k] = rand_poisson(&g, exp(lik->logtheta[k]) * logistic_(fi[k]) / sumth * theta / (1.0 - p0));
            for (int k = 0; k < Lf; ++k)
                om[k] = rand_pg(&g, k, (double)(nn[k] + (int64_t)y[i * Lf + k]), fabs(fi[k]), &nt);
        } break;
        case LIK_POISSON: {
            int64_t n1 = rand_poisson(&g, lik->p[0] * logistic_(-fi[0]));
            nn[0] = n1;
            om[0] = rand_pg(&g, 0, (double)(n1 + ((const int32_t *)yv)[i]), fabs(fi[0]), &nt);
        } break;
        case LIK_LAPLACE: {
            double beta = lik->p[0], lam = 1.0 / ((2.0 * beta) * (2.0 * beta));
            om[0] = rand_invgaussian(&g, 1.0 / (2.0 * beta * fabs(((const double *)yv)[i] - fi[0])), 2.0 * lam);
        } break;
        case LIK_HETEROGAUSS: {
            double yy = ((const double *)yv)[i];
            int64_t n1 = rand_poisson(&g, lik->p[0] * logistic_(-fi[1]) * (fi[0] - yy) * (fi[0] - yy) / 2.0);
            nn[0] = n1;
            om[0] = rand_pg(&g, 0, 0.5 + (double)n1, fabs(fi[1]), &nt);
        } break;
        default:
            bad |= 2;
        }
        if (nuni_out) nuni_out[i] = g.nuni;
    }
    if (bad) return -bad;
    /* potentials / precisions of the draw, rounded to float32 */
    double *b64 = (double *)malloc(sizeof(double) * 2 * (size_t)Lf * N);
    if (!b64) return -4;
    double *g64 = b64 + (size_t)Lf * N;
    agplo_potential_precision(lik, N, yv, omega, nout, f_out, b64, g64);
    for (int64_t k = 0; k < (int64_t)Lf * N; ++k) {
        beta32[k] = (float)b64[k];
        gamma32[k] = (float)g64[k];
    }
    free(b64);
    return 0;
}

AGPLO_API int agplo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
