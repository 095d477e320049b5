// agpl_random.h -- per-lane counter RNG and the scalar samplers of the Gibbs half, device side.
//
// One Philox4x32-10 stream per (context seed, point index, sweep), so a sweep is reproducible for any launch
// geometry and resumable from (seed, sweep) alone.  Every PG(1, c) draw of a point lives on a SUB-STREAM of the
// point's stream (counter word 3 = (point index >> 32) in its low 8 bits, sub-stream id above: draw j of latent k has
// id 1 + (k << 16) + (j mod 65535) and starts its block counter (counter word 0) at (j div 65535) << 20 -- zero for the first
// 65535 draws, whose streams are what they were before round 6 --, the residual Gamma series id 1 + (k << 16) + 0xFFFF,
// id 0 is the main stream; b < 2^22 = kPgMaxB, polyagamma.jl:129-134 sums any b): the
// b = y + r draws of a negative-binomial point are independent work items that the kernels deal across the lanes of
// a workgroup, sorted by the sampler's branch (pg_int_sum_block, agpl_ops.hip), and sum left to right in draw order -- the order of the sequential draw_sum loop (polyagamma.jl:129-134).  Uniform -> double
// conversion, randexp and randn are fixed transforms of the stream (52-bit open-interval uniform,
// inversion, cosine Box-Muller) so that a float64 host evaluation of the same formulas consumes the
// stream identically.
//
// Sampler algorithms (reference file:line, paths relative to the reference repo):
//   sample_pg1 / a / mass_texpon / rand_truncated_inverse_gaussian / draw_sum / rand_gamma_sum
//       src/SpecialDistributions/polyagamma.jl:121-257
//   Gamma  (Distributions.jl 0.25, un-vendored): Marsaglia-Tsang, shape<1 boosted by exp(-E/shape)
//   Poisson(Distributions.jl 0.25, un-vendored): exponential-arrival count for mu<6; PTRS for mu>=6
//   InverseGaussian (Distributions.jl 0.25, un-vendored): Michael-Schucany-Haas
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace agpl {

constexpr double kPgT = 0.64;                        // polyagamma.jl:3
constexpr double kPi = 3.14159265358979323846;
constexpr double kPi2_8 = kPi * kPi / 8.0;           // polyagamma.jl:4
constexpr double kLogTwo = 0.69314718055994530942;
constexpr double kLog2Pi = 1.83787706640934548356;
constexpr double kSqrtHalf = 0.70710678118654752440;

struct Philox {
    uint32_t k0, k1;
    uint32_t c0, c1, c2, c3;
    uint32_t b0, b1, b2, b3;
    int pos;
    uint32_t nuni;

    __device__ __forceinline__ void init(uint64_t seed, uint64_t stream, uint32_t sweep) {
        k0 = (uint32_t)seed;
        k1 = (uint32_t)(seed >> 32);
        c0 = 0;
        c1 = sweep;
        c2 = (uint32_t)stream;
        c3 = (uint32_t)(stream >> 32);
        pos = 4;
        nuni = 0;
    }
    // sub-stream `id` of this point's stream (same key, sweep and point; fresh counter, starting at block `block0`)
    __device__ __forceinline__ Philox sub(uint32_t id, uint32_t block0 = 0u) const {
        Philox s = *this;
        s.c0 = block0;
        s.c3 = (c3 & 0xFFu) + (id << 8);
        s.pos = 4;
        s.nuni = 0;
        return s;
    }
    __device__ __forceinline__ void refill() {
        uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3, q0 = k0, q1 = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            // (64-bit products: one v_mad_u64_u32 gives the high and the low word; the 32-bit integer multiplies are quarter rate)
            const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
            const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
            uint32_t n0 = hi1 ^ x1 ^ q0;
            uint32_t n2 = hi0 ^ x3 ^ q1;
            x0 = n0; x1 = lo1; x2 = n2; x3 = lo0;
            q0 += 0x9E3779B9u;
            q1 += 0xBB67AE85u;
        }
        b0 = x0; b1 = x1; b2 = x2; b3 = x3;
        c0 += 1u;
        pos = 0;
    }
    // a fresh (sub-)stream whose first two uniforms -- all four words of block 0 -- are not needed: the state after them, without
    // the ten rounds of the block
    __device__ __forceinline__ void skip_first_block() {
        c0 += 1u; // (a fresh sub-stream: c0 is its first block, pg_draw_block0)
        pos = 4;
        nuni = 2u;
    }
    // uniform in the open interval (0,1): (k + 1/2) 2^-52 for a 52-bit k -- every value is exact in float64
    // (k + 1/2 needs 53 bits), so neither 0 nor 1 can come out of the rounding of the conversion
    __device__ __forceinline__ double u01() {
        if (pos >= 4) refill();
        uint32_t w0, w1;
        // (all four words read unconditionally: loads in the two arms of a branch are merged into ONE load through a selected
        //  address, which pins the whole stream in scratch memory -- 40 bytes stored per refill and two scratch loads per
        //  uniform inside the trial loops of the PG sampler, round 5)
        const uint32_t a0 = b0, a1 = b1, a2 = b2, a3 = b3;
        const bool first = pos == 0;
        w0 = first ? a0 : a2;
        w1 = first ? a1 : a3;
        pos += 2;
        nuni += 1u;
        uint64_t k = ((uint64_t)(w0 >> 6) << 26) | (uint64_t)(w1 >> 6);
        return ((double)k + 0.5) * 0x1.0p-52;
    }
    __device__ __forceinline__ double exp1() { return -log(u01()); }
    __device__ __forceinline__ double normal() {
        double u1 = u01();
        double u2 = u01();
        return sqrt(-2.0 * log(u1)) * cos(2.0 * kPi * u2);
    }
};

// StatsFuns.normlogcdf as called at polyagamma.jl:186-187
__device__ __forceinline__ double normlogcdf(double z) {
    if (z < -1.0) {
        if (z > -35.0) return log(0.5 * erfc(-z * kSqrtHalf));
        double iz2 = 1.0 / (z * z);
        double ser = 1.0 - iz2 * (1.0 - 3.0 * iz2 * (1.0 - 5.0 * iz2 * (1.0 - 7.0 * iz2)));
        return -0.5 * z * z - log(-z) - 0.5 * kLog2Pi + log(ser);
    }
    return log1p(-0.5 * erfc(z * kSqrtHalf));
}

// mean(::PolyaGamma) polyagamma.jl:25-31
template <typename T>
__device__ __forceinline__ T pg_mean(T b, T c) {
    if (c == T(0)) return b / T(4);
    return b / (T(2) * c) * tanh(c / T(2));
}

// a(n,x) polyagamma.jl:167-177
__device__ __forceinline__ double pg_a(int n, double x) {
    double k = (n + 0.5) * kPi;
    if (x > kPgT) return k * exp(-k * k * x / 2.0);
    if (x > 0.0) {
        double expnt = -3.0 / 2.0 * (log(kPi / 2.0) + log(x)) - 2.0 * (n + 0.5) * (n + 0.5) / x;
        return k * exp(expnt);
    }
    return __builtin_nan(""); // DomainError in the reference; unreachable from the sampler
}

// mass_texpon(z,K) polyagamma.jl:179-192
__device__ __forceinline__ double pg_mass_texpon(double z, double K) {
    const double t = kPgT;
    double b = sqrt(1.0 / t) * (t * z - 1.0);
    double a = -sqrt(1.0 / t) * (t * z + 1.0);
    if (z < 8.0) {
        // the same quantity without the round trip through logarithms: q / p = (4 / pi) K e^{K t} (e^{-z} Phi(b) + e^{z} Phi(a)),
        // Phi(x) = erfc(-x / sqrt 2) / 2: all terms positive (no cancellation), every factor far inside the float64 range for
        // z < 8 (K t < 22).  r is only ever compared with a uniform: the last-bit difference to the log-space form below is the
        // same order as the difference between two libm's evaluating that form (round 3: identical draws over 1.6e8 PG(1)).
        const double ez = exp(-z);
        const double qdivp = (4.0 / kPi) * (K * exp(K * t)) * (ez * (0.5 * erfc(-b * kSqrtHalf)) + (0.5 * erfc(-a * kSqrtHalf)) / ez);
        return 1.0 / (1.0 + qdivp);
    }
    double x0 = log(K) + K * t;
    double xb = x0 - z + normlogcdf(b);
    double xa = x0 + z + normlogcdf(a);
    double qdivp = (4.0 / kPi) * (exp(xb) + exp(xa));
    return 1.0 / (1.0 + qdivp);
}

// A bracket of r(z) = mass_texpon(z, K(z)) for the branch test `r > u` of sample_pg1 (polyagamma.jl:238): Chebyshev fit on
// z in [0, 8] (tools/fit_pg_mass.py: max |fit - exact| = 8.0e-11 against a 60-digit evaluation) widened by 1e-8 on both sides.
// u below the bracket takes the truncated-exponential branch, u above it the inverse-Gaussian branch, and a draw whose u falls
// INSIDE it (2e-8 of the draws) is handed to the sequential sampler with the exact formula (phase C of the kernels): the decisions
// are those of the exact formula, at a twentieth of its ~900 instructions (two erfc, three exp, a division) per point.
#define AGPL_PG_MASS_CHEB                                                                                           \
    0.17891350136487277, -0.29110615717444843, 0.14413720467193056, -0.019096796213082485, -0.030408673533667163,  \
        0.02413478068743331, -0.005763120734068512, -0.0025295518306110215, 0.0022193291620839425,                 \
        -0.00036720422556672745, -0.0002554337564001494, 0.0001344480400196417, 2.6363411059142603e-06,            \
        -1.989863672582531e-05, 3.945796121778492e-06, 1.8751767350099586e-06, -8.57276875529128e-07,              \
        -9.114466427503223e-08, 1.221518168946105e-07, -8.29437512251279e-09, -1.3409793302941801e-08,             \
        3.0516811956701023e-09, 1.0826812382200682e-09, -5.089763581593449e-10, -3.546254472440936e-11
constexpr double kPgMassChebLit[25] = {AGPL_PG_MASS_CHEB};
// The same table in constant memory (weak: one definition per code object that includes this header), read through scalar loads.  As
// compile-time literals the 25 coefficients are hoisted out of a kernel's point loop into 50 VGPRs: fine where the fit runs per
// DRAW (the multi-latent engine: literals measured 3-7 % faster there), the largest block of register pressure where it runs once
// per point (the PG(1) kernels and the one-latent engine, which spilled because of it) -- round 5.
__attribute__((weak)) __constant__ double kPgMassChebMem[25] = {AGPL_PG_MASS_CHEB};
#undef AGPL_PG_MASS_CHEB
constexpr double kPgMassSlack = 1e-8;
template <bool LITERAL = false>
__device__ __forceinline__ double pg_mass_fit(double z) { // z in [0, 8)
    const double x = z * 0.25 - 1.0;
    const double x2 = 2.0 * x;
    double b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int j = 24; j > 0; --j) {
        const double t = __builtin_fma(x2, b1, (LITERAL ? kPgMassChebLit[j] : kPgMassChebMem[j]) - b2);
        b2 = b1;
        b1 = t;
    }
    return __builtin_fma(x, b1, (LITERAL ? kPgMassChebLit[0] : kPgMassChebMem[0]) - b2);
}
// the bracket [rlo, rhi] of r for tilt c ([0, 1] where the fit does not apply: decided by the exact formula, phase C)
__device__ __forceinline__ void pg_mass_bracket(double c, double &z, double &K, double &rlo, double &rhi);

// rand_truncated_inverse_gaussian(rng,z) polyagamma.jl:195-221
__device__ inline double rand_tig(Philox &g, double z) {
    double mu = 1.0 / z;
    double x = 1.0 + kPgT;
    if (mu > kPgT) {
        double alpha = 0.0;
        while (alpha < g.u01()) {
            double E = g.exp1();
            double Ep = g.exp1();
            while (E * E > (2.0 * Ep / kPgT)) {
                E = g.exp1();
                Ep = g.exp1();
            }
            double d = 1.0 + E * kPgT;
            x = kPgT / (d * d);
            alpha = exp(-z * z * x / 2.0);
        }
    } else {
        while (x > kPgT) {
            double nrm = g.normal();
            double y = nrm * nrm;
            double muy = mu * y;
            x = mu + mu * muy / 2.0 - mu * sqrt(4.0 * muy + muy * muy) / 2.0;
            if (mu / (mu + x) < g.u01()) x = mu * mu / x;
        }
    }
    return x;
}

// Parameters of sample_pg1 that depend only on c (polyagamma.jl:226-236): hoisted so that the b draws of
// draw_sum (polyagamma.jl:129-134) evaluate mass_texpon once.
struct Pg1Params {
    double z, K, r;
    __device__ __forceinline__ void set(double c) {
        z = fabs(c) / 2.0;
        if (z == 0.0) {
            r = 0.5776972428360435; // polyagamma.jl:231
            K = kPi2_8;
        } else {
            K = kPi2_8 + z * z / 2.0;
            r = pg_mass_texpon(z, K);
        }
    }
};

__device__ __forceinline__ void pg_mass_bracket(double c, double &z, double &K, double &rlo, double &rhi) {
    z = fabs(c) / 2.0;
    K = kPi2_8 + z * z / 2.0; // (= Pg1Params::set, bit for bit: z = 0 adds an exact zero)
    if (z < 8.0) {
        const double r = pg_mass_fit(z);
        rlo = r - kPgMassSlack;
        rhi = r + kPgMassSlack;
    } else { // no fit: the bracket is all of (0, 1), i.e. every draw of this owner goes to the sequential sampler, which forms the exact
        rlo = 0.0, rhi = 1.0; // r (the ~900 instructions of mass_texpon inline here cost every point's prologue their registers)
    }
}
// sample_pg1(rng,c) polyagamma.jl:237-257.  The alternating-series accept loop runs with the wave's
// exec mask shrinking as lanes accept (the compiler's divergent-loop lowering is the wave ballot: the
// loop back-edge is s_cbranch on exec != 0).
__device__ inline double sample_pg1(Philox &g, const Pg1Params &p, uint32_t &nterms) {
    for (;;) {
        double x;
        if (p.r > g.u01())
            x = kPgT + g.exp1() / p.K;
        else
            x = rand_tig(g, p.z);
        double s = pg_a(0, x);
        double y = g.u01() * s;
        int n = 0;
        bool accepted = false;
        for (;;) {
            n += 1;
            if (n & 1) {
                s -= pg_a(n, x);
                if (y <= s) { accepted = true; break; }
            } else {
                s += pg_a(n, x);
                if (y > s) break;
            }
        }
        nterms += (uint32_t)n;
        if (accepted) return x / 4.0;
    }
}

// Marsaglia-Tsang
__device__ inline double rand_gamma_mt(Philox &g, double shape) {
    double d = shape - 1.0 / 3.0;
    double c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = g.normal();
        double v = 1.0 + c * x;
        while (v <= 0.0) {
            x = g.normal();
            v = 1.0 + c * x;
        }
        v = v * v * v;
        double u = g.u01();
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2 || log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return d * v;
    }
}
__device__ inline double rand_gamma(Philox &g, double shape) {
    if (!(shape > 0.0)) return __builtin_nan(""); // NaN / invalid shape: no draw (the rejection loops would not end)
    if (shape >= 1.0) return rand_gamma_mt(g, shape);
    double x = rand_gamma_mt(g, shape + 1.0);
    double e = g.exp1();
    return x * exp(-e / shape);
}

// rand_gamma_sum polyagamma.jl:157-164
__device__ inline double rand_gamma_sum(Philox &g, double c, double e) {
    const double inv2pi2 = (1.0 / (2.0 * kPi)) * (1.0 / kPi);
    double w = (c * (1.0 / (2.0 * kPi)));
    w = w * w;
    double acc = 0.0;
    for (int k = 1; k <= 200; ++k) acc += rand_gamma(g, e) / ((k - 0.5) * (k - 0.5) + w);
    return inv2pi2 * acc;
}

constexpr uint32_t kSubResidual = 0xFFFFu;
// PG(1, c) draw j of a latent: sub-stream id (sub_base =) 1 + (latent << 16) plus pg_draw_id(j), first block pg_draw_block0(j).
// The id has 16 bits for the draw (0xFFFF is the residual series): draws 0 .. 65534 are numbered by it alone, the block counter of
// their streams starts at zero -- the layout of rounds 2-5, bit for bit.  Draw j >= 65535 reuses id j mod 65535 and starts its
// block counter at (j div 65535) << 20: a PG(1, c) draw consumes a handful of blocks (2^20 blocks would be two million uniforms),
// so the streams of one id never meet.  kPgMaxB = 2^22 bounds b = y + r (the engine's per-wave draw offsets are 32-bit sums over
// up to 256 owners); beyond it AGPL_ERR_UNSUPPORTED as before.
constexpr double kPgMaxB = 4194304.0;
__device__ __forceinline__ uint32_t pg_draw_id(uint32_t j) { return j % 65535u; }
__device__ __forceinline__ uint32_t pg_draw_block0(uint32_t j) { return (j / 65535u) << 20; }

// rand(PolyaGamma(b,c)) polyagamma.jl:121-154, one lane doing all of a point's draws: draw j of `latent` on sub-stream
// 1 + (latent << 16) + pg_draw_id(j) from block pg_draw_block0(j), the residual series on 1 + (latent << 16) + 0xFFFF; uniforms
// consumed are added to g.nuni.
__device__ inline double rand_pg(Philox &g, int latent, double b, double c, uint32_t &nterms) {
    // NaN / Inf in, NaN out: the accept loops never terminate on a non-finite tilt (the reference would spin or
    // throw its DomainError from a(n, 0), polyagamma.jl:175); b >= kPgMaxB is outside the numbering of the draws
    if (!(b >= 0.0) || !(fabs(c) < __builtin_inf()) || !(b < kPgMaxB)) return __builtin_nan("");
    if (b == 0.0) return 0.0;
    const long tb = (long)floor(b);
    const uint32_t base = 1u + ((uint32_t)latent << 16);
    double acc = 0.0;
    if (tb > 0) {
        Pg1Params p;
        p.set(c);
        for (long j = 0; j < tb; ++j) {
            Philox s = g.sub(base + pg_draw_id((uint32_t)j), pg_draw_block0((uint32_t)j));
            acc += sample_pg1(s, p, nterms);
            g.nuni += s.nuni;
        }
    }
    const double res = b - (double)tb;
    if (res == 0.0) return acc;
    Philox s = g.sub(base + kSubResidual);
    acc += rand_gamma_sum(s, c, res);
    g.nuni += s.nuni;
    return acc;
}

// draw_sum(rng, PolyaGamma{<:Integer}) polyagamma.jl:129-134 for a compile-time-integer b (Bernoulli: b = 1):
// identical draws to rand_pg(g, 0, b, c, .) without carrying the non-integer Gamma-series code.
__device__ inline double rand_pg_int(Philox &g, int b, double c, uint32_t &nterms) {
    if (!(fabs(c) < __builtin_inf())) return __builtin_nan("");
    Pg1Params p;
    p.set(c);
    double acc = 0.0;
    for (int j = 0; j < b; ++j) {
        Philox s = g.sub(1u + (uint32_t)j);
        acc += sample_pg1(s, p, nterms);
        g.nuni += s.nuni;
    }
    return acc;
}

__device__ inline int64_t rand_poisson(Philox &g, double mu) {
    if (!(mu > 0.0)) return 0;
    if (mu < 6.0) {
        int64_t n = 0;
        double c = g.exp1();
        while (c < mu) {
            n += 1;
            c += g.exp1();
        }
        return n;
    }
    double slam = sqrt(mu), loglam = log(mu);
    double b = 0.931 + 2.53 * slam;
    double a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4);
    double vr = 0.9277 - 3.6224 / (b - 2.0);
    for (;;) {
        double U = g.u01() - 0.5;
        double V = g.u01();
        double us = 0.5 - fabs(U);
        double kf = floor((2.0 * a / us + b) * U + mu + 0.43);
        if (us >= 0.07 && V <= vr) return (int64_t)kf;
        if (kf < 0.0 || (us < 0.013 && V > us)) continue;
        if ((log(V) + log(invalpha) - log(a / (us * us) + b)) <= (-mu + kf * loglam - lgamma(kf + 1.0)))
            return (int64_t)kf;
    }
}

__device__ inline double rand_invgaussian(Philox &g, double mu, double lambda) {
    double z = g.normal();
    double v = z * z;
    double w = mu * v;
    double x1 = mu + mu / (2.0 * lambda) * (w - sqrt(w * (4.0 * lambda + w)));
    double p1 = mu / (mu + x1);
    double u = g.u01();
    return u >= p1 ? mu * mu / x1 : x1;
}

template <typename T>
__device__ __forceinline__ T logistic(T x) { return T(1) / (T(1) + exp(-x)); }

// approx_expected_logistic src/utils.jl:11-14 with LogExpFunctions._logistic_bounds per element type
__device__ __forceinline__ double approx_expected_logistic(double mu, double c) {
    if (mu < -744.4400719213812) return 0.0;
    if (mu > 36.7368005696771) return 1.0;
    return exp(mu / 2.0) * (1.0 / cosh(c / 2.0)) / 2.0;
}
__device__ __forceinline__ float approx_expected_logistic(float mu, float c) {
    if (mu < -103.27893f) return 0.0f;
    if (mu > 16.635532f) return 1.0f;
    return expf(mu / 2.0f) * (1.0f / coshf(c / 2.0f)) / 2.0f;
}

} // namespace agpl
