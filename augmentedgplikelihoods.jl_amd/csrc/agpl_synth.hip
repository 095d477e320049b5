// agpl_synth.hip -- the step before the hot path: squared-exponential cross-covariance features and the
// synthetic workloads of SURVEY.md 8(d).  Every synthetic value is a pure function of (seed, index)
// through Philox4x32-10, so any index range can be regenerated anywhere (no transfer of N-sized inputs).
#include <math.h>

#include "agpl_common.h"
#include "agpl_random.h"

using namespace agpl;

namespace {

constexpr uint32_t kSynthSweep = 0xD47Au;

__device__ __forceinline__ double fstar(double x) { return 2.0 * sin(0.7 * x) + cos(0.23 * x); }

__global__ __launch_bounds__(256) void synth_xy_kernel(agpl_lik_dev lik, uint64_t seed, int64_t i0, int64_t n,
                                                       double *__restrict__ x_out, void *__restrict__ yv) {
    const int L = lik.nlatent;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        Philox g;
        g.init(seed, (uint64_t)(i0 + i), kSynthSweep);
        double x = -10.0 + 20.0 * g.u01();
        if (x_out) x_out[i] = x;
        if (!yv) continue;
        double fs = fstar(x);
        switch (lik.kind) {
        case AGPL_LIK_BERNOULLI_LOGISTIC:
            ((uint8_t *)yv)[i] = g.u01() < logistic(fs) ? 1 : 0;
            break;
        case AGPL_LIK_NEGBINOMIAL: {
            double p = logistic(0.5 * fs);
            double lam = rand_gamma(g, lik.p[0]) * p / (1.0 - p);
            ((int32_t *)yv)[i] = (int32_t)rand_poisson(g, lam);
        } break;
        case AGPL_LIK_STUDENTT: {
            double z = g.normal();
            double ch = 2.0 * rand_gamma(g, lik.p[0] / 2.0);
            ((float *)yv)[i] = (float)(fs + lik.p[1] * z / sqrt(ch / lik.p[0]));
        } break;
        case AGPL_LIK_CATEGORICAL:
        case AGPL_LIK_CATEGORICAL_BIJ: {
            // class weights w_k = theta_k * logistic(f*(x + 2k)); the bijective link's last class has the
            // constant weight theta_K / 2 and no one-hot row
            double tot = lik.kind == AGPL_LIK_CATEGORICAL_BIJ ? lik.cat_const : 0.0;
            for (int k = 0; k < L; ++k) tot += exp(lik.logtheta[k]) * logistic(fstar(x + 2.0 * k));
            double u = g.u01() * tot, cum = 0.0;
            int cls = L; // falls through to the implicit class
            for (int k = 0; k < L; ++k) {
                cum += exp(lik.logtheta[k]) * logistic(fstar(x + 2.0 * k));
                if (u < cum) {
                    cls = k;
                    break;
                }
            }
            if (lik.kind == AGPL_LIK_CATEGORICAL && cls == L) cls = L - 1;
            for (int k = 0; k < L; ++k) ((uint8_t *)yv)[i * L + k] = (k == cls) ? 1 : 0;
        } break;
        default:
            break;
        }
    }
}

// one thread per (4 features, point): float4 stores, coalesced along the feature index
__global__ __launch_bounds__(256) void se_features_kernel(int64_t N, int M, int ld, const double *__restrict__ x,
                                                          const double *__restrict__ z, double ell,
                                                          float *__restrict__ out) {
    const int q4 = ld >> 2;
    const int64_t total = N * (int64_t)q4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / q4;
        const int a = (int)(t - i * q4) << 2;
        const double xi = x[i];
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (a + e < M) {
                double d = (xi - z[a + e]) / ell;
                v[e] = (float)exp(-0.5 * d * d);
            } else {
                v[e] = 0.f;
            }
        }
        *reinterpret_cast<float4 *>(out + i * (int64_t)ld + a) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

} // namespace

extern "C" int32_t agpl_synth_xy(agpl_ctx *ctx, const agpl_lik_desc *lik, uint64_t seed, int64_t i0, int64_t n,
                                 double *x_out, void *y_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    agpl_lik_dev ld;
    int32_t rc = agpl_lik_to_device(ctx, lik, &ld);
    if (rc) return rc;
    if (n < 0 || i0 < 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "negative range");
    if (n == 0) return AGPL_OK;
    if (y_out && !(ld.kind == AGPL_LIK_BERNOULLI_LOGISTIC || ld.kind == AGPL_LIK_NEGBINOMIAL ||
                   ld.kind == AGPL_LIK_STUDENTT || ld.kind == AGPL_LIK_CATEGORICAL ||
                   ld.kind == AGPL_LIK_CATEGORICAL_BIJ))
        AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED, "no synthetic workload is defined for likelihood kind %d", ld.kind);
    int64_t nb = agpl_cdiv(n, 256);
    if (nb > 4096) nb = 4096;
    synth_xy_kernel<<<(unsigned)nb, 256, 0, ctx->stream>>>(ld, seed, i0, n, x_out, y_out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

extern "C" int32_t agpl_se_features(agpl_ctx *ctx, int64_t N, int32_t M, int32_t ld, const double *x,
                                    const double *z, double ell, float *out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (N < 0 || M <= 0 || ld < M || (ld & 3) || !(ell > 0.0))
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "need N >= 0, 0 < M <= ld, ld %% 4 == 0, ell > 0");
    if (N == 0) return AGPL_OK;
    if (!x || !z || !out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    int64_t nb = agpl_cdiv(N * (int64_t)(ld >> 2), 256);
    if (nb > 16384) nb = 16384;
    se_features_kernel<<<(unsigned)nb, 256, 0, ctx->stream>>>(N, M, ld, x, z, ell, out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}
