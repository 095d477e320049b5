// bench/agpl_bench.cpp -- the sparse CAVI sweep driven from C++ through nothing but include/agpl.h and the HIP runtime
// (SURVEY.md 8b caller (2): no Python, no torch): the loop of /root/reference examples/bernoulli/script.jl:32-38 in the
// whitened sparse form, on the synthetic workload of SURVEY.md 8d.  Mirrors bench.py's build_workload + SparseCAVI call
// for call (augmentedgplikelihoods.jl_amd/sparse.py), so that tests/test_gpu_cxx_driver.py can hold its natural parameters
// against the Python host's on the same seed -- the Python layer adds no arithmetic of its own.
//
//   agpl_bench [--n 10000000] [--m 512] [--sweeps 10] [--warmup 2] [--seed 20240807] [--dump file]
//
// prints ONE JSON line; --dump writes G [M*M] then g [M] (float64, little endian) after the last sweep.
// Build: see __graft_entry__.build() (hipcc -std=c++17 bench/agpl_bench.cpp -Iinclude -L<pkg> -lagpl).
#include "agpl.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

[[noreturn]] void die(const char *what, const char *detail) {
    std::fprintf(stderr, "agpl_bench: %s: %s\n", what, detail);
    std::exit(1);
}

#define HIP(call)                                                     \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess) die(#call, hipGetErrorString(e_));      \
    } while (0)

agpl_ctx *g_ctx = nullptr;
#define AGPL(call)                                                    \
    do {                                                              \
        int32_t rc_ = (call);                                         \
        if (rc_ != AGPL_OK) die(#call, g_ctx ? agpl_last_error(g_ctx) : "no context"); \
    } while (0)

template <typename T>
T *dalloc(size_t n, bool zero = false) {
    T *p = nullptr;
    HIP(hipMalloc((void **)&p, n * sizeof(T)));
    if (zero) HIP(hipMemset(p, 0, n * sizeof(T)));
    return p;
}

// lower Cholesky factor of the symmetric positive definite A (row-major, in place) and its inverse: the `_chol_cov` of
// examples/bernoulli/script.jl:30 with the 1e-8 jitter of :44, float64 on the host (M x M, once per data set)
void cholesky_and_inverse(int M, std::vector<double> &A, std::vector<double> &Linv) {
    for (int j = 0; j < M; ++j) {
        double d = A[(size_t)j * M + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * M + k] * A[(size_t)j * M + k];
        if (!(d > 0.0)) die("cholesky", "K_Z + jitter I is not positive definite");
        const double ljj = std::sqrt(d);
        A[(size_t)j * M + j] = ljj;
        for (int i = j + 1; i < M; ++i) {
            double s = A[(size_t)i * M + j];
            for (int k = 0; k < j; ++k) s -= A[(size_t)i * M + k] * A[(size_t)j * M + k];
            A[(size_t)i * M + j] = s / ljj;
        }
    }
    Linv.assign((size_t)M * M, 0.0);
    for (int c = 0; c < M; ++c) { // forward substitution, column c of the identity
        for (int i = c; i < M; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= A[(size_t)i * M + k] * Linv[(size_t)k * M + c];
            Linv[(size_t)i * M + c] = s / A[(size_t)i * M + i];
        }
    }
}

} // namespace

int main(int argc, char **argv) {
    int64_t N = 10000000;
    int M = 512, sweeps = 10, warmup = 2;
    uint64_t seed = 20240807ull; // bench.py SEED
    std::string dump;
    for (int i = 1; i < argc; ++i) {
        auto next = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) die(flag, "missing value");
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--n")) N = std::atoll(next("--n"));
        else if (!std::strcmp(argv[i], "--m")) M = std::atoi(next("--m"));
        else if (!std::strcmp(argv[i], "--sweeps")) sweeps = std::atoi(next("--sweeps"));
        else if (!std::strcmp(argv[i], "--warmup")) warmup = std::atoi(next("--warmup"));
        else if (!std::strcmp(argv[i], "--seed")) seed = std::strtoull(next("--seed"), nullptr, 10);
        else if (!std::strcmp(argv[i], "--dump")) dump = next("--dump");
        else die("unknown argument", argv[i]);
    }
    if (N <= 0 || M <= 0 || M % 256 || sweeps <= 0 || warmup < 0) die("arguments", "need N > 0, M a positive multiple of 256");
    if (agpl_version() != AGPL_VERSION) die("agpl_version", "header and library disagree");

    AGPL(agpl_ctx_create(&g_ctx, 0, seed));
    agpl_ctx *ctx = g_ctx;
    agpl_lik_desc lik;
    std::memset(&lik, 0, sizeof lik);
    lik.kind = AGPL_LIK_BERNOULLI_LOGISTIC; // examples/bernoulli/script.jl:20
    lik.nlatent = 1;

    // ---- setup (untimed): synthetic (x, y), K_ZX, whitening, Nystrom residual -- bench.py build_workload
    double *x = dalloc<double>((size_t)N);
    uint8_t *y = dalloc<uint8_t>((size_t)N);
    AGPL(agpl_synth_xy(ctx, &lik, seed, 0, N, x, y));
    std::vector<double> z(M), Kzz((size_t)M * M), Linv;
    const double step = 20.0 / (M - 1); // numpy.linspace(-10, 10, M): start + j * step, the last point exact
    for (int j = 0; j < M; ++j) z[j] = -10.0 + j * step;
    z[M - 1] = 10.0;
    const double ell = 1.5 * (z[1] - z[0]);
    for (int a = 0; a < M; ++a)
        for (int b = 0; b < M; ++b) {
            const double d = (z[a] - z[b]) / ell;
            Kzz[(size_t)a * M + b] = std::exp(-0.5 * d * d) + (a == b ? 1e-8 : 0.0);
        }
    cholesky_and_inverse(M, Kzz, Linv);
    double *zd = dalloc<double>((size_t)M);
    HIP(hipMemcpy(zd, z.data(), sizeof(double) * M, hipMemcpyHostToDevice));
    float *Kzx = dalloc<float>((size_t)N * M), *Phi = dalloc<float>((size_t)N * M);
    AGPL(agpl_se_features(ctx, N, M, M, x, zd, ell, Kzx));
    std::vector<float> At((size_t)M * M); // column-major L^-1 = row-major transpose
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) At[(size_t)j * M + i] = (float)Linv[(size_t)i * M + j];
    float *Atd = dalloc<float>((size_t)M * M);
    HIP(hipMemcpy(Atd, At.data(), sizeof(float) * M * M, hipMemcpyHostToDevice));
    AGPL(agpl_transform_features(ctx, N, M, Atd, Kzx, Phi));
    AGPL(agpl_ctx_synchronize(ctx));
    HIP(hipFree(Kzx));
    HIP(hipFree(x));
    // d_i = k_ii - |phi_i|^2 (agpl_feature_residual; sparse.py nystrom_residual), k_ii = 1 for the unit-variance kernel
    float *resid = dalloc<float>((size_t)N);
    {
        std::vector<float> ones((size_t)N, 1.0f);
        float *kxx = dalloc<float>((size_t)N);
        HIP(hipMemcpy(kxx, ones.data(), sizeof(float) * N, hipMemcpyHostToDevice));
        AGPL(agpl_feature_residual(ctx, N, M, Phi, kxx, resid));
        AGPL(agpl_ctx_synchronize(ctx));
        HIP(hipFree(kxx));
    }

    // ---- state of the sweep -- SparseCAVI.__init__: ONE plan (both split-float16 images of Phi with one scale, the residual,
    //      q(v) = N(0, I) in factor form, script.jl:41-42); the float32 features are not read by the sweep after this
    agpl_plan *plan = nullptr;
    AGPL(agpl_plan_create(ctx, N, M, 1, Phi, resid, 0u, nullptr, &plan));
    AGPL(agpl_ctx_synchronize(ctx));
    HIP(hipFree(Phi));
    HIP(hipFree(resid));
    double *Gg = dalloc<double>((size_t)M * M + M, true); // one flat buffer: the exchange step is one collective
    double *G = Gg, *g = Gg + (size_t)M * M;

    auto sweep = [&]() {
        // marginals -> aux_posterior! -> expected potential / precision -> (G, g)   script.jl:32-34
        AGPL(agpl_cavi_pass_plan(plan, &lik, nullptr, y, G, g, nullptr, nullptr, nullptr, nullptr));
        // S = (I + G)^-1, m = S g   script.jl:35-36
        AGPL(agpl_plan_update(plan, G, g, nullptr, nullptr));
    };
    for (int s = 0; s < warmup; ++s) sweep();
    AGPL(agpl_ctx_synchronize(ctx));
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < sweeps; ++s) sweep();
    AGPL(agpl_ctx_synchronize(ctx)); // also reports a failed factorisation
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<double> host((size_t)M * M + M);
    HIP(hipMemcpy(host.data(), Gg, sizeof(double) * host.size(), hipMemcpyDeviceToHost));
    double sumG = 0.0, sumg = 0.0;
    for (size_t i = 0; i < (size_t)M * M; ++i) sumG += host[i];
    for (int i = 0; i < M; ++i) sumg += host[(size_t)M * M + i];
    if (!dump.empty()) {
        FILE *f = std::fopen(dump.c_str(), "wb");
        if (!f || std::fwrite(host.data(), sizeof(double), host.size(), f) != host.size()) die("--dump", dump.c_str());
        std::fclose(f);
    }
    std::printf("{\"driver\": \"bench/agpl_bench.cpp (C ABI only)\", \"metric\": \"CAVI sweeps/sec (N obs, M inducing)\", "
                "\"value\": %.4f, \"unit\": \"sweeps/s\", \"ms_per_step\": %.3f, \"N\": %lld, \"M\": %d, \"sweeps\": %d, "
                "\"warmup\": %d, \"seed\": %llu, \"sum_G\": %.17g, \"sum_g\": %.17g, \"G00\": %.17g, \"g0\": %.17g}\n",
                sweeps / dt, dt / sweeps * 1e3, (long long)N, M, sweeps, warmup, (unsigned long long)seed, sumG, sumg,
                host[0], host[(size_t)M * M]);
    AGPL(agpl_plan_destroy(plan));
    AGPL(agpl_ctx_destroy(ctx));
    return 0;
}
