// agpl_update.hip -- the M x M Gaussian update of a sweep (a12, examples/bernoulli/script.jl:35-36 in the
// sparse whitened form of docs/src/index.md:154-163) and the fused-pass orchestration.
//
//   S = (I + G)^-1   (float64 Cholesky + inverse: rocSOLVER potrf / potri, plain library calls)
//   m = S (g + eta0)
//   Wpack = packed(-S) float32 for agpl_marginals ; alpha = m float32
//
// M <= a few thousand: M^3 work is < 1 % of the N M^2 contractions and stays on the sweep's stream.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <cstdlib>
#include <cstring>

#include "agpl_common.h"

// agpl_ops.hip / agpl_mfma.hip internals
int32_t agpl_pack_factor_split_info(agpl_ctx *ctx, int32_t M, int32_t L, const double *A, void *U_hi, void *U_lo,
                                    const int *info, int *info_host, int ninfo, int u_scale_exp); // agpl_split.hip
int32_t agpl_marginals_factor_internal(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const void *Phi_hi, const void *Phi_lo,
                                       const float *resid, const float *mu0, const void *U_hi, const void *U_lo, const float *v,
                                       float *mu_out, float *var_out, int image_scale_exp);
int32_t agpl_marginals_factor_parts(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const void *Phi_hi, const void *Phi_lo,
                                    const void *U_hi, const void *U_lo, const float *v, unsigned *zero2, float **qpart_out,
                                    float **mpart_out, unsigned **queues_out, int image_scale_exp); // agpl_split.hip
int32_t agpl_launch_fused_point(agpl_ctx *ctx, const agpl_lik_dev &ld, int64_t n, int64_t npad, int nb2, const void *y,
                                const float *resid, const float *mu0, const float *qpart, const float *mpart,
                                float *gamma, float *beta, float *c_out, float *gb, unsigned *scal, unsigned *queues,
                                double *elbo_terms_out); // agpl_ops.hip
void agpl_accumulate_records(int64_t N, int32_t M, int32_t L, void *slab_mem, float **gb, unsigned **scal); // agpl_mfma.hip
int32_t agpl_launch_fused_elementwise(agpl_ctx *ctx, const agpl_lik_dev &ld, int64_t n, const void *y,
                                      const float *mu, const float *var, float *gamma, float *beta, float *c_out);
int32_t agpl_accumulate_impl(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi, const void *acc_image, const float *beta,
                             const float *gamma, double *G_out, double *g_out, void *slab_mem, bool records_ready = false);
int32_t agpl_launch_gibbs_project_sample(agpl_ctx *ctx, const agpl_lik_dev &ld, int64_t N, int M, const float *Phi, const void *image,
                                         const float *kdiag, const float *mu0, const void *y, const double *v,
                                         uint32_t sweep, float *gamma, float *beta, double *f_out,
                                         double *omega_out, int64_t *n_out, uint32_t *nuni_out, int *bad,
                                         double *proj_work);
int32_t agpl_launch_randn(agpl_ctx *ctx, int64_t n, uint32_t sweep, double *out);
int32_t agpl_sampler_outcome(agpl_ctx *ctx, int32_t kind, const int *bad);
size_t agpl_slab_bytes(int64_t N, int32_t M, int32_t L);

namespace {

#define AGPL_ROCBLAS(ctx, call)                                                                     \
    do {                                                                                            \
        rocblas_status s__ = (call);                                                                \
        if (s__ != rocblas_status_success)                                                          \
            AGPL_FAIL(ctx, AGPL_ERR_HIP, "%s failed: rocblas_status %d (%s:%d)", #call, (int)s__,   \
                      __FILE__, __LINE__);                                                          \
    } while (0)

__global__ void add_identity_kernel(int M, const double *__restrict__ G, double *__restrict__ A) {
    const int l = blockIdx.z, row = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= M) return;
    const int64_t idx = ((int64_t)l * M + row) * M + col;
    A[idx] = G[idx] + (row == col ? 1.0 : 0.0);
}

// potri leaves one triangle valid.  With rocblas_fill_lower on our row-major storage (= upper in the
// column-major view of rocSOLVER... the matrix is symmetric so either reading is the same matrix) the
// valid entries are those with (column-major) row >= col, i.e. row-major A[c][r] for r >= c: element
// (i, j) of the row-major array is valid when j >= i.
__global__ void symmetrize_kernel(int M, double *__restrict__ A) {
    const int l = blockIdx.z, row = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= M || col >= row) return;
    double *Al = A + (int64_t)l * M * M;
    Al[(int64_t)row * M + col] = Al[(int64_t)col * M + row];
}

// m = S (g + eta0): one block per row, fixed-order tree reduce
__global__ __launch_bounds__(256) void symv_kernel(int M, const double *__restrict__ S, const double *__restrict__ g,
                                                   const double *__restrict__ eta0, double *__restrict__ m_out,
                                                   float *__restrict__ alpha_out) {
    __shared__ double sm[256];
    const int l = blockIdx.y, row = blockIdx.x;
    const double *Sr = S + ((int64_t)l * M + row) * M;
    const double *gl = g + (int64_t)l * M;
    const double *el = eta0 ? eta0 + (int64_t)l * M : nullptr;
    double acc = 0.0;
    for (int c = threadIdx.x; c < M; c += 256) acc += Sr[c] * (gl[c] + (el ? el[c] : 0.0));
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (m_out) m_out[(int64_t)l * M + row] = sm[0];
        if (alpha_out) alpha_out[(int64_t)l * M + row] = (float)sm[0];
    }
}

// Wpack[b][a] = scale * (b > a ? 2 W[a][b] : (b == a ? W[a][a] : 0))
__global__ void pack_w_kernel(int M, const double *__restrict__ W, double scale, float *__restrict__ Wp) {
    const int l = blockIdx.z, b = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= M) return;
    const double *Wl = W + (int64_t)l * M * M;
    double v = 0.0;
    if (b > a)
        v = 2.0 * Wl[(int64_t)b * M + a]; // symmetric: W[a][b] == W[b][a]; row b is the coalesced read
    else if (b == a)
        v = Wl[(int64_t)b * M + a];
    Wp[((int64_t)l * M + b) * M + a] = (float)(scale * v);
}

int32_t get_handle(agpl_ctx *ctx, rocblas_handle *h) {
    if (!ctx->rocblas) {
        rocblas_handle hh = nullptr;
        AGPL_ROCBLAS(ctx, rocblas_create_handle(&hh));
        ctx->rocblas = hh;
    }
    *h = (rocblas_handle)ctx->rocblas;
    AGPL_ROCBLAS(ctx, rocblas_set_stream(*h, ctx->stream));
    return AGPL_OK;
}

} // namespace

int32_t agpl_get_rocblas(agpl_ctx *ctx, void **handle_out) {
    rocblas_handle h;
    int32_t rc = get_handle(ctx, &h);
    *handle_out = h;
    return rc;
}

extern "C" void agpl_update_release(agpl_ctx *ctx) {
    if (ctx && ctx->rocblas) {
        rocblas_destroy_handle((rocblas_handle)ctx->rocblas);
        ctx->rocblas = nullptr;
    }
}

namespace {
// logdet[l] = 2 sum_i log C_ii of the Cholesky factor (read between potrf and potri), fixed-order tree
__global__ __launch_bounds__(256) void logdet_kernel(int M, const double *__restrict__ A, double *__restrict__ out) {
    __shared__ double sm[256];
    const int l = blockIdx.x;
    const double *Al = A + (int64_t)l * M * M;
    double acc = 0.0;
    for (int i = threadIdx.x; i < M; i += 256) acc += log(Al[(int64_t)i * M + i]);
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[l] = 2.0 * sm[0];
}
// KL(N(m, S) || N(0, I)) = (tr S + m'm - M + logdet(I + G)) / 2 per latent
__global__ __launch_bounds__(256) void gauss_kl_kernel(int M, const double *__restrict__ S, const double *__restrict__ m,
                                                       const double *__restrict__ logdet, double *__restrict__ out) {
    __shared__ double sm[256];
    const int l = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < M; i += 256) {
        const double mi = m[(int64_t)l * M + i];
        acc += S[((int64_t)l * M + i) * M + i] + mi * mi;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[l] = 0.5 * (sm[0] - (double)M + logdet[l]);
}
__global__ void sum_latents_kernel(int L, const double *__restrict__ per, double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int l = 0; l < L; ++l) t += per[l]; // fixed order
        *out = t;
    }
}
} // namespace

int32_t agpl_factor_fused(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g, const double *eta0,
                          double *T_work, double *A_work, double *v_out, float *v32_out, double *logdet_out,
                          int *info_dev, void *coop_work);
size_t agpl_factor_coop_bytes(int32_t M, int32_t L); // agpl_factor.hip
// the hand-written factorisation (agpl_factor.hip) takes this shape; everything else -- M > 2048, or a feature count that is not a
// multiple of 32 (128 beyond 512) -- goes to rocSOLVER.  A plan pads to a multiple of 256, so its sweeps take the library route only
// beyond M = 2048.  (Rounds 4-5 had a third route, two block rows of the M <= 512 kernel around four library GEMMs, for
// 512 < M <= 1024 with M % 128 != 0 or L > 8: removed in round 6 -- the pipeline form runs eight latents per launch instead.)
static inline bool factor_one_launch(int32_t M, int32_t L) {
    if (M % 32 || L > 64) return false;
    if (M <= 512) return true;
    return M <= 2048 && M % 128 == 0; // (beyond 1024: two block rows around the one-launch kernel, agpl_factor_two_block)
}
// agpl_dense.hip: U = chol(I + G)^-1 for 1024 < M <= 2048 as two block rows of the one-launch kernel and four products on the
// float64 tile routine -- no library call
size_t agpl_factor_two_block_bytes(int32_t M);
int32_t agpl_factor_two_block(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, double *T_work, double *A_work, double *logdet_out,
                              int *info_dev, void *work);
static inline size_t factor_work_bytes(int32_t M, int32_t L) {
    return M <= 1024 ? agpl_factor_coop_bytes(M, L) : agpl_factor_two_block_bytes(M);
}
// agpl_factor_fused's contract for every M factor_one_launch accepts (defined behind factor_apply_kernel)
static int32_t factor_any(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g, const double *eta0, double *T_work,
                          double *A_work, double *v_out, float *v32_out, double *logdet_out, int *info_dev, void *work);

namespace {
// Uz = the factor with its foreign triangle zeroed: Uz[i][j] = A[i][j] for i <= j (row-major; = U[j][i]), else 0
__global__ void factor_clean_kernel(int M, const double *__restrict__ A, double *__restrict__ Uz) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, l = blockIdx.z;
    if (j < M) {
        const int64_t idx = ((int64_t)l * M + i) * M + j;
        Uz[idx] = i <= j ? A[idx] : 0.0;
    }
}
} // namespace

static int32_t gaussian_update_impl(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                    const double *eta0, double *S_out, double *m_out, float *Wpack_out,
                                    float *alpha_out, double *logdet_dev) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (M <= 0 || L <= 0 || !G || !g) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    rocblas_handle h;
    int32_t rc = get_handle(ctx, &h);
    if (rc) return rc;
    const size_t mat_bytes = sizeof(double) * (size_t)L * M * M;
    const size_t info_off = 16384; // ws2 head is used by the reductions
    if (factor_one_launch(M, L)) {
        // hand-written factorisation (agpl_factor.hip): U = chol(I + G)^-1, then S = U'U as one float64 GEMM -- against 3.4 ms for the
        // ~300 launches of potrf + potri
        const size_t coop_bytes = (factor_work_bytes(M, L) + 255) & ~(size_t)255;
        const size_t own = info_off + 1024;
        rc = agpl_ws2_reserve(ctx, own + 3 * mat_bytes + (S_out ? 0 : mat_bytes) + coop_bytes + 1024);
        if (rc) return rc;
        int *info = (int *)((char *)ctx->ws2 + info_off);
        char *p = (char *)ctx->ws2 + own;
        double *T = (double *)p, *Aw = (double *)(p + mat_bytes), *Uz = (double *)(p + 2 * mat_bytes);
        double *S = S_out ? S_out : (double *)(p + 3 * mat_bytes);
        void *coop = (void *)(p + 3 * mat_bytes + (S_out ? 0 : mat_bytes));
        rc = factor_any(ctx, M, L, G, g, eta0, T, Aw, nullptr, nullptr, logdet_dev, info, coop);
        if (rc) return rc;
        dim3 grid((unsigned)agpl_cdiv(M, 128), (unsigned)M, (unsigned)L);
        factor_clean_kernel<<<grid, 128, 0, ctx->stream>>>(M, Aw, Uz);
        AGPL_LAUNCH_CHECK(ctx);
        const double one = 1.0, zero = 0.0;
        AGPL_ROCBLAS(ctx, rocblas_set_pointer_mode(h, rocblas_pointer_mode_host));
        // the array Uz read column-major is U: S = U'U
        AGPL_ROCBLAS(ctx, rocblas_dgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, M, M, M,
                                                        &one, Uz, M, (rocblas_stride)M * M, Uz, M, (rocblas_stride)M * M,
                                                        &zero, S, M, (rocblas_stride)M * M, L));
        symmetrize_kernel<<<grid, 128, 0, ctx->stream>>>(M, S);
        AGPL_LAUNCH_CHECK(ctx);
        if (m_out || alpha_out) {
            dim3 g2((unsigned)M, (unsigned)L);
            symv_kernel<<<g2, 256, 0, ctx->stream>>>(M, S, g, eta0, m_out, alpha_out);
            AGPL_LAUNCH_CHECK(ctx);
        }
        if (Wpack_out) {
            pack_w_kernel<<<grid, 128, 0, ctx->stream>>>(M, S, -1.0, Wpack_out);
            AGPL_LAUNCH_CHECK(ctx);
        }
        int hinfo[64];
        const int ni = L;
        AGPL_HIP(ctx, hipMemcpyAsync(hinfo, info, sizeof(int) * ni, hipMemcpyDeviceToHost, ctx->stream));
        AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < ni; ++i)
            if (hinfo[i] != 0)
                AGPL_FAIL(ctx, hinfo[i] < 0 ? AGPL_ERR_HIP : AGPL_ERR_NOT_POSDEF,
                          hinfo[i] < 0 ? "factor kernel: a cooperating workgroup never arrived (latent %d, %d)"
                                       : "I + G is not positive definite (latent %d, pivot at row %d)",
                          i, (int)hinfo[i] - 1);
        return AGPL_OK;
    }
    rc = agpl_ws2_reserve(ctx, info_off + sizeof(rocblas_int) * 2 * (size_t)L + 256 + (S_out ? 0 : mat_bytes));
    if (rc) return rc;
    rocblas_int *info = (rocblas_int *)((char *)ctx->ws2 + info_off);
    double *A = S_out ? S_out : (double *)((char *)ctx->ws2 + info_off + 256 + sizeof(rocblas_int) * 2 * (size_t)L);
    A = (double *)(((uintptr_t)A + 255) & ~(uintptr_t)255);

    dim3 grid((unsigned)agpl_cdiv(M, 128), (unsigned)M, (unsigned)L);
    add_identity_kernel<<<grid, 128, 0, ctx->stream>>>(M, G, A);
    AGPL_LAUNCH_CHECK(ctx);
    const rocblas_stride stride = (rocblas_stride)M * M;
    AGPL_ROCBLAS(ctx, rocsolver_dpotrf_strided_batched(h, rocblas_fill_lower, M, A, M, stride, info, L));
    if (logdet_dev) {
        logdet_kernel<<<(unsigned)L, 256, 0, ctx->stream>>>(M, A, logdet_dev);
        AGPL_LAUNCH_CHECK(ctx);
    }
    AGPL_ROCBLAS(ctx, rocsolver_dpotri_strided_batched(h, rocblas_fill_lower, M, A, M, stride, info + L, L));
    symmetrize_kernel<<<grid, 128, 0, ctx->stream>>>(M, A);
    AGPL_LAUNCH_CHECK(ctx);
    if (m_out || alpha_out) {
        dim3 g2((unsigned)M, (unsigned)L);
        symv_kernel<<<g2, 256, 0, ctx->stream>>>(M, A, g, eta0, m_out, alpha_out);
        AGPL_LAUNCH_CHECK(ctx);
    }
    if (Wpack_out) {
        pack_w_kernel<<<grid, 128, 0, ctx->stream>>>(M, A, -1.0, Wpack_out);
        AGPL_LAUNCH_CHECK(ctx);
    }
    // PosDefException check (one small D2H + sync per sweep)
    rocblas_int hinfo[128];
    const int ni = 2 * L > 128 ? 128 : 2 * L;
    AGPL_HIP(ctx, hipMemcpyAsync(hinfo, info, sizeof(rocblas_int) * ni, hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < ni; ++i)
        if (hinfo[i] != 0)
            AGPL_FAIL(ctx, AGPL_ERR_NOT_POSDEF, "I + G is not positive definite (latent %d, %s info = %d)", i % L,
                      i < L ? "potrf" : "potri", (int)hinfo[i]);
    return AGPL_OK;
}

namespace {
// v[a] = sum_{b <= a} U[a][b] (g + eta0)[b] with U[a][b] = A[b * M + a] (column-major lower triangle), one wave-row
// of the fixed-order tree per output; also the float32 copy the marginal kernel stages
__global__ __launch_bounds__(256) void factor_apply_kernel(int M, const double *__restrict__ A,
                                                           const double *__restrict__ g,
                                                           const double *__restrict__ eta0, double *__restrict__ v,
                                                           float *__restrict__ v32) {
    __shared__ double sm[256];
    const int a = blockIdx.x, l = blockIdx.y;
    const double *Al = A + (int64_t)l * M * M;
    double acc = 0.0;
    for (int b = threadIdx.x; b <= a; b += 256) {
        const double r = g[(int64_t)l * M + b] + (eta0 ? eta0[(int64_t)l * M + b] : 0.0);
        acc += Al[(int64_t)b * M + a] * r;
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (v) v[(int64_t)l * M + a] = sm[0];
        if (v32) v32[(int64_t)l * M + a] = (float)sm[0];
    }
}
} // namespace

static int32_t factor_any(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g, const double *eta0, double *T_work,
                          double *A_work, double *v_out, float *v32_out, double *logdet_out, int *info_dev, void *work) {
    if (M <= 1024) return agpl_factor_fused(ctx, M, L, G, g, eta0, T_work, A_work, v_out, v32_out, logdet_out, info_dev, work);
    const int32_t rc = agpl_factor_two_block(ctx, M, L, G, T_work, A_work, logdet_out, info_dev, work);
    if (rc) return rc;
    if (v_out || v32_out) {
        factor_apply_kernel<<<dim3((unsigned)M, (unsigned)L), 256, 0, ctx->stream>>>(M, A_work, g, eta0, v_out, v32_out);
        AGPL_LAUNCH_CHECK(ctx);
    }
    return AGPL_OK;
}

// ---- feature counts the hand-written kernels do not take as they are: zero-padded copies (round 6) ------------------------------
// Zero features change nothing: G, g have zero rows / columns there, I + G is the identity there, U = chol(I + G)^-1 and v are
// the caller's in their leading block.  A plan pads to a multiple of 256 (agpl_plan.hip); agpl_gibbs_draw_v pads to the next count
// the factor kernels take.
namespace {
// dst [L, Mp, Mp] <- src [L, Mc, Mc] in the leading block, zero elsewhere (vectors: [L, Mp] <- [L, Mc]); a null source leaves zeros
__global__ __launch_bounds__(256) void pad_natural_kernel(int L, int Mc, int Mp, const double *__restrict__ G, const double *__restrict__ g,
                                                       const double *__restrict__ e, const double *__restrict__ v,
                                                       double *__restrict__ Gp, double *__restrict__ gp, double *__restrict__ ep,
                                                       double *__restrict__ vp) {
    const int64_t nm = (int64_t)L * Mp * Mp, stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (G)
        for (int64_t i = t0; i < nm; i += stride) {
            const int l = (int)(i / ((int64_t)Mp * Mp));
            const int64_t r = i - (int64_t)l * Mp * Mp;
            const int a = (int)(r / Mp), b = (int)(r - (int64_t)a * Mp);
            Gp[i] = (a < Mc && b < Mc) ? G[((int64_t)l * Mc + a) * Mc + b] : 0.0;
        }
    for (int64_t i = t0; i < (int64_t)L * Mp; i += stride) {
        const int l = (int)(i / Mp), a = (int)(i - (int64_t)l * Mp);
        const int64_t j = (int64_t)l * Mc + a;
        if (g) gp[i] = a < Mc ? g[j] : 0.0;
        if (ep) ep[i] = (e && a < Mc) ? e[j] : 0.0;
        if (v) vp[i] = a < Mc ? v[j] : 0.0;
    }
}
// the caller's G [L, Mc, Mc], g [L, Mc] <- the leading blocks of the padded ones
__global__ __launch_bounds__(256) void unpad_natural_kernel(int L, int Mc, int Mp, const double *__restrict__ Gp, const double *__restrict__ gp,
                                                         double *__restrict__ G, double *__restrict__ g) {
    const int64_t nm = (int64_t)L * Mc * Mc, stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < nm; i += stride) {
        const int l = (int)(i / ((int64_t)Mc * Mc));
        const int64_t r = i - (int64_t)l * Mc * Mc;
        const int a = (int)(r / Mc), b = (int)(r - (int64_t)a * Mc);
        G[i] = Gp[((int64_t)l * Mp + a) * Mp + b];
    }
    for (int64_t i = t0; i < (int64_t)L * Mc; i += stride) {
        const int l = (int)(i / Mc), a = (int)(i - (int64_t)l * Mc);
        g[i] = gp[(int64_t)l * Mp + a];
    }
}

} // namespace
int32_t agpl_pad_natural(agpl_ctx *ctx, int L, int Mc, int Mp, const double *G, const double *g, const double *e, const double *v,
                         double *Gp, double *gp, double *ep, double *vp) {
    pad_natural_kernel<<<G ? 256 : 8, 256, 0, ctx->stream>>>(L, Mc, Mp, G, g, e, v, Gp, gp, ep, vp);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}
int32_t agpl_unpad_natural(agpl_ctx *ctx, int L, int Mc, int Mp, const double *Gp, const double *gp, double *G, double *g) {
    unpad_natural_kernel<<<256, 256, 0, ctx->stream>>>(L, Mc, Mp, Gp, gp, G, g);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

int32_t agpl_pending_resolve(agpl_ctx *ctx) {
    if (!ctx->pend) return AGPL_OK;
    ctx->pend = false;
    AGPL_HIP(ctx, hipEventSynchronize(ctx->pend_ev));
    const int L = ctx->pend_latents;
    if (ctx->pend_gamma_word) {
        const unsigned bad = (unsigned)ctx->pend_host[127];
        ctx->pend_host[127] = 0;
        if (bad)
            AGPL_FAIL(ctx, AGPL_ERR_DOMAIN,
                      "the expected precision gamma of a sweep is negative or not finite (first at flat index %u of the "
                      "[latent][point] array): observations or marginals outside the likelihood's domain",
                      bad - 1u);
    }
    for (int i = 0; i < ctx->pend_n; ++i) {
        const int info = ctx->pend_host[i];
        if (info < 0)
            AGPL_FAIL(ctx, AGPL_ERR_HIP, "factor kernel: a cooperating workgroup never arrived (latent %d)", i % L);
        if (info != 0)
            AGPL_FAIL(ctx, AGPL_ERR_NOT_POSDEF, "I + G is not positive definite (latent %d, pivot block at row %d)", i % L,
                      info - 1 + (i < L ? 0 : 512));
    }
    return AGPL_OK;
}

// the info words of a factorisation that was just enqueued: copied to pinned host memory behind it, checked by
// agpl_pending_resolve
static int32_t pending_prepare(agpl_ctx *ctx) {
    if (!ctx->pend_host) {
        AGPL_HIP(ctx, hipHostMalloc((void **)&ctx->pend_host, sizeof(int) * 128, hipHostMallocMapped));
        AGPL_HIP(ctx, hipHostGetDevicePointer((void **)&ctx->pend_host_dev, ctx->pend_host, 0));
        memset(ctx->pend_host, 0, sizeof(int) * 128);
    }
    if (!ctx->pend_ev) AGPL_HIP(ctx, hipEventCreateWithFlags(&ctx->pend_ev, hipEventDisableTiming));
    return AGPL_OK;
}
// info_dev == nullptr: the last kernel enqueued has written the words to pend_host itself (agpl_pack_factor_split_info)
static int32_t pending_arm(agpl_ctx *ctx, const int *info_dev, int n, int L) {
    int32_t rc = pending_prepare(ctx);
    if (rc) return rc;
    ctx->pend_gamma_word = info_dev == nullptr;
    if (info_dev)
        AGPL_HIP(ctx, hipMemcpyAsync(ctx->pend_host, info_dev, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipEventRecord(ctx->pend_ev, ctx->stream));
    ctx->pend = true;
    ctx->pend_n = n;
    ctx->pend_latents = L;
    return AGPL_OK;
}

static int32_t gaussian_factor_enqueue(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                       const double *eta0, double *A_work, double *v_out, float *v32_out, void *U_hi,
                                       void *U_lo, double *logdet_out, bool *armed, int u_scale_exp = 0);

// I + G = R R' ; U = R^-1 ; v = U (g + eta0).  S = U'U and m = U'v are never formed: the factor form of the marginal
// pass consumes U and v directly.  Asynchronous: the outcome (AGPL_ERR_NOT_POSDEF, ...) is reported by the next plan pass
// (after it has enqueued its own kernels -- the host never idles the GPU between the update and the next pass), the next
// agpl_gaussian_factor / agpl_plan_update, or agpl_ctx_synchronize on this context.  Everything enqueued behind a failed
// factorisation computes on NaNs; nothing is lost but the timing of the report.
extern "C" int32_t agpl_gaussian_factor(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                        const double *eta0, double *A_work, double *v_out, double *logdet_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    bool armed = false;
    int32_t rc = gaussian_factor_enqueue(ctx, M, L, G, g, eta0, A_work, v_out, nullptr, nullptr, nullptr, logdet_out, &armed);
    if (rc) return rc;
    return armed ? AGPL_OK : agpl_pending_resolve(ctx);
}

// (agpl_plan.hip) the asynchronous form with the images of 2^u_scale_exp U
int32_t agpl_gaussian_factor_async_scaled(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                          const double *eta0, double *A_work, double *v_out, float *v32_out, void *U_hi,
                                          void *U_lo, double *logdet_out, int u_scale_exp) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    bool armed = false;
    int32_t rc = gaussian_factor_enqueue(ctx, M, L, G, g, eta0, A_work, v_out, v32_out, U_hi, U_lo, logdet_out, &armed, u_scale_exp);
    if (rc) return rc;
    return armed ? AGPL_OK : agpl_pending_resolve(ctx);
}

static int32_t gaussian_factor_enqueue(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                       const double *eta0, double *A_work, double *v_out, float *v32_out, void *U_hi,
                                       void *U_lo, double *logdet_out, bool *armed, int u_scale_exp) {
    if (M <= 0 || L <= 0 || L > 64 || !G || !g || !A_work) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    if ((U_hi == nullptr) != (U_lo == nullptr)) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "U_hi and U_lo go together");
    {
        const int32_t rp = agpl_pending_resolve(ctx); // the previous factorisation's outcome, before its slot is reused
        if (rp) return rp;
    }
    if (factor_one_launch(M, L)) {
        // one launch: blocked Cholesky + inverse factor + v + logdet (agpl_factor.hip)
        const size_t info_off = 16384, mat_bytes = sizeof(double) * (size_t)L * M * M;
        const size_t coop_bytes = factor_work_bytes(M, L);
        int32_t rc = agpl_ws2_reserve(ctx, info_off + 1024 + mat_bytes + coop_bytes);
        if (rc) return rc;
        int *info = (int *)((char *)ctx->ws2 + info_off);
        double *T = (double *)((char *)ctx->ws2 + info_off + 1024);
        void *coop = (char *)ctx->ws2 + info_off + 1024 + mat_bytes;
        rc = factor_any(ctx, M, L, G, g, eta0, T, A_work, v_out, v32_out, logdet_out, info, coop);
        if (rc) return rc;
        if (U_hi) {
            rc = pending_prepare(ctx);
            if (rc) return rc;
            rc = agpl_pack_factor_split_info(ctx, M, L, A_work, U_hi, U_lo, info, ctx->pend_host_dev, L, u_scale_exp);
            if (rc) return rc;
        }
        rc = pending_arm(ctx, U_hi ? nullptr : info, L, L);
        if (rc) return rc;
        *armed = true;
        return AGPL_OK;
    }
    rocblas_handle h;
    int32_t rc = get_handle(ctx, &h);
    if (rc) return rc;
    const size_t info_off = 16384;
    rc = agpl_ws2_reserve(ctx, info_off + sizeof(rocblas_int) * 2 * (size_t)L + 256);
    if (rc) return rc;
    rocblas_int *info = (rocblas_int *)((char *)ctx->ws2 + info_off);
    dim3 grid((unsigned)agpl_cdiv(M, 128), (unsigned)M, (unsigned)L);
    add_identity_kernel<<<grid, 128, 0, ctx->stream>>>(M, G, A_work);
    AGPL_LAUNCH_CHECK(ctx);
    const rocblas_stride stride = (rocblas_stride)M * M;
    AGPL_ROCBLAS(ctx, rocsolver_dpotrf_strided_batched(h, rocblas_fill_lower, M, A_work, M, stride, info, L));
    if (logdet_out) {
        logdet_kernel<<<(unsigned)L, 256, 0, ctx->stream>>>(M, A_work, logdet_out);
        AGPL_LAUNCH_CHECK(ctx);
    }
    AGPL_ROCBLAS(ctx, rocsolver_dtrtri_strided_batched(h, rocblas_fill_lower, rocblas_diagonal_non_unit, M, A_work, M,
                                                       stride, info + L, L));
    if (v_out || v32_out) {
        dim3 g2((unsigned)M, (unsigned)L);
        factor_apply_kernel<<<g2, 256, 0, ctx->stream>>>(M, A_work, g, eta0, v_out, v32_out);
        AGPL_LAUNCH_CHECK(ctx);
    }
    if (U_hi) {
        rc = agpl_pack_factor_split_info(ctx, M, L, A_work, U_hi, U_lo, nullptr, nullptr, 0, u_scale_exp);
        if (rc) return rc;
    }
    rocblas_int hinfo[128];
    const int ni = 2 * L;
    AGPL_HIP(ctx, hipMemcpyAsync(hinfo, info, sizeof(rocblas_int) * ni, hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < ni; ++i)
        if (hinfo[i] != 0)
            AGPL_FAIL(ctx, AGPL_ERR_NOT_POSDEF, "I + G is not positive definite (latent %d, %s info = %d)", i % L,
                      i < L ? "potrf" : "trtri", (int)hinfo[i]);
    return AGPL_OK;
}

extern "C" int32_t agpl_gaussian_update(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                        const double *eta0, double *S_out, double *m_out, float *Wpack_out,
                                        float *alpha_out, double *kl_out) {
    if (!kl_out) return gaussian_update_impl(ctx, M, L, G, g, eta0, S_out, m_out, Wpack_out, alpha_out, nullptr);
    // KL(q(v) || N(0, I)) = sum_l (tr S_l + m_l'm_l - M + logdet(I + G_l)) / 2 of the q(v) just formed: needs S, m and the
    // log-determinant read off the Cholesky factor; what the caller does not take lives in the large workspace
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (M <= 0 || L <= 0 || L > 64 || !G || !g) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    const size_t mat = sizeof(double) * (size_t)L * M * M, vec = sizeof(double) * (size_t)L * M;
    int32_t rc = agpl_ws_reserve(ctx, mat + vec + 4096);
    if (rc) return rc;
    double *S = S_out ? S_out : (double *)ctx->ws, *m = m_out ? m_out : (double *)((char *)ctx->ws + mat);
    double *ld = (double *)((char *)ctx->ws + mat + vec), *kl = ld + 64;
    rc = gaussian_update_impl(ctx, M, L, G, g, eta0, S, m, Wpack_out, alpha_out, ld);
    if (rc) return rc;
    gauss_kl_kernel<<<(unsigned)L, 256, 0, ctx->stream>>>(M, S, m, ld, kl);
    AGPL_LAUNCH_CHECK(ctx);
    sum_latents_kernel<<<1, 64, 0, ctx->stream>>>(L, kl, kl_out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

extern "C" int32_t agpl_cavi_pass(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M, const float *Phi,
                                  const float *kdiag, const float *mu0, const void *y, const float *Wpack,
                                  const float *alpha, double *G_out, double *g_out, float *c_out, float *gamma_out,
                                  float *beta_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    agpl_lik_dev ld;
    int32_t rc = agpl_lik_to_device(ctx, lik, &ld);
    if (rc) return rc;
    const int L = ld.nlatent;
    if (N <= 0 || M <= 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes N=%lld M=%d", (long long)N, M);
    if (M % 128) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "M = %d must be a multiple of 128 (zero-pad the features)", M);
    if (!Phi || !kdiag || !y || !Wpack || !alpha || !G_out || !g_out)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    const size_t slab = (agpl_slab_bytes(N, M, L) + 255) & ~(size_t)255;
    const size_t vec = (sizeof(float) * (size_t)L * N + 255) & ~(size_t)255;
    rc = agpl_ws_reserve(ctx, slab + 4 * vec);
    if (rc) return rc;
    char *base = (char *)ctx->ws;
    float *mu = (float *)(base + slab);
    float *var = (float *)(base + slab + vec);
    float *gam = gamma_out ? gamma_out : (float *)(base + slab + 2 * vec);
    float *bet = beta_out ? beta_out : (float *)(base + slab + 3 * vec);
    rc = agpl_marginals(ctx, N, M, L, Phi, kdiag, mu0, Wpack, alpha, mu, var);
    if (rc) return rc;
    rc = agpl_launch_fused_elementwise(ctx, ld, N, y, mu, var, gam, bet, c_out);
    if (rc) return rc;
    return agpl_accumulate_impl(ctx, N, M, L, Phi, nullptr, bet, gam, G_out, g_out, base);
}

// (also agpl_plan.hip)
int32_t agpl_gibbs_pass_internal(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M, const float *Phi, const void *acc_image, bool force_split,
                                   const float *kdiag, const float *mu0, const void *y, const double *v,
                                   uint32_t sweep, double *G_out, double *g_out, double *f_out, double *omega_out,
                                   int64_t *n_out, uint32_t *nuni_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    agpl_lik_dev ld;
    int32_t rc = agpl_lik_to_device(ctx, lik, &ld);
    if (rc) return rc;
    const int L = ld.nlatent;
    if (N <= 0 || M <= 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes N=%lld M=%d", (long long)N, M);
    if (M % 128) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "M = %d must be a multiple of 128 (zero-pad the features)", M);
    // Phi == nullptr (a plan's pass): projection and accumulation both read the accumulate image
    if ((!Phi && !(acc_image && M % 256 == 0)) || !kdiag || !v || !G_out || !g_out)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    if (ld.kind != AGPL_LIK_BERNOULLI_LOGISTIC && !y) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null y");
    if (sweep & 0x80000000u) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "sweep must be < 2^31");
    const size_t slab = (agpl_slab_bytes(N, M, L) + 255) & ~(size_t)255;
    const size_t vec = (sizeof(float) * (size_t)L * N + 255) & ~(size_t)255;
    rc = agpl_ws_reserve(ctx, slab + 2 * vec);
    if (rc) return rc;
    rc = agpl_ws2_reserve(ctx, sizeof(double) * (1024 + 8));
    if (rc) return rc;
    int *bad = (int *)ctx->ws2;
    AGPL_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    char *base = (char *)ctx->ws;
    float *gam = (float *)(base + slab);
    float *bet = (float *)(base + slab + vec);
    // the slab region (>= 16 L N bytes) is free until the accumulation starts: it lends the N x L doubles of the
    // projections between the two kernels of the point pass
    rc = agpl_launch_gibbs_project_sample(ctx, ld, N, M, Phi, acc_image, kdiag, mu0, y, v, sweep, gam, bet, f_out, omega_out,
                                          n_out, nuni_out, bad, (double *)base);
    if (rc) return rc;
    const int keep_split = ctx->accumulate_split;
    if (force_split) ctx->accumulate_split = 1;
    rc = agpl_accumulate_impl(ctx, N, M, L, Phi, acc_image, bet, gam, G_out, g_out, base);
    ctx->accumulate_split = keep_split;
    if (rc) return rc;
    return agpl_sampler_outcome(ctx, ld.kind, bad);
}

extern "C" int32_t agpl_gibbs_pass(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M, const float *Phi,
                                   const float *kdiag, const float *mu0, const void *y, const double *v,
                                   uint32_t sweep, double *G_out, double *g_out, double *f_out, double *omega_out,
                                   int64_t *n_out, uint32_t *nuni_out) {
    return agpl_gibbs_pass_internal(ctx, lik, N, M, Phi, nullptr, false, kdiag, mu0, y, v, sweep, G_out, g_out, f_out, omega_out,
                           n_out, nuni_out);
}

namespace {
__global__ void add_vec_kernel(int n, const double *__restrict__ a, const double *__restrict__ b,
                               double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + (b ? b[i] : 0.0);
}
// m[b] = sum_{a >= b} U[a][b] vf[a],  v[b] = sum_{a >= b} U[a][b] (vf[a] + z[a])   with U[a][b] = A[b * M + a]:
// one wave per b reads its row of A contiguously; fixed-order butterfly
// (ld >= M: A [L, ld, ld] and vf [L, ld] are those of a zero-padded problem -- U is block diagonal, the rows a >= M contribute nothing
//  to the first M entries; z, v_out, m_out are M-sized)
__global__ __launch_bounds__(256) void factor_draw_kernel(int M, int ld, const double *__restrict__ A,
                                                          const double *__restrict__ vf, const double *__restrict__ z,
                                                          double *__restrict__ v_out, double *__restrict__ m_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), l = blockIdx.y;
    if (b >= M) return;
    const double *row = A + ((size_t)l * ld + b) * ld;
    const double *vl = vf + (size_t)l * ld, *zl = z + (size_t)l * M;
    double am = 0.0, av = 0.0;
    for (int a = b + lane; a < M; a += 64) {
        const double u = row[a], f = vl[a];
        am += u * f;
        av += u * (f + zl[a]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        am += __shfl_xor(am, o);
        av += __shfl_xor(av, o);
    }
    if (lane == 0) {
        v_out[(size_t)l * M + b] = av;
        if (m_out) m_out[(size_t)l * M + b] = am;
    }
}
} // namespace

extern "C" int32_t agpl_gibbs_draw_v(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g,
                                     const double *eta0, uint32_t sweep, double *v_out, double *m_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (M <= 0 || L <= 0 || !G || !g || !v_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    if (sweep & 0x80000000u) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "sweep must be < 2^31");
    // the feature count the hand-written factorisation runs on: M itself, or (round 6) M zero-padded to the next count it takes --
    // a multiple of 32 up to 512, of 128 up to 2048.  The draw z ~ N(0, I) stays M-sized (stream index l M + a, whatever the padding)
    const int32_t Mf = M <= 512 ? (M + 31) / 32 * 32 : (M <= 2048 ? (M + 127) / 128 * 128 : M);
    if (factor_one_launch(Mf, L)) {
        // I + G = C C', U = C^-1:  m = U'(U r),  v = m + C^-T z = U'(U r + z)   (one fused factor launch + one matvec)
        const size_t mat_bytes = sizeof(double) * (size_t)L * Mf * Mf;
        const size_t vec_bytes = (sizeof(double) * (size_t)L * Mf + 255) & ~(size_t)255;
        const size_t info_off = 16384;
        const size_t coop_bytes = (factor_work_bytes(Mf, L) + 255) & ~(size_t)255;
        const size_t pad_bytes = Mf != M ? mat_bytes + 2 * vec_bytes : 0;
        int32_t rc = agpl_ws2_reserve(ctx, info_off + 1024 + 2 * mat_bytes + 2 * vec_bytes + 512 + coop_bytes + pad_bytes);
        if (rc) return rc;
        int *info = (int *)((char *)ctx->ws2 + info_off);
        char *p = (char *)ctx->ws2 + info_off + 1024;
        double *T = (double *)p, *A = (double *)(p + mat_bytes);
        double *vf = (double *)(p + 2 * mat_bytes), *z = (double *)(p + 2 * mat_bytes + vec_bytes);
        void *coop = p + 2 * mat_bytes + 2 * vec_bytes + 512;
        if (Mf != M) {
            double *Gp = (double *)((char *)coop + coop_bytes), *gp = (double *)((char *)Gp + mat_bytes),
                   *ep = (double *)((char *)gp + vec_bytes);
            rc = agpl_pad_natural(ctx, L, M, Mf, G, g, eta0, nullptr, Gp, gp, eta0 ? ep : nullptr, nullptr);
            if (rc) return rc;
            G = Gp, g = gp, eta0 = eta0 ? ep : nullptr;
        }
        rc = factor_any(ctx, Mf, L, G, g, eta0, T, A, vf, nullptr, nullptr, info, coop);
        if (rc) return rc;
        rc = agpl_launch_randn(ctx, (int64_t)L * M, sweep | 0x80000000u, z);
        if (rc) return rc;
        dim3 gd((unsigned)agpl_cdiv(M, 4), (unsigned)L);
        factor_draw_kernel<<<gd, 256, 0, ctx->stream>>>(M, Mf, A, vf, z, v_out, m_out);
        AGPL_LAUNCH_CHECK(ctx);
        int hinfo[64];
        AGPL_HIP(ctx, hipMemcpyAsync(hinfo, info, sizeof(int) * L, hipMemcpyDeviceToHost, ctx->stream));
        AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < L; ++i)
            if (hinfo[i] != 0)
                AGPL_FAIL(ctx, hinfo[i] < 0 ? AGPL_ERR_HIP : AGPL_ERR_NOT_POSDEF,
                          hinfo[i] < 0 ? "factor kernel: a cooperating workgroup never arrived (latent %d, %d)"
                                       : "I + G is not positive definite (latent %d, pivot at row %d)",
                          i, (int)hinfo[i] - 1);
        return AGPL_OK;
    }
    rocblas_handle h;
    int32_t rc = get_handle(ctx, &h);
    if (rc) return rc;
    const size_t mat_bytes = sizeof(double) * (size_t)L * M * M;
    const size_t vec_bytes = (sizeof(double) * (size_t)L * M + 255) & ~(size_t)255;
    const size_t info_off = 16384;
    rc = agpl_ws2_reserve(ctx, info_off + 512 + mat_bytes + 2 * vec_bytes + 256);
    if (rc) return rc;
    rocblas_int *info = (rocblas_int *)((char *)ctx->ws2 + info_off);
    char *p = (char *)ctx->ws2 + info_off + 512;
    p = (char *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    double *A = (double *)p;
    double *mvec = (double *)(p + mat_bytes);
    double *z = (double *)(p + mat_bytes + vec_bytes);

    dim3 grid((unsigned)agpl_cdiv(M, 128), (unsigned)M, (unsigned)L);
    add_identity_kernel<<<grid, 128, 0, ctx->stream>>>(M, G, A);
    AGPL_LAUNCH_CHECK(ctx);
    const rocblas_stride stride = (rocblas_stride)M * M;
    AGPL_ROCBLAS(ctx, rocsolver_dpotrf_strided_batched(h, rocblas_fill_lower, M, A, M, stride, info, L));
    add_vec_kernel<<<(unsigned)agpl_cdiv((int64_t)L * M, 256), 256, 0, ctx->stream>>>(L * M, g, eta0, mvec);
    AGPL_LAUNCH_CHECK(ctx);
    rc = agpl_launch_randn(ctx, (int64_t)L * M, sweep | 0x80000000u, z);
    if (rc) return rc;
    for (int l = 0; l < L; ++l) {
        double *Al = A + (size_t)l * M * M;
        // m = (C C')^-1 (g + eta0)
        AGPL_ROCBLAS(ctx, rocsolver_dpotrs(h, rocblas_fill_lower, M, 1, Al, M, mvec + (size_t)l * M, M));
        // x = C^-T z  (in place)
        AGPL_ROCBLAS(ctx, rocblas_dtrsv(h, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                                        M, Al, M, z + (size_t)l * M, 1));
    }
    add_vec_kernel<<<(unsigned)agpl_cdiv((int64_t)L * M, 256), 256, 0, ctx->stream>>>(L * M, mvec, z, v_out);
    AGPL_LAUNCH_CHECK(ctx);
    if (m_out)
        AGPL_HIP(ctx, hipMemcpyAsync(m_out, mvec, sizeof(double) * (size_t)L * M, hipMemcpyDeviceToDevice, ctx->stream));
    rocblas_int hinfo[64];
    const int ni = L > 64 ? 64 : L;
    AGPL_HIP(ctx, hipMemcpyAsync(hinfo, info, sizeof(rocblas_int) * ni, hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < ni; ++i)
        if (hinfo[i] != 0)
            AGPL_FAIL(ctx, AGPL_ERR_NOT_POSDEF, "I + G is not positive definite (latent %d, potrf info = %d)", i,
                      (int)hinfo[i]);
    return AGPL_OK;
}

// the plan's CAVI pass (agpl_plan.hip).  image_scale_exp: the marginal images hold 2^e Phi (times the U images' 2^15).  elbo_terms_out (device, may be
// null; image path only): sum over the points of expected_logtilt_i - aux_kldivergence_i for the q(v) this pass used.
int32_t agpl_cavi_pass_factor_internal(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, int32_t M, const float *Phi,
                                       const void *Phi_hi, const void *Phi_lo, const void *acc_image, const float *resid,
                                       const float *mu0, const void *y, const void *U_hi, const void *U_lo, const float *v,
                                       double *G_out, double *g_out, float *c_out, float *gamma_out, float *beta_out,
                                       int image_scale_exp, double *elbo_terms_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    agpl_lik_dev ld;
    int32_t rc = agpl_lik_to_device(ctx, lik, &ld);
    if (rc) return rc;
    const int L = ld.nlatent;
    if (N <= 0 || M <= 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes N=%lld M=%d", (long long)N, M);
    if ((!Phi && !acc_image) || !Phi_hi || !Phi_lo || !resid || !y || !U_hi || !U_lo || !v || !G_out || !g_out)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    const size_t slab = (agpl_slab_bytes(N, M, L) + 255) & ~(size_t)255;
    const size_t vec = (sizeof(float) * (size_t)L * N + 255) & ~(size_t)255;
    rc = agpl_ws_reserve(ctx, slab + 4 * vec);
    if (rc) return rc;
    char *base = (char *)ctx->ws;
    // a split entry point implies the split-float16 accumulation (ctx->accumulate_split is internal: the float32-named entry points leave it 0)
    const int keep = ctx->accumulate_split;
    (void)slab;
    (void)vec;
    if (acc_image && M % 256 == 0) {
        // three launches up to the slabs: marginal partial sums (MFMA) -> ONE per-point kernel (q(f_i), aux_posterior!,
        // expected potential / precision, written as the accumulation's gamma | beta records, and max gamma) -> accumulation.
        // Neither mu / var nor (unless the caller asks for them) gamma / beta exist as arrays.
        float *gb, *qpart, *mpart;
        unsigned *scal, *queues;
        agpl_accumulate_records(N, M, L, base, &gb, &scal);
        rc = agpl_timing_begin(ctx, 0);
        if (rc) return rc;
        rc = agpl_marginals_factor_parts(ctx, N, M, L, Phi_hi, Phi_lo, U_hi, U_lo, v, scal, &qpart, &mpart, &queues,
                                         image_scale_exp);
        if (rc) return rc;
        rc = agpl_timing_end(ctx, 0);
        if (rc) return rc;
        if ((char *)gb < (char *)mpart + sizeof(float) * (size_t)(M / 256) * L * N)
            AGPL_FAIL(ctx, AGPL_ERR_HIP, "workspace layout: the records overlap the marginal partial sums");
        const int64_t Npad = ((N + 31) & ~(int64_t)31) + 32;
        rc = agpl_launch_fused_point(ctx, ld, N, Npad, M / 256, y, resid, mu0, qpart, mpart, gamma_out, beta_out, c_out, gb,
                                     scal, queues, elbo_terms_out);
        if (rc) return rc;
        ctx->accumulate_split = 1;
        rc = agpl_accumulate_impl(ctx, N, M, L, nullptr, acc_image, nullptr, nullptr, G_out, g_out, base, true);
        ctx->accumulate_split = keep;
        if (rc) return rc;
        return agpl_pending_resolve(ctx);
    }
    AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "the plan sweep needs the accumulate image and M %% 256 == 0");
}

