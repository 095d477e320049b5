// agpl_dense.hip -- the full-rank Gibbs step the reference actually executes (`gibbs_sample`,
// examples/bernoulli/script.jl:76-87; examples/studentt/script.jl is the same loop), for N points with a dense
// N x N prior covariance K (BASELINE config C5: StudentT, N = 65 536).
//
//   Omega <- aux_sample!(lik, y, f)                                              script.jl:81
//   Sigma  = inv(Symmetric(inv(K) + Diagonal(gamma)))                            script.jl:82
//   mu     = Sigma * (beta + K \ mu0)                                            script.jl:83
//   f      ~ MvNormal(mu, Sigma)                                                 script.jl:84
//
// evaluated without any inverse (the reference forms two O(N^3) inverses and a Cholesky per sweep):
//   B = I + D^1/2 K D^1/2 = C C'      (D = Diag(gamma); ONE float64 Cholesky per sweep: rocSOLVER potrf)
//   f = f0 + K D^1/2 B^-1 (D^-1/2 beta - D^1/2 f0 - z2),   f0 = mu0 + L_K z1,  z1, z2 ~ N(0, I)
// which is an exact draw from N(mu, Sigma) (Matheron's rule with pseudo-observations yhat = D^-1 beta of noise
// variance D^-1).  The N^3 / 3 of the Cholesky is the hand-written float64-MFMA trailing update below; the 2048-wide
// diagonal blocks, the panel solves and the matrix-vector products are rocSOLVER / rocBLAS calls; the sampler (agpl_ops.hip)
// and the fused elementwise steps are hand-written.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "agpl_common.h"

int32_t agpl_launch_randn(agpl_ctx *ctx, int64_t n, uint32_t sweep, double *out);
int32_t agpl_get_rocblas(agpl_ctx *ctx, void **handle_out);

namespace {

#define AGPL_ROCBLAS(ctx, call)                                                                     \
    do {                                                                                            \
        rocblas_status s__ = (call);                                                                \
        if (s__ != rocblas_status_success)                                                          \
            AGPL_FAIL(ctx, AGPL_ERR_HIP, "%s failed: rocblas_status %d (%s:%d)", #call, (int)s__,   \
                      __FILE__, __LINE__);                                                          \
    } while (0)

// B[i][j] = (i == j) + sqrt(gamma_i) K[i][j] sqrt(gamma_j) : one streaming pass, 16 B per lane
__global__ __launch_bounds__(256) void build_b_kernel(int64_t N, const double *__restrict__ K,
                                                      const double *__restrict__ gamma, double *__restrict__ B) {
    const int64_t row = blockIdx.y;
    const double sr = sqrt(gamma[row]);
    const double *Kr = K + row * N;
    double *Br = B + row * N;
    for (int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; c < N;
         c += (int64_t)gridDim.x * blockDim.x * 2) {
        if (c + 1 < N) {
            const double2 k = *reinterpret_cast<const double2 *>(Kr + c);
            double2 o;
            o.x = sr * k.x * sqrt(gamma[c]) + (c == row ? 1.0 : 0.0);
            o.y = sr * k.y * sqrt(gamma[c + 1]) + (c + 1 == row ? 1.0 : 0.0);
            *reinterpret_cast<double2 *>(Br + c) = o;
        } else {
            Br[c] = sr * Kr[c] * sqrt(gamma[c]) + (c == row ? 1.0 : 0.0);
        }
    }
}

// f0 = mu0 + (L_K z1) ; r = beta / sqrt(gamma) - sqrt(gamma) f0 - z2.  gamma_i = 0 happens (Poisson: y_i = 0 and a
// drawn n_i = 0 give omega_i = PG(0, c) = 0) and implies beta_i = 0: row i of B is then e_i and D^1/2 zeroes that
// component of the solve on output, so beta / sqrt(gamma) := 0 there is exact, not a patch.
__global__ void prep_rhs_kernel(int64_t N, const double *__restrict__ mu0, const double *__restrict__ lz,
                                const double *__restrict__ beta, const double *__restrict__ gamma,
                                const double *__restrict__ z2, double *__restrict__ f0, double *__restrict__ r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double f = lz[i] + (mu0 ? mu0[i] : 0.0);
    const double sg = sqrt(gamma[i]);
    f0[i] = f;
    r[i] = (sg > 0.0 ? beta[i] / sg : 0.0) - sg * f - z2[i];
}
// t = sqrt(gamma) .* s
__global__ void scale_kernel(int64_t N, const double *__restrict__ gamma, double *__restrict__ s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) s[i] *= sqrt(gamma[i]);
}
__global__ void copy_kernel(int64_t n, const double *__restrict__ a, double *__restrict__ b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        b[i] = a[i];
}

// or_info: info[0] |= info[1] (keeps the first failing block's flag across the blocked factorisation)
__global__ void or_info_kernel(rocblas_int *info, int block_start) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && info[1] != 0 && info[0] == 0) info[0] = block_start + info[1];
}

// ------------------------------------------------------------------------------------------------
// trailing_update_kernel: the N^3 of the factorisation, hand-written on the float64 matrix cores.
//   A[r][c] -= sum_{q < w} A[r][k0 + q] A[c][k0 + q]   for e <= c <= r < N   (column-major, ld = N; e = k0 + w)
// i.e. the symmetric rank-w update of the LOWER triangle of the trailing matrix by the panel P = A[e:, k0:e] that the
// triangular solve has just produced -- only the lower triangle is computed (tiles with row block >= column block).
// One workgroup = one 128 x 128 tile, four waves as 2 x 2 of 64 x 64 = 4 x 4 accumulators of v_mfma_f64_16x16x4_f64 (128
// VGPRs).  Both operands are rows of P ([panel row][q], the row index contiguous in memory): 16 columns of P at a time go
// through a double-buffered LDS tile [q][row] (row pitch 144 doubles: the four q-groups of a fragment read fall in distinct
// bank halves), the next tile's global loads in flight during the 64 MFMAs of the current one.  An MFMA of this shape takes
// 64 cycles for 2 x 8 bytes of operands per lane: the kernel is bound by the matrix pipe, not by LDS or HBM
// (16 KB + 16 KB per 4096 MFMA-cycles and workgroup).  The product is formed transposed, D[c][r], so that the 16 lanes of a
// result register run along the contiguous (row) index of A: 128-byte read-modify-writes.  The fixed summation order makes
// the factor bitwise reproducible.
// ------------------------------------------------------------------------------------------------
typedef double d4t __attribute__((ext_vector_type(4)));
constexpr int kTT = 128;      // tile edge
constexpr int kTK = 16;       // panel columns per stage
constexpr int kTPitch = 144;  // doubles per LDS row of a stage
constexpr int kTStage = kTK * kTPitch; // doubles per operand and stage

template <bool DIAG>
__device__ __forceinline__ void trailing_tile(double *smem, int64_t N, double *__restrict__ A, int64_t k0, int w, int64_t r0,
                                              int64_t c0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    const int li = lane & 15, lq = lane >> 4;
    // staging role: rows (tid & 63) * 2, +1 of the tile; panel columns (tid >> 6) + 4 i
    const int srow = (tid & 63) * 2, sq = tid >> 6;
    int64_t gr = r0 + srow, gc = c0 + srow;
    if (gr > N - 2) gr = N - 2; // (rows beyond N are never stored: any in-range address will do)
    if (gc > N - 2) gc = N - 2;
    const double *pr = A + gr + (k0 + sq) * N, *pc = A + gc + (k0 + sq) * N;
    double *sR = smem, *sC = smem + 2 * kTStage; // [stage][q][row]
    double2 vr0, vr1, vr2, vr3, vc0, vc1, vc2, vc3; // the next stage, in flight during the MFMAs of the current one
#define AGPL_T_GLOAD(kt_)                                                                                       \
    do {                                                                                                        \
        const double *qr_ = pr + (int64_t)(kt_) * kTK * N, *qc_ = pc + (int64_t)(kt_) * kTK * N;                \
        vr0 = *reinterpret_cast<const double2 *>(qr_);                                                          \
        vr1 = *reinterpret_cast<const double2 *>(qr_ + 4 * N);                                                  \
        vr2 = *reinterpret_cast<const double2 *>(qr_ + 8 * N);                                                  \
        vr3 = *reinterpret_cast<const double2 *>(qr_ + 12 * N);                                                 \
        if (!DIAG) {                                                                                            \
            vc0 = *reinterpret_cast<const double2 *>(qc_);                                                      \
            vc1 = *reinterpret_cast<const double2 *>(qc_ + 4 * N);                                              \
            vc2 = *reinterpret_cast<const double2 *>(qc_ + 8 * N);                                              \
            vc3 = *reinterpret_cast<const double2 *>(qc_ + 12 * N);                                             \
        }                                                                                                       \
    } while (0)
#define AGPL_T_SWRITE(st_)                                                                                      \
    do {                                                                                                        \
        double *dr_ = sR + (st_) * kTStage + sq * kTPitch + srow, *dc_ = sC + (st_) * kTStage + sq * kTPitch + srow; \
        *reinterpret_cast<double2 *>(dr_) = vr0;                                                                \
        *reinterpret_cast<double2 *>(dr_ + 4 * kTPitch) = vr1;                                                  \
        *reinterpret_cast<double2 *>(dr_ + 8 * kTPitch) = vr2;                                                  \
        *reinterpret_cast<double2 *>(dr_ + 12 * kTPitch) = vr3;                                                 \
        if (!DIAG) {                                                                                            \
            *reinterpret_cast<double2 *>(dc_) = vc0;                                                            \
            *reinterpret_cast<double2 *>(dc_ + 4 * kTPitch) = vc1;                                              \
            *reinterpret_cast<double2 *>(dc_ + 8 * kTPitch) = vc2;                                              \
            *reinterpret_cast<double2 *>(dc_ + 12 * kTPitch) = vc3;                                             \
        }                                                                                                       \
    } while (0)
    d4t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = d4t{0.0, 0.0, 0.0, 0.0};
    const int nkt = w / kTK;
    AGPL_T_GLOAD(0);
    AGPL_T_SWRITE(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        AGPL_T_GLOAD(kt + 1 < nkt ? kt + 1 : kt); // (unconditional: the last iteration re-reads its own stage, unused)
        const double *bR = sR + st * kTStage + lq * kTPitch + wr * 64 + li;           // B operand: rows of the row block
        const double *bC = (DIAG ? sR : sC) + st * kTStage + lq * kTPitch + wc * 64 + li; // A operand: rows of the column block
#pragma unroll
        for (int ks = 0; ks < kTK / 4; ++ks) {
            double a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = bC[ks * 4 * kTPitch + 16 * m];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = bR[ks * 4 * kTPitch + 16 * n];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        AGPL_T_SWRITE(st ^ 1);
        __syncthreads();
    }
#undef AGPL_T_GLOAD
#undef AGPL_T_SWRITE
    // A[r][c] -= D[c][r]: register rr of lane l holds D[16 m + 4 rr + lq][16 n + li]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t c = c0 + wc * 64 + 16 * m + 4 * rr + lq;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int64_t r = r0 + wr * 64 + 16 * n + li;
                if (r < N && c < N && (!DIAG || r >= c)) A[r + c * N] -= acc[m][n][rr];
            }
        }
}

__global__ __launch_bounds__(256, 2) void trailing_update_kernel(int64_t N, double *__restrict__ A, int64_t k0, int w, int nt) {
    extern __shared__ __attribute__((aligned(16))) double tsm[];
    // tile (I, J), J <= I, of the trailing matrix, row by row of the lower block triangle
    const int64_t p = blockIdx.x;
    int64_t I = (int64_t)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= p) ++I;
    while (I * (I + 1) / 2 > p) --I;
    const int64_t J = p - I * (I + 1) / 2;
    const int64_t e = k0 + w;
    if (I == J) trailing_tile<true>(tsm, N, A, k0, w, e + I * kTT, e + J * kTT);
    else trailing_tile<false>(tsm, N, A, k0, w, e + I * kTT, e + J * kTT);
}

// Lower Cholesky (column-major view, in place) of an N x N float64 matrix as a right-looking blocked factorisation:
// rocsolver_dpotrf on the 2048-wide diagonal blocks, one rocblas dtrsm per panel (together ~12 % of the flops at N = 65536),
// and the trailing lower triangle -- the N^3 / 3 -- by trailing_update_kernel above (rounds 1-2: one rocblas dgemm per block
// column, 54.7 TF/s at C5).  Round 3, N = 65536: the kernel runs at 63.4 TF/s = 0.96 of the probe's 65.9 TF/s
// (profiles/r03_c5_kernel_stats.csv: 1404 ms of a 1741 ms step); what remains is the chain of small rocSOLVER / rocBLAS
// kernels of the diagonal blocks and panel solves, serialised with the updates.
// info[0] = 0 or 1-based index of the first non-positive pivot, as potrf.
#ifndef AGPL_DENSE_NB
#define AGPL_DENSE_NB 2048 // measured at C5 on one box (profiles/r03_c5_block_width.txt): 512 / 1024 / 2048 / 4096 -> 1876 / 1763 / 1741 / 1743 ms per step
#endif
int32_t blocked_potrf(agpl_ctx *ctx, rocblas_handle h, int64_t N, double *A, rocblas_int *info) {
    constexpr int64_t nb = AGPL_DENSE_NB;
    static_assert(nb % kTK == 0, "panel width must be whole stages");
    AGPL_HIP(ctx, hipMemsetAsync(info, 0, 2 * sizeof(rocblas_int), ctx->stream));
    AGPL_ROCBLAS(ctx, rocblas_set_pointer_mode(h, rocblas_pointer_mode_host));
    const double one = 1.0;
    const size_t lds = sizeof(double) * 4 * kTStage;
    AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&trailing_update_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int64_t k = 0; k < N; k += nb) {
        const int64_t e = k + nb < N ? k + nb : N, w = e - k;
        double *Akk = A + k + k * N;
        AGPL_ROCBLAS(ctx, rocsolver_dpotrf(h, rocblas_fill_lower, (rocblas_int)w, Akk, (rocblas_int)N, info + 1));
        or_info_kernel<<<1, 64, 0, ctx->stream>>>(info, (int)k);
        AGPL_LAUNCH_CHECK(ctx);
        if (e == N) break;
        const int64_t m = N - e;
        double *A21 = A + e + k * N;
        // A21 <- A21 L11^-T
        AGPL_ROCBLAS(ctx, rocblas_dtrsm(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose,
                                        rocblas_diagonal_non_unit, (rocblas_int)m, (rocblas_int)w, &one, Akk,
                                        (rocblas_int)N, A21, (rocblas_int)N));
        // trailing lower triangle: A[e:, e:] -= A21 A21'
        const int64_t nt = (m + kTT - 1) / kTT;
        trailing_update_kernel<<<(unsigned)(nt * (nt + 1) / 2), 256, lds, ctx->stream>>>(N, A, k, (int)w, (int)nt);
        AGPL_LAUNCH_CHECK(ctx);
    }
    return AGPL_OK;
}

} // namespace

namespace {
// x <- (L L')^-1 x for one right-hand side, L the lower Cholesky factor (column-major, ld N) -- rocsolver_dpotrs /
// rocblas_dtrsv walk a single dependent chain down the whole matrix (188 ms per triangular solve at N = 65 536:
// 90 GB/s); here only the 2048-wide diagonal blocks are solved that way and everything off the diagonal is a dgemv at
// HBM speed (two sweeps over the 17 GB triangle).
int32_t blocked_potrs_vec(agpl_ctx *ctx, rocblas_handle h, int64_t N, const double *L, double *x) {
    constexpr int64_t nb = 2048;
    AGPL_ROCBLAS(ctx, rocblas_set_pointer_mode(h, rocblas_pointer_mode_host));
    const double one = 1.0, mone = -1.0;
    // forward: L y = x
    for (int64_t k = 0; k < N; k += nb) {
        const int64_t e = k + nb < N ? k + nb : N, w = e - k;
        AGPL_ROCBLAS(ctx, rocblas_dtrsv(h, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                                        (rocblas_int)w, L + k + k * N, (rocblas_int)N, x + k, 1));
        if (e < N)
            AGPL_ROCBLAS(ctx, rocblas_dgemv(h, rocblas_operation_none, (rocblas_int)(N - e), (rocblas_int)w, &mone,
                                            L + e + k * N, (rocblas_int)N, x + k, 1, &one, x + e, 1));
    }
    // backward: L' z = y
    const int64_t last = ((N - 1) / nb) * nb;
    for (int64_t k = last; k >= 0; k -= nb) {
        const int64_t e = k + nb < N ? k + nb : N, w = e - k;
        AGPL_ROCBLAS(ctx, rocblas_dtrsv(h, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                                        (rocblas_int)w, L + k + k * N, (rocblas_int)N, x + k, 1));
        if (k > 0)
            AGPL_ROCBLAS(ctx, rocblas_dgemv(h, rocblas_operation_transpose, (rocblas_int)w, (rocblas_int)k, &mone,
                                            L + k, (rocblas_int)N, x + k, 1, &one, x, 1));
    }
    return AGPL_OK;
}
} // namespace

extern "C" int32_t agpl_dense_cholesky(agpl_ctx *ctx, int64_t N, const double *A, double *L_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (N <= 0 || N > 0x7fffffff || !A || !L_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    void *hv;
    int32_t rc = agpl_get_rocblas(ctx, &hv);
    if (rc) return rc;
    rocblas_handle h = (rocblas_handle)hv;
    if (A != L_out) {
        copy_kernel<<<4096, 256, 0, ctx->stream>>>(N * N, A, L_out);
        AGPL_LAUNCH_CHECK(ctx);
    }
    rc = agpl_ws2_reserve(ctx, 32768);
    if (rc) return rc;
    rocblas_int *info = (rocblas_int *)((char *)ctx->ws2 + 16384);
    if (N >= 8192) {
        rc = blocked_potrf(ctx, h, N, L_out, info);
        if (rc) return rc;
    } else {
        AGPL_ROCBLAS(ctx, rocsolver_dpotrf(h, rocblas_fill_lower, (rocblas_int)N, L_out, (rocblas_int)N, info));
    }
    rocblas_int hinfo = 0;
    AGPL_HIP(ctx, hipMemcpyAsync(&hinfo, info, sizeof(hinfo), hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hinfo != 0) AGPL_FAIL(ctx, AGPL_ERR_NOT_POSDEF, "matrix is not positive definite (potrf info = %d)", (int)hinfo);
    return AGPL_OK;
}

extern "C" int32_t agpl_dense_gibbs_step(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t N, const double *K,
                                         const double *Lk, const double *mu0, const void *y, double *f_inout,
                                         double *B_work, uint32_t sweep, double *omega_out, int64_t *n_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (!lik) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null likelihood descriptor");
    if (lik->nlatent != 1)
        AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED, "the dense Gibbs step handles single-latent likelihoods (nlatent = %d)",
                  lik->nlatent);
    if (N <= 0 || N > 0x7fffffff || !K || !Lk || !f_inout || !B_work || !omega_out)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    if (sweep & 0x80000000u) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "sweep must be < 2^31");
    void *hv;
    int32_t rc = agpl_get_rocblas(ctx, &hv);
    if (rc) return rc;
    rocblas_handle h = (rocblas_handle)hv;
    // scratch: beta, gamma, z (2N), f0, r
    rc = agpl_ws_reserve(ctx, sizeof(double) * 6 * (size_t)N + 1024);
    if (rc) return rc;
    double *beta = (double *)ctx->ws, *gamma = beta + N, *z = gamma + N, *f0 = z + 2 * N, *r = f0 + N;

    // 1. Omega <- aux_sample!(lik, y, f) ; beta, gamma = auglik_potential / auglik_precision     script.jl:81-83
    rc = agpl_aux_sample(ctx, lik, N, y, f_inout, omega_out, n_out, sweep, nullptr, nullptr);
    if (rc) return rc;
    rc = agpl_potential_precision(ctx, lik, N, y, omega_out, n_out, nullptr, beta, gamma);
    if (rc) return rc;
    // 2. z1 | z2 from the streams (seed, 0..2N-1, sweep | 2^31)
    rc = agpl_launch_randn(ctx, 2 * N, sweep | 0x80000000u, z);
    if (rc) return rc;
    // 3. f0 = mu0 + L_K z1   (in place on z1)
    AGPL_ROCBLAS(ctx, rocblas_dtrmv(h, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit,
                                    (rocblas_int)N, Lk, (rocblas_int)N, z, 1));
    const unsigned nb = (unsigned)agpl_cdiv(N, 256);
    prep_rhs_kernel<<<nb, 256, 0, ctx->stream>>>(N, mu0, z, beta, gamma, z + N, f0, r);
    AGPL_LAUNCH_CHECK(ctx);
    // 4. B = I + D^1/2 K D^1/2, Cholesky, solve
    dim3 gb((unsigned)(agpl_cdiv(N, 512) < 64 ? agpl_cdiv(N, 512) : 64), (unsigned)N);
    build_b_kernel<<<gb, 256, 0, ctx->stream>>>(N, K, gamma, B_work);
    AGPL_LAUNCH_CHECK(ctx);
    rc = agpl_ws2_reserve(ctx, 32768);
    if (rc) return rc;
    rocblas_int *info = (rocblas_int *)((char *)ctx->ws2 + 16384);
    if (N >= 8192) {
        rc = blocked_potrf(ctx, h, N, B_work, info);
        if (rc) return rc;
    } else {
        AGPL_ROCBLAS(ctx, rocsolver_dpotrf(h, rocblas_fill_lower, (rocblas_int)N, B_work, (rocblas_int)N, info));
    }
    if (N >= 8192) {
        rc = blocked_potrs_vec(ctx, h, N, B_work, r);
        if (rc) return rc;
    } else {
        AGPL_ROCBLAS(ctx, rocsolver_dpotrs(h, rocblas_fill_lower, (rocblas_int)N, 1, B_work, (rocblas_int)N, r,
                                           (rocblas_int)N));
    }
    // 5. f = f0 + K (D^1/2 s)
    scale_kernel<<<nb, 256, 0, ctx->stream>>>(N, gamma, r);
    AGPL_LAUNCH_CHECK(ctx);
    const double one = 1.0;
    AGPL_ROCBLAS(ctx, rocblas_set_pointer_mode(h, rocblas_pointer_mode_host));
    AGPL_ROCBLAS(ctx, rocblas_dsymv(h, rocblas_fill_lower, (rocblas_int)N, &one, K, (rocblas_int)N, r, 1, &one, f0, 1));
    copy_kernel<<<nb, 256, 0, ctx->stream>>>(N, f0, f_inout);
    AGPL_LAUNCH_CHECK(ctx);
    rocblas_int hinfo = 0;
    AGPL_HIP(ctx, hipMemcpyAsync(&hinfo, info, sizeof(hinfo), hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hinfo != 0)
        AGPL_FAIL(ctx, AGPL_ERR_NOT_POSDEF, "I + D^1/2 K D^1/2 is not positive definite (potrf info = %d)", (int)hinfo);
    return AGPL_OK;
}

// ------------------------------------------------------------------------------------------------
// agpl_probe_mfma_f64: the float64 matrix rate this device sustains (the local hardware guide gives no FP64 MFMA
// peak: "measure, don't assume") -- v_mfma_f64_16x16x4_f64 issued back to back on four independent
// accumulators by one wave per SIMD of every CU, on non-trivial operands; 2 * 16 * 16 * 4 flop per instruction.
// ------------------------------------------------------------------------------------------------
namespace {
typedef double d4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_f64_probe_kernel(int iters, double *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + 1e-3 * lane, b = 1.0 - 1e-3 * lane;
    d4v c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
        }
        a = -a; // keeps the sums bounded and the operands changing
    }
    const d4v s = c0 + c1 + c2 + c3;
    if (s[0] + s[1] + s[2] + s[3] == 1.2345e300) sink[0] = s[0]; // never true: keeps the chain alive
}
} // namespace

extern "C" int32_t agpl_probe_mfma_f64(agpl_ctx *ctx, int32_t iters, double *tflops_host) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (iters <= 0 || !tflops_host) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    int32_t rc = agpl_ws2_reserve(ctx, 4096);
    if (rc) return rc;
    hipDeviceProp_t prop;
    AGPL_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    hipEvent_t e0, e1;
    AGPL_HIP(ctx, hipEventCreate(&e0));
    AGPL_HIP(ctx, hipEventCreate(&e1));
    double best_tf = 0.0;
    // 1, 2 and 4 waves per SIMD (4-wave workgroups, 1 / 2 / 4 per CU): the rate a wave alone cannot reach because of the
    // instruction's dependent-issue latency shows up with more waves; the best of the three is the device's rate
    for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {
        const int blocks = prop.multiProcessorCount * per_cu;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) { // first launch warms the clocks; the fastest of the rest counts
            AGPL_HIP(ctx, hipEventRecord(e0, ctx->stream));
            mfma_f64_probe_kernel<<<blocks, 256, 0, ctx->stream>>>(iters, (double *)ctx->ws2);
            AGPL_HIP(ctx, hipEventRecord(e1, ctx->stream));
            AGPL_HIP(ctx, hipEventSynchronize(e1));
            float ms = 0.f;
            AGPL_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double flop = (double)blocks * 4.0 * (double)iters * 16.0 * (2.0 * 16 * 16 * 4);
        const double tf = flop / ((double)best * 1e-3) / 1e12;
        if (tf > best_tf) best_tf = tf;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    AGPL_LAUNCH_CHECK(ctx);
    *tflops_host = best_tf;
    return AGPL_OK;
}
