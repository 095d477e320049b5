// agpl_mfma.hip -- the two O(N M^2) contractions of a sparse sweep on the gfx950 matrix cores.
//
// Feature matrix Phi: float32 [M, N] column-major (one point = M contiguous floats), M % 128 == 0.
//
//   marginal_kernel  (a11)  T = W' Phi_tile  (f32 MFMA 32x32x2),  q_i = sum_a Phi[a,i] T[a,i],
//                           mu_i = alpha' phi_i ; var_i = kdiag_i - q_i.
//                           W' = upper-triangular with doubled off-diagonal, so only the block pairs
//                           cb >= rb are visited: (nb+1)/(2 nb) of a full GEMM.
//   syrk_kernel      (a12)  G = Phi Diag(gamma) Phi' on 128 x 128 output tiles of the lower triangle,
//                           N split over workgroups (split-K), g = Phi beta on the diagonal tiles.
//   reduce kernels          fixed-order float64 sum of the per-split slabs.
//
// Both MFMA kernels: 256 threads = 4 waves, each wave a 64 x 64 sub-tile = 2 x 2 accumulators of
// v_mfma_f32_32x32x2_f32 (64 accumulator VGPRs); operand tiles are staged global -> registers -> LDS
// with one barrier per 32-deep k-slice and the next slice's global loads in flight during the MFMAs;
// 2 workgroups per CU (<= 74 KB LDS each) so one workgroup's barrier hides behind the other's MFMAs.
// LDS images: [k][row] with the row index contiguous (conflict-free ds_read_b32: lanes 0-31 read
// consecutive rows at k, lanes 32-63 at k+1), except the Phi operand of marginal_kernel whose k index
// is the contiguous one in memory: image [point][33] (odd pitch -> conflict-free).
#include "agpl_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BS = 128; // block of feature rows
constexpr int KT = 32;  // k-slice per stage
constexpr int NT = 128; // points per marginal tile
constexpr int KPITCH = KT + 1;
constexpr int HPITCH = BS + 4;
constexpr int kFlushStages = 64; // f32 accumulation run: 64 stages x 32 points = 2048 points

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// marginal / transform kernel
//   MODE 0: quadratic form + mean (outputs mu, var);  P = Wpack (block pairs cb >= rb only)
//   MODE 1: transform out[:, i] = A in[:, i];         P = A' (all block pairs), writes float4 per lane
// grid = (tiles of 128 points, L)
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256, 2) void marginal_kernel(int64_t N, int M, const float *__restrict__ Phi,
                                                          const float *__restrict__ kdiag,
                                                          const float *__restrict__ mu0,
                                                          const float *__restrict__ Pall,
                                                          const float *__restrict__ alpha_all,
                                                          float *__restrict__ mu_out, float *__restrict__ var_out,
                                                          float *__restrict__ t_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // region 0: staging (2 x (Pt 4096 + Kt 4224)) aliased with the Hadamard image (128 x 132)
    float *stage0 = smem;
    constexpr int kStageFloats = KT * BS + NT * KPITCH; // 8320
    constexpr int kRegion0 = (NT * HPITCH > 2 * kStageFloats) ? NT * HPITCH : 2 * kStageFloats;
    float *alpha_s = smem + kRegion0;          // M floats
    float *qred = alpha_s + M;                 // 2 x 128
    float *mred = qred + 2 * NT;               // 2 x 128

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    const int l = blockIdx.y;
    const int nb = M / BS;
    const int64_t n0 = (int64_t)blockIdx.x * NT;
    const float *P = Pall + (int64_t)l * M * M;

    if (MODE == 0) {
        const float *alpha = alpha_all + (int64_t)l * M;
        for (int a = tid; a < M; a += 256) alpha_s[a] = alpha[a];
    }

    // per-thread staging coordinates
    // Pt: q = tid + 256 j -> kk = q >> 5, a4 = q & 31
    // Kt: q = tid + 256 j -> n = q >> 3, k4 = q & 7
    int64_t krow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int64_t n = n0 + ((tid + 256 * j) >> 3);
        if (n > N - 1) n = N - 1;
        krow[j] = n * (int64_t)M + ((tid & 7) << 2);
    }

    float qacc[2] = {0.f, 0.f};
    float macc = 0.f;

    float4 pr[4], kr[4];
    auto load_stage = [&](int rb, int cb, int ks) {
        const int b0 = cb * BS + ks * KT;
        const float *psrc = P + (int64_t)b0 * M + rb * BS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int q = tid + 256 * j;
            pr[j] = *reinterpret_cast<const float4 *>(psrc + (int64_t)(q >> 5) * M + ((q & 31) << 2));
            kr[j] = *reinterpret_cast<const float4 *>(Phi + krow[j] + b0);
        }
    };
    auto store_stage = [&](int buf) {
        float *Pt = stage0 + buf * kStageFloats;
        float *Kt = Pt + KT * BS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int q = tid + 256 * j;
            *reinterpret_cast<float4 *>(Pt + (q >> 5) * BS + ((q & 31) << 2)) = pr[j];
            float *kd = Kt + (q >> 3) * KPITCH + ((q & 7) << 2);
            kd[0] = kr[j].x;
            kd[1] = kr[j].y;
            kd[2] = kr[j].z;
            kd[3] = kr[j].w;
        }
    };

    for (int rb = 0; rb < nb; ++rb) {
        f32x16 acc[2][2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ii][jj][r] = 0.f;

        const int cb_first = (MODE == 0) ? rb : 0;
        const int nstage = (nb - cb_first) * (BS / KT);
        load_stage(rb, cb_first, 0);
        __syncthreads(); // previous rb's Hadamard reads of region 0 are done
        store_stage(0);
        __syncthreads();
        for (int s = 0; s < nstage; ++s) {
            const int buf = s & 1;
            if (s + 1 < nstage) load_stage(rb, cb_first + ((s + 1) >> 2), (s + 1) & 3);
            const float *Pt = stage0 + buf * kStageFloats;
            const float *Kt = Pt + KT * BS;
            const float *pa = Pt + lk * BS + wr * 64 + li;
            const float *pb = Kt + (wc * 64 + li) * KPITCH + lk;
#pragma unroll
            for (int k0 = 0; k0 < KT; k0 += 2) {
                float a0 = pa[k0 * BS], a1 = pa[k0 * BS + 32];
                float b0 = pb[k0], b1 = pb[k0 + 32 * KPITCH];
                acc[0][0] = mfma(a0, b0, acc[0][0]);
                acc[0][1] = mfma(a0, b1, acc[0][1]);
                acc[1][0] = mfma(a1, b0, acc[1][0]);
                acc[1][1] = mfma(a1, b1, acc[1][1]);
            }
            if (MODE == 0 && rb == 0) { // mean: every feature row passes through exactly once when rb == 0
                const int bbase = (s >> 2) * BS + (s & 3) * KT + (tid >> 7) * 16;
                const float *kp = Kt + (tid & 127) * KPITCH + (tid >> 7) * 16;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) macc += alpha_s[bbase + kk] * kp[kk];
            }
            if (s + 1 < nstage) store_stage(buf ^ 1);
            __syncthreads();
        }

        if (MODE == 0) {
            // Hadamard epilogue: q_n += sum_{a in rb} Phi[a, n] * T[a, n]
            float *Ht = stage0;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                float4 h[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int q = tid + 256 * (jb * 4 + j);
                    int64_t n = n0 + (q >> 5);
                    if (n > N - 1) n = N - 1;
                    h[j] = *reinterpret_cast<const float4 *>(Phi + n * (int64_t)M + rb * BS + ((q & 31) << 2));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int q = tid + 256 * (jb * 4 + j);
                    *reinterpret_cast<float4 *>(Ht + (q >> 5) * HPITCH + ((q & 31) << 2)) = h[j];
                }
            }
            __syncthreads();
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int a = wr * 64 + ii * 32 + 8 * g4 + 4 * lk;
                        const int n = wc * 64 + jj * 32 + li;
                        float4 h = *reinterpret_cast<const float4 *>(Ht + n * HPITCH + a);
                        qacc[jj] += acc[ii][jj][4 * g4 + 0] * h.x + acc[ii][jj][4 * g4 + 1] * h.y +
                                    acc[ii][jj][4 * g4 + 2] * h.z + acc[ii][jj][4 * g4 + 3] * h.w;
                    }
            // the __syncthreads() at the top of the next rb iteration protects region 0
        } else {
            // transform: out[n, rb*128 + a] = T[a][n]; one float4 (4 consecutive a) per lane
            float *outl = t_out;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int a = wr * 64 + ii * 32 + 8 * g4 + 4 * lk;
                        const int64_t n = n0 + wc * 64 + jj * 32 + li;
                        if (n < N) {
                            float4 v = make_float4(acc[ii][jj][4 * g4 + 0], acc[ii][jj][4 * g4 + 1],
                                                   acc[ii][jj][4 * g4 + 2], acc[ii][jj][4 * g4 + 3]);
                            *reinterpret_cast<float4 *>(outl + n * (int64_t)M + rb * BS + a) = v;
                        }
                    }
        }
    }

    if (MODE == 0) {
        // combine: lane halves (rows 4*lk), the two wr waves, the two mean halves
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) qacc[jj] += __shfl_xor(qacc[jj], 32);
        __syncthreads();
        if (lk == 0) {
            qred[wr * NT + wc * 64 + li] = qacc[0];
            qred[wr * NT + wc * 64 + 32 + li] = qacc[1];
        }
        mred[(tid >> 7) * NT + (tid & 127)] = macc;
        __syncthreads();
        if (tid < NT) {
            const int64_t n = n0 + tid;
            if (n < N) {
                float q = qred[tid] + qred[NT + tid];
                float m = mred[tid] + mred[NT + tid];
                if (mu0) m += mu0[(int64_t)l * N + n];
                mu_out[(int64_t)l * N + n] = m;
                var_out[(int64_t)l * N + n] = kdiag[n] - q;
            }
        }
    }
}

size_t marginal_lds_bytes(int M) {
    constexpr int kStageFloats = KT * BS + NT * KPITCH;
    constexpr int kRegion0 = (NT * HPITCH > 2 * kStageFloats) ? NT * HPITCH : 2 * kStageFloats;
    return sizeof(float) * (size_t)(kRegion0 + M + 4 * NT);
}

// ------------------------------------------------------------------------------------------------
// syrk kernel: grid = (pairs, splits, L)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void syrk_kernel(int64_t N, int M, const float *__restrict__ Phi,
                                                      const float *__restrict__ gamma_all,
                                                      const float *__restrict__ beta_all,
                                                      double *__restrict__ slabG, double *__restrict__ slabg,
                                                      int64_t chunk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kPanel = KT * BS; // 4096 floats
    float *panels = smem;           // [2 buf][2 panel][kPanel]
    float *sgam = smem + 4 * kPanel; // [2][KT]
    float *sbet = sgam + 2 * KT;     // [2][KT]
    float *gred = sbet + 2 * KT;     // [128] doubles worth of floats x2 (used as double[128])

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    const int p = blockIdx.x, s = blockIdx.y, l = blockIdx.z;
    const int nsplit = gridDim.y, npairs = gridDim.x;
    const int nb = M / BS;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= p) ++bi;
    const int bj = p - bi * (bi + 1) / 2;
    const bool diag = (bi == bj);

    const int64_t nbeg = (int64_t)s * chunk;
    int64_t nend = nbeg + chunk;
    if (nend > N) nend = N;
    const int nstage = nbeg < nend ? (int)((nend - nbeg + KT - 1) / KT) : 0;
    const float *gam = gamma_all + (int64_t)l * N;
    const float *bet = beta_all + (int64_t)l * N;

    f32x16 acc[2][2];
    double dacc[2][2][16];
    float gacc = 0.f;
    double gdacc = 0.0;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[ii][jj][r] = 0.f;
                dacc[ii][jj][r] = 0.0;
            }

    float4 ar[4], br[4];
    float gv = 0.f;
    auto load_stage = [&](int st) {
        const int64_t nbase = nbeg + (int64_t)st * KT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int q = tid + 256 * j;
            int64_t n = nbase + (q >> 5);
            if (n > N - 1) n = N - 1;
            const float *src = Phi + n * (int64_t)M + ((q & 31) << 2);
            ar[j] = *reinterpret_cast<const float4 *>(src + bi * BS);
            if (!diag) br[j] = *reinterpret_cast<const float4 *>(src + bj * BS);
        }
        if (tid < 2 * KT) {
            int64_t n = nbase + (tid & (KT - 1));
            gv = 0.f;
            if (n < nend) gv = (tid < KT) ? gam[n] : bet[n];
        }
    };
    auto store_stage = [&](int buf) {
        float *A = panels + buf * 2 * kPanel;
        float *B = A + kPanel;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int q = tid + 256 * j;
            *reinterpret_cast<float4 *>(A + (q >> 5) * BS + ((q & 31) << 2)) = ar[j];
            if (!diag) *reinterpret_cast<float4 *>(B + (q >> 5) * BS + ((q & 31) << 2)) = br[j];
        }
        if (tid < KT)
            sgam[buf * KT + tid] = gv;
        else if (tid < 2 * KT)
            sbet[buf * KT + tid - KT] = gv;
    };

    if (nstage > 0) {
        load_stage(0);
        store_stage(0);
    }
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstage) load_stage(st + 1);
        const float *A = panels + buf * 2 * kPanel;
        const float *B = diag ? A : A + kPanel;
        const float *pa = A + lk * BS + wr * 64 + li;
        const float *pb = B + lk * BS + wc * 64 + li;
        const float *pg = sgam + buf * KT + lk;
#pragma unroll
        for (int k0 = 0; k0 < KT; k0 += 2) {
            float ga = pg[k0];
            float a0 = pa[k0 * BS] * ga, a1 = pa[k0 * BS + 32] * ga;
            float b0 = pb[k0 * BS], b1 = pb[k0 * BS + 32];
            acc[0][0] = mfma(a0, b0, acc[0][0]);
            acc[0][1] = mfma(a0, b1, acc[0][1]);
            acc[1][0] = mfma(a1, b0, acc[1][0]);
            acc[1][1] = mfma(a1, b1, acc[1][1]);
        }
        if (diag) { // g = Phi beta for the rows of this diagonal block
            const float *ga = A + (tid >> 7) * 16 * BS + (tid & 127);
            const float *gb = sbet + buf * KT + (tid >> 7) * 16;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) gacc += gb[kk] * ga[kk * BS];
        }
        if (((st + 1) % kFlushStages) == 0) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        dacc[ii][jj][r] += (double)acc[ii][jj][r];
                        acc[ii][jj][r] = 0.f;
                    }
            gdacc += (double)gacc;
            gacc = 0.f;
        }
        if (st + 1 < nstage) store_stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[ii][jj][r] += (double)acc[ii][jj][r];
    gdacc += (double)gacc;

    double *slab = slabG + (((int64_t)l * npairs + p) * nsplit + s) * (int64_t)(BS * BS);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 64 + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int col = wc * 64 + jj * 32 + li;
                slab[row * BS + col] = dacc[ii][jj][r];
            }
    if (diag) {
        double *gd = reinterpret_cast<double *>(gred);
        if (tid >= 128) gd[tid - 128] = gdacc;
        __syncthreads();
        if (tid < 128) slabg[(((int64_t)l * nb + bi) * nsplit + s) * BS + tid] = gdacc + gd[tid];
    }
}

size_t syrk_lds_bytes() { return sizeof(float) * (size_t)(4 * KT * BS + 4 * KT) + sizeof(double) * 128; }

// fixed-order reduction of the slabs into the full symmetric G and g
__global__ void reduce_G_kernel(int M, int nsplit, const double *__restrict__ slabG, double *__restrict__ G) {
    const int nb = M / BS;
    const int npairs = nb * (nb + 1) / 2;
    const int l = blockIdx.z;
    const int row = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= M) return;
    int r = row, c = col;
    if (r < c) { // mirror: always read the lower-triangle element
        int t = r;
        r = c;
        c = t;
    }
    const int rb = r / BS, cb = c / BS;
    const int p = rb * (rb + 1) / 2 + cb;
    const double *slab = slabG + (((int64_t)l * npairs + p) * nsplit) * (int64_t)(BS * BS) + (r % BS) * BS + (c % BS);
    double acc = 0.0;
    for (int s = 0; s < nsplit; ++s) acc += slab[(int64_t)s * BS * BS];
    G[((int64_t)l * M + row) * M + col] = acc;
}

__global__ void reduce_g_kernel(int M, int nsplit, const double *__restrict__ slabg, double *__restrict__ g) {
    const int l = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= M) return;
    const int nb = M / BS;
    const double *sl = slabg + (((int64_t)l * nb + a / BS) * nsplit) * BS + (a % BS);
    double acc = 0.0;
    for (int s = 0; s < nsplit; ++s) acc += sl[(int64_t)s * BS];
    g[(int64_t)l * M + a] = acc;
}

int syrk_nsplit(int64_t N, int M, int L, int64_t *chunk_out) {
    const int nb = M / BS;
    const int npairs = nb * (nb + 1) / 2;
    // ~2 resident workgroups per CU x 256 CUs, a few waves of them for balance
    int64_t target = (int64_t)(3 * 512) / ((int64_t)npairs * L);
    if (target < 1) target = 1;
    int64_t chunk = agpl_cdiv(agpl_cdiv(N, target), KT) * KT;
    if (chunk < 4 * KT) chunk = 4 * KT;
    int64_t ns = agpl_cdiv(N, chunk);
    if (ns < 1) ns = 1;
    *chunk_out = chunk;
    return (int)ns;
}

} // namespace

extern "C" int64_t agpl_workspace_bytes(int64_t N, int32_t M, int32_t L) {
    if (N <= 0 || M <= 0 || M % BS || L <= 0) return 0;
    int64_t chunk;
    const int ns = syrk_nsplit(N, M, L, &chunk);
    const int nb = M / BS;
    const int64_t npairs = (int64_t)nb * (nb + 1) / 2;
    int64_t bytes = sizeof(double) * ((int64_t)L * npairs * ns * BS * BS + (int64_t)L * nb * ns * BS);
    bytes += sizeof(float) * 4 * (int64_t)L * N; // mu, var, gamma, beta of the fused pass
    return bytes + 1024;
}

extern "C" int32_t agpl_marginals(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi,
                                  const float *kdiag, const float *mu0, const float *Wpack, const float *alpha,
                                  float *mu_out, float *var_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (N < 0 || M <= 0 || L <= 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes N=%lld M=%d L=%d", (long long)N, M, L);
    if (M % BS) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "M = %d must be a multiple of %d (zero-pad the features)", M, BS);
    if (N == 0) return AGPL_OK;
    if (!Phi || !kdiag || !Wpack || !alpha || !mu_out || !var_out)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    const size_t lds = marginal_lds_bytes(M);
    AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&marginal_kernel<0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)agpl_cdiv(N, NT), (unsigned)L);
    int32_t rc = agpl_timing_begin(ctx, 0);
    if (rc) return rc;
    marginal_kernel<0><<<grid, 256, lds, ctx->stream>>>(N, M, Phi, kdiag, mu0, Wpack, alpha, mu_out, var_out, nullptr);
    AGPL_LAUNCH_CHECK(ctx);
    rc = agpl_timing_end(ctx, 0);
    if (rc) return rc;
    return AGPL_OK;
}

extern "C" int32_t agpl_transform_features(agpl_ctx *ctx, int64_t N, int32_t M, const float *A_colmajor,
                                           const float *in, float *out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (N < 0 || M <= 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes");
    if (M % BS) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "M = %d must be a multiple of %d", M, BS);
    if (N == 0) return AGPL_OK;
    if (!A_colmajor || !in || !out || in == out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null or aliased argument");
    const size_t lds = marginal_lds_bytes(M);
    AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&marginal_kernel<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)agpl_cdiv(N, NT), 1);
    marginal_kernel<1><<<grid, 256, lds, ctx->stream>>>(N, M, in, nullptr, nullptr, A_colmajor, nullptr, nullptr,
                                                        nullptr, out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

// internal: accumulate with caller-provided slab storage (used by agpl_accumulate and agpl_cavi_pass)
int32_t agpl_accumulate_impl(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi, const float *beta,
                             const float *gamma, double *G_out, double *g_out, void *slab_mem) {
    int64_t chunk;
    const int ns = syrk_nsplit(N, M, L, &chunk);
    const int nb = M / BS;
    const int npairs = nb * (nb + 1) / 2;
    double *slabG = (double *)slab_mem;
    double *slabg = slabG + (int64_t)L * npairs * ns * BS * BS;
    const size_t lds = syrk_lds_bytes();
    AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&syrk_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)npairs, (unsigned)ns, (unsigned)L);
    int32_t rc = agpl_timing_begin(ctx, 1);
    if (rc) return rc;
    syrk_kernel<<<grid, 256, lds, ctx->stream>>>(N, M, Phi, gamma, beta, slabG, slabg, chunk);
    AGPL_LAUNCH_CHECK(ctx);
    rc = agpl_timing_end(ctx, 1);
    if (rc) return rc;
    dim3 rg((unsigned)agpl_cdiv(M, 128), (unsigned)M, (unsigned)L);
    reduce_G_kernel<<<rg, 128, 0, ctx->stream>>>(M, ns, slabG, G_out);
    AGPL_LAUNCH_CHECK(ctx);
    dim3 rg2((unsigned)agpl_cdiv(M, 128), (unsigned)L);
    reduce_g_kernel<<<rg2, 128, 0, ctx->stream>>>(M, ns, slabg, g_out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

size_t agpl_slab_bytes(int64_t N, int32_t M, int32_t L) {
    int64_t chunk;
    const int ns = syrk_nsplit(N, M, L, &chunk);
    const int nb = M / BS;
    const int64_t npairs = (int64_t)nb * (nb + 1) / 2;
    return sizeof(double) * (size_t)((int64_t)L * npairs * ns * BS * BS + (int64_t)L * nb * ns * BS);
}

extern "C" int32_t agpl_accumulate(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi,
                                   const float *beta, const float *gamma, double *G_out, double *g_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (N <= 0 || M <= 0 || L <= 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes N=%lld M=%d L=%d", (long long)N, M, L);
    if (M % BS) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "M = %d must be a multiple of %d (zero-pad the features)", M, BS);
    if (!Phi || !beta || !gamma || !G_out || !g_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    int32_t rc = agpl_ws_reserve(ctx, agpl_slab_bytes(N, M, L));
    if (rc) return rc;
    return agpl_accumulate_impl(ctx, N, M, L, Phi, beta, gamma, G_out, g_out, ctx->ws);
}
