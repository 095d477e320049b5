// agpl_mfma.hip -- the two O(N M^2) contractions of a sparse sweep on the gfx950 matrix cores.
//
// Feature matrix Phi: float32 [M, N] column-major (one point = M contiguous floats), M % 128 == 0.
//
//   marginal_kernel  (a11)  T = W' Phi_tile  (f32 MFMA 32x32x2),  q_i = sum_a Phi[a,i] T[a,i],
//                           mu_i = alpha' phi_i ; var_i = kdiag_i - q_i.
//                           W' = upper-triangular with doubled off-diagonal, so only the block pairs
//                           cb >= rb are visited: (nb+1)/(2 nb) of a full GEMM.
//   syrk_kernel      (a12)  G = Phi Diag(gamma) Phi' on 128 x 128 output tiles of the lower triangle,
//                           N split over workgroups in slices of agpl_chunk_points(M) = 4096 points (8192 at M <= 256) (one f32 accumulation run
//                           each), g = Phi beta on the diagonal tiles.
//   reduce kernels          fixed-order float64 sum of the per-slice f32 slabs.
//
// Both MFMA kernels: 256 threads = 4 waves, each wave a 64 x 64 sub-tile = 2 x 2 accumulators of
// v_mfma_f32_32x32x2_f32 (64 accumulator VGPRs); operand tiles are staged global -> registers -> LDS
// with one barrier per 16-deep k-slice and the next slice's global loads in flight during the MFMAs;
// <= 128 VGPRs and <= 40 KB LDS per workgroup: 4 workgroups = 16 waves per CU (4 per SIMD), so the LDS
// latency and barrier waits of one wave hide behind the other waves' MFMAs (each MFMA holds the pipe 64 cycles).
// LDS images: [k][row] with the row index contiguous (conflict-free ds_read_b32: lanes 0-31 read
// consecutive rows at k, lanes 32-63 at k+1), except the Phi operand of marginal_kernel whose k index
// is the contiguous one in memory: image [point][17] (odd pitch -> conflict-free).
#include <cstdlib>

#include "agpl_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BS = 128;      // block of feature rows
constexpr int KT = 16;       // k-slice per stage
constexpr int NT = 128;      // points per marginal tile
constexpr int KPITCH = KT + 1;

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// marginal / transform kernel
//   MODE 0: quadratic form + mean (outputs mu, var);  P = Wpack (block pairs cb >= rb only)
//   MODE 1: transform out[:, i] = A in[:, i];         P = A' (all block pairs), writes float4 per lane
// grid = (tiles of 128 points, L).  ~37-39 KB LDS, <= 128 VGPRs: 4 workgroups (16 waves) per CU.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256, 4) void marginal_kernel(int64_t N, int M, const float *__restrict__ Phi,
                                                          const float *__restrict__ kdiag,
                                                          const float *__restrict__ mu0,
                                                          const float *__restrict__ Pall,
                                                          const float *__restrict__ alpha_all,
                                                          float *__restrict__ mu_out, float *__restrict__ var_out,
                                                          float *__restrict__ t_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kStageFloats = KT * BS + NT * KPITCH; // Pt [16][128] + Kt [128][17]
    float *stage0 = smem;                                // [2][kStageFloats]
    float *alpha_s = smem + 2 * kStageFloats;            // M floats
    float *qred = alpha_s + M;                           // 2 x 128
    float *mred = qred + 2 * NT;                         // 2 x 128

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

    // per-thread staging coordinates (2 float4 of each operand per stage), 32-bit offsets from uniform bases
    //   Pt: q = tid + 256 j -> kk = q >> 5, a4 = q & 31     Kt: q -> n = q >> 2, k4 = q & 3
    const float *tile = Phi + n0 * (int64_t)M;               // uniform
    const int nlim = (int)((N - 1 - n0) < (NT - 1) ? (N - 1 - n0) : (NT - 1)); // last valid local point
    int koff[2], poff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + 256 * j;
        int n = q >> 2;
        n = n > nlim ? nlim : n;
        koff[j] = n * M + ((q & 3) << 2);
        poff[j] = (q >> 5) * M + ((q & 31) << 2);
    }
    // this lane's Hadamard / output columns
    int hoff[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        int n = wc * 64 + jj * 32 + li;
        n = n > nlim ? nlim : n;
        hoff[jj] = n * M;
    }

    float qacc[2] = {0.f, 0.f};
    float macc = 0.f;
    float4 pr0, pr1, kr0, kr1;

#define AGPL_MARG_LOAD(rb_, b0_)                                                        \
    do {                                                                                \
        const float *psrc_ = P + (int64_t)(b0_) * M + (rb_) * BS;                       \
        const float *ksrc_ = tile + (b0_);                                              \
        pr0 = *reinterpret_cast<const float4 *>(psrc_ + poff[0]);                       \
        pr1 = *reinterpret_cast<const float4 *>(psrc_ + poff[1]);                       \
        kr0 = *reinterpret_cast<const float4 *>(ksrc_ + koff[0]);                       \
        kr1 = *reinterpret_cast<const float4 *>(ksrc_ + koff[1]);                       \
    } while (0)
#define AGPL_MARG_STORE(buf_)                                                           \
    do {                                                                                \
        float *Pt_ = stage0 + (buf_) * kStageFloats;                                    \
        float *Kt_ = Pt_ + KT * BS;                                                     \
        *reinterpret_cast<float4 *>(Pt_ + (tid >> 5) * BS + ((tid & 31) << 2)) = pr0;   \
        *reinterpret_cast<float4 *>(Pt_ + ((tid >> 5) + 8) * BS + ((tid & 31) << 2)) = pr1; \
        float *kd_ = Kt_ + (tid >> 2) * KPITCH + ((tid & 3) << 2);                      \
        kd_[0] = kr0.x; kd_[1] = kr0.y; kd_[2] = kr0.z; kd_[3] = kr0.w;                 \
        kd_ += 64 * KPITCH;                                                             \
        kd_[0] = kr1.x; kd_[1] = kr1.y; kd_[2] = kr1.z; kd_[3] = kr1.w;                 \
    } while (0)

    for (int rb = 0; rb < nb; ++rb) {
        f32x16 acc[2][2];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ii][jj][r] = 0.f;

        const int bfirst = (MODE == 0) ? rb * BS : 0;
        const int nstage = (M - bfirst) / KT;
        AGPL_MARG_LOAD(rb, bfirst);
        AGPL_MARG_STORE(0);
        __syncthreads();
        for (int s = 0; s < nstage; ++s) {
            const int buf = s & 1;
            if (s + 1 < nstage) AGPL_MARG_LOAD(rb, bfirst + (s + 1) * KT);
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
                const int bbase = s * KT + (tid >> 7) * 8;
                const float *kp = Kt + (tid & 127) * KPITCH + (tid >> 7) * 8;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) macc += alpha_s[bbase + kk] * kp[kk];
            }
            if (s + 1 < nstage) AGPL_MARG_STORE(buf ^ 1);
            __syncthreads();
        }

        // epilogue for this row block: the lane's accumulators hold T[a][n] for 4 consecutive a (r & 3) at
        // a = rb*128 + wr*64 + ii*32 + 8*g4 + 4*lk and n = column li of sub-tile jj.
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int abase = rb * BS + wr * 64 + ii * 32 + 4 * lk;
                if (MODE == 0) {
                    // Hadamard with Phi straight from global/L2 (the tile was streamed moments ago):
                    // q_n += sum_a Phi[a, n] * T[a, n]
#pragma unroll
                    for (int g2 = 0; g2 < 4; g2 += 2) { // two float4 in flight: keeps the kernel at 128 VGPRs
                        const float4 h0 = *reinterpret_cast<const float4 *>(tile + hoff[jj] + abase + 8 * g2);
                        const float4 h1 = *reinterpret_cast<const float4 *>(tile + hoff[jj] + abase + 8 * g2 + 8);
                        qacc[jj] += acc[ii][jj][4 * g2 + 0] * h0.x + acc[ii][jj][4 * g2 + 1] * h0.y +
                                    acc[ii][jj][4 * g2 + 2] * h0.z + acc[ii][jj][4 * g2 + 3] * h0.w;
                        qacc[jj] += acc[ii][jj][4 * g2 + 4] * h1.x + acc[ii][jj][4 * g2 + 5] * h1.y +
                                    acc[ii][jj][4 * g2 + 6] * h1.z + acc[ii][jj][4 * g2 + 7] * h1.w;
                    }
                } else {
                    const int64_t n = n0 + wc * 64 + jj * 32 + li;
                    if (n < N) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
                            *reinterpret_cast<float4 *>(t_out + n * (int64_t)M + abase + 8 * g4) =
                                make_float4(acc[ii][jj][4 * g4 + 0], acc[ii][jj][4 * g4 + 1],
                                            acc[ii][jj][4 * g4 + 2], acc[ii][jj][4 * g4 + 3]);
                    }
                }
            }
    }

    if (MODE == 0) {
        // combine: lane halves (rows 4*lk), the two wr waves, the two mean halves
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) qacc[jj] += __shfl_xor(qacc[jj], 32);
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

#undef AGPL_MARG_LOAD
#undef AGPL_MARG_STORE

size_t marginal_lds_bytes(int M) {
    constexpr int kStageFloats = KT * BS + NT * KPITCH;
    return sizeof(float) * (size_t)(2 * kStageFloats + M + 4 * NT);
}

// ------------------------------------------------------------------------------------------------
// syrk kernel: 1-D grid of npairs * nsplit8 * L workgroups, remapped so that the workgroups that share an
// XCD (blockIdx % 8) walk whole point-slices: all tile pairs of a slice read the same Phi panels through
// one L2.  One workgroup = one 128 x 128 tile pair x one slice of agpl_chunk_points(M) points, accumulated in f32
// (chains of <= 8192 terms) and written as an f32 slab; the slabs are summed in float64 by the reduce kernels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void syrk_kernel(int64_t N, int M, int npairs, int nsplit, int chunk, int nbig, int small,
                                                      const float *__restrict__ Phi,
                                                      const float *__restrict__ gamma_all,
                                                      const float *__restrict__ beta_all,
                                                      float *__restrict__ slabG, float *__restrict__ slabg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kPanel = KT * BS;  // 2048 floats
    float *panels = smem;            // [2 buf][2 panel][kPanel]
    float *sgb = smem + 4 * kPanel;  // [2 buf][gamma KT | beta KT]
    float *gred = sgb + 4 * KT;      // [128]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    // XCD-aware decode (speed only): id % 8 labels the XCD group; each group owns slices s = 8 t + xcd
    const int nsplit8 = (nsplit + 7) / 8;
    const int per_l = npairs * nsplit8 * 8;
    const int l = blockIdx.x / per_l;
    const int id = blockIdx.x - l * per_l;
    const int xcd = id & 7, j = id >> 3;
    const int s = (j / npairs) * 8 + xcd;
    const int p = j % npairs;
    if (s >= nsplit) return;
    const int nb = M / BS;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= p) ++bi;
    const int bj = p - bi * (bi + 1) / 2;
    const bool diag = (bi == bj);
    // On a diagonal tile the upper 64 x 64 wave tile is the transpose of the lower one and is never read by the
    // reduction (it mirrors from the lower triangle): that wave stages and synchronises but issues no MFMA, which
    // frees its SIMD's matrix pipe for the other resident workgroups.
    const bool active = !(diag && wr < wc);

    int64_t nbeg, nend;
    agpl_slice_range(s, chunk, nbig, small, N, nbeg, nend);
    const int nstage = (int)((nend - nbeg + KT - 1) / KT);
    const float *gam = gamma_all + (int64_t)l * N;
    const float *bet = beta_all + (int64_t)l * N;

    f32x16 acc[2][2];
    float gacc = 0.f;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ii][jj][r] = 0.f;

    // branch-free staging: every load is unconditional (a diagonal pair loads its panel twice -- the second
    // read is an L1 hit -- and out-of-range points are clamped, their gamma / beta zeroed), so the compiler
    // keeps all of a stage's global loads in flight behind one counted wait.  Uniform base + 32-bit offsets.
    float4 ar0, ar1, br0, br1;
    float gv = 0.f;
    const float *gbsrc = (tid & KT) ? bet : gam;
    const int r0 = tid >> 5, c4 = (tid & 31) << 2; // staging row (point within the slice) and column
#define AGPL_SYRK_LOAD(st_)                                                                  \
    do {                                                                                     \
        const int64_t nbase_ = nbeg + (int64_t)(st_) * KT;                                   \
        const int lim_ = (int)((N - 1 - nbase_) < (KT - 1) ? (N - 1 - nbase_) : (KT - 1));   \
        const float *base_ = Phi + nbase_ * (int64_t)M;                                      \
        const int ra_ = (r0 > lim_ ? lim_ : r0) * M + c4;                                    \
        const int rb_ = ((r0 + 8) > lim_ ? lim_ : (r0 + 8)) * M + c4;                        \
        ar0 = *reinterpret_cast<const float4 *>(base_ + ra_ + bi * BS);                      \
        ar1 = *reinterpret_cast<const float4 *>(base_ + rb_ + bi * BS);                      \
        br0 = *reinterpret_cast<const float4 *>(base_ + ra_ + bj * BS);                      \
        br1 = *reinterpret_cast<const float4 *>(base_ + rb_ + bj * BS);                      \
        int64_t n_ = nbase_ + (tid & (KT - 1));                                              \
        const float keep_ = n_ < nend ? 1.f : 0.f;                                           \
        n_ = n_ > N - 1 ? N - 1 : n_;                                                        \
        gv = gbsrc[n_] * keep_;                                                              \
    } while (0)
#define AGPL_SYRK_STORE(buf_)                                                                \
    do {                                                                                     \
        float *A_ = panels + (buf_) * 2 * kPanel;                                            \
        float *B_ = A_ + kPanel;                                                             \
        *reinterpret_cast<float4 *>(A_ + r0 * BS + c4) = ar0;                                \
        *reinterpret_cast<float4 *>(A_ + (r0 + 8) * BS + c4) = ar1;                          \
        *reinterpret_cast<float4 *>(B_ + r0 * BS + c4) = br0;                                \
        *reinterpret_cast<float4 *>(B_ + (r0 + 8) * BS + c4) = br1;                          \
        if (tid < 2 * KT) sgb[(buf_) * 2 * KT + tid] = gv; /* [gamma 0..15 | beta 0..15] */  \
    } while (0)

    AGPL_SYRK_LOAD(0);
    AGPL_SYRK_STORE(0);
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstage) AGPL_SYRK_LOAD(st + 1);
        const float *A = panels + buf * 2 * kPanel;
        const float *B = A + kPanel;
        const float *pa = A + lk * BS + wr * 64 + li;
        const float *pb = B + lk * BS + wc * 64 + li;
        const float *pg = sgb + buf * 2 * KT + lk;
        if (active) { // wave-uniform: the (wr < wc) wave of a diagonal tile would only recompute the mirror image
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
        }
        if (diag) { // g = Phi beta for the rows of this diagonal block (wave-uniform branch)
            const float *ga = A + (tid >> 7) * 8 * BS + (tid & 127);
            const float *gb = sgb + buf * 2 * KT + KT + (tid >> 7) * 8;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) gacc += gb[kk] * ga[kk * BS];
        }
        if (st + 1 < nstage) AGPL_SYRK_STORE(buf ^ 1);
        __syncthreads();
    }
#undef AGPL_SYRK_LOAD
#undef AGPL_SYRK_STORE

    float *slab = slabG + (((int64_t)l * npairs + p) * nsplit + s) * (int64_t)(BS * BS);
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 64 + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int col = wc * 64 + jj * 32 + li;
                slab[row * BS + col] = acc[ii][jj][r];
            }
    if (diag) {
        if (tid >= 128) gred[tid - 128] = gacc;
        __syncthreads();
        if (tid < 128) slabg[(((int64_t)l * nb + bi) * nsplit + s) * BS + tid] = gacc + gred[tid];
    }
}

// float16 fragment types of the matrix-core probe kernels below
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma16(h8v a, h8v b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}



size_t syrk_lds_bytes() { return sizeof(float) * (size_t)(4 * KT * BS + 4 * KT + 128); }

// Fixed-order float64 reduction of the f32 slabs, two levels in ONE launch (HBM-bound streaming, 16 B per lane):
//   level 1: one workgroup = 1 KB of one tile (256 consecutive elements) x one group of kRedGroup slices;
//            wave w sums slices w, w+4, ... of the group (float4 loads, 8 in flight), waves combined in LDS
//            in wave order -> partial[p][group][16384] (float64)
//   level 2: G[row][col] = sum over groups (in order) of the lower-triangle partial, mirrored -- by whichever workgroup of
//            the 1 KB segment arrives LAST at the segment's counter (release / acquire at device scope).  The sum runs over the
//            groups in index order whoever performs it, so the result does not depend on the arrival order.
// (Rounds 1-5 ran level 2 as a second kernel, reduce_G_kernel: one launch and 24-49 us more per sweep.)
constexpr int kRedGroup = 64;

__device__ __forceinline__ double red_load(const double *p) { // a partial another workgroup (possibly on another XCD's L2) wrote
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void red_store(double *p, double x) { // write-through (sc1): visible to every XCD once the store has drained
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// the 16-byte form (the 8-byte one costs 2.7 x per byte on the fabric: MI355X_MICROARCH.md, visibility table).  s_nop 1: a VMEM
// store of more than 8 bytes reads its data registers for two cycles after issue and the hazard recogniser does not look into inline asm
typedef double red_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void red_store2(double *p, double a, double b) {
    red_d2 v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// True in every lane of the calling wave iff this wave's arrival is the ngroup-th at `cnt` (which it then zeroes).  The wave's
// partials were stored write-through and are drained here before the arrival; the consumer reads them by sc1 loads behind the
// counter: no fence on either side (a device-scope fence per workgroup writes back and invalidates the whole L2 -- measured in
// round 6: the C2 sweep 13.1 -> 17.2 ms with 25 000 such workgroups per sweep; cdna_hip_programming.md Guideline 16).
__device__ __forceinline__ bool red_last_arrival(unsigned *cnt, int ngroup, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0u;
    if (lane == 0) {
        old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(ngroup - 1)) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
    return old == (unsigned)(ngroup - 1);
}

// blockIdx.z < npl: tile pl = l * npairs + p of G; beyond: 128-row block bl = l * nb + bi of g (one launch for both)
__global__ __launch_bounds__(256) void reduce_slab_kernel(int M, int nsplit, int ngroup, int npairs, int npl,
                                                          const float *__restrict__ slab, double *partial,
                                                          const float *__restrict__ slabg, double *partialg,
                                                          unsigned *__restrict__ cnt, double *__restrict__ G,
                                                          double *__restrict__ g) {
    __shared__ double sm[3][64][4];
    const int seg = blockIdx.x;  // 1 KB segment of the 128 x 128 tile (64 of them)
    const int grp = blockIdx.y;
    const int pl = blockIdx.z;   // l * npairs + p
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (pl >= npl) { // the g slabs: 128 floats per (row block, slice)
        if (seg != 0 || threadIdx.x >= BS) return;
        const int bl = pl - npl;
        const float *base = slabg + ((int64_t)bl * nsplit) * BS + threadIdx.x;
        const int s0 = grp * kRedGroup;
        int s1 = s0 + kRedGroup;
        if (s1 > nsplit) s1 = nsplit;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int s = s0;
        for (; s + 4 <= s1; s += 4) {
            a0 += (double)base[(int64_t)(s + 0) * BS];
            a1 += (double)base[(int64_t)(s + 1) * BS];
            a2 += (double)base[(int64_t)(s + 2) * BS];
            a3 += (double)base[(int64_t)(s + 3) * BS];
        }
        for (; s < s1; ++s) a0 += (double)base[(int64_t)s * BS];
        red_store(partialg + ((int64_t)bl * ngroup + grp) * BS + threadIdx.x, (a0 + a1) + (a2 + a3));
        // (each of the two waves counts for its own 64 entries)
        if (!red_last_arrival(cnt + (int64_t)npl * 64 + 2 * bl + wave, ngroup, lane)) return;
        const double *src = partialg + ((int64_t)bl * ngroup) * BS + threadIdx.x;
        double acc = 0.0;
        for (int g0 = 0; g0 < ngroup; g0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = red_load(src + (int64_t)(g0 + u < ngroup ? g0 + u : ngroup - 1) * BS);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (g0 + u < ngroup) acc += v[u];
        }
        g[(int64_t)bl * BS + threadIdx.x] = acc; // (bl * BS + t = l * M + bi * BS + t)
        return;
    }
    const float *base = slab + ((int64_t)pl * nsplit) * (int64_t)(BS * BS) + seg * 256 + lane * 4;
    const int s0 = grp * kRedGroup;
    int s1 = s0 + kRedGroup;
    if (s1 > nsplit) s1 = nsplit;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int s = s0 + wave;
    for (; s + 28 < s1; s += 32) { // 8 independent 16-byte loads in flight
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(base + (int64_t)(s + 4 * u) * BS * BS);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 += (double)v[u].x;
            a1 += (double)v[u].y;
            a2 += (double)v[u].z;
            a3 += (double)v[u].w;
        }
    }
    for (; s < s1; s += 4) {
        float4 v = *reinterpret_cast<const float4 *>(base + (int64_t)s * BS * BS);
        a0 += (double)v.x;
        a1 += (double)v.y;
        a2 += (double)v.z;
        a3 += (double)v.w;
    }
    if (wave > 0) {
        sm[wave - 1][lane][0] = a0;
        sm[wave - 1][lane][1] = a1;
        sm[wave - 1][lane][2] = a2;
        sm[wave - 1][lane][3] = a3;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        a0 += sm[w][lane][0];
        a1 += sm[w][lane][1];
        a2 += sm[w][lane][2];
        a3 += sm[w][lane][3];
    }
    double *dst = partial + ((int64_t)pl * ngroup + grp) * (int64_t)(BS * BS) + seg * 256 + lane * 4;
    red_store2(dst + 0, a0, a1);
    red_store2(dst + 2, a2, a3);
    if (!red_last_arrival(cnt + (int64_t)pl * 64 + seg, ngroup, lane)) return;
    // level 2 for this segment: rows 2 seg, 2 seg + 1 of the tile, columns 4 (lane % 32) .. + 3
    const int l = pl / npairs, p = pl - l * npairs;
    int rb = 0;
    while ((rb + 1) * (rb + 2) / 2 <= p) ++rb;
    const int cb = p - rb * (rb + 1) / 2;
    const int tr = 2 * seg + (lane >> 5), tc = 4 * (lane & 31);
    const double *src = partial + ((int64_t)pl * ngroup) * (int64_t)(BS * BS) + seg * 256 + lane * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int g0 = 0; g0 < ngroup; g0 += 8) { // 32 loads in flight (each is a trip to the memory side), summed in group order
        double v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int gi = g0 + u < ngroup ? g0 + u : ngroup - 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[u][e] = red_load(src + (int64_t)gi * BS * BS + e);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (g0 + u < ngroup) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[u][e];
            }
    }
    double *Gl = G + (int64_t)l * M * M;
    const int r = rb * BS + tr;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cb * BS + tc + e;
        if (r < c) continue; // (a diagonal tile's upper half: its mirror image is what G takes -- exact symmetry)
        Gl[(int64_t)r * M + c] = acc[e];
        if (r != c) Gl[(int64_t)c * M + r] = acc[e];
    }
}

// g: level 1 = one workgroup of 128 threads per (row block, group of slices); level 2 sums the groups


} // namespace

size_t agpl_slab_bytes(int64_t N, int32_t M, int32_t L);

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

// scratch layout of one accumulation: [f32 slabs G][f32 slabs g][f64 partials G][f64 partials g]
struct SlabLayout {
    int ns, ng, nb, chunk, nbig, small;
    int64_t npairs;
    size_t slabG, slabg, partG, partg, sgam, ctr; // byte offsets
    size_t total;
};
static SlabLayout slab_layout(int64_t N, int32_t M, int32_t L) {
    SlabLayout o;
    const agpl_slices sl = agpl_slice_plan(N, M, L);
    o.chunk = sl.chunk;
    o.nbig = sl.nbig;
    o.small = sl.small;
    o.ns = sl.ns;
    o.ng = (o.ns + kRedGroup - 1) / kRedGroup;
    o.nb = M / BS;
    o.npairs = (int64_t)o.nb * (o.nb + 1) / 2;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    o.slabG = 0;
    o.slabg = al(o.slabG + sizeof(float) * (size_t)(L * o.npairs * o.ns * BS * BS));
    o.partG = al(o.slabg + sizeof(float) * (size_t)((int64_t)L * o.nb * o.ns * BS));
    o.partg = al(o.partG + sizeof(double) * (size_t)(L * o.npairs * o.ng * BS * BS));
    o.sgam = al(o.partg + sizeof(double) * (size_t)((int64_t)L * o.nb * o.ng * BS));
    // padded 2^8 sqrt(gamma) | beta (split tile kernel) or the gamma | beta records (image kernel): N rounded up to a 32-point
    // stage + one stage of zeros
    o.ctr = al(o.sgam + 2 * sizeof(float) * (size_t)((int64_t)L * (((N + 31) & ~(int64_t)31) + 32)));
    o.total = al(o.ctr + sizeof(unsigned) * 64); // scale / flag words of the image kernel
    return o;
}

size_t agpl_slab_bytes(int64_t N, int32_t M, int32_t L) { return slab_layout(N, M, L).total; }

// internal: accumulate with caller-provided slab storage (used by agpl_accumulate and agpl_cavi_pass)
int32_t agpl_syrk_image_launch(agpl_ctx *ctx, int64_t N, int64_t Npad, int32_t M, int32_t L, const void *image,
                               const float *gamma, const float *beta, float *gb, unsigned *scal, float *slabG,
                               float *slabg, int ns, int chunk, int nbig, int small, bool records_ready); // agpl_syrk.hip

// acc_image != nullptr (and M % 256 == 0): the point-major split-float16 image of agpl_accumulate_image is the operand
// (syrk_strip_kernel, agpl_syrk.hip) and Phi is not read; otherwise Phi is, by the kernel ctx->accumulate_split selects.
// internal (agpl_update.hip): where the gamma | beta records and the two scale words of the image path live in slab_mem
void agpl_accumulate_records(int64_t N, int32_t M, int32_t L, void *slab_mem, float **gb, unsigned **scal) {
    const SlabLayout lo = slab_layout(N, M, L);
    *gb = (float *)((char *)slab_mem + lo.sgam);
    *scal = (unsigned *)((char *)slab_mem + lo.ctr);
}

// records_ready: the caller's per-point kernel has filled agpl_accumulate_records already (image path; beta / gamma unread)
int32_t agpl_accumulate_impl(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi, const void *acc_image,
                             const float *beta, const float *gamma, double *G_out, double *g_out, void *slab_mem,
                             bool records_ready) {
    const bool use_image = acc_image && M % 256 == 0;
    const SlabLayout lo = slab_layout(N, M, L);
    const int ns = lo.ns, nb = lo.nb, npairs = (int)lo.npairs, ng = lo.ng;
    float *slabG = (float *)((char *)slab_mem + lo.slabG);
    float *slabg = (float *)((char *)slab_mem + lo.slabg);
    double *partG = (double *)((char *)slab_mem + lo.partG);
    double *partg = (double *)((char *)slab_mem + lo.partg);
    const size_t lds = syrk_lds_bytes();
    int32_t rc_cnt = agpl_red_cnt_reserve(ctx, (int64_t)L * npairs * 64 + 2 * (int64_t)L * nb);
    if (rc_cnt) return rc_cnt;
    const int64_t nwg = (int64_t)L * npairs * ((ns + 7) / 8) * 8;
    if (nwg > 0x7fffffffLL) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "problem too large for one launch");
    if (!use_image && !Phi)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "the accumulation needs the float32 features (no image, or M %% 256 != 0)");
    int32_t rc = agpl_timing_begin(ctx, 1);
    if (rc) return rc;
    int32_t launch_rc = AGPL_OK; // reported behind agpl_timing_end: a failed launch must not leave an open event pair
    if (use_image) {
        const int64_t Npad = ((N + 31) & ~(int64_t)31) + 32;
        launch_rc = agpl_syrk_image_launch(ctx, N, Npad, M, L, acc_image, gamma, beta, (float *)((char *)slab_mem + lo.sgam),
                                           (unsigned *)((char *)slab_mem + lo.ctr), slabG, slabg, ns, lo.chunk, lo.nbig, lo.small, records_ready);
    } else if (ctx->accumulate_split) {
        launch_rc = AGPL_ERR_INVALID_ARGUMENT; // (internal: the split-float16 accumulation exists on the image only)
        snprintf(ctx->err, sizeof(ctx->err), "the split-float16 accumulation needs the accumulate image and M %% 256 == 0");
    } else
        syrk_kernel<<<(unsigned)nwg, 256, lds, ctx->stream>>>(N, M, npairs, ns, lo.chunk, lo.nbig, lo.small, Phi, gamma, beta, slabG, slabg);
    rc = agpl_timing_end(ctx, 1);
    if (launch_rc) return launch_rc;
    AGPL_LAUNCH_CHECK(ctx);
    if (rc) return rc;
    // fixed-order float64 reduction of the slabs, G and g together: slices -> groups of kRedGroup, groups -> G, g
    dim3 r1(64, (unsigned)ng, (unsigned)(L * npairs + L * nb));
    reduce_slab_kernel<<<r1, 256, 0, ctx->stream>>>(M, ns, ng, npairs, L * npairs, slabG, partG, slabg, partg, ctx->red_cnt, G_out,
                                                    g_out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

extern "C" int32_t agpl_accumulate(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const float *Phi,
                                   const float *beta, const float *gamma, double *G_out, double *g_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (N <= 0 || M <= 0 || L <= 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes N=%lld M=%d L=%d", (long long)N, M, L);
    if (M % BS) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "M = %d must be a multiple of %d (zero-pad the features)", M, BS);
    if (!Phi || !beta || !gamma || !G_out || !g_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    int32_t rc = agpl_ws_reserve(ctx, agpl_slab_bytes(N, M, L));
    if (rc) return rc;
    return agpl_accumulate_impl(ctx, N, M, L, Phi, nullptr, beta, gamma, G_out, g_out, ctx->ws, false);
}

// ------------------------------------------------------------------------------------------------
// Measured float16 MFMA rate under sustained load (the ceiling the split contractions are priced against next to the
// data-sheet peak): v_mfma_f32_32x32x16_f16 back to back, 12 per step into four 32 x 32 accumulators -- the instruction
// mix of one 16-point stage of the accumulation's wave -- with (mode 1) or without (mode 0) the stage's eight 16-byte
// fragment reads from LDS.  Launches are long (milliseconds) and repeated so that the clock the power management
// settles on is the one measured.
// ------------------------------------------------------------------------------------------------
namespace {
template <int MODE>
__global__ __launch_bounds__(256, 4) void mfma_f16_probe_kernel(int iters, float *__restrict__ sink) {
    __shared__ h8v frag[8][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int f = 0; f < 8; ++f) {
        h8v v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (_Float16)((((lane * 7 + k * 13 + f * 29 + wave) & 63) - 31.5f) * 0.03125f);
        frag[f][threadIdx.x] = v;
    }
    __syncthreads();
    h8v ah0 = frag[0][threadIdx.x], ah1 = frag[1][threadIdx.x], al0 = frag[2][threadIdx.x], al1 = frag[3][threadIdx.x];
    h8v bh0 = frag[4][threadIdx.x], bh1 = frag[5][threadIdx.x], bl0 = frag[6][threadIdx.x], bl1 = frag[7][threadIdx.x];
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1) {
            const int o = (threadIdx.x + i) & 255; // a changing address: the reads cannot be hoisted
            ah0 = frag[0][o]; ah1 = frag[1][o]; bh0 = frag[4][o]; bh1 = frag[5][o];
        }
        c00 = mfma16(ah0, bh0, c00); c01 = mfma16(ah0, bh1, c01); c10 = mfma16(ah1, bh0, c10); c11 = mfma16(ah1, bh1, c11);
        if (MODE == 1) { const int o = (threadIdx.x + i + 1) & 255; bl0 = frag[6][o]; bl1 = frag[7][o]; }
        c00 = mfma16(ah0, bl0, c00); c01 = mfma16(ah0, bl1, c01); c10 = mfma16(ah1, bl0, c10); c11 = mfma16(ah1, bl1, c11);
        if (MODE == 1) { const int o = (threadIdx.x + i + 2) & 255; al0 = frag[2][o]; al1 = frag[3][o]; }
        c00 = mfma16(al0, bh0, c00); c01 = mfma16(al0, bh1, c01); c10 = mfma16(al1, bh0, c10); c11 = mfma16(al1, bh1, c11);
        if (MODE == 0) { ah0 = -ah0; bl1 = -bl1; } // bounded sums, changing operands
    }
    const f32x16 s = c00 + c01 + c10 + c11;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += s[k];
    if (t == 1.2345e30f) sink[0] = t; // never true: keeps the chain alive
}

// The same for the shape the shipped kernels issue (round 4): v_mfma_f32_16x16x32_f16, 48 per step into sixteen 16 x 16
// accumulators = one 32-deep stage of marginal_factor_queue_kernel's wave (4 x 4 blocks x hi hi' + hi lo' + lo hi'), with
// (MODE 1) or without (MODE 0) the stage's sixteen 16-byte fragment reads from LDS; operands are hashed bit patterns of
// float16 normals in [-2, 2) (the clock a matrix loop holds depends on how the operand bits toggle: MI355X_MICROARCH.md, DVFS).
template <int MODE>
__global__ __launch_bounds__(256, 2) void mfma_f16_probe32_kernel(int iters, float *__restrict__ sink) {
    __shared__ h8v frag[16][64 * 4];
    for (int f = 0; f < 16; ++f) {
        h8v v;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned x = (unsigned)(threadIdx.x * 8 + k) * 2654435761u + (unsigned)f * 40503u + blockIdx.x * 97u;
            x ^= x >> 15, x *= 2246822519u, x ^= x >> 13;
            const unsigned short bits = (unsigned short)((x & 0x8000u) | (((x >> 16) & 0x3FFu)) | ((12u + ((x >> 26) & 3u)) << 10));
            v[k] = __builtin_bit_cast(_Float16, bits);
        }
        frag[f][threadIdx.x] = v;
    }
    __syncthreads();
    h8v ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ah[q] = frag[q][threadIdx.x], al[q] = frag[4 + q][threadIdx.x];
        bh[q] = frag[8 + q][threadIdx.x], bl[q] = frag[12 + q][threadIdx.x];
    }
    typedef float f32x4p __attribute__((ext_vector_type(4)));
    f32x4p acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4p{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            const int o = (threadIdx.x + it) & 255; // a changing address: the reads cannot be hoisted
#pragma unroll
            for (int q = 0; q < 4; ++q) ah[q] = frag[q][o], al[q] = frag[4 + q][o];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 1) {
                const int o = (threadIdx.x + it + j) & 255;
                bh[j] = frag[8 + j][o], bl[j] = frag[12 + j][o];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
        }
        if (MODE == 0) { ah[0] = -ah[0]; bl[3] = -bl[3]; al[2] = -al[2]; } // bounded sums, changing operands
        else if ((it & 63) == 63) { // (random-sign products: a random walk; rescaled now and then so that nothing reaches inf)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] *= 0.03125f;
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 1.2345e30f) sink[0] = t;
}
} // namespace

int32_t agpl_probe_mfma_f64_impl(agpl_ctx *ctx, int32_t iters, double *tflops_host); // agpl_dense.hip

extern "C" int32_t agpl_probe_mfma(agpl_ctx *ctx, int32_t dtype, int32_t iters, int32_t mode, int32_t workgroups_per_cu,
                                   double *tflops_host, double *ms_host) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (dtype == AGPL_F64) {
        if (ms_host) *ms_host = 0.0;
        return agpl_probe_mfma_f64_impl(ctx, iters, tflops_host);
    }
    if (dtype != AGPL_F32) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "dtype must be AGPL_F64 or AGPL_F32 (float16 operands, float32 accumulate)");
    if (iters <= 0 || !tflops_host || mode < 0 || mode > 3 || workgroups_per_cu < 1 || workgroups_per_cu > 4)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    int32_t rc = agpl_ws2_reserve(ctx, 4096);
    if (rc) return rc;
    hipDeviceProp_t prop;
    AGPL_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    hipEvent_t e0, e1;
    AGPL_HIP(ctx, hipEventCreate(&e0));
    AGPL_HIP(ctx, hipEventCreate(&e1));
    const int blocks = prop.multiProcessorCount * workgroups_per_cu, reps = 6;
    for (int rep = 0; rep < 2 + reps; ++rep) { // two untimed launches settle the clocks, `reps` launches are one timed region
        if (rep == 2) AGPL_HIP(ctx, hipEventRecord(e0, ctx->stream));
        if (mode == 0) mfma_f16_probe_kernel<0><<<blocks, 256, 0, ctx->stream>>>(iters, (float *)ctx->ws2);
        else if (mode == 1) mfma_f16_probe_kernel<1><<<blocks, 256, 0, ctx->stream>>>(iters, (float *)ctx->ws2);
        else if (mode == 2) mfma_f16_probe32_kernel<0><<<blocks, 256, 0, ctx->stream>>>(iters, (float *)ctx->ws2);
        else mfma_f16_probe32_kernel<1><<<blocks, 256, 0, ctx->stream>>>(iters, (float *)ctx->ws2);
    }
    AGPL_HIP(ctx, hipEventRecord(e1, ctx->stream));
    AGPL_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    AGPL_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    AGPL_LAUNCH_CHECK(ctx);
    // per wave and iteration: modes 0, 1: 12 x 32x32x16; modes 2, 3: 48 x 16x16x32
    const double flop = (double)reps * blocks * 4.0 * (double)iters * (mode < 2 ? 12.0 * (2.0 * 32 * 32 * 16) : 48.0 * (2.0 * 16 * 16 * 32));
    *tflops_host = flop / ((double)ms * 1e-3) / 1e12;
    if (ms_host) *ms_host = ms / reps;
    return AGPL_OK;
}

