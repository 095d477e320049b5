// agpl_syrk.hip -- the accumulation G = Phi Diag(gamma) Phi', g = Phi beta (a12, docs/src/index.md:154-163:
// S = (K_Z^-1 + kappa Diag(r) kappa')^-1 in the whitened basis) on the float16 matrix cores from a STATIC point-major image.
//
// The reduction index of this product is the POINT, the strided index of the float32 features (one point = M contiguous
// floats).  agpl_accumulate_image writes, once per data set, a second split-float16 image of Phi whose 16-byte granule is
// (one feature, 8 consecutive points) -- exactly one MFMA operand fragment:
//
//   block (point slice ps of 16 points, feature block fb of 128, part hl)  =  [plane 2][feature 128][8 halves]  = 4 KB,
//   blocks ordered [ps][fb][hl]: everything a run of points needs is one contiguous stretch of HBM; plane p holds points
//   8p .. 8p+7 of the slice; hl = 0: hi = f16(s_A phi), hl = 1: lo = f16(s_A phi - hi); s_A = 2^e_A chosen from max |Phi| so
//   that hi and lo stay float16 normals over the widest range (header word scale_exp).
//
// syrk_strip_kernel (below): one 512-thread workgroup (8 waves, two per SIMD, one workgroup per CU) owns a 256 x 256 tile of
// the lower triangle of G for one slice of agpl_chunk_points(M) = 4096 (8192 at M = 256) points (one f32 accumulation run, one slab set -- the slabs and the fixed-
// order float64 reduction behind them are those of agpl_mfma.hip).  A = rows of panel I: image blocks moved HBM -> LDS by the
// DMA path, no VGPRs, no VALU.  B = gamma_n * (rows of panel J): the granules of a wave's own 32 columns, loaded (or, on a
// diagonal tile, read back from the A image in LDS) into registers, rebuilt (x = hi + lo, exact in float32), scaled
// (y = (s_B gamma_n) x) and re-split there (3 v_fma_mix per value) -- the granule is already the MFMA fragment, nothing is
// transposed and B never touches LDS.  G_tile += A B' as hi hi' + hi lo' + lo hi' with v_mfma_f32_16x16x32_f16.
// Compared with syrk_split_kernel (agpl_mfma.hip: 128 x 128 tiles, both panels converted from float32 every stage by every
// tile pair): a quarter of the conversions per flop, half the operand bytes per flop (a third on diagonal tiles), no register
// staging of A.  g = Phi beta rides the B conversion of the diagonal tiles (x is the float32 feature there).
#include <cstdlib>

#include "agpl_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BS = 128;       // feature rows per image block / per slab tile (the slab layout of agpl_mfma.hip)
constexpr int kPanel = 256;   // feature rows per operand panel
constexpr int kStagePts = 32; // points per stage (one MFMA K)
constexpr uint32_t kImageMagic = 0x41474951u; // "AGIQ"

struct AccImageHeader { // 256 bytes in front of the blocks
    uint32_t magic;
    int32_t scale_exp; // e_A: the image holds 2^e_A phi
    float max_abs;     // max |Phi| the scale was chosen for
    uint32_t reserved;
    int64_t N;
    int32_t M;
    int32_t pad[57];
};
static_assert(sizeof(AccImageHeader) == 256, "header size");

__device__ __forceinline__ f32x4 mfma32(h8 a, h8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// hi / lo float16 pair of two products x s, packed (see agpl_mfma.hip AGPL_SPLIT2): one rounding of the exact fma per part
#define AGPL_E_SPLIT2(x0_, s0_, x1_, s1_, H_, L_)                                                              \
    do {                                                                                                       \
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(H_) : "v"(x0_), "v"(s0_));                                  \
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(H_) : "v"(x1_), "v"(s1_));                                  \
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(L_) : "v"(x0_), "v"(s0_), "v"(H_));     \
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"                                 \
            : "+v"(L_)                                                                                         \
            : "v"(x1_), "v"(s1_), "v"(H_));                                                                    \
    } while (0)
// g += b x, one v_fmac_f32 per term in program order (agpl_mfma.hip AGPL_GFMA: no compiler-formed packed float32 forms)
#define AGPL_E_GFMA(acc_, b_, x_) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc_) : "v"(b_), "v"(x_))
// x = hi + lo of the two halves of a packed pair (exact: hi and lo do not overlap and span <= 24 bits)
#define AGPL_Q_JOIN2(H_, L_, x0_, x1_)                                                                         \
    do {                                                                                                       \
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(x0_) : "v"(H_), "v"(L_));                 \
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(x1_) : "v"(H_), "v"(L_));  \
    } while (0)

// ------------------------------------------------------------------------------------------------
// image construction
// ------------------------------------------------------------------------------------------------
// max |x| over n floats as the bit pattern of a non-negative float (orders like an unsigned); a NaN ends up above the
// pattern of +inf, so `bits >= 0x7F800000` says "something is not finite"
// (ntail < 4 floats behind the n4 float4s -- a feature count that is not a multiple of 4 -- are taken by the first threads)
__global__ __launch_bounds__(256) void absmax_kernel(int64_t n4, int ntail, const float4 *__restrict__ x, unsigned *__restrict__ out) {
    unsigned m = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        m = max(m, __float_as_uint(v.x) & 0x7FFFFFFFu);
        m = max(m, __float_as_uint(v.y) & 0x7FFFFFFFu);
        m = max(m, __float_as_uint(v.z) & 0x7FFFFFFFu);
        m = max(m, __float_as_uint(v.w) & 0x7FFFFFFFu);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail)
        m = max(m, __float_as_uint(reinterpret_cast<const float *>(x)[4 * n4 + threadIdx.x]) & 0x7FFFFFFFu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// first element (linear index) that is not finite or whose magnitude is >= limit
__global__ __launch_bounds__(256) void find_bad_kernel(int64_t n, const float *__restrict__ x, float limit,
                                                       unsigned long long *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = fabsf(x[i]);
        if (!(a < limit)) atomicMin(out, (unsigned long long)i);
    }
}

// one workgroup = one (point slice, feature block): 16 points x 128 features through an LDS transpose
// Msrc: the features the caller's rows hold (row pitch Msrc floats); the image's features Msrc .. M - 1 are zero (round 6: a plan
// pads any feature count to the next multiple of 256 itself)
__global__ __launch_bounds__(256) void accumulate_image_kernel(int64_t N, int M, int Msrc, int64_t nps, float scale, int scale_exp,
                                                               float max_abs, const float *__restrict__ Phi,
                                                               unsigned char *__restrict__ image) {
    __shared__ float tile[16][BS + 1];
    const int nb = M / BS;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        AccImageHeader *h = reinterpret_cast<AccImageHeader *>(image);
        h->magic = kImageMagic;
        h->scale_exp = scale_exp;
        h->max_abs = max_abs;
        h->reserved = 0;
        h->N = N;
        h->M = M;
    }
    h8 *blocks = reinterpret_cast<h8 *>(image + sizeof(AccImageHeader));
    const int64_t nblk = nps * nb;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t ps = blk / nb;
        const int fb = (int)(blk - ps * nb);
        {
            const int pt = threadIdx.x >> 4, f0 = (threadIdx.x & 15) * 8;
            const int64_t n = ps * 16 + pt;
            float4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
            if (n < N) {
                const int fg = fb * BS + f0;
                const float *src = Phi + n * (int64_t)Msrc + fg;
                if (!(Msrc & 3) && fg + 8 <= Msrc) {
                    a = *reinterpret_cast<const float4 *>(src);
                    b = *reinterpret_cast<const float4 *>(src + 4);
                } else { // ragged rows (unaligned, or the row ends inside these eight features)
                    float t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = fg + j < Msrc ? src[j] : 0.f;
                    a = float4{t[0], t[1], t[2], t[3]};
                    b = float4{t[4], t[5], t[6], t[7]};
                }
            }
            float *d = &tile[pt][f0];
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
            d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        }
        __syncthreads();
        const int plane = threadIdx.x >> 7, row = threadIdx.x & 127;
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = tile[plane * 8 + j][row] * scale;
            const _Float16 a = (_Float16)x;
            hi[j] = a;
            lo[j] = (_Float16)(x - (float)a);
        }
        blocks[(blk * 2 + 0) * 256 + threadIdx.x] = hi;
        blocks[(blk * 2 + 1) * 256 + threadIdx.x] = lo;
        __syncthreads();
    }
}

// The accumulation kernel takes gamma and beta of a stage from ONE 256-byte record (gamma x 32 | beta x 32, zeros beyond N), which
// each wave moves into LDS by DMA three steps ahead -- a scalar or vector load at the point of use would pay the HBM latency
// of a cold, once-read array in every step.  In a sweep the per-point kernel writes the records itself
// (agpl_fused_point_kernel, agpl_ops.hip); acc_prep_kernel is the stand-alone form (agpl_accumulate_split, the Gibbs pass):
// records + max gamma, one atomic per workgroup (8192 per-wave atomics on one word cost ~90 us, round 3).  The accumulation
// kernel multiplies gamma by s_B = 2^e_B (exact) as it converts.
__global__ __launch_bounds__(256) void acc_prep_kernel(int64_t N, int64_t Npad, int L, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float *__restrict__ gb,
                                                        unsigned *__restrict__ scal) {
    __shared__ unsigned red[2][4];
    const int64_t total = (int64_t)L * Npad;
    unsigned m = 0u, bd = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = i / Npad, n = i - l * Npad;
        const bool in = n < N;
        const float gv = in ? gamma[l * N + n] : 0.f;
        const unsigned gbits = __float_as_uint(gv), ab = gbits & 0x7FFFFFFFu;
        const bool bad = ab >= 0x7F800000u || ((gbits >> 31) && ab != 0u); // inf, NaN or negative
        if (bad) bd = max(bd, (unsigned)min((int64_t)0x7FFFFFFE, l * N + n) + 1u);
        m = max(m, bad ? 0u : ab);
        float *rec = gb + (l * (Npad / 32) + n / 32) * 64 + (n & 31);
        rec[0] = gv;
        rec[32] = in ? beta[l * N + n] : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m = max(m, (unsigned)__shfl_xor((int)m, o));
        bd = max(bd, (unsigned)__shfl_xor((int)bd, o));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = m;
        red[1][threadIdx.x >> 6] = bd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            m = max(m, red[0][w]);
            bd = max(bd, red[1][w]);
        }
        if (m) atomicMax(scal, m);
        if (bd) atomicMax(scal + 1, bd);
    }
}

__device__ __forceinline__ int acc_scale_exp(unsigned gmax) { // e_B: 2^e_B max(gamma) in [1/2, 1)
    int eB = gmax ? 126 - (int)(gmax >> 23) : 0;
    return eB < -60 ? -60 : (eB > 60 ? 60 : eB);
}

#ifdef AGPL_QTRACE // diagnostic build (make QTRACE=1): per-wave cycle sums of the stage loop, tools/qtrace.py
__device__ unsigned long long g_qtrace[64 * 16 * 8];
constexpr bool kTrace = true;
#else
constexpr bool kTrace = false;
#endif

// ------------------------------------------------------------------------------------------------
// syrk_strip_kernel: the same tile, image and slabs with the B operand kept OUT of LDS.
// A wave owns a column strip of the tile -- all 256 rows x 32 columns (16 x 2 accumulators of 16 x 16) -- so the B
// fragments it multiplies are ITS OWN: lane l loads the granules (feature 16 cb + (l & 15) of the strip, points of k-group
// l >> 4) of its two column blocks straight from the image into registers (the image granule IS the MFMA B fragment),
// scales them by gamma and re-splits them there; nothing of B is stored to or read from LDS, no wave converts what another
// one multiplies, and the B bytes are fetched once per CU.  LDS holds only A: a ring of four 32-point steps (32 KB each),
// filled by DMA three steps ahead, plus each wave's private copies of the steps' gamma | beta records.
// Per step and wave: 32 A-fragment reads (one 16-row block ahead of its six MFMAs), 96 MFMAs, 5 DMA pieces, 4 granule
// loads, 48 v_fma_mix.  The granule loads are inline asm: beside an LDS-DMA in flight the compiler drains the whole queue
// (vmcnt(0)) at the first use of an ordinary load; here the one wait of a step is `vmcnt(5)` = everything but this step's
// DMA pieces.  Diagonal tiles: strip c needs row blocks i >= 2 c; the waves of a SIMD (w, w + 4) take the strips (s, 7 - s):
// 34 blocks per SIMD against 64 of an off-diagonal tile.
// ------------------------------------------------------------------------------------------------
// Measurement build (make L2PROBE=1): every step re-reads one of the slice's first four steps, i.e. every operand load hits the
// XCD's L2 -- wrong sums, same instruction stream and data statistics.  The upper bound of what ANY scheme of sharing panels
// between the workgroups of a slice could buy (round 4: C2 6.87 -> 6.42 ms, M = 1024 30.6 -> 26.6 ms).
// (make NODMA: -DAGPL_SYRK_NODMA: no operand load at all after the first steps -- the step loop's compute side alone.)
#ifdef AGPL_SYRK_L2PROBE
#define AGPL_PROBE_T(t_) ((t_) & 3)
#else
#define AGPL_PROBE_T(t_) (t_)
#endif
#ifdef AGPL_SYRK_NODMA
#define AGPL_PROBE_LOAD(t_) ((t_) < 3)
#else
#define AGPL_PROBE_LOAD(t_) true
#endif
#ifndef AGPL_S_PRE
#define AGPL_S_PRE 3 // row blocks of A-fragment lead (2, 3, 4 measured: 3.73 / 3.69 / 3.74 ms at N = 4e6, M = 512)
#endif
constexpr int kStepBytes = 32768;                  // A image of one 32-point step: [slice 2][(128-row block, hi | lo) 4][4096]
constexpr int kRing = 4;                           // steps in the ring
constexpr int kRecBytes = 256;                     // one gamma | beta record
constexpr int kStripLds = kRing * kStepBytes + 8 * kRing * kRecBytes;

template <bool DIAG>
__device__ __forceinline__ void syrk_strip_body(unsigned char *smem_raw, int64_t N, int64_t Npad, int M, int nsplit, int chunk, int nbig, int small, int l,
                                                int s, int I, int J, const unsigned char *__restrict__ image,
                                                const float *__restrict__ gb_all, const unsigned *__restrict__ scal,
                                                float *__restrict__ slabG, float *__restrict__ slabg) {
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0..7
    const int nb = M / BS;
    // strip of this wave: on a diagonal tile the two waves of a SIMD (w, w + 4) take strips (s, 7 - s)
    const int c = DIAG ? ((wave & 4) ? 7 - (wave & 3) : (wave & 3)) : wave;
    const int i0 = DIAG ? 2 * c : 0; // first needed 16-row block
    const AccImageHeader *hdr = reinterpret_cast<const AccImageHeader *>(image);
    const h8 *blocks = reinterpret_cast<const h8 *>(image + sizeof(AccImageHeader));
    const int eA = hdr->scale_exp;
    const int eB = acc_scale_exp(scal[0]);
    const float sB = __uint_as_float((unsigned)(127 + eB) << 23); // gamma is scaled as it is used (exact)

    int64_t nbeg, nend;
    agpl_slice_range(s, chunk, nbig, small, N, nbeg, nend);
    const int nstep = (int)((nend - nbeg + kStagePts - 1) / kStagePts);
    const int64_t ps0 = nbeg / 16;
    const int64_t slice_pitch = (int64_t)nb * 2 * 256; // h8 units between consecutive point slices

    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int kg = ln >> 4, lr = ln & 15;
    // A pieces of this wave: 128-row block rbS, quarter qd, both slices, hi and lo
    const int rbS = (wave >> 2) & 1, qd = wave & 3;
    const h8 *a_src = blocks + ((ps0 * nb + 2 * I + rbS) * 2) * 256 + qd * 64 + lane;
    const int a_dst = rbS * 2 * 4096 + qd * 1024;
    // B granules of this lane: strip rows 32 c + 16 cb + lr of panel J, k-group kg = (slice kg >> 1, plane kg & 1)
    const int R0 = 32 * c + lr;
    const h8 *b_src = blocks + (((ps0 + (kg >> 1)) * nb + 2 * J + (R0 >> 7)) * 2) * 256 + (kg & 1) * 128 + (R0 & 127);
    // gamma | beta records: every wave keeps its own copy of a step's record (no wave waits for another's DMA)
    const float *gb_src = gb_all + ((int64_t)l * (Npad / 32) + nbeg / 32) * 64 + lane;
    float *gbuf = reinterpret_cast<float *>(smem_raw + kRing * kStepBytes + wave * kRing * kRecBytes);

    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gacc0 = 0.f, gacc1 = 0.f;
    u32x4 rh0 = {0, 0, 0, 0}, rl0 = rh0, rh1 = rh0, rl1 = rh0; // raw granules of the two column blocks (hi, lo), then their converted form
    h8 bh0, bl0, bh1, bl1;    // the B fragments in use

#define AGPL_S_DMA(t_, k_) /* piece k_ = 0..3 of step t_: (slice k_ >> 1, hi / lo k_ & 1) */                    \
    do {                                                                                                        \
        unsigned char *d_ = smem_raw + ((t_) & (kRing - 1)) * kStepBytes + ((k_) >> 1) * 16384 + a_dst + ((k_) & 1) * 4096; \
        const h8 *src_ = a_src + (int64_t)(2 * AGPL_PROBE_T(t_) + ((k_) >> 1)) * slice_pitch + ((k_) & 1) * 256; \
        if (AGPL_PROBE_LOAD(t_)) __builtin_amdgcn_global_load_lds(src_, (lds_void *)d_, 16, 0, 0);              \
    } while (0)
#define AGPL_S_DMAG(t_)                                                                                         \
    if (AGPL_PROBE_LOAD(t_))                                                                                    \
    __builtin_amdgcn_global_load_lds(gb_src + (int64_t)(t_) * 64, (lds_void *)(gbuf + ((t_) & (kRing - 1)) * 64), 4, 0, 0)
    // The compiler does not see these loads in the memory queue (it believes their results are there at once): the waits
    // are written out below, and the destinations are read-write operands of every statement that touches them and are never
    // behind a condition -- a copy of an in-flight destination (the phi of a conditional load) would copy stale registers
#define AGPL_S_LOADB(t_)                                                                                        \
    do {                                                                                                        \
        const h8 *src_ = b_src + (int64_t)(2 * AGPL_PROBE_T(t_)) * slice_pitch;                                 \
        const h8 *srcl_ = src_ + 256; /* the lo block follows the hi block; column block 1 = 16 rows = 256 bytes on */ \
        if (AGPL_PROBE_LOAD(t_))                                                                                \
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"                   \
                     "global_load_dwordx4 %2, %4, off offset:256\n\tglobal_load_dwordx4 %3, %5, off offset:256"  \
                     : "+v"(rh0), "+v"(rl0), "+v"(rh1), "+v"(rl1)                                               \
                     : "v"(src_), "v"(srcl_)                                                                    \
                     : "memory");                                                                               \
    } while (0)
#define AGPL_S_CVT(RH_, RL_, g0_, g1_, b0_, b1_, GA_)                                                           \
    do {                                                                                                        \
        float x0_, x1_;                                                                                         \
        AGPL_Q_JOIN2(RH_, RL_, x0_, x1_);                                                                       \
        if (DIAG) {                                                                                             \
            AGPL_E_GFMA(GA_, b0_, x0_);                                                                         \
            AGPL_E_GFMA(GA_, b1_, x1_);                                                                         \
        }                                                                                                       \
        AGPL_E_SPLIT2(x0_, g0_, x1_, g1_, RH_, RL_);                                                            \
    } while (0)
    // a quarter of the conversion: points 4 q .. 4 q + 3 of the lane's k-group, both column blocks; tt_ = the step the raw
    // granules belong to (its record gives gamma, and beta on a diagonal tile; keepf zeroes the beta of a duplicate)
#define AGPL_S_CVTQ(q_, tt_, keepf_)                                                                            \
    do {                                                                                                        \
        const float *gq_ = gbuf + ((tt_) & (kRing - 1)) * 64 + 8 * kg + 4 * ((q_) & 1);                         \
        float4 g4_ = *reinterpret_cast<const float4 *>(gq_);                                                    \
        g4_.x *= sB, g4_.y *= sB, g4_.z *= sB, g4_.w *= sB;                                                     \
        float4 b4_ = {0.f, 0.f, 0.f, 0.f};                                                                      \
        if (DIAG) {                                                                                             \
            b4_ = *reinterpret_cast<const float4 *>(gq_ + 32);                                                  \
            if (!(keepf_)) b4_ = float4{0.f, 0.f, 0.f, 0.f};                                                    \
        }                                                                                                       \
        if ((q_) == 0) {                                                                                        \
            AGPL_S_CVT(rh0.x, rl0.x, g4_.x, g4_.y, b4_.x, b4_.y, gacc0);                                        \
            AGPL_S_CVT(rh0.y, rl0.y, g4_.z, g4_.w, b4_.z, b4_.w, gacc0);                                        \
            AGPL_S_CVT(rh1.x, rl1.x, g4_.x, g4_.y, b4_.x, b4_.y, gacc1);                                        \
            AGPL_S_CVT(rh1.y, rl1.y, g4_.z, g4_.w, b4_.z, b4_.w, gacc1);                                        \
        } else {                                                                                                \
            AGPL_S_CVT(rh0.z, rl0.z, g4_.x, g4_.y, b4_.x, b4_.y, gacc0);                                        \
            AGPL_S_CVT(rh0.w, rl0.w, g4_.z, g4_.w, b4_.z, b4_.w, gacc0);                                        \
            AGPL_S_CVT(rh1.z, rl1.z, g4_.x, g4_.y, b4_.x, b4_.y, gacc1);                                        \
            AGPL_S_CVT(rh1.w, rl1.w, g4_.z, g4_.w, b4_.z, b4_.w, gacc1);                                        \
        }                                                                                                       \
    } while (0)
#define AGPL_S_TAKEB() /* the converted granules become the fragments of the next step */                      \
    do {                                                                                                        \
        bh0 = __builtin_bit_cast(h8, rh0);                                                                      \
        bl0 = __builtin_bit_cast(h8, rl0);                                                                      \
        bh1 = __builtin_bit_cast(h8, rh1);                                                                      \
        bl1 = __builtin_bit_cast(h8, rl1);                                                                      \
    } while (0)

    // fragment slot of (row block i, part) inside a step: [slice kg >> 1][(rb = i >> 3, hl)][plane kg & 1][row]
    const int fa = (kg >> 1) * 1024 + (kg & 1) * 128 + lr;
#define AGPL_S_AOFF(i_) (fa + ((i_) >> 3) * 512 + ((i_) & 7) * 16)
    // On a DIAGONAL tile the B panel is the A panel: the raw granules of the strip are row blocks 2 c, 2 c + 1 of the A
    // image already in LDS -- read from there (step t_'s slot), not from memory: a diagonal tile moves 32 KB per step through
    // the CU's vector-memory path instead of 64 (that path, ~15-20 bytes per clock and CU, is what bounds this kernel)
#define AGPL_S_LDSB(t_)                                                                                         \
    do {                                                                                                        \
        const u32x4 *sn_ = reinterpret_cast<const u32x4 *>(smem_raw + ((t_) & (kRing - 1)) * kStepBytes);       \
        rh0 = sn_[AGPL_S_AOFF(2 * c)];                                                                          \
        rl0 = sn_[256 + AGPL_S_AOFF(2 * c)];                                                                    \
        rh1 = sn_[AGPL_S_AOFF(2 * c + 1)];                                                                      \
        rl1 = sn_[256 + AGPL_S_AOFF(2 * c + 1)];                                                                \
    } while (0)

    // ---- prologue: records and A of steps 0..2; B of step 0 converted; (off the diagonal) B of step 1 in flight
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < nstep) {
            AGPL_S_DMAG(t);
            AGPL_S_DMA(t, 0);
            AGPL_S_DMA(t, 1);
            AGPL_S_DMA(t, 2);
            AGPL_S_DMA(t, 3);
        }
    if (!DIAG) AGPL_S_LOADB(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (DIAG) {
        __builtin_amdgcn_s_barrier(); // every wave's pieces of step 0 are in LDS
        AGPL_S_LDSB(0);
    }
    AGPL_S_CVTQ(0, 0, true);
    AGPL_S_CVTQ(1, 0, true);
    AGPL_S_TAKEB();
    if (!DIAG) AGPL_S_LOADB(nstep > 1 ? 1 : 0);

    [[maybe_unused]] unsigned long long q0 = 0, q1 = 0, qw = 0, qa = 0, qb = 0, qr0 = 0;
    if (kTrace) {
        q0 = __builtin_amdgcn_s_memtime();
        qr0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int t = 0; t < nstep; ++t) {
        if (kTrace) qa = __builtin_amdgcn_s_memtime();
        if (!DIAG) __builtin_amdgcn_s_barrier(); // every wave's pieces of step t have landed: each waited for them a step ago
        if (kTrace && !DIAG) qw += __builtin_amdgcn_s_memtime() - qa;
        const bool more = t + 1 < nstep; // (else the conversion below works on a duplicate: its beta reads as zero)
        const int tl = t + 2 < nstep ? t + 2 : nstep - 1;
        const h8 *st = reinterpret_cast<const h8 *>(smem_raw + (t & (kRing - 1)) * kStepBytes);
        // hook h sits behind the MFMAs of row block h: the step's staging work in pieces
        //   0..3  one DMA piece of step t + 3 each (hook 0: its record as well)
        //   5     wait for everything but those five pieces: the raw granules of step t + 1 (off the diagonal), the A pieces
        //         of step t + 2
        //   6, 8  half of the conversion of step t + 1's granules each (a diagonal tile reads them from its A slot at 6)
        //   10    they are set aside as the next fragments; off the diagonal the loads of step t + 2 go out
#define AGPL_S_HOOK(h_)                                                                                         \
    do {                                                                                                        \
        if (!DIAG) {                                                                                            \
            if ((h_) < 4 && t + 3 < nstep) {                                                                    \
                if ((h_) == 0) AGPL_S_DMAG(t + 3);                                                              \
                AGPL_S_DMA(t + 3, h_);                                                                          \
            }                                                                                                   \
            if ((h_) == 5) { /* (the last three steps issue no DMA pieces: everything outstanding is older) */ \
                __builtin_amdgcn_sched_barrier(0);                                                              \
                if (t + 3 < nstep) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                             \
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
                __builtin_amdgcn_sched_barrier(0);                                                              \
            }                                                                                                   \
            if ((h_) == 6) AGPL_S_CVTQ(0, t + 1, more);                                                         \
            if ((h_) == 8) AGPL_S_CVTQ(1, t + 1, more);                                                         \
        } else {                                                                                                \
            /* a diagonal tile: ONE barrier per step, in its middle.  Before it a wave waits for its own pieces of */ \
            /* step t + 1 (everything but the five of step t + 2); behind it A(t + 1) is in LDS for every wave -- for the */ \
            /* next step's fragment reads and for this wave's raw B granules -- and every wave is through with step */ \
            /* t - 1, whose slot takes the pieces of step t + 3: two full steps of flight for every piece */     \
            if ((h_) == 7) {                                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                              \
                if (kTrace) qb = __builtin_amdgcn_s_memtime();                                                  \
                if (t + 2 < nstep) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                             \
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
                if (kTrace) q1 += __builtin_amdgcn_s_memtime() - qb;                                            \
                if (kTrace) qb = __builtin_amdgcn_s_memtime();                                                  \
                __builtin_amdgcn_s_barrier();                                                                   \
                if (kTrace) qw += __builtin_amdgcn_s_memtime() - qb;                                            \
                __builtin_amdgcn_sched_barrier(0);                                                              \
                AGPL_S_LDSB(more ? t + 1 : t); /* (behind the last step: a duplicate, finite) */                \
            }                                                                                                   \
            if ((h_) >= 8 && (h_) < 12 && t + 3 < nstep) {                                                      \
                if ((h_) == 8) AGPL_S_DMAG(t + 3);                                                              \
                AGPL_S_DMA(t + 3, (h_) - 8);                                                                    \
            }                                                                                                   \
            if ((h_) == 9) AGPL_S_CVTQ(0, t + 1, more);                                                         \
            if ((h_) == 11) AGPL_S_CVTQ(1, t + 1, more);                                                        \
        }                                                                                                       \
    } while (0)
        h8 nbh0, nbl0, nbh1, nbl1;
        // A fragments kPre row blocks ahead of the six MFMAs that use them (a wave alone must cover the LDS latency: with
        // one block of lead a wave ran at ~46 cycles per MFMA -- in-kernel stamps); the reads are unconditional, only the
        // MFMAs of a diagonal tile's unneeded blocks (i < i0) are skipped
        constexpr int kPre = AGPL_S_PRE;
        h8 af[kPre + 1][2];
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            af[j][0] = st[AGPL_S_AOFF(j)];
            af[j][1] = st[256 + AGPL_S_AOFF(j)];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i + kPre < 16) {
                af[(i + kPre) % (kPre + 1)][0] = st[AGPL_S_AOFF(i + kPre)];
                af[(i + kPre) % (kPre + 1)][1] = st[256 + AGPL_S_AOFF(i + kPre)];
            }
            if (!DIAG || i >= i0) {
                const h8 ahc = af[i % (kPre + 1)][0], alc = af[i % (kPre + 1)][1];
                acc[i][0] = mfma32(ahc, bh0, acc[i][0]);
                acc[i][1] = mfma32(ahc, bh1, acc[i][1]);
                acc[i][0] = mfma32(ahc, bl0, acc[i][0]);
                acc[i][1] = mfma32(ahc, bl1, acc[i][1]);
                acc[i][0] = mfma32(alc, bh0, acc[i][0]);
                acc[i][1] = mfma32(alc, bh1, acc[i][1]);
            }
            AGPL_S_HOOK(i);
            if (i == (DIAG ? 12 : 10)) { // the converted granules are kept aside until this step's MFMAs are through with the old ones
                nbh0 = __builtin_bit_cast(h8, rh0);
                nbl0 = __builtin_bit_cast(h8, rl0);
                nbh1 = __builtin_bit_cast(h8, rh1);
                nbl1 = __builtin_bit_cast(h8, rl1);
                if (!DIAG) AGPL_S_LOADB(tl); // unconditional (clamped: the last steps re-load the last one)
            }
        }
#undef AGPL_S_HOOK
        bh0 = nbh0;
        bl0 = nbl0;
        bh1 = nbh1;
        bl1 = nbl1;
    }
#ifdef AGPL_QTRACE
    if (blockIdx.x < 64 && lane == 0) {
        unsigned long long *o = g_qtrace + ((size_t)blockIdx.x * 16 + wave) * 8;
        o[0] = q1;                                   // waiting for the memory queue
        o[1] = __builtin_amdgcn_s_memtime() - q0;    // loop cycles
        o[2] = qw;                                   // waiting at the barrier
        o[5] = __builtin_amdgcn_s_memrealtime() - qr0;
        o[6] = (unsigned long long)nstep | ((unsigned long long)(DIAG ? 1 : 0) << 32) | ((unsigned long long)c << 40);
        o[7] = o[1];
    }
#endif
    // the clamped loads of the last steps land in registers that must not look dead (and be reused) while they are in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(rh0), "v"(rl0), "v"(rh1), "v"(rl1));
#undef AGPL_S_DMA
#undef AGPL_S_DMAG
#undef AGPL_S_LOADB
#undef AGPL_S_CVT
#undef AGPL_S_CVTQ
#undef AGPL_S_TAKEB
#undef AGPL_S_LDSB
#undef AGPL_S_AOFF

    // ---- slabs: [l][128-pair][slice][128 x 128] float32; the accumulators carry (s_A phi)(s_B gamma s_A phi)'
    const float unscale = __uint_as_float((unsigned)(127 - (2 * eA + eB)) << 23);
    const int npairs = nb * (nb + 1) / 2;
    const int bj = 2 * J + (c >> 2);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) { // the two 128-row blocks of the tile
        const int bi = 2 * I + hb;
        if (DIAG && bi < bj) continue; // the 128 x 128 block above the diagonal is never read back
        const int p128 = bi * (bi + 1) / 2 + bj;
        float *slab = slabG + (((int64_t)l * npairs + p128) * nsplit + s) * (int64_t)(BS * BS);
#pragma unroll
        for (int ii = 0; ii < 8; ++ii)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * ii + 4 * kg + r;
                    const int col = (c & 3) * 32 + 16 * cb + lr;
                    slab[row * BS + col] = acc[8 * hb + ii][cb][r] * unscale;
                }
    }
    if (DIAG) {
        // g of the strip's 32 rows: the four k-groups of a row sit in lanes lr, lr + 16, lr + 32, lr + 48
        gacc0 += __shfl_xor(gacc0, 16);
        gacc0 += __shfl_xor(gacc0, 32);
        gacc1 += __shfl_xor(gacc1, 16);
        gacc1 += __shfl_xor(gacc1, 32);
        const float g1s = __shfl(gacc1, (lane - 16) & 63); // (outside the branch: a shuffle reads live lanes only)
        if (lane < 32) {
            const int rr = 32 * c + lane; // row of the panel: column block lane >> 4, feature lane & 15
            slabg[(((int64_t)l * nb + 2 * I + (rr >> 7)) * nsplit + s) * BS + (rr & 127)] =
                (lane < 16 ? gacc0 : g1s) * __uint_as_float((unsigned)(127 - eA) << 23);
        }
    }
}


// Round 6 built one complete alternative to this kernel and removed it again -- record: profiles/NOTES_r06.md,
// profiles/r06_ab_syrk_quad.jsonl, commit "syrk_quad_kernel v3".  syrk_quad_kernel: ONE wave per SIMD owning the two strips (w, 7 - w)
// the two waves of a SIMD share here (256 x 64 = 256 accumulator registers in AGPRs; every A fragment read once per SIMD; fragments
// three row blocks ahead across the step boundary; the step's conversions and DMA issue cut into <= 2-instruction slices behind
// each MFMA; LDS reads by inline asm with counted waits).  G and g bit-identical; 10.3 ms against 6.5 ms at C2, 16.2 against 10.7 at
// N = 5e6, M = 1024.  Its bare probe (no operand loads, no conversions: MFMAs + fragment reads + barrier, a near-ideal instruction
// stream) still took 7.7 ms = 2.1 x the matrix-pipe floor: one wave per SIMD does not issue v_mfma_f32_16x16x32_f16 back to back on
// this device -- the second wave of a SIMD is what fills the pipe, whatever the first one's stream looks like.  Pinning the order
// of THIS kernel's row blocks with sched_barrier (the compiler sinks the fragment reads to within four MFMAs of their use) is worth
// 0-1 % (profiles/r06_ab_syrk_pinned.jsonl): with two waves per SIMD the partner covers it.
__global__ __launch_bounds__(512, 2) void syrk_strip_kernel(int64_t N, int64_t Npad, int M, int nlat, int npairs2, int nsplit, int chunk, int nbig, int small,
                                                            const unsigned char *__restrict__ image,
                                                            const float *__restrict__ gb_all,
                                                            const unsigned *__restrict__ scal,
                                                            float *__restrict__ slabG, float *__restrict__ slabg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // workgroup -> (slice, latent, tile): id % 8 labels the XCD (speed only); an XCD walks the slices s = 8 t + xcd, and for every slice
    // all latents' tiles before the next slice (round 6: latent-major order streamed the image L times -- C4, L = 10, fetched 5.2 x
    // its image -- and put every latent's short tail slices in the middle of the launch)
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int per_s = nlat * npairs2;
    const int s = (jj / per_s) * 8 + xcd;
    const int rem = jj % per_s;
    const int l = rem / npairs2;
    const int p2 = rem - l * npairs2;
    if (s >= nsplit) return;
    const int nb2 = M / kPanel;
    const int noff = nb2 * (nb2 - 1) / 2;
    int I, J;
    if (p2 < noff) {
        I = 1;
        while ((I + 1) * I / 2 <= p2) ++I;
        J = p2 - I * (I - 1) / 2;
    } else {
        I = J = p2 - noff;
    }
    if (I == J) syrk_strip_body<true>(smem_raw, N, Npad, M, nsplit, chunk, nbig, small, l, s, I, J, image, gb_all, scal, slabG, slabg);
    else syrk_strip_body<false>(smem_raw, N, Npad, M, nsplit, chunk, nbig, small, l, s, I, J, image, gb_all, scal, slabG, slabg);
}

} // namespace

#ifdef AGPL_QTRACE
extern "C" __attribute__((visibility("default"))) int agpl_debug_qtrace(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qtrace), sizeof(g_qtrace));
}
#endif


// internal (agpl_plan.hip): bytes of the accumulate image (256-byte header + 4 KB blocks) for N points, M features
int64_t agpl_accumulate_image_bytes(int64_t N, int32_t M) {
    if (N <= 0 || M <= 0 || M % BS) return 0;
    const int64_t nps = ((N + kStagePts - 1) / kStagePts) * 2; // whole 32-point stages
    return (int64_t)sizeof(AccImageHeader) + nps * (M / BS) * 2 * 4096;
}

// internal: max |Phi| (bit pattern of a non-negative float) after checking every value finite and |x| < limit; otherwise
// AGPL_ERR_DOMAIN naming the first offending (point, feature).  One stream synchronisation; features are static, so this
// runs once per image, not per sweep.  `what` names the image in the message.
int32_t agpl_feature_range_check(agpl_ctx *ctx, int64_t N, int32_t M, const float *Phi, float limit, const char *what,
                                 unsigned *max_bits_out) {
    int32_t rc = agpl_ws2_reserve(ctx, 4096);
    if (rc) return rc;
    if ((uintptr_t)Phi & 15) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "the feature matrix must be 16-byte aligned");
    unsigned *mx = (unsigned *)ctx->ws2 + 512; // words 512.. of the small scratch (the info flags live below)
    AGPL_HIP(ctx, hipMemsetAsync(mx, 0, 16, ctx->stream));
    const int64_t n = N * (int64_t)M, n4 = n / 4; // (any M: the rows are contiguous, the last n % 4 floats are taken one by one)
    absmax_kernel<<<4096, 256, 0, ctx->stream>>>(n4, (int)(n - 4 * n4), reinterpret_cast<const float4 *>(Phi), mx);
    AGPL_LAUNCH_CHECK(ctx);
    unsigned hmx = 0, hlim;
    memcpy(&hlim, &limit, 4);
    AGPL_HIP(ctx, hipMemcpyAsync(&hmx, mx, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hmx >= 0x7F800000u || hmx >= hlim) {
        unsigned long long *bad = (unsigned long long *)(mx + 2), hbad = ~0ull;
        AGPL_HIP(ctx, hipMemsetAsync(bad, 0xFF, 8, ctx->stream));
        find_bad_kernel<<<4096, 256, 0, ctx->stream>>>(n, Phi, limit, bad);
        AGPL_HIP(ctx, hipMemcpyAsync(&hbad, bad, 8, hipMemcpyDeviceToHost, ctx->stream));
        AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const long long pt = (long long)(hbad / (unsigned long long)M), ft = (long long)(hbad % (unsigned long long)M);
        if (hmx >= 0x7F800000u && !(limit < __builtin_inff()))
            AGPL_FAIL(ctx, AGPL_ERR_DOMAIN, "feature value is not finite (point %lld, feature %lld)", pt, ft);
        AGPL_FAIL(ctx, AGPL_ERR_DOMAIN,
                  "feature value is not finite or its magnitude is not below %g, the range of %s (point %lld, feature %lld)",
                  (double)limit, what, pt, ft);
    }
    *max_bits_out = hmx;
    return AGPL_OK;
}

// internal (also agpl_plan.hip): the scale exponent e_A for a feature matrix whose max |x| has the float32 bit pattern `hmx`:
// 2^e_A max|Phi| in [2^13, 2^14) -- hi stays finite, and lo = f16(x - hi) is a float16 normal for |x| down to 2^-17 max|Phi|.
// |e_A| <= 30 (2 e_A + e_B must stay an exponent of a normal float32, e_B in [-60, 60]): a larger max|Phi| (>= 2^44) would turn
// into inf, and one below 2^-24 would leave fewer than ten of those seventeen octaves -- both AGPL_ERR_DOMAIN; between 2^-24 and
// 2^-17 the exponent stays at 30 and the image sits up to seven octaves lower in the float16 range.
int32_t agpl_image_scale_exp(agpl_ctx *ctx, unsigned hmx, int *eA_out) {
    int eA = 0;
    if (hmx) {
        const int ex = (int)(hmx >> 23) - 127; // max_abs in [2^ex, 2^(ex+1))   (a subnormal max reads ex = -127)
        eA = 13 - ex;
        if (eA > 37 || eA < -30) {
            float mx;
            memcpy(&mx, &hmx, 4);
            AGPL_FAIL(ctx, AGPL_ERR_DOMAIN,
                      "max |Phi| = %g is outside the range the split-float16 images can be scaled for (2^-24 .. 2^44): rescale the "
                      "features", (double)mx);
        }
        if (eA > 30) eA = 30;
    }
    *eA_out = eA;
    return AGPL_OK;
}
// internal (also agpl_plan.hip): the image of 2^eA Phi; the features have been range-checked
// (Msrc <= M: the features the rows of Phi hold; the image's features beyond them are zero)
int32_t agpl_accumulate_image_build(agpl_ctx *ctx, int64_t N, int32_t M, int32_t Msrc, const float *Phi, int eA, unsigned hmx,
                                    void *image_out) {
    float max_abs;
    memcpy(&max_abs, &hmx, 4);
    const float scale = ldexpf(1.f, eA);
    const int64_t nps = ((N + kStagePts - 1) / kStagePts) * 2;
    accumulate_image_kernel<<<16384, 256, 0, ctx->stream>>>(N, M, Msrc, nps, scale, eA, max_abs, Phi, (unsigned char *)image_out);
    AGPL_LAUNCH_CHECK(ctx);
    if (ctx->checked_image == image_out) ctx->checked_image = nullptr; // (rebuilt in place: its header is looked at again)
    return AGPL_OK;
}

// internal (agpl_accumulate_impl): prep + accumulation kernel; slabs as agpl_mfma.hip lays them out.
// gb: 2 L Npad floats (the gamma | beta records), Npad = N rounded up to 32 (+ 32); scal: 2 words (max gamma bits, 1 + index
// of a gamma that is negative or not finite).  records_ready: both are filled already; gamma / beta are not read.
int32_t agpl_syrk_image_launch(agpl_ctx *ctx, int64_t N, int64_t Npad, int32_t M, int32_t L, const void *image,
                               const float *gamma, const float *beta, float *gb, unsigned *scal, float *slabG,
                               float *slabg, int ns, int chunk, int nbig, int small, bool records_ready) {
    if (M % kPanel) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "the image accumulation needs M %% 256 == 0 (M = %d)", M);
    if (ctx->checked_image != image || ctx->checked_image_N != N || ctx->checked_image_M != M) {
        // the header, once per (image, N, M): an image of another (N, M), or a buffer that never was one, would otherwise give
        // out-of-range DMA reads and a garbage scale.  One small copy and one synchronisation, at the first sweep only.
        AccImageHeader h;
        AGPL_HIP(ctx, hipMemcpyAsync(&h, image, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h.magic != kImageMagic || h.N != N || h.M != M)
            AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT,
                      "not an accumulate image of this problem (agpl_plan_create): header says N = %lld, "
                      "M = %d, magic %#x; the call has N = %lld, M = %d",
                      (long long)h.N, (int)h.M, (unsigned)h.magic, (long long)N, (int)M);
        ctx->checked_image = image;
        ctx->checked_image_N = N;
        ctx->checked_image_M = M;
    }
    const int nb2 = M / kPanel;
    const int npairs2 = nb2 * (nb2 + 1) / 2;
    const int64_t nwg = (int64_t)L * npairs2 * ((ns + 7) / 8) * 8;
    if (nwg > 0x7fffffffLL) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "problem too large for one launch");
    if (!records_ready) { // (a sweep's per-point kernel has written the records and max gamma already)
        AGPL_HIP(ctx, hipMemsetAsync(scal, 0, 2 * sizeof(unsigned), ctx->stream));
        int64_t nblk = agpl_cdiv((int64_t)L * Npad, 256);
        if (nblk > 1024) nblk = 1024;
        acc_prep_kernel<<<(unsigned)nblk, 256, 0, ctx->stream>>>(N, Npad, L, gamma, beta, gb, scal);
        AGPL_LAUNCH_CHECK(ctx);
    }
    if (!ctx->strip_attr) { // once per context
        AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&syrk_strip_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kStripLds));
        ctx->strip_attr = 1;
    }
    syrk_strip_kernel<<<(unsigned)nwg, 512, kStripLds, ctx->stream>>>(N, Npad, M, L, npairs2, ns, chunk, nbig, small, (const unsigned char *)image, gb,
                                                                     scal, slabG, slabg);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}
