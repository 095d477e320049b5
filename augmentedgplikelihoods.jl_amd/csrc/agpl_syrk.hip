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
// syrk_image8_kernel: one 512-thread workgroup (8 waves, two per SIMD, one workgroup per CU) owns a 256 x 256 tile of the
// lower triangle of G for one slice of 4096 points (one f32 accumulation run, one slab set -- the slabs and the fixed-order
// float64 reduction behind them are those of agpl_mfma.hip).  Per 32-point stage:
//   A = rows of panel I:  image blocks moved HBM -> LDS by the DMA path (global_load_lds_dwordx4), no VGPRs, no VALU;
//   B = gamma_n * (rows of panel J):  every thread loads TWO granules (hi and lo: 4 x 16 B) of the same image into registers,
//       rebuilds x = hi + lo (exact in float32), forms y = (s_B gamma_n) x and splits it into hi / lo again
//       (3 v_fma_mix per value) and stores the 16-byte results into the B half of the next stage's LDS slot -- the granule is
//       already the MFMA fragment, nothing is transposed;
//   G_tile += A B'  as  hi hi' + hi lo' + lo hi'  with v_mfma_f32_16x16x32_f16 (96 per wave and stage, 128 x 64 per wave).
// Compared with syrk_split_kernel (agpl_mfma.hip: 128 x 128 tiles, both panels converted from float32 every stage by every
// tile pair): a quarter of the conversions per flop, half the operand bytes per flop, no register staging of A.
// g = Phi beta rides the B conversion of the diagonal tiles (x is the float32 feature there).
#include <cstdlib>

#include "agpl_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BS = 128;       // feature rows per image block / per slab tile (the slab layout of agpl_mfma.hip)
constexpr int kChunk = 4096;  // points per slice (must match agpl_mfma.hip)
constexpr int kPanel = 256;   // feature rows per operand panel
constexpr int kStagePts = 32; // points per stage (one MFMA K)
constexpr int kSliceBytes = 8 * 4096;       // LDS bytes of one 16-point slice of a stage: A (rb, hl) x 4 | B (rb, hl) x 4
constexpr int kSlot = 2 * kSliceBytes;      // one stage
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
__global__ __launch_bounds__(256) void absmax_kernel(int64_t n4, const float4 *__restrict__ x, unsigned *__restrict__ out) {
    unsigned m = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        m = max(m, __float_as_uint(v.x) & 0x7FFFFFFFu);
        m = max(m, __float_as_uint(v.y) & 0x7FFFFFFFu);
        m = max(m, __float_as_uint(v.z) & 0x7FFFFFFFu);
        m = max(m, __float_as_uint(v.w) & 0x7FFFFFFFu);
    }
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
__global__ __launch_bounds__(256) void accumulate_image_kernel(int64_t N, int M, int64_t nps, float scale, int scale_exp,
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
                const float *src = Phi + n * (int64_t)M + fb * BS + f0;
                a = *reinterpret_cast<const float4 *>(src);
                b = *reinterpret_cast<const float4 *>(src + 4);
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

// The 8-wave kernel takes gamma and beta of a stage from ONE 256-byte record (gamma x 32 | beta x 32, zeros beyond N), which
// one wave moves into LDS by DMA a stage ahead -- a scalar or vector load at the point of use would pay the HBM latency of a
// cold, once-read array in every stage.  acc_prep8_kernel writes the records and max gamma, acc_scale_kernel then multiplies
// gamma by s_B = 2^e_B (exact), so that the accumulation kernel applies no scale of its own.
__global__ __launch_bounds__(256) void acc_prep8_kernel(int64_t N, int64_t Npad, int L, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float *__restrict__ gb,
                                                        unsigned *__restrict__ scal) {
    const int64_t total = (int64_t)L * Npad;
    unsigned m = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t l = i / Npad, n = i - l * Npad;
        const bool in = n < N;
        const float gv = in ? gamma[l * N + n] : 0.f;
        const unsigned gbits = __float_as_uint(gv), ab = gbits & 0x7FFFFFFFu;
        const bool bad = ab >= 0x7F800000u || ((gbits >> 31) && ab != 0u); // inf, NaN or negative
        if (bad) atomicMax(scal + 1, (unsigned)min((int64_t)0x7FFFFFFE, l * N + n) + 1u);
        m = max(m, bad ? 0u : ab);
        float *rec = gb + (l * (Npad / 32) + n / 32) * 64 + (n & 31);
        rec[0] = gv;
        rec[32] = in ? beta[l * N + n] : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(scal, m);
}

__device__ __forceinline__ int acc_scale_exp(unsigned gmax) { // e_B: 2^e_B max(gamma) in [1/2, 1)
    int eB = gmax ? 126 - (int)(gmax >> 23) : 0;
    return eB < -60 ? -60 : (eB > 60 ? 60 : eB);
}

__global__ __launch_bounds__(256) void acc_scale_kernel(int64_t nrec, float *__restrict__ gb, const unsigned *__restrict__ scal) {
    const int eB = acc_scale_exp(scal[0]);
    const float sB = __uint_as_float((unsigned)(127 + eB) << 23);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec * 32; i += (int64_t)gridDim.x * blockDim.x) {
        float *p = gb + (i >> 5) * 64 + (i & 31);
        const float v = *p * sB;
        *p = v < 1.17549435e-38f ? 0.f : v; // (a product below the smallest normal is dropped)
    }
}

// ------------------------------------------------------------------------------------------------
// syrk_image8_kernel: the same tile, slots, images and slabs with EIGHT waves (512 threads, two per SIMD, up to 256
// registers each): a wave owns 128 x 64 of the tile (8 x 4 accumulators), keeps the B fragments of its four column blocks
// for the whole stage and streams the A fragments one 16-row block ahead of the twelve MFMAs that use them, so that a
// wave alone keeps the matrix pipe fed (with sixteen 128-register waves the fragment reads of a group could not be issued
// ahead of the previous group's MFMAs: a wave alone ran at ~40 cycles per MFMA instead of 16, and since a SIMD serves
// its oldest wave first, the youngest finished each stage alone at that pace -- in-kernel stamps).
// Staging per thread and stage: 4 DMA pieces of A, two granules of B (same row and plane in the stage's two slices).
// Diagonal tiles: block (i, j) of sub-tile (wr, wc) is needed iff 8 wr + i >= 4 wc + j; the six sub-tiles that hold such
// blocks are dealt over the SIMDs as 32 | 32 | 26 + 10 | 26 + 10 blocks (an off-diagonal tile: 64 per SIMD).
// ------------------------------------------------------------------------------------------------
#ifdef AGPL_QTRACE // diagnostic build (make QTRACE=1): per-wave cycle sums of the stage loop, tools/qtrace.py
__device__ unsigned long long g_qtrace[64 * 16 * 8];
constexpr bool kTrace = true;
#else
constexpr bool kTrace = false;
#endif

__constant__ unsigned char kDiagWave8[8] = {1 | 16, 1 | 4 | 16, 0 | 16, 1 | 8 | 16, 0, 0, 0 | 4 | 16, 1 | 12 | 16}; // wr | wc << 2 | active << 4

template <bool DIAG>
__device__ __forceinline__ void syrk_image8_body(unsigned char *smem_raw, int64_t N, int64_t Npad, int M, int nsplit, int l,
                                                 int s, int I, int J, const unsigned char *__restrict__ image,
                                                 const float *__restrict__ gb_all,
                                                 const unsigned *__restrict__ scal, float *__restrict__ slabG,
                                                 float *__restrict__ slabg) {
    constexpr bool TRACE = kTrace;
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0..7
    const int nb = M / BS;
    int wr = wave >> 2, wc = wave & 3;
    bool active = true;
    if (DIAG) {
        const unsigned e = kDiagWave8[wave];
        wr = e & 1;
        wc = (e >> 2) & 3;
        active = (e >> 4) & 1;
    }
    const AccImageHeader *hdr = reinterpret_cast<const AccImageHeader *>(image);
    const h8 *blocks = reinterpret_cast<const h8 *>(image + sizeof(AccImageHeader));
    const int eA = hdr->scale_exp;
    const int eB = acc_scale_exp(scal[0]); // (gamma arrives scaled by 2^e_B: acc_scale_kernel)

    const int64_t nbeg = (int64_t)s * kChunk;
    int64_t nend = nbeg + kChunk;
    if (nend > N) nend = N;
    const int nstage = (int)((nend - nbeg + kStagePts - 1) / kStagePts);
    const int64_t ps0 = nbeg / 16;

    // staging duty of a wave: 128-row block rbS = (wave >> 2) & 1 of BOTH panels, quarter qd = wave & 3 = (plane, 64-row half)
    const int rbS = (wave >> 2) & 1, qd = wave & 3;
    const int64_t slice_pitch = (int64_t)nb * 2 * 256; // h8 units between consecutive point slices
    const h8 *a_src = blocks + ((ps0 * nb + 2 * I + rbS) * 2) * 256 + qd * 64 + lane; // + 256: lo; + slice_pitch: slice 1
    const int a_dst = rbS * 2 * 4096 + qd * 1024;                                     // + 4096: lo; + kSliceBytes: slice 1
    const h8 *b_src = blocks + ((ps0 * nb + 2 * J + rbS) * 2) * 256 + qd * 64 + lane;
    const int b_dst = (4 + rbS * 2) * 4096 + qd * 1024 + lane * 16;
    // gamma | beta records of the slice's stages; the 8 points of this wave's granules are 8 (qd >> 1) + 0..7 of each
    // 16-point slice of a stage
    const float *gb_src = gb_all + ((int64_t)l * (Npad / 32) + nbeg / 32) * 64 + lane;
    float *gbuf = reinterpret_cast<float *>(smem_raw + 2 * kSlot); // [2 (stage parity)][64]
    const int gofs = 8 * (qd >> 1);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gacc = 0.f;

    u32x4 rh0, rl0, rh1, rl1; // the two raw granules in flight (slice 0, slice 1)

#define AGPL_E_DMA(t_)                                                                                          \
    do {                                                                                                        \
        unsigned char *slot_ = smem_raw + ((t_) & 1) * kSlot + a_dst;                                           \
        const h8 *src_ = a_src + (int64_t)(2 * (t_)) * slice_pitch;                                             \
        __builtin_amdgcn_global_load_lds(src_, (lds_void *)slot_, 16, 0, 0);                                    \
        __builtin_amdgcn_global_load_lds(src_ + 256, (lds_void *)(slot_ + 4096), 16, 0, 0);                     \
        __builtin_amdgcn_global_load_lds(src_ + slice_pitch, (lds_void *)(slot_ + kSliceBytes), 16, 0, 0);      \
        __builtin_amdgcn_global_load_lds(src_ + slice_pitch + 256, (lds_void *)(slot_ + kSliceBytes + 4096), 16, 0, 0); \
    } while (0)
#define AGPL_E_DMA1(t_, k_) /* piece k_ = 0..3 of the four */                                                   \
    do {                                                                                                        \
        unsigned char *slot_ = smem_raw + ((t_) & 1) * kSlot + a_dst + ((k_) & 1) * 4096 + ((k_) >> 1) * kSliceBytes; \
        const h8 *src_ = a_src + (int64_t)(2 * (t_)) * slice_pitch + ((k_) & 1) * 256 + ((k_) >> 1) * slice_pitch; \
        __builtin_amdgcn_global_load_lds(src_, (lds_void *)slot_, 16, 0, 0);                                    \
    } while (0)
#define AGPL_E_LOADB(t_)                                                                                        \
    do {                                                                                                        \
        const h8 *src_ = b_src + (int64_t)(2 * (t_)) * slice_pitch;                                             \
        rh0 = *reinterpret_cast<const u32x4 *>(src_);                                                           \
        rl0 = *reinterpret_cast<const u32x4 *>(src_ + 256);                                                     \
        rh1 = *reinterpret_cast<const u32x4 *>(src_ + slice_pitch);                                             \
        rl1 = *reinterpret_cast<const u32x4 *>(src_ + slice_pitch + 256);                                       \
    } while (0)
#define AGPL_E_DMAG(t_) /* wave 0: the gamma | beta record of stage t_ -> gbuf[t_ & 1] */                    \
    do {                                                                                                        \
        if (wave == 0)                                                                                          \
            __builtin_amdgcn_global_load_lds(gb_src + (int64_t)(t_) * 64, (lds_void *)(gbuf + ((t_) & 1) * 64), 4, 0, 0); \
    } while (0)
#define AGPL_E_CVT(RH_, RL_, g0_, g1_, b0_, b1_)                                                                \
    do {                                                                                                        \
        float x0_, x1_;                                                                                         \
        AGPL_Q_JOIN2(RH_, RL_, x0_, x1_);                                                                       \
        if (DIAG) {                                                                                             \
            AGPL_E_GFMA(gacc, b0_, x0_);                                                                        \
            AGPL_E_GFMA(gacc, b1_, x1_);                                                                        \
        }                                                                                                       \
        AGPL_E_SPLIT2(x0_, g0_, x1_, g1_, RH_, RL_);                                                            \
    } while (0)
    // half a granule (4 points): its gamma (and beta on a diagonal tile) come out of the stage's record in LDS (one address
    // for the whole wave: a broadcast read); keepf zeroes the beta of the clamped duplicate behind the last stage
#define AGPL_E_CVT_HALF(c_, tt_, keepf_)                                                                        \
    do {                                                                                                        \
        const float *gq_ = gbuf + ((tt_) & 1) * 64 + gofs + 16 * ((c_) >> 1) + 4 * ((c_) & 1);                  \
        const float4 g4_ = *reinterpret_cast<const float4 *>(gq_);                                              \
        float4 b4_ = {0.f, 0.f, 0.f, 0.f};                                                                      \
        if (DIAG) {                                                                                             \
            b4_ = *reinterpret_cast<const float4 *>(gq_ + 32);                                                  \
            b4_.x *= (keepf_); b4_.y *= (keepf_); b4_.z *= (keepf_); b4_.w *= (keepf_);                         \
        }                                                                                                       \
        if ((c_) == 0) { AGPL_E_CVT(rh0.x, rl0.x, g4_.x, g4_.y, b4_.x, b4_.y); AGPL_E_CVT(rh0.y, rl0.y, g4_.z, g4_.w, b4_.z, b4_.w); } \
        if ((c_) == 1) { AGPL_E_CVT(rh0.z, rl0.z, g4_.x, g4_.y, b4_.x, b4_.y); AGPL_E_CVT(rh0.w, rl0.w, g4_.z, g4_.w, b4_.z, b4_.w); } \
        if ((c_) == 2) { AGPL_E_CVT(rh1.x, rl1.x, g4_.x, g4_.y, b4_.x, b4_.y); AGPL_E_CVT(rh1.y, rl1.y, g4_.z, g4_.w, b4_.z, b4_.w); } \
        if ((c_) == 3) { AGPL_E_CVT(rh1.z, rl1.z, g4_.x, g4_.y, b4_.x, b4_.y); AGPL_E_CVT(rh1.w, rl1.w, g4_.z, g4_.w, b4_.z, b4_.w); } \
    } while (0)
#define AGPL_E_STOREB(t_)                                                                                       \
    do {                                                                                                        \
        const unsigned d_ = lds_base + (unsigned)(((t_) & 1) * kSlot + b_dst);                                  \
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:4096\n\t"                             \
                     "ds_write_b128 %0, %3 offset:32768\n\tds_write_b128 %0, %4 offset:36864" ::"v"(d_),        \
                     "v"(rh0), "v"(rl0), "v"(rh1), "v"(rl1)                                                     \
                     : "memory");                                                                               \
    } while (0)

    [[maybe_unused]] unsigned long long tr0 = 0, tr_wait = 0, tr_body = 0, tr_loop = 0, rt0 = 0, tr_h[4] = {0, 0, 0, 0};
    if (TRACE) {
        tr0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem_raw;
    AGPL_E_DMAG(0);
    AGPL_E_DMA(0);
    AGPL_E_LOADB(0);
    __builtin_amdgcn_s_waitcnt(0x0070); // the record of stage 0 is in LDS ...
    __builtin_amdgcn_s_barrier();       // ... for every wave
    if (nstage > 1) AGPL_E_DMAG(1);
    AGPL_E_CVT_HALF(0, 0, 1.f);
    AGPL_E_CVT_HALF(1, 0, 1.f);
    AGPL_E_CVT_HALF(2, 0, 1.f);
    AGPL_E_CVT_HALF(3, 0, 1.f);
    AGPL_E_STOREB(0);
    AGPL_E_LOADB(nstage > 1 ? 1 : 0);

    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int kg = ln >> 4;
    const int fbase = (kg >> 1) * 2048 + (kg & 1) * 128 + (ln & 15);
    const int fa = fbase + wr * 512;
    const int fb = fbase + 1024 + (wc >> 1) * 512 + (wc & 1) * 64;
    const int jbase = DIAG ? 8 * wr - 4 * wc + 1 : 4; // blocks j < jbase + i of row block i are needed


    if (TRACE) tr_loop = __builtin_amdgcn_s_memtime();
    // hook i sits behind the MFMAs of row block i: the staging work of stage t + 1 in pieces
#define AGPL_E_HOOK(i_)                                                                                         \
    do {                                                                                                        \
        if (TRACE && ((i_) == 0 || (i_) == 2 || (i_) == 4 || (i_) == 6)) {                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            tr_h[(i_) / 2] += __builtin_amdgcn_s_memtime() - tb;                                                \
        }                                                                                                       \
        if ((i_) == 0 && t + 2 < nstage) AGPL_E_DMAG(t + 2);                                                    \
        if (more && (i_) < 4) AGPL_E_DMA1(t + 1, i_); /* one piece per hook: four at once queue behind each other */ \
        if ((i_) >= 1 && (i_) <= 4) AGPL_E_CVT_HALF((i_) - 1, t + 1, keepf);                                    \
        if ((i_) == 4) {                                                                                        \
            AGPL_E_STOREB(t + 1);                                                                               \
            AGPL_E_LOADB(tl);                                                                                   \
        }                                                                                                       \
    } while (0)
    // the MFMA block with the set of needed blocks fixed at compile time: JB_ = 4 all of them, 1 / -3 the two kinds of
    // sub-tile that straddle the diagonal (blocks j < JB_ + i of row block i), 100 = a wave with no sub-tile (staging only)
#define AGPL_E_BLOCK(JB_)                                                                                       \
    do {                                                                                                        \
        h8 bh[4], bl[4];                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
            bh[j] = st[fb + 16 * j];                                                                            \
            bl[j] = st[256 + fb + 16 * j];                                                                      \
        }                                                                                                       \
        constexpr int i0_ = (JB_) >= 1 ? 0 : 1 - (JB_); /* first row block with a needed block */              \
        h8 ah = st[fa + 16 * i0_], al = st[256 + fa + 16 * i0_];                                                \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                         \
            if (i >= i0_) {                                                                                     \
                const h8 ahc = ah, alc = al;                                                                    \
                if (i < 7) { /* the next row block's fragments, ahead of this one's MFMAs */                    \
                    ah = st[fa + 16 * (i + 1)];                                                                 \
                    al = st[256 + fa + 16 * (i + 1)];                                                           \
                }                                                                                               \
                __builtin_amdgcn_s_setprio(1);                                                                  \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) if (j < (JB_) + i) {                              \
                    acc[i][j] = mfma32(ahc, bh[j], acc[i][j]);                                                  \
                    acc[i][j] = mfma32(ahc, bl[j], acc[i][j]);                                                  \
                    acc[i][j] = mfma32(alc, bh[j], acc[i][j]);                                                  \
                }                                                                                               \
                __builtin_amdgcn_s_setprio(0);                                                                  \
            }                                                                                                   \
            AGPL_E_HOOK(i);                                                                                     \
        }                                                                                                       \
    } while (0)
    // one stage loop per kind of wave, chosen once: a branch inside the loop would make the accumulators a merge of the
    // bodies' and spill them
#define AGPL_E_LOOP(JB_)                                                                                        \
    for (int t = 0; t < nstage; ++t) {                                                                          \
        unsigned long long ta = 0, tb = 0;                                                                      \
        if (TRACE) ta = __builtin_amdgcn_s_memtime();                                                           \
        /* stage t's A pieces, the raw granules of stage t + 1 and the B' stores of stage t are done */        \
        __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) lgkmcnt(0) */                                           \
        __builtin_amdgcn_s_barrier();                                                                           \
        if (TRACE) tb = __builtin_amdgcn_s_memtime();                                                           \
        const bool more = t + 1 < nstage;                                                                       \
        const int tl = t + 2 < nstage ? t + 2 : nstage - 1; /* unconditional loads (clamped) */                \
        const float keepf = more ? 1.f : 0.f;               /* behind the last stage the conversion is a duplicate */ \
        const h8 *st = reinterpret_cast<const h8 *>(smem_raw + (t & 1) * kSlot);                                \
        if ((JB_) == 100) {                                                                                     \
            if (more) AGPL_E_DMA(t + 1);                                                                        \
            if (t + 2 < nstage) AGPL_E_DMAG(t + 2);                                                             \
            AGPL_E_CVT_HALF(0, t + 1, keepf);                                                                   \
            AGPL_E_CVT_HALF(1, t + 1, keepf);                                                                   \
            AGPL_E_CVT_HALF(2, t + 1, keepf);                                                                   \
            AGPL_E_CVT_HALF(3, t + 1, keepf);                                                                   \
            AGPL_E_STOREB(t + 1);                                                                               \
            AGPL_E_LOADB(tl);                                                                                   \
        } else {                                                                                                \
            AGPL_E_BLOCK(JB_);                                                                                  \
        }                                                                                                       \
        if (TRACE) {                                                                                            \
            const unsigned long long td = __builtin_amdgcn_s_memtime();                                         \
            tr_wait += tb - ta;                                                                                 \
            tr_body += td - tb;                                                                                 \
        }                                                                                                       \
    }
    if (!DIAG || (active && jbase >= 4)) {
        AGPL_E_LOOP(4)
    } else if (!active) {
        AGPL_E_LOOP(100)
    } else if (jbase == 1) {
        AGPL_E_LOOP(1)
    } else {
        AGPL_E_LOOP(-3)
    }
#undef AGPL_E_LOOP
#undef AGPL_E_BLOCK
#undef AGPL_E_HOOK
#ifdef AGPL_QTRACE
    if (blockIdx.x < 64 && lane == 0) {
        const unsigned long long te = __builtin_amdgcn_s_memtime(), rte = __builtin_amdgcn_s_memrealtime();
        unsigned long long *o = g_qtrace + ((size_t)blockIdx.x * 16 + wave) * 8;
        o[0] = tr_h[0] | (tr_h[1] << 32); // (sums over <= 128 stages of < 2^24 cycles each: 32 bits are enough)
        o[1] = te - tr_loop;
        o[2] = tr_wait;
        o[3] = tr_h[2] | (tr_h[3] << 32);
        o[4] = tr_body;
        o[5] = rte - rt0;
        o[6] = (unsigned long long)nstage | ((unsigned long long)(DIAG ? 1 : 0) << 32) | ((unsigned long long)(active ? 1 : 0) << 33);
        o[7] = te - tr0;
    }
#endif
#undef AGPL_E_DMA
#undef AGPL_E_LOADB
#undef AGPL_E_DMAG
#undef AGPL_E_DMA1
#undef AGPL_E_CVT
#undef AGPL_E_CVT_HALF
#undef AGPL_E_STOREB

    // ---- slabs: [l][128-pair][slice][128 x 128] float32; the accumulators carry (s_A phi)(s_B gamma s_A phi)'
    const float unscale = __uint_as_float((unsigned)(127 - (2 * eA + eB)) << 23);
    if (active) {
        const int bi = 2 * I + wr, bj = 2 * J + (wc >> 1);
        const int npairs = nb * (nb + 1) / 2;
        const int p128 = bi * (bi + 1) / 2 + bj;
        float *slab = slabG + (((int64_t)l * npairs + p128) * nsplit + s) * (int64_t)(BS * BS);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + 4 * kg + r;
                    const int col = (wc & 1) * 64 + 16 * j + (ln & 15);
                    slab[row * BS + col] = acc[i][j][r] * unscale;
                }
    }
    if (DIAG) {
        // the two plane waves of a (128-row block, half) hold partial sums of the same rows
        float *gw = reinterpret_cast<float *>(smem_raw); // [2][256]; the stage slots are dead behind this barrier
        __syncthreads();
        gw[(qd >> 1) * 256 + rbS * 128 + (qd & 1) * 64 + lane] = gacc;
        __syncthreads();
        if (threadIdx.x < 256) {
            const int r = threadIdx.x;
            slabg[(((int64_t)l * nb + 2 * I + (r >> 7)) * nsplit + s) * BS + (r & 127)] =
                (gw[r] + gw[256 + r]) * __uint_as_float((unsigned)(127 - eA) << 23);
        }
    }
}

__global__ __launch_bounds__(512, 2) void syrk_image8_kernel(int64_t N, int64_t Npad, int M, int npairs2, int nsplit,
                                                             const unsigned char *__restrict__ image,
                                                             const float *__restrict__ gb_all,
                                                             const unsigned *__restrict__ scal,
                                                             float *__restrict__ slabG, float *__restrict__ slabg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // item decode: workgroups that share blockIdx % 8 (one XCD under round-robin dispatch; speed only) walk whole slices
    const int nsplit8 = (nsplit + 7) / 8;
    const int per_l = npairs2 * nsplit8 * 8;
    const int l = blockIdx.x / per_l;
    const int id = blockIdx.x - l * per_l;
    const int xcd = id & 7, jj = id >> 3;
    const int s = (jj / npairs2) * 8 + xcd;
    const int p2 = jj % npairs2;
    if (s >= nsplit) return;
    const int nb2 = M / kPanel;
    const int noff = nb2 * (nb2 - 1) / 2;
    int I, J;
    if (p2 < noff) { // off-diagonal tiles first (the long items of a slice)
        I = 1;
        while ((I + 1) * I / 2 <= p2) ++I;
        J = p2 - I * (I - 1) / 2;
    } else {
        I = J = p2 - noff;
    }
    if (I == J) syrk_image8_body<true>(smem_raw, N, Npad, M, nsplit, l, s, I, J, image, gb_all, scal, slabG, slabg);
    else syrk_image8_body<false>(smem_raw, N, Npad, M, nsplit, l, s, I, J, image, gb_all, scal, slabG, slabg);
}

} // namespace

#ifdef AGPL_QTRACE
extern "C" __attribute__((visibility("default"))) int agpl_debug_qtrace(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qtrace), sizeof(g_qtrace));
}
#endif

size_t agpl_syrk_image_lds_bytes() { return 2 * (size_t)kSlot + 512; } // two stage slots + two gamma | beta records

extern "C" int64_t agpl_accumulate_image_bytes(int64_t N, int32_t M) {
    if (N <= 0 || M <= 0 || M % BS) return 0;
    const int64_t nps = ((N + kStagePts - 1) / kStagePts) * 2; // whole 32-point stages
    return (int64_t)sizeof(AccImageHeader) + nps * (M / BS) * 2 * 4096;
}

extern "C" int32_t agpl_accumulate_image(agpl_ctx *ctx, int64_t N, int32_t M, const float *Phi, void *image_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (N <= 0 || M <= 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes N=%lld M=%d", (long long)N, M);
    if (M % BS) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "M = %d must be a multiple of %d (zero-pad the features)", M, BS);
    if (!Phi || !image_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    int32_t rc = agpl_ws2_reserve(ctx, 4096);
    if (rc) return rc;
    unsigned *mx = (unsigned *)ctx->ws2 + 512; // words 512.. of the small scratch (the info flags live below)
    AGPL_HIP(ctx, hipMemsetAsync(mx, 0, 16, ctx->stream));
    const int64_t n4 = N * (int64_t)M / 4;
    absmax_kernel<<<4096, 256, 0, ctx->stream>>>(n4, reinterpret_cast<const float4 *>(Phi), mx);
    AGPL_LAUNCH_CHECK(ctx);
    unsigned hmx = 0;
    AGPL_HIP(ctx, hipMemcpyAsync(&hmx, mx, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hmx >= 0x7F800000u) {
        unsigned long long *bad = (unsigned long long *)(mx + 2), hbad = ~0ull;
        AGPL_HIP(ctx, hipMemsetAsync(bad, 0xFF, 8, ctx->stream));
        find_bad_kernel<<<4096, 256, 0, ctx->stream>>>(N * (int64_t)M, Phi, __builtin_inff(), bad);
        AGPL_HIP(ctx, hipMemcpyAsync(&hbad, bad, 8, hipMemcpyDeviceToHost, ctx->stream));
        AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        AGPL_FAIL(ctx, AGPL_ERR_DOMAIN, "feature value is not finite (point %lld, feature %lld)",
                  (long long)(hbad / (unsigned long long)M), (long long)(hbad % (unsigned long long)M));
    }
    float max_abs;
    memcpy(&max_abs, &hmx, 4);
    // 2^e_A max|Phi| in [2^13, 2^14): hi stays finite, and lo = f16(x - hi) is a float16 normal for |x| down to 2^-17 max|Phi|
    int eA = 0;
    if (hmx) {
        const int ex = (int)(hmx >> 23) - 127; // max_abs in [2^ex, 2^(ex+1))   (a subnormal max reads ex = -127)
        eA = 13 - ex;
        if (eA > 30) eA = 30; // 2 e_A + e_B stays an exponent of a normal float32 (e_B in [-60, 60])
        if (eA < -30) eA = -30;
    }
    const float scale = ldexpf(1.f, eA);
    const int64_t nps = ((N + kStagePts - 1) / kStagePts) * 2;
    accumulate_image_kernel<<<16384, 256, 0, ctx->stream>>>(N, M, nps, scale, eA, max_abs, Phi, (unsigned char *)image_out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

// internal (agpl_accumulate_impl): prep + accumulation kernel; slabs as agpl_mfma.hip lays them out.
// gb: 2 L Npad floats of scratch (the gamma | beta records), Npad = N rounded up to 32 (+ 32); scal: 2 words of scratch.
int32_t agpl_syrk_image_launch(agpl_ctx *ctx, int64_t N, int64_t Npad, int32_t M, int32_t L, const void *image,
                               const float *gamma, const float *beta, float *gb, unsigned *scal, float *slabG,
                               float *slabg, int ns) {
    if (M % kPanel) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "the image accumulation needs M %% 256 == 0 (M = %d)", M);
    const int nb2 = M / kPanel;
    const int npairs2 = nb2 * (nb2 + 1) / 2;
    const int64_t nwg = (int64_t)L * npairs2 * ((ns + 7) / 8) * 8;
    if (nwg > 0x7fffffffLL) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "problem too large for one launch");
    AGPL_HIP(ctx, hipMemsetAsync(scal, 0, 2 * sizeof(unsigned), ctx->stream));
    acc_prep8_kernel<<<2048, 256, 0, ctx->stream>>>(N, Npad, L, gamma, beta, gb, scal);
    acc_scale_kernel<<<1024, 256, 0, ctx->stream>>>((int64_t)L * (Npad / 32), gb, scal);
    AGPL_LAUNCH_CHECK(ctx);
    const size_t lds = agpl_syrk_image_lds_bytes();
    AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&syrk_image8_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    syrk_image8_kernel<<<(unsigned)nwg, 512, lds, ctx->stream>>>(N, Npad, M, npairs2, ns, (const unsigned char *)image, gb,
                                                                 scal, slabG, slabg);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}
