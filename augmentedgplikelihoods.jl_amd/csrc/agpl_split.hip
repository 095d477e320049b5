// agpl_split.hip -- the marginal pass (a11) on the fast matrix cores: every float32 operand x is carried as two
// float16 values, x ~= hi + lo with hi = f16(x), lo = f16(x - hi) (error <= max(2^-22 |x|, 3e-8), |x| < 6e4), and
//   W' Phi  ~=  W'hi Phihi + W'hi Philo + W'lo Phihi          (the lo*lo term, <= 2^-22 relative, is dropped)
// runs as three v_mfma_f32_32x32x16_f16 per 32x32x16 sub-product, accumulated in float32: 16x the rate of the
// f32-input MFMA for 3x the instructions.  Both operands of this product have the reduction index (the feature b)
// contiguous in memory, so each is stored ONCE in a blocked image that is both the global and the LDS layout:
//
//   block (row-block of 128, k-slice of 16)  =  [plane h = 2][row 128][8 halves]  = 4 KB contiguous
//   (plane h holds k = 8h .. 8h+7 of the slice: the 16 bytes one MFMA lane needs; lanes 0-31 read plane 0, lanes
//   32-63 plane 1, each a contiguous 512 B -> conflict-free ds_read_b128; a workgroup stages a block with one
//   coalesced 16-byte load per thread.)
//
//   Phi: blocks [tile][M/16], written once by agpl_split_features;  W': blocks [l][rb][M/16], written each sweep by
//   agpl_pack_w_split (upper-triangular, doubled off-diagonal, as the f32 Wpack).
// The Hadamard epilogue and the output are float32 (the exact float32 Phi is read for it).
#include <cstdlib>
#include <type_traits>

#include "agpl_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BS = 128;
constexpr int KS = 16;  // reduction slice per stage
constexpr int NT = 128; // points per tile

__device__ __forceinline__ void split_f16(float x, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Phi [N][M] float32  ->  blocked hi / lo images (zero rows past N)
// (scale = 2^e: 1 for the unscaled image of agpl_split_features, the accumulate image's 2^e_A for a plan's self-scaled image)
// Msrc: the features the caller's rows hold (row pitch Msrc floats); features Msrc .. M - 1 of the images are zero
__global__ __launch_bounds__(256) void split_features_kernel(int64_t N, int M, int Msrc, const float *__restrict__ Phi, float scale,
                                                             h8 *__restrict__ Ph, h8 *__restrict__ Pl) {
    const int nks = M / KS;
    const int64_t nblk = ((N + NT - 1) / NT) * nks;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t tile = blk / nks;
        const int ks = (int)(blk - tile * nks);
        const int plane = threadIdx.x >> 7, row = threadIdx.x & 127;
        const int64_t n = tile * NT + row;
        h8 hi, lo;
        if (n < N) {
            const int fg = ks * KS + plane * 8;
            const float *src = Phi + n * (int64_t)Msrc + fg;
            float xs[8];
            if (!(Msrc & 3) && fg + 8 <= Msrc) {
                const float4 x0 = *reinterpret_cast<const float4 *>(src), x1 = *reinterpret_cast<const float4 *>(src + 4);
                xs[0] = x0.x, xs[1] = x0.y, xs[2] = x0.z, xs[3] = x0.w, xs[4] = x1.x, xs[5] = x1.y, xs[6] = x1.z, xs[7] = x1.w;
            } else { // ragged rows
#pragma unroll
                for (int j = 0; j < 8; ++j) xs[j] = fg + j < Msrc ? src[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 a, b;
                split_f16(xs[j] * scale, a, b);
                hi[j] = a;
                lo[j] = b;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) hi[j] = lo[j] = (_Float16)0.f;
        }
        Ph[blk * 256 + threadIdx.x] = hi;
        Pl[blk * 256 + threadIdx.x] = lo;
    }
}

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_f16: the two 16-deep slices of a stage ARE the 32-deep reduction of one instruction (lane l reads row
// l & 15 of k-group l >> 4 = slice l >> 5, plane (l >> 4) & 1 of the same 4 KB blocks; conflict-free ds_read_b128)
__device__ __forceinline__ f32x4 mfma32(h8 a, h8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

constexpr int NT2 = 256; // points per item of the queue kernel (two 128-point tiles of the image)

// ------------------------------------------------------------------------------------------------
// marginal_factor_queue_kernel: marginal_factor_persist_kernel<true> with the work cut one level finer and handed out
// dynamically.  An item is ONE 256-row block of U against one (tile, latent); the items of a tile are queued back to
// back, longest first, in eight queues (tiles t = q mod 8 in queue q; a workgroup serves queue blockIdx.x & 7, i.e.
// under round-robin dispatch the queues line up with the XCDs -- for speed only, never correctness).  Resident
// workgroups take the next item of their queue whenever they finish one, so the row blocks of a tile are picked up
// within about a microsecond of each other by different CUs of one XCD and stream the tile's point images (B) through
// the same L2 at the same time: the images reach the fabric once per tile instead of once per row block (round-1 / persistent
// kernels: 1.56 x the image bytes at M = 512, 2.5 x at M = 1024).  The per-point sums of the row blocks go to two partial
// arrays [row block][latent][N]; marginal_combine_kernel adds them to the residual in a fixed order (bitwise
// reproducible: no atomics on the outputs).  The next-but-one item is fetched by ONE returning atomic per item, issued in
// the first stage of an item and written to LDS in the second: its latency hides behind a whole stage.
// ------------------------------------------------------------------------------------------------
// x over the lanes l, l ^ 16, l ^ 32, l ^ 48 (in every one of them), as (x_l + x_{l^16}) + (the same of l ^ 32): two VALU swaps
// (gfx950 v_permlane16_swap / v_permlane32_swap), no LDS crossbar round trip
__device__ __forceinline__ float kgroup_sum(float x) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned u = __float_as_uint(x);
    const u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned v = __float_as_uint(y);
    const u32x2 t = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// A wave's share of an item's per-point sums, q_n = sum_a T[a,n]^2 and m_n = sum_a v_a T[a,n], over the 16-row blocks I0 .. I1 - 1
// of its 64 rows (lane rows a = 16 i + 4 kg + 0..3), into the item's parity buffers qr, mr ([row group][256 points]); those
// accumulators are cleared.
// Round 4 (in-kernel stamps and probe builds, profiles/r04_mtrace_marginal.txt): all sixteen waves used to do this behind the
// item's last stage, with the matrix pipe idle -- ~250 VALU instructions each in three-deep dependent runs behind a vmcnt(0) the
// compiler put in front of the LDS read of v (= the wait for the DMA pieces just issued): with the sums compiled out the kernel
// ran 14-17 % faster.  Now
//  * row groups 0..2 take their sums in the first stage in which they have nothing else to do (stages 3, 5, 7 of the item's
//    diagonal block), beside the MFMAs of the waves still at work on their SIMD; row group 3, whose rows are complete with the
//    item, behind the last stage.  (Not free beside MFMAs -- the stages that carry a wave's sums are 350-450 cycles longer --
//    but the item's last stage lost 2 k.  Tried and measured slower: at the END of a wave's last active stage (the stage waits for
//    that wave); row group 3's in the first stage of the next item, behind its DMA issue (that stage then waits for row group 3:
//    +1.7-2.8 k cycles against -1.3 k); two halves in two idle stages (three instances of this routine: spills).)
//  * v comes from LDS by inline-asm reads (no vmcnt(0): the buffer is not a DMA target of this or the previous stage);
//  * sixteen independent chains of scalar FMAs (v_pk_fma_f32 measured: -1 % at M = 512, +0.5 % at M = 1024 -- and the library keeps
//    packed-float32 arithmetic out of its sums: DESIGN 4.4e, Reproducibility);
//  * the sums over the four k-groups of a column (lanes l, l ^ 16, l ^ 32, l ^ 48) by v_permlane16/32_swap: no LDS round trip.
template <int I0, int I1, bool NOSUMS>
__device__ __forceinline__ void item_sums_rows(f32x4 (&acc)[4][4], const float *vrow, float *qr, float *mr) {
    typedef __attribute__((address_space(3))) void lds_void;
    f32x4 a0[I1 - I0];
    {
        const unsigned aaddr = (unsigned)(uintptr_t)(lds_void *)(vrow + 16 * I0);
        if (I1 - I0 == 1)
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a0[0]) : "v"(aaddr));
        else if (I1 - I0 == 2)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a0[0]), "=&v"(a0[(I1 - I0) > 1 ? 1 : 0])
                         : "v"(aaddr));
        else
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                         "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a0[0]), "=&v"(a0[(I1 - I0) > 1 ? 1 : 0]), "=&v"(a0[(I1 - I0) > 2 ? 2 : 0]), "=&v"(a0[(I1 - I0) > 3 ? 3 : 0])
                         : "v"(aaddr));
    }
    f32x2 q2[4], m2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) q2[j] = m2[j] = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const f32x2 alo = {a0[i - I0][0], a0[i - I0][1]}, ahi = {a0[i - I0][2], a0[i - I0][3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 lo = {acc[i][j][0], acc[i][j][1]};
            if (!NOSUMS) {
                q2[j][0] = fmaf(lo[0], lo[0], q2[j][0]), q2[j][1] = fmaf(lo[1], lo[1], q2[j][1]);
                m2[j][0] = fmaf(alo[0], lo[0], m2[j][0]), m2[j][1] = fmaf(alo[1], lo[1], m2[j][1]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 hi = {acc[i][j][2], acc[i][j][3]};
            if (!NOSUMS) {
                q2[j][0] = fmaf(hi[0], hi[0], q2[j][0]), q2[j][1] = fmaf(hi[1], hi[1], q2[j][1]);
                m2[j][0] = fmaf(ahi[0], hi[0], m2[j][0]), m2[j][1] = fmaf(ahi[1], hi[1], m2[j][1]);
            }
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    float qacc[4], macc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        qacc[j] = kgroup_sum(q2[j][0] + q2[j][1]);
        macc[j] = kgroup_sum(m2[j][0] + m2[j][1]);
    }
    if (lane_e < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qr[16 * j + lane_e] = qacc[j];
            mr[16 * j + lane_e] = macc[j];
        }
    }
}

#if defined(AGPL_MCLOCK) && !defined(AGPL_MTRACE) // diagnostic build: two stamps around the stage loop only (the in-kernel clock of the kernel as shipped)
__device__ unsigned long long g_mtrace[256 * 16 * 4];
__device__ unsigned long long g_mtrace_phase[256 * 16 * 24];
constexpr bool kMTrace = false;
constexpr bool kMClock = true;
#elif defined(AGPL_MTRACE) // diagnostic build (make MTRACE=1): per-wave cycle sums of the stage loop, tools/mtrace.py
__device__ unsigned long long g_mtrace[256 * 16 * 4];
__device__ unsigned long long g_mtrace_phase[256 * 16 * 24]; // per wave: [kind 3][segment 7 + count]: kind 0 full stage, 1 diagonal stage 1..6, 2 diagonal stage 7, 8
constexpr bool kMTrace = true;
constexpr bool kMClock = true;
#else
constexpr bool kMTrace = false;
constexpr bool kMClock = false;
#endif
__global__ __launch_bounds__(1024, 1) void marginal_factor_queue_kernel(
    int64_t N, int M, int L, int64_t ntiles128, int ntiles2, const h8 *__restrict__ Ph, const h8 *__restrict__ Pl,
    const h8 *__restrict__ Wh, const h8 *__restrict__ Wl, const float *__restrict__ v_all,
    float *__restrict__ qpart, float *__restrict__ mpart, unsigned *__restrict__ queues, unsigned *__restrict__ zero2,
    float unq, float unm) {
    // unq, unm: the point image may carry s Phi (s = 2^e, a plan's self-scaled image): the sums come out as s^2 q and s m and are
    // written as unq (s^2 q), unm (s m) -- exact powers of two, 1 for the unscaled image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int R = 2, KU = 2;
    if (zero2 && blockIdx.x == 0 && threadIdx.x < 2) zero2[threadIdx.x] = 0u; // (the caller's scale words: see the launch)
    // ... and the sweep's "bad gamma" word (queues[8]): the update behind the previous sweep has forwarded it by now, and an update
    // route that does not forward it (M > 1024: library factorisation) must not leave a stale flag for a later problem
    if (zero2 && blockIdx.x == 0 && threadIdx.x == 2) queues[8] = 0u;
    constexpr int kSlot = KU * 8 * 4096;
    float *alpha_s = reinterpret_cast<float *>(smem_raw + R * kSlot); // [2][256] floats (v of the item's 256-row block, by item parity)
    float *qred = alpha_s + 2 * NT2;                                   // [2][4 x 256] by item parity
    float *mred = qred + 2 * 4 * NT2;                                  // [2][4 x 256]
    int *qi = reinterpret_cast<int *>(mred + 2 * 4 * NT2);             // [4][4] this workgroup's items, decoded

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0..15
    const unsigned lane_v = (unsigned)lane;
    const int wr = wave >> 2, wc = wave & 3;
    // Issue priority by row group (round 5).  The hardware issues oldest wave first: waves 12..15 (row group 3, whose rows carry work to
    // the last stage of an item's diagonal block and which therefore end every such stage) got the leftover slots (round-4 stamps: they
    // arrive last at every barrier whatever their share).  With priority = row group the wave that ends the stage goes first:
    // 7.66 -> 7.49 ms at N = 1e7, M = 512 and 13.28 -> 13.07 ms at N = 5e6, M = 1024 (random features, tools/kbench.py,
    // profiles/r05_ab_marginal_priority.txt).  Measured beside it (-DAGPL_MPRIO=k): 0 none, 2 reversed (no gain), 3 only row
    // group 3 raised (7.50 / 13.12), 4 graded with the item sums dropped to priority 0 (7.58 / 13.06).
#ifndef AGPL_MPRIO
#define AGPL_MPRIO 1
#endif
    {
        const int pr = AGPL_MPRIO == 0 ? 0 : AGPL_MPRIO == 2 ? 3 - wr : AGPL_MPRIO == 3 ? (wr == 3 ? 3 : 0) : wr;
        if (pr == 3) __builtin_amdgcn_s_setprio(3);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    }
    const int nb = M / BS, nks = M / KS, nb2 = M / NT2;
    const int qn = (int)(blockIdx.x & 7); // (the launch has >= 8 workgroups: every queue is served)
    const int ntq = qn < ntiles2 ? (ntiles2 - qn + 7) / 8 : 0;
    const int nitems = ntq * L * nb2;
    unsigned *queue = queues + qn;

    // item k of this workgroup = queue index j: (tile, latent, 256-row block), longest row block first.  Thread 0 decodes an index
    // once, when it stores it (two integer divisions: by every wave they cost the SIMDs ~1.5 k cycles per item); the waves read
    // qi[k & 3] = {tile, latent, row block, valid}
#define AGPL_Q_STORE(k_, j_)                                                                                \
    do {                                                                                                    \
        const int jj_ = (int)(j_);                                                                          \
        const int jl_ = jj_ / nb2;                                                                          \
        const int tq_ = jl_ / L;                                                                            \
        int4 e_;                                                                                            \
        e_.x = tq_ * 8 + qn;                                                                                \
        e_.y = jl_ - tq_ * L;                                                                               \
        e_.z = nb2 - 1 - (jj_ - jl_ * nb2);                                                                 \
        e_.w = jj_ >= 0 && jj_ < nitems;                                                                    \
        *reinterpret_cast<int4 *>(qi + 4 * ((k_) & 3)) = e_;                                                \
    } while (0)
    typedef __attribute__((address_space(3))) void lds_void;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    if (threadIdx.x == 0) AGPL_Q_STORE(0, atomicAdd(queue, 1u));
    __syncthreads();

    const int ia = wave >> 2, qd = wave & 3;
    const h8 *a_img = (ia & 1) ? Wl : Wh, *b_img = (ia & 1) ? Pl : Ph;
    const int dma_off = ia * 4096 + qd * 1024;
    const int wra_ = 2 * (ia >> 1) + (qd & 1);          // the 64-row block of the item this wave stages (image layout: quarter qd = k half qd >> 1 of rows 64 (qd & 1) ..)
    constexpr bool skip_zero_rows = true; // (-3.5 % at C2 with / without: profiles/r02_ab_marginal_zero_row_skip.jsonl)

#define AGPL_Q_DECODE(k_, valid_, tile_, l_, rb_)                                                           \
    do {                                                                                                    \
        /* (inline asm: in front of a plain LDS read the compiler waits for vmcnt(0) -- the DMA pieces just issued) */ \
        const unsigned qaddr_ = (unsigned)(uintptr_t)(lds_void *)(qi + 4 * ((k_) & 3));                     \
        i32x4 e_;                                                                                           \
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e_) : "v"(qaddr_));               \
        tile_ = __builtin_amdgcn_readfirstlane(e_[0]);                                                      \
        l_ = __builtin_amdgcn_readfirstlane(e_[1]);                                                         \
        rb_ = __builtin_amdgcn_readfirstlane(e_[2]);                                                        \
        valid_ = __builtin_amdgcn_readfirstlane(e_[3]) != 0;                                                \
    } while (0)

    int ik = 0, irb = 0, iks = 0, il = 0, itile = 0; // issue pointer
    bool ivalid;
    AGPL_Q_DECODE(0, ivalid, itile, il, irb);
    if (!ivalid) return; // this queue is empty (uniform over the workgroup)
    const h8 *a_src, *b_src;
#define AGPL_Q_SRC()                                                                                        \
    do {                                                                                                    \
        const int64_t t128_ = AGPL_MPROBE == 2 ? (ia >> 1)                                                  \
                              : 2 * (int64_t)itile + (ia >> 1) < ntiles128 ? 2 * (int64_t)itile + (ia >> 1) \
                                                                           : ntiles128 - 1;                 \
        a_src = a_img + ((int64_t)il * nb + 2 * irb + (ia >> 1)) * nks * 256 + qd * 64;                     \
        b_src = b_img + t128_ * nks * 256 + qd * 64;                                                        \
    } while (0)
    // Measurement builds (tools/build_variant.sh ... -DAGPL_MPROBE=k; wrong sums, same instruction stream otherwise):
    //   1  no DMA after the first two stages (the stage loop's compute side alone)
    //   2  every point image read is tile 0's (all of B from L2: the loop without its HBM stream)
    //   3  the 64 pieces of a stage issued by four waves (one per SIMD, 16 pieces each) instead of 4 by each of the 16
    //   4  no per-point sums at the end of an item (the accumulators are only cleared)
    //   5  every stage multiplied in full (no zero block of U skipped)
#ifndef AGPL_MPROBE
#define AGPL_MPROBE 0
#endif
#if AGPL_MPROBE == 3
#define AGPL_Q_PIECES(t_)                                                                                   \
    do {                                                                                                    \
        if (wave < 4) {                                                                                     \
            _Pragma("unroll") for (int ia_ = 0; ia_ < 4; ++ia_) {                                           \
                const int64_t t128_ = 2 * (int64_t)itile + (ia_ >> 1) < ntiles128 ? 2 * (int64_t)itile + (ia_ >> 1) : ntiles128 - 1; \
                const h8 *as_ = ((ia_ & 1) ? Wl : Wh) + ((int64_t)il * nb + 2 * irb + (ia_ >> 1)) * nks * 256 + wave * 64; \
                const h8 *bs_ = ((ia_ & 1) ? Pl : Ph) + t128_ * nks * 256 + wave * 64;                      \
                unsigned char *slot_ = smem_raw + ((t_) & 1) * kSlot + ia_ * 4096 + wave * 1024;            \
                const int wr2_ = 2 * (ia_ >> 1) + (wave & 1);                                               \
                _Pragma("unroll") for (int u_ = 0; u_ < KU; ++u_) {                                         \
                    if (!skip_zero_rows || iks + u_ < 16 * irb + 4 * (wr2_ + 1))                            \
                        __builtin_amdgcn_global_load_lds(as_ + (int64_t)(iks + u_) * 256 + lane_v,          \
                                                         (lds_void *)(slot_ + u_ * 8 * 4096), 16, 0, 0);    \
                    __builtin_amdgcn_global_load_lds(bs_ + (int64_t)(iks + u_) * 256 + lane_v,              \
                                                     (lds_void *)(slot_ + u_ * 8 * 4096 + 4 * 4096), 16, 0, 0); \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
#else
#define AGPL_Q_PIECES(t_)                                                                                   \
    do {                                                                                                    \
        if (AGPL_MPROBE != 1 || (t_) < 2) {                                                                 \
            unsigned char *slot_ = smem_raw + ((t_) & 1) * kSlot + dma_off;                                 \
            _Pragma("unroll") for (int u_ = 0; u_ < KU; ++u_) {                                             \
                /* U is lower triangular: the 64 rows x 8 k this wave stages are the 64-row block wra_ of the item,  */  \
                /* which is zero from slice 16 irb + 4 (wra_ + 1) on -- nobody reads it there (`act` below) */       \
                if (!skip_zero_rows || iks + u_ < 16 * irb + 4 * (wra_ + 1))                                \
                    __builtin_amdgcn_global_load_lds(a_src + (int64_t)(iks + u_) * 256 + lane_v,            \
                                                     (lds_void *)(slot_ + u_ * 8 * 4096), 16, 0, 0);        \
                __builtin_amdgcn_global_load_lds(b_src + (int64_t)(iks + u_) * 256 + lane_v,                \
                                                 (lds_void *)(slot_ + u_ * 8 * 4096 + 4 * 4096), 16, 0, 0); \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
#endif
#define AGPL_Q_ISSUE(t_)                                                                                    \
    do {                                                                                                    \
        if (ivalid) {                                                                                       \
            AGPL_Q_PIECES(t_);                                                                              \
            iks += KU;                                                                                      \
            if (iks == 16 * (irb + 1)) {                                                                    \
                iks = 0;                                                                                    \
                ++ik;                                                                                       \
                AGPL_Q_DECODE(ik, ivalid, itile, il, irb);                                                  \
                if (ivalid) AGPL_Q_SRC();                                                                   \
            }                                                                                               \
        }                                                                                                   \
    } while (0)

    AGPL_Q_SRC();
    AGPL_Q_ISSUE(0);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int ck = 0, rb = 0, ks = 0, cl = 0, ctile = 0; // consume pointer
    bool cvalid;
    AGPL_Q_DECODE(0, cvalid, ctile, cl, rb);
    int pend = -1, pl = 0, prb = 0, ptile = 0;
    unsigned fetched = 0u; // thread 0: the queue index fetched in the first stage of the item, stored in its second
    bool fetch_pending = false;
    [[maybe_unused]] unsigned long long mt0 = 0, mtw = 0, mtb = 0, mta = 0, mtn = 0, mtl = 0;
    // segment sums by stage kind (scalars: every one of them is wave-uniform and lives in SGPRs)
    //   seg 0 dma wait | 1 barrier | 2 barrier exit -> before the DMA issue (bookkeeping, first fragment reads, first MFMAs) |
    //   3 the DMA issue | 4 the rest of the MFMAs | 5 item-end work | 6 (unused)
#define AGPL_MT_DECL(k_) [[maybe_unused]] unsigned long long mp##k_##0 = 0, mp##k_##1 = 0, mp##k_##2 = 0, mp##k_##3 = 0, mp##k_##4 = 0, mp##k_##5 = 0, mp##k_##n = 0
    AGPL_MT_DECL(0);
    AGPL_MT_DECL(1);
    AGPL_MT_DECL(2);
#undef AGPL_MT_DECL
    [[maybe_unused]] int mkind = 0;
    [[maybe_unused]] unsigned long long mtx = 0;
#define AGPL_MT_SEG(seg_)                                                                                   \
    do {                                                                                                    \
        if (kMTrace) {                                                                                      \
            const unsigned long long y_ = __builtin_amdgcn_s_memtime();                                     \
            const unsigned long long d_ = y_ - mtx;                                                         \
            mtx = y_;                                                                                       \
            if (mkind == 0) mp0##seg_ += d_;                                                                \
            else if (mkind == 1) mp1##seg_ += d_;                                                           \
            else mp2##seg_ += d_;                                                                           \
        }                                                                                                   \
    } while (0)
    [[maybe_unused]] unsigned long long mr0 = 0, mkl = 0;
    [[maybe_unused]] int mkp = -1;
    [[maybe_unused]] unsigned long long mk0 = 0, mk1 = 0, mk2 = 0, mk3 = 0, mk4 = 0, mk5 = 0, mk6 = 0, mk7 = 0, mk8 = 0;
    [[maybe_unused]] unsigned long long mn0 = 0, mn1 = 0, mn2 = 0, mn3 = 0, mn4 = 0, mn5 = 0, mn6 = 0, mn7 = 0, mn8 = 0;
    if (kMClock) {
        mt0 = __builtin_amdgcn_s_memtime();
        mr0 = __builtin_amdgcn_s_memrealtime();
    }
    for (int t = 0; cvalid; ++t) {
        if (kMTrace) {
            mta = __builtin_amdgcn_s_memtime();
            mtx = mta;
            mkind = ks < 16 * rb ? 0 : (ks - 16 * rb < 12 ? 1 : 2);
            if (mkind == 0) mp0n += 1;
            else if (mkind == 1) mp1n += 1;
            else mp2n += 1;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): stage t has landed (R = 2: nothing younger is in flight)
        if (kMTrace) {
            const unsigned long long x = __builtin_amdgcn_s_memtime();
            mtw += x - mta;
            mta = x;
        }
        AGPL_MT_SEG(0);
        __builtin_amdgcn_s_barrier();
        if (kMTrace) {
            const unsigned long long x = __builtin_amdgcn_s_memtime();
            mtb += x - mta;
            ++mtn;
        }
        AGPL_MT_SEG(1);
        // (AGPL_MCLOCK build) one stamp per stage at the barrier exit, consumed at the bottom of the loop body, where this wave's
        // LDS reads have all been waited for anyway: the stage lengths by kind without a wait the shipped kernel does not have
        [[maybe_unused]] unsigned long long mks = 0;
        [[maybe_unused]] int mkk = 0;
        if (kMClock && !kMTrace) {
            mks = __builtin_amdgcn_s_memtime();
            mkk = ks < 16 * rb ? 0 : 1 + (ks - 16 * rb) / KU;
        }
        if (fetch_pending) {
            if (threadIdx.x == 0) AGPL_Q_STORE(ck + 1, fetched);
            fetch_pending = false;
        }
        if (ks == 0 && wave == 0) {
            // first stage of an item: the 256 entries of v that belong to the item's row block into the item's parity buffer, one
            // 1 KB piece by the DMA path (in LDS behind the next barrier; first read at the end of the item's second stage).
            // (Rounds 4-5 kept all M entries: 8 M bytes of LDS, which put M >= 2048 beyond the 160 KB of a CU -- round 6.)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const f32x4 *>(v_all + (int64_t)cl * M + rb * NT2) + lane_v,
                                             (lds_void *)(alpha_s + (ck & 1) * NT2), 16, 0, 0);
        }
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int kg = ln >> 4;
        if (pend >= 0) {
            // the previous item's per-wave partial sums are in LDS (written before this barrier): rows out
            const int tid_e = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            if (tid_e < NT2) {
                const int64_t n = (int64_t)ptile * NT2 + tid_e;
                if (n < N) {
                    const float *qr = qred + (pend & 1) * 4 * NT2, *mr = mred + (pend & 1) * 4 * NT2;
                    const int64_t o = ((int64_t)prb * L + pl) * N + n;
                    qpart[o] = unq * ((qr[tid_e] + qr[NT2 + tid_e]) + (qr[2 * NT2 + tid_e] + qr[3 * NT2 + tid_e]));
                    mpart[o] = unm * ((mr[tid_e] + mr[NT2 + tid_e]) + (mr[2 * NT2 + tid_e] + mr[3 * NT2 + tid_e]));
                }
            }
            pend = -1;
        }
        // U is lower triangular: this wave's rows 64 wr .. 64 wr + 63 of block rb are zero from slice 16 rb + 4 (wr + 1)
        const bool act = AGPL_MPROBE == 5 || ks < rb * 16 + 4 * (wr + 1);
        const h8 *st = reinterpret_cast<const h8 *>(smem_raw + (t & 1) * kSlot);
        const int fbase16 = (kg >> 1) * 2048 + (kg & 1) * 128 + (ln & 15);
        const int fa = fbase16 + (wr >> 1) * 512 + (wr & 1) * 64;
        const int fb = fbase16 + 1024 + (wc >> 1) * 512 + (wc & 1) * 64;
        // (Round 4 measured a CYCLIC assignment of the 16-row blocks to the four waves of a SIMD -- every wave runs out of work
        // together in the diagonal part of an item: same MFMAs, 6.51 -> 7.55 ms at C2, not kept.  What the stage loop spends its
        // time on (stamps and probe builds -DAGPL_MPROBE=k, tools/mtrace.py; profiles/r04_mtrace_marginal.txt): a stage takes
        // ~3.9 k cycles whether full (3.07 k of MFMA per SIMD) or diagonal (0.8-2.3 k); without ANY DMA after the prologue (probe 1)
        // still ~3.6 k (-9 % cycles, -5 % time: the clock drops); with all of Phi served by L2 (probe 2) -5 %; the 64 pieces of a stage
        // issued by four waves instead of sixteen (probe 3) no change; without the per-point sums at item ends (probe 4) -14..17 %.
        // I.e. neither HBM nor the DMA path bounds it: every barrier costs the SIMD ~0.5-0.9 k cycles of fragment-read latency,
        // bookkeeping and waiting for the slowest wave, and the item-end sums were serial VALU work -- the part that was cut.)
        // the wave's LAST active stage covers k = its own rows 32..63 of the diagonal 64 x 64 block of U: rows 0..31 (i = 0, 1)
        // are zero there -- their MFMAs would add exact zeros and are skipped (1 + 32 / M instead of 1 + 64 / M executed).
        // One code path with wave-uniform branches around the i = 0, 1 groups (two copies of the loop body spill).
        if (act) {
            const bool lo_rows = AGPL_MPROBE == 5 || !(skip_zero_rows && ks + KU == rb * 16 + 4 * (wr + 1));
            h8 ah[4], al[4];
            if (lo_rows) {
                ah[0] = st[fa];
                ah[1] = st[fa + 16];
            }
            ah[2] = st[fa + 32];
            ah[3] = st[fa + 48];
            h8 bh = st[fb], bl = st[256 + fb];
            if (lo_rows) {
                acc[0][0] = mfma32(ah[0], bh, acc[0][0]);
                acc[1][0] = mfma32(ah[1], bh, acc[1][0]);
            }
            acc[2][0] = mfma32(ah[2], bh, acc[2][0]);
            acc[3][0] = mfma32(ah[3], bh, acc[3][0]);
            __builtin_amdgcn_sched_barrier(0);
            AGPL_MT_SEG(2);
            AGPL_Q_ISSUE(t + 1); // into the slot read in iteration t - 1, behind the first MFMAs
            AGPL_MT_SEG(3);
            __builtin_amdgcn_sched_barrier(0);
            if (lo_rows) {
                al[0] = st[256 + fa];
                al[1] = st[256 + fa + 16];
            }
            al[2] = st[256 + fa + 32];
            al[3] = st[256 + fa + 48];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j > 0) {
                    bh = st[fb + 16 * j];
                    bl = st[256 + fb + 16 * j];
                    if (lo_rows) {
                        acc[0][j] = mfma32(ah[0], bh, acc[0][j]);
                        acc[1][j] = mfma32(ah[1], bh, acc[1][j]);
                    }
                    acc[2][j] = mfma32(ah[2], bh, acc[2][j]);
                    acc[3][j] = mfma32(ah[3], bh, acc[3][j]);
                }
                if (lo_rows) {
                    acc[0][j] = mfma32(ah[0], bl, acc[0][j]);
                    acc[1][j] = mfma32(ah[1], bl, acc[1][j]);
                    acc[0][j] = mfma32(al[0], bh, acc[0][j]);
                    acc[1][j] = mfma32(al[1], bh, acc[1][j]);
                }
                acc[2][j] = mfma32(ah[2], bl, acc[2][j]);
                acc[3][j] = mfma32(ah[3], bl, acc[3][j]);
                acc[2][j] = mfma32(al[2], bh, acc[2][j]);
                acc[3][j] = mfma32(al[3], bh, acc[3][j]);
            }
        } else {
            AGPL_MT_SEG(2);
            AGPL_Q_ISSUE(t + 1);
            AGPL_MT_SEG(3);
        }
        if (kMTrace) __builtin_amdgcn_sched_barrier(0);
        AGPL_MT_SEG(4);
        if (ks == 16 * (rb + 1) - 4 * KU) {
            // four stages before the item ends: take the NEXT item of the queue -- as late as the pipeline allows (its
            // index is stored next stage, visible the stage after, decoded by the issue pointer in the stage after that),
            // so that the workgroups that take the row blocks of one tile start them within a stage or two of each other.
            // Behind the stage's DMA issue: the compiler waits for the returned value at once (vmcnt(0)), and wave 0 -- idle in
            // this stage -- must have its pieces of the next stage out before it sits in that wait.
            if (threadIdx.x == 0) fetched = atomicAdd(queue, 1u);
            fetch_pending = true;
        }
        {
            // row groups 0..2: in their first idle stage of the item's diagonal block (their rows are complete); row group 3: behind
            // its (= the item's) last stage
            const int s_idle = ks - (rb * 16 + 4 * (wr + 1));
            const float *vrow = alpha_s + (ck & 1) * NT2 + wr * 64 + 4 * kg;
            float *qr = qred + (ck & 1) * 4 * NT2 + wr * NT2 + wc * 64, *mr = mred + (ck & 1) * 4 * NT2 + wr * NT2 + wc * 64;
            constexpr bool kNoSums = AGPL_MPROBE == 4;
            if (wr < 3 ? s_idle == 0 : ks + KU == 16 * (rb + 1)) {
#if AGPL_MPRIO == 4
                __builtin_amdgcn_s_setprio(0);
#endif
                item_sums_rows<0, 4, kNoSums>(acc, vrow, qr, mr);
#if AGPL_MPRIO == 4
                if (wr == 3) __builtin_amdgcn_s_setprio(3);
                else if (wr == 2) __builtin_amdgcn_s_setprio(2);
                else if (wr == 1) __builtin_amdgcn_s_setprio(1);
#endif
            }
        }
        ks += KU;
        if (ks == 16 * (rb + 1)) {
            // item finished: every wave's partial sums are in LDS behind the next barrier
            pend = ck;
            pl = cl;
            prb = rb;
            ptile = ctile;
            ks = 0;
            ++ck;
            AGPL_Q_DECODE(ck, cvalid, ctile, cl, rb);
        }
        AGPL_MT_SEG(5);
        if (kMClock && !kMTrace) {
            // mks - mkl = the length of the PREVIOUS stage (barrier exit to barrier exit), of kind mkp
            if (mkp >= 0) {
                const unsigned long long d = mks - mkl;
#define AGPL_MK(k_) if (mkp == k_) { mk##k_ += d; mn##k_ += 1; }
                AGPL_MK(0) else AGPL_MK(1) else AGPL_MK(2) else AGPL_MK(3) else AGPL_MK(4) else AGPL_MK(5) else AGPL_MK(6) else AGPL_MK(7) else AGPL_MK(8)
#undef AGPL_MK
            }
            mkl = mks;
            mkp = mkk;
        }
    }
#undef AGPL_Q_ISSUE
#undef AGPL_Q_PIECES
#undef AGPL_Q_SRC
#undef AGPL_Q_DECODE
#undef AGPL_Q_STORE
#undef AGPL_MT_SEG
#if defined(AGPL_MTRACE) || defined(AGPL_MCLOCK)
    if (blockIdx.x < 256 && lane == 0) {
        unsigned long long *o = g_mtrace + ((size_t)blockIdx.x * 16 + wave) * 4;
        o[0] = __builtin_amdgcn_s_memtime() - mt0; // loop cycles
        o[1] = mtw;                                 // waiting for the DMA of the stage
        o[2] = mtb;                                 // waiting at the barrier
        o[3] = kMTrace ? mtn : 1;                   // stages
        unsigned long long *ph = g_mtrace_phase + ((size_t)blockIdx.x * 16 + wave) * 24;
#define AGPL_MT_OUT(k_)                                                                                     \
    ph[8 * k_ + 0] = mp##k_##0, ph[8 * k_ + 1] = mp##k_##1, ph[8 * k_ + 2] = mp##k_##2, ph[8 * k_ + 3] = mp##k_##3,          \
                ph[8 * k_ + 4] = mp##k_##4, ph[8 * k_ + 5] = mp##k_##5, ph[8 * k_ + 6] = 0, ph[8 * k_ + 7] = mp##k_##n
        AGPL_MT_OUT(0);
        AGPL_MT_OUT(1);
        AGPL_MT_OUT(2);
        if (!kMTrace) { // stage lengths by kind: sums in [0..8] (but [6], moved to [18]), counts in [9..17]
            ph[0] = mk0, ph[1] = mk1, ph[2] = mk2, ph[3] = mk3, ph[4] = mk4, ph[5] = mk5, ph[18] = mk6, ph[7] = mk7, ph[8] = mk8;
            ph[9] = mn0, ph[10] = mn1, ph[11] = mn2, ph[12] = mn3, ph[13] = mn4, ph[14] = mn5, ph[15] = mn6, ph[16] = mn7, ph[17] = mn8;
        }
        ph[6] = __builtin_amdgcn_s_memrealtime() - mr0; // 100 MHz ticks of the loop (in-kernel clock = o[0] / ph[6] x 100 MHz)
#undef AGPL_MT_OUT
    }
#endif
    __syncthreads();
    if (pend >= 0) {
        const int tid_e = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (tid_e < NT2) {
            const int64_t n = (int64_t)ptile * NT2 + tid_e;
            if (n < N) {
                const float *qr = qred + (pend & 1) * 4 * NT2, *mr = mred + (pend & 1) * 4 * NT2;
                const int64_t o = ((int64_t)prb * L + pl) * N + n;
                qpart[o] = unq * ((qr[tid_e] + qr[NT2 + tid_e]) + (qr[2 * NT2 + tid_e] + qr[3 * NT2 + tid_e]));
                mpart[o] = unm * ((mr[tid_e] + mr[NT2 + tid_e]) + (mr[2 * NT2 + tid_e] + mr[3 * NT2 + tid_e]));
            }
        }
    }
}

// Round 6 built the alternative the round-5 analysis called for and removed it again -- record:
// profiles/NOTES_r06.md section 2, profiles/r06_ab_marginal_pair.jsonl, commit "marginal_factor_pair_kernel".  Same items, queues,
// images, ring and arithmetic with 8 waves of 224 registers (two per SIMD) instead of 16 of 128: a wave owned 128 rows x 64 points
// (8 x 4 accumulators), the two waves of a SIMD took the item's 16-row blocks alternately so that both kept the same share of the
// diagonal block's shrinking work to the item's last stage, B fragments held for the stage, A fragments two row blocks ahead by
// inline asm with counted waits.  Ten-sweep parity and bitwise-repeat tests green; 7.03-7.09 ms against 5.89-5.99 at C2, 11.58
// against 9.75 at N = 5e6, M = 1024 (+ 19 % at both): fewer, fatter waves issue MFMAs at a lower rate on this device (the quad
// accumulation experiment: one wave per SIMD reaches half the pipe's rate) -- the four waves per SIMD of this kernel are what keeps
// the pipe at 58 %, and the diagonal stages' idle waves cost less than halving the wave count does.

// var_n = resid_n + sum_rb qpart[rb][l][n], mu_n = mu0_n + sum_rb mpart[rb][l][n]  (row blocks in ascending order)
__global__ __launch_bounds__(256) void marginal_combine_kernel(int64_t N, int L, int nb2, const float *__restrict__ resid,
                                                               const float *__restrict__ mu0,
                                                               const float *__restrict__ qpart,
                                                               const float *__restrict__ mpart,
                                                               float *__restrict__ mu_out, float *__restrict__ var_out,
                                                               unsigned *__restrict__ queues) {
    if (blockIdx.x == 0 && threadIdx.x < 8) queues[threadIdx.x] = 0u; // the item queues, for the next marginal launch
    const int64_t total = (int64_t)L * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i % N;
        float q = 0.f, m = 0.f;
        for (int rb = 0; rb < nb2; ++rb) {
            q += qpart[(int64_t)rb * total + i];
            m += mpart[(int64_t)rb * total + i];
        }
        if (mu0) m += mu0[i];
        mu_out[i] = m;
        var_out[i] = resid[n] + q;
    }
}

// U = R^-1 as rocSOLVER leaves it: column-major lower triangle of A, i.e. U[a][b] = A[b * M + a] for b <= a
// (the other triangle of A still holds I + G and is never read) -> blocked hi / lo images of U
// info_host (may be null): pinned host memory the factorisation's info words are forwarded to -- the last kernel of an update
// reports its outcome itself instead of a copy command behind it
__global__ __launch_bounds__(256) void pack_factor_split_kernel(int M, const double *__restrict__ A,
                                                                h8 *__restrict__ Wh, h8 *__restrict__ Wl,
                                                                const int *__restrict__ info, int *__restrict__ info_host,
                                                                int ninfo, unsigned *__restrict__ bad_gamma, double scale) {
    // scale = 2^e (exact): 1 for the images of agpl_pack_factor_split; a plan carries 2^15 U -- every |U[a][b]| <= 1 (I + G >= I),
    // so 2^15 U is finite in float16 whatever G is, and entries down to 2^-29 keep a normal hi part (unscaled, an inverse
    // factor of a strongly informed posterior, |U| ~ 1e-5, sat in the float16 subnormals)
    const int nks = M / KS, nb = M / BS;
    const int l = blockIdx.z, rb = blockIdx.y, ks = blockIdx.x;
    if (info_host && l == 0 && rb == 0 && ks == 0) {
        if ((int)threadIdx.x < ninfo) info_host[threadIdx.x] = info[threadIdx.x];
        if (threadIdx.x == 127) { // the sweep's "bad gamma" word (agpl_fused_point_kernel), reported with the outcome
            info_host[127] = (int)*bad_gamma;
            *bad_gamma = 0u;
        }
    }
    const int plane = threadIdx.x >> 7, row = threadIdx.x & 127;
    const int a = rb * BS + row;
    const double *Al = A + (int64_t)l * M * M;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int b = ks * KS + plane * 8 + j;
        const double v = b <= a ? Al[(int64_t)b * M + a] * scale : 0.0;
        _Float16 x, y;
        split_f16((float)v, x, y);
        hi[j] = x;
        lo[j] = y;
    }
    const int64_t blk = ((int64_t)l * nb + rb) * nks + ks;
    Wh[blk * 256 + threadIdx.x] = hi;
    Wl[blk * 256 + threadIdx.x] = lo;
}

// resid_n = k_nn - |phi_n|^2 (float64 accumulation, one wave per point)
__global__ __launch_bounds__(256) void feature_residual_kernel(int64_t N, int M, const float *__restrict__ Phi,
                                                               const float *__restrict__ kdiag,
                                                               float *__restrict__ resid) {
    const int lane = threadIdx.x & 63;
    for (int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); n < N; n += (int64_t)gridDim.x * 4) {
        const float *row = Phi + n * (int64_t)M;
        double acc = 0.0;
        for (int a = lane * 4; a < M; a += 256) {
            const float4 x = *reinterpret_cast<const float4 *>(row + a);
            acc += (double)x.x * x.x + (double)x.y * x.y + (double)x.z * x.z + (double)x.w * x.w;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) resid[n] = (float)((double)kdiag[n] - acc);
    }
}

} // namespace

#if defined(AGPL_MTRACE) || defined(AGPL_MCLOCK)
extern "C" __attribute__((visibility("default"))) int agpl_debug_mtrace(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mtrace), sizeof(g_mtrace));
}
extern "C" __attribute__((visibility("default"))) int agpl_debug_mtrace_phase(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mtrace_phase), sizeof(g_mtrace_phase));
}
#endif

// internal (agpl_plan.hip): bytes of ONE marginal image (hi or lo) for N points, M features
int64_t agpl_split_features_bytes(int64_t N, int32_t M) {
    if (N <= 0 || M <= 0 || M % BS) return 0;
    return (int64_t)sizeof(_Float16) * ((N + NT - 1) / NT) * NT * M; // per image (hi and lo each)
}

int32_t agpl_feature_range_check(agpl_ctx *ctx, int64_t N, int32_t M, const float *Phi, float limit, const char *what,
                                 unsigned *max_bits_out); // agpl_syrk.hip

// internal (agpl_plan.hip): the marginal image of scale * Phi (the plan has checked the features and chosen the scale)
int32_t agpl_split_features_build(agpl_ctx *ctx, int64_t N, int32_t M, int32_t Msrc, const float *Phi, float scale, void *Phi_hi,
                                  void *Phi_lo) {
    int64_t nblk = ((N + NT - 1) / NT) * (M / KS);
    if (nblk > 65535 * 16) nblk = 65535 * 16;
    split_features_kernel<<<(unsigned)nblk, 256, 0, ctx->stream>>>(N, M, Msrc, Phi, scale, (h8 *)Phi_hi, (h8 *)Phi_lo);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

// internal (agpl_update.hip, agpl_plan.hip): split-float16 images of 2^u_scale_exp U from A_work, forwarding ninfo <= 128 info words
// of the factorisation to pinned host memory
int32_t agpl_pack_factor_split_info(agpl_ctx *ctx, int32_t M, int32_t L, const double *A, void *U_hi, void *U_lo,
                                    const int *info, int *info_host, int ninfo, int u_scale_exp) {
    if (M <= 0 || M % BS || L <= 0 || !A || !U_hi || !U_lo || ninfo > 127) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad argument");
    dim3 grid((unsigned)(M / KS), (unsigned)(M / BS), (unsigned)L);
    pack_factor_split_kernel<<<grid, 256, 0, ctx->stream>>>(M, A, (h8 *)U_hi, (h8 *)U_lo, info, info_host, ninfo,
                                                            (unsigned *)((char *)ctx->ws2 + 8192) + 8, ldexp(1.0, u_scale_exp));
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

extern "C" int32_t agpl_feature_residual(agpl_ctx *ctx, int64_t N, int32_t M, const float *Phi, const float *kdiag,
                                         float *resid_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (N < 0 || M <= 0 || M % 4) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "need N >= 0 and M %% 4 == 0");
    if (N == 0) return AGPL_OK;
    if (!Phi || !kdiag || !resid_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    int64_t nblk = agpl_cdiv(N, 4);
    if (nblk > 262144) nblk = 262144;
    feature_residual_kernel<<<(unsigned)nblk, 256, 0, ctx->stream>>>(N, M, Phi, kdiag, resid_out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

// internal: the MFMA part of the factor-form marginals -- per row block rb of U the partial sums
// qpart[rb][l][n] = sum_{rows of rb} (U Phi_n)^2 and mpart[rb][l][n] = v . (U Phi_n) over those rows, at the base of the
// workspace.  The caller's point kernel adds them up in ascending rb (marginal_combine_kernel, or agpl_fused_point_kernel
// of a sweep) and zeroes the eight queue words again.  zero2 (may be null): two words this kernel zeroes for the caller
// (the accumulation's scale words, which no kernel between the previous accumulation and this sweep's point kernel reads).
int32_t agpl_marginals_factor_parts(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const void *Phi_hi, const void *Phi_lo,
                                    const void *U_hi, const void *U_lo, const float *v, unsigned *zero2, float **qpart_out,
                                    float **mpart_out, unsigned **queues_out, int image_scale_exp) {
    // resident workgroups on 16x16x32 MFMA serving per-XCD queues of (tile, latent, row block) items: the row blocks of a tile
    // share its images through L2 (round 2: 6.31-6.36 against 6.67-6.78 ms for the per-tile kernel at C2 on one box, 1.92
    // against 2.63 ms at C4, 20.6 GB fetched over the fabric instead of 31.9 GB; the other forms measured then -- static
    // persistent runs, 32x32x16, L2 touch prefetch, all 512 rows resident -- are in DESIGN 4.3c and no longer in the tree)
    const int nb2 = M / NT2;
    const int64_t ntiles2 = agpl_cdiv(N, NT2);
    if (ntiles2 * L * nb2 > 0x3fffffffLL) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "problem too large for one launch");
    const size_t part_bytes = sizeof(float) * (size_t)nb2 * L * N;
    // the partial sums live at the base of the workspace (the slab region of the accumulation that follows them in a
    // sweep; a caller that keeps mu / var in the workspace has reserved more than this already: no reallocation)
    int32_t rc = agpl_ws_reserve(ctx, 2 * part_bytes + 256);
    if (rc) return rc;
    rc = agpl_ws2_reserve(ctx, 16384);
    if (rc) return rc;
    float *qpart = (float *)ctx->ws, *mpart = (float *)((char *)ctx->ws + ((part_bytes + 255) & ~(size_t)255));
    unsigned *queues = (unsigned *)((char *)ctx->ws2 + 8192); // zero between launches (agpl_ws2_reserve)
    const size_t ldsq = (size_t)2 * 2 * 8 * 4096 + sizeof(float) * (size_t)(2 * NT2 + 16 * NT2) + 64; // (+ 64: the four decoded items)
    if (!ctx->queue_attr) { // once per context
        AGPL_HIP(ctx, hipDeviceGetAttribute(&ctx->ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
        AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&marginal_factor_queue_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->queue_attr = 1;
    }
    // one resident workgroup per CU, and never fewer than the eight queues (a masked / partitioned device would otherwise
    // leave queues unserved and their tiles unwritten)
    const int nwg = ctx->ncu < 8 ? 8 : ctx->ncu;
    marginal_factor_queue_kernel<<<(unsigned)nwg, 1024, ldsq, ctx->stream>>>(
        N, M, L, agpl_cdiv(N, NT), (int)ntiles2, (const h8 *)Phi_hi, (const h8 *)Phi_lo, (const h8 *)U_hi, (const h8 *)U_lo,
        v, qpart, mpart, queues, zero2, ldexpf(1.f, -2 * image_scale_exp), ldexpf(1.f, -image_scale_exp));
    AGPL_LAUNCH_CHECK(ctx);
    *qpart_out = qpart;
    *mpart_out = mpart;
    *queues_out = queues;
    return AGPL_OK;
}

// (agpl_plan.hip) image_scale_exp: the point images hold 2^e Phi
int32_t agpl_marginals_factor_internal(agpl_ctx *ctx, int64_t N, int32_t M, int32_t L, const void *Phi_hi, const void *Phi_lo,
                                       const float *resid, const float *mu0, const void *U_hi, const void *U_lo, const float *v,
                                       float *mu_out, float *var_out, int image_scale_exp) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (N < 0 || M <= 0 || L <= 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "bad sizes N=%lld M=%d L=%d", (long long)N, M, L);
    if (M % NT2)
        AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED, "the factor form runs on 256-row blocks: M = %d must be a multiple of 256", M);
    if (N == 0) return AGPL_OK;
    if (!Phi_hi || !Phi_lo || !resid || !U_hi || !U_lo || !v || !mu_out || !var_out)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    int32_t rc = agpl_timing_begin(ctx, 0);
    if (rc) return rc;
    float *qpart, *mpart;
    unsigned *queues;
    rc = agpl_marginals_factor_parts(ctx, N, M, L, Phi_hi, Phi_lo, U_hi, U_lo, v, nullptr, &qpart, &mpart, &queues, image_scale_exp);
    if (rc) return rc;
    int64_t nbk = agpl_cdiv((int64_t)L * N, 256);
    if (nbk > 8192) nbk = 8192;
    marginal_combine_kernel<<<(unsigned)nbk, 256, 0, ctx->stream>>>(N, L, M / NT2, resid, mu0, qpart, mpart, mu_out, var_out,
                                                                    queues);
    AGPL_LAUNCH_CHECK(ctx);
    return agpl_timing_end(ctx, 0);
}
