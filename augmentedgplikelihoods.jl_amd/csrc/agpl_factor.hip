// agpl_factor.hip -- the M x M half of a sweep in ONE launch (M <= 512):
//
//     I + G = R R'   (Cholesky, float64)        U = R^-1        v = U (g + eta0)        logdet(I + G)
//
// which is what the factor form of the marginal pass consumes (agpl_marginals_factor_split: S = U'U and m = U'v are
// never formed).  The rocSOLVER route (potrf + trtri) costs ~300 dependent launches of 5-140 us = 3.4 ms at M = 512
// and at M = 1024 alike: latency, not flops (2 x M^3/3 = 90 MFLOP).  Here one 1024-thread workgroup per latent runs a
// right-looking blocked Cholesky (block 32) and carries the identity along as right-hand sides, so the inverse factor
// falls out of the same sweep over k:
//
//   for block k:   D  = T[k,k] -> R_kk = chol(D), W = R_kk^-1   (32 x 32 thread grid, one barrier per column)
//                  P  = T[k+1:, k] W'            (the panel of R below the block: M' x 32, kept in LDS only)
//                  X_k = W RHS[k, :]             (rows k of U: final; 32 x 32 (k+1), kept in LDS as X_k')
//                  T[k+1:, k+1:]  -= P P'        (trailing update, lower triangle)
//                  RHS[k+1:, :]   -= P X_k       (the identity's forward elimination)
//
// P has M - 32 (k+1) rows and X_k' has 32 (k+1): together always M rows of 32 doubles -- one LDS region of M x 33 doubles.
// The panel / X_k products and both updates are the same routine C[u][w] (-)= sum_m Uop[u][m] Vop[w][m] on 32 x 32 wave
// tiles of v_mfma_f64_16x16x4_f64 with both operands in LDS (pitch 33); the lanes of a result run along the
// contiguous index of the output (T row-major, U column-major).
// R itself is never stored: no later step needs an old panel.
//
// Output convention = rocSOLVER's (so agpl_pack_factor_split and the S / m accessors are shared with the library
// route): U[a][b] (b <= a) at A[b * M + a]; the other triangle of A is left untouched.
#include <cstdlib>

#include "agpl_common.h"

#ifdef AGPL_FTRACE
// debug build only (tools/scratch/ftrace.py): wall-clock stamps of the phases of every block step, per workgroup
__device__ unsigned long long g_ftrace[16 * 32 * 8];
#define AGPL_TS(id_)                                                                                                  \
    do {                                                                                                              \
        if (tid == 0 && blockIdx.y == 0 && wg < 16 && k < 32) g_ftrace[(wg * 32 + k) * 8 + (id_)] = wall_clock64();    \
    } while (0)
extern "C" __attribute__((visibility("default"))) int agpl_debug_ftrace(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ftrace), sizeof(g_ftrace));
}
#else
#define AGPL_TS(id_)
#endif

namespace {

constexpr int FB = 32;      // block size
constexpr int FP = FB + 1;  // LDS pitch (doubles)

__device__ __forceinline__ double readlane_f64(double x, int srclane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), srclane);
    return __hiloint2double(hi, lo);
}

typedef double d4 __attribute__((ext_vector_type(4)));

// R_kk = chol(D) and W = R_kk^-1 of one 32 x 32 diagonal block, by the whole workgroup (1024 threads; called under uniform control
// flow): sixteen steps over column PAIRS (c, c + 1), one barrier each.  Ds: the block (lower triangle + diagonal, pitch FP), Y: 32 x FP
// of LDS for the identity being eliminated, Wf: W on return (row c final behind step c / 2).  Behind the previous step columns c, c + 1
// of D and rows c, c + 1 of Y are final; every working thread rebuilds the two pivots and its own entries of R from D:
//     R[r][c] = D[r][c] / sqrt(D[c][c]),  R[r][c+1] = (D[r][c+1] - R[r][c] R[c+1][c]) / R[c+1][c+1]
//     D[r][cc] -= R[r][c] R[cc][c] + R[r][c+1] R[cc][c+1]                              (cc > c + 1)
//     W[c][:] = Y[c][:] / R[c][c],  W[c+1][:] = (Y[c+1][:] - R[c+1][c] W[c][:]) / R[c+1][c+1]
//     Y[r][:] -= R[r][c] W[c][:] + R[r][c+1] W[c+1][:]                                 (r > c + 1)
// The reciprocal pivots are v_rsq_f64 + two Newton steps (rounding-limited) instead of the ~400-cycle sqrt + divide sequences, the
// two of a step from independent chains (det = D[c][c] D[c+1][c+1] - D[c+1][c]^2: the second pivot is det / D[c][c]).  Four waves work
// (one per SIMD): thread (rq, cc) keeps D and Y of rows rq, rq + 8, rq + 16, rq + 24 at column cc in registers.  Rounds 1-4 ran it on
// eight waves (rows rq, rq + 16) on the reading that a step was bound by instruction issue -- most of a wave's instructions are the
// two chains every wave repeats; with four waves a SIMD issues them once instead of twice and a block takes the SAME 8.2-8.3 us
// (0.52 us per step, round-5 stamps): the step is bound by its dependent chain (barrier, LDS round trip, ~20 dependent float64
// operations behind v_rsq_f64, LDS write), not by issue.  Every element sees the same formula in either form: results unchanged to the
// bit.  Written branch-free (masks: a read inside a branch exposes its latency behind the chains); LDS only carries what other
// threads need next (columns c + 2, c + 3 of D, rows c + 2, c + 3 of Y).  sink(c, p0, p1): thread 0, the two pivots of the step (log
// det, first-bad-pivot bookkeeping of the caller).  One routine for factor_kernel and factor_pipe_kernel's F.
template <class SINK>
__device__ __forceinline__ void block_factor_32(int tk, double *__restrict__ Ds, double *__restrict__ Y, double *__restrict__ Wf,
                                                SINK sink) {
    const bool act = tk < 256;
    const int cc = tk & 31, rq = (tk >> 5) & 7;
    double d[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        d[j] = Ds[(rq + 8 * j) * FP + cc];
        y[j] = rq + 8 * j == cc ? 1.0 : 0.0;
    }
    Y[(tk >> 5) * FP + cc] = (tk >> 5) == cc ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < FB; c += 2) {
        __syncthreads();
        if (act) {
            double p0 = Ds[c * FP + c], b10 = Ds[(c + 1) * FP + c], d11 = Ds[(c + 1) * FP + c + 1];
            double x0 = Ds[cc * FP + c], x1 = Ds[cc * FP + c + 1]; // D[cc][c], D[cc][c+1]
            double ya = Y[c * FP + cc], yb = Y[(c + 1) * FP + cc];
            double a0[4], a1[4]; // D[row][c], D[row][c+1] of the thread's four rows
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0[j] = Ds[(rq + 8 * j) * FP + c];
                a1[j] = Ds[(rq + 8 * j) * FP + c + 1];
            }
            // (every read above is issued before the first use: the asm keeps them out of later branches)
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(ya), "+v"(yb), "+v"(a0[0]), "+v"(a1[0]), "+v"(a0[1]), "+v"(a1[1]), "+v"(a0[2]),
                         "+v"(a1[2]), "+v"(a0[3]), "+v"(a1[3]));
            const double det = __builtin_fma(d11, p0, -(b10 * b10));
            double r0 = __builtin_amdgcn_rsq(p0), rd = __builtin_amdgcn_rsq(det);
            r0 = r0 * (1.5 - 0.5 * p0 * r0 * r0);
            rd = rd * (1.5 - 0.5 * det * rd * rd);
            r0 = r0 * (1.5 - 0.5 * p0 * r0 * r0); // 1 / R[c][c]
            rd = rd * (1.5 - 0.5 * det * rd * rd);
            const double l10 = b10 * r0;          // R[c+1][c]
            const double r1 = rd * (p0 * r0);     // 1 / R[c+1][c+1]
            const double p1 = det * (r0 * r0);    // the second pivot
            const double lc0 = x0 * r0;                   // R[cc][c]
            const double lc1 = (x1 - lc0 * l10) * r1;     // R[cc][c+1]
            const double w0 = ya * r0;                    // W[c][cc]
            const double w1 = (yb - l10 * w0) * r1;       // W[c+1][cc]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = rq + 8 * j;
                const double la0 = a0[j] * r0, la1 = (a1[j] - la0 * l10) * r1; // R[row][c], R[row][c+1]
                // masks instead of branches (rows above the block being eliminated hold finite leftovers)
                const double md = (cc > c + 1 && row >= cc) ? 1.0 : 0.0, my = row > c + 1 ? 1.0 : 0.0;
                d[j] -= md * (la0 * lc0 + la1 * lc1);
                y[j] -= my * (la0 * w0 + la1 * w1);
                if (row == c) Wf[c * FP + cc] = w0;
                if (row == c + 1) Wf[(c + 1) * FP + cc] = w1;
                if (c + 2 < FB) {
                    if (cc == c + 2 || cc == c + 3) Ds[row * FP + cc] = d[j];
                    if (row == c + 2 || row == c + 3) Y[row * FP + cc] = y[j];
                }
            }
            if (tk == 0) sink(c, p0, p1);
        }
    }
}

// 32 x 32 macro tile on the float64 matrix cores:  acc[ti][tj] += Uop[16 ti + i][:] . Vop[16 tj + j][:]  over the 32
// columns of both LDS operands (row pitch FP).  v_mfma_f64_16x16x4_f64 (layout probed on gfx950, tools/scratch):
// lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; result register r of lane l is
// D[4 r + (l >> 4)][l & 15].
// VTRI: Vop is lower triangular (Vop[w][m] = 0 for m > w): its first 16 rows end at column 15, so the tj = 0 products
// of the upper half of the columns are skipped (24 MFMAs instead of 32).
template <bool VTRI>
__device__ __forceinline__ void macro_mac(const double *__restrict__ Uop, const double *__restrict__ Vop, int lane,
                                          d4 (&acc)[2][2]) {
    const int off = (lane & 15) * FP + (lane >> 4);
    const double *u = Uop + off, *v = Vop + off;
#pragma unroll
    for (int kk = 0; kk < FB / 4; ++kk) {
        const double a0 = u[4 * kk], a1 = u[16 * FP + 4 * kk];
        const double b1 = v[16 * FP + 4 * kk];
        if (!VTRI || kk < FB / 8) {
            const double b0 = v[4 * kk];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        }
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// 32 x 16 tile (two MFMA tiles sharing the B operand): the update tiles also hold the prefetched old values, and
// 2 x (16 + 16) result / old VGPRs is what fits beside them under the 128-VGPR cap of a 1024-thread workgroup
__device__ __forceinline__ void mac_2x1(const double *__restrict__ Uop, const double *__restrict__ Vop, int lane,
                                        d4 (&acc)[2]) {
    const int off = (lane & 15) * FP + (lane >> 4);
    const double *u = Uop + off, *v = Vop + off;
#pragma unroll
    for (int kk = 0; kk < FB / 4; ++kk) {
        const double a0 = u[4 * kk], a1 = u[16 * FP + 4 * kk], b0 = v[4 * kk];
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1], 0, 0, 0);
    }
}

// spin (one lane) until *p >= target, acquire at agent scope; bounded: a lost partner must not hang the device
__device__ __forceinline__ bool spin_until_ge(unsigned *p, unsigned target) {
    for (int it = 0; it < (1 << 22); ++it) {
        if (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// NW = 1: the whole factorisation in one workgroup.  NW > 1: NW workgroups per latent share the update tiles of every
// block step.  Workgroup 0 runs the serial spine (stage, block factor + inverse, panel / U rows), publishes P | X_k'
// (M rows of 32 doubles) through global memory and a `ready` counter; all NW workgroups take their share of the
// 32 x 16 update tiles and bump a `done` counter; workgroup 0 waits for it before staging block k + 1.  Release =
// barrier (all stores drained) + one lane's agent-scope atomic, acquire = agent-scope atomic load (invalidates the
// CU's L1) + barrier.
// The launch is 8 NW workgroups wide per latent and only those with blockIdx.x % 8 == latent % 8 work: the dispatcher
// deals workgroups round-robin to the 8 XCDs by linear id, so the cooperating ones of a latent share one L2 and the
// latents spread over the XCDs (all on XCD 0, the first form, put the 50 workgroups of a 10-latent update on 32 CUs).
//
// LA (look-ahead; NW = 2, 3, 5): workgroup 0 runs ONLY the spine, the other NW - 1 ONLY tiles, two block columns
// ahead of it.  At step k the spine factors block k, forms P | X_k' and publishes them, then applies this step's update
// itself -- in LDS, straight into the layout of step k + 1 -- to the next block column of T (diagonal block k + 1 and
// its panel) and to the next 32 rows of the eliminated identity.  The tile workgroups update everything from block
// column k + 2 on; the block column the spine will take over at step k + 1 comes first and is counted in `crit`, which
// the spine reads a whole step later: no round trip through another workgroup sits on the spine's path (measured:
// 29 -> 20 us per block step).  A T / U location is updated by one tile workgroup until the spine takes it over
// (ownership by absolute block coordinates), so no other hand-off is needed; P | X_k' travels through two alternating
// global buffers (buffer k & 1 is rewritten at step k + 2, by when every tile workgroup has loaded it: it counted
// crit(k) after doing so).
template <int NW, bool LA>
__global__ __launch_bounds__(1024, 1) void factor_kernel(int M, const double *__restrict__ Gall,
                                                         const double *__restrict__ gall,
                                                         const double *__restrict__ eta0all, double *__restrict__ Tall,
                                                         double *__restrict__ Aall, double *__restrict__ vall,
                                                         float *__restrict__ v32all, double *__restrict__ logdet,
                                                         int *__restrict__ info, double *__restrict__ PXg_all,
                                                         unsigned *__restrict__ sync_all, int rescue) {
    if (NW > 1 && (blockIdx.x & 7) != (blockIdx.y & 7)) return; // latent l works on XCD l % 8 (see the launch)
    if (NW > 1 && rescue == 2) { // fault injection (agpl_debug_force_factor_rescue): behave as if a partner never arrived
        if ((blockIdx.x >> 3) == 0 && threadIdx.x == 0) info[blockIdx.y] = -1;
        return;
    }
    // rescue launch (NW = 1, queued behind every multi-workgroup launch): redo latent l alone iff the cooperative
    // launch gave up on a partner that was not resident (info = -1: the device is shared with other work); G, g are
    // untouched inputs, so the result is the one the cooperative launch would have produced
    if (NW == 1 && rescue) { // ... and leaves the hand-off flags of its latent zero for the next factorisation (no memset)
        if (blockIdx.x == 0) // the whole flag area of the small workspace (bytes 8448 .. 16383: PipeFlags records, or 4 words per latent)
            for (int i = threadIdx.x; i < 1984; i += 1024) sync_all[i] = 0u;
        if (info[blockIdx.x] != -1 || M > 512) return; // (beyond 512 one workgroup cannot hold the panel: the loss is reported)
    }
    const int wg = NW > 1 ? (int)(blockIdx.x >> 3) : 0;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *PX = sm;                 // [M][FP]: row c < ncx = X_k'[c] (column c of U), row g >= ncx = P of global row g
    double *Ds = PX + (size_t)M * FP; // [32][FP] diagonal block being eliminated, then the identity's elimination
    double *Rs = Ds + FB * FP;       // [32][FP] the identity block being eliminated alongside
    double *Wf = Rs + FB * FP;       // [32][FP] W = R_kk^-1
    __shared__ int bad;
    __shared__ int lost;
    __shared__ double ldsum;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = NW > 1 ? (int)blockIdx.y : (int)blockIdx.x;
    double *PXg0 = PXg_all + (size_t)l * 2 * M * FB;       // NW > 1: P | X_k' of a block step (LA: two buffers)
    unsigned *ready = sync_all + 4 * l, *done = ready + 1, *crit = ready + 2; // steps published / tiles finished
    const double *G = Gall + (size_t)l * M * M;
    double *T = Tall + (size_t)l * M * M;
    double *A = Aall + (size_t)l * M * M;
    const int nb = M / FB;

    if (tid == 0) {
        bad = 0;
        lost = 0;
        ldsum = 0.0;
    }
    // T holds the updated trailing matrix; nothing has been updated before step 0, whose reads therefore come straight
    // from G (+ I): no M x M copy up front
    __syncthreads();

    for (int k = 0; k < nb; ++k) {
        // the thread and lane indices pass through an empty asm once per block step: without it every address that
        // depends on them is hoisted out of this loop as an invariant and the kernel spills ~100 VGPRs
        int tk = tid, ln = lane;
        asm volatile("" : "+v"(tk), "+v"(ln));
        const int kb = k * FB;
        const int Mp = M - kb - FB; // trailing rows
        const int ncx = kb + FB;    // columns of U that rows kb.. can touch
        if (wg == 0) {
            AGPL_TS(0);
            if (!LA && NW > 1 && k > 0) {
                // every workgroup's tiles of step k - 1 (they touch the block and panel staged next) are finished
                if (tk == 0 && !spin_until_ge(done, (unsigned)(NW * k))) lost = 1;
                __syncthreads();
                if (lost) break;
            }
            AGPL_TS(1);
            if (!LA || k == 0) { // look-ahead: block k > 0 is already in LDS, updated by this workgroup at step k - 1
        // ---- stage the diagonal block and the raw panel
            //      (PX rows Mp.. = rows kb..kb+31 of the eliminated identity, one LDS row per column c; block (k,k) is
            //      still the identity).  All of a thread's loads are issued before the first LDS store: one round trip
            //      to L2 instead of one per element.
            {
                const double *Tsrc = k == 0 ? G : T;
                const double dadd = k == 0 ? 1.0 : 0.0;
                const int r = tk >> 5, c = tk & 31;
                const double dval = Tsrc[(size_t)(kb + r) * M + kb + c];
                // thread (r, c) takes element c of rows r, r + 32, ...: row block u is all rows of U already eliminated
                // into (u < k), the identity block (u == k) or all P rows (u > k): uniform branches
                const unsigned toff = (unsigned)((kb + FB + r) * M + kb + c), aoff = (unsigned)(r * M + kb + c);
                // (8 loads in flight per thread: 16 would spill under the 128-VGPR cap)
#pragma unroll 1
                for (int u0 = 0; u0 < nb; u0 += 8) {
                    double tmp[8];
#pragma unroll
                    for (int uu = 0; uu < 8; ++uu) {
                        const int u = u0 + uu;
                        tmp[uu] = 0.0;
                        if (u > k && u < nb) tmp[uu] = Tsrc[toff + (unsigned)((u - k - 1) * 32 * M)];
                        else if (u < k) tmp[uu] = A[aoff + (unsigned)(u * 32 * M)];
                    }
#pragma unroll
                    for (int uu = 0; uu < 8; ++uu) {
                        const int u = u0 + uu;
                        if (u < nb) PX[(size_t)(u * 32 + r) * FP + c] = (u == k && r == c) ? 1.0 : tmp[uu];
                    }
                }
                Ds[r * FP + c] = c < r ? dval : (c == r ? dval + dadd : 0.0);
            }
            __syncthreads();
            }
            AGPL_TS(2);
            // ---- R_kk = chol(D) and W = R_kk^-1 together (block_factor_32: sixteen column-pair steps, the serial spine of the
            //      factorisation).  The pivots go to the padding column of PX (log det at the end, off the serial path)
            block_factor_32(tk, Ds, Rs, Wf, [&](int c, double p0, double p1) {
                PX[(size_t)(kb + c) * FP + FB] = p0;
                PX[(size_t)(kb + c + 1) * FP + FB] = p1;
                if (!(p0 > 0.0)) bad = kb + c + 1;
                else if (!(p1 > 0.0)) bad = kb + c + 2;
            });
            __syncthreads();
            AGPL_TS(3);
            // ---- every row of PX times W':  P[i'][c] = sum_m Araw[i'][m] W[c][m]  (panel of R below the block) and
            //      X_k'[c][m] = sum_q RHS[kb+q][c] W[m][q]  (rows kb..kb+31 of U, final).  Wave w owns rows 32 w..32 w + 31
            //      and nobody else touches them: in place without a barrier between its reads and its writes.
            if (wave * 32 < M) {
                d4 acc[2][2];
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = (d4){0.0, 0.0, 0.0, 0.0};
                double *rows = PX + (size_t)wave * 32 * FP;
                macro_mac<true>(rows, Wf, ln, acc); // W is lower triangular
                const bool xrows = wave * 32 < ncx; // a wave's rows are all X rows or all P rows
                const bool p0rows = LA && wave * 32 == ncx; // the first 32 rows of P: a copy for the look-ahead update
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * ti + 4 * r + (ln >> 4), m = 16 * tj + (ln & 15);
                            const double val = acc[ti][tj][r];
                            rows[i * FP + m] = val;
                            if (p0rows) Rs[i * FP + m] = val;
                            if (xrows) {
                                const int c = wave * 32 + i;
                                if (c <= kb + m) A[(size_t)c * M + kb + m] = val; // (the other triangle of A is not ours)
                            }
                        }
            }
            __syncthreads();
            AGPL_TS(4);
            if (NW > 1 && (!LA || Mp >= 2 * FB)) {
                // publish P | X_k' (the LDS rows without their padding) and the step counter
                double *PXg = PXg0 + (LA ? (size_t)(k & 1) * M * FB : 0);
                for (int idx = tk; idx < M * FB; idx += 1024) PXg[idx] = PX[(size_t)(idx >> 5) * FP + (idx & 31)];
                __syncthreads(); // every wave's stores have left the CU (vmcnt(0) + barrier; the L1 is write-through)
                if (tk == 0) __hip_atomic_store(ready, (unsigned)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            AGPL_TS(5);
            if (LA && Mp > 0) {
                // ---- look-ahead on the spine: apply THIS step's update to the next block column of T (diagonal block
                //      k + 1 and the panel below it) and to the next 32 rows of the eliminated identity, straight into
                //      the LDS layout of step k + 1.  Their updates of steps < k came from the tile workgroups, which
                //      never touch them again (they work two block columns ahead); those of step k - 1 are the tiles
                //      they do first and count in `crit` -- a whole spine step ago, so this wait is normally over.
                const int nbp = Mp / 32, ncb = ncx / 32; // nbp + ncb = nb - 1 <= 15 row blocks: one wave each
                if (k > 0) {
                    if (tk == 0 && !spin_until_ge(crit, (unsigned)((NW - 1) * k))) lost = 1;
                    __syncthreads();
                    if (lost) break;
                }
                AGPL_TS(6);
                // wave w < nbp: rows 32 w.. of P against the first 32 rows of P (their copy in Rs), old values from T,
                // result = rows of the next diagonal block (w = 0, into Ds) or of the next raw panel (in place: the
                // same LDS rows, which only this wave reads).  Wave nbp + cb: rows 32 cb.. of X_k' against the same,
                // old values from U (column-major), result = the next raw rows of the eliminated identity, in place.
                if (wave < nbp + ncb) {
                    const bool ta = wave < nbp;
                    const int hb = ta ? wave : wave - nbp;
                    const double *oldp = ta ? (k == 0 ? G : T) + (size_t)(kb + FB + 32 * hb) * M + kb + FB
                                            : A + (size_t)(32 * hb) * M + kb + FB;
                    const bool zero_old = !ta && 32 * hb >= kb; // column block k of U: nothing eliminated into it yet
                    const bool tri = ta && hb == 0;
                    double *rowsU = tri ? Rs : PX + (size_t)(ta ? ncx + 32 * hb : 32 * hb) * FP;
                    double *dst = tri ? Ds : rowsU;
                    const unsigned lo_ = (unsigned)(ln >> 4) * (unsigned)M + (unsigned)(ln & 15);
                    d4 acc[2][2], old[2][2];
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj) {
                            acc[ti][tj] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                old[ti][tj][r] = oldp[(unsigned)(16 * ti + 4 * r) * (unsigned)M + 16 * tj + lo_];
                        }
                    macro_mac<false>(rowsU, Rs, ln, acc);
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = 16 * ti + 4 * r + (ln >> 4), j = 16 * tj + (ln & 15);
                                double val = (zero_old ? 0.0 : old[ti][tj][r]) - acc[ti][tj][r];
                                if (tri) val = j < i ? val : (j == i ? val + (k == 0 ? 1.0 : 0.0) : 0.0); // T = I + G
                                dst[i * FP + j] = val;
                            }
                }
                // block (k + 1, k + 1) of the eliminated identity is still the identity
                // (rows ncx.. of PX held the first 32 rows of P: the waves above read their copy in Rs)
                PX[(size_t)(ncx + (tk >> 5)) * FP + (tk & 31)] = (tk >> 5) == (tk & 31) ? 1.0 : 0.0;
                __syncthreads();
                AGPL_TS(7);
            }
        } else {
            if (LA && Mp < 2 * FB) break; // the tile workgroups work two block columns ahead of the spine
            AGPL_TS(0);
            if (tk == 0 && !spin_until_ge(ready, (unsigned)(k + 1))) lost = 1;
            __syncthreads();
            if (lost) break;
            AGPL_TS(1);
            const double *PXg = PXg0 + (LA ? (size_t)(k & 1) * M * FB : 0);
#pragma unroll 1
            for (int i0 = tk; i0 < M * FB; i0 += 8 * 1024) { // 8 loads in flight per thread
                double tmp[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) tmp[u] = PXg[min(i0 + u * 1024, M * FB - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = i0 + u * 1024;
                    if (idx < M * FB) PX[(size_t)(idx >> 5) * FP + (idx & 31)] = tmp[u];
                }
            }
            __syncthreads();
            AGPL_TS(2);
        }
        if (Mp > 0) {
            // ---- 32 x 16 wave tiles: (a) T[i][j] -= P_i . P_j over the lower block triangle (16 wb <= 32 ub + 31);
            //                          (b) RHS[i][c] -= X_k'[c] . P_i for every 32-column block cb (U is column-major:
            //                              the lanes of a result run along i).
            // The old values are loaded first (uniform tile origin + one 32-bit ln offset): their latency hides
            // behind the 16 MFMAs.
            const int nbp = Mp / 32, ncb = ncx / 32;
            const int ntile_a = nbp * (nbp + 1), ntile_b = ncb * 2 * nbp;
            const unsigned lo_ = (unsigned)(ln >> 4) * (unsigned)M + (unsigned)(ln & 15);
            for (int pass = 0; pass < (LA ? 2 : 1); ++pass) {
            if (LA && wg == 0) break; // look-ahead: the spine workgroup does no tiles
            int ub = 0;
            for (int w = (LA ? 0 : wg * 16) + wave; w < ntile_a + ntile_b; w += (LA ? 1 : NW) * 16) {
                d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}}, old[2];
                if (w < ntile_a) {
                    while ((ub + 1) * (ub + 2) <= w) ++ub;
                    const int wb = w - ub * (ub + 1); // 16-column block, 0 .. 2 ub + 1
                    if (LA && wb < 2) continue;                  // the next block column is the spine's own
                    if (LA && (wb < 4) != (pass == 0)) continue; // pass 0: the block column the spine takes next step
                    // look-ahead with several tile workgroups: a location keeps its owner across steps (absolute
                    // 32-row / 16-column block coordinates), so no workgroup ever waits for another one's update
                    if (LA && NW > 2 && (k + 1 + ub + 2 * (k + 1) + wb) % (NW - 1) != wg - 1) continue;
                    const size_t to = (size_t)(kb + FB + 32 * ub) * M + kb + FB + 16 * wb;
                    double *tp = T + to;
                    const double *tsrc = (k == 0 ? G : T) + to;
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) old[ti][r] = tsrc[(unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_];
                    mac_2x1(PX + (size_t)(ncx + 32 * ub) * FP, PX + (size_t)(ncx + 16 * wb) * FP, ln, acc);
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ip = 32 * ub + 16 * ti + 4 * r + (ln >> 4), jp = 16 * wb + (ln & 15);
                            // (elements above the diagonal of a diagonal tile were read too: inside T, unused)
                            if (jp <= ip)
                                tp[(unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_] =
                                    old[ti][r] - acc[ti][r] + (k == 0 && jp == ip ? 1.0 : 0.0);
                        }
                } else {
                    const int wbi = w - ntile_a;
                    const int cb = wbi / (2 * nbp), ib = wbi - cb * 2 * nbp; // 32 columns c x 16 rows i
                    if (LA && ib < 2) continue;                  // (the same for the rows of the eliminated identity)
                    if (LA && (ib < 4) != (pass == 0)) continue;
                    if (LA && NW > 2 && (cb + 2 * (k + 1) + ib) % (NW - 1) != wg - 1) continue;
                    double *ap = A + (size_t)(32 * cb) * M + kb + FB + 16 * ib;
                    const bool fresh = 32 * cb >= kb; // column block k: nothing eliminated into it yet
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            old[ti][r] = fresh ? 0.0 : ap[(unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_];
                    mac_2x1(PX + (size_t)(32 * cb) * FP, PX + (size_t)(ncx + 16 * ib) * FP, ln, acc);
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            ap[(unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_] = old[ti][r] - acc[ti][r];
                }
            }
            if (LA && pass == 0) {
                __syncthreads();
                if (tk == 0) __hip_atomic_fetch_add(crit, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                AGPL_TS(3);
            }
            }
        }
        if (LA) {
            if (wg != 0) {
                __syncthreads();
                if (tk == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                AGPL_TS(4);
            }
        } else if (NW > 1) {
            __syncthreads(); // as above: one ln's agent-scope release then covers the whole workgroup's stores
            if (tk == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();
        }
    }
    if (NW > 1) {
        if (wg != 0) return;
        // U is complete once every workgroup has finished the last step's tiles
        if (tid == 0 && !lost && (!LA || nb > 2) && !spin_until_ge(done, (unsigned)(LA ? (NW - 1) * (nb - 2) : NW * nb))) lost = 1;
        __syncthreads();
        if (lost) {
            if (tid == 0) info[l] = -1; // a partner workgroup never arrived: reported as an internal error by the host
            return;
        }
    }

    // ---- log det(I + G) = sum of the log pivots (fixed order: lanes of wave 0 over a stride, then a shuffle tree)
    if (wave == 0) {
        double acc = 0.0;
        for (int i = lane; i < M; i += 64) acc += log(PX[(size_t)i * FP + FB]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) ldsum = acc;
    }
    __syncthreads();
    // ---- v = U (g + eta0):  v[a] = sum_{b <= a} A[b * M + a] r[b].  Wave w takes b = w, w + 16, ...; lanes run along
    //      a (coalesced); the 16 partial sums of an a are combined in wave order (fixed order: reproducible)
    {
        double *rs = PX;     // [M]
        double *part = PX + M; // [16][M]
        const double *g = gall + (size_t)l * M;
        for (int b = tid; b < M; b += 1024) rs[b] = g[b] + (eta0all ? eta0all[(size_t)l * M + b] : 0.0);
        __syncthreads();
        for (int a = lane; a < M; a += 64) {
            double acc = 0.0;
#pragma unroll 4
            for (int b = wave; b <= a; b += 16) acc += A[(size_t)b * M + a] * rs[b];
            part[wave * M + a] = acc;
        }
        __syncthreads();
        for (int a = tid; a < M; a += 1024) {
            double acc = 0.0;
#pragma unroll
            for (int w = 0; w < 16; ++w) acc += part[w * M + a];
            if (vall) vall[(size_t)l * M + a] = acc;
            if (v32all) v32all[(size_t)l * M + a] = (float)acc;
        }
    }
    if (tid == 0) {
        if (logdet) logdet[l] = ldsum;
        info[l] = bad;
    }
}

// =====================================================================================================================
// factor_pipe_kernel (round 5): the same factorisation as factor_kernel, M % 128 == 0, M <= 1024, ONE launch, with the serial
// chain cut down to the diagonal blocks.  Per latent the workgroups take fixed roles:
//
//   F  (1 workgroup)   the diagonal chain: D_k -> R_kk = chol(D_k), W_k = R_kk^-1 (the 16 column-pair steps of factor_kernel),
//                      publishes W_k, then P0 = T[k+1,k] W_k' and D_{k+1} = T[k+1,k+1] - P0 P0' from two 32 x 32 blocks that a P
//                      workgroup handed over a whole step earlier.  Nothing else is on its path: 12-13 us per block step (factor
//                      8.2, publish 0.45, wait for the hand-over 1-3, the two blocks 0.65, P0 and D 0.8, 0.55 to the next step:
//                      -DAGPL_FTRACE stamps) instead of the ~21 us of factor_kernel's spine.
//   P  (ceil(M / 256)) the block column: each owns 256 rows of the LDS image PX (row c < 32 (k + 1): X_k'[c], column c of U; row
//                      g >= 32 (k + 1): the raw panel T[g, k]), two images by step parity.  Waves 0-7 (owners): load W_k, every row
//                      times W_k' in place (the panel of R and the final rows of U, which go to A), then this step's update of the
//                      NEXT block column and of the next 32 rows of the eliminated identity into the other image (factor_kernel's
//                      look-ahead; the old values by write-through loads).  The owner whose rows are block k + 2 stores
//                      T[k+2,k+1] for F.  Waves 8-15 (helpers), beside the owners' look-ahead: fetch P0, publish the
//                      workgroup's rows (LDS-staged 16-byte write-through stores), update the diagonal tiles T[j,j] of its
//                      rows (the one for F write-through), drain; the last helper to finish stores `ready[j]`.
//   T  (CY^2 x TS)     the trailing update: cell (tr, tc) of a CY x CY block-cyclic grid over the 32 x 32 tiles (CY = 4; 2 where
//                      several latents share an XCD at M <= 512), TS = 2 workgroups per cell beyond M = 512 (they split a cell's
//                      row blocks by parity).  A cell holds the strictly-lower tiles (ib, jb) of T with ib = tr, jb = tc (mod CY)
//                      and the tiles of the eliminated identity with row block = tr, column block = tc; it stages the two
//                      residue classes of P | X' it needs (<= 2 x 256 rows).  Two passes per step: first the block column / row
//                      P takes over at its next step (one tile per wave, stored write-through, then `crit[w]`), then the rest on
//                      a skewed walk that gives every wave a mix of long and short columns.  Cyclic, not contiguous regions:
//                      every workgroup keeps its share of the work to the last steps.
//
// Hand-offs inside the launch follow the local guide's recipe (cdna_hip_programming.md Guideline 16):
//   * everything another workgroup reads inside the launch -- W_k, P0, the blocks for F, P | X', the first-pass tiles of T and U -- is
//     stored write-through (sc1: relaxed agent-scope atomic stores; 16-byte inline-asm stores where a wave has pairs), every
//     storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at a barrier (or counts its helper waves in LDS), ONE lane
//     stores the flag; every load of those bytes in the consumer is an sc1 load behind ONE relaxed poll: no fence on either side
//     (an acquire fence in a wave with publish stores in flight waits for its own drain: measured, 4 us per step);
//   * second-pass tiles stay with the workgroup that wrote them (plain stores and loads through its own L2; a cell's parts keep the
//     row blocks they own in BOTH passes -- with chip-wide placement another part's plain stores sit in another XCD's L2);
//   * the end of the factorisation is one agent-scope release add per workgroup (`alldone`) and an acquire behind its poll; then
//     every workgroup takes its 64-wide blocks of v = U (g + eta0).
// Placement: M <= 512: the <= 19 workgroups of a latent on the XCD l % 8 (8 x wide launch, 7 of 8 leave at once), so that the
// hand-offs stay in one L2; beyond, up to 37 workgroups a latent do not fit 32 CUs: chip-wide (`spread`), every hand-off agent-scope.
// Nothing is reused inside a launch (W, P0, P | X' have a slot per step: M x M doubles of scratch), so there is no
// write-after-read hazard to reason about; the flag words are zeroed by the clean-up launch queued behind every launch.
// Arithmetic: every element sees exactly factor_kernel's sequence of float64 MFMA accumulations and subtractions, in the same
// order -- the results are bit for bit those of factor_kernel<1, false> (the rescue path and tests rely on it).
// =====================================================================================================================
// Measured as builds in round 5 and not shipped: (1) the T workgroups' second pass in batches of 2 / 4 tiles whose old values are in
// flight together (to hide the ~1 us round trip per tile): at the 128-VGPR cap the compiler spills the loaded values (172 / 660
// bytes of scratch) and reloads them one by one; (2) every owner wave of P polling the T workgroups' first passes itself and issuing
// the look-ahead's old-value loads BEFORE the barrier behind the panel (its poll first waits for its own publish stores to drain):
// same box, 100 updates each, 0.206-0.215 / 0.426-0.428 ms as shipped against 0.207-0.221 / 0.424-0.428 at M = 512 / 1024; (3) the same
// with the loads in front of the publish stores (`ready` comes later: 0.226 / 0.437); (4) the hand-over flag stored by whichever of its
// two waves drains last (a pair through one LDS word) instead of behind the workgroup's end-of-step barrier: 0.210 / 0.421, no change.
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ double ld_sc1(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const u64 *)p, RLX_AGENT));
}
__device__ __forceinline__ void st_sc1(double *p, double x) {
    __hip_atomic_store((u64 *)p, (u64)__double_as_longlong(x), RLX_AGENT);
}
// ONE lane polls ONE word, relaxed, bounded (a lost partner must not hang the device)
__device__ __forceinline__ bool poll_ge(unsigned *p, unsigned target) {
    for (int it = 0; it < (1 << 22); ++it) {
        if (__hip_atomic_load(p, RLX_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}
#define AGPL_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

struct PipeFlags { // one 192-byte record per latent (at most 32 latents take this form), zero between launches.  Every word has ONE writer (or is a final count):
                   // a sum over producers could be reached by a fast one running a step ahead of a slow one
    unsigned wready;   // F: W_k published                                 (value k + 1)
    unsigned p0ready;  // F: P0 of step k published                        (value k + 1)
    unsigned hand;     // P: T[k+2,k+1], T[k+2,k+2] of step k stored        (value k + 1)
    unsigned alldone;  // every workgroup: U complete                       (+1 each; release)
    unsigned lost;     // any workgroup: a partner never arrived (which wait: diagnostic)
    unsigned ready[4]; // P workgroup j: its rows of step k published       (value k + 1)
    unsigned pad[3];
    unsigned crit[36]; // T workgroup w (<= 32): first pass of step k finished (value k + 1)
};

// 16-byte write-through store (global_store_dwordx4 ... sc1): the 8-byte form costs 2.7 x per byte on the fabric
// (MI355X_MICROARCH.md, visibility table) -- publishing a block column by 8-byte sc1 stores took 12 us of P's 20 us step
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_sc1_x2(double *p, double a, double b) {
#ifdef AGPL_PIPE_PUB8
    st_sc1(p, a);
    st_sc1(p + 1, b);
#else
    d2v v = {a, b};
    // (s_nop 1: a VMEM store of more than 8 bytes still reads its data registers for two cycles after issue, and the compiler's
    //  hazard recogniser does not look inside inline asm: without it the next VALU write of those registers corrupted the low
    //  words of the stored doubles -- 2e-7 relative errors in U; cdna_hip_programming.md 5.7)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}

template <int DUMMY>
__global__ __launch_bounds__(1024, 1) void factor_pipe_kernel(int M, int NP, int CY, int TS, int spread, const double *__restrict__ Gall,
                                                              const double *__restrict__ gall,
                                                              const double *__restrict__ eta0all, double *__restrict__ Tall,
                                                              double *__restrict__ Aall, double *__restrict__ vall,
                                                              float *__restrict__ v32all, double *__restrict__ logdet,
                                                              int *__restrict__ info, double *__restrict__ scratch_all,
                                                              PipeFlags *__restrict__ flags_all, int rescue) {
    // spread == 0: latent l works on XCD l % 8 (as factor_kernel: the launch is 8 x wide and 7 of 8 workgroups leave at once);
    // spread == 1 (M > 512: up to 37 workgroups a latent): wherever the dispatcher puts them -- every hand-off is agent-scope
    if (!spread && (blockIdx.x & 7) != (blockIdx.y & 7)) return;
    const int l = (int)blockIdx.y;
    const int wg = spread ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
    if (rescue == 2) { // fault injection (agpl_debug_force_factor_rescue): behave as if a partner never arrived
        if (wg == 0 && threadIdx.x == 0) info[l] = -1;
        return;
    }
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int bad, lostf;
    __shared__ double ldsum;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = M / FB;
    const int NT = CY * CY * TS, NWG = 1 + NP + NT; // T workgroups: a CY x CY (4 x 4, or 2 x 2) block-cyclic grid over the 32 x 32
                                                    // tiles, TS (1, 2) workgroups per cell
    const double *G = Gall + (size_t)l * M * M;
    double *T = Tall + (size_t)l * M * M;
    double *A = Aall + (size_t)l * M * M;
    // scratch of this latent: W [nb][32][32] | P0 [nb][32][32] | PXg [nb][M][32] | pivots [M]
    double *Wg = scratch_all + (size_t)l * ((size_t)2 * nb * FB * FB + (size_t)M * M + M);
    double *P0g = Wg + (size_t)nb * FB * FB;
    double *PXg = P0g + (size_t)nb * FB * FB;
    PipeFlags *fl = flags_all + l;
    if (tid == 0) {
        bad = 0;
        lostf = 0;
        ldsum = 0.0;
    }
    __syncthreads();

    if (wg == 0) {
        // =================================================================================== F: the diagonal chain
        double *Ds = sm;              // [32][FP] the diagonal block being eliminated
        double *Ys = Ds + FB * FP;    // [32][FP] the identity being eliminated alongside
        double *Wf = Ys + FB * FP;    // [32][FP] W = R_kk^-1
        double *T1 = Wf + FB * FP;    // [32][FP] T[k+1, k]   (raw rows of the next block, column block k)
        double *T2 = T1 + FB * FP;    // [32][FP] T[k+1, k+1] (the next diagonal block before this step's update)
        double *Ps = T2 + FB * FP;    // [32][FP] P0 = T1 W'
        double *piv = Ps + FB * FP;   // [M] pivots (log det at the end)
        {   // D_0 = (I + G)[0:32, 0:32], lower triangle
            const int r = tid >> 5, c = tid & 31;
            const double dval = G[(size_t)r * M + c];
            Ds[r * FP + c] = c < r ? dval : (c == r ? dval + 1.0 : 0.0);
        }
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            int tk = tid, ln = lane;
            asm volatile("" : "+v"(tk), "+v"(ln));
            const int kb = k * FB;
            AGPL_TS(0);
            // ---- R_kk = chol(D), W = R_kk^-1: factor_kernel's sixteen column-pair steps (the same routine)
            block_factor_32(tk, Ds, Ys, Wf, [&](int c, double p0, double p1) {
                piv[kb + c] = p0;
                piv[kb + c + 1] = p1;
                if (!(p0 > 0.0)) bad = bad ? bad : kb + c + 1;
                else if (!(p1 > 0.0)) bad = bad ? bad : kb + c + 2;
            });
            __syncthreads();
            AGPL_TS(1);
            // ---- publish W_k (write-through); fetch what the next diagonal block needs
            st_sc1(Wg + (size_t)k * FB * FB + tk, Wf[(tk >> 5) * FP + (tk & 31)]);
            const bool more = k + 1 < nb;
            double t1 = 0.0, t2 = 0.0;
            if (more && k == 0) { // nothing has been updated before step 0: straight from G (+ I)
                const int r = tk >> 5, c = tk & 31;
                t1 = G[(size_t)(FB + r) * M + c];
                t2 = G[(size_t)(FB + r) * M + FB + c]; // (T = I + G: the 1 is added behind the subtraction, as factor_kernel does)
            }
            AGPL_DRAIN(); // W stores (and the k == 0 loads) have completed
            __syncthreads();
            if (tk == 0) __hip_atomic_store(&fl->wready, (unsigned)(k + 1), RLX_AGENT);
            AGPL_TS(2);
            if (!more) break;
            if (k > 0) {
                // T[k+1, k] and T[k+1, k+1] as P's step k - 1 left them (a whole step ago: normally long there)
                if (tk == 0 && !poll_ge(&fl->hand, (unsigned)k)) lostf = 1;
                __syncthreads();
                AGPL_TS(3);
                if (lostf) break;
                const int r = tk >> 5, c = tk & 31;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (the loads stay below the poll; every one is sc1)
                t1 = ld_sc1(T + (size_t)(kb + FB + r) * M + kb + c);
                t2 = ld_sc1(T + (size_t)(kb + FB + r) * M + kb + FB + c);
            }
            {
                const int r = tk >> 5, c = tk & 31;
                T1[r * FP + c] = t1;
                T2[r * FP + c] = t2;
            }
            __syncthreads();
            AGPL_TS(4);
            // P0 = T1 W' (waves 0..3: one 16 x 16 tile each; W lower triangular: the tj = 0 tiles stop at column 15)
            if (wave < 4) {
                const int ti = wave >> 1, tj = wave & 1;
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
                const double *u = T1 + (16 * ti + (ln & 15)) * FP + (ln >> 4), *v = Wf + (16 * tj + (ln & 15)) * FP + (ln >> 4);
#pragma unroll
                for (int kk = 0; kk < FB / 4; ++kk)
                    if (tj == 1 || kk < FB / 8) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(u[4 * kk], v[4 * kk], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ti + 4 * r + (ln >> 4), m = 16 * tj + (ln & 15);
                    Ps[i * FP + m] = acc[r];
                    st_sc1(P0g + (size_t)k * FB * FB + i * FB + m, acc[r]);
                }
            }
            AGPL_DRAIN();
            __syncthreads();
            if (tk == 0) __hip_atomic_store(&fl->p0ready, (unsigned)(k + 1), RLX_AGENT);
            AGPL_TS(5);
            // D_{k+1} = T2 - P0 P0' (lower triangle; upper zero)
            if (wave < 4) {
                const int ti = wave >> 1, tj = wave & 1;
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
                const double *u = Ps + (16 * ti + (ln & 15)) * FP + (ln >> 4), *v = Ps + (16 * tj + (ln & 15)) * FP + (ln >> 4);
#pragma unroll
                for (int kk = 0; kk < FB / 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(u[4 * kk], v[4 * kk], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * ti + 4 * r + (ln >> 4), j = 16 * tj + (ln & 15);
                    const double val = T2[i * FP + j] - acc[r];
                    Ds[i * FP + j] = j < i ? val : (j == i ? val + (k == 0 ? 1.0 : 0.0) : 0.0);
                }
            }
            __syncthreads();
        }
        // log det(I + G) = sum of the log pivots (fixed order), the outcome word
        if (wave == 0 && !lostf) {
            double acc = 0.0;
            for (int i = lane; i < M; i += 64) acc += log(piv[i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) ldsum = acc;
        }
        __syncthreads();
    } else if (wg <= NP) {
        // =================================================================================== P: the block column
        // 256 rows per workgroup: waves 0..7 OWN 32 rows each (panel, look-ahead), waves 8..15 HELP (wave w + 8 publishes the rows
        // of wave w, updates their diagonal tile, waits for the write-through and signals) -- so the ~4 us a block column takes to
        // reach the fabric, and the diagonal tiles, run beside the look-ahead instead of in front of it.  The rows live in two LDS
        // images: the panel of step k is formed in place in image k & 1 (read by the helpers), the look-ahead writes the raw rows
        // of step k + 1 into the other one.
        const int j = wg - 1, R0 = 256 * j;           // this workgroup's rows: global R0 .. R0 + 255 (< M)
        const int nrow = min(256, M - R0);
        double *PXbuf = sm;                        // [2][256][FP]
        double *Wf = PXbuf + (size_t)2 * 256 * FP; // [32][FP]
        double *Rs = Wf + FB * FP;                 // [32][FP] P0
        __shared__ unsigned pubcnt;                // owner waves whose published rows have reached the fabric (monotonic)
        if (tid == 0) pubcnt = 0u;
        // ---- stage step 0: rows R >= 32: G[R][0:32] (T = I + G, off the diagonal block); rows R < 32: the identity block
        for (int idx = tid; idx < nrow * FB; idx += 1024) {
            const int r = idx >> 5, c = idx & 31, R = R0 + r;
            PXbuf[(size_t)r * FP + c] = R >= FB ? G[(size_t)R * M + c] : (R == c ? 1.0 : 0.0);
        }
        __syncthreads();
        const bool owner = wave < 8;
        const int ow = wave & 7;                 // the owner wave this wave is or helps
        const int myR = R0 + 32 * ow;            // first global row of its 32 rows
        const bool have = 32 * ow < nrow;
        const int nhelp = (nrow + 31) / 32;      // helpers with rows
        for (int k = 0; k < nb; ++k) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int kb = k * FB, ncx = kb + FB, Mp = M - ncx;
            const int Rb = myR >> 5; // global 32-row block of the rows
            double *rows = PXbuf + (size_t)(k & 1) * 256 * FP + (size_t)ow * 32 * FP;        // this step's rows (panel in place)
            double *rowsN = PXbuf + (size_t)((k + 1) & 1) * 256 * FP + (size_t)ow * 32 * FP; // the raw rows of step k + 1
            const bool xrows = myR < ncx;
            const bool pub = Mp >= 2 * FB;       // (no T workgroup reads the last two steps)
            const unsigned lo_ = (unsigned)(ln >> 4) * (unsigned)M + (unsigned)(ln & 15);
            // ---- W_k
            AGPL_TS(0);
            if (tid == 0 && !poll_ge(&fl->wready, (unsigned)(k + 1))) lostf = 3;
            __syncthreads();
            AGPL_TS(1);
            if (lostf) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            Wf[(tid >> 5) * FP + (tid & 31)] = ld_sc1(Wg + (size_t)k * FB * FB + tid);
            // helpers: the old values of the diagonal tile, on their way while the panel is formed (the owners' come behind the panel:
            // they wait for the T workgroups)
            const bool upd = owner && have && Mp > 0 && Rb != k + 1;
            const bool dgt = !owner && have && !xrows && Rb >= k + 2; // (P owns every diagonal tile; Rb == k + 1 is F's)
            d4 uold[2][2];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) uold[ti][tj] = (d4){0.0, 0.0, 0.0, 0.0};
            if (dgt) { // P's own data (this wave wrote it last step): plain loads
                const double *oldp = (k == 0 ? G : T) + (size_t)myR * M + myR;
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) uold[ti][tj][r] = oldp[(unsigned)(16 * ti + 4 * r) * (unsigned)M + 16 * tj + lo_];
            }
            __syncthreads();
            AGPL_TS(2);
            // ---- owners: every row times W': P (panel of R below the block) and X_k' (rows kb .. kb + 31 of U, final)
            if (owner && have) {
                d4 acc[2][2];
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = (d4){0.0, 0.0, 0.0, 0.0};
                macro_mac<true>(rows, Wf, ln, acc);
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * ti + 4 * r + (ln >> 4), m = 16 * tj + (ln & 15);
                            const double val = acc[ti][tj][r];
                            rows[i * FP + m] = val;
                            if (xrows) {
                                const int c = myR + i;
                                if (c <= kb + m) A[(size_t)c * M + kb + m] = val; // final rows of U
                            }
                        }
                if (pub) {
                    // publish its own 32 rows at once (8 x (64 lanes x 16 bytes), write-through, re-read from LDS as pairs: the wave's
                    // own LDS writes, in order).  Not waited for here: the stores drain beside the wait for P0 / the T workgroups and
                    // the look-ahead; the signal goes out behind the look-ahead.  (Until round 5's last pass the HELPERS published,
                    // behind the barrier below: `ready` then came 6.3 us behind it instead of 3, and through the T workgroups' first
                    // pass that wait closed a 13.4 us cycle around this barrier; now F's 10.7 us chain is the longest.)
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    double *pubp = PXg + ((size_t)k * M + myR) * FB;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = 2 * (64 * q + ln); // element pair (row e >> 5, columns e & 31, + 1)
                        st_sc1_x2(pubp + e, rows[(e >> 5) * FP + (e & 31)], rows[(e >> 5) * FP + (e & 31) + 1]);
                    }
                }
            }
            if (!owner && Mp > 0) {
                // helpers, beside the panel: P0 of this step (F publishes it ~2 us behind W_k) into LDS -- every helper wave polls
                // for itself and loads its eighth; wave 8 also waits for the T workgroups' first passes of step k - 1 (the
                // look-ahead's old values: stored write-through there, read by sc1 loads here -- no fence on either side)
                bool ok = true;
                if (lane == 0) ok = poll_ge(&fl->p0ready, (unsigned)(k + 1));
                else if (wave == 8 && lane >= 16 && lane < 16 + NT && k > 0) ok = poll_ge(&fl->crit[lane - 16], (unsigned)k);
                if (!__all(ok) && lane == 0) lostf = 4;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = (wave - 8) * 128 + 64 * u + lane; // 1024 elements over 8 waves
                    Rs[(e >> 5) * FP + (e & 31)] = ld_sc1(P0g + (size_t)k * FB * FB + e);
                }
            }
            __syncthreads(); // the panel rows and P0 are in LDS
            AGPL_TS(3);
            if (Mp <= 0) break; // last step: the final rows of U are written (drained behind the loop)
            if (lostf) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (upd && !(xrows && myR >= kb)) { // (column block k of U: nothing eliminated into it yet -> zeros)
                const double *oldp = !xrows ? (k == 0 ? G : T) + (size_t)myR * M + ncx : A + (size_t)myR * M + ncx;
                if (k == 0) {
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) uold[ti][tj][r] = oldp[(unsigned)(16 * ti + 4 * r) * (unsigned)M + 16 * tj + lo_];
                } else {
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                uold[ti][tj][r] = ld_sc1(oldp + (unsigned)(16 * ti + 4 * r) * (unsigned)M + 16 * tj + lo_);
                }
            }
            AGPL_TS(4);
            if (upd) {
                // ---- owners, look-ahead: this step's update of the NEXT block column of T and of the next 32 rows of the
                //      eliminated identity, into the other LDS image = the layout of step k + 1
                d4 acc[2][2];
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = (d4){0.0, 0.0, 0.0, 0.0};
                macro_mac<false>(rows, Rs, ln, acc);
                const bool crit_wave = !xrows && Rb == k + 2; // F needs these rows (T[k+2, k+1]) at the end of ITS next step
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 16 * ti + 4 * r + (ln >> 4), jj = 16 * tj + (ln & 15);
                            const double val = uold[ti][tj][r] - acc[ti][tj][r];
                            rowsN[i * FP + jj] = val;
                            if (crit_wave) st_sc1(T + (size_t)(myR + i) * M + ncx + jj, val);
                        }
            } else if (owner && have) {
                // rows ncx .. ncx + 31 held the first 32 rows of P: block (k + 1, k + 1) of the eliminated identity is the identity
                for (int e = ln; e < FB * FB; e += 64) rowsN[(e >> 5) * FP + (e & 31)] = (e >> 5) == (e & 31) ? 1.0 : 0.0;
            }
            if (owner && have) {
                // its published rows (and, for the wave of block k + 2, its half of the hand-over to F) have reached the fabric; the
                // owner whose rows arrive last signals for the workgroup
                AGPL_DRAIN();
                if (pub && ln == 0) {
                    const unsigned n = __hip_atomic_fetch_add(&pubcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
                    if (n == (unsigned)(nhelp * (k + 1))) __hip_atomic_store(&fl->ready[j], (unsigned)(k + 1), RLX_AGENT);
                }
            } else if (!owner && have) {
                // ---- helpers: the diagonal tile of their rows
                if (dgt) { // T[Rb,Rb] -= P_k[Rb] P_k[Rb]'; the tile of block k + 2 goes to F next step: write-through
                    d4 dg[2][2];
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj) dg[ti][tj] = (d4){0.0, 0.0, 0.0, 0.0};
                    macro_mac<false>(rows, rows, ln, dg);
                    const size_t to = (size_t)myR * M + myR;
                    const bool toF = Rb == k + 2;
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = 16 * ti + 4 * r + (ln >> 4), jj = 16 * tj + (ln & 15);
                                if (jj <= i) {
                                    double *dst = T + to + (unsigned)(16 * ti + 4 * r) * (unsigned)M + 16 * tj + lo_;
                                    const double val = uold[ti][tj][r] - dg[ti][tj][r] + (k == 0 && jj == i ? 1.0 : 0.0);
                                    if (toF) st_sc1(dst, val);
                                    else *dst = val;
                                }
                            }
                }
                AGPL_DRAIN(); // its diagonal tile (the one of block k + 2: F's) has reached the fabric
            }
            __syncthreads(); // (Wf, Rs and image k & 1 are free again; both halves of the hand-over have drained)
            if (tid == 0 && Mp >= 2 * FB && ((ncx + FB) >> 8) == j) __hip_atomic_store(&fl->hand, (unsigned)(k + 1), RLX_AGENT);
            AGPL_TS(6);
        }
    } else {
        // =================================================================================== T: the trailing update
        // Workgroup (tr, tc) of a CY x CY block-cyclic grid over the 32 x 32 tiles: the strictly-lower tiles (ib, jb) of T with
        // ib = tr, jb = tc (mod CY) (the diagonal tiles and block column k + 1 are P's), and the tiles of the eliminated identity
        // with row block ib = tr and column block cb = tc (mod CY).  Cyclic: every workgroup keeps its share of the work to the
        // last steps (contiguous regions left most of them idle half-way, and the bottom ones with 25 us per step at M = 1024).
        // It stages the 2 x nb / CY row blocks of P | X' it needs: <= 2 x 256 rows (CY = 2 only for M <= 512).
        const int tw = wg - 1 - NP;            // 0 .. CY^2 TS - 1
        const int cell = tw / TS, part = tw - cell * TS; // TS workgroups share a cell: they split its tiles
        const int tr = cell / CY, tc = cell - tr * CY;
        const int nq = nb / CY;                // row blocks per residue class (ib = CY q + tr); nb % 4 == 0
        double *PXa = sm;                      // [nq][32][FP] row blocks ib = tr (mod 4)
        double *PXb = tr == tc ? PXa : PXa + (size_t)nq * FB * FP; // [nq][32][FP] row blocks = tc (mod 4)
        for (int k = 0; k + 2 < nb; ++k) { // (the last two steps leave no tile: block column k + 1 is P's)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            AGPL_TS(0);
            if (wave == 0) {
                bool ok = true;
                if (lane < NP) ok = poll_ge(&fl->ready[lane], (unsigned)(k + 1));
                if (!__all(ok) && lane == 0) lostf = 5;
            }
            __syncthreads();
            AGPL_TS(1);
            if (lostf) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {   // stage: thread t takes 16-byte pairs; 8 loads in flight per thread and class
                const double *src = PXg + (size_t)k * M * FB;
                const int npair = nq * FB * FB / 2;
#pragma unroll 1
                for (int i0 = tid; i0 < npair; i0 += 4 * 1024) {
                    double ta_[4][2], tb_[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e = 2 * min(i0 + u * 1024, npair - 1); // element (q, row, col) of the class image
                        const int q = e >> 10, rc = e & 1023;
                        const int ba = min(CY * q + tr, nb - 1), bb = min(CY * q + tc, nb - 1);
                        ta_[u][0] = ld_sc1(src + (size_t)ba * FB * FB + rc);
                        ta_[u][1] = ld_sc1(src + (size_t)ba * FB * FB + rc + 1);
                        tb_[u][0] = tr == tc ? 0.0 : ld_sc1(src + (size_t)bb * FB * FB + rc);
                        tb_[u][1] = tr == tc ? 0.0 : ld_sc1(src + (size_t)bb * FB * FB + rc + 1);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = i0 + u * 1024;
                        if (idx < npair) {
                            const int e = 2 * idx, row = e >> 5, col = e & 31;
                            PXa[(size_t)row * FP + col] = ta_[u][0];
                            PXa[(size_t)row * FP + col + 1] = ta_[u][1];
                            if (tr != tc) {
                                PXb[(size_t)row * FP + col] = tb_[u][0];
                                PXb[(size_t)row * FP + col + 1] = tb_[u][1];
                            }
                        }
                    }
                }
            }
            __syncthreads();
            AGPL_TS(2);
            const unsigned lo_ = (unsigned)(ln >> 4) * (unsigned)M + (unsigned)(ln & 15);
            // Tiles per wave.  (a) 32 rows (block ib = CY qa + tr) x 16 columns (half h of block jb = CY qb + tc), jb < ib, jb >= k + 2;
            // (b) 32 columns of U (block cb = CY qb + tc <= k) x 16 rows (half h of block ib = CY qa + tr >= k + 2).  First pass:
            // (a) with jb == k + 2 and (b) with ib == k + 2 -- what the P workgroups take over at their next step: ONE block
            // column / row, its <= 2 nq tiles dealt one per wave, stored write-through (P reads them by sc1 loads: no fence on
            // either side).  Second pass: wave w takes, for every q, the half-column hb = (w - 3 q) mod 16 -- each wave a mix of
            // long and short columns of the triangle, no division in the walk.  (Walking t = w, w + 16, ... over the rectangle gave a
            // wave one fixed half-column: the first pass ran on 2 waves, 6-10 us.)
            const int ncol = 2 * nq;
            for (int pass = 0; pass < 2; ++pass) {
                for (int it = 0; it < (pass == 0 ? 1 : nq); ++it) {
                    int qa, qb, h;
                    if (pass == 0) { // jb == k + 2 fixes qb: tiles (qa, h), wave = 2 qa + h
                        if ((k + 2) % CY != tc) break;
                        // one tile per wave; a part keeps the row blocks it owns in the second pass (qa % TS == part): its
                        // earlier plain stores to the tile sit in ITS XCD's L2
                        qb = (k + 2) / CY, qa = (wave >> 1) * TS + part, h = wave & 1;
                        if (qa >= nq) break;
                    } else {
                        qa = it;
                        if (qa % TS != part) continue;
                        const int hb = (wave + 16 - ((3 * qa) & 15)) & 15;
                        if (hb >= ncol) continue;
                        qb = hb >> 1, h = hb & 1;
                    }
                    const int ib = CY * qa + tr, jb = CY * qb + tc;
                    if (jb >= ib || jb < k + 2 || (pass == 1 && jb == k + 2)) continue;
                    d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}}, old[2];
                    const size_t to = (size_t)(32 * ib) * M + 32 * jb + 16 * h;
                    const double *tsrc = (k == 0 ? G : T) + to;
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r) old[ti][r] = tsrc[(unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_];
                    mac_2x1(PXa + (size_t)(32 * qa) * FP, PXb + (size_t)(32 * qb + 16 * h) * FP, ln, acc);
                    if (pass == 0) {
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                st_sc1(T + to + (unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_, old[ti][r] - acc[ti][r]);
                    } else {
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                T[to + (unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_] = old[ti][r] - acc[ti][r];
                    }
                }
                for (int it = 0; it < (pass == 0 ? 1 : nq); ++it) {
                    int qa, qb, h;
                    if (pass == 0) { // ib == k + 2 fixes qa: tiles (qb, h), wave = 2 qb + h
                        if ((k + 2) % CY != tr) break;
                        qa = (k + 2) / CY, qb = (wave >> 1) * TS + part, h = wave & 1;
                        if (qb >= nq) break;
                    } else {
                        qb = it;
                        if (qb % TS != part) continue;
                        const int ha = (wave + 16 - ((3 * qb) & 15)) & 15;
                        if (ha >= ncol) continue;
                        qa = ha >> 1, h = ha & 1;
                    }
                    const int ib = CY * qa + tr, cb = CY * qb + tc;
                    if (ib < k + 2 || cb > k || (pass == 1 && ib == k + 2)) continue;
                    d4 acc[2] = {(d4){0.0, 0.0, 0.0, 0.0}, (d4){0.0, 0.0, 0.0, 0.0}}, old[2];
                    double *ap = A + (size_t)(32 * cb) * M + 32 * ib + 16 * h;
                    const bool fresh = cb == k; // column block k: nothing eliminated into it yet
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            old[ti][r] = fresh ? 0.0 : ap[(unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_];
                    mac_2x1(PXb + (size_t)(32 * qb) * FP, PXa + (size_t)(32 * qa + 16 * h) * FP, ln, acc);
                    if (pass == 0) {
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                st_sc1(ap + (unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_, old[ti][r] - acc[ti][r]);
                    } else {
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                ap[(unsigned)(16 * ti + 4 * r) * (unsigned)M + lo_] = old[ti][r] - acc[ti][r];
                    }
                }
                if (pass == 0) {
                    AGPL_DRAIN(); // (every wave: its write-through stores have completed)
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(&fl->crit[tw], (unsigned)(k + 1), RLX_AGENT);
                    AGPL_TS(3);
                }
            }
            AGPL_DRAIN();
            __syncthreads();
            AGPL_TS(4);
        }
    }

    // ======================================================================================= all: U complete -> v = U (g + eta0)
    AGPL_DRAIN();
    __syncthreads();
#ifdef AGPL_FTRACE
    const int k = 31; // (the tail's stamps go to the free slots 5, 6, 7 of the last step's row)
#endif
    AGPL_TS(5);
    if (tid == 0) {
        if (lostf) __hip_atomic_store(&fl->lost, (unsigned)(100 * wg + lostf), RLX_AGENT); // (which wait gave up: diagnostic)
        __hip_atomic_fetch_add(&fl->alldone, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (!poll_ge(&fl->alldone, (unsigned)NWG)) lostf = 1;
        if (__hip_atomic_load(&fl->lost, RLX_AGENT)) lostf = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        AGPL_DRAIN();
    }
    __syncthreads();
    AGPL_TS(6);
    if (lostf) {
        if (wg == 0 && tid == 0) info[l] = -1; // a partner workgroup never arrived: the clean-up launch redoes M <= 512 alone
        return;
    }
    {
        // v[a] = sum_{b <= a} A[b * M + a] r[b]: workgroup w takes the 64-wide blocks of a with index = w (mod NWG); wave w' the
        // b = w', w' + 16, ...; the 16 partial sums of an a are combined in wave order (fixed: reproducible, and the same
        // for every NWG)
        double *rs = sm;           // [M]
        double *part = sm + M;     // [16][64]
        const double *g = gall + (size_t)l * M;
        for (int b = tid; b < M; b += 1024) rs[b] = g[b] + (eta0all ? eta0all[(size_t)l * M + b] : 0.0);
        __syncthreads();
        for (int ab = wg; ab < M / 64; ab += NWG) {
            const int a = ab * 64 + lane;
            double acc = 0.0;
            // sixteen loads in flight per lane, the sums in the order of the plain loop (the last column block is 64 dependent L2
            // round trips per wave otherwise: ~25 us behind the last block step at M = 1024 with four in flight)
            const int bend = ab * 64 + 63;
            int b = wave;
            for (; b + 16 * 31 <= bend; b += 16 * 32) { // (the long column blocks: thirty-two in flight)
                double u[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) u[j] = A[(size_t)(b + 16 * j) * M + a];
#pragma unroll
                for (int j = 0; j < 32; ++j)
                    if (b + 16 * j <= a) acc += u[j] * rs[b + 16 * j];
            }
            for (; b + 16 * 15 <= bend; b += 16 * 16) {
                double u[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) u[j] = A[(size_t)(b + 16 * j) * M + a]; // (rows b > a hold the other triangle: unused)
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (b + 16 * j <= a) acc += u[j] * rs[b + 16 * j];
            }
#pragma unroll 4
            for (; b <= bend; b += 16) {
                const double u = A[(size_t)b * M + a];
                if (b <= a) acc += u * rs[b];
            }
            part[wave * 64 + lane] = acc;
            __syncthreads();
            if (tid < 64) {
                double s = 0.0;
#pragma unroll
                for (int w = 0; w < 16; ++w) s += part[w * 64 + tid];
                if (vall) vall[(size_t)l * M + ab * 64 + tid] = s;
                if (v32all) v32all[(size_t)l * M + ab * 64 + tid] = (float)s;
            }
            __syncthreads();
        }
    }
    AGPL_TS(7);
    if (wg == 0 && tid == 0) {
        if (logdet) logdet[l] = ldsum;
        info[l] = bad;
    }
}

} // namespace

// bytes of `coop_work` for the cooperative forms of agpl_factor_fused
size_t agpl_factor_coop_bytes(int32_t M, int32_t L) {
    const size_t nb = (size_t)M / FB;
    const size_t pipe = sizeof(double) * (size_t)L * (2 * nb * FB * FB + (size_t)M * M + M); // W | P0 | P|X' per step | pivots
    const size_t la = sizeof(double) * (size_t)L * 2 * M * FB;                               // factor_kernel's two P|X' buffers
    return (pipe > la ? pipe : la) + 1024;
}

// internal: fused factorisation (M % 32 == 0; M <= 512, or M <= 1024 with M % 128 == 0); T_work / A_work [L][M][M] float64,
// info [L] int (device), coop_work: agpl_factor_coop_bytes(M, L) for the multi-workgroup forms (may be null: single workgroup,
// M <= 512 only)
int32_t agpl_factor_fused(agpl_ctx *ctx, int32_t M, int32_t L, const double *G, const double *g, const double *eta0,
                          double *T_work, double *A_work, double *v_out, float *v32_out, double *logdet_out,
                          int *info_dev, void *coop_work) {
    if (M > 512 && L > 8) {
        // Beyond M = 512 a latent's pipeline is 21-37 workgroups spread over the chip: eight latents at a time (round 6; rounds 4-5
        // sent L > 8 to two block rows around four library GEMMs).  Every array is indexed by the latent inside the launch, the flag
        // words are zero again behind each clean-up launch, and the launches are in stream order: the batches share coop_work.
        for (int l0 = 0; l0 < L; l0 += 8) {
            const int lb = L - l0 < 8 ? L - l0 : 8;
            const size_t mm = (size_t)l0 * M * M, mv = (size_t)l0 * M;
            const int32_t rc = agpl_factor_fused(ctx, M, lb, G + mm, g + mv, eta0 ? eta0 + mv : nullptr, T_work + mm, A_work + mm,
                                                 v_out ? v_out + mv : nullptr, v32_out ? v32_out + mv : nullptr,
                                                 logdet_out ? logdet_out + l0 : nullptr, info_dev + l0, coop_work);
            if (rc) return rc;
        }
        return AGPL_OK;
    }
    const size_t lds = sizeof(double) * ((size_t)(M < 512 ? M : 512) * FP + 3 * FB * FP);
    // hand-off flags: fixed words of the small workspace (8 per latent) that are zero between launches (agpl_ws2_reserve; the
    // clean-up launch behind every cooperative launch zeroes them again)
    unsigned *sync = coop_work ? (unsigned *)((char *)ctx->ws2 + 8448) : nullptr;
    const int per_xcd = (L + 7) / 8; // latent l runs on XCD l % 8 (32 CUs each)
    const int coop_mode = ctx->debug_force_rescue ? 2 : 0;
#define AGPL_LAUNCH_FACTOR(NW_, LA_, GRID_, RESCUE_)                                                                         \
    do {                                                                                                             \
        AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&factor_kernel<NW_, LA_>),                 \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                    \
        factor_kernel<NW_, LA_><<<GRID_, 1024, lds, ctx->stream>>>(M, G, g, eta0, T_work, A_work, v_out, v32_out,    \
                                                                   logdet_out, info_dev, (double *)coop_work, sync, RESCUE_); \
    } while (0)
    // ---- the pipeline form (round 5): roles F | P x NP | T x 16.  All of a latent's workgroups must be resident at once (one
    //      per CU: the 150 KB of LDS see to that) within the 32 CUs of its XCD
    if (coop_work && M % 128 == 0 && M <= 1024) {
        const int NP = (M + 255) / 256; // P workgroups of 256 rows
        // T workgroups: 4 x 4 block-cyclic cells; 2 x 2 where several latents share an XCD and a class still fits the LDS (M <= 512);
        // beyond 512 two workgroups per cell (the trailing update of the first steps is float64-MFMA-bound on 16 CUs), and the
        // workgroups go wherever the dispatcher puts them (37 per latent do not fit the 32 CUs of one XCD)
        int CY = 4, TS = 1, spread = 0;
        if (M > 512) {
            spread = 1;
            TS = L * (1 + NP + 32) <= 200 ? 2 : 1;
        } else if (per_xcd * (1 + NP + 16) > 28)
            CY = 2;
        const int nwg = 1 + NP + CY * CY * TS;
        if (L <= 32 && (spread ? L * nwg <= 200 : per_xcd * nwg <= 28)) {
            const size_t ldsp = sizeof(double) * ((size_t)(512 + 2 * FB) * FP); // P: two images of 256 rows + W + P0; T: two classes of <= 256 rows
            if (!ctx->pipe_attr) {
                AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&factor_pipe_kernel<0>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
                ctx->pipe_attr = 1;
            }
            factor_pipe_kernel<0><<<dim3(spread ? nwg : 8 * nwg, (unsigned)L), 1024, ldsp, ctx->stream>>>(
                M, NP, CY, TS, spread, G, g, eta0, T_work, A_work, v_out, v32_out, logdet_out, info_dev, (double *)coop_work,
                (PipeFlags *)sync, coop_mode);
            AGPL_LAUNCH_CHECK(ctx);
            // the clean-up launch: zeroes the flag words; redoes a latent whose partners never arrived in ONE workgroup (M <= 512)
            AGPL_LAUNCH_FACTOR(1, false, dim3((unsigned)L), 1);
            AGPL_LAUNCH_CHECK(ctx);
            return AGPL_OK;
        }
    }
    if (M > 512) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "agpl_factor_fused: M = %d, L = %d has no one-launch form", M, L);
    // workgroups per latent: 5 = look-ahead, one spine workgroup + 4 tile workgroups (2, 3 = narrower look-ahead forms used when
    // several latents share an XCD): the round 1-4 form, kept for feature counts that are not a multiple of 128
    constexpr int nw_env = 5;
    int nw = 1;
    if (coop_work) {
        if (per_xcd * nw_env <= 24) nw = nw_env;
        else if (nw_env == 5 && per_xcd * 3 <= 24) nw = 3;
        else if ((nw_env == 5 || nw_env == 3) && per_xcd * 2 <= 24) nw = 2;
    }
    if (nw == 2) AGPL_LAUNCH_FACTOR(2, true, dim3(16, (unsigned)L), coop_mode);
    else if (nw == 3) AGPL_LAUNCH_FACTOR(3, true, dim3(24, (unsigned)L), coop_mode);
    else if (nw == 5) AGPL_LAUNCH_FACTOR(5, true, dim3(40, (unsigned)L), coop_mode);
    else AGPL_LAUNCH_FACTOR(1, false, dim3((unsigned)L), 0);
    AGPL_LAUNCH_CHECK(ctx);
    // The multi-workgroup forms are plain launches that assume their partners co-resident (true when this process has
    // the device to itself).  Should other work hold those CUs for longer than the bounded spin, the spine reports info = -1 and
    // the rescue launch behind it redoes that latent in one workgroup, in stream order and without the host; otherwise it exits
    // at once (~2 us per update).
    if (nw > 1) AGPL_LAUNCH_FACTOR(1, false, dim3((unsigned)L), 1);
#undef AGPL_LAUNCH_FACTOR
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

// Test hook: make every cooperative factor launch of this context report "a partner never arrived" (info = -1) at once, so that
// the rescue launch behind it -- the path taken when other work holds the CUs the cooperating workgroups need -- does the
// factorisation.  The result must be the one the cooperative launch produces (tests/test_gpu_parity.py).
extern "C" int32_t agpl_debug_force_factor_rescue(agpl_ctx *ctx, int32_t on) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    ctx->debug_force_rescue = on != 0;
    return AGPL_OK;
}
