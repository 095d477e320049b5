// agpl_ops.hip -- the per-datapoint half of a sweep: aux_sample!, aux_posterior!,
// (expected_)auglik_{potential,precision}, and the ELBO N-reductions.  HBM-bound streaming kernels
// (the PG sampler is transcendental-heavy: see DESIGN.md for its measured fraction of the HBM roof).
#include <math.h>
#include <cstdlib>

#include "agpl_common.h"
#include "agpl_random.h"

using namespace agpl;

namespace {

constexpr int kBlock = 256;
constexpr int kRedParts = 1016; // reduction partials: doubles 8..1023 of the small scratch (bytes 64..8191)

inline int grid_for(int64_t n) {
    int64_t b = agpl_cdiv(n, kBlock);
    if (b > 256 * 16) b = 256 * 16; // 256 CUs x 16 resident blocks, grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------------------------------------
// aux_sample!  src/generic.jl:5-12.  A workgroup of four waves owns 256 consecutive points; lane = point for everything
// drawn on the point's main Philox stream (seed, i, sweep).  The PG(1, c) draws -- one per Bernoulli point, b_i = y_i + r
// for a negative-binomial point, n_ik + y_ik per latent for the categorical / Poisson families (polyagamma.jl:129-134
// draw_sum) -- live on sub-streams (agpl_random.h) and are independent work items.  They are processed in three phases that
// sort them by the branch the Devroye sampler (polyagamma.jl:225-257) takes, instead of every wave running both proposal
// branches and both forms of a(n, x) one after the other with part of its lanes idle:
//   A  each wave deals its points' draws over its lanes (exclusive prefix sum of floor(b) numbers them; lane l takes draws
//      l, l + 64, ...; the owner is found by bisection), draws the first uniform and, if the proposal is the truncated
//      exponential (x > t: a(n, x) in its exponential form), finishes the draw; otherwise the draw goes to the WORKGROUP's queue;
//   B  the queue -- truncated inverse-Gaussian proposals (x <= t: a(n, x) in its logarithmic form) -- is served by as many
//      whole waves as it fills: 108 of 256 Bernoulli draws on average = two waves, the other two wait at the barrier;
//   C  the (rare: 8e-4) draws whose series rejected the proposal are redone by the sequential sampler sample_pg1.
// A draw's value, uniforms consumed and series index do not depend on who runs it (its sub-stream is its own); the owner adds
// its draws up left to right -- the order of the sequential draw_sum loop: bit-identical sums.  Round 2 dealt the draws per
// wave but ran sample_pg1 whole in every lane: 7.9e9 Bernoulli PG(1) draws/s (profiles/r02_*), ~20 000 cycles per 64 draws.
// sample_point_wave is called by ALL threads of the workgroup under uniform control flow (valid = the lane has a point);
// f / om / nn point at the lane's own latent values (global memory for agpl_aux_sample, LDS scratch for the Gibbs pass).
// ------------------------------------------------------------------------------------------------
#ifdef AGPL_PG_TRACE // diagnostic build (make PGTRACE=1): per-phase cycle sums over all waves, tools/ab_sampler.py --trace
__device__ unsigned long long g_pgtrace[8];
#define PGT_DECL() long long pgt_ = clock64()
#define PGT_MARK(k_)                                                                                              \
    do {                                                                                                          \
        const long long now_ = clock64();                                                                         \
        if ((threadIdx.x & 63) == 0) atomicAdd(&scr->trace[k_], (unsigned long long)(now_ - pgt_));               \
        pgt_ = now_;                                                                                              \
    } while (0)
#else
#define PGT_DECL()
#define PGT_MARK(k_)
#endif
constexpr int kPgWaves = kBlock / 64;
// draws per chunk of the engine (one wave's queues and draw slots).  512 was measured again in round 6, with nothing spilling any
// more (profiles/NOTES_r06.md): build with -DAGPL_PG_CHUNK_LOG=9.
#ifndef AGPL_PG_CHUNK_LOG
#define AGPL_PG_CHUNK_LOG 8
#endif
constexpr int kPgChunkLog = AGPL_PG_CHUNK_LOG;
constexpr int kPgChunk = 1 << kPgChunkLog;
static_assert(kPgChunk >= 256 && kPgWaves * kPgChunk <= 65536, "queue entries are 16-bit (wave, slot) words; a chunk holds every owner's parameters");
constexpr int kPgMaxLat = 4;                 // latents dealt together by one engine call (owner o = 64 j + lane fits the owner byte)
constexpr int kPgMaxOwners = 64 * kPgMaxLat;
struct PgBlockScratch {
    double draws[kPgWaves][kPgChunk];        // the dealt draws of each wave's current chunk
    double par[kPgWaves][256];          // owners' parameters: one latent per call -- (z, K, bracket of r) of lane l at [4 l ..];
                                        // NB > 1 latents per call -- z of owner o = 64 j + lane at [o] (the rest per draw)
    int off[kPgWaves][kPgMaxOwners + 1]; // exclusive prefix sums of floor(b) over the owners; off[.][owners] = total
    unsigned nuni[kPgWaves][64], nterms[kPgWaves][64];
    unsigned long long index0[kPgWaves]; // point index (40 bits) of lane 0 of each wave
    unsigned short queue[kPgWaves * kPgChunk]; // phase-B queue: (wave << 8) | slot
    unsigned short queue2[kPgWaves * kPgChunk]; // phase-B queue of the mu <= t branch
    unsigned short retry[kPgWaves * kPgChunk]; // phase-C queue
    unsigned st[kPgWaves * kPgChunk];         // stream position of a parked proposal: (refills << 3) | words used
    unsigned char owner[kPgWaves][kPgChunk];  // owner lane of each dealt draw
    double etheta[64];                   // categorical kinds: exp(log theta_k), filled once per kernel (pg_scratch_init)
    int qn, q2n, qhead, rn, tmax;
    int stats;                           // nonzero: the caller reads the uniform / series-term counts (set once per kernel)
    int wqn[kPgWaves], wq2n[kPgWaves], wqhead[kPgWaves], wrn[kPgWaves]; // the same counters per wave (kPgWaveLocal)
#ifdef AGPL_PG_TRACE
    unsigned long long trace[8];
#endif
};

// once per kernel, by the whole workgroup
__device__ __forceinline__ void pg_scratch_init(PgBlockScratch *scr, const agpl_lik_dev &lik, bool stats) {
    if (threadIdx.x == 0) scr->stats = stats ? 1 : 0;
    if ((lik.kind == AGPL_LIK_CATEGORICAL || lik.kind == AGPL_LIK_CATEGORICAL_BIJ) && (int)threadIdx.x < lik.nlatent)
        scr->etheta[threadIdx.x] = exp(lik.logtheta[threadIdx.x]);
    __syncthreads();
}

// a(n, x) polyagamma.jl:167-177 with the branch known: x > t ...
__device__ __forceinline__ double pg_a_hi(int n, double x) {
    const double k = (n + 0.5) * kPi;
    return k * exp(-k * k * x / 2.0);
}
// ... and 0 < x <= t, with the term shared by all n: lx = -3/2 (log(pi / 2) + log(x))
__device__ __forceinline__ double pg_a_lo(int n, double x, double lx) {
    const double k = (n + 0.5) * kPi;
    const double expnt = lx - 2.0 * (n + 0.5) * (n + 0.5) / x;
    return k * exp(expnt);
}
// the alternating-series test of polyagamma.jl:243-256 for a proposal x; false = rejected.  HI: x > t.
template <bool HI>
__device__ __forceinline__ bool pg_series_accept(Philox &g, double x, uint32_t &nterms) {
    const double u = g.u01();
    // The first test of the series (n = 1) is  u a_0 <= a_0 - a_1,  i.e.  u <= 1 - a_1 / a_0  with  a_1 / a_0 = 3 exp(-pi^2 x)
    // (x > t) or 3 exp(-4 / x) (x <= t) -- at most 0.006: 99.5 % of the proposals are accepted here, and neither a_0 nor a_1 (one
    // more exp; a log as well for x <= t) is needed to know it.  Decided through a bracket like the branch test: a u within 1e-9
    // of the boundary (or beyond it) runs the series as the reference writes it, so the outcome and the series index are its.
    // Round 4: the bracket is first formed with ONE single-instruction float32 exponential (v_exp_f32).  rho <= 0.0058, the float32
    // argument is off by <= 2e-7 |arg| and the exponential by <= 4e-7 relative, so |rho32 - rho| < 1e-8 everywhere (|arg| e^-|arg|
    // is largest at the branch point |arg| = 6.25): a u more than 1e-7 below 1 - rho32 is accepted without a float64
    // exponential or division; 1e-7 of the proposals go on to the float64 bracket, 0.6 % to the series.  Outcome, uniforms and
    // series index are those of the plain evaluation.
    const float rho32 = 3.0f * __expf(HI ? (float)(-(kPi * kPi)) * (float)x : -4.0f / (float)x);
    if (u < 1.0 - (double)rho32 - 1e-7) {
        nterms += 1u;
        return true;
    }
    const double rho = 3.0 * exp(HI ? -(kPi * kPi) * x : -4.0 / x);
    if (u < 1.0 - rho - 1e-9) {
        nterms += 1u;
        return true;
    }
    const double lx = HI ? 0.0 : -3.0 / 2.0 * (log(kPi / 2.0) + log(x));
    double s = HI ? pg_a_hi(0, x) : pg_a_lo(0, x, lx);
    const double y = u * s;
    int n = 0;
    bool accepted = false;
    for (;;) {
        n += 1;
        const double a = HI ? pg_a_hi(n, x) : pg_a_lo(n, x, lx);
        if (n & 1) {
            s -= a;
            if (y <= s) { accepted = true; break; }
        } else {
            s += a;
            if (y > s) break;
        }
    }
    nterms += (uint32_t)n;
    return accepted;
}

// The test `alpha < u` of rand_truncated_inverse_gaussian (polyagamma.jl:205-212), alpha = exp(-w), w = z^2 x / 2 >= 0, through the
// bounds 1 - w <= exp(-w) <= 1 - w + w^2 / 2: the exponential is evaluated only when u falls between them (w^2 / 2 wide, widened by
// 1e-12 against the rounding of the bounds) -- the outcome is the exact comparison's.
__device__ __forceinline__ bool pg_alpha_below(double w, double u) {
    const double lo = 1.0 - w;
    if (u < lo - 1e-12) return false;                    // u < 1 - w <= alpha
    if (u > lo + 0.5 * w * w + 1e-12) return true;       // u > 1 - w + w^2 / 2 >= alpha
    return exp(-w) < u;
}

// One trial of the rejection loop at polyagamma.jl:200-204: E = -log(u1), E' = -log(u2) from the stream, accepted when
// E^2 <= 2 E' / t.  The two float64 logarithms (117 instructions each) are the bulk of a trial, 38 % of the trials fail, and a
// trial that passes never uses E' again -- so the test is first made on single-instruction float32 logarithms with an error
// allowance of 1e-5 (1 + |log|) each (v_log_f32 is good to ~1e-7; the conversion of u to float32 adds 6e-8 absolute): sure to
// fail -> no float64 log at all; sure to pass -> only E; in between (1e-4 of the trials) both, and the reference's comparison.
// The uniforms consumed, the outcome and E are those of the plain evaluation.
__device__ __forceinline__ bool pg_trial_passes(Philox &s, double &E) {
    const double u1 = s.u01(), u2 = s.u01();
    const float e1 = -__logf((float)u1), e2 = -__logf((float)u2);
    const float d1 = 1e-5f * (1.0f + e1), d2 = 1e-5f * (1.0f + e2);
    const float lo1 = fmaxf(e1 - d1, 0.0f), hi1 = e1 + d1, lo2 = fmaxf(e2 - d2, 0.0f), hi2 = e2 + d2;
    constexpr float c = (float)(2.0 / kPgT);
    if (lo1 * lo1 > c * hi2 * 1.000001f) return false;
    E = -log(u1);
    if (hi1 * hi1 <= c * lo2 * 0.999999f) return true;
    const double Ep = -log(u2);
    return !(E * E > (2.0 * Ep / kPgT));
}

// sub-stream of PG(1, c) draw j of point `index`, sub_base = 1 + (latent << 16) (agpl_random.h: pg_draw_id / pg_draw_block0)
__device__ __forceinline__ Philox pg_substream(const Philox &g, uint64_t index, uint32_t sub_base, uint32_t j = 0u) {
    Philox s = g;
    s.c0 = pg_draw_block0(j);
    s.c2 = (uint32_t)index;
    s.c3 = (uint32_t)(index >> 32) + ((sub_base + pg_draw_id(j)) << 8);
    s.pos = 4;
    s.nuni = 0;
    return s;
}

// sum of tb PG(1, c) draws on the sub-streams sub_base + 0 .. tb - 1 of the calling lane's point (tb = 0: the lane only helps).
// sub_base is uniform over the workgroup.  __forceinline__ (and its callers): left to its heuristics the inliner turned this into a
// real call in some builds -- the negative-binomial kernel then ran 18.0 instead of 8.6 ms per 4e6 points.
// kPgWaveLocal (round 5): every wave runs the phases on its OWN draws, queues and counters -- no workgroup barrier anywhere in the
// engine (a wave's LDS operations execute in order: a fence for the compiler is all a phase boundary needs).  The workgroup-wide
// queue balanced the trial loop between the four waves, but a lane gets the same ~1.7 entries per chunk either way, and the six
// barriers per 256-draw chunk were 29 % of a negative-binomial wave's cycles (PGTRACE, profiles/r05_ab_sampler.txt).
#ifndef AGPL_PG_WAVE_LOCAL
#define AGPL_PG_WAVE_LOCAL 1
#endif
constexpr bool kPgWaveLocal = AGPL_PG_WAVE_LOCAL != 0;
#define PG_SYNC()                                                                                                     \
    do {                                                                                                              \
        if (kPgWaveLocal) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                      \
        else __syncthreads();                                                                                         \
    } while (0)
#define PG_CNT(name_) (kPgWaveLocal ? scr->w##name_[wave] : scr->name_)
#define PG_Q(name_) (kPgWaveLocal ? scr->name_ + wave * kPgChunk : scr->name_)
#define PG_QSTRIDE (kPgWaveLocal ? 64 : kBlock)
#define PG_QFIRST (kPgWaveLocal ? lane : (int)threadIdx.x)
#define PG_LEADER (kPgWaveLocal ? lane == 0 : threadIdx.x == 0)
template <int NB>
__device__ __forceinline__ void pg_int_sum_block(PgBlockScratch *scr, int wave, int lane, const Philox &g, int latent0,
                                                 const int (&tb)[NB], const double (&c)[NB], double (&acc)[NB], uint32_t &nuni,
                                                 uint32_t &nterms) {
    static_assert(NB >= 1 && NB <= kPgMaxLat, "latents per engine call");
    constexpr int kOwners = 64 * NB;
    PGT_DECL();
#pragma unroll
    for (int j = 0; j < NB; ++j)
        if (tb[j] > 0) {
            if (NB == 1) {
                double z, K, rlo, rhi;
                pg_mass_bracket(c[0], z, K, rlo, rhi);
                scr->par[wave][4 * lane + 0] = z;
                scr->par[wave][4 * lane + 1] = K;
                scr->par[wave][4 * lane + 2] = rlo;
                scr->par[wave][4 * lane + 3] = rhi;
            } else
                scr->par[wave][64 * j + lane] = fabs(c[j]) / 2.0;
        }
    scr->nuni[wave][lane] = 0u;
    scr->nterms[wave][lane] = 0u;
    // the per-owner counts are same-address LDS atomics (about floor(b) lanes per owner): taken only when a caller reads them --
    // they were most of the engine's LDS bank conflicts (profiles/NOTES_r06.md); the draws do not depend on them.  (The flag is
    // read from LDS at each site: held in a register across the phases it cost the Gibbs point pass a spill.)
    int off[NB], T = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) { // owners in the order (latent, lane)
        int incl = tb[j];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        off[j] = T + incl - tb[j];
        T += __shfl(incl, 63);
        scr->off[wave][64 * j + lane] = off[j];
    }
    if (lane == 63) scr->off[wave][kOwners] = T;
    if (lane == 0) scr->index0[wave] = (uint64_t)g.c2 | ((uint64_t)(g.c3 & 0xFFu) << 32); // lanes = consecutive points
    int tmax;
    if (kPgWaveLocal) {
        if (lane == 0) PG_CNT(qn) = PG_CNT(q2n) = PG_CNT(qhead) = PG_CNT(rn) = 0;
        PGT_MARK(0);
        PG_SYNC();
        tmax = T; // (wave-uniform: every wave walks its own chunks)
    } else {
        if (threadIdx.x == 0) scr->tmax = scr->qn = scr->q2n = scr->qhead = scr->rn = 0;
        PGT_MARK(0);
        __syncthreads();
        if (lane == 0 && T > 0) atomicMax(&scr->tmax, T);
        __syncthreads();
        tmax = scr->tmax;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = 0.0;
    PGT_MARK(1);
    // the parameters of owner o_ of wave w_ (z always; K, the bracket of r: stored for NB == 1, formed per draw otherwise)
#define PG_OWNER_Z(w_, o_) (NB == 1 ? scr->par[w_][4 * (o_)] : scr->par[w_][o_])
#define PG_ST(e_) scr->st[e_] /* (e = (wave << 8) | slot: every wave's own 256 words in either form) */
#define PG_OWNER_SUB(o_) (1u + ((uint32_t)(latent0 + ((o_) >> 6)) << 16))
    for (int cb = 0; cb < tmax; cb += kPgChunk) {
        asm volatile("; agpl-pg-phases-begin"); // (markers in the generated code: tests/test_isa_guards.py looks between them)
        // ---- phase A
        for (int r = 0; r < kPgChunk / 64; ++r) {
            const int t = cb + 64 * r + lane;
            bool to_b = false, to_b2 = false, to_c = false;
            if (t < T) {
                int lo = 0, hi = kOwners - 1; // owner = last one whose first draw is <= t (owners without draws share an offset
                while (lo < hi) {             // with their successor and are skipped by "last")
                    const int mid = (lo + hi + 1) >> 1;
                    if (scr->off[wave][mid] <= t) lo = mid;
                    else hi = mid - 1;
                }
                scr->owner[wave][t - cb] = (unsigned char)lo;
                Philox s = pg_substream(g, scr->index0[wave] + (uint64_t)(lo & 63), PG_OWNER_SUB(lo), (uint32_t)(t - scr->off[wave][lo]));
                const double u = s.u01();
                const double z = PG_OWNER_Z(wave, lo);
                double K, rlo, rhi;
                if (NB == 1) {
                    K = scr->par[wave][4 * lo + 1], rlo = scr->par[wave][4 * lo + 2], rhi = scr->par[wave][4 * lo + 3];
                } else { // (no fit for z >= 8: an empty bracket around u sends the draw to the sequential sampler)
                    K = kPi2_8 + z * z / 2.0;
                    const double rf = pg_mass_fit<true>(z < 8.0 ? z : 0.0);
                    rlo = z < 8.0 ? rf - kPgMassSlack : u, rhi = z < 8.0 ? rf + kPgMassSlack : u;
                }
                if (!(u < rlo) && !(u > rhi)) // inside the bracket of r: decided exactly
                    to_c = true;
                else if (u < rlo) { // truncated exponential proposal, polyagamma.jl:239-240
                    const double x = kPgT + s.exp1() / K;
                    uint32_t nt = 0;
                    if (pg_series_accept<true>(s, x, nt)) {
                        scr->draws[wave][t - cb] = x / 4.0;
                        if (scr->stats) {
                            atomicAdd(&scr->nuni[wave][lo & 63], s.nuni);
                            atomicAdd(&scr->nterms[wave][lo & 63], nt);
                        }
                    } else
                        to_c = true;
                } else if (1.0 / z > kPgT) // (the test of rand_truncated_inverse_gaussian, polyagamma.jl:197)
                    to_b = true;
                else
                    to_b2 = true;
            }
            const unsigned long long mb = __ballot(to_b), mc = __ballot(to_c), mb2 = __ballot(to_b2);
            if (mb2) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&PG_CNT(q2n), __popcll(mb2));
                base = __shfl(base, 0);
                if (to_b2) PG_Q(queue2)[base + __popcll(mb2 & ((1ull << lane) - 1ull))] = (unsigned short)((wave << kPgChunkLog) | (t - cb));
            }
            if (mb) { // (wave-uniform)
                int base = 0;
                if (lane == 0) base = atomicAdd(&PG_CNT(qn), __popcll(mb));
                base = __shfl(base, 0);
                if (to_b) PG_Q(queue)[base + __popcll(mb & ((1ull << lane) - 1ull))] = (unsigned short)((wave << kPgChunkLog) | (t - cb));
            }
            if (mc) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&PG_CNT(rn), __popcll(mc));
                base = __shfl(base, 0);
                if (to_c) PG_Q(retry)[base + __popcll(mc & ((1ull << lane) - 1ull))] = (unsigned short)((wave << kPgChunkLog) | (t - cb));
            }
        }
        PGT_MARK(2);
        PG_SYNC();
        PGT_MARK(3);
        // ---- phase B: truncated inverse-Gaussian proposals, polyagamma.jl:241-242 (rand_truncated_inverse_gaussian :195-221)
        // B1: the mu > t branch as a stream of TRIALS (E, E' -> test -> x, alpha -> test).  A wave running the two nested
        // rejection loops waits for its unluckiest lane (measured: 4.9 inner iterations per pass against a mean of 1.6 per lane);
        // here a lane whose proposal is accepted parks it (x in the draw's slot, the stream position beside it) and takes the next
        // entry of the queue, so every iteration of the loop is a trial for (nearly) all 64 lanes of all four waves.
        const int qn = PG_CNT(qn), q2n = PG_CNT(q2n);
        {
            int e = -1, w = 0, slot = 0;
            double z = 0.0;
            uint32_t c0first = 0u;
            Philox s = g;
            // the lanes that need a new entry take them with ONE returning LDS atomic per wave and iteration (round 6: every lane adding
            // 1 to the same word was a 64-way serialised atomic at the start of the loop and a several-way one in every iteration --
            // SQ_LDS_BANK_CONFLICT ~ 50 % of the LDS-active cycles of the sampler kernels, profiles/r05_pmc_sq2_c2.json); which lane
            // runs which entry does not matter (a draw's value is a function of its own sub-stream).  Called by the whole wave.
            auto fetch = [&](bool need) {
                const unsigned long long mneed = __ballot(need);
                if (mneed) { // (wave-uniform)
                    const int first = __ffsll((long long)mneed) - 1;
                    int idx = 0;
                    if (lane == first) idx = atomicAdd(&PG_CNT(qhead), __popcll(mneed));
                    idx = __shfl(idx, first) + __popcll(mneed & ((1ull << lane) - 1ull));
                    if (need) {
                        e = -1;
                        if (idx < qn) {
                            e = PG_Q(queue)[idx];
                            w = e >> kPgChunkLog, slot = e & (kPgChunk - 1);
                            const int lo = scr->owner[w][slot];
                            s = pg_substream(g, scr->index0[w] + (uint64_t)(lo & 63), PG_OWNER_SUB(lo), (uint32_t)(cb + slot - scr->off[w][lo]));
                            // the branch uniform (drawn in phase A) and the first `while (alpha < rand())` (alpha = 0: always entered, u is
                            // in the open interval) are block 0 of the sub-stream: passed over without its ten rounds
                            c0first = s.c0;
                            s.skip_first_block();
                            z = PG_OWNER_Z(w, lo);
                        }
                    }
                }
            };
            fetch(true);
            while (__ballot(e >= 0)) {
                bool need = false;
                if (e >= 0) {
                    double E;
                    if (pg_trial_passes(s, E)) { // E, E' ~ Exp(1) with E^2 <= 2 E' / t (polyagamma.jl:200-204)
                        const double d = 1.0 + E * kPgT;
                        const double x = kPgT / (d * d);
                        const double ua = s.u01();
                        if (!pg_alpha_below(z * z * x / 2.0, ua)) { // `alpha < u` with alpha = exp(-z^2 x / 2) is false: accepted
                            scr->draws[w][slot] = x;
                            PG_ST(e) = ((s.c0 - c0first) << 3) | (uint32_t)s.pos; // (blocks since the draw's first one)
                            need = true;
                        }
                    }
                }
                fetch(need);
            }
        }
        // the mu <= t branch (|c| >= 3.125: Michael-Schucany-Haas proposals until x <= t) is rare: whole passes of the loop as is
        if (q2n) {
            for (int q = PG_QFIRST; q < q2n; q += PG_QSTRIDE) {
                const int e = PG_Q(queue2)[q];
                const int w = e >> kPgChunkLog, slot = e & (kPgChunk - 1), lo = scr->owner[w][slot];
                Philox s = pg_substream(g, scr->index0[w] + (uint64_t)(lo & 63), PG_OWNER_SUB(lo), (uint32_t)(cb + slot - scr->off[w][lo]));
                const uint32_t c0first = s.c0;
                (void)s.u01();
                scr->draws[w][slot] = rand_tig(s, PG_OWNER_Z(w, lo));
                PG_ST(e) = ((s.c0 - c0first) << 3) | (uint32_t)s.pos;
            }
        }
        PGT_MARK(4);
        PG_SYNC();
        PGT_MARK(5);
        // B2: the series test of the parked proposals (x <= t: a(n, x) in its logarithmic form), whole waves
        for (int q = PG_QFIRST; q < ((qn + q2n + 63) & ~63); q += PG_QSTRIDE) {
            bool to_c = false;
            int e = 0;
            if (q < qn + q2n) {
                e = q < qn ? PG_Q(queue)[q] : PG_Q(queue2)[q - qn];
                const int w = e >> kPgChunkLog, slot = e & (kPgChunk - 1), lo = scr->owner[w][slot];
                Philox s = pg_substream(g, scr->index0[w] + (uint64_t)(lo & 63), PG_OWNER_SUB(lo), (uint32_t)(cb + slot - scr->off[w][lo]));
                const uint32_t st = PG_ST(e), c0 = st >> 3, pos = st & 7u; // the stream where the proposal left it (blocks since its first)
                if (pos < 4u) {
                    s.c0 += c0 - 1u;
                    s.refill();
                } else
                    s.c0 += c0;
                s.pos = (int)pos;
                s.nuni = 2u * c0 - (4u - pos) / 2u; // (every uniform takes two of the four words of a block)
                const double x = scr->draws[w][slot];
                uint32_t nt = 0;
                if (pg_series_accept<false>(s, x, nt)) {
                    scr->draws[w][slot] = x / 4.0;
                    if (scr->stats) {
                        atomicAdd(&scr->nuni[w][lo & 63], s.nuni);
                        atomicAdd(&scr->nterms[w][lo & 63], nt);
                    }
                } else
                    to_c = true;
            }
            const unsigned long long mc = __ballot(to_c);
            if (mc) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&PG_CNT(rn), __popcll(mc));
                base = __shfl(base, 0);
                if (to_c) PG_Q(retry)[base + __popcll(mc & ((1ull << lane) - 1ull))] = (unsigned short)e;
            }
        }
        PGT_MARK(4);
        PG_SYNC();
        asm volatile("; agpl-pg-phases-end");
        PGT_MARK(5);
        // ---- phase C: the draws whose first proposal was rejected, from the start of their sub-stream
        const int rn = PG_CNT(rn);
        for (int q = PG_QFIRST; q < rn; q += PG_QSTRIDE) {
            const int e = PG_Q(retry)[q];
            const int w = e >> kPgChunkLog, slot = e & (kPgChunk - 1), lo = scr->owner[w][slot];
            Philox s = pg_substream(g, scr->index0[w] + (uint64_t)(lo & 63), PG_OWNER_SUB(lo), (uint32_t)(cb + slot - scr->off[w][lo]));
            Pg1Params p;
            p.set(2.0 * PG_OWNER_Z(w, lo)); // (z = |c| / 2 exactly)
            uint32_t nt = 0;
            scr->draws[w][slot] = sample_pg1(s, p, nt);
            if (scr->stats) {
                atomicAdd(&scr->nuni[w][lo & 63], s.nuni);
                atomicAdd(&scr->nterms[w][lo & 63], nt);
            }
        }
        PG_SYNC();
        PGT_MARK(6);
        if (PG_LEADER) PG_CNT(qn) = PG_CNT(q2n) = PG_CNT(qhead) = PG_CNT(rn) = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int a0 = off[j] > cb ? off[j] : cb, a1 = (off[j] + tb[j]) < (cb + kPgChunk) ? (off[j] + tb[j]) : (cb + kPgChunk);
            for (int t = a0; t < a1; ++t) acc[j] += scr->draws[wave][t - cb];
        }
        PG_SYNC();
        PGT_MARK(1);
    }
    nuni += scr->nuni[wave][lane];
    nterms += scr->nterms[wave][lane];
#undef PG_OWNER_Z
#undef PG_OWNER_SUB
#undef PG_ST
}

// rand(PolyaGamma(b_j, c_j)) for latents latent0 .. latent0 + nk - 1 of the lane's point (nk <= NB), integer parts dealt across
// the workgroup in ONE pass of the phases: the same values, uniforms consumed and series indices as agpl::rand_pg(g, latent, b, c, .)
// run by one lane, latent after latent.  Called by all threads of the workgroup.  INT: every b is known to be an integer (no
// Gamma-series code in the kernel).  The categorical likelihood's K latents go through in groups of kPgMaxLat: per call the
// engine costs six workgroup barriers whatever it has to draw, and a latent of that likelihood brings ~13 draws per wave.
template <int NB, bool INT = false>
__device__ __forceinline__ void pg_points_wave(PgBlockScratch *scr, int lane, bool valid, Philox &g, int latent0, int nk,
                                               const double (&b)[NB], const double (&c)[NB], double (&w)[NB], uint32_t &nterms,
                                               int *bad) {
    const int wave = (int)(threadIdx.x >> 6);
    int tb[NB];
    bool ok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const bool on = valid && j < nk;
        // b >= kPgMaxB = 2^22 is outside the numbering of a point's draws (agpl_random.h): flagged (bit 1), reported by the
        // host as AGPL_ERR_UNSUPPORTED instead of a silent NaN
        if (on && b[j] >= kPgMaxB) atomicOr(bad, 2);
        ok[j] = on && (b[j] >= 0.0) && (fabs(c[j]) < __builtin_inf()) && (b[j] < kPgMaxB);
        tb[j] = ok[j] ? (int)floor(b[j]) : 0;
    }
    pg_int_sum_block<NB>(scr, wave, lane, g, latent0, tb, c, w, g.nuni, nterms);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (!(valid && j < nk)) w[j] = 0.0;
        else if (!ok[j]) w[j] = __builtin_nan("");
        else if (b[j] == 0.0) w[j] = 0.0;
        else if (!INT) {
            const double res = b[j] - (double)tb[j];
            if (res != 0.0) {
                Philox s = g.sub(1u + ((uint32_t)(latent0 + j) << 16) + kSubResidual);
                w[j] += rand_gamma_sum(s, c[j], res);
                g.nuni += s.nuni;
            }
        }
    }
}
// one latent
template <bool INT = false>
__device__ __forceinline__ double pg_point_wave(PgBlockScratch *scr, int lane, bool valid, Philox &g, int latent, double b,
                                       double c, uint32_t &nterms, int *bad) {
    const double bb[1] = {b}, cc[1] = {c};
    double w[1];
    pg_points_wave<1, INT>(scr, lane, valid, g, latent, 1, bb, cc, w, nterms, bad);
    return w[0];
}

template <typename NN>
__device__ __forceinline__ NN count_cast(int64_t n) { // (saturating: a count beyond int32 is far beyond the PG(b) limit and is flagged there)
    return sizeof(NN) == 4 && n > 0x7fffffffLL ? (NN)0x7fffffff : (NN)n;
}
// NN: the type the latent counts n are kept in (int64_t in the caller's n_out; int32_t in the Gibbs kernel's LDS scratch, which with
// 64-bit counts pinned that kernel to one workgroup per CU at K = 10)
template <int KIND, typename NN>
__device__ __forceinline__ void sample_point_wave(const agpl_lik_dev &lik, PgBlockScratch *scr, int lane, bool valid, Philox &g,
                                         int64_t i, const void *yv, const double *f, double *om, NN *nn,
                                         uint32_t &nt, int *bad) {
    const int L = lik.nlatent;
    switch (KIND) { // compile-time: each kernel instantiation carries one likelihood's sampler only
    case AGPL_LIK_BERNOULLI_LOGISTIC: { // bernoulli.jl:13-15: one draw per point (= rand_pg_int(g, 1, |f|, .))
        const double w = pg_point_wave<true>(scr, lane, valid, g, 0, 1.0, valid ? fabs(f[0]) : 0.0, nt, bad);
        if (valid) om[0] = w;
    } break;
    case AGPL_LIK_NEGBINOMIAL: { // negativebinomial.jl:20-22
        const int32_t *y = (const int32_t *)yv;
        const double b = valid ? (double)y[i] + lik.p[0] : 0.0, c = valid ? fabs(f[0]) : 0.0;
        const double w = pg_point_wave(scr, lane, valid, g, 0, b, c, nt, bad);
        if (valid) om[0] = w;
    } break;
    case AGPL_LIK_STUDENTT: { // studentt.jl:46-48
        if (valid) {
            const double *y = (const double *)yv;
            double nu = lik.p[0], sg = lik.p[1];
            double d = y[i] - f[0];
            double scale = 2.0 / (nu / (sg * sg) + d * d);
            om[0] = scale * rand_gamma(g, (nu + 1.0) / 2.0);
        }
    } break;
    case AGPL_LIK_CATEGORICAL:
    case AGPL_LIK_CATEGORICAL_BIJ: { // categorical.jl:72-78, polyagammanegativemultinomial.jl:27-31,
                                     // negativemultinomial.jl:35-45
        const uint8_t *y = (const uint8_t *)yv;
        bool good = valid;
        if (valid) {
            // p_k = theta_k logistic(f_k) / sum(theta): formed once, kept in the point's omega slots until the draws replace them
            // (the same values the reference forms twice); theta_k = exp(log theta_k) comes from the workgroup's table
            double sp = 0.0;
            for (int k = 0; k < L; ++k) {
                const double pk = scr->etheta[k] * logistic(f[k]) / lik.sum_theta;
                om[k] = pk;
                sp += pk;
            }
            double p0 = 1.0 - sp;
            if (!(sp < 1.0)) { // ArgumentError negativemultinomial.jl:17-22
                atomicOr(bad, 1);
                good = false;
            } else {
                double theta = (1.0 / p0 - 1.0) * rand_gamma(g, 1.0);
                for (int k = 0; k < L; ++k) {
                    const double pk = om[k];
                    double lam = pk * theta / (1.0 - p0);
                    nn[k] = count_cast<NN>(rand_poisson(g, lam));
                }
            }
        }
        for (int k0 = 0; k0 < L; k0 += kPgMaxLat) { // L is uniform over the workgroup: every lane deals for every group of latents
            const int nk = L - k0 < kPgMaxLat ? L - k0 : kPgMaxLat;
            double b[kPgMaxLat], c[kPgMaxLat], w[kPgMaxLat];
#pragma unroll
            for (int j = 0; j < kPgMaxLat; ++j) {
                const bool on = good && j < nk;
                b[j] = on ? (double)(nn[k0 + j] + (int64_t)y[i * L + k0 + j]) : 0.0;
                c[j] = on ? fabs(f[k0 + j]) : 0.0;
            }
            pg_points_wave<kPgMaxLat, true>(scr, lane, good, g, k0, nk, b, c, w, nt, bad); // (counts + one-hot labels: integers)
#pragma unroll
            for (int j = 0; j < kPgMaxLat; ++j)
                if (good && j < nk) om[k0 + j] = w[j];
        }
    } break;
    case AGPL_LIK_POISSON: { // poisson.jl:26-28, polyagammapoisson.jl:23-27
        const int32_t *y = (const int32_t *)yv;
        double b = 0.0, c = 0.0;
        if (valid) {
            double lam = lik.p[0] * logistic(-f[0]);
            int64_t n1 = rand_poisson(g, lam);
            nn[0] = count_cast<NN>(n1);
            b = (double)(n1 + y[i]);
            c = fabs(f[0]);
        }
        const double w = pg_point_wave(scr, lane, valid, g, 0, b, c, nt, bad);
        if (valid) om[0] = w;
    } break;
    case AGPL_LIK_LAPLACE: { // laplace.jl:40-42
        if (valid) {
            const double *y = (const double *)yv;
            double beta = lik.p[0];
            double lam = 1.0 / ((2.0 * beta) * (2.0 * beta));
            om[0] = rand_invgaussian(g, 1.0 / (2.0 * beta * fabs(y[i] - f[0])), 2.0 * lam);
        }
    } break;
    case AGPL_LIK_HETEROGAUSS: { // heteroscedasticgaussian.jl:28-32
        const double *y = (const double *)yv;
        double b = 0.0, c = 0.0;
        if (valid) {
            double ff = f[0], gg = f[1];
            double lam = lik.p[0] * logistic(-gg) * (ff - y[i]) * (ff - y[i]) / 2.0;
            int64_t n1 = rand_poisson(g, lam);
            nn[0] = count_cast<NN>(n1);
            b = 0.5 + (double)n1;
            c = fabs(gg);
        }
        const double w = pg_point_wave(scr, lane, valid, g, 0, b, c, nt, bad);
        if (valid) om[0] = w;
    } break;
    default:
        break;
    }
}

// Waves per SIMD the sampler kernels are compiled for (register allocation only: the draws are bit-identical).  The phased
// sampler has workgroup barriers between its phases, which only other workgroups on the same SIMDs can fill: four waves per
// SIMD (128 VGPRs, some spills) beat 1 / 2 / 3 on one box in round 3 -- Bernoulli 2.36 / 1.42 / 1.31 / 1.26 ms per 1e7 points,
// negative binomial r = 15 16.9 / 10.4 / 8.9 / 8.5 ms per 4e6 points (profiles/r03_ab_sampler.txt).  The kinds without PG draws
// have no barriers and keep their registers.
constexpr int sampler_wps(int kind) {
    // (round 5, after the scratch was gone: 3 / 4 / 5 waves -- Bernoulli PG(1) kernel 0.540 / 0.533 / 0.532 ms, negative binomial
    //  7.26 / 6.30 / 7.23 ms, categorical K = 10 0.525 / 0.665 / 0.646 ms: the categorical kinds take 3)
#ifdef AGPL_SAMPLER_WPS // measurement builds
    return (kind == AGPL_LIK_STUDENTT || kind == AGPL_LIK_LAPLACE) ? 1 : AGPL_SAMPLER_WPS;
#else
    return (kind == AGPL_LIK_STUDENTT || kind == AGPL_LIK_LAPLACE) ? 1 : (kind == AGPL_LIK_CATEGORICAL || kind == AGPL_LIK_CATEGORICAL_BIJ) ? 3 : 4;
#endif
}

template <int KIND>
__global__ __launch_bounds__(kBlock, sampler_wps(KIND)) void aux_sample_kernel(agpl_lik_dev lik, int64_t n, const void *yv,
                                                            const double *__restrict__ f,
                                                            double *__restrict__ omega,
                                                            int64_t *__restrict__ nout, uint64_t seed,
                                                            uint64_t i0, uint32_t sweep,
                                                            uint32_t *__restrict__ nuni_out,
                                                            uint32_t *__restrict__ nterms_out,
                                                            int *__restrict__ bad) {
    __shared__ PgBlockScratch scratch;
    pg_scratch_init(&scratch, lik, nuni_out != nullptr || nterms_out != nullptr);
#ifdef AGPL_PG_TRACE
    if (threadIdx.x < 8) scratch.trace[threadIdx.x] = 0ull;
    __syncthreads();
#endif
    const int Lf = lik.nlatent;
    const int Lo = lik.kind == AGPL_LIK_HETEROGAUSS ? 1 : lik.nlatent;
    const int lane = threadIdx.x & 63;
    const int64_t nblocks = (n + kBlock - 1) / kBlock; // (the trip count is uniform over the workgroup: the sampler has barriers)
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t i = blk * kBlock + threadIdx.x;
        const bool valid = i < n;
        Philox g;
        g.init(seed, i0 + (uint64_t)i, sweep);
        uint32_t nt = 0;
        sample_point_wave<KIND, int64_t>(lik, &scratch, lane, valid, g, i, yv, f + i * Lf, omega + i * Lo,
                                nout ? nout + i * Lo : nullptr, nt, bad);
        if (valid) {
            if (nuni_out) nuni_out[i] = g.nuni;
            if (nterms_out) nterms_out[i] = nt;
        }
    }
#ifdef AGPL_PG_TRACE
    __syncthreads();
    if (threadIdx.x < 8) atomicAdd(&g_pgtrace[threadIdx.x], scratch.trace[threadIdx.x]);
#endif
}

// ------------------------------------------------------------------------------------------------
// aux_sample! of the Bernoulli likelihood (bernoulli.jl:13-15): ONE PG(1, |f_i|) draw per point, so the draw is its own owner
// and the dealing machinery above has nothing to deal -- but the trial queue wants to be long (a lane that finds the queue
// empty idles until the unluckiest lane of its wave is done).  A workgroup therefore takes kPg1Pts points per thread through
// the same phases: A (branch uniform; truncated-exponential proposals finished), B1 (trials of the truncated inverse-Gaussian
// proposals with refill from the workgroup's queue), B2 (their series test).  The 8e-4 whose proposal the series rejects (and the
// points without a fitted branch mass, |f| >= 16) are redone from the start of their sub-stream by the sequential sampler -- in a
// launch of their own (aux_sample_pg1_retry_kernel, round 5): the point goes on a list in global memory.  With that loop inside
// this kernel its live state cost the trial loops 300 bytes of scratch per lane, and every workgroup iteration waited at two more
// barriers for the one wave that had a redo.  Same sub-streams (draw 0 of latent 0 = id 1), same values, uniforms consumed and
// series indices as sample_point_wave's case.
// ------------------------------------------------------------------------------------------------
constexpr int kPg1Pts = 8;                   // points per thread and workgroup iteration
constexpr int kPg1Slots = kPg1Pts * kBlock;  // 2048 draws per iteration
struct Pg1BlockScratch {
    double x[kPg1Slots];                     // parked proposals
    unsigned st[kPg1Slots];                  // their stream positions
    unsigned short queue[kPg1Slots], queue2[kPg1Slots]; // (kPg1WaveLocal: wave w's entries at [512 w ..])
    int qn, q2n, qhead;
    int wqn[kBlock / 64], wq2n[kBlock / 64], wqhead[kBlock / 64];
};
#ifndef AGPL_PG1_WAVE_LOCAL
#define AGPL_PG1_WAVE_LOCAL 0
#endif
// as kPgWaveLocal (every wave its own 512 points, queues and counters, no workgroup barrier) -- measured and NOT shipped for this kernel:
// 0.554-0.560 against 0.534-0.536 ms per 1e7 points (its 2048-entry workgroup queue keeps the trial loop fuller than four 512-entry ones,
// and it has four barriers per 2048 points, not six per 256 draws)
constexpr bool kPg1WaveLocal = AGPL_PG1_WAVE_LOCAL != 0;

// GIBBS: the Bernoulli point pass of a sparse Gibbs sweep in the same kernel -- f_i = projection_i + sqrt(d_i) eps_i (+ mu0_i) on the
// point's main stream (two uniforms) is formed in phase A and written over the projection (fbuf, read back by the later phases),
// and a finished draw also leaves gamma_i = omega_i, beta_i = +-1/2 (auglik_precision / auglik_potential, bernoulli.jl:27-33).
template <bool GIBBS>
__global__ __launch_bounds__(kBlock, 4) void aux_sample_pg1_kernel(const unsigned n, const double *__restrict__ f,
                                                                   double *__restrict__ omega, uint64_t seed, uint64_t i0,
                                                                   uint32_t sweep, uint32_t *__restrict__ nuni_out,
                                                                   uint32_t *__restrict__ nterms_out, double *fbuf,
                                                                   const float *__restrict__ kdiag, const float *__restrict__ mu0,
                                                                   const uint8_t *__restrict__ y, float *__restrict__ gamma,
                                                                   float *__restrict__ beta, double *__restrict__ f_out,
                                                                   unsigned *__restrict__ retry) {
    if (GIBBS) f = fbuf;
    constexpr uint32_t kMain = GIBBS ? 2u : 0u; // uniforms the point's main stream has consumed (the normal of f)
    __shared__ Pg1BlockScratch scr;
    const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6);
    int &qn_c = kPg1WaveLocal ? scr.wqn[wave] : scr.qn, &q2n_c = kPg1WaveLocal ? scr.wq2n[wave] : scr.q2n,
        &qhead_c = kPg1WaveLocal ? scr.wqhead[wave] : scr.qhead;
    unsigned short *const queue_w = kPg1WaveLocal ? scr.queue + wave * (kPg1Slots / (kBlock / 64)) : scr.queue;
    unsigned short *const queue2_w = kPg1WaveLocal ? scr.queue2 + wave * (kPg1Slots / (kBlock / 64)) : scr.queue2;
    const bool leader = kPg1WaveLocal ? lane == 0 : threadIdx.x == 0;
    const int qfirst = kPg1WaveLocal ? lane : (int)threadIdx.x, qstride = kPg1WaveLocal ? 64 : kBlock;
#define PG1_SYNC()                                                                                                    \
    do {                                                                                                              \
        if (kPg1WaveLocal) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");                                     \
        else __syncthreads();                                                                                         \
    } while (0)
    const unsigned nblocks = (n + kPg1Slots - 1u) / kPg1Slots; // (n <= 2^30 per launch: 32-bit point indices, launch_pg1)
    if (leader) qn_c = q2n_c = qhead_c = 0;
    PG1_SYNC();
    auto push = [&](bool flag, unsigned short *q, int *cnt, int slot) { // compacted append of the wave's flagged lanes
        const unsigned long long m = __ballot(flag);
        if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(cnt, __popcll(m));
            base = __shfl(base, 0);
            if (flag) q[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)slot;
        }
    };
    auto push_retry = [&](bool flag, unsigned i) { // the same onto the launch's retry list: [0] entries, [2 ..] point indices
        const unsigned long long m = __ballot(flag);
        if (m) {
            unsigned at = 0;
            if (lane == 0) at = atomicAdd(&retry[0], (unsigned)__popcll(m));
            at = __shfl(at, 0);
            if (flag) retry[2u + at + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = i;
        }
    };
    auto finish = [&](unsigned i, double w, const Philox &s, uint32_t nt) {
        if (!GIBBS || omega) omega[i] = w;
        if (nuni_out) nuni_out[i] = kMain + s.nuni;
        if (nterms_out) nterms_out[i] = nt;
        if (GIBBS) {
            gamma[i] = (float)w;
            beta[i] = y[i] ? 0.5f : -0.5f;
        }
    };
    for (unsigned blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const unsigned base = blk * kPg1Slots;
        Philox g0; // key and sweep of every stream of this launch; point and sub-stream set by pg_substream
        g0.init(seed, 0, sweep);
        // ---- phase A
        if (GIBBS) { // (the point half of gibbs_sample_points for this likelihood; a loop of its own: the normal's registers -- log,
                     //  sqrt, cos -- are free again when the branch test below starts, which left 8 bytes of scratch otherwise)
            for (int p = 0; p < kPg1Pts; ++p) {
                const unsigned i = base + (unsigned)(p * kBlock) + threadIdx.x;
                if (i < n) {
                    Philox g;
                    g.init(seed, i0 + (uint64_t)i, sweep);
                    const double kd = (double)kdiag[i];
                    double fi = fbuf[i] + sqrt(kd > 0.0 ? kd : 0.0) * g.normal();
                    if (mu0) fi += (double)mu0[i];
                    fbuf[i] = fi;
                    if (f_out) f_out[i] = fi;
                }
            }
        }
        for (int p = 0; p < kPg1Pts; ++p) {
            const int slot = p * kBlock + (int)threadIdx.x;
            const unsigned i = base + (unsigned)slot;
            bool to_b = false, to_b2 = false, to_c = false;
            if (i < n) {
                const double fi = GIBBS ? fbuf[i] : f[i];
                const double c = fabs(fi);
                if (!(c < __builtin_inf())) { // NaN / Inf in, NaN out (rand_pg_int)
                    finish(i, __builtin_nan(""), g0, 0u);
                    if (nuni_out) nuni_out[i] = kMain;
                } else {
                    const double z = c / 2.0, K = kPi2_8 + z * z / 2.0; // (= Pg1Params::set, bit for bit)
                    const double r = pg_mass_fit(z < 8.0 ? z : 0.0);
                    Philox s = pg_substream(g0, i0 + (uint64_t)i, 1u);
                    const double u = s.u01();
                    if (!(z < 8.0) || (!(u < r - kPgMassSlack) && !(u > r + kPgMassSlack))) // no fit, or u inside the bracket of r:
                        to_c = true;                                                        // the sequential sampler decides
                    else if (u < r - kPgMassSlack) { // truncated exponential proposal, polyagamma.jl:239-240
                        const double x = kPgT + s.exp1() / K;
                        uint32_t nt = 0;
                        if (pg_series_accept<true>(s, x, nt)) finish(i, x / 4.0, s, nt);
                        else to_c = true;
                    } else if (1.0 / z > kPgT)
                        to_b = true;
                    else
                        to_b2 = true;
                }
            }
            push(to_b, queue_w, &qn_c, slot);
            push(to_b2, queue2_w, &q2n_c, slot);
            push_retry(to_c, i);
        }
        PG1_SYNC();
        // ---- phase B1: trials with refill (see pg_int_sum_block)
        const int qn = qn_c, q2n = q2n_c;
        {
            int e = -1;
            double z = 0.0;
            Philox s = g0;
            auto fetch = [&](bool need) { // ONE returning LDS atomic per wave and iteration (see pg_int_sum_block); whole wave calls
                const unsigned long long mneed = __ballot(need);
                if (mneed) {
                    const int first = __ffsll((long long)mneed) - 1;
                    int idx = 0;
                    if (lane == first) idx = atomicAdd(&qhead_c, __popcll(mneed));
                    idx = __shfl(idx, first) + __popcll(mneed & ((1ull << lane) - 1ull));
                    if (need) {
                        e = -1;
                        if (idx < qn) {
                            e = queue_w[idx];
                            const unsigned i = base + (unsigned)e;
                            s = pg_substream(g0, i0 + (uint64_t)i, 1u);
                            s.skip_first_block(); // (the branch uniform and the always-entered first `alpha < rand()`: see pg_int_sum_block)
                            z = fabs(f[i]) / 2.0; // (Pg1Params::set)
                        }
                    }
                }
            };
            fetch(true);
            while (__ballot(e >= 0)) {
                bool need = false;
                if (e >= 0) {
                    const double E = s.exp1();
                    const double Ep = s.exp1();
                    if (!(E * E > (2.0 * Ep / kPgT))) { // (pg_trial_passes costs this kernel registers: 0.80 against 0.70 ms per 1e7 points)
                        const double d = 1.0 + E * kPgT;
                        const double x = kPgT / (d * d);
                        const double ua = s.u01();
                        if (!pg_alpha_below(z * z * x / 2.0, ua)) { // `alpha < u` with alpha = exp(-z^2 x / 2) is false: accepted
                            scr.x[e] = x;
                            scr.st[e] = (s.c0 << 3) | (uint32_t)s.pos;
                            need = true;
                        }
                    }
                }
                fetch(need);
            }
        }
        if (q2n) {
            for (int q = qfirst; q < q2n; q += qstride) {
                const int e = queue2_w[q];
                const unsigned i = base + (unsigned)e;
                Philox s = pg_substream(g0, i0 + (uint64_t)i, 1u);
                (void)s.u01();
                scr.x[e] = rand_tig(s, fabs(f[i]) / 2.0);
                scr.st[e] = (s.c0 << 3) | (uint32_t)s.pos;
            }
        }
        PG1_SYNC();
        // ---- phase B2: series test of the parked proposals
        for (int q = qfirst; q < ((qn + q2n + 63) & ~63); q += qstride) {
            bool to_c = false;
            int e = 0;
            if (q < qn + q2n) {
                e = q < qn ? queue_w[q] : queue2_w[q - qn];
                const unsigned i = base + (unsigned)e;
                Philox s = pg_substream(g0, i0 + (uint64_t)i, 1u);
                const uint32_t st = scr.st[e], c0 = st >> 3, pos = st & 7u;
                if (pos < 4u) {
                    s.c0 = c0 - 1u;
                    s.refill();
                } else
                    s.c0 = c0;
                s.pos = (int)pos;
                s.nuni = 2u * c0 - (4u - pos) / 2u;
                const double x = scr.x[e];
                uint32_t nt = 0;
                if (pg_series_accept<false>(s, x, nt)) finish(i, x / 4.0, s, nt);
                else to_c = true;
            }
            push_retry(to_c, base + e);
        }
        PG1_SYNC();
        if (leader) {
            int zero = 0; // (formed here: as a loop invariant the compiler kept three zeroed registers alive over the whole iteration --
            asm volatile("" : "+v"(zero)); // and spilled them, the kernel's last 12 bytes of scratch)
            qn_c = zero, q2n_c = zero, qhead_c = zero;
        }
        PG1_SYNC();
    }
#undef PG1_SYNC
}

// The points aux_sample_pg1_kernel left on its list: the sequential sampler from the start of the draw's sub-stream
// (polyagamma.jl:223-257 as sample_pg1 restates it).  The last workgroup to finish leaves the two counter words zero for the next launch.
template <bool GIBBS>
__global__ __launch_bounds__(kBlock) void aux_sample_pg1_retry_kernel(unsigned *__restrict__ retry, const double *__restrict__ f,
                                                                      double *__restrict__ omega, uint64_t seed, uint64_t i0,
                                                                      uint32_t sweep, uint32_t *__restrict__ nuni_out,
                                                                      uint32_t *__restrict__ nterms_out,
                                                                      const uint8_t *__restrict__ y, float *__restrict__ gamma,
                                                                      float *__restrict__ beta) {
    constexpr uint32_t kMain = GIBBS ? 2u : 0u;
    const unsigned cnt = retry[0];
    Philox g0;
    g0.init(seed, 0, sweep);
    for (unsigned q = blockIdx.x * kBlock + threadIdx.x; q < cnt; q += gridDim.x * kBlock) {
        const int64_t i = (int64_t)retry[2u + q];
        Pg1Params prm;
        prm.set(fabs(f[i]));
        Philox s = pg_substream(g0, i0 + (uint64_t)i, 1u);
        uint32_t nt = 0;
        const double w = sample_pg1(s, prm, nt);
        if (!GIBBS || omega) omega[i] = w;
        if (nuni_out) nuni_out[i] = kMain + s.nuni;
        if (nterms_out) nterms_out[i] = nt;
        if (GIBBS) {
            gamma[i] = (float)w;
            beta[i] = y[i] ? 0.5f : -0.5f;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&retry[1], 1u) == gridDim.x - 1u) { // (every other workgroup has read retry[0] before its add)
        retry[0] = 0u;
        retry[1] = 0u;
    }
}

// host side of the pair: kernel, then the retry kernel over its list.  Point indices on the list are 32-bit: launches of at most
// 2^30 points (pointers and the stream offset advanced per launch).
#ifndef AGPL_PG1_MAX_LAUNCH
#define AGPL_PG1_MAX_LAUNCH ((int64_t)1 << 30)
#endif
constexpr int64_t kPg1MaxLaunch = AGPL_PG1_MAX_LAUNCH; // (a test build sets 5000: tests/test_gpu_regressions_r5.py)
template <bool GIBBS>
static void launch_pg1(agpl_ctx *ctx, int64_t n, const double *f, double *omega, uint32_t sweep, uint32_t *nuni_out,
                          uint32_t *nterms_out, double *fbuf, const float *kdiag, const float *mu0, const uint8_t *y, float *gamma,
                          float *beta, double *f_out) {
    constexpr int64_t kMaxLaunch = kPg1MaxLaunch; // (the caller has reserved the list: agpl_pg_retry_reserve(ctx, min(n, this)))
    for (int64_t o = 0; o < n; o += kMaxLaunch) {
        const int64_t m = n - o < kMaxLaunch ? n - o : kMaxLaunch;
        int64_t nb = agpl_cdiv(m, kPg1Slots);
        if (nb > 256 * 4 * 8) nb = 256 * 4 * 8;
        int64_t nr = agpl_cdiv(m, 16 * kBlock);
        if (nr > 1024) nr = 1024;
#define AGPL_O(p_) ((p_) ? (p_) + o : nullptr)
        aux_sample_pg1_kernel<GIBBS><<<(unsigned)nb, kBlock, 0, ctx->stream>>>(
            (unsigned)m, AGPL_O(f), AGPL_O(omega), ctx->seed, (uint64_t)ctx->point_offset + (uint64_t)o, sweep, AGPL_O(nuni_out), AGPL_O(nterms_out),
            AGPL_O(fbuf), AGPL_O(kdiag), AGPL_O(mu0), AGPL_O(y), AGPL_O(gamma), AGPL_O(beta), AGPL_O(f_out), ctx->pg_retry);
        aux_sample_pg1_retry_kernel<GIBBS><<<(unsigned)nr, kBlock, 0, ctx->stream>>>(
            ctx->pg_retry, GIBBS ? AGPL_O(fbuf) : AGPL_O(f), AGPL_O(omega), ctx->seed, (uint64_t)ctx->point_offset + (uint64_t)o, sweep,
            AGPL_O(nuni_out), AGPL_O(nterms_out), AGPL_O(y), AGPL_O(gamma), AGPL_O(beta));
#undef AGPL_O
    }
}

#ifdef AGPL_PG_TRACE
extern "C" __attribute__((visibility("default"))) int32_t agpl_debug_pgtrace(unsigned long long *out_host, int32_t reset) {
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_pgtrace), sizeof(unsigned long long) * 8) != hipSuccess) return -4;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_pgtrace), z, sizeof(z)) != hipSuccess) return -4;
    }
    return 0;
}
#endif

__global__ __launch_bounds__(kBlock) void rand_pg_kernel(double b, double c, int64_t n, uint64_t seed,
                                                         uint32_t sweep, double *__restrict__ out,
                                                         uint32_t *__restrict__ nuni_out,
                                                         uint32_t *__restrict__ nterms_out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        Philox g;
        g.init(seed, (uint64_t)i, sweep);
        uint32_t nt = 0;
        out[i] = rand_pg(g, 0, b, c, nt);
        if (nuni_out) nuni_out[i] = g.nuni;
        if (nterms_out) nterms_out[i] = nt;
    }
}

// ------------------------------------------------------------------------------------------------
// aux_posterior!  (bernoulli.jl:17-25, negativebinomial.jl:24-33, studentt.jl:50-58,
// categorical.jl:80-110, poisson.jl:30-39, laplace.jl:44-52, heteroscedasticgaussian.jl:34-46)
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T second_moment(T mu, T var) { return mu * mu + var; } // utils.jl:1-3
template <typename T>
__device__ __forceinline__ T second_moment_y(T mu, T var, T y) { return (mu - y) * (mu - y) + var; } // :5-7

template <typename T>
__global__ __launch_bounds__(kBlock) void aux_posterior_kernel(agpl_lik_dev lik, int64_t n, const void *yv,
                                                               const T *__restrict__ mu,
                                                               const T *__restrict__ var, T *__restrict__ out1,
                                                               T *__restrict__ out2, T *__restrict__ out3) {
    const int L = lik.nlatent;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        switch (lik.kind) {
        case AGPL_LIK_BERNOULLI_LOGISTIC:
        case AGPL_LIK_NEGBINOMIAL:
            out1[i] = sqrt(second_moment(mu[i], var[i]));
            break;
        case AGPL_LIK_STUDENTT: {
            const T *y = (const T *)yv;
            T nu = (T)lik.p[0], sg = (T)lik.p[1];
            out1[i] = (nu / (sg * sg) + second_moment_y(mu[i], var[i], y[i])) / T(2);
        } break;
        case AGPL_LIK_CATEGORICAL:
        case AGPL_LIK_CATEGORICAL_BIJ: {
            T den = lik.kind == AGPL_LIK_CATEGORICAL ? (T)L : (T)(lik.cat_const + (double)L);
            for (int k = 0; k < L; ++k) {
                T c = sqrt(second_moment(mu[i * L + k], var[i * L + k]));
                out1[i * L + k] = c;
                out2[i * L + k] = approx_expected_logistic(-mu[i * L + k], c) / den;
            }
        } break;
        case AGPL_LIK_POISSON: {
            T c = sqrt(second_moment(mu[i], var[i]));
            out1[i] = c;
            out2[i] = (T)lik.p[0] * approx_expected_logistic(-mu[i], c);
        } break;
        case AGPL_LIK_LAPLACE: {
            const T *y = (const T *)yv;
            out1[i] = T(1) / (T(2) * (T)lik.p[0] * sqrt(second_moment_y(mu[i], var[i], y[i])));
        } break;
        case AGPL_LIK_HETEROGAUSS: {
            const T *y = (const T *)yv;
            T psi = second_moment_y(mu[2 * i], var[2 * i], y[i]) / T(2);
            T c = sqrt(second_moment(mu[2 * i + 1], var[2 * i + 1]));
            out3[i] = psi;
            out1[i] = c;
            out2[i] = (T)lik.p[0] * approx_expected_logistic(-mu[2 * i + 1], c) * psi;
        } break;
        default:
            break;
        }
    }
}

// expected_auglik_potential / expected_auglik_precision for one point (shared with the fused pass)
// q1, q2 indexed [i*L + k]; outputs [k*n + i].
template <typename T>
__device__ __forceinline__ void expected_pp_point(const agpl_lik_dev &lik, int64_t n, int64_t i, const void *yv,
                                                  const T *q1, const T *q2, const T *mu_g, T *beta, T *gamma) {
    const int L = lik.nlatent;
    switch (lik.kind) {
    case AGPL_LIK_BERNOULLI_LOGISTIC: { // bernoulli.jl:27-29,41-45
        const uint8_t *y = (const uint8_t *)yv;
        beta[i] = y[i] ? T(0.5) : T(-0.5);
        gamma[i] = pg_mean(T(1), q1[i]);
    } break;
    case AGPL_LIK_NEGBINOMIAL: { // negativebinomial.jl:35-37,47-49
        const int32_t *y = (const int32_t *)yv;
        beta[i] = ((T)y[i] - (T)lik.p[0]) / T(2);
        gamma[i] = pg_mean((T)y[i] + (T)lik.p[0], q1[i]);
    } break;
    case AGPL_LIK_STUDENTT: { // studentt.jl:41-43,68-74
        const T *y = (const T *)yv;
        T w = (((T)lik.p[0] + T(1)) / T(2)) * (T(1) / q1[i]);
        gamma[i] = w;
        beta[i] = w * y[i];
    } break;
    case AGPL_LIK_CATEGORICAL:
    case AGPL_LIK_CATEGORICAL_BIJ: { // categorical.jl:121-136, polyagammanegativemultinomial.jl:41-49
        const uint8_t *y = (const uint8_t *)yv;
        T sp = T(0);
        for (int k = 0; k < L; ++k) sp += q2[i * L + k];
        T p0 = T(1) - sp;
        for (int k = 0; k < L; ++k) {
            T nbar = T(1) / p0 * q2[i * L + k];
            T yk = (T)y[i * L + k];
            beta[(int64_t)k * n + i] = (yk - nbar) / T(2);
            gamma[(int64_t)k * n + i] = pg_mean(yk + nbar, q1[i * L + k]);
        }
    } break;
    case AGPL_LIK_POISSON: { // poisson.jl:49-60, polyagammapoisson.jl:35-41
        const int32_t *y = (const int32_t *)yv;
        T nbar = q2[i];
        beta[i] = ((T)y[i] - nbar) / T(2);
        gamma[i] = pg_mean((T)y[i] + nbar, q1[i]);
    } break;
    case AGPL_LIK_LAPLACE: { // laplace.jl:62-68
        const T *y = (const T *)yv;
        gamma[i] = T(2) * q1[i];
        beta[i] = T(2) * q1[i] * y[i];
    } break;
    case AGPL_LIK_HETEROGAUSS: { // heteroscedasticgaussian.jl:94-104
        const T *y = (const T *)yv;
        T lsg = (T)lik.p[0] * (T(1) - approx_expected_logistic(-mu_g[i], q1[i]));
        T nbar = q2[i];
        beta[i] = y[i] * lsg / T(2);
        gamma[i] = lsg;
        beta[n + i] = (T(0.5) - nbar) / T(2);
        gamma[n + i] = pg_mean(T(0.5) + nbar, q1[i]);
    } break;
    default:
        break;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void expected_pp_kernel(agpl_lik_dev lik, int64_t n, const void *yv,
                                                             const T *__restrict__ q1, const T *__restrict__ q2,
                                                             const T *__restrict__ mu_g, T *__restrict__ beta,
                                                             T *__restrict__ gamma) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        expected_pp_point<T>(lik, n, i, yv, q1, q2, mu_g, beta, gamma);
}

// auglik_potential / auglik_precision (sampled twins)
__global__ __launch_bounds__(kBlock) void potential_precision_kernel(agpl_lik_dev lik, int64_t n, const void *yv,
                                                                     const double *__restrict__ omega,
                                                                     const int64_t *__restrict__ nn,
                                                                     const double *__restrict__ fg,
                                                                     double *__restrict__ beta,
                                                                     double *__restrict__ gamma) {
    const int L = lik.nlatent;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        switch (lik.kind) {
        case AGPL_LIK_BERNOULLI_LOGISTIC: {
            const uint8_t *y = (const uint8_t *)yv;
            beta[i] = y[i] ? 0.5 : -0.5;
            gamma[i] = omega[i];
        } break;
        case AGPL_LIK_NEGBINOMIAL: {
            const int32_t *y = (const int32_t *)yv;
            beta[i] = ((double)y[i] - lik.p[0]) / 2.0;
            gamma[i] = omega[i];
        } break;
        case AGPL_LIK_STUDENTT: {
            const double *y = (const double *)yv;
            beta[i] = y[i] * omega[i];
            gamma[i] = omega[i];
        } break;
        case AGPL_LIK_CATEGORICAL:
        case AGPL_LIK_CATEGORICAL_BIJ: {
            const uint8_t *y = (const uint8_t *)yv;
            for (int k = 0; k < L; ++k) {
                beta[(int64_t)k * n + i] = ((double)y[i * L + k] - (double)nn[i * L + k]) / 2.0;
                gamma[(int64_t)k * n + i] = omega[i * L + k];
            }
        } break;
        case AGPL_LIK_POISSON: {
            const int32_t *y = (const int32_t *)yv;
            beta[i] = ((double)y[i] - (double)nn[i]) / 2.0;
            gamma[i] = omega[i];
        } break;
        case AGPL_LIK_LAPLACE: {
            const double *y = (const double *)yv;
            beta[i] = 2.0 * omega[i] * y[i];
            gamma[i] = 2.0 * omega[i];
        } break;
        case AGPL_LIK_HETEROGAUSS: {
            const double *y = (const double *)yv;
            double il = lik.p[0] * logistic(fg[2 * i + 1]);
            beta[i] = y[i] * il;
            gamma[i] = il;
            beta[n + i] = (0.5 - (double)nn[i]) / 2.0;
            gamma[n + i] = omega[i];
        } break;
        default:
            break;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ELBO N-reductions.  Per-point terms (float64), block tree-reduce, fixed-order final sum:
// bitwise reproducible.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double logcosh_(double x) { // LogExpFunctions.logcosh
    double ax = fabs(x);
    return ax + log1p(exp(-2.0 * ax)) - kLogTwo;
}
__device__ __forceinline__ double pg_logtilt(double omega, double b, double c) { // polyagamma.jl:108-110
    return b * logcosh_(c / 2.0) - c * c * omega / 2.0;
}
__device__ __forceinline__ double pg_kl(double b, double c) { // polyagamma.jl:99-106
    return pg_logtilt(pg_mean(b, c), b, c);
}
__device__ __forceinline__ double negbin_logconst(double y, double r) { // negativebinomial.jl:51-52
    return lgamma(y + r) - lgamma(y + 1.0) - lgamma(r);
}
__device__ __forceinline__ double digamma_(double x) {
    double r = 0.0;
    while (x < 6.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x -
           f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f / 132.0))));
}

// logpdf(PolyaGamma(b, c), x) -- polyagamma.jl:37-91: exponential tilt + (b-1) log 2 - (log 2pi + 3 log x) / 2 + the
// log of the 101-term alternating series (n = 0, 2, .., 200), evaluated in the log domain (logsumexp) for x < 1e-2
// exactly as the reference does.  The running product prod_{m<=n} (1 + (b-1)/m) is carried along; the log-domain
// branch makes two passes over the terms (maximum, then sum) instead of materialising them.
__device__ __forceinline__ double log1mexp_(double x) { // LogExpFunctions.log1mexp, x < 0
    return x < -kLogTwo ? log1p(-exp(x)) : log(-expm1(x));
}
__device__ double pg_log_series_term(double x, double b, int n, double logprod) {
    const double Rn = 2.0 * n + b;
    const double log_c_nb = log(n + b) - log(n + 1.0) + log(2.0 / Rn + 1.0);
    const double log_inner = log1mexp_(log_c_nb + ((Rn + 1.0) / (-2.0 * x)));
    return (n == 0 ? 0.0 : logprod) + log(Rn) + Rn * Rn / (-8.0 * x) + log_inner;
}
__device__ double pg_logpdf(double b, double c, double x) {
    if (b == 0.0) return x == 0.0 ? 0.0 : -__builtin_inf();
    const double ext = b * logcosh_(c / 2.0) - c * c * x / 2.0 + (b - 1.0) * kLogTwo - (kLog2Pi + 3.0 * log(x)) / 2.0;
    if (x < 1e-2) {
        double mx = -__builtin_inf(), logprod = 0.0;
        int m = 0;
        for (int n = 0; n <= 200; n += 2) {
            while (m < n) {
                m += 1;
                logprod += log(1.0 + (b - 1.0) / m);
            }
            const double t = pg_log_series_term(x, b, n, logprod);
            mx = t > mx ? t : mx;
        }
        double ssum = 0.0;
        logprod = 0.0;
        m = 0;
        for (int n = 0; n <= 200; n += 2) {
            while (m < n) {
                m += 1;
                logprod += log(1.0 + (b - 1.0) / m);
            }
            ssum += exp(pg_log_series_term(x, b, n, logprod) - mx);
        }
        return ext + mx + log(ssum);
    }
    double prod = 1.0, acc = 0.0;
    int m = 0;
    for (int n = 0; n <= 200; n += 2) {
        while (m < n) {
            m += 1;
            prod *= 1.0 + (b - 1.0) / m;
        }
        const double Rn = 2.0 * n + b;
        const double c_nb = ((n + b) / (n + 1.0)) * (2.0 / Rn + 1.0);
        acc += (n == 0 ? 1.0 : prod) * Rn * exp(Rn * Rn / (-8.0 * x)) * (1.0 - c_nb * exp((Rn + 1.0) / (-2.0 * x)));
    }
    if (!(acc > 2.2250738585072014e-308)) acc = 2.2250738585072014e-308; // max(s, floatmin)
    return ext + log(acc);
}

enum { RED_LOGTILT = 0, RED_EXPECTED_LOGTILT = 1, RED_KL = 2, RED_AUX_PRIOR_LOGPDF = 3, RED_AUG_LOGLIK = 4, RED_EXPECTED_AUG_LOGLIK = 5 };

struct RedArgs {
    const void *y;
    const double *a1; // omega | q1
    const double *a2; // (unused) | q2
    const int64_t *nn;
    const double *f;   // f | mu
    const double *var; // var
};

// One point's expected_logtilt (bernoulli.jl:59-65, negativebinomial.jl:59-65, studentt.jl:80-83, categorical.jl:172-180,
// poisson.jl:76-85, laplace.jl:83-88) and aux_kldivergence (generic.jl:56-62) term from accessors -- y(k), q1(k), q2(k), mu(k),
// var(k), all double, k = latent -- shared by the reduction kernels (accessors over arrays) and by the sweep's per-point kernel
// (accessors over the marginals it has just formed): the same expressions, hence the same float64 results.
template <class Y, class Q1, class Q2, class MU, class VAR>
__device__ __forceinline__ double expected_logtilt_point(const agpl_lik_dev &lik, Y y, Q1 q1, Q2 q2, MU mu, VAR var) {
    const int L = lik.nlatent;
    switch (lik.kind) {
    case AGPL_LIK_BERNOULLI_LOGISTIC: { // bernoulli.jl:59-65
        double s = y(0) != 0.0 ? 1.0 : -1.0;
        double th = pg_mean(1.0, q1(0));
        return -kLogTwo + (s * mu(0) - (mu(0) * mu(0) + var(0)) * th) / 2.0;
    }
    case AGPL_LIK_NEGBINOMIAL: { // negativebinomial.jl:59-65
        double r = lik.p[0], yy = y(0);
        double th = pg_mean(yy + r, q1(0));
        return negbin_logconst(yy, r) - (yy + r) * kLogTwo + (mu(0) * (yy - r) - (mu(0) * mu(0) + var(0)) * th) / 2.0;
    }
    case AGPL_LIK_STUDENTT: { // studentt.jl:80-83
        double th = ((lik.p[0] + 1.0) / 2.0) / q1(0);
        double d = mu(0) - y(0);
        return -0.5 * kLog2Pi + 0.5 * log(th) - 0.5 * d * d * th - var(0) * th / 2.0;
    }
    case AGPL_LIK_CATEGORICAL:
    case AGPL_LIK_CATEGORICAL_BIJ: { // categorical.jl:172-180
        double sp = 0.0;
        for (int k = 0; k < L; ++k) sp += q2(k);
        double p0 = 1.0 - sp, s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < L; ++k) {
            double yk = y(k), nbar = q2(k) / p0;
            double w = pg_mean(yk + nbar, q1(k));
            double m = mu(k), v = var(k);
            s1 += yk + nbar;
            s2 += ((yk - nbar) * m - (m * m + v) * w) / 2.0;
        }
        return -s1 * kLogTwo + s2;
    }
    case AGPL_LIK_POISSON: { // poisson.jl:76-85
        double yy = y(0), nbar = q2(0);
        double w = pg_mean(yy + nbar, q1(0));
        return -(yy + nbar) * kLogTwo + ((yy - nbar) * mu(0) - (mu(0) * mu(0) + var(0)) * w) / 2.0 + yy * log(lik.p[0]) -
               lgamma(yy + 1.0);
    }
    case AGPL_LIK_LAPLACE: { // laplace.jl:83-88
        double yy = y(0);
        return lgamma(0.5) - 0.5 * log(kPi) - log(2.0 * lik.p[0]) - ((mu(0) - yy) * (mu(0) - yy) + var(0)) * q1(0);
    }
    default:
        return __builtin_nan("");
    }
}
template <class Y, class Q1, class Q2>
__device__ __forceinline__ double aux_kl_point(const agpl_lik_dev &lik, Y y, Q1 q1, Q2 q2) {
    const int L = lik.nlatent;
    switch (lik.kind) {
    case AGPL_LIK_BERNOULLI_LOGISTIC:
        return pg_kl(1.0, q1(0));
    case AGPL_LIK_NEGBINOMIAL:
        return pg_kl(y(0) + lik.p[0], q1(0));
    case AGPL_LIK_STUDENTT: { // KL(Gamma(alpha, 1/beta_i) || Gamma(nu/2, 2 sigma^2/nu)) studentt.jl:85-91
        double nu = lik.p[0], sg = lik.p[1];
        double ap = (nu + 1.0) / 2.0, thp = 1.0 / q1(0);
        double aq = nu / 2.0, thq = sg * sg / (nu / 2.0);
        return (ap - aq) * digamma_(ap) - lgamma(ap) + lgamma(aq) + aq * (log(thq) - log(thp)) + ap * (thp - thq) / thq;
    }
    case AGPL_LIK_POISSON: { // polyagammapoisson.jl:47-51
        double lq = q2(0), lp = lik.p[0];
        double klp = lq > 0 ? lq * (log(lq) - log(lp)) - lq + lp : lp;
        return pg_kl(y(0) + lq, q1(0)) + klp;
    }
    case AGPL_LIK_LAPLACE: { // laplace.jl:96-104
        double lam = 1.0 / ((2.0 * lik.p[0]) * (2.0 * lik.p[0]));
        return log(2.0 * lam) / 2.0 - log(2.0 * kPi) / 2.0 - log(lam) / 2.0 + lgamma(0.5) + lam / q1(0);
    }
    case AGPL_LIK_CATEGORICAL_BIJ: { // polyagammanegativemultinomial.jl:56-65, negativemultinomial.jl:72-82
        double sp = 0.0;
        for (int k = 0; k < L; ++k) sp += q2(k);
        double p0 = 1.0 - sp;
        double pp = 1.0 / lik.sum_theta;
        double p0p = 1.0 - L * pp;
        double s = 0.0, acc = 0.0;
        for (int k = 0; k < L; ++k) {
            double nbar = q2(k) / p0;
            acc += pg_kl(y(k) + nbar, q1(k));
            s += q2(k) * (log(q2(k)) - log(pp));
        }
        return acc + log(p0) - log(p0p) + s / p0;
    }
    default:
        return __builtin_nan("");
    }
}
// y of (point i, latent k) as a double, by the likelihood's observation type (REAL: the element type of real-valued y)
template <typename REAL>
struct YAcc {
    const void *y;
    int64_t i;
    int kind, L;
    __device__ __forceinline__ double operator()(int k) const {
        switch (kind) {
        case AGPL_LIK_BERNOULLI_LOGISTIC:
            return (double)((const uint8_t *)y)[i];
        case AGPL_LIK_NEGBINOMIAL:
        case AGPL_LIK_POISSON:
            return (double)((const int32_t *)y)[i];
        case AGPL_LIK_CATEGORICAL:
        case AGPL_LIK_CATEGORICAL_BIJ:
            return (double)((const uint8_t *)y)[i * L + k];
        default:
            return (double)((const REAL *)y)[i];
        }
    }
};

__device__ double red_term(int mode, const agpl_lik_dev &lik, int64_t i, const RedArgs &A);

// logpdf(Poisson(lam), n) -- Distributions.jl closed form (upstream, unpinned)
__device__ __forceinline__ double poisson_logpdf(double lam, double n) {
    if (lam == 0.0) return n == 0.0 ? 0.0 : -__builtin_inf();
    return n * log(lam) - lam - lgamma(n + 1.0);
}
// logdensity_def(aux_prior(lik, y), Omega) per point -- the second half of aug_loglik (generic.jl:48-50).
// PG(1, 0) bernoulli.jl:51-57 ; PG(y + r, 0) negativebinomial.jl:67-73 ; Gamma(nu/2, scale 2 sigma^2/nu) studentt.jl:91 ;
// PolyaGammaPoisson(y, 0, lambda) poisson.jl:67-76 with the joint density of polyagammapoisson.jl:29-33 ;
// InverseGamma(1/2, (2 beta)^-2) laplace.jl:90-96.  The categorical prior goes through the reference's broken logdensity_def
// (polyagammanegativemultinomial.jl:33-39): unsupported.  The heteroscedastic likelihood has no aux_prior
// (its aug_loglik is its own method, below).
__device__ double aux_prior_logpdf_term(const agpl_lik_dev &lik, int64_t i, const RedArgs &A) {
    const double *omega = A.a1;
    switch (lik.kind) {
    case AGPL_LIK_BERNOULLI_LOGISTIC:
        return pg_logpdf(1.0, 0.0, omega[i]);
    case AGPL_LIK_NEGBINOMIAL:
        return pg_logpdf((double)((const int32_t *)A.y)[i] + lik.p[0], 0.0, omega[i]);
    case AGPL_LIK_STUDENTT: {
        const double a = lik.p[0] / 2.0, th = lik.p[1] * lik.p[1] / a;
        return -lgamma(a) - a * log(th) + (a - 1.0) * log(omega[i]) - omega[i] / th;
    }
    case AGPL_LIK_POISSON: {
        const double nk = (double)A.nn[i];
        return poisson_logpdf(lik.p[0], nk) + pg_logpdf((double)((const int32_t *)A.y)[i] + nk, 0.0, omega[i]);
    }
    case AGPL_LIK_LAPLACE: {
        const double lam = 1.0 / ((2.0 * lik.p[0]) * (2.0 * lik.p[0]));
        return 0.5 * log(lam) - lgamma(0.5) - 1.5 * log(omega[i]) - lam / omega[i];
    }
    default:
        return __builtin_nan("");
    }
}
// aug_loglik(lik::AugHeteroGaussian, (omega, n), y, (f, g)) heteroscedasticgaussian.jl:118-128 ; fg = [2, N]
__device__ double hetero_aug_loglik_term(const agpl_lik_dev &lik, int64_t i, const RedArgs &A) {
    const double ff = A.f[2 * i], gg = A.f[2 * i + 1], yy = ((const double *)A.y)[i];
    const double nk = (double)A.nn[i], om = A.a1[i];
    return -(0.5 + nk) * kLogTwo + ((0.5 - nk) * gg - gg * gg * om) / 2.0 + pg_logpdf(0.5 + nk, 0.0, om) +
           poisson_logpdf(lik.p[0] / 2.0 * (yy - ff) * (yy - ff), nk);
}
// expected_aug_loglik(lik::AugHeteroGaussian, qOmega, y, qfg) heteroscedasticgaussian.jl:130-145 ; q1 = c, q2 = lambda of
// aux_posterior!, (mu, var) = q(f), q(g) as [2, N]; `var(first(qg))` is read as var(qg)
__device__ double hetero_expected_aug_loglik_term(const agpl_lik_dev &lik, int64_t i, const RedArgs &A) {
    const double lam = lik.p[0], yy = ((const double *)A.y)[i];
    const double mf = A.f[2 * i], vf = A.var[2 * i], g = A.f[2 * i + 1], vg = A.var[2 * i + 1];
    const double tn = A.a2[i], tw = pg_mean(0.5 + tn, A.a1[i]);
    const double lp = lam / 2.0 * ((yy - mf) * (yy - mf) + vf);
    const double klp = tn > 0 ? tn * (log(tn) - log(lp)) - tn + lp : lp;
    return 0.5 * (log(lam) + log(2.0 / kPi)) - (0.5 + tn) * kLogTwo + ((0.5 - tn) * g - (g * g + vg) * tw) / 2.0 +
           pg_kl(0.5 + tn, A.a1[i]) + klp;
}

__device__ double red_term(int mode, const agpl_lik_dev &lik, int64_t i, const RedArgs &A) {
    const int L = lik.nlatent;
    const double nanv = __builtin_nan("");
    if (mode == RED_AUX_PRIOR_LOGPDF) return aux_prior_logpdf_term(lik, i, A);
    if (lik.kind == AGPL_LIK_HETEROGAUSS) // the two methods the reference defines for it; everything else is refused on the host
        return mode == RED_AUG_LOGLIK ? hetero_aug_loglik_term(lik, i, A) : hetero_expected_aug_loglik_term(lik, i, A);
    if (mode == RED_AUG_LOGLIK) return red_term(RED_LOGTILT, lik, i, A) + aux_prior_logpdf_term(lik, i, A);
    if (mode == RED_EXPECTED_AUG_LOGLIK) // generic.jl:52-54: expected_logtilt + aux_kldivergence (the sign is the reference's)
        return red_term(RED_EXPECTED_LOGTILT, lik, i, A) + red_term(RED_KL, lik, i, A);
    if (mode == RED_LOGTILT) {
        const double *omega = A.a1, *f = A.f;
        switch (lik.kind) {
        case AGPL_LIK_BERNOULLI_LOGISTIC: { // bernoulli.jl:47-49
            double s = ((const uint8_t *)A.y)[i] ? 1.0 : -1.0;
            return -kLogTwo + (s * f[i] - f[i] * f[i] * omega[i]) / 2.0;
        }
        case AGPL_LIK_NEGBINOMIAL: { // negativebinomial.jl:54-57
            double r = lik.p[0], yy = (double)((const int32_t *)A.y)[i];
            return negbin_logconst(yy, r) - (yy + r) * kLogTwo + (f[i] * (yy - r) - f[i] * f[i] * omega[i]) / 2.0;
        }
        case AGPL_LIK_STUDENTT: { // studentt.jl:76-78
            double d = ((const double *)A.y)[i] - f[i];
            return -0.5 * kLog2Pi + 0.5 * log(omega[i]) - 0.5 * d * d * omega[i];
        }
        case AGPL_LIK_CATEGORICAL:
        case AGPL_LIK_CATEGORICAL_BIJ: { // categorical.jl:138-145
            const uint8_t *y = (const uint8_t *)A.y;
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < L; ++k) {
                double yk = (double)y[i * L + k], nk = (double)A.nn[i * L + k], fk = f[i * L + k];
                s1 += yk + nk;
                s2 += (yk - nk) * fk - fk * fk * omega[i * L + k];
            }
            return -s1 * kLogTwo + s2 / 2.0;
        }
        case AGPL_LIK_POISSON: { // poisson.jl:62-65
            double yy = (double)((const int32_t *)A.y)[i], nk = (double)A.nn[i];
            return yy * log(lik.p[0]) - (yy + nk) * kLogTwo - lgamma(yy + 1.0) +
                   ((yy - nk) * f[i] - f[i] * f[i] * omega[i]) / 2.0;
        }
        case AGPL_LIK_LAPLACE: { // laplace.jl:78-81
            double d = ((const double *)A.y)[i] - f[i];
            return lgamma(0.5) - 0.5 * log(kPi) - log(2.0 * lik.p[0]) - d * d * omega[i];
        }
        default:
            return nanv;
        }
    }
    const double *q1 = A.a1, *q2 = A.a2;
    const YAcc<double> y{A.y, i, lik.kind, L};
    auto q1a = [&](int k) { return q1[i * L + k]; };
    auto q2a = [&](int k) { return q2[i * L + k]; };
    if (mode == RED_EXPECTED_LOGTILT) {
        const double *mu = A.f, *var = A.var;
        return expected_logtilt_point(lik, y, q1a, q2a, [&](int k) { return mu[i * L + k]; }, [&](int k) { return var[i * L + k]; });
    }
    return aux_kl_point(lik, y, q1a, q2a); // RED_KL: aux_kldivergence generic.jl:56-62
}

__global__ __launch_bounds__(kBlock) void reduce_terms_kernel(int mode, agpl_lik_dev lik, int64_t n, RedArgs A,
                                                              double *__restrict__ partial) {
    __shared__ double sm[kBlock];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        acc += red_term(mode, lik, i, A);
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ void reduce_final_kernel(int nparts, const double *__restrict__ partial, double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double acc = 0.0;
        for (int i = 0; i < nparts; ++i) acc += partial[i];
        *out = acc;
    }
}

int32_t run_reduction(agpl_ctx *ctx, int mode, const agpl_lik_desc *lik, int64_t n, const RedArgs &A,
                      double *out_host) {
    if (!ctx || !out_host) return AGPL_ERR_INVALID_ARGUMENT;
    agpl_lik_dev ld;
    int32_t rc = agpl_lik_to_device(ctx, lik, &ld);
    if (rc) return rc;
    if ((mode == RED_KL || mode == RED_EXPECTED_AUG_LOGLIK) && lik->kind == AGPL_LIK_CATEGORICAL)
        AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED,
                  "the kl-divergence cannot be computed for the non-bijective LogisticSoftMaxLink "
                  "(categorical.jl:165-170); use the bijective link");
    if (lik->kind == AGPL_LIK_HETEROGAUSS && mode != RED_AUG_LOGLIK && mode != RED_EXPECTED_AUG_LOGLIK)
        AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED,
                  "the heteroscedastic likelihood defines aug_loglik and expected_aug_loglik only "
                  "(heteroscedasticgaussian.jl:106-145): its tilt, prior and KL are not split in the reference");
    if ((mode == RED_AUX_PRIOR_LOGPDF || mode == RED_AUG_LOGLIK) &&
        (lik->kind == AGPL_LIK_CATEGORICAL || lik->kind == AGPL_LIK_CATEGORICAL_BIJ))
        AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED,
                  "aug_loglik / the aux-prior log-density of the categorical likelihood: the reference's logdensity_def of "
                  "PolyaGammaNegativeMultinomial is broken (polyagammanegativemultinomial.jl:33-39, SURVEY App. B)");
    const bool wants_n = lik->kind == AGPL_LIK_POISSON || lik->kind == AGPL_LIK_HETEROGAUSS;
    if ((mode == RED_AUX_PRIOR_LOGPDF || mode == RED_AUG_LOGLIK) && wants_n && n > 0 && !A.nn)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "this likelihood's prior density needs the counts n_aux");
    if (n <= 0) {
        *out_host = 0.0;
        return AGPL_OK;
    }
    // partials live in bytes 64..8191 of the small scratch: bytes 8192..16383 belong to the marginal item queues and the
    // factor hand-off flags, which must read zero between launches (agpl_split.hip, agpl_factor.hip)
    int nb = grid_for(n);
    if (nb > kRedParts) nb = kRedParts;
    rc = agpl_ws2_reserve(ctx, 16384);
    if (rc) return rc;
    double *partial = (double *)ctx->ws2;
    reduce_terms_kernel<<<nb, kBlock, 0, ctx->stream>>>(mode, ld, n, A, partial + 8);
    AGPL_LAUNCH_CHECK(ctx);
    reduce_final_kernel<<<1, 64, 0, ctx->stream>>>(nb, partial + 8, partial);
    AGPL_LAUNCH_CHECK(ctx);
    AGPL_HIP(ctx, hipMemcpyAsync(out_host, partial, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AGPL_OK;
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
// the sampler kernels' flag word: bit 0 = invalid NegativeMultinomial parameters, bit 1 = a PolyaGamma(b, c) draw with
// b >= 2^22 (agpl_random.h kPgMaxB: outside the numbering of a point's draws).  Read (one stream synchronisation) for the likelihoods that can set it.
int32_t agpl_sampler_outcome(agpl_ctx *ctx, int32_t kind, const int *bad) {
    if (kind != AGPL_LIK_CATEGORICAL && kind != AGPL_LIK_CATEGORICAL_BIJ && kind != AGPL_LIK_NEGBINOMIAL &&
        kind != AGPL_LIK_POISSON && kind != AGPL_LIK_HETEROGAUSS)
        return AGPL_OK;
    int hbad = 0;
    AGPL_HIP(ctx, hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    AGPL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hbad & 1)
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT,
                  "NegativeMultinomial: all p should be positive and their sum strictly smaller than 1 "
                  "(negativemultinomial.jl:17-22)");
    if (hbad & 2)
        AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED,
                  "PolyaGamma(b, c) with b >= 4194304 = 2^22 (y + r, or y + n): outside this build's numbering of the PG(1, c) "
                  "draws of a point; the outputs of that point are NaN");
    return AGPL_OK;
}

extern "C" int32_t agpl_aux_sample(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                   const double *f, double *omega_out, int64_t *n_out, uint32_t sweep,
                                   uint32_t *nuni_out, uint32_t *nterms_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    agpl_lik_dev ld;
    int32_t rc = agpl_lik_to_device(ctx, lik, &ld);
    if (rc) return rc;
    if (n < 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "n < 0");
    if (n == 0) return AGPL_OK;
    if (!f || !omega_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null f / omega_out");
    const bool needs_n = ld.kind == AGPL_LIK_CATEGORICAL || ld.kind == AGPL_LIK_CATEGORICAL_BIJ ||
                         ld.kind == AGPL_LIK_POISSON || ld.kind == AGPL_LIK_HETEROGAUSS;
    if (needs_n && !n_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "this likelihood needs n_out");
    if (ld.kind != AGPL_LIK_BERNOULLI_LOGISTIC && !y) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null y");
    rc = agpl_ws2_reserve(ctx, sizeof(double) * (1024 + 8));
    if (rc) return rc;
    int *bad = (int *)ctx->ws2;
    AGPL_HIP(ctx, hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    if (ld.kind == AGPL_LIK_BERNOULLI_LOGISTIC) {
        rc = agpl_pg_retry_reserve(ctx, n < kPg1MaxLaunch ? n : kPg1MaxLaunch);
        if (rc) return rc;
    }
    rc = agpl_timing_begin(ctx, 3);
    if (rc) return rc;
#define AGPL_LAUNCH_AUX(K)                                                                                     \
    case K:                                                                                                    \
        aux_sample_kernel<K><<<grid_for(n), kBlock, 0, ctx->stream>>>(ld, n, y, f, omega_out, n_out, ctx->seed,   \
                                                                      (uint64_t)ctx->point_offset, sweep,      \
                                                                      nuni_out, nterms_out, bad);              \
        break;
    switch (ld.kind) {
    case AGPL_LIK_BERNOULLI_LOGISTIC: { // one draw per point: the kernel with the long trial queue
        launch_pg1<false>(ctx, n, f, omega_out, sweep, nuni_out, nterms_out, nullptr, nullptr, nullptr, nullptr, nullptr,
                          nullptr, nullptr);
    } break;
        AGPL_LAUNCH_AUX(AGPL_LIK_NEGBINOMIAL)
        AGPL_LAUNCH_AUX(AGPL_LIK_STUDENTT)
        AGPL_LAUNCH_AUX(AGPL_LIK_CATEGORICAL)
        AGPL_LAUNCH_AUX(AGPL_LIK_CATEGORICAL_BIJ)
        AGPL_LAUNCH_AUX(AGPL_LIK_POISSON)
        AGPL_LAUNCH_AUX(AGPL_LIK_LAPLACE)
        AGPL_LAUNCH_AUX(AGPL_LIK_HETEROGAUSS)
    default:
        break;
    }
#undef AGPL_LAUNCH_AUX
    AGPL_LAUNCH_CHECK(ctx);
    rc = agpl_timing_end(ctx, 3);
    if (rc) return rc;
    return agpl_sampler_outcome(ctx, ld.kind, bad);
}

extern "C" int32_t agpl_rand_polyagamma(agpl_ctx *ctx, double b, double c, int64_t n, uint32_t sweep,
                                        double *out, uint32_t *nuni_out, uint32_t *nterms_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    if (n < 0 || !(b >= 0.0)) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "need n >= 0 and b >= 0");
    if (b >= kPgMaxB)
        AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED, "PolyaGamma(b, c) with b >= 4194304 = 2^22: outside this build's numbering of the "
                                            "PG(1, c) draws of a point (b = %g)", b);
    if (n == 0) return AGPL_OK;
    if (!out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null out");
    rand_pg_kernel<<<grid_for(n), kBlock, 0, ctx->stream>>>(b, c, n, ctx->seed, sweep, out, nuni_out, nterms_out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

extern "C" int32_t agpl_potential_precision(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                            const double *omega, const int64_t *n_aux, const double *fg,
                                            double *beta_out, double *gamma_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    agpl_lik_dev ld;
    int32_t rc = agpl_lik_to_device(ctx, lik, &ld);
    if (rc) return rc;
    if (n < 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "n < 0");
    if (n == 0) return AGPL_OK;
    if (!y || !omega || !beta_out || !gamma_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    const bool needs_n = ld.kind == AGPL_LIK_CATEGORICAL || ld.kind == AGPL_LIK_CATEGORICAL_BIJ ||
                         ld.kind == AGPL_LIK_POISSON || ld.kind == AGPL_LIK_HETEROGAUSS;
    if (needs_n && !n_aux) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "this likelihood needs n_aux");
    if (ld.kind == AGPL_LIK_HETEROGAUSS && !fg) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "heterogauss needs fg");
    potential_precision_kernel<<<grid_for(n), kBlock, 0, ctx->stream>>>(ld, n, y, omega, n_aux, fg, beta_out,
                                                                        gamma_out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

extern "C" int32_t agpl_aux_posterior(agpl_ctx *ctx, const agpl_lik_desc *lik, int32_t dtype, int64_t n,
                                      const void *y, const void *mu, const void *var, void *out1, void *out2,
                                      void *out3) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    agpl_lik_dev ld;
    int32_t rc = agpl_lik_to_device(ctx, lik, &ld);
    if (rc) return rc;
    if (n < 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "n < 0");
    if (n == 0) return AGPL_OK;
    if (!mu || !var || !out1) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null mu / var / out1");
    const bool needs2 = ld.kind == AGPL_LIK_CATEGORICAL || ld.kind == AGPL_LIK_CATEGORICAL_BIJ ||
                        ld.kind == AGPL_LIK_POISSON || ld.kind == AGPL_LIK_HETEROGAUSS;
    if (needs2 && !out2) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "this likelihood needs out2");
    if (ld.kind == AGPL_LIK_HETEROGAUSS && !out3) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "heterogauss needs out3");
    const bool needs_y = ld.kind == AGPL_LIK_STUDENTT || ld.kind == AGPL_LIK_LAPLACE || ld.kind == AGPL_LIK_HETEROGAUSS;
    if (needs_y && !y) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null y");
    if (dtype == AGPL_F64)
        aux_posterior_kernel<double><<<grid_for(n), kBlock, 0, ctx->stream>>>(
            ld, n, y, (const double *)mu, (const double *)var, (double *)out1, (double *)out2, (double *)out3);
    else if (dtype == AGPL_F32)
        aux_posterior_kernel<float><<<grid_for(n), kBlock, 0, ctx->stream>>>(
            ld, n, y, (const float *)mu, (const float *)var, (float *)out1, (float *)out2, (float *)out3);
    else
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "dtype must be AGPL_F32 or AGPL_F64");
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

extern "C" int32_t agpl_expected_potential_precision(agpl_ctx *ctx, const agpl_lik_desc *lik, int32_t dtype,
                                                     int64_t n, const void *y, const void *q1, const void *q2,
                                                     const void *mu_g, void *beta_out, void *gamma_out) {
    if (!ctx) return AGPL_ERR_INVALID_ARGUMENT;
    agpl_lik_dev ld;
    int32_t rc = agpl_lik_to_device(ctx, lik, &ld);
    if (rc) return rc;
    if (n < 0) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "n < 0");
    if (n == 0) return AGPL_OK;
    if (!y || !q1 || !beta_out || !gamma_out) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "null argument");
    const bool needs2 = ld.kind == AGPL_LIK_CATEGORICAL || ld.kind == AGPL_LIK_CATEGORICAL_BIJ ||
                        ld.kind == AGPL_LIK_POISSON || ld.kind == AGPL_LIK_HETEROGAUSS;
    if (needs2 && !q2) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "this likelihood needs q2");
    if (ld.kind == AGPL_LIK_HETEROGAUSS && !mu_g) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "heterogauss needs mu_g");
    if (dtype == AGPL_F64)
        expected_pp_kernel<double><<<grid_for(n), kBlock, 0, ctx->stream>>>(
            ld, n, y, (const double *)q1, (const double *)q2, (const double *)mu_g, (double *)beta_out,
            (double *)gamma_out);
    else if (dtype == AGPL_F32)
        expected_pp_kernel<float><<<grid_for(n), kBlock, 0, ctx->stream>>>(
            ld, n, y, (const float *)q1, (const float *)q2, (const float *)mu_g, (float *)beta_out,
            (float *)gamma_out);
    else
        AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "dtype must be AGPL_F32 or AGPL_F64");
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

extern "C" int32_t agpl_logtilt(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                const double *omega, const int64_t *n_aux, const double *f, double *out_host) {
    RedArgs A{y, omega, nullptr, n_aux, f, nullptr};
    return run_reduction(ctx, RED_LOGTILT, lik, n, A, out_host);
}
extern "C" int32_t agpl_aux_prior_logpdf(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                         const double *omega, const int64_t *n_aux, double *out_host) {
    RedArgs A{y, omega, nullptr, n_aux, nullptr, nullptr};
    return run_reduction(ctx, RED_AUX_PRIOR_LOGPDF, lik, n, A, out_host);
}
extern "C" int32_t agpl_aug_loglik(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                   const double *omega, const int64_t *n_aux, const double *f, double *out_host) {
    RedArgs A{y, omega, nullptr, n_aux, f, nullptr};
    return run_reduction(ctx, RED_AUG_LOGLIK, lik, n, A, out_host);
}
extern "C" int32_t agpl_expected_logtilt(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                         const double *q1, const double *q2, const double *mu, const double *var,
                                         double *out_host) {
    RedArgs A{y, q1, q2, nullptr, mu, var};
    return run_reduction(ctx, RED_EXPECTED_LOGTILT, lik, n, A, out_host);
}
extern "C" int32_t agpl_aux_kldivergence(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                         const double *q1, const double *q2, double *out_host) {
    RedArgs A{y, q1, q2, nullptr, nullptr, nullptr};
    return run_reduction(ctx, RED_KL, lik, n, A, out_host);
}
extern "C" int32_t agpl_expected_aug_loglik(agpl_ctx *ctx, const agpl_lik_desc *lik, int64_t n, const void *y,
                                            const double *q1, const double *q2, const double *mu, const double *var,
                                            double *out_host) {
    RedArgs A{y, q1, q2, nullptr, mu, var};
    return run_reduction(ctx, RED_EXPECTED_AUG_LOGLIK, lik, n, A, out_host);
}

// fused elementwise step of a sweep: aux_posterior! + expected potential / precision of point i from its marginals.
// MG gives the marginal of latent k (m(k, i), v(k, i)); OUT takes (gamma, beta) of latent k.  One code path for both callers:
// agpl_fused_elementwise_kernel (marginals in arrays, outputs in arrays) and agpl_fused_point_kernel (marginals summed on
// the fly from the marginal kernel's row-block partials, outputs as the accumulation's gamma | beta records).
template <class MG, class OUT>
__device__ __forceinline__ void fused_point(const agpl_lik_dev &lik, int64_t n, int64_t i, const void *yv, const MG &mg,
                                            OUT &out, float *__restrict__ c_out) {
    const int L = lik.nlatent;
    float og_[2], ob_[2]; // (single- and two-latent kinds; the categorical kinds put per k)
#define out_g(k_) og_[k_]
#define out_b(k_) ob_[k_]
    {
        switch (lik.kind) {
        case AGPL_LIK_BERNOULLI_LOGISTIC: {
            const uint8_t *y = (const uint8_t *)yv;
            float c = sqrtf(second_moment(mg.m(0, i), mg.v(0, i)));
            out_g(0) = pg_mean(1.0f, c);
            out_b(0) = y[i] ? 0.5f : -0.5f;
            if (c_out) c_out[i] = c;
        } break;
        case AGPL_LIK_NEGBINOMIAL: {
            const int32_t *y = (const int32_t *)yv;
            float c = sqrtf(second_moment(mg.m(0, i), mg.v(0, i)));
            float r = (float)lik.p[0];
            out_g(0) = pg_mean((float)y[i] + r, c);
            out_b(0) = ((float)y[i] - r) / 2.0f;
            if (c_out) c_out[i] = c;
        } break;
        case AGPL_LIK_STUDENTT: {
            const float *y = (const float *)yv;
            float nu = (float)lik.p[0], sg = (float)lik.p[1];
            float bi = (nu / (sg * sg) + second_moment_y(mg.m(0, i), mg.v(0, i), y[i])) / 2.0f;
            float w = ((nu + 1.0f) / 2.0f) * (1.0f / bi);
            out_g(0) = w;
            out_b(0) = w * y[i];
            if (c_out) c_out[i] = bi;
        } break;
        case AGPL_LIK_POISSON: {
            const int32_t *y = (const int32_t *)yv;
            const float m0 = mg.m(0, i);
            float c = sqrtf(second_moment(m0, mg.v(0, i)));
            float nbar = (float)lik.p[0] * approx_expected_logistic(-m0, c);
            out_g(0) = pg_mean((float)y[i] + nbar, c);
            out_b(0) = ((float)y[i] - nbar) / 2.0f;
            if (c_out) c_out[i] = c;
        } break;
        case AGPL_LIK_LAPLACE: {
            const float *y = (const float *)yv;
            float m = 1.0f / (2.0f * (float)lik.p[0] * sqrtf(second_moment_y(mg.m(0, i), mg.v(0, i), y[i])));
            out_g(0) = 2.0f * m;
            out_b(0) = 2.0f * m * y[i];
            if (c_out) c_out[i] = m;
        } break;
        case AGPL_LIK_CATEGORICAL:
        case AGPL_LIK_CATEGORICAL_BIJ: {
            const uint8_t *y = (const uint8_t *)yv;
            float den = lik.kind == AGPL_LIK_CATEGORICAL ? (float)L : (float)(lik.cat_const + (double)L);
            float sp = 0.0f;
            for (int k = 0; k < L; ++k) {
                float m = mg.m(k, i);
                float c = sqrtf(second_moment(m, mg.v(k, i)));
                sp += approx_expected_logistic(-m, c) / den;
            }
            float p0 = 1.0f - sp;
            for (int k = 0; k < L; ++k) {
                float m = mg.m(k, i);
                float c = sqrtf(second_moment(m, mg.v(k, i)));
                float p = approx_expected_logistic(-m, c) / den;
                float nbar = 1.0f / p0 * p;
                float yk = (float)y[i * L + k];
                const float bk_ = (yk - nbar) / 2.0f;
                const float gk_ = pg_mean(yk + nbar, c);
                out.put(k, i, gk_, bk_);
                if (c_out) c_out[i * L + k] = c;
            }
        } break;
        case AGPL_LIK_HETEROGAUSS: {
            const float *y = (const float *)yv;
            float psi = second_moment_y(mg.m(0, i), mg.v(0, i), y[i]) / 2.0f;
            float mgs = mg.m(1, i);
            float c = sqrtf(second_moment(mgs, mg.v(1, i)));
            float ael = approx_expected_logistic(-mgs, c);
            float lam = (float)lik.p[0];
            float nbar = lam * ael * psi;
            float lsg = lam * (1.0f - ael);
            out_b(0) = y[i] * lsg / 2.0f;
            out_g(0) = lsg;
            out_b(1) = (0.5f - nbar) / 2.0f;
            out_g(1) = pg_mean(0.5f + nbar, c);
            if (c_out) c_out[i] = c;
        } break;
        default:
            break;
        }
    }
#undef out_g
#undef out_b
    if (lik.kind != AGPL_LIK_CATEGORICAL && lik.kind != AGPL_LIK_CATEGORICAL_BIJ) {
        out.put(0, i, og_[0], ob_[0]);
        if (lik.kind == AGPL_LIK_HETEROGAUSS) out.put(1, i, og_[1], ob_[1]);
    }
}

// expected_logtilt_i - aux_kldivergence_i (the per-point part of aug_elbo, examples/bernoulli/script.jl:65-70) for q(f_i) = the
// marginal `mg` gives and qOmega_i = aux_posterior(lik, y_i, q(f_i)), evaluated in float64 FROM the float32 marginals -- the value
// the float64 operator kernels (aux_posterior_kernel<double>, reduce_terms_kernel) give for the same marginals.  NaN for the
// likelihoods whose terms the reference does not define (non-bijective categorical KL, heteroscedastic).
template <class MG>
__device__ __forceinline__ double elbo_point(const agpl_lik_dev &lik, int64_t i, const void *yv, const MG &mg) {
    const int L = lik.nlatent;
    const YAcc<float> y{yv, i, lik.kind, L};
    // Bernoulli / negative binomial: with theta = E[omega] = b tanh(c / 2) / (2 c) and c^2 = mu^2 + sigma^2 the theta terms of
    // expected_logtilt (.. - c^2 theta / 2) and of KL(PG(b, c) || PG(b, 0)) = b logcosh(c / 2) - c^2 theta / 2 cancel: what is
    // left needs one logcosh (0.78 -> ~0.1 ms per 1e7 points against the literal expressions; equal to them to rounding)
    if (lik.kind == AGPL_LIK_BERNOULLI_LOGISTIC) {
        const double m = (double)mg.m(0, i), c = sqrt(second_moment(m, (double)mg.v(0, i)));
        return -kLogTwo + (y(0) != 0.0 ? m : -m) / 2.0 - logcosh_(c / 2.0);
    }
    if (lik.kind == AGPL_LIK_NEGBINOMIAL) {
        const double m = (double)mg.m(0, i), c = sqrt(second_moment(m, (double)mg.v(0, i)));
        const double r = lik.p[0], yy = y(0);
        return negbin_logconst(yy, r) - (yy + r) * kLogTwo + m * (yy - r) / 2.0 - (yy + r) * logcosh_(c / 2.0);
    }
    auto mu = [&](int k) { return (double)mg.m(k, i); };
    auto var = [&](int k) { return (double)mg.v(k, i); };
    auto q1 = [&](int k) -> double { // out1 of aux_posterior!
        switch (lik.kind) {
        case AGPL_LIK_STUDENTT: {
            const double nu = lik.p[0], sg = lik.p[1];
            return (nu / (sg * sg) + second_moment_y(mu(0), var(0), y(0))) / 2.0;
        }
        case AGPL_LIK_LAPLACE:
            return 1.0 / (2.0 * lik.p[0] * sqrt(second_moment_y(mu(0), var(0), y(0))));
        default:
            return sqrt(second_moment(mu(k), var(k)));
        }
    };
    auto q2 = [&](int k) -> double { // out2 of aux_posterior!
        if (lik.kind == AGPL_LIK_POISSON) return lik.p[0] * approx_expected_logistic(-mu(0), q1(0));
        const double den = lik.kind == AGPL_LIK_CATEGORICAL ? (double)L : (lik.cat_const + (double)L);
        return approx_expected_logistic(-mu(k), q1(k)) / den;
    };
    return expected_logtilt_point(lik, y, q1, q2, mu, var) - aux_kl_point(lik, y, q1, q2);
}

struct MargArrays { // marginals latent-major [L][N]
    const float *mu, *var;
    int64_t n;
    __device__ __forceinline__ float m(int k, int64_t i) const { return mu[(int64_t)k * n + i]; }
    __device__ __forceinline__ float v(int k, int64_t i) const { return var[(int64_t)k * n + i]; }
};
struct OutArrays {
    float *gamma, *beta;
    int64_t n;
    __device__ __forceinline__ void put(int k, int64_t i, float g, float b) {
        gamma[(int64_t)k * n + i] = g;
        beta[(int64_t)k * n + i] = b;
    }
};

__global__ __launch_bounds__(kBlock) void agpl_fused_elementwise_kernel(agpl_lik_dev lik, int64_t n, const void *yv,
                                                                        const float *__restrict__ mu,
                                                                        const float *__restrict__ var,
                                                                        float *__restrict__ gamma,
                                                                        float *__restrict__ beta,
                                                                        float *__restrict__ c_out) {
    const MargArrays mg{mu, var, n};
    OutArrays out{gamma, beta, n};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        fused_point(lik, n, i, yv, mg, out, c_out);
}

// The sweep's ONE per-point kernel (image path): the marginal kernel's row-block partial sums -> q(f_i) -> aux_posterior! ->
// expected potential / precision -> the accumulation's gamma | beta records (256 bytes per 32-point step: gamma x 32 |
// beta x 32, zeros beyond N) and max gamma (one atomic per workgroup) -- what marginal_combine_kernel,
// agpl_fused_elementwise_kernel and acc_prep_kernel did in three launches and three round trips through HBM.
struct MargParts {
    const float *resid, *mu0, *qpart, *mpart;
    int64_t n;
    int L, nb2;
    __device__ __forceinline__ float m(int k, int64_t i) const {
        float s = 0.f;
        for (int rb = 0; rb < nb2; ++rb) s += mpart[((int64_t)rb * L + k) * n + i]; // (row blocks in ascending order)
        return mu0 ? s + mu0[(int64_t)k * n + i] : s;
    }
    __device__ __forceinline__ float v(int k, int64_t i) const {
        float q = 0.f;
        for (int rb = 0; rb < nb2; ++rb) q += qpart[((int64_t)rb * L + k) * n + i];
        return resid[i] + q;
    }
};
struct OutRecords {
    float *gamma, *beta; // optional [L][N] copies
    float *gb;
    int64_t n, nrec;     // nrec = records per latent
    unsigned gmax, bad;
    __device__ __forceinline__ void put(int k, int64_t i, float g, float b) {
        if (gamma) gamma[(int64_t)k * n + i] = g;
        if (beta) beta[(int64_t)k * n + i] = b;
        float *rec = gb + ((int64_t)k * nrec + (i >> 5)) * 64 + (i & 31);
        rec[0] = g;
        rec[32] = b;
        const unsigned gbits = __float_as_uint(g), ab = gbits & 0x7FFFFFFFu;
        if (ab >= 0x7F800000u || ((gbits >> 31) && ab != 0u)) bad = max(bad, (unsigned)min((int64_t)0x7FFFFFFE, k * n + i) + 1u);
        else gmax = max(gmax, ab);
    }
};

// ELBO: the instantiation that also sums the ELBO terms (float64 transcendental code: kept out of the plain kernel, whose
// register footprint and 0.13 ms per 1e7 points it would otherwise cost -- 0.47 ms with the branch compiled in, measured)
// (KIND: the likelihood of an ELBO instantiation, so that only its own float64 terms are compiled in; -1: taken from `lik`)
template <bool ELBO, int KIND>
__global__ __launch_bounds__(kBlock) void agpl_fused_point_kernel(agpl_lik_dev lik_arg, int64_t n, int64_t npad, int nb2,
                                                                  const void *yv, const float *__restrict__ resid,
                                                                  const float *__restrict__ mu0,
                                                                  const float *__restrict__ qpart,
                                                                  const float *__restrict__ mpart,
                                                                  float *__restrict__ gamma, float *__restrict__ beta,
                                                                  float *__restrict__ c_out, float *__restrict__ gb,
                                                                  unsigned *__restrict__ scal,
                                                                  unsigned *__restrict__ queues,
                                                                  double *__restrict__ elbo_part) {
    // queues[0..7]: the marginal kernel's item queues; queues[8]: 1 + index of a gamma that is negative or not finite, kept
    // until the update's last kernel forwards it to the host (agpl_pending_resolve reports AGPL_ERR_DOMAIN)
    __shared__ unsigned red[2][kBlock / 64];
    agpl_lik_dev lik = lik_arg;
    if (KIND >= 0) lik.kind = KIND; // (a compile-time constant from here on: the switches over the kind fold)
    const int L = lik.nlatent;
    if (blockIdx.x == 0 && threadIdx.x < 8) queues[threadIdx.x] = 0u; // the marginal kernel's item queues, for its next launch
    const MargParts mg{resid, mu0, qpart, mpart, n, L, nb2};
    OutRecords out{gamma, beta, gb, n, npad / 32, 0u, 0u};
    double eacc = 0.0; // (elbo_part != nullptr) this thread's ELBO terms, points in ascending order
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n) {
            fused_point(lik, n, i, yv, mg, out, c_out);
            if (ELBO) eacc += elbo_point(lik, i, yv, mg);
        } else { // the zero tail of the records
            for (int k = 0; k < L; ++k) {
                float *rec = gb + ((int64_t)k * out.nrec + (i >> 5)) * 64 + (i & 31);
                rec[0] = 0.f;
                rec[32] = 0.f;
            }
        }
    }
    unsigned m = out.gmax, b = out.bad;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m = max(m, (unsigned)__shfl_xor((int)m, o));
        b = max(b, (unsigned)__shfl_xor((int)b, o));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = m;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            m = max(m, red[0][w]);
            b = max(b, red[1][w]);
        }
        if (m) atomicMax(scal, m);
        if (b) {
            atomicMax(scal + 1, b);
            atomicMax(queues + 8, b);
        }
    }
    if (ELBO) { // the ELBO rides the pass: fixed-order tree over the workgroup, one partial per workgroup
        __shared__ double esum[kBlock];
        esum[threadIdx.x] = eacc;
        __syncthreads();
        for (int st = kBlock / 2; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) esum[threadIdx.x] += esum[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) elbo_part[blockIdx.x] = esum[0];
    }
}

// internal (agpl_update.hip): the per-point kernel of the image sweep; scal must be zero (the marginal kernel zeroes it)
int32_t agpl_launch_fused_point(agpl_ctx *ctx, const agpl_lik_dev &ld, int64_t n, int64_t npad, int nb2, const void *y,
                                const float *resid, const float *mu0, const float *qpart, const float *mpart,
                                float *gamma, float *beta, float *c_out, float *gb, unsigned *scal, unsigned *queues,
                                double *elbo_terms_out) {
    int64_t nblk = agpl_cdiv(npad, kBlock);
    if (nblk > 1024) nblk = 1024; // (one atomic per workgroup on the max-gamma word)
    double *part = nullptr;
    if (elbo_terms_out) { // the sum over points of expected_logtilt_i - aux_kldivergence_i rides the pass
        if (ld.kind == AGPL_LIK_CATEGORICAL || ld.kind == AGPL_LIK_HETEROGAUSS)
            AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED,
                      "the ELBO terms are not defined for this likelihood (categorical.jl:165-170: non-bijective link; "
                      "heteroscedastic: not split in the reference)");
        if (!ctx->elbo_part) {
            AGPL_HIP(ctx, hipMalloc((void **)&ctx->elbo_part, sizeof(double) * 1024));
        }
        part = ctx->elbo_part;
    }
#define AGPL_LAUNCH_FUSED(E_, K_)                                                                               \
    agpl_fused_point_kernel<E_, K_><<<(unsigned)nblk, kBlock, 0, ctx->stream>>>(ld, n, npad, nb2, y, resid, mu0, qpart, mpart, \
                                                                               gamma, beta, c_out, gb, scal, queues, part)
    if (!part) AGPL_LAUNCH_FUSED(false, -1);
    else
        switch (ld.kind) {
        case AGPL_LIK_BERNOULLI_LOGISTIC: AGPL_LAUNCH_FUSED(true, AGPL_LIK_BERNOULLI_LOGISTIC); break;
        case AGPL_LIK_NEGBINOMIAL: AGPL_LAUNCH_FUSED(true, AGPL_LIK_NEGBINOMIAL); break;
        case AGPL_LIK_STUDENTT: AGPL_LAUNCH_FUSED(true, AGPL_LIK_STUDENTT); break;
        case AGPL_LIK_CATEGORICAL_BIJ: AGPL_LAUNCH_FUSED(true, AGPL_LIK_CATEGORICAL_BIJ); break;
        case AGPL_LIK_POISSON: AGPL_LAUNCH_FUSED(true, AGPL_LIK_POISSON); break;
        default: AGPL_LAUNCH_FUSED(true, AGPL_LIK_LAPLACE); break;
        }
#undef AGPL_LAUNCH_FUSED
    AGPL_LAUNCH_CHECK(ctx);
    if (part) {
        reduce_final_kernel<<<1, 64, 0, ctx->stream>>>((int)nblk, part, elbo_terms_out);
        AGPL_LAUNCH_CHECK(ctx);
    }
    return AGPL_OK;
}

int32_t agpl_launch_fused_elementwise(agpl_ctx *ctx, const agpl_lik_dev &ld, int64_t n, const void *y,
                                      const float *mu, const float *var, float *gamma, float *beta, float *c_out) {
    agpl_fused_elementwise_kernel<<<grid_for(n), kBlock, 0, ctx->stream>>>(ld, n, y, mu, var, gamma, beta, c_out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}

// The per-point half of the Gibbs point pass: lane `lane` of a wave owns point base + lane, whose noiseless projection(s)
// sit in fS[lane * Lf + l]; f = projection + sqrt(d) eps (+ mu0), then aux_sample! and the sampled potential / precision.
template <int KIND>
__device__ __forceinline__ void gibbs_sample_points(const agpl_lik_dev &lik, int64_t N, int64_t base, int lane, int np,
                                                    int Lf, int Lo, const float *__restrict__ kdiag,
                                                    const float *__restrict__ mu0, const void *yv, uint64_t seed,
                                                    uint64_t i0, uint32_t sweep, PgBlockScratch *scr, double *fS, double *omS,
                                                    int32_t *nnS,
                                                    float *__restrict__ gamma, float *__restrict__ beta,
                                                    double *__restrict__ f_out, double *__restrict__ omega_out,
                                                    int64_t *__restrict__ n_out, uint32_t *__restrict__ nuni_out,
                                                    int *__restrict__ bad) {
    const int64_t i = base + lane;
    const bool valid = lane < np;
    Philox g;
    g.init(seed, i0 + (uint64_t)i, sweep);
    uint32_t nt = 0;
    if (valid) {
        const double kd = (double)kdiag[i];
        const double sd = sqrt(kd > 0.0 ? kd : 0.0); // a float32 Nystrom residual can round below zero
        for (int l = 0; l < Lf; ++l) {
            double f = fS[lane * Lf + l] + sd * g.normal();
            if (mu0) f += (double)mu0[(int64_t)l * N + i];
            fS[lane * Lf + l] = f;
            if (f_out) f_out[i * Lf + l] = f;
        }
    }
    sample_point_wave<KIND, int32_t>(lik, scr, lane, valid, g, i, yv, fS + lane * Lf, omS + lane * Lo, nnS + lane * Lo, nt, bad);
    if (valid) {
        if (nuni_out) nuni_out[i] = g.nuni;
        // auglik_potential / auglik_precision of the draw (same formulas as potential_precision_kernel)
        switch (KIND) {
        case AGPL_LIK_BERNOULLI_LOGISTIC:
            beta[i] = ((const uint8_t *)yv)[i] ? 0.5f : -0.5f;
            gamma[i] = (float)omS[lane];
            break;
        case AGPL_LIK_NEGBINOMIAL:
            beta[i] = (float)(((double)((const int32_t *)yv)[i] - lik.p[0]) / 2.0);
            gamma[i] = (float)omS[lane];
            break;
        case AGPL_LIK_STUDENTT:
            beta[i] = (float)(((const double *)yv)[i] * omS[lane]);
            gamma[i] = (float)omS[lane];
            break;
        case AGPL_LIK_CATEGORICAL:
        case AGPL_LIK_CATEGORICAL_BIJ:
            for (int k = 0; k < Lf; ++k) {
                beta[(int64_t)k * N + i] =
                    (float)(((double)((const uint8_t *)yv)[i * Lf + k] - (double)nnS[lane * Lo + k]) / 2.0);
                gamma[(int64_t)k * N + i] = (float)omS[lane * Lo + k];
            }
            break;
        case AGPL_LIK_POISSON:
            beta[i] = (float)(((double)((const int32_t *)yv)[i] - (double)nnS[lane]) / 2.0);
            gamma[i] = (float)omS[lane];
            break;
        case AGPL_LIK_LAPLACE:
            beta[i] = (float)(2.0 * omS[lane] * ((const double *)yv)[i]);
            gamma[i] = (float)(2.0 * omS[lane]);
            break;
        case AGPL_LIK_HETEROGAUSS: {
            const double il = lik.p[0] * logistic(fS[lane * 2 + 1]);
            beta[i] = (float)(((const double *)yv)[i] * il);
            gamma[i] = (float)il;
            beta[N + i] = (float)((0.5 - (double)nnS[lane]) / 2.0);
            gamma[N + i] = (float)omS[lane];
        } break;
        default:
            break;
        }
        if (omega_out)
            for (int k = 0; k < Lo; ++k) omega_out[i * Lo + k] = omS[lane * Lo + k];
        if (n_out)
            for (int k = 0; k < Lo; ++k) n_out[i * Lo + k] = (int64_t)nnS[lane * Lo + k];
    }
}

// ------------------------------------------------------------------------------------------------
// Gibbs point pass (agpl_gibbs_pass; sparse form of examples/bernoulli/script.jl:81-84):
//   f_il = mu0_il + phi_i' v_l + sqrt(kdiag_i) eps_il      (the draw of f given v under the sparse model)
//   Omega_i ~ aux_full_conditional(lik, y_i, f_i)           (aux_sample!, same per-point Philox stream)
//   beta_i, gamma_i = auglik_potential / auglik_precision   (float32, [L][N], feed agpl_accumulate)
// Two kernels around N * L doubles of scratch.  gibbs_project_kernel streams Phi once: one wave per 64-point chunk,
// 16 lanes per point form phi_i' v in float64 in a fixed order (lane q owns features 4q.., 4q+64..; xor butterfly
// 8,4,2,1 -- the order is part of the contract, the CPU check reproduces it bit for bit).  gibbs_sample_kernel<KIND>
// then gives every lane its own point (one instantiation per likelihood keeps the sampler's register footprint down).
// Fused into one kernel (the first form) the sampler's ~255 VGPRs held the streaming half at one wave per SIMD:
// 6.2 ms at C2 against 4.9 ms for the pair.
// ------------------------------------------------------------------------------------------------
// NV > 0: M = 64 NV and the lane's NV float4 of a row are loaded ONCE and kept in registers for all latents (the generic form,
// NV = 0, re-reads the row per latent: with K = 10 latents that was 10 x the loads of a pass that is otherwise free); the
// summation order per latent is the same, so the projections are bit for bit those of the generic form.
template <int NV>
__global__ __launch_bounds__(256) void gibbs_project_kernel(int64_t N, int M, int Lf, const float *__restrict__ Phi,
                                                            const double *__restrict__ v, double *__restrict__ proj) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *v_s = sh; // [Lf][M]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = threadIdx.x; a < Lf * M; a += blockDim.x) v_s[a] = v[a];
    __syncthreads();
    const int q = lane & 15, grp = lane >> 4;
    const int64_t nchunks = (N + 63) >> 6; // (a wave per 64-point chunk; no barrier in the loop: waves run out independently)
    for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave; chunk < nchunks; chunk += (int64_t)gridDim.x * 4) {
        const int64_t base = chunk << 6;
        const int np = (int)((N - base) < 64 ? (N - base) : 64); // >= 1
        for (int r = 0; r < 16; ++r) {
            int p = 4 * r + grp;
            const int pc = p < np ? p : np - 1; // clamp: keeps every lane in the shuffles
            const float *row = Phi + (base + pc) * (int64_t)M;
            double xd[NV > 0 ? 4 * NV : 1]; // (converted once, not per latent)
            if (NV > 0) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const float4 x = *reinterpret_cast<const float4 *>(row + (q << 2) + 64 * i);
                    xd[4 * i] = (double)x.x, xd[4 * i + 1] = (double)x.y, xd[4 * i + 2] = (double)x.z, xd[4 * i + 3] = (double)x.w;
                }
            }
            int l = 0;
            if (NV > 0) { // four latents at a time: their accumulation chains are independent and overlap (each sum in its own order)
                for (; l + 4 <= Lf; l += 4) {
                    const double *v0 = v_s + (size_t)l * M, *v1 = v0 + M, *v2 = v1 + M, *v3 = v2 + M;
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int a = (q << 2) + 64 * i;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a0 += xd[4 * i + e] * v0[a + e];
                            a1 += xd[4 * i + e] * v1[a + e];
                            a2 += xd[4 * i + e] * v2[a + e];
                            a3 += xd[4 * i + e] * v3[a + e];
                        }
                    }
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) {
                        a0 += __shfl_xor(a0, off);
                        a1 += __shfl_xor(a1, off);
                        a2 += __shfl_xor(a2, off);
                        a3 += __shfl_xor(a3, off);
                    }
                    if (q == 0 && p < np) {
                        double *dst = proj + (base + p) * Lf + l;
                        dst[0] = a0, dst[1] = a1, dst[2] = a2, dst[3] = a3;
                    }
                }
            }
            for (; l < Lf; ++l) {
                const double *vl = v_s + (size_t)l * M;
                double acc = 0.0;
                if (NV > 0) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int a = (q << 2) + 64 * i;
                        acc += xd[4 * i] * vl[a];
                        acc += xd[4 * i + 1] * vl[a + 1];
                        acc += xd[4 * i + 2] * vl[a + 2];
                        acc += xd[4 * i + 3] * vl[a + 3];
                    }
                } else {
#pragma unroll 8
                    for (int a = q << 2; a < M; a += 64) {
                        const float4 x = *reinterpret_cast<const float4 *>(row + a);
                        acc += (double)x.x * vl[a];
                        acc += (double)x.y * vl[a + 1];
                        acc += (double)x.z * vl[a + 2];
                        acc += (double)x.w * vl[a + 3];
                    }
                }
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
                if (q == 0 && p < np) proj[(base + p) * Lf + l] = acc;
            }
        }
    }
}

// the Gibbs point pass: the categorical kinds at two waves (C4 Gibbs sweep 4.19 / 3.90 / 3.84 ms at 4 / 3 / 2, round 5; its per-wave
// LDS scratch sets the occupancy at K = 10 anyway), the other PG kinds at four
constexpr int gibbs_wps(int kind) {
#ifdef AGPL_GIBBS_WPS_CAT // measurement builds
    if (kind == AGPL_LIK_CATEGORICAL || kind == AGPL_LIK_CATEGORICAL_BIJ) return AGPL_GIBBS_WPS_CAT;
#endif
    return (kind == AGPL_LIK_STUDENTT || kind == AGPL_LIK_LAPLACE) ? 1 : (kind == AGPL_LIK_CATEGORICAL || kind == AGPL_LIK_CATEGORICAL_BIJ) ? 2 : 4;
}
template <int KIND>
__global__ __launch_bounds__(256, gibbs_wps(KIND)) void gibbs_sample_kernel(
    agpl_lik_dev lik, int64_t N, const double *__restrict__ proj, const float *__restrict__ kdiag,
    const float *__restrict__ mu0, const void *yv, uint64_t seed, uint64_t i0, uint32_t sweep,
    float *__restrict__ gamma, float *__restrict__ beta, double *__restrict__ f_out, double *__restrict__ omega_out, int64_t *__restrict__ n_out,
    uint32_t *__restrict__ nuni_out, int *__restrict__ bad) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    __shared__ PgBlockScratch scratch;
    pg_scratch_init(&scratch, lik, nuni_out != nullptr);
    const int Lf = lik.nlatent;
    const int Lo = KIND == AGPL_LIK_HETEROGAUSS ? 1 : lik.nlatent;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per wave: f [64][Lf] and omega [64][Lo] in float64, the latent counts n [64][Lo] in int32 (a Poisson count that does not
    // fit 31 bits would need a rate no float64 logistic produces): 12.8 KB per wave at K = 10 -- with 64-bit counts the kernel's
    // 84 KB per workgroup pinned it to one workgroup per CU
    double *fS = sh + (size_t)wave * (64 * (Lf + Lo) + 32 * Lo); // [64][Lf]
    double *omS = fS + 64 * Lf;                                   // [64][Lo]
    int32_t *nnS = reinterpret_cast<int32_t *>(omS + 64 * Lo);    // [64][Lo]
    const int64_t nblocks = (N + 255) >> 8; // (uniform trip count over the workgroup: the sampler has barriers)
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t base = (blk * 4 + wave) << 6;
        const int np = (N - base) < 0 ? 0 : (int)((N - base) < 64 ? (N - base) : 64);
        if (lane < np)
            for (int l = 0; l < Lf; ++l) fS[lane * Lf + l] = proj[(base + lane) * Lf + l];
        gibbs_sample_points<KIND>(lik, N, base, lane, np, Lf, Lo, kdiag, mu0, yv, seed, i0, sweep, &scratch, fS, omS, nnS, gamma,
                                  beta,
                                  f_out, omega_out, n_out, nuni_out, bad);
    }
}

// The same projection from the accumulate image (agpl_syrk.hip: 4 KB blocks [point slice of 16][feature block of 128][hi | lo] =
// [plane 2 of 8 points][feature 128][8 halves], the image holding 2^e Phi), so that a plan's Gibbs pass reads nothing but its
// images: x = (hi + lo) 2^-e is the feature the accumulation multiplies (Phi to 2^-22 relative).  One wave per 16-point slice:
// lanes 0-31 take plane 0 (points 0..7), lanes 32-63 plane 1; a lane walks the features (lane & 31) + 32 j, j = 0..3, of every
// feature block in ascending order and keeps the partial sums of its 8 points; the 32 lanes of a plane are then added by an
// exchange tree (a lane hands half of its points to its partner at distances 16, 8, 4 -- 4, 2, 1 values left -- and the last two
// steps add the remaining value): fixed order, bitwise reproducible.  float64 accumulation; several latents share the loads.
typedef _Float16 gp_h8 __attribute__((ext_vector_type(8)));
template <int LB> // latents per pass over the slice
__device__ __forceinline__ void gibbs_project_slice(int64_t N, int M, int Lf, int l0, const gp_h8 *__restrict__ blocks, int64_t ps,
                                                    double unscale, const double *__restrict__ v_s, double *__restrict__ proj,
                                                    int lane) {
    const int nb = M / 128;
    const int plane = lane >> 5, fl = lane & 31;
    double acc[LB][8];
#pragma unroll
    for (int l = 0; l < LB; ++l)
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[l][p] = 0.0;
    for (int fb = 0; fb < nb; ++fb) {
        const gp_h8 *bh = blocks + ((ps * nb + fb) * 2) * 256 + plane * 128 + fl; // hi block; the lo block follows (+ 256 granules)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const gp_h8 h = bh[32 * j], lo = bh[256 + 32 * j];
            const int f = fb * 128 + fl + 32 * j;
            double x[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) x[p] = (double)((float)h[p] + (float)lo[p]); // exact: hi and lo do not overlap
#pragma unroll
            for (int l = 0; l < LB; ++l) {
                const double vv = v_s[(size_t)(l0 + l) * M + f];
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[l][p] += x[p] * vv;
            }
        }
    }
#pragma unroll
    for (int l = 0; l < LB; ++l) {
        // exchange tree over the 32 lanes of the plane: after distance d a lane keeps the points whose bit (d >> 2 ... ) matches it
        double a4[4], a2[2], a1;
        const bool up16 = fl & 16, up8 = fl & 8, up4 = fl & 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) { // keep points 4..7 in the upper half of the 16-pairs, 0..3 in the lower
            const double mine = up16 ? acc[l][4 + p] : acc[l][p], give = up16 ? acc[l][p] : acc[l][4 + p];
            a4[p] = mine + __shfl_xor(give, 16);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const double mine = up8 ? a4[2 + p] : a4[p], give = up8 ? a4[p] : a4[2 + p];
            a2[p] = mine + __shfl_xor(give, 8);
        }
        {
            const double mine = up4 ? a2[1] : a2[0], give = up4 ? a2[0] : a2[1];
            a1 = mine + __shfl_xor(give, 4);
        }
        a1 += __shfl_xor(a1, 2);
        a1 += __shfl_xor(a1, 1);
        // the lane with (fl & 3) == 0 holds point 4 [fl & 16] + 2 [fl & 8] + [fl & 4] of its plane
        if ((fl & 3) == 0) {
            const int pt = plane * 8 + (up16 ? 4 : 0) + (up8 ? 2 : 0) + (up4 ? 1 : 0);
            const int64_t n = ps * 16 + pt;
            if (n < N) proj[n * Lf + l0 + l] = a1 * unscale;
        }
    }
}
__global__ __launch_bounds__(256) void gibbs_project_image_kernel(int64_t N, int M, int Lf, const unsigned char *__restrict__ image,
                                                                  const double *__restrict__ v, double *__restrict__ proj) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *v_s = sh; // [Lf][M]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int a = threadIdx.x; a < Lf * M; a += blockDim.x) v_s[a] = v[a];
    __syncthreads();
    const int e = reinterpret_cast<const int32_t *>(image)[1]; // header word 1: the image holds 2^e Phi
    const double unscale = __hiloint2double((1023 - e) << 20, 0);
    const gp_h8 *blocks = reinterpret_cast<const gp_h8 *>(image + 256);
    const int64_t nslices = (N + 15) >> 4;
    for (int64_t ps = (int64_t)blockIdx.x * 4 + wave; ps < nslices; ps += (int64_t)gridDim.x * 4) {
        int l = 0;
        for (; l + 4 <= Lf; l += 4) gibbs_project_slice<4>(N, M, Lf, l, blocks, ps, unscale, v_s, proj, lane);
        for (; l + 2 <= Lf; l += 2) gibbs_project_slice<2>(N, M, Lf, l, blocks, ps, unscale, v_s, proj, lane);
        for (; l < Lf; ++l) gibbs_project_slice<1>(N, M, Lf, l, blocks, ps, unscale, v_s, proj, lane);
    }
}

// Phi == nullptr: the projection reads `image` (the accumulate image of the same features) instead
int32_t agpl_launch_gibbs_project_sample(agpl_ctx *ctx, const agpl_lik_dev &ld, int64_t N, int M, const float *Phi,
                                         const void *image, const float *kdiag, const float *mu0, const void *y, const double *v,
                                         uint32_t sweep, float *gamma, float *beta, double *f_out,
                                         double *omega_out, int64_t *n_out, uint32_t *nuni_out, int *bad,
                                         double *proj_work /* N * L doubles of scratch */) {
    const int Lf = ld.nlatent;
    const int Lo = ld.kind == AGPL_LIK_HETEROGAUSS ? 1 : ld.nlatent;
    if (sizeof(double) * (size_t)Lf * M > 160 * 1024)
        AGPL_FAIL(ctx, AGPL_ERR_UNSUPPORTED, "Gibbs pass: L * M = %d x %d does not fit the LDS working set", Lf, M);
    int64_t nb = agpl_cdiv(agpl_cdiv(N, 64), 4);
    if (nb > 256 * 8) nb = 256 * 8;
    if (!proj_work) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "Gibbs point pass: no projection scratch");
    int32_t rc = AGPL_OK;
    if (ld.kind == AGPL_LIK_BERNOULLI_LOGISTIC) rc = agpl_pg_retry_reserve(ctx, N < kPg1MaxLaunch ? N : kPg1MaxLaunch);
    if (rc) return rc;
    rc = agpl_timing_begin(ctx, 2);
    if (rc) return rc;
    {
        const size_t lds_p = sizeof(double) * (size_t)Lf * M, lds_s = sizeof(double) * 4 * (size_t)(64 * (Lf + Lo) + 32 * Lo);
        int64_t nbp = agpl_cdiv(agpl_cdiv(N, 64), 4);
        if (nbp > 256 * 16) nbp = 256 * 16;
#define AGPL_LAUNCH_PROJECT(NV_)                                                                                    \
    do {                                                                                                            \
        AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&gibbs_project_kernel<NV_>),               \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));                 \
        gibbs_project_kernel<NV_><<<(unsigned)nbp, 256, lds_p, ctx->stream>>>(N, M, Lf, Phi, v, proj_work);         \
    } while (0)
        if (!Phi) {
            if (!image) AGPL_FAIL(ctx, AGPL_ERR_INVALID_ARGUMENT, "Gibbs point pass: neither features nor their image");
            AGPL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&gibbs_project_image_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
            int64_t nbi = agpl_cdiv(agpl_cdiv(N, 16), 4);
            if (nbi > 256 * 16) nbi = 256 * 16;
            gibbs_project_image_kernel<<<(unsigned)nbi, 256, lds_p, ctx->stream>>>(N, M, Lf, (const unsigned char *)image, v,
                                                                                proj_work);
        } else if (Lf > 1 && M == 256) AGPL_LAUNCH_PROJECT(4); // several latents: the row stays in registers (M = 64 NV)
        else if (Lf > 1 && M == 512) AGPL_LAUNCH_PROJECT(8);
        else if (Lf > 1 && M == 1024) AGPL_LAUNCH_PROJECT(16);
        else AGPL_LAUNCH_PROJECT(0);
#undef AGPL_LAUNCH_PROJECT
        AGPL_LAUNCH_CHECK(ctx);
#define AGPL_LAUNCH_GIBBS_S(K)                                                                                        \
    case K:                                                                                                           \
        gibbs_sample_kernel<K><<<(unsigned)nb, 256, lds_s, ctx->stream>>>(ld, N, proj_work, kdiag, mu0, y, ctx->seed, \
                                                                           (uint64_t)ctx->point_offset, sweep, gamma, \
                                                                           beta, f_out, omega_out, n_out, nuni_out,  \
                                                                           bad);                                     \
        break;
        switch (ld.kind) {
        case AGPL_LIK_BERNOULLI_LOGISTIC: { // one PG(1, |f_i|) draw per point: the kernel with the long trial queue (f over proj_work)
            launch_pg1<true>(ctx, N, nullptr, omega_out, sweep, nuni_out, nullptr, proj_work, kdiag, mu0, (const uint8_t *)y,
                             gamma, beta, f_out);
        } break;
            AGPL_LAUNCH_GIBBS_S(AGPL_LIK_NEGBINOMIAL)
            AGPL_LAUNCH_GIBBS_S(AGPL_LIK_STUDENTT)
            AGPL_LAUNCH_GIBBS_S(AGPL_LIK_CATEGORICAL)
            AGPL_LAUNCH_GIBBS_S(AGPL_LIK_CATEGORICAL_BIJ)
            AGPL_LAUNCH_GIBBS_S(AGPL_LIK_POISSON)
            AGPL_LAUNCH_GIBBS_S(AGPL_LIK_LAPLACE)
            AGPL_LAUNCH_GIBBS_S(AGPL_LIK_HETEROGAUSS)
        default:
            break;
        }
#undef AGPL_LAUNCH_GIBBS_S
        AGPL_LAUNCH_CHECK(ctx);
        return agpl_timing_end(ctx, 2);
    }
}

// z_a = standard normal from the stream (seed, a, sweep): the randn!(...) of the Gaussian conditional draw
__global__ void randn_kernel(int64_t n, uint64_t seed, uint32_t sweep, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        Philox g;
        g.init(seed, (uint64_t)i, sweep);
        out[i] = g.normal();
    }
}
int32_t agpl_launch_randn(agpl_ctx *ctx, int64_t n, uint32_t sweep, double *out) {
    randn_kernel<<<grid_for(n), kBlock, 0, ctx->stream>>>(n, ctx->seed, sweep, out);
    AGPL_LAUNCH_CHECK(ctx);
    return AGPL_OK;
}
