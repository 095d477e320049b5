// agpl_slices.h -- how one accumulation cuts its N points into slices (one float32 slab per slice and tile).  No HIP dependency:
// tests/test_slice_plan.py compiles this header with g++ and checks that the slices tile [0, N) for thousands of (N, M, L).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define AGPL_HD __host__ __device__
#else
#define AGPL_HD
#endif

// Points per accumulation slice (one workgroup's float32 accumulation run; one slab per slice and tile).  4096, and 8192 where the
// feature matrix is a single 256-tile wide: there the slab write + reduction is a larger share of a slice (C4-shape accumulate
// 2.92 against 3.27 ms on one box); at M >= 512 the longer slice measured slower (round 3).
AGPL_HD constexpr int agpl_chunk_points(int M) { return M <= 256 ? 8192 : 4096; }
// Points per accumulation slice (float32 sums within a slice, float64 across slices): the figure above, doubled while the launch
// would still have >= 32 workgroups per CU -- N = 1e7 at M = 1024: 2442 slices of 36 x 64 KB are 5.8 GB of slabs written and read
// again per sweep (the reduction alone 0.98-1.05 ms); 1221 slices (47 workgroups per CU) halve that.  A function of (N, M, L) only:
// results repeat.
#ifndef AGPL_SLICE_MIN_WG
#define AGPL_SLICE_MIN_WG 32 // workgroups per CU the launch must keep after a doubling.  (Measured with 8: C2 then runs 8192-point slices at 14
                             // workgroups per CU -- the accumulation kernel loses 0.24-0.34 ms to its coarser tail, the reduction gains 0.14.)
#endif
inline int agpl_slice_points(int64_t N, int M, int L) {
    int chunk = agpl_chunk_points(M);
    const int64_t nb2 = (M + 255) / 256, pairs = nb2 * (nb2 + 1) / 2;
    while (chunk < 16384 && (int64_t)L * pairs * ((N + 2 * chunk - 1) / (2 * chunk)) >= AGPL_SLICE_MIN_WG * 256) chunk *= 2;
    return chunk;
}
// The slices of one accumulation (round 6): `nbig` slices of `chunk` points, then the rest of the points in slices of chunk / 4.
// A launch of equal workgroups of 0.2-0.45 ms each ends with a tail of about one of them on a mostly idle device -- measured as a
// fixed 0.21 ms per launch at M = 512 whatever N is (kernel time = 0.21 + 0.217 ms x rounds of 256 workgroups:
// profiles/NOTES_r06.md), which is 3 % of the launch at N = 1e7 and 23 % of it at a rank's N / 8.  So the last round's worth of
// workgroups (256 / (L x 256-tile pairs) slices) is cut four times finer; a slab is written per slice and tile whatever its length, so
// the finer tail adds about one round of slabs.  A function of (N, M, L) only: results repeat.
struct agpl_slices {
    int chunk, small; // points per big / small slice (multiples of 32)
    int nbig, ns;     // big slices; slices in all
};
#ifndef AGPL_SLICE_TAIL_DIV
#define AGPL_SLICE_TAIL_DIV 4 // (1: no finer tail -- the slices of rounds 1-5)
#endif
inline agpl_slices agpl_slice_plan(int64_t N, int M, int L) {
    agpl_slices o;
    o.chunk = agpl_slice_points(N, M, L);
    o.small = o.chunk / AGPL_SLICE_TAIL_DIV;
    const int64_t nfull = (N + o.chunk - 1) / o.chunk;
    const int64_t nb2 = (M + 255) / 256, wg_per_slice = (int64_t)L * nb2 * (nb2 + 1) / 2;
    int64_t tail = (256 + wg_per_slice - 1) / wg_per_slice; // big slices that make one round of workgroups
    // measured (profiles/NOTES_r06.md): worth 0.05-0.11 ms per launch at M >= 512 and for launches of up to two rounds at any M;
    // a longer launch of diagonal tiles only (M = 256: C4, ten latents, 4.8 rounds) ran 0.08 ms slower with it
    if (AGPL_SLICE_TAIL_DIV == 1 || (M < 512 && nfull * wg_per_slice > 512)) tail = 0;
    o.nbig = (int)(nfull > tail ? nfull - tail : 0);
    const int64_t rest = N - (int64_t)o.nbig * o.chunk; // (<= 0 without a fine tail: the last big slice is the ragged one)
    o.ns = o.nbig + (rest > 0 ? (int)((rest + o.small - 1) / o.small) : 0);
    return o;
}
// points [nbeg, nend) of slice s
AGPL_HD inline void agpl_slice_range(int s, int chunk, int nbig, int small, int64_t N, int64_t &nbeg, int64_t &nend) {
    if (s < nbig) {
        nbeg = (int64_t)s * chunk;
        nend = nbeg + chunk;
    } else {
        nbeg = (int64_t)nbig * chunk + (int64_t)(s - nbig) * small;
        nend = nbeg + small;
    }
    if (nend > N) nend = N;
}

