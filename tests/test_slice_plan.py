"""How one accumulation cuts N into slices (csrc/agpl_slices.h, round 6: the last round's worth of slices four times finer).  The header
has no HIP dependency, so it is compiled here with g++ and checked on the host: for thousands of (N, M, L) the slices tile [0, N)
exactly, in order, every boundary on a 32-point stage, none empty, none longer than the float32 chains allow (16 384 points), and the
plan is a function of (N, M, L) only.  (The first version of the plan lost the last slices of any launch WITHOUT a fine tail -- the C4
shape -- which only a full-size property test on the GPU caught.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r"""
#include <stdio.h>
#include <stdlib.h>
#include "agpl_slices.h"
static int check(int64_t N, int M, int L) {
    const agpl_slices p = agpl_slice_plan(N, M, L), q = agpl_slice_plan(N, M, L);
    if (p.chunk != q.chunk || p.ns != q.ns || p.nbig != q.nbig || p.small != q.small) return 1;
    if (p.chunk % 32 || p.small % 32 || p.chunk > 16384 || p.small <= 0 || p.nbig < 0 || p.ns < 1 || p.nbig > p.ns) return 2;
    int64_t at = 0;
    for (int s = 0; s < p.ns; ++s) {
        int64_t b, e;
        agpl_slice_range(s, p.chunk, p.nbig, p.small, N, b, e);
        if (b != at || e <= b || e > N || b % 32 || e - b > 16384) return 3;
        at = e;
    }
    return at == N ? 0 : 4;
}
int main() {
    const int Ms[] = {128, 256, 384, 512, 768, 1024, 1280, 2048};
    const int Ls[] = {1, 2, 3, 10};
    uint64_t x = 88172645463325252ull;
    long cases = 0;
    for (int mi = 0; mi < 8; ++mi)
        for (int li = 0; li < 4; ++li) {
            const int M = Ms[mi], L = Ls[li];
            // around every granularity, the configured sizes, and random N
            const int64_t fixed[] = {1, 31, 32, 33, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 8193, 16384, 16385, 352256, 352257, 1000000,
                                     1250000, 10000000, 10000001, 123456789};
            for (size_t i = 0; i < sizeof(fixed) / sizeof(fixed[0]); ++i, ++cases) {
                const int rc = check(fixed[i], M, L);
                if (rc) { printf("FAIL rc=%d N=%lld M=%d L=%d\n", rc, (long long)fixed[i], M, L); return 1; }
            }
            for (int r = 0; r < 400; ++r, ++cases) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                const int64_t N = 1 + (int64_t)(x % (r % 4 == 0 ? 20000000ull : 600000ull));
                const int rc = check(N, M, L);
                if (rc) { printf("FAIL rc=%d N=%lld M=%d L=%d\n", rc, (long long)N, M, L); return 1; }
            }
        }
    // the shapes the round-6 records quote
    const agpl_slices c2 = agpl_slice_plan(10000000, 512, 1), c4 = agpl_slice_plan(1000000, 256, 10), n8 = agpl_slice_plan(1250000, 512, 1);
    printf("OK %ld cases; C2 chunk %d nbig %d ns %d small %d; C4 chunk %d nbig %d ns %d; N/8 nbig %d ns %d\n", cases, c2.chunk, c2.nbig, c2.ns,
           c2.small, c4.chunk, c4.nbig, c4.ns, n8.nbig, n8.ns);
    return 0;
}
"""


def test_slices_tile_the_points(tmp_path):
    src = tmp_path / "slices.cpp"
    src.write_text(SRC)
    exe = tmp_path / "slices"
    inc = os.path.join(ROOT, "augmentedgplikelihoods.jl_amd", "csrc")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("OK"), out.stdout
    # C4 (ten latents, M = 256, 4.8 rounds): no fine tail, 123 slices of 8192 with a ragged last one; C2: 2442 - 86 big + the rest fine
    assert "C4 chunk 8192 nbig 123 ns 123" in out.stdout, out.stdout
    assert "C2 chunk 4096 nbig 2356 ns" in out.stdout, out.stdout
