"""The host-built decomposition that syrk_strip_kernel executes (agpl_mfma.hip syrk_strip_plan), checked on the CPU
through the library's test hook: for every number of 128-row blocks the plan must cover the lower triangle of
G = Phi Diag(gamma) Phi' (docs/src/index.md:154-163, `kappa Diag(r) kappa'`) exactly once per slice, within the
kernel's limits (<= 16 sub-tiles and <= 4 staged panel instances per workgroup), and produce g = Phi beta once per
(panel, slice)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPE_WORDS, SUPER = 16 + 16 * 8, 4


def plan(nb, max_sub=16):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    lib_path = os.path.join(ROOT, "augmentedgplikelihoods.jl_amd", "libagpl.so")
    if not os.path.exists(lib_path):
        g.build()
    import torch  # noqa: F401  (the library's NEEDED entries resolve against torch's bundled ROCm runtime)

    lib = C.CDLL(lib_path)
    buf = (C.c_int32 * 200000)()
    nt, ne = C.c_int32(), C.c_int32()
    n = lib.agpl_debug_strip_plan(C.c_int32(nb | ((8 << 16) if max_sub == 8 else 0)), buf, C.c_int32(len(buf)), C.byref(nt), C.byref(ne))
    assert n > 0
    w = np.frombuffer(buf, dtype=np.int32, count=n).copy()
    types = w[: nt.value * TYPE_WORDS].reshape(nt.value, TYPE_WORDS)
    ents = w[nt.value * TYPE_WORDS:].reshape(ne.value, 2)
    return types, ents


@pytest.mark.parametrize("max_sub", [16, 8])
@pytest.mark.parametrize("nb", list(range(1, 11)) + [16])
def test_strip_plan_covers_the_lower_triangle_exactly_once(nb, max_sub):
    types, ents = plan(nb, max_sub)
    cover = {}   # (slice, unit, wr, wc) -> count
    gcount = {}  # (slice, panel) -> count
    stagings = 0
    for t, k0 in ents:
        ty = types[t]
        r, npan, nsub = ty[0], ty[1], ty[14]
        assert 1 <= npan <= 4 and 1 <= nsub <= max_sub and r in (1, 2, 4) and k0 % r == 0 and k0 + r <= SUPER
        pan, psl, gfl = ty[2:6], ty[6:10], ty[10:14]
        assert all(0 <= pan[i] < nb and 0 <= psl[i] < r for i in range(npan))
        assert len({(pan[i], psl[i]) for i in range(npan)}) == npan  # no instance staged twice
        stagings += npan
        for i in range(npan):
            if gfl[i]:
                gcount[(k0 + psl[i], pan[i])] = gcount.get((k0 + psl[i], pan[i]), 0) + 1
        for s in range(nsub):
            act, qa, ra, qb, rb, unit, wr, wc = ty[16 + 8 * s: 24 + 8 * s]
            assert act == 1 and qa < npan and qb < npan and psl[qa] == psl[qb]
            bi, bj = pan[qa], pan[qb]
            assert bj <= bi and unit == bi * (bi + 1) // 2 + bj and (ra, rb) == (wr, wc)
            assert not (bi == bj and wr < wc)  # the sub-tile above the diagonal of a diagonal unit is never computed
            key = (k0 + psl[qa], unit, wr, wc)
            cover[key] = cover.get(key, 0) + 1
    for sl in range(SUPER):
        for bi in range(nb):
            for bj in range(bi + 1):
                for wr in range(2):
                    for wc in range(2):
                        need = 0 if (bi == bj and wr < wc) else 1
                        assert cover.get((sl, bi * (bi + 1) // 2 + bj, wr, wc), 0) == need, (nb, sl, bi, bj, wr, wc)
            assert gcount.get((sl, bi), 0) == 1, (nb, sl, bi)
    assert sum(cover.values()) == SUPER * (4 * nb * (nb + 1) // 2 - nb)
    # what the form is for: fewer panel stagings per slice than one 128 x 128 tile per workgroup (nb diagonal tiles
    # stage one panel, nb (nb - 1) / 2 off-diagonal tiles two)
    if nb >= 4:
        assert stagings / SUPER <= (0.7 if max_sub == 16 else 0.8) * nb * nb


def test_strip_plan_m512_is_the_documented_decomposition():
    types, ents = plan(4)
    assert len(types) == 3 and len(ents) == 4 + 4 + 2  # T0, T1 per slice; T2 per two slices
    by_panels = sorted((tuple(int(p) for p in t[2:2 + t[1]]), int(t[0]), int(t[14])) for t in types)
    assert by_panels == [((1, 0, 1, 0), 2, 14), ((2, 1, 0), 1, 14), ((3, 2, 1, 0), 1, 15)]


def test_strip_plan_m512_eight_wave_form_folds_the_diagonal_tiles():
    """The 8-wave form (<= 8 sub-tiles): every diagonal unit shares a workgroup with an off-diagonal unit of one of its
    panels, so no workgroup stages a panel for a diagonal tile alone: 11 stagings per slice instead of 16."""
    types, ents = plan(4, 8)
    per_slice = sum(types[t][1] for t, _ in ents) / SUPER
    assert per_slice == 11
    for ty in types:
        units = {int(ty[16 + 8 * s + 5]) for s in range(ty[14])}
        diag = {u for u in units if u in (0, 2, 5, 9)}
        assert not diag or len(units) > len(diag)  # a diagonal unit never sits alone
