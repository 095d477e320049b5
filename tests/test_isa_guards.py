"""Static guards on the generated gfx950 code of the accumulation kernel (agpl_syrk.hip, syrk_strip_kernel).

Its B granules are fetched by inline-asm `global_load_dwordx4`: the compiler does not see those loads in the memory queue
and believes their destination registers hold the data at once (that is the point: beside an LDS-DMA in flight it would
otherwise drain the whole queue at their first use).  What makes that safe is a property of the generated code, so it is
checked on the generated code (hipcc cross-compiles here, no GPU needed):
  * between such a load and the next `s_waitcnt vmcnt`, no instruction reads or writes its destination registers (a register
    copy placed there -- the phi of a conditional load -- would copy stale data: exactly the failure seen during bring-up);
  * no compiler-generated scratch (spill) traffic inside the step loops (a reload's own wait is computed without the asm loads);
  * the loops wait with a counted `vmcnt(5)`, i.e. the DMA pieces of the step stay in flight."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "augmentedgplikelihoods.jl_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


KERNELS = ("syrk_strip_kernel",)


@pytest.fixture(scope="module")
def syrk_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("isa")
    flags = re.search(r"^COMMON\s*:=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), flags=re.M).group(1)
    flags = flags.replace("$(ARCH)", "gfx950").split()
    subprocess.check_call([HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(CSRC, "agpl_syrk.hip"), "-o", "syrk.s"],
                          cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(d, "syrk.s")).read().splitlines()
    out = {}
    for k in KERNELS:
        start = next(i for i, ln in enumerate(asm) if re.match(r"^_Z\w*" + k + r"\w*:", ln))
        end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
        out[k] = asm[start:end + 1]  # raw lines: the block labels carry the compiler's loop annotations as comments
    return out


def _code(lines):
    return [ln.split(";")[0].rstrip() for ln in lines]


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _operands(line):
    toks = re.split(r"[,\s]+", line.strip())
    return toks[0], [t for t in toks[1:] if t]


def _step_loops(raw):
    """Innermost loops that hold a whole step of MFMAs: the step loops of the off-diagonal and of the diagonal body (96 MFMAs
    each).  A loop = the blocks the compiler annotates
    with its header (`; =>This Inner Loop Header` on the header, `; in Loop: Header=BBx_y Depth=1` on the others); the blocks
    of a loop need not be contiguous or in program order."""
    loops, cur = {}, None
    i = 0
    while i < len(raw):
        ln = raw[i]
        m = re.match(r"^\.L(BB\d+_\d+):(.*)$", ln)
        if m:
            note = m.group(2)
            while i + 1 < len(raw) and re.match(r"^\s*;", raw[i + 1]):  # the annotation continues on comment-only lines
                i += 1
                note += raw[i]
            if re.search(r"This (Inner )?Loop Header", note):
                cur = m.group(1)
            else:
                h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=\d+", note)
                cur = h.group(1) if h else None
        elif re.match(r"^\.L\w+:", ln):
            cur = None
        elif cur is not None:
            loops.setdefault(cur, []).append(ln.split(";")[0].rstrip())
        i += 1
    out = []
    for seg in loops.values():
        n = sum("v_mfma_f32_16x16x32_f16" in x for x in seg)
        if 90 <= n <= 110:
            out.append(seg)
    return out


@pytest.mark.parametrize("kernel", KERNELS)
def test_asm_load_destinations_are_untouched_until_the_wait(syrk_isa, kernel):
    isa = _code(syrk_isa[kernel])
    loads = [i for i, ln in enumerate(isa) if re.match(r"\s*global_load_dwordx4\s", ln)]
    assert len(loads) >= 8  # prologue + step loop, four each (off-diagonal body)
    for i in loads:
        dst = _regs(_operands(isa[i])[1][0])
        assert len(dst) == 4
        j = i + 1
        # to the end of the basic block (a label, a branch, the barrier or the wait itself): exact inside a block, and the
        # failure seen was a block of v_mov copies directly behind the loads
        while not re.search(r"s_waitcnt.*vmcnt|^\s*s_c?branch|^\s*s_barrier|^\.?\w+:\s*$|s_endpgm", isa[j]):
            if isa[j].strip():
                op, args = _operands(isa[j])
                if not op.startswith("global_load_dwordx4"):
                    touched = set().union(*[_regs(a) for a in args]) if args else set()
                    assert not (touched & dst), f"line {j}: `{isa[j].strip()}` touches an in-flight load destination"
            j += 1
        assert j > i + 1


@pytest.mark.parametrize("kernel", KERNELS)
def test_no_scratch_traffic_inside_the_step_loops(syrk_isa, kernel):
    loops = _step_loops(syrk_isa[kernel])
    assert len(loops) >= 2  # the off-diagonal and the diagonal body
    for seg in loops:
        assert not [ln for ln in seg if "scratch_" in ln], "spill traffic inside a step loop"
        assert any("global_load_lds_dwordx4" in ln for ln in seg)


@pytest.mark.parametrize("kernel", KERNELS)
def test_the_step_loops_wait_with_a_counted_vmcnt(syrk_isa, kernel):
    for seg in _step_loops(syrk_isa[kernel]):
        waits = [ln.strip() for ln in seg if re.search(r"s_waitcnt.*vmcnt", ln)]
        assert "s_waitcnt vmcnt(5)" in waits  # the step's five DMA pieces stay in flight


@pytest.fixture(scope="module")
def ops_asm(tmp_path_factory):
    """agpl_ops.hip compiled to gfx950 assembly with the flags the Makefile gives that file (COMMON, NOFMA and its own EXTRA +=)."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("ops_isa")
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^COMMON\s*:=\s*(.*)$", mk, flags=re.M).group(1).replace("$(ARCH)", "gfx950").split()
    flags += re.search(r"^NOFMA\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    for extra in re.findall(r"^agpl_ops\.o:\s*EXTRA\s*\+=\s*(.*)$", mk, flags=re.M):
        if "AGPL_PG_TRACE" not in extra:  # (the diagnostic build's flag sits behind an ifdef)
            flags += extra.split()
    assert "-disable-machine-licm" in flags  # (the flag the scratch guard below depends on)
    subprocess.check_call([HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(CSRC, "agpl_ops.hip"), "-o", "ops.s"],
                          cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(os.path.join(d, "ops.s")).read()


def test_pg_sampler_kernels_keep_scratch_out_of_their_phases(ops_asm):
    """Round 5.  The PG(1) kernels (Bernoulli `aux_sample!` and the Bernoulli point pass of a Gibbs sweep) use
    no scratch memory at all; the general engine's negative-binomial kernels have no scratch instruction between the markers that
    delimit phases A, B1 and B2 (what remains: saves around the engine call and the sequential phase C, whose `cos` carries the
    math library's 24-byte argument-reduction array).  Three things made the difference and each can silently come back: the Philox
    word selection through a selected address (the whole stream lived in scratch), phase C inside the PG(1) kernels, and machine
    LICM hoisting ~100 registers of float64 polynomial coefficients out of the point loops only to spill them."""
    meta = dict(re.findall(r"\.amdhsa_kernel (\S+).*?\.amdhsa_private_segment_fixed_size (\d+)", ops_asm, flags=re.S))
    pg1 = [k for k in meta if "aux_sample_pg1_kernel" in k]
    assert len(pg1) == 2
    for k in pg1:
        assert int(meta[k]) == 0, (k, meta[k])
    checked = 0
    for name, body in re.findall(r"^(_Z\w*(?:aux_sample_kernelILi1E|gibbs_sample_kernelILi1E)\w*):.*?\n(.*?)s_endpgm", ops_asm,
                                 flags=re.S | re.M):
        lines = body.splitlines()
        # the engine has no workgroup barrier since it runs per wave (kPgWaveLocal): the phases are delimited by two marker comments
        # that pg_int_sum_block emits through empty inline asm (the chunk loop's body is laid out between them)
        beg = [i for i, ln in enumerate(lines) if "agpl-pg-phases-begin" in ln]
        end = [i for i, ln in enumerate(lines) if "agpl-pg-phases-end" in ln]
        assert len(beg) == 1 and len(end) == 1 and beg[0] < end[0], (name, beg, end)
        assert sum("s_barrier" in ln for ln in lines) == 1  # (pg_scratch_init's, once per kernel)
        phases = lines[beg[0]:end[0]]
        stray = [ln.strip() for ln in phases if re.search(r"\b(scratch_|buffer_(load|store))", ln)]
        assert not stray, (name, stray[:5])
        assert len(phases) > 2000  # (the three phases really are between those barriers)
        checked += 1
    assert checked == 2


def test_sampler_kernels_contain_no_function_call(ops_asm):
    """The PG sampler engine (pg_int_sum_block and its callers) must be inlined into its kernels: left to its heuristics the
    inliner once turned it into a real call (`s_swappc_b64`), and the negative-binomial aux_sample_kernel went from 8.6 to 18.0 ms
    per 4e6 points without a single source line of it changing.  Checked on the generated code of agpl_ops.hip."""
    asm = ops_asm
    kernels = re.findall(r"^(_Z\w*(?:aux_sample_kernel|gibbs_sample_kernel|aux_sample_pg1_kernel)\w*):.*?\n(.*?)s_endpgm", asm,
                         flags=re.S | re.M)
    assert len(kernels) >= 16  # 8 + 7 likelihood instantiations + the two PG(1) kernels
    for name, body in kernels:
        assert "s_swappc" not in body and "s_call" not in body, name


def test_marginal_kernel_stage_loop_has_no_spill_and_no_stray_vmcnt0(tmp_path):
    """marginal_factor_queue_kernel (agpl_split.hip), round 4: with LDS-DMA pieces in flight the compiler puts `s_waitcnt vmcnt(0)`
    in front of every plain LDS read it cannot prove disjoint from them -- i.e. it waits for the whole flight of the next stage's
    pieces.  The item-end sums and the queue decode read LDS through inline asm for that reason; what must remain in the stage loop
    is the loop's own wait at its top and the wait behind the queue's returning atomic (wave 0, behind its DMA issue).  Also: no
    scratch (the kernel sits at the 128-VGPR cap of a 1024-thread workgroup), and one copy of the 48-MFMA stage body."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    flags = re.search(r"^COMMON\s*:=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), flags=re.M).group(1)
    flags = flags.replace("$(ARCH)", "gfx950").split()
    subprocess.check_call([HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(CSRC, "agpl_split.hip"), "-o", "split.s"],
                          cwd=tmp_path, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(tmp_path, "split.s")).read().splitlines()
    start = next(i for i, ln in enumerate(asm) if re.match(r"^_Z\w*marginal_factor_queue_kernel\w*:", ln))
    end = next(i for i in range(start, len(asm)) if ".end_amdhsa_kernel" in asm[i])
    body = asm[start:end]
    assert any(".amdhsa_private_segment_fixed_size 0" in ln for ln in body), "the marginal kernel spills"
    code = _code(body)
    assert not [ln for ln in code if "scratch_" in ln]
    assert sum("v_mfma_f32_16x16x32_f16" in ln for ln in code) == 48
    # the stage loop: from the first barrier that follows a vmcnt(0) (its top) to the kernel's last barrier
    bars = [i for i, ln in enumerate(code) if re.match(r"\s*s_barrier", ln)]
    assert len(bars) == 3  # the queue's first item, the stage loop's, the one in front of the last rows out
    loop = code[bars[1]:bars[2]]
    waits = [ln.strip() for ln in loop if re.search(r"s_waitcnt.*vmcnt\(0\)", ln)]
    # behind the atomic + (at the loop bottom, in front of the last barrier) the drain of the final item
    assert len(waits) <= 2, waits
    assert sum("global_load_lds_dwordx4" in ln for ln in loop) >= 8
