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


@pytest.fixture(scope="module")
def strip_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    d = tmp_path_factory.mktemp("isa")
    flags = re.search(r"^COMMON\s*:=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), flags=re.M).group(1)
    flags = flags.replace("$(ARCH)", "gfx950").split()
    subprocess.check_call([HIPCC] + flags + ["-save-temps", "-c", os.path.join(CSRC, "agpl_syrk.hip"), "-o", "/dev/null"],
                          cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(d, "agpl_syrk-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    start = next(i for i, ln in enumerate(asm) if re.match(r"^_Z\w*syrk_strip_kernel\w*:", ln))
    end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
    return [ln.split(";")[0].rstrip() for ln in asm[start:end + 1]]


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _operands(line):
    toks = re.split(r"[,\s]+", line.strip())
    return toks[0], [t for t in toks[1:] if t]


def test_asm_load_destinations_are_untouched_until_the_wait(strip_isa):
    loads = [i for i, ln in enumerate(strip_isa) if re.match(r"\s*global_load_dwordx4\s", ln)]
    assert len(loads) >= 8  # prologue + step loop, four each (off-diagonal body)
    for i in loads:
        dst = _regs(_operands(strip_isa[i])[1][0])
        assert len(dst) == 4
        j = i + 1
        # to the end of the basic block (a label, a branch, the barrier or the wait itself): exact inside a block, and the
        # failure seen was a block of v_mov copies directly behind the loads
        while not re.search(r"s_waitcnt.*vmcnt|^\s*s_c?branch|^\s*s_barrier|^\.?\w+:\s*$|s_endpgm", strip_isa[j]):
            if strip_isa[j].strip():
                op, args = _operands(strip_isa[j])
                if not op.startswith("global_load_dwordx4"):
                    touched = set().union(*[_regs(a) for a in args]) if args else set()
                    assert not (touched & dst), f"line {j}: `{strip_isa[j].strip()}` touches an in-flight load destination"
            j += 1
        assert j > i + 1


def test_no_scratch_traffic_inside_the_step_loops(strip_isa):
    # loop bodies: from a line that carries s_barrier to the next backward branch; coarse but sufficient: no scratch_*
    # instruction may sit between the first and the last v_mfma of the kernel's loops that also hold an LDS-DMA
    idx_dma = [i for i, ln in enumerate(strip_isa) if "global_load_lds_dwordx4" in ln]
    idx_mfma = [i for i, ln in enumerate(strip_isa) if "v_mfma_f32_16x16x32_f16" in ln]
    assert idx_dma and idx_mfma
    # segments of MFMA code (two bodies: diagonal and off-diagonal tiles)
    segs, cur = [], [idx_mfma[0], idx_mfma[0]]
    for i in idx_mfma[1:]:
        if i - cur[1] > 400:
            segs.append(cur)
            cur = [i, i]
        cur[1] = i
    segs.append(cur)
    assert len(segs) >= 2
    for a, b in segs:
        body = strip_isa[a:b + 1]
        assert not [ln for ln in body if "scratch_" in ln], "spill traffic inside a step loop"
        assert sum("v_mfma_f32_16x16x32_f16" in ln for ln in body) >= 96


def test_the_step_loops_wait_with_a_counted_vmcnt(strip_isa):
    waits = [ln.strip() for ln in strip_isa if re.search(r"s_waitcnt.*vmcnt", ln)]
    assert sum(w == "s_waitcnt vmcnt(5)" for w in waits) >= 2  # one per body: the step's five DMA pieces stay in flight


def test_sampler_kernels_contain_no_function_call(tmp_path):
    """The PG sampler engine (pg_int_sum_block and its callers) must be inlined into its kernels: left to its heuristics the
    inliner once turned it into a real call (`s_swappc_b64`), and the negative-binomial aux_sample_kernel went from 8.6 to 18.0 ms
    per 4e6 points without a single source line of it changing.  Checked on the generated code of agpl_ops.hip."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^COMMON\s*:=\s*(.*)$", mk, flags=re.M).group(1).replace("$(ARCH)", "gfx950").split()
    flags += re.search(r"^NOFMA\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    subprocess.check_call([HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(CSRC, "agpl_ops.hip"), "-o", "ops.s"],
                          cwd=tmp_path, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(tmp_path, "ops.s")).read()
    kernels = re.findall(r"^(_Z\w*(?:aux_sample_kernel|gibbs_sample_kernel|aux_sample_pg1_kernel)\w*):.*?\n(.*?)s_endpgm", asm,
                         flags=re.S | re.M)
    assert len(kernels) >= 16  # 8 + 7 likelihood instantiations + the two PG(1) kernels
    for name, body in kernels:
        assert "s_swappc" not in body and "s_call" not in body, name
