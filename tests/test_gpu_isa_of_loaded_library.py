"""-m gpu twins of two ISA guards (tests/test_isa_guards.py is CPU tier: it only ever sees the CONTAINER's compiler).  A rebuild on
the GPU box under its own ROCm -- or a stale object that travelled with the snapshot -- could bring the sampler kernels' scratch
back, or turn the sampler engine into a real function call, without any CPU-tier test noticing.  These
tests take the library the process has actually LOADED (the path of the mapped libagpl.so from /proc/self/maps), pull the gfx950
code objects out of it (llvm-objdump --offloading, on a copy in a temporary directory) and check the kernels' metadata notes and
disassembly:
  * aux_sample_pg1_kernel<false / true>: private_segment_fixed_size == 0 (no scratch at all), no register spilled;
  * no s_swappc / s_call in any sampler kernel (the engine is inlined);
  * marginal_factor_queue_kernel and syrk_strip_kernel: the shipped contraction kernels are present, the marginal kernel without
    scratch and at <= 128 VGPRs (16 waves per CU), the accumulation at <= 256."""
import glob
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def loaded_code_objects(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    import agpl_amd as A

    A._ffi.lib()
    A.Context(0, seed=1)  # (the code objects are registered with the runtime: this really is the library in use)
    paths = {ln.split()[-1] for ln in open("/proc/self/maps") if ln.rstrip().endswith("libagpl.so")}
    assert len(paths) == 1, paths
    so = paths.pop()
    assert os.path.realpath(so) == os.path.realpath(A._ffi.LIB_PATH)
    d = tmp_path_factory.mktemp("loaded_so")
    shutil.copy(so, os.path.join(d, "lib.so"))
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    cos = sorted(glob.glob(os.path.join(d, "lib.so.*gfx950")))
    assert cos, "no gfx950 code object in the loaded library"
    notes, disasm = "", {}
    for co in cos:
        n = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        notes += n
        if "aux_sample" in n or "gibbs_sample" in n:
            disasm[co] = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
    return notes, disasm


def _kernel_meta(notes):
    """{kernel name: {field: int}} from the amdhsa metadata notes."""
    out = {}
    for blk in re.split(r"\n\s+- \.agpr_count:", notes):
        m = re.search(r"\.name:\s+(\S+)", blk)
        if not m:
            continue
        fields = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\d+)", blk)}
        out[m.group(1)] = fields
    return out


def test_pg1_kernels_of_the_loaded_library_use_no_scratch(loaded_code_objects):
    meta = _kernel_meta(loaded_code_objects[0])
    pg1 = {k: v for k, v in meta.items() if "aux_sample_pg1_kernel" in k}
    assert len(pg1) == 2, sorted(meta)[:5]
    for k, v in pg1.items():
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0, (k, v)
        assert v["vgpr_count"] <= 128, (k, v)  # four waves per SIMD


def test_sampler_kernels_of_the_loaded_library_contain_no_call(loaded_code_objects):
    notes, disasm = loaded_code_objects
    assert disasm
    seen = 0
    for text in disasm.values():
        for name, body in re.findall(r"^[0-9a-f]+ <(_Z\w*(?:aux_sample_kernel|gibbs_sample_kernel|aux_sample_pg1_kernel)\w*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)",
                                     text, flags=re.S | re.M):
            assert "s_swappc" not in body and "s_call" not in body, name
            seen += 1
    assert seen >= 16  # 8 + 7 likelihood instantiations + the two PG(1) kernels


def test_contraction_kernels_of_the_loaded_library(loaded_code_objects):
    meta = _kernel_meta(loaded_code_objects[0])
    marg = [v for k, v in meta.items() if "marginal_factor_queue_kernel" in k]
    acc = [v for k, v in meta.items() if "syrk_strip_kernel" in k]
    assert len(marg) == 1 and len(acc) == 1
    assert marg[0]["private_segment_fixed_size"] == 0 and marg[0]["vgpr_count"] <= 128, marg
    assert acc[0]["vgpr_count"] <= 256 and acc[0]["private_segment_fixed_size"] <= 64, acc
