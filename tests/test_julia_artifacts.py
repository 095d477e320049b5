"""The boundary made literal, CPU-only checks:

* julia/AGPLDeviceExt.jl: every `ccall` names a symbol that include/agpl.h declares and libagpl.so exports, with as
  many argument types as the C prototype has parameters; the operator surface of the reference
  (/root/reference/src/generic.jl:5-24,64-72) is overloaded method by method.
* bench/julia_ref.jl exists and runs iff `julia` is on PATH (it is not in the build image: skip, not fail).
* include/agpl.h is self-contained for a non-ctypes FFI: a C11 and a C++17 translation unit that include nothing else
  compile with -Wall -Wextra -Werror -pedantic, link against libagpl.so and run (no GPU needed: agpl_version()).
"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "agpl.h")
EXT = os.path.join(ROOT, "julia", "AGPLDeviceExt.jl")
PKG = os.path.join(ROOT, "augmentedgplikelihoods.jl_amd")


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"AGPL_API\s+[\w\s\*]+?\b(agpl_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return protos


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def julia_ccalls():
    src = open(EXT).read()
    src = re.sub(r"#=.*?=#", "", src, flags=re.S)
    src = "\n".join(ln.split("#")[0] for ln in src.splitlines())
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*libagpl\)", src):
        # scan the balanced argument list of this ccall
        i = src.index("(", m.start())  # the ccall( itself
        depth, j = 0, i
        while True:
            if src[j] == "(":
                depth += 1
            elif src[j] == ")":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        parts = split_top(src[i + 1:j])  # (:sym, lib), Ret, (types...), args...
        types = split_top(parts[2].strip()[1:-1])
        calls.append((m.group(1), len(types), len(parts) - 3))
    return calls


def test_every_julia_ccall_matches_the_header():
    protos = header_prototypes()
    assert len(protos) == 45
    calls = julia_ccalls()
    assert len(calls) >= 20
    for name, ntypes, nargs in calls:
        assert name in protos, f"{name} is not declared in include/agpl.h"
        assert ntypes == protos[name], f"{name}: {ntypes} ccall argument types, header has {protos[name]} parameters"
        assert nargs == ntypes, f"{name}: {nargs} arguments passed for {ntypes} declared types"
    used = {c[0] for c in calls}
    # the operator surface and the sweep must all be routed
    for need in ("agpl_aux_sample", "agpl_aux_posterior", "agpl_potential_precision",
                 "agpl_expected_potential_precision", "agpl_logtilt", "agpl_aug_loglik", "agpl_expected_logtilt",
                 "agpl_aux_kldivergence", "agpl_plan_bytes", "agpl_plan_create", "agpl_plan_destroy", "agpl_cavi_pass_plan",
                 "agpl_plan_update", "agpl_marginals_plan", "agpl_plan_state", "agpl_expected_aug_loglik",
                 "agpl_allreduce_nat", "agpl_ctx_set_point_offset"):
        assert need in used, need


def test_julia_shim_overloads_the_reference_operator_surface():
    src = open(EXT).read()
    for fn in ("aux_sample!", "aux_posterior!", "auglik_potential", "auglik_precision",
               "auglik_potential_and_precision", "expected_auglik_potential", "expected_auglik_precision",
               "expected_auglik_potential_and_precision", "logtilt", "aug_loglik", "expected_logtilt",
               "expected_aug_loglik", "aux_kldivergence"):
        assert re.search(r"^import AugmentedGPLikelihoods:.*?\b" + re.escape(fn), src, flags=re.S | re.M), fn
        assert re.search(r"^(function )?" + re.escape(fn) + r"\(", src, flags=re.M), f"no method of {fn}"
    # all eight likelihood families (the files of src/likelihoods/) have a descriptor, the categorical ones with logtheta
    for kind in range(8):
        assert re.search(r"LikDesc\(%d," % kind, src), kind
    assert "Vector{<:Normal}" in src or "AbstractVector{<:Normal}" in src  # the reference's own qf argument type
    assert src.count("(") == src.count(")") and src.count("[") == src.count("]")



def test_julia_reference_bench_runs_iff_julia_is_present():
    script = os.path.join(ROOT, "bench", "julia_ref.jl")
    assert os.path.exists(script)
    if shutil.which("julia") is None:
        pytest.skip("no julia on PATH (SURVEY.md F6): bench/julia_ref.jl is written, not executed here")
    r = subprocess.run(["julia", script, "20000", "64", "1"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    assert '"kind": "reference"' in r.stdout


C_TU = r"""
#include "agpl.h"
#include <stdio.h>
int main(void) {
    agpl_lik_desc d;
    d.kind = AGPL_LIK_BERNOULLI_LOGISTIC; d.nlatent = 1; d.p[0] = d.p[1] = d.p[2] = d.p[3] = 0.0; d.logtheta = 0;
    int32_t (*sample)(agpl_ctx *, const agpl_lik_desc *, int64_t, const void *, const double *, double *, int64_t *,
                      uint32_t, uint32_t *, uint32_t *) = agpl_aux_sample;
    int32_t (*pass)(agpl_plan *, const agpl_lik_desc *, const float *, const void *, double *, double *, float *, float *,
                    float *, double *) = agpl_cavi_pass_plan;
    int32_t rc = agpl_aux_sample((agpl_ctx *)0, &d, 0, 0, 0, 0, 0, 0u, 0, 0); /* null context: argument error */
    printf("%d %d %d %d\n", (int)agpl_version(), (int)rc, sample != 0, pass != 0);
    return (agpl_version() == AGPL_VERSION && rc == AGPL_ERR_INVALID_ARGUMENT && (int)AGPL_F64 == 1) ? 0 : 1;
}
"""


@pytest.mark.parametrize("lang,compiler,std", [("c", "gcc", "-std=c11"), ("cpp", "g++", "-std=c++17")])
def test_header_is_self_contained_for_c_and_cxx(tmp_path, lang, compiler, std):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    lib = os.path.join(PKG, "libagpl.so")
    if not os.path.exists(lib):
        g.build()
    src = tmp_path / f"tu.{lang}"
    src.write_text(C_TU)
    exe = tmp_path / f"tu_{lang}"
    cmd = [compiler, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
           "-o", str(exe), "-L", PKG, "-lagpl", "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib",
           "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert r.stdout.split()[0] == "121"  # AGPL_VERSION: 45 exports
