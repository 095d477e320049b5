"""Round-5 regression tests.

The PG(1) kernel pair (Bernoulli `aux_sample!`, bernoulli.jl:13-15, and the Bernoulli point pass of a sparse Gibbs sweep) keeps 32-bit point
indices on its retry list, so inputs beyond 2^30 points go through in several launches with every pointer and the Philox stream
offset advanced (launch_pg1, agpl_ops.hip).  No test can afford 2^30 points: a build of the library with the limit set to 5000
(-DAGPL_PG1_MAX_LAUNCH=5000, compiled here by hipcc) must reproduce the shipped library bit for bit at n = 12 345 -- three launches,
the last one ragged -- for the draws, the uniforms consumed and the series indices, and for the Gibbs pass's f, gamma, beta."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "augmentedgplikelihoods.jl_amd")

CHILD = r'''
import hashlib, json, os, sys
sys.path.insert(0, os.environ["AGPL_ROOT"])
import numpy as np, torch
import agpl_amd as A
from agpl_amd import _ffi
if os.environ.get("AGPL_LIB_AB"):
    _ffi.LIB_PATH = os.environ["AGPL_LIB_AB"]
ctx = A.Context(0, seed=77)
n = 12345
g = torch.Generator(device="cuda").manual_seed(3)
f = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 3.0
f[::97] = 25.0  # (no fitted branch mass: these go to the retry list)
lik = A.BernoulliLikelihood()
Om, nuni, nterms = A.aux_sample_(A.init_aux_variables(lik, n, ctx=ctx), lik, None, f, ctx=ctx, sweep=4, stats=True)
h = hashlib.sha256()
for t in (Om.ω, nuni, nterms):
    h.update(t.detach().cpu().numpy().tobytes())
# the Bernoulli point pass of a sparse Gibbs sweep through the same kernel pair
M = 256
Phi = torch.randn((n, M), device="cuda", generator=g) * 0.1
kd = (Phi * Phi).sum(1) + 0.3
y = (torch.rand(n, device="cuda", generator=g) < 0.5).to(torch.uint8)
gib = A.SparseGibbs(lik, Phi, kd, y, ctx=ctx)
for _ in range(2):
    gib.sweep()
torch.cuda.synchronize()
for t in (gib.G, gib.g):
    h.update(t.detach().cpu().numpy().tobytes())
print(json.dumps({"sha": h.hexdigest(), "omega_sum": float(Om.ω.sum().item())}))
'''


def _run(lib):
    env = dict(os.environ, AGPL_ROOT=ROOT)
    if lib:
        env["AGPL_LIB_AB"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.timeout(1800)
def test_pg1_kernel_pair_in_several_launches_reproduces_one_launch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    lib = os.path.join(PKG, "libagpl_pg1chunk.so")
    # (the variant links against the regular build's objects: a no-op where they travelled with the tree, a full build where not)
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=800)
    try:
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "pg1chunk", "agpl_ops.hip", "-DAGPL_PG1_MAX_LAUNCH=5000"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        ref = _run(None)
        got = _run(lib)
    finally:
        if os.path.exists(lib):
            os.remove(lib)
    assert got["sha"] == ref["sha"], (got, ref)
    assert ref["omega_sum"] > 0
