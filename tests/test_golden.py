"""Oracle against the committed self-generated vectors (tests/golden/oracle_selfcheck.json, made by
tests/golden/make_golden.py).  These are NOT reference (Julia) outputs -- see the generator's docstring."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "oracle_selfcheck.json")) as f:
        return json.load(f)


def test_stream_and_samplers_are_frozen(oracle, golden):
    O, S = oracle, golden["seed"]
    assert O.uniforms(S, 3, 1, 6).tolist() == golden["uniforms_stream3_sweep1"]  # bit-exact stream mapping
    for key, ref in golden["rand_pg"].items():
        b, c = (float(v) for v in key.split(","))
        x, nu, nt = O.rand_pg(b, c, 6, seed=S, stats=True)
        assert nu.tolist() == ref["nuni"] and nt.tolist() == ref["nterms"]  # integer bookkeeping: bit-exact
        assert np.allclose(x, ref["omega"], rtol=1e-13, atol=0)
    assert np.allclose(O.rand_gamma(2.25, 1.0, 5, seed=S), golden["rand_gamma_2.25"], rtol=1e-13)
    assert O.rand_poisson(0.7, 12, seed=S).tolist() == golden["rand_poisson_0.7"]
    assert O.rand_poisson(40.0, 8, seed=S).tolist() == golden["rand_poisson_40"]


def test_closed_forms_and_synthetic_inputs(oracle, golden):
    O, S = oracle, golden["seed"]
    for z, ref in golden["mass_texpon"].items():
        assert O.pg_mass_texpon(float(z)) == pytest.approx(ref, rel=1e-14)
    for key, ref in golden["pg_mean"].items():
        b, c = (float(v) for v in key.split(","))
        assert O.pg_mean(b, c) == ref
    assert np.allclose(O.synth_x(S, 0, 5), golden["synth_x"], rtol=0, atol=1e-15)
    assert O.synth_y(O.bernoulli(), S, 0, 24).tolist() == golden["synth_y_bernoulli"]
    assert O.synth_y(O.negbinomial(15.0), S, 0, 12).tolist() == golden["synth_y_negbin15"]
    g = golden["bernoulli_cavi_ops"]
    c, _, _ = O.aux_posterior(O.bernoulli(), np.array(g["y"], np.uint8), np.array(g["mu"]), np.array(g["var"]))
    beta, gamma = O.expected_potential_precision(O.bernoulli(), np.array(g["y"], np.uint8), c)
    assert np.allclose(c, g["c"], rtol=1e-14) and np.allclose(gamma[0], g["gamma"], rtol=1e-14)
    assert beta[0].tolist() == g["beta"]


@pytest.mark.gpu
def test_device_reproduces_the_golden_draws(oracle, golden):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge

    ge.build()
    import agpl_amd as A

    ctx = A.Context(0, seed=golden["seed"])
    for key, ref in golden["rand_pg"].items():
        b, c = (float(v) for v in key.split(","))
        out = torch.empty(6, dtype=torch.float64, device="cuda")
        _, nuni, nterms = A.rand_polyagamma(b, c, out, ctx=ctx, sweep=0, stats=True)
        assert nuni.cpu().tolist() == ref["nuni"] and nterms.cpu().tolist() == ref["nterms"]
        assert np.allclose(out.cpu().numpy(), ref["omega"], rtol=1e-10)
    lik = A.BernoulliLikelihood()
    _, y = A.synth_xy(lik, golden["seed"], 0, 24, ctx=ctx)
    assert y.cpu().tolist() == golden["synth_y_bernoulli"]
