"""Pins the CPU oracle against every known answer the reference's own tests hold.

The reference has no golden vectors (its tests check closed forms, self-consistency identities and
one unseeded sample mean), so these are the pins available:
  1. test/SpecialDistributions/polyagamma.jl:27-28,30   closed-form PG means
  2. test/SpecialDistributions/polyagamma.jl:36         |mean(10000 draws) - E| <= 1e-2, six (b,c)
  3. test/SpecialDistributions/polyagamma.jl:3-19,34-35 density series == independent 4001-term series
  4. src/SpecialDistributions/polyagamma.jl:231         hard-coded r(z=0)
  5. test/utils.jl:6-13                                 second_moment, approx_expected_logistic
  6. src/TestUtils.jl:80-88,107-148,162-171             shape/sign rules + the two full-conditional identities
  7. test/likelihoods/laplace.jl:6-9                    Laplace KL closed form
plus the Random123 known-answer vectors for Philox4x32-10.
"""
import numpy as np
import pytest
from scipy import special, stats

PAIRS = ((1, 0), (1, 2.0), (3, 0), (3, 2.5), (3, 3.2), (1.2, 3.2))


def test_philox_known_answers(oracle):
    # Random123 kat_vectors: philox4x32 10
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_uniform_stream_open_interval_and_reproducible(oracle):
    u = oracle.uniforms(7, 3, 1, 10001)
    assert np.all(u > 0) and np.all(u < 1)
    assert np.array_equal(u, oracle.uniforms(7, 3, 1, 10001))
    assert not np.array_equal(u[:100], oracle.uniforms(7, 4, 1, 100))
    assert abs(u.mean() - 0.5) < 0.01


def test_pg_mean_closed_forms(oracle):
    assert oracle.pg_mean(1, 0) == 1 / 4  # polyagamma.jl test :27
    assert oracle.pg_mean(1, 2.0) == np.tanh(1.0) / 4  # :28
    for b, c in PAIRS:
        ref = b / 4 if c == 0 else b / (2 * c) * np.tanh(c / 2)
        assert oracle.pg_mean(b, c) == ref


def test_mass_texpon_hardcoded_constant(oracle):
    # polyagamma.jl:231 hard-codes r at z = 0; mass_texpon(z -> 0) must reproduce it bit for bit
    assert oracle.pg_mass_texpon(1e-300) == 0.5776972428360435
    # independent evaluation through scipy's normal logcdf
    for z in (0.5, 1.0, 2.5, 10.0, 40.0, 60.0):
        K = np.pi ** 2 / 8 + z * z / 2
        t = 0.64
        b = (t * z - 1) / np.sqrt(t)
        a = -(t * z + 1) / np.sqrt(t)
        x0 = np.log(K) + K * t
        q = 4 / np.pi * (np.exp(x0 - z + stats.norm.logcdf(b)) + np.exp(x0 + z + stats.norm.logcdf(a)))
        assert oracle.pg_mass_texpon(z) == pytest.approx(1 / (1 + q), rel=1e-11)


def test_normlogcdf_matches_scipy(oracle):
    for z in np.concatenate([np.linspace(-60, 8, 137), [-35.0, -34.999, -1.0, -0.999]]):
        assert oracle.normlogcdf(z) == pytest.approx(stats.norm.logcdf(z), rel=2e-12, abs=1e-300)


def test_a_coefficients_continuous_at_truncation(oracle):
    # the two branches of a(n,x) (polyagamma.jl:167-177) are two series for the same function and agree
    # to O(1e-4) at the truncation point t = 0.64 (Devroye); the piecewise choice is what is tested here.
    for n in range(4):
        lo, hi = oracle.pg_a(n, 0.64), oracle.pg_a(n, 0.64 + 1e-12)
        k = (n + 0.5) * np.pi
        assert hi == pytest.approx(k * np.exp(-k * k * 0.64 / 2), rel=1e-12)
        assert lo == pytest.approx(k * np.exp(-1.5 * (np.log(np.pi / 2) + np.log(0.64)) - 2 * (n + 0.5) ** 2 / 0.64),
                                   rel=1e-12)
    assert np.isnan(oracle.pg_a(0, 0.0)) and np.isnan(oracle.pg_a(1, -1.0))  # DomainError in the reference


@pytest.mark.parametrize("b,c", PAIRS)
def test_sampler_mean_within_reference_tolerance(oracle, b, c):
    # test/SpecialDistributions/polyagamma.jl:36 -- mean(rand(p, 10000)) ~ mean(p) atol 1e-2
    s = oracle.rand_pg(b, c, 10000, seed=20240807)
    assert abs(s.mean() - oracle.pg_mean(b, c)) <= 1e-2
    assert np.all(s > 0)


def _ref_logpdf(b, c, x):
    # test/SpecialDistributions/polyagamma.jl:3-23 independent 4001-term series
    n = np.arange(0, 4002)
    terms = special.gammaln(n + b) - special.gammaln(n + 1) - (2 * n + b) ** 2 / (8 * x) + np.log(2 * n + b)
    pos = special.logsumexp(terms[0::2])
    neg = special.logsumexp(terms[1::2])
    logcosh = abs(c / 2) + np.log1p(np.exp(-abs(c))) - np.log(2)
    ext = b * logcosh - c * c * x / 2 + (b - 1) * np.log(2) - special.gammaln(b) - (np.log(2 * np.pi) + 3 * np.log(x)) / 2
    return ext + np.log(np.exp(pos) - np.exp(neg))


@pytest.mark.parametrize("b,c", PAIRS)
def test_logpdf_series_against_independent_series(oracle, b, c):
    xs = 10.0 ** np.arange(-2.5, 0.5001, 0.1)  # :34
    got = np.array([oracle.pg_logpdf(b, c, x) for x in xs])
    ref = np.array([_ref_logpdf(b, c, x) for x in xs])
    assert np.allclose(got, ref, rtol=1e-8, atol=0)  # Julia isapprox default rtol = sqrt(eps)
    wide = np.array([oracle.pg_logpdf(b, c, x) for x in 10.0 ** np.arange(-7, 7.01, 0.1)])  # :32
    assert not np.any(np.isnan(wide))


@pytest.mark.parametrize("b,c", ((1, 0), (1, 2.0), (3, 2.5), (1, 9.0)))
def test_sampler_distribution_ks_against_density_series(oracle, b, c):
    # strengthens the reference's mean-only check: KS distance of 20000 draws against the CDF obtained by
    # integrating the reference's own density series (polyagamma.jl:37-91)
    s = np.sort(oracle.rand_pg(b, c, 20000, seed=99))
    grid = np.concatenate([[1e-6], np.geomspace(1e-4, s.max() * 1.5 + 1, 6000)])
    pdf = np.exp([oracle.pg_logpdf(b, c, x) for x in grid])
    cdf = np.concatenate([[0], np.cumsum((pdf[1:] + pdf[:-1]) / 2 * np.diff(grid))])
    assert cdf[-1] == pytest.approx(1.0, abs=2e-3)
    emp = np.arange(1, s.size + 1) / s.size
    d = np.max(np.abs(emp - np.interp(s, grid, cdf / cdf[-1])))
    assert d < 1.63 / np.sqrt(s.size) + 2e-3  # 1% KS critical value + quadrature slack


def test_third_party_samplers_distribution(oracle):
    # Distributions.jl samplers are not under /root/reference ("upstream, unpinned"): check the laws.
    for shape in (0.3, 1.0, 2.25, 7.5):
        g = oracle.rand_gamma(shape, 1.7, 20000, seed=5)
        assert stats.kstest(g, stats.gamma(shape, scale=1.7).cdf).pvalue > 1e-3
    for mu in (0.05, 0.9, 5.5, 6.0, 40.0, 300.0):
        k = oracle.rand_poisson(mu, 40000, seed=6)
        assert abs(k.mean() - mu) < 5 * np.sqrt(mu / k.size)
        assert abs(k.var() - mu) < 0.06 * mu + 0.01
        hi = int(stats.poisson(mu).ppf(0.999)) + 1
        obs = np.bincount(np.minimum(k, hi), minlength=hi + 1)
        exp = stats.poisson(mu).pmf(np.arange(hi + 1)) * k.size
        exp[-1] = k.size - exp[:-1].sum()
        keep = exp > 5
        chi = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
        assert chi < stats.chi2(keep.sum()).ppf(0.9999)
    for mu, lam in ((0.5, 0.5), (2.0, 0.125), (0.1, 3.0)):
        x = oracle.rand_invgaussian(mu, lam, 20000, seed=8)
        assert stats.kstest(x, stats.invgauss(mu / lam, scale=lam).cdf).pvalue > 1e-3


def test_utils(oracle):
    rng = np.random.default_rng(42)
    m, s = rng.normal(), rng.uniform()
    c = m * m + s * s
    # test/utils.jl:8
    assert oracle.approx_expected_logistic(m, c) == pytest.approx(np.exp(m / 2) / np.cosh(c / 2) / 2, abs=1e-5)
    # test/utils.jl:9-13: saturation for Float64 and Float32
    for f32 in (False, True):
        big = 1000.0
        assert oracle.approx_expected_logistic(big, big + abs(rng.normal()), f32=f32) == pytest.approx(1.0)
        assert oracle.approx_expected_logistic(-big, big + 1.0, f32=f32) == 0.0


def _liks(O):
    return {
        "bernoulli": O.bernoulli(),
        "negbin10": O.negbinomial(10),
        "negbin5.5": O.negbinomial(5.5),
        "studentt": O.studentt(3.0, 1.5),
        "poisson": O.poisson(10.0),
        "laplace": O.laplace(1.0),
        "cat3": O.categorical(np.zeros(3)),
        "cat3bij": O.categorical(np.zeros(3), bijective=True),
        "hetero": O.heterogauss(5.0),
    }


def _gen_y(O, lik, f, rng):
    n = f.size // lik.nlatent
    if lik.kind == O.BERNOULLI:
        return (rng.uniform(size=n) < 1 / (1 + np.exp(-f))).astype(np.uint8)
    if lik.kind in (O.NEGBINOMIAL, O.POISSON):
        return rng.poisson(3.0, size=n).astype(np.int32)
    if lik.kind in (O.CATEGORICAL, O.CATEGORICAL_BIJ):
        L = lik.nlatent
        K = L + (1 if lik.kind == O.CATEGORICAL_BIJ else 0)
        lab = rng.integers(0, K, size=n)
        return (lab[:, None] == np.arange(L)[None, :]).astype(np.uint8)
    return rng.normal(size=n)


@pytest.mark.parametrize("name", ["bernoulli", "negbin10", "negbin5.5", "studentt", "poisson", "laplace", "cat3",
                                  "cat3bij", "hetero"])
def test_auglik_conformance(oracle, name):
    """Restates src/TestUtils.jl:57-206 test_auglik(lik; n=10): arity, shapes, gamma >= 0,
    potential_and_precision agreement (trivially the same call here), VI outputs finite."""
    O = oracle
    lik = _liks(O)[name]
    rng = np.random.default_rng(1)
    n, L = 10, lik.nlatent
    f = rng.normal(size=(n, L))
    y = _gen_y(O, lik, f.ravel() if L == 1 else f, rng)
    draw = O.aux_sample(lik, y, f, seed=11)
    assert draw["omega"].shape[0] == n
    beta, gamma = O.potential_precision(lik, y, draw["omega"], draw.get("n"), fg=f if name == "hetero" else None)
    assert beta.shape == gamma.shape == (L, n)  # TestUtils.jl:83 length == nlatent
    assert np.all(gamma >= 0)  # :88
    assert np.all(np.isfinite(beta))
    qmu, qvar = rng.normal(size=(n, L)), np.ones((n, L))
    q1, q2, q3 = O.aux_posterior(lik, y, qmu, qvar)
    eb, eg = O.expected_potential_precision(lik, y, q1, q2, mu_g=qmu[:, -1] if name == "hetero" else None)
    assert eb.shape == eg.shape == (L, n)  # :165
    assert np.all(eg >= 0)  # :171
    if name not in ("cat3", "hetero"):
        assert np.isfinite(O.aux_kl(lik, y, q1, q2))  # :201-203
    else:
        assert np.isnan(O.aux_kl(lik, y, q1, q2))  # error() categorical.jl:165-170
    if name != "hetero":
        assert np.isfinite(O.expected_logtilt(lik, y, q1, q2, qmu, qvar))  # :202


def _scipy_cond_logpdf(O, name, lik, y, f, om, nn):
    """The conditional's log-density from scipy / the PG series alone (no oracle density code besides pg_logpdf):
    an independent statement of what agplo_full_conditional_logpdf restates."""
    if name == "bernoulli":
        return sum(O.pg_logpdf(1, abs(fi), w) for fi, w in zip(f, om))
    if name.startswith("negbin"):
        r = lik.p[0]
        return sum(O.pg_logpdf(yi + r, abs(fi), w) for yi, fi, w in zip(y, f, om))
    if name == "studentt":
        a = (3.0 + 1) / 2
        sc = 2 / (3.0 / 1.5 ** 2 + (y - f) ** 2)
        return stats.gamma(a, scale=sc).logpdf(om).sum()
    if name == "poisson":  # PolyaGammaPoisson(y, |f|, lambda sigma(-f)): poisson.jl:26-28, polyagammapoisson.jl:29-33
        lam = lik.p[0] / (1 + np.exp(f))
        return stats.poisson(lam).logpmf(nn).sum() + sum(O.pg_logpdf(yi + ni, abs(fi), w) for yi, ni, fi, w in zip(y, nn, f, om))
    if name == "laplace":  # InverseGaussian(1 / (2 beta |y - f|), 2 (2 beta)^-2): laplace.jl:40-42
        beta = lik.p[0]
        mu, lam = 1 / (2 * beta * np.abs(y - f)), 2 / (2 * beta) ** 2
        return stats.invgauss(mu / lam, scale=lam).logpdf(om).sum()
    if name == "hetero":  # PolyaGammaPoisson(1/2, |g|, lambda sigma(-g) (f - y)^2 / 2): heteroscedasticgaussian.jl:28-32
        ff, gg = f[:, 0], f[:, 1]
        lam = lik.p[0] / (1 + np.exp(gg)) * (ff - y) ** 2 / 2
        return stats.poisson(lam).logpmf(nn).sum() + sum(O.pg_logpdf(0.5 + ni, abs(gi), w) for ni, gi, w in zip(nn, gg, om))
    raise AssertionError(name)


@pytest.mark.parametrize("name", ["bernoulli", "negbin10", "negbin5.5", "studentt", "poisson", "laplace", "hetero"])
def test_full_conditional_omega_identity(oracle, name):
    """src/TestUtils.jl:107-116: log p(y,Omega|f) - log p(Omega|y,f) is the same for two independent
    draws of Omega (atol 1e-5) -- pins the conditional's parameters against tilt + prior through the
    density series.  The reference runs it for Bernoulli, NegBin r = 10 and r = 5.5 (non-integer b ->
    rand_gamma_sum), Poisson, StudentT and Laplace (test/likelihoods/*.jl); the heteroscedastic file is not in
    its runtests.jl but the identity holds for heteroscedasticgaussian.jl:106-128 all the same (the difference
    is log N(y | f, 1/(lambda sigma(g))) up to a constant) and pins that method."""
    O = oracle
    lik = _liks(O)[name]
    rng = np.random.default_rng(3)
    n, L = 10, lik.nlatent
    f = rng.normal(size=n) if L == 1 else rng.normal(size=(n, L))
    y = _gen_y(O, lik, f, rng)
    if name == "poisson":
        y = rng.poisson(3.0, size=n).astype(np.int32)
    vals = []
    for seed in (1, 2):
        d = O.aux_sample(lik, y, f, seed=seed)
        om, nn = d["omega"], d.get("n")
        cond = O.full_conditional_logpdf(lik, y, f, om, nn)
        assert cond == pytest.approx(_scipy_cond_logpdf(O, name, lik, y, f, om, nn), rel=1e-11, abs=1e-10)
        vals.append(O.aug_loglik(lik, y, om, f, nn) - cond)
    assert vals[0] == pytest.approx(vals[1], abs=1e-5)
    # what the constant is: log p(y | f) of the likelihood itself (the marginal over Omega)
    if name == "poisson":
        rate = lik.p[0] / (1 + np.exp(-f))
        assert vals[0] == pytest.approx(stats.poisson(rate).logpmf(y).sum(), abs=1e-5)
    if name == "laplace":
        assert vals[0] == pytest.approx(stats.laplace(f, lik.p[0]).logpdf(y).sum(), abs=1e-5)
    if name == "negbin5.5":
        r, p = 5.5, 1 / (1 + np.exp(-f))  # NBParamFailure(r): y successes before r failures, success prob sigma(f)
        assert vals[0] == pytest.approx(stats.nbinom(r, 1 - p).logpmf(y).sum(), abs=1e-5)
    if name == "bernoulli":
        assert vals[0] == pytest.approx(stats.bernoulli(1 / (1 + np.exp(-f))).logpmf(y).sum(), abs=1e-5)
    if name == "hetero":
        prec = lik.p[0] / (1 + np.exp(-f[:, 1]))
        # aug_loglik (heteroscedasticgaussian.jl:118-128) leaves the normalising constant log(lambda / 2 pi) / 2 out
        assert vals[0] == pytest.approx(stats.norm(f[:, 0], 1 / np.sqrt(prec)).logpdf(y).sum()
                                        - n * 0.5 * np.log(lik.p[0] / (2 * np.pi)), abs=1e-5)


def test_aux_prior_logpdf_closed_forms(oracle):
    """aux_prior densities against scipy: PolyaGammaPoisson(y, 0, lambda) (poisson.jl:67-76, polyagammapoisson.jl:29-33),
    InverseGamma(1/2, (2 beta)^-2) (laplace.jl:90-96), Gamma(nu/2, scale 2 sigma^2/nu) (studentt.jl:91)."""
    O = oracle
    rng = np.random.default_rng(5)
    n = 7
    om = rng.uniform(0.05, 2.0, size=n)
    y = rng.poisson(3.0, size=n).astype(np.int32)
    nn = rng.poisson(2.0, size=n).astype(np.int64)
    ref = stats.poisson(10.0).logpmf(nn).sum() + sum(O.pg_logpdf(yi + ni, 0.0, w) for yi, ni, w in zip(y, nn, om))
    assert O.aux_prior_logpdf(O.poisson(10.0), y, om, nn) == pytest.approx(ref, rel=1e-12)
    lam = 1 / (2 * 0.8) ** 2
    assert O.aux_prior_logpdf(O.laplace(0.8), np.zeros(n), om) == pytest.approx(stats.invgamma(0.5, scale=lam).logpdf(om).sum(), rel=1e-12)
    assert O.aux_prior_logpdf(O.studentt(3.0, 1.5), np.zeros(n), om) == pytest.approx(
        stats.gamma(1.5, scale=1.5 ** 2 / 1.5).logpdf(om).sum(), rel=1e-12)
    assert np.isnan(O.aux_prior_logpdf(O.categorical(np.zeros(3)), np.zeros((n, 3), np.uint8), np.ones((n, 3)), np.zeros((n, 3), np.int64)))


@pytest.mark.parametrize("name", ["bernoulli", "negbin10", "studentt", "poisson", "laplace", "cat3bij", "hetero"])
def test_expected_aug_loglik(oracle, name):
    """expected_aug_loglik = expected_logtilt + aux_kldivergence (generic.jl:52-54, sign as coded); the heteroscedastic method
    (heteroscedasticgaussian.jl:130-145) against a numpy statement of the same expression."""
    O = oracle
    lik = _liks(O)[name]
    rng = np.random.default_rng(6)
    n, L = 9, lik.nlatent
    qmu, qvar = rng.normal(size=(n, L)), rng.uniform(0.2, 1.5, size=(n, L))
    y = _gen_y(O, lik, qmu.ravel() if L == 1 else qmu, rng)
    q1, q2, q3 = O.aux_posterior(lik, y, qmu, qvar)
    got = O.expected_aug_loglik(lik, y, q1, q2, qmu, qvar)
    if name != "hetero":
        assert got == pytest.approx(O.expected_logtilt(lik, y, q1, q2, qmu, qvar) + O.aux_kl(lik, y, q1, q2), rel=1e-13)
        return
    lam = lik.p[0]
    c, lq = q1.ravel(), q2.ravel()
    tw = np.array([O.pg_mean(0.5 + a, b) for a, b in zip(lq, c)])
    mf, vf, g, vg = qmu[:, 0], qvar[:, 0], qmu[:, 1], qvar[:, 1]
    lp = lam / 2 * ((y - mf) ** 2 + vf)
    kl = np.array([O.pg_kl(0.5 + a, b) for a, b in zip(lq, c)]) + lq * (np.log(lq) - np.log(lp)) - lq + lp
    ref = np.sum(0.5 * (np.log(lam) + np.log(2 / np.pi)) - (0.5 + lq) * np.log(2) + ((0.5 - lq) * g - (g ** 2 + vg) * tw) / 2 + kl)
    assert got == pytest.approx(ref, rel=1e-12)


@pytest.mark.parametrize("name", ["bernoulli", "negbin10", "studentt", "poisson", "laplace"])
def test_full_conditional_f_identity(oracle, name):
    """src/TestUtils.jl:118-131: with K = AA', S = (K^-1 + Diag gamma)^-1, m = S beta:
    logtilt(f) + log p(f) - log q(f) is the same for two draws f ~ q (atol 1e-5) -- pins beta, gamma
    against logtilt."""
    O = oracle
    lik = _liks(O)[name]
    rng = np.random.default_rng(4)
    n = 10
    f0 = rng.normal(size=n)
    y = _gen_y(O, lik, f0, rng)
    d = O.aux_sample(lik, y, f0, seed=21)
    beta, gamma = O.potential_precision(lik, y, d["omega"], d.get("n"))
    A = rng.uniform(size=(n, n))
    K = A @ A.T
    S = np.linalg.inv(np.linalg.inv(K) + np.diag(gamma[0]))
    S = (S + S.T) / 2
    m = S @ beta[0]
    qF, pF = stats.multivariate_normal(m, S), stats.multivariate_normal(np.zeros(n), K, allow_singular=True)
    vals = []
    for _ in range(2):
        f = qF.rvs(random_state=rng)
        vals.append(O.logtilt(lik, y, d["omega"], f, d.get("n")) + pF.logpdf(f) - qF.logpdf(f))
    assert vals[0] == pytest.approx(vals[1], abs=1e-5)


def test_laplace_kl_closed_form(oracle):
    # test/likelihoods/laplace.jl:6-9
    rng = np.random.default_rng(0)
    mu = rng.uniform(0.1, 1.0, size=5)
    beta = 0.8
    lam = 1 / (2 * beta) ** 2
    lik = oracle.laplace(beta)
    ref = np.sum(np.log(2 * lam) / 2 - np.log(2 * np.pi) / 2 - np.log(lam) / 2 + special.gammaln(0.5) + lam / mu)
    assert oracle.aux_kl(lik, np.zeros(5), mu) == pytest.approx(ref, rel=1e-13)


def test_expected_values_match_sampled_means(oracle):
    """expected_auglik_precision == E_q[auglik_precision] (tvmean, ntdist.jl:63-65): Monte-Carlo check of
    the categorical / poisson joint means (polyagammanegativemultinomial.jl:41-49, polyagammapoisson.jl:35-41)
    using the conditional sampler with f fixed so that the conditional equals q."""
    O = oracle
    lik = O.poisson(4.0)
    n = 60000
    f = np.full(n, 0.7)
    y = np.full(n, 2, dtype=np.int32)
    d = O.aux_sample(lik, y, f, seed=3)
    lam = 4.0 / (1 + np.exp(0.7))
    assert d["n"].mean() == pytest.approx(lam, abs=4 * np.sqrt(lam / n))
    assert d["omega"].mean() == pytest.approx(O.pg_mean(2 + lam, 0.7), abs=5e-3)
    lik = O.categorical(np.zeros(3))
    f = np.tile(np.array([0.3, -0.5, 1.0]), (n, 1))
    y = np.tile(np.array([0, 1, 0], dtype=np.uint8), (n, 1))
    d = O.aux_sample(lik, y, f, seed=4)
    p = 1 / (1 + np.exp(-f[0])) / 3
    nbar = p / (1 - p.sum())
    assert np.allclose(d["n"].mean(axis=0), nbar, atol=0.02)
    for k in range(3):
        assert d["omega"][:, k].mean() == pytest.approx(O.pg_mean(y[0, k] + nbar[k], abs(f[0, k])), abs=6e-3)


def test_cavi_pass_matches_dense_numpy(oracle):
    """The sparse pass (a11+a9+a10+a12) against a direct float64 numpy evaluation of the same formulas."""
    O = oracle
    rng = np.random.default_rng(7)
    N, M = 300, 16
    Phi = rng.normal(size=(N, M)).astype(np.float32) / 4
    W = rng.normal(size=(M, M))
    W = (W + W.T) / 20
    alpha = rng.normal(size=M)
    kd = 1.0 + rng.uniform(size=N)
    y = (rng.uniform(size=N) < 0.5).astype(np.uint8)
    G, g, pts = O.cavi_pass(O.bernoulli(), Phi, kd, y, W, alpha, want_points=True)
    P = Phi.astype(np.float64)
    mu = P @ alpha
    var = kd - np.einsum("ia,ab,ib->i", P, W, P)
    c = np.sqrt(mu ** 2 + var)
    gam = np.tanh(c / 2) / (2 * c)
    bet = (y.astype(float) - 0.5)
    assert np.allclose(pts["mu"][:, 0], mu, rtol=1e-12, atol=1e-14)
    assert np.allclose(pts["var"][:, 0], var, rtol=1e-12, atol=1e-14)
    assert np.allclose(pts["gamma"][0], gam, rtol=1e-12)
    assert np.allclose(G[0], (P * gam[:, None]).T @ P, rtol=1e-12, atol=1e-14)
    assert np.allclose(g[0], P.T @ bet, rtol=1e-12, atol=1e-14)
    G2, g2 = O.accumulate(Phi, pts["beta"], pts["gamma"])
    assert np.allclose(G2, G, rtol=1e-13) and np.allclose(g2, g, rtol=1e-13)


def test_sparse_update_equals_dense_reference_update(oracle):
    """The whitened sparse update used by the product equals the reference's dense CAVI update
    (examples/bernoulli/script.jl:35-36) when Z = X (the only case the reference ever executes, F3)."""
    O = oracle
    rng = np.random.default_rng(9)
    n = 25
    x = np.linspace(-10, 10, n)
    K = np.exp(-0.5 * ((x[:, None] - x[None, :]) / 2.0) ** 2) + 1e-6 * np.eye(n)
    lam = rng.uniform(0.05, 0.25, size=n)
    h = rng.choice([-0.5, 0.5], size=n)
    S_ref = np.linalg.inv(np.linalg.inv(K) + np.diag(lam))  # script.jl:35
    m_ref = S_ref @ h  # :36 with zero prior mean
    Lc = np.linalg.cholesky(K)
    Phi = np.linalg.solve(Lc, K).T  # whitened features phi_i = L^-1 k_i   [N, M]
    G = (Phi * lam[:, None]).T @ Phi
    g = Phi.T @ h
    Sv, mv = O.gaussian_update(G[None], g[None])
    assert np.allclose(Lc @ Sv[0] @ Lc.T, S_ref, rtol=1e-6, atol=1e-9)
    assert np.allclose(Lc @ mv[0], m_ref, rtol=1e-6, atol=1e-9)


def test_dense_gibbs_step_is_an_exact_draw_from_the_reference_conditional(oracle):
    """The inverse-free step used on the device (Matheron's rule through B = I + D^1/2 K D^1/2) samples
    N(mu, Sigma) with the reference's mu, Sigma (examples/bernoulli/script.jl:82-84): for a fixed Omega the map
    z -> f is affine, so its mean and covariance are checked exactly (no Monte-Carlo)."""
    O = oracle
    rng = np.random.default_rng(5)
    n = 12
    x = np.linspace(-3, 3, n)
    K = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2) + 1e-6 * np.eye(n)
    Lk = np.linalg.cholesky(K)
    gamma = rng.uniform(0.1, 2.0, size=n)
    beta = rng.normal(size=n)
    mu0 = rng.normal(size=n) * 0.3
    sg = np.sqrt(gamma)
    B = np.eye(n) + sg[:, None] * K * sg[None, :]
    Binv = np.linalg.inv(B)
    # f = f0 + K D^1/2 B^-1 (beta/sg - sg f0 - z2),  f0 = mu0 + Lk z1  => affine in (z1, z2)
    A0 = K @ (sg[:, None] * Binv)
    J1 = (np.eye(n) - A0 * sg[None, :]) @ Lk
    J2 = -A0
    mean = (np.eye(n) - A0 * sg[None, :]) @ mu0 + A0 @ (beta / sg)
    cov = J1 @ J1.T + J2 @ J2.T
    mu_ref, Sigma_ref = O.dense_conditional(K, beta, gamma, mu0)
    assert np.allclose(mean, mu_ref, rtol=1e-7, atol=1e-9)
    assert np.allclose(cov, Sigma_ref, rtol=1e-6, atol=1e-9)


# ------------------------------------------------------------------------------------------------
# Tightening what can be tightened on the unpinned oracle: constructions that do not share a
# line of code with the oracle's samplers.
# ------------------------------------------------------------------------------------------------
def _pg_from_definition(b, c, n, rng, terms=400):
    """PG(b, c) straight from its definition (Polson, Scott & Windle 2013, eq. 2; the law polyagamma.jl:1-20 documents):
    omega = 1/(2 pi^2) sum_k g_k / ((k - 1/2)^2 + c^2 / (4 pi^2)),  g_k ~ Gamma(b, 1) iid -- numpy's Gamma sampler, numpy's
    bit generator, truncated after `terms` terms with the tail replaced by its mean (relative tail mass ~ 1/(pi^2 terms))."""
    k = np.arange(1, terms + 1)
    den = (k - 0.5) ** 2 + c * c / (4 * np.pi ** 2)
    g = rng.gamma(b, 1.0, size=(n, terms))
    ktail = np.arange(terms + 1, 200_000)
    tail = b * np.sum(1.0 / ((ktail - 0.5) ** 2 + c * c / (4 * np.pi ** 2)))
    return (g / den).sum(1) / (2 * np.pi ** 2) + tail / (2 * np.pi ** 2)


@pytest.mark.parametrize("b,c", [(1, 0.0), (1, 2.0), (3, 0.0), (3, 2.5), (3, 3.2), (1.2, 3.2), (1, 9.0), (2, 60.0)])
def test_rand_pg_against_independent_construction_from_the_definition(oracle, b, c):
    """Two-sample test of the oracle's Devroye sampler (polyagamma.jl:121-257 restated) against the infinite-Gamma-sum
    definition of the law: the six (b, c) pairs of the reference's own test (test/SpecialDistributions/polyagamma.jl:30)
    plus c = 9 and c = 60 (both truncated-inverse-Gaussian branches)."""
    n = 20_000
    x = oracle.rand_pg(b, c, n, seed=1234)
    ref = _pg_from_definition(b, c, n, np.random.default_rng(4321))
    mean = b / (2 * c) * np.tanh(c / 2) if c else b / 4
    assert ref.mean() == pytest.approx(mean, rel=0.02)  # the construction itself is sound
    assert stats.ks_2samp(x, ref).pvalue > 1e-3
    assert abs(x.mean() - ref.mean()) < 5 * np.sqrt((x.var() + ref.var()) / n)
    assert x.var() == pytest.approx(ref.var(), rel=0.1)


def test_negative_multinomial_counts_against_numpy_construction(oracle):
    """The categorical Gibbs counts (negativemultinomial.jl:35-45: theta ~ Gamma(x0 = 1, 1/p0 - 1), n_k ~ Poisson(p_k
    theta / (1 - p0))) against a numpy construction of NegativeMultinomial(1, p): marginally n_k ~ Geometric-type
    NegBin(1, p0 / (p0 + p_k)) and sum_k n_k ~ NegBin(1, p0); two-sample chi-square on the pooled counts."""
    O = oracle
    L = 4
    lik = O.categorical(np.array([0.1, -0.2, 0.3, 0.0]))
    f = np.array([0.4, -1.0, 0.2, 1.3])
    n = 40_000
    fs = np.tile(f, (n, 1))
    y = np.zeros((n, L), dtype=np.uint8)
    d = O.aux_sample(lik, y, fs, seed=77)
    counts = d["n"]
    theta = np.exp(lik.logtheta[:L]) if hasattr(lik, "logtheta") else np.exp(np.array([0.1, -0.2, 0.3, 0.0]))
    p = theta / theta.sum() / (1 + np.exp(-f))
    p0 = 1 - p.sum()
    rng = np.random.default_rng(99)
    th = rng.gamma(1.0, 1 / p0 - 1, size=n)
    ref = rng.poisson(p[None, :] * th[:, None] / (1 - p0))
    for k in range(L):
        hi = int(max(counts[:, k].max(), ref[:, k].max()))
        a = np.bincount(counts[:, k], minlength=hi + 1).astype(float)
        b = np.bincount(ref[:, k], minlength=hi + 1).astype(float)
        keep = (a + b) > 10
        chi = ((a[keep] - b[keep]) ** 2 / (a[keep] + b[keep])).sum()
        assert chi < stats.chi2(keep.sum()).ppf(0.9999), k
        # closed form: E n_k = x0 p_k / p0 (negativemultinomial.jl:54)
        assert counts[:, k].mean() == pytest.approx(p[k] / p0, rel=0.05)
    tot = counts.sum(1)
    assert tot.mean() == pytest.approx((1 - p0) / p0, rel=0.05)
    assert stats.ks_2samp(tot, ref.sum(1)).pvalue > 1e-3


def test_oracle_pins_pass_under_address_and_ub_sanitizers():
    """The plain-C restatement built with -fsanitize=address,undefined (oracle/Makefile, libagpl_oracle_asan.so) runs
    the sampler / operator / sweep entry points in a child process with the sanitizer runtime preloaded: any
    out-of-bounds access, use-after-free or undefined shift / overflow in the checker aborts the child."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "-s", "libagpl_oracle_asan.so"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan_rt = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(asan_rt):
        pytest.skip("no libasan runtime in this image")
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
from oracle import oracle as O
import ctypes as C
O._lib = None
O.build = lambda force=False: os.path.join(%r, "oracle", "libagpl_oracle_asan.so")
rng = np.random.default_rng(3)
n = 300
for lik in (O.bernoulli(), O.negbinomial(15.0), O.negbinomial(5.5), O.studentt(3.0, 1.5), O.poisson(10.0), O.laplace(1.0),
            O.categorical(np.zeros(3)), O.categorical(np.zeros(3), bijective=True), O.heterogauss(5.0)):
    L = lik.nlatent
    f = rng.normal(size=(n, L)) if L > 1 else rng.normal(size=n)
    if lik.kind == O.BERNOULLI:
        y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    elif lik.kind in (O.NEGBINOMIAL, O.POISSON):
        y = rng.poisson(3.0, size=n).astype(np.int32)
    elif lik.kind in (O.CATEGORICAL, O.CATEGORICAL_BIJ):
        K = L + (1 if lik.kind == O.CATEGORICAL_BIJ else 0)
        y = (rng.integers(0, K, size=n)[:, None] == np.arange(L)[None, :]).astype(np.uint8)
    else:
        y = rng.normal(size=n)
    d = O.aux_sample(lik, y, f, seed=5, sweep=2, stats=True, i0=7)
    mu = rng.normal(size=f.shape); var = rng.uniform(0.1, 1.0, size=f.shape)
    q1, q2, q3 = O.aux_posterior(lik, y, mu, var)
    O.expected_potential_precision(lik, y, q1, q2, mu_g=(mu[:, 1] if lik.kind == O.HETEROGAUSS else None))
    O.potential_precision(lik, y, d["omega"], d.get("n"), fg=(f if lik.kind == O.HETEROGAUSS else None))
    O.aux_kl(lik, y, q1, q2) if lik.kind != O.CATEGORICAL else None
    M = 32
    Phi = (rng.normal(size=(n, M)) * 0.3).astype(np.float32)
    O.cavi_pass(lik, Phi, np.full(n, 0.5), y, -np.tile(np.eye(M), (L, 1, 1)) * 0.5, np.zeros((L, M)))
    O.gibbs_pass(lik, Phi, np.full(n, 0.5), y, rng.normal(size=(L, M)), seed=5, sweep=1, i0=11)
O.rand_pg(1.2, 3.2, 500, seed=1); O.rand_pg(3, 0.0, 500, seed=1); O.rand_gamma(0.3, 1.0, 500, seed=1)
O.rand_poisson(0.7, 500, seed=1); O.rand_poisson(40.0, 500, seed=1); O.rand_invgaussian(0.5, 0.5, 500, seed=1)
O.synth_y(O.negbinomial(15.0), 1, 3, 200); O.synth_x(1, 0, 10)
[O.pg_logpdf(1, 2.0, x) for x in (1e-4, 0.01, 0.3, 5.0)]
print("ASAN_RUN_OK")
""" % (root, root)
    env = dict(os.environ, LD_PRELOAD=asan_rt + (":" + ubsan_rt if os.path.exists(ubsan_rt) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ASAN_RUN_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_conditioning_budget_of_an_unwhitened_on_the_fly_marginal_pass():
    """SURVEY.md 8(f3) / VERDICT r1 item 9: generating K_ZX tiles in-kernel means the marginal pass has to run as
    T = (U L^-1) K_ZX instead of T = U Phi with the stored whitened features Phi = L^-1 K_ZX.  This is its conditioning
    budget on the bench's own kernel (SE, lengthscale 1.5 x inducing spacing, jitter 1e-8), with the device's operand
    format emulated exactly (every float32 operand carried as hi + lo float16, the lo x lo product dropped): the
    un-whitened product loses more than an order of magnitude of accuracy in sum_a T[a,n]^2 (the variance projection),
    which puts it AT the 1e-5 natural-parameter bar with no margin where the whitened path clears it by ~10x.
    That is why row (f3) is not built: the images it would save cost the path its tolerance."""
    import scipy.linalg as sla

    rng = np.random.default_rng(0)
    M, n = 256, 1500
    z = np.linspace(-10, 10, M)
    ell = 1.5 * (z[1] - z[0])
    Kzz = np.exp(-0.5 * ((z[:, None] - z[None, :]) / ell) ** 2) + 1e-8 * np.eye(M)
    Linv = sla.solve_triangular(np.linalg.cholesky(Kzz), np.eye(M), lower=True)
    x = rng.uniform(-10, 10, n)
    K = np.exp(-0.5 * ((z[:, None] - x[None, :]) / ell) ** 2)
    Phi = Linv @ K
    G = (Phi * (rng.uniform(0.05, 0.25, n) * 5000)) @ Phi.T  # a posterior as sharp as N = 1e7 points make it
    U = sla.solve_triangular(np.linalg.cholesky(np.eye(M) + G), np.eye(M), lower=True)

    def split(a):
        a = a.astype(np.float32).astype(np.float64)
        hi = a.astype(np.float16).astype(np.float64)
        return hi, (a - hi).astype(np.float16).astype(np.float64)

    def prod3(A, B):
        Ah, Al = split(A)
        Bh, Bl = split(B)
        return Ah @ Bh + Ah @ Bl + Al @ Bh

    q_ref = ((U @ Phi) ** 2).sum(0)
    err_w = np.abs((prod3(U, Phi) ** 2).sum(0) - q_ref).max() / q_ref.max()
    err_u = np.abs((prod3(U @ Linv, K) ** 2).sum(0) - q_ref).max() / q_ref.max()
    assert err_w < 5e-6            # the shipped (whitened) operand pair
    assert err_u > 10 * err_w      # the un-whitened pair: measured 18-30x worse (|U L^-1| ~ 12 against |U| ~ 0.4)
    assert err_u > 1e-5            # i.e. at / above the bar before any accumulation error is added


def test_pg_draw_index_beyond_sixteen_bits(oracle):
    """polyagamma.jl:129-134 sums any integer b.  Draw j of a point keeps 16 bits of the Philox sub-stream id (j mod 65535) and, from
    j = 65535 on, starts its block counter at (j div 65535) << 20: (i) the draws of b < 65535 are those of the round 2-5 layout (the
    golden fixtures pin them; here: the b-th draw is the difference of consecutive sums); (ii) draw 65535 reuses the id of draw 0
    but is NOT draw 0; (iii) the sample mean at b = 70 000 sits on b / (2c) tanh(c / 2) (polyagamma.jl:25-31); (iv) b >= 2^22 is
    refused (NaN), as the device build refuses it."""
    O = oracle
    c, seed = 1.0, 5
    s = {b: O.rand_pg(float(b), c, 4, seed=seed) for b in (1, 65534, 65535, 65536, 65537)}
    d0, d65534, d65535, d65536 = s[1], s[65535] - s[65534], s[65536] - s[65535], s[65537] - s[65536]
    assert (d65534 > 0).all() and (d65535 > 0).all() and (d65536 > 0).all()
    assert not np.allclose(d65535, d0, rtol=1e-6) and not np.allclose(d65536, O.rand_pg(2.0, c, 4, seed=seed) - d0, rtol=1e-6)
    big = O.rand_pg(70_000.0, c, 8, seed=seed)
    mean, sd = 70_000 / (2 * c) * np.tanh(c / 2), np.sqrt(70_000 * 0.0363)  # var PG(1, 1) ~ 0.036
    assert np.abs(big - mean).max() < 5 * sd
    assert np.isnan(O.rand_pg(4194304.0, c, 1, seed=seed)[0]) and np.isfinite(O.rand_pg(4194303.0, 30.0, 1, seed=seed)[0])
