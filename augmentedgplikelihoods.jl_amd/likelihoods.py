"""Likelihood types = constructor arguments of the reference's likelihoods (src/likelihoods/*.jl and the
re-exported GPLikelihoods types).  They only carry parameters; all arithmetic is in libagpl.so."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from ._ffi import LikDesc

(KIND_BERNOULLI, KIND_NEGBINOMIAL, KIND_STUDENTT, KIND_CATEGORICAL, KIND_CATEGORICAL_BIJ, KIND_POISSON,
 KIND_LAPLACE, KIND_HETEROGAUSS) = range(8)


class AbstractLikelihood:
    kind: int = -1
    _nlatent: int = 1

    def _params(self):
        return ()

    def _logtheta(self):
        return None

    def desc(self) -> LikDesc:
        d = LikDesc()
        d.kind, d.nlatent = self.kind, self._nlatent
        p = list(self._params()) + [0.0] * 4
        for i in range(4):
            d.p[i] = float(p[i])
        lt = self._logtheta()
        if lt is not None:
            self._lt_keep = np.ascontiguousarray(lt, dtype=np.float64)
            d.logtheta = self._lt_keep.ctypes.data_as(C.POINTER(C.c_double))
        return d

    # dtype of y on the device: 'u8' | 'i32' | 'real'
    ykind = "real"


def nlatent(lik: AbstractLikelihood) -> int:
    """nlatent(lik): src/generic.jl:87, src/likelihoods/categorical.jl:46-47,
    src/likelihoods/heteroscedasticgaussian.jl:11."""
    return lik._nlatent


@dataclass
class BernoulliLikelihood(AbstractLikelihood):
    """BernoulliLikelihood(LogisticLink()) -- src/likelihoods/bernoulli.jl."""
    kind = KIND_BERNOULLI
    ykind = "u8"


@dataclass
class NegativeBinomialLikelihood(AbstractLikelihood):
    """NegativeBinomialLikelihood(NBParamFailure(r), LogisticLink()) -- src/likelihoods/negativebinomial.jl."""
    failures: float = 1.0
    kind = KIND_NEGBINOMIAL
    ykind = "i32"

    def _params(self):
        return (self.failures,)


@dataclass
class StudentTLikelihood(AbstractLikelihood):
    """StudentTLikelihood(nu, sigma) -- src/likelihoods/studentt.jl:14-21."""
    nu: float = 3.0
    sigma: float = 1.0
    kind = KIND_STUDENTT

    def _params(self):
        return (self.nu, self.sigma)


@dataclass
class CategoricalLikelihood(AbstractLikelihood):
    """CategoricalLikelihood(LogisticSoftMaxLink(logtheta)) or, with ``bijective=True``,
    CategoricalLikelihood(BijectiveSimplexLink(LogisticSoftMaxLink(logtheta))) --
    src/likelihoods/categorical.jl:6-47.  ``CategoricalLikelihood(nclass)`` = zeros(nclass) (:10)."""
    logtheta: object = 2
    bijective: bool = False
    ykind = "u8"

    def __post_init__(self):
        if isinstance(self.logtheta, (int, np.integer)):
            self.logtheta = np.zeros(int(self.logtheta))
        self.logtheta = np.asarray(self.logtheta, dtype=np.float64)
        self.kind = KIND_CATEGORICAL_BIJ if self.bijective else KIND_CATEGORICAL
        self._nlatent = len(self.logtheta) - (1 if self.bijective else 0)

    def _logtheta(self):
        return self.logtheta


@dataclass
class PoissonLikelihood(AbstractLikelihood):
    """PoissonLikelihood(ScaledLogistic(lambda)) -- src/likelihoods/poisson.jl."""
    lam: float = 1.0
    kind = KIND_POISSON
    ykind = "i32"

    def _params(self):
        return (self.lam,)


@dataclass
class LaplaceLikelihood(AbstractLikelihood):
    """LaplaceLikelihood(beta) -- src/likelihoods/laplace.jl:13-17."""
    beta: float = 1.0
    kind = KIND_LAPLACE

    def _params(self):
        return (self.beta,)


@dataclass
class HeteroscedasticGaussianLikelihood(AbstractLikelihood):
    """HeteroscedasticGaussianLikelihood(InvScaledLogistic(lambda)) --
    src/likelihoods/heteroscedasticgaussian.jl."""
    lam: float = 1.0
    kind = KIND_HETEROGAUSS
    _nlatent = 2

    def _params(self):
        return (self.lam,)
