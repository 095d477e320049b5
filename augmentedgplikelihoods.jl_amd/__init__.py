"""MI355X-native drop-in for the inner inference loop of AugmentedGPLikelihoods.jl.

Host side above the C ABI (include/agpl.h): the reference's Likelihood operator surface --
``init_aux_variables, init_aux_posterior, aux_sample(!), aux_posterior(!), auglik_potential,
auglik_precision, expected_auglik_potential, expected_auglik_precision, *_and_precision, logtilt,
expected_logtilt, aux_kldivergence, nlatent`` (exports of src/AugmentedGPLikelihoods.jl:18-30; Julia's
``f!`` is spelled ``f_``) -- over torch CUDA tensors, plus the sparse sweep drivers (``SparseCAVI``,
``SparseGibbs``) that restate ``cavi!`` / ``gibbs_sample`` of examples/bernoulli/script.jl:29-39,76-87 in
the sparse form of docs/src/index.md:154-163.

torch is used for device memory, streams and torch.distributed only; every operator runs in libagpl.so
(hand-written HIP for gfx950).  There is no CPU fallback: without the built library imports succeed but
the first operator call raises.
"""
from __future__ import annotations

from . import _ffi
from ._ffi import AGPLError, ArgumentError, DomainError, PosDefException, build
from .likelihoods import (BernoulliLikelihood, CategoricalLikelihood, HeteroscedasticGaussianLikelihood,
                          LaplaceLikelihood, NegativeBinomialLikelihood, PoissonLikelihood, StudentTLikelihood,
                          nlatent)
from .operators import (AuxPosterior, Context, TupleVector, aug_loglik, aux_prior_logpdf, auglik_potential,
                        auglik_potential_and_precision, auglik_precision, aux_kldivergence, aux_posterior,
                        aux_posterior_, aux_sample, aux_sample_, default_context, expected_auglik_potential,
                        expected_auglik_potential_and_precision, expected_auglik_precision, expected_aug_loglik, expected_logtilt,
                        init_aux_posterior, init_aux_variables, logtilt, rand_polyagamma)
from . import sparse
from .sparse import (DenseGibbs, SparseCAVI, SparseGibbs, exchange_natural_parameters, se_features, shard_range, synth_xy,
                     whiten_features)

__all__ = [
    "AGPLError", "ArgumentError", "DomainError", "PosDefException", "build",
    "BernoulliLikelihood", "NegativeBinomialLikelihood", "StudentTLikelihood", "CategoricalLikelihood",
    "PoissonLikelihood", "LaplaceLikelihood", "HeteroscedasticGaussianLikelihood", "nlatent",
    "Context", "default_context", "TupleVector", "AuxPosterior",
    "init_aux_variables", "init_aux_posterior", "aux_sample", "aux_sample_", "aux_posterior", "aux_posterior_",
    "auglik_potential", "auglik_precision", "auglik_potential_and_precision",
    "expected_auglik_potential", "expected_auglik_precision", "expected_auglik_potential_and_precision",
    "logtilt", "expected_logtilt", "aux_kldivergence", "aug_loglik", "expected_aug_loglik", "aux_prior_logpdf", "rand_polyagamma",
    "SparseCAVI", "SparseGibbs", "DenseGibbs", "se_features", "whiten_features", "synth_xy", "shard_range",
    "exchange_natural_parameters",
]
