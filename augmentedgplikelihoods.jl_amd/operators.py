"""The reference's Likelihood operator surface over torch CUDA tensors, backed by libagpl.so.

Containers mirror the reference's: ``TupleVector`` is the SoA named-field container the reference gets from
TupleVectors.jl (``TupleVector(; ω=Vector)`` bernoulli.jl:4), ``AuxPosterior`` stands for the ``For`` of
variational factors whose ``.inds`` is a TupleVector of parameters (bernoulli.jl:7-11).  Multi-latent
per-point fields are tensors of shape [N, L] (C-order) = the reference's flat [L, N] column-major storage
(categorical.jl:52-70); potentials / precisions come back as a tuple of L vectors of N (utils.jl:24).
"""
from __future__ import annotations

import ctypes as C

from . import _ffi
from .likelihoods import (KIND_BERNOULLI, KIND_CATEGORICAL, KIND_CATEGORICAL_BIJ, KIND_HETEROGAUSS, KIND_LAPLACE,
                          KIND_NEGBINOMIAL, KIND_POISSON, KIND_STUDENTT, AbstractLikelihood)


def _torch():
    import torch

    return torch


class Context:
    """One per GPU; replaces GLOBAL_RNG / the ``rng`` argument (src/generic.jl:1-3): Philox key = seed,
    and a draw counter (``sweep``) that advances with every aux_sample call."""

    def __init__(self, device=None, seed: int = 0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("augmentedgplikelihoods.jl_amd needs a ROCm GPU (MI355X); there is no CPU path")
        idx = torch.cuda.current_device() if device is None else torch.device(device).index or 0
        self.device = torch.device("cuda", idx)
        self._h = C.c_void_p()
        rc = _ffi.lib().agpl_ctx_create(C.byref(self._h), C.c_int32(idx), C.c_uint64(seed))
        if rc != 0:
            raise _ffi.AGPLError(rc, "agpl_ctx_create failed")
        self._bound = None
        self.seed = seed
        self.sweep = 0
        self.point_offset = 0

    def bind(self):
        """Enqueue on torch's current stream for this device."""
        s = _torch().cuda.current_stream(self.device).cuda_stream
        if s != self._bound:
            _ffi.check(self._h, _ffi.lib().agpl_ctx_set_stream(self._h, C.c_void_p(s)))
            self._bound = s
        return self._h

    def set_seed(self, seed: int):
        _ffi.check(self._h, _ffi.lib().agpl_ctx_set_seed(self._h, C.c_uint64(seed)))
        self.seed = seed
        self.sweep = 0

    def next_sweep(self) -> int:
        """The next unused draw counter of this context (the ``sweep`` word of every Philox stream key).  Every
        sampler entry point -- aux_sample, rand_polyagamma, SparseGibbs, DenseGibbs -- takes its indices from here,
        so two users of one Context never replay each other's streams."""
        s = self.sweep
        self.sweep += 1
        return s

    def set_point_offset(self, i0: int):
        """Global index of this rank's local point 0: per-point streams are keyed (seed, i0 + i, sweep), so N sharded
        over ranks (same seed everywhere) draws what a single process would (agpl_ctx_set_point_offset)."""
        _ffi.check(self._h, _ffi.lib().agpl_ctx_set_point_offset(self._h, C.c_int64(i0)))
        self.point_offset = i0

    def synchronize(self):
        _ffi.check(self._h, _ffi.lib().agpl_ctx_synchronize(self.bind()))

    def call(self, name, *args):
        h = self.bind()
        _ffi.check(h, getattr(_ffi.lib(), name)(h, *args))

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().agpl_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass


_default = {}


def default_context(device=None) -> Context:
    torch = _torch()
    idx = torch.cuda.current_device() if device is None else torch.device(device).index or 0
    if idx not in _default:
        _default[idx] = Context(idx, seed=0)
    return _default[idx]


class TupleVector:
    """SoA container with named fields of equal length (TupleVectors.jl as used by the reference)."""

    def __init__(self, **fields):
        self._fields = dict(fields)

    def __getattr__(self, k):
        f = self.__dict__.get("_fields", {})
        if k in f:
            return f[k]
        raise AttributeError(k)

    def keys(self):
        return self._fields.keys()

    def __getitem__(self, k):
        return self._fields[k]

    def __len__(self):
        return next(iter(self._fields.values())).shape[0]

    def __repr__(self):
        return "TupleVector(" + ", ".join(f"{k}={tuple(v.shape)}" for k, v in self._fields.items()) + ")"


class AuxPosterior:
    """The ``For(TupleVector(...)) do φ ... end`` object of init_aux_posterior: ``.inds`` is a 1-tuple holding
    the TupleVector of variational parameters (``only(qΩ.inds)`` in the reference)."""

    def __init__(self, lik, params: TupleVector):
        self.lik = lik
        self.inds = (params,)

    def __len__(self):
        return len(self.inds[0])


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _prep(t, dtype, name):
    torch = _torch()
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA tensor (device arrays only; no host fallback)")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _ydtype(lik, real):
    torch = _torch()
    return {"u8": torch.uint8, "i32": torch.int32, "real": real}[lik.ykind]


def _prep_y(lik, y, real):
    torch = _torch()
    if y is not None and isinstance(y, torch.Tensor) and y.dtype == torch.bool:
        y = y.to(torch.uint8)
    return _prep(y, _ydtype(lik, real), "y")


def _npoints(lik, t):
    L = lik._nlatent
    if L == 1:
        return t.numel()
    if t.dim() != 2 or t.shape[1] != L:
        raise _ffi.ArgumentError(-1, f"expected a [N, {L}] tensor, got {tuple(t.shape)}")
    return t.shape[0]


_HAS_N = (KIND_CATEGORICAL, KIND_CATEGORICAL_BIJ, KIND_POISSON, KIND_HETEROGAUSS)


# ------------------------------------------------------------------------------------------ Gibbs half
def init_aux_variables(lik: AbstractLikelihood, n: int, ctx: Context | None = None) -> TupleVector:
    """init_aux_variables(rng, lik, n): bernoulli.jl:3-5 (PG(1,0) draws), negativebinomial.jl:10-12,
    studentt.jl:33-35 (the reference draws Gamma(); any positive start works since aux_sample! overwrites),
    categorical.jl:52-57, poisson.jl:14-18.  Here: PG(1, 0) draws for ω, zeros for n."""
    torch = _torch()
    ctx = ctx or default_context()
    L = lik._nlatent
    shape = (n,) if (L == 1 or lik.kind == KIND_HETEROGAUSS) else (n, L)
    om = torch.empty(shape, dtype=torch.float64, device=ctx.device)
    rand_polyagamma(1.0, 0.0, om, ctx=ctx)
    if lik.kind in _HAS_N:
        return TupleVector(ω=om, n=torch.zeros(shape, dtype=torch.int64, device=ctx.device))
    return TupleVector(ω=om)


def rand_polyagamma(b: float, c: float, out, ctx: Context | None = None, sweep: int | None = None,
                    stats: bool = False):
    """rand(rng, PolyaGamma(b, c), n) -- polyagamma.jl:121-126; fills ``out`` (float64 CUDA tensor)."""
    torch = _torch()
    ctx = ctx or default_context()
    if sweep is None:
        sweep = ctx.next_sweep()
    n = out.numel()
    nuni = torch.empty(n, dtype=torch.int32, device=out.device) if stats else None
    nterms = torch.empty(n, dtype=torch.int32, device=out.device) if stats else None
    ctx.call("agpl_rand_polyagamma", C.c_double(b), C.c_double(c), C.c_int64(n), C.c_uint32(sweep), _ptr(out),
             _ptr(nuni), _ptr(nterms))
    return (out, nuni, nterms) if stats else out


def aux_sample_(Ω: TupleVector, lik, y, f, ctx: Context | None = None, sweep: int | None = None,
                stats: bool = False):
    """aux_sample!(rng, Ω, lik, y, f) -- src/generic.jl:5-12: Ωᵢ ← draw from aux_full_conditional(lik, yᵢ, fᵢ),
    in place; returns Ω."""
    torch = _torch()
    ctx = ctx or default_context()
    f = _prep(f, torch.float64, "f")
    y = _prep_y(lik, y, torch.float64)
    n = _npoints(lik, f)
    om = Ω.ω
    if om.dtype != torch.float64 or not om.is_contiguous():
        raise TypeError("Ω.ω must be a contiguous float64 CUDA tensor")
    nn = Ω.n if lik.kind in _HAS_N else None
    if sweep is None:
        sweep = ctx.next_sweep()
    nuni = torch.empty(n, dtype=torch.int32, device=f.device) if stats else None
    nterms = torch.empty(n, dtype=torch.int32, device=f.device) if stats else None
    d = lik.desc()
    ctx.call("agpl_aux_sample", C.byref(d), C.c_int64(n), _ptr(y), _ptr(f), _ptr(om), _ptr(nn),
             C.c_uint32(sweep), _ptr(nuni), _ptr(nterms))
    if stats:
        return Ω, nuni, nterms
    return Ω


def aux_sample(lik, y, f, ctx: Context | None = None, sweep: int | None = None):
    """aux_sample(rng, lik, y, f) -- src/generic.jl:14-20."""
    torch = _torch()
    ctx = ctx or default_context()
    L = lik._nlatent
    n = f.numel() // L
    shape = (n,) if (L == 1 or lik.kind == KIND_HETEROGAUSS) else (n, L)
    fields = dict(ω=torch.empty(shape, dtype=torch.float64, device=ctx.device))
    if lik.kind in _HAS_N:
        fields["n"] = torch.zeros(shape, dtype=torch.int64, device=ctx.device)
    return aux_sample_(TupleVector(**fields), lik, y, f, ctx=ctx, sweep=sweep)


def _transposed(t, L):
    """[L][N] buffer -> tuple of L vectors (the reference's Tuple of per-latent Vectors)."""
    return tuple(t[k] for k in range(L))


def auglik_potential_and_precision(lik, Ω: TupleVector, y, f=None, ctx: Context | None = None):
    """auglik_potential_and_precision(lik, Ω, y, f) -- src/generic.jl:64-66 (+ the per-likelihood methods
    cited in include/agpl.h)."""
    torch = _torch()
    ctx = ctx or default_context()
    L = lik._nlatent
    y = _prep_y(lik, y, torch.float64)
    om = _prep(Ω.ω, torch.float64, "Ω.ω")
    nn = Ω.n if lik.kind in _HAS_N else None
    fg = _prep(f, torch.float64, "f") if lik.kind == KIND_HETEROGAUSS else None
    n = om.shape[0]
    beta = torch.empty((L, n), dtype=torch.float64, device=om.device)
    gamma = torch.empty((L, n), dtype=torch.float64, device=om.device)
    d = lik.desc()
    ctx.call("agpl_potential_precision", C.byref(d), C.c_int64(n), _ptr(y), _ptr(om), _ptr(nn), _ptr(fg),
             _ptr(beta), _ptr(gamma))
    return _transposed(beta, L), _transposed(gamma, L)


def auglik_potential(lik, Ω, y, f=None, ctx=None):
    return auglik_potential_and_precision(lik, Ω, y, f, ctx)[0]


def auglik_precision(lik, Ω, y, f=None, ctx=None):
    return auglik_potential_and_precision(lik, Ω, y, f, ctx)[1]


# ------------------------------------------------------------------------------------------ CAVI half
_POSTERIOR_FIELDS = {
    KIND_BERNOULLI: ("c",), KIND_NEGBINOMIAL: ("y", "c"), KIND_STUDENTT: ("β",),
    KIND_CATEGORICAL: ("y", "c", "p"), KIND_CATEGORICAL_BIJ: ("y", "c", "p"), KIND_POISSON: ("y", "c", "λ"),
    KIND_LAPLACE: ("μ",), KIND_HETEROGAUSS: ("c", "λ", "ψ"),
}


def init_aux_posterior(lik, n: int, dtype=None, ctx: Context | None = None) -> AuxPosterior:
    """init_aux_posterior(T, lik, n): zero-initialised variational parameters (bernoulli.jl:7-11,
    negativebinomial.jl:14-18, studentt.jl:37-44, categorical.jl:59-70, poisson.jl:20-24, laplace.jl:33-38,
    heteroscedasticgaussian.jl:22-26).  T defaults to Float64 (generic.jl:36-38)."""
    torch = _torch()
    ctx = ctx or default_context()
    dtype = dtype or torch.float64
    L = lik._nlatent
    shape = (n,) if (L == 1 or lik.kind == KIND_HETEROGAUSS) else (n, L)
    fields = {}
    for name in _POSTERIOR_FIELDS[lik.kind]:
        if name == "y":
            fields[name] = torch.zeros(shape, dtype=_ydtype(lik, dtype), device=ctx.device)
        else:
            fields[name] = torch.zeros(shape, dtype=dtype, device=ctx.device)
    return AuxPosterior(lik, TupleVector(**fields))


def _qf_parts(qf):
    """q(f) marginals as (mean, var): accepts a (mean, var) pair or any object with .mean/.var (or
    .loc/.scale like torch.distributions.Normal) -- the reference passes a Vector{Normal}."""
    if isinstance(qf, (tuple, list)) and len(qf) == 2:
        return qf[0], qf[1]
    if hasattr(qf, "loc") and hasattr(qf, "scale"):
        return qf.loc, qf.scale ** 2
    return qf.mean, qf.var


def aux_posterior_(qΩ: AuxPosterior, lik, y, qf, ctx: Context | None = None) -> AuxPosterior:
    """aux_posterior!(qΩ, lik, y, qf): optimal q(Ω) parameters from the marginals q(fᵢ) = N(μᵢ, σᵢ²), in place."""
    torch = _torch()
    ctx = ctx or default_context()
    φ = qΩ.inds[0]
    names = [k for k in _POSTERIOR_FIELDS[lik.kind] if k != "y"]
    real = φ[names[0]].dtype
    mu, var = _qf_parts(qf)
    mu, var = _prep(mu, real, "mean(qf)"), _prep(var, real, "var(qf)")
    y = _prep_y(lik, y, real)
    n = _npoints(lik, mu)
    outs = [φ[k] for k in names] + [None, None]
    d = lik.desc()
    ctx.call("agpl_aux_posterior", C.byref(d), C.c_int32(_ffi.F64 if real == torch.float64 else _ffi.F32),
             C.c_int64(n), _ptr(y), _ptr(mu), _ptr(var), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]))
    if "y" in φ.keys():  # negativebinomial.jl:28 `φ.y .= y`, categorical.jl:89,106, poisson.jl:36
        φ["y"].copy_(y.reshape(φ["y"].shape))
    qΩ._mu_g = mu[:, 1].contiguous() if lik.kind == KIND_HETEROGAUSS else None
    return qΩ


def aux_posterior(lik, y, qf, ctx: Context | None = None) -> AuxPosterior:
    """aux_posterior(lik, y, qf) -- src/generic.jl:22-24."""
    mu, _ = _qf_parts(qf)
    n = mu.numel() // lik._nlatent
    return aux_posterior_(init_aux_posterior(lik, n, dtype=mu.dtype, ctx=ctx), lik, y, qf, ctx=ctx)


def expected_auglik_potential_and_precision(lik, qΩ: AuxPosterior, y, qf=None, ctx: Context | None = None):
    """expected_auglik_potential_and_precision(lik, qΩ, y, qf) -- src/generic.jl:68-72."""
    torch = _torch()
    ctx = ctx or default_context()
    L = lik._nlatent
    φ = qΩ.inds[0]
    names = [k for k in _POSTERIOR_FIELDS[lik.kind] if k != "y"]
    q1 = φ[names[0]]
    q2 = φ[names[1]] if len(names) > 1 else None
    real = q1.dtype
    y = _prep_y(lik, y, real)
    n = q1.shape[0]
    mu_g = None
    if lik.kind == KIND_HETEROGAUSS:
        mu_g = _prep(_qf_parts(qf)[0], real, "mean(qf)")[:, 1].contiguous() if qf is not None else qΩ._mu_g
    beta = torch.empty((L, n), dtype=real, device=q1.device)
    gamma = torch.empty((L, n), dtype=real, device=q1.device)
    d = lik.desc()
    ctx.call("agpl_expected_potential_precision", C.byref(d),
             C.c_int32(_ffi.F64 if real == torch.float64 else _ffi.F32), C.c_int64(n), _ptr(y), _ptr(q1), _ptr(q2),
             _ptr(mu_g), _ptr(beta), _ptr(gamma))
    return _transposed(beta, L), _transposed(gamma, L)


def expected_auglik_potential(lik, qΩ, y, qf=None, ctx=None):
    return expected_auglik_potential_and_precision(lik, qΩ, y, qf, ctx)[0]


def expected_auglik_precision(lik, qΩ, y, qf=None, ctx=None):
    return expected_auglik_potential_and_precision(lik, qΩ, y, qf, ctx)[1]


# ------------------------------------------------------------------------------------------ ELBO terms
def logtilt(lik, Ω: TupleVector, y, f, ctx: Context | None = None) -> float:
    """logtilt(lik, Ω, y, f) -- src/generic.jl:40-46."""
    torch = _torch()
    ctx = ctx or default_context()
    f = _prep(f, torch.float64, "f")
    y = _prep_y(lik, y, torch.float64)
    out = C.c_double()
    d = lik.desc()
    nn = Ω.n if lik.kind in _HAS_N else None
    ctx.call("agpl_logtilt", C.byref(d), C.c_int64(_npoints(lik, f)), _ptr(y), _ptr(_prep(Ω.ω, torch.float64, "ω")),
             _ptr(nn), _ptr(f), C.byref(out))
    return out.value


def aug_loglik(lik, Ω: TupleVector, y, f, ctx: Context | None = None) -> float:
    """aug_loglik(lik, Ω, y, f) = logtilt + logdensity_def(aux_prior(lik, y), Ω) -- src/generic.jl:48-50; the
    heteroscedastic likelihood's own method heteroscedasticgaussian.jl:106-128 (f = fg [N, 2]).  Categorical: UnsupportedError
    (the reference's prior density is broken)."""
    torch = _torch()
    ctx = ctx or default_context()
    f = _prep(f, torch.float64, "f")
    y = _prep_y(lik, y, torch.float64)
    out = C.c_double()
    d = lik.desc()
    nn = Ω.n if lik.kind in _HAS_N else None
    ctx.call("agpl_aug_loglik", C.byref(d), C.c_int64(_npoints(lik, f)), _ptr(y), _ptr(_prep(Ω.ω, torch.float64, "ω")),
             _ptr(nn), _ptr(f), C.byref(out))
    return out.value


def aux_prior_logpdf(lik, Ω: TupleVector, y, ctx: Context | None = None) -> float:
    """logdensity_def(aux_prior(lik, y), Ω): the second term of aug_loglik (priors: bernoulli.jl:51-57,
    negativebinomial.jl:67-73, studentt.jl:85-91, poisson.jl:67-76, laplace.jl:90-96)."""
    torch = _torch()
    ctx = ctx or default_context()
    y = _prep_y(lik, y, torch.float64)
    ω = _prep(Ω.ω, torch.float64, "ω")
    nn = Ω.n if lik.kind in _HAS_N else None
    out = C.c_double()
    d = lik.desc()
    ctx.call("agpl_aux_prior_logpdf", C.byref(d), C.c_int64(ω.numel()), _ptr(y), _ptr(ω), _ptr(nn), C.byref(out))
    return out.value


def expected_aug_loglik(lik, qΩ: AuxPosterior, y, qf, ctx: Context | None = None) -> float:
    """expected_aug_loglik(lik, qΩ, y, qf) = expected_logtilt + aux_kldivergence -- src/generic.jl:52-54 (sign as coded);
    heteroscedastic: heteroscedasticgaussian.jl:130-145."""
    torch = _torch()
    ctx = ctx or default_context()
    φ = qΩ.inds[0]
    names = [k for k in _POSTERIOR_FIELDS[lik.kind] if k != "y"]
    q1 = _prep(φ[names[0]], torch.float64, "q1")
    q2 = _prep(φ[names[1]], torch.float64, "q2") if len(names) > 1 else None
    mu, var = _qf_parts(qf)
    mu, var = _prep(mu, torch.float64, "mean(qf)"), _prep(var, torch.float64, "var(qf)")
    y = _prep_y(lik, y, torch.float64)
    out = C.c_double()
    d = lik.desc()
    ctx.call("agpl_expected_aug_loglik", C.byref(d), C.c_int64(_npoints(lik, mu)), _ptr(y), _ptr(q1), _ptr(q2),
             _ptr(mu), _ptr(var), C.byref(out))
    return out.value


def expected_logtilt(lik, qΩ: AuxPosterior, y, qf, ctx: Context | None = None) -> float:
    """expected_logtilt(lik, qΩ, y, qf) -- src/api.jl:219-223."""
    torch = _torch()
    ctx = ctx or default_context()
    φ = qΩ.inds[0]
    names = [k for k in _POSTERIOR_FIELDS[lik.kind] if k != "y"]
    q1 = _prep(φ[names[0]], torch.float64, "q1")
    q2 = _prep(φ[names[1]], torch.float64, "q2") if len(names) > 1 else None
    mu, var = _qf_parts(qf)
    mu, var = _prep(mu, torch.float64, "mean(qf)"), _prep(var, torch.float64, "var(qf)")
    y = _prep_y(lik, y, torch.float64)
    out = C.c_double()
    d = lik.desc()
    ctx.call("agpl_expected_logtilt", C.byref(d), C.c_int64(_npoints(lik, mu)), _ptr(y), _ptr(q1), _ptr(q2),
             _ptr(mu), _ptr(var), C.byref(out))
    return out.value


def aux_kldivergence(lik, qΩ: AuxPosterior, y, ctx: Context | None = None) -> float:
    """aux_kldivergence(lik, qΩ, y) = KL(q(Ω) || aux_prior(lik, y)) -- src/generic.jl:56-62."""
    torch = _torch()
    ctx = ctx or default_context()
    φ = qΩ.inds[0]
    names = [k for k in _POSTERIOR_FIELDS[lik.kind] if k != "y"]
    q1 = _prep(φ[names[0]], torch.float64, "q1")
    q2 = _prep(φ[names[1]], torch.float64, "q2") if len(names) > 1 else None
    y = _prep_y(lik, y, torch.float64)
    out = C.c_double()
    d = lik.desc()
    ctx.call("agpl_aux_kldivergence", C.byref(d), C.c_int64(q1.shape[0]), _ptr(y), _ptr(q1), _ptr(q2), C.byref(out))
    return out.value

