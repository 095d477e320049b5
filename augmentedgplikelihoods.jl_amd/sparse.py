"""Sparse sweep drivers: the reference's user-level loops (``cavi!`` examples/bernoulli/script.jl:29-39,
multi-latent examples/categorical/script.jl:64-77, ``gibbs_sample`` :76-87) restated in the sparse form the
docs give only as a formula (docs/src/index.md:154-163), in the whitened feature basis

    Phi = L^-1 K_ZX  (K_Z = L L'),   u = L v,   q(v) = N(m, S),
    S = (I + G)^-1,  m = S g,        G = Phi Diag(gamma) Phi',  g = Phi beta,
    q(f_i) = N(phi_i' m, d_i + phi_i' S phi_i),   d_i = k_ii - |phi_i|^2.

All O(N) work is two MFMA passes over Phi per sweep (agpl_cavi_pass) plus an M x M update
(agpl_gaussian_update).  N is sharded over ranks with one all-reduce of (G, g) per sweep.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from .operators import Context, _prep, _prep_y, _ptr, _torch, default_context

PAD = 128  # feature rows are padded to a multiple of the MFMA block


def padded(M: int) -> int:
    return (M + PAD - 1) // PAD * PAD


def synth_xy(lik, seed: int, i0: int, n: int, ctx: Context | None = None, want_x: bool = True):
    """Synthetic workload of SURVEY.md 8(d): x_i = -10 + 20 u_i, y_i ~ lik(f*(x_i)), pure function of
    (seed, i0 + i).  Real-valued y comes back as float32 (the fused pass's element type)."""
    torch = _torch()
    ctx = ctx or default_context()
    L = lik._nlatent
    x = torch.empty(n, dtype=torch.float64, device=ctx.device) if want_x else None
    ydt = {"u8": torch.uint8, "i32": torch.int32, "real": torch.float32}[lik.ykind]
    y = torch.empty((n,) if L == 1 else (n, L), dtype=ydt, device=ctx.device)
    d = lik.desc()
    ctx.call("agpl_synth_xy", C.byref(d), C.c_uint64(seed), C.c_int64(i0), C.c_int64(n), _ptr(x), _ptr(y))
    return x, y


def se_features(x, z, ell: float, ctx: Context | None = None, out=None):
    """K_ZX for the squared-exponential kernel with lengthscale ``ell`` (examples/bernoulli/script.jl:15):
    float32 [N, Mp] (= [Mp, N] column-major), rows M..Mp zero."""
    torch = _torch()
    ctx = ctx or default_context()
    x = _prep(x, torch.float64, "x")
    z = _prep(z, torch.float64, "z")
    N, M = x.numel(), z.numel()
    Mp = padded(M)
    if out is None:
        out = torch.empty((N, Mp), dtype=torch.float32, device=x.device)
    ctx.call("agpl_se_features", C.c_int64(N), C.c_int32(M), C.c_int32(Mp), _ptr(x), _ptr(z), C.c_double(ell),
             _ptr(out))
    return out


def whitening_matrix(Kzz: np.ndarray, jitter: float = 0.0):
    """Host float64 setup (M x M, once): L = chol(K_Z + jitter I) (``_chol_cov`` of
    examples/bernoulli/script.jl:30 with the 1e-8 of :44) and L^-1."""
    Kzz = np.asarray(Kzz, dtype=np.float64)
    Lc = np.linalg.cholesky(Kzz + jitter * np.eye(Kzz.shape[0]))
    import scipy.linalg as sla

    Linv = sla.solve_triangular(Lc, np.eye(Kzz.shape[0]), lower=True)
    return Lc, Linv


def whiten_features(Kzx, Linv: np.ndarray, ctx: Context | None = None, out=None):
    """Phi = L^-1 K_ZX on the matrix cores (agpl_transform_features).  Kzx: float32 [N, Mp]."""
    torch = _torch()
    ctx = ctx or default_context()
    Kzx = _prep(Kzx, torch.float32, "Kzx")
    N, Mp = Kzx.shape
    M = Linv.shape[0]
    A = np.zeros((Mp, Mp), dtype=np.float32)
    A[:M, :M] = Linv
    At = torch.from_numpy(np.ascontiguousarray(A.T)).to(Kzx.device)  # column-major A
    if out is None:
        out = torch.empty_like(Kzx)
    ctx.call("agpl_transform_features", C.c_int64(N), C.c_int32(Mp), _ptr(At), _ptr(Kzx), _ptr(out))
    return out


def shard_range(N: int, rank: int, world: int):
    """Points [i0, i1) owned by ``rank`` when N observations are sharded over ``world`` ranks:
    contiguous, sizes differ by at most one, a pure function of (N, rank, world)."""
    return rank * N // world, (rank + 1) * N // world


def natural_parameter_buffers(L, M, device, extra=0):
    """(flat, G, g): G [L, M, M] and g [L, M] float64 as views of ONE flat buffer, so that the exchange step of a
    sweep is a single collective of L (M^2 + M) doubles.  ``extra`` more doubles ride at its end (the ELBO terms of a sweep
    are summed over ranks by the same collective)."""
    torch = _torch()
    flat = torch.zeros(L * M * M + L * M + extra, dtype=torch.float64, device=device)
    return flat, flat[: L * M * M].view(L, M, M), flat[L * M * M: L * M * M + L * M].view(L, M)


def plan_padded(M: int) -> int:
    """The feature count a plan works on: M rounded up to a multiple of 256 (include/agpl.h: the state arrays of agpl_plan_state
    are sized by it; the images carry zero features beyond M)."""
    return (M + 255) // 256 * 256


class Plan:
    """agpl_plan (include/agpl.h): the two split-float16 images of Phi (one scale), the Nystrom residual and q(v) in factor form,
    in ONE torch-owned block of device memory; the float32 features are not referenced after construction.  ANY feature count M:
    the plan pads to ``Mp = plan_padded(M)`` itself (round 6); ``U_colmajor`` / ``v`` are the Mp-sized state (checkpoints),
    ``U_lead`` / ``v_lead`` their leading M x M / M blocks -- the caller's q(v)."""

    NO_MARGINALS = 1  # AGPL_PLAN_NO_MARGINALS: Gibbs passes only (no marginal image)

    def __init__(self, Phi, resid, L, ctx: Context, flags: int = 0):
        torch = _torch()
        self.ctx = ctx
        self.N, self.M = Phi.shape
        self.L = L
        self.flags = flags
        self.Mp = plan_padded(self.M)
        nbytes = _ffi.lib().agpl_plan_bytes(C.c_int64(self.N), C.c_int32(self.M), C.c_int32(L), C.c_uint32(flags))
        if nbytes <= 0:
            raise _ffi.ArgumentError(-1, f"a plan needs N >= 1 points, M >= 1 features (got {self.N}, {self.M}) and at most 64 "
                                         f"latents (got {L})")
        self.mem = torch.empty(nbytes, dtype=torch.uint8, device=Phi.device)
        self._h = C.c_void_p()
        ctx.call("agpl_plan_create", C.c_int64(self.N), C.c_int32(self.M), C.c_int32(L), _ptr(Phi), _ptr(resid),
                 C.c_uint32(flags), _ptr(self.mem), C.byref(self._h))
        U, v, uh, ul, v32, ld = (C.c_void_p() for _ in range(6))
        _ffi.check(ctx._h, _ffi.lib().agpl_plan_state(self._h, C.byref(U), C.byref(v), C.byref(uh), C.byref(ul), C.byref(v32),
                                                      C.byref(ld), None))
        base = self.mem.data_ptr()
        M = self.Mp  # the state arrays are sized by the padded feature count
        view = lambda ptr, nb, dt: self.mem[ptr.value - base: ptr.value - base + nb].view(dt)
        self.U_colmajor = view(U, 8 * L * M * M, torch.float64).view(L, M, M)
        self.v = view(v, 8 * L * M, torch.float64).view(L, M)
        self.U_hi, self.U_lo = view(uh, 2 * L * M * M, torch.float16), view(ul, 2 * L * M * M, torch.float16)
        self.v32 = view(v32, 4 * L * M, torch.float32).view(L, M)
        self.logdet = view(ld, 8 * L, torch.float64)
        e = C.c_int32()
        _ffi.check(ctx._h, _ffi.lib().agpl_plan_info(self._h, None, None, None, C.byref(e), None))
        self.scale_exp = e.value
        self.nbytes = nbytes
        self.U_lead = self.U_colmajor[:, : self.M, : self.M]
        self.v_lead = self.v[:, : self.M]

    def call(self, name, *args):
        self.ctx.bind()
        _ffi.check(self.ctx._h, getattr(_ffi.lib(), name)(self._h, *args))

    _STATE = ("U_colmajor", "v", "U_hi", "U_lo", "v32", "logdet")

    def state(self):
        """Checkpoint of what an update rewrites (agpl_plan_state: U, v, their float16 / float32 images,
        log det(I + G)) as host tensors.  The images of Phi are static: a restoring process rebuilds them from the same features."""
        self.ctx.synchronize()
        return {k: getattr(self, k).detach().cpu().clone() for k in self._STATE}

    def load_state(self, st):
        """Restore a checkpoint into a plan created for the same data: the next pass is bit for bit the saved run's."""
        self.ctx.synchronize()
        for k in self._STATE:
            dst = getattr(self, k)
            if tuple(st[k].shape) != tuple(dst.shape) or st[k].dtype != dst.dtype:
                raise _ffi.ArgumentError(-1, f"checkpoint field {k}: {tuple(st[k].shape)} {st[k].dtype}, the plan holds "
                                             f"{tuple(dst.shape)} {dst.dtype}")
            dst.copy_(st[k])

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().agpl_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


def exchange_natural_parameters(G, g, group=None, flat=None):
    """The one exchange step of a sweep: sum the per-rank partials of the M x M natural-parameter
    accumulators over the ranks that shard N (torch.distributed all-reduce; backend "nccl" = RCCL over
    xGMI on the GPU box, "gloo" in the CPU tests).  float64 on the wire; in place.  ``flat``: the buffer G and g are
    views of (natural_parameter_buffers) -- one collective instead of two."""
    if group is None:
        return G, g
    import torch.distributed as dist

    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return G, g
    dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return G, g


class SparseCAVI:
    """CAVI sweeps over N local points and M features.

    Phi: float32 [N, M] CUDA tensor (any M on the plan path, which pads to a multiple of 256 itself; M % 128 == 0 for the
    float32-input pair); kdiag: float32 [N] (d_i above); y as the likelihood
    wants it (real-valued y: float32); mu0: optional prior mean at the data, float32 [L][N].
    ``group``: a torch.distributed process group over which N is sharded (None = single GPU).
    """

    def __init__(self, lik, Phi, kdiag, y, mu0=None, ctx: Context | None = None, group=None, keep_points=False,
                 marginal_precision: str = "auto", accumulate_precision: str = "auto", track_elbo: bool = False):
        """Two arithmetics, chosen for both contractions together:
          * the shipped path ("f16x2-factor" marginals + "f16x2" accumulation; "auto"): ONE plan (agpl_plan_create: both
            split-float16 images of Phi with one scale, q(v) in factor form; any feature count -- the plan zero-pads to a multiple
            of 256 itself, G / g / S / m stay M-sized) and agpl_cavi_pass_plan / agpl_plan_update per sweep;
          * "f32" / "f32" (on request): the float32-input MFMA kernels (agpl_cavi_pass + agpl_gaussian_update), the
            arithmetic SURVEY.md 8(d) prices; feature count a multiple of 128.
        ``track_elbo``: the per-point ELBO terms ride the pass and the Gaussian KL the update (plan path only); see
        ``elbo_entering``."""
        torch = _torch()
        self.ctx = ctx or default_context()
        self.lik = lik
        self.Phi = _prep(Phi, torch.float32, "Phi")
        self.N, self.M = self.Phi.shape
        self.L = lik._nlatent
        self.kdiag = _prep(kdiag, torch.float32, "kdiag")
        self.y = _prep_y(lik, y, torch.float32)
        self.mu0 = _prep(mu0, torch.float32, "mu0")
        self.group = group
        split_names = ("f16x2", "f16x2-factor")
        if marginal_precision == "auto":
            marginal_precision = "f32" if accumulate_precision == "f32" else "f16x2-factor"
        if accumulate_precision == "auto":
            accumulate_precision = "f32" if marginal_precision == "f32" else "f16x2"
        if marginal_precision not in ("f32", "f16x2-factor") or accumulate_precision not in ("f32", "f16x2"):
            raise _ffi.ArgumentError(-1, "marginal_precision must be 'auto', 'f32' or 'f16x2-factor', accumulate_precision 'auto', "
                                         "'f32' or 'f16x2'")
        if (marginal_precision in split_names) != (accumulate_precision == "f16x2"):
            # the entry points come in two arithmetics: float32-input MFMA for both contractions (agpl_cavi_pass), or
            # split-float16 for both (the plan)
            raise _ffi.ArgumentError(-1, "marginal_precision and accumulate_precision must both be 'f32' or both be split-float16")
        self.marginal_precision = marginal_precision
        self.factor = marginal_precision == "f16x2-factor"
        if not self.factor and self.M % PAD:
            raise _ffi.ArgumentError(-1, f"the float32-input kernels need a feature count that is a multiple of {PAD} (got {self.M}; "
                                         "zero-pad, or take the plan path)")
        dev = self.Phi.device
        L, M = self.L, self.M
        f64, f32 = torch.float64, torch.float32
        self.plan = None
        self.track_elbo = bool(track_elbo)
        self.gamma = self.beta = self.c = None
        if keep_points:
            self.gamma = torch.empty((L, self.N), dtype=f32, device=dev)
            self.beta = torch.empty((L, self.N), dtype=f32, device=dev)
            self.c = torch.empty((self.N,) if L == 1 else (self.N, L), dtype=f32, device=dev)
        self.nsweeps = 0
        self._Gg, self.G, self.g = natural_parameter_buffers(L, M, dev, extra=1 if (self.track_elbo and self.factor) else 0)
        self._kl = torch.zeros(2, dtype=f64, device=dev)  # KL(q(v) || p(v)) of the last two updates (q = N(0, I) at start)
        if self.factor:
            self.resid = self.kdiag  # kdiag is already d_i = k_ii - |phi_i|^2 (agpl_feature_residual / nystrom_residual)
            self.plan = Plan(self.Phi, self.resid, L, self.ctx)
            self._elbo_terms = self._Gg[-1:] if self.track_elbo else None
            self.A_work, self.v = self.plan.U_lead, self.plan.v_lead  # (views of the plan's state: the caller's M x M / M blocks)
            self._S = self._m = self.Wpack = self.alpha = None
            return
        if self.track_elbo:
            raise _ffi.ArgumentError(-1, "track_elbo rides the plan path (the default arithmetic)")
        # q(v) = N(0, I) (script.jl:41-42): the update of G = 0, g = 0 gives S = I, m = 0 and the packed -I the first pass reads
        self._S = torch.empty((L, M, M), dtype=f64, device=dev)
        self._m = torch.empty((L, M), dtype=f64, device=dev)
        self.Wpack = torch.empty((L, M, M), dtype=f32, device=dev)
        self.alpha = torch.empty((L, M), dtype=f32, device=dev)
        self.update()

    @property
    def S(self):
        """Covariance of q(v).  In the factor form it is materialised on demand: S = U'U, U' = triu of A_work viewed
        row-major (the column-major lower triangle holds U)."""
        if not self.factor:
            return self._S
        self.check()
        Ut = _torch().triu(self.A_work)
        return Ut @ Ut.transpose(1, 2)

    @property
    def m(self):
        if not self.factor:
            return self._m
        self.check()
        return (_torch().triu(self.A_work) @ self.v.unsqueeze(-1)).squeeze(-1)

    def accumulate(self):
        """marginals -> aux_posterior! -> expected potential/precision -> local (G, g)."""
        d = self.lik.desc()
        if self.plan is not None:  # the shipped path
            self.plan.call("agpl_cavi_pass_plan", C.byref(d), _ptr(self.mu0), _ptr(self.y), _ptr(self.G), _ptr(self.g),
                           _ptr(self.c), _ptr(self.gamma), _ptr(self.beta), _ptr(self._elbo_terms))
            return
        self.ctx.call("agpl_cavi_pass", C.byref(d), C.c_int64(self.N), C.c_int32(self.M), _ptr(self.Phi),
                      _ptr(self.kdiag), _ptr(self.mu0), _ptr(self.y), _ptr(self.Wpack), _ptr(self.alpha),
                      _ptr(self.G), _ptr(self.g), _ptr(self.c), _ptr(self.gamma), _ptr(self.beta))

    exchange_timing = None  # a list: exchange() appends a (start, stop) pair of timing events per call (bench.py)

    def exchange(self):
        """Sum (G, g) over the ranks that shard N; every rank then performs the identical M x M update."""
        if self.exchange_timing is not None and self.group is not None:
            torch = _torch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            exchange_natural_parameters(self.G, self.g, self.group, flat=getattr(self, "_Gg", None))
            e1.record()
            self.exchange_timing.append((e0, e1))
            return
        exchange_natural_parameters(self.G, self.g, self.group, flat=getattr(self, "_Gg", None))

    def update(self):
        """S = (I + G)^-1, m = S g (examples/bernoulli/script.jl:35-36 in sparse whitened form); the plan keeps
        U = chol(I + G)^-1 and v = U g (S = U'U, m = U'v) and the images of U."""
        if self.plan is not None:
            # enqueue only: a failed factorisation surfaces in the next accumulate() (once its kernels are queued),
            # in check(), or when S / m / elbo() are read -- the host never idles the GPU between update and pass
            kl = self._kl[self.nsweeps & 1:(self.nsweeps & 1) + 1] if self.track_elbo else None
            self.plan.call("agpl_plan_update", _ptr(self.G), _ptr(self.g), C.c_void_p(0), _ptr(kl))
            return
        self.ctx.call("agpl_gaussian_update", C.c_int32(self.M), C.c_int32(self.L), _ptr(self.G), _ptr(self.g),
                      C.c_void_p(0), _ptr(self._S), _ptr(self._m), _ptr(self.Wpack), _ptr(self.alpha), _ptr(self._kl[:1]))

    def sweep(self):
        self.accumulate()
        self.exchange()
        self.update()
        self.nsweeps += 1

    def check(self):
        """Wait for the stream and raise what a deferred factorisation has to report (PosDefException, ...)."""
        self.ctx.synchronize()

    def state_dict(self):
        """Checkpoint of the sweep loop (SURVEY.md 5 "checkpoint / resume"; the reference's state is the (m, S, qΩ) of
        examples/bernoulli/script.jl:41-43): q(v) in the plan's factor form, the reduced (G, g) of the last sweep, the sweep count
        and the context's Philox key / draw counter.  qΩ is a function of q(v) and is not stored.  Plan path only."""
        if self.plan is None:
            raise _ffi.ArgumentError(-1, "state_dict is the plan path's (the default arithmetic)")
        self.check()
        return {"plan": self.plan.state(), "Gg": self._Gg.detach().cpu().clone(), "kl": self._kl.detach().cpu().clone(),
                "nsweeps": self.nsweeps, "seed": int(self.ctx.seed), "sweep_counter": int(self.ctx.sweep),
                "shape": (self.N, self.M, self.L)}

    def load_state_dict(self, st, restore_rng: bool = True):
        if self.plan is None:
            raise _ffi.ArgumentError(-1, "load_state_dict is the plan path's")
        if tuple(st["shape"]) != (self.N, self.M, self.L):
            raise _ffi.ArgumentError(-1, f"the checkpoint is of a {tuple(st['shape'])} problem, this one is {(self.N, self.M, self.L)}")
        self.plan.load_state(st["plan"])
        self._Gg[: st["Gg"].numel()].copy_(st["Gg"][: self._Gg.numel()])
        self._kl.copy_(st["kl"])
        self.nsweeps = int(st["nsweeps"])
        if restore_rng:
            self.ctx.set_seed(int(st["seed"]))
            self.ctx.sweep = int(st["sweep_counter"])

    def elbo_entering(self):
        """aug_elbo (examples/bernoulli/script.jl:65-70) of the q(v) that ENTERED the last sweep, at no extra pass over the
        features (``track_elbo=True``): the per-point terms expected_logtilt - aux_kldivergence rode that sweep's per-point kernel
        (summed over ranks by the sweep's one all-reduce), and KL(q(v) || p(v)) rode the update that produced that q(v).
        After sweep k this is the ELBO after k - 1 updates: a convergence monitor that lags one sweep."""
        if not self.track_elbo or self.nsweeps == 0:
            raise _ffi.ArgumentError(-1, "elbo_entering needs track_elbo=True and at least one sweep")
        self.check()
        # sweep k (1-based) wrote kl[(k - 1) & 1] for its NEW q(v); the q(v) that entered it was made by sweep k - 1 -> kl[k & 1]
        # (k = 1: the initial N(0, I), whose slot still holds 0)
        return float(self._elbo_terms.item()) - float(self._kl[self.nsweeps & 1].item())

    def run(self, niter: int = 10):
        for _ in range(niter):
            self.sweep()
        self.check()
        return self.m, self.S

    def elbo(self):
        """aug_elbo of examples/bernoulli/script.jl:65-70 for the current q(v):
        expected_logtilt(lik, qΩ, y, qf) - aux_kldivergence(lik, qΩ, y) - KL(q(v) || p(v)), with qf the current
        marginals and qΩ = aux_posterior(lik, y, qf) (float64 operator kernels; the Gaussian KL from the factor).  Local points
        only: with N sharded, sum the first two terms over ranks and count the KL once."""
        from . import operators as ops

        torch = _torch()
        mu, var = self.marginals()  # [L][N] float32
        if self.L == 1:
            qf = (mu[0].to(torch.float64), var[0].to(torch.float64))
        else:
            qf = (mu.t().contiguous().to(torch.float64), var.t().contiguous().to(torch.float64))
        y = self.y.to(torch.float64) if self.lik.ykind == "real" else self.y
        qΩ = ops.aux_posterior(self.lik, y, qf, ctx=self.ctx)
        elt = ops.expected_logtilt(self.lik, qΩ, y, qf, ctx=self.ctx)
        kl_aux = ops.aux_kldivergence(self.lik, qΩ, y, ctx=self.ctx)
        if self.plan is not None:
            self.check()
            Ut = torch.triu(self.A_work)  # U' (row-major view of the column-major lower triangle)
            m = (Ut @ self.v.unsqueeze(-1)).squeeze(-1)
            kl = 0.5 * float(((Ut * Ut).sum() + (m * m).sum() - self.L * self.M + self.plan.logdet.sum()).item())
        else:
            kl = float(self._kl[0].item())  # of the last agpl_gaussian_update (its kl_out)
        return elt - kl_aux - kl

    def natural_parameters(self):
        """(Lambda_v, eta_v) = (I + G, g): the whitened natural parameters of q(v)."""
        torch = _torch()
        eye = torch.eye(self.M, dtype=torch.float64, device=self.G.device)
        return self.G + eye, self.g

    def marginals(self):
        """q(f_i) for the current (m, S): (mu, var) float32 [L][N]."""
        torch = _torch()
        mu = torch.empty((self.L, self.N), dtype=torch.float32, device=self.y.device)
        var = torch.empty_like(mu)
        if self.plan is not None:
            self.plan.call("agpl_marginals_plan", _ptr(self.mu0), _ptr(mu), _ptr(var))
            return mu, var
        self.ctx.call("agpl_marginals", C.c_int64(self.N), C.c_int32(self.M), C.c_int32(self.L), _ptr(self.Phi),
                      _ptr(self.kdiag), _ptr(self.mu0), _ptr(self.Wpack), _ptr(self.alpha), _ptr(mu), _ptr(var))
        return mu, var


def nystrom_residual(Phi, kxx, ctx: Context | None = None):
    """d_i = k_ii - |phi_i|^2 (agpl_feature_residual: float64 accumulation), the `resid` of a plan / `kdiag` of the sweeps."""
    torch = _torch()
    ctx = ctx or default_context()
    Phi = _prep(Phi, torch.float32, "Phi")
    N, M = Phi.shape
    kxx = _prep(kxx, torch.float32, "kxx")
    out = torch.empty(N, dtype=torch.float32, device=Phi.device)
    ctx.call("agpl_feature_residual", C.c_int64(N), C.c_int32(M), _ptr(Phi), _ptr(kxx), _ptr(out))
    return out


class SparseGibbs:
    """Gibbs sweeps of the sparse model (``gibbs_sample`` of examples/bernoulli/script.jl:76-87 restated for M
    inducing coordinates in the whitened basis):

        f_i | v  ~ N(phi_i' v, d_i)            (agpl_gibbs_pass: projection + noise, per-point Philox stream)
        Ω_i | f_i ~ aux_full_conditional       (aux_sample!, same stream)
        v | Ω    ~ N(m, S), S = (I + G)^-1, m = S g,  G = Phi Diag(γ(Ω)) Phi', g = Phi β(Ω)   (agpl_gibbs_draw_v)

    N is sharded over ``group`` exactly as in SparseCAVI: (G, g) are all-reduced, every rank draws the identical v
    (same Philox key / counter: every rank's Context must carry the same seed and be at the same draw counter).
    ``point_offset`` = global index of this rank's first point (``shard_range(N, rank, world)[0]``): the per-point
    streams are keyed on the GLOBAL point index, so the sharded chain is the single-process chain.
    Real-valued y must be float64 here (the Gibbs operators are Float64).
    """

    def __init__(self, lik, Phi, kdiag, y, mu0=None, ctx: Context | None = None, group=None, keep_points=False,
                 accumulate_precision: str = "auto", point_offset: int = 0, plan: "Plan | None" = None):
        """``accumulate_precision``: "f16x2" (the plan's split-float16 image accumulation; "auto"; any feature count) or "f32"
        (agpl_gibbs_pass: float32-input MFMA, feature count a multiple of 128).  ``plan``: the plan of a SparseCAVI over the same
        features (its accumulate image and residual are shared); by default a plan without the marginal image is built here."""
        torch = _torch()
        self.ctx = ctx or default_context()
        if accumulate_precision == "auto":
            accumulate_precision = "f16x2"
        if accumulate_precision not in ("f32", "f16x2"):
            raise _ffi.ArgumentError(-1, "accumulate_precision must be 'auto', 'f32' or 'f16x2'")
        self.acc_split = 1 if accumulate_precision == "f16x2" else 0
        self.lik = lik
        self.Phi = _prep(Phi, torch.float32, "Phi")
        self.N, self.M = self.Phi.shape
        if not self.acc_split and self.M % PAD:
            raise _ffi.ArgumentError(-1, f"the float32-input kernels need a feature count that is a multiple of {PAD} (got {self.M}; "
                                         "zero-pad, or accumulate_precision='f16x2')")
        self.L = lik._nlatent
        self.kdiag = _prep(kdiag, torch.float32, "kdiag")
        self.y = _prep_y(lik, y, torch.float64)
        self.mu0 = _prep(mu0, torch.float32, "mu0")
        self.group = group
        dev = self.Phi.device
        L, M = self.L, self.M
        f64 = torch.float64
        self.plan = None
        if self.acc_split:
            if plan is not None and (plan.N, plan.M, plan.L) != (self.N, M, L):
                raise _ffi.ArgumentError(-1, "the plan was created for another problem size")
            if plan is not None and plan.ctx is not self.ctx:
                # the pass runs on the PLAN's context (its seed, its point offset): a plan from another Context would draw that
                # context's Philox streams while this object sets the offset on its own
                raise _ffi.ArgumentError(-1, "the plan belongs to another Context: pass ctx=plan.ctx (or build the plan on this one)")
            self.plan = plan if plan is not None else Plan(self.Phi, self.kdiag, L, self.ctx, flags=Plan.NO_MARGINALS)
        self._Gg, self.G, self.g = natural_parameter_buffers(L, M, dev)
        self.v = torch.empty((L, M), dtype=f64, device=dev)
        self.m = torch.empty((L, M), dtype=f64, device=dev)
        self.point_offset = int(point_offset)
        if group is not None and self.point_offset == 0:
            import torch.distributed as dist

            if dist.get_rank(group) != 0:
                raise _ffi.ArgumentError(-1, "SparseGibbs(group=...) needs point_offset = the global index of this "
                                             "rank's first point on every rank but the first")
        self.sweep_index = None  # draw counter of the sweep in flight (taken from the Context)
        self.f = self.omega = self.n = None
        if keep_points:
            Lo = 1 if lik.kind == 7 else L
            self.f = torch.empty((self.N, L), dtype=f64, device=dev)
            self.omega = torch.empty((self.N, Lo), dtype=f64, device=dev)
            self.n = torch.zeros((self.N, Lo), dtype=torch.int64, device=dev)
        self.sweep_index = self.ctx.next_sweep()
        self._check_ranks_agree()
        self.draw()  # G = 0, g = 0: v ~ N(0, I), the prior draw (script.jl:89 `f = randn(N)`)

    def _check_ranks_agree(self):
        """Every rank draws v itself: the ranks' Philox key (seed) and draw counter must be the same, or the ranks sample
        different chains behind an all-reduce that still "works".  One tiny MIN / MAX exchange at construction."""
        if self.group is None:
            return
        import torch.distributed as dist

        torch = _torch()
        dev = self.Phi.device if dist.get_backend(self.group) == "nccl" else "cpu"
        seed = int(self.ctx.seed) & 0xFFFFFFFFFFFFFFFF
        mine = torch.tensor([seed & 0xFFFFFFFF, seed >> 32, int(self.sweep_index)], dtype=torch.int64, device=dev)
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not bool((lo == hi).all()):
            raise _ffi.ArgumentError(-1, "SparseGibbs(group=...): the ranks' Contexts differ in seed or draw counter "
                                         f"(this rank: seed {seed}, sweep {self.sweep_index}); create every rank's "
                                         "Context with the same seed and use them for the same sequence of draws")

    def draw(self):
        """v ~ N(m, S) on the streams (seed, l M + a, sweep_index | 2^31): identical on every rank."""
        self.ctx.call("agpl_gibbs_draw_v", C.c_int32(self.M), C.c_int32(self.L), _ptr(self.G), _ptr(self.g),
                      C.c_void_p(0), C.c_uint32(self.sweep_index), _ptr(self.v), _ptr(self.m))

    def accumulate(self):
        """Point pass of one sweep on the streams (seed, point_offset + i, sweep_index)."""
        d = self.lik.desc()
        self.sweep_index = self.ctx.next_sweep()
        prev_offset = self.ctx.point_offset
        self.ctx.set_point_offset(self.point_offset)
        try:  # (a raising pass must not leave the context's point offset shifted for its other users)
            if self.plan is not None:  # the shipped path
                self.plan.call("agpl_gibbs_pass_plan", C.byref(d), _ptr(self.mu0), _ptr(self.y), _ptr(self.v),
                               C.c_uint32(self.sweep_index), _ptr(self.G), _ptr(self.g), _ptr(self.f), _ptr(self.omega),
                               _ptr(self.n), C.c_void_p(0))
            else:
                self.ctx.call("agpl_gibbs_pass", C.byref(d), C.c_int64(self.N), C.c_int32(self.M), _ptr(self.Phi),
                              _ptr(self.kdiag), _ptr(self.mu0), _ptr(self.y), _ptr(self.v), C.c_uint32(self.sweep_index),
                              _ptr(self.G), _ptr(self.g), _ptr(self.f), _ptr(self.omega), _ptr(self.n), C.c_void_p(0))
        finally:
            self.ctx.set_point_offset(prev_offset)

    def exchange(self):
        exchange_natural_parameters(self.G, self.g, self.group, flat=getattr(self, "_Gg", None))

    def state_dict(self):
        """Checkpoint of the chain (the reference's state is the (f, Ω) of examples/bernoulli/script.jl:89-90; here f and Ω are
        redrawn from v every sweep): the inducing draw v, and the context's Philox key and draw counter -- counter-based streams
        make the resumed chain the uninterrupted one, bit for bit."""
        self.ctx.synchronize()
        return {"v": self.v.detach().cpu().clone(), "m": self.m.detach().cpu().clone(), "seed": int(self.ctx.seed),
                "sweep_counter": int(self.ctx.sweep), "sweep_index": int(self.sweep_index), "point_offset": self.point_offset,
                "shape": (self.N, self.M, self.L)}

    def load_state_dict(self, st):
        if tuple(st["shape"]) != (self.N, self.M, self.L):
            raise _ffi.ArgumentError(-1, f"the checkpoint is of a {tuple(st['shape'])} problem, this one is {(self.N, self.M, self.L)}")
        self.ctx.synchronize()
        self.v.copy_(st["v"])
        self.m.copy_(st["m"])
        self.ctx.set_seed(int(st["seed"]))
        self.ctx.sweep = int(st["sweep_counter"])
        self.sweep_index = int(st["sweep_index"])
        self._check_ranks_agree()

    def sweep(self):
        self.accumulate()
        self.exchange()
        self.draw()
        return self.v

    def run(self, nsamples: int = 200):
        """Returns the [nsamples, L, M] chain of inducing draws."""
        torch = _torch()
        out = torch.empty((nsamples, self.L, self.M), dtype=torch.float64, device=self.Phi.device)
        for t in range(nsamples):
            out[t] = self.sweep()
        return out


class DenseGibbs:
    """``gibbs_sample(fz, f, Ω)`` of examples/bernoulli/script.jl:76-87 (studentt/script.jl is the same loop):
    full-rank Gibbs over N points with a dense prior covariance K [N, N] float64 (BASELINE config C5).
    One float64 Cholesky of I + D^1/2 K D^1/2 per sweep instead of the reference's two inverses + Cholesky."""

    def __init__(self, lik, K, y, mu0=None, f0=None, ctx: Context | None = None):
        torch = _torch()
        self.ctx = ctx or default_context()
        self.lik = lik
        self.K = _prep(K, torch.float64, "K")
        self.N = self.K.shape[0]
        self.y = _prep_y(lik, y, torch.float64)
        self.mu0 = _prep(mu0, torch.float64, "mu0")
        dev = self.K.device
        self.Lk = torch.empty_like(self.K)
        self.ctx.call("agpl_dense_cholesky", C.c_int64(self.N), _ptr(self.K), _ptr(self.Lk))  # script.jl:77
        self.B = torch.empty_like(self.K)
        self.f = (torch.zeros(self.N, dtype=torch.float64, device=dev) if f0 is None
                  else _prep(f0, torch.float64, "f0").clone())
        self.omega = torch.empty(self.N, dtype=torch.float64, device=dev)
        self.n = torch.zeros(self.N, dtype=torch.int64, device=dev) if lik.kind == 5 else None
        self.sweep_index = None

    def sweep(self):
        d = self.lik.desc()
        self.sweep_index = self.ctx.next_sweep()
        self.ctx.call("agpl_dense_gibbs_step", C.byref(d), C.c_int64(self.N), _ptr(self.K), _ptr(self.Lk),
                      _ptr(self.mu0), _ptr(self.y), _ptr(self.f), _ptr(self.B), C.c_uint32(self.sweep_index),
                      _ptr(self.omega), _ptr(self.n))
        return self.f

    def run(self, nsamples: int = 200):
        torch = _torch()
        out = torch.empty((nsamples, self.N), dtype=torch.float64, device=self.K.device)
        for t in range(nsamples):
            out[t] = self.sweep()
        return out
