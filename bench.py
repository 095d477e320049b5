#!/usr/bin/env python3
"""bench.py -- CAVI sweeps/s of the sparse augmented-likelihood sweep on MI355X.

Workload (BASELINE.json configs[1], "C2"): BernoulliLikelihood(LogisticLink) SVGP, N = 1e7 observations,
M = 512 inducing points, synthetic data (x ~ U(-10, 10), y ~ Bernoulli(logistic(f*(x))),
squared-exponential kernel with lengthscale 1.5 * inducing spacing, whitened features).
One "step" = one CAVI sweep = marginals (MFMA) -> aux_posterior! / expected_auglik_* (elementwise) ->
G = Phi diag(gamma) Phi', g = Phi beta (MFMA, split over N) -> [all-reduce over ranks] -> M x M update.
All inputs are resident in HBM before the timed region.

N is a fixed total sharded over ranks (strong scaling), with one RCCL all-reduce of (G, g) per sweep.

At N = 1 the same JSON line also carries (each skippable by a --no-... flag):
  "m1024"  the north-star target configuration (Bernoulli, N = 1e7, M = 1024) with the same timed-loop discipline,
           its own roofline and a 10-sweep parity slice;
  "gibbs"  the Gibbs half on the resident workload (point pass, PG sampler rates for Bernoulli AND NegBin r = 15);
  "parity" 10 sweeps on a 20 000-point slice against the oracle; "full_size_check"; "cpu_baseline";
  "c5"     BASELINE configs[4]: StudentT full-rank Gibbs step at N = 65 536 (float64).
  "n8", "c3r" (+ "n8_m1024"), "c4"  one rank's share of C2 / of C3 and the north-star configuration at 8 GPUs (N / 8 points), and
           BASELINE configs[3] (categorical K = 10, N = 1e6, M = 256): ms_per_step, roofline.kernels, ten-sweep parity, Gibbs
           sweep, projected_scaling_8 = full-N ms / per-rank ms (--no-extra skips them and the full-N CPU sweep).

At N > 1 (default configuration, or --sharded-legs on) the ranks go on, after the sharded C2 headline, to BASELINE configs[2] itself
("c3": NegBin r = 15, N sharded, M = 1024, ten CAVI sweeps + five sparse Gibbs sweeps, all-reduce time, per-rank min / max, ten-sweep
parity on a slice) and to the Bernoulli M = 1024 north-star target ("m1024"), same timed-loop discipline.
The LAST key of the line is "summary": every configuration's ms per step / sweeps/s / dominant-kernel frac / parity in <= 1.5 KB.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 10000000] [--m 512] [--lik bernoulli]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import datetime
import gc
import json
import os
import sys
import threading
import time

import numpy as np

# RCCL / device-tensor sharing between the ranks of one node needs dmabuf IPC on this driver stack (set before anything touches HIP)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20240807
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, f32 in / f32 accumulate
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/F16 MFMA ~2.5 PF dense
PMC_PROFILE = "r06_pmc_traffic_c2.json"  # HBM bytes per launch of the two contraction kernels (separate --pmc passes)
PMC_PROFILE_M1024 = "r06_pmc_traffic_m1024.json"  # the same at the north-star target configuration (N = 1e7, M = 1024)
RENDEZVOUS_TIMEOUT_S = 300  # a rank that cannot join (or whose first collective hangs) exits non-zero after this


def make_lik(A, name):
    if name == "bernoulli":
        return A.BernoulliLikelihood()
    if name == "negbin":
        return A.NegativeBinomialLikelihood(15.0)  # examples/negativebinomial/script.jl:17
    if name == "studentt":
        return A.StudentTLikelihood(3.5, 2.0)  # examples/studentt/script.jl:17-19
    if name == "categorical":
        return A.CategoricalLikelihood(np.zeros(10))  # LogisticSoftMaxLink(zeros(10))
    raise SystemExit(f"unknown --lik {name}")


def make_olik(O, name):
    return {"bernoulli": O.bernoulli, "negbin": lambda: O.negbinomial(15.0), "studentt": lambda: O.studentt(3.5, 2.0),
            "categorical": lambda: O.categorical(np.zeros(10))}[name]()


def build_workload(A, ctx, lik, i0, n_loc, M):
    """Setup (untimed): synthetic data, K_ZX, whitening, Nystrom residual -- all on device."""
    import torch

    x, y = A.synth_xy(lik, SEED, i0, n_loc, ctx=ctx)
    z = np.linspace(-10.0, 10.0, M)
    ell = 1.5 * (z[1] - z[0])
    Kzz = np.exp(-0.5 * ((z[:, None] - z[None, :]) / ell) ** 2)
    _, Linv = A.sparse.whitening_matrix(Kzz, 1e-8)
    Kzx = A.se_features(x, torch.from_numpy(z).cuda(), ell, ctx=ctx)
    Phi = A.whiten_features(Kzx, Linv, ctx=ctx)
    del Kzx
    kd = A.sparse.nystrom_residual(Phi, torch.ones(n_loc, device="cuda"), ctx=ctx)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return y, Phi, kd


def read_timing(ctx, which):
    from agpl_amd import _ffi

    ms, cnt = C.c_double(), C.c_int64()
    _ffi.check(ctx.bind(), _ffi.lib().agpl_timing(ctx.bind(), C.c_int32(which), C.byref(ms), C.byref(cnt)))
    return ms.value, cnt.value


def timed_sweeps(ctx, cavi, steps, warmup, barrier):
    """W untimed sweeps, then exactly K sweeps bracketed by barrier + synchronize on both sides; the deferred outcome of
    the last factorisation is read inside the timed region.  Returns (seconds, per-kernel hipEvent timings)."""
    from agpl_amd import _ffi

    # a full (generation-2) pass of Python's cyclic collector over the ~1e6 objects torch imports takes ~75 ms and
    # used to land in one random sweep of the timed loop: collect now (before the warm-up: 75 ms of idle device between the warm-up
    # and the timed sweeps would let the clocks drop again), keep the collector off while timing
    gc.collect()
    gc.disable()
    for _ in range(warmup):
        cavi.sweep()
    barrier()
    cavi.exchange_timing = [] if cavi.group is not None else None  # hipEvent pairs around every exchange() of the timed sweeps
    _ffi.lib().agpl_timing(ctx.bind(), C.c_int32(-1), None, None)
    read_timing(ctx, 0)  # discard what earlier legs left in the kernel timers (the sampler legs call cavi.marginals() with the
    read_timing(ctx, 1)  # timers on: two C2-size launches used to be averaged into the m1024 leg's marginal kernel: 16.8 for 20.1 ms)
    trace = os.environ.get("AGPL_BENCH_TRACE")  # debug: per-step wall times
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        cavi.sweep()
        if trace:
            print(f"[trace] step {1e3 * (time.perf_counter() - ts):.2f} ms", file=sys.stderr)
    barrier()
    cavi.check()  # inside the timed region: the last sweep's deferred factorisation outcome (raises if it failed)
    dt = time.perf_counter() - t0
    gc.enable()
    kt = [read_timing(ctx, 0), read_timing(ctx, 1)]
    _ffi.lib().agpl_timing(ctx.bind(), C.c_int32(-2), None, None)
    if cavi.exchange_timing is not None:
        xs = [a.elapsed_time(b) for a, b in cavi.exchange_timing]
        cavi.exchange_ms = (sum(xs) / max(len(xs), 1), max(xs) if xs else 0.0)
        cavi.exchange_timing = None
    return dt, kt


def roofline_of(kt, L, n_loc, M, Mp, marginal, accumulate, ms_per_step, world, N, traffic_key=None, profile=None):
    """Roofline of the dominant kernel from the in-library hipEvent timings.  Algorithmic flops per launch
    (SURVEY.md 8d): marginal pass 2 L n M^2, accumulation L n M^2 (n = local points)."""
    flops = (2.0 * L * n_loc * M * M, 1.0 * L * n_loc * M * M)
    nbk = Mp // 128
    msplit = marginal == "f16x2-factor"
    # executed flops: factor-form marginal kernel -- a wave (64 rows) stops at its own diagonal and skips the all-zero upper
    # half of its last stage -> (1 + 32 / M) M^2 per point; accumulation -- lower tile pairs only, on a diagonal tile the
    # sub-tile above the diagonal idles and (split tile kernel) the two diagonal sub-tiles skip their upper 32 x 32 block
    ex_m = (1.0 + 32.0 / Mp) if (marginal == "f16x2-factor" and Mp % 256 == 0) else (1.0 + 1.0 / nbk)
    image_acc = accumulate == "f16x2" and Mp % 256 == 0  # syrk_strip_kernel (agpl_syrk.hip): 16 x 16 blocks on or below the diagonal
    ex_s = (1.0 + 16.0 / Mp) if image_acc else (nbk + (0.25 if accumulate == "f16x2" else 0.5)) / nbk
    executed = (ex_m * L * n_loc * Mp * Mp, ex_s * L * n_loc * Mp * Mp)
    # kernel names as the library picks them: the plan's marginal pass runs resident workgroups on 16x16x32 MFMA serving per-XCD
    # item queues (agpl_split.hip), its accumulation reads the point-major image (syrk_strip_kernel, agpl_syrk.hip); the
    # float32-input pair is marginal_kernel<0> / syrk_kernel (agpl_mfma.hip)
    names = ("marginal_factor_queue_kernel" if msplit else "marginal_kernel<0>",
             "syrk_strip_kernel" if accumulate == "f16x2" else "syrk_kernel")
    mult = (3.0 if msplit else 1.0, 3.0 if accumulate == "f16x2" else 1.0)
    peaks = (PEAK_F16_MFMA_TFLOPS if msplit else PEAK_F32_MFMA_TFLOPS,
             PEAK_F16_MFMA_TFLOPS if accumulate == "f16x2" else PEAK_F32_MFMA_TFLOPS)
    per = []
    for (ms, cnt), fl, ex, nm, mu, pk in zip(kt, flops, executed, names, mult, peaks):
        avg = ms / max(cnt, 1)
        per.append({"kernel": nm, "avg_ms": round(avg, 4), "launches": cnt,
                    "algorithmic_tflops": round(fl / (avg * 1e-3) / 1e12, 2) if avg > 0 else None,
                    "mfma_dtype": "f16 (hi/lo split, x3 products)" if mu == 3.0 else "f32",
                    "executed_mfma_tflops": round(mu * ex / (avg * 1e-3) / 1e12, 2) if avg > 0 else None,
                    "peak_tflops": pk,
                    "executed_frac_of_peak": round(mu * ex / (avg * 1e-3) / 1e12 / pk, 4) if avg > 0 else None})
    dom = 0 if kt[0][0] >= kt[1][0] else 1
    achieved = per[dom]["algorithmic_tflops"]
    # HBM/fabric bytes per launch come from a separate rocprofv3 --pmc pass (cannot be taken inside this process):
    # the committed summary is attached when it was collected on exactly this configuration, else null.
    traffic = None
    profile = profile or PMC_PROFILE
    if traffic_key is not None and world == 1:
        try:
            with open(os.path.join(ROOT, "profiles", profile)) as fh:
                pm = json.load(fh)
            if pm["config"] == traffic_key:
                traffic = pm["kernels"].get(names[dom], {}).get("traffic_bytes")
        except Exception:
            traffic = None
    return {"kernel": names[dom], "bound": "mfma", "achieved": achieved, "peak": peaks[dom], "unit": "TFLOP/s",
            "frac": round(achieved / peaks[dom], 4) if achieved else None,
            "mfma_products_per_algorithmic_product": mult[dom],
            "traffic": traffic, "traffic_source": "profiles/" + profile if traffic else None, "kernels": per,
            "sweep_algorithmic_tflops": round(3.0 * L * N * M * M / (ms_per_step * 1e-3) / 1e12 / world, 2)}


def sustained_f16(ctx, roofline):
    """The float16 MFMA rate this device SUSTAINS (agpl_probe_mfma: the accumulation kernel's own instruction mix, 12
    MFMAs + 8 LDS fragment reads per step, nothing else, four waves per SIMD, ~40 ms of back-to-back launches): under
    matrix load the clock settles well below the boost clock the 2.5 PFLOP/s data-sheet peak assumes, so this -- not
    `peak` -- is what a perfect kernel of this shape could execute.  Added next to `peak`/`frac`, which stay the guide's."""
    import ctypes as C
    from agpl_amd import _ffi

    tf, ms = C.c_double(0), C.c_double(0)
    _ffi.check(ctx.bind(), _ffi.lib().agpl_probe_mfma(ctx.bind(), C.c_int32(_ffi.F32), C.c_int32(5000), C.c_int32(1), C.c_int32(4),
                                                          C.byref(tf), C.byref(ms)))
    roofline["sustained_mfma_f16"] = {
        "tflops": round(tf.value, 1), "frac_of_peak": round(tf.value / PEAK_F16_MFMA_TFLOPS, 3),
        "probe": "agpl_probe_mfma(float16, mode 1: 12 v_mfma_f32_32x32x16_f16 + 8 ds_read_b128 per step, 4 waves/SIMD), "
                 "6 launches of %.1f ms timed as one region" % ms.value}
    # the shape the shipped kernels issue (16x16x32, one stage of the marginal kernel's wave, hashed operands): the ceiling the
    # executed fractions below are taken against
    tf2, ms2 = C.c_double(0), C.c_double(0)
    _ffi.check(ctx.bind(), _ffi.lib().agpl_probe_mfma(ctx.bind(), C.c_int32(_ffi.F32), C.c_int32(3000), C.c_int32(3), C.c_int32(2),
                                                          C.byref(tf2), C.byref(ms2)))
    roofline["sustained_mfma_f16_16x16x32"] = {
        "tflops": round(tf2.value, 1), "frac_of_peak": round(tf2.value / PEAK_F16_MFMA_TFLOPS, 3),
        "probe": "agpl_probe_mfma(float16, mode 3: 48 v_mfma_f32_16x16x32_f16 + 16 ds_read_b128 per step, hashed operands, "
                 "2 waves/SIMD), 6 launches of %.1f ms timed as one region" % ms2.value}
    ceiling = max(tf.value, tf2.value)
    for k in roofline["kernels"]:
        if k.get("executed_mfma_tflops") and k["mfma_dtype"].startswith("f16"):
            k["executed_frac_of_sustained"] = round(k["executed_mfma_tflops"] / ceiling, 4)


def parity_slice(A, ctx, lik, likname, Phi, kd, y, marginal, accumulate, nsweeps=10, ns=20_000):
    """GPU vs oracle on a slice of the same workload: `nsweeps` full CAVI sweeps (10), natural parameters
    compared at the end (and after the first sweep)."""
    import torch
    from oracle import oracle as O

    olik = make_olik(O, likname)
    ns = min(ns, Phi.shape[0])
    Mp, L = Phi.shape[1], A.nlatent(lik)
    Phi_s, kd_s, y_s = Phi[:ns].contiguous(), kd[:ns].contiguous(), y[:ns].contiguous()
    cs = A.SparseCAVI(lik, Phi_s, kd_s, y_s, ctx=ctx, marginal_precision=marginal, accumulate_precision=accumulate)
    Ph, kh, yh = Phi_s.cpu().numpy(), kd_s.cpu().numpy().astype(np.float64), y_s.cpu().numpy()
    if lik.ykind == "real":
        yh = yh.astype(np.float64)
    S, m = np.tile(np.eye(Mp), (L, 1, 1)), np.zeros((L, Mp))
    first = None
    for it in range(nsweeps):
        cs.sweep()
        G, g = O.cavi_pass(olik, Ph, kh, yh, -S, m)
        S, m = O.gaussian_update(G, g)
        if it == 0:
            torch.cuda.synchronize()
            first = (float(np.abs(cs.G.cpu().numpy() - G).max() / np.abs(G).max()),
                     float(np.abs(cs.g.cpu().numpy() - g).max() / np.abs(g).max()))
    cs.check()
    dG = float(np.abs(cs.G.cpu().numpy() - G).max() / np.abs(G).max())
    dg = float(np.abs(cs.g.cpu().numpy() - g).max() / np.abs(g).max())
    return {"max_rel_dG": dG, "max_rel_dg": dg, "after_first_sweep": {"max_rel_dG": first[0], "max_rel_dg": first[1]},
            "points": ns, "sweeps": nsweeps, "tolerance": 1e-5,
            "pass": bool(max(dG, dg, first[0], first[1]) < 1e-5)}


def full_size_quadratic_check(Phi, gamma, beta, G, g, nvec=4, seed=7):
    """Size-independent properties of (G, g) = (Phi Diag(gamma) Phi', Phi beta) at ANY N, against float64 torch reductions
    over the same gamma, beta (docs/src/index.md:154-163 in the whitened basis): g, tr G, and v'Gv = sum_n gamma_n
    (phi_n . v)^2 for `nvec` random v -- O(N M) each, and unlike the trace they see every off-diagonal tile."""
    import torch

    L, n = gamma.shape
    Mp = Phi.shape[1]
    gen = torch.Generator(device="cuda").manual_seed(seed)
    V = torch.randn((Mp, nvec), dtype=torch.float64, device="cuda", generator=gen)
    step = max(1, (1 << 29) // Mp)
    rel_g = rel_tr = rel_q = 0.0
    for l in range(L):
        tr = torch.zeros((), dtype=torch.float64, device="cuda")
        gref = torch.zeros(Mp, dtype=torch.float64, device="cuda")
        qref = torch.zeros(nvec, dtype=torch.float64, device="cuda")
        for j0 in range(0, n, step):
            P = Phi[j0:j0 + step].double()
            gm = gamma[l, j0:j0 + step].double()
            tr += (gm * (P * P).sum(1)).sum()
            gref += P.T @ beta[l, j0:j0 + step].double()
            qref += (gm[:, None] * (P @ V) ** 2).sum(0)
            del P
        q = ((G[l] @ V) * V).sum(0)
        rel_g = max(rel_g, float(((g[l] - gref).abs().max() / gref.abs().max().clamp_min(1e-300)).item()))
        rel_tr = max(rel_tr, float(((torch.diagonal(G[l]).sum() - tr).abs() / tr.abs().clamp_min(1e-300)).item()))
        rel_q = max(rel_q, float(((q - qref).abs() / qref.abs().clamp_min(1e-300)).max().item()))
    return {"max_rel_dg": rel_g, "rel_d_trace_G": rel_tr, "max_rel_d_vGv": rel_q, "quadratic_forms": nvec,
            "G_symmetric": bool(torch.equal(G, G.transpose(1, 2)))}


def marginal_sample_indices(n, seed=11, ntiles=64, nrandom=4096):
    """Point indices for a sampled check of the marginal kernel at any N: whole 128-point tiles -- one from each of `ntiles`
    strata of the tile range, its index chosen so that every residue mod 8 (= the kernel's eight per-XCD item queues, tile t in
    queue t mod 8) occurs --, the first tile, the last full tile and the ragged tail tile, plus `nrandom` single points."""
    import torch

    T = (n + 127) // 128
    rng = np.random.default_rng(seed)
    tiles = {0, max(T - 1, 0), max(T - 2, 0)}
    for k in range(ntiles):
        lo, hi = k * T // ntiles, max((k + 1) * T // ntiles, k * T // ntiles + 1)
        t = int(rng.integers(lo, hi))
        t = min(T - 1, t - (t % 8) + (k % 8)) if hi - lo >= 8 else t
        tiles.add(max(t, 0))
    idx = [np.arange(t * 128, min(n, t * 128 + 128)) for t in sorted(tiles)]
    idx.append(rng.integers(0, n, size=nrandom))
    idx = np.unique(np.concatenate(idx))
    return torch.from_numpy(idx).cuda(), sorted(tiles)


def full_size_marginal_check(cavi, Phi, mu=None, var=None):
    """The marginal kernel of the plan path at ANY N with the REAL posterior factor: after >= 1 update, q(v) = (U, v) is pulled from
    the plan (agpl_plan_state: U float64, column-major lower triangle), and mu_n = sum_a v_a T[a,n], var_n = d_n + sum_a T[a,n]^2,
    T = U Phi (the `marginals(post_u(x))` of examples/bernoulli/script.jl:32-33 in factor form, include/agpl.h) is evaluated in
    float64 from the float32 feature rows of a sample of points (marginal_sample_indices) and compared with agpl_marginals_plan's
    float32 mu, var.  Then gamma, beta of those points (if the object exports them) against the float64 Bernoulli / NegBin / ...
    operators is left to the tests; here: max |d mu| / max |mu|, max |d var| / max |var|."""
    import torch

    assert cavi.plan is not None, "full_size_marginal_check is for the plan path"
    cavi.check()
    if mu is None:
        mu, var = cavi.marginals()
    idx, tiles = marginal_sample_indices(cavi.N)
    L = cavi.L
    rel_mu = rel_var = 0.0
    P = Phi[idx].double()
    d = cavi.kdiag[idx].double().clamp_min(0.0)  # (the plan stores round-off below zero as 0)
    for l in range(L):
        Ut = torch.triu(cavi.plan.U_lead[l])  # row-major view of the column-major lower triangle = U' (the caller's M x M block)
        T = P @ Ut  # T[n, a] = sum_b phi_n[b] U[a][b]
        mref = T @ cavi.plan.v_lead[l]
        vref = d + (T * T).sum(1)
        if cavi.mu0 is not None:
            mref = mref + cavi.mu0[l][idx].double()
        rel_mu = max(rel_mu, float(((mu[l][idx].double() - mref).abs().max() / mref.abs().max().clamp_min(1e-300)).item()))
        rel_var = max(rel_var, float(((var[l][idx].double() - vref).abs().max() / vref.abs().max().clamp_min(1e-300)).item()))
    U0 = cavi.plan.U_lead[0]
    offdiag = float((torch.triu(U0, 1).abs().max()).item())
    return {"max_rel_d_mu": rel_mu, "max_rel_d_var": rel_var, "sampled_points": int(idx.numel()), "sampled_tiles": len(tiles),
            "tile_residues_mod_8": sorted({t % 8 for t in tiles}), "last_tile_points": int(cavi.N - (tiles[-1]) * 128),
            "max_abs_offdiag_U": offdiag, "sweeps_before": int(cavi.nsweeps)}


def f32_contract_leg(A, ctx, lik, likname, Phi, kd, y, N, M, Mp, L, args):
    """The same C2 sweep at the arithmetic SURVEY.md 8(d) prices: float32 features contracted by v_mfma_f32_32x32x2_f32
    (agpl_cavi_pass: marginal_kernel<0> + syrk_kernel of agpl_mfma.hip), float64 reductions and M x M update as everywhere
    (the reference computes in Float64, src/generic.jl:36-38; 8(d) contracts float32 K_ZX against a 157.3 TFLOP/s roof)."""
    import torch

    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f32", accumulate_precision="f32")
    steps = 3
    dt, kt = timed_sweeps(ctx, cavi, steps, 1, torch.cuda.synchronize)
    ms = dt / steps * 1e3
    roof = roofline_of(kt, L, N, M, Mp, "f32", "f32", ms, 1, N)
    # SURVEY 8(d) counts the variance projection as a dense product (2 N M^2); marginal_kernel<0> uses the symmetry of W (packed
    # triangle, doubled off-diagonal) and executes (1 + 128 / M) N M^2 -- its algorithmic rate can exceed the float32 MFMA roof.
    # `frac` here is therefore the EXECUTED MFMA rate of the dominant kernel over the roof; the algorithmic one is kept beside it.
    dom = max(roof["kernels"], key=lambda k: k["avg_ms"])
    roof["frac_of_algorithmic_flops"] = roof["frac"]
    roof["frac"] = dom["executed_frac_of_peak"]
    roof["achieved_executed"] = dom["executed_mfma_tflops"]
    out = {"config": {"workload": f"{likname}-logistic SVGP CAVI sweep, N={N}, M={M}, L={L}, 1 GPU, float32-input MFMA kernels"},
           "dtype": "f32", "value": round(steps / dt, 4), "unit": "sweeps/s", "ms_per_step": round(ms, 3), "steps": steps,
           "warmup": 1, "roofline": roof}
    del cavi
    if not args.no_parity:
        out["parity"] = parity_slice(A, ctx, lik, likname, Phi, kd, y, "f32", "f32")
    return out


def elbo_leg(A, ctx, lik, Phi, kd, y, base_ms, args, plain=None):
    """aug_elbo (examples/bernoulli/script.jl:65-70) riding the sweep: the per-point terms in the pass's one
    per-point kernel, the Gaussian KL behind the update; what it adds to a sweep, and the values of the last sweeps.  The
    difference of two ~14 ms sweeps measured minutes apart moves by +-0.15 ms with the device's clock: the plain sweep (`plain`,
    the headline run's object) is timed again right behind the ELBO run, and the added time is taken against the mean of the two
    plain timings that bracket it."""
    import torch

    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, track_elbo=True)
    steps = max(5, args.steps)
    dt, _ = timed_sweeps(ctx, cavi, steps, 2, torch.cuda.synchronize)
    vals = [cavi.elbo_entering()]
    for _ in range(2):
        cavi.sweep()
        vals.append(cavi.elbo_entering())
    ms = dt / steps * 1e3
    after_ms = None
    if plain is not None:
        dt2, _ = timed_sweeps(ctx, plain, steps, 1, torch.cuda.synchronize)
        after_ms = dt2 / steps * 1e3
        base_ms = 0.5 * (base_ms + after_ms)
    return {"ms_per_step_with_elbo": round(ms, 3), "added_ms_per_step": round(ms - base_ms, 3), "steps": steps,
            "plain_ms_per_step_before_and_after": [round(2 * base_ms - after_ms, 3), round(after_ms, 3)] if after_ms else None,
            "elbo_entering_last_sweeps": vals,
            "non_decreasing": bool(all(b >= a - 1e-9 * abs(a) for a, b in zip(vals, vals[1:])))}


def config_leg(A, ctx, likname, N, M, steps=5, gibbs=False, parity_points=10_000, no_parity=False, workload=None, label=None,
               warmup=1):
    """One more BASELINE configuration on this GPU with the headline's timed-loop discipline: `steps` plan sweeps after `warmup`
    untimed ones (the N/8 legs take 8: their sweeps are 2-6 ms and the first five after the set-up run up to 14 % slower --
    kernel trace of round 5: 6.84, 6.44, 6.22, 6.06, 6.02, 6.00, 6.00 ms -- while the device clocks come up), the two contraction kernels from the in-library events (roofline.kernels), a ten-sweep parity slice against the
    oracle, optionally the sparse Gibbs sweep on the same plan.  `workload`: (y, Phi, kd) to reuse instead of building one."""
    import torch

    lik = make_lik(A, likname)
    L = A.nlatent(lik)
    t0 = time.time()
    y, Phi, kd = workload if workload is not None else build_workload(A, ctx, lik, 0, N, M)
    Mp = Phi.shape[1]
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx)
    t_setup = time.time() - t0
    dt, kt = timed_sweeps(ctx, cavi, steps, warmup, torch.cuda.synchronize)
    ms = dt / steps * 1e3
    out = {"config": {"workload": label or f"{likname} SVGP CAVI sweep, N={N}, M={M} (padded {Mp}), L={L}, 1 GPU", "N": N, "M": M, "L": L},
           "value": round(steps / dt, 4), "unit": "sweeps/s", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
           "roofline": roofline_of(kt, L, N, M, Mp, "f16x2-factor", "f16x2", ms, 1, N), "setup_s": round(t_setup, 2)}
    if gibbs:
        yg = y.to(torch.float64) if lik.ykind == "real" else y
        gib = A.SparseGibbs(lik, Phi, kd, yg, ctx=ctx, plan=cavi.plan)
        gib.sweep()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(steps):
            gib.sweep()
        torch.cuda.synchronize()
        out["gibbs_ms_per_sweep"] = round((time.perf_counter() - tg) / steps * 1e3, 3)
        del gib
    del cavi
    if not no_parity:
        out["parity"] = parity_slice(A, ctx, lik, likname, Phi, kd, y, "f16x2-factor", "f16x2", ns=parity_points)
    return out, (y, Phi, kd)


def m1024_leg(A, ctx, args):
    """BASELINE.json north_star's target configuration: Bernoulli-logistic CAVI, N = 1e7, M = 1024, 1 GPU -- same
    timed-loop discipline as the headline value, its own roofline, and a 10-sweep parity slice."""
    import torch

    lik = make_lik(A, "bernoulli")
    N, M = args.m1024_n, 1024
    t0 = time.time()
    y, Phi, kd = build_workload(A, ctx, lik, 0, N, M)
    Mp = Phi.shape[1]
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, marginal_precision="f16x2-factor", accumulate_precision="f16x2")
    t_setup = time.time() - t0
    steps = max(3, min(args.steps, 6))
    dt, kt = timed_sweeps(ctx, cavi, steps, 1, torch.cuda.synchronize)
    ms = dt / steps * 1e3
    out = {"config": {"workload": f"bernoulli-logistic SVGP CAVI sweep, N={N}, M={M}, L=1, 1 GPU", "N": N, "M": M, "L": 1,
                      "marginal_pass": "f16x2-factor", "accumulate_pass": "f16x2"},
           "value": round(steps / dt, 4), "unit": "sweeps/s", "ms_per_step": round(ms, 3), "steps": steps, "warmup": 1,
           "roofline": roofline_of(kt, 1, N, M, Mp, "f16x2-factor", "f16x2", ms, 1, N,
                                   traffic_key={"lik": "bernoulli", "N": N, "M": M, "L": 1}, profile=PMC_PROFILE_M1024),
           "setup_s": round(t_setup, 2),
           "hbm_gb": {"plan": round(cavi.plan.nbytes / 1e9, 2) if getattr(cavi, "plan", None) is not None else None,
                      "float32_features": round(Phi.numel() * 4 / 1e9, 2)}}
    if not args.no_parity:
        # the same full-size self-check as the headline's, at the north-star size (the accumulation runs 8192-point slices here:
        # agpl_slice_points): g, tr G and four quadratic forms against float64 reductions over the exported gamma, beta, and the
        # sampled marginals with the real q(v) of the timed sweeps
        cavi.gamma = torch.empty((1, N), dtype=torch.float32, device="cuda")
        cavi.beta = torch.empty((1, N), dtype=torch.float32, device="cuda")
        cavi.c = torch.empty((N,), dtype=torch.float32, device="cuda")
        cavi.accumulate()
        torch.cuda.synchronize()
        chk = full_size_quadratic_check(Phi, cavi.gamma, cavi.beta, cavi.G, cavi.g)
        cavi.gamma = cavi.beta = cavi.c = None
        mchk = full_size_marginal_check(cavi, Phi)
        out["full_size_check"] = {"points": N, **chk, "marginals": mchk, "tolerance": {"G_g": 2e-6, "marginals": 2e-5},
                                  "pass": bool(chk["max_rel_dg"] < 2e-6 and chk["rel_d_trace_G"] < 2e-6 and chk["max_rel_d_vGv"] < 2e-6
                                               and chk["G_symmetric"] and mchk["max_rel_d_mu"] < 2e-5 and mchk["max_rel_d_var"] < 2e-5)}
    del cavi
    if not args.no_parity:
        out["parity"] = parity_slice(A, ctx, lik, "bernoulli", Phi, kd, y, "f16x2-factor", "f16x2", ns=10_000)
    if not args.no_extra:
        # BASELINE configs[2] (C3: NegBin r = 15, N = 1e7, M = 1024) on ONE GPU, on the same features (x_i depends on (seed, i)
        # only): the numerator of c3r's projected 8-GPU scaling
        try:
            nlik = make_lik(A, "negbin")
            _, yn = A.synth_xy(nlik, SEED, 0, N, ctx=ctx, want_x=False)
            torch.cuda.empty_cache()
            c3, _ = config_leg(A, ctx, "negbin", N, M, steps=3, no_parity=True, workload=(yn, Phi, kd),
                               label=f"C3 on one GPU: NegBin(r=15) SVGP CAVI sweep, N={N}, M={M}, L=1")
            out["c3_full_one_gpu"] = c3
            del yn
        except Exception as e:
            out["c3_full_one_gpu"] = {"error": f"{type(e).__name__}: {e}"}
    del Phi, kd, y
    torch.cuda.empty_cache()
    return out


def c5_leg(A, args):
    """BASELINE configs[4]: StudentTLikelihood Gibbs path, aug_sample (Gamma) + the full-rank N = 65 536 conditional
    solve of examples/studentt/script.jl (gibbs_sample, :76-87 of the bernoulli example), float64.  The N^3 / 3 of the
    Cholesky is the hand-written float64-MFMA trailing update of agpl_dense.hip (diagonal blocks and panel solves are
    rocSOLVER / rocBLAS calls); the rate is priced against the float64 MFMA rate measured by the library's own
    probe kernel on this device."""
    import torch
    from agpl_amd import _ffi

    N = args.c5_n
    ctx = A.Context(0, seed=SEED)
    lik = A.StudentTLikelihood(3.5, 2.0)  # examples/studentt/script.jl:17-19
    x, y32 = A.synth_xy(lik, SEED, 0, N, ctx=ctx)
    x, order = torch.sort(x)
    y = y32.to(torch.float64)[order].contiguous()
    K = torch.empty((N, N), dtype=torch.float64, device="cuda")
    for r0 in range(0, N, 4096):  # row blocks: no N x N temporaries
        blk = K[r0:min(N, r0 + 4096)]
        torch.sub(x[r0:r0 + 4096, None], x[None, :], out=blk)
        blk.div_(2.0).pow_(2).mul_(-0.5).exp_()  # with_lengthscale(SqExponentialKernel(), 2.0), script.jl:15
    K.diagonal().add_(1e-6)  # LatentGP(gp, lik, 1e-6), script.jl:18
    dg = A.DenseGibbs(lik, K, y, ctx=ctx)
    dg.sweep()
    torch.cuda.synchronize()
    steps = 2
    t0 = time.perf_counter()
    for _ in range(steps):
        dg.sweep()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    peak = C.c_double()
    _ffi.check(ctx.bind(), _ffi.lib().agpl_probe_mfma(ctx.bind(), C.c_int32(_ffi.F64), C.c_int32(4096), C.c_int32(0), C.c_int32(1), C.byref(peak), None))
    tf = N ** 3 / 3 / dt / 1e12
    out = {"config": {"workload": f"StudentT(3.5, 2.0) full-rank Gibbs step, N={N}, SE kernel lengthscale 2.0, jitter 1e-6"},
           "value": round(1.0 / dt, 4), "unit": "sweeps/s", "ms_per_step": round(dt * 1e3, 1), "steps": steps, "dtype": "f64",
           "n3_work": "block by block (1024 wide), one stream: U_k = chol(D_k)^-1 by the sparse sweep's one-launch factorisation "
                      "(factor_pipe_kernel), panel <- panel U_k' and the trailing update (the N^3/3) on the hand-written float64-MFMA "
                      "tile routine (agpl_dense.hip: gemm_nt_assign_kernel, trailing_update_kernel on a 1-D grid of the lower-triangle "
                      "tiles); solves with the kept U_k (dtrmv) and one dgemv per panel",
           "roofline": {"bound": "mfma", "achieved": round(tf, 2), "unit": "TFLOP/s (N^3/3 per sweep, float64)",
                        "peak": round(peak.value, 1), "peak_source": "agpl_probe_mfma(float64): v_mfma_f64_16x16x4_f64 "
                        "back-to-back on every SIMD, measured on this device in this run",
                        "frac": round(tf / peak.value, 4) if peak.value > 0 else None},
           "f_finite": bool(torch.isfinite(dg.f).all().item()),
           "hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
    del dg, K
    torch.cuda.empty_cache()
    return out


def gather_over_ranks(mine, dt, group, device):
    """What a multi-rank leg reports: every rank's record (all_gather_object) and the MAX over ranks of the timed region.
    `device`: where the reduced scalar lives ("cuda" under RCCL, "cpu" under gloo).  Plain torch.distributed, no GPU needed
    (tests/test_bench_launcher_cpu.py runs it with 2 and 8 gloo ranks)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine, group=group)
    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
    return per_rank, float(tt.item())


def sharded_leg(A, ctx, likname, N, M, rank, world, group, barrier, steps=10, warmup=2, gibbs_steps=5, parity_points=5_000,
                workload=None, label=None, red_device="cuda"):
    """One more BASELINE configuration on the ranks of an N > 1 run: N points sharded with shard_range, one
    all-reduce of L (M^2 + M) float64 per sweep -- CAVI `steps` sweeps after `warmup`, then `gibbs_steps` sparse Gibbs sweeps on the
    same plan (global point index = this rank's offset), timed with the headline's barrier + MAX-over-ranks discipline; rank 0
    adds a ten-sweep oracle parity slice of its own shard.  Every rank calls this; rank 0 gets the object, the others None.
    `workload`: this rank's (Phi, kd) to reuse (the features depend on (seed, point index) only)."""
    import torch

    lik = make_lik(A, likname)
    L = A.nlatent(lik)
    i0, i1 = A.shard_range(N, rank, world)
    n_loc = i1 - i0
    t0 = time.time()
    if workload is None:
        y, Phi, kd = build_workload(A, ctx, lik, i0, n_loc, M)
    else:
        Phi, kd = workload
        _, y = A.synth_xy(lik, SEED, i0, n_loc, ctx=ctx, want_x=False)
    Mp = Phi.shape[1]
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, group=group)
    t_setup = time.time() - t0
    dt, kt = timed_sweeps(ctx, cavi, steps, warmup, barrier)
    xms = getattr(cavi, "exchange_ms", (0.0, 0.0))
    mine = {"rank": rank, "points": n_loc, "ms_per_step": round(dt / steps * 1e3, 3), "allreduce_ms_avg": round(xms[0], 4),
            "allreduce_ms_max": round(xms[1], 4), "marginal_kernel_ms": round(kt[0][0] / max(kt[0][1], 1), 4),
            "accumulate_kernel_ms": round(kt[1][0] / max(kt[1][1], 1), 4), "setup_s": round(t_setup, 2)}
    per_rank, dt = gather_over_ranks(mine, dt, group, red_device)
    gibbs_ms = None
    if gibbs_steps > 0:
        yg = y.to(torch.float64) if lik.ykind == "real" else y
        gib = A.SparseGibbs(lik, Phi, kd, yg, ctx=ctx, group=group, plan=cavi.plan, point_offset=i0)
        gib.sweep()
        barrier()
        tg = time.perf_counter()
        for _ in range(gibbs_steps):
            gib.sweep()
        barrier()
        _, tg = gather_over_ranks(None, time.perf_counter() - tg, group, red_device)
        gibbs_ms = round(tg / gibbs_steps * 1e3, 3)
        del gib
    del cavi
    parity = None
    if rank == 0 and parity_points > 0:
        parity = parity_slice(A, ctx, lik, likname, Phi, kd, y, "f16x2-factor", "f16x2", ns=parity_points)
    barrier()
    if rank != 0:
        return None, (Phi, kd)
    ms = dt / steps * 1e3
    steps_ms = [r["ms_per_step"] for r in per_rank]
    out = {"config": {"workload": label or f"{likname} SVGP CAVI sweep, N={N}, M={M} (padded {Mp}), L={L}, N sharded over {world} ranks, "
                                           f"1 all-reduce of L*(M^2+M) f64 per sweep", "N": N, "M": M, "L": L,
                      "parallelism": f"N-shard x{world}"},
           "value": round(steps / dt, 4), "unit": "sweeps/s", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
           "world": world, "n_gpus": world,
           "allreduce_ms": round(sum(r["allreduce_ms_avg"] for r in per_rank) / world, 4), "allreduce_bytes": 8 * L * (Mp * Mp + Mp),
           "ms_per_step_min_rank": min(steps_ms), "ms_per_step_max_rank": max(steps_ms), "per_rank": per_rank,
           "roofline": roofline_of(kt, L, n_loc, M, Mp, "f16x2-factor", "f16x2", ms, world, N)}
    if gibbs_ms is not None:
        out["gibbs_ms_per_sweep"] = gibbs_ms
        out["gibbs_steps"] = gibbs_steps
    if parity is not None:
        out["parity"] = parity
    return out, (Phi, kd)


def summarize(out):
    """The LAST key of the line, <= 1.5 KB: every configuration's driver-timed number in one compact object (the driver's record
    keeps the headline object and only the last ~2 KB of the line -- VERDICT r5 item 2).  Per leg: ms = ms per step (sweep),
    sps = sweeps/s, frac = the dominant kernel's algorithmic fraction of the guide's peak, k = [marginal, accumulation] kernel ms,
    par = max relative error of (G, g) after ten sweeps against the oracle, gibbs_ms = ms per sparse Gibbs sweep,
    proj8 = projected 8-GPU scaling (full-N ms / per-rank ms, before the all-reduce), ar_ms = all-reduce ms per sweep (N > 1)."""
    def r3(x):
        return None if x is None else float(f"{x:.3g}")

    def leg(o):
        if not isinstance(o, dict):
            return None
        if "error" in o:
            return {"error": str(o["error"])[:60]}
        if "ms_per_step" not in o:
            return None
        rf = o.get("roofline") or {}
        s = {"ms": o["ms_per_step"], "sps": r3(o.get("value")), "frac": rf.get("frac")}
        ks = rf.get("kernels")
        if ks:
            s["k"] = [round(k["avg_ms"], 2) for k in ks]
        p = o.get("parity")
        if p:
            s["par"] = r3(max(p["max_rel_dG"], p["max_rel_dg"]))
        if "gibbs_ms_per_sweep" in o:
            s["gibbs_ms"] = o["gibbs_ms_per_sweep"]
        if o.get("projected_scaling_8") is not None:
            s["proj8"] = o["projected_scaling_8"]
        if "allreduce_ms" in o:
            s["ar_ms"] = o["allreduce_ms"]
            s["ranks"] = o.get("world", o.get("n_gpus"))
        return s

    sm = {"c2": leg(out)}
    for name in ("m1024", "n8", "n8_m1024", "c3r", "c3", "c4", "c5", "f32_contract"):
        if name in out:
            sm[name] = leg(out[name])
    c3f = out.get("m1024", {}).get("c3_full_one_gpu") if isinstance(out.get("m1024"), dict) else None
    if c3f is not None:
        sm["c3_full_one_gpu"] = leg(c3f)
    g = out.get("gibbs")
    if isinstance(g, dict):
        sm["gibbs"] = {"ms": g.get("ms_per_sweep"), "pg1_per_s": r3((g.get("sampler") or {}).get("pg1_draws_per_s")),
                       "pg1_per_s_negbin": r3((g.get("sampler_negbin") or {}).get("pg1_draws_per_s"))}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        sm["cpu"] = {"sps": r3(cb.get("value")), "cores": cb.get("cores"), "gpu_over_cpu": cb.get("gpu_over_cpu")}
    fs = out.get("full_size_check")
    if isinstance(fs, dict):
        sm["full_size_pass"] = fs.get("pass")
    sm["n_gpus"] = out.get("n_gpus")
    return {k: v for k, v in sm.items() if v is not None}


def visible_gpu_count():
    """GPUs visible to a rank, counted WITHOUT loading a HIP runtime into this process: the launching parent goes on to
    start torch.distributed.run, and a process that has initialised the GPU must not start (exec) other programs on this
    pool.  torch.cuda.device_count() only avoids hipInit when amdsmi answers; otherwise it falls back to hipGetDeviceCount.
    So: a throw-away child counts with torch (it honours HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES exactly as the ranks
    will) and exits; if that child fails, the kfd topology in sysfs is read (GPU nodes have simd_count > 0)."""
    import glob
    import subprocess

    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                           text=True, timeout=600)
        if r.returncode == 0:
            return int(r.stdout.strip().splitlines()[-1])
    except Exception:
        pass
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(ln.split(None, 1) for ln in open(f).read().splitlines() if " " in ln)
            n += int(props.get("simd_count", "0")) > 0
        except Exception:
            pass
    return n


def self_launch(ngpus):
    """Run this script under torch.distributed.run with one rank per GPU as a child process; returns its exit code
    (2: fewer than `ngpus` devices are visible).  This parent never loads torch or a HIP runtime: the devices are counted
    by a throw-away child (visible_gpu_count)."""
    import socket
    import subprocess

    single_dev = os.environ.get("AGPL_BENCH_SINGLE_DEVICE") == "1"  # test hook, see main()
    have = visible_gpu_count()
    if have < (1 if single_dev else ngpus):
        print(f"[bench] --gpus {ngpus} but only {have} device(s) are visible: refusing to measure fewer", file=sys.stderr,
              flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    # (--points / --inducing: spellings that torch.distributed.run does not mistake for its own --n*/--m* options)
    ap.add_argument("--n", "--points", dest="n", type=int, default=10_000_000,
                    help="total observations (sharded over ranks)")
    ap.add_argument("--m", "--inducing", dest="m", type=int, default=512, help="inducing points")
    ap.add_argument("--lik", default="bernoulli")
    ap.add_argument("--cpu-sample", type=int, default=250_000, help="points of the CPU-baseline cross-check sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-gibbs", action="store_true")
    ap.add_argument("--no-m1024", action="store_true", help="skip the north-star target leg (Bernoulli N=1e7 M=1024)")
    ap.add_argument("--m1024-n", type=int, default=10_000_000)
    ap.add_argument("--no-c5", action="store_true", help="skip the full-rank StudentT Gibbs leg (BASELINE configs[4])")
    ap.add_argument("--no-f32", action="store_true", help="skip the float32-MFMA leg (the sweep at SURVEY 8d's stated arithmetic)")
    ap.add_argument("--no-elbo", action="store_true", help="skip the ELBO-riding-the-sweep leg")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the per-rank / other-configuration legs (n8, c3r, c4) and the full-N CPU sweep")
    ap.add_argument("--cpu-full", dest="cpu_full", action="store_true", default=None,
                    help="time ONE full-N sweep of the CPU twin (default: on for the default configuration)")
    ap.add_argument("--c5-n", type=int, default=65_536)
    ap.add_argument("--sharded-legs", choices=["auto", "on", "off"], default="auto",
                    help="N > 1 runs: BASELINE C3 (NegBin r=15, M=1024, CAVI + Gibbs) and the Bernoulli M=1024 north-star target on the "
                         "same ranks after the headline (auto: on for the default configuration)")
    ap.add_argument("--sharded-m", type=int, default=1024, help="inducing points of the sharded legs")
    ap.add_argument("--accumulate", default="f16x2", choices=["f32", "f16x2"],
                    help="K_ZX diag(gamma) K_XZ accumulation: f32-input MFMA, or split-float16 MFMA")
    ap.add_argument("--marginal", default="auto", choices=["auto", "f32", "f16x2-factor"],
                    help="marginal pass: f32-input MFMA (with --accumulate f32), or the plan's split-float16 factor form "
                         "(3 f16 products per f32 product; needs padded M %% 256 == 0; auto picks it when it applies)")
    args = ap.parse_args()
    default_config = (args.n, args.m, args.lik) == (10_000_000, 512, "bernoulli")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL over
        # xGMI) as a CHILD of this process -- before anything here has touched a GPU -- and relay rank 0's JSON line and
        # the exit code.  A run that asks for N GPUs can therefore never silently measure one.
        raise SystemExit(self_launch(args.gpus))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # AGPL_BENCH_SINGLE_DEVICE=1 is a test hook for 1-GPU boxes: every rank uses cuda:0 and the exchange goes over
    # gloo, so that the multi-rank code path (sharding, all-reduce, max-over-ranks timing, teardown) can be run
    # where RCCL would refuse two ranks on one device.  Never set by the driver.
    single_dev = os.environ.get("AGPL_BENCH_SINGLE_DEVICE") == "1"
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    group = None
    watchdog = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # A rank that never arrives (or a first collective that hangs) must end the job with a non-zero exit instead of
        # stalling the node: the rendezvous and every collective carry a timeout, and a watchdog thread ends THIS
        # process (os._exit: no re-exec, nothing started from a process that has touched the GPU) if the first
        # all-reduce has not completed in time.
        tmo = datetime.timedelta(seconds=int(os.environ.get("AGPL_BENCH_TIMEOUT_S", RENDEZVOUS_TIMEOUT_S)))

        def _expired():
            print(f"[bench rank {rank}] rendezvous / first all-reduce did not complete in {tmo}: exiting 3",
                  file=sys.stderr, flush=True)
            os._exit(3)

        watchdog = threading.Timer(tmo.total_seconds(), _expired)
        watchdog.daemon = True
        watchdog.start()
        if single_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=tmo,
                                    device_id=torch.device("cuda", local_rank))
        group = dist.group.WORLD
        probe = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(probe, group=group)  # the first collective builds the communicator: under the watchdog
        torch.cuda.synchronize()
        if probe.item() != world:
            print(f"[bench rank {rank}] first all-reduce returned {probe.item()} != {world}", file=sys.stderr, flush=True)
            os._exit(4)
        rccl_ranks = int(probe.item())  # what the communicator actually summed over (reported in the JSON line)
        watchdog.cancel()

    import agpl_amd as A
    from agpl_amd import _ffi

    ctx = A.Context(local_rank, seed=SEED)
    lik = make_lik(A, args.lik)
    N, M, L = args.n, args.m, A.nlatent(lik)
    i0, i1 = A.shard_range(N, rank, world)
    n_loc = i1 - i0

    t_setup = time.time()
    y, Phi, kd = build_workload(A, ctx, lik, i0, n_loc, M)
    t_setup = time.time() - t_setup
    Mp = Phi.shape[1]

    if args.marginal == "auto":
        args.marginal = "f16x2-factor" if (Mp % 256 == 0 and args.accumulate == "f16x2") else "f32"
    if args.marginal == "f32":
        args.accumulate = "f32"
    cavi = A.SparseCAVI(lik, Phi, kd, y, ctx=ctx, group=group, marginal_precision=args.marginal,
                        accumulate_precision=args.accumulate)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.reset_peak_memory_stats()
    dt, kt = timed_sweeps(ctx, cavi, args.steps, args.warmup, barrier)
    hbm_sweep_gb = round(torch.cuda.max_memory_allocated() / 1e9, 1)
    plan_gb = round(cavi.plan.nbytes / 1e9, 2) if getattr(cavi, "plan", None) is not None else None
    feat_gb = round(Phi.numel() * 4 / 1e9, 2)
    if world > 1:
        import torch.distributed as dist

        # per-rank record for the diagnosis of a scaling run: this rank's own time per sweep, the exchange step as the
        # stream saw it (hipEvents around exchange(): the wait for the slowest rank + the RCCL all-reduce), the two
        # contraction kernels from the in-library events
        mine = {"rank": rank, "points": n_loc, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "allreduce_ms_avg": round(cavi.exchange_ms[0], 4), "allreduce_ms_max": round(cavi.exchange_ms[1], 4),
                "marginal_kernel_ms": round(kt[0][0] / max(kt[0][1], 1), 4),
                "accumulate_kernel_ms": round(kt[1][0] / max(kt[1][1], 1), 4)}
        per_rank, dt = gather_over_ranks(mine, dt, group, "cuda")
        dist.barrier()
        # BASELINE configs[2] itself (C3: NegBin r = 15, N sharded, M = 1024, CAVI + Gibbs) and the north-star target (Bernoulli,
        # M = 1024) on the same ranks, so that a scaling run speaks to the configuration north_star shards -- every rank takes part,
        # rank 0 keeps the objects.  Budget at 8 ranks (N = 1e7): two workload builds of 1.25e6 x 1024 (~1 s each), 2 x 12 CAVI
        # sweeps of ~6 ms, 6 Gibbs sweeps, two 5000-point parity slices on rank 0 (~10-20 s each): well under 120 s.
        sharded = {}
        want_sharded = args.sharded_legs == "on" or (args.sharded_legs == "auto" and default_config and not args.no_extra)
        if want_sharded:
            del cavi, y, Phi, kd
            gc.collect()
            torch.cuda.empty_cache()
            Ms = args.sharded_m
            pp = 0 if args.no_parity else min(5_000, max(1, N // world))
            c3o, wl = sharded_leg(A, ctx, "negbin", N, Ms, rank, world, group, barrier, steps=10, warmup=2, gibbs_steps=5,
                                  parity_points=pp,
                                  label=f"C3: NegBin(r=15) SVGP CAVI sweep, N={N}, M={Ms}, L=1, N sharded over {world} ranks, "
                                        f"1 all-reduce of L*(M^2+M) f64 per sweep (+ 5 sparse Gibbs sweeps on the same plan)")
            m1o, _ = sharded_leg(A, ctx, "bernoulli", N, Ms, rank, world, group, barrier, steps=10, warmup=2, gibbs_steps=0,
                                 parity_points=pp, workload=wl,
                                 label=f"north-star target: bernoulli-logistic SVGP CAVI sweep, N={N}, M={Ms}, L=1, N sharded over "
                                       f"{world} ranks, 1 all-reduce of L*(M^2+M) f64 per sweep")
            del wl
            sharded = {"c3": c3o, "m1024": m1o}
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
        # the single-GPU extra legs (Gibbs, parity slice, CPU baseline, C5) are reported at N = 1 only
        args.no_gibbs = args.no_cpu = args.no_parity = args.no_m1024 = args.no_c5 = args.no_f32 = args.no_elbo = args.no_extra = True

    ms_per_step = dt / args.steps * 1e3
    value = args.steps / dt
    roofline = roofline_of(kt, L, n_loc, M, Mp, args.marginal, args.accumulate, ms_per_step, world, N,
                           traffic_key={"lik": args.lik, "N": N, "M": M, "L": L})
    if world == 1:
        sustained_f16(ctx, roofline)

    out = {
        "metric": "CAVI sweeps/sec (N obs, M inducing) + max |Δnat-param| vs CPU ref",
        "value": round(value, 4), "unit": "sweeps/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if (args.marginal == "f32" and args.accumulate == "f32")
                 else "f32 via f16 hi/lo split operands (3 f16 MFMA products per f32 product, f32 accumulate, "
                      "f64 reductions and M x M update)",
        "data": "synthetic",
        "config": {"workload": f"{args.lik}-logistic SVGP CAVI sweep, N={N}, M={M} (padded {Mp}), L={L}, "
                               f"N sharded over {world} GPU(s), 1 all-reduce of L*(M^2+M) f64 per sweep",
                   "N": N, "M": M, "L": L, "parallelism": f"N-shard x{world}", "marginal_pass": args.marginal, "accumulate_pass": args.accumulate},
        "roofline": roofline, "setup_s": round(t_setup, 2),
        "hbm_gb": {"timed_sweeps_high_water": hbm_sweep_gb,
                   "plan": plan_gb,
                   "float32_features": feat_gb,
                   "note": "the plan (both split-float16 images + q(v)) is all that CAVI and Gibbs sweeps read; the float32 features "
                           "stay resident here only because the parity, float32-contract and CPU-baseline legs of this run use them"},
    }
    if world > 1:
        steps_ms = [r["ms_per_step"] for r in per_rank]
        out["allreduce_ms"] = round(sum(r["allreduce_ms_avg"] for r in per_rank) / world, 4)
        out["allreduce_bytes"] = 8 * L * (Mp * Mp + Mp)
        out["ms_per_step_min_rank"], out["ms_per_step_max_rank"] = min(steps_ms), max(steps_ms)
        out["per_rank"] = per_rank
        out["per_rank_kernels"] = [{"rank": r["rank"], "marginal_kernel_ms": r["marginal_kernel_ms"],
                                    "accumulate_kernel_ms": r["accumulate_kernel_ms"]} for r in per_rank]
        for k_, v_ in sharded.items():
            if v_ is not None:
                out[k_] = v_
        out["scaling_curve_note"] = ("this line is ONE point of a scaling curve; efficiency is the driver's to compute from its own "
                                     "N = 1, 2, 4, 8 runs")

    # ---- Gibbs half on the same resident workload (extra legs, not the headline value) ------------------------
    if not args.no_gibbs:
        yg = y.to(torch.float64) if lik.ykind == "real" else y
        gib = A.SparseGibbs(lik, Phi, kd, yg, ctx=ctx, group=None, accumulate_precision=args.accumulate,
                            plan=getattr(cavi, "plan", None))  # (the CAVI leg's plan: its accumulate image and residual are shared)
        for _ in range(2):
            gib.sweep()
        torch.cuda.synchronize()
        _ffi.lib().agpl_timing(ctx.bind(), C.c_int32(-1), None, None)
        nsw = max(3, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(nsw):
            gib.sweep()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / nsw
        pms, pcnt = read_timing(ctx, 2)
        read_timing(ctx, 1)
        proj_ms = pms / max(pcnt, 1)
        out["gibbs"] = {
            "sweeps_per_s": round(1.0 / tg, 3), "ms_per_sweep": round(tg * 1e3, 3),
            "point_pass": {"kernel": ("gibbs_project_image_kernel + " if gib.plan is not None else "gibbs_project_kernel + ") + ("aux_sample_pg1_kernel<gibbs>" if args.lik == "bernoulli"
                                                                  else "gibbs_sample_kernel"), "avg_ms": round(proj_ms, 3), "bound": "hbm",
                           "algorithmic_bytes": n_loc * (Mp * 4 + 16),
                           "achieved_GBps": round(n_loc * (Mp * 4 + 16) / (proj_ms * 1e-3) / 1e9, 1), "peak_GBps": 8000,
                           "projection_reads": "the plan's accumulate image (hi | lo float16 = 4 bytes per feature and point)"
                                               if gib.plan is not None else "the float32 features"}}
        del gib

        # standalone aux_sample! (src/generic.jl:5-12) at the marginal means: the PG sampler kernel alone, for the
        # bench likelihood and for NegBin r = 15, whose b = y + r PG(1, c) draws per point are the
        # load-balance case; PG(1) draws/s = sum_i floor(b_i) / kernel time
        def sampler_leg(slik, sname, ys, nd):
            f64 = (cavi.marginals()[0].to(torch.float64).t().contiguous() if A.nlatent(slik) > 1
                   else cavi.marginals()[0][0].to(torch.float64))[:nd].contiguous()
            Om = A.aux_sample(slik, ys, f64, ctx=ctx)
            for _ in range(3):
                A.aux_sample_(Om, slik, ys, f64, ctx=ctx)
            sms, scnt = read_timing(ctx, 3)
            sms /= max(scnt, 1)
            Ls = A.nlatent(slik)
            bytes_pt = {"bernoulli": 16, "negbin": 20, "studentt": 24, "categorical": Ls * (8 + 1 + 8 + 8)}[sname]
            if sname == "bernoulli":
                pg1 = nd
            elif sname == "negbin":
                pg1 = int(ys.to(torch.int64).sum().item()) + 15 * nd
            else:
                pg1 = None
            leg = {"kernel": "aux_sample_pg1_kernel" if sname == "bernoulli" else "aux_sample_kernel", "likelihood": sname, "points": nd, "avg_ms": round(sms, 4),
                   "bound": "hbm (by contract)", "algorithmic_bytes_per_point": bytes_pt,
                   "points_per_s": round(nd / (sms * 1e-3), 0),
                   "achieved_GBps": round(nd * bytes_pt / (sms * 1e-3) / 1e9, 1), "peak_GBps": 8000,
                   "frac": round(nd * bytes_pt / (sms * 1e-3) / 8e12, 4)}
            if pg1 is not None:
                leg["pg1_draws_per_s"] = round(pg1 / (sms * 1e-3), 0)
            return leg

        yg = y.to(torch.float64) if lik.ykind == "real" else y
        read_timing(ctx, 3)
        out["gibbs"]["sampler"] = sampler_leg(lik, args.lik, yg, n_loc)
        out["gibbs"]["sampler"]["draws_per_s"] = out["gibbs"]["sampler"]["points_per_s"]
        if args.lik == "bernoulli":
            nlik = make_lik(A, "negbin")
            nd = min(n_loc, 4_000_000)
            _, yn = A.synth_xy(nlik, SEED, i0, nd, ctx=ctx, want_x=False)
            out["gibbs"]["sampler_negbin"] = sampler_leg(nlik, "negbin", yn, nd)
            a, b = out["gibbs"]["sampler"].get("pg1_draws_per_s"), out["gibbs"]["sampler_negbin"].get("pg1_draws_per_s")
            out["gibbs"]["negbin_over_bernoulli_pg1_rate"] = round(b / a, 3) if a and b else None
            del yn
        _ffi.lib().agpl_timing(ctx.bind(), C.c_int32(-2), None, None)

    # ---- parity leg: a slice of the same workload, GPU vs oracle, 10 sweeps ---------------------
    if not args.no_parity:
        out["parity"] = parity_slice(A, ctx, lik, args.lik, Phi, kd, y, args.marginal, args.accumulate)

    # ---- full-size self-check (N = 1 only): the accumulation of the timed object at the measured size and load,
    #      against float64 torch reductions over the same gamma, beta (a slice cannot see load-dependent faults)
    if world == 1 and not args.no_parity:
        cavi.gamma = torch.empty((L, n_loc), dtype=torch.float32, device="cuda")
        cavi.beta = torch.empty((L, n_loc), dtype=torch.float32, device="cuda")
        cavi.c = torch.empty((n_loc,) if L == 1 else (n_loc, L), dtype=torch.float32, device="cuda")
        cavi.accumulate()
        torch.cuda.synchronize()
        chk = full_size_quadratic_check(Phi, cavi.gamma, cavi.beta, cavi.G, cavi.g)
        out["full_size_check"] = {"points": n_loc, **chk,
                                  "reference": "float64 torch reductions over the exported gamma, beta: g, tr G, and the "
                                               "quadratic forms v'Gv = sum_n gamma_n (phi_n . v)^2 for 4 random v (every "
                                               "tile of G, off-diagonal ones included, enters each of them)",
                                  "tolerance": 2e-6, "pass": bool(chk["max_rel_dg"] < 2e-6 and chk["rel_d_trace_G"] < 2e-6
                                                                  and chk["max_rel_d_vGv"] < 2e-6 and chk["G_symmetric"])}
        cavi.gamma = cavi.beta = cavi.c = None
        if getattr(cavi, "plan", None) is not None and cavi.nsweeps > 0:
            # ... and the marginal kernel at the measured size with the REAL q(v) of the timed sweeps (not U = I, v = 0)
            mchk = full_size_marginal_check(cavi, Phi)
            out["full_size_check"]["marginals"] = {
                **mchk, "reference": "float64 T = U Phi from the float32 feature rows of the sampled points, U, v from "
                                     "agpl_plan_state after the timed sweeps", "tolerance": 2e-5,
                "pass": bool(mchk["max_rel_d_mu"] < 2e-5 and mchk["max_rel_d_var"] < 2e-5 and mchk["max_abs_offdiag_U"] > 1e-3)}
            out["full_size_check"]["pass"] = bool(out["full_size_check"]["pass"] and out["full_size_check"]["marginals"]["pass"])

    # ---- CPU baseline: the oracle on a bounded sample of the same workload (rank 0, N = 1 only) -------------
    if world == 1 and not args.no_cpu:
        from oracle import oracle as O

        olik = make_olik(O, args.lik)
        ns = min(args.cpu_sample, n_loc)
        Ph, kh, yh = Phi[:ns].cpu().numpy(), kd[:ns].cpu().numpy().astype(np.float64), y[:ns].cpu().numpy()
        if lik.ykind == "real":
            yh = yh.astype(np.float64)
        S, m = np.tile(np.eye(Mp), (L, 1, 1)), np.zeros((L, Mp))
        t0 = time.perf_counter()
        G, g = O.cavi_pass(olik, Ph, kh, yh, -S, m)
        S, m = O.gaussian_update(G, g)
        t_cpu = time.perf_counter() - t0
        cpu_value = 1.0 / (t_cpu * (n_loc / ns))
        note = (f"1 sweep of the float64 oracle (OpenMP, scalar per-point loops) on the first {ns} of the {N} points, "
                f"{t_cpu:.2f} s; value extrapolated linearly in N (labelled extrapolation)")
        extra = {}
        if L == 1:
            # the same sweep with its two contractions through the host BLAS (numpy / OpenBLAS, float64) -- what the
            # reference's dense algebra runs on -- and the per-point operators through the oracle's vector entry
            # points; checked against the oracle's pass on the sample, and the faster of the two is `value`
            S0, m0 = np.tile(np.eye(Mp), (1, 1, 1)), np.zeros((1, Mp))
            t0 = time.perf_counter()
            P = Ph.astype(np.float64)
            q = np.einsum("ij,ij->i", P @ (-S0[0]), P)
            mu_b, var_b = P @ m0[0], kh - q
            q1, q2, _ = O.aux_posterior(olik, yh, mu_b, var_b)
            bt, gm = O.expected_potential_precision(olik, yh, q1, q2)
            Gb = (P * gm[0][:, None]).T @ P
            gb = P.T @ bt[0]
            Sb, mb = O.gaussian_update(Gb[None], gb[None])
            t_blas = time.perf_counter() - t0
            ok = bool(np.abs(Gb - G[0]).max() <= 1e-9 * np.abs(G).max() and np.abs(gb - g[0]).max() <= 1e-9 * np.abs(g).max())
            blas_value = 1.0 / (t_blas * (n_loc / ns))
            extra = {"oracle_openmp_value": cpu_value, "blas_twin_value": blas_value, "blas_twin_matches_oracle": ok}
            if ok and blas_value > cpu_value:
                cpu_value = blas_value
                note = (f"1 sweep on the first {ns} of the {N} points with the two contractions through numpy/OpenBLAS "
                        f"(float64) and the oracle's per-point operators, {t_blas:.2f} s (the oracle's own scalar pass: "
                        f"{t_cpu:.2f} s; both agree to 1e-9); value extrapolated linearly in N (labelled extrapolation)")
            del P
        full = None
        want_full = args.cpu_full if args.cpu_full is not None else (default_config and not args.no_extra)
        if want_full and L == 1 and extra.get("blas_twin_matches_oracle"):
            # ONE sweep of the BLAS twin over ALL n_loc points, in chunks of 1e6 points (the float32 features live on the
            # device: each chunk is copied to the host outside the timed regions; only the CPU work is timed) -- a measured
            # figure, the 1e6-point sample above stays as its cross-check
            Gf, gf = np.zeros((Mp, Mp)), np.zeros(Mp)
            t_full = 0.0
            chunk = 1_000_000
            for c0 in range(0, n_loc, chunk):
                c1 = min(n_loc, c0 + chunk)
                Pc, kc, yc = Phi[c0:c1].cpu().numpy(), kd[c0:c1].cpu().numpy().astype(np.float64), y[c0:c1].cpu().numpy()
                t0 = time.perf_counter()
                P = Pc.astype(np.float64)
                q = np.einsum("ij,ij->i", P @ (-S0[0]), P)
                q1, q2, _ = O.aux_posterior(olik, yc, P @ m0[0], kc - q)
                bt, gm = O.expected_potential_precision(olik, yc, q1, q2)
                Gf += (P * gm[0][:, None]).T @ P
                gf += P.T @ bt[0]
                t_full += time.perf_counter() - t0
                del P, Pc
            t0 = time.perf_counter()
            O.gaussian_update(Gf[None], gf[None])
            t_full += time.perf_counter() - t0
            full = {"value": 1.0 / t_full, "seconds": round(t_full, 2), "points": n_loc, "chunks": -(-n_loc // chunk),
                    "extrapolated_from_sample": blas_value, "measured_over_extrapolated": round((1.0 / t_full) / blas_value, 3)}
            cpu_value = full["value"]
            note = (f"MEASURED: one full sweep over all {n_loc} points with the two contractions through numpy/OpenBLAS (float64) "
                    f"and the oracle's per-point operators, {t_full:.1f} s of CPU time in {full['chunks']} chunks (host copies of the "
                    f"device-resident features not timed); the {ns}-point sample extrapolates to {1.0 / blas_value:.1f} s")
            extra["full_sweep"] = full
        try:
            with open("/proc/cpuinfo") as fh:
                phys = {ln.split(":")[1].strip() for ln in fh if ln.startswith("physical id")}
            sockets = len(phys) or None
        except Exception:
            sockets = None
        out["cpu_baseline"] = {
            "value": cpu_value, "unit": "sweeps/s", "cores": O.num_threads(), "sockets": sockets, "kind": "port",
            "sample": note, "julia": "unavailable (no julia on PATH; bench/julia_ref.jl runs the literal reference "
                                     "operators where it is)",
            "gpu_over_cpu": round(value / cpu_value, 1), **extra}
        del Ph, kh, yh

    # ---- ELBO riding the sweep, and the sweep at the contract's own arithmetic (N = 1 only) -------------------
    if world == 1 and not args.no_elbo and getattr(cavi, "plan", None) is not None:
        try:
            out["elbo"] = elbo_leg(A, ctx, lik, Phi, kd, y, ms_per_step, args, plain=cavi)
        except Exception as e:
            out["elbo"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_f32:
        try:
            out["f32_contract"] = f32_contract_leg(A, ctx, lik, args.lik, Phi, kd, y, N, M, Mp, L, args)
        except Exception as e:
            out["f32_contract"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- the other configurations the driver should see (N = 1, default C2 run only) ------------------------
    if world == 1 and default_config and not (args.no_m1024 and args.no_c5 and args.no_extra):
        del cavi, Phi, kd, y
        gc.collect()
        torch.cuda.empty_cache()
        t_legs = time.time()

        def leg(name, fn):  # an extra leg must not take the headline line down with it
            try:
                out[name] = fn()
            except Exception as e:
                out[name] = {"error": f"{type(e).__name__}: {e}"}
            gc.collect()
            torch.cuda.empty_cache()

        if not args.no_extra:
            # one rank's share of C2 at 8 GPUs (N / 8 points, everything but the all-reduce): the per-rank fixed cost decides scaling
            def n8():
                o, _ = config_leg(A, ctx, "bernoulli", N // 8, 512, steps=10, warmup=8, no_parity=args.no_parity)
                o["projected_scaling_8"] = round(ms_per_step / o["ms_per_step"], 3)
                o["projected_scaling_8_note"] = "ms_per_step of the full-N headline / ms_per_step of one rank's N/8 share, before the all-reduce (2 MB of float64)"
                return o
            leg("n8", n8)

            # BASELINE configs[3] (C4): categorical K = 10, N = 1e6, M = 256 -- CAVI and Gibbs
            def c4():
                o, _ = config_leg(A, ctx, "categorical", 1_000_000, 256, steps=10, gibbs=True, no_parity=args.no_parity, parity_points=10_000)
                return o
            leg("c4", c4)
        if not args.no_m1024:
            try:
                out["m1024"] = m1024_leg(A, ctx, args)
                if "cpu_baseline" in out and out["cpu_baseline"].get("blas_twin_value"):
                    # CPU cost of a sweep scales with M^2 at fixed N (both contractions): the M = 512 sample x 4
                    out["m1024"]["gpu_over_cpu_estimate"] = round(out["m1024"]["value"] / (out["cpu_baseline"]["value"] / 4.0), 1)
            except Exception as e:  # an extra leg must not take the headline line down with it
                out["m1024"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_extra:
            # one rank's share of the north-star configuration (Bernoulli, N = 1.25e6, M = 1024) and of C3 (NegBin r = 15, same
            # features): CAVI (+ Gibbs for C3), ten-sweep parity, projected 8-GPU scaling against the full-N single-GPU legs above
            def n8_m1024():
                o, wl = config_leg(A, ctx, "bernoulli", N // 8, 1024, steps=10, warmup=8, no_parity=args.no_parity, parity_points=5_000)
                full = out.get("m1024", {}).get("ms_per_step")
                o["projected_scaling_8"] = round(full / o["ms_per_step"], 3) if full else None
                out["n8_m1024"] = o
                nlik = make_lik(A, "negbin")
                _, yn = A.synth_xy(nlik, SEED, 0, N // 8, ctx=ctx, want_x=False)
                o3, _ = config_leg(A, ctx, "negbin", N // 8, 1024, steps=10, warmup=8, gibbs=True, no_parity=args.no_parity, parity_points=5_000,
                                   workload=(yn, wl[1], wl[2]),
                                   label=f"one rank's share of C3: NegBin(r=15) SVGP CAVI sweep, N={N // 8}, M=1024, L=1")
                full3 = out.get("m1024", {}).get("c3_full_one_gpu", {}).get("ms_per_step")
                o3["projected_scaling_8"] = round(full3 / o3["ms_per_step"], 3) if full3 else None
                o3["projected_scaling_8_note"] = ("ms_per_step of C3 at its full N on one GPU (m1024.c3_full_one_gpu) / this leg, before the "
                                                  "all-reduce (8.4 MB of float64 per sweep)")
                return o3
            leg("c3r", n8_m1024)
        if not args.no_c5:
            try:
                out["c5"] = c5_leg(A, args)
            except Exception as e:
                out["c5"] = {"error": f"{type(e).__name__}: {e}"}
        out["extra_legs_s"] = round(time.time() - t_legs, 1)
    out["summary"] = summarize(out)  # LAST key: the driver's record keeps the tail of the line
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
