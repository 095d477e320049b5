#!/usr/bin/env julia
# bench/julia_ref.jl -- the reference Julia CPU path timed on the same host (BASELINE.md tier B3; SURVEY.md 8d
# "CPU reference timing", tier 3).  Runs the LITERAL reference operators (aux_posterior!,
# expected_auglik_potential_and_precision of AugmentedGPLikelihoods.jl) inside the sparse CAVI sweep of
# docs/src/index.md:154-163 in the whitened basis the GPU path uses, on the synthetic workload of bench.py
# (Bernoulli-logistic, x ~ U(-10, 10), f*(x) = 2 sin(0.7 x) + cos(0.23 x), SE kernel, lengthscale 1.5 x the inducing
# spacing, M inducing points on a grid) and prints ONE JSON line:
#
#     julia --project=/path/to/AugmentedGPLikelihoods.jl -t auto bench/julia_ref.jl [N] [M] [sweeps]
#
# It needs a Julia toolchain with the reference package instantiated; the build image has none (SURVEY.md F6), so
# this script is WRITTEN, NOT EXECUTED here: tests/test_julia_artifacts.py runs it iff `julia` is on PATH, and
# bench.py reports the Julia tier as "unavailable" otherwise.  Data are drawn with Julia's own RNG (the GPU run
# uses Philox): the timing is what this script is for, not stream equality.
using AugmentedGPLikelihoods
using Distributions
using LinearAlgebra
using Random

const N = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 1_000_000
const M = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 512
const SWEEPS = length(ARGS) >= 3 ? parse(Int, ARGS[3]) : 2

fstar(x) = 2sin(0.7x) + cos(0.23x)
logistic(x) = inv(1 + exp(-x))

function main()
    rng = MersenneTwister(20240807)
    x = -10 .+ 20 .* rand(rng, N)
    y = rand(rng, N) .< logistic.(fstar.(x))
    z = collect(range(-10, 10; length=M))
    ell = 1.5 * (z[2] - z[1])
    k(a, b) = exp(-abs2(a - b) / (2 * ell^2))
    Kz = [k(a, b) for a in z, b in z] + 1e-8I          # examples/bernoulli/script.jl:44 jitter
    Lz = cholesky(Symmetric(Kz)).L
    Kzx = [k(a, b) for a in z, b in x]                  # M x N, column-major: the layout the kernels read
    Φ = Lz \ Kzx                                        # whitened features
    d = max.(1 .- vec(sum(abs2, Φ; dims=1)), 0)         # Nyström residual k_nn - |φ_n|²

    lik = BernoulliLikelihood()
    qΩ = init_aux_posterior(lik, N)
    S = Matrix{Float64}(I, M, M)                        # script.jl:41-42: m = 0, S = I
    m = zeros(M)
    t = @elapsed for _ in 1:SWEEPS
        # marginals(post_u(x)) (script.jl:32-33) in feature form
        μ = Φ' * m
        σ² = d .+ vec(sum(Φ .* (S * Φ); dims=1))
        qf = Normal.(μ, sqrt.(σ²))
        aux_posterior!(qΩ, lik, y, qf)                  # script.jl:34
        β, γ = expected_auglik_potential_and_precision(lik, qΩ, y)
        G = Φ * (only(γ) .* Φ')                         # κ Diag(r) κᵀ  (docs/src/index.md:154-163)
        g = Φ * only(β)
        S = inv(Symmetric(I + G))                       # script.jl:35
        m = S * g                                       # script.jl:36
    end
    sweeps_per_s = SWEEPS / t
    println("{\"metric\": \"CAVI sweeps/sec (reference Julia CPU path)\", \"value\": $(sweeps_per_s), ",
            "\"unit\": \"sweeps/s\", \"N\": $N, \"M\": $M, \"sweeps\": $SWEEPS, ",
            "\"julia_threads\": $(Threads.nthreads()), \"blas_threads\": $(BLAS.get_num_threads()), ",
            "\"cpu_threads\": $(Sys.CPU_THREADS), \"kind\": \"reference\", ",
            "\"max_abs_m\": $(maximum(abs, m))}")
end

main()
