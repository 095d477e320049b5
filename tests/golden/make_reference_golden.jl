# make_reference_golden.jl -- run on any box that has Julia and the reference package:
#
#     julia --project=<env with AugmentedGPLikelihoods> tests/golden/make_reference_golden.jl [outdir = tests/golden]
#
# writes tests/golden/reference_<likelihood>.json and reference_polyagamma.json: inputs and the REFERENCE's own outputs for
# every deterministic operator on the hot path (aux_posterior!, expected_auglik_*, auglik_* on a fixed Ω, logtilt,
# expected_logtilt, aux_kldivergence; mean / logpdf of PolyaGamma, mass_texpon, a(n, x)) plus 1e5-draw moments of
# rand(PolyaGamma).  tests/test_reference_golden.py consumes the files when they are present (oracle leg on CPU, device leg
# under -m gpu) and reports "skipped" otherwise.  Until such files exist the repo's parity is "partial": the oracle is pinned
# on the reference's known answers only (DESIGN.md 3).  Inputs are fixed numbers (no RNG involved) so that a re-run on
# another Julia / Distributions version reproduces the same file up to floating point.
using AugmentedGPLikelihoods
using AugmentedGPLikelihoods: nlatent, aux_posterior, expected_auglik_potential, expected_auglik_precision,
    auglik_potential, auglik_precision, logtilt, expected_logtilt, aux_kldivergence, aux_prior
using AugmentedGPLikelihoods.SpecialDistributions: PolyaGamma
const SD = AugmentedGPLikelihoods.SpecialDistributions
using Distributions
using GPLikelihoods
using ArraysOfArrays
using TupleVectors
using Random
using Statistics

outdir = length(ARGS) >= 1 ? ARGS[1] : @__DIR__

# ---- a tiny JSON writer (numbers, strings, vectors, dicts with string keys; no dependency) -------------------------
jnum(x::Integer) = string(x)
jnum(x::Bool) = x ? "1" : "0"
jnum(x::AbstractFloat) = isfinite(x) ? repr(Float64(x)) : (isnan(x) ? "\"nan\"" : (x > 0 ? "\"inf\"" : "\"-inf\""))
json(x::Real) = jnum(x)
json(x::AbstractString) = "\"" * x * "\""
json(x::Union{AbstractVector,Tuple}) = "[" * join((json(v) for v in x), ", ") * "]"
json(x::AbstractMatrix) = json([collect(r) for r in eachrow(x)])
json(d::AbstractDict) = "{" * join(("\"" * string(k) * "\": " * json(v) for (k, v) in d), ",\n ") * "}"

# ---- fixed inputs ---------------------------------------------------------------------------------------------------
const n = 48
grid(a, b, k = n) = collect(range(a, b; length = k))
fvals(shift = 0.0) = [2.3 * sin(0.37 * i + shift) + 0.8 * cos(1.1 * i) for i in 1:n]      # latent values in (-3.1, 3.1)
muvals(shift = 0.0) = [1.9 * sin(0.53 * i + shift) - 0.4 for i in 1:n]
varvals(shift = 0.0) = [0.05 + 1.7 * abs(cos(0.29 * i + shift)) for i in 1:n]

function dump_lik(name, lik, y, f, mu, var, Ω; params = Dict{String,Any}())
    L = nlatent(lik)
    qf = L == 1 ? Normal.(mu, sqrt.(var)) : [Normal.(mu[l], sqrt.(var[l])) for l in 1:L]
    ft = L == 1 ? f : invert(ArrayOfSimilarArrays(f))       # what TestUtils.test_auglik passes (src/TestUtils.jl:65-66)
    qft = L == 1 ? qf : invert(ArrayOfSimilarArrays(qf))
    d = Dict{String,Any}("name" => name, "n" => n, "nlatent" => L, "params" => params,
                         "y" => (y isa AbstractVector{<:AbstractVector} ? [collect(Int.(v)) for v in y] : collect(y)),
                         "f" => L == 1 ? f : f, "qf_mean" => mu, "qf_var" => var)
    qΩ = aux_posterior(lik, y, qft)
    φ = only(qΩ.inds)
    d["aux_posterior"] = Dict{String,Any}(string(k) => collect(getproperty(φ, k)) for k in propertynames(φ))
    d["expected_auglik_potential"] = [collect(v) for v in expected_auglik_potential(lik, qΩ, y, qft)]
    d["expected_auglik_precision"] = [collect(v) for v in expected_auglik_precision(lik, qΩ, y, qft)]
    d["omega"] = Dict{String,Any}(string(k) => collect(getproperty(Ω, k)) for k in propertynames(Ω))
    d["auglik_potential"] = [collect(v) for v in auglik_potential(lik, Ω, y, ft)]
    d["auglik_precision"] = [collect(v) for v in auglik_precision(lik, Ω, y, ft)]
    try
        d["logtilt"] = logtilt(lik, Ω, y, ft)
    catch err
        d["logtilt_error"] = sprint(showerror, err)
    end
    try
        d["expected_logtilt"] = expected_logtilt(lik, qΩ, y, qft)
    catch err
        d["expected_logtilt_error"] = sprint(showerror, err)
    end
    try
        d["aux_kldivergence"] = aux_kldivergence(lik, qΩ, aux_prior(lik, y))
    catch err
        d["aux_kldivergence_error"] = sprint(showerror, err)
    end
    open(joinpath(outdir, "reference_$(name).json"), "w") do io
        write(io, json(d), "\n")
    end
    println("wrote reference_$(name).json")
end

posω(shift = 0.0) = [0.02 + 0.31 * abs(sin(0.41 * i + shift)) for i in 1:n]   # a fixed positive auxiliary field
counts(k, shift = 0) = [mod(7 * i + shift, k) for i in 1:n]

# Bernoulli (src/likelihoods/bernoulli.jl)
dump_lik("bernoulli", BernoulliLikelihood(LogisticLink()), [isodd(div(3i, 2)) for i in 1:n], fvals(), muvals(), varvals(),
         TupleVector(; ω = posω()))
# negative binomial, integer and real failures (src/likelihoods/negativebinomial.jl; test/likelihoods/negativebinomial.jl)
for (tag, r) in (("negbin_r10", 10), ("negbin_r5p5", 5.5), ("negbin_r15", 15.0))
    dump_lik(tag, NegativeBinomialLikelihood(NBParamFailure(r), LogisticLink()), counts(23, 3), fvals(0.3), muvals(0.2),
             varvals(0.1), TupleVector(; ω = posω(0.5)); params = Dict{String,Any}("failures" => r))
end
# Student-t (src/likelihoods/studentt.jl; examples/studentt/script.jl:17-19)
for (tag, ν, σ) in (("studentt_3_1p5", 3.0, 1.5), ("studentt_3p5_2", 3.5, 2.0))
    dump_lik(tag, StudentTLikelihood(ν, σ), [1.3 * sin(0.77 * i) + 0.2 * i / n for i in 1:n], fvals(0.6), muvals(0.4),
             varvals(0.3), TupleVector(; ω = posω(0.9)); params = Dict{String,Any}("nu" => ν, "sigma" => σ))
end
# Laplace (src/likelihoods/laplace.jl)
dump_lik("laplace_1", LaplaceLikelihood(1.0), [0.9 * cos(0.61 * i) for i in 1:n], fvals(0.8), muvals(0.5), varvals(0.6),
         TupleVector(; ω = posω(1.3)); params = Dict{String,Any}("beta" => 1.0))
# Poisson with the scaled logistic link (src/likelihoods/poisson.jl)
dump_lik("poisson_10", PoissonLikelihood(ScaledLogistic(10.0)), counts(9, 1), fvals(1.0), muvals(0.7), varvals(0.8),
         TupleVector(; ω = posω(1.7), n = counts(5, 2)); params = Dict{String,Any}("lambda" => 10.0))
# heteroscedastic Gaussian (src/likelihoods/heteroscedasticgaussian.jl), two latents (f, g)
let lik = HeteroscedasticGaussianLikelihood(InvScaledLogistic(3.0))
    dump_lik("heterogauss_3", lik, [0.7 * sin(0.45 * i) for i in 1:n], [fvals(0.2), fvals(1.4)], [muvals(0.1), muvals(1.1)],
             [varvals(0.2), varvals(1.2)], TupleVector(; ω = posω(2.1), n = counts(4, 1));
             params = Dict{String,Any}("lambda" => 3.0))
end
# categorical, logistic-softmax link, non-bijective (K = L = 4) and bijective (K = 4, L = 3) (src/likelihoods/categorical.jl)
for (tag, bij) in (("categorical_4", false), ("categorical_bij_4", true))
    K = 4
    link = LogisticSoftMaxLink(zeros(K))
    lik = CategoricalLikelihood(bij ? BijectiveSimplexLink(link) : link)
    L = nlatent(lik)
    cls = [mod(3i + 1, K) + 1 for i in 1:n]
    y = nestedview(Matrix{Bool}((1:K) .== cls')[1:L, :])     # one-hot rows, as TestUtils.gen_y (src/TestUtils.jl:52-55)
    f = [fvals(0.5 * l) for l in 1:L]
    mu = [muvals(0.3 * l) for l in 1:L]
    var = [varvals(0.2 * l) for l in 1:L]
    Ω = init_aux_variables(lik, n)                            # the reference's own container for this likelihood ...
    for k in propertynames(Ω)                                 # ... filled with fixed values
        v = getproperty(Ω, k)
        fl = v isa AbstractVector{<:AbstractVector} ? flatview(v) : v
        if eltype(fl) <: Integer
            fl .= reshape([mod(5j + 2, 4) for j in 1:length(fl)], size(fl))
        else
            fl .= reshape([0.03 + 0.27 * abs(sin(0.23 * j)) for j in 1:length(fl)], size(fl))
        end
    end
    dump_lik(tag, lik, y, f, mu, var, Ω; params = Dict{String,Any}("K" => K, "bijective" => bij ? 1 : 0))
end

# ---- PolyaGamma (src/SpecialDistributions/polyagamma.jl) ---------------------------------------------------------------
pg = Dict{String,Any}()
bcs = [(1, 0.0), (1, 2.0), (3, 0.0), (3, 2.5), (3, 3.2), (1.2, 3.2), (1, 9.0), (2, 60.0), (0.4, 1.0), (15, 0.7)]
pg["mean"] = [[b, c, mean(PolyaGamma(b, c))] for (b, c) in bcs]                                  # polyagamma.jl:25-31
xs = [10.0^e for e in -2.5:0.25:0.5]
pg["logpdf"] = [[b, c, x, logpdf(PolyaGamma(b, c), x)] for (b, c) in bcs[1:8] for x in xs]      # polyagamma.jl:37-91
pg["mass_texpon"] = [[z, SD.mass_texpon(z, π^2 / 8 + z^2 / 2)] for z in (1e-8, 0.05, 0.5, 1.0, 2.5, 6.0, 20.0)]  # :179-193
pg["a"] = [[k, x, SD.a(k, x)] for k in (0, 1, 2, 5, 9) for x in (0.01, 0.2, 0.64, 0.65, 1.0, 3.0)]                  # :167-177
rng = MersenneTwister(42)                                                                         # test/utils.jl:2
pg["rand_moments"] = [begin
        s = rand(rng, PolyaGamma(b, c), 100_000)
        [b, c, mean(s), var(s), length(s)]
    end for (b, c) in bcs]
pg["kl_to_prior"] = [[b, c, Distributions.kldivergence(PolyaGamma(b, c), PolyaGamma(b, 0.0))] for (b, c) in bcs if c > 0]
open(joinpath(outdir, "reference_polyagamma.json"), "w") do io
    write(io, json(pg), "\n")
end
println("wrote reference_polyagamma.json")
