# AGPLDeviceExt.jl -- package extension routing AugmentedGPLikelihoods.jl's operator surface to libagpl.so
# (include/agpl.h) for AMDGPU.jl device arrays.  Drop it into the package's `ext/` directory with
#
#     [weakdeps]   AMDGPU = "21141c5a-9bdb-4563-92ae-f87d6854732e"
#     [extensions] AGPLDeviceExt = "AMDGPU"
#
# Nothing in the package's own methods changes: these are additional methods of the SAME generic functions
# (src/generic.jl:5-72, src/likelihoods/*.jl) dispatching on ROCArray arguments, so every existing call site
# (examples/*/script.jl, an SVGP loop over ApproximateGPs) keeps its code and gets the device path by passing
# device arrays.  One `ccall` per operator; status codes become the exception types the reference throws.
#
# STATUS: written against include/agpl.h v110 (the plan API), not executed -- the build image has no `julia` (SURVEY.md F6).  The
# executed twin is the Python host (augmentedgplikelihoods.jl_amd/operators.py, sparse.py): same entry point per
# operator, same argument order.  tests/test_julia_artifacts.py checks that every `ccall` below names an exported
# symbol with the right number of arguments.
module AGPLDeviceExt

using AugmentedGPLikelihoods
using AMDGPU
using Random
using LinearAlgebra
using Distributions: Normal, mean, var
using TupleVectors: TupleVector
using ArraysOfArrays: flatview

using AugmentedGPLikelihoods: AbstractLikelihood, BernoulliLikelihood, NegativeBinomialLikelihood, NBParamFailure,
    StudentTLikelihood, CategoricalLikelihood, PoissonLikelihood, LaplaceLikelihood,
    HeteroscedasticGaussianLikelihood, LogisticLink, LogisticSoftMaxLink, BijectiveSimplexLink, ScaledLogistic,
    InvScaledLogistic, nlatent
import AugmentedGPLikelihoods: aux_sample!, aux_posterior!, auglik_potential, auglik_precision,
    auglik_potential_and_precision, expected_auglik_potential, expected_auglik_precision,
    expected_auglik_potential_and_precision, logtilt, aug_loglik, expected_logtilt, expected_aug_loglik, aux_kldivergence

const libagpl = get(ENV, "AGPL_LIB", "libagpl.so")

# ------------------------------------------------------------------------------------------------ descriptor
# mirrors agpl_lik_desc; logtheta is a HOST pointer that must stay alive across the call (GC.@preserve below)
struct LikDesc
    kind::Int32
    nlatent::Int32
    p::NTuple{4,Float64}
    logtheta::Ptr{Float64}
end
const P0 = (0.0, 0.0, 0.0, 0.0)

# (descriptor, object to GC.@preserve) per likelihood family = per file of src/likelihoods/
desc(::BernoulliLikelihood{<:LogisticLink}) = (LikDesc(0, 1, P0, C_NULL), nothing)
function desc(l::NegativeBinomialLikelihood{<:NBParamFailure})
    return (LikDesc(1, 1, (Float64(l.params.failures), 0.0, 0.0, 0.0), C_NULL), nothing)
end
desc(l::StudentTLikelihood) = (LikDesc(2, 1, (Float64(l.ν), Float64(l.σ), 0.0, 0.0), C_NULL), nothing)
function desc(l::CategoricalLikelihood{<:LogisticSoftMaxLink})        # categorical.jl:9-10, nlatent = K
    θ = convert(Vector{Float64}, l.invlink.logθ)
    return (LikDesc(3, length(θ), P0, pointer(θ)), θ)
end
function desc(l::CategoricalLikelihood{<:BijectiveSimplexLink{<:LogisticSoftMaxLink}})   # nlatent = K - 1
    θ = convert(Vector{Float64}, l.invlink.link.logθ)
    return (LikDesc(4, length(θ) - 1, P0, pointer(θ)), θ)
end
desc(l::PoissonLikelihood{<:ScaledLogistic}) = (LikDesc(5, 1, (Float64(l.invlink.λ), 0.0, 0.0, 0.0), C_NULL), nothing)
desc(l::LaplaceLikelihood) = (LikDesc(6, 1, (Float64(l.β), 0.0, 0.0, 0.0), C_NULL), nothing)
function desc(l::HeteroscedasticGaussianLikelihood{<:InvScaledLogistic})
    return (LikDesc(7, 2, (Float64(l.invlink.λ), 0.0, 0.0, 0.0), C_NULL), nothing)
end

# ------------------------------------------------------------------------------------------------ context
# replaces GLOBAL_RNG / the `rng` argument (src/generic.jl:1-3): Philox key = seed, `sweep` = draw counter
mutable struct Ctx
    h::Ptr{Cvoid}
    sweep::UInt32
end

function check(h::Ptr{Cvoid}, rc::Integer)
    rc == 0 && return nothing
    msg = h == C_NULL ? "libagpl status $rc" :
          unsafe_string(ccall((:agpl_last_error, libagpl), Cstring, (Ptr{Cvoid},), h))
    rc == -1 && throw(ArgumentError(msg))       # negativemultinomial.jl:17-22
    rc == -2 && throw(DomainError(NaN, msg))     # polyagamma.jl:175
    rc == -5 && throw(PosDefException(0))        # the M x M Cholesky
    rc == -6 && throw(OutOfMemoryError())
    return error(msg)                            # categorical.jl:165-170, polyagamma.jl:100-104
end

function Ctx(dev::Integer=0; seed::Integer=rand(UInt64))
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:agpl_ctx_create, libagpl), Int32, (Ref{Ptr{Cvoid}}, Int32, UInt64), r, dev, seed))
    c = Ctx(r[], 0)
    finalizer(c -> ccall((:agpl_ctx_destroy, libagpl), Int32, (Ptr{Cvoid},), c.h), c)
    return c
end

const CTX = Ref{Ctx}()
ctx() = isassigned(CTX) ? CTX[] : (CTX[] = Ctx())
function next_sweep!(c::Ctx)
    s = c.sweep
    c.sweep += one(UInt32)
    return s
end
synchronize(c::Ctx=ctx()) = check(c.h, ccall((:agpl_ctx_synchronize, libagpl), Int32, (Ptr{Cvoid},), c.h))
"Global index of this rank's first point: per-point Philox streams are keyed on offset + i (N sharded over ranks)."
function set_point_offset!(c::Ctx, i0::Integer)
    return check(c.h, ccall((:agpl_ctx_set_point_offset, libagpl), Int32, (Ptr{Cvoid}, Int64), c.h, i0))
end

dptr(a::ROCArray) = Ptr{Cvoid}(UInt(pointer(a)))
dptr(::Nothing) = Ptr{Cvoid}(0)
field(tv, s::Symbol) = hasproperty(tv, s) ? getproperty(tv, s) : nothing
flat(a) = a isa ROCArray ? a : flatview(a)      # nestedview([L, N]) containers (categorical.jl:52-70) -> flat [L, N]
npoints(lik, f) = length(flat(f)) ÷ nlatent(lik)

# ------------------------------------------------------------------------------------------------ device marginals
"""
    DeviceNormals(mean, var)

What the device `marginals` returns in place of the `Vector{Normal}` the reference's methods receive
(`bernoulli.jl:17-25`): the same q(f_i) = N(mean_i, var_i) as two device arrays ([N], or [L, N] column-major for
multi-latent likelihoods).  `DeviceNormals(qf::AbstractVector{<:Normal})` uploads a host vector, so a call site
that still computes `marginals(...)` on the CPU keeps working.
"""
struct DeviceNormals{T<:ROCArray{Float64}}
    mean::T
    var::T
end
DeviceNormals(qf::AbstractVector{<:Normal}) = DeviceNormals(ROCArray(mean.(qf)), ROCArray(var.(qf)))
Base.length(q::DeviceNormals) = length(q.mean)

# ------------------------------------------------------------------------------------------------ Gibbs half
# aux_sample!(rng, Ω, lik, y, f)  src/generic.jl:5-12  ->  agpl_aux_sample.  The rng argument is accepted for
# signature compatibility; the draws come from the context's Philox streams (seed, point, sweep).
function aux_sample!(::AbstractRNG, Ω::TupleVector, lik::AbstractLikelihood, y::ROCArray, f::ROCArray{Float64})
    c = ctx()
    d, keep = desc(lik)
    ω, n = flat(Ω.ω), field(Ω, :n)
    n = n === nothing ? nothing : flat(n)
    GC.@preserve keep check(c.h, ccall((:agpl_aux_sample, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ptr{Cvoid}, Ptr{Cvoid}),
        c.h, d, npoints(lik, f), dptr(y), dptr(f), dptr(ω), dptr(n), next_sweep!(c), C_NULL, C_NULL))
    return Ω                                      # mutated in place and returned, as generic.jl:11
end

# auglik_potential / auglik_precision / _and_precision  (generic.jl:64-66; bernoulli.jl:27-33 ...)  -> agpl_potential_precision
function auglik_potential_and_precision(lik::AbstractLikelihood, Ω::TupleVector, y::ROCArray, f=nothing)
    c = ctx()
    d, keep = desc(lik)
    L = nlatent(lik)
    N = length(y) ÷ (lik isa CategoricalLikelihood ? L : 1)
    β = ROCArray{Float64}(undef, N, L)            # L contiguous vectors of N (utils.jl:24)
    γ = similar(β)
    n = field(Ω, :n)
    GC.@preserve keep check(c.h, ccall((:agpl_potential_precision, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        c.h, d, N, dptr(y), dptr(flat(Ω.ω)), dptr(n === nothing ? nothing : flat(n)),
        dptr(f === nothing ? nothing : flat(f)), dptr(β), dptr(γ)))
    return ntuple(l -> view(β, :, l), L), ntuple(l -> view(γ, :, l), L)
end
auglik_potential(lik::AbstractLikelihood, Ω::TupleVector, y::ROCArray, f=nothing) =
    first(auglik_potential_and_precision(lik, Ω, y, f))
auglik_precision(lik::AbstractLikelihood, Ω::TupleVector, y::ROCArray, f=nothing) =
    last(auglik_potential_and_precision(lik, Ω, y, f))

# ------------------------------------------------------------------------------------------------ CAVI half
# which fields of only(qΩ.inds) are out1 / out2 / out3 of agpl_aux_posterior (include/agpl.h)
out1(lik, φ) = lik isa StudentTLikelihood ? φ.β : lik isa LaplaceLikelihood ? φ.μ : φ.c
out2(φ) = hasproperty(φ, :p) ? φ.p : field(φ, :λ)
out3(φ) = field(φ, :ψ)

# aux_posterior!(qΩ, lik, y, qf)  bernoulli.jl:17-25, negativebinomial.jl:24-33, studentt.jl:50-58,
# categorical.jl:80-110, poisson.jl:30-39, laplace.jl:44-52, heteroscedasticgaussian.jl:34-46 -> agpl_aux_posterior
function aux_posterior!(qΩ, lik::AbstractLikelihood, y::ROCArray, qf::DeviceNormals)
    c = ctx()
    d, keep = desc(lik)
    φ = only(qΩ.inds)
    o2, o3 = out2(φ), out3(φ)
    GC.@preserve keep check(c.h, ccall((:agpl_aux_posterior, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        c.h, d, 1 #= AGPL_F64 =#, npoints(lik, qf.mean), dptr(y), dptr(qf.mean), dptr(qf.var),
        dptr(flat(out1(lik, φ))), dptr(o2 === nothing ? nothing : flat(o2)), dptr(o3)))
    hasproperty(φ, :y) && copyto!(flat(φ.y), y)   # negativebinomial.jl:28, poisson.jl:35, categorical.jl:84
    return qΩ
end
# the reference's own argument type: a host Vector{Normal} is uploaded and routed to the method above
aux_posterior!(qΩ, lik::AbstractLikelihood, y::ROCArray, qf::AbstractVector{<:Normal}) =
    aux_posterior!(qΩ, lik, y, DeviceNormals(qf))

# expected_auglik_potential / _precision / _and_precision  (generic.jl:68-72; bernoulli.jl:35-45 ...)
# f = DeviceNormals of the latent(s) for the heteroscedastic likelihood (mu_g), else unused
function expected_auglik_potential_and_precision(lik::AbstractLikelihood, qΩ, y::ROCArray, f=nothing)
    c = ctx()
    d, keep = desc(lik)
    φ = only(qΩ.inds)
    L = nlatent(lik)
    N = length(y) ÷ (lik isa CategoricalLikelihood ? L : 1)
    β = ROCArray{Float64}(undef, N, L)
    γ = similar(β)
    o2 = out2(φ)
    μg = f isa DeviceNormals ? f.mean : nothing
    GC.@preserve keep check(c.h, ccall((:agpl_expected_potential_precision, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        c.h, d, 1, N, dptr(y), dptr(flat(out1(lik, φ))), dptr(o2 === nothing ? nothing : flat(o2)), dptr(μg),
        dptr(β), dptr(γ)))
    return ntuple(l -> view(β, :, l), L), ntuple(l -> view(γ, :, l), L)
end
expected_auglik_potential(lik::AbstractLikelihood, qΩ, y::ROCArray, f=nothing) =
    first(expected_auglik_potential_and_precision(lik, qΩ, y, f))
expected_auglik_precision(lik::AbstractLikelihood, qΩ, y::ROCArray, f=nothing) =
    last(expected_auglik_potential_and_precision(lik, qΩ, y, f))

# ------------------------------------------------------------------------------------------------ ELBO terms
# logtilt generic.jl:40-46 -> agpl_logtilt ; aug_loglik generic.jl:48-50 -> agpl_aug_loglik
function logtilt(lik::AbstractLikelihood, Ω::TupleVector, y::ROCArray, f::ROCArray{Float64})
    c = ctx()
    d, keep = desc(lik)
    out = Ref{Float64}(0.0)
    n = field(Ω, :n)
    GC.@preserve keep check(c.h, ccall((:agpl_logtilt, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}),
        c.h, d, npoints(lik, f), dptr(y), dptr(flat(Ω.ω)), dptr(n === nothing ? nothing : flat(n)), dptr(f), out))
    return out[]
end
function aug_loglik(lik::AbstractLikelihood, Ω::TupleVector, y::ROCArray, f::ROCArray{Float64})
    c = ctx()
    d, keep = desc(lik)
    out = Ref{Float64}(0.0)
    n = field(Ω, :n)
    GC.@preserve keep check(c.h, ccall((:agpl_aug_loglik, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}),
        c.h, d, npoints(lik, f), dptr(y), dptr(flat(Ω.ω)), dptr(n === nothing ? nothing : flat(n)), dptr(f), out))
    return out[]
end
# expected_logtilt api.jl:219-223 -> agpl_expected_logtilt
function expected_logtilt(lik::AbstractLikelihood, qΩ, y::ROCArray, qf::DeviceNormals)
    c = ctx()
    d, keep = desc(lik)
    φ = only(qΩ.inds)
    o2 = out2(φ)
    out = Ref{Float64}(0.0)
    GC.@preserve keep check(c.h, ccall((:agpl_expected_logtilt, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}),
        c.h, d, npoints(lik, qf.mean), dptr(y), dptr(flat(out1(lik, φ))), dptr(o2 === nothing ? nothing : flat(o2)),
        dptr(qf.mean), dptr(qf.var), out))
    return out[]
end
# aux_kldivergence(lik, qΩ, y) generic.jl:56-62 -> agpl_aux_kldivergence
function aux_kldivergence(lik::AbstractLikelihood, qΩ, y::ROCArray)
    c = ctx()
    d, keep = desc(lik)
    φ = only(qΩ.inds)
    o2 = out2(φ)
    out = Ref{Float64}(0.0)
    N = length(y) ÷ (lik isa CategoricalLikelihood ? nlatent(lik) : 1)
    GC.@preserve keep check(c.h, ccall((:agpl_aux_kldivergence, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}),
        c.h, d, N, dptr(y), dptr(flat(out1(lik, φ))), dptr(o2 === nothing ? nothing : flat(o2)), out))
    return out[]
end

# expected_aug_loglik(lik, qΩ, y, qf) generic.jl:52-54 (heteroscedasticgaussian.jl:130-145 its own method) -> agpl_expected_aug_loglik
function expected_aug_loglik(lik::AbstractLikelihood, qΩ, y::ROCArray, qf::DeviceNormals)
    c = ctx()
    d, keep = desc(lik)
    φ = only(qΩ.inds)
    o2 = out2(φ)
    out = Ref{Float64}(0.0)
    GC.@preserve keep check(c.h, ccall((:agpl_expected_aug_loglik, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}),
        c.h, d, npoints(lik, qf.mean), dptr(y), dptr(flat(out1(lik, φ))), dptr(o2 === nothing ? nothing : flat(o2)),
        dptr(qf.mean), dptr(qf.var), out))
    return out[]
end

# ------------------------------------------------------------------------------------------------ sparse sweep
"""
    SparseSweep(lik, Φ, d, y; comm=C_NULL, track_elbo=false)

State of the sparse CAVI loop (`cavi!`, examples/bernoulli/script.jl:29-39, in the sparse whitened form of
docs/src/index.md:154-163) on the shipped device path, i.e. ONE plan (include/agpl.h, agpl_plan_create): Φ::ROCMatrix{Float32}
is M x N column-major -- exactly Julia's layout of K_ZX (whitened: Φ = L⁻¹ K_ZX), any M (the plan zero-pads to a multiple of 256
inside; G, g stay M-sized) -- `d` the
Nyström residual k_nn - |φ_n|².  The plan builds both split-float16 images of Φ (one scale) and carries q(v) as (U, v) with
S = UᵀU, m = Uᵀv; after construction Φ itself is no longer read by the sweep and may be freed by the caller.
"""
mutable struct SparseSweep{Tlik}
    lik::Tlik
    plan::Ptr{Cvoid}            # agpl_plan*
    mem::ROCVector{UInt8}       # the plan's device memory (caller-owned storage: freed with this object)
    N::Int
    M::Int
    y::ROCArray
    Gg::ROCVector{Float64}      # [G (L M M) | g (L M) | ELBO terms (1)]: ONE buffer, one all-reduce per sweep
    kl::ROCVector{Float64}      # KL(q(v) || p(v)) of the last two updates
    nsweeps::Int
    track_elbo::Bool
    comm::Ptr{Cvoid}            # ncclComm_t when N is sharded over ranks, else C_NULL
end

function SparseSweep(lik, Φ::ROCMatrix{Float32}, d::ROCVector{Float32}, y::ROCArray; comm::Ptr{Cvoid}=C_NULL,
                     track_elbo::Bool=false)
    c = ctx()
    M, N = size(Φ)
    L = nlatent(lik)
    nbytes = ccall((:agpl_plan_bytes, libagpl), Int64, (Int64, Int32, Int32, UInt32), N, M, L, 0)
    nbytes > 0 || throw(ArgumentError("need N >= 1 points, M >= 1 features and nlatent <= 64 (got N = $N, M = $M)"))
    mem = ROCVector{UInt8}(undef, nbytes)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(c.h, ccall((:agpl_plan_create, libagpl), Int32,
        (Ptr{Cvoid}, Int64, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
        c.h, N, M, L, dptr(Φ), dptr(d), 0, dptr(mem), h))          # DomainError: a feature is not finite / out of range
    s = SparseSweep(lik, h[], mem, N, M, y, AMDGPU.zeros(Float64, L * M * M + L * M + 1), AMDGPU.zeros(Float64, 2), 0,
                    track_elbo, comm)
    finalizer(x -> ccall((:agpl_plan_destroy, libagpl), Int32, (Ptr{Cvoid},), x.plan), s)
    return s                                                       # q(v) = N(0, I) (script.jl:41-42)
end
Gview(s::SparseSweep) = (M = s.M; L = nlatent(s.lik); reshape(view(s.Gg, 1:(L * M * M)), M, M, L))
gview(s::SparseSweep) = (M = s.M; L = nlatent(s.lik); reshape(view(s.Gg, (L * M * M + 1):(L * M * M + L * M)), M, L))
Gptr(s::SparseSweep) = Ptr{Cvoid}(UInt(pointer(s.Gg)))
gptr(s::SparseSweep) = Ptr{Cvoid}(UInt(pointer(s.Gg)) + 8 * nlatent(s.lik) * s.M * s.M)
eptr(s::SparseSweep) = s.track_elbo ? Ptr{Cvoid}(UInt(pointer(s.Gg)) + 8 * (length(s.Gg) - 1)) : C_NULL

# S = (I + G)⁻¹, m = S g (script.jl:35-36) in factor form; enqueue only (a PosDefException surfaces in the next pass)
function update!(s::SparseSweep)
    c = ctx()
    kl = s.track_elbo ? Ptr{Cvoid}(UInt(pointer(s.kl)) + 8 * (s.nsweeps & 1)) : C_NULL
    return check(c.h, ccall((:agpl_plan_update, libagpl), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), s.plan, Gptr(s), gptr(s), C_NULL, kl))
end

# one sweep: marginals -> aux_posterior! -> expected potential / precision -> (G, g) -> [all-reduce] -> update
function sweep!(s::SparseSweep)
    c = ctx()
    L = nlatent(s.lik)
    dsc, keep = desc(s.lik)
    GC.@preserve keep check(c.h, ccall((:agpl_cavi_pass_plan, libagpl), Int32,
        (Ptr{Cvoid}, Ref{LikDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
         Ptr{Cvoid}),
        s.plan, dsc, C_NULL, dptr(s.y), Gptr(s), gptr(s), C_NULL, C_NULL, C_NULL, eptr(s)))
    if s.comm != C_NULL                           # the one exchange step of the N-sharded sweep (SURVEY.md 8e)
        check(c.h, ccall((:agpl_allreduce_nat, libagpl), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                         c.h, s.comm, Gptr(s), length(s.Gg)))
    end
    update!(s)
    s.nsweeps += 1
    return s
end

function cavi!(s::SparseSweep; niter::Integer=10)
    for _ in 1:niter
        sweep!(s)
    end
    synchronize()                                 # the last update's outcome (PosDefException, ...)
    return moments(s)
end

"q(v) = N(m, S): m = Uᵀ v, S = Uᵀ U per latent, on the host (M x M)."
function moments(s::SparseSweep)
    synchronize()
    L, M = nlatent(s.lik), s.M
    U_p, v_p = Ref{Ptr{Float64}}(C_NULL), Ref{Ptr{Float64}}(C_NULL)
    check(ctx().h, ccall((:agpl_plan_state, libagpl), Int32,
                         (Ptr{Cvoid}, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                         s.plan, U_p, v_p, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL))
    Mp = 256 * cld(M, 256)      # the plan's state arrays are sized by M rounded up to a multiple of 256 (include/agpl.h)
    A = Array(unsafe_wrap(ROCArray, U_p[], (Mp, Mp, L)))[1:M, 1:M, :]   # the caller's U: the leading M x M block
    v = Array(unsafe_wrap(ROCArray, v_p[], (Mp, L)))[1:M, :]
    U = [LowerTriangular(A[:, :, l]) for l in 1:L]
    return [U[l]' * v[:, l] for l in 1:L], [U[l]' * U[l] for l in 1:L]
end

"`marginals(post_u(x))` on the device (examples/bernoulli/script.jl:32-33): q(f_n) for the current (U, v)."
function device_marginals(s::SparseSweep)
    c = ctx()
    L = nlatent(s.lik)
    μ, σ² = ROCArray{Float32}(undef, s.N, L), ROCArray{Float32}(undef, s.N, L)
    check(c.h, ccall((:agpl_marginals_plan, libagpl), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        s.plan, C_NULL, dptr(μ), dptr(σ²)))
    # [N, L] latent-major out; the per-point operators take [L, N] (L contiguous per point)
    qm, qv = Float64.(permutedims(μ)), Float64.(permutedims(σ²))
    return L == 1 ? DeviceNormals(vec(qm), vec(qv)) : DeviceNormals(qm, qv)
end

"""
`aug_elbo` (examples/bernoulli/script.jl:65-70) of the q(v) that ENTERED the last sweep, at no extra pass over Φ
(`track_elbo=true`): the per-point terms rode that sweep's per-point kernel (and its all-reduce), KL(q(v) ‖ p(v)) the update
that made that q(v).
"""
function elbo_entering(s::SparseSweep)
    (s.track_elbo && s.nsweeps > 0) || throw(ArgumentError("elbo_entering needs track_elbo=true and one sweep"))
    synchronize()
    return Array(view(s.Gg, length(s.Gg):length(s.Gg)))[1] - Array(s.kl)[(s.nsweeps & 1) + 1]
end

"`aug_elbo` of examples/bernoulli/script.jl:65-70 for the current q(v), by separate passes (the reference's own operators)."
function aug_elbo(s::SparseSweep)
    c = ctx()
    M = s.M
    L = nlatent(s.lik)
    qf = device_marginals(s)
    qΩ = AugmentedGPLikelihoods.init_aux_posterior(s.lik, s.N)
    qΩdev = to_device(qΩ)
    aux_posterior!(qΩdev, s.lik, s.y, qf)
    # KL(q(v) ‖ N(0, I)) = (tr S + mᵀm − M + log det(I + G)) / 2 from the plan's factor: S = UᵀU, m = Uᵀv, log det(S) = 2 Σ log U_aa
    ms, Ss = moments(s)
    kl = sum(0.5 * (tr(Ss[l]) + dot(ms[l], ms[l]) - M - logdet(Symmetric(Ss[l]))) for l in 1:L)
    return expected_logtilt(s.lik, qΩdev, s.y, qf) - aux_kldivergence(s.lik, qΩdev, s.y) - kl
end

"A `For(TupleVector(...))` of host vectors -> the same container over device arrays (fields keep their names)."
function to_device(qΩ)
    φ = only(qΩ.inds)
    names = propertynames(φ)
    return (; inds=(TupleVectorLike(NamedTuple{names}(map(n -> ROCArray(flat(getproperty(φ, n))), names))),))
end
struct TupleVectorLike{NT<:NamedTuple}
    fields::NT
end
Base.getproperty(t::TupleVectorLike, s::Symbol) = s === :fields ? getfield(t, :fields) : getfield(t, :fields)[s]
Base.hasproperty(t::TupleVectorLike, s::Symbol) = haskey(getfield(t, :fields), s)
Base.propertynames(t::TupleVectorLike) = keys(getfield(t, :fields))

end # module
