# MCMCDiagnosticToolsROC.jl -- thin Julia shim over libmdt_hip.so (include/mdt_hip.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: Julia is not available in the build image.  It shows the
# reference-side binding a maintainer would add: same function names, keyword arguments, array
# convention and error types as MCMCDiagnosticTools.jl (src/MCMCDiagnosticTools.jl:17-25), one
# `ccall` per request.  The tested host mirror is the Python package next to it.
#
# Front door, as in the reference:
#   * `Union{Missing,T}` eltypes: a parameter with any missing draw yields `missing`, the others are computed
#     (src/ess_rhat.jl:382-385, 519-523; src/utils.jl:176-179); an all-`Missing` eltype skips the device (:356, 480);
#   * any number of parameter axes and any axes offsets: outputs are `similar(x, T, axes(x)[3:end])`
#     (src/ess_rhat.jl:464-467, src/utils.jl:197-215; test/ess_rhat.jl:134-165 with OffsetArrays);
#   * `Int` input is promoted to Float64 (src/ess_rhat.jl:353, 465);
#   * device arrays (AMDGPU.jl `ROCArray`) are passed by pointer with `location = MDT_DEVICE`: no host copy;
#   * `devices = [0, 1, ...]`: the multi-GPU context (mdt_ctx_create_multi), params axis cut into contiguous blocks.
module MCMCDiagnosticToolsROC

using Statistics: Statistics

export ess, rhat, ess_rhat, rhat_nested, mcse, bfmi, gelmandiag, status, UnsupportedOnDevice
export AbstractAutocovMethod, AutocovMethod, FFTAutocovMethod, BDAAutocovMethod

const libmdt = get(ENV, "MDT_HIP_LIB", "libmdt_hip.so")

abstract type AbstractAutocovMethod end
struct AutocovMethod <: AbstractAutocovMethod end      # src/ess_rhat.jl:38
struct FFTAutocovMethod <: AbstractAutocovMethod end   # src/ess_rhat.jl:55
struct BDAAutocovMethod <: AbstractAutocovMethod end   # src/ess_rhat.jl:73
_code(::AutocovMethod) = Cint(0)
_code(::FFTAutocovMethod) = Cint(1)
_code(::BDAAutocovMethod) = Cint(2)
_code(m::AbstractAutocovMethod) =
    throw(ArgumentError("$(typeof(m)) is a CPU plugin; use MCMCDiagnosticTools.jl for it"))

# mirrors `struct mdt_args` field for field
struct MdtArgs
    draws::Int64
    chains::Int64
    params::Int64
    dtype::Int32      # 0 Float64, 1 Float32
    location::Int32   # 0 host, 1 device
    kind::Int32       # 0 basic 1 bulk 2 tail 3 rank
    outputs::Int32    # 1 ESS | 2 RHAT
    autocov::Int32
    split_chains::Int32
    relative::Int32
    sync::Int32
    maxlag::Int64
    tail_prob::Float64
    stream::Ptr{Cvoid}
    chain_stride::Int64   # 0: dense
    param_stride::Int64   # 0: dense
end
MdtArgs(D, C, P, dtype, location, kind, outputs, autocov, split, relative, sync, maxlag, tail_prob, stream) =
    MdtArgs(D, C, P, dtype, location, kind, outputs, autocov, split, relative, sync, maxlag, tail_prob, stream, 0, 0)

const _ctx = Ref{Ptr{Cvoid}}(C_NULL)
const _multi = Dict{Vector{Cint},Ptr{Cvoid}}()
function ctx()
    if _ctx[] == C_NULL
        rc = ccall((:mdt_ctx_create, libmdt), Cint, (Ref{Ptr{Cvoid}}, Cint), _ctx, 0)
        rc == 0 || error("mdt_ctx_create failed with status $rc (no MI355X visible?)")
    end
    return _ctx[]
end
# one context over several GPUs of the node: contiguous blocks of the params axis, one per device (mdt_ctx_create_multi)
function ctx(devices::AbstractVector{<:Integer})
    ids = collect(Cint, devices)
    return get!(_multi, ids) do
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:mdt_ctx_create_multi, libmdt), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cint}, Cint), h, ids, length(ids))
        rc == 0 || error("mdt_ctx_create_multi failed with status $rc")
        h[]
    end
end
_ctx_for(devices) = devices === nothing ? ctx() : ctx(devices)

"""    status(; devices=nothing, clear=true) -> flags

Sticky condition flags of the context (mdt_ctx_status): 1 ranking tables overflowed, 2 a quantile of data holding
NaN (the reference throws ArgumentError there), 4 FDist.  Only requests issued with `sync = 0` leave flags behind."""
function status(; devices=nothing, clear::Bool=true)
    f = Ref{Int32}(0)
    _check(ccall((:mdt_ctx_status, libmdt), Cint, (Ptr{Cvoid}, Ref{Int32}, Int32), _ctx_for(devices), f, clear))
    return f[]
end

"""    UnsupportedOnDevice(msg)

A valid request outside the engine's limits (MDT_EUNSUPPORTED; DESIGN.md §7: effective `maxlag > 1000`, slices beyond
the ranking's round tables).  Never an approximation: a caller that wants every request of the reference served can
catch it and hand the request to MCMCDiagnosticTools.jl on the host (INTEGRATION.md)."""
struct UnsupportedOnDevice <: Exception
    msg::String
end
Base.showerror(io::IO, e::UnsupportedOnDevice) = print(io, "unsupported on the device path: ", e.msg)

function _check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:mdt_last_error, libmdt), Cstring, (Ptr{Cvoid},), ctx()))
    rc == 1 && throw(ArgumentError(msg))
    rc == 2 && throw(DomainError(0, msg))
    rc == 3 && throw(DimensionMismatch(msg))
    rc == 6 && throw(UnsupportedOnDevice(msg))
    error("libmdt_hip status $rc: $msg")
end

const _KIND = Dict(:basic => 0, :bulk => 1, :tail => 2, :rank => 3)
# estimators with an expectand proxy (src/ess_rhat.jl:626-659): MDT_KIND_MEDIAN / STD / MAD / QUANTILE
const _EST = Dict(:median => 4, :std => 5, :mad => 6, :quantile => 7)

# ---- the reference's front door (src/utils.jl:197-215): sample / parameter axes, Missing, device arrays
_is_device(x) = nameof(typeof(x)) === :ROCArray                 # AMDGPU.jl, without depending on it
_param_axes(x::AbstractArray) = ntuple(i -> axes(x, i + 2), max(ndims(x) - 2, 0))
_valtype(x) = (T = float(nonmissingtype(eltype(x))); T in (Float32, Float64) ? T : Float64)

struct Prepared{T,A}
    x3::A                       # (draws, chains, params), dense, eltype T: Array or ROCArray
    missing_param::Union{Nothing,BitVector}
    location::Int32             # 0 host, 1 device
end
function _prepare(x::AbstractArray)
    T = _valtype(x)
    if _is_device(x)
        eltype(x) === T || throw(ArgumentError("device arrays must be Float32 or Float64"))
        return Prepared{T,typeof(x)}(reshape(x, size(x, 1), size(x, 2), :), nothing, Int32(1))
    end
    xa = collect(x)                                              # drops axes offsets, keeps the memory order
    x3 = reshape(xa, size(xa, 1), size(xa, 2), :)
    miss = nothing
    if Missing <: eltype(x)
        miss = BitVector(any(ismissing, view(x3, :, :, p)) for p in axes(x3, 3))
        x3 = map(v -> ismissing(v) ? zero(T) : T(v), x3)         # the affected parameters are overwritten with `missing` below
    end
    return Prepared{T,Array{T,3}}(convert(Array{T,3}, x3), miss, Int32(0))
end
_newout(pr::Prepared{T}, n) where {T} = pr.location == 1 ? similar(pr.x3, T, n) : Vector{T}(undef, n)
_ptr(a) = a === nothing ? C_NULL : reinterpret(Ptr{Cvoid}, pointer(a))
_maybescalar(x::AbstractArray{<:Any,0}) = x[]
_maybescalar(x::AbstractArray) = x
# similar(x, T, axes(x)[3:end]) filled with v, `missing` where a draw was missing (src/ess_rhat.jl:464-467, 519-523)
function _out(x, pr::Prepared{T}, v) where {T}
    pr.location == 1 && return _maybescalar(reshape(v, map(length, _param_axes(x))...))
    TV = pr.missing_param === nothing ? T : Union{Missing,T}
    out = similar(x, TV, _param_axes(x))
    for (i, p) in zip(eachindex(out), eachindex(v))
        out[i] = (pr.missing_param !== nothing && pr.missing_param[p]) ? missing : v[p]
    end
    return _maybescalar(out)
end
_all_missing(x) = eltype(x) === Missing                          # nothing to compute (src/ess_rhat.jl:356, 480)
_missing_out(x) = _maybescalar(similar(x, Missing, _param_axes(x)) .= missing)

function _ess_rhat(x, kind::Symbol, outputs; kindcode::Int=_KIND[kind], relative::Bool=false,
                   autocov_method::AbstractAutocovMethod=AutocovMethod(), split_chains::Int=2,
                   maxlag::Int=250, tail_prob::Real=1//10, devices=nothing)
    _all_missing(x) && return _missing_out(x), _missing_out(x)
    pr = _prepare(x)
    T = eltype(pr.x3)
    D, C, P = size(pr.x3)
    niter = D ÷ split_chains
    (outputs & 1) != 0 && !(niter > 4) &&
        @warn "number of draws after splitting must be >4 but is $niter. ESS cannot be computed."
    S = (outputs & 1) != 0 ? _newout(pr, P) : nothing
    R = (outputs & 2) != 0 ? _newout(pr, P) : nothing
    a = MdtArgs(D, C, P, T === Float32 ? 1 : 0, pr.location, kindcode, outputs, _code(autocov_method),
                split_chains, relative, 1, maxlag, Float64(tail_prob), C_NULL)
    x3 = pr.x3
    GC.@preserve x3 S R begin
        _check(ccall((:mdt_ess_rhat, libmdt), Cint,
                     (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                     _ctx_for(devices), a, _ptr(x3), _ptr(S), _ptr(R)))
    end
    return S === nothing ? nothing : _out(x, pr, S), R === nothing ? nothing : _out(x, pr, R)
end

function ess(x::AbstractArray{<:Union{Missing,Real}}; kind=:bulk, kwargs...)
    kind === Statistics.mean && return first(_ess_rhat(x, :basic, 1; kwargs...))  # src/ess_rhat.jl:629
    # the other estimators with a proxy (src/ess_rhat.jl:630-659); the probability of a quantile travels in tail_prob
    kind === Statistics.median && return first(_ess_rhat(x, :basic, 1; kindcode=_EST[:median], kwargs...))
    kind === Statistics.std && return first(_ess_rhat(x, :basic, 1; kindcode=_EST[:std], kwargs...))
    nameof(kind isa Function ? kind : typeof(kind)) === :mad &&  # StatsBase.mad (not a dependency of this shim)
        return first(_ess_rhat(x, :basic, 1; kindcode=_EST[:mad], kwargs...))
    kind isa Base.Fix2{typeof(Statistics.quantile),<:Real} &&
        return first(_ess_rhat(x, :basic, 1; kindcode=_EST[:quantile], kwargs..., tail_prob=kind.x))
    kind isa Symbol && haskey(_KIND, kind) && kind !== :rank ||
        throw(ArgumentError("the `kind` `$kind` is not supported by `ess`"))
    return first(_ess_rhat(x, kind, 1; kwargs...))
end
function rhat(x::AbstractArray{<:Union{Missing,Real}}; kind::Symbol=:rank, split_chains::Int=2, devices=nothing)
    haskey(_KIND, kind) || throw(ArgumentError("the `kind` `$kind` is not supported by `rhat`"))
    return last(_ess_rhat(x, kind, 2; split_chains, devices))
end
function ess_rhat(x::AbstractArray{<:Union{Missing,Real}}; kind::Symbol=:rank, kwargs...)
    haskey(_KIND, kind) || throw(ArgumentError("the `kind` `$kind` is not supported by `ess_rhat`"))
    S, R = _ess_rhat(x, kind, 3; kwargs...)
    return (; ess=S, rhat=R)
end

function rhat_nested(x::AbstractArray{<:Union{Missing,Real}}, superchain_ids::AbstractVector; kind::Symbol=:rank,
                     split_chains::Int=2, devices=nothing)
    ndims(x) >= 2 || throw(ArgumentError("`samples` must have at least 2 dimensions `(draws, chains[, parameters…])`"))
    length(superchain_ids) == size(x, 2) || throw(DimensionMismatch("`superchain_ids` has length $(length(superchain_ids)) but `samples` has $(size(x, 2)) chains"))
    ids = sort(unique(superchain_ids))
    length(ids) >= 2 || throw(ArgumentError("at least 2 superchains are required, got $(length(ids))"))
    groups = [findall(==(i), superchain_ids) for i in ids]
    allequal(length, groups) || throw(ArgumentError("all superchains must contain the same number of chains"))
    inds = Int32.(reduce(hcat, groups) .- 1)             # K x S, 0-based, column-major
    haskey(_KIND, kind) || throw(ArgumentError("the `kind` `$kind` is not supported by `rhat_nested`"))
    _all_missing(x) && return _missing_out(x)
    pr = _prepare(x)
    T = eltype(pr.x3)
    D, C, P = size(pr.x3)
    R = _newout(pr, P)
    a = MdtArgs(D, C, P, T === Float32 ? 1 : 0, pr.location, _KIND[kind], 2, 0, split_chains, 0, 1, 250, 0.1, C_NULL)
    x3 = pr.x3
    GC.@preserve x3 R inds begin
        _check(ccall((:mdt_rhat_nested, libmdt), Cint,
                     (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Int32}, Int32, Int32, Ptr{Cvoid}),
                     _ctx_for(devices), a, _ptr(x3), inds, size(inds, 1), size(inds, 2), _ptr(R)))
    end
    return _out(x, pr, R)
end

function mcse(x::AbstractArray{<:Union{Missing,Real}}; kind=Statistics.mean, autocov_method=AutocovMethod(),
              split_chains::Int=2, maxlag::Int=250, devices=nothing)
    isq = kind === Statistics.median || kind isa Base.Fix2{typeof(Statistics.quantile),<:Real}
    kind === Statistics.mean || kind === Statistics.std || isq ||
        throw(ArgumentError("kind must be mean, std, median or Fix2(quantile, p) on the device path"))  # src/mcse.jl:45-118
    _all_missing(x) && return _missing_out(x)
    pr = _prepare(x)
    T = eltype(pr.x3)
    D, C, P = size(pr.x3)
    out = _newout(pr, P)
    dt = T === Float32 ? 1 : 0
    a = MdtArgs(D, C, P, dt, pr.location, 0, 1, _code(autocov_method), split_chains, 0, 1, maxlag, 0.1, C_NULL)
    x3 = pr.x3
    GC.@preserve x3 out begin
        if isq  # kind code and probability travel in the args (src/mcse.jl:66-92)
            aq = MdtArgs(D, C, P, dt, pr.location, kind === Statistics.median ? _EST[:median] : _EST[:quantile], 1,
                         _code(autocov_method), split_chains, 0, 1, maxlag, kind === Statistics.median ? 0.5 : Float64(kind.x), C_NULL)
            _check(ccall((:mdt_mcse_quantile, libmdt), Cint, (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}), ctx(), aq, _ptr(x3), _ptr(out)))
        elseif kind === Statistics.mean
            _check(ccall((:mdt_mcse_mean, libmdt), Cint, (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}), _ctx_for(devices), a, _ptr(x3), _ptr(out)))
        else
            _check(ccall((:mdt_mcse_std, libmdt), Cint, (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}), ctx(), a, _ptr(x3), _ptr(out)))
        end
    end
    return _out(x, pr, out)
end

# bfmi(energy; dims = 1) (src/bfmi.jl:38-45): one value per chain
function bfmi(energy::AbstractVector{<:Real})
    return bfmi(reshape(energy, :, 1))[1]
end
function bfmi(energy::AbstractMatrix{<:Real}; dims::Int=1)
    e = dims == 1 ? energy : permutedims(energy)
    T = float(eltype(e)); T in (Float32, Float64) || (T = Float64)
    e2 = convert(Matrix{T}, e)
    D, C = size(e2)
    out = Vector{T}(undef, C)
    a = MdtArgs(D, C, 1, T === Float32 ? 1 : 0, 0, 0, 0, 0, 1, 0, 1, 250, 0.1, C_NULL)
    GC.@preserve e2 out _check(ccall((:mdt_bfmi, libmdt), Cint, (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}), ctx(), a, e2, out))
    return out
end

# gelmandiag(samples; alpha) (src/gelmandiag.jl:76-80); Float64 results whatever the eltype (:41-42)
function gelmandiag(chains::AbstractArray{<:Real,3}; alpha::Real=0.05)
    size(chains, 2) > 1 || error("Gelman diagnostic requires at least 2 chains")  # :3
    pr = _prepare(chains)
    x3 = pr.x3
    T = eltype(x3)
    pr.location == 0 || throw(ArgumentError("gelmandiag takes host arrays in this shim"))
    D, C, P = size(x3)
    psrf, psrfci = Vector{Float64}(undef, P), Vector{Float64}(undef, P)
    a = MdtArgs(D, C, P, T === Float32 ? 1 : 0, 0, 0, 0, 0, 1, 0, 1, 250, 0.1, C_NULL)
    GC.@preserve x3 psrf psrfci _check(ccall((:mdt_gelmandiag, libmdt), Cint,
        (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}), ctx(), a, x3, Float64(alpha), psrf, psrfci))
    return (psrf=psrf, psrfci=psrfci)
end

end # module
