# MCMCDiagnosticToolsROC.jl -- thin Julia shim over libmdt_hip.so (include/mdt_hip.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: Julia is not available in the build image.  It shows the
# reference-side binding a maintainer would add: same function names, keyword arguments, array
# convention and error types as MCMCDiagnosticTools.jl (src/MCMCDiagnosticTools.jl:17-25), one
# `ccall` per request.  The tested host mirror is the Python package next to it.  Where Julia and a
# GPU exist, `julia julia/test/runtests.jl` checks this file against the API behaviour the reference's
# own suite demands -- and against MCMCDiagnosticTools.jl itself when that package is installed.
#
# Front door, as in the reference:
#   * `Union{Missing,T}` eltypes: a parameter with any missing draw yields `missing`, the others are computed
#     (src/ess_rhat.jl:382-385, 519-523; src/utils.jl:176-179); an all-`Missing` eltype skips the device (:356, 480);
#   * any number of parameter axes and any axes offsets: outputs are `similar(x, T, axes(x)[3:end])`
#     (src/ess_rhat.jl:464-467, src/utils.jl:197-215; test/ess_rhat.jl:134-165 with OffsetArrays);
#   * `Int` input is promoted to Float64 (src/ess_rhat.jl:353, 465);
#   * device arrays (AMDGPU.jl `ROCArray`) are passed by pointer with `location = MDT_DEVICE`: no host copy;
#   * `devices = [0, 1, ...]`: the multi-GPU context (mdt_ctx_create_multi), params axis cut into contiguous blocks
#     (block i = `shard_bounds(P, i, n)`, one rule everywhere); a `Vector` of ROCArrays, block i on GPU devices[i], is
#     taken PRE-SHARDED (mdt_args.x_shards): no input moves between GPUs;
#   * a `PermutedDimsArray` view of (iterations, parameters, chains) storage -- what MCMCChains keeps -- is read in
#     place through chain_stride / param_stride (`samples_from_chains`), no `permutedims` copy.
module MCMCDiagnosticToolsROC

using Statistics: Statistics

export ess, rhat, ess_rhat, bulk_tail, rhat_nested, mcse, mcse_rhat_nested, bfmi, gelmandiag, status, reserve, shard_bounds, samples_from_chains, UnsupportedOnDevice
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
    x_shards::Ptr{Ptr{Cvoid}}   # multi-GPU context: block i of the params axis on device i (pre-sharded), or C_NULL
    tail_p_lo::Float64          # pl, pu of src/ess_rhat.jl:306-307 evaluated HERE, in the caller's arithmetic (0, 0: derived in the library)
    tail_p_hi::Float64
end
MdtArgs(D, C, P, dtype, location, kind, outputs, autocov, split, relative, sync, maxlag, tail_prob, stream) =
    MdtArgs(D, C, P, dtype, location, kind, outputs, autocov, split, relative, sync, maxlag, tail_prob, stream, 0, 0, C_NULL, 0.0, 0.0)
_with_layout(a::MdtArgs, cs, ps, shards=C_NULL, pl=a.tail_p_lo, pu=a.tail_p_hi) =
    MdtArgs(a.draws, a.chains, a.params, a.dtype, a.location, a.kind, a.outputs, a.autocov, a.split_chains, a.relative,
            a.sync, a.maxlag, a.tail_prob, a.stream, cs, ps, shards, pl, pu)

# An mdt_ctx is single-owner (include/mdt_hip.h: one call at a time per context): the handles are created under this lock
# and every request holds it for the duration of its ccall (`_locked`), so tasks on several threads queue up instead of
# entering the library together.  (The Python mirror keeps one lock per context; one lock for the module is enough here:
# a request occupies the whole GPU anyway.)
const _lock = ReentrantLock()
_locked(f) = lock(f, _lock)
const _ctx = Ref{Ptr{Cvoid}}(C_NULL)
const _multi = Dict{Vector{Cint},Ptr{Cvoid}}()
function ctx()
    _locked() do
        if _ctx[] == C_NULL
            rc = ccall((:mdt_ctx_create, libmdt), Cint, (Ref{Ptr{Cvoid}}, Cint), _ctx, 0)
            rc == 0 || error("mdt_ctx_create failed with status $rc (no MI355X visible?)")
        end
        _ctx[]
    end
end
# one context over several GPUs of the node: contiguous blocks of the params axis, one per device (mdt_ctx_create_multi)
function ctx(devices::AbstractVector{<:Integer})
    ids = collect(Cint, devices)
    _locked() do
        get!(_multi, ids) do
            h = Ref{Ptr{Cvoid}}(C_NULL)
            rc = ccall((:mdt_ctx_create_multi, libmdt), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cint}, Cint), h, ids, length(ids))
            rc == 0 || error("mdt_ctx_create_multi failed with status $rc")
            h[]
        end
    end
end
_ctx_for(devices) = devices === nothing ? ctx() : ctx(devices)
# the context of a prepared request: pre-sharded input lives on several GPUs, and only the caller knows which
function _ctx_for(devices, pr)
    pr.shards === nothing || devices !== nothing ||
        throw(ArgumentError("pre-sharded input (a Vector of device arrays) needs `devices = [id of the GPU holding block 1, ...]`"))
    pr.shards === nothing || length(devices) == length(pr.shards) ||
        throw(DimensionMismatch("$(length(pr.shards)) blocks but $(length(devices)) devices"))
    return _ctx_for(devices)
end
# what every function here accepts: the reference's samples arrays, or the blocks of a pre-sharded one
const Samples = Union{AbstractArray{<:Union{Missing,Real}},AbstractVector{<:AbstractArray{<:Real}}}

"""    status(; devices=nothing, clear=true) -> flags

Sticky condition flags of the context (mdt_ctx_status): 1 ranking tables overflowed, 2 a quantile of data holding
NaN (the reference throws ArgumentError there), 4 FDist.  Only requests issued with `sync = 0` leave flags behind."""
function status(; devices=nothing, clear::Bool=true)
    f = Ref{Int32}(0)
    h = _ctx_for(devices)
    _locked() do
        _check(h, ccall((:mdt_ctx_status, libmdt), Cint, (Ptr{Cvoid}, Ref{Int32}, Int32), h, f, clear))
    end
    return f[]
end

"""    reserve(draws, chains, params; T=Float64, devices=nothing, split_chains=2, host=false)

Allocate and build, before the first request, what a `(draws, chains, params)` array of eltype `T` needs on the context
(mdt_ctx_reserve): scratch, the normal-score table of the shape, ticket tables.  Optional -- a diagnostic is usually
called once per array, and without this its one call also pays for the allocations.  The context warmed is the one a
request with the same `devices` resolves to (`_ctx_for`: this shim keeps one context per device set, whatever the
stream); `host = true` announces an `Array` in host memory (its staging buffers are reserved too), `split_chains` the
keyword the request will carry (it decides which kernel family the shape is handed to)."""
function reserve(draws::Integer, chains::Integer, params::Integer; T::Type{<:Union{Float64,Float32}}=Float64, devices=nothing,
                 split_chains::Integer=2, host::Bool=false)
    h = _ctx_for(devices)
    a = MdtArgs(draws, chains, params, T === Float32 ? 1 : 0, host ? 0 : 1, 0, 0, 0, split_chains, 0, 1, 250, 0.1, C_NULL)
    _locked() do
        _check(h, ccall((:mdt_ctx_reserve, libmdt), Cint, (Ptr{Cvoid}, Ref{MdtArgs}), h, a))
    end
    return nothing
end

"""    shard_bounds(P, i, n) -> (lo, hi)

Block `i` (0-based) of `n` of a params axis of length `P`, as the half-open 0-based range the library cuts by
(mdt_shard_bounds): the first `P mod n` blocks hold one parameter more.  A caller that keeps its array pre-sharded
over the GPUs of a node must cut by the same rule."""
function shard_bounds(P::Integer, i::Integer, n::Integer)
    lo, hi = Ref{Int64}(0), Ref{Int64}(0)
    rc = ccall((:mdt_shard_bounds, libmdt), Cint, (Int64, Int32, Int32, Ref{Int64}, Ref{Int64}), P, i, n, lo, hi)
    rc == 0 || throw(ArgumentError("shard_bounds($P, $i, $n)"))
    return lo[], hi[]
end

"""    UnsupportedOnDevice(msg)

A valid request outside the engine's limits (MDT_EUNSUPPORTED; DESIGN.md §7: effective `maxlag > 1000`, slices beyond
the ranking's round tables).  Never an approximation: a caller that wants every request of the reference served can
catch it and hand the request to MCMCDiagnosticTools.jl on the host (INTEGRATION.md)."""
struct UnsupportedOnDevice <: Exception
    msg::String
end
Base.showerror(io::IO, e::UnsupportedOnDevice) = print(io, "unsupported on the device path: ", e.msg)

# status -> exception; the message is read from the context the call was made on (single- or multi-GPU).
# `domval`: the offending value of a DomainError -- the only one on this path is `maxlag` (src/ess_rhat.jl:481,
# `DomainError(maxlag, "maxlag must be >0.")`), so the callers that take a maxlag pass it.
function _check(h::Ptr{Cvoid}, rc::Cint, domval=0)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:mdt_last_error, libmdt), Cstring, (Ptr{Cvoid},), h))
    rc == 1 && throw(ArgumentError(msg))
    rc == 2 && throw(DomainError(domval, msg))
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
    x3::A                       # the memory handed to the library: (draws, chains, params) dense, or the parent of a strided view
    dims::NTuple{3,Int}         # logical (draws, chains, params)
    cs::Int64                   # chain_stride / param_stride in elements (0, 0: dense)
    ps::Int64
    missing_param::Union{Nothing,BitVector}
    location::Int32             # 0 host, 1 device
    shards::Union{Nothing,Vector{Ptr{Cvoid}}}   # pre-sharded input of a multi-GPU context
end
Prepared{T,A}(x3::A, miss, loc) where {T,A} = Prepared{T,A}(x3, size(x3), 0, 0, miss, loc, nothing)
Base.size(pr::Prepared) = pr.dims

"""    samples_from_chains(value) -> view

`value` is (iterations, parameters, chains) -- the storage order of `MCMCChains.Chains` (`chn.value.data`).  The
result is the (draws, chains, parameters) view every function here takes, WITHOUT a `permutedims` copy: the library
reads the parent in place through `chain_stride = iterations * parameters`, `param_stride = iterations`."""
samples_from_chains(value::AbstractArray{<:Any,3}) = PermutedDimsArray(value, (1, 3, 2))

function _prepare(x::AbstractArray)
    T = _valtype(x)
    # (iterations, parameters, chains) storage viewed as (draws, chains, params): read in place through the strides
    if x isa PermutedDimsArray{T,3,(1, 3, 2)} && (parent(x) isa Array{T,3} || _is_device(parent(x)))
        par = parent(x)
        D, Pn, C = size(par)
        return Prepared{T,typeof(par)}(par, (D, C, Pn), Int64(D) * Pn, Int64(D), nothing, Int32(_is_device(par) ? 1 : 0), nothing)
    end
    if _is_device(x)
        eltype(x) === T || throw(ArgumentError("device arrays must be Float32 or Float64"))
        x3 = reshape(x, size(x, 1), size(x, 2), :)                   # (typeof(x3), not typeof(x): a 2-d or 4-d device array changes rank here)
        return Prepared{T,typeof(x3)}(x3, nothing, Int32(1))
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
# pre-sharded input: blocks[i] is block i of the params axis, a ROCArray on GPU devices[i] (mdt_args.x_shards)
function _prepare(blocks::AbstractVector{<:AbstractArray})
    isempty(blocks) && throw(ArgumentError("no blocks"))
    all(_is_device, blocks) || throw(ArgumentError("pre-sharded input must be device arrays, block i on GPU devices[i]"))
    T = _valtype(first(blocks))
    b3 = [reshape(b, size(b, 1), size(b, 2), :) for b in blocks]
    D, C = size(first(b3), 1), size(first(b3), 2)
    all(b -> eltype(b) === T && size(b, 1) == D && size(b, 2) == C, b3) ||
        throw(DimensionMismatch("every block must be (draws, chains, params_i) with the same draws, chains and eltype"))
    P = sum(b -> size(b, 3), b3)
    for (i, b) in enumerate(b3)
        lo, hi = shard_bounds(P, i - 1, length(b3))
        size(b, 3) == hi - lo || throw(DimensionMismatch("block $i holds $(size(b, 3)) parameters; shard_bounds gives it $(hi - lo) of $P"))
    end
    return Prepared{T,typeof(b3)}(b3, (D, C, P), 0, 0, nothing, Int32(1), [_ptr(b) for b in b3])
end
_newout(pr::Prepared{T}, n) where {T} = pr.location == 1 ? similar(pr.shards === nothing ? pr.x3 : first(pr.x3), T, n) : Vector{T}(undef, n)
_xptr(pr::Prepared) = pr.shards === nothing ? _ptr(pr.x3) : C_NULL
# the args of a request on `pr`: layout and (pre-sharded) block pointers filled in; `sh` must stay GC-preserved during the call
_args_for(pr::Prepared, a::MdtArgs, sh) = _with_layout(a, pr.cs, pr.ps, sh === nothing ? Ptr{Ptr{Cvoid}}(C_NULL) : pointer(sh))
_ptr(a) = a === nothing ? C_NULL : reinterpret(Ptr{Cvoid}, pointer(a))
_maybescalar(x::AbstractArray{<:Any,0}) = x[]
_maybescalar(x::AbstractArray) = x
# similar(x, T, axes(x)[3:end]) filled with v, `missing` where a draw was missing (src/ess_rhat.jl:464-467, 519-523)
_param_axes(x::AbstractVector{<:AbstractArray}) = (Base.OneTo(sum(b -> size(reshape(b, size(b, 1), size(b, 2), :), 3), x)),)
function _out(x, pr::Prepared{T}, v) where {T}
    if pr.location == 1   # device results stay on the device; a 0-dimensional result comes to the host as a scalar (no scalar indexing on the GPU)
        pa = _param_axes(x)
        return isempty(pa) ? Array(v)[1] : reshape(v, map(length, pa)...)
    end
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
    T = _valtype(x isa AbstractVector{<:AbstractArray} ? first(x) : x)
    D, C, P = size(pr)
    niter = D ÷ split_chains
    (outputs & 1) != 0 && !(niter > 4) &&
        @warn "number of draws after splitting must be >4 but is $niter. ESS cannot be computed."
    S = (outputs & 1) != 0 ? _newout(pr, P) : nothing
    R = (outputs & 2) != 0 ? _newout(pr, P) : nothing
    # src/ess_rhat.jl:305-307, evaluated here exactly as the reference does (a Rational tail_prob halves exactly):
    #   T = float(promote_type(eltype(x), typeof(tail_prob))); pl = convert(T, tail_prob / 2); pu = convert(T, 1 - tail_prob / 2)
    # and handed over as they are (tail_p_lo / tail_p_hi): the library does not redo the arithmetic in double
    TP = float(promote_type(T, typeof(tail_prob)))
    pl, pu = Float64(convert(TP, tail_prob / 2)), Float64(convert(TP, 1 - tail_prob / 2))
    tp = kindcode == _EST[:quantile] ? Float64(T(tail_prob)) : Float64(tail_prob)
    a0 = _args_for(pr, MdtArgs(D, C, P, T === Float32 ? 1 : 0, pr.location, kindcode, outputs, _code(autocov_method),
                               split_chains, relative, 1, maxlag, tp, C_NULL), pr.shards)
    a = kindcode == _EST[:quantile] ? a0 : _with_layout(a0, a0.chain_stride, a0.param_stride, a0.x_shards, pl, pu)
    x3, sh, h = pr.x3, pr.shards, _ctx_for(devices, pr)
    GC.@preserve x3 sh S R begin
        _locked() do
            _check(h, ccall((:mdt_ess_rhat, libmdt), Cint,
                            (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                            h, a, _xptr(pr), _ptr(S), _ptr(R)), maxlag)
        end
    end
    return S === nothing ? nothing : _out(x, pr, S), R === nothing ? nothing : _out(x, pr, R)
end

function ess(x::Samples; kind=:bulk, kwargs...)
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
function rhat(x::Samples; kind::Symbol=:rank, split_chains::Int=2, devices=nothing)
    haskey(_KIND, kind) || throw(ArgumentError("the `kind` `$kind` is not supported by `rhat`"))
    return last(_ess_rhat(x, kind, 2; split_chains, devices))
end
function ess_rhat(x::Samples; kind::Symbol=:rank, kwargs...)
    haskey(_KIND, kind) || throw(ArgumentError("the `kind` `$kind` is not supported by `ess_rhat`"))
    S, R = _ess_rhat(x, kind, 3; kwargs...)
    return (; ess=S, rhat=R)
end

"""
    bulk_tail(samples; relative=false, autocov_method=AutocovMethod(), split_chains=2, maxlag=250, tail_prob=1//10, devices=nothing)

`ess_rhat(samples; kind=:bulk)` and `ess_rhat(samples; kind=:tail)` from ONE pass over `samples` (`mdt_bulk_tail`: the pooled
ranking of `x` is shared; `src/ess_rhat.jl:604-616`) -- what a summary table asks for, and the BASELINE metric.  Returns
`(; ess_bulk, ess_tail, rhat_bulk, rhat_tail)`; `rhat(samples)` (`kind = :rank`) is `max.(rhat_tail, rhat_bulk)` (`:416-420`).
The reference has no fused counterpart: it runs the two kinds one after the other.
"""
function bulk_tail(x::Samples; relative::Bool=false, autocov_method::AbstractAutocovMethod=AutocovMethod(), split_chains::Int=2,
                   maxlag::Int=250, tail_prob::Real=1//10, devices=nothing)
    if _all_missing(x)
        m = _missing_out(x)
        return (; ess_bulk=m, ess_tail=m, rhat_bulk=m, rhat_tail=m)
    end
    pr = _prepare(x)
    T = _valtype(x isa AbstractVector{<:AbstractArray} ? first(x) : x)
    D, C, P = size(pr)
    niter = D ÷ split_chains
    !(niter > 4) && @warn "number of draws after splitting must be >4 but is $niter. ESS cannot be computed."
    Sb, St, Rb, Rt = _newout(pr, P), _newout(pr, P), _newout(pr, P), _newout(pr, P)
    TP = float(promote_type(T, typeof(tail_prob)))          # src/ess_rhat.jl:305-307, in the caller's arithmetic (see _ess_rhat)
    pl, pu = Float64(convert(TP, tail_prob / 2)), Float64(convert(TP, 1 - tail_prob / 2))
    a0 = _args_for(pr, MdtArgs(D, C, P, T === Float32 ? 1 : 0, pr.location, 0, 3, _code(autocov_method),
                               split_chains, relative, 1, maxlag, Float64(tail_prob), C_NULL), pr.shards)
    a = _with_layout(a0, a0.chain_stride, a0.param_stride, a0.x_shards, pl, pu)
    x3, sh, h = pr.x3, pr.shards, _ctx_for(devices, pr)
    GC.@preserve x3 sh Sb St Rb Rt begin
        _locked() do
            _check(h, ccall((:mdt_bulk_tail, libmdt), Cint,
                            (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                            h, a, _xptr(pr), _ptr(Sb), _ptr(St), _ptr(Rb), _ptr(Rt)), maxlag)
        end
    end
    return (; ess_bulk=_out(x, pr, Sb), ess_tail=_out(x, pr, St), rhat_bulk=_out(x, pr, Rb), rhat_tail=_out(x, pr, Rt))
end

# _validate_superchain_ids (src/rhat_nested.jl:68-81) -> K x S, 0-based, column-major
function _superchain_inds(superchain_ids::AbstractVector, nchains::Integer)
    length(superchain_ids) == nchains || throw(DimensionMismatch("`superchain_ids` has length $(length(superchain_ids)) but `samples` has $nchains chains"))
    ids = sort(unique(superchain_ids))
    length(ids) >= 2 || throw(ArgumentError("at least 2 superchains are required, got $(length(ids))"))
    groups = [findall(==(i), superchain_ids) for i in ids]
    allequal(length, groups) || throw(ArgumentError("all superchains must contain the same number of chains"))
    return Int32.(reduce(hcat, groups) .- 1)
end
_nchains(x::AbstractArray) = size(x, 2)
_nchains(x::AbstractVector{<:AbstractArray}) = size(first(x), 2)

function rhat_nested(x::Samples, superchain_ids::AbstractVector; kind::Symbol=:rank,
                     split_chains::Int=2, devices=nothing)
    (x isa AbstractVector{<:AbstractArray} || ndims(x) >= 2) || throw(ArgumentError("`samples` must have at least 2 dimensions `(draws, chains[, parameters…])`"))
    inds = _superchain_inds(superchain_ids, _nchains(x))
    haskey(_KIND, kind) || throw(ArgumentError("the `kind` `$kind` is not supported by `rhat_nested`"))
    _all_missing(x) && return _missing_out(x)
    pr = _prepare(x)
    T = _valtype(x isa AbstractVector{<:AbstractArray} ? first(x) : x)
    D, C, P = size(pr)
    R = _newout(pr, P)
    a = _args_for(pr, MdtArgs(D, C, P, T === Float32 ? 1 : 0, pr.location, _KIND[kind], 2, 0, split_chains, 0, 1, 250, 0.1, C_NULL), pr.shards)
    x3, sh, h = pr.x3, pr.shards, _ctx_for(devices, pr)
    GC.@preserve x3 sh R inds begin
        _locked() do
            _check(h, ccall((:mdt_rhat_nested, libmdt), Cint,
                            (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Int32}, Int32, Int32, Ptr{Cvoid}),
                            h, a, _xptr(pr), inds, size(inds, 1), size(inds, 2), _ptr(R)))
        end
    end
    return _out(x, pr, R)
end

"""    mcse_rhat_nested(x, superchain_ids; kind=:rank, ...) -> (mcse, rhat_nested)

`mcse(x)` (kind = mean) and `rhat_nested(x, superchain_ids; kind)` from ONE launch (mdt_mcse_nested): a parameter's
slice is read from HBM once for both.  Bit-identical to the two calls; no counterpart in the reference."""
function mcse_rhat_nested(x::Samples, superchain_ids::AbstractVector; kind::Symbol=:rank, split_chains::Int=2,
                          maxlag::Int=250, autocov_method::AbstractAutocovMethod=AutocovMethod(), devices=nothing)
    inds = _superchain_inds(superchain_ids, _nchains(x))
    haskey(_KIND, kind) || throw(ArgumentError("the `kind` `$kind` is not supported by `rhat_nested`"))
    pr = _prepare(x)
    T = _valtype(x isa AbstractVector{<:AbstractArray} ? first(x) : x)
    D, C, P = size(pr)
    M, R = _newout(pr, P), _newout(pr, P)
    a = _args_for(pr, MdtArgs(D, C, P, T === Float32 ? 1 : 0, pr.location, _KIND[kind], 3, _code(autocov_method), split_chains, 0, 1, maxlag, 0.1, C_NULL), pr.shards)
    x3, sh, h = pr.x3, pr.shards, _ctx_for(devices, pr)
    GC.@preserve x3 sh M R inds begin
        _locked() do
            _check(h, ccall((:mdt_mcse_nested, libmdt), Cint,
                            (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Int32}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
                            h, a, _xptr(pr), inds, size(inds, 1), size(inds, 2), _ptr(M), _ptr(R)), maxlag)
        end
    end
    return _out(x, pr, M), _out(x, pr, R)
end

function mcse(x::Samples; kind=Statistics.mean, autocov_method=AutocovMethod(),
              split_chains::Int=2, maxlag::Int=250, devices=nothing)
    isq = kind === Statistics.median || kind isa Base.Fix2{typeof(Statistics.quantile),<:Real}
    kind === Statistics.mean || kind === Statistics.std || isq ||
        throw(ArgumentError("kind must be mean, std, median or Fix2(quantile, p) on the device path"))  # src/mcse.jl:45-118
    _all_missing(x) && return _missing_out(x)
    pr = _prepare(x)
    T = _valtype(x isa AbstractVector{<:AbstractArray} ? first(x) : x)
    D, C, P = size(pr)
    out = _newout(pr, P)
    dt = T === Float32 ? 1 : 0
    a = _args_for(pr, MdtArgs(D, C, P, dt, pr.location, 0, 1, _code(autocov_method), split_chains, 0, 1, maxlag, 0.1, C_NULL), pr.shards)
    x3, sh, h = pr.x3, pr.shards, _ctx_for(devices, pr)   # every kind runs on a multi-GPU context (each block on its own GPU)
    GC.@preserve x3 sh out begin
        _locked() do
            if isq  # kind code and probability travel in the args (src/mcse.jl:66-92)
                aq = _args_for(pr, MdtArgs(D, C, P, dt, pr.location, kind === Statistics.median ? _EST[:median] : _EST[:quantile], 1,
                                           _code(autocov_method), split_chains, 0, 1, maxlag, kind === Statistics.median ? 0.5 : Float64(T(kind.x)), C_NULL), pr.shards)
                _check(h, ccall((:mdt_mcse_quantile, libmdt), Cint, (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}), h, aq, _xptr(pr), _ptr(out)), maxlag)
            elseif kind === Statistics.mean
                _check(h, ccall((:mdt_mcse_mean, libmdt), Cint, (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}), h, a, _xptr(pr), _ptr(out)), maxlag)
            else
                _check(h, ccall((:mdt_mcse_std, libmdt), Cint, (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}), h, a, _xptr(pr), _ptr(out)), maxlag)
            end
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
    h = ctx()
    GC.@preserve e2 out begin
        _locked() do
            _check(h, ccall((:mdt_bfmi, libmdt), Cint, (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Ptr{Cvoid}), h, a, e2, out))
        end
    end
    return out
end

# gelmandiag(samples; alpha) (src/gelmandiag.jl:76-80); Float64 results whatever the eltype (:41-42)
function gelmandiag(chains::AbstractArray{<:Real,3}; alpha::Real=0.05)
    size(chains, 2) > 1 || error("Gelman diagnostic requires at least 2 chains")  # :3
    pr = _prepare(chains isa PermutedDimsArray ? collect(chains) : chains)   # (the column pass reads dense arrays only)
    x3 = pr.x3
    T = eltype(x3)
    pr.location == 0 || throw(ArgumentError("gelmandiag takes host arrays in this shim"))
    D, C, P = size(x3)
    psrf, psrfci = Vector{Float64}(undef, P), Vector{Float64}(undef, P)
    a = MdtArgs(D, C, P, T === Float32 ? 1 : 0, 0, 0, 0, 0, 1, 0, 1, 250, 0.1, C_NULL)
    h = ctx()
    GC.@preserve x3 psrf psrfci begin
        _locked() do
            _check(h, ccall((:mdt_gelmandiag, libmdt), Cint,
                            (Ptr{Cvoid}, Ref{MdtArgs}, Ptr{Cvoid}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}), h, a, x3, Float64(alpha), psrf, psrfci))
        end
    end
    return (psrf=psrf, psrfci=psrfci)
end

end # module
