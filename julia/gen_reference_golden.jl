# gen_reference_golden.jl -- pin the oracle to numbers MCMCDiagnosticTools.jl ITSELF returns.
#
# The build image has no Julia, so the committed golden vectors (tests/golden/*.npz) were minted by two independent
# restatements of the reference (oracle/mdt_oracle.c, oracle/numpy_oracle.py; DESIGN.md section 2) and parity is
# "unpinned" by reference output.  This script closes that: run it once on any machine with Julia, commit the
# files it writes, and tests/test_reference_pin.py compares the oracle -- and, on a GPU box, the HIP path -- with them.
#
#   julia -e 'using Pkg; Pkg.add(["MCMCDiagnosticTools", "NPZ", "StatsBase"])'      # reference v0.3.19 (Project.toml compat)
#   julia julia/gen_reference_golden.jl [tests/golden]
#
# For every fixture <name>.npz it reads the stored INPUT `x` (and nothing else) and writes ref_<name>.npz with the
# reference's answers under the same keys the fixture uses:
#   ess|kind|method|split|maxlag, rhat|...      ess_rhat(x; kind, autocov_method, split_chains, maxlag)   src/ess_rhat.jl:438-624
#   nested|kind|split, mcse_mean                rhat_nested / mcse                                         src/rhat_nested.jl:43-125, src/mcse.jl:40-51
#   essF|..., rhatF|...                         the same calls (Float32 fixtures: the reference computes in Float32 -- the "faithful" keys)
#   bulk_tail_fft                               [ess bulk; ess tail; rhat bulk; rhat tail] with FFTAutocovMethod, 4 x params
#   est|estimator|p|method|split, err|est|...   ess(x; kind = f); err|... = the fixture expects the call to throw               src/ess_rhat.jl:291-297
#   mcse_std, mcseq|estimator|p                 mcse(x; kind = std) / mcse(x; kind = median | Fix2(quantile, p))                 src/mcse.jl:52-118
#   gelman|alpha|psrf(ci), bfmi                 gelmandiag / bfmi over all chains x params columns         src/gelmandiag.jl:76-80, src/bfmi.jl:38-45
#   stage_rank[0|1], stage_z, stage_fold, stage_q05   StatsBase.tiedrank, _rank_normalize, _fold_around_median, quantile   src/utils.jl:148-193
# Not written (no exported counterpart): stage_acov_* (the per-lag mean_autocov of the internal caches), inds, x.
# Keys whose call throws in the reference (e.g. quantile of data with NaN) are written as a 1-element array holding
# the sentinel -12345.0 so the comparison can check that the device path refuses them too.
using MCMCDiagnosticTools, NPZ, Statistics, StatsBase

const MDT = MCMCDiagnosticTools
const METHODS = Dict("direct" => AutocovMethod(), "fft" => FFTAutocovMethod(), "bda" => BDAAutocovMethod())
const THREW = [-12345.0]

guard(f) = try
    f()
catch err
    err isa Union{ArgumentError,DomainError,DimensionMismatch} || rethrow()
    THREW
end
arr(v) = v isa Number ? [Float64(v)] : collect(Float64, v)

function estimator(name, p)
    name == "median" && return Statistics.median
    name == "std" && return Statistics.std
    name == "mad" && return StatsBase.mad
    name == "mean" && return Statistics.mean
    return Base.Fix2(Statistics.quantile, p)
end

function reference_answers(name::AbstractString, d::Dict)
    out = Dict{String,Any}()
    haskey(d, "x") || return out
    x = d["x"]                                   # (draws, chains, params), the eltype the fixture was minted with
    for key in keys(d)
        parts = split(key, "|")
        if parts[1] == "ess" && length(parts) == 5
            kind, m, s, ml = Symbol(parts[2]), METHODS[parts[3]], parse(Int, parts[4]), parse(Int, parts[5])
            r = guard(() -> ess_rhat(x; kind, autocov_method=m, split_chains=s, maxlag=ml))
            out[key] = r === THREW ? THREW : arr(r.ess)
            out["rhat|" * join(parts[2:end], "|")] = r === THREW ? THREW : arr(r.rhat)
        elseif parts[1] == "nested" && length(parts) == 3
            ids = repeat(1:4; inner=size(x, 2) ÷ 4)   # the fixtures use four superchains of equal size
            out[key] = arr(guard(() -> rhat_nested(x, ids; kind=Symbol(parts[2]), split_chains=parse(Int, parts[3]))))
        elseif key == "mcse_mean"
            out[key] = arr(guard(() -> mcse(x)))
        elseif (parts[1] == "essF" || parts[1] == "rhatF") && length(parts) == 5
            kind, m, s, ml = Symbol(parts[2]), METHODS[parts[3]], parse(Int, parts[4]), parse(Int, parts[5])
            r = guard(() -> ess_rhat(x; kind, autocov_method=m, split_chains=s, maxlag=ml))
            out[key] = r === THREW ? THREW : arr(parts[1] == "essF" ? r.ess : r.rhat)
        elseif key == "bulk_tail_fft"
            b = guard(() -> ess_rhat(x; kind=:bulk, autocov_method=FFTAutocovMethod()))
            t = guard(() -> ess_rhat(x; kind=:tail, autocov_method=FFTAutocovMethod()))
            out[key] = (b === THREW || t === THREW) ? THREW : permutedims(hcat(arr(b.ess), arr(t.ess), arr(b.rhat), arr(t.rhat)))  # 4 x params
        elseif (parts[1] == "est" && length(parts) == 5) || (parts[1] == "err" && length(parts) == 6 && parts[2] == "est")
            q = parts[1] == "est" ? parts[2:end] : parts[3:end]
            f = estimator(q[1], parse(Float64, q[2]))
            out[key] = arr(guard(() -> ess(x; kind=f, autocov_method=METHODS[q[3]], split_chains=parse(Int, q[4]))))
        elseif key == "mcse_std"
            out[key] = arr(guard(() -> mcse(x; kind=Statistics.std)))
        elseif parts[1] == "mcseq" && length(parts) == 3
            f = estimator(parts[2], parse(Float64, parts[3]))
            out[key] = arr(guard(() -> mcse(x; kind=f)))
        elseif parts[1] == "gelman" && length(parts) == 3
            g = guard(() -> gelmandiag(x; alpha=parse(Float64, parts[2])))
            out[key] = g === THREW ? THREW : arr(getproperty(g, Symbol(parts[3])))
        elseif key == "bfmi"
            out[key] = arr(guard(() -> bfmi(reshape(x, size(x, 1), :))))   # one value per (chain, param) column
        end
    end
    for (key, ip) in (("stage_rank0", 1), ("stage_rank1", 2))
        haskey(d, key) && (out[key] = reshape(StatsBase.tiedrank(vec(x[:, :, ip])), size(x[:, :, ip])))
    end
    if haskey(d, "stage_rank")                     # the third-party boundaries, parameter 1
        xp = x[:, :, 1]
        out["stage_rank"] = reshape(StatsBase.tiedrank(vec(xp)), size(xp))
        out["stage_z"] = MDT._rank_normalize(xp)
        out["stage_fold"] = MDT._fold_around_median(xp)
        out["stage_q05"] = [Statistics.quantile(vec(xp), 0.05), Statistics.quantile(vec(xp), 0.95)]
    end
    return out
end

function main(dir)
    for f in sort(readdir(dir))
        (endswith(f, ".npz") && !startswith(f, "ref_")) || continue
        d = npzread(joinpath(dir, f))
        src = d
        if !haskey(d, "x")                          # est_<name> / diag_<name> share the input of <name>
            base = replace(f, r"^(est|diag)_" => "")
            isfile(joinpath(dir, base)) || continue
            src = merge(d, Dict("x" => npzread(joinpath(dir, base))["x"]))
        end
        out = reference_answers(f, src)
        isempty(out) && continue
        npzwrite(joinpath(dir, "ref_" * f), out)
        println("ref_", f, ": ", length(out), " keys")
    end
end

main(length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden"))
