# julia/test/runtests.jl -- the API behaviour of MCMCDiagnosticToolsROC (the ccall shim over libmdt_hip.so).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI (no Julia in the build image).  One command where Julia and an MI355X exist:
#
#     MDT_HIP_LIB=/path/to/libmdt_hip.so julia julia/test/runtests.jl
#
# Part 1 demands of the shim what the reference's own suite demands of MCMCDiagnosticTools.jl on this path -- result
# types and shapes, error types, Missing, offset axes, slice-wise `==`, the known answers -- each testset citing the
# reference test it answers to (test/ess_rhat.jl:42-204, 242-257, 314-335; test/rhat_nested.jl:26-155;
# test/mcse.jl:9-101).  Part 2 runs when MCMCDiagnosticTools.jl itself is installed: the same random inputs through
# both packages, 1e-6 relative (BASELINE.json) -- the reference-output pin that the build image could not produce --
# and then julia/gen_reference_golden.jl writes tests/golden/ref_*.npz for the Python side (tests/test_reference_pin.py).
using Test
using Random
using Statistics

include(joinpath(@__DIR__, "..", "MCMCDiagnosticToolsROC.jl"))
const ROC = MCMCDiagnosticToolsROC

const HAVE_OFFSET = try
    @eval using OffsetArrays
    true
catch
    false
end
const HAVE_REF = try
    @eval import MCMCDiagnosticTools
    true
catch
    false
end
const KINDS = (:rank, :bulk, :tail, :basic)
esskind(k) = k === :rank ? :bulk : k          # `ess` has no :rank (src/ess_rhat.jl:276-290)
outtype(T, sz) = length(sz) < 3 ? float(T) : Array{float(T),length(sz) - 2}
sample(T, sz) = T <: Integer ? rand(1:10, sz...) : rand(T, sz...)

Random.seed!(20261004)

@testset "MCMCDiagnosticToolsROC" begin
    @testset "result types follow the input (test/ess_rhat.jl:42-64)" begin
        for sz in ((100,), (100, 4), (100, 4, 2), (100, 4, 2, 3)), T in (Float32, Float64, Int)
            x = sample(T, sz)
            TV = outtype(T, sz)
            for kind in KINDS
                kind === :rank || @test ROC.ess(x; kind) isa TV
                @test ROC.rhat(x; kind) isa TV
                er = ROC.ess_rhat(x; kind)
                @test er isa NamedTuple{(:ess, :rhat),Tuple{TV,TV}}
            end
            for est in (mean, median, std, Base.Fix2(quantile, 0.25))
                @test ROC.ess(x; kind=est) isa TV
            end
        end
    end

    @testset "too few draws, maxlag domain, unknown kinds (test/ess_rhat.jl:66-98)" begin
        short1, short2, long, one = rand(4, 3, 5), rand(5, 3, 5), rand(100, 3, 5), rand(1, 3, 5)
        for kind in KINDS
            # niter <= 4: ESS is NaN (with a warning), R-hat is still rhat's value bit for bit
            er = @test_logs (:warn,) match_mode = :any ROC.ess_rhat(short1; split_chains=1, kind)
            @test all(isnan, er.ess)
            @test er.rhat == ROC.rhat(short1; split_chains=1, kind)
            @test all(isnan, ROC.ess_rhat(one; split_chains=2, kind).ess)
            er2 = ROC.ess_rhat(short2; split_chains=2, kind)
            @test all(isnan, er2.ess) && er2.rhat == ROC.rhat(short2; split_chains=2, kind)
            @test !any(isnan, ROC.ess_rhat(short2; split_chains=1, kind).ess)   # niter = 5 is enough
            @test !any(isnan, ROC.ess_rhat(long; maxlag=1, kind).ess)
            err = try
                ROC.ess_rhat(long; maxlag=0, kind)
                nothing
            catch e
                e
            end
            @test err isa DomainError && err.val == 0      # DomainError(maxlag, ...), src/ess_rhat.jl:481
            @test_throws DomainError ROC.ess_rhat(long; maxlag=-3, kind)
            if kind !== :rank
                @test all(isnan, ROC.ess(short1; split_chains=1, kind))
                @test_throws DomainError ROC.ess(long; maxlag=0, kind)
            end
        end
        @test_throws ArgumentError ROC.ess(short2; kind=:foo)
        @test_throws ArgumentError ROC.ess(short2; kind=:rank)
        @test_throws ArgumentError ROC.ess_rhat(short2; kind=:foo)
        @test_throws ArgumentError ROC.rhat(short2; kind=:foo)
        @test_throws ArgumentError ROC.ess(short2; kind=x -> mean(x))      # no expectand proxy for an arbitrary function
        @test_throws MethodError ROC.ess(["a" "b"; "c" "d"])               # non-numeric input fails at the door, as in the reference
    end

    @testset "relative = true (test/ess_rhat.jl:100-113)" begin
        for kind in KINDS, niter in (50, 100), nchains in (2, 4)
            x = rand(niter, nchains, 2)
            ss = niter * nchains
            kind === :rank || @test ROC.ess(x; kind, relative=true) ≈ ROC.ess(x; kind) / ss
            S, R = ROC.ess_rhat(x; kind)
            S2, R2 = ROC.ess_rhat(x; kind, relative=true)
            @test S2 ≈ S / ss
            @test R2 == R
        end
    end

    @testset "a missing draw makes its parameter missing (test/ess_rhat.jl:115-132)" begin
        for kind in KINDS
            x = Array{Union{Missing,Float64}}(undef, 1000, 4, 3)
            x .= randn.()
            x[1, 1, 1] = missing
            S1, R1 = ROC.ess(x; kind=esskind(kind)), ROC.rhat(x; kind)
            S2, R2 = ROC.ess_rhat(x; kind)
            for v in (S1, R1, S2, R2)
                @test ismissing(v[1]) && !any(ismissing, v[2:3])
            end
            # the computed parameters are those of the array without the missing one
            @test collect(skipmissing(R1)) == ROC.rhat(Float64.(x[:, :, 2:3]); kind)
        end
    end

    HAVE_OFFSET && @testset "offset axes and any number of parameter axes (test/ess_rhat.jl:134-165)" begin
        for kind in KINDS, ax in ((-5:94, 2:5, 11:15), (-5:94, 2:5, 11:15, 0:2))
            x = randn(map(length, ax)...)
            N = ndims(x)
            y = OffsetArray(x, ax...)
            S1, R1 = ROC.ess(y; kind=esskind(kind)), ROC.rhat(y; kind)
            S2, R2 = ROC.ess_rhat(y; kind)
            for v in (S1, R1, S2, R2)
                @test v isa OffsetArray{Float64,N - 2} && axes(v) == ax[3:end]
            end
            @test ROC.ess(x; kind=esskind(kind)) == collect(S1) == collect(S2)
            @test ROC.rhat(x; kind) == collect(R1) == collect(R2)
            ym = OffsetArray(similar(x, Missing), ax...)
            for v in (ROC.ess(ym; kind=esskind(kind)), ROC.rhat(ym; kind), ROC.ess_rhat(ym; kind)...)
                @test v isa OffsetArray{Missing,N - 2} && axes(v) == ax[3:end]
            end
        end
    end

    @testset "a slice gives the bits of the batch (test/ess_rhat.jl:167-204)" begin
        x = randn(1000, 4, 10, 3)
        for kind in KINDS, split_chains in (1, 2)
            R1 = ROC.rhat(x; kind, split_chains)
            for i in axes(x, 4)
                @test ROC.rhat(x[:, :, :, i]; kind, split_chains) == R1[:, i]
                @test all(ROC.rhat(x[:, :, j, i]; kind, split_chains) == R1[j, i] for j in axes(x, 3))
            end
            for autocov_method in (ROC.AutocovMethod(), ROC.BDAAutocovMethod()), maxlag in (100, 10)
                kw = (; split_chains, autocov_method, maxlag)
                S1 = ROC.ess(x; kind=esskind(kind), kw...)
                S2, R2 = ROC.ess_rhat(x; kind, kw...)
                @test S1 == S2 && R1 == R2
                for i in axes(x, 4)
                    xi = x[:, :, :, i]
                    @test ROC.ess(xi; kind=esskind(kind), kw...) == S1[:, i]
                    @test ROC.ess_rhat(xi; kind, kw...) == (ess=S1[:, i], rhat=R1[:, i])
                    for j in axes(x, 3)
                        @test ROC.ess_rhat(x[:, :, j, i]; kind, kw...) == (ess=S1[j, i], rhat=R1[j, i])
                    end
                end
            end
        end
    end

    @testset "known answers (test/ess_rhat.jl:210-257, 314-335)" begin
        # IID draws: ESS within 10 % of the sample size... the reference allows a wide band; so do we
        x = randn(10_000, 4, 20)
        for m in (ROC.AutocovMethod(), ROC.FFTAutocovMethod(), ROC.BDAAutocovMethod())
            S, R = ROC.ess_rhat(x; kind=:basic, autocov_method=m)
            @test all(s -> 0.85 * 40_000 < s < 1.15 * 40_000, S) && all(r -> isapprox(r, 1; rtol=0.01), R)
        end
        Rd = ROC.rhat(x; kind=:basic)
        @test Rd == ROC.ess_rhat(x; kind=:basic, autocov_method=ROC.FFTAutocovMethod()).rhat   # identical across methods
        @test isapprox(ROC.ess(x; kind=:basic, autocov_method=ROC.FFTAutocovMethod()), ROC.ess(x; kind=:basic); rtol=1e-9)
        # constant chains: NaN, NaN
        c = ones(100, 4, 3)
        @test all(isnan, ROC.ess_rhat(c; kind=:basic).ess) && all(isnan, ROC.rhat(c; kind=:basic))
        # antithetic chains: ESS capped at ntotal * log10(ntotal), exactly
        nd, nc = 1000, 4
        z = randn(nd, nc, 5)
        z[2:2:end, :, :] .= .-z[1:2:end, :, :]
        @test all(==(nd * nc * log10(nd * nc)), ROC.ess(z; kind=:basic))
        # bulk == basic of the rank-normalised draws <=> invariant under strictly increasing maps
        y = randn(500, 4, 6)
        for kind in (:bulk, :tail, :rank)
            @test ROC.rhat(exp.(y); kind) == ROC.rhat(y; kind)
        end
        @test ROC.ess(exp.(y); kind=:bulk) == ROC.ess(y; kind=:bulk)
    end

    @testset "rhat_nested: types, errors, Missing (test/rhat_nested.jl:26-75)" begin
        ids = [1, 1, 2, 2]
        for kind in KINDS, sz in ((100, 4), (100, 4, 2), (100, 4, 2, 3)), T in (Float32, Float64, Int)
            @test ROC.rhat_nested(sample(T, sz), ids; kind) isa outtype(T, sz)
        end
        x = rand(100, 4)
        @test_throws ArgumentError ROC.rhat_nested(rand(100), [1])
        @test_throws DimensionMismatch ROC.rhat_nested(x, [1, 1, 2])
        @test_throws DimensionMismatch ROC.rhat_nested(x, [1, 1, 2, 2, 3])
        @test_throws ArgumentError ROC.rhat_nested(x, [1, 1, 1, 1])
        @test_throws ArgumentError ROC.rhat_nested(x, [1, 1, 1, 2])
        @test_throws ArgumentError ROC.rhat_nested(x, ids; kind=:foo)
        for kind in KINDS
            xm = Array{Union{Missing,Float64}}(undef, 1000, 4, 3)
            xm .= randn.()
            xm[1, 1, 1] = missing
            R = ROC.rhat_nested(xm, ids; kind)
            @test ismissing(R[1]) && !any(ismissing, R[2:3])
            @test ROC.rhat_nested(randn(100, 4), ids; kind) isa Float64
        end
        if HAVE_OFFSET
            for kind in KINDS, ax in ((-5:94, 2:5, 11:15), (-5:94, 2:5, 11:15, 0:2))
                xo = randn(map(length, ax)...)
                R = ROC.rhat_nested(OffsetArray(xo, ax...), ids; kind)
                @test R isa OffsetArray{Float64,ndims(xo) - 2} && axes(R) == ax[3:end]
                @test ROC.rhat_nested(xo, ids; kind) == collect(R)
                Rm = ROC.rhat_nested(OffsetArray(similar(xo, Missing), ax...), ids; kind)
                @test Rm isa OffsetArray{Missing,ndims(xo) - 2} && axes(Rm) == ax[3:end]
            end
        end
    end

    @testset "rhat_nested: invariances and the link to rhat (test/rhat_nested.jl:102-155)" begin
        x = randn(100, 4, 10)
        for kind in KINDS
            R = ROC.rhat_nested(x, [1, 1, 2, 2]; kind)
            @test R == ROC.rhat_nested(x, [42, 42, 99, 99]; kind) == ROC.rhat_nested(x, ['a', 'a', 'b', 'b']; kind)
        end
        for nsuper in (8, 16), per in (2, 4), split_chains in (1, 2), kind in KINDS
            nchains = nsuper * per
            xs = randn(100, nchains, 10)
            ids = repeat(1:nsuper; inner=per)
            perm = randperm(nchains)
            @test ROC.rhat_nested(xs, ids; kind, split_chains) == ROC.rhat_nested(xs[:, perm, :], ids[perm]; kind, split_chains)
        end
        for kind in (:basic, :bulk, :tail), nd in (10, 20), nc in (4, 8)
            xs = randn(nd, nc, 10)
            Rn = ROC.rhat_nested(xs, collect(1:nc); kind, split_chains=1)
            @test Rn ≈ sqrt.(ROC.rhat(xs; kind, split_chains=1) .^ 2 .+ 1 / nd)
        end
        Rb, Rt = ROC.rhat_nested(x, [1, 1, 2, 2]; kind=:bulk), ROC.rhat_nested(x, [1, 1, 2, 2]; kind=:tail)
        @test ROC.rhat_nested(x, [1, 1, 2, 2]; kind=:rank) == max.(Rb, Rt)
        many = randn(10, 2048, 4)                       # the reference's many-short-chains case
        @test all(r -> isapprox(r, 1; atol=0.05), ROC.rhat_nested(many, repeat(1:8; inner=256); kind=:basic))
    end

    @testset "mcse (test/mcse.jl:9-101)" begin
        x = randn(100, 4, 10)
        @test ROC.mcse(x) == ROC.mcse(x; kind=mean)
        for f in (mean, median, std, Base.Fix2(quantile, 0.1))
            @test ROC.mcse(x; kind=f, split_chains=1) != ROC.mcse(x; kind=f)
            x4 = randn(100, 4, 3, 2)
            se = ROC.mcse(x4; kind=f)
            for i in axes(x4, 4), j in axes(x4, 3)
                @test ROC.mcse(x4[:, :, :, i]; kind=f) ≈ se[:, i]
                @test ROC.mcse(x4[:, :, j, i]; kind=f) ≈ se[j, i]
            end
            xv = x4[:, 1, 1, 1]
            @test ROC.mcse(xv; kind=f) ≈ ROC.mcse(reshape(xv, :, 1); kind=f)
            @test isnan(ROC.mcse(ones(100); kind=f))
        end
        @test_throws ArgumentError ROC.mcse(x; kind=x -> mean(x))   # the SBM fallback stays with the reference (DESIGN.md 8)
        for sz in ((1000, 4), (1000, 4, 3)), f in (mean, median, std)
            xm = Array{Union{Missing,Float64}}(undef, sz...)
            xm .= randn.()
            xm[1] = missing
            se = ROC.mcse(xm; kind=f)
            ndims(xm) > 2 ? (@test ismissing(se[1]) && !any(ismissing, se[2:end])) : (@test ismissing(se))
        end
        if HAVE_OFFSET
            for T in (Float32, Float64), f in (mean, median, std, Base.Fix2(quantile, T(0.1))), ax in ((-5:94, 2:5, 11:15), (-5:94, 2:5, 11:15, 0:1))
                xo = randn(T, map(length, ax)...)
                se = ROC.mcse(OffsetArray(xo, ax...); kind=f)
                @test se isa OffsetArray{T,length(ax) - 2} && axes(se) == ax[3:end]
                @test ROC.mcse(xo; kind=f) ≈ collect(se)
            end
        end
        # the fused request is the two calls, bit for bit
        x16 = randn(Float32, 500, 16, 6)
        ids = repeat(1:4; inner=4)
        m, r = ROC.mcse_rhat_nested(x16, ids)
        @test m == ROC.mcse(x16) && r == ROC.rhat_nested(x16, ids)
    end

    @testset "storage order of MCMCChains read in place; bfmi known answers (test/bfmi.jl:2-41)" begin
        v = randn(300, 7, 4)                                # (iterations, parameters, chains)
        @test ROC.ess_rhat(ROC.samples_from_chains(v)) == ROC.ess_rhat(permutedims(v, (1, 3, 2)))
        energy = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
        @test ROC.bfmi(energy) ≈ 0.6
        @test ROC.bfmi([2.0, 3.0, 7.0, 8.0, 9.0, 10.0, 4.0, 1.0, 6.0, 5.0]) ≈ 0.2406937229
        e2 = hcat(energy, energy)
        @test ROC.bfmi(e2) ≈ [0.6, 0.6] && ROC.bfmi(permutedims(e2); dims=2) ≈ [0.6, 0.6]
    end

    @testset "bulk_tail: the two kinds from one pass are the two kinds (src/ess_rhat.jl:604-616)" begin
        x = randn(400, 4, 6)
        for split_chains in (1, 2, 3), am in (ROC.AutocovMethod(), ROC.FFTAutocovMethod())
            bt = ROC.bulk_tail(x; split_chains, autocov_method=am)
            b = ROC.ess_rhat(x; kind=:bulk, split_chains, autocov_method=am)
            t = ROC.ess_rhat(x; kind=:tail, split_chains, autocov_method=am)
            @test bt.rhat_bulk == b.rhat && bt.rhat_tail == t.rhat              # R-hat: the same bits on every path
            @test isapprox(bt.ess_bulk, b.ess; rtol=1e-9) && isapprox(bt.ess_tail, t.ess; rtol=1e-9)
            @test ROC.rhat(x; split_chains) == max.(bt.rhat_tail, bt.rhat_bulk)  # kind = :rank (src/ess_rhat.jl:416-420)
        end
        @test ROC.bulk_tail(rand(Float32, 100, 2, 3)).ess_bulk isa Vector{Float32}
        @test ROC.bulk_tail(randn(100, 4)).rhat_tail isa Float64
        @test_throws DomainError ROC.bulk_tail(x; maxlag=0)
    end
    @testset "requests from several tasks queue on the context" begin
        x = randn(400, 4, 8)
        want = ROC.ess_rhat(x)
        got = fetch.([Threads.@spawn ROC.ess_rhat(x) for _ in 1:8])
        @test all(==(want), got)
    end

    HAVE_REF && @testset "against MCMCDiagnosticTools.jl itself (the reference-output pin)" begin
        MDT = MCMCDiagnosticTools
        close6(a, b) = all(isapprox.(a, b; rtol=1e-6, nans=true))
        for (D, C, P) in ((1000, 4, 10), (501, 3, 4), (64, 8, 5)), T in (Float64, Float32)
            phi = T(0.7)
            x = randn(T, D, C, P)
            for t in 2:D
                x[t, :, 1:(P ÷ 2)] .= phi .* x[t - 1, :, 1:(P ÷ 2)] .+ sqrt(1 - phi^2) .* x[t, :, 1:(P ÷ 2)]
            end
            tol = T === Float64 ? 1e-6 : 1e-3   # Float32: the reference rounds every step, the device only the result (SURVEY.md 8c)
            cl(a, b) = all(isapprox.(a, b; rtol=tol, nans=true))
            for kind in KINDS, split_chains in (1, 2, 3)
                @test cl(ROC.rhat(x; kind, split_chains), MDT.rhat(x; kind, split_chains))
                for (mr, mm) in ((ROC.AutocovMethod(), MDT.AutocovMethod()), (ROC.FFTAutocovMethod(), MDT.FFTAutocovMethod()),
                                 (ROC.BDAAutocovMethod(), MDT.BDAAutocovMethod())), maxlag in (250, 10)
                    a = ROC.ess_rhat(x; kind, split_chains, maxlag, autocov_method=mr)
                    b = MDT.ess_rhat(x; kind, split_chains, maxlag, autocov_method=mm)
                    @test cl(a.ess, b.ess) && cl(a.rhat, b.rhat)
                end
            end
            for f in (mean, median, std, Base.Fix2(quantile, T(0.1)))
                @test cl(ROC.ess(x; kind=f), MDT.ess(x; kind=f))
                @test cl(ROC.mcse(x; kind=f), MDT.mcse(x; kind=f))
            end
            if C % 2 == 0
                ids = repeat(1:(C ÷ 2); inner=2)
                for kind in KINDS
                    @test cl(ROC.rhat_nested(x, ids; kind), MDT.rhat_nested(x, ids; kind))
                end
            end
            T === Float64 && @test close6(ROC.bfmi(x[:, :, 1]), MDT.bfmi(x[:, :, 1]))
        end
        # ties, signed zeros, infinities: the ranks are the contract (bit-exact), so R-hat must agree to rounding
        xt = Float64.(rand(0:9, 400, 4, 6))
        xt[1, 1, 1] = -0.0; xt[2, 1, 1] = 0.0; xt[3, 1, 2] = Inf; xt[4, 1, 2] = -Inf
        for kind in (:bulk, :tail, :rank)
            @test close6(ROC.rhat(xt; kind), MDT.rhat(xt; kind))
        end
        include(joinpath(@__DIR__, "..", "gen_reference_golden.jl"))   # writes tests/golden/ref_*.npz for tests/test_reference_pin.py
    end
    HAVE_REF || @info "MCMCDiagnosticTools.jl is not installed: the reference-output pin (part 2) was skipped"
end
