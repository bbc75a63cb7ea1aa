"""MI355X-native ESS / R-hat engine behind MCMCDiagnosticTools.jl's ess / rhat / ess_rhat /
rhat_nested / mcse(kind=mean) API (reference: src/MCMCDiagnosticTools.jl:17-25 exports).

The directory name contains a dot, so import it through `__graft_entry__.load_package()`
(module name `mcmcdiagnostictools_jl_amd`).
"""
from ._lib import ArgumentError, DimensionMismatch, DomainError, MdtRuntimeError  # noqa: F401
from .api import (  # noqa: F401
    AbstractAutocovMethod,
    AutocovMethod,
    BDAAutocovMethod,
    ErrorException,
    ESSRhat,
    FFTAutocovMethod,
    GelmanDiag,
    Quantile,
    bfmi,
    bulk_tail,
    ess,
    ess_rhat,
    from_mcmcchains_layout,
    gelmandiag,
    mcse,
    mcse_rhat_nested,
    rank_normalize,
    reserve,
    rhat,
    rhat_nested,
    status,
    FLAG_RANK_OVERFLOW,
    FLAG_NAN_QUANTILE,
    FLAG_FDIST,
)
from .dist import ess_rhat_sharded, shard_bounds  # noqa: F401
