"""Host-side mirror of the reference's operator interface for the hot path.

Same names, keyword meaning and error behaviour as MCMCDiagnosticTools.jl (the reference is Julia;
Julia is not in this image, so the tested host side is Python and the Julia shim in julia/ is the
thin ccall layer a maintainer would ship -- INTEGRATION.md):

    ess(samples; kind=:bulk, relative=false, autocov_method=AutocovMethod(), split_chains=2,
        maxlag=250, tail_prob=1//10)                      src/ess_rhat.jl:276-311
    rhat(samples; kind=:rank, split_chains=2)             src/ess_rhat.jl:335-420
    ess_rhat(samples; kind=:rank, kwargs...)              src/ess_rhat.jl:438-624
    rhat_nested(samples, superchain_ids; kind, split_chains)   src/rhat_nested.jl:43-125
    mcse(samples; kind=mean, kwargs...)                   src/mcse.jl:40-51

`samples` has shape (draws, [chains, [parameters...]]) -- the reference's array convention.  It may
be a torch tensor on the GPU (zero-copy when its memory is draws-fastest, i.e. it was built as
a (params, chains, draws) tensor and permuted), a CPU tensor or a numpy array (staged through the
library's device buffers).  All compute happens in libmdt_hip.so; nothing here computes a
diagnostic on the CPU.
"""
import ctypes
import statistics
import sys
import warnings
from collections import namedtuple

import numpy as np

from . import _lib
from ._lib import ArgumentError, DimensionMismatch

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class AbstractAutocovMethod:
    """src/ess_rhat.jl:2"""
    _code = None


class AutocovMethod(AbstractAutocovMethod):
    """src/ess_rhat.jl:38 -- direct lagged dot products (the default)."""
    _code = 0


class FFTAutocovMethod(AbstractAutocovMethod):
    """src/ess_rhat.jl:55"""
    _code = 1


class BDAAutocovMethod(AbstractAutocovMethod):
    """src/ess_rhat.jl:73 -- variogram estimator"""
    _code = 2


ESSRhat = namedtuple("ESSRhat", ["ess", "rhat"])  # NamedTuple{(:ess, :rhat)}

_KINDS = ("rank", "bulk", "tail", "basic")


def _norm_kind(kind):
    if isinstance(kind, str):
        return kind[1:] if kind.startswith(":") else kind
    return kind


def _is_mean(f):
    return f is np.mean or f is statistics.mean or f is statistics.fmean or (
        torch is not None and f is torch.mean) or (isinstance(f, str) and f == "mean")


class Quantile:
    """The estimator `Base.Fix2(Statistics.quantile, p)` of the reference (src/ess_rhat.jl:647): `ess(x, kind=Quantile(0.25))`."""

    def __init__(self, p):
        self.p = float(p)

    def __repr__(self):
        return f"Quantile({self.p})"


def _estimator(kind):
    """estimator -> (C-ABI kind name, probability) for the estimators with an expectand proxy, else None"""
    if isinstance(kind, Quantile):
        return "quantile", kind.p
    if isinstance(kind, tuple) and len(kind) == 2 and kind[0] in ("quantile", np.quantile):
        return "quantile", float(kind[1])
    if kind is np.median or kind is statistics.median or (torch is not None and kind is torch.median) or (
            isinstance(kind, str) and kind == "median"):
        return "median", 0.5
    if kind is np.std or kind is statistics.stdev or (torch is not None and kind is torch.std) or (
            isinstance(kind, str) and kind == "std"):
        return "std", 0.5
    if (isinstance(kind, str) and kind == "mad") or getattr(kind, "__name__", "") == "median_abs_deviation":
        return "mad", 0.5
    return None


def _is_shard_list(samples):
    """a list of torch tensors: the pre-sharded input of a multi-GPU context (block i of the params axis on GPU i)"""
    return isinstance(samples, (list, tuple)) and len(samples) > 0 and torch is not None and all(isinstance(t, torch.Tensor) for t in samples)


class _Prepared:
    """samples flattened to (draws, chains, params) with draws-fastest memory"""
    pass


def _prepare(samples, allow_strided=False):
    """allow_strided: a CUDA tensor whose draws are contiguous but whose chains / parameters are strided is handed
    over in place (chain_stride / param_stride of mdt_args) instead of being copied dense.  Only the entry points that
    go through the library's run() read those strides (a multi-GPU context included); everything else (bfmi,
    gelmandiag, the stage output) gets the dense copy.  A list of CUDA tensors is the pre-sharded input of a multi-GPU
    context (_prepare_shards)."""
    if _is_shard_list(samples):
        return _prepare_shards(samples)
    pr = _Prepared()
    mask = None
    # the reference's Union{Missing,T} eltype.  (numpy imports numpy.ma on first touch -- 40 - 55 ms, measured as most of a process's
    # first request -- and an array can only be a MaskedArray if somebody has imported it already)
    ma = sys.modules.get("numpy.ma")
    if ma is not None and isinstance(samples, ma.MaskedArray):
        mask = np.ma.getmaskarray(samples)
        samples = samples.filled(0.0)
    is_torch = torch is not None and isinstance(samples, torch.Tensor)
    if not is_torch:
        samples = np.asarray(samples)
        if samples.dtype not in (np.float32, np.float64):
            samples = samples.astype(np.float64)  # promote_type(eltype, typeof(zero/1))
    else:
        if samples.dtype not in (torch.float32, torch.float64):
            samples = samples.to(torch.float64 if not samples.dtype.is_floating_point else torch.float32)
    if samples.ndim < 1:
        raise ArgumentError("`samples` must have at least 1 dimension")
    shape = tuple(samples.shape)
    pr.orig_shape = shape
    pr.param_shape = shape[2:]
    pr.scalar = len(shape) < 3  # _maybescalar (src/utils.jl:214-215)
    D = shape[0]
    C = shape[1] if len(shape) > 1 else 1
    P = 1
    for d in pr.param_shape:
        P *= d
    pr.D, pr.C, pr.P = D, C, P
    x3 = samples.reshape(D, C, P)
    pr.cs = pr.ps = 0  # chain_stride / param_stride in elements; 0 = dense (draws, chains, params)
    if is_torch:
        st = x3.stride() if D * C * P > 0 else (1, D, D * C)
        dense = (st[0] == 1 or D == 1) and (st[1] == D or C == 1) and (st[2] == D * C or P == 1)
        if allow_strided and x3.is_cuda and not dense and (st[0] == 1 or D == 1) and (st[1] >= D or C == 1) and (st[2] >= 1 or P == 1):
            # draws contiguous, chains / parameters strided (e.g. MCMCChains' (iterations, parameters, chains)
            # storage viewed as (draws, chains, params)): read in place through the strides of mdt_args
            pr.cs = int(st[1]) if C > 1 else D
            pr.ps = int(st[2]) if P > 1 else max(1, (C - 1) * pr.cs + D)
            xt = x3
        else:
            xt = x3.permute(2, 1, 0).contiguous()  # no copy when memory is already draws-fastest
        pr.dev = xt.is_cuda
        pr.device_index = xt.device.index if xt.is_cuda else 0
        pr.buf = xt
        pr.ptr = xt.data_ptr()
        pr.f32 = xt.dtype == torch.float32
        pr.torch = True
        pr.like = samples
    else:
        xt = np.ascontiguousarray(np.transpose(x3, (2, 1, 0)))
        pr.dev = False
        pr.device_index = 0
        pr.buf = xt
        pr.ptr = xt.ctypes.data
        pr.f32 = xt.dtype == np.float32
        pr.torch = False
        pr.like = None
    pr.missing = None
    if mask is not None:
        pr.missing = mask.reshape(D, C, P).any(axis=(0, 1))
    return pr


def _describe(samples):
    """shape, eltype, location and layout of the request `samples` would make -- what _prepare finds, without copying
    anything (reserve(like=...))"""
    ma = sys.modules.get("numpy.ma")
    if ma is not None and isinstance(samples, ma.MaskedArray):
        samples = samples.data
    pr = _Prepared()
    is_torch = torch is not None and isinstance(samples, torch.Tensor)
    if not is_torch:
        samples = np.asarray(samples)
    if samples.ndim < 1:
        raise ArgumentError("`samples` must have at least 1 dimension")
    shape = tuple(samples.shape)
    D = shape[0]
    C = shape[1] if len(shape) > 1 else 1
    P = 1
    for d in shape[2:]:
        P *= d
    pr.D, pr.C, pr.P = D, C, P
    pr.cs = pr.ps = 0
    pr.shards = None
    if is_torch and samples.is_cuda:
        pr.dev, pr.device_index, pr.buf = True, samples.device.index, samples
        pr.f32 = samples.dtype == torch.float32 or (samples.dtype != torch.float64 and samples.dtype.is_floating_point)
        if D * C * P > 0:
            st = samples.reshape(D, C, P).stride() if samples.is_contiguous() or samples.dim() == 3 else None
            if st is not None:
                dense = (st[0] == 1 or D == 1) and (st[1] == D or C == 1) and (st[2] == D * C or P == 1)
                if not dense and (st[0] == 1 or D == 1) and (st[1] >= D or C == 1) and (st[2] >= 1 or P == 1):
                    pr.cs = int(st[1]) if C > 1 else D
                    pr.ps = int(st[2]) if P > 1 else max(1, (C - 1) * pr.cs + D)
    else:
        pr.dev, pr.device_index, pr.buf = False, 0, None
        dtn = samples.dtype
        pr.f32 = (dtn == torch.float32) if is_torch else (dtn == np.float32)
    return pr


def _prepare_shards(shards):
    """PRE-SHARDED input of a multi-GPU context (mdt_args.x_shards): a list of CUDA tensors, block i of the params axis on
    GPU devices[i], each of shape (draws, chains, p_i) with p_i = shard_bounds(sum p_i, i, len(shards)) -- no input
    moves between GPUs; the outputs are gathered on the first block's GPU."""
    prs = [_prepare(t, allow_strided=False) for t in shards]
    p0 = prs[0]
    if any(not q.dev for q in prs):
        raise ArgumentError("pre-sharded input must be CUDA tensors, block i on GPU devices[i]")
    if any((q.D, q.C, q.f32) != (p0.D, p0.C, p0.f32) for q in prs) or any(len(q.orig_shape) != 3 for q in prs):
        raise DimensionMismatch("every block of a pre-sharded array must be (draws, chains, params_i) with the same draws, chains and eltype")
    from .dist import shard_bounds
    P = sum(q.P for q in prs)
    for i, q in enumerate(prs):
        lo, hi = shard_bounds(P, i, len(prs))
        if q.P != hi - lo:
            raise DimensionMismatch(f"block {i} holds {q.P} parameters; the partition rule (shard_bounds) gives it {hi - lo} of {P}")
    pr = _Prepared()
    pr.__dict__.update(p0.__dict__)
    pr.P, pr.param_shape, pr.orig_shape, pr.scalar = P, (P,), (p0.D, p0.C, P), False
    pr.ptr, pr.shards, pr.missing = None, prs, None
    pr.xsh = (ctypes.c_void_p * len(prs))(*[q.ptr for q in prs])   # (kept alive with pr for the duration of the call)
    # mdt_args has ONE stream slot: the producers of the other blocks run on streams of other GPUs.  The library
    # synchronises every device before it reads its block (include/mdt_hip.h); the streams torch is using are waited for
    # here as well, so the contract does not rest on one side only.
    for q in prs:
        torch.cuda.current_stream(q.buf.device).synchronize()
    return pr


def _new_out(pr, n=None, dtype=None):
    n = pr.P if n is None else n
    if pr.torch:
        dt = dtype or (torch.float32 if pr.f32 else torch.float64)
        return torch.empty(n, dtype=dt, device=pr.buf.device)
    dt = dtype or (np.float32 if pr.f32 else np.float64)
    return np.empty(n, dtype=dt)


def _optr(o):
    if o is None:
        return None
    return o.data_ptr() if (torch is not None and isinstance(o, torch.Tensor)) else o.ctypes.data


def _args(pr, kind, outputs, autocov, split_chains, maxlag, relative, tail_prob, sync=True):
    stream = None
    if pr.dev:
        stream = torch.cuda.current_stream(pr.buf.device).cuda_stream
    a = _lib.MdtArgs(pr.D, pr.C, pr.P, _lib.MDT_F32 if pr.f32 else _lib.MDT_F64,
                     _lib.MDT_DEVICE if pr.dev else _lib.MDT_HOST, kind, outputs, autocov, int(split_chains),
                     1 if relative else 0, 1 if sync else 0, int(maxlag), float(tail_prob), stream,
                     getattr(pr, "cs", 0), getattr(pr, "ps", 0))
    if getattr(pr, "shards", None) is not None:
        a.x_shards = ctypes.cast(pr.xsh, ctypes.POINTER(ctypes.c_void_p))
    return a


def _ctx_key(pr, devices=None):
    """(device, stream) of the context a request runs on: concurrent streams never share scratch.  `devices`: a sequence
    of GPU ordinals -> the multi-GPU context over them (params axis cut into contiguous blocks, mdt_ctx_create_multi)"""
    if getattr(pr, "shards", None) is not None:
        if devices is None:
            devices = [q.device_index for q in pr.shards]
        if [int(d) for d in devices] != [q.device_index for q in pr.shards]:
            raise ArgumentError("block i of a pre-sharded array must live on devices[i]")
    if devices is not None:
        return tuple(int(d) for d in devices), None
    stream = torch.cuda.current_stream(pr.buf.device).cuda_stream if pr.dev else 0
    return (pr.device_index if pr.dev else 0), stream


def _call(key, fn, *args):
    """one library call on the context of `key`: held against eviction and under its lock (an mdt_ctx is single-owner)
    for the whole call, status -> exception while the context (and its error message) is still held"""
    with _lib.using(*key) as h:
        _lib.check(fn(h, *args), h)


def status(device=None, *, devices=None, clear=True):
    """Sticky condition flags of the context of (device, current stream) -- or of the multi-GPU context over
    `devices` -- after waiting for its queued work (mdt_ctx_status): what requests issued with sync=False could not
    raise.  Bits: FLAG_RANK_OVERFLOW, FLAG_NAN_QUANTILE (the reference throws ArgumentError there), FLAG_FDIST."""
    if devices is not None:
        return _lib.status((tuple(int(d) for d in devices), None), clear)
    dev = torch.cuda.current_device() if device is None else int(device)
    return _lib.status((dev, torch.cuda.current_stream(dev).cuda_stream), clear)


FLAG_RANK_OVERFLOW, FLAG_NAN_QUANTILE, FLAG_FDIST = _lib.FLAG_RANK_OVERFLOW, _lib.FLAG_NAN_QUANTILE, _lib.FLAG_FDIST


def reserve(draws, chains, params, *, dtype=np.float64, device=None, devices=None, like=None, split_chains=2):
    """Allocate and build, before the first request, what a (draws, chains, params) array of `dtype` needs: scratch, the
    normal-score table of the shape, ticket tables (mdt_ctx_reserve).  Optional: a diagnostic is usually called once per
    array, and without this its one call also pays for the allocations.  No counterpart in the reference (its caches are
    per-call temporaries, src/ess_rhat.jl:511).

    Which context is warmed -- the same rule as a request (`_ctx_key`): with `like=` (the array the request will be made on,
    or any array of its kind) exactly the context that request resolves to, shape / dtype / layout taken from it; a host
    (numpy) array lives on context (device 0, stream 0) and also gets its staging buffers.  Without `like` a DEVICE request
    is assumed: the context of (`device` or the current device, that device's current stream) -- reserve under the stream
    the request will run on --, or the multi-GPU context over `devices`."""
    a_like = None
    if like is not None:
        pr = _describe(like)
        draws, chains, params = pr.D, pr.C, pr.P
        dt = np.dtype(np.float32 if pr.f32 else np.float64)
        key = _ctx_key(pr, devices)
        a_like = _args(pr, 0, 0, 0, split_chains, 250, False, 0.1)
    else:
        dt = np.dtype(dtype)
    if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
        raise ArgumentError("dtype must be float64 or float32")
    if a_like is not None:
        a = a_like
        a.x_shards = None
    else:
        if devices is not None:
            key, stream = (tuple(int(d) for d in devices), None), None
        else:
            dev = torch.cuda.current_device() if device is None else int(device)
            stream = torch.cuda.current_stream(dev).cuda_stream
            key = (dev, stream)
        a = _lib.MdtArgs(int(draws), int(chains), int(params), _lib.MDT_F32 if dt == np.dtype(np.float32) else _lib.MDT_F64,
                         _lib.MDT_DEVICE, 0, 0, 0, int(split_chains), 0, 1, 250, 0.1, stream, 0, 0)
    _call(key, _lib.load().mdt_ctx_reserve, ctypes.byref(a))


def _finish(pr, out):
    """param axes, Missing and scalar conventions of the reference (src/utils.jl:197-215)"""
    if out is None:
        return None
    if pr.missing is not None:
        arr = out.cpu().numpy() if pr.torch else out
        arr = np.ma.MaskedArray(arr.reshape(pr.param_shape), mask=pr.missing.reshape(pr.param_shape))
        return arr if not pr.scalar else (np.ma.masked if pr.missing.any() else float(arr))
    if pr.scalar:
        return float(out[0])
    return out.reshape(pr.param_shape)


def _autocov_code(m):
    if isinstance(m, AbstractAutocovMethod) and m._code is not None:
        return m._code
    if isinstance(m, type) and issubclass(m, AbstractAutocovMethod) and m._code is not None:
        return m._code
    raise ArgumentError(
        "autocov_method must be AutocovMethod(), FFTAutocovMethod() or BDAAutocovMethod(); "
        "user-defined AbstractAutocovMethod plugins run on the reference's CPU path only")


def _check_int(name, v):
    if isinstance(v, bool) or not isinstance(v, (int, np.integer)):
        raise TypeError(f"`{name}` must be an integer")  # Julia: MethodError on ::Int keyword
    return int(v)


def _warn_niter(pr, split_chains):
    niter = pr.D // split_chains
    if not niter > 4:  # src/ess_rhat.jl:472-479
        warnings.warn(f"number of draws after splitting must be >4 but is {niter}. ESS cannot be computed.")


def _call_ess_rhat(samples, kind, want_ess, want_rhat, relative=False, autocov_method=None, split_chains=2,
                   maxlag=250, tail_prob=0.1, sync=True, devices=None):
    split_chains = _check_int("split_chains", split_chains)
    maxlag = _check_int("maxlag", maxlag)
    if split_chains < 1:
        raise ArgumentError("split_chains must be >= 1")
    autocov = _autocov_code(AutocovMethod() if autocov_method is None else autocov_method)
    pr = _prepare(samples, allow_strided=True)
    if want_ess:
        _warn_niter(pr, split_chains)
    ess_o = _new_out(pr) if want_ess else None
    rhat_o = _new_out(pr) if want_rhat else None
    if pr.P > 0:
        h = _ctx_key(pr, devices)
        a = _args(pr, _lib.KIND[kind], (1 if want_ess else 0) | (2 if want_rhat else 0), autocov, split_chains,
                  maxlag, relative, tail_prob, sync)
        _call(h, _lib.load().mdt_ess_rhat, ctypes.byref(a), pr.ptr, _optr(ess_o), _optr(rhat_o))
    return pr, ess_o, rhat_o


def ess(samples, *, kind="bulk", **kwargs):
    """Effective sample size (src/ess_rhat.jl:276-311).  `kind` is "bulk", "tail", "basic" or an estimator
    with an expectand proxy (src/ess_rhat.jl:626-659): mean (`numpy.mean`, "mean"), median (`numpy.median`,
    "median"), std (`numpy.std`, "std"), mad ("mad", `scipy.stats.median_abs_deviation`) or `Quantile(p)`."""
    kind = _norm_kind(kind)
    if isinstance(kind, str) and kind in ("bulk", "tail", "basic"):
        pr, e, _ = _call_ess_rhat(samples, kind, True, False, **kwargs)
        return _finish(pr, e)
    if _is_mean(kind):
        kwargs.pop("tail_prob", None)
        pr, e, _ = _call_ess_rhat(samples, "basic", True, False, **kwargs)
        return _finish(pr, e)
    est = _estimator(kind)
    if est is not None:
        kwargs.pop("tail_prob", None)
        pr, e, _ = _call_ess_rhat(samples, est[0], True, False, tail_prob=est[1], **kwargs)
        return _finish(pr, e)
    if isinstance(kind, str):
        raise ArgumentError(f"the `kind` `{kind}` is not supported by `ess`")
    raise ArgumentError(f"the estimator {kind} is not yet supported by `ess`")


def rhat(samples, *, kind="rank", split_chains=2, devices=None):
    """R-hat (src/ess_rhat.jl:335-420)."""
    kind = _norm_kind(kind)
    if kind not in _KINDS:
        raise ArgumentError(f"the `kind` `{kind}` is not supported by `rhat`")
    pr, _, r = _call_ess_rhat(samples, kind, False, True, split_chains=split_chains, devices=devices)
    return _finish(pr, r)


def ess_rhat(samples, *, kind="rank", **kwargs):
    """(ess=..., rhat=...) (src/ess_rhat.jl:438-624)."""
    kind = _norm_kind(kind)
    if kind not in _KINDS:
        raise ArgumentError(f"the `kind` `{kind}` is not supported by `ess_rhat`")
    pr, e, r = _call_ess_rhat(samples, kind, True, True, **kwargs)
    return ESSRhat(_finish(pr, e), _finish(pr, r))


def _check_out(pr, out, n=4):
    """caller-supplied output buffers of bulk_tail: the library writes pr.P elements of the input's dtype through the
    raw pointers, so count, shape, dtype, contiguity and placement are checked here"""
    if len(out) != n:
        raise ArgumentError(f"`out` must hold {n} buffers, got {len(out)}")
    for o in out:
        if pr.torch:
            ok = isinstance(o, torch.Tensor) and o.dim() == 1 and o.is_contiguous() and o.numel() == pr.P and \
                o.dtype == (torch.float32 if pr.f32 else torch.float64) and o.device == pr.buf.device
        else:
            ok = isinstance(o, np.ndarray) and o.ndim == 1 and o.flags.c_contiguous and o.size == pr.P and \
                o.dtype == (np.float32 if pr.f32 else np.float64)
        if not ok:
            raise DimensionMismatch(f"every `out` buffer must be a contiguous 1-d array of {pr.P} "
                                    f"{'float32' if pr.f32 else 'float64'} values in the same memory (host / GPU) as `samples`")


def bulk_tail(samples, *, relative=False, autocov_method=None, split_chains=2, maxlag=250, tail_prob=0.1,
              sync=True, out=None, devices=None):
    """The BASELINE metric in one pass: ess_rhat(kind=:bulk) and ess_rhat(kind=:tail) fused.
    Returns (ess_bulk, ess_tail, rhat_bulk, rhat_tail).  `out`: four 1-d buffers to write into (validated);
    `devices`: run on the multi-GPU context over these ordinals (params axis in contiguous blocks)."""
    split_chains = _check_int("split_chains", split_chains)
    maxlag = _check_int("maxlag", maxlag)
    autocov = _autocov_code(AutocovMethod() if autocov_method is None else autocov_method)
    pr = _prepare(samples, allow_strided=True)
    _warn_niter(pr, split_chains)
    if out is not None:
        _check_out(pr, out)
    outs = out if out is not None else [_new_out(pr) for _ in range(4)]
    if pr.P > 0:
        h = _ctx_key(pr, devices)
        a = _args(pr, 0, 3, autocov, split_chains, maxlag, relative, tail_prob, sync)
        _call(h, _lib.load().mdt_bulk_tail, ctypes.byref(a), pr.ptr, *[_optr(o) for o in outs])
    if out is not None:
        return out
    return tuple(_finish(pr, o) for o in outs)


def _validate_superchain_ids(superchain_ids, nchains):
    """src/rhat_nested.jl:68-81 + unique_indices (src/utils.jl:50-64): K x S, 0-based"""
    ids = list(superchain_ids)
    if len(ids) != nchains:
        raise DimensionMismatch(f"`superchain_ids` has length {len(ids)} but `samples` has {nchains} chains")
    groups = {}
    for i, v in enumerate(ids):
        groups.setdefault(v, []).append(i)
    keys = sorted(groups)  # SortedDict order
    if len(keys) < 2:
        raise ArgumentError(f"at least 2 superchains are required, got {len(keys)}")
    sizes = {len(groups[k]) for k in keys}
    if len(sizes) != 1:
        raise ArgumentError("all superchains must contain the same number of chains")
    return np.asfortranarray(np.array([groups[k] for k in keys], dtype=np.int32).T)


def rhat_nested(samples, superchain_ids, *, kind="rank", split_chains=2, devices=None):
    """Nested R-hat (src/rhat_nested.jl:43-188)."""
    first = samples[0] if _is_shard_list(samples) else samples
    nd = first.ndim if hasattr(first, "ndim") else np.asarray(first).ndim
    if nd < 2:
        raise ArgumentError("`samples` must have at least 2 dimensions `(draws, chains[, parameters…])`")
    split_chains = _check_int("split_chains", split_chains)
    inds = _validate_superchain_ids(superchain_ids, first.shape[1])
    kind = _norm_kind(kind)
    if kind not in _KINDS:
        raise ArgumentError(f"the `kind` `{kind}` is not supported by `rhat_nested`")
    pr = _prepare(samples, allow_strided=True)
    out = _new_out(pr)
    if pr.P > 0:
        h = _ctx_key(pr, devices)
        a = _args(pr, _lib.KIND[kind], 2, 0, split_chains, 250, False, 0.1)
        K, S = inds.shape
        _call(h, _lib.load().mdt_rhat_nested, ctypes.byref(a), pr.ptr, inds.ctypes.data, K, S, _optr(out))
    return _finish(pr, out)


def mcse_rhat_nested(samples, superchain_ids, *, kind="rank", split_chains=2, maxlag=250, autocov_method=None,
                     devices=None):
    """(mcse(samples; kind=mean), rhat_nested(samples, superchain_ids; kind)) from ONE launch (mdt_mcse_nested): every
    parameter's slice is read from HBM once for both (BASELINE config 5).  Bit-identical to the two separate calls."""
    first = samples[0] if _is_shard_list(samples) else samples
    nd = first.ndim if hasattr(first, "ndim") else np.asarray(first).ndim
    if nd < 2:
        raise ArgumentError("`samples` must have at least 2 dimensions `(draws, chains[, parameters…])`")
    split_chains = _check_int("split_chains", split_chains)
    maxlag = _check_int("maxlag", maxlag)
    autocov = _autocov_code(AutocovMethod() if autocov_method is None else autocov_method)
    inds = _validate_superchain_ids(superchain_ids, first.shape[1])
    kind = _norm_kind(kind)
    if kind not in _KINDS:
        raise ArgumentError(f"the `kind` `{kind}` is not supported by `rhat_nested`")
    pr = _prepare(samples, allow_strided=True)
    _warn_niter(pr, split_chains)
    m_out, r_out = _new_out(pr), _new_out(pr)
    if pr.P > 0:
        h = _ctx_key(pr, devices)
        a = _args(pr, _lib.KIND[kind], 3, autocov, split_chains, maxlag, False, 0.1)
        K, S = inds.shape
        _call(h, _lib.load().mdt_mcse_nested, ctypes.byref(a), pr.ptr, inds.ctypes.data, K, S, _optr(m_out), _optr(r_out))
    return _finish(pr, m_out), _finish(pr, r_out)


def mcse(samples, *, kind=np.mean, **kwargs):
    """Monte Carlo standard error for kind = mean (src/mcse.jl:45-51), std (:52-65), median or Quantile(p)
    (:66-118: Beta quadrature points + two order statistics); the SBM fallback for arbitrary estimators is
    not on the device path."""
    est = None if _is_mean(kind) else _estimator(kind)
    if not _is_mean(kind) and (est is None or est[0] == "mad"):
        raise NotImplementedError("mcse: kind must be mean, std, median or Quantile(p) on the device path (SURVEY.md §8f)")
    split_chains = _check_int("split_chains", kwargs.pop("split_chains", 2))
    maxlag = _check_int("maxlag", kwargs.pop("maxlag", 250))
    autocov = _autocov_code(kwargs.pop("autocov_method", AutocovMethod()))
    relative = kwargs.pop("relative", False)
    devices = kwargs.pop("devices", None)
    if kwargs:
        raise TypeError(f"unsupported keyword arguments {sorted(kwargs)}")
    pr = _prepare(samples, allow_strided=True)
    _warn_niter(pr, split_chains)
    out = _new_out(pr)
    if pr.P > 0:
        h = _ctx_key(pr, devices)
        lib = _lib.load()
        if est is None or est[0] == "std":
            a = _args(pr, 0, 1, autocov, split_chains, maxlag, relative, 0.1)
            fn = lib.mdt_mcse_mean if est is None else lib.mdt_mcse_std
        else:
            a = _args(pr, _lib.KIND[est[0]], 1, autocov, split_chains, maxlag, relative, est[1])
            fn = lib.mdt_mcse_quantile
        _call(h, fn, ctypes.byref(a), pr.ptr, _optr(out))
    return _finish(pr, out)


def rank_normalize(samples, *, fold=False, want_ranks=True):
    """Stage output: (tied ranks as float64, normal scores) of `_rank_normalize(samples)` or of
    `_rank_normalize(_fold_around_median(samples))` (src/utils.jl:148-184); shape of `samples`."""
    pr = _prepare(samples)
    n = pr.D * pr.C * pr.P
    ranks = _new_out(pr, n, (torch.float64 if pr.torch else np.float64)) if want_ranks else None
    z = _new_out(pr, n)
    if pr.P > 0:
        h = _ctx_key(pr)
        a = _args(pr, 0, 0, 0, 1, 250, False, 0.1)
        _call(h, _lib.load().mdt_rank_normalize, ctypes.byref(a), pr.ptr, 1 if fold else 0, _optr(ranks), _optr(z))

    def back(o):
        if o is None:
            return None
        o3 = o.reshape(pr.P, pr.C, pr.D)
        o3 = o3.permute(2, 1, 0) if pr.torch else np.transpose(o3, (2, 1, 0))
        return o3.reshape(pr.orig_shape)
    return back(ranks), back(z)


def from_mcmcchains_layout(samples):
    """(iterations, parameters, chains) device tensor, as MCMCChains stores it -> the (draws, chains, params)
    view every function here takes, through ONE coalesced device pass (mdt_layout_from_dpc) instead of a
    host-side permutedims.  The result is a zero-copy view of a (params, chains, draws)-contiguous buffer."""
    if torch is None or not isinstance(samples, torch.Tensor) or not samples.is_cuda:
        raise ArgumentError("from_mcmcchains_layout takes a CUDA tensor of shape (iterations, parameters, chains)")
    if samples.dim() != 3 or samples.dtype not in (torch.float32, torch.float64):
        raise ArgumentError("expected a 3-d float32/float64 tensor")
    D, P, C = samples.shape
    src = samples.permute(2, 1, 0).contiguous()  # memory = Julia's column-major (D, P, C): no-op if it already is
    out = torch.empty((P, C, D), dtype=samples.dtype, device=samples.device)
    pr = _Prepared()
    pr.D, pr.C, pr.P = D, C, P
    pr.f32 = samples.dtype == torch.float32
    pr.dev, pr.buf = True, out
    pr.device_index = samples.device.index or 0
    h = _ctx_key(pr)
    a = _args(pr, 0, 0, 0, 1, 250, False, 0.1)
    _call(h, _lib.load().mdt_layout_from_dpc, ctypes.byref(a), ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(out.data_ptr()))
    return out.permute(2, 1, 0)


class ErrorException(RuntimeError):
    """Julia's ErrorException (`error("...")`), src/gelmandiag.jl:3"""


GelmanDiag = namedtuple("GelmanDiag", ["psrf", "psrfci"])


def bfmi(energy, *, dims=1):
    """Estimated Bayesian fraction of missing information (src/bfmi.jl:38-45): per chain,
    mean(abs2, diff(energy)) / var(energy).  `energy` is a vector of draws (-> a scalar) or a matrix;
    `dims` is the axis holding the draws in the reference's 1-based numbering (default 1: (draws, chains))."""
    is_torch = torch is not None and isinstance(energy, torch.Tensor)
    e = energy if is_torch else np.asarray(energy)
    if e.ndim == 1:
        e2 = e.reshape(-1, 1)
    elif e.ndim == 2:
        if dims not in (1, 2):
            raise ArgumentError("dims must be 1 or 2 for a matrix of energies")
        e2 = e if dims == 1 else (e.t() if is_torch else e.T)
    else:
        raise ArgumentError("`energy` must be a vector or a matrix")  # MethodError in the reference
    pr = _prepare(e2)
    out = _new_out(pr, pr.C)
    if pr.C > 0:
        h = _ctx_key(pr)
        a = _args(pr, 0, 0, 0, 1, 250, False, 0.1)
        _call(h, _lib.load().mdt_bfmi, ctypes.byref(a), pr.ptr, _optr(out))
    return float(out[0]) if e.ndim == 1 else out


def gelmandiag(samples, *, alpha=0.05):
    """Gelman, Rubin and Brooks diagnostic (src/gelmandiag.jl:76-80): (psrf, psrfci) per parameter for
    `samples` of shape (draws, chains, parameters); both Float64 (:41-42).  The multivariate variant is not
    provided."""
    if getattr(samples, "ndim", np.ndim(samples)) != 3:
        raise ArgumentError("`samples` must have shape (draws, chains, parameters)")  # MethodError in the reference
    pr = _prepare(samples)
    if pr.C < 2:
        raise ErrorException("Gelman diagnostic requires at least 2 chains")
    f64 = torch.float64 if pr.torch else np.float64
    psrf, ci = _new_out(pr, pr.P, f64), _new_out(pr, pr.P, f64)
    if pr.P > 0:
        h = _ctx_key(pr)
        a = _args(pr, 0, 0, 0, 1, 250, False, 0.1)
        _call(h, _lib.load().mdt_gelmandiag, ctypes.byref(a), pr.ptr, float(alpha), _optr(psrf), _optr(ci))
    return GelmanDiag(psrf, ci)
