"""ctypes binding of oracle/libmdt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of oracle/mdt_oracle.c).  Arrays follow the reference's Julia convention:
logical shape (draws, chains, params), memory column-major (draws fastest).  numpy callers pass
arrays of shape (draws, chains, params) in ANY layout; they are copied to Fortran order here.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KINDS = {"basic": 0, "bulk": 1, "tail": 2, "rank": 3, "median": 4, "std": 5, "mad": 6, "quantile": 7}
AUTOCOV = {"direct": 0, "fft": 1, "bda": 2}


class OracleArgs(ctypes.Structure):
    _fields_ = [
        ("draws", ctypes.c_int64),
        ("chains", ctypes.c_int64),
        ("params", ctypes.c_int64),
        ("kind", ctypes.c_int32),
        ("autocov", ctypes.c_int32),
        ("split_chains", ctypes.c_int32),
        ("relative", ctypes.c_int32),
        ("maxlag", ctypes.c_int64),
        ("tail_prob", ctypes.c_double),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libmdt_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("mdt_oracle.c", "mdt_oracle_impl.inc")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libmdt_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.mdt_oracle_norminvcdf.restype = ctypes.c_double
        _LIB.mdt_oracle_norminvcdf.argtypes = [ctypes.c_double]
    return _LIB


def _sfx(dtype, faithful=False):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "_f64"
    if dtype == np.float32:
        return "_f32_faithful" if faithful else "_f32"
    raise TypeError(dtype)


def _as3d(x):
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None, None]
    elif x.ndim == 2:
        x = x[:, :, None]
    elif x.ndim > 3:
        x = x.reshape(x.shape[0], x.shape[1], -1, order="F")
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    return np.asfortranarray(x)


def _args(x, kind, autocov, split_chains, maxlag, relative, tail_prob):
    D, C, P = x.shape
    return OracleArgs(D, C, P, KINDS[kind], AUTOCOV[autocov], split_chains, int(relative), maxlag,
                      float(tail_prob))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleError(Exception):
    def __init__(self, code):
        super().__init__({1: "ArgumentError", 2: "DomainError", 3: "DimensionMismatch"}.get(code, str(code)))
        self.code = code


def ess_rhat(x, kind="rank", autocov="direct", split_chains=2, maxlag=250, relative=False,
             tail_prob=0.1, want_ess=True, want_rhat=True, faithful=False):
    x = _as3d(x)
    a = _args(x, kind, autocov, split_chains, maxlag, relative, tail_prob)
    P = x.shape[2]
    ess = np.empty(P, x.dtype) if want_ess else None
    rhat = np.empty(P, x.dtype) if want_rhat else None
    f = getattr(lib(), "mdt_oracle_ess_rhat" + _sfx(x.dtype, faithful))
    rc = f(ctypes.byref(a), _ptr(x), _ptr(ess) if want_ess else None, _ptr(rhat) if want_rhat else None)
    if rc:
        raise OracleError(rc)
    return ess, rhat


def bulk_tail(x, autocov="direct", split_chains=2, maxlag=250, relative=False, tail_prob=0.1,
              faithful=False):
    """returns (ess_bulk, ess_tail, rhat_bulk, rhat_tail)"""
    x = _as3d(x)
    a = _args(x, "bulk", autocov, split_chains, maxlag, relative, tail_prob)
    P = x.shape[2]
    outs = [np.empty(P, x.dtype) for _ in range(4)]
    f = getattr(lib(), "mdt_oracle_bulk_tail" + _sfx(x.dtype, faithful))
    rc = f(ctypes.byref(a), _ptr(x), *[_ptr(o) for o in outs])
    if rc:
        raise OracleError(rc)
    return tuple(outs)


def bulk_tail_all_cores(x, autocov="direct", split_chains=2, maxlag=250, tail_prob=0.1, nthreads=0, chunk=8):
    """bulk_tail over an OpenMP team: blocks of `chunk` parameters per thread (the all-core CPU baseline of bench.py).
    Float64 only.  Returns (ess_bulk, ess_tail, rhat_bulk, rhat_tail), the same numbers as bulk_tail."""
    x = _as3d(x)
    assert x.dtype == np.float64
    a = _args(x, "bulk", autocov, split_chains, maxlag, False, tail_prob)
    P = x.shape[2]
    outs = [np.empty(P, np.float64) for _ in range(4)]
    f = lib().mdt_oracle_bulk_tail_omp_f64
    rc = f(ctypes.byref(a), _ptr(x), *[_ptr(o) for o in outs], ctypes.c_int(int(nthreads)), ctypes.c_int64(int(chunk)))
    if rc:
        raise OracleError(rc)
    return tuple(outs)


def omp_threads():
    return int(lib().mdt_oracle_omp_threads())


def rhat_nested(x, chain_inds, kind="rank", split_chains=2, faithful=False):
    """chain_inds: (K, S) int array, 0-based; column k lists the chains of superchain k."""
    x = _as3d(x)
    a = _args(x, kind, "direct", split_chains, 250, False, 0.1)
    inds = np.asfortranarray(np.asarray(chain_inds, dtype=np.int32))
    K, S = inds.shape
    out = np.empty(x.shape[2], x.dtype)
    f = getattr(lib(), "mdt_oracle_rhat_nested" + _sfx(x.dtype, faithful))
    rc = f(ctypes.byref(a), _ptr(x), _ptr(inds), K, S, _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def mcse_mean(x, autocov="direct", split_chains=2, maxlag=250, faithful=False):
    x = _as3d(x)
    a = _args(x, "basic", autocov, split_chains, maxlag, False, 0.1)
    out = np.empty(x.shape[2], x.dtype)
    f = getattr(lib(), "mdt_oracle_mcse_mean" + _sfx(x.dtype, faithful))
    rc = f(ctypes.byref(a), _ptr(x), _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def ess_estimator(x, estimator, prob=0.5, autocov="direct", split_chains=2, maxlag=250, relative=False,
                  faithful=False):
    """ess(x; kind=estimator) for 'median' | 'std' | 'mad' | 'quantile' (with `prob`), src/ess_rhat.jl:291-297"""
    x = _as3d(x)
    a = _args(x, estimator, autocov, split_chains, maxlag, relative, prob)
    out = np.empty(x.shape[2], x.dtype)
    f = getattr(lib(), "mdt_oracle_ess_rhat" + _sfx(x.dtype, faithful))
    rc = f(ctypes.byref(a), _ptr(x), _ptr(out), None)
    if rc:
        raise OracleError(rc)
    return out


def mcse_std(x, autocov="direct", split_chains=2, maxlag=250, faithful=False):
    x = _as3d(x)
    a = _args(x, "basic", autocov, split_chains, maxlag, False, 0.1)
    out = np.empty(x.shape[2], x.dtype)
    f = getattr(lib(), "mdt_oracle_mcse_std" + _sfx(x.dtype, faithful))
    rc = f(ctypes.byref(a), _ptr(x), _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def mcse_quantile(x, estimator="median", prob=0.5, autocov="direct", split_chains=2, maxlag=250, faithful=False):
    """mcse(x; kind = median) / mcse(x; kind = Fix2(quantile, prob)), src/mcse.jl:66-118"""
    x = _as3d(x)
    a = _args(x, estimator, autocov, split_chains, maxlag, False, prob)
    out = np.empty(x.shape[2], x.dtype)
    f = getattr(lib(), "mdt_oracle_mcse_quantile" + _sfx(x.dtype, faithful))
    rc = f(ctypes.byref(a), _ptr(x), _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def bfmi(energy, faithful=False):
    """bfmi(energy) for a (draws,) vector or a (draws, chains) matrix, src/bfmi.jl:38-45"""
    e = np.asarray(energy)
    if e.dtype not in (np.float32, np.float64):
        e = e.astype(np.float64)
    vec = e.ndim == 1
    e2 = np.asfortranarray(e.reshape(e.shape[0], -1))
    out = np.empty(e2.shape[1], e2.dtype)
    f = getattr(lib(), "mdt_oracle_bfmi" + _sfx(e2.dtype, faithful))
    rc = f(_ptr(e2), ctypes.c_int64(e2.shape[0]), ctypes.c_int64(e2.shape[1]), _ptr(out))
    if rc:
        raise OracleError(rc)
    return out[0] if vec else out


def gelmandiag(x, alpha=0.05, faithful=False):
    """gelmandiag(samples; alpha) -> (psrf, psrfci), both Float64, src/gelmandiag.jl:1-60"""
    x = _as3d(x)
    P = x.shape[2]
    psrf, ci = np.empty(P), np.empty(P)
    f = getattr(lib(), "mdt_oracle_gelmandiag" + _sfx(x.dtype, faithful))
    rc = f(_ptr(x), ctypes.c_int64(x.shape[0]), ctypes.c_int64(x.shape[1]), ctypes.c_int64(P), ctypes.c_double(alpha),
           _ptr(psrf), _ptr(ci))
    if rc:
        raise OracleError(rc)
    return psrf, ci


# ---- stage functions (one parameter slice, pooled over draws and chains) ----
def tiedrank(v):
    v = np.ascontiguousarray(np.asarray(v).ravel(order="F"))
    out = np.empty(v.size, np.float64)
    getattr(lib(), "mdt_oracle_tiedrank" + _sfx(v.dtype))(_ptr(v), ctypes.c_int64(v.size), _ptr(out))
    return out


def rank_normalize(v):
    v = np.asarray(v)
    flat = np.ascontiguousarray(v.ravel(order="F"))
    out = np.empty_like(flat)
    getattr(lib(), "mdt_oracle_rank_normalize" + _sfx(flat.dtype))(_ptr(flat), ctypes.c_int64(flat.size), _ptr(out))
    return out.reshape(v.shape, order="F")


def fold_around_median(v):
    v = np.asarray(v)
    flat = np.ascontiguousarray(v.ravel(order="F"))
    out = np.empty_like(flat)
    getattr(lib(), "mdt_oracle_fold" + _sfx(flat.dtype))(_ptr(flat), ctypes.c_int64(flat.size), _ptr(out))
    return out.reshape(v.shape, order="F")


def quantile(v, p):
    flat = np.ascontiguousarray(np.asarray(v).ravel(order="F"))
    out = np.empty(1, flat.dtype)
    rc = getattr(lib(), "mdt_oracle_quantile" + _sfx(flat.dtype))(_ptr(flat), ctypes.c_int64(flat.size),
                                                               ctypes.c_double(p), _ptr(out))
    if rc:
        raise OracleError(1)
    return out[0]


def copyto_split(x2d, split):
    x2d = np.asfortranarray(np.asarray(x2d))
    if x2d.dtype not in (np.float32, np.float64):
        x2d = np.asfortranarray(x2d.astype(np.float64))
    D, C = x2d.shape
    out = np.empty((D // split, C * split), x2d.dtype, order="F")
    getattr(lib(), "mdt_oracle_copyto_split" + _sfx(x2d.dtype))(_ptr(x2d), ctypes.c_int64(D), ctypes.c_int64(C),
                                                                ctypes.c_int64(split), _ptr(out))
    return out


def mean_autocov(x2d, split=2, autocov="direct", nlags=None):
    x2d = np.asfortranarray(np.asarray(x2d))
    D, C = x2d.shape
    niter = D // split
    nlags = niter if nlags is None else nlags
    out = np.empty(nlags, x2d.dtype)
    rc = getattr(lib(), "mdt_oracle_mean_autocov" + _sfx(x2d.dtype))(
        _ptr(x2d), ctypes.c_int64(D), ctypes.c_int64(C), ctypes.c_int64(split), AUTOCOV[autocov],
        ctypes.c_int64(nlags), _ptr(out))
    if rc:
        raise OracleError(1)
    return out


def norminvcdf(p):
    return lib().mdt_oracle_norminvcdf(float(p))
