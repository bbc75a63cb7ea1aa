"""ctypes binding of csrc/libmdt_hip.so (the C-ABI declared in include/mdt_hip.h).

There is no CPU fallback: if the shared library is missing or no GPU is visible, every compute
entry point raises.  Build with `python -c "import __graft_entry__ as g; g.build()"`.
"""
import contextlib
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# MDT_LIB: another build of the same library (A/B timing, diagnostics builds); never a different implementation
LIB_PATH = os.environ.get("MDT_LIB") or os.path.join(_HERE, "csrc", "libmdt_hip.so")

MDT_OK, MDT_EARG, MDT_EDOMAIN, MDT_EDIM, MDT_ERUNTIME, MDT_ENOMEM, MDT_EUNSUPPORTED = range(7)
MDT_F64, MDT_F32 = 0, 1
MDT_HOST, MDT_DEVICE = 0, 1
KIND = {"basic": 0, "bulk": 1, "tail": 2, "rank": 3, "median": 4, "std": 5, "mad": 6, "quantile": 7}
OUT_ESS, OUT_RHAT = 1, 2

# every symbol include/mdt_hip.h declares
SYMBOLS = ("mdt_version", "mdt_ctx_create", "mdt_ctx_reserve", "mdt_ctx_create_multi", "mdt_shard_bounds", "mdt_ctx_destroy", "mdt_ctx_status", "mdt_last_error", "mdt_ess_rhat",
           "mdt_bulk_tail", "mdt_rhat_nested", "mdt_mcse_nested", "mdt_mcse_mean", "mdt_mcse_std", "mdt_mcse_quantile", "mdt_rank_normalize", "mdt_layout_from_dpc",
           "mdt_bfmi", "mdt_gelmandiag", "mdt_debug_phase_cycles", "mdt_debug_beta")


class MdtArgs(ctypes.Structure):
    _fields_ = [
        ("draws", ctypes.c_int64),
        ("chains", ctypes.c_int64),
        ("params", ctypes.c_int64),
        ("dtype", ctypes.c_int32),
        ("location", ctypes.c_int32),
        ("kind", ctypes.c_int32),
        ("outputs", ctypes.c_int32),
        ("autocov", ctypes.c_int32),
        ("split_chains", ctypes.c_int32),
        ("relative", ctypes.c_int32),
        ("sync", ctypes.c_int32),
        ("maxlag", ctypes.c_int64),
        ("tail_prob", ctypes.c_double),
        ("stream", ctypes.c_void_p),
        ("chain_stride", ctypes.c_int64),
        ("param_stride", ctypes.c_int64),
        ("x_shards", ctypes.POINTER(ctypes.c_void_p)),
        ("tail_p_lo", ctypes.c_double),
        ("tail_p_hi", ctypes.c_double),
    ]


class ArgumentError(ValueError):
    """Julia's ArgumentError (MDT_EARG)."""


class DomainError(ValueError):
    """Julia's DomainError (MDT_EDOMAIN), e.g. maxlag <= 0 (src/ess_rhat.jl:481)."""


class DimensionMismatch(ValueError):
    """Julia's DimensionMismatch (MDT_EDIM)."""


class MdtRuntimeError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen the HIP library; loud failure when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MdtRuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run __graft_entry__.build()); there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        ap = ctypes.POINTER(MdtArgs)
        lib.mdt_version.restype = ctypes.c_int
        lib.mdt_ctx_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
        lib.mdt_ctx_create_multi.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        lib.mdt_ctx_status.argtypes = [vp, ctypes.POINTER(i32), i32]
        lib.mdt_ctx_reserve.argtypes = [vp, ap]
        lib.mdt_shard_bounds.argtypes = [i64, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
        lib.mdt_ctx_destroy.argtypes = [vp]
        lib.mdt_last_error.argtypes = [vp]
        lib.mdt_last_error.restype = ctypes.c_char_p
        lib.mdt_ess_rhat.argtypes = [vp, ap, vp, vp, vp]
        lib.mdt_bulk_tail.argtypes = [vp, ap, vp, vp, vp, vp, vp]
        lib.mdt_rhat_nested.argtypes = [vp, ap, vp, vp, i32, i32, vp]
        lib.mdt_mcse_nested.argtypes = [vp, ap, vp, vp, i32, i32, vp, vp]
        lib.mdt_mcse_mean.argtypes = [vp, ap, vp, vp]
        lib.mdt_mcse_std.argtypes = [vp, ap, vp, vp]
        lib.mdt_mcse_quantile.argtypes = [vp, ap, vp, vp]
        lib.mdt_layout_from_dpc.argtypes = [vp, ap, vp, vp]
        lib.mdt_bfmi.argtypes = [vp, ap, vp, vp]
        lib.mdt_gelmandiag.argtypes = [vp, ap, vp, ctypes.c_double, vp, vp]
        lib.mdt_rank_normalize.argtypes = [vp, ap, vp, i32, vp, vp]
        lib.mdt_debug_phase_cycles.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), i32]
        lib.mdt_debug_beta.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, i32, ctypes.POINTER(ctypes.c_double)]
        _ = i64
        _lib = lib
    return _lib


class _Entry:
    """a registered context: handle, its lock (an mdt_ctx is single-owner) and the number of calls that hold it"""
    __slots__ = ("h", "lock", "pins")

    def __init__(self, h):
        self.h, self.lock, self.pins = h, threading.Lock(), 0


_ctxs = {}
_ctxs_lock = threading.Lock()
MAX_CTX_PER_DEVICE = 4  # every context owns ~0.5 GB of per-workgroup scratch at the config-2 slice size
FLAG_RANK_OVERFLOW, FLAG_NAN_QUANTILE, FLAG_FDIST = 1, 2, 4


def _key(device, stream):
    multi = isinstance(device, (tuple, list))
    return (tuple(int(d) for d in device), 0) if multi else (int(device), int(stream or 0))


def _entry(key):
    """the registered context of `key`, created on first use.  Caller holds _ctxs_lock.
    At most MAX_CTX_PER_DEVICE live per GPU; beyond that one that nobody holds (pins == 0), that is not the default
    stream's and that has no unread sticky flags (what sync=False requests could not report) is destroyed first --
    mdt_ctx_status waits for its queued work, mdt_ctx_destroy synchronises the device.  When every candidate is held
    or has flags to report, nothing is evicted and the cap is exceeded."""
    lib = load()
    e = _ctxs.get(key)
    if e is not None:
        return e
    h = ctypes.c_void_p()
    if isinstance(key[0], tuple):
        ids = (ctypes.c_int * len(key[0]))(*key[0])
        rc = lib.mdt_ctx_create_multi(ctypes.byref(h), ids, len(key[0]))
    else:
        mine = [k for k in _ctxs if k[0] == key[0]]
        if len(mine) >= MAX_CTX_PER_DEVICE:
            for victim in mine:
                ve = _ctxs[victim]
                if victim[1] == 0 or ve.pins != 0:
                    continue
                f = ctypes.c_int32(0)
                with ve.lock:  # (pins == 0 under _ctxs_lock: nobody is inside the library with it, nobody can get it)
                    rcs = lib.mdt_ctx_status(ve.h, ctypes.byref(f), 0)
                    if rcs == MDT_OK and f.value == 0:
                        del _ctxs[victim]
                        lib.mdt_ctx_destroy(ve.h)
                        break
        rc = lib.mdt_ctx_create(ctypes.byref(h), key[0])
    if rc != MDT_OK:
        raise MdtRuntimeError(f"mdt_ctx_create(device={key[0]}) failed with status {rc} "
                              "(no MI355X visible? there is no CPU fallback)")
    e = _Entry(h)
    _ctxs[key] = e
    return e


@contextlib.contextmanager
def using(device, stream=None):
    """`with using(device, stream) as h:` -- the context of (GPU, stream) (or of a tuple of GPUs: one multi-GPU context
    per tuple), held for the duration of the block: it cannot be evicted while held, and the block runs under the
    context's lock (ctypes releases the GIL during a call, so two Python threads on the same (device, stream) must not
    enter the library together)."""
    key = _key(device, stream)
    with _ctxs_lock:
        e = _entry(key)
        e.pins += 1
    try:
        with e.lock:
            yield e.h
    finally:
        with _ctxs_lock:
            e.pins -= 1


def ctx(device, stream=None):
    """The raw handle of the context of (GPU, stream) -- diagnostics and tests; calls go through `using`, which also
    protects the handle against eviction.  One mdt_ctx per (GPU, stream) per process: a context owns the scratch its
    kernels work in, so requests that may run concurrently -- different streams -- must not share one."""
    with _ctxs_lock:
        return _entry(_key(device, stream)).h


def lock(h):
    """the lock of a registered context (diagnostics; `using` takes it itself).  A handle that is no longer
    registered was destroyed: using it would be a use-after-free, so this raises instead of inventing a lock."""
    with _ctxs_lock:
        for e in _ctxs.values():
            if e.h.value == h.value:
                return e.lock
    raise MdtRuntimeError("this mdt_ctx handle is no longer registered (it was evicted and destroyed)")


def status(key, clear=True):
    """sticky condition flags of the context of key = (device, stream) (mdt_ctx_status): what asynchronous requests
    could not report"""
    f = ctypes.c_int32(0)
    with using(*key) as h:
        check(load().mdt_ctx_status(h, ctypes.byref(f), 1 if clear else 0), h)
    return int(f.value)


def check(rc, h):
    if rc == MDT_OK:
        return
    msg = load().mdt_last_error(h).decode()
    if rc == MDT_EARG:
        raise ArgumentError(msg)
    if rc == MDT_EDOMAIN:
        raise DomainError(msg)
    if rc == MDT_EDIM:
        raise DimensionMismatch(msg)
    if rc == MDT_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == MDT_ENOMEM:
        raise MemoryError(msg)
    raise MdtRuntimeError(f"status {rc}: {msg}")
