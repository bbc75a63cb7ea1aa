#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric: param-series/sec for ess_rhat (bulk+tail) on
draws x chains x params Float64, with the HBM roofline fraction and a CPU baseline beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...` (one rank per GPU over RCCL); started plainly as `python bench.py --gpus N` it launches that same command
itself, as a child process, before anything touches a GPU, and exits with the child's code -- the line always reports
`n_gpus` = `--gpus`, or the run fails.  A "step" is one fused bulk+tail pass
(mdt_bulk_tail through the C-ABI) over one synthetic (draws, chains, params) array that is already
resident in HBM.  Weak scaling: every GPU owns `--params` parameters (the params axis shards with
no exchange); the only collective is the all-gather of the four small output
vectors, which is inside the timed region.

Workload at N=1 = BASELINE.json configs[1]: 10_000 draws x 4 chains x 10_000 params Float64
(3.2 GB), half IID N(0,1), half AR(1) with phi cycling (-0.3, 0.3, 0.7, 0.9) and per-chain offsets,
generated on the device (synthetic; there are no datasets in the image).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# The library reads its test / A-B switches only behind MDT_TEST_HOOKS=1.  The bench uses two of them for the diagnostic figures
# beside the headline (ms_no_order, the workgroup-per-parameter path beside the wave-per-parameter one); every OTHER MDT_* switch
# is removed from the environment here, so that every reported number is the default path's whatever the caller's shell holds.
def _own_the_switches():
    """(called by main(), not at import: the diagnostics under tools/ import synth / timed_ms from here and set switches of their own)"""
    for k in [k for k in os.environ if k.startswith("MDT_") and k not in ("MDT_LIB", "MDT_BENCH_DRYRUN", "MDT_TEST_HOOKS")]:
        del os.environ[k]
    os.environ["MDT_TEST_HOOKS"] = "1"


# ---------------------------------------------------------------- synthetic inputs
# Counter-based Philox-4x32-10 (Salmon et al., SC'11), keyed by the seed, counted by (draw, chain, GLOBAL parameter index):
# an element depends on where it sits in the global array and on nothing else, so the shard of rank r in an N-GPU run is a
# slice of the array the N = 1 run sees, whatever N is -- and a host twin (numpy) generates the same stream.
PHILOX_M0, PHILOX_M1, PHILOX_W0, PHILOX_W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_MASK32 = 0xFFFFFFFF


def _mulhilo(m, c):
    """(hi, lo) 32-bit halves of the 64-bit product m * c; c: int64 array/tensor holding 32-bit values.  Torch has no
    unsigned 64-bit arithmetic: the product is assembled from 16-bit halves of c (all intermediates < 2^49)."""
    cl, ch = c & 0xFFFF, c >> 16
    pl, ph = m * cl, m * ch
    lo = (pl + ((ph & 0xFFFF) << 16)) & _MASK32
    hi = (ph + (pl >> 16)) >> 16
    return hi, lo


def philox4x32(c0, c1, c2, c3, k0, k1):
    """ten rounds; counters are int64 arrays/tensors of 32-bit values (numpy or torch), keys Python ints"""
    for _ in range(10):
        h0, l0 = _mulhilo(PHILOX_M0, c0)
        h1, l1 = _mulhilo(PHILOX_M1, c2)
        c0, c1, c2, c3 = (h1 ^ c1 ^ k0), l1, (h0 ^ c3 ^ k1), l0
        k0, k1 = (k0 + PHILOX_W0) & _MASK32, (k1 + PHILOX_W1) & _MASK32
    return c0, c1, c2, c3


def philox_normal(draws, chains, p0, p1, seed, device=None, dtype=None):
    """standard normals of the global elements (draw d, chain c, parameter p), p0 <= p < p1, shape (p1 - p0, chains, draws):
    Box-Muller on the first two words of philox(counter = (d, c, p mod 2^32, p div 2^32), key = (seed mod 2^32, seed div 2^32)).
    device None: numpy (the host twin); otherwise a torch tensor on that device."""
    if device is None:
        xp, kw = np, {}
        d = np.arange(draws, dtype=np.int64)[None, None, :]
        c = np.arange(chains, dtype=np.int64)[None, :, None]
        p = np.arange(p0, p1, dtype=np.int64)[:, None, None]
    else:
        xp, kw = torch, {"device": device}
        d = torch.arange(draws, dtype=torch.int64, **kw)[None, None, :]
        c = torch.arange(chains, dtype=torch.int64, **kw)[None, :, None]
        p = torch.arange(p0, p1, dtype=torch.int64, **kw)[:, None, None]
    shape = (p1 - p0, chains, draws)
    z = xp.zeros(shape, dtype=xp.int64, **kw)
    r0, r1, _, _ = philox4x32(d + z, c + z, (p & _MASK32) + z, (p >> 32) + z, seed & _MASK32, (seed >> 32) & _MASK32)
    f64 = xp.float64
    u1 = (r0.to(f64) if xp is torch else r0.astype(f64)) * 2.0 ** -32 + 2.0 ** -33   # in (0, 1)
    u2 = (r1.to(f64) if xp is torch else r1.astype(f64)) * 2.0 ** -32
    n = xp.sqrt(-2.0 * xp.log(u1)) * xp.cos(2.0 * np.pi * u2)
    if dtype is not None:
        n = n.to(dtype) if xp is torch else n.astype(dtype)
    return n


def _ar1_filter(e, phi, B=100):
    """x[t] = phi x[t-1] + e[t], x[-1] = 0, along the last axis of e (rows, draws): blocks of B draws by one matmul with the
    block's response matrix, the carry between blocks by a rank-one update (torch or numpy)"""
    xp = torch if isinstance(e, torch.Tensor) else np
    rows, draws = e.shape
    if draws % B:
        B = draws
    nblk = draws // B
    kw = {"device": e.device} if xp is torch else {}
    idx = xp.arange(B, dtype=e.dtype, **kw)
    dlt = idx[:, None] - idx[None, :]
    zero = xp.zeros((), dtype=e.dtype, **kw) if xp is torch else 0.0
    L = xp.where(dlt >= 0, phi ** (dlt.clamp(min=0) if xp is torch else np.clip(dlt, 0, None)), zero)
    carry = phi ** (idx + 1.0)
    y = e.reshape(rows, nblk, B) @ L.T
    prev = xp.zeros((rows, 1), dtype=e.dtype, **kw)
    for b in range(nblk):
        y[:, b, :] += prev * carry
        prev = y[:, b, -1:].clone() if xp is torch else y[:, b, -1:].copy()
    return y.reshape(rows, draws)


PHIS = (-0.3, 0.3, 0.7, 0.9)


def synth(draws, chains, params, device, seed, p0=0, chunk=500):
    """Parameters [p0, p0 + params) of the GLOBAL config-2/3/4 array, shape (params, chains, draws) float64 (draws fastest):
    global parameter p is IID N(0,1) when p mod 8 < 4, else AR(1) with phi = PHIS[p mod 4], unit stationary variance and
    per-chain offsets (so that R-hat and the Geyer truncation length vary): half IID, half AR(1), in every slice.
    device None: the numpy twin."""
    host = device is None
    out = (np.empty if host else torch.empty)((params, chains, draws), dtype=(np.float64 if host else torch.float64),
                                              **({} if host else {"device": device}))
    for q0 in range(0, params, chunk):
        q1 = min(params, q0 + chunk)
        x = philox_normal(draws, chains, p0 + q0, p0 + q1, seed, device)
        pg = np.arange(p0 + q0, p0 + q1)
        for gi, phi in enumerate(PHIS):
            sel = np.nonzero((pg % 8 >= 4) & (pg % 4 == gi))[0]
            if len(sel) == 0:
                continue
            si = sel if host else torch.as_tensor(sel, device=device)
            e = x[si] * (1.0 - phi * phi) ** 0.5
            y = _ar1_filter(e.reshape(len(sel) * chains, draws), phi).reshape(len(sel), chains, draws)
            off = 0.05 * (np.arange(chains, dtype=np.float64) if host else torch.arange(chains, device=device, dtype=torch.float64))[None, :, None]
            x[si] = y + off
        out[q0:q1] = x
    return out


def synth_cfg5(draws, chains, params, device, seed, p0=0, chunk=500):
    """Parameters [p0, p0 + params) of the GLOBAL config-5 array: (params, chains, draws) float32; chains grouped in
    superchains of 4 that share their initial value (the ar1_superchains recipe of test/rhat_nested.jl:8-21:
    x[1] = init, x[t] = phi x[t-1] + eps[t]); phi = 0 for even global parameters (IID after the first draw), 0.6 for odd
    ones, unit stationary variance.  Philox-keyed like synth()."""
    host = device is None
    out = (np.empty if host else torch.empty)((params, chains, draws), dtype=(np.float32 if host else torch.float32),
                                              **({} if host else {"device": device}))
    phi = 0.6
    for q0 in range(0, params, chunk):
        q1 = min(params, q0 + chunk)
        x = philox_normal(draws, chains, p0 + q0, p0 + q1, seed, device)
        # the first "innovation" of a chain is the shared initial value of its superchain: that of the superchain's first chain
        lead = x[:, ::4, :1]
        x[:, :, :1] = (np.repeat(lead, 4, axis=1) if host else lead.repeat_interleave(4, dim=1))
        pg = np.arange(p0 + q0, p0 + q1)
        sel = np.nonzero(pg % 2 == 1)[0]
        if len(sel):
            si = sel if host else torch.as_tensor(sel, device=device)
            e = x[si]
            e[:, :, 1:] *= (1 - phi * phi) ** 0.5
            x[si] = _ar1_filter(e.reshape(len(sel) * chains, draws), phi).reshape(len(sel), chains, draws)
        out[q0:q1] = x.to(torch.float32) if not host else x.astype(np.float32)
    return out


def timed_ms(fn, steps, warmup):
    """average HIP-event duration of fn() on the current stream"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / steps


def max_rel_err(got, want):
    err = 0.0
    for a, b in zip(got, want):
        m = np.isfinite(b)
        if m.any():
            err = max(err, float(np.max(np.abs(np.asarray(a, dtype=np.float64)[m] - b[m]) / np.abs(b[m]))))
    return err


N_SIMD, MAX_CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs; /opt/skills/guides/MI355X_MICROARCH.md


def committed_counters():
    """profiles/traffic.json: calibrated memory-side bytes and SQ instruction counters per launch of the measured workloads, from
    the committed rocprofv3 --pmc passes of the same workloads (tools/collect_profiles.sh + tools/summarize_profiles.py); not
    measured in this run"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)
    except Exception:
        return {}


def counter_bounds(tj, key, shape, ms):
    """(traffic, issue) of workload `key` when the committed counters are of this shape.  issue: the second roofline of the
    line -- these launches are bound by vector instruction issue, not by HBM: a SIMD issues one wave64 vector instruction per
    4 cycles at best (one wave's stream; the f64 operations that dominate here run below that), and while a v_mfma_f64 executes
    (SQ_VALU_MFMA_BUSY_CYCLES) nothing else issues on its SIMD (measured): floor = (4 (VALU - MFMA instructions) +
    MFMA busy cycles) / 1024 SIMDs / 2.4 GHz; frac = floor / measured duration."""
    w = (tj.get("workloads") or {}).get(key)
    if not w or (w.get("draws"), w.get("chains"), w.get("params")) != tuple(shape):
        return None, None
    src = f"profiles/traffic.json ({tj.get('profile')}: separate rocprofv3 --pmc passes of this workload; FETCH_SIZE / WRITE_SIZE calibrated on a copy kernel of known size in the same session; not measured in this run)"
    traffic = {"bytes_per_launch": w.get("memory_side_bytes_per_launch"), "over_algorithmic": w.get("traffic_over_algorithmic"), "source": src}
    issue = None
    if w.get("valu_wave_insts"):
        mf, mb = w.get("mfma_wave_insts") or 0.0, w.get("mfma_busy_cycles") or 0.0
        floor_ms = (4.0 * (w["valu_wave_insts"] - mf) + mb) / N_SIMD / MAX_CLOCK_HZ * 1e3
        issue = {"bound": "vector instruction issue", "valu_wave_insts": w["valu_wave_insts"], "mfma_wave_insts": mf, "mfma_busy_cycles": mb,
                 "floor_ms": floor_ms, "frac": floor_ms / ms if ms else None,
                 "model": "(4 cycles x (VALU - MFMA wave-instructions) + MFMA busy cycles) / 1024 SIMDs / 2.4 GHz", "source": src}
        if w.get("valu_active_quad_cycles") and w.get("wave_quad_cycles"):
            issue["valu_active_over_wave_cycles"] = w["valu_active_quad_cycles"] / w["wave_quad_cycles"]
    return traffic, issue


def other_configs(pkg, device, steps, warmup):
    """BASELINE configs 3 and 5 (per-GPU shard) measured in the same run: time per call, param-series/s, the
    HBM-roofline fraction on algorithmic bytes, and the oracle on a few parameters of the timed array."""
    from oracle import c_oracle
    out = {}
    # ---- config 3: 100 000 draws x 8 chains x 1 000 params Float64, FFTAutocovMethod (the long-chain case)
    D, C, P = 100_000, 8, 1_000
    x_pcd = synth(D, C, P, device, seed=4321)
    x = x_pcd.permute(2, 1, 0)
    m = pkg.FFTAutocovMethod()
    outs = [torch.empty(P, dtype=torch.float64, device=device) for _ in range(4)]
    ms = timed_ms(lambda: pkg.bulk_tail(x, autocov_method=m, sync=False, out=outs), steps, warmup)
    flags = pkg.status()
    pick = np.array([1, P // 2 + 3, P - 1])
    xs = x_pcd[torch.as_tensor(pick, device=device)].permute(2, 1, 0).cpu().numpy()
    t1 = time.perf_counter()
    want = c_oracle.bulk_tail(xs, autocov="fft")
    cpu_t = time.perf_counter() - t1
    b_alg = D * C * 8 + 4 * 8
    out["config3"] = {
        "workload": "ess_rhat bulk+tail, 100000 draws x 8 chains x 1000 params Float64, FFTAutocovMethod (BASELINE.json configs[2]); "
                    "evaluated from lag products -- no device transform runs: zero padding makes the transform's result exactly those "
                    "sums, checked here against the oracle's real FFT",
        "ms": ms, "param_series_per_s": P / (ms * 1e-3), "kernel": "mdt_big::mdt_kernel<double>",
        "roofline": {"bound": "hbm", "achieved": P * b_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": P * b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_param": b_alg},
        "parity_max_rel_err_vs_oracle": max_rel_err([o.cpu().numpy()[pick] for o in outs], want),
        "parity_sample": f"{len(pick)} parameters against oracle/mdt_oracle.c with its real FFT",
        "cpu_port_param_series_per_s_1core": len(pick) / cpu_t, "status_flags": flags,
    }
    del x, x_pcd, outs
    torch.cuda.empty_cache()
    # ---- config 5, one GPU's shard: 10 000 draws x 16 chains (4 superchains) x 12 500 params Float32
    D, C, P = 10_000, 16, 12_500
    x_pcd = synth_cfg5(D, C, P, device, seed=55)
    x = x_pcd.permute(2, 1, 0)
    ids = np.repeat(np.arange(1, 5), 4)
    res = {}

    def step5():
        res["mcse"], res["nested"] = pkg.mcse_rhat_nested(x, ids, kind="rank")   # ONE launch (mdt_mcse_nested)
    step5()
    torch.cuda.synchronize()
    ms = timed_ms(step5, steps, 1)
    t_m = timed_ms(lambda: pkg.mcse(x), steps, 1)
    t_n = timed_ms(lambda: pkg.rhat_nested(x, ids, kind="rank"), steps, 1)
    flags = pkg.status()
    pick = np.array([0, 7, P // 2 + 1, P - 2])
    xs = x_pcd[torch.as_tensor(pick, device=device)].permute(2, 1, 0).cpu().numpy()
    inds = np.arange(16, dtype=np.int32).reshape(4, 4, order="F")
    want = [c_oracle.mcse_mean(xs), c_oracle.rhat_nested(xs, inds, kind="rank")]
    faithful = [c_oracle.mcse_mean(xs, faithful=True), c_oracle.rhat_nested(xs, inds, kind="rank", faithful=True)]
    got = [res["mcse"].cpu().numpy()[pick], res["nested"].cpu().numpy()[pick]]
    b_alg = D * C * 4 + 2 * 4  # the slice read once, two output values
    out["config5_shard"] = {
        "workload": "mcse(kind=mean) + rhat_nested(kind=rank), 10000 draws x 16 chains (4 superchains) x 12500 params Float32: "
                    "one GPU's shard of BASELINE.json configs[4] (100000 params over 8 GPUs); ONE fused request (mdt_mcse_nested)",
        "ms": ms, "ms_separate_calls": {"mcse_mean": t_m, "rhat_nested_rank": t_n}, "param_series_per_s": P / (ms * 1e-3),
        "kernel": "mdt_small::mdt_kernel<float>",
        "roofline": {"bound": "hbm", "achieved": P * b_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": P * b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_param": b_alg,
                     "note": "one fused request: the slice is counted once (SURVEY.md 8d)"},
        "parity_max_rel_err_vs_oracle": max_rel_err(got, want),
        "parity_sample": f"{len(pick)} parameters against oracle/mdt_oracle.c (Float32 input, Float64 internal comparator)",
        "distance_to_float32_faithful_oracle": {
            "max_rel": max_rel_err(got, faithful),
            "note": "reported, not gated on (SURVEY.md 8c): the reference computes Float32 data in Float32; the device accumulates in "
                    "Float64 and rounds the result, the faithful restatement carries the reference's ~1e-4 rounding",
        },
        "status_flags": flags,
    }
    del x, x_pcd
    torch.cuda.empty_cache()
    # ---- the headline request on two more shapes: config 1's per-parameter shape at scale (what a sampler emits: 1000 draws x
    # 4 chains), and the shard BASELINE config 4 really puts on one GPU (1 000 000 params over 8 GPUs = 125 000 each, 40 GB)
    for key, (D, C, P), label, nstep in (
            ("config1_at_scale", (1_000, 4, 100_000), "config 1's slice shape (BASELINE.json configs[0]: 1000 draws x 4 chains) at 100000 params", steps),
            ("short_500x4", (500, 4, 100_000), "500 draws x 4 chains at 100000 params (a short run of a sampler)", steps),
            ("config4_shard", (10_000, 4, 125_000), "one GPU's shard of BASELINE.json configs[3] (1000000 params over 8 GPUs), 40 GB resident", 3)):
        x_pcd = synth(D, C, P, device, seed=1234, chunk=1000 if D >= 10_000 else 10_000)
        x = x_pcd.permute(2, 1, 0)
        outs = [torch.empty(P, dtype=torch.float64, device=device) for _ in range(4)]
        ms = timed_ms(lambda: pkg.bulk_tail(x, sync=False, out=outs), nstep, warmup)
        flags = pkg.status()
        pick = np.array([0, 5, P // 2 + 6, P - 1])
        xs = x_pcd[torch.as_tensor(pick, device=device)].permute(2, 1, 0).cpu().numpy()
        want = c_oracle.bulk_tail(xs)
        b_alg = D * C * 8 + 4 * 8
        wave = D * C <= 8192 and D // 2 <= 640 and 2 * C <= 16
        extra = {}
        if wave:  # the same request on the workgroup-per-parameter kernels (what served it until round 4), beside it
            os.environ["MDT_WAVE"] = "0"
            try:
                extra["ms_workgroup_per_parameter_path"] = timed_ms(lambda: pkg.bulk_tail(x, sync=False, out=outs), nstep, 1)
            finally:
                del os.environ["MDT_WAVE"]
        out[key] = {
            "workload": f"ess_rhat bulk+tail, {D} draws x {C} chains x {P} params Float64: " + label,
            "ms": ms, "param_series_per_s": P / (ms * 1e-3),
            "kernel": "mdtw::wave_kernel<double> (one parameter per wave, csrc/mdt_wave.hip)" if wave else "mdt_small::mdt_kernel<double>",
            **extra,
            "roofline": {"bound": "hbm", "achieved": P * b_alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": P * b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_param": b_alg},
            "parity_max_rel_err_vs_oracle": max_rel_err([o.cpu().numpy()[pick] for o in outs], want),
            "parity_sample": f"{len(pick)} parameters against oracle/mdt_oracle.c", "status_flags": flags,
        }
        del x, x_pcd, outs
        torch.cuda.empty_cache()
    tj = committed_counters()
    for key, ckey, shape in (("config3", "cfg3", (100_000, 8, 1_000)), ("config5_shard", "cfg5", (10_000, 16, 12_500)),
                             ("config1_at_scale", "cfg1", (1_000, 4, 100_000)), ("short_500x4", "s500", (500, 4, 100_000))):
        traffic, issue = counter_bounds(tj, ckey, shape, out[key]["ms"])
        out[key]["roofline"]["traffic"] = traffic["bytes_per_launch"] if traffic else None
        if traffic:
            out[key]["roofline"]["traffic_over_algorithmic"] = traffic["over_algorithmic"]
            out[key]["roofline"]["traffic_source"] = traffic["source"]
            out[key]["roofline"]["traffic_rate_gbs"] = traffic["bytes_per_launch"] / (out[key]["ms"] * 1e-3) / 1e9
        if issue:
            out[key]["roofline"]["issue"] = issue
    return out


def request_kinds(pkg, device, steps=3):
    """Partial requests -- what the reference's API hands out one call at a time (`ess(kind = :tail)`, `rhat(kind = :tail)`,
    `mcse(kind = mean)`, `kind = :basic`, `ess(kind = median | mad)`, `mcse(kind = median)`) -- on config 1's slice shape at scale and on config 2's: time per
    call on the default path and, beside it, on the path that served the request until round 5 (the library's own switch, behind
    the test-hooks gate the bench sets): round 6 stopped such requests from paying for pipeline stages they do not need."""
    import warnings
    from oracle import c_oracle
    out = {"note": "ms per call (HIP events, synchronous API calls), 3 timed calls after one warm-up; `before` = the same request on the "
                   "route of rounds 1-5 (MDT_WAVE=0: workgroup-per-parameter kernels; MDT_WAVE_OSONLY=0: full ranking on the wave kernel; "
                   "MDT_TAIL_SRT=0: order-statistics-only ranking + a second full ranking of |x - median|)"}

    def both(fn, env):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            now = timed_ms(fn, steps, 1)
            old = dict((k, os.environ.get(k)) for k in env)
            os.environ.update(env)
            try:
                before = timed_ms(fn, steps, 1)
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        return {"ms": round(now, 3), "ms_before": round(before, 3)}

    # config 1's slice shape at scale: the wave-per-parameter kernel
    D, C, P = 1_000, 4, 100_000
    x_pcd = synth(D, C, P, device, seed=1234, chunk=10_000)
    x = x_pcd.permute(2, 1, 0)
    rows = {}
    rows["rhat_basic"] = both(lambda: pkg.rhat(x, kind="basic"), {"MDT_WAVE": "0"})
    rows["ess_rhat_basic"] = both(lambda: pkg.ess_rhat(x, kind="basic"), {"MDT_WAVE": "0"})
    rows["mcse_mean"] = both(lambda: pkg.mcse(x), {"MDT_WAVE": "0"})
    rows["ess_tail"] = both(lambda: pkg.ess(x, kind="tail"), {"MDT_WAVE_OSONLY": "0"})
    rows["ess_median"] = both(lambda: pkg.ess(x, kind=np.median), {"MDT_WAVE": "0"})
    rows["ess_mad"] = both(lambda: pkg.ess(x, kind="mad"), {"MDT_WAVE": "0"})
    rows["mcse_median"] = both(lambda: pkg.mcse(x, kind=np.median), {"MDT_WAVE": "0"})
    pick = np.array([0, 5, P // 2 + 6, P - 1])
    xs = x_pcd[torch.as_tensor(pick, device=device)].permute(2, 1, 0).cpu().numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = [pkg.mcse(x).cpu().numpy()[pick], pkg.ess(x, kind="tail").cpu().numpy()[pick], pkg.rhat(x, kind="basic").cpu().numpy()[pick]]
    want = [c_oracle.mcse_mean(xs), c_oracle.ess_rhat(xs, kind="tail")[0], c_oracle.ess_rhat(xs, kind="basic")[1]]
    out["config1_shape"] = {"workload": f"{D} draws x {C} chains x {P} params Float64", "rows": rows,
                            "parity_max_rel_err_vs_oracle": max_rel_err(got, want),
                            "parity_sample": f"mcse(mean), ess(tail), rhat(basic) of {len(pick)} parameters against oracle/mdt_oracle.c"}
    del x, x_pcd
    torch.cuda.empty_cache()
    # config 2's shape: the workgroup-per-parameter kernels
    D, C, P = 10_000, 4, 10_000
    x_pcd = synth(D, C, P, device, seed=1234)
    x = x_pcd.permute(2, 1, 0)
    rows = {}
    rows["ess_rhat_tail"] = both(lambda: pkg.ess_rhat(x, kind="tail"), {"MDT_TAIL_SRT": "0"})
    rows["rhat_tail"] = both(lambda: pkg.rhat(x, kind="tail"), {"MDT_TAIL_SRT": "0"})
    pick = np.array([1, P // 2 + 3, P - 1])
    xs = x_pcd[torch.as_tensor(pick, device=device)].permute(2, 1, 0).cpu().numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        e, r = pkg.ess_rhat(x, kind="tail")
    we, wr = c_oracle.ess_rhat(xs, kind="tail")
    out["config2_shape"] = {"workload": f"{D} draws x {C} chains x {P} params Float64", "rows": rows,
                            "parity_max_rel_err_vs_oracle": max_rel_err([e.cpu().numpy()[pick], r.cpu().numpy()[pick]], [we, wr]),
                            "parity_sample": f"ess_rhat(kind = tail) of {len(pick)} parameters against oracle/mdt_oracle.c"}
    out["status_flags"] = pkg.status()
    del x, x_pcd
    torch.cuda.empty_cache()
    return out


def wave_rule(D, C, P, split=2):
    """the host's dispatch rule for the wave-per-parameter kernel (run() in csrc/mdt_hip.hip): slices of 256 ... 8192
    values in at most 16 split chains of 8 ... 1024 draws, from the parameter count at which a wave per parameter pays"""
    n, niter = D * C, D // split
    return 256 <= n <= 8192 and 8 <= niter <= 1024 and split * C <= 16 and P >= (3 * n if (n > 4096 or niter > 640) else 700 + (4 * n) // 3)


def shape_sweep(pkg, device, target_bytes=1.6e9, steps=3):
    """the fused bulk+tail request over draws x chains shapes at ~1.6 GB of Float64 each: every boundary between the paths
    (wave-per-parameter kernel up to 8192 values per slice in split chains of at most 1024 draws, 256- / 512-thread workgroups per parameter beyond) is visible
    in the driver's record"""
    rows = []
    for D in (250, 500, 1_000, 2_000, 5_000, 10_000):
        for C in (2, 4, 8):
            P = max(256, int(target_bytes / (8 * D * C)))
            x_pcd = synth(D, C, P, device, seed=99, chunk=max(500, min(20_000, int(2e8 / (8 * D * C)))))
            x = x_pcd.permute(2, 1, 0)
            outs = [torch.empty(P, dtype=torch.float64, device=device) for _ in range(4)]
            ms = timed_ms(lambda: pkg.bulk_tail(x, sync=False, out=outs), steps, 1)
            b_alg = D * C * 8 + 4 * 8
            rows.append({"draws": D, "chains": C, "params": P, "ms": round(ms, 3), "param_series_per_s": round(P / (ms * 1e-3), 1),
                         "hbm_frac": round(P * b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "path": "wave" if wave_rule(D, C, P) else ("wg256" if D * C <= 12288 and D // 2 <= 1392 else "wg512")})
            del x, x_pcd, outs
            torch.cuda.empty_cache()
    return {"note": "bulk+tail, split_chains 2, maxlag 250, ~1.6 GB of Float64 per shape, 3 timed steps after 1 warm-up; path = which kernel "
                    "family serves the shape by default", "rows": rows, "status_flags": pkg.status()}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a CHILD process (never an exec: this process has not touched the GPU, and must not replace itself once
    it has) and return its exit code.  Rank 0 of the child prints the JSON line."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # --standalone: the launcher picks its own rendezvous port (binding a socket here to find a free one, closing it and passing
    # the number on left a window in which another process could take it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class DryRunEngine:
    """MDT_BENCH_DRYRUN=1 (tests/test_bench_launch.py): the launch, sharding, gather, timing and reporting plumbing of this
    file on CPU tensors over gloo, with the compute INJECTED -- every output is the global index of its parameter -- so that
    `--gpus 2` can be shown to run two ranks and to report n_gpus 2 where there is no GPU.  Nothing here is a fallback of
    the product: the line says "dry_run": true and carries no roofline, and bench.py refuses to run without a GPU otherwise."""

    def __init__(self, p0):
        self.p0 = p0

    def bulk_tail(self, x, autocov_method=None, sync=False, out=None):
        P = x.shape[2]
        for o in out:
            o.copy_(torch.arange(self.p0, self.p0 + P, dtype=o.dtype))
        return out

    def status(self):
        return 0

    class _M:
        pass
    AutocovMethod = FFTAutocovMethod = BDAAutocovMethod = _M


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--draws", type=int, default=10_000)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--params", type=int, default=10_000, help="parameters per GPU")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE.json configs[3]: 125000 parameters per GPU (1000000 over 8 GPUs, 40 GB resident per GPU); same as --params 125000")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the BASELINE configs 3 and 5 that the N = 1 line reports beside the headline")
    ap.add_argument("--streams", type=int, default=1,
                    help="pipelining experiment (DESIGN.md §5): round-robin the steps over this many HIP streams, each with "
                         "its own context, so that the tail of one persistent launch overlaps the head of the next; the "
                         "default 1 is the configuration every reported number and profile uses")
    ap.add_argument("--autocov", choices=("direct", "fft", "bda"), default="direct",
                    help="autocov_method of the request (BASELINE config 3 forces fft)")
    args = ap.parse_args()
    _own_the_switches()
    if args.config4:
        args.params = 125_000
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one (a child process, started before any GPU call), so that the line printed
        # is the N-GPU line the command asked for
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to "
                         f"report a line whose n_gpus is not the one asked for")
    if args.streams > 1 and world > 1:
        raise SystemExit("--streams is a single-GPU experiment")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = os.environ.get("MDT_BENCH_DRYRUN") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dry:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
    else:
        dist = None
    dev_sync = (lambda: None) if dry else torch.cuda.synchronize

    D, C, P = args.draws, args.chains, args.params
    if dry:
        pkg = DryRunEngine(rank * P)
        x_pcd = torch.from_numpy(synth(D, C, P, None, seed=1234, p0=rank * P))
    else:
        import __graft_entry__ as g
        pkg = g.load_package()
        x_pcd = synth(D, C, P, device, seed=1234, p0=rank * P)   # rank r owns global parameters [r P, (r + 1) P): generated on its GPU
    x = x_pcd.permute(2, 1, 0)                          # logical (draws, chains, params), zero-copy
    outs4 = torch.empty((4, P), dtype=torch.float64, device=device)   # the four output vectors, packed: ONE all-gather per step
    outs = [outs4[k] for k in range(4)]
    gathered = torch.empty((world, 4, P), dtype=torch.float64, device=device) if world > 1 else None

    method = {"direct": pkg.AutocovMethod, "fft": pkg.FFTAutocovMethod, "bda": pkg.BDAAutocovMethod}[args.autocov]()

    streams = [torch.cuda.Stream(device=device) for _ in range(args.streams)] if args.streams > 1 else None
    outs_s = [[torch.empty(P, dtype=torch.float64, device=device) for _ in range(4)] for _ in range(args.streams)] if streams else None
    counter = [0]

    def step():
        if streams:  # (N = 1 only: the gather below reads `outs` on the current stream)
            i = counter[0] % len(streams)
            counter[0] += 1
            with torch.cuda.stream(streams[i]):
                pkg.bulk_tail(x, autocov_method=method, sync=False, out=outs_s[i])
            return
        pkg.bulk_tail(x, autocov_method=method, sync=False, out=outs)
        if world > 1:
            dist.all_gather_into_tensor(gathered.view(-1), outs4.view(-1))   # (flat views: one contiguous [4 P] block per rank)

    # A diagnostic is normally called ONCE per array.  The first request of a process pays for the context (scratch
    # allocation, the score table of the shape) and runs its tickets in parameter order; the next ones run them in the
    # order of the costs the launch before measured ON THE SAME ARRAY.  Both are reported beside the
    # steady state: the wall time of the very first call, of the second (context warm, first cost-ordered launch) and
    # -- after the timed loop -- the steady state with the order switched off (ms_no_order).
    cold = {}
    if not dry and not streams and world == 1:
        def wall_ms(fn):
            dev_sync()
            tc = time.perf_counter()
            fn()
            dev_sync()
            return (time.perf_counter() - tc) * 1e3
        # (a) what a user's ONE call costs on a fresh context: context (code object, status words), scratch and ticket tables,
        # the score table of the shape, the launch -- measured on a side stream, whose context nothing else here uses
        side = torch.cuda.Stream(device=device)
        outs_side = [torch.empty(P, dtype=torch.float64, device=device) for _ in range(4)]

        def first_on_side():
            with torch.cuda.stream(side):
                pkg.bulk_tail(x, autocov_method=method, sync=False, out=outs_side)
        cold["first_call_ms"] = wall_ms(first_on_side)
        # (b) the same with the shape announced first (mdt_ctx_reserve): creating the context, reserving, then the first request
        cur = torch.cuda.current_stream(device).cuda_stream
        cold["ctx_create_ms"] = wall_ms(lambda: pkg._lib.ctx(device.index, cur))
        cold["ctx_reserve_ms"] = wall_ms(lambda: pkg.reserve(D, C, P, dtype=np.float64))
        cold["first_request_ms"] = wall_ms(step)
        cold["second_call_ms"] = wall_ms(step)
        cold["first_call_note"] = ("first_call_ms: one call on a fresh context (context + allocations + score table + launch in parameter order); "
                                   "ctx_create_ms / ctx_reserve_ms / first_request_ms: the same split up -- with mdt_ctx_reserve(shape) called "
                                   "beforehand the first request pays the launch only (tickets in parameter order: no cost history yet)")
        del outs_side
    for _ in range(args.warmup):
        step()
    dev_sync()
    if world > 1:
        dist.barrier()
    dev_sync()
    if not dry:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if not dry:
        ev0.record()
    for _ in range(args.steps):
        step()
    if not dry:
        ev1.record()
    dev_sync()
    if world > 1:
        dist.barrier()
    dev_sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps if not dry else elapsed / args.steps * 1e3  # HIP events on the stream the kernel runs on
    if world > 1:  # the gather did what it is there for: every rank holds every rank's block of every output
        for r in range(world):
            for k in range(4):
                blk = gathered[r, k]
                if dry:
                    assert torch.equal(blk, torch.arange(r * P, (r + 1) * P, dtype=blk.dtype)), "gathered block mismatch"
        if not dry:
            assert torch.equal(gathered[rank], outs4), "gathered block mismatch"
    # the timed requests were asynchronous: what they could not raise is on the context (mdt_ctx_status)
    status_flags = pkg.status() if not streams else 0
    if status_flags:
        raise SystemExit(f"bench.py: the timed launches raised condition flags {status_flags} (see include/mdt_hip.h MDT_FLAG_*)")
    if streams:  # overlapping launches: the per-step figure is the wall time; per-launch durations are in rocprofv3
        kernel_ms = elapsed / args.steps * 1e3
        outs = outs_s[0]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    res = None
    if rank == 0:
        total_params = P * world * args.steps
        value = total_params / elapsed
        b_alg = D * C * 8 + 4 * 8  # algorithmic bytes per param-series: read x once + 4 outputs
        achieved = P * b_alg / (kernel_ms * 1e-3) / 1e9
        res = {
            "metric": "param-series/sec for ess_rhat (bulk+tail) on draws x chains x params; HBM GB/s fraction",
            "value": value,
            "unit": "param-series/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"ess_rhat bulk+tail, {D} draws x {C} chains x {P} params Float64 per GPU "
                            + ("(BASELINE.json configs[1])" if (D, C, P) == (10000, 4, 10000) else "(non-default shape)")
                            + "; half IID N(0,1), half AR(1) phi in (-0.3,0.3,0.7,0.9), Philox-4x32-10 keyed by (seed, draw, chain, global parameter)",
                "draws": D, "chains": C, "params_per_gpu": P, "autocov_method": type(method).__name__,
                "split_chains": 2, "maxlag": 250, "tail_prob": 0.1, "sharding": f"params axis x{world}",
                "streams": args.streams, "status_flags_after_timed_loop": status_flags,
                "ticket_order": ("steady state: the timed launches hand out their tickets in the order of the per-parameter costs the launch "
                                 "before measured on this same array (warm-up passes included; DESIGN.md 4.1d); a first call runs in parameter "
                                 "order (first_call_ms), ms_no_order is the steady state with MDT_NO_ORDER=1"),
            },
            **cold,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "traffic_source": None,
                "kernel": "mdt_small::mdt_kernel<double>",
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_param": b_alg,
                "limiter_note": ("the nominal bound of this scan/sort/reduce path is HBM (SURVEY.md 8d); what limits the launch in practice is instruction "
                                 "issue and dependent LDS / L2 round trips between barriers -- about 490 vector instructions per element, vector ALUs "
                                 "about 60 % busy at four waves per SIMD (profiles/r04f_pmc_phases_cfg2.txt, DESIGN.md 5d)"),
            },
        }
        # traffic per launch and the instruction-issue bound from the committed PMC profile of this same workload, if present
        tj = committed_counters()
        traffic, issue = counter_bounds(tj, "cfg2", (D, C, P), kernel_ms)
        if traffic:
            res["roofline"]["traffic"] = traffic["bytes_per_launch"]
            res["roofline"]["traffic_source"] = traffic["source"]
            # the same roofline on the bytes the launch really moves (memory side, from the committed counters) over the kernel time
            # measured HERE: how busy the memory system is, as opposed to how far the algorithm's traffic is from its input size
            res["roofline"]["traffic_rate"] = {"achieved": traffic["bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": traffic["bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "note": "memory-side bytes per launch (roofline.traffic) / kernel_ms; a plain copy reaches ~7200 GB/s "
                                                       "on the same counters; by phase: profiles/r06_pmc_phases_cfg2.txt (the first ranking pass runs at 5700 GB/s)"}
        if issue:
            res["roofline"]["issue"] = issue
            res["roofline"]["limiter_note"] = (
                "the nominal bound of this scan/sort/reduce path is HBM (SURVEY.md 8d).  The launch moves ~12x its input through the memory side "
                "(roofline.traffic, roofline.traffic_rate: about half of the HBM peak over the whole launch); its first ranking pass is bound by that "
                "bandwidth (5.7 TB/s where a copy reaches 7.2: profiles/r06_pmc_phases_cfg2.txt), the other phases by dependent LDS / L2 round trips "
                f"between barriers at four waves per SIMD (vector pipes 56 % busy by SQ_ACTIVE_INST_VALU; roofline.issue: {issue['valu_wave_insts']:.3g} vector "
                f"wave-instructions per launch, floor {issue['floor_ms']:.2f} ms); counters: profiles/{tj.get('profile')}_counters.txt, "
                "profiles/r06_instruction_mix.txt, phases: DESIGN.md 5b")

        if dry:
            res["dry_run"] = True
            del res["roofline"]
    if rank == 0 and world == 1 and not dry and not streams:
        # the steady state once more with the ticket order switched off (read per request): parameter order every launch
        os.environ["MDT_NO_ORDER"] = "1"
        try:
            res["ms_no_order"] = timed_ms(step, args.steps, 1)
        finally:
            del os.environ["MDT_NO_ORDER"]

    # ---- CPU baseline (rank 0, N=1 only): the oracle ("port") on a bounded sample of the same data
    if rank == 0 and world == 1 and not dry and not args.no_cpu_baseline:
        from oracle import c_oracle
        c_oracle.build()
        got = [o.cpu().numpy() for o in outs]
        probe = 2
        xs = x[:, :, :probe].cpu().numpy()
        t1 = time.perf_counter()
        c_oracle.bulk_tail(xs, autocov=args.autocov)
        per = (time.perf_counter() - t1) / probe
        ns = int(max(4, min(P, args.cpu_seconds / max(per, 1e-6))))
        sel = np.linspace(0, P - 1, ns).astype(np.int64)  # spread over the array (IID and AR(1) parameters alternate in blocks of 4)
        xs = x_pcd[torch.as_tensor(sel, device=device)].permute(2, 1, 0).cpu().numpy()
        t1 = time.perf_counter()
        want = c_oracle.bulk_tail(xs, autocov=args.autocov)
        cpu_t = time.perf_counter() - t1
        err = 0.0
        for a, b in zip(got, want):
            a = a[sel]
            m = np.isfinite(b)
            if m.any():
                err = max(err, float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))))
        res["cpu_baseline"] = {
            "value": ns / cpu_t,
            "unit": "param-series/s",
            "cores": 1,
            "kind": "port",
            "sample": f"{ns} of the {P} parameters of the same device-generated array (evenly spaced, "
                      f"copied D2H), oracle/mdt_oracle.c single-threaded, {cpu_t:.1f} s",
        }
        res["parity_max_rel_err_vs_oracle"] = err
        res["speedup_vs_cpu_1core"] = res["value"] / res["cpu_baseline"]["value"]
        # the same oracle on all host cores: an OpenMP team inside the oracle (oracle/mdt_oracle.c, mdt_oracle_bulk_tail_omp_f64),
        # blocks of 8 parameters handed out dynamically, at least 64 parameters per thread or the whole array
        try:
            nvis = len(os.sched_getaffinity(0))
            ncore, quota_note = nvis, ""
            try:  # the container's CPU quota (cgroup v2 cpu.max = "<quota> <period>"): more threads than that only throttle each other
                q, per = open("/sys/fs/cgroup/cpu.max").read().split()
                if q != "max":
                    ncore = max(1, min(nvis, int(int(q) // int(per))))
                    quota_note = f"; the box shows {nvis} hardware threads but its cgroup cpu.max grants {int(q) / int(per):g} CPUs: one thread per granted CPU"
            except Exception:
                pass
            if ncore > 1:
                nall = int(min(P, max(64 * ncore, ns)))
                sel_all = np.arange(P) if nall == P else np.linspace(0, P - 1, nall).astype(np.int64)
                xa = x_pcd[torch.as_tensor(sel_all, device=device)].permute(2, 1, 0).cpu().numpy()
                c_oracle.bulk_tail_all_cores(xa[:, :, : min(nall, 2 * ncore)], autocov=args.autocov, nthreads=ncore)  # team start-up, page faults
                t1 = time.perf_counter()
                wall = c_oracle.bulk_tail_all_cores(xa, autocov=args.autocov, nthreads=ncore, chunk=8)
                tall = time.perf_counter() - t1
                errA = 0.0
                for a_, b_ in zip(got, wall):
                    a_ = a_[sel_all]
                    m_ = np.isfinite(b_)
                    if m_.any():
                        errA = max(errA, float(np.max(np.abs(a_[m_] - b_[m_]) / np.abs(b_[m_]))))
                res["cpu_baseline_all_cores"] = {
                    "value": nall / tall, "unit": "param-series/s", "cores": ncore, "kind": "port",
                    "sample": f"{nall} of the {P} parameters of the same array ({nall // ncore} per thread), OpenMP parallel for over blocks of 8 "
                              f"parameters inside oracle/mdt_oracle.c, {ncore} threads, {tall:.1f} s" + quota_note,
                    "scaling_vs_1core": (nall / tall) / res["cpu_baseline"]["value"],
                    "parity_max_rel_err_vs_oracle": errA,
                }
                del xa
        except Exception as e:  # the single-core figure above is the contractual one
            res["cpu_baseline_all_cores"] = {"error": str(e)}
    if rank == 0 and world == 1 and not dry and not args.no_other_configs and (D, C, P) == (10000, 4, 10000) and not streams:
        del x, x_pcd
        torch.cuda.empty_cache()
        res["other_configs"] = other_configs(pkg, device, steps=5, warmup=1)
        res["other_configs"]["shape_sweep"] = shape_sweep(pkg, device)
        res["other_configs"]["request_kinds"] = request_kinds(pkg, device)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
