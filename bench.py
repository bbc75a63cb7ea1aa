#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric: param-series/sec for ess_rhat (bulk+tail) on
draws x chains x params Float64, with the HBM roofline fraction and a CPU baseline beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...` (one rank per GPU over RCCL).  A "step" is one fused bulk+tail pass
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


def synth(draws, chains, params, device, seed):
    """(params, chains, draws) float64 tensor (draws fastest): half IID, half AR(1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(params, chains, draws, generator=g, dtype=torch.float64, device=device)
    half = params // 2
    phis = (-0.3, 0.3, 0.7, 0.9)
    B = 100 if draws % 100 == 0 else draws
    nblk = draws // B
    idx = torch.arange(B, device=device, dtype=torch.float64)
    for gi, phi in enumerate(phis):
        sl = slice(half + gi, params, len(phis))
        e = x[sl] * (1.0 - phi * phi) ** 0.5
        npar = e.shape[0]
        if npar == 0:
            continue
        # within-block response: y = e @ L^T, L[i, j] = phi^(i-j) for i >= j
        d = idx[:, None] - idx[None, :]
        L = torch.where(d >= 0, torch.tensor(phi, dtype=torch.float64, device=device) ** d.clamp(min=0), torch.zeros((), dtype=torch.float64, device=device))
        y = (e.reshape(npar * chains, nblk, B) @ L.T)
        carry = torch.tensor(phi, dtype=torch.float64, device=device) ** (idx + 1.0)
        prev = torch.zeros(npar * chains, 1, dtype=torch.float64, device=device)
        for b in range(nblk):
            y[:, b, :] += prev * carry
            prev = y[:, b, -1:].clone()
        y = y.reshape(npar, chains, draws)
        # per-chain offsets so that R-hat and the Geyer truncation length vary
        off = 0.05 * torch.arange(chains, device=device, dtype=torch.float64)[None, :, None]
        x[sl] = y + off
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--draws", type=int, default=10_000)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--params", type=int, default=10_000, help="parameters per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="pipelining experiment (DESIGN.md §5): round-robin the steps over this many HIP streams, each with "
                         "its own context, so that the tail of one persistent launch overlaps the head of the next; the "
                         "default 1 is the configuration every reported number and profile uses")
    ap.add_argument("--autocov", choices=("direct", "fft", "bda"), default="direct",
                    help="autocov_method of the request (BASELINE config 3 forces fft)")
    args = ap.parse_args()
    if args.streams > 1 and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("--streams is a single-GPU experiment")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=device)
    else:
        dist = None

    import __graft_entry__ as g
    pkg = g.load_package()

    D, C, P = args.draws, args.chains, args.params
    x_pcd = synth(D, C, P, device, seed=1234 + rank)   # generated per shard on its GPU
    x = x_pcd.permute(2, 1, 0)                          # logical (draws, chains, params), zero-copy
    outs = [torch.empty(P, dtype=torch.float64, device=device) for _ in range(4)]
    gathered = [torch.empty(world * P, dtype=torch.float64, device=device) for _ in range(4)] if world > 1 else None

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
            for o, gbuf in zip(outs, gathered):
                dist.all_gather_into_tensor(gbuf, o)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the stream the kernel runs on
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
                            + "; half IID N(0,1), half AR(1) phi in (-0.3,0.3,0.7,0.9)",
                "draws": D, "chains": C, "params_per_gpu": P, "autocov_method": type(method).__name__,
                "split_chains": 2, "maxlag": 250, "tail_prob": 0.1, "sharding": f"params axis x{world}",
                "streams": args.streams,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "kernel": "mdt_kernel<double>",
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_param": b_alg,
            },
        }
        # traffic per launch from the committed PMC profile of this same command, if present
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("draws") == D and tj.get("chains") == C and tj.get("params") == P:
                    res["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
            except Exception:
                pass

    # ---- CPU baseline (rank 0, N=1 only): the oracle ("port") on a bounded sample of the same data
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle
        c_oracle.build()
        got = [o.cpu().numpy() for o in outs]
        probe = 2
        xs = x[:, :, :probe].cpu().numpy()
        t1 = time.perf_counter()
        c_oracle.bulk_tail(xs, autocov=args.autocov)
        per = (time.perf_counter() - t1) / probe
        ns = int(max(4, min(P, args.cpu_seconds / max(per, 1e-6))))
        sel = np.linspace(0, P - 1, ns).astype(np.int64)  # spread over the IID and AR(1) halves
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
        # the same oracle on all host cores (parameters are independent; ctypes releases the GIL)
        try:
            from concurrent.futures import ThreadPoolExecutor
            ncore = len(os.sched_getaffinity(0))
            if ncore > 1:
                chunks = [c for c in np.array_split(np.arange(ns), ncore * 2) if len(c)]
                parts = [np.ascontiguousarray(xs[:, :, c]) for c in chunks]
                t1 = time.perf_counter()
                with ThreadPoolExecutor(ncore) as ex:
                    list(ex.map(lambda a: c_oracle.bulk_tail(a, autocov=args.autocov), parts))
                tall = time.perf_counter() - t1
                res["cpu_baseline_all_cores"] = {"value": ns / tall, "unit": "param-series/s", "cores": ncore, "kind": "port",
                                                 "sample": f"the same {ns} parameters split over {ncore} threads, {tall:.1f} s"}
        except Exception as e:  # the single-core figure above is the contractual one
            res["cpu_baseline_all_cores"] = {"error": str(e)}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
