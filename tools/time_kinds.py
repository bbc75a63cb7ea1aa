#!/usr/bin/env python3
"""Timing of every kind / estimator on the BASELINE config-2 shape (10_000 x 4 x P Float64), or any other.
Usage: python tools/time_kinds.py [P [DRAWS [CHAINS]]]"""
import os, sys, time, warnings
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
from bench import synth
pkg = g.load_package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
C = int(sys.argv[3]) if len(sys.argv) > 3 else 4
x = synth(D, C, P, torch.device("cuda", 0), 1234).permute(2, 1, 0)
warnings.simplefilter("ignore")
calls = [("ess_rhat basic", lambda: pkg.ess_rhat(x, kind="basic")), ("rhat basic", lambda: pkg.rhat(x, kind="basic")),
         ("ess_rhat bulk", lambda: pkg.ess_rhat(x, kind="bulk")), ("ess_rhat tail", lambda: pkg.ess_rhat(x, kind="tail")),
         ("ess tail", lambda: pkg.ess(x, kind="tail")), ("rhat tail", lambda: pkg.rhat(x, kind="tail")),
         ("ess bulk", lambda: pkg.ess(x, kind="bulk")), ("rhat bulk", lambda: pkg.rhat(x, kind="bulk")),
         ("ess_rhat rank", lambda: pkg.ess_rhat(x, kind="rank")), ("rhat rank", lambda: pkg.rhat(x, kind="rank")),
         ("bulk_tail", lambda: pkg.bulk_tail(x)), ("ess median", lambda: pkg.ess(x, kind=np.median)),
         ("ess mad", lambda: pkg.ess(x, kind="mad")), ("ess std", lambda: pkg.ess(x, kind=np.std)),
         ("mcse mean", lambda: pkg.mcse(x)), ("mcse median", lambda: pkg.mcse(x, kind=np.median)),
         ("ess_rhat bulk fft", lambda: pkg.ess_rhat(x, kind="bulk", autocov_method=pkg.FFTAutocovMethod()))]
for name, f in calls:
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name:20s} {dt*1e3:8.2f} ms  {P/dt:10.0f} param-series/s  {P*D*C*8/dt/1e9:7.1f} GB/s of input")
