#!/usr/bin/env python3
"""Timing of the BASELINE config-5 shape (10_000 draws x 16 chains x P params Float32): mcse(kind=mean), then
rhat_nested(kind=:rank) with superchain_ids = repeat(1:4, inner=4).  Usage: python tools/time_cfg5.py [P]"""
import os, sys, time
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
D, C = 10_000, 16
gen = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(P, C, D, generator=gen, dtype=torch.float32, device="cuda").permute(2, 1, 0)
ids = np.repeat(np.arange(1, 5), 4)
for _ in range(2):
    pkg.mcse(x); pkg.rhat_nested(x, ids, kind="rank")
torch.cuda.synchronize()
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    pkg.mcse(x)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(reps):
    pkg.rhat_nested(x, ids, kind="rank")
torch.cuda.synchronize()
t2 = time.perf_counter()
a, b = (t1 - t0) / reps, (t2 - t1) / reps
print(f"config-5 shape, P={P}: mcse(mean) {a*1e3:.2f} ms ({P/a:.0f} param-series/s), rhat_nested(rank) {b*1e3:.2f} ms ({P/b:.0f} param-series/s); "
      f"together {P/(a+b):.0f} param-series/s, {P*D*C*4/(a+b)/1e9:.1f} GB/s of input")
