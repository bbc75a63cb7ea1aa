#!/usr/bin/env python3
"""Where the first call of a process goes (needs a GPU): context (code objects, kernel attributes), reserve (scratch, score table),
first request, second request -- wall clock, each followed by a device synchronise.  python tools/time_first_call.py [DRAWS CHAINS PARAMS]"""
import os, sys, time
os.environ.setdefault("MDT_TEST_HOOKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
from bench import synth
D, C, P = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 4, 10000)
dev = torch.device("cuda", 0)
x = synth(D, C, P, dev, 99).permute(2, 1, 0)
torch.cuda.synchronize()


def wall(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


t_load = wall(lambda: g.load_package()._lib.load())
pkg = g.load_package()
cur = torch.cuda.current_stream(dev).cuda_stream
t_ctx = wall(lambda: pkg._lib.ctx(0, cur))
t_res = wall(lambda: pkg.reserve(D, C, P, dtype=np.float64))
t_small = float('nan')
if os.environ.get("FIRST_TINY"):  # a tiny request of ANOTHER kernel of the same code object first: does it pay for the load?
    e = torch.randn(4, 100, dtype=torch.float64, device=dev).permute(1, 0)
    t_small = wall(lambda: pkg.bfmi(e))
t_1 = wall(lambda: pkg.bulk_tail(x, sync=False))
t_2 = wall(lambda: pkg.bulk_tail(x, sync=False))
print(f"{D}x{C}x{P}: dlopen {t_load:.2f} ms, first context of the process {t_ctx:.2f} ms, reserve {t_res:.2f} ms, tiny request of another kernel first {t_small:.2f} ms, first request {t_1:.2f} ms, second {t_2:.2f} ms")
