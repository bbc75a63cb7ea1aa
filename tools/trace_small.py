#!/usr/bin/env python3
"""Diagnostics: kernel durations and gaps of small requests queued back to back (run under rocprofv3 --kernel-trace)."""
import sys, time, torch
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
pkg = g.load_package()
x = torch.randn(10, 4, 1000, dtype=torch.float64, device="cuda").permute(2, 1, 0)
pkg.bulk_tail(x); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    pkg.bulk_tail(x, sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"100 async calls: issued in {(t1 - t0) * 1e3:.2f} ms, done after {(t2 - t0) * 1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(100):
    pkg.bulk_tail(x)
print(f"100 sync calls: {(time.perf_counter() - t0) * 1e3:.2f} ms")
