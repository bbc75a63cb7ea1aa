#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline request (needs a GPU): the same 10 000 x 4 x P Float64 array handed over as a HOST
buffer (location = MDT_HOST: the library stages it to the device, runs, copies the outputs back), pageable and pinned.
Usage: python tools/time_host.py [P]"""
import os, sys, time
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
from bench import synth, timed_ms
pkg = g.load_package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
D, C = 10000, 4
xd = synth(D, C, P, torch.device("cuda", 0), 1234)
dev_ms = timed_ms(lambda: pkg.bulk_tail(xd.permute(2, 1, 0), sync=False), 5, 2)
xh_t = xd.cpu()
for name, t in (("pageable", xh_t), ("pinned", xh_t.pin_memory())):
    xh = t.numpy().transpose(2, 1, 0)  # (draws, chains, params) view of the same memory
    pkg.bulk_tail(xh)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        pkg.bulk_tail(xh)
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"host buffer ({name}): {ms:.1f} ms per call, {P / ms * 1e3:.0f} param-series/s, {D * C * P * 8 / ms / 1e6:.1f} GB/s of input "
          f"(device-resident: {dev_ms:.2f} ms)")
