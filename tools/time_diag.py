#!/usr/bin/env python3
"""Timing of the streaming diagnostics (needs a GPU): gelmandiag and bfmi on the BASELINE config-2 shape.

    python tools/time_diag.py [--draws 10000 --chains 4 --params 10000]

Prints the wall time of the whole call (moments kernel + D2H of the moments + host finish) and, from HIP events
around the launch on a raw ctypes call, nothing more than that: the kernel itself is in the rocprofv3 stats
(`column_moments_kernel`)."""
import argparse
import os
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
from bench import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=10000)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--params", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    pkg = g.load_package()
    x = synth(a.draws, a.chains, a.params, torch.device("cuda", 0), 1234).permute(2, 1, 0)
    xe = x.permute(2, 1, 0).reshape(a.params * a.chains, a.draws).t()  # (draws, columns), zero-copy
    nbytes = a.draws * a.chains * a.params * 8
    for name, fn in (("gelmandiag", lambda: pkg.gelmandiag(x)), ("bfmi (every column)", lambda: pkg.bfmi(xe))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        print(f"{name:22s} {dt * 1e3:8.3f} ms per call  ({nbytes / dt / 1e9:7.1f} GB/s of input, {a.params / dt:10.0f} params/s)")


if __name__ == "__main__":
    main()
