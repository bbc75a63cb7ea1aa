#!/usr/bin/env python3
"""Timing of the BASELINE config-3 shape (100_000 draws x 8 chains x P params Float64, FFTAutocovMethod): fused bulk+tail,
rhat(kind=rank), ess(kind=bulk).  Usage: python tools/time_cfg3.py [P]"""
import os, sys
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
from bench import synth, timed_ms
pkg = g.load_package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
x = synth(100000, 8, P, torch.device("cuda", 0), 4321).permute(2, 1, 0)
m = pkg.FFTAutocovMethod()
print(f"config-3 shape, P={P}: bulk+tail (fft) {timed_ms(lambda: pkg.bulk_tail(x, autocov_method=m, sync=False), 5, 1):.2f} ms, "
      f"rhat(rank) {timed_ms(lambda: pkg.rhat(x), 5, 1):.2f} ms, ess(bulk) {timed_ms(lambda: pkg.ess(x, kind='bulk'), 5, 1):.2f} ms")
