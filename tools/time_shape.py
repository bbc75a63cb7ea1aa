#!/usr/bin/env python3
"""Timing of the fused bulk+tail request on any shape (needs a GPU): python tools/time_shape.py DRAWS CHAINS PARAMS [fft]
(MDT_CONFIG / MDT_SMALL_ROUNDS / MDT_LIB in the environment select geometry and library, as everywhere.)"""
import os, sys
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
from bench import synth, timed_ms
pkg = g.load_package()
D, C, P = (int(v) for v in sys.argv[1:4])
x = synth(D, C, P, torch.device("cuda", 0), 4321).permute(2, 1, 0)
kw = {"autocov_method": pkg.FFTAutocovMethod()} if len(sys.argv) > 4 and sys.argv[4] == "fft" else {}
ms = timed_ms(lambda: pkg.bulk_tail(x, sync=False, **kw), 5, 1)
print(f"{D}x{C}x{P} [{os.environ.get('MDT_CONFIG', 'auto')}]: bulk+tail {ms:.2f} ms, {ms * 1e6 * 256 / (D * C * P):.2f} ns per element per CU, status {pkg.status() if hasattr(pkg, 'status') else '-'}")
