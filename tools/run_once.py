#!/usr/bin/env python3
"""a few calls on synthetic data (for rocprofv3 --pmc runs): tools/run_once.py [params] [reps] [cfg2|cfg3|cfg5|cfg1|s500]"""
import os, sys
import os as _os
_os.environ.setdefault("MDT_TEST_HOOKS", "1")  # diagnostics: the library reads its A/B switches only behind this gate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
from bench import synth, synth_cfg5
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
pkg = g.load_package()
dev = torch.device("cuda", 0)
if cfg == "cfg3":
    x = synth(100000, 8, P, dev, 4321).permute(2, 1, 0)
    m = pkg.FFTAutocovMethod()
    for _ in range(reps):
        pkg.bulk_tail(x, autocov_method=m)
elif cfg == "cfg5":
    x = synth_cfg5(10000, 16, P, dev, 55).permute(2, 1, 0)
    ids = np.repeat(np.arange(1, 5), 4)
    for _ in range(reps):
        pkg.mcse_rhat_nested(x, ids, kind="rank")   # one fused launch (mdt_mcse_nested)
elif cfg in ("cfg1", "s500"):   # short slices: the wave-per-parameter kernel (config 1's slice shape at scale; 500 x 4)
    x = synth(1000 if cfg == "cfg1" else 500, 4, P, dev, 1234).permute(2, 1, 0)
    for _ in range(reps):
        pkg.bulk_tail(x)
else:
    x = synth(10000, 4, P, dev, 1234).permute(2, 1, 0)
    for _ in range(reps):
        pkg.bulk_tail(x)
torch.cuda.synchronize()
print("done")
