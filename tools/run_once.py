#!/usr/bin/env python3
"""one bulk+tail call on synthetic data (for rocprofv3 --pmc runs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
from bench import synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pkg = g.load_package()
x = synth(10000, 4, P, torch.device("cuda", 0), 1234).permute(2, 1, 0)
for _ in range(reps):
    pkg.bulk_tail(x)
torch.cuda.synchronize()
print("done")
