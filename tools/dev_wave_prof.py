#!/usr/bin/env python3
"""Cycles per phase of the wave-per-parameter kernel (diagnostics build: tools/build_wave_variant.sh waveprof -DWV_PROF):
MDT_LIB=tools/exp/libwaveprof.so python tools/dev_wave_prof.py DRAWS CHAINS PARAMS"""
import os, sys
os.environ["MDT_TEST_HOOKS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
buf = torch.zeros(8192 + 2 * 24 * 4096, dtype=torch.int32).pin_memory()
os.environ["MDT_WAVE_DBG"] = hex(buf.data_ptr())
import __graft_entry__ as g
from bench import synth, timed_ms
pkg = g.load_package()
D, C, P = (int(v) for v in sys.argv[1:4])
x = synth(D, C, P, torch.device("cuda", 0), 4321).permute(2, 1, 0)
pkg.bulk_tail(x)
torch.cuda.synchronize()
buf.zero_()
ms = timed_ms(lambda: pkg.bulk_tail(x, sync=False), 1, 0)
torch.cuda.synchronize()
prof = buf[8192:].numpy().view(np.uint64).reshape(-1, 24)
nw = int((prof.sum(axis=1) > 0).sum())
tot = prof.sum(axis=0).astype(float)
names = ["ticket", "sample+sort", "pass A", "prefix/rounds table", "pass B", "rounds: rest", "ostat+fold", "bulk series: rest", "indicators", "outputs",
         "round: load+bins+count", "round: prefix", "round: placement", "round: dense count+outputs", "series: stage+moments", "series: centre+mfma",
         "-", "fold: median place + tile boundaries", "fold: staging", "fold: diagonal search", "fold: merge + atomics", "-", "-", "-"]
print(f"{D}x{C}x{P}: {ms:.2f} ms, {nw} waves, {tot.sum() / max(nw, 1) / 1e6:.2f} Mcycles per wave")
for i in np.argsort(-tot):
    if tot[i] > 0:
        print(f"  {names[i]:32s} {100 * tot[i] / tot.sum():5.1f} %   {tot[i] / P:9.0f} cycles per parameter")
