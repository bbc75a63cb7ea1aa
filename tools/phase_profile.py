#!/usr/bin/env python3
"""Diagnostics: where the fused kernel spends its cycles, by phase (needs a GPU).

    MDT_PROF=1 python tools/phase_profile.py [--params 2560] [--draws 10000] [--chains 4]

Reads the per-phase shader-clock totals the kernel accumulates when the context is created with
MDT_PROF=1 (mdt_debug_phase_cycles) and prints each phase's share.  Shares, not durations: the
stamps themselves perturb nothing but they are summed over all workgroups.
"""
import argparse
import ctypes
import os
import sys

os.environ.setdefault("MDT_PROF", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402
from bench import synth  # noqa: E402

NAMES = {0: "load x -> registers", 1: "rank x (tail)", 2: "series bulk (moments+autocov+Geyer)", 3: "rank |x-med| (tail)",
         4: "series tail (moments)", 5: "indicators (tail ESS)", 6: "outputs",
         8: "  rank: sample+sort+splitters", 9: "  rank: histogram", 10: "  rank: scan+rounds", 11: "  rank: place",
         12: "  rank: count", 13: "  zseries: loads+stores", 14: "  zseries: barrier", 15: "  zseries: products+reductions",
         7: "  zseries: finalize"}


NAMES_Q = {13: "  rank: partition pass", 14: "  rank: round load + local histogram", 15: "  rank: round prefix",
           7: "  rank: round output (scores / tags)", 11: "  rank: round placement", 12: "  rank: round scores: table lookup, back to partition order",
           0: "  rank: round count loop (+ sorted value store)", 6: "  rank: round sorted tags (+ outputs)"}


NAMES_F = {13: "  fold merge: tile boundaries", 14: "  fold merge: staging (loads, fold, LDS stores)", 15: "  fold merge: diagonal search + merge",
           7: "  fold merge: moment atomics"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--f", action="store_true", help="the library was built with -DMDT_PROF_F: the z-series sub-slots are lent to fold_merge_phase")
    ap.add_argument("--q", action="store_true", help="the library was built with -DMDT_PROF_Q: the z-series sub-slots are lent to the rank phase")
    ap.add_argument("--params", type=int, default=2560)
    ap.add_argument("--draws", type=int, default=10000)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cfg5", action="store_true", help="BASELINE config 5's call: rhat_nested(kind = rank) on Float32, 16 chains in 4 superchains")
    a = ap.parse_args()
    pkg = g.load_package()
    if a.cfg5:
        import numpy as np
        from bench import synth_cfg5
        a.chains = 16
        x = synth_cfg5(a.draws, 16, a.params, torch.device("cuda", 0), 55).permute(2, 1, 0)
        ids = np.repeat(np.arange(1, 5), 4)
        call = lambda **kw: pkg.rhat_nested(x, ids, kind="rank")  # (synchronous: the entry point has no sync switch)
    else:
        x = synth(a.draws, a.chains, a.params, torch.device("cuda", 0), 1234).permute(2, 1, 0)
        call = lambda **kw: pkg.bulk_tail(x, **kw)
    call()
    lib, h = pkg._lib.load(), pkg._lib.ctx(0, torch.cuda.current_stream().cuda_stream)
    buf = (ctypes.c_uint64 * 16)()
    lib.mdt_debug_phase_cycles(h, buf, 1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(a.reps):
        call(sync=False)
    ev1.record()
    torch.cuda.synchronize()
    rc = lib.mdt_debug_phase_cycles(h, buf, 1)
    assert rc == 0, rc
    ms = ev0.elapsed_time(ev1) / a.reps
    top = sum(buf[i] for i in range(16))  # every stamp advances the shared clock: the slots partition the time
    print(f"{a.draws}x{a.chains}x{a.params}: {ms:.3f} ms per call, {a.params / ms * 1e3:.0f} param-series/s")
    names = {**NAMES, **NAMES_Q} if a.q else ({**NAMES, **NAMES_F} if a.f else NAMES)
    for i in sorted(names):
        print(f"{names[i]:42s} {100.0 * buf[i] / top:6.2f} %")


if __name__ == "__main__":
    main()
